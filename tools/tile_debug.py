#!/usr/bin/env python3
"""Tile LZ4 decode (lz4_tile.hip) block by block against the input: per S-mix class and for the edge inputs, the result code, the
first differing byte, and what the walk / executor left in the block's workspace slot (tail position, resume point).
Debug aid; run on the GPU box:  python tools/tile_debug.py [--edges]"""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if not os.environ.get("FOURMC_LIB"): p.use_research(True); p.gpu_init(0)     # debug exports: research side build
B = p.BLOCKSIZE
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
def one(tag, data, path):
    r, comp = helpers.orc_compress(data)
    if r <= 0:
        print(f"{tag:14s} stored"); return True
    p.lib().fourmc_gpu_set_lz4_decode_path(path)
    d_src = torch.from_numpy(np.concatenate([comp, np.zeros(64, np.uint8)])).cuda()
    d_dst = torch.full((len(data) + 256,), 0xA5, dtype=torch.uint8, device="cuda")
    blk = p.DeviceBatch(p.make_blocks([0], [64], [len(comp)], [len(data)]))
    p.lz4_decompress(d_src, d_dst, blk); torch.cuda.synchronize()
    res = int(blk.download()["result"][0])
    out = d_dst.cpu().numpy()
    meta = np.zeros(48, np.uint32)
    p.binding.check(p.lib().fourmc_gpu_debug_read_workspace(meta.ctypes.data, 0, meta.nbytes), "ws")
    got = out[64: 64 + len(data)]
    bad = np.nonzero(got != data)[0]
    guard = bool(np.all(out[:64] == 0xA5) and np.all(out[64 + len(data):] == 0xA5))
    ok = res == len(data) and len(bad) == 0 and guard
    print(f"{tag:14s} path {path} csize {len(comp):8d} res {res:11d} {'OK ' if ok else 'BAD'} first diff {int(bad[0]) if len(bad) else -1} ndiff {len(bad)} guard {guard}"
          f" | status {meta[0]} tail_ip {meta[1]} (csize-{len(comp) - int(meta[1])}) res_ip {meta[2]} res_op {meta[3]}", flush=True)
    return ok
allok = True
data = helpers.corpus(12 * B)
for path in (14, 13):
    for b in range(12):
        allok &= one(names[b], data[b * B:(b + 1) * B].copy(), path)
if "--edges" in sys.argv:
    import test_gpu_lz4par as par
    for k, v in par._inputs().items():
        if len(v) >= 300: allok &= one(k[:14], np.ascontiguousarray(v), 13)
print("ALL OK" if allok else "SOME BAD")
