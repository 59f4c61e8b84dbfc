#!/usr/bin/env python3
"""LZ4 decode paths (wx / trio / rows) on one launch of a given corpus and encoder: python tools/decode_paths_corpus.py <smix|logs> <fast|hc4> [blocks]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
corpus, encoder = sys.argv[1], sys.argv[2]; nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
bn = 24 if corpus == "logs" else 48
base = helpers.corpus(bn * B, logs=True) if corpus == "logs" else helpers.corpus(bn * B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb // bn))[: nb * B].contiguous()
offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, dtype=np.uint32)
enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens)); st = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
if encoder == "hc4": p.encode_blocks(d_src, st, enc, codec=p.CODEC_LZ4_HC, level=4)
else: p.encode_blocks(d_src, st, enc)
e = enc.download()
ratio = nb * B / float(e["result"].astype(np.int64).sum())
for path, name in ((6, "auto"), (9, "wx"), (0, "trio"), (4, "rows")):
    p.lib().fourmc_gpu_set_lz4_decode_path(path)
    out = torch.zeros(nb * B + 64, dtype=torch.uint8, device="cuda"); ts = []
    for it in range(3):
        dec = p.DeviceBatch(p.make_blocks(offs, offs, e["result"].astype(np.uint32), lens, e["xxh32"]))
        s = torch.cuda.Event(enable_timing=True); t = torch.cuda.Event(enable_timing=True)
        s.record(); p.decode_blocks(st, out, dec, codec=(p.CODEC_LZ4_HC if encoder == "hc4" and os.environ.get("DEC_HC") else p.CODEC_LZ4_FAST)); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t))
    print(f"{corpus} {encoder} ratio {ratio:.2f} blocks {nb}: {name} {min(ts):.2f} ms", torch.equal(out[: nb * B], d_src))
