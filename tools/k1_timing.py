#!/usr/bin/env python3
"""Times the LZ4 decode kernel (K1) per S-mix block class (1 block per launch) and on the bench batch
(FOURMC_BENCH_BLOCKS replicas of the 48-block S-mix); checks the round trip."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if os.environ.get("FOURMC_DECODE") is None and "k1x" in __file__: p.lib().fourmc_gpu_set_lz4_decode_path(1)
B = p.BLOCKSIZE
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e))
    return best
def run(src, nb, tag):
    S = (B + B // 255 + 16 + 63) & ~63
    offs = np.arange(nb, dtype=np.uint64) * B; soffs = np.arange(nb, dtype=np.uint64) * S; lens = np.full(nb, B, np.uint32)
    enc = p.DeviceBatch(p.make_blocks(offs, soffs, lens, np.full(nb, S, np.uint32)))
    stage = torch.empty(nb * S, dtype=torch.uint8, device="cuda")
    p.lz4_compress_fast(src, stage, enc)
    r = enc.download()
    dec = p.DeviceBatch(p.make_blocks(soffs, offs, r["result"].astype(np.uint32), lens))
    out = torch.zeros(nb * B + 64, dtype=torch.uint8, device="cuda")
    td = t(lambda: p.lz4_decompress(stage, out, dec))
    ok = torch.equal(out[: nb * B], src[: nb * B])
    print(f"{tag:12s} blocks {nb:5d} dec {td:9.2f} ms ({nb * B / td / 1e6:7.2f} GB/s) roundtrip {'ok' if ok else 'BAD'}", flush=True)
if "--classes" in sys.argv:
    data = helpers.corpus(12 * B)
    for b in range(12):
        run(torch.from_numpy(data[b * B:(b + 1) * B].copy()).cuda(), 1, names[b])
nb = int(os.environ.get("FOURMC_BENCH_BLOCKS", "2048"))
if "--class-load" in sys.argv:                      # the whole batch made of one class
    data = helpers.corpus(12 * B)
    for b in (0, 1, 3, 5, 10):
        run(torch.from_numpy(data[b * B:(b + 1) * B].copy()).cuda().repeat(nb), nb, names[b] + " x%d" % nb)
base = helpers.corpus(48 * B)
src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
run(src, nb, "S-mix")
