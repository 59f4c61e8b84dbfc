#!/usr/bin/env python3
"""Time of one zstd encode launch (level, blocks, corpus) and a size check against a base run.  python tools/zenc_time.py <level> <blocks> [logs]"""
import importlib, os, sys, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
level = int(sys.argv[1]); nb = int(sys.argv[2]); logs = len(sys.argv) > 3 and sys.argv[3] == "logs"
bn = 24 if logs else 48
base = helpers.corpus(bn * B, logs=True) if logs else helpers.corpus(bn * B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb // bn))[: nb * B].contiguous()
offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, dtype=np.uint32)
st = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
for it in range(2):
    enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
    s = torch.cuda.Event(enable_timing=True); t = torch.cuda.Event(enable_timing=True)
    s.record(); p.encode_blocks(d_src, st, enc, codec=p.CODEC_ZSTD, level=level); t.record(); torch.cuda.synchronize()
e = enc.download()
h = hashlib.sha256(e["result"].tobytes() + e["xxh32"].tobytes()).hexdigest()[:16]
print(f"level {level} blocks {nb}: {s.elapsed_time(t):9.1f} ms  {nb * B / s.elapsed_time(t) / 1e6:6.2f} GB/s  sizes+checksums {h}")
