#!/usr/bin/env python3
"""XXH32 of n payloads of the 4mc Fast container (S-mix, sizes as the headline launch has them), kernel time by HIP events.
   python tools/xxh_timing.py [blocks]"""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = 2048
base = helpers.corpus(48 * B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, np.uint32)
enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
p.encode_blocks(d_src, d_stage, enc)
cs = enc.download()["result"].astype(np.uint32)
reps = -(-n // nb)
big = d_stage.repeat(min(reps, 8))                                    # up to 64 GiB of staging would not fit: 8 copies, reused
so = (np.tile(offs, reps)[:n] + (np.arange(n, dtype=np.uint64) // nb % 8) * np.uint64(nb * B))
hb = p.DeviceBatch(p.make_blocks(so, so, np.tile(cs, reps)[:n], np.tile(lens, reps)[:n]))
best = 1e9
for it in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record(); p.xxh32(big, hb, 0); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
tot = float(np.tile(cs, reps)[:n].astype(np.float64).sum())
print(f"xxh32 of {n} payloads ({tot / 1e9:.2f} GB): {best:.2f} ms = {tot / best / 1e6:.1f} GB/s")
