#!/usr/bin/env python3
"""One container-mode encode launch of a codec at N blocks of the S-mix (kernel time by HIP events).
   python tools/enc_time.py <hc4|mc|z1|z3> [blocks]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
what = sys.argv[1]; nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
codec, level = {"hc4": (p.CODEC_LZ4_HC, 4), "mc": (p.CODEC_LZ4_MC, 0), "z1": (p.CODEC_ZSTD, 1), "z3": (p.CODEC_ZSTD, 3)}[what]
base = helpers.corpus(48 * B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, dtype=np.uint32)
st = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
for it in range(2):
    enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
    s = torch.cuda.Event(enable_timing=True); t = torch.cuda.Event(enable_timing=True)
    s.record(); p.encode_blocks(d_src, st, enc, codec=codec, level=level); t.record(); torch.cuda.synchronize()
cs = enc.download()["result"].astype(np.int64)
print(f"{what} blocks {nb}: {s.elapsed_time(t):9.1f} ms  {nb * B / s.elapsed_time(t) / 1e6:6.2f} GB/s  ratio {nb * B / float(cs.sum()):.4f}")
