#!/usr/bin/env python3
"""Per-phase cycle split of the LZ4 fast encoder (K2) per S-mix block class.  Uses a side build of the library
compiled with -DK2_PROF (cycle counters written behind the compressed payload):
    make -C 4mc_amd/csrc prof      # -> 4mc_amd/lib/libhadoop-4mc-prof.so
    FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-prof.so python tools/k2_phases.py"""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
data = helpers.corpus(12 * B)
for b in range(12):
    src = torch.from_numpy(data[b * B:(b + 1) * B].copy()).cuda()
    enc = p.DeviceBatch(p.make_blocks([0], [0], [B], [B]))
    stage = torch.zeros(B, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    p.encode_blocks(src, stage, enc); torch.cuda.synchronize()
    s.record(); p.encode_blocks(src, stage, enc); e.record(); torch.cuda.synchronize()
    r = int(enc.download()["result"][0])
    c = stage[B - 32: B - 8].cpu().numpy().view(np.uint64) if r < B - 256 else np.zeros(3, np.uint64)
    d = stage[B - 128: B - 72].cpu().numpy().view(np.uint64) if r < B - 256 else np.zeros(7, np.uint64)
    f = stage[B - 192: B - 136].cpu().numpy().view(np.uint64) if r < B - 256 else np.zeros(7, np.uint64)
    tot = float(c.sum() + d[:4].sum() + f[4] + f[5]) or 1.0
    dec = p.DeviceBatch(p.make_blocks([0], [0], [r], [B], enc.download()["xxh32"]))
    out = torch.zeros(B + 64, dtype=torch.uint8, device="cuda")
    ds = torch.cuda.Event(enable_timing=True); de = torch.cuda.Event(enable_timing=True)
    p.decode_blocks(stage, out, dec); torch.cuda.synchronize()
    ds.record(); p.decode_blocks(stage, out, dec); de.record(); torch.cuda.synchronize()
    print(f"        K1 dec {ds.elapsed_time(de):7.2f} ms")
    print(f"{names[b]:7s} csize {r:8d} enc {s.elapsed_time(e):8.2f} ms   dense: cursor {100*d[0]/tot:4.1f}% table {100*d[1]/tot:4.1f}% gather {100*d[2]/tot:4.1f}% prep {100*f[4]/tot:4.1f}% general {100*f[5]/tot:4.1f}% walk {100*c[1]/tot:4.1f}% emit {100*d[3]/tot:4.1f}% commit {100*c[2]/tot:4.1f}% | sparse {100*c[0]/tot:4.1f}%  windows {int(d[4])} seq {int(d[5])} slow-seq {int(d[6])} | window ends: none {int(f[0])} cross {int(f[1])} dirty-cut {int(f[2])} stride-cut {int(f[3])} inner-iterations {int(f[6])} total-clk {tot:.3g}")
