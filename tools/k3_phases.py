#!/usr/bin/env python3
"""Per-phase cycle split of the LZ4 HC encoder (K3) per S-mix block class, from the side build (make prof):
    FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-prof.so python tools/k3_phases.py [level]"""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE; level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
data = helpers.corpus(12 * B)
for b in (0, 1, 3, 5, 7, 10):
    src = torch.from_numpy(data[b * B:(b + 1) * B].copy()).cuda()
    enc = p.DeviceBatch(p.make_blocks([0], [0], [B], [B]))
    stage = torch.zeros(B, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    p.encode_blocks(src, stage, enc, codec=p.CODEC_LZ4_HC, level=level); torch.cuda.synchronize()
    s.record(); p.encode_blocks(src, stage, enc, codec=p.CODEC_LZ4_HC, level=level); e.record(); torch.cuda.synchronize()
    r = int(enc.download()["result"][0])
    c = stage[B - 128: B - 40].cpu().numpy().view(np.uint64)
    tot = float(c[:3].sum()) or 1.0
    print(f"{names[b]:7s} csize {r:8d} hc{level} {s.elapsed_time(e):8.2f} ms  window build {100*c[0]/tot:4.1f}%  searches {100*c[1]/tot:4.1f}%  parser+emit {100*c[2]/tot:4.1f}%"
          f"  windows {int(c[3])} searches {int(c[4])} memory extensions {int(c[5])}  clk/search {c[1]/max(1,int(c[4])):.0f} clk/window {c[0]/max(1,int(c[3])):.0f}"
          f"  | emits {int(c[9])} x {c[6]/max(1,int(c[9])):.0f} clk, no-match steps {int(c[10])} x {c[7]/max(1,int(c[10])):.0f} clk, first-search hits {c[8]/max(1,int(c[9])):.0f} clk/seq; Mclk total {tot/1e6:.0f} emit {c[6]/1e6:.0f} skip {c[7]/1e6:.0f} first {c[8]/1e6:.0f}")
