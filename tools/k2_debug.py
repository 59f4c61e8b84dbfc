#!/usr/bin/env python3
"""Debug aid: encode the edge inputs and corpus blocks with K2 and report the first byte that differs from the oracle."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
inputs = dict(helpers.edge_inputs())
data = helpers.corpus(12 * B)
for b in range(12): inputs[f"corpus{b}"] = data[b * B:(b + 1) * B]
bad = 0
for k, s in inputs.items():
    n = len(s)
    if n == 0: continue
    cap = helpers.oracle().orc_lz4_compress_bound(n)
    src = torch.from_numpy(np.ascontiguousarray(s)).cuda()
    dst = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda")
    batch = p.DeviceBatch(p.make_blocks([0], [0], [n], [cap]))
    p.lz4_compress_fast(src, dst, batch) if hasattr(p, "lz4_compress_fast") else p.encode_blocks(src, dst, batch)
    r = int(batch.download()["result"][0])
    want_r, want = helpers.orc_compress(s, cap)
    got = dst[:max(r, 0)].cpu().numpy()
    if r != want_r or not np.array_equal(got, want):
        bad += 1
        m = min(len(got), len(want))
        d = np.nonzero(got[:m] != want[:m])[0]
        print(f"{k}: n={n} r={r} want={want_r} first diff at {int(d[0]) if len(d) else m}")
print("mismatches:", bad)
