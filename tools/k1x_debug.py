#!/usr/bin/env python3
"""Decodes inputs one at a time through the C ABI and reports where the output differs from the input (debug aid)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers, test_gpu_lz4par as T
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if os.environ.get("FOURMC_DECODE") is None and "k1x" in __file__: p.lib().fourmc_gpu_set_lz4_decode_path(1)
ins = T._inputs()
for name, src in ins.items():
    if len(src) < 64: continue
    r, comp = helpers.orc_compress(src)
    n, tok, opos, total = T._sequences(comp, len(src))
    d_src = torch.from_numpy(comp).cuda(); d_dst = torch.full((len(src) + 256,), 0xA5, dtype=torch.uint8, device="cuda")
    batch = p.DeviceBatch(p.make_blocks([0], [0], [len(comp)], [len(src)]))
    p.lz4_decompress(d_src, d_dst, batch); torch.cuda.synchronize()
    res = int(batch.download()["result"][0])
    out = d_dst.cpu().numpy()[: len(src)]
    bad = np.nonzero(out != src)[0]
    msg = f"{name:14s} n={len(src):8d} nseq={n:7d} result={res} bad={len(bad)}"
    if len(bad):
        b0 = int(bad[0]); i = int(np.searchsorted(opos, b0, side="right")) - 1
        # literal or match byte?
        t = int(tok[i]); tk = int(comp[t]); ll = tk >> 4; q = t + 1
        if ll == 15:
            while True:
                bb = int(comp[q]); q += 1; ll += bb
                if bb != 255: break
        kind = "literal" if b0 < opos[i] + ll else "match"
        off = int(comp[q + ll]) | (int(comp[q + ll + 1]) << 8) if i < n - 1 else 0
        runs = np.split(bad, np.nonzero(np.diff(bad) != 1)[0] + 1)
        msg += f" first={b0} (window {b0 >> 10}, seq {i}, {kind}, ll={ll}, off={off}, seq start {int(opos[i])}) runs={len(runs)} firstrun={len(runs[0])} got={out[b0]} want={src[b0]}"
    print(msg, flush=True)
