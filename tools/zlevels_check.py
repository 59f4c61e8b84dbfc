#!/usr/bin/env python3
"""Every zstd level the device has (1..12), every size class of the level table: device frames against the oracle port's.
   python tools/zlevels_check.py [levels...]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
data = helpers.corpus(12 * B)
srcs = []
for b in (0, 1, 3, 5, 10):
    blk = data[b * B:(b + 1) * B]
    for n in (700, 5000, 16384, 16385, 60000, 131072, 131073, 200000, 262144, 262145, 600000):
        srcs.append(np.ascontiguousarray(blk[1000:1000 + n]))
srcs += [np.ascontiguousarray(data[b * B:(b + 1) * B]) for b in (0, 3, 7)]
ed = helpers.edge_inputs()
srcs += [np.ascontiguousarray(ed[k]) for k in ("text_60k", "period37", "lit_then_run", "two_symbols", "hello10", "one")]
levels = [int(x) for x in sys.argv[1:]] or list(range(1, 13))
offs = np.cumsum([0] + [(len(s) + 63) & ~63 for s in srcs]); caps = [helpers.zstd_bound(len(s)) for s in srcs]
doffs = np.cumsum([0] + [(c + 63) & ~63 for c in caps])
buf = np.zeros(int(offs[-1]) + 64, np.uint8)
for s, o in zip(srcs, offs): buf[o:o + len(s)] = s
d_src = torch.from_numpy(buf).cuda()
total = 0
for level in levels:
    bad = 0
    for tag, cc in (("bound", caps), ("n-1", [max(len(s) - 1, 0) for s in srcs])):
        batch = p.DeviceBatch(p.make_blocks(offs[:-1].astype(np.uint64), doffs[:-1].astype(np.uint64), np.array([len(s) for s in srcs], np.uint32), np.array(cc, np.uint32)))
        d_out = torch.zeros(int(doffs[-1]) + 64, dtype=torch.uint8, device="cuda")
        p.zstd_compress(d_src, d_out, batch, level)
        res = batch.download()["result"].astype(np.int64); out = d_out.cpu().numpy()
        for i, s in enumerate(srcs):
            r, comp = helpers.orc_zstd_compress(s, level, cc[i])
            if int(res[i]) != r or (r > 0 and bytes(out[doffs[i]:doffs[i] + r]) != bytes(comp[:r])):
                print("MISMATCH level", level, tag, "case", i, "n", len(s), "device", int(res[i]), "oracle", r); bad += 1
    print("level", level, "cases", 2 * len(srcs), "bad", bad, flush=True); total += bad
print("total bad", total)
sys.exit(1 if total else 0)
