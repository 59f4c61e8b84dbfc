#!/usr/bin/env python3
"""How much does the order of the blocks in a launch matter?  Same 2048 blocks (S-mix pattern of 48, replicated), block
descriptors in pattern order, shuffled, and sorted by class; LZ4 fast encode and decode launches timed with HIP events."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE; NB = 2048
pat = helpers.corpus(48 * B)
d_src = torch.from_numpy(pat.copy()).cuda().repeat((NB + 47) // 48)[: NB * B].contiguous()
d_dst = torch.zeros(NB * B, dtype=torch.uint8, device="cuda")
d_out = torch.zeros(NB * B + 64, dtype=torch.uint8, device="cuda")
rng = np.random.default_rng(3)
orders = {"pattern": np.arange(NB), "shuffled": rng.permutation(NB), "by class": np.argsort(np.arange(NB) % 12, kind="stable"),
          "stride 7": (np.arange(NB) * 7) % NB}
def timed(f):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)
for name, o in orders.items():
    offs = (o.astype(np.int64) * B).tolist()
    eb = p.DeviceBatch(p.make_blocks(offs, offs, [B] * NB, [B] * NB))
    t_enc = timed(lambda: p.encode_blocks(d_src, d_dst, eb))
    enc = eb.download()
    db = p.DeviceBatch(p.make_blocks(offs, offs, [int(r) for r in enc["result"]], [B] * NB, enc["xxh32"]))
    t_dec = timed(lambda: p.decode_blocks(d_dst, d_out, db))
    print(f"{name:10s} encode {t_enc:8.2f} ms   decode {t_dec:8.2f} ms")
