#!/usr/bin/env python3
"""4mz decode, one-wave kernel against entropy + execute kernels: time per launch and equality with the input.
    python tools/zdec_ab.py [blocks] [level]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
logs = len(sys.argv) > 3 and sys.argv[3] == "logs"
base_n = 48 if not logs else 24
base = helpers.corpus(base_n * B, logs=logs) if logs else helpers.corpus(base_n * B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb // base_n))[: nb * B].contiguous()
offs = np.arange(nb, dtype=np.uint64) * B
lens = np.full(nb, B, dtype=np.uint32)
enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
p.encode_blocks(d_src, d_stage, enc, codec=p.CODEC_ZSTD, level=level)
e = enc.download()
lib = p.lib()
for split in ((1, 1) if os.environ.get("ZDEC_SPLIT_ONLY") else (0, 1, 0, 1)):
    lib.fourmc_gpu_set_zstd_decode_split(split)
    d_out = torch.zeros(nb * B + 64, dtype=torch.uint8, device="cuda")
    ts = []
    for it in range(3):
        dec = p.DeviceBatch(p.make_blocks(offs, offs, e["result"].astype(np.uint32), lens, e["xxh32"]))
        s = torch.cuda.Event(enable_timing=True); t = torch.cuda.Event(enable_timing=True)
        s.record(); p.decode_blocks(d_stage, d_out, dec, codec=p.CODEC_ZSTD); t.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(t))
    r = dec.download()["result"].astype(np.int64)
    ok = bool((r == B).all()) and torch.equal(d_out[: nb * B], d_src)
    print(f"split={split} blocks={nb} level={level}: {min(ts):8.2f} ms  ok={ok}  bad results: {np.unique(r[r != B])[:5]}")
