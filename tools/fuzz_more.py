#!/usr/bin/env python3
"""Extra seeds of tests/test_gpu_fuzz.py for the LZ4 encoders (fast, MC, HC 4/8) and the zstd encoder (levels 1 .. 12; FUZZ_ZLEVELS="2 4 5" picks; half of the
inputs of the levels with tree strategies - 9 .. 12 - cut to the btopt / btlazy2 size classes) - run by hand after kernel changes:
    python tools/fuzz_more.py [first_seed] [count]"""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers, test_gpu_fuzz as tf
gpu = importlib.import_module("4mc_amd"); gpu.gpu_init(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
bad = 0
for seed in range(first, first + count):
    srcs = tf._inputs(seed, 120)
    rng = np.random.default_rng(seed + 7)
    pick = rng.integers(0, 3, len(srcs))
    bound = [helpers.oracle().orc_lz4_compress_bound(len(s)) for s in srcs]
    caps = [b if p == 0 else max(len(s) - 1, 0) if p == 1 else len(s) // 2 for s, b, p in zip(srcs, bound, pick)]
    for name, launch, orc in () if os.environ.get("FUZZ_ZSTD_ONLY") else (("fast", lambda a, b, c: gpu.lz4_compress_fast(a, b, c), lambda s, cap: helpers.orc_compress(s, cap)),
                              ("mc", lambda a, b, c: gpu.lz4_compress_mc(a, b, c), lambda s, cap: helpers.orc_compress_mc(s, cap)),
                              ("hc4", lambda a, b, c: gpu.lz4_compress_hc(a, b, c, 4), lambda s, cap: helpers.orc_compress_hc(s, 4, cap)),
                              ("hc8", lambda a, b, c: gpu.lz4_compress_hc(a, b, c, 8), lambda s, cap: helpers.orc_compress_hc(s, 8, cap))):
        res, outs, d_out, dsts = tf._run(gpu, srcs, caps, launch)
        for i, (s, cap, r, o) in enumerate(zip(srcs, caps, res, outs)):
            wr, wb = orc(s, cap)
            if r != wr or not np.array_equal(o, wb):
                bad += 1; print("MISMATCH", name, seed, i, len(s), cap, r, wr)
        ok = [i for i, r in enumerate(res) if r > 0]
        blocks = gpu.make_blocks([dsts[i] for i in ok], np.cumsum([0] + [len(srcs[i]) + 8 for i in ok[:-1]]).tolist(), [res[i] for i in ok], [len(srcs[i]) for i in ok])
        db = gpu.DeviceBatch(blocks)
        d_back = torch.zeros(int(sum(len(srcs[i]) + 8 for i in ok)) + 64, dtype=torch.uint8, device="cuda")
        gpu.lz4_decompress(d_out, d_back, db); torch.cuda.synchronize()
        got = db.download(); back = d_back.cpu().numpy()
        for k, i in enumerate(ok):
            o = int(blocks["dst_off"][k])
            if int(got["result"][k]) != len(srcs[i]) or not np.array_equal(back[o:o + len(srcs[i])], srcs[i]):
                bad += 1; print("DECODE MISMATCH", name, seed, i)
    for level in [int(x) for x in os.environ.get("FUZZ_ZLEVELS", "1 3 6 12").split()]:
        zs = srcs
        if level >= 9:                                   # small last blocks: optimal parser (<= 16 KiB), binary tree (<= 256 KiB)
            zs = [s[: int(rng.integers(0, 16385))] if k % 2 else s for k, s in enumerate(srcs)]
        zcaps = [helpers.zstd_bound(len(s)) if p == 0 else max(len(s) - 1, 0) if p == 1 else len(s) // 2 for s, p in zip(zs, pick)]
        res, outs, d_out, dsts = tf._run(gpu, zs, zcaps, lambda a, b, c: gpu.zstd_compress(a, b, c, level))
        for i, (s, cap, r, o) in enumerate(zip(zs, zcaps, res, outs)):
            wr, wb = helpers.orc_zstd_compress(s, level, cap)
            if r != wr or not np.array_equal(o, wb):
                bad += 1; print("MISMATCH zstd", level, seed, i, len(s), cap, r, wr)
        ok = [i for i, r in enumerate(res) if r > 0]
        blocks = gpu.make_blocks([dsts[i] for i in ok], np.cumsum([0] + [len(zs[i]) + 8 for i in ok[:-1]]).tolist(), [res[i] for i in ok], [len(zs[i]) for i in ok])
        db = gpu.DeviceBatch(blocks)
        d_back = torch.zeros(int(sum(len(zs[i]) + 8 for i in ok)) + 64, dtype=torch.uint8, device="cuda")
        gpu.zstd_decompress(d_out, d_back, db); torch.cuda.synchronize()
        got = db.download(); back = d_back.cpu().numpy()
        for k, i in enumerate(ok):
            o = int(blocks["dst_off"][k])
            if int(got["result"][k]) != len(zs[i]) or not np.array_equal(back[o:o + len(zs[i])], zs[i]):
                bad += 1; print("DECODE MISMATCH zstd", level, seed, i)
    print("seed", seed, "done", flush=True)
print("mismatches:", bad)
