#!/usr/bin/env python3
"""Damaged golden frames through the reference's ZSTD_decompress (oracle/_ref, this container only) and the CPU restatement
(oracle/zstd_port.cpp): verdicts and accepted bytes have to be the same.   python tools/zstd_damaged_vs_ref.py [seed ...]"""
import sys, json, os, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers
ref = helpers.ref(); assert ref is not None, "oracle/_ref not built"
z = json.load(open(os.path.join(ROOT, "tests", "golden", "zstd_frames.json")))
tot = L = X = D = 0
for seed in [int(a) for a in sys.argv[1:]] or [99, 7, 2025, 11, 12]:
    rng = np.random.default_rng(seed)
    cnt = collections.Counter(); n = extra = diff = 0
    for name in z:
        for lvl, hx in z[name]["frames"].items():
            base = np.frombuffer(bytes.fromhex(hx), np.uint8); cap = z[name]["input_bytes"]
            for t in range(150):
                m = base.copy(); k = t % 4
                if k == 0: m[rng.integers(0, len(m))] ^= 1 << rng.integers(0, 8)
                elif k == 1: m = m[: rng.integers(1, len(m))]
                elif k == 2:
                    i = rng.integers(4, len(m)); m[i:i + 2] = rng.integers(0, 256, len(m[i:i + 2]), dtype=np.uint8)
                else: m = np.concatenate([m, rng.integers(0, 256, rng.integers(1, 6), dtype=np.uint8)])
                m = np.ascontiguousarray(m)
                dst = np.zeros(cap + 64, np.uint8)
                rr = ref.ZSTD_decompress(dst.ctypes.data, cap, m.ctypes.data, len(m))
                r, out = helpers.orc_zstd_decompress(m.tobytes(), cap); n += 1
                if ref.ZSTD_isError(rr):
                    if r >= 0: extra += 1
                elif r < 0: cnt[(name, lvl, k)] += 1
                elif not (r == rr and np.array_equal(out, dst[:rr])): diff += 1
    print("seed", seed, n, "frames: reference accepts / port rejects", sum(cnt.values()), "| port accepts / reference rejects", extra, "| both accept, bytes differ", diff, dict(cnt), flush=True)
    tot += n; L += sum(cnt.values()); X += extra; D += diff
print("total", tot, "frames:", L, X, D)
sys.exit(1 if (X or D) else 0)
