#!/usr/bin/env python3
"""Seeded fuzz of the ratio-tolerance LZ4 encoder (lz4_par_encode.hip): inputs BUILT to hit its corners - sizes around the window
(64), the segment (65536) and the block's end rules, runs and periods of every length below and above a window, records with short
strides (what a lane's own group hides from it), literal runs long enough for the general route, matches long enough for many
length bytes, incompressible stretches, text - many per launch, at random misaligned offsets, with capacities above, at and below
the size.  Every payload has to (a) decode to the input with the oracle's LZ4_decompress_safe restatement (and the reference's own
decoder when oracle/_ref is there), (b) equal the bytes of tools/model/lz4p_model.c, (c) respect the capacity rule.
    python tools/fuzz_k2p.py [first_seed] [count]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
gpu = importlib.import_module("4mc_amd"); gpu.gpu_init(0)
gpu.lib().fourmc_gpu_set_lz4_encode_mode(1)
TEXT = helpers.corpus(2 * helpers.B, first_block=0)[:3 << 20]
DB = helpers.corpus(helpers.B, first_block=4)


def build(rng):
    kind = int(rng.integers(0, 10))
    szc = int(rng.integers(0, 8))
    n = [int(rng.integers(0, 80)), int(rng.integers(60, 200)), int(rng.integers(200, 5000)), int(rng.integers(65500, 65600)), int(rng.integers(5000, 140000)),
         int(rng.integers(131000, 131200)), int(rng.integers(140000, 600000)), int(rng.integers(600000, 1500000))][szc]
    if kind == 0: return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1: return np.full(n, int(rng.integers(0, 256)), np.uint8)
    if kind == 2:
        per = int(rng.integers(1, 200)); return np.tile(rng.integers(0, 256, per, dtype=np.uint8), n // per + 1)[:n].copy()
    if kind == 3:
        o = int(rng.integers(0, len(TEXT) - n - 1)); return TEXT[o:o + n].copy()
    if kind == 4:
        o = int(rng.integers(0, len(DB) - n - 1)) if n < len(DB) - 1 else 0; return DB[o:o + min(n, len(DB))].copy()
    if kind == 5:                                   # records: a stride of 4..40 bytes, a few fields changing
        st = int(rng.integers(4, 41)); rec = rng.integers(0, 256, st, dtype=np.uint8)
        a = np.tile(rec, n // st + 1)[:n].copy()
        if n:
            idx = rng.integers(0, n, max(n // int(rng.integers(3, 30)), 1)); a[idx] = rng.integers(0, 256, len(idx), dtype=np.uint8)
        return a
    if kind == 6:                                   # long literal runs between repeats (the general route)
        parts = []; tot = 0
        blob = rng.integers(0, 256, int(rng.integers(20, 3000)), dtype=np.uint8)
        while tot < n:
            p = rng.integers(0, 256, int(rng.integers(1, 4000)), dtype=np.uint8) if rng.integers(0, 2) else blob
            parts.append(p); tot += len(p)
        return np.concatenate(parts)[:n].copy() if parts else np.zeros(0, np.uint8)
    if kind == 7:                                   # two symbols / small alphabet
        return rng.integers(0, int(rng.integers(2, 5)), n, dtype=np.uint8)
    if kind == 8:                                   # a far repeat: the same stretch 1..70000 bytes later
        h = rng.integers(0, 256, max(n // 3, 1), dtype=np.uint8)
        gap = rng.integers(0, 256, int(rng.integers(1, 70000)), dtype=np.uint8)
        return np.concatenate([h, gap, h])[:max(n, 1)].copy()
    a = TEXT[:n].copy()                             # text with a run dropped in at a random place
    if n > 300:
        o = int(rng.integers(0, n - 200)); a[o:o + int(rng.integers(1, 200))] = 7
    return a


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    orc = helpers.oracle(); ref = helpers.ref()
    tot = bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        arrays = [build(rng) for _ in range(24)]
        sizes = [helpers.lz4p_model_encode(a, orc.orc_lz4_compress_bound(len(a)) + 64) for a in arrays]
        capkind = [int(rng.integers(0, 4)) for _ in arrays]              # 0, 1: roomy; 2: exactly the size; 3: one byte short
        caps = [max(r + (64 if k < 2 else 0 if k == 2 else -1), 0) for (r, _), k in zip(sizes, capkind)]
        offs, pos = [], int(rng.integers(0, 16))
        for a in arrays: offs.append(pos); pos += len(a) + int(rng.integers(0, 9))
        buf = np.zeros(pos + 64, np.uint8)
        for a, o in zip(arrays, offs): buf[o:o + len(a)] = a
        doffs, dpos = [], 0
        for c in caps: doffs.append(dpos); dpos += c + 16
        d_dst = torch.full((dpos + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        batch = gpu.DeviceBatch(gpu.make_blocks(offs, doffs, [len(a) for a in arrays], caps))
        gpu.lz4_compress_fast(torch.from_numpy(buf).cuda(), d_dst, batch)
        res = batch.download()["result"]; out = d_dst.cpu().numpy()
        for i, a in enumerate(arrays):
            tot += 1
            r = int(res[i]); mr, mb = sizes[i]; cap = caps[i]
            want_r = mr if cap >= mr else 0
            ok = r == want_r and np.all(out[doffs[i] + cap:doffs[i] + cap + 16] == 0x5A)
            if ok and r > 0:
                got = out[doffs[i]:doffs[i] + r]
                ok = np.array_equal(got, mb)
                if ok:
                    dr, back = helpers.orc_decompress(got, len(a)); ok = dr == len(a) and np.array_equal(back[:len(a)], a)
                if ok and ref is not None:
                    dst = np.zeros(len(a) + 8, np.uint8); g = np.ascontiguousarray(got)
                    ok = ref.LZ4_decompress_safe(g.ctypes.data, dst.ctypes.data, len(g), len(a)) == len(a) and np.array_equal(dst[:len(a)], a)
            if not ok:
                bad += 1; print(f"seed {seed} input {i}: n={len(a)} r={r} model={mr} cap={cap}")
    print(f"fuzz_k2p: seeds {first}..{first + count - 1}, {tot} inputs, {bad} bad")
    sys.exit(1 if bad else 0)


main()
