#!/usr/bin/env python3
"""Damaged 4mz frames through both decode paths (entropy + execute kernels | one-wave kernel) and the oracle: same verdicts, same
bytes where accepted.   python tools/fuzz_zdecode.py [rounds] [seed]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if not os.environ.get("FOURMC_LIB"): p.use_research(True); p.gpu_init(0)     # debug exports: research side build
B = p.BLOCKSIZE
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
data = helpers.corpus(12 * B)
lib = p.lib()

def decode(frames, caps):
    offs, pos = [], 3
    for f in frames: offs.append(pos); pos += len(f) + 5
    buf = np.zeros(pos + 64, np.uint8)
    for f, o in zip(frames, offs): buf[o:o + len(f)] = np.frombuffer(f, np.uint8)
    dsts, dpos = [], 0
    for c in caps: dsts.append(dpos); dpos += c + 64
    batch = p.DeviceBatch(p.make_blocks(offs, dsts, [len(f) for f in frames], caps))
    d_out = torch.full((dpos + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    p.zstd_decompress(torch.from_numpy(buf).cuda(), d_out, batch)
    res = batch.download()["result"]; out = d_out.cpu().numpy()
    return res, [out[d:d + max(int(r), 0)] for d, r in zip(dsts, res)]

total = accepted = 0
import ctypes as C
execd, back = C.c_ulonglong(0), C.c_ulonglong(0); n_exec = n_back = 0
for rd in range(rounds):
    frames, caps = [], []
    for k in range(48):
        b = int(rng.integers(0, 12)); n = int(rng.choice([3000, 40000, 200000, 700000, 1 << 20]))
        at = int(rng.integers(0, B - n)); src = data[b * B + at: b * B + at + n]
        level = int(rng.choice([1, 3, 6]))
        r, comp = helpers.orc_zstd_compress(src, level, n + 1024)
        if r <= 0: continue
        m = np.array(comp[:r], dtype=np.uint8)
        for _ in range(int(rng.integers(1, 3))):
            kind = int(rng.integers(0, 5)) if rng.integers(0, 3) == 0 else 0
            if kind == 0: m[rng.integers(0, len(m))] ^= 1 << rng.integers(0, 8)
            elif kind == 1: j = int(rng.integers(4, len(m))); m[j:j + 4] = rng.integers(0, 256, len(m[j:j + 4]), dtype=np.uint8)
            elif kind == 2 and len(m) > 40: m = m[: int(rng.integers(20, len(m)))]
            elif kind == 3: j = int(rng.integers(4, len(m))); m = np.concatenate([m[:j], m[j + 1:]])
            else: j = int(rng.integers(4, len(m))); m = np.concatenate([m[:j], rng.integers(0, 256, 1, dtype=np.uint8), m[j:]])
        frames.append(m.tobytes()); caps.append(n)
    got = {}
    for split in (1, 0):
        lib.fourmc_gpu_set_zstd_decode_split(split)
        lib.fourmc_gpu_debug_zstd_exec_counts(C.byref(execd), C.byref(back))
        got[split] = decode(frames, caps)
        lib.fourmc_gpu_debug_zstd_exec_counts(C.byref(execd), C.byref(back)); n_exec += execd.value; n_back += back.value
    for i, (f, c) in enumerate(zip(frames, caps)):
        r1, r0 = int(got[1][0][i]), int(got[0][0][i])
        assert r1 == r0, (rd, i, r1, r0)
        if r1 >= 0: assert np.array_equal(got[1][1][i], got[0][1][i]), (rd, i)
        wr, want = helpers.orc_zstd_decompress(f, c)
        assert (r1 < 0) == (wr < 0), (rd, i, r1, wr)
        if r1 >= 0: assert r1 == wr and np.array_equal(got[1][1][i], want), (rd, i)
        total += 1; accepted += r1 >= 0
    print(f"round {rd}: {total} damaged frames so far, {accepted} accepted by all three, no difference; execute kernel completed {n_exec}, handed back {n_back}", flush=True)
