#!/usr/bin/env python3
"""Seeded fuzz of the LZ4 decode paths (default: the launcher's choice; FOURMC_DECODE selects): streams BUILT from random
sequences - literal runs of every length class (0, < 15, one / two / many continuation bytes), matches of every length class
(4 .. 18, one continuation byte, several), offsets from 1 (runs) to 65535, rows of the stream with 21 tokens and rows without
any, long literal runs that skip whole rows - and mutated copies of them (byte flips, truncations, capacities above and below
the decoded size).  Every result (return code, bytes, bytes outside the block untouched) has to equal the oracle's
(oracle/lz4_port.c, pinned to the reference's LZ4_decompress_safe).   python tools/fuzz_decode.py [first_seed] [count]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
gpu = importlib.import_module("4mc_amd"); gpu.gpu_init(0)


def build_stream(rng, target):
    """a valid LZ4 block of about `target` decoded bytes from random sequences; returns (stream, decoded)"""
    out = bytearray(); dec = bytearray()
    style = int(rng.integers(0, 6))
    def lenbytes(v):
        b = bytearray()
        while v >= 255: b.append(255); v -= 255
        b.append(v); return b
    while len(dec) < target:
        lit_cls = int(rng.integers(0, 10))
        if style == 0: L = int(rng.integers(0, 4))                                   # dense tokens: up to 21 per row
        elif style == 1: L = int(rng.integers(0, 40))
        elif lit_cls < 5: L = int(rng.integers(0, 15))
        elif lit_cls < 8: L = int(rng.integers(15, 270))
        elif lit_cls < 9: L = int(rng.integers(270, 2000))
        else: L = int(rng.integers(2000, 70000))
        if len(dec) == 0 and L == 0: L = 1                                            # the first match needs something behind it
        m_cls = int(rng.integers(0, 10))
        if style == 0: M = int(rng.integers(4, 8))
        elif m_cls < 5: M = int(rng.integers(4, 19))
        elif m_cls < 8: M = int(rng.integers(19, 274))
        elif m_cls < 9: M = int(rng.integers(274, 3000))
        else: M = int(rng.integers(3000, 200000))
        lits = rng.integers(0, 256, L, dtype=np.uint8).tobytes() if style != 5 else bytes([int(rng.integers(0, 3))]) * L
        o_cls = int(rng.integers(0, 6))
        avail = len(dec) + L
        hi = min(avail, 65535)
        off = 1 if o_cls == 0 else int(rng.integers(1, min(hi, 8) + 1)) if o_cls == 1 else int(rng.integers(1, min(hi, 300) + 1)) if o_cls < 4 else int(rng.integers(1, hi + 1))
        out.append((min(L, 15) << 4) | min(M - 4, 15))
        if L >= 15: out += lenbytes(L - 15)
        out += lits; dec += lits
        out += bytes([off & 255, off >> 8])
        if M - 4 >= 15: out += lenbytes(M - 4 - 15)
        start = len(dec) - off
        for k in range(M): dec.append(dec[start + k])
    # the block's end: a last sequence of literals only, at least 5, the last match 12 bytes before the end (lz4.c:243-247)
    L = int(rng.integers(12, 40))
    lits = rng.integers(0, 256, L, dtype=np.uint8).tobytes()
    out.append(min(L, 15) << 4)
    if L >= 15: out += lenbytes(L - 15)
    out += lits; dec += lits
    return np.frombuffer(bytes(out), np.uint8).copy(), np.frombuffer(bytes(dec), np.uint8).copy()


def decode_batch(comps, caps):
    offs, pos = [], 0
    for c in comps: offs.append(pos); pos += len(c) + 13
    src = np.zeros(pos + 64, np.uint8)
    for c, o in zip(comps, offs): src[o:o + len(c)] = c
    doffs, dpos = [], 0
    for i, cap in enumerate(caps):
        dpos = (dpos + 127) // 128 * 128 + (i * 29) % 128
        doffs.append(dpos); dpos += cap + 7
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.full((dpos + 256,), 0xA5, dtype=torch.uint8, device="cuda")
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, doffs, [len(c) for c in comps], caps))
    gpu.lz4_decompress(d_src, d_dst, batch)
    torch.cuda.synchronize()
    return batch.download()["result"], d_dst.cpu().numpy(), doffs


first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bad = 0; total = 0; handed = 0
ALONE = os.environ.get("FOURMC_DECODE", "").endswith("only")          # a fast path without the exact walker behind it: it may hand a stream
RETRY = -1000000003                                                  # back (kRetry), it may never answer differently from the oracle
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    comps, caps, kinds = [], [], []
    for i in range(48):
        target = int(rng.choice([200, 3000, 70000, 300000, 1500000, 4000000]))
        c, d = build_stream(rng, target)
        if len(d) > (4 << 20): continue
        comps.append(c); caps.append(len(d) + (0 if i % 3 else int(rng.integers(1, 500)))); kinds.append("built")
        for _ in range(2):                                   # mutations
            m = c.copy()
            for _ in range(int(rng.integers(1, 4))): m[int(rng.integers(0, len(m)))] = rng.integers(0, 256)
            if rng.integers(0, 4) == 0: m = m[: int(rng.integers(1, len(m)))]
            comps.append(m); caps.append(len(d) if rng.integers(0, 3) else max(1, len(d) - int(rng.integers(1, 50)))); kinds.append("mutated")
    res, out, doffs = decode_batch(comps, caps)
    for i, c in enumerate(comps):
        wr, want = helpers.orc_decompress(c, caps[i])
        total += 1
        if ALONE and int(res[i]) == RETRY:
            handed += 1
            ok = (doffs[i] == 0 or out[doffs[i] - 1] == 0xA5) and bool(np.all(out[doffs[i] + caps[i]: doffs[i] + caps[i] + 7] == 0xA5))   # even then: nothing outside the block
            if not ok: bad += 1; print("MISMATCH (outside the block) seed", seed, "stream", i, flush=True)
            continue
        ok = int(res[i]) == wr and (wr <= 0 or np.array_equal(out[doffs[i]: doffs[i] + wr], want))
        ok = ok and (doffs[i] == 0 or out[doffs[i] - 1] == 0xA5) and bool(np.all(out[doffs[i] + caps[i]: doffs[i] + caps[i] + 7] == 0xA5))
        if not ok:
            bad += 1; print("MISMATCH seed", seed, "stream", i, kinds[i], "len", len(c), "cap", caps[i], "got", int(res[i]), "want", wr, flush=True)
    print("seed", seed, "streams", len(comps), "bad so far", bad, flush=True)
print("fuzz_decode:", total, "streams,", bad, "mismatches" + (f", {handed} handed back by the fast path alone" if ALONE else ""))
sys.exit(1 if bad else 0)
