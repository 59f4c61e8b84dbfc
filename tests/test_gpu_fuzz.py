"""GPU: seeded fuzz of every encoder against the oracle — many small/medium inputs of mixed structure
(runs, periodic, text-like, sparse, random, concatenations) through one batched launch per codec;
payload bytes and return codes must equal the oracle's, and every payload must decode back on the GPU."""
import numpy as np
import pytest
import torch

import helpers
from helpers import B

pytestmark = pytest.mark.gpu


def _inputs(seed, count):
    rng = np.random.default_rng(seed)
    base = helpers.corpus(2 * B, first_block=int(rng.integers(0, 40)))
    out = []
    for i in range(count):
        n = int(rng.choice([rng.integers(0, 64), rng.integers(64, 4096), rng.integers(4096, 70000), rng.integers(70000, 400000)]))
        kind = int(rng.integers(0, 7))
        if kind == 0:
            d = base[(o := int(rng.integers(0, B))): o + n].copy()
        elif kind == 1:
            d = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 2:
            d = np.resize(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8), n)
        elif kind == 3:
            d = rng.integers(0, int(rng.integers(2, 17)), n, dtype=np.uint8)
        elif kind == 4:
            d = np.zeros(n, np.uint8); k = max(1, n // int(rng.integers(5, 200))); d[::k] = rng.integers(1, 256, len(d[::k]), dtype=np.uint8)
        elif kind == 5:                                   # text with a long-distance repeat
            a = base[(o := int(rng.integers(0, B))): o + n // 2]; d = np.concatenate([a, rng.integers(0, 256, n - 2 * len(a), dtype=np.uint8), a])
        else:                                             # runs of varying bytes and lengths
            parts = []
            while sum(map(len, parts)) < n:
                parts.append(np.full(int(rng.integers(1, 600)), int(rng.integers(0, 256)), np.uint8))
            d = np.concatenate(parts)[:n] if parts else np.zeros(0, np.uint8)
        out.append(np.ascontiguousarray(d, dtype=np.uint8))
    return out


def _run(gpu, srcs, caps, launch):
    offs, pos = [], 0
    for s in srcs:
        offs.append(pos); pos += len(s) + 7
    buf = np.zeros(pos + 64, np.uint8)
    for s, o in zip(srcs, offs):
        buf[o:o + len(s)] = s
    dsts, dpos = [], 0
    for c in caps:
        dsts.append(dpos); dpos += c + 24
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, dsts, [len(s) for s in srcs], caps))
    d_out = torch.zeros(dpos + 64, dtype=torch.uint8, device="cuda")
    launch(torch.from_numpy(buf).cuda(), d_out, batch)
    torch.cuda.synchronize()
    res = [int(r) for r in batch.download()["result"]]
    out = d_out.cpu().numpy()
    return res, [out[d:d + max(r, 0)] for d, r in zip(dsts, res)], d_out, dsts


@pytest.mark.parametrize("codec", ["lz4_fast", "lz4_mc", "lz4_hc4", "zstd1", "zstd3", "zstd6", "zstd12"])
def test_fuzz_encoders_equal_oracle(gpu, codec):
    srcs = _inputs({"lz4_fast": 1, "lz4_mc": 2, "lz4_hc4": 3, "zstd1": 4, "zstd3": 5, "zstd6": 6, "zstd12": 7}[codec],
                   60 if codec == "lz4_hc4" else 240 if codec == "zstd12" else 160)
    lz4_bound = [helpers.oracle().orc_lz4_compress_bound(len(s)) for s in srcs]
    rng = np.random.default_rng(99)
    # a mix of capacities: bound, n-1 (container), and something smaller
    pick = rng.integers(0, 3, len(srcs))
    if codec.startswith("lz4"):
        caps = [b if p == 0 else max(len(s) - 1, 0) if p == 1 else len(s) // 2 for s, b, p in zip(srcs, lz4_bound, pick)]
    else:
        caps = [helpers.zstd_bound(len(s)) if p == 0 else max(len(s) - 1, 0) if p == 1 else len(s) // 2 for s, p in zip(srcs, pick)]
    launch = {
        "lz4_fast": lambda a, b, c: gpu.lz4_compress_fast(a, b, c),
        "lz4_mc": lambda a, b, c: gpu.lz4_compress_mc(a, b, c),
        "lz4_hc4": lambda a, b, c: gpu.lz4_compress_hc(a, b, c, 4),
        "zstd1": lambda a, b, c: gpu.zstd_compress(a, b, c, 1),
        "zstd3": lambda a, b, c: gpu.zstd_compress(a, b, c, 3),
        "zstd6": lambda a, b, c: gpu.zstd_compress(a, b, c, 6),
        "zstd12": lambda a, b, c: gpu.zstd_compress(a, b, c, 12),
    }[codec]
    oracle = {
        "lz4_fast": lambda s, cap: helpers.orc_compress(s, cap),
        "lz4_mc": lambda s, cap: helpers.orc_compress_mc(s, cap),
        "lz4_hc4": lambda s, cap: helpers.orc_compress_hc(s, 4, cap),
        "zstd1": lambda s, cap: helpers.orc_zstd_compress(s, 1, cap),
        "zstd3": lambda s, cap: helpers.orc_zstd_compress(s, 3, cap),
        "zstd6": lambda s, cap: helpers.orc_zstd_compress(s, 6, cap),
        "zstd12": lambda s, cap: helpers.orc_zstd_compress(s, 12, cap),
    }[codec]
    res, outs, d_out, dsts = _run(gpu, srcs, caps, launch)
    for i, (s, cap, r, o) in enumerate(zip(srcs, caps, res, outs)):
        wr, wb = oracle(s, cap)
        assert r == wr, (codec, i, len(s), cap, r, wr)
        assert np.array_equal(o, wb), (codec, i, len(s), cap)
    # decode every successful payload back on the GPU
    ok = [i for i, r in enumerate(res) if r > 0]
    blocks = gpu.make_blocks([dsts[i] for i in ok], np.cumsum([0] + [len(srcs[i]) + 8 for i in ok[:-1]]).tolist(),
                             [res[i] for i in ok], [len(srcs[i]) for i in ok])
    dbatch = gpu.DeviceBatch(blocks)
    d_back = torch.zeros(int(sum(len(srcs[i]) + 8 for i in ok)) + 64, dtype=torch.uint8, device="cuda")
    (gpu.zstd_decompress if codec.startswith("zstd") else gpu.lz4_decompress)(d_out, d_back, dbatch)
    torch.cuda.synchronize()
    got = dbatch.download()
    back = d_back.cpu().numpy()
    for k, i in enumerate(ok):
        assert int(got["result"][k]) == len(srcs[i]), (codec, i)
        o = int(blocks["dst_off"][k])
        assert np.array_equal(back[o:o + len(srcs[i])], srcs[i]), (codec, i)
