"""GPU: ZSTD level-1 / 3 / 6 frame encode (4mz "fast" / "medium" / "high") byte parity against the oracle port
(oracle/zstd_enc_port.c, itself pinned to the reference's ZSTD_compress) and the reference CLI's
`4mc -z -1` golden manifest; every frame also decodes back on the GPU."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import helpers
from helpers import B

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _encode(gpu, srcs, caps, level=1):
    offs, pos = [], 0
    for s in srcs:
        offs.append(pos); pos += len(s) + 5
    buf = np.zeros(pos + 64, np.uint8)
    for s, o in zip(srcs, offs):
        buf[o:o + len(s)] = s
    dsts, dpos = [], 0
    for c in caps:
        dsts.append(dpos); dpos += c + 40
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, dsts, [len(s) for s in srcs], caps))
    d_out = torch.full((dpos + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    gpu.zstd_compress(torch.from_numpy(buf).cuda(), d_out, batch, level)
    torch.cuda.synchronize()
    res = batch.download()["result"]
    out = d_out.cpu().numpy()
    # a frame never writes past its capacity
    for d, c in zip(dsts, caps):
        assert (out[d + c: d + c + 40] == 0x5A).all()
    return res, [out[d:d + max(int(r), 0)] for d, r in zip(dsts, res)]


def _check(gpu, names, srcs, caps, tag, level=1):
    res, outs = _encode(gpu, srcs, caps, level)
    for k, s, cap, r, o in zip(names, srcs, caps, res, outs):
        want_r, want = helpers.orc_zstd_compress(s, level, cap)
        assert int(r) == want_r, (tag, level, k, len(s), cap, int(r), want_r)
        assert np.array_equal(o, want), (tag, k, len(s), cap)


def test_zstd_golden_frames(gpu):
    """The device writes the committed frames of the reference's ZSTD_compress (tests/golden/zstd_frames.json) at every level the device has (1..12)."""
    z = json.load(open(os.path.join(G, "zstd_frames.json")))
    inputs = helpers.golden_zstd_inputs()
    names = list(z)
    for lvl in range(1, 13):
        srcs = [np.ascontiguousarray(inputs[k]) for k in names]
        res, outs = _encode(gpu, srcs, [len(s) + 1024 for s in srcs], lvl)
        for k, r, o in zip(names, res, outs):
            hx = z[k]["frames"][str(lvl)]
            assert int(r) == len(hx) // 2 and o.tobytes().hex() == hx, (k, lvl)


@pytest.mark.parametrize("level", [2, 4, 5, 7, 8, 9, 10, 11])
def test_zstd_levels_4mz_does_not_use(gpu, level):
    """compressBytesDirectHC(level) passes any level through (native/jniZstdCompressor.c:143-173): levels 1..12 run on the device - their
    rows of the level table name fast, dfast, greedy, lazy, lazy2, btlazy2 and btopt by input size (clevels.h:25-130).  Every size class,
    the capacities 4mz and the bound give, edge inputs and two full 4 MiB blocks against the oracle port (pinned to the reference's
    ZSTD_compress at these levels by tests/test_oracle_golden.py and by the frames of tests/golden/zstd_frames.json)."""
    data = helpers.corpus(8 * B)
    srcs, names = [], []
    for b in (0, 3, 5):
        for n in (700, 5000, 16384, 16385, 60000, 131072, 131073, 200000, 262144, 262145, 600000):
            srcs.append(np.ascontiguousarray(data[b * B + 1000: b * B + 1000 + n])); names.append("c%d_%d" % (b, n))
    for b in (1, 7):
        srcs.append(np.ascontiguousarray(data[b * B:(b + 1) * B])); names.append("block%d" % b)
    for k, v in helpers.edge_inputs().items():
        srcs.append(np.ascontiguousarray(v)); names.append(k)
    _check(gpu, names, srcs, [helpers.zstd_bound(len(s)) for s in srcs], "bound", level)
    _check(gpu, names, srcs, [max(len(s) - 1, 0) for s in srcs], "n-1", level)
    _check(gpu, names, srcs, [len(s) // 3 for s in srcs], "n/3", level)


@pytest.mark.parametrize("level", [1, 3, 6])
def test_zstd_bytes_identical_edge_inputs(gpu, level):
    inputs = helpers.edge_inputs()
    names = list(inputs)
    srcs = [inputs[k] for k in names]
    _check(gpu, names, srcs, [helpers.zstd_bound(len(s)) for s in srcs], "bound", level)
    _check(gpu, names, srcs, [max(len(s) - 1, 0) for s in srcs], "n-1", level)   # the capacity 4mz passes
    _check(gpu, names, srcs, [len(s) // 3 for s in srcs], "n/3", level)           # mostly dstSize_tooSmall


@pytest.mark.parametrize("level", [1, 3, 6])
def test_zstd_size_classes_and_tails(gpu, level):
    """Level-table size classes (16 KiB / 128 KiB / 256 KiB), 128 KiB sub-block boundaries and tails."""
    rng = np.random.default_rng(5)
    src = helpers.corpus(3 * B, first_block=5)
    sizes = [7, 8, 18, 19, 20, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 16383, 16384, 16385, 65791, 65792,
             131071, 131072, 131073, 131078, 131079, 131080, 262144, 262145, 262151, 393216 + 3, 524288, 524289, 1500001,
             2 * 1024 * 1024 + 77, 3 * 1024 * 1024]
    srcs = [src[int(o): int(o) + n].copy() for n, o in zip(sizes, rng.integers(0, B, len(sizes)))]
    names = ["n=%d" % n for n in sizes]
    _check(gpu, names, srcs, [max(n - 1, 0) for n in sizes], "n-1", level)
    _check(gpu, names, srcs, [helpers.zstd_bound(n) for n in sizes], "bound", level)


@pytest.mark.parametrize("level", [1, 3])
def test_zstd_matches_running_to_block_ends(gpu, level):
    """A match that runs to the end of a 128 KiB inner block (or of the input) ends behind ilimit: the reference makes none of
    the refills that follow a match there (zstd_fast.c:263, zstd_double_fast.c:262).  Zero runs placed on those ends, reached
    from inside the dense window of the level-1 / level-3 finders."""
    rng = np.random.default_rng(11)
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9))).astype(np.uint8)) for _ in range(300)]
    text = np.frombuffer(b" ".join(words[int(i)] for i in rng.integers(0, 300, 120000)), np.uint8)
    srcs, names = [], []
    for n in (131072, 262144 + 5000, 400000):
        for run in (40, 230, 700, 5000):
            d = text[:n].copy()
            for edge in list(range(131072, n + 1, 131072)) + [n]:
                d[max(edge - run, 0):edge] = 0
                if edge + 16 <= n: d[edge:edge + 3] = 0                      # ... and a little of the run on the other side
            srcs.append(d); names.append("n=%d run=%d" % (n, run))
    _check(gpu, names, srcs, [helpers.zstd_bound(len(s)) for s in srcs], "ends", level)
    _check(gpu, names, srcs, [len(s) - 1 for s in srcs], "ends n-1", level)


@pytest.mark.parametrize("level", [1, 3, 6])
def test_zstd_capacity_sweep(gpu, level):
    """Capacities around the real frame size: every overflow rule of the bit and byte writers."""
    src = helpers.corpus(B, first_block=2)
    for n in (100, 1000, 20000, 140000):
        d = src[7 * n: 8 * n]
        c, _ = helpers.orc_zstd_compress(d, level, helpers.zstd_bound(n))
        caps = list(range(max(0, c - 24), c + 10))
        _check(gpu, ["cap=%d" % x for x in caps], [d] * len(caps), caps, "tight n=%d" % n, level)


@pytest.mark.parametrize("level,key", [(1, "4mz-1"), (3, "4mz-2"), (6, "4mz-3")])
def test_zstd_corpus_blocks_golden_manifest_and_roundtrip(gpu, level, key):
    m = json.load(open(os.path.join(G, "corpus_manifest.json")))
    n = m["corpus"]["bytes"]
    data = helpers.corpus(n)
    assert hashlib.sha256(data.tobytes()).hexdigest() == m["corpus"]["sha256"]
    nb = (n + B - 1) // B
    lens = [min(B, n - b * B) for b in range(nb)]
    blocks = gpu.make_blocks([b * B for b in range(nb)], [b * (B + 64) for b in range(nb)], lens, lens)
    batch = gpu.DeviceBatch(blocks)
    d_src = torch.from_numpy(data).cuda()
    d_dst = torch.zeros(nb * (B + 64), dtype=torch.uint8, device="cuda")
    gpu.encode_blocks(d_src, d_dst, batch, codec=gpu.CODEC_ZSTD, level=level)
    torch.cuda.synchronize()
    got = batch.download()
    want = m["levels"][key]["blocks"]
    for b, (u, c, x) in enumerate(want):
        assert (int(got["src_len"][b]), int(got["result"][b]), int(got["xxh32"][b])) == (u, c, x), b
    # decode the encoded payloads back on the GPU
    csz = [int(r) for r in got["result"]]
    dblocks = gpu.make_blocks([b * (B + 64) for b in range(nb)], [b * B for b in range(nb)], csz, lens)
    dblocks["xxh32"] = got["xxh32"]
    dbatch = gpu.DeviceBatch(dblocks)
    d_back = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    gpu.decode_blocks(d_dst, d_back, dbatch, codec=gpu.CODEC_ZSTD)
    torch.cuda.synchronize()
    res = dbatch.download()["result"]
    assert [int(r) for r in res] == lens
    assert torch.equal(d_back[:n], d_src)


@pytest.mark.parametrize("flag,key", [("-1", "4mz-1"), ("-2", "4mz-2"), ("-3", "4mz-3")])
def test_cli_4mz_file_equals_reference(gpu, tmp_path, flag, key):
    """`4mc -z -1|-2|-3 file` writes the reference CLI's .4mz bytes; -z -4 (zstd level 12) fails loudly, no CPU fallback."""
    import subprocess
    m = json.load(open(os.path.join(G, "corpus_manifest.json")))
    data = helpers.corpus(m["corpus"]["bytes"])
    src = tmp_path / "c.bin"; src.write_bytes(data.tobytes())
    out = tmp_path / "c.4mz"
    r = subprocess.run([gpu.cli_path(), "-z", flag, "-f", str(src), str(out)], capture_output=True)
    assert r.returncode == 0, r.stderr
    img = out.read_bytes()
    assert len(img) == m["levels"][key]["file_bytes"]
    assert hashlib.sha256(img).hexdigest() == m["levels"][key]["sha256"], "file differs from the reference CLI's"
    back = tmp_path / "back.bin"
    assert subprocess.run([gpu.cli_path(), "-d", "-z", "-f", str(out), str(back)], capture_output=True).returncode == 0
    assert back.read_bytes() == data.tobytes()


def test_host_zstd_compress_entry_point(gpu):
    """fourmc_ZSTD_compress / fourmc_ZSTD_compressBound: what jniZstdCompressor.c:93 binds (capacity 1 GiB)."""
    import ctypes as C
    L = gpu.lib()
    d = helpers.corpus(300000, first_block=9)
    for n in (0, 1, 19, 1000, 300000):
        assert L.fourmc_ZSTD_compressBound(n) == helpers.zstd_bound(n)
        out = np.zeros(helpers.zstd_bound(n) + 64, np.uint8)
        r = L.fourmc_ZSTD_compress(out.ctypes.data, 1 << 30, d.ctypes.data, n, 1)
        want_r, want = helpers.orc_zstd_compress(d[:n], 1)
        assert r == want_r and np.array_equal(out[:r], want), n
    out = np.zeros(64, np.uint8)
    r = L.fourmc_ZSTD_compress(out.ctypes.data, 30, d.ctypes.data, 1000, 1)
    assert r == (1 << 64) - 70                                     # (size_t)-ZSTD_error_dstSize_tooSmall
    r = L.fourmc_ZSTD_compress(out.ctypes.data, 64, d.ctypes.data, 10, 13)
    assert r > (1 << 64) - 120                                     # level 13: ZSTD_isError(), no CPU fallback
    out = np.zeros(helpers.zstd_bound(300000) + 64, np.uint8)
    for lvl in (3, 6):
        r = L.fourmc_ZSTD_compress(out.ctypes.data, 1 << 30, d.ctypes.data, 300000, lvl)
        want_r, want = helpers.orc_zstd_compress(d, lvl)
        assert r == want_r and np.array_equal(out[:r], want), lvl


def test_zstd12_golden_manifest_and_every_size_class(gpu, tmp_path):
    """4mz Ultra (zstd level 12) on the device: lazy2 + 64-entry rows above 256 KiB, the binary-tree finder (btlazy2)
    from 16 KiB + 1 to 256 KiB, the optimal parser (btopt) at 16 KiB and less.  The 12 full corpus blocks AND the
    123457-byte tail equal the reference CLI's manifest; a level 4mz never uses is refused loudly."""
    import subprocess
    m = json.load(open(os.path.join(G, "corpus_manifest.json")))
    nb = 12
    data = helpers.corpus(nb * B)
    blocks = gpu.make_blocks([b * B for b in range(nb)], [b * (B + 64) for b in range(nb)], [B] * nb, [B] * nb)
    batch = gpu.DeviceBatch(blocks)
    d_src = torch.from_numpy(data).cuda()
    d_dst = torch.zeros(nb * (B + 64), dtype=torch.uint8, device="cuda")
    gpu.encode_blocks(d_src, d_dst, batch, codec=gpu.CODEC_ZSTD, level=12)
    torch.cuda.synchronize()
    got = batch.download()
    for b, (u, c, x) in enumerate(m["levels"]["4mz-4"]["blocks"][:nb]):
        assert (int(got["src_len"][b]), int(got["result"][b]), int(got["xxh32"][b])) == (u, c, x), b
    # raw frames against the oracle on a few sizes just above the 256 KiB class boundary, incl. a capacity that fails
    src = helpers.corpus(2 * B, first_block=21)
    sizes = [262145, 300001, 1000003, 2 * 1024 * 1024 + 5]
    _check(gpu, ["n=%d" % n for n in sizes], [src[n: 2 * n].copy() for n in sizes], [n - 1 for n in sizes], "n-1", 12)
    _check(gpu, ["n=%d" % n for n in sizes], [src[n: 2 * n].copy() for n in sizes], [n // 4 for n in sizes], "n/4", 12)
    # btlazy2 size classes (<= 128 KiB, <= 256 KiB), window / table clamps for small inputs, capacities that fail
    sizes = [16385, 20000, 65536, 65537, 100000, 131072, 131073, 200000, 262144]
    _check(gpu, ["n=%d" % n for n in sizes], [src[n: 2 * n].copy() for n in sizes], [helpers.zstd_bound(n) for n in sizes], "bound", 12)
    _check(gpu, ["n=%d" % n for n in sizes], [src[n: 2 * n].copy() for n in sizes], [n - 1 for n in sizes], "n-1", 12)
    _check(gpu, ["n=%d" % n for n in sizes], [src[n: 2 * n].copy() for n in sizes], [n // 4 for n in sizes], "n/4", 12)
    edge = {k: v[:262144] for k, v in helpers.edge_inputs().items() if len(v) > 16384}
    _check(gpu, list(edge), [v.copy() for v in edge.values()], [len(v) - 1 for v in edge.values()], "edge n-1", 12)
    # btopt sizes: predefined prices up to 1024 bytes, statistics above; window / table clamps; capacities that fail
    sizes = [0, 1, 2, 3, 7, 8, 9, 12, 13, 17, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 2048, 4095, 4096, 4097, 8192, 10000, 16383, 16384]
    _check(gpu, ["n=%d" % n for n in sizes], [src[3 * n: 4 * n].copy() for n in sizes], [helpers.zstd_bound(n) for n in sizes], "btopt bound", 12)
    _check(gpu, ["n=%d" % n for n in sizes], [src[5 * n: 6 * n].copy() for n in sizes], [max(n - 1, 0) for n in sizes], "btopt n-1", 12)
    _check(gpu, ["n=%d" % n for n in sizes], [src[7 * n: 8 * n].copy() for n in sizes], [n // 3 for n in sizes], "btopt n/3", 12)
    edge = {k: v[:16384] for k, v in helpers.edge_inputs().items()}
    _check(gpu, list(edge), [v.copy() for v in edge.values()], [max(len(v) - 1, 0) for v in edge.values()], "btopt edge n-1", 12)
    edge = {k: v[:900] for k, v in helpers.edge_inputs().items()}
    _check(gpu, list(edge), [v.copy() for v in edge.values()], [helpers.zstd_bound(len(v)) for v in edge.values()], "btopt edge 900", 12)
    # a level the device does not have (13 and above need btlazy2 on full blocks / btultra): refused, never a silently different payload
    small = gpu.DeviceBatch(gpu.make_blocks([0], [0], [16384], [16384]))
    for lvl in (13, 19, 0, -1):
        with pytest.raises(gpu.EngineError, match="not on the device"):
            gpu.zstd_compress(d_src, d_dst, small, lvl)
    # CLI: the full corpus file (12 blocks + a 123457-byte tail) equals the reference CLI's; a file with a tiny (btopt) tail round-trips
    full = tmp_path / "full.bin"; full.write_bytes(helpers.corpus(m["corpus"]["bytes"]).tobytes())
    fo = tmp_path / "full.4mz"
    assert subprocess.run([gpu.cli_path(), "-z", "-4", "-f", str(full), str(fo)], capture_output=True).returncode == 0
    img = fo.read_bytes()
    assert len(img) == m["levels"]["4mz-4"]["file_bytes"]
    import hashlib
    assert hashlib.sha256(img).hexdigest() == m["levels"]["4mz-4"]["sha256"], "file differs from the reference CLI's"
    f = tmp_path / "whole.bin"; f.write_bytes(data[: 3 * B].tobytes())
    out = tmp_path / "whole.4mz"
    assert subprocess.run([gpu.cli_path(), "-z", "-4", "-f", str(f), str(out)], capture_output=True).returncode == 0
    back = tmp_path / "whole.back"
    assert subprocess.run([gpu.cli_path(), "-d", "-z", "-f", str(out), str(back)], capture_output=True).returncode == 0
    assert back.read_bytes() == data[: 3 * B].tobytes()
    tail = helpers.corpus(B + 12345)
    g = tmp_path / "tail.bin"; g.write_bytes(tail.tobytes())
    assert subprocess.run([gpu.cli_path(), "-z", "-4", "-f", str(g), str(tmp_path / "tail.4mz")], capture_output=True).returncode == 0
    img = np.frombuffer((tmp_path / "tail.4mz").read_bytes(), np.uint8)
    blocks, _ = gpu.split_container(img, gpu.MAGIC_4MZ)
    b1 = blocks[1]
    want_r, want = helpers.orc_zstd_compress(tail[B:], 12, 12345 - 1)              # the tail block is the reference's btopt frame
    assert int(b1["src_len"]) == want_r and np.array_equal(img[int(b1["src_off"]): int(b1["src_off"]) + want_r], want)
    assert subprocess.run([gpu.cli_path(), "-d", "-z", "-f", str(tmp_path / "tail.4mz"), str(tmp_path / "tail.back")], capture_output=True).returncode == 0
    assert (tmp_path / "tail.back").read_bytes() == tail.tobytes()


def test_zstd12_log_corpus_blocks(gpu):
    """BASELINE configs[4]'s workload (4mz Ultra on the synthetic log corpus) at a size the oracle covers: frames of whole
    4 MiB log blocks and of a short tail equal the oracle's (itself pinned to the reference's ZSTD_compress level 12)."""
    logs = helpers.corpus(2 * B + 150001, first_block=5, logs=True)
    srcs = [logs[:B], logs[B:2 * B], logs[2 * B:], logs[1000:1000 + 300000]]
    caps = [len(s) - 1 for s in srcs]                    # the container's capacity (native/4mc.c:467)
    _check(gpu, ["log0", "log1", "tail", "mid300k"], srcs, caps, "logs", level=12)
    ref = helpers.ref()
    if ref is not None:                                  # and the reference itself, when it is built
        out = np.empty(B + 4096, np.uint8)
        r = ref.ZSTD_compress(out.ctypes.data, B - 1, srcs[0].ctypes.data, B, 12)
        res, outs = _encode(gpu, [srcs[0]], [B - 1], 12)
        assert int(res[0]) == int(r) and np.array_equal(outs[0], out[:r])


@pytest.mark.parametrize("level", [1, 3])
def test_zstd_dense_window_equals_the_batched_search(gpu, level, monkeypatch):
    """The two shapes of the level-1 / level-3 match finders (dense window; batched search alone, FOURMC_ZSTD_SERIAL=1: at
    level 1 the wave-uniform transcription of the reference loop) write the same frames: six 4 MiB blocks of different classes."""
    data = helpers.corpus(12 * B)
    picks = [0, 1, 2, 3, 5, 7]
    srcs = [data[b * B:(b + 1) * B] for b in picks]
    caps = [B - 1] * len(srcs)
    monkeypatch.setenv("FOURMC_ZSTD_SERIAL", "0")
    res_a, outs_a = _encode(gpu, srcs, caps, level)
    monkeypatch.setenv("FOURMC_ZSTD_SERIAL", "1")
    res_b, outs_b = _encode(gpu, srcs, caps, level)
    for b, ra, oa, rb, ob in zip(picks, res_a, outs_a, res_b, outs_b):
        assert int(ra) == int(rb) and np.array_equal(oa, ob), b
