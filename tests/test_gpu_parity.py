"""Parity of the HIP path against the oracle, through the C ABI (include/fourmc_gpu.h).

Bit-exact bar: every byte, every return code.  Sizes here are ones the oracle finishes in
seconds; full-size properties are in test_gpu_fullsize.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

import helpers
from helpers import B

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _pack(arrays, align=1, pad=64):
    """Concatenate byte arrays at offsets aligned to `align`; returns (buffer, offsets)."""
    offs, pos = [], 0
    for a in arrays:
        pos = -(-pos // align) * align
        offs.append(pos)
        pos += len(a)
    buf = np.zeros(pos + pad, dtype=np.uint8)
    for a, o in zip(arrays, offs):
        buf[o:o + len(a)] = a
    return buf, offs


# ------------------------------------------------------------------------------------------ XXH32
def test_xxh32_lengths_and_alignments(gpu):
    rng = np.random.default_rng(1)
    lens = list(range(0, 70)) + [255, 256, 1008, 1023, 1024, 1025, 4095, 4096, 4097, 65536, 100003, B]
    arrays = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    for align in (1, 16):                       # payloads sit at arbitrary byte offsets in a .4mc
        buf, offs = _pack(arrays, align)
        if align == 1:
            buf = np.concatenate([np.zeros(3, np.uint8), buf]); offs = [o + 3 for o in offs]
        blocks = gpu.make_blocks(offs, [0] * len(lens), lens, [0] * len(lens))
        for seed in (0, 0x9E3779B1):
            batch = gpu.DeviceBatch(blocks)
            gpu.xxh32(_dev(buf), batch, seed)
            got = batch.download()["xxh32"]
            want = [helpers.orc_xxh32(a, seed) for a in arrays]
            assert list(got) == want, (align, seed)


@pytest.mark.parametrize("count", [3, 1500, 3000, 6000, 17000])
def test_xxh32_many_blocks_per_wavefront(gpu, count):
    """Batches above 1024 blocks put 2, 4, 8 blocks on one wavefront (staged kernel up to 4): ragged lengths around the
    64-stripe bank size, blocks that end while their neighbours go on, a last wavefront that is not full (17000: 16 per wavefront)."""
    rng = np.random.default_rng(count)
    lens = [int(rng.choice([0, 5, 16, 1023, 1024, 1025, 2047, 2048, 2049, int(rng.integers(0, 9000))])) for _ in range(count)]
    lens[-1] = 40000; lens[0] = 70001
    arrays = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    buf, offs = _pack(arrays, 1)
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, [0] * count, lens, [0] * count))
    gpu.xxh32(_dev(buf), batch, 7)
    got = batch.download()["xxh32"]
    want = [helpers.orc_xxh32(a, 7) for a in arrays]
    assert list(got) == want


# ------------------------------------------------------------------------------------------ decode
def _decode_batch(gpu, comps, caps, align=1):
    buf, offs = _pack(comps, align)
    dsts, pos = [], 0
    for c in caps:
        dsts.append(pos); pos += c + 32
    blocks = gpu.make_blocks(offs, dsts, [len(c) for c in comps], caps)
    batch = gpu.DeviceBatch(blocks)
    d_out = torch.full((pos + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    gpu.lz4_decompress(_dev(buf), d_out, batch)
    res = batch.download()["result"]
    out = d_out.cpu().numpy()
    return res, [out[d:d + max(r, 0)] for d, r in zip(dsts, res)], out, dsts


def test_lz4_decode_edge_inputs(gpu):
    inputs = helpers.edge_inputs()
    names = list(inputs)
    comps = []
    for k in names:
        r, c = helpers.orc_compress(inputs[k])
        assert r > 0
        comps.append(c)
    for slack in (0, 100, 4096):                 # exact capacity (CLI) and oversize (JNI: 4 MiB)
        caps = [len(inputs[k]) + slack for k in names]
        res, outs, raw, dsts = _decode_batch(gpu, comps, caps)
        for k, r, o, c, d in zip(names, res, outs, caps, dsts):
            want_r, want = helpers.orc_decompress(comps[names.index(k)], c)
            assert r == want_r == len(inputs[k]), (k, slack, r, want_r)
            assert np.array_equal(o, inputs[k]), (k, slack)
            assert np.all(raw[d + c:d + c + 32] == 0xA5), f"{k}: wrote past its capacity"


def test_lz4_decode_corpus_blocks(gpu):
    data = helpers.corpus(12 * B)
    comps, caps = [], []
    for b in range(12):
        r, c = helpers.orc_compress(data[b * B:(b + 1) * B])
        comps.append(c); caps.append(B)
    res, outs, _, _ = _decode_batch(gpu, comps, caps)
    for b in range(12):
        assert res[b] == B
        assert np.array_equal(outs[b], data[b * B:(b + 1) * B]), b


def test_lz4_decode_corrupt_streams_match_oracle_codes(gpu):
    """Accept/reject set and negative return codes equal the reference's (oracle/lz4_port.c
    mirrors both decode loops of native/lz4/lz4.c:1995-2325)."""
    rng = np.random.default_rng(5)
    base = helpers.edge_inputs()
    srcs = [base["text_60k"], base["period37"][:30000], base["lit_then_run"][:60000], base["hello10"], base["text_300k"][:200000]]
    comps, caps = [], []
    for s in srcs:
        _, c = helpers.orc_compress(s)
        for trial in range(60):
            m = c.copy()
            kind = trial % 4
            if kind == 0:
                m[rng.integers(0, len(m))] ^= 1 << rng.integers(0, 8)
            elif kind == 1:
                m = m[: rng.integers(1, len(m))]
            elif kind == 2:
                i = rng.integers(0, len(m)); m[i:i + 4] = rng.integers(0, 256, len(m[i:i + 4]), dtype=np.uint8)
            else:
                m = np.concatenate([m, rng.integers(0, 256, rng.integers(1, 20), dtype=np.uint8)])
            comps.append(m)
            caps.append(len(s) if trial % 3 else len(s) + 77)
    res, outs, _, _ = _decode_batch(gpu, comps, caps)
    checked = 0
    for i, (m, cap) in enumerate(zip(comps, caps)):
        want_r, want = helpers.orc_decompress(m, cap)
        if want_r == -(2 ** 31):                 # offset 0: undefined in the reference, both reject
            assert res[i] == -(2 ** 31)
            continue
        assert res[i] == want_r, (i, res[i], want_r)
        if want_r >= 0:
            assert np.array_equal(outs[i], want), i
        checked += 1
    assert checked > 200


# ------------------------------------------------------------------------------------------ encode
def _encode_batch(gpu, srcs, caps):
    buf, offs = _pack(srcs, 1)
    dsts, pos = [], 0
    for c in caps:
        dsts.append(pos); pos += c + 40
    blocks = gpu.make_blocks(offs, dsts, [len(s) for s in srcs], caps)
    batch = gpu.DeviceBatch(blocks)
    d_out = torch.full((pos + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    gpu.lz4_compress_fast(_dev(buf), d_out, batch)
    res = batch.download()["result"]
    out = d_out.cpu().numpy()
    return res, [out[d:d + max(r, 0)] for d, r in zip(dsts, res)]


def test_lz4_encode_bytes_identical_edge_inputs(gpu):
    inputs = helpers.edge_inputs()
    names = list(inputs)
    srcs = [inputs[k] for k in names]
    bound = [helpers.oracle().orc_lz4_compress_bound(len(s)) for s in srcs]
    for mode, caps in (("bound", bound), ("n-1", [max(len(s) - 1, 0) for s in srcs])):
        res, outs = _encode_batch(gpu, srcs, caps)
        for k, s, cap, r, o in zip(names, srcs, caps, res, outs):
            want_r, want = helpers.orc_compress(s, cap)
            assert r == want_r, (mode, k, r, want_r)
            assert np.array_equal(o, want), (mode, k)


def test_lz4_encode_capacity_sweep(gpu):
    """limitedOutput (lz4.c:1083-1107,1184-1196,1266-1293): the same input at capacities around its compressed size
    and at tiny ones - return value (0 = does not fit) and every byte must equal the reference's."""
    inputs = helpers.edge_inputs()
    rng = np.random.default_rng(11)
    for name in ("text_300k", "two_symbols", "far_repeat", "random_100k", "period37", "zeros_64k_limit"):
        s = inputs[name]
        full, _ = helpers.orc_compress(s, helpers.oracle().orc_lz4_compress_bound(len(s)))
        caps = sorted(set([0, 1, 5, 13, 64, 200, 1000] + list(range(max(full - 40, 1), full + 6)) +
                          [int(c) for c in rng.integers(1, full + 300, 24)] + [full - 300, full - 1000, len(s) - 1, len(s)]))
        caps = [c for c in caps if c >= 0]
        res, outs = _encode_batch(gpu, [s] * len(caps), caps)
        for cap, r, o in zip(caps, res, outs):
            want_r, want = helpers.orc_compress(s, cap)
            assert r == want_r, (name, cap, r, want_r)
            assert np.array_equal(o, want), (name, cap)


def test_lz4_encode_sizes_around_the_dense_window_limits(gpu):
    """Block lengths around the byU16/byU32 switch and around every length at which the dense window hands over to the
    strided batch near the end of a block (a few bytes more or less of tail change which path takes the last sequences)."""
    text = helpers.corpus(B)[:140000]
    sizes = list(range(13, 40)) + list(range(120, 150)) + list(range(65530, 65560)) + list(range(70000, 70140, 7)) + [139999, 140000]
    srcs = [text[:n] for n in sizes] + [np.tile(text[1000:1064], 40)[:n] for n in range(130, 400, 9)]
    caps = [helpers.oracle().orc_lz4_compress_bound(len(s)) for s in srcs]
    for mode, cc in (("bound", caps), ("n-1", [len(s) - 1 for s in srcs])):
        res, outs = _encode_batch(gpu, srcs, cc)
        for s, cap, r, o in zip(srcs, cc, res, outs):
            want_r, want = helpers.orc_compress(s, cap)
            assert r == want_r, (mode, len(s), r, want_r)
            assert np.array_equal(o, want), (mode, len(s))


def test_lz4_encode_bytes_identical_corpus(gpu):
    data = helpers.corpus(12 * B + 123457)       # 12 full blocks + a ragged tail block
    srcs = [data[b * B:(b + 1) * B] for b in range(13)]
    res, outs = _encode_batch(gpu, srcs, [max(len(s) - 1, 0) for s in srcs])
    for b, s in enumerate(srcs):
        want_r, want = helpers.orc_compress(s, len(s) - 1)
        assert res[b] == want_r, (b, res[b], want_r)
        assert np.array_equal(outs[b], want), b


# ------------------------------------------------------------------------------------------ container
def test_container_encode_decode_blocks(gpu):
    n = 12 * B + 54321
    data = helpers.corpus(n)
    nb = -(-n // B)
    lens = [min(B, n - b * B) for b in range(nb)]
    blocks = gpu.make_blocks([b * B for b in range(nb)], [b * B for b in range(nb)], lens, lens)
    batch = gpu.DeviceBatch(blocks)
    d_src = _dev(data)
    d_dst = torch.zeros(nb * B, dtype=torch.uint8, device="cuda")
    gpu.encode_blocks(d_src, d_dst, batch)
    enc = batch.download()
    out = d_dst.cpu().numpy()
    payloads = [out[b * B: b * B + enc["result"][b]] for b in range(nb)]
    image = gpu.assemble_container(gpu.MAGIC_4MC, lens, enc["result"], enc["xxh32"], payloads)
    want = helpers.orc_container(data)
    assert image == want.tobytes(), "container bytes differ from the oracle's"
    assert any(enc["result"][b] == lens[b] for b in range(nb)), "corpus should exercise a stored block"

    # decode the image in place from HBM
    img = np.frombuffer(image, dtype=np.uint8)
    dblocks, used = gpu.split_container(img, gpu.MAGIC_4MC)
    assert used == len(image)
    dbatch = gpu.DeviceBatch(dblocks)
    d_img = _dev(np.concatenate([img, np.zeros(64, np.uint8)]))
    d_out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    gpu.decode_blocks(d_img, d_out, dbatch)
    dec = dbatch.download()
    assert list(dec["result"]) == lens
    assert np.array_equal(d_out.cpu().numpy()[:n], data)

    # a flipped payload byte must be reported as a checksum failure for that block only
    bad = img.copy(); bad[int(dblocks["src_off"][3]) + 1000] ^= 0x10
    dbatch = gpu.DeviceBatch(dblocks)
    gpu.decode_blocks(_dev(np.concatenate([bad, np.zeros(64, np.uint8)])), d_out, dbatch)
    r = dbatch.download()["result"]
    assert r[3] == gpu.BLK_BADSUM and all(r[b] == lens[b] for b in range(nb) if b != 3)


# ------------------------------------------------------------------------------------------ host / CLI
def test_host_block_calls(gpu):
    L = gpu.lib()
    s = helpers.edge_inputs()["text_300k"]
    bound = L.fourmc_LZ4_compressBound(len(s))
    dst = np.zeros(bound, np.uint8)
    r = L.fourmc_LZ4_compress_default(s.ctypes.data, dst.ctypes.data, len(s), bound)
    want_r, want = helpers.orc_compress(s)
    assert r == want_r and np.array_equal(dst[:r], want)
    back = np.zeros(len(s), np.uint8)
    assert L.fourmc_LZ4_decompress_safe(dst.ctypes.data, back.ctypes.data, r, len(s)) == len(s)
    assert np.array_equal(back, s)
    assert L.fourmc_LZ4_decompress_safe(dst.ctypes.data, back.ctypes.data, r - 1, len(s)) < 0


def test_cli_roundtrip_and_bytes(gpu, tmp_path):
    cli = gpu.cli_path()
    n = 5 * B + 777
    data = helpers.corpus(n, first_block=7)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    out = tmp_path / "in.bin.4mc"; back = tmp_path / "back.bin"
    env = dict(os.environ, FOURMC_BATCH_BLOCKS="4")
    r = subprocess.run([cli, "-f", str(src), str(out)], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr
    assert b"Compressed (fast)" in r.stderr
    assert out.read_bytes() == helpers.orc_container(data).tobytes()
    r = subprocess.run([cli, "-d", "-f", str(out), str(back)], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == data.tobytes()
    # corruption -> exit 4 with the reference's message
    img = bytearray(out.read_bytes()); img[5000] ^= 1
    bad = tmp_path / "bad.4mc"; bad.write_bytes(bytes(img))
    r = subprocess.run([cli, "-d", "-f", str(bad), str(back)], capture_output=True, env=env)
    assert r.returncode == 4 and b"invalid block checksum detected" in r.stderr
    # empty input -> 44-byte file, golden vector from SURVEY.md §8(c)
    e = tmp_path / "empty"; e.write_bytes(b"")
    r = subprocess.run([cli, "-f", str(e), str(tmp_path / "e.4mc")], capture_output=True, env=env)
    assert r.returncode == 0
    assert (tmp_path / "e.4mc").read_bytes().hex() == (
        "344d430000000001a4b73443" + "00" * 12 + "00000014" "00000001" "00000014" "344d4300" "849b8d65")


def test_footer_index_random_access_decode(gpu, tmp_path):
    """§8(f)1: decode an arbitrary block range of a .4mc / .4mz file through its footer index (a Hadoop split)."""
    import ctypes as C
    n = 9 * B + 12345
    data = helpers.corpus(n, first_block=3)
    src = tmp_path / "c.bin"; src.write_bytes(data.tobytes())
    L = gpu.lib()
    for flags, ext in (([], ".4mc"), (["-z"], ".4mz")):
        out = tmp_path / ("c" + ext)
        assert subprocess.run([gpu.cli_path(), *flags, "-1", "-f", str(src), str(out)], capture_output=True).returncode == 0
        isz = C.c_int(-1)
        assert L.fourmc_file_block_count(str(out).encode(), C.byref(isz)) == 10 and isz.value == (ext == ".4mz")
        for first, count in ((0, 1), (3, 4), (9, 1), (0, 10), (7, 3)):
            buf = np.zeros(count * B, np.uint8)
            r = L.fourmc_file_decode_blocks(str(out).encode(), first, count, buf.ctypes.data, buf.size)
            want = data[first * B: min(n, (first + count) * B)]
            assert r == len(want) and np.array_equal(buf[:r], want), (ext, first, count, r)
        buf = np.zeros(B, np.uint8)
        assert L.fourmc_file_decode_blocks(str(out).encode(), 8, 3, buf.ctypes.data, buf.size) == -3      # range
        assert L.fourmc_file_decode_blocks(str(out).encode(), 2, 2, buf.ctypes.data, buf.size) == -5      # capacity
        raw = bytearray(out.read_bytes()); raw[12 + 12 + 1000] ^= 0x40                                   # corrupt block 0's payload
        bad = tmp_path / ("bad" + ext); bad.write_bytes(bytes(raw))
        assert L.fourmc_file_decode_blocks(str(bad).encode(), 0, 1, buf.ctypes.data, buf.size) == -4
        assert L.fourmc_file_decode_blocks(str(bad).encode(), 1, 1, buf.ctypes.data, buf.size) == B       # other blocks unaffected
    assert L.fourmc_file_block_count(str(src).encode(), None) == -2                                      # not a 4mc file
