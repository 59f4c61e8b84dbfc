"""GPU: ZSTD frame decode (4mz payloads) against the reference's own output.
Checker = committed golden frames written by the reference's ZSTD_compress (tests/golden/zstd_frames.json)
and, when oracle/_ref travelled with the snapshot, the reference codec itself on corpus blocks."""
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


def _decode(gpu, frames, caps):
    offs, pos = [], 3                                  # odd base offset: payloads are unaligned in a .4mz
    for f in frames:
        offs.append(pos); pos += len(f) + 5
    buf = np.zeros(pos + 64, np.uint8)
    for f, o in zip(frames, offs):
        buf[o:o + len(f)] = np.frombuffer(bytes(f), np.uint8)
    dsts, dpos = [], 0
    for c in caps:
        dsts.append(dpos); dpos += c + 64
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, dsts, [len(f) for f in frames], caps))
    d_out = torch.full((dpos + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    gpu.zstd_decompress(torch.from_numpy(buf).cuda(), d_out, batch)
    res = batch.download()["result"]
    out = d_out.cpu().numpy()
    return res, [out[d:d + max(r, 0)] for d, r in zip(dsts, res)], out, dsts


def test_zstd_golden_frames(gpu):
    z = json.load(open(os.path.join(G, "zstd_frames.json")))
    names, frames, caps = [], [], []
    for name, e in z.items():
        for lvl, hx in e["frames"].items():
            for slack in (0, 300):
                names.append((name, lvl, slack)); frames.append(bytes.fromhex(hx)); caps.append(e["input_bytes"] + slack)
    res, outs, raw, dsts = _decode(gpu, frames, caps)
    for (name, lvl, slack), r, o, c, d in zip(names, res, outs, caps, dsts):
        e = z[name]
        assert r == e["input_bytes"], (name, lvl, slack, r)
        assert hashlib.sha256(o.tobytes()).hexdigest() == e["input_sha256"], (name, lvl)
        assert np.all(raw[d + c:d + c + 64] == 0xA5), "wrote past capacity"


def test_zstd_small_4mz_container_files(gpu):
    files = json.load(open(os.path.join(G, "small_files.json")))
    for name, f in files.items():
        img = np.frombuffer(bytes.fromhex(f["4mz_fast_hex"]), np.uint8)
        want = bytes.fromhex(f["input_hex"])
        blocks, used = gpu.split_container(img, gpu.MAGIC_4MZ)
        assert used == len(img)
        if len(blocks) == 0:
            assert want == b""; continue
        batch = gpu.DeviceBatch(blocks)
        d_out = torch.zeros(len(want) + 64, dtype=torch.uint8, device="cuda")
        gpu.decode_blocks(torch.from_numpy(np.concatenate([img, np.zeros(64, np.uint8)])).cuda(), d_out, batch, codec=gpu.CODEC_ZSTD)
        r = batch.download()["result"]
        assert int(r.sum()) == len(want), (name, r)
        assert d_out.cpu().numpy()[: len(want)].tobytes() == want, name


@pytest.mark.skipif(helpers.ref() is None, reason="oracle/_ref not present")
def test_zstd_corpus_blocks_all_levels_vs_reference(gpu):
    ref = helpers.ref()
    data = helpers.corpus(12 * B)
    frames, caps, srcs = [], [], []
    for lvl in (1, 3, 6, 12):
        for b in range(12):
            if lvl == 12 and b % 3:            # level 12 is slow on the host: sample it
                continue
            src = data[b * B:(b + 1) * B]
            out = np.zeros(B + 65536, np.uint8)
            r = ref.ZSTD_compress(out.ctypes.data, len(out), src.ctypes.data, B, lvl)
            assert not ref.ZSTD_isError(r)
            frames.append(out[:r].tobytes()); caps.append(B); srcs.append(src)
    res, outs, _, _ = _decode(gpu, frames, caps)
    for i, (r, o, s) in enumerate(zip(res, outs, srcs)):
        assert r == B, (i, r)
        assert np.array_equal(o, s), i


@pytest.mark.skipif(helpers.ref() is None, reason="oracle/_ref not present")
def test_zstd_corrupt_frames_rejected_like_reference(gpu):
    """Accept/reject agrees with ZSTD_decompress on mutated frames; accepted outputs are identical."""
    ref = helpers.ref()
    rng = np.random.default_rng(17)
    z = json.load(open(os.path.join(G, "zstd_frames.json")))
    frames, caps = [], []
    for name in ("text_30k", "lit_then_run_30k", "two_symbols_30k"):
        for lvl in ("1", "6"):
            base = np.frombuffer(bytes.fromhex(z[name]["frames"][lvl]), np.uint8)
            for t in range(40):
                m = base.copy()
                k = t % 3
                if k == 0: m[rng.integers(0, len(m))] ^= 1 << rng.integers(0, 8)
                elif k == 1: m = m[: rng.integers(1, len(m))]
                else:
                    i = rng.integers(4, len(m)); m[i:i + 2] = rng.integers(0, 256, len(m[i:i + 2]), dtype=np.uint8)
                frames.append(m.tobytes()); caps.append(z[name]["input_bytes"])
    res, outs, _, _ = _decode(gpu, frames, caps)
    lenient = 0
    for f, c, r, o in zip(frames, caps, res, outs):
        src = np.frombuffer(f, np.uint8); dst = np.zeros(c + 64, np.uint8)
        rr = ref.ZSTD_decompress(dst.ctypes.data, c, src.ctypes.data, len(src))
        want_r, want = helpers.orc_zstd_decompress(f, c)
        assert (r < 0) == (want_r < 0) and (r < 0 or (r == want_r and np.array_equal(o, want)))   # == oracle, always
        if ref.ZSTD_isError(rr):
            assert r < 0, (r, rr)
        elif r >= 0:
            assert r == rr and np.array_equal(o, dst[:rr])
        else:
            lenient += 1            # see tests/test_oracle_golden.py: none may be left since round 3
    assert lenient == 0


def test_cli_decodes_4mz_files(gpu, tmp_path):
    """`4mc -d file.4mz` through the GPU engine: golden small files + (if the reference CLI travelled)
    multi-block files written by the reference at every zstd level."""
    import subprocess
    cli = gpu.cli_path()
    files = json.load(open(os.path.join(G, "small_files.json")))
    for name, f in files.items():
        src = tmp_path / f"{name}.4mz"; src.write_bytes(bytes.fromhex(f["4mz_fast_hex"]))
        out = tmp_path / f"{name}.out"
        r = subprocess.run([cli, "-d", "-z", "-f", str(src), str(out)], capture_output=True)
        assert r.returncode == 0, (name, r.stderr)
        assert out.read_bytes() == bytes.fromhex(f["input_hex"]), name
    ref_cli = helpers.ref_cli()
    if ref_cli:
        data = helpers.corpus(3 * B + 4567, first_block=9)
        raw = tmp_path / "c.bin"; raw.write_bytes(data.tobytes())
        for lvl in (1, 2, 3, 4):
            z = tmp_path / f"c{lvl}.4mz"
            assert subprocess.run([ref_cli, "-z", f"-{lvl}", "-f", str(raw), str(z)], capture_output=True).returncode == 0
            back = tmp_path / f"c{lvl}.out"
            r = subprocess.run([cli, "-d", "-z", "-f", str(z), str(back)], capture_output=True, env=dict(os.environ, FOURMC_BATCH_BLOCKS="2"))
            assert r.returncode == 0, r.stderr
            assert back.read_bytes() == data.tobytes(), lvl
        # a .4mc written by the reference at the HC levels decodes too (decode is level independent)
        for lvl in (2, 3, 4):
            z = tmp_path / f"c{lvl}.4mc"
            assert subprocess.run([ref_cli, f"-{lvl}", "-f", str(raw), str(z)], capture_output=True).returncode == 0
            back = tmp_path / f"c{lvl}.lz4out"
            r = subprocess.run([cli, "-d", "-f", str(z), str(back)], capture_output=True)
            assert r.returncode == 0 and back.read_bytes() == data.tobytes(), lvl
        bad = bytearray((tmp_path / "c1.4mz").read_bytes()); bad[100000] ^= 4
        (tmp_path / "bad.4mz").write_bytes(bytes(bad))
        r = subprocess.run([cli, "-d", "-z", "-f", str(tmp_path / "bad.4mz"), str(tmp_path / "x")], capture_output=True)
        assert r.returncode == 4 and b"invalid block checksum detected" in r.stderr


def _shaped_blocks():
    """Inputs chosen for the execute kernel (zstd_exec.inc): bulk records (stored and RLE inner blocks, literal runs above 1 KiB,
    literals behind the last sequence), matches above 1 KiB and at distances up to the whole block, overlapping matches of
    small periods, groups cut by the literal / match caps, ragged sizes."""
    rng = np.random.default_rng(2024)
    text = helpers.corpus(B)[: B]
    out = []
    out.append(np.zeros(B, np.uint8))                                                     # RLE blocks
    out.append(rng.integers(0, 256, B, dtype=np.uint8))                                   # stored (the container keeps it raw)
    a = rng.integers(0, 256, 300 << 10, dtype=np.uint8)
    out.append(np.concatenate([a, text[: 200 << 10], a, a[: 100 << 10], text[: 200 << 10], a])[: B].copy())   # long matches, far away
    b = text[: B].copy()
    for i in range(0, B - 8192, 8192): b[i:i + 3000] = rng.integers(0, 256, 3000, dtype=np.uint8)              # literal runs above 1 KiB
    out.append(b)
    c = np.zeros(B, np.uint8)
    pos = 0
    for k, per in enumerate([1, 2, 3, 5, 7, 13, 31, 64, 100, 257, 1000] * 40):          # overlapping matches of many periods
        n = 2000 + 137 * (k % 11)
        if pos + n > B: break
        c[pos:pos + per] = rng.integers(0, 256, per, dtype=np.uint8)
        for i in range(pos + per, pos + n): c[i] = c[i - per]
        pos += n
    c[pos:] = text[: B - pos]
    out.append(c)
    d = np.concatenate([text[: 1 << 20], np.zeros(1 << 20, np.uint8), rng.integers(0, 256, 1 << 20, dtype=np.uint8), text[: 1 << 20]])
    out.append(d)                                                                         # compressed, RLE and stored inner blocks in one frame
    for n in (1, 100, 16384, (128 << 10) + 1, (1 << 20) + 12345):
        out.append(text[5000: 5000 + n].copy())
    return out


def test_zstd_execute_kernel_shapes_both_paths(gpu):
    """The two-kernel decode (entropy stage + execute kernel, zstd_exec.inc) and the one-wave kernel give the input back, on frames
    of every shape the execute kernel treats specially, at levels 1 and 3; on damaged frames they give the same verdicts and bytes."""
    srcs = _shaped_blocks()
    gpu.use_research(True); gpu.gpu_init()             # the counters of the execute kernel are a debug export: research side build
    lib = gpu.lib()
    before = lib.fourmc_gpu_get_zstd_decode_split()
    try:
        for level in (1, 3):
            frames = []
            for s in srcs:
                r, comp = helpers.orc_zstd_compress(s, level, len(s) + 1024)
                assert r > 0
                frames.append(bytes(comp[:r]))
            caps = [len(s) for s in srcs]
            got = {}
            import ctypes as C
            done, back = C.c_ulonglong(0), C.c_ulonglong(0)
            for split in (1, 0):
                lib.fourmc_gpu_set_zstd_decode_split(split)
                lib.fourmc_gpu_debug_zstd_exec_counts(C.byref(done), C.byref(back))     # (reset)
                res, outs, _, _ = _decode(gpu, frames, caps)
                for i, (r, o, s) in enumerate(zip(res, outs, srcs)):
                    assert r == len(s), (level, split, i, r)
                    assert np.array_equal(o, s), (level, split, i)
                assert lib.fourmc_gpu_debug_zstd_exec_counts(C.byref(done), C.byref(back)) == 0
                # every frame is one the execute kernel takes, and none comes back (valid input)
                assert (done.value, back.value) == ((len(frames), 0) if split else (0, 0)), (split, done.value, back.value)
            # damaged copies of the big frames: the verdict of the two paths is the same, and so are accepted bytes
            rng = np.random.default_rng(5 + level)
            bad, bcaps = [], []
            for i in (2, 3, 4, 5):
                base = np.frombuffer(frames[i], np.uint8)
                for t in range(6):
                    m = base.copy()
                    if t % 2 == 0: m[rng.integers(16, len(m))] ^= 1 << rng.integers(0, 8)
                    else: j = rng.integers(16, len(m) - 4); m[j:j + 3] = rng.integers(0, 256, 3, dtype=np.uint8)
                    bad.append(m.tobytes()); bcaps.append(caps[i])
            for split in (1, 0):
                lib.fourmc_gpu_set_zstd_decode_split(split)
                res, outs, _, _ = _decode(gpu, bad, bcaps)
                got[split] = (res.copy(), [o.copy() for o in outs])
            assert np.array_equal(got[0][0], got[1][0]), (got[0][0], got[1][0])
            for a, b2, r in zip(got[0][1], got[1][1], got[0][0]):
                if r >= 0: assert np.array_equal(a, b2)
    finally:
        lib.fourmc_gpu_set_zstd_decode_split(before)
        gpu.use_research(False)
