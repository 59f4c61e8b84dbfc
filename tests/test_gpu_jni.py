"""GPU: the 35 JNI entry points of libhadoop-4mc.so driven through a mock JNIEnv (tests/jni_mock/mock_jni.c,
no JVM): field protocol, return values, InternalError text, and bytes equal to the oracle's for every codec call
(native/jniCompressor.c:72-168, jniDecompressor.c:67-100, jniZstdCompressor.c:74-173, jniZstdDecompressor.c:69-102)."""
import os
import subprocess

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
ROOT = helpers.ROOT


def test_jni_entry_points_through_mock_env(gpu, tmp_path):
    exe = tmp_path / "mock_jni"
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "jni_mock", "mock_jni.c"),
                    "-ldl", "-o", str(exe)], check=True)
    n = 1_500_000
    data = helpers.corpus(n, first_block=6)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    r = subprocess.run([str(exe), gpu.lib_path(), str(src), str(n), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    check_jni_protocol(r.stdout, data, tmp_path, level15_on_device=False)


def test_jni_lz4_compressor_under_the_tolerance_switch(gpu, tmp_path):
    """FOURMC_LZ4_ENCODE=parallel in the JVM's environment: Lz4Compressor.compressBytesDirect hands out the ratio-tolerance encoder's
    payload (a valid LZ4 block - the model's bytes - that Lz4Decompressor and the oracle decode); HC / MC / zstd entry points are
    untouched, the protocol (length fields reset, no exception) is the same."""
    exe = tmp_path / "mock_jni"
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "jni_mock", "mock_jni.c"),
                    "-ldl", "-o", str(exe)], check=True)
    n = 1_500_000
    data = helpers.corpus(n, first_block=6)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    r = subprocess.run([str(exe), gpu.lib_path(), str(src), str(n), str(tmp_path)], capture_output=True, text=True,
                       env=dict(os.environ, FOURMC_LZ4_ENCODE="parallel"))
    assert r.returncode == 0, r.stderr
    out = {}
    for line in r.stdout.splitlines():
        name, val, rest = line.split(" ", 2)
        out[name] = (int(val), rest)
    bound = helpers.oracle().orc_lz4_compress_bound(n)
    got_r, rest = out["Lz4_compressBytesDirect"]
    mr, mb = helpers.lz4p_model_encode(data, bound)
    assert got_r == mr and "ulen_after=0" in rest
    payload = np.fromfile(os.path.join(str(tmp_path), "Lz4_compressBytesDirect.bin"), dtype=np.uint8)
    assert np.array_equal(payload, mb)
    dr, back = helpers.orc_decompress(payload, n)
    assert dr == n and np.array_equal(back[:n], data)
    d, rest = out["Lz4_compressBytesDirect_roundtrip"]
    assert d == n and "same=1" in rest
    wr, wbytes = helpers.orc_compress_hc(data, 4, bound)                              # the other encoders: the reference's bytes, as ever
    assert out["Lz4_compressBytesDirectHC"][0] == wr
    assert np.array_equal(np.fromfile(os.path.join(str(tmp_path), "Lz4_compressBytesDirectHC.bin"), dtype=np.uint8), wbytes)


def check_jni_protocol(stdout, data, out_dir, level15_on_device):
    """What the mock driver's output has to show, whichever library it drove: this repository's (on the GPU) or the reference's
    shipped artefact (tests/test_jni_reference_artifact.py, on the CPU) - the same expectations for both."""
    n = len(data)
    out = {}
    for line in stdout.splitlines():
        name, val, rest = line.split(" ", 2)
        out[name] = (int(val), rest)
    bound = helpers.oracle().orc_lz4_compress_bound(n)
    assert out["Lz4_compressBound"][0] == bound and out["Zstd_compressBound"][0] == helpers.zstd_bound(n)
    assert out["Lz4_xxhash32"][0] == np.int32(np.uint32(helpers.orc_xxh32(data[3:103]))) == out["Zstd_xxhash32"][0]
    want = {
        "Lz4_compressBytesDirect": helpers.orc_compress(data, bound),
        "Lz4_compressBytesDirectMC": helpers.orc_compress_mc(data, -1),
        "Lz4_compressBytesDirectHC": helpers.orc_compress_hc(data, 4, bound),
        "Zstd_compressBytesDirect": helpers.orc_zstd_compress(data, 1),
        "Zstd_compressBytesDirectMC": helpers.orc_zstd_compress(data, 3),      # zstd level 3 (jniZstdCompressor.c:126)
        "Zstd_compressBytesDirectHC": helpers.orc_zstd_compress(data, 1),      # the driver passes level 1
    }
    for name, (wr, wbytes) in want.items():
        got_r, rest = out[name]
        assert got_r == wr and rest.startswith("- ") and "ulen_after=0" in rest, (name, got_r, wr, rest)
        assert np.array_equal(np.fromfile(os.path.join(str(out_dir), name + ".bin"), dtype=np.uint8), wbytes), name
        d, rest = out[name + "_roundtrip"]
        assert d == n and "same=1" in rest and "clen_after=0" in rest, (name, d, rest)
    # level 9: a level the shipped Java classes never pass - the reference serves every level, the device levels 1..12: a frame, no
    # exception, the buffer length reset
    r9, rest = out["Zstd_compressBytesDirectHC_level9"]
    assert r9 > 0 and "InternalError" not in rest and "ulen_after=0" in rest, (r9, rest)
    r15, rest = out["Zstd_compressBytesDirectHC_level15"]
    if level15_on_device:
        assert r15 > 0 and "InternalError" not in rest and "ulen_after=0" in rest, (r15, rest)
    else:
        # a zstd level that is not on the device: error code returned AND InternalError thrown, buffer length untouched
        assert "java/lang/InternalError: ZSTD_compress returned: " in rest and ("ulen_after=%d" % n) in rest
    for codec, fn in (("Lz4", "LZ4_decompress_safe"), ("Zstd", "LZ4_decompress_safe")):   # zstd reuses the text (jniZstdDecompressor.c:96)
        d, rest = out[codec + "_decompress_garbage"]
        assert d < 0 and ("java/lang/InternalError: %s returned: %d" % (fn, d)) in rest, (codec, d, rest)


def check_block_stream(stdout, data, out_dir, chunk):
    """BlockCompressorStream-shaped use of the raw codecs (Lz4Codec.java:95-104): every small buffer's compressed bytes equal the
    oracle's for that buffer, every one decompresses back, nothing is thrown, the length fields are reset after every call."""
    n = len(data)
    calls = -(-n // chunk)
    for codec, orc in (("Lz4", lambda s: helpers.orc_compress(s, helpers.oracle().orc_lz4_compress_bound(len(s)))),
                       ("Zstd", lambda s: helpers.orc_zstd_compress(s, 1))):
        line = [l for l in stdout.splitlines() if l.startswith(codec + "_stream ")][0]
        assert line.split()[1] == str(calls) and "bad=0 thrown=0" in line, line
        sizes = [int(x) for x in open(os.path.join(str(out_dir), codec + "_stream.sizes")).read().split()]
        blob = np.fromfile(os.path.join(str(out_dir), codec + "_stream.bin"), dtype=np.uint8)
        assert len(sizes) == calls and sum(sizes) == len(blob)
        at = 0
        for k, sz in enumerate(sizes):
            r, want = orc(data[k * chunk: min(n, (k + 1) * chunk)])
            assert sz == r and np.array_equal(blob[at: at + sz], want), (codec, k, sz, r)
            at += sz


def test_raw_codecs_in_block_compressor_stream_shape(gpu, tmp_path):
    """many small blocks through the JNI entry points (64 KiB-class buffers, directBufferSize far from 4 MiB): the call pattern of
    Hadoop's BlockCompressorStream around Lz4Codec / ZstdCodec"""
    exe = tmp_path / "mock_jni"
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "jni_mock", "mock_jni.c"),
                    "-ldl", "-o", str(exe)], check=True)
    n, chunk = 1_000_000, 65536 - 65536 // 255 - 16
    data = helpers.corpus(n, first_block=9)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    r = subprocess.run([str(exe), gpu.lib_path(), str(src), str(n), str(tmp_path), str(chunk)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    check_block_stream(r.stdout, data, tmp_path, chunk)


def test_concurrent_one_block_calls_share_launches(gpu):
    """One JNI call is one block; calls from concurrent threads are combined inside the library into
    shared launches.  Results must equal the oracle's for every call, whatever was batched with what."""
    import ctypes as C, threading
    import numpy as np
    import helpers
    L = gpu.binding.lib()                              # the PRODUCT library: its queue statistics are a read-only export (fourmc_gpu_one_block_stats)
    data = helpers.corpus(helpers.B)
    rng = np.random.default_rng(5)
    jobs = []
    for i in range(96):
        n = int(rng.integers(1000, 300000)); o = int(rng.integers(0, helpers.B - n))
        jobs.append((i, np.ascontiguousarray(data[o:o + n]), ("fast", "hc", "zstd", "dec")[i % 4]))
    c0, l0 = C.c_ulonglong(), C.c_ulonglong()
    L.fourmc_gpu_one_block_stats(C.byref(c0), C.byref(l0))
    results = {}
    def work(job):
        i, s, kind = job
        bound = helpers.oracle().orc_lz4_compress_bound(len(s))
        if kind == "fast":
            dst = np.empty(bound, np.uint8)
            r = L.fourmc_LZ4_compress_default(s.ctypes.data, dst.ctypes.data, len(s), bound)
            results[i] = (r, dst[:max(r, 0)].copy())
        elif kind == "hc":
            dst = np.empty(bound, np.uint8)
            r = L.fourmc_LZ4_compress_HC(s.ctypes.data, dst.ctypes.data, len(s), bound, 4)
            results[i] = (r, dst[:max(r, 0)].copy())
        elif kind == "zstd":
            cap = helpers.zstd_bound(len(s)); dst = np.empty(cap, np.uint8)
            r = L.fourmc_ZSTD_compress(dst.ctypes.data, cap, s.ctypes.data, len(s), 1)
            results[i] = (int(r), dst[:max(int(r), 0)].copy())
        else:
            wr, wb = helpers.orc_compress(s, bound)
            out = np.empty(len(s), np.uint8)
            r = L.fourmc_LZ4_decompress_safe(wb.ctypes.data, out.ctypes.data, wr, len(s))
            results[i] = (r, out[:max(r, 0)].copy())
    threads = [threading.Thread(target=lambda part=jobs[t::12]: [work(j) for j in part]) for t in range(12)]
    for t in threads: t.start()
    for t in threads: t.join()
    for i, s, kind in jobs:
        r, out = results[i]
        if kind == "fast": wr, wb = helpers.orc_compress(s, helpers.oracle().orc_lz4_compress_bound(len(s)))
        elif kind == "hc": wr, wb = helpers.orc_compress_hc(s, 4)
        elif kind == "zstd": wr, wb = helpers.orc_zstd_compress(s, 1, helpers.zstd_bound(len(s)))
        else: wr, wb = len(s), s
        assert r == wr, (i, kind, r, wr)
        assert np.array_equal(out, wb), (i, kind)
    c1, l1 = C.c_ulonglong(), C.c_ulonglong()
    L.fourmc_gpu_one_block_stats(C.byref(c1), C.byref(l1))
    assert c1.value - c0.value == len(jobs)
    assert l1.value - l0.value < len(jobs), "no two concurrent calls ever shared a launch"
