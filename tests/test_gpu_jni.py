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
    out = {}
    for line in r.stdout.splitlines():
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
        assert np.array_equal(np.fromfile(tmp_path / (name + ".bin"), dtype=np.uint8), wbytes), name
        d, rest = out[name + "_roundtrip"]
        assert d == n and "same=1" in rest and "clen_after=0" in rest, (name, d, rest)
    # a zstd level that is not on the device: error code returned AND InternalError thrown, buffer length untouched
    r6, rest = out["Zstd_compressBytesDirectHC12"]
    assert "java/lang/InternalError: ZSTD_compress returned: " in rest and ("ulen_after=%d" % n) in rest
    for codec, fn in (("Lz4", "LZ4_decompress_safe"), ("Zstd", "LZ4_decompress_safe")):   # zstd reuses the text (jniZstdDecompressor.c:96)
        d, rest = out[codec + "_decompress_garbage"]
        assert d < 0 and ("java/lang/InternalError: %s returned: %d" % (fn, d)) in rest, (codec, d, rest)
