"""CPU: the mock-JNIEnv driver (tests/jni_mock/mock_jni.c) run against the REFERENCE's shipped libhadoop-4mc.so with the
expectations the GPU test applies to this repository's library (tests/test_gpu_jni.py::check_jni_protocol): the JNI protocol -
field names and types read, buffer lengths reset, InternalError texts, return values - is thereby checked against the artefact,
not against a reading of native/jniCompressor.c:72-168.  The compressed bytes the
artefact returns are compared with the oracle's as well: one more pin of the oracle, on the prebuilt binary.
Skipped where the reference tree is absent (the GPU box)."""
import os
import subprocess

import pytest

import helpers
from test_gpu_jni import check_jni_protocol, check_block_stream

REF_SO = "/root/reference/java/hadoop-4mc/src/main/resources/com/fing/compression/fourmc/linux/amd64/libhadoop-4mc.so"


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="the reference's shipped library is not on this machine")
def test_mock_jni_driver_against_the_reference_artifact(tmp_path):
    exe = tmp_path / "mock_jni"
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(helpers.ROOT, "include"), os.path.join(helpers.ROOT, "tests", "jni_mock", "mock_jni.c"),
                    "-ldl", "-o", str(exe)], check=True)
    n = 1_500_000
    data = helpers.corpus(n, first_block=6)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    r = subprocess.run([str(exe), REF_SO, str(src), str(n), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    check_jni_protocol(r.stdout, data, tmp_path, level15_on_device=True)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="the reference's shipped library is not on this machine")
def test_block_stream_shape_against_the_reference_artifact(tmp_path):
    """the BlockCompressorStream call pattern (64 KiB-class buffers) on the artefact: the expectations of the GPU test hold there"""
    exe = tmp_path / "mock_jni"
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(helpers.ROOT, "include"), os.path.join(helpers.ROOT, "tests", "jni_mock", "mock_jni.c"),
                    "-ldl", "-o", str(exe)], check=True)
    n, chunk = 1_000_000, 65536 - 65536 // 255 - 16
    data = helpers.corpus(n, first_block=9)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    r = subprocess.run([str(exe), REF_SO, str(src), str(n), str(tmp_path), str(chunk)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    check_block_stream(r.stdout, data, tmp_path, chunk)


def _zstream(exe, lib, src, n, outdir, chunk, decode=None):
    cmd = [str(exe), lib, str(src), str(n), str(outdir), "zstream", str(chunk)] + ([str(decode)] if decode else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return {l.split()[0]: l for l in r.stdout.splitlines()}


def test_streaming_zstcodec_pass_through(tmp_path):
    """The streaming ZstCodec entry points (a host pass-through; native/jniZStreamCompressor.c:65-134,
    jniZStreamDecompressor.c:66-112, jniZstd.c:49-104) through the mock JNIEnv in the call pattern of ZstdStreamOutputStream /
    InputStream: this repository's library (libzstd.so.1 of the host behind it) gives the input back, reports the reference's stream
    buffer sizes and error protocol - and, where the reference's shipped library is present, each side reads the other's stream
    (standard zstd frames) and the artefact meets the same expectations."""
    exe = tmp_path / "mock_jni"
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(helpers.ROOT, "include"), os.path.join(helpers.ROOT, "tests", "jni_mock", "mock_jni.c"),
                    "-ldl", "-o", str(exe)], check=True)
    mine = helpers.pkg().lib_path()
    n = 2_500_000
    data = helpers.corpus(n, first_block=4)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    libs = [("mine", mine)] + ([("ref", REF_SO)] if os.path.exists(REF_SO) else [])
    outs = {}
    for tag, lib in libs:
        d = tmp_path / tag; d.mkdir()
        for chunk in (70_000, 1_000_003):                   # below and above the stream's own buffer size
            got = _zstream(exe, lib, src, n, d, chunk)
            assert got["zstream_sizes"].split()[1:5] == ["131072", "131591", "131075", "131072"], (tag, got["zstream_sizes"])   # ZSTD_CStreamInSize .. DStreamOutSize
            assert "err(-1)=1" in got["zstream_sizes"], tag
            assert "bad=0" in got["zstream_compress"], (tag, chunk, got)
            assert "bad=0 same=1" in got["zstream_roundtrip"] and got["zstream_roundtrip"].split()[1] == str(n), (tag, chunk, got)
        outs[tag] = d / "zstream.bin"
        assert 0 < os.path.getsize(outs[tag]) < n // 2, tag
    if len(libs) == 2:
        for reader, writer in (("mine", "ref"), ("ref", "mine")):
            got = _zstream(exe, dict(libs)[reader], src, n, tmp_path / reader, 70_000, decode=outs[writer])
            assert "bad=0 same=1" in got["zstream_roundtrip"], (reader, writer, got)


def test_streaming_without_libzstd_fails_loudly(tmp_path):
    """no usable libzstd (FOURMC_LIBZSTD names a file that is not one): the constructors throw, nothing returns a null handle silently"""
    exe = tmp_path / "mock_jni"
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(helpers.ROOT, "include"), os.path.join(helpers.ROOT, "tests", "jni_mock", "mock_jni.c"),
                    "-ldl", "-o", str(exe)], check=True)
    src = tmp_path / "in.bin"; src.write_bytes(helpers.corpus(100_000).tobytes())
    r = subprocess.run([str(exe), helpers.pkg().lib_path(), str(src), "100000", str(tmp_path), "zstream", "50000"], capture_output=True, text=True,
                       env=dict(os.environ, FOURMC_LIBZSTD="/nonexistent/libzstd.so"))
    assert r.returncode == 0, r.stderr
    assert "zstream_create 0 java/lang/UnsupportedOperationException" in r.stdout, r.stdout
