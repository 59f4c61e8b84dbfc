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
    check_jni_protocol(r.stdout, data, tmp_path, level9_on_device=True)


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
