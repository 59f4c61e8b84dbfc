#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REFERENCE ITSELF, run in the build container:
  - oracle/_ref/4mc_ref     (reference CLI built from /root/reference/native by oracle/Makefile)
  - oracle/_ref/libref4mc.so (reference codecs)
The fixtures are DATA only: inputs (or their generator parameters) and expected outputs.
Re-run:  make -C oracle ref && python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

B = 4 << 20
cli = helpers.ref_cli()
ref = helpers.ref()
assert cli and ref, "build the reference first: make -C oracle ref"


def run_cli(data, args):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "in"); dst = os.path.join(d, "out")
        open(src, "wb").write(bytes(data))
        r = subprocess.run([cli, "-f", *args, src, dst], capture_output=True)
        assert r.returncode == 0, r.stderr
        return open(dst, "rb").read()


small = {
    "empty": b"",
    "abc": b"abc",
    "hello10": b"hello " * 9 + b"hello",
    "thirteen_a": b"a" * 13,
    "zeros_4096": bytes(4096),
    "text_2000": bytes(helpers.corpus(B)[:2000]),
}
files = {}
for name, data in small.items():
    files[name] = {"input_hex": data.hex(),
                   "4mc_fast_hex": run_cli(data, ["-1"]).hex(),
                   "4mz_fast_hex": run_cli(data, ["-z", "-1"]).hex()}

# corpus-scale: per-block manifest (usize, csize, XXH32 of the stored payload) for every level,
# plus SHA-256 of the whole container written by the reference CLI.
n = 12 * B + 123457
data = helpers.corpus(n)
manifest = {"corpus": {"seed": helpers.CORPUS_SEED, "first_block": 0, "bytes": n,
                       "sha256": hashlib.sha256(data.tobytes()).hexdigest()}, "levels": {}}
for fmt, flag in (("4mc", []), ("4mz", ["-z"])):
    for lvl in (1, 2, 3, 4):
        img = run_cli(data, flag + [f"-{lvl}"])
        blocks, pos = [], 12
        while True:
            u, c, s = (int.from_bytes(img[pos + 4 * i: pos + 4 * i + 4], "big") for i in range(3))
            pos += 12
            if u == 0 and c == 0 and s == 0:
                break
            blocks.append([u, c, s]); pos += c
        manifest["levels"][f"{fmt}-{lvl}"] = {"file_bytes": len(img), "sha256": hashlib.sha256(img).hexdigest(),
                                              "blocks": blocks, "footer_hex": img[pos:].hex()}

# XXH32 known answers from the reference's XXH32()
rng = np.random.default_rng(11)
xx = []
for ln in (0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 100, 1000, 4099):
    d = rng.integers(0, 256, ln, dtype=np.uint8)
    for seed in (0, 1, 0x9E3779B1):
        xx.append({"hex": d.tobytes().hex(), "seed": seed, "xxh32": int(ref.XXH32(d.ctypes.data, ln, seed))})

# zstd frames written by the reference's ZSTD_compress (what a 4mz block payload is): levels 1/3/6/12 are 4mz's, the others are what
# the JNI name compressBytesDirectHC(level) may ask for (levels 1..12 run on the device)
zin = helpers.golden_zstd_inputs()
zf = {}
for name, d in zin.items():
    d = np.ascontiguousarray(d)
    zf[name] = {"input_sha256": hashlib.sha256(d.tobytes()).hexdigest(), "input_bytes": len(d), "frames": {}}
    for lvl in range(1, 13):
        out = np.zeros(len(d) + 1024, np.uint8)
        r = ref.ZSTD_compress(out.ctypes.data, len(out), d.ctypes.data, len(d), lvl)
        assert not ref.ZSTD_isError(r)
        zf[name]["frames"][str(lvl)] = out[:r].tobytes().hex()
json.dump(zf, open(os.path.join(HERE, "zstd_frames.json"), "w"))

json.dump(files, open(os.path.join(HERE, "small_files.json"), "w"), indent=0)
json.dump(manifest, open(os.path.join(HERE, "corpus_manifest.json"), "w"))
json.dump(xx, open(os.path.join(HERE, "xxh32_kat.json"), "w"))
print("golden fixtures written:", {k: len(v["blocks"]) for k, v in manifest["levels"].items()})
