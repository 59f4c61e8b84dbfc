"""CPU: the product's host logic (framing, footer index, exports, CLI plumbing) without a GPU."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np

import helpers

ROOT = helpers.ROOT
G = os.path.join(ROOT, "tests", "golden")

JNI = [f"Java_com_fing_compression_fourmc_{c}_{m}" for c, ms in {
    "Lz4Compressor": ["initIDs", "compressBytesDirect", "compressBytesDirectMC", "compressBytesDirectHC", "compressBound", "xxhash32"],
    "ZstdCompressor": ["initIDs", "compressBytesDirect", "compressBytesDirectMC", "compressBytesDirectHC", "compressBound", "xxhash32"],
    "Lz4Decompressor": ["initIDs", "decompressBytesDirect", "xxhash32"],
    "ZstdDecompressor": ["initIDs", "decompressBytesDirect", "xxhash32"],
    "zstd_Zstd": ["isError", "getErrorName", "cStreamInSize", "cStreamOutSize", "dStreamInSize", "dStreamOutSize"],
    "zstd_ZstdStreamCompressor": ["initIDs", "createCStream", "freeCStream", "initCStream", "compressStream", "endStream"],
    "zstd_ZstdStreamDecompressor": ["initIDs", "createDStream", "freeDStream", "initDStream", "decompressStream"],
}.items() for m in ms]


def test_library_loads_and_exports_every_declared_symbol():
    p = helpers.pkg()
    L = p.lib()
    for name in p.exported_symbols():
        assert getattr(L, name) is not None, name
    assert len(JNI) == 35
    raw = C.CDLL(p.lib_path())
    for name in JNI:
        assert getattr(raw, name) is not None, name
    # every prototype in the public headers is exported; the block under FOURMC_RESEARCH (debug / profiling exports) by the research
    # side build ONLY - the drop-in library's dynamic table carries the boundary and nothing else
    pat = r"\b(four[mM][cC][A-Za-z0-9_]*|fourM[cZ][A-Za-z]+)\s*\("
    research = C.CDLL(p.research_lib_path())
    for hdr in ("fourmc_gpu.h", "fourmc.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        rs = re.search(r"#ifdef FOURMC_RESEARCH(.*?)#endif", text, re.S)
        rnames = set(re.findall(pat, rs.group(1))) if rs else set()
        for name in set(re.findall(pat, text)):
            assert getattr(research, name) is not None, (hdr, name)
            if name in rnames:
                assert not hasattr(raw, name), (hdr, name, "a debug export in the product library")
            else:
                assert getattr(raw, name) is not None, (hdr, name)
    import subprocess as sp
    dyn = sp.run(["nm", "-D", "--defined-only", p.lib_path()], capture_output=True, text=True).stdout
    assert "debug" not in dyn and "fourmc_launch" not in dyn, "the product library exports debug symbols or internal launchers"


def test_no_gpu_calls_fail_loudly_not_silently():
    """In this container there is no device: the engine must refuse, never fall back."""
    import torch
    if torch.cuda.is_available():
        return
    p = helpers.pkg()
    L = p.lib()
    assert L.fourmc_gpu_init(0) < 0 and L.fourmc_gpu_last_error()
    src = np.zeros(100, np.uint8); dst = np.zeros(200, np.uint8)
    assert L.fourmc_LZ4_compress_default(src.ctypes.data, dst.ctypes.data, 100, 200) == 0
    assert L.fourmc_LZ4_decompress_safe(src.ctypes.data, dst.ctypes.data, 100, 200) < 0
    r = subprocess.run([p.cli_path(), "-f", __file__, "/tmp/_x.4mc"], capture_output=True)
    assert r.returncode == 1 and b"GPU engine error" in r.stderr


def test_product_does_not_touch_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "4mc_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                text = open(os.path.join(d, f), errors="ignore").read()
                assert "liboracle" not in text and "orc_" not in text and "_ref/" not in text, os.path.join(d, f)


def test_framing_bytes_match_golden():
    p = helpers.pkg()
    files = json.load(open(os.path.join(G, "small_files.json")))
    assert p.frame_header(p.MAGIC_4MC).hex() == "344d430000000001a4b73443"
    assert p.frame_header(p.MAGIC_4MZ).hex() == "344d5a0000000001289a1c9a"
    # empty file = header + end mark + N=0 footer
    assert (p.frame_header(p.MAGIC_4MC) + b"\0" * 12 + p.frame_footer(p.MAGIC_4MC, [])).hex() == files["empty"]["4mc_fast_hex"]
    assert (p.frame_header(p.MAGIC_4MZ) + b"\0" * 12 + p.frame_footer(p.MAGIC_4MZ, [])).hex() == files["empty"]["4mz_fast_hex"]
    # re-assemble every golden file from its own parsed pieces (writer == reader inverse)
    for name, f in files.items():
        for key, magic in (("4mc_fast_hex", p.MAGIC_4MC), ("4mz_fast_hex", p.MAGIC_4MZ)):
            img = np.frombuffer(bytes.fromhex(f[key]), dtype=np.uint8)
            blocks, used = p.split_container(img, magic)
            assert used == len(img)
            pay = [img[int(b["src_off"]): int(b["src_off"]) + int(b["src_len"])] for b in blocks]
            again = p.assemble_container(magic, blocks["dst_cap"], blocks["src_len"], blocks["xxh32"], pay)
            assert again == img.tobytes(), (name, key)
            # host XXH32 equals the stored payload checksums
            L = p.lib()
            for b, q in zip(blocks, pay):
                q = np.ascontiguousarray(q)
                assert L.fourmc_XXH32(q.ctypes.data, len(q), 0) == int(b["xxh32"])


def test_footer_of_reference_files_and_manifest():
    p = helpers.pkg()
    m = json.load(open(os.path.join(G, "corpus_manifest.json")))
    for key, lvl in m["levels"].items():
        magic = p.MAGIC_4MC if key.startswith("4mc") else p.MAGIC_4MZ
        foot = bytes.fromhex(lvl["footer_hex"])
        offs = p.parse_footer(foot, magic)
        csizes = [c for _, c, _ in lvl["blocks"]]
        assert list(offs) == list(p.container.block_offsets(csizes)), key
        assert p.frame_footer(magic, offs) == foot, key
        bad = bytearray(foot); bad[9] ^= 1
        try:
            p.parse_footer(bytes(bad), magic); assert False
        except ValueError:
            pass


def test_block_index_queries_like_reference_unit_test():
    """Same cases as TestFourMcBlockIndex.java:41-84 (index {100,200,300,400})."""
    p = helpers.pkg(); L = p.lib()
    off = np.array([100, 200, 300, 400], dtype=np.uint64); a = off.ctypes.data
    assert [L.fourmc_index_find_next(a, 4, x) for x in (0, 100, 101, 400, 401)] == [100, 100, 200, 400, -1]
    assert [L.fourmc_index_find_block(a, 4, x) for x in (99, 100, 150, 399, 400, 5000)] == [-1, 0, 0, 2, 3, 3]
    NF = 2 ** 64 - 1
    assert L.fourmc_index_align_start(a, 4, 0, 1000) == 0
    assert L.fourmc_index_align_start(a, 4, 150, 250) == 200
    assert L.fourmc_index_align_start(a, 4, 150, 200) == NF
    assert L.fourmc_index_align_start(a, 4, 401, 1000) == NF
    assert L.fourmc_index_align_end(a, 4, 250, 1000) == 300
    assert L.fourmc_index_align_end(a, 4, 450, 1000) == 1000


def test_cli_flags_without_gpu():
    p = helpers.pkg()
    r = subprocess.run([p.cli_path(), "-V"], capture_output=True)
    assert r.returncode == 0 and b"4mc CLI 64-bits" in r.stderr
    r = subprocess.run([p.cli_path(), "-h"], capture_output=True)
    assert r.returncode == 0 and b"-z     : zstd compression" in r.stderr
    r = subprocess.run([p.cli_path(), "-Q", "x"], capture_output=True)
    assert r.returncode == 1 and b"Incorrect command line arguments" in r.stderr
    r = subprocess.run([p.cli_path(), "-f", "/nonexistent/file", "/tmp/_y.4mc"], capture_output=True)
    assert r.returncode == 2 and b"Cannot open input file" in r.stderr
