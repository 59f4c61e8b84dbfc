"""Test-side access to the checkers: oracle/liboracle.so (this repo's CPU restatement),
oracle/_ref/libref4mc.so (the reference's own sources, when built) and tools/libcorpus.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
B = 4 << 20
CORPUS_SEED = 0x4D43

_BLOCK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int)


def pkg():
    return importlib.import_module("4mc_amd")


def _build_if_missing(path, cmd, cwd):
    srcs = [os.path.join(cwd, f) for f in os.listdir(cwd) if f.endswith((".c", ".cpp", ".h"))]
    if not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in srcs):
        subprocess.run(cmd, cwd=cwd, check=True, stdout=subprocess.DEVNULL)
    return path


_cache = {}


def corpus_lib():
    if "corpus" not in _cache:
        p = _build_if_missing(os.path.join(ROOT, "tools", "libcorpus.so"),
                              ["gcc", "-O2", "-fPIC", "-shared", "corpus.c", "-o", "libcorpus.so"],
                              os.path.join(ROOT, "tools"))
        L = C.CDLL(p)
        for f in (L.corpus_fill, L.corpus_fill_logs):
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64]
            f.restype = None
        _cache["corpus"] = L
    return _cache["corpus"]


def corpus(nbytes, first_block=0, seed=CORPUS_SEED, logs=False):
    buf = np.empty(nbytes, dtype=np.uint8)
    (corpus_lib().corpus_fill_logs if logs else corpus_lib().corpus_fill)(buf.ctypes.data, nbytes, seed, first_block)
    return buf


def oracle():
    if "oracle" not in _cache:
        p = _build_if_missing(os.path.join(ROOT, "oracle", "liboracle.so"), ["make", "port"], os.path.join(ROOT, "oracle"))
        L = C.CDLL(p)
        L.orc_xxh32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]; L.orc_xxh32.restype = C.c_uint32
        L.orc_lz4_compress_bound.argtypes = [C.c_int]; L.orc_lz4_compress_bound.restype = C.c_int
        for f in (L.orc_lz4_compress_fast, L.orc_lz4_decompress_safe):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]; f.restype = C.c_int
        L.orc_container_bound.argtypes = [C.c_size_t]; L.orc_container_bound.restype = C.c_size_t
        L.orc_container_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_container_compress.restype = C.c_int64
        L.orc_container_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_container_decompress.restype = C.c_int64
        L.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]; L.orc_zstd_decompress.restype = C.c_int64
        _cache["oracle"] = L
    return _cache["oracle"]


def ref():
    """The reference's own codec sources compiled by oracle/Makefile; None when not built."""
    if "ref" not in _cache:
        p = os.path.join(ROOT, "oracle", "_ref", "libref4mc.so")
        L = None
        if os.path.exists(p):
            L = C.CDLL(p)
            for f in (L.LZ4_compress_default, L.LZ4_decompress_safe):
                f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]; f.restype = C.c_int
            L.LZ4_compress_HC.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]; L.LZ4_compress_HC.restype = C.c_int
            L.XXH32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]; L.XXH32.restype = C.c_uint32
            L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]; L.ZSTD_compress.restype = C.c_size_t
            L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]; L.ZSTD_decompress.restype = C.c_size_t
            L.ZSTD_isError.argtypes = [C.c_size_t]; L.ZSTD_isError.restype = C.c_uint
        _cache["ref"] = L
    return _cache["ref"]


def ref_cli():
    p = os.path.join(ROOT, "oracle", "_ref", "4mc_ref")
    return p if os.path.exists(p) else None


# ---- oracle conveniences (numpy in / numpy out) ----------------------------------------------
def orc_compress(src, cap=None):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    bound = oracle().orc_lz4_compress_bound(len(src))
    cap = bound if cap is None else cap
    dst = np.empty(max(cap, 1) + 8, dtype=np.uint8)
    r = oracle().orc_lz4_compress_fast(src.ctypes.data, dst.ctypes.data, len(src), cap)
    return r, dst[:max(r, 0)].copy()


def orc_decompress(comp, cap):
    comp = np.ascontiguousarray(comp, dtype=np.uint8)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    r = oracle().orc_lz4_decompress_safe(comp.ctypes.data, dst.ctypes.data, len(comp), cap)
    return r, dst[:max(r, 0)].copy()


def orc_zstd_decompress(frame, cap):
    frame = np.ascontiguousarray(np.frombuffer(bytes(frame), dtype=np.uint8))
    dst = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
    r = oracle().orc_zstd_decompress(frame.ctypes.data, len(frame), dst.ctypes.data, cap)
    return r, dst[:max(r, 0)].copy()


def orc_xxh32(data, seed=0):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return oracle().orc_xxh32(data.ctypes.data, len(data), seed)


def orc_container(src, magic=0x344D4300):
    """Whole .4mc image of `src` by the oracle (LZ4 fast)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    L = oracle()
    cap = L.orc_container_bound(len(src))
    dst = np.empty(cap, dtype=np.uint8)
    fn = C.cast(L.orc_codec_lz4_fast, C.c_void_p)
    n = L.orc_container_compress(src.ctypes.data, len(src), dst.ctypes.data, cap, magic, fn, None)
    assert n > 0
    return dst[:n].copy()


def lz4p_model():
    """tools/model/lz4p_model.c: the executable statement of the ratio-tolerance LZ4 encoder's rules (test infrastructure, like
    oracle/: the product never links it)."""
    if "lz4p" not in _cache:
        d = os.path.join(ROOT, "tools", "model")
        so = os.path.join(d, "liblz4p_model.so")
        src = os.path.join(d, "lz4p_model.c")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-DLZ4P_NO_MAIN", "-o", so, src], check=True)
        L = C.CDLL(so)
        L.lz4p_model_encode.restype = C.c_int
        L.lz4p_model_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        _cache["lz4p"] = L
    return _cache["lz4p"]


def lz4p_model_encode(src, cap):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.zeros(cap + 8, dtype=np.uint8)
    r = lz4p_model().lz4p_model_encode(src.ctypes.data if len(src) else None, len(src), out.ctypes.data, cap)
    return r, out[:max(r, 0)].copy()


def orc_container_decode(img, cap, magic=0x344D4300):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    L = oracle()
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    used = C.c_size_t(0)
    fn = C.cast(L.orc_codec_lz4_decode, C.c_void_p)
    n = L.orc_container_decompress(img.ctypes.data, len(img), dst.ctypes.data, cap, magic, fn, None, C.byref(used))
    return n, dst[:max(n, 0)].copy(), used.value


def edge_inputs():
    """Small inputs that walk the encoder/decoder corner cases (names -> uint8 arrays)."""
    rng = np.random.default_rng(7)
    text = corpus(B)[: 300000]
    out = {
        "empty": np.zeros(0, np.uint8),
        "one": np.array([65], np.uint8),
        "abc": np.frombuffer(b"abc", np.uint8),
        "twelve": np.frombuffer(b"abcabcabcabc", np.uint8),
        "thirteen": np.frombuffer(b"aaaaaaaaaaaaa", np.uint8),
        "hello10": np.frombuffer(b"hello " * 9 + b"hello", np.uint8),
        "zeros_64k": np.zeros(65536, np.uint8),
        "zeros_64k_limit-1": np.zeros(65546, np.uint8),     # last byU16 size (lz4.c:689)
        "zeros_64k_limit": np.zeros(65547, np.uint8),       # first byU32 size
        "zeros_1m": np.zeros(1 << 20, np.uint8),
        "period3": np.tile(np.frombuffer(b"xyz", np.uint8), 50000),
        "period7": np.tile(np.frombuffer(b"abcdefg", np.uint8), 30000),
        "period37": np.tile(rng.integers(0, 256, 37, dtype=np.uint8), 5000),
        "period200": np.tile(rng.integers(0, 256, 200, dtype=np.uint8), 2000),
        "random_100k": rng.integers(0, 256, 100000, dtype=np.uint8),
        "random_small": rng.integers(0, 256, 40000, dtype=np.uint8),
        "text_300k": text.copy(),
        "text_60k": text[:60000].copy(),
        "lit_then_run": np.concatenate([rng.integers(0, 256, 5000, dtype=np.uint8), np.zeros(90000, np.uint8),
                                        rng.integers(0, 256, 300, dtype=np.uint8)]),
        "two_symbols": rng.integers(0, 2, 200000, dtype=np.uint8),
        "far_repeat": np.concatenate([text[:70000], rng.integers(0, 256, 70000, dtype=np.uint8), text[:70000]]),
    }
    return out


def orc_compress_hc(src, level, cap=None):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    L = oracle()
    L.orc_lz4hc_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]; L.orc_lz4hc_compress.restype = C.c_int
    bound = L.orc_lz4_compress_bound(len(src))
    cap = bound if cap is None else cap
    dst = np.empty(max(cap, 1) + 8, dtype=np.uint8)
    r = L.orc_lz4hc_compress(src.ctypes.data, dst.ctypes.data, len(src), cap, level)
    return r, dst[:max(r, 0)].copy()


def orc_compress_mc(src, cap=-1):
    """cap < 0: LZ4_compressMC (no limit); else LZ4_compressMC_limitedOutput."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    L = oracle()
    L.orc_lz4mc_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]; L.orc_lz4mc_compress.restype = C.c_int
    dst = np.empty(L.orc_lz4_compress_bound(len(src)) + 64, dtype=np.uint8)
    r = L.orc_lz4mc_compress(src.ctypes.data, dst.ctypes.data, len(src), cap)
    return r, dst[:max(r, 0)].copy()


def zstd_bound(n):
    """ZSTD_compressBound (zstd.h:206 ZSTD_COMPRESSBOUND)."""
    return n + (n >> 8) + (((128 << 10) - n) >> 11 if n < (128 << 10) else 0)


def orc_zstd_compress(src, level=1, cap=None):
    """orc_zstd_compress -> (result, frame bytes); result < 0 is -(ZSTD error number)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    L = oracle()
    L.orc_zstd_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]; L.orc_zstd_compress.restype = C.c_int64
    cap = zstd_bound(len(src)) if cap is None else cap
    dst = np.empty(max(cap, 1) + 64, dtype=np.uint8)
    r = L.orc_zstd_compress(src.ctypes.data, len(src), dst.ctypes.data, cap, level)
    return r, dst[:max(r, 0)].copy()


def golden_zstd_inputs():
    """The inputs of tests/golden/zstd_frames.json (frames written by the reference's ZSTD_compress at levels 1..12);
    tests/golden/make_golden.py generates the fixture from exactly these."""
    ed = edge_inputs()
    return {"text_30k": ed["text_60k"][:30000], "period37_20k": ed["period37"][:20000], "lit_then_run_30k": ed["lit_then_run"][:30000],
            "two_symbols_30k": ed["two_symbols"][:30000], "zeros_20k": np.zeros(20000, np.uint8), "random_3k": ed["random_small"][:3000],
            "hello10": ed["hello10"], "one": ed["one"],
            # level 12 at 16 KiB and less is the optimal parser (btopt): predefined prices up to 1024 bytes, statistics above
            "text_700": ed["text_60k"][:700], "text_5k": ed["text_60k"][1000:6000], "text_16k": ed["text_60k"][:16384],
            "period37_9k": ed["period37"][:9000], "two_symbols_12k": ed["two_symbols"][:12000], "lit_then_run_16k": ed["lit_then_run"][:16384]}
