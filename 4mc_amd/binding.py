"""ctypes view of the C ABI declared in include/fourmc_gpu.h and include/fourmc.h."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

BLOCKSIZE = 4 << 20                   # native/4mc.c:116
MAGIC_4MC = 0x344D4300                # native/4mc.c:111
MAGIC_4MZ = 0x344D5A00                # native/4mc.c:112
CODEC_LZ4_FAST, CODEC_LZ4_MC, CODEC_LZ4_HC, CODEC_ZSTD = 0, 1, 2, 3
BLK_BADSUM = -1000000001
BLK_CORRUPT = -1000000002

# struct fourmc_block (32 bytes)
BLOCK_DTYPE = np.dtype([("src_off", "<u8"), ("dst_off", "<u8"), ("src_len", "<u4"),
                        ("dst_cap", "<u4"), ("result", "<i4"), ("xxh32", "<u4")])
assert BLOCK_DTYPE.itemsize == 32


class EngineError(RuntimeError):
    pass


def lib_path():
    # FOURMC_LIB: load an alternative build (research / profiling variants); default = the in-tree product library
    return os.environ.get("FOURMC_LIB") or os.path.join(_HERE, "lib", "libhadoop-4mc.so")


def research_lib_path():
    # the product plus the alternative LZ4 decode designs and the debug exports (make -C 4mc_amd/csrc research)
    return os.path.join(_HERE, "lib", "libhadoop-4mc-research.so")


def cli_path():
    return os.path.join(_HERE, "bin", "4mc")


_lib = None
_product_lib = None
# declared by include/fourmc_gpu.h under FOURMC_RESEARCH only: the product library does not export them
_RESEARCH_ONLY = ("fourmc_gpu_debug_read_workspace", "fourmc_gpu_debug_zstd_exec_counts", "fourmc_gpu_debug_lz4_parse", "fourmc_debug_one_block_counters")

# every symbol include/fourmc_gpu.h and include/fourmc.h declare
_GPU_API = {
    "fourmc_gpu_device_count": (C.c_int, []),
    "fourmc_gpu_init": (C.c_int, [C.c_int]),
    "fourmc_gpu_last_error": (C.c_char_p, []),
    "fourmc_gpu_arch": (C.c_char_p, []),
    "fourmc_gpu_lz4_decompress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "fourmc_gpu_lz4_compress_fast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "fourmc_gpu_lz4_compress_hc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "fourmc_gpu_lz4_compress_mc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "fourmc_gpu_zstd_decompress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "fourmc_gpu_zstd_compress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "fourmc_gpu_debug_read_workspace": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]),
    "fourmc_gpu_set_lz4_encode_mode": (None, [C.c_int]),
    "fourmc_gpu_get_lz4_encode_mode": (C.c_int, []),
    "fourmc_gpu_set_lz4_decode_path": (None, [C.c_int]),
    "fourmc_gpu_get_lz4_decode_path": (C.c_int, []),
    "fourmc_gpu_set_zstd_decode_split": (None, [C.c_int]),
    "fourmc_gpu_get_zstd_decode_split": (C.c_int, []),
    "fourmc_gpu_debug_zstd_exec_counts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fourmc_gpu_debug_lz4_parse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fourmc_debug_one_block_counters": (None, [C.c_void_p, C.c_void_p]),
    "fourmc_gpu_one_block_stats": (None, [C.c_void_p, C.c_void_p]),
    "fourmc_gpu_release_workspaces": (C.c_int, []),
    "fourmc_gpu_xxh32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "fourmc_gpu_4mc_encode_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    "fourmc_gpu_4mc_decode_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "fourmc_gpu_4mc_pack_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "fourmc_LZ4_compressBound": (C.c_int, [C.c_int]),
    "fourmc_LZ4_compress_default": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "fourmc_LZ4_compressMC": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "fourmc_LZ4_compressMC_limitedOutput": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "fourmc_LZ4_compress_HC": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fourmc_LZ4_decompress_safe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "fourmc_ZSTD_decompress": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "fourmc_ZSTD_compress": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]),
    "fourmc_ZSTD_compressBound": (C.c_size_t, [C.c_size_t]),
    "fourmc_XXH32": (C.c_uint32, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "fourmc_host_alloc": (C.c_void_p, [C.c_size_t]),
    "fourmc_host_free": (None, [C.c_void_p]),
    "fourmc_host_4mc_encode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "fourmc_host_4mc_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_int]),
}
_FILE_API = {
    "fourMCcompressFilename": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int]),
    "fourMcDecompressFileName": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p]),
    "fourMZcompressFilename": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int]),
    "fourMZDecompressFileName": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p]),
    "fourmc_file_block_count": (C.c_int64, [C.c_char_p, C.c_void_p]),
    "fourmc_file_decode_blocks": (C.c_int64, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
    "fourmc_shard_range": (None, [C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fourmc_shard_offsets": (None, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "fourmc_shard_write": (C.c_int, [C.c_int, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fourmc_file_compress_sharded": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fourmc_frame_header": (None, [C.c_void_p, C.c_uint32]),
    "fourmc_frame_check_header": (C.c_int, [C.c_void_p, C.c_uint32]),
    "fourmc_frame_block_header": (None, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "fourmc_frame_parse_block_header": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fourmc_frame_footer": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "fourmc_frame_parse_footer": (C.c_int64, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "fourmc_index_find_next": (C.c_int64, [C.c_void_p, C.c_uint32, C.c_uint64]),
    "fourmc_index_find_block": (C.c_int64, [C.c_void_p, C.c_uint32, C.c_uint64]),
    "fourmc_index_align_start": (C.c_uint64, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "fourmc_index_align_end": (C.c_uint64, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
}


def exported_symbols():
    """Names the headers declare for the PRODUCT (C-ABI + file API); the JNI names are listed in tests."""
    return [n for n in list(_GPU_API) + list(_FILE_API) if n not in _RESEARCH_ONLY]


def lib():
    """Load libhadoop-4mc.so (built in-tree by __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise EngineError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP extension is mandatory; there is no fallback path)")
        _lib = _load(path)
    return _lib


def _load(path):
    L = C.CDLL(path)
    for name, (res, args) in {**_GPU_API, **_FILE_API}.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            if name in _RESEARCH_ONLY:
                continue
            raise
        fn.restype, fn.argtypes = res, args
    return L


def use_research(on=True):
    """Point lib() at the research side build (a superset of the product: every call keeps working) or back at the product.
    Tests of the alternative decode designs and of the debug counters switch for the duration of a module."""
    global _lib, _product_lib
    if on:
        path = research_lib_path()
        if not os.path.exists(path):
            raise EngineError(f"{path} is missing: make -C 4mc_amd/csrc research")
        if _product_lib is None:
            _product_lib = _lib
        _lib = _load(path)
    else:
        _lib = _product_lib
        _product_lib = None
    return _lib


def check(rc, what):
    if rc != 0:
        raise EngineError(f"{what} failed ({rc}): {lib().fourmc_gpu_last_error().decode()}")


def gpu_init(device=-1):
    check(lib().fourmc_gpu_init(device), "fourmc_gpu_init")
    return lib().fourmc_gpu_arch().decode()


def make_blocks(src_off, dst_off, src_len, dst_cap, xxh32=None):
    """Host-side descriptor array (numpy structured, one row per block)."""
    n = len(src_len)
    b = np.zeros(n, dtype=BLOCK_DTYPE)
    b["src_off"], b["dst_off"], b["src_len"], b["dst_cap"] = src_off, dst_off, src_len, dst_cap
    if xxh32 is not None:
        b["xxh32"] = xxh32
    return b
