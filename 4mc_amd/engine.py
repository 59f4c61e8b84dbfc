"""Device-resident calls: torch tensors are the HBM buffers, the C ABI does the work.

Mirrors the reference's per-block codec calls (LZ4_decompress_safe / LZ4_compress_default /
XXH32, native/4mc.c:301,311,637,661) in their batched form (include/fourmc_gpu.h).
"""
import numpy as np
import torch

from .binding import BLOCK_DTYPE, CODEC_LZ4_FAST, check, lib


def _stream_ptr(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    return int(stream.cuda_stream)


def _ptr(t):
    assert t.is_cuda and t.is_contiguous()
    return int(t.data_ptr())


class DeviceBatch:
    """A descriptor array resident in HBM (n x struct fourmc_block)."""

    def __init__(self, blocks_np, device="cuda"):
        assert blocks_np.dtype == BLOCK_DTYPE
        self.n = len(blocks_np)
        raw = torch.from_numpy(np.ascontiguousarray(blocks_np).view(np.uint8).reshape(-1).copy())
        self.d = raw.to(device) if self.n else torch.empty(0, dtype=torch.uint8, device=device)

    def download(self):
        return self.d.cpu().numpy().view(BLOCK_DTYPE).copy()

    @property
    def ptr(self):
        return _ptr(self.d) if self.n else 0


def lz4_decompress(d_src, d_dst, batch, stream=None):
    """result[b] = LZ4_decompress_safe(src+src_off, dst+dst_off, src_len, dst_cap)."""
    check(lib().fourmc_gpu_lz4_decompress(_ptr(d_src), _ptr(d_dst), batch.ptr, batch.n, _stream_ptr(stream)),
          "fourmc_gpu_lz4_decompress")


def lz4_compress_hc(d_src, d_dst, batch, level, stream=None):
    """result[b] = LZ4_compress_HC(src+src_off, dst+dst_off, src_len, dst_cap, level)."""
    check(lib().fourmc_gpu_lz4_compress_hc(_ptr(d_src), _ptr(d_dst), batch.ptr, batch.n, level, _stream_ptr(stream)),
          "fourmc_gpu_lz4_compress_hc")


def lz4_compress_mc(d_src, d_dst, batch, stream=None):
    """result[b] = LZ4_compressMC_limitedOutput(...) (dst_cap 0xFFFFFFFF: LZ4_compressMC, unlimited)."""
    check(lib().fourmc_gpu_lz4_compress_mc(_ptr(d_src), _ptr(d_dst), batch.ptr, batch.n, _stream_ptr(stream)),
          "fourmc_gpu_lz4_compress_mc")


def zstd_decompress(d_src, d_dst, batch, stream=None):
    """result[b] = ZSTD_decompress(dst+dst_off, dst_cap, src+src_off, src_len) (negative on error)."""
    check(lib().fourmc_gpu_zstd_decompress(_ptr(d_src), _ptr(d_dst), batch.ptr, batch.n, _stream_ptr(stream)),
          "fourmc_gpu_zstd_decompress")


def zstd_compress(d_src, d_dst, batch, level=1, stream=None):
    """result[b] = ZSTD_compress(dst+dst_off, dst_cap, src+src_off, src_len, level) (-(error number) on error)."""
    check(lib().fourmc_gpu_zstd_compress(_ptr(d_src), _ptr(d_dst), batch.ptr, batch.n, level, _stream_ptr(stream)),
          "fourmc_gpu_zstd_compress")


def lz4_compress_fast(d_src, d_dst, batch, stream=None):
    """result[b] = LZ4_compress_default(src+src_off, dst+dst_off, src_len, dst_cap)."""
    check(lib().fourmc_gpu_lz4_compress_fast(_ptr(d_src), _ptr(d_dst), batch.ptr, batch.n, _stream_ptr(stream)),
          "fourmc_gpu_lz4_compress_fast")


def xxh32(d_src, batch, seed=0, stream=None):
    """xxh32[b] = XXH32(src+src_off, src_len, seed)."""
    check(lib().fourmc_gpu_xxh32(_ptr(d_src), batch.ptr, batch.n, seed, _stream_ptr(stream)), "fourmc_gpu_xxh32")


def encode_blocks(d_src, d_dst, batch, codec=CODEC_LZ4_FAST, level=0, stream=None):
    """One iteration of the reference's compress loop per block (native/4mc.c:301-329)."""
    check(lib().fourmc_gpu_4mc_encode_blocks(_ptr(d_src), _ptr(d_dst), batch.ptr, batch.n, codec, level,
                                             _stream_ptr(stream)), "fourmc_gpu_4mc_encode_blocks")


def decode_blocks(d_src, d_dst, batch, codec=CODEC_LZ4_FAST, stream=None):
    """One iteration of the reference's decode loop per block (native/4mc.c:603-668)."""
    check(lib().fourmc_gpu_4mc_decode_blocks(_ptr(d_src), _ptr(d_dst), batch.ptr, batch.n, codec,
                                             _stream_ptr(stream)), "fourmc_gpu_4mc_decode_blocks")


def pack_image(d_staging, d_image, batch, d_image_off, stream=None):
    """Block headers + payloads -> contiguous file image (native/4mc.c:309-315); d_image_off: int64/uint64 tensor."""
    check(lib().fourmc_gpu_4mc_pack_image(_ptr(d_staging), _ptr(d_image), batch.ptr, _ptr(d_image_off), batch.n,
                                          _stream_ptr(stream)), "fourmc_gpu_4mc_pack_image")


def release_workspaces():
    """Frees the device workspaces the engine keeps per stream (fourmc_gpu_release_workspaces); the next call allocates again."""
    check(lib().fourmc_gpu_release_workspaces(), "fourmc_gpu_release_workspaces")
