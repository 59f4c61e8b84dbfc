"""4mc_amd — MI355X (gfx950) block engine for the 4mc / 4mz splittable container.

The product is the C-ABI shared library ``4mc_amd/lib/libhadoop-4mc.so`` (HIP kernels + C host
code, see ``include/fourmc_gpu.h`` / ``include/fourmc.h``) and the ``4mc_amd/bin/4mc`` CLI.  This
package is only the thin Python plumbing used by tests, ``bench.py`` and multi-GPU launches:
ctypes bindings over the C ABI, PyTorch tensors as device memory / streams, ``torch.distributed``
(RCCL) for the one collective of the path (the per-rank block-index gather).

There is no CPU fallback anywhere in this package: every codec call goes through the C ABI into
the HIP kernels, and a missing library or missing GPU raises.

The directory name starts with a digit (the reference is called 4mc), so import it with
``importlib.import_module("4mc_amd")``.
"""
from .binding import (  # noqa: F401
    BLOCK_DTYPE, BLOCKSIZE, MAGIC_4MC, MAGIC_4MZ, CODEC_LZ4_FAST, CODEC_LZ4_MC, CODEC_LZ4_HC,
    CODEC_ZSTD, BLK_BADSUM, BLK_CORRUPT, EngineError, lib, lib_path, research_lib_path, use_research, cli_path, exported_symbols,
    make_blocks, gpu_init,
)
from .engine import (  # noqa: F401
    lz4_decompress, zstd_decompress, zstd_compress, lz4_compress_fast, lz4_compress_hc, lz4_compress_mc, xxh32, encode_blocks, decode_blocks, pack_image, DeviceBatch, release_workspaces,
)
from .container import (  # noqa: F401
    frame_header, frame_footer, parse_footer, assemble_container, split_container, shard_range,
    gather_block_index,
)
