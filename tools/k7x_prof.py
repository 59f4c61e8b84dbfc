#!/usr/bin/env python3
"""Per-role cycle counters of the 4mz execute kernel (zstd_exec.inc built with -DK7X_PROF into libhadoop-4mc-zx.so):
    FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-zx.so python tools/k7x_prof.py [blocks]
prints, for the first 12 blocks of the launch, each role's lifetime and the part of it spent waiting."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if not os.environ.get("FOURMC_LIB"): p.use_research(True); p.gpu_init(0)     # debug exports: research side build
B = p.BLOCKSIZE
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
base = helpers.corpus(48 * B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, dtype=np.uint32)
enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
p.encode_blocks(d_src, d_stage, enc, codec=p.CODEC_ZSTD, level=int(os.environ.get("ZLEVEL", "1")))
e = enc.download()
d_out = torch.zeros(nb * B + 64, dtype=torch.uint8, device="cuda")
for it in range(2):
    dec = p.DeviceBatch(p.make_blocks(offs, offs, e["result"].astype(np.uint32), lens, e["xxh32"]))
    s = torch.cuda.Event(enable_timing=True); t = torch.cuda.Event(enable_timing=True)
    s.record(); p.decode_blocks(d_stage, d_out, dec, codec=p.CODEC_ZSTD); t.record(); torch.cuda.synchronize()
print(f"{nb} blocks: {s.elapsed_time(t):.2f} ms, equal: {torch.equal(d_out[: nb * B], d_src)}")
lib = p.lib()
lib.fourmc_zstd_scratch_bytes.restype = C.c_size_t; lib.fourmc_zstd_scratch_bytes.argtypes = [C.c_uint32]
slot = lib.fourmc_zstd_scratch_bytes(1)
for b in range(min(nb, 12)):
    buf = (C.c_uint64 * 8)()
    assert lib.fourmc_gpu_debug_read_workspace(buf, b * slot + slot - 192, 64) == 0
    v = [int(x) for x in buf]
    eb = (C.c_uint64 * 4)()
    assert lib.fourmc_gpu_debug_read_workspace(eb, b * slot + slot - 64, 32) == 0
    print(f"{names[b % 12]:7s} entropy kernel: literals {eb[0]/1e6:6.2f} headers {eb[1]/1e6:5.2f} sequences {eb[2]/1e6:6.2f} Mclk (cycle counter) || " + "  ".join(f"{r} {v[2*i]/1e6:7.2f} Mclk (waiting {100*v[2*i+1]/max(v[2*i],1):4.1f}%)" for i, r in enumerate(("SEQ", "LIT", "PLAN", "EXEC"))))
