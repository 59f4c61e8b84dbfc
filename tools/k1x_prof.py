#!/usr/bin/env python3
"""Per-wave cycle counters of the LZ4 executor kernel (profiling build: `make -C 4mc_amd/csrc prof`, K1X_PROF):
one block per S-mix class alone on the chip.  usage: FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-prof.so python tools/k1x_prof.py [class ...]"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if not os.environ.get("FOURMC_LIB"): p.use_research(True); p.gpu_init(0)     # debug exports: research side build
if os.environ.get("FOURMC_DECODE") is None and "k1x" in __file__: p.lib().fourmc_gpu_set_lz4_decode_path(1)
B = p.BLOCKSIZE
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
layout = (C.c_size_t * 3)()
p.lib().fourmc_gpu_debug_lz4_parse(0, 0, 0, 0, 0, 0, 0, layout)
slot = layout[0]; dbg_off = slot - 1024
NW = 6
want = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 5, 10]
data = helpers.corpus(12 * B)
for b in want:
    src = data[b * B:(b + 1) * B]
    r, comp = helpers.orc_compress(src, B - 1)
    if r <= 0:
        continue
    d_src = torch.from_numpy(comp).cuda(); d_dst = torch.zeros(B + 64, dtype=torch.uint8, device="cuda")
    batch = p.DeviceBatch(p.make_blocks([0], [0], [len(comp)], [B]))
    for _ in range(2):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); p.lz4_decompress(d_src, d_dst, batch); e.record(); torch.cuda.synchronize()
    hh = np.zeros(64, np.uint8)
    p.binding.check(p.lib().fourmc_gpu_debug_read_workspace(hh.ctypes.data, 0, 64), "read")
    pd = hh.view(np.uint32)[5:15].astype(np.float64)
    names_p = ["stage", "A next", "B exits", "C entries", "D1 count", "scan", "D2 records", "loop"]
    print("  parse (Mclk): " + "  ".join(f"{nm} {pd[i] * 256 / 1e6:.2f}" for i, nm in enumerate(names_p)) + f"   repair rounds {int(pd[8])}  pieces {int(pd[9])}")
    host = np.zeros(1024, np.uint8)
    p.binding.check(p.lib().fourmc_gpu_debug_read_workspace(host.ctypes.data, dbg_off, 1024), "read")
    t = host.view(np.uint64).reshape(-1, 8)
    ok = bool(torch.equal(d_dst[:B].cpu(), torch.from_numpy(src)))
    print(f"== {names[b]}: {s.elapsed_time(e):.2f} ms, roundtrip {'ok' if ok else 'BAD'}; counters in Mclk")
    v = t[0] / 1e6
    print(f"  literal: gate {v[0]:7.2f}  decode {v[1]:7.2f}  wait-slot {v[2]:7.2f}  literals {v[3]:7.2f}  far matches {v[4]:7.2f}  publish {v[5]:7.2f}")
    v = t[1] / 1e6
    print(f"  chain:   wait {v[0]:7.2f}  copy {v[1]:7.2f}  publish {v[2]:7.2f}   rounds {int(t[1][3])}  entries {int(t[1][4])}  slots {int(t[1][5])}")
    v = t[2] / 1e6
    print(f"  flush:   wait {v[0]:7.2f}  work {v[1]:7.2f}")
