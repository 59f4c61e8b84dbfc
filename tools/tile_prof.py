#!/usr/bin/env python3
"""Per-phase cycle counters of the tile LZ4 decoder (profiling build: `make -C 4mc_amd/csrc tprof`, K1T_PROF).
usage: FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-tprof.so python tools/tile_prof.py [--full N]
Per S-mix class, one block alone (and, with --full N, block 0 of every class inside a launch of N blocks): Mclk per phase
(thread 0's view).  walk: chains / threading / check;  exec: chunk (stage + bitmap) / sequences / marks + scan / pass 1 / pass 2 / flush."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if not os.environ.get("FOURMC_LIB"): p.use_research(True); p.gpu_init(0)     # debug exports: research side build
p.lib().fourmc_gpu_set_lz4_decode_path(13)
B = p.BLOCKSIZE
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
def slot_bytes():
    # lz4tile.h: (kMetaWords + kBmWords + 3) & ~3, kBmWords = (kMaxSrc + 31) / 32 + 96, kMaxSrc = 4210768 + 32
    return ((48 + (4210768 + 32 + 31) // 32 + 96 + 3) & ~3) * 4
def run(comps, caps, which, tag):
    nb = len(comps)
    offs, pos = [], 0
    for c in comps: offs.append(pos); pos += (len(c) + 15) // 16 * 16
    src = np.zeros(pos + 64, np.uint8)
    for c, o in zip(comps, offs): src[o:o + len(c)] = c
    d_src = torch.from_numpy(src).cuda(); d_dst = torch.empty(nb * B + 64, dtype=torch.uint8, device="cuda")
    blk = p.DeviceBatch(p.make_blocks(offs, [i * B for i in range(nb)], [len(c) for c in comps], caps))
    p.lz4_decompress(d_src, d_dst, blk); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    blk2 = p.DeviceBatch(p.make_blocks(offs, [i * B for i in range(nb)], [len(c) for c in comps], caps))
    s.record(); p.lz4_decompress(d_src, d_dst, blk2); e.record(); torch.cuda.synchronize()
    print(f"== {tag}: {nb} blocks, {s.elapsed_time(e):.2f} ms")
    for b, name in which:
        meta = np.zeros(48, np.uint32)
        p.binding.check(p.lib().fourmc_gpu_debug_read_workspace(meta.ctypes.data, b * slot_bytes(), meta.nbytes), "ws")
        w = meta[4:12].view(np.uint64).astype(np.float64); x = meta[12:38].view(np.uint64).astype(np.float64)
        print(f"  {name:7s} walk Mclk chains {w[0]/1e6:7.2f} thread {w[1]/1e6:6.2f} check {w[2]/1e6:6.2f} (redone {int(w[3])})"
              f" | exec chunk {x[0]/1e6:6.2f} seqs {x[1]/1e6:6.2f} marks+scan {x[2]/1e6:6.2f} pass1 {x[3]/1e6:6.2f} pass2 {x[4]/1e6:6.2f} flush {x[5]/1e6:6.2f}"
              f" sum {x[:6].sum()/1e6:7.2f} | tiles {int(x[6])} clk/tile {x[:6].sum()/max(x[6],1):.0f} | fused walk: own chain {x[8]/1e6:5.2f} wait {x[9]/1e6:5.2f} threading {x[10]/1e6:5.2f} chain {x[11]/1e6:5.2f} end {x[12]/1e6:5.2f}", flush=True)
data = helpers.corpus(12 * B)
comps = []
for b in range(12):
    r, c = helpers.orc_compress(data[b * B:(b + 1) * B].copy()); comps.append(c)
if "--noalone" not in sys.argv:
    for b in range(12):
        run([comps[b]], [B], [(0, names[b])], names[b] + " alone")
if "--full" in sys.argv:
    n = int(sys.argv[sys.argv.index("--full") + 1])
    cc = [comps[i % 12] for i in range(n)]
    run(cc, [B] * n, [(b, names[b]) for b in range(12)], f"full launch")
