#!/usr/bin/env python3
"""Per-phase cycle counters of the group executor of the LZ4 decode, lz4_ring.hip (profiling build: `make -C 4mc_amd/csrc gprof`, K1G_PROF).
usage: FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-gprof.so python tools/ring_prof.py [--full N]
Per S-mix class, one block alone (and, with --full N, block 0 of every class inside a launch of N blocks): Mclk per phase.
exec (thread 0's clock): records+scan+barrier / literals / rounds / flush; steps, rounds, escapes."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if not os.environ.get("FOURMC_LIB"): p.use_research(True); p.gpu_init(0)     # debug exports: research side build
p.lib().fourmc_gpu_set_lz4_decode_path(15)
B = p.BLOCKSIZE
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
WS = None
def ws_words():
    import re
    h = open(os.path.join(ROOT, "4mc_amd", "csrc", "lz4seg.h")).read()
    return 5 * 1024 * 1024   # read generously; slot size comes from the header below
def slot_bytes():
    # lz4seg.h: (kMetaWords + kRecWords * (kSegs * (kFixCap + 8) + kMaxSrc / 3 + 512) + 3) & ~3, kMaxSrc = 4210768 + 32
    return ((320 + 2 * (64 * 136 + (4210768 + 32) // 3 + 512) + 3) & ~3) * 4
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
        meta = np.zeros(320, np.uint32)
        p.binding.check(p.lib().fourmc_gpu_debug_read_workspace(meta.ctypes.data, b * slot_bytes(), meta.nbytes), "ws")
        t = meta[272:320].view(np.uint64).astype(np.float64)
        w, x = t[:12], t[12:]
        print(f"  {name:7s} walk Mclk {(w[0]+w[1]+w[2])/1e6:7.2f} | exec scan {x[0]/1e6:7.2f} lits {x[1]/1e6:7.2f} rounds {x[2]/1e6:7.2f} flush {x[3]/1e6:7.2f}"
              f" | steps {int(x[6])} rounds {int(x[7])} ({x[7]/max(x[6],1):.2f} per step) escapes {int(x[9])} | clk per step {(x[0]+x[1]+x[2]+x[3])/max(x[6],1):.0f}", flush=True)
data = helpers.corpus(12 * B)
comps = []
for b in range(12):
    r, c = helpers.orc_compress(data[b * B:(b + 1) * B].copy()); comps.append(c)
for b in range(12):
    run([comps[b]], [B], [(0, names[b])], names[b] + " alone")
if "--full" in sys.argv:
    n = int(sys.argv[sys.argv.index("--full") + 1])
    cc = [comps[i % 12] for i in range(n)]
    run(cc, [B] * n, [(b, names[b]) for b in range(12)], f"full launch")
