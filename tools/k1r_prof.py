#!/usr/bin/env python3
"""Per-role cycle counters of the row-parallel LZ4 decoder (profiling build: `make -C 4mc_amd/csrc rprof`, K1R_PROF).
One block per S-mix class alone on the chip, then (--full N) N blocks of S-mix in one launch (role averages).
usage: FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-rprof.so python tools/k1r_prof.py [--full N] [class ...]"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
LANES = os.environ.get("LANES") == "1"                                  # the lane-per-sequence path (two roles) instead of the row pipeline
WX = os.environ.get("WX") == "1"                                        # walk + window copier (four roles)
p.lib().fourmc_gpu_set_lz4_decode_path(10 if WX else (8 if LANES else 5))
L = C.CDLL(os.environ["FOURMC_LIB"])
L.fourmc_gpu_debug_rows_prof.argtypes = [C.c_void_p, C.c_uint32]; L.fourmc_gpu_debug_rows_prof.restype = C.c_int
B = p.BLOCKSIZE
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
ROLE = {0: ("pre ", ["wait ring", "", "", "", "", "", ""]),
        1: ("walk", ["wait pre", "wait res slot", "wait copier", "general tokens", "#general", "", ""]),
        2: ("post", ["walk wait", "row reads", "marking", "sizes", "copier room", "records+literals", "publish"]),
        3: ("copy", ["stores+rounds", "#iterations", "#match bytes", "#passes", "admit+owners", "scratch phase", "loads"])}

if LANES:
    ROLE = {0: ("walk", ["wait queue room", "#windows", "#window tokens", "#general tokens", "", "", ""]),
            1: ("exec", ["wait tokens", "#batches", "#batch tokens", "#passes", "#overlapping (whole wave)", "#general", "matches"]),
            2: ("-", [""] * 7), 3: ("-", [""] * 7)}
if WX:
    ROLE = {0: ("walk", ["wait queue room", "#windows", "#window tokens", "#general tokens", "", "", "#polls"]),
            1: ("seq+lit", ["", "", "", "", "", "", "#polls"]), 2: ("plan", ["", "", "", "", "", "", "#polls"]), 3: ("exec", ["", "", "", "", "", "", "#polls"])}

def show(t, ms, tag):
    print(f"== {tag}: {ms:.2f} ms; Mclk (counts plain)")
    for role in range(4):
        nm, sites = ROLE[role]
        v = t[role]
        parts = [f"total {v[7] / 1e6:8.2f}"]
        for i, sname in enumerate(sites):
            if sname:
                parts.append(f"{sname} {v[i]:.0f}" if sname.startswith("#") else f"{sname} {v[i] / 1e6:.2f}")
        print(f"  {nm}: " + "  ".join(parts))

args = sys.argv[1:]
full = 0
if "--full" in args:
    i = args.index("--full"); full = int(args[i + 1]); del args[i:i + 2]
want = [int(a) for a in args] or ([] if full else [0, 1, 2, 3, 5, 11])
data = helpers.corpus(12 * B)
for b in want:
    src = data[b * B:(b + 1) * B]
    r, comp = helpers.orc_compress_hc(src, 4, len(src)) if os.environ.get("HC") == "1" else helpers.orc_compress(src)
    d_src = torch.from_numpy(comp).cuda(); d_dst = torch.zeros(B + 64, dtype=torch.uint8, device="cuda")
    batch = p.DeviceBatch(p.make_blocks([0], [0], [len(comp)], [B]))
    for _ in range(2):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); p.lz4_decompress(d_src, d_dst, batch); e.record(); torch.cuda.synchronize()
    host = np.zeros(32, np.uint64)
    assert L.fourmc_gpu_debug_rows_prof(host.ctypes.data, 1) == 0
    ok = bool(torch.equal(d_dst[:B].cpu(), torch.from_numpy(src))) and int(batch.download()["result"][0]) == B
    show(host.reshape(4, 8).astype(np.float64), s.elapsed_time(e), f"{names[b]} (csize {len(comp)}, rows {len(comp) // 64}) {'ok' if ok else 'BAD'}")
if full:
    nb = full
    base = helpers.corpus(48 * B)
    src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
    S = (B + B // 255 + 16 + 63) & ~63
    offs = np.arange(nb, dtype=np.uint64) * B; soffs = np.arange(nb, dtype=np.uint64) * S; lens = np.full(nb, B, np.uint32)
    enc = p.DeviceBatch(p.make_blocks(offs, soffs, lens, np.full(nb, S, np.uint32)))
    stage = torch.empty(nb * S, dtype=torch.uint8, device="cuda")
    p.lz4_compress_fast(src, stage, enc)
    r = enc.download()
    dec = p.DeviceBatch(p.make_blocks(soffs, offs, r["result"].astype(np.uint32), lens))
    out = torch.zeros(nb * B + 64, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); p.lz4_decompress(stage, out, dec); e.record(); torch.cuda.synchronize()
    host = np.zeros(nb * 32, np.uint64)
    assert L.fourmc_gpu_debug_rows_prof(host.ctypes.data, nb) == 0
    t = host.reshape(nb, 4, 8).astype(np.float64)
    res = dec.download()["result"]
    print(f"blocks handed back: {int((res != B).sum())} of {nb}")
    show(t.mean(axis=0), s.elapsed_time(e), f"S-mix x{nb}: mean over blocks")
    for cls in (0, 1, 3, 5):
        show(t[cls::48].mean(axis=0), s.elapsed_time(e), f"  class {names[cls % 12]} inside the full launch")
