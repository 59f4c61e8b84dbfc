#!/usr/bin/env python3
"""Times the zstd level-1 encode kernel and the zstd decode kernel per S-mix block class (1 block per
launch) and on the bench batch (FOURMC_BENCH_BLOCKS replicas of the 48-block S-mix)."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if not os.environ.get("FOURMC_LIB"): p.use_research(True); p.gpu_init(0)     # debug exports: research side build
B = p.BLOCKSIZE
LEVEL = int(os.environ.get("ZLEVEL", "1"))
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
def t(fn):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); s.record(); fn(); e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)
ENC_CTR = 3 * 131328 + 3 * 32832 + 64 + 131072 + 64                                    # zstd_encode.hip: kOffLit + kLitPad + kSub + 64
ENC_WORK = 0     
DEC_WORK, DEC_CTR = 131072 + 64, 131072                                                 # zstd_decode.hip
def counters(work, off, names):
    import ctypes as C
    buf = (C.c_uint64 * 4)()
    assert p.lib().fourmc_gpu_debug_read_workspace(buf, off, 32) == 0
    tot = sum(buf) or 1
    return " ".join(f"{n} {100 * v / tot:4.1f}%" for n, v in zip(names, buf))
def run(src, nb, tag):
    offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, np.uint32)
    enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
    stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
    te = t(lambda: p.encode_blocks(src, stage, enc, codec=p.CODEC_ZSTD, level=LEVEL))
    r = enc.download()
    dec = p.DeviceBatch(p.make_blocks(offs, offs, r["result"].astype(np.uint32), lens, r["xxh32"]))
    out = torch.empty(nb * B + 64, dtype=torch.uint8, device="cuda")
    ec = counters(ENC_WORK, ENC_CTR, ("match", "literals", "sequences", "frame")) if nb == 1 else ""
    dn = ""
    if "zprof" in os.environ.get("FOURMC_LIB", ""):                                     # (of block 0 when there are several)                         # make -C 4mc_amd/csrc zprof: dense-window counters of the level-1 finder
        import ctypes as C
        buf = (C.c_uint64 * 17)()
        assert p.lib().fourmc_gpu_debug_read_workspace(buf, ENC_CTR + 32, 136) == 0
        k = ("windows", "rep-seq", "hash-seq", "batched-seq", "dirty-cut", "end:go-on", "end:fresh", "end:batched", "end:rep2-loop", "rep-reads", "fast-steps")
        dn = "             dense: " + " ".join(f"{n} {int(v)}" for n, v in zip(k, buf))
        ph = [int(v) for v in buf[11:17]]; tot = sum(ph) or 1
        dn += "\n" + ("             dense cycles: " + " ".join(f"{n} {100 * v / tot:4.1f}%" for n, v in zip(("input", "table+slots", "candidates", "walk", "store+commit", "outside(batched,entropy)"), ph)) + f"  per window {tot / max(int(buf[0]), 1):.0f} clk")
    td = t(lambda: p.decode_blocks(stage, out, dec, codec=p.CODEC_ZSTD))
    import ctypes as C
    p.lib().fourmc_zstd_dec_counter_offset.restype = C.c_size_t
    dc = counters(0, p.lib().fourmc_zstd_dec_counter_offset(), ("literals", "headers", "sequences", "execute")) if (nb == 1 or os.environ.get("FOURMC_ZDECODE") == "single") else ""
    ok = torch.equal(out[: nb * B], src[: nb * B])
    cs = int(r["result"].astype(np.int64).sum())
    print(f"{tag:12s} blocks {nb:5d} ratio {nb * B / cs:6.3f}  enc {te:9.2f} ms ({nb * B / te / 1e6:7.2f} GB/s)  dec {td:9.2f} ms ({nb * B / td / 1e6:7.2f} GB/s) roundtrip {'ok' if ok else 'BAD'}", flush=True)
    if nb == 1: print(f"             enc phases: {ec}\n             dec phases: {dc}", flush=True)
    elif dc: print(f"             dec phases of block 0 (one-wave kernel): {dc}", flush=True)
    if dn: print(dn, flush=True)
if "--classes" in sys.argv:
    data = helpers.corpus(12 * B)
    for b in range(12):
        run(torch.from_numpy(data[b * B:(b + 1) * B].copy()).cuda(), 1, names[b])
nb = int(os.environ.get("FOURMC_BENCH_BLOCKS", "2048"))
base = helpers.corpus(48 * B)
src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
run(src, nb, "S-mix")
