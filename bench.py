#!/usr/bin/env python3
"""bench.py — 4mc-Fast (LZ4 fast) compress + decompress of the silesia-like S-mix corpus replicated
to 8 GiB per GPU in 4 MiB blocks, HBM resident (BASELINE.json configs[1]).

One step = one pass of the hot path over the whole per-GPU batch:
  compress  : LZ4 encode (cap n-1, stored fallback) -> XXH32(payload) -> prefix sum of sizes
              (all_gather over RCCL when world > 1) -> pack into a contiguous .4mc image in HBM
  decompress: XXH32 verify of every payload IN PLACE in the image -> LZ4 decode / stored copy
value = uncompressed bytes of all ranks / wall time of a step (compress + decompress), GB/s
(10^9 B/s); compress and decompress rates are reported next to it.  Weak scaling: every rank holds
its own 8 GiB shard of blocks (block ranges are independent; the only collective is the gather of
per-block compressed sizes for the footer index).

Run:  python bench.py [--gpus N --steps K --warmup W]   (N>1 via torch.distributed.run)
"""
import argparse
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(helpers, base, nblk_sample, budget_s=12.0):
    """The reference's own code (oracle/_ref, kind 'reference') or, if it is not built, the oracle
    port, timed on the host cores on a bounded sample of the same corpus: per block
    LZ4_compress_default(cap n-1) + XXH32 + LZ4_decompress_safe, one block per thread."""
    import ctypes as C
    B = helpers.B
    ref = helpers.ref()
    kind = "reference" if ref is not None else "port"
    if ref is not None:
        comp, dec, xxh = ref.LZ4_compress_default, ref.LZ4_decompress_safe, ref.XXH32
    else:
        o = helpers.oracle()
        comp, dec, xxh = o.orc_lz4_compress_fast, o.orc_lz4_decompress_safe, o.orc_xxh32
    cores = max(1, min(os.cpu_count() or 1, nblk_sample))
    done = {"c": 0.0, "d": 0.0, "bytes": 0}
    lock = threading.Lock()

    def work(blocks):
        out = np.empty(B + 64, np.uint8); back = np.empty(B, np.uint8)
        tc = td = 0.0; nb = 0
        for b in blocks:
            src = base[b * B:(b + 1) * B]
            t0 = time.perf_counter()
            r = comp(src.ctypes.data, out.ctypes.data, B, B - 1)
            if r > 0:
                xxh(out.ctypes.data, r, 0)
            else:
                xxh(src.ctypes.data, B, 0)
            t1 = time.perf_counter()
            if r > 0:
                xxh(out.ctypes.data, r, 0)
                dec(out.ctypes.data, back.ctypes.data, r, B)
            else:
                xxh(src.ctypes.data, B, 0)
                back[:] = src
            t2 = time.perf_counter()
            tc += t1 - t0; td += t2 - t1; nb += B
        with lock:
            done["c"] += tc; done["d"] += td; done["bytes"] += nb

    passes = 0
    t_start = time.perf_counter()
    while True:
        parts = [list(range(i, nblk_sample, cores)) for i in range(cores)]
        th = [threading.Thread(target=work, args=(p,)) for p in parts]
        [t.start() for t in th]; [t.join() for t in th]
        passes += 1
        if time.perf_counter() - t_start > budget_s or passes >= 8:
            break
    wall = time.perf_counter() - t_start
    return {
        "value": round(done["bytes"] / wall / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": kind,
        "sample": f"{passes} pass(es) over {nblk_sample} S-mix blocks ({done['bytes'] >> 20} MiB), "
                  f"compress+xxh32+decompress per block, {cores} threads",
        "compress_GBps_per_core": round(done["bytes"] / done["c"] / 1e9, 4),
        "decompress_GBps_per_core": round(done["bytes"] / done["d"] / 1e9, 4),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("FOURMC_BENCH_BLOCKS", 2048)),
                    help="4 MiB blocks per GPU (2048 = 8 GiB, BASELINE configs[1])")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configs (4mz Fast/Medium, 4mc High) timed after the headline")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import helpers
    p = importlib.import_module("4mc_amd")

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    arch = p.gpu_init(local)
    dev = torch.device("cuda", local)
    B = p.BLOCKSIZE
    nb = args.blocks
    base_blocks = 48                                   # 4 cycles of the 12-class S-mix, ~201 MB (silesia: 212 MB)

    # ---- corpus: S-mix generated once, replicated in HBM to nb blocks (physically distinct copies)
    base = helpers.corpus(base_blocks * B, first_block=0)
    d_base = torch.from_numpy(base).to(dev)
    reps = -(-nb // base_blocks)
    d_src = d_base.repeat(reps)[: nb * B].contiguous()
    del d_base
    lens = np.full(nb, B, dtype=np.uint32)
    offs = np.arange(nb, dtype=np.uint64) * B
    enc_blocks = p.make_blocks(offs, offs, lens, lens)
    enc = p.DeviceBatch(enc_blocks, dev)
    d_stage = torch.empty(nb * B, dtype=torch.uint8, device=dev)
    d_image = torch.empty(nb * (B + 12) + 4096, dtype=torch.uint8, device=dev)
    d_out = torch.empty(nb * B + 64, dtype=torch.uint8, device=dev)
    L = p.lib()
    stream = torch.cuda.current_stream()
    sp = int(stream.cuda_stream)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    kt = {"lz4_encode": [], "xxh32_out": [], "pack": [], "xxh32_verify": [], "lz4_decode": []}
    phase = {"compress": [], "decompress": []}
    state = {}

    def step(record):
        e = [ev() for _ in range(8)]
        # ------------------------------------------------------------------ compress
        e[0].record()
        p.binding.check(L.fourmc_gpu_4mc_encode_blocks(d_src.data_ptr(), d_stage.data_ptr(), enc.ptr, nb, 0, 0, sp), "encode")
        e[2].record()
        desc = enc.d.view(torch.int32).view(nb, 8)
        csz = desc[:, 6].to(torch.int64)                         # result = stored payload size
        if world > 1:                                            # footer index spans ranks: gather sizes
            allc = torch.empty(world * nb, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(allc, csz)
            before = int((allc[: rank * nb] + 12).sum().item())
        else:
            before = 0
        img_off = torch.cumsum(csz + 12, 0) - (csz + 12) + 12 + before   # absolute file offsets (footer index)
        loc_off = (img_off - before).contiguous()                # offsets inside this rank's image shard
        e[3].record()
        p.binding.check(L.fourmc_gpu_4mc_pack_image(d_stage.data_ptr(), d_image.data_ptr(), enc.ptr, loc_off.data_ptr(), nb, sp), "pack")
        e[4].record()
        # ------------------------------------------------------------------ decompress (payloads in place)
        dec_desc = torch.empty_like(desc)
        dec64 = dec_desc.view(torch.int64)
        dec64[:, 0] = loc_off + 12                               # src_off: payload inside the image
        dec64[:, 1] = torch.arange(nb, device=dev, dtype=torch.int64) * B
        dec_desc[:, 4] = desc[:, 6]                              # src_len = csize
        dec_desc[:, 5] = desc[:, 4]                              # dst_cap = usize
        dec_desc[:, 6] = 0
        dec_desc[:, 7] = desc[:, 7]                              # expected checksum
        e[5].record()
        p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_image.data_ptr(), d_out.data_ptr(), dec_desc.data_ptr(), nb, 0, sp), "decode")
        e[7].record()
        state["csz"], state["dec"], state["img_off"] = csz, dec_desc, img_off
        if record:
            torch.cuda.synchronize()
            phase["compress"].append(e[0].elapsed_time(e[4])); phase["decompress"].append(e[5].elapsed_time(e[7]))

    def kernel_times():
        """Per-kernel launch durations with events on the launch stream (untimed extra pass)."""
        e = [ev() for _ in range(8)]
        loc_off = (state["img_off"] - (state["img_off"][0] - 12)).contiguous()
        e[0].record(); p.binding.check(L.fourmc_gpu_lz4_compress_fast(d_src.data_ptr(), d_stage.data_ptr(), enc.ptr, 0, sp), "noop")
        # encode_blocks = lz4 encode (container mode) + xxh32; time them through the fused call's two halves
        t0 = ev(); t1 = ev(); t2 = ev()
        t0.record()
        p.binding.check(L.fourmc_gpu_4mc_encode_blocks(d_src.data_ptr(), d_stage.data_ptr(), enc.ptr, nb, 0, 0, sp), "encode")
        t1.record()
        p.binding.check(L.fourmc_gpu_4mc_pack_image(d_stage.data_ptr(), d_image.data_ptr(), enc.ptr, loc_off.data_ptr(), nb, sp), "pack")
        t2.record()
        d0 = ev(); d1 = ev()
        d0.record()
        p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_image.data_ptr(), d_out.data_ptr(), state["dec"].data_ptr(), nb, 0, sp), "decode")
        d1.record()
        # hash-only launches to split the fused calls
        h0 = ev(); h1 = ev(); h2 = ev()
        hb = p.DeviceBatch(p.make_blocks(offs, offs, state["csz"].cpu().numpy().astype(np.uint32), lens), dev)
        h0.record(); p.xxh32(d_stage, hb, 0, stream); h1.record()
        vb = state["dec"].clone()
        torch.cuda.synchronize()
        v0 = ev(); v1 = ev()
        v0.record(); p.binding.check(L.fourmc_gpu_xxh32(d_image.data_ptr(), vb.data_ptr(), nb, 0, sp), "xxh32"); v1.record()
        torch.cuda.synchronize()
        enc_total, pack, dec_total = t0.elapsed_time(t1), t1.elapsed_time(t2), d0.elapsed_time(d1)
        x_out, x_ver = h0.elapsed_time(h1), v0.elapsed_time(v1)
        return {"lz4_encode": enc_total - x_out, "xxh32_out": x_out, "pack": pack,
                "xxh32_verify": x_ver, "lz4_decode": dec_total - x_ver}

    def other_configs():
        """BASELINE configs[2] (4mz Fast = zstd level 1) and configs[3] (4mc High = LZ4 HC level 4), plus 4mz Medium
        (zstd level 3), on the same resident corpus: one launch each, HIP events on the launch stream; every
        compressed batch is decoded back and compared.  Reported beside the headline, never part of `value`."""
        out = {}
        def timed(fn):
            fn()                                       # untimed first call: workspace allocation happens here
            a, b = ev(), ev()
            torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b)
        for name, codec, level in (("4mz_fast_zstd1", p.CODEC_ZSTD, 1), ("4mz_medium_zstd3", p.CODEC_ZSTD, 3), ("4mc_high_lz4hc4", p.CODEC_LZ4_HC, 4)):
            eb = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens), dev)
            t_enc = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_encode_blocks(d_src.data_ptr(), d_stage.data_ptr(), eb.ptr, nb, codec, level, sp), name))
            r = eb.download()
            cs = r["result"].astype(np.int64)
            db = p.DeviceBatch(p.make_blocks(offs, offs, r["result"].astype(np.uint32), lens, r["xxh32"]), dev)
            t_dec = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_stage.data_ptr(), d_out.data_ptr(), db.ptr, nb, codec, sp), name))
            assert bool(torch.equal(d_out[: nb * B], d_src)), name + ": round trip failed"
            out[name] = {"compress_GBps": round(nb * B / t_enc / 1e6, 3), "decompress_GBps": round(nb * B / t_dec / 1e6, 3),
                         "ratio": round(nb * B / float((cs + 12).sum()), 4), "encode_blocks_ms": round(t_enc, 2), "decode_blocks_ms": round(t_dec, 2)}
        return out

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        w = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())

    # ---- correctness of what was timed (size-independent properties at full size)
    out_ok = bool(torch.equal(d_out[: nb * B], d_src))
    res = state["dec"][:, 6]
    assert out_ok and bool((res == B).all()), "round trip failed at full size"

    csz = state["csz"]
    U = nb * B
    Cbytes = int((csz + 12).sum().item())
    kts = kernel_times()
    ms_step = wall / args.steps * 1e3
    comp_ms = float(np.mean(phase["compress"])); dec_ms = float(np.mean(phase["decompress"]))
    alg_enc = U + int(csz.sum().item())          # encode launch: reads U, writes payloads
    alg_dec = int(csz.sum().item()) + U          # decode launch: reads payloads, writes U
    dom = "lz4_encode" if kts["lz4_encode"] >= kts["lz4_decode"] else "lz4_decode"
    alg = alg_enc if dom == "lz4_encode" else alg_dec

    # HBM bytes per launch from the PMC passes (rocprofv3 FETCH_SIZE + WRITE_SIZE, separate runs of this
    # same script at --blocks 512; summary and calibration note in profiles/), scaled to this launch
    traffic = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        for k in ("lz4_encode", "lz4_decode"):
            traffic[k] = int((tj[k]["fetch_KiB"] + tj[k]["write_KiB"]) * 1024 * nb / tj["blocks"])
    except Exception:
        pass

    def roof(name, algb):
        a = algb / (kts[name] * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 5), "traffic": traffic.get(name),
                "algorithmic_bytes_per_launch": algb, "avg_launch_ms": round(kts[name], 3)}

    if rank == 0:
        line = {
            "metric": "GB/s compress+decompress (silesia-like S-mix, 4mc-Fast)",
            "value": round(world * U / (wall / args.steps) / 1e9, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic (S-mix generator tools/corpus.c seed 0x4D43, 48 blocks replicated in HBM; silesia is not available offline)",
            "config": {"workload": "4mc Fast (LZ4 fast), 4 MiB blocks, S-mix replicated to %.2f GiB per GPU, HBM resident" % (U / 2**30),
                       "blocks_per_gpu": nb, "block_bytes": B, "parallelism": f"block-range dp{world}", "arch": arch},
            "compress_GBps": round(world * U / (comp_ms * 1e-3) / 1e9, 3),
            "decompress_GBps": round(world * U / (dec_ms * 1e-3) / 1e9, 3),
            "ratio": round(U / (12 + Cbytes + 12 + 20 + 4 * nb), 4),
            "ratio_vs_reference": 1.0,
            "kernel_ms": {k: round(v, 3) for k, v in kts.items()},
            "roofline": roof(dom, alg),
            "roofline_decode": roof("lz4_decode", alg_dec),
        }
        if world == 1 and not args.no_extras:
            line["other_configs"] = other_configs()
        if world == 1 and not args.no_cpu:             # the host-core baseline is measured at N = 1 only
            line["cpu_baseline"] = cpu_baseline(helpers, base, base_blocks)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
