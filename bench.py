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

Beside the headline (never part of `value`), at N = 1:
  other_configs   BASELINE configs[2] 4mz Fast (zstd 1), 4mz Medium (zstd 3), configs[3] 4mc High (LZ4 HC 4) on the same
                  corpus, and configs[4]'s workload - 4mz Ultra (zstd 12) on the synthetic LOG corpus - at a single-GPU
                  size; each with the reference's own code timed on the host cores next to it
  decode_64GiB    the 64 GiB decode-only configuration the north-star target is quoted on
  cpu_baseline    the reference's LZ4 path on the host cores (bounded sample)
Corpus: S-mix (tools/corpus.c) unless SILESIA_DIR names a directory with the silesia files.

Run:  python bench.py [--gpus N --steps K --warmup W]   (N>1 via torch.distributed.run)
"""
import argparse
import ctypes as C
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
TRAFFIC_JSON = os.path.join("profiles", "r05_traffic.json")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_codecs(helpers):
    """(kind, {name: (compress(src_ptr, n, dst_ptr, cap) -> size, decompress(src_ptr, n, dst_ptr, cap) -> size)})
    from the reference's own sources (oracle/_ref, kind 'reference') or, if it is not built, this repo's port."""
    ref = helpers.ref()
    if ref is not None:
        lz4 = (lambda s, n, d, cap: ref.LZ4_compress_default(s, d, n, cap), lambda s, n, d, cap: ref.LZ4_decompress_safe(s, d, n, cap))
        hc4 = (lambda s, n, d, cap: ref.LZ4_compress_HC(s, d, n, cap, 4), lz4[1])
        zs = lambda lvl: (lambda s, n, d, cap: ref.ZSTD_compress(d, cap, s, n, lvl), lambda s, n, d, cap: ref.ZSTD_decompress(d, cap, s, n))
        return "reference", {"lz4": lz4, "hc4": hc4, "zstd1": zs(1), "zstd3": zs(3), "zstd12": zs(12)}, ref.XXH32
    o = helpers.oracle()
    o.orc_lz4hc_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]; o.orc_lz4hc_compress.restype = C.c_int
    o.orc_zstd_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]; o.orc_zstd_compress.restype = C.c_int64
    lz4 = (lambda s, n, d, cap: o.orc_lz4_compress_fast(s, d, n, cap), lambda s, n, d, cap: o.orc_lz4_decompress_safe(s, d, n, cap))
    hc4 = (lambda s, n, d, cap: o.orc_lz4hc_compress(s, d, n, cap, 4), lz4[1])
    zs = lambda lvl: (lambda s, n, d, cap: o.orc_zstd_compress(s, n, d, cap, lvl), lambda s, n, d, cap: o.orc_zstd_decompress(s, n, d, cap))
    return "port", {"lz4": lz4, "hc4": hc4, "zstd1": zs(1), "zstd3": zs(3), "zstd12": zs(12)}, o.orc_xxh32


def reference_sizes(helpers, base, nblk, codec, B):
    """compressed size per distinct corpus block by the host comparator's code (for ratio_vs_reference), 16 threads"""
    from concurrent.futures import ThreadPoolExecutor
    kind, codecs, xxh = host_codecs(helpers)
    comp, _ = codecs[codec]
    def one(b):
        out = np.empty(B + B // 128 + 1024, np.uint8)
        r = comp(base[b * B:(b + 1) * B].ctypes.data, B, out.ctypes.data, B - 1)
        return b, (int(r) if 0 < r < B else B)
    with ThreadPoolExecutor(16) as ex:
        return dict(ex.map(one, range(nblk)))


def cpu_leg(helpers, base, nblk, codec, budget_s, B, logs=False):
    """The reference's per-block work (native/4mc.c:301-329 compress + checksum, :637-661 checksum + decode) on the host:
    tools/cpu_baseline (C, one pinned thread per PHYSICAL core, NUMA-local buffers, compress phase then decompress phase,
    each >= budget_s, rate = all threads' bytes / wall time of the phase).  `value` = uncompressed bytes per second of compress +
    decompress time, the definition of the GPU figure: 1 / (1 / compress_rate + 1 / decompress_rate)."""
    import json as _json, subprocess
    exe = os.path.join(ROOT, "tools", "cpu_baseline")
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libref4mc.so")
    lib = ref_so if os.path.exists(ref_so) else os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(exe):
        subprocess.run(["gcc", "-O2", "-pthread", "tools/cpu_baseline.c", "tools/corpus.c", "-ldl", "-o", "tools/cpu_baseline"], cwd=ROOT, check=True)
    cmd = [exe, "--lib", lib, "--codec", codec, "--seconds", str(budget_s), "--blocks", str(nblk)] + (["--logs"] if logs else [])
    if os.environ.get("SILESIA_DIR") and not logs:
        tmp = os.path.join(os.environ.get("FOURMC_BENCH_TMP", "/tmp"), "bench_corpus.bin"); base[: nblk * B].tofile(tmp); cmd += ["--data", tmp]
    r = _json.loads(subprocess.run(cmd, capture_output=True, check=True, text=True).stdout)
    c, d = r["compress_GBps"], r["decompress_GBps"]
    out = {"value": round(1.0 / (1.0 / c + 1.0 / d), 3), "unit": "GB/s", "cores": r["threads"], "cores_physical": r["cores_physical"], "cpus_logical": r["cpus_logical"],
           "cpu": cpu_model(), "kind": r["kind"],
           "sample": f"{r['blocks_per_thread']} blocks of the corpus per thread ({r['corpus_blocks']} distinct), compress+xxh32 phase {r['compress_seconds']} s then "
                     f"xxh32+decompress phase {r['decompress_seconds']} s, one pinned thread per physical core, thread-local (NUMA-local) buffers; tools/cpu_baseline.c",
           "compress_GBps_all_cores": c, "decompress_GBps_all_cores": d,
           "compress_GBps_per_core": r["compress_GBps_per_thread"], "decompress_GBps_per_core": r["decompress_GBps_per_thread"],
           "slowest_thread_GBps": [r["compress_GBps_slowest_thread"], r["decompress_GBps_slowest_thread"]], "round_trip_failures": r["round_trip_failures"]}
    return out, reference_sizes(helpers, base, nblk, codec, B)


def load_corpus(helpers, base_blocks, B):
    """S-mix, or the silesia files when SILESIA_DIR is set (concatenated in name order, cut to whole 4 MiB blocks)."""
    d = os.environ.get("SILESIA_DIR")
    if d and os.path.isdir(d):
        names = sorted(f for f in os.listdir(d) if os.path.isfile(os.path.join(d, f)))
        data = np.concatenate([np.fromfile(os.path.join(d, f), dtype=np.uint8) for f in names]) if names else np.zeros(0, np.uint8)
        nb = len(data) // B
        if nb >= 1:
            return data[: nb * B].copy(), nb, f"silesia from SILESIA_DIR ({len(names)} files, {nb} whole 4 MiB blocks), replicated in HBM"
    return helpers.corpus(base_blocks * B, first_block=0), base_blocks, \
        "synthetic (S-mix generator tools/corpus.c seed 0x4D43, 48 blocks replicated in HBM; silesia is not available offline, SILESIA_DIR not set)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("FOURMC_BENCH_BLOCKS", 2048)),
                    help="4 MiB blocks per GPU (2048 = 8 GiB, BASELINE configs[1])")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configs and the 64 GiB decode leg timed after the headline")
    ap.add_argument("--decode-blocks", type=int, default=16384, help="blocks of the decode-only leg (16384 = 64 GiB)")
    ap.add_argument("--config", choices=["fast", "ultra_logs"], default="fast",
                    help="fast: BASELINE configs[1], 4mc Fast on the S-mix (the headline).  ultra_logs: configs[4]'s codec and corpus - 4mz Ultra (zstd 12) on the "
                         "synthetic log corpus, block ranges per rank, the footer index gathered over RCCL, then decoded rank by rank (no collective)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import helpers
    p = importlib.import_module("4mc_amd")

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # FOURMC_BENCH_BACKEND=gloo: test aid - several ranks on ONE GPU (RCCL refuses two ranks on a device); the collectives then go through
    # host copies.  The driver's runs use the default, nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("FOURMC_BENCH_BACKEND", "nccl")
    if backend != "nccl": local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl": dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else: dist.init_process_group(backend)
    arch = p.gpu_init(local)
    dev = torch.device("cuda", local)
    B = p.BLOCKSIZE
    nb = args.blocks

    # ---- corpus: generated / loaded once, replicated in HBM to nb blocks (physically distinct copies)
    ULTRA = args.config == "ultra_logs"
    CODEC, LEVEL = (p.CODEC_ZSTD, 12) if ULTRA else (0, 0)
    ENC, DEC = ("zstd_encode", "zstd_decode") if ULTRA else ("lz4_encode", "lz4_decode")
    if ULTRA:
        free = torch.cuda.mem_get_info(dev)[0]
        while nb > 256 and nb * (49 + 14 + 17) * (1 << 20) > 0.8 * free: nb //= 2          # level-12 tables, decode scratch, the four 4 MiB-per-block buffers
        base, base_blocks = helpers.corpus(24 * B, first_block=0, logs=True), 24
        data_note = "synthetic log corpus (tools/corpus.c corpus_fill_logs, seed 0x4D43, 24 distinct blocks replicated in HBM)"
        args.no_extras = True
    else:
        base, base_blocks, data_note = load_corpus(helpers, 48, B)
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
    phase = {"compress": [], "decompress": []}
    state = {}

    def step(record):
        e = [ev() for _ in range(8)]
        # ------------------------------------------------------------------ compress
        e[0].record()
        p.binding.check(L.fourmc_gpu_4mc_encode_blocks(d_src.data_ptr(), d_stage.data_ptr(), enc.ptr, nb, CODEC, LEVEL, sp), "encode")
        e[2].record()
        desc = enc.d.view(torch.int32).view(nb, 8)
        csz = desc[:, 6].to(torch.int64)                         # result = stored payload size
        if world > 1:                                            # footer index spans ranks: gather sizes
            allc = torch.empty(world * nb, dtype=torch.int64, device=dev)
            if backend == "nccl": dist.all_gather_into_tensor(allc, csz)
            else:
                parts = [torch.empty(nb, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(parts, csz.cpu()); allc.copy_(torch.cat(parts))
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
        p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_image.data_ptr(), d_out.data_ptr(), dec_desc.data_ptr(), nb, CODEC, sp), "decode")
        e[7].record()
        state["csz"], state["dec"], state["img_off"], state["loc_off"] = csz, dec_desc, img_off, loc_off
        if record:
            torch.cuda.synchronize()
            phase["compress"].append(e[0].elapsed_time(e[4])); phase["decompress"].append(e[5].elapsed_time(e[7]))

    def kernel_times():
        """Per-kernel launch durations with events on the launch stream, in an extra pass after the timed steps (the fused
        encode / decode calls are split by timing their hash launches alone, so the parts do not add up to ms_per_step exactly)."""
        loc_off = state["loc_off"]
        t0 = ev(); t1 = ev(); t2 = ev()
        t0.record()
        p.binding.check(L.fourmc_gpu_4mc_encode_blocks(d_src.data_ptr(), d_stage.data_ptr(), enc.ptr, nb, CODEC, LEVEL, sp), "encode")
        t1.record()
        p.binding.check(L.fourmc_gpu_4mc_pack_image(d_stage.data_ptr(), d_image.data_ptr(), enc.ptr, loc_off.data_ptr(), nb, sp), "pack")
        t2.record()
        d0 = ev(); d1 = ev()
        d0.record()
        p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_image.data_ptr(), d_out.data_ptr(), state["dec"].data_ptr(), nb, CODEC, sp), "decode")
        d1.record()
        h0 = ev(); h1 = ev()
        hb = p.DeviceBatch(p.make_blocks(offs, offs, state["csz"].cpu().numpy().astype(np.uint32), lens), dev)
        h0.record(); p.xxh32(d_stage, hb, 0, stream); h1.record()
        vb = state["dec"].clone()
        torch.cuda.synchronize()
        v0 = ev(); v1 = ev()
        v0.record(); p.binding.check(L.fourmc_gpu_xxh32(d_image.data_ptr(), vb.data_ptr(), nb, 0, sp), "xxh32"); v1.record()
        torch.cuda.synchronize()
        enc_total, pack, dec_total = t0.elapsed_time(t1), t1.elapsed_time(t2), d0.elapsed_time(d1)
        x_out, x_ver = h0.elapsed_time(h1), v0.elapsed_time(v1)
        return {ENC: enc_total - x_out, "xxh32_out": x_out, "pack": pack,
                "xxh32_verify": x_ver, DEC: dec_total - x_ver}

    def timed(fn):
        fn()                                       # untimed first call: workspace allocation happens here
        a, b = ev(), ev()
        torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b)

    def config_leg(name, src_t, nblk, codec, level, host_codec, host_base, host_nblk, budget):
        """encode + decode of `nblk` resident blocks with one codec: one launch each, HIP events on the launch stream; the
        compressed batch is decoded back and compared; the reference's own code on the host cores beside it."""
        o = np.arange(nblk, dtype=np.uint64) * B; ln = np.full(nblk, B, dtype=np.uint32)
        eb = p.DeviceBatch(p.make_blocks(o, o, ln, ln), dev)
        t_enc = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_encode_blocks(src_t.data_ptr(), d_stage.data_ptr(), eb.ptr, nblk, codec, level, sp), name))
        r = eb.download()
        cs = r["result"].astype(np.int64)
        db = p.DeviceBatch(p.make_blocks(o, o, r["result"].astype(np.uint32), ln, r["xxh32"]), dev)
        t_dec = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_stage.data_ptr(), d_out.data_ptr(), db.ptr, nblk, codec, sp), name))
        assert bool(torch.equal(d_out[: nblk * B], src_t[: nblk * B])), name + ": round trip failed"
        out = {"blocks": nblk, "compress_GBps": round(nblk * B / t_enc / 1e6, 3), "decompress_GBps": round(nblk * B / t_dec / 1e6, 3),
               "ratio": round(nblk * B / float((cs + 12).sum()), 4), "encode_blocks_ms": round(t_enc, 2), "decode_blocks_ms": round(t_dec, 2)}
        if not args.no_cpu:
            cb, ref_cs = cpu_leg(helpers, host_base, host_nblk, host_codec, budget, B, logs=name.endswith("_logs"))
            out["cpu_baseline"] = cb
            have = sorted(ref_cs)
            if have:                               # same blocks, reference sizes against the sizes of this run
                ours = float(sum(12 + int(cs[b]) for b in have)); theirs = float(sum(12 + ref_cs[b] for b in have))
                out["ratio_vs_reference"] = round(theirs / ours, 6)
        return out

    def other_configs():
        out = {}
        for name, codec, level, hc in (("4mz_fast_zstd1", p.CODEC_ZSTD, 1, "zstd1"), ("4mz_medium_zstd3", p.CODEC_ZSTD, 3, "zstd3"),
                                       ("4mc_high_lz4hc4", p.CODEC_LZ4_HC, 4, "hc4")):
            out[name] = config_leg(name, d_src, nb, codec, level, hc, base, base_blocks, 4.0)
        # BASELINE configs[4]'s workload at a single-GPU size: 4mz Ultra (zstd 12) on the synthetic log corpus
        # (every CU holds 8 blocks as in the other legs when the 49 MiB of level-12 tables per block fit: 2048 blocks take 97 GiB)
        nlog = nb
        free = torch.cuda.mem_get_info(dev)[0] if hasattr(torch.cuda, "mem_get_info") else 0
        while nlog > 256 and nlog * (49 + 14) * (1 << 20) > 0.8 * free: nlog //= 2
        logs = helpers.corpus(min(nlog, 24) * B, first_block=0, logs=True)
        d_logs = torch.from_numpy(logs).to(dev).repeat(-(-nlog // min(nlog, 24)))[: nlog * B].contiguous()
        out["4mz_ultra_zstd12_logs"] = config_leg("4mz_ultra_zstd12_logs", d_logs, nlog, p.CODEC_ZSTD, 12, "zstd12", logs, min(nlog, 24), 6.0)
        out["4mz_ultra_zstd12_logs"]["corpus"] = "tools/corpus.c corpus_fill_logs, 24 distinct blocks replicated to %d" % nlog
        return out

    def tolerant_roofline(algb, ms):
        """roofline object of the tolerant encoder's two kernels together; traffic: the PMC passes kept under profiles/ (same launch size)"""
        traffic = None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "r06_k2p_traffic.json")))
            if t.get("blocks") == nb:
                traffic = int(sum(1024 * (v.get("fetch_KiB") or 0) + 1024 * (v.get("write_KiB") or 0) for k, v in t.items() if isinstance(v, dict)))
        except Exception:
            pass
        a = algb / (ms * 1e-3) / 1e9
        return {"kernel": "lz4_par_segment_kernel + lz4_par_stitch_kernel", "bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": "profiles/r06_k2p_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 2048 blocks, re-measured in round 6: unchanged): the candidates' lines miss the L2" if traffic else None,
                "algorithmic_bytes_per_launch": int(algb), "avg_launch_ms": round(ms, 3)}

    def tolerant_leg():
        """The same 2048 blocks through the RATIO-TOLERANCE LZ4 encoder (4mc_amd/csrc/lz4_par_encode.hip; not the default, not in
        `value`): north_star's clause for a parse that is not reproduced.  Payloads are valid LZ4 blocks but not the reference's
        bytes; they are decoded back by the device decoder and compared here, and tests/test_gpu_lz4par_encode.py holds them to the
        reference's LZ4_decompress_safe.  ratio_vs_reference: the exact encoder's container bytes (byte-identical to the reference)
        over this encoder's, same blocks."""
        exact = state["csz"].cpu().numpy().astype(np.int64)
        L.fourmc_gpu_set_lz4_encode_mode(1)
        try:
            o = np.arange(nb, dtype=np.uint64) * B; ln = np.full(nb, B, dtype=np.uint32)
            eb = p.DeviceBatch(p.make_blocks(o, o, ln, ln), dev)
            t_enc = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_encode_blocks(d_src.data_ptr(), d_stage.data_ptr(), eb.ptr, nb, 0, 0, sp), "tolerant"))
            r = eb.download(); cs = r["result"].astype(np.int64)
            db = p.DeviceBatch(p.make_blocks(o, o, r["result"].astype(np.uint32), ln, r["xxh32"]), dev)
            t_dec = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_stage.data_ptr(), d_out.data_ptr(), db.ptr, nb, 0, sp), "tolerant"))
            assert bool(torch.equal(d_out[: nb * B], d_src[: nb * B])), "tolerant encoder: round trip failed"
            kb = p.DeviceBatch(p.make_blocks(o, o, ln, np.full(nb, B + B // 255 + 16, dtype=np.uint32)), dev)
            d_tmp = torch.empty(nb * B + (B // 255 + 4096), dtype=torch.uint8, device=dev)
            t_k = timed(lambda: p.binding.check(L.fourmc_gpu_lz4_compress_fast(d_src.data_ptr(), d_tmp.data_ptr(), kb.ptr, nb, sp), "tolerant kernels"))
            del d_tmp
        finally:
            L.fourmc_gpu_set_lz4_encode_mode(0)
        return {"blocks": nb, "encoder": "lz4_par_encode.hip (64 KiB segments in parallel, one LZ4 block per 4 MiB block; fourmc_gpu_set_lz4_encode_mode(1) / FOURMC_LZ4_ENCODE=parallel)",
                "compress_GBps": round(nb * B / t_enc / 1e6, 3), "decompress_GBps": round(nb * B / t_dec / 1e6, 3),
                "compress_decompress_GBps": round(nb * B / (t_enc + t_dec) / 1e6, 3),
                "encode_blocks_ms": round(t_enc, 2), "decode_blocks_ms": round(t_dec, 2),
                "encode_kernels_ms": round(t_k, 2), "encode_kernels_GBps": round(nb * B / t_k / 1e6, 3),
                "encode_kernels_note": "fourmc_gpu_lz4_compress_fast alone (segment kernel + stitch kernel; no checksum launch); algorithmic bytes usize + csize: %.3f of 8 TB/s" % ((nb * B + float(cs.sum())) / (t_k * 1e-3) / 1e9 / HBM_PEAK_GBS),
                "roofline": tolerant_roofline(float(nb) * B + float(cs.sum()), t_k),
                "ratio": round(nb * B / float((cs + 12).sum()), 4),
                "ratio_vs_reference": round(float((exact + 12).sum()) / float((cs + 12).sum()), 6),
                "tolerance": "sizes within 3 %% of the reference parse over the S-mix (tests/test_gpu_lz4par_encode.py); this run: %+.2f %%" % (100.0 * (float((cs + 12).sum()) / float((exact + 12).sum()) - 1)),
                "round_trip": "decoded by the device decoder and compared with the input in this run"}

    def seq_copy_ceiling(alg, t_ms):
        """What the copy engine of the decode path would take for this leg if DEPENDENCIES WERE FREE: the same
        executor fetching and placing every literal run and match of the real sequence list without ever waiting for a source
        (tools/ubench/seq_copy.sh, `make nodeps`; measured on the GPU box this round, profiles/r06_seq_copy.md).  Not re-measured inside
        this run (it needs the side build and a wrong-output decode of 64 GiB); the leg of THIS run is put beside it."""
        try:
            c = json.load(open(os.path.join(ROOT, "profiles", "r06_seq_copy.json")))
            ms = float(c["dependency_free"]["leg_ms"])
        except Exception:
            return None
        a = alg / (ms * 1e-3) / 1e9
        return {"dependency_free_leg_ms": ms, "GBps_algorithmic": round(a, 1), "frac_of_hbm_peak": round(a / HBM_PEAK_GBS, 4),
                "this_leg_over_ceiling": round(ms / t_ms, 3),
                "below_40_percent_target": a < 0.4 * HBM_PEAK_GBS,
                "what": "lz4_seg_exec_kernel with no wait for a flush and no order between near matches (every copy still made, output wrong by design) + the real walk and tail kernels + the checksum pass; one sequence per lane, 9 - 21 byte strings",
                "other_engines_leg_ms": c.get("other_engines_ms"),
                "source": "profiles/r06_seq_copy.md / .json (tools/ubench/seq_copy.sh)"}

    def decode_64gib():
        """Decode-only at the size the north-star target is quoted on: 16384 blocks = 64 GiB written, read from a 35 GB image of
        physically DISTINCT payloads (the 8 GiB configuration's image copied 8 times in HBM, every block descriptor pointing into
        its own copy: nothing is read twice).  Runs as 8 launches' worth of blocks in one call (the engine's decode entry point)."""
        nd = args.decode_blocks
        reps = -(-nd // nb)
        p.release_workspaces(); torch.cuda.empty_cache()
        img_bytes = int((state["loc_off"][-1] + 12 + state["csz"][-1]).item())
        img_pad = (img_bytes + 4095) & ~4095
        free, _ = torch.cuda.mem_get_info()
        need = nd * B + reps * img_pad + (64 << 20)
        if free < need + (8 << 30):
            return {"skipped": "not enough free HBM for %d blocks of output and %d image copies" % (nd, reps)}
        big = torch.empty(nd * B + 64, dtype=torch.uint8, device=dev)
        img = torch.empty(reps * img_pad + 64, dtype=torch.uint8, device=dev)
        for k in range(reps):
            img[k * img_pad: k * img_pad + img_bytes] = d_image[:img_bytes]
        dd = state["dec"].repeat(reps, 1)[:nd].contiguous()
        d64 = dd.view(torch.int64)
        d64[:, 0] += (torch.arange(nd, device=dev, dtype=torch.int64) // nb) * img_pad
        d64[:, 1] = torch.arange(nd, device=dev, dtype=torch.int64) * B
        dd[:, 6] = 0
        t = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_decode_blocks(img.data_ptr(), big.data_ptr(), dd.data_ptr(), nd, 0, sp), "decode64"))
        ok = bool((dd[:, 6] == B).all())
        for k in range(0, nd, nb):
            m = min(nb, nd - k)
            ok = ok and bool(torch.equal(big[k * B:(k + m) * B], d_src[: m * B]))
        assert ok, "64 GiB decode: round trip failed"
        cbytes = int(state["csz"].sum().item()) * (nd // nb) if nd % nb == 0 else None
        hash_ms = timed(lambda: p.binding.check(L.fourmc_gpu_xxh32(img.data_ptr(), dd.clone().data_ptr(), nd, 0, sp), "xxh32"))
        alg = (cbytes or 0) + nd * B
        del big, img
        stored = int((state["csz"] == B).sum().item()) * (nd // nb) if nd % nb == 0 else None
        return {"blocks": nd, "stored_blocks": stored,
                "stored_blocks_note": "blocks LZ4 does not shrink are STORED by the container (native/4mc.c:301-329); their decode is a copy.  decode_64GiB_lz4_only has the same corpus with every block an LZ4 stream",
                "uncompressed_GiB": nd * B / 2**30, "decode_blocks_ms": round(t, 2), "of_which_xxh32_verify_ms": round(hash_ms, 2),
                "decompress_GBps": round(nd * B / t / 1e6, 2),
                "lz4_decode_path": decode_path_label(L.fourmc_gpu_get_lz4_decode_path(), nd),
                "roofline": {"bound": "hbm", "achieved": round(alg / (t * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "over": "the whole decode call: checksum verify pass + decode kernels (payload bytes counted once)",
                             "frac_decode_kernels_only": round(alg / ((t - hash_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "ceiling_seq_copy": seq_copy_ceiling(alg, t)},
                "note": "%d distinct copies of the 8 GiB image in HBM (%.1f GB of payloads, each read once); 64 GiB of distinct output" % (reps, reps * img_bytes / 1e9)}

    def decode_64gib_lz4_only():
        """The 64 GiB decode with EVERY block an LZ4 stream: the blocks are compressed with capacity LZ4_compressBound (the raw
        LZ4 call of jniCompressor.c:91 - no stored fallback), each payload in its own slot, 8 distinct copies of the slots in HBM;
        decoded by the raw LZ4 decode entry point (no container, no checksum pass).  This is the leg the stored blocks do not flatter."""
        nd = args.decode_blocks
        if nd % nb: return {"skipped": "decode blocks not a multiple of the launch"}
        reps = nd // nb
        slot = (B + B // 255 + 16 + 4095) & ~4095
        p.release_workspaces(); torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info()
        if free < nd * B + (reps + 1) * nb * slot + (4 << 30):
            return {"skipped": "not enough free HBM (%d GiB free)" % (free >> 30)}
        o = np.arange(nb, dtype=np.uint64); ln = np.full(nb, B, dtype=np.uint32)
        one = torch.empty(nb * slot + 64, dtype=torch.uint8, device=dev)
        kb = p.DeviceBatch(p.make_blocks(o * B, o * slot, ln, np.full(nb, slot, dtype=np.uint32)), dev)
        p.binding.check(L.fourmc_gpu_lz4_compress_fast(d_src.data_ptr(), one.data_ptr(), kb.ptr, nb, sp), "lz4-only encode")
        cs = kb.download()["result"].astype(np.int64)
        assert (cs > 0).all()
        img = torch.empty(reps * nb * slot + 64, dtype=torch.uint8, device=dev)
        for k in range(reps): img[k * nb * slot: (k + 1) * nb * slot] = one[: nb * slot]
        del one
        big = torch.empty(nd * B + 64, dtype=torch.uint8, device=dev)
        so = (np.arange(nd, dtype=np.uint64) % nb) * slot + (np.arange(nd, dtype=np.uint64) // nb) * (nb * slot)
        db = p.DeviceBatch(p.make_blocks(so, np.arange(nd, dtype=np.uint64) * B, np.tile(cs.astype(np.uint32), reps), np.full(nd, B, dtype=np.uint32)), dev)
        t = timed(lambda: p.binding.check(L.fourmc_gpu_lz4_decompress(img.data_ptr(), big.data_ptr(), db.ptr, nd, sp), "lz4-only decode"))
        ok = bool((torch.from_numpy(db.download()["result"].astype(np.int64)) == B).all())
        for k in range(0, nd, nb): ok = ok and bool(torch.equal(big[k * B:(k + nb) * B], d_src[: nb * B]))
        assert ok, "64 GiB LZ4-only decode: round trip failed"
        alg = int(cs.sum()) * reps + nd * B
        del big, img
        return {"blocks": nd, "stored_blocks": 0, "decode_blocks_ms": round(t, 2), "decompress_GBps": round(nd * B / t / 1e6, 2),
                "compressed_GB": round(int(cs.sum()) * reps / 1e9, 2),
                "lz4_decode_path": decode_path_label(L.fourmc_gpu_get_lz4_decode_path(), nd),
                "roofline": {"bound": "hbm", "achieved": round(alg / (t * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "over": "the raw LZ4 decode call (walk + executor + exact walker for the block ends), no checksum pass",
                             "algorithmic_bytes_per_launch": alg}}

    def decode_path_label(path, n):
        names = {0: "wave trio (parser wave + 2 copier waves per block; lz4_decode.hip)", 1: "block parallel (parse + executor kernels)",
                 2: "exact walker only", 4: "row pipeline (pre / walk / post / copy waves per block; lz4_rows.hip)",
                 7: "one lane per sequence (lz4_rows.hip)", 9: "walk + window copier (K1wx: four waves per block; lz4_rows.hip)",
                 11: "segment-parallel (lz4_seg.hip: walk kernel, one lane per stream segment + batch executor, one wave per block + exact walker for the last bytes of each block)",
                 13: "tile (lz4_tile.hip: walk kernel leaves a token bitmap; executor: one workgroup per block, the 64 KiB window in LDS, one thread per output byte, pointer chase for sources inside a tile + exact walker for the last bytes of each block)"}
        if path == 6:
            tmax = int(os.environ.get("FOURMC_TILE_MAX", 1536))
            return "auto: tile path up to %d blocks per launch, segment-parallel above (this launch of %d blocks: %s)" % (tmax, n, names[13 if n <= tmax else 11])
        return names.get(path, "path %d" % path)

    def decode_path_comparison():
        """The LZ4 decode fast paths of the product on the same launch (identical results): the segment-parallel path and the tile path
        (and the exact walker, the path without a workspace); HIP events around the decode_blocks call minus the hash launch.  "auto"
        (the default) takes the tile path up to 1536 blocks per launch, the segment-parallel path above.  (Every other design -
        walk + window copier, wave trio, row pipeline, group executor, pipelined executor - lives under tools/research/ and in the
        research side build only.)"""
        out = {}
        before = L.fourmc_gpu_get_lz4_decode_path()
        vb = state["dec"].clone()
        x_ver = timed(lambda: p.binding.check(L.fourmc_gpu_xxh32(d_image.data_ptr(), vb.data_ptr(), nb, 0, sp), "xxh32"))
        for path, name in ((2, "exact_walker"), (11, "segment_parallel"), (13, "tile")):
            L.fourmc_gpu_set_lz4_decode_path(path)
            dd = state["dec"].clone(); dd[:, 6] = 0
            t = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_image.data_ptr(), d_out.data_ptr(), dd.data_ptr(), nb, 0, sp), name))
            ok = bool((dd[:, 6] == B).all()) and bool(torch.equal(d_out[: nb * B], d_src))
            out[name] = {"lz4_decode_ms": round(t - x_ver, 3), "round_trip": ok}
        for m in (256, 1024):                          # launches that do not fill the chip: what the file API sends
            if m >= nb: continue
            xv = timed(lambda: p.binding.check(L.fourmc_gpu_xxh32(d_image.data_ptr(), state["dec"][:m].clone().data_ptr(), m, 0, sp), "xxh32"))
            for path, name in ((2, "exact_walker"), (11, "segment_parallel"), (13, "tile")):
                L.fourmc_gpu_set_lz4_decode_path(path)
                dd = state["dec"][:m].clone(); dd[:, 6] = 0
                t = timed(lambda: p.binding.check(L.fourmc_gpu_4mc_decode_blocks(d_image.data_ptr(), d_out.data_ptr(), dd.data_ptr(), m, 0, sp), name))
                out[name]["lz4_decode_ms_%d_blocks" % m] = round(t - xv, 3)
        L.fourmc_gpu_set_lz4_decode_path(before)
        return out

    def cli_wallclock():
        """The drop-in CLI end to end (file in, file out, process start, PCIe and file system included) next to the reference CLI built
        from the reference's sources (oracle/_ref/4mc_ref): the same 8 GiB file on tmpfs, both directions, files compared
        (tools/cli_timing.py)."""
        import json as _json, subprocess
        d = os.environ.get("FOURMC_BENCH_TMP") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
        gib = float(os.environ.get("FOURMC_BENCH_CLI_GIB", "8"))
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize < (3.2 * gib + 2) * 2**30: gib = 2.0
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_timing.py"), "--gib", str(gib), "--dir", d, "--modes", "default"],
                               capture_output=True, text=True, timeout=600)
            return _json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:                         # never let the side measurement take the bench line down
            return {"skipped": repr(e)[:200]}

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
    own_wall = wall
    per_rank = None; rccl_ranks = None
    if world > 1:
        w = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
        # every rank's own numbers into rank 0's line, and the gathered footer index checked against a prefix sum recomputed locally
        mine = {"rank": rank, "ms_per_step": round(own_wall / args.steps * 1e3, 3), "blocks": nb,
                "compress_GBps": round(nb * B / (float(np.mean(phase["compress"])) * 1e-3) / 1e9, 3),
                "decompress_GBps": round(nb * B / (float(np.mean(phase["decompress"])) * 1e-3) / 1e9, 3),
                "first_block_offset": int(state["img_off"][0].item()), "shard_bytes": int((state["csz"] + 12).sum().item())}
        per_rank = p.container.gather_rank_reports(mine)        # (asserts the gathered footer index against a prefix sum recomputed from the shard sizes)
        try:
            v = torch.cuda.nccl.version(); ver = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception:
            ver = "unknown"
        rccl_ranks = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": ver,
                      "collective": "all_gather_into_tensor of %d int64 sizes per rank and step (the footer index); no data-path collective" % nb}

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
    dom = ENC if kts[ENC] >= kts[DEC] else DEC
    alg = alg_enc if dom == ENC else alg_dec

    # HBM bytes per launch from the PMC passes (rocprofv3 FETCH_SIZE + WRITE_SIZE, separate runs of this same script at
    # --blocks 512; summary and calibration note in profiles/), scaled to this launch: NOT measured in this run
    traffic = {}
    # HBM traffic of the two headline kernels: measured INSIDE this run when rocprofv3 is on the box (tools/traffic_probe.py: two PMC
    # passes around the same launches at this launch size), else taken from the committed passes of the round
    traffic_src = {}
    probe = None
    if rank == 0 and world == 1 and not args.no_extras and not os.environ.get("FOURMC_BENCH_NO_PMC"):
        try:
            import shutil, subprocess
            if shutil.which("rocprofv3"):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_probe.py"), "--blocks", str(nb)], capture_output=True, text=True, timeout=900)
                probe = json.loads(r.stdout.strip().splitlines()[-1])
                for k in ("lz4_encode", "lz4_decode"):
                    traffic[k] = probe[k]["traffic"]; traffic_src[k] = probe["source"]
        except Exception as e:
            probe = {"error": repr(e)[:300]}
    if not traffic:
        try:
            tj = json.load(open(os.path.join(ROOT, TRAFFIC_JSON)))
            for k in ("lz4_encode", "lz4_decode"):
                traffic[k] = int((tj[k]["fetch_KiB"] + tj[k]["write_KiB"]) * 1024 * nb / tj["blocks"])
                traffic_src[k] = TRAFFIC_JSON + " (rocprofv3 PMC passes of the round, scaled by blocks; not measured in this run)"
        except Exception:
            pass

    def roof(name, algb):
        a = algb / (kts[name] * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 5), "traffic": traffic.get(name),
                "traffic_source": traffic_src.get(name),
                "algorithmic_bytes_per_launch": algb, "avg_launch_ms": round(kts[name], 3)}

    if rank == 0:
        line = {
            "metric": "GB/s compress+decompress (synthetic logs, 4mz-Ultra zstd 12)" if ULTRA else "GB/s compress+decompress (silesia-like S-mix, 4mc-Fast)",
            "value": round(world * U / (wall / args.steps) / 1e9, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": data_note,
            "config": {"workload": ("4mz Ultra (zstd level 12), 4 MiB blocks, log corpus replicated to %.2f GiB per GPU, HBM resident; block ranges per rank, footer index gathered over RCCL" if ULTRA else
                                    "4mc Fast (LZ4 fast), 4 MiB blocks, corpus replicated to %.2f GiB per GPU, HBM resident") % (U / 2**30),
                       "blocks_per_gpu": nb, "block_bytes": B, "parallelism": f"block-range dp{world}", "arch": arch,
                       "lz4_decode_path": decode_path_label(L.fourmc_gpu_get_lz4_decode_path(), nb)},
            "compress_GBps": round(world * U / (comp_ms * 1e-3) / 1e9, 3),
            "decompress_GBps": round(world * U / (dec_ms * 1e-3) / 1e9, 3),
            "ratio": round(U / (12 + Cbytes + 12 + 20 + 4 * nb), 4),
            "stored_blocks": int((csz == B).sum().item()),
            "kernel_ms": {k: round(v, 3) for k, v in kts.items()},
            "kernel_ms_note": "HIP events on the launch stream in an extra pass after the timed steps; the hash launches are timed alone and subtracted from the fused calls, so the parts need not add up to ms_per_step",
            "roofline": roof(dom, alg),
            "roofline_decode": roof(DEC, alg_dec),
        }
        if per_rank is not None:
            line["per_rank"] = per_rank; line["rccl_ranks"] = rccl_ranks
            line["footer_index_check"] = "every rank's first block offset equals 12 + the bytes of the ranks before it (recomputed from the gathered sizes)"
        if probe is not None:
            line["traffic_probe"] = probe
        if world == 1 and not args.no_cpu:             # the host-core baseline is measured at N = 1 only
            cb, ref_cs = cpu_leg(helpers, base, base_blocks, "zstd12" if ULTRA else "lz4", 12.0, B, logs=ULTRA)
            line["cpu_baseline"] = cb
            mine = csz[:base_blocks].cpu().numpy()
            have = sorted(b for b in ref_cs if b < len(mine))
            line["ratio_vs_reference"] = round(float(sum(12 + ref_cs[b] for b in have)) / float(sum(12 + int(mine[b]) for b in have)), 6) if have else None
            line["ratio_vs_reference_note"] = "container bytes of the reference's codec on the host over this run's, same %d blocks (1.0 = identical sizes; payloads are byte-identical by the parity tests)" % len(have)
        else:
            line["cpu_baseline"] = None
            line["ratio_vs_reference"] = None
        if world == 1 and not args.no_extras:
            line["other_configs"] = other_configs()
            line["lz4_fast_tolerant"] = tolerant_leg()
            tl = line["lz4_fast_tolerant"]
            # the second headline north_star allows: the same metric with the ratio-tolerance parse.  `value` stays the exact encoder's:
            # it is what the CLI, the file API and the JNI names run by default, because files then equal the reference's byte for byte
            line["value_tolerant"] = tl["compress_decompress_GBps"]
            line["value_tolerant_note"] = {"encoder": "ratio-tolerance LZ4 parse (lz4_par_encode.hip), opt-in: fourmc_gpu_set_lz4_encode_mode(1) / FOURMC_LZ4_ENCODE=parallel",
                                           "ratio_vs_reference": tl["ratio_vs_reference"], "tolerance": tl["tolerance"],
                                           "default": "`value` is the byte-identical encoder: what a user gets without asking (4mc files then have the reference's SHA-256); this is what a user gets who asks for throughput and accepts payloads up to 3 % larger (valid LZ4 blocks, read by the reference's decoder)"}
            line["decode_paths"] = decode_path_comparison()
            line["decode_64GiB"] = decode_64gib()
            try:
                line["decode_64GiB_lz4_only"] = decode_64gib_lz4_only()
            except Exception as e:
                line["decode_64GiB_lz4_only"] = {"error": repr(e)[:300]}
            try:
                line["cli_wallclock"] = cli_wallclock()
            except Exception as e:                          # never lose the line over a temp-file problem
                line["cli_wallclock"] = {"error": str(e)[:200]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
