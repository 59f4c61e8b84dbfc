#!/usr/bin/env python3
"""Turns what tools/profile_round.sh left under gpurun_out/<tag>/ into the files kept under profiles/:
    python tools/profile_publish.py <tag> <name>     e.g.  r04a r04_a
Round 4: the LZ4 decode path of a chip-filling launch is three kernels (walk, executor, exact walker for the tails); they are listed
one by one and summed."""
import json, os, re, sys
tag, name = sys.argv[1], sys.argv[2]
o = f"gpurun_out/{tag}/"
def rows(f, per=5):
    d = {}
    if not os.path.exists(f): return d
    for l in open(f):
        m = re.match(r"\| ([\w<>]+)[^|]* \| (\w+) \| (\d+) \| (\d+) \| (\d+) \|", l)
        if m: d.setdefault(m.group(1).split('<')[0], {})[m.group(2)] = int(m.group(per))
    return d
DEC = ("lz4_seg_walk_kernel", "lz4_seg_exec_kernel", "lz4_tile_walk_kernel", "lz4_tile_exec_kernel", "lz4_decode_resume_kernel", "lz4_decode_wx_kernel", "lz4_decode_retry_kernel")
fe, wr = rows(o + "summary_fetch.md"), rows(o + "summary_write.md")
xk = [k for k in fe if k.startswith("xxh32")][0]
dks = [k for k in DEC if k in fe and (fe[k].get("FETCH_SIZE", 0) + wr.get(k, {}).get("WRITE_SIZE", 0)) > 64]
rnd = re.match(r"(r\d+)", name).group(1)
traffic_json = f"profiles/{rnd}_traffic.json"
prev = sorted(f for f in os.listdir("profiles") if re.match(r"r\d+_traffic.json", f))
t = json.load(open(traffic_json if os.path.exists(traffic_json) else "profiles/" + prev[-1]))
t["lz4_encode"] = {"fetch_KiB": fe["lz4_encode_fast_kernel"]["FETCH_SIZE"], "write_KiB": wr["lz4_encode_fast_kernel"]["WRITE_SIZE"]}
t["lz4_decode"] = {"fetch_KiB": sum(fe[k]["FETCH_SIZE"] for k in dks), "write_KiB": sum(wr[k]["WRITE_SIZE"] for k in dks), "kernel": "+".join(dks),
                   "per_kernel": {k: {"fetch_KiB": fe[k]["FETCH_SIZE"], "write_KiB": wr[k]["WRITE_SIZE"]} for k in dks}}
t["xxh32"] = {"fetch_KiB": fe[xk]["FETCH_SIZE"], "write_KiB": wr[xk]["WRITE_SIZE"]}
t["pack"] = {"fetch_KiB": fe["pack_image_kernel"]["FETCH_SIZE"], "write_KiB": wr["pack_image_kernel"]["WRITE_SIZE"]}
json.dump(t, open(traffic_json, "w"), indent=1)
open(f"profiles/{name}_kernel_stats.md", "w").write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu   (tools/profile_round.sh)\n\n" + open(o + "summary_stats.md").read())
open(f"profiles/{name}_hbm_traffic.md", "w").write("# rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python bench.py --no-extras --no-cpu --steps 2 --warmup 1   (2048 blocks: the kernels of the headline launch)\n# per-dispatch values are KiB (see the calibration note in the round's *_traffic.json); the LZ4 decode path = " + " + ".join(dks) + "\n\n" + open(o + "summary_fetch.md").read() + "\n" + open(o + "summary_write.md").read())
open(f"profiles/{name}_bench_under_rocprof.json", "w").write(open(o + "bench_stats.json").read())
HDR = "| kernel | waves' time parked (s_waitcnt) | issuing | issue stalls | VALU busy (of 1024 SIMDs) | SALU : VALU instructions | instructions per output byte (VALU+SALU+LDS+VMEM) | LDS instr | VMEM rd / wr instr |\n|---|---|---|---|---|---|---|---|---|\n"
def sqrow(label, x, y, out_bytes):
    wc = x["SQ_WAVE_CYCLES"]; cyc = y.get("GRBM_GUI_ACTIVE", 8) / 8
    lds = y.get("SQ_INSTS_LDS", x.get("SQ_INSTS_LDS", 0))
    n = x["SQ_INSTS_VALU"] + x["SQ_INSTS_SALU"] + lds + y.get("SQ_INSTS_VMEM_RD", 0) + y.get("SQ_INSTS_VMEM_WR", 0)
    av = x.get("SQ_ACTIVE_INST_VALU", y.get("SQ_ACTIVE_INST_VALU", 0))
    return "| %s | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.2f | %.3f | %.2e | %.2e / %.2e |\n" % (label, 100 * x["SQ_WAIT_ANY"] / wc, 100 * x["SQ_ACTIVE_INST_ANY"] / wc, 100 * x["SQ_WAIT_INST_ANY"] / wc,
            100 * 4 * av / (1024 * cyc), x["SQ_INSTS_SALU"] / max(x["SQ_INSTS_VALU"], 1), n / out_bytes, lds, y.get("SQ_INSTS_VMEM_RD", 0), y.get("SQ_INSTS_VMEM_WR", 0))
a, b = rows(o + "summary_sq.md"), rows(o + "summary_sq2.md")
out = "# rocprofv3 --pmc <SQ group> -- python bench.py --no-extras --no-cpu --steps 2 --warmup 1   (2048 blocks; two passes, tools/profile_round.sh)\n# per-dispatch sums over all waves; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs\n\n" + HDR
for k in ["lz4_encode_fast_kernel"] + [k for k in DEC if k in a] + [k for k in a if k.startswith("xxh32")][:1]:
    out += sqrow(k, a[k], b.get(k, {}), 2048 * 4194304.0)
out += "\n" + open(o + "summary_sq.md").read() + "\n" + open(o + "summary_sq2.md").read()
open(f"profiles/{name}_sq_counters.md", "w").write(out)
# the LZ4 decode paths alone (tools/k1_timing.py under FOURMC_DECODE=seg / wx)
txt = "# FOURMC_DECODE=seg | tile  FOURMC_BENCH_BLOCKS=N  rocprofv3 ... -- python tools/k1_timing.py   (decode only, S-mix; per configuration: kernel trace, two SQ groups; seg at 8192 blocks also FETCH / WRITE_SIZE and TCP / TCC groups)\n"
txt += "# (tools/k1_timing.py compresses with the bound as capacity, so the incompressible blocks of the corpus are LZ4 streams here, not stored blocks as in the container: its times are not the bench's)\n\n" + HDR
for cfg, nb in (("seg2048", 2048), ("seg8192", 8192), ("tile2048", 2048), ("tile512", 512)):
    pa, pb = rows(o + f"summary_{cfg}_sq.md"), rows(o + f"summary_{cfg}_sq2.md")
    for k in DEC:
        if k in pa and pa[k]["SQ_INSTS_VALU"] > 1000: txt += sqrow(f"{cfg}: {k}", pa[k], pb.get(k, {}), nb * 4194304.0)
for cfg in ("seg2048", "seg8192", "tile2048", "tile512"):
    for sfx in ("_stats", "_sq", "_sq2", "_fetch", "_write", "_tcp", "_tcc"):
        f = o + f"summary_{cfg}{sfx}.md"
        if os.path.exists(f):
            keep = [l for l in open(f) if not l.startswith("| void at::") and "elementwise" not in l]
            txt += f"\n## {cfg}{sfx}\n\n" + "".join(keep)
    f = o + f"{cfg}_stats.log"
    if os.path.exists(f): txt += "\n```\n" + "".join(l for l in open(f) if "S-mix" in l) + "```\n"
open(f"profiles/{name}_lz4_decode_paths.md", "w").write(txt)
if os.path.exists(o + "summary_z1_stats.md"):
    open(f"profiles/{name}_4mz_fast.md", "w").write("# rocprofv3 --kernel-trace --stats -- python tools/zstd_timing.py   (FOURMC_BENCH_BLOCKS=2048: 4mz Fast encode + decode of 2048 blocks of S-mix)\n\n" + open(o + "summary_z1_stats.md").read())
d = json.load(open(o + "bench_stats.json"))
print(d["value"], d["ms_per_step"], d["kernel_ms"], d["compress_GBps"], d["decompress_GBps"])
