#!/usr/bin/env python3
"""Turns what tools/profile_round.sh left under gpurun_out/<tag>/ into the files kept under profiles/:
    python tools/profile_publish.py <tag> <name>     e.g.  r01e6 r01_end3"""
import json, os, re, sys
tag, name = sys.argv[1], sys.argv[2]
o = f"gpurun_out/{tag}/"
def rows(f, per=5):
    d = {}
    for l in open(f):
        m = re.match(r"\| ([\w<>]+)[^|]* \| (\w+) \| (\d+) \| (\d+) \| (\d+) \|", l)
        if m: d.setdefault(m.group(1), {})[m.group(2)] = int(m.group(per))
    return d
fe, wr = rows(o + "summary_fetch.md"), rows(o + "summary_write.md")
xk = [k for k in fe if k.startswith("xxh32")][0]
dk = [k for k in ("lz4_decode_wx_kernel", "lz4_decode_rows_kernel", "lz4_decode_fast_kernel") if k in fe][0]     # the default decode path of the bench
rnd = re.match(r"(r\d+)", name).group(1)
traffic_json = f"profiles/{rnd}_traffic.json"
t = json.load(open(traffic_json if os.path.exists(traffic_json) else "profiles/r01_traffic.json"))
t["lz4_encode"] = {"fetch_KiB": fe["lz4_encode_fast_kernel"]["FETCH_SIZE"], "write_KiB": wr["lz4_encode_fast_kernel"]["WRITE_SIZE"]}
t["lz4_decode"] = {"fetch_KiB": fe[dk]["FETCH_SIZE"] + fe.get("lz4_decode_retry_kernel", {}).get("FETCH_SIZE", 0),
                   "write_KiB": wr[dk]["WRITE_SIZE"], "kernel": dk}
t["xxh32"] = {"fetch_KiB": fe[xk]["FETCH_SIZE"], "write_KiB": wr[xk]["WRITE_SIZE"]}
t["pack"] = {"fetch_KiB": fe["pack_image_kernel"]["FETCH_SIZE"], "write_KiB": wr["pack_image_kernel"]["WRITE_SIZE"]}
json.dump(t, open(traffic_json, "w"), indent=1)
open(f"profiles/{name}_kernel_stats.md", "w").write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1   (tools/profile_round.sh)\n\n" + open(o + "summary_stats.md").read())
open(f"profiles/{name}_hbm_traffic.md", "w").write("# rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python bench.py --no-extras --no-cpu --steps 2 --warmup 1   (2048 blocks: the kernels of the headline launch)\n# per-dispatch values are KiB (see the calibration note in the round's *_traffic.json)\n\n" + open(o + "summary_fetch.md").read() + "\n" + open(o + "summary_write.md").read())
open(f"profiles/{name}_bench_under_rocprof.json", "w").write(open(o + "bench_stats.json").read())
open(f"profiles/{name}_bench.json", "w").write(open(o + "bench_stats.json").read())
a, b = rows(o + "summary_sq.md"), rows(o + "summary_sq2.md")
out = "# rocprofv3 --pmc <SQ group> -- python bench.py --no-extras --no-cpu --steps 2 --warmup 1   (2048 blocks; two passes, tools/profile_round.sh)\n# per-dispatch sums over all waves; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs\n\n"
out += "| kernel | waves' time parked (s_waitcnt) | issuing | issue stalls | VALU busy (of 1024 SIMDs) | SALU : VALU instructions | cycles per issued instruction | LDS instr | VMEM rd / wr instr |\n|---|---|---|---|---|---|---|---|---|\n"
for k in ("lz4_encode_fast_kernel", dk, [k for k in a if k.startswith("xxh32")][0]):
    x, y = a[k], b[k]; wc = x["SQ_WAVE_CYCLES"]; cyc = y["GRBM_GUI_ACTIVE"] / 8
    n = x["SQ_INSTS_VALU"] + x["SQ_INSTS_SALU"] + y["SQ_INSTS_LDS"] + y["SQ_INSTS_VMEM_RD"] + y["SQ_INSTS_VMEM_WR"]
    out += "| %s | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.2f | %.1f | %.2e | %.2e / %.2e |\n" % (k, 100 * x["SQ_WAIT_ANY"] / wc, 100 * x["SQ_ACTIVE_INST_ANY"] / wc, 100 * x["SQ_WAIT_INST_ANY"] / wc, 100 * 4 * x["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), x["SQ_INSTS_SALU"] / x["SQ_INSTS_VALU"], 4 * x["SQ_ACTIVE_INST_ANY"] / n, y["SQ_INSTS_LDS"], y["SQ_INSTS_VMEM_RD"], y["SQ_INSTS_VMEM_WR"])
out += "\n" + open(o + "summary_sq.md").read() + "\n" + open(o + "summary_sq2.md").read()
open(f"profiles/{name}_sq_counters.md", "w").write(out)
# the LZ4 decode paths side by side (tools/k1_timing.py under FOURMC_DECODE=rows / trio)
if os.path.exists(o + "summary_rows_stats.md") or os.path.exists(o + "summary_wx_stats.md"):
    txt = "# FOURMC_DECODE=wx | rows | trio | lanes  rocprofv3 ... -- python tools/k1_timing.py   (2048 and 256 blocks of S-mix, decode only; per mode: two kernel traces, two SQ groups)\n\n"
    txt += "| path | kernel | waves' time parked (s_waitcnt) | issuing | issue stalls | VALU busy (of 1024 SIMDs) | SALU : VALU | instructions per output byte (VALU+SALU+LDS+VMEM) | LDS instr | LDS busy (of 256 CUs) | LDS bank-conflict / LDS active |\n|---|---|---|---|---|---|---|---|---|---|---|\n"
    for mode, kern in (("wx", "lz4_decode_wx_kernel"), ("rows", "lz4_decode_rows_kernel"), ("trio", "lz4_decode_fast_kernel"), ("lanes", "lz4_decode_lanes_kernel")):
        pa, pb = rows(o + f"summary_{mode}_sq.md"), rows(o + f"summary_{mode}_sq2.md")
        if kern not in pa: continue
        x, y = pa[kern], pb.get(kern, {})
        wc = x["SQ_WAVE_CYCLES"]; cyc = y.get("GRBM_GUI_ACTIVE", 8) / 8
        n = x["SQ_INSTS_VALU"] + x["SQ_INSTS_SALU"] + x["SQ_INSTS_LDS"] + y.get("SQ_INSTS_VMEM_RD", 0) + y.get("SQ_INSTS_VMEM_WR", 0)
        txt += "| %s | %s | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.2f | %.2f | %.2e | %.0f %% | %.2f |\n" % (mode, kern, 100 * x["SQ_WAIT_ANY"] / wc, 100 * x["SQ_ACTIVE_INST_ANY"] / wc, 100 * x["SQ_WAIT_INST_ANY"] / wc,
                100 * 4 * y.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * cyc), x["SQ_INSTS_SALU"] / x["SQ_INSTS_VALU"], n / (2048 * 4194304.0), x["SQ_INSTS_LDS"],
                100 * 4 * y.get("SQ_ACTIVE_INST_LDS", 0) / (256 * cyc), y.get("SQ_LDS_BANK_CONFLICT", 0) / max(y.get("SQ_LDS_IDX_ACTIVE", 1), 1))
    for mode in ("wx", "rows", "trio", "lanes"):
        for sfx in ("_stats", "256_stats", "_sq", "_sq2"):
            f = o + f"summary_{mode}{sfx}.md"
            if os.path.exists(f): txt += f"\n## {mode}{sfx}\n\n" + open(f).read()
        for sfx in ("_stats", "256_stats"):
            f = o + f"{mode}{sfx}.log"
            if os.path.exists(f): txt += "\n```\n" + "".join(l for l in open(f) if "S-mix" in l) + "```\n"
    open(f"profiles/{name}_lz4_decode_paths.md", "w").write(txt)
# 4mz Fast: tools/zstd_timing.py at 2048 blocks
if os.path.exists(o + "summary_z1_stats.md"):
    za, zb = rows(o + "summary_z1_sq.md"), rows(o + "summary_z1_sq2.md")
    txt = "# rocprofv3 ... -- python tools/zstd_timing.py   (FOURMC_BENCH_BLOCKS=2048: 4mz Fast encode + decode of 2048 blocks of S-mix; three passes: kernel trace, two SQ groups)\n\n"
    txt += open(o + "summary_z1_stats.md").read() + "\n"
    txt += "| kernel | waves' time parked (s_waitcnt) | issuing | issue stalls | VALU busy (of 1024 SIMDs) | SALU : VALU instructions | cycles per issued instruction | LDS instr | VMEM rd / wr instr |\n|---|---|---|---|---|---|---|---|---|\n"
    for k in za:
        if not k.startswith("zstd_"): continue
        x, y = za[k], zb.get(k, {}); wc = x["SQ_WAVE_CYCLES"]; cyc = y.get("GRBM_GUI_ACTIVE", 8) / 8
        n = x["SQ_INSTS_VALU"] + x["SQ_INSTS_SALU"] + y.get("SQ_INSTS_LDS", 0) + y.get("SQ_INSTS_VMEM_RD", 0) + y.get("SQ_INSTS_VMEM_WR", 0)
        txt += "| %s | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.2f | %.1f | %.2e | %.2e / %.2e |\n" % (k, 100 * x["SQ_WAIT_ANY"] / wc, 100 * x["SQ_ACTIVE_INST_ANY"] / wc, 100 * x["SQ_WAIT_INST_ANY"] / wc, 100 * 4 * x["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), x["SQ_INSTS_SALU"] / x["SQ_INSTS_VALU"], 4 * x["SQ_ACTIVE_INST_ANY"] / n, y.get("SQ_INSTS_LDS", 0), y.get("SQ_INSTS_VMEM_RD", 0), y.get("SQ_INSTS_VMEM_WR", 0))
    txt += "\n" + open(o + "summary_z1_sq.md").read() + "\n" + open(o + "summary_z1_sq2.md").read()
    open(f"profiles/{name}_4mz_fast.md", "w").write(txt)
d = json.load(open(o + "bench_stats.json"))
print(d["value"], d["ms_per_step"], d["kernel_ms"], d["compress_GBps"], d["decompress_GBps"])
