#!/usr/bin/env python3
"""gpurun_out/k2p_sq/summary_*.md (tools/k2p_sq.sh) -> profiles/<name>.md: the ratio-tolerance LZ4 encoder's kernels at 2048 blocks.
    python tools/k2p_publish.py r04_k2p"""
import os, re, sys
name = sys.argv[1]; o = "gpurun_out/k2p_sq/"
def rows(f, per=5):
    d = {}
    for l in open(f) if os.path.exists(f) else []:
        m = re.match(r"\| ([\w<>]+)[^|]* \| (\w+) \| (\d+) \| (\d+) \| (\d+) \|", l)
        if m: d.setdefault(m.group(1), {})[m.group(2)] = int(m.group(per))
    return d
c = {}
for f in ("sq", "sq2", "fetch", "write", "tcp"):
    for k, v in rows(o + f"summary_{f}.md").items(): c.setdefault(k, {}).update(v)
K = "lz4_par_segment_kernel"; S = "lz4_par_stitch_kernel"
x = c[K]; win = 2048 * 65536.0; inb = 2048 * 4194304.0
txt = "# rocprofv3 (--kernel-trace --stats | --pmc <one group per pass>) -- python tools/k2p_debug.py --no-edge --blocks 2048     (tools/k2p_sq.sh)\n"
txt += "# the ratio-tolerance LZ4 encoder (4mc_amd/csrc/lz4_par_encode.hip) on 2048 blocks of the S-mix; the run also times the exact encoder (lz4_encode_fast_kernel) for comparison\n"
txt += "# per-dispatch sums over all waves; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; FETCH_SIZE / WRITE_SIZE in KiB (see profiles/r04_traffic.json for the calibration)\n\n"
txt += "## derived, %s (a window = 64 input positions; %d windows per launch)\n\n" % (K, win)
wc = x["SQ_WAVE_CYCLES"]
txt += "| per window | VALU | SALU | LDS | VMEM rd | VMEM wr | wave-cycles | of which waiting (s_waitcnt) | issuing | issue stalls | L1 accesses | L1 -> L2 read requests |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n"
txt += "| | %.0f | %.0f | %.1f | %.1f | %.2f | %.0f | %.0f %% | %.0f %% | %.0f %% | %.0f | %.1f |\n\n" % (
    x["SQ_INSTS_VALU"] / win, x["SQ_INSTS_SALU"] / win, x["SQ_INSTS_LDS"] / win, x["SQ_INSTS_VMEM_RD"] / win, x["SQ_INSTS_VMEM_WR"] / win, 4.0 * wc / win,
    100.0 * x["SQ_WAIT_ANY"] / wc, 100.0 * x["SQ_ACTIVE_INST_ANY"] / wc, 100.0 * x["SQ_WAIT_INST_ANY"] / wc,
    x.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / win, x.get("TCP_TCC_READ_REQ_sum", 0) / win)
for k in (K, S):
    if "FETCH_SIZE" in c.get(k, {}):
        txt += "%s: HBM-side traffic %.2f GB read + %.2f GB written per launch (input %.2f GB)\n" % (k, c[k]["FETCH_SIZE"] * 1024 / 1e9, c[k].get("WRITE_SIZE", 0) * 1024 / 1e9, inb / 1e9)
txt += "\n"
for f in ("st", "sq", "sq2", "fetch", "write", "tcp"):
    p = o + f"summary_{f}.md"
    if os.path.exists(p):
        keep = [l for l in open(p) if not l.startswith("| void at::") and "elementwise" not in l and "at::native" not in l]
        txt += f"## {f}\n\n" + "".join(keep) + "\n"
open(f"profiles/{name}.md", "w").write(txt)
import json
json.dump({"_source": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, tools/k2p_sq.sh) at 2048 blocks; KiB per dispatch, calibration in profiles/r04_traffic.json",
           "blocks": 2048,
           K: {"fetch_KiB": c[K].get("FETCH_SIZE"), "write_KiB": c[K].get("WRITE_SIZE")},
           S: {"fetch_KiB": c.get(S, {}).get("FETCH_SIZE"), "write_KiB": c.get(S, {}).get("WRITE_SIZE")}}, open(f"profiles/{name}_traffic.json", "w"), indent=1)
print(txt[:1800])
