#!/usr/bin/env python3
"""gpurun_out/enc_sq/summary_*.md (tools/enc_sq.sh) -> profiles/<name>.md      python tools/enc_publish.py r04_encoders"""
import os, re, sys
name = sys.argv[1]; o = "gpurun_out/enc_sq/"
def rows(f, per=5):
    d = {}
    for l in open(f) if os.path.exists(f) else []:
        m = re.match(r"\| ([\w<>]+)[^|]* \| (\w+) \| (\d+) \| (\d+) \| (\d+) \|", l)
        if m: d.setdefault(m.group(1), {})[m.group(2)] = int(m.group(per))
    return d
txt = "# rocprofv3 (--kernel-trace --stats | --pmc <one group per pass>) -- python tools/enc_time.py <hc4|z1|z3> 2048     (tools/enc_sq.sh)\n"
txt += "# the encoders the headline does not time, 2048 blocks of the S-mix (8 GiB) in one launch; per-dispatch sums over all waves, quad-cycles for SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*;\n# FETCH_SIZE / WRITE_SIZE in KiB (profiles/r04_traffic.json has the calibration)\n\n"
txt += "| kernel | ms | waves' time waiting (s_waitcnt) | issuing | issue stalls | instructions per input byte: VALU | SALU | LDS | VMEM rd + wr | HBM-side read / written (GB; input 8.59) |\n|---|---|---|---|---|---|---|---|---|---|\n"
inb = 2048 * 4194304.0
for what, kern in (("hc4", "lz4hc_encode_kernel"), ("z1", "zstd_encode_fast_kernel"), ("z3", "zstd_encode_dfast_kernel")):
    c = {}
    for f in ("sq", "sq2", "fetch", "write"):
        c.update(rows(o + f"summary_{what}_{f}.md").get(kern, {}))
    st = rows(o + f"summary_{what}_stats.md", 4)
    ms = None
    for l in open(o + f"summary_{what}_stats.md"):
        m = re.match(r"\| %s \| (\d+) \| ([\d.]+) \| ([\d.]+) \|" % kern, l)
        if m: ms = float(m.group(3))
    if not c: continue
    wc = c["SQ_WAVE_CYCLES"]
    txt += "| %s | %s | %.0f %% | %.0f %% | %.0f %% | %.2f | %.2f | %.3f | %.3f | %.1f / %.1f |\n" % (kern, ms, 100.0 * c["SQ_WAIT_ANY"] / wc, 100.0 * c["SQ_ACTIVE_INST_ANY"] / wc,
        100.0 * c["SQ_WAIT_INST_ANY"] / wc, c["SQ_INSTS_VALU"] / inb, c["SQ_INSTS_SALU"] / inb, c["SQ_INSTS_LDS"] / inb, (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / inb,
        c.get("FETCH_SIZE", 0) * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9)
txt += "\n"
for what in ("hc4", "z1", "z3"):
    for f in ("stats", "sq", "sq2", "fetch", "write"):
        p = o + f"summary_{what}_{f}.md"
        if os.path.exists(p):
            keep = [l for l in open(p) if "at::native" not in l and "elementwise" not in l and not l.startswith("| void")]
            txt += f"## {what}_{f}\n\n" + "".join(keep) + "\n"
open(f"profiles/{name}.md", "w").write(txt)
print(txt[:2500])
