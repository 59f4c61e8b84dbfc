#!/bin/bash
# kernel stats and SQ counters of the lazy zstd levels (6: S-mix, 12: log corpus) at 1024 blocks; run on the GPU box from the repo root
export TMPDIR=/tmp
out=gpurun_out/zlazy; mkdir -p $out
run() { name=$1; lvl=$2; corpus=$3; shift 3; timeout 600 rocprofv3 "$@" -d /tmp/p_$name -o $name -- python tools/zdec_ab.py 1024 $lvl $corpus > $out/$name.log 2>&1; db=$(find /tmp/p_$name -name "*_results.db" | head -1); [ -n "$db" ] && timeout 60 python tools/rocpd_summary.py $db > $out/summary_$name.md 2>&1; }
for cfg in "6 smix" "12 logs"; do set -- $cfg
  run l$1_stats $1 $2 --kernel-trace --stats
  run l$1_sq $1 $2 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
  run l$1_sq2 $1 $2 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
done
grep -h "zstd_encode" $out/summary_*.md | cut -c1-200
