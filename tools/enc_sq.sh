#!/bin/bash
# kernel stats and SQ counters of the encoders the headline does not time: LZ4 HC level 4 (K3), zstd 1 and 3 (K6), 2048 blocks of the S-mix.
# run on the GPU box from the repo root; -> gpurun_out/enc_sq/summary_*.md (tools/enc_publish.py puts them under profiles/)
export TMPDIR=/tmp
out=gpurun_out/enc_sq; mkdir -p $out; here=$(pwd)
run() { name=$1; what=$2; shift 2; (cd /tmp && timeout 600 rocprofv3 "$@" -d /tmp/p_$name -o $name -- python $here/tools/enc_time.py $what 2048 > $here/$out/$name.log 2>&1); db=$(find /tmp/p_$name -name "*_results.db" | head -1); [ -n "$db" ] && timeout 60 python tools/rocpd_summary.py $db > $out/summary_$name.md 2>&1; }
for what in hc4 z1 z3; do
  run ${what}_stats $what --kernel-trace --stats
  run ${what}_sq $what --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
  run ${what}_sq2 $what --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
  run ${what}_fetch $what --pmc FETCH_SIZE
  run ${what}_write $what --pmc WRITE_SIZE
done
grep -h "lz4hc_encode\|zstd_encode" $out/summary_*_stats.md | cut -c1-120
