#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/{stats,fetch,write,sq}/ + gpurun_out/<tag>/summary_*.md
# Kernel trace + stats in one pass; every PMC group in a pass of its own (no trace domains next to --pmc).
set -u
tag=${1:-prof}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 "$@" -d $out/$name -o $name -- python bench.py --steps 2 --warmup 1 ${BENCH_ARGS:-} > $out/$name.log 2>&1; }
run stats --kernel-trace --stats
BENCH_ARGS="--blocks 512 --no-extras --no-cpu" run fetch --pmc FETCH_SIZE
BENCH_ARGS="--blocks 512 --no-extras --no-cpu" run write --pmc WRITE_SIZE
BENCH_ARGS="--no-extras --no-cpu" run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
BENCH_ARGS="--no-extras --no-cpu" run sq2 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
for p in stats fetch write sq sq2; do
    db=$(find $out/$p -name "*_results.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/summary_$p.md
    grep "^{\"metric\"" $out/$p.log | tail -1 > $out/bench_$p.json
done
