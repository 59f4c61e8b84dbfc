#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/summary_*.md + bench_*.json
# Kernel trace + stats in one pass; every PMC group in a pass of its own (no trace domains next to --pmc).
# The decode path of a launch is several kernels (segment-parallel: lz4_seg_walk_kernel, lz4_seg_exec_kernel; tile: lz4_tile_exec_kernel with
# the walk fused in; both: lz4_decode_resume_kernel); tools/profile_publish.py adds them up.
set -u
tag=${1:-prof}; out=gpurun_out/$tag; raw=/tmp/prof_$tag; mkdir -p $out $raw     # raw rocprofv3 output stays on the box (only <= 64 MiB come back)
export TMPDIR=/tmp
here=$(pwd)
sum() { db=$(find $raw/$1 -name "*_results.db" | head -1); [ -n "$db" ] && python $here/tools/rocpd_summary.py $db > $here/$out/summary_$1.md; }
run() { name=$1; shift; (cd /tmp && rocprofv3 "$@" -d $raw/$name -o $name -- python $here/bench.py --steps 2 --warmup 1 ${BENCH_ARGS:-} > $here/$out/$name.log 2>&1); sum $name; grep "^{\"metric\"" $out/$name.log | tail -1 > $out/bench_$name.json; }
BENCH_ARGS="--no-cpu" run stats --kernel-trace --stats
BENCH_ARGS="--no-extras --no-cpu" run fetch --pmc FETCH_SIZE
BENCH_ARGS="--no-extras --no-cpu" run write --pmc WRITE_SIZE
BENCH_ARGS="--no-extras --no-cpu" run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
BENCH_ARGS="--no-extras --no-cpu" run sq2 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
# the LZ4 decode paths alone (tools/k1_timing.py, decode only): the segment-parallel path at 2048 and at 8192 blocks (one workspace-full:
# the regime of the 64 GiB leg), the tile path at 2048 and at 512 blocks (what the file API sends: auto takes it up to 1536)
k1() { name=$1; mode=$2; nb=$3; shift 3; (cd /tmp && FOURMC_DECODE=$mode FOURMC_BENCH_BLOCKS=$nb rocprofv3 "$@" -d $raw/$name -o $name -- python $here/tools/k1_timing.py > $here/$out/$name.log 2>&1); sum $name; }
for cfg in "seg 2048" "seg 8192" "tile 2048" "tile 512"; do
  set -- $cfg; m=$1; n=$2
  k1 ${m}${n}_stats $m $n --kernel-trace --stats
  k1 ${m}${n}_sq $m $n --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
  k1 ${m}${n}_sq2 $m $n --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
done
k1 seg8192_fetch seg 8192 --pmc FETCH_SIZE
k1 seg8192_write seg 8192 --pmc WRITE_SIZE
k1 seg8192_tcp seg 8192 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
k1 seg8192_tcc seg 8192 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
# 4mz Fast (zstd level 1): kernel stats of tools/zstd_timing.py at 2048 blocks
(cd /tmp && FOURMC_BENCH_BLOCKS=2048 rocprofv3 --kernel-trace --stats -d $raw/z1_stats -o z1_stats -- python $here/tools/zstd_timing.py > $here/$out/z1_stats.log 2>&1); sum z1_stats
ls $out
