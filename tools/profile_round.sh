#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/{stats,fetch,write,sq}/ + gpurun_out/<tag>/summary_*.md
# Kernel trace + stats in one pass; every PMC group in a pass of its own (no trace domains next to --pmc).
set -u
tag=${1:-prof}; out=gpurun_out/$tag; raw=/tmp/prof_$tag; mkdir -p $out $raw     # raw rocprofv3 output stays on the box (only <= 64 MiB come back)
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 "$@" -d $raw/$name -o $name -- python bench.py --steps 2 --warmup 1 ${BENCH_ARGS:-} > $out/$name.log 2>&1; }
BENCH_ARGS="--no-cpu" run stats --kernel-trace --stats
BENCH_ARGS="--no-extras --no-cpu" run fetch --pmc FETCH_SIZE
BENCH_ARGS="--no-extras --no-cpu" run write --pmc WRITE_SIZE
BENCH_ARGS="--no-extras --no-cpu" run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
BENCH_ARGS="--no-extras --no-cpu" run sq2 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
for p in stats fetch write sq sq2; do
    db=$(find $raw/$p -name "*_results.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/summary_$p.md
    grep "^{\"metric\"" $out/$p.log | tail -1 > $out/bench_$p.json
done
# the row-parallel LZ4 decode pipeline (lz4_rows.hip; default up to 1536 blocks per launch): kernel stats and SQ counters of the
# decode-only timing tool at 2048 blocks (a full chip) and 256 blocks (what the file API sends), the wave trio beside it
for mode in wx rows trio lanes; do
  FOURMC_DECODE=$mode rocprofv3 --kernel-trace --stats -d $raw/${mode}_stats -o ${mode}_stats -- python tools/k1_timing.py > $out/${mode}_stats.log 2>&1
  FOURMC_BENCH_BLOCKS=256 FOURMC_DECODE=$mode rocprofv3 --kernel-trace --stats -d $raw/${mode}256_stats -o ${mode}256_stats -- python tools/k1_timing.py > $out/${mode}256_stats.log 2>&1
  FOURMC_DECODE=$mode rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $raw/${mode}_sq -o ${mode}_sq -- python tools/k1_timing.py > $out/${mode}_sq.log 2>&1
  FOURMC_DECODE=$mode rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $raw/${mode}_sq2 -o ${mode}_sq2 -- python tools/k1_timing.py > $out/${mode}_sq2.log 2>&1
  for p in ${mode}_stats ${mode}256_stats ${mode}_sq ${mode}_sq2; do
    db=$(find $raw/$p -name "*_results.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/summary_$p.md
  done
done
# 4mz Fast (zstd level 1: zstd_encode_fast_kernel, zstd_decode_kernel): kernel stats and SQ counters of tools/zstd_timing.py at 2048 blocks
FOURMC_BENCH_BLOCKS=2048 rocprofv3 --kernel-trace --stats -d $raw/z1_stats -o z1_stats -- python tools/zstd_timing.py > $out/z1_stats.log 2>&1
FOURMC_BENCH_BLOCKS=2048 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $raw/z1_sq -o z1_sq -- python tools/zstd_timing.py > $out/z1_sq.log 2>&1
FOURMC_BENCH_BLOCKS=2048 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $raw/z1_sq2 -o z1_sq2 -- python tools/zstd_timing.py > $out/z1_sq2.log 2>&1
for p in z1_stats z1_sq z1_sq2; do
    db=$(find $raw/$p -name "*_results.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/summary_$p.md
done
