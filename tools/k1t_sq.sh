# SQ counters of the LZ4 decode kernels of path MODE (tile | seg) at FOURMC_BENCH_BLOCKS blocks (default 2048): three passes.
set -u
export TMPDIR=/tmp
NB=${FOURMC_BENCH_BLOCKS:-2048}
out=gpurun_out/r5sq_${MODE:-tile}_$NB; raw=/tmp/r5sq_${MODE:-tile}_$NB; mkdir -p $out $raw
here=$(pwd)
cd /tmp
FOURMC_DECODE=${MODE:-tile} timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $raw/sq -o sq -- python $here/tools/k1_timing.py > $here/$out/sq.log 2>&1
FOURMC_DECODE=${MODE:-tile} timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $raw/sq2 -o sq2 -- python $here/tools/k1_timing.py > $here/$out/sq2.log 2>&1
FOURMC_DECODE=${MODE:-tile} timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $raw/mem -o mem -- python $here/tools/k1_timing.py > $here/$out/mem.log 2>&1
cd $here
for p in sq sq2 mem; do db=$(find $raw/$p -name "*_results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/summary_$p.md; done
cat $out/summary_*.md | grep -i "seg\|tile\|resume" | grep -v "^| void" 
