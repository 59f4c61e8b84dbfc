# kernel times and SQ / memory counters of the ratio-tolerance LZ4 encoder (lz4_par_encode.hip) at 2048 blocks.
set -u
export TMPDIR=/tmp
out=gpurun_out/k2p_sq; raw=/tmp/k2p_sq; mkdir -p $out $raw
here=$(pwd)
cd /tmp
cmd="python $here/tools/k2p_debug.py --no-edge --blocks 2048"
timeout 300 rocprofv3 --kernel-trace --stats -d $raw/st -o st -- $cmd > $here/$out/st.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $raw/sq -o sq -- $cmd > $here/$out/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $raw/sq2 -o sq2 -- $cmd > $here/$out/sq2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $raw/fetch -o fetch -- $cmd > $here/$out/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $raw/write -o write -- $cmd > $here/$out/write.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY -d $raw/tcp -o tcp -- $cmd > $here/$out/tcp.log 2>&1
cd $here
for p in st sq sq2 fetch write tcp; do db=$(find $raw/$p -name "*_results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/summary_$p.md; done
cat $out/summary_*.md | grep -i "lz4_par" | grep -v "^| void"
