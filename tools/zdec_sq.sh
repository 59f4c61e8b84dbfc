export TMPDIR=/tmp
run() { name=$1; shift; FOURMC_ZHELPER=0 timeout 250 rocprofv3 "$@" -d /tmp/p_$name -o $name -- python tools/zdec_ab.py 2048 1 > /tmp/$name.log 2>&1; db=$(find /tmp/p_$name -name "*_results.db" | head -1); [ -n "$db" ] && timeout 60 python tools/rocpd_summary.py $db 2>&1 | grep -i "zstd_exec" | cut -c1-400; }
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
