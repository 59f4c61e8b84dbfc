export TMPDIR=/tmp
run() { name=$1; shift; FOURMC_DECODE=wx timeout 250 rocprofv3 "$@" -d /tmp/p_$name -o $name -- python tools/k1_timing.py > /tmp/$name.log 2>&1; db=$(find /tmp/p_$name -name "*_results.db" | head -1); [ -n "$db" ] && timeout 60 python tools/rocpd_summary.py $db 2>&1 | grep -i "lz4_decode_wx" | cut -c1-200; }
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
