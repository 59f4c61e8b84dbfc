set -u
export TMPDIR=/tmp
out=gpurun_out/r3sq; raw=/tmp/r3sq; mkdir -p $out $raw
for mode in rows trio; do
FOURMC_DECODE=$mode timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $raw/${mode}_sq -o sq -- python tools/k1_timing.py > $out/${mode}_sq.log 2>&1
FOURMC_DECODE=$mode timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $raw/${mode}_sq2 -o sq2 -- python tools/k1_timing.py > $out/${mode}_sq2.log 2>&1
for p in sq sq2; do db=$(find $raw/${mode}_$p -name "*_results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/summary_${mode}_$p.md; done
done
cat $out/summary_*.md | grep -i "decode\|kernel\|---" | head -60
