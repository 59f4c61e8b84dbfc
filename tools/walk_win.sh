# walk kernels with smaller per-lane stream windows (more waves per CU): kernel times of the 64 GiB leg per variant library
export TMPDIR=/tmp
here=$(pwd); mkdir -p gpurun_out/r5
for v in "" segring32 segring16; do
  lib=4mc_amd/lib/libhadoop-4mc${v:+-$v}.so
  rm -rf /tmp/ww; cd /tmp; FOURMC_LIB=$here/$lib FOURMC_DECODE=seg timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ww -o ww -- python $here/tools/k1_big.py ${NB:-16384} > /tmp/ww.log 2>&1; cd $here
  grep blocks /tmp/ww.log; db=$(find /tmp/ww -name "*_results.db" | head -1); echo "== seg ${v:-default}"; python tools/rocpd_summary.py $db | grep -i "walk\|exec\|resume" | cut -c1-120
done
for v in "" tilewin32 tilewin16; do
  lib=4mc_amd/lib/libhadoop-4mc${v:+-$v}.so
  rm -rf /tmp/ww; cd /tmp; FOURMC_LIB=$here/$lib FOURMC_DECODE=tile timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ww -o ww -- python $here/tools/k1_big.py ${NB:-16384} > /tmp/ww.log 2>&1; cd $here
  grep blocks /tmp/ww.log; db=$(find /tmp/ww -name "*_results.db" | head -1); echo "== tile ${v:-default}"; python tools/rocpd_summary.py $db | grep -i "walk\|exec\|resume" | cut -c1-120
done
