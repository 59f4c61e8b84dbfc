# CEILING of the sequence-per-lane copy engine: the segment-parallel executor with every dependency switched
# off (make -C 4mc_amd/csrc nodeps: no wait for the flush before a far match, no order between the near matches of a step - the same
# loads, the same LDS assembly, the same stores, wrong bytes) against the real one, kernel times by rocprofv3 on the 64 GiB leg.
#   gpurun -- bash tools/ubench/seq_copy.sh      -> gpurun_out/r6b/seq_copy.txt
export TMPDIR=/tmp
here=$(pwd); out=$here/gpurun_out/r6b; mkdir -p $out; : > $out/seq_copy.txt
for v in "" nodeps; do
  lib=4mc_amd/lib/libhadoop-4mc${v:+-$v}.so
  for nb in 2048 16384; do
    rm -rf /tmp/sc; cd /tmp; FOURMC_LIB=$here/$lib FOURMC_DECODE=seg timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/sc -o sc -- python $here/tools/k1_big.py $nb > /tmp/sc.log 2>&1; cd $here
    echo "== ${v:-real} $nb blocks" >> $out/seq_copy.txt; grep blocks /tmp/sc.log >> $out/seq_copy.txt
    db=$(find /tmp/sc -name "*_results.db" | head -1); python tools/rocpd_summary.py $db | grep -i "walk\|exec\|resume\|xxh" | cut -c1-140 >> $out/seq_copy.txt
  done
done
cat $out/seq_copy.txt
