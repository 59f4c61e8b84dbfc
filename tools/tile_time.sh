# tile LZ4 decoder: per-phase counters (make tprof) alone and in a full launch, kernel times of the 2048-block launch (rocprofv3), the two legs
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-tprof.so timeout 600 python tools/tile_prof.py --full 2048 > gpurun_out/r5/tile_prof.txt 2>&1; cat gpurun_out/r5/tile_prof.txt | cut -c1-300
here=$(pwd); cd /tmp
FOURMC_DECODE=tile timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tk -o tk -- python $here/tools/k1_big.py 2048 > $here/gpurun_out/r5/tk.log 2>&1
cd $here; f=$(find /tmp/tk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-6 $f | head -12 | cut -c1-200
for m in tile; do FOURMC_DECODE=$m timeout 300 python tools/k1_big.py 2048 2>&1 | grep blocks; FOURMC_DECODE=$m timeout 300 python tools/k1_big.py 2>&1 | grep blocks; done
