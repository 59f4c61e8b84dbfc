mkdir -p gpurun_out/r4a
timeout 600 python tools/seg_debug.py --edges > gpurun_out/r4a/seg_debug.txt 2>&1; grep -c "OK " gpurun_out/r4a/seg_debug.txt; grep "BAD" gpurun_out/r4a/seg_debug.txt | head
FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-sprof.so timeout 600 python tools/seg_prof.py --full 2048 > gpurun_out/r4a/seg_prof.txt 2>&1; grep -A13 "full launch" gpurun_out/r4a/seg_prof.txt | cut -c1-330
python tools/k1_big.py 2048 2>&1 | grep blocks; python tools/k1_big.py 2>&1 | grep blocks
