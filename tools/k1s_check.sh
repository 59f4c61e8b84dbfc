# the segment-parallel LZ4 decoder after a change: edge streams (seg_debug), its tests, per-phase cycle counters (make sprof), the two decode legs
# (run on the GPU box from the repo root: gpurun -- bash tools/k1s_check.sh)
mkdir -p gpurun_out/r4a
timeout 300 python tools/seg_debug.py --edges > gpurun_out/r4a/seg_debug.txt 2>&1; grep -c "OK " gpurun_out/r4a/seg_debug.txt; grep "BAD" gpurun_out/r4a/seg_debug.txt | head -8 | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_lz4rows.py tests/test_gpu_fullsize.py -x -q -k "seg or lz4 or fast" > gpurun_out/r4a/pytest_seg.txt 2>&1; tail -3 gpurun_out/r4a/pytest_seg.txt
FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-sprof.so timeout 600 python tools/seg_prof.py --full 2048 > gpurun_out/r4a/seg_prof.txt 2>&1; grep -A13 "full launch" gpurun_out/r4a/seg_prof.txt | cut -c1-330
timeout 300 python tools/k1_big.py 2048 2>&1 | grep blocks; timeout 300 python tools/k1_big.py 2>&1 | grep blocks
