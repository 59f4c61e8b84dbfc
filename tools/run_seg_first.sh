mkdir -p gpurun_out/r4a
timeout 600 python tools/seg_debug.py --edges > gpurun_out/r4a/seg_debug.txt 2>&1; grep -c "OK " gpurun_out/r4a/seg_debug.txt; grep "BAD" gpurun_out/r4a/seg_debug.txt | head
timeout 900 python -m pytest tests/test_gpu_lz4rows.py -x -q -k "seg" > gpurun_out/r4a/pytest_seg.txt 2>&1; tail -3 gpurun_out/r4a/pytest_seg.txt
FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-sprof.so timeout 600 python tools/seg_prof.py --full 2048 > gpurun_out/r4a/seg_prof.txt 2>&1; cat gpurun_out/r4a/seg_prof.txt | grep -v amdgpu.ids | cut -c1-330
FOURMC_DECODE=seg FOURMC_BENCH_BLOCKS=8192 timeout 600 python tools/k1_timing.py > gpurun_out/r4a/k1_seg_8192.txt 2>&1; tail -1 gpurun_out/r4a/k1_seg_8192.txt
