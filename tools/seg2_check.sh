# the pipelined executor of the segment-parallel path (lz4_seg.hip: lz4_seg_exec2_kernel) after a change: parity tests, fuzz, the decode legs
# against the first executor (FOURMC_SEG_EXEC=1)          gpurun -- bash tools/seg2_check.sh
mkdir -p gpurun_out/r6c
timeout 900 python -m pytest tests/test_gpu_lz4rows.py -x -q -k "seg" > gpurun_out/r6c/pytest_seg2.txt 2>&1; tail -5 gpurun_out/r6c/pytest_seg2.txt
FOURMC_DECODE=seg timeout 300 python tools/fuzz_decode.py 1 60 > gpurun_out/r6c/fuzz_seg2.txt 2>&1; tail -3 gpurun_out/r6c/fuzz_seg2.txt
FOURMC_DECODE=segonly timeout 300 python tools/fuzz_decode.py 100 40 > gpurun_out/r6c/fuzz_seg2only.txt 2>&1; tail -3 gpurun_out/r6c/fuzz_seg2only.txt
for x in 2 1; do FOURMC_SEG_EXEC=$x FOURMC_DECODE=seg timeout 300 python tools/k1_big.py 2048 2>&1 | grep blocks; done
for x in 2 1; do FOURMC_SEG_EXEC=$x FOURMC_DECODE=seg timeout 300 python tools/k1_big.py 2>&1 | grep blocks; done
