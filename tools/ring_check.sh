# the group executor (lz4_ring.hip) after a change: its parity tests, the fuzz, the two decode legs against the segment-parallel path
# (run on the GPU box from the repo root: gpurun -- bash tools/ring_check.sh)
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_lz4rows.py -x -q -k "ring" > gpurun_out/r6a/pytest_ring.txt 2>&1; tail -5 gpurun_out/r6a/pytest_ring.txt
FOURMC_DECODE=ring timeout 300 python tools/fuzz_decode.py 1 40 > gpurun_out/r6a/fuzz_ring.txt 2>&1; tail -3 gpurun_out/r6a/fuzz_ring.txt
for m in ring seg; do FOURMC_DECODE=$m timeout 300 python tools/k1_big.py 2048 2>&1 | grep blocks; done
for m in ring seg; do FOURMC_DECODE=$m timeout 300 python tools/k1_big.py 2>&1 | grep blocks; done
