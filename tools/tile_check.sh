# the tile LZ4 decoder after a change: edge streams (tile_debug), its tests, per-phase counters, launch sizes
# (run on the GPU box from the repo root: gpurun -- bash tools/tile_check.sh)
mkdir -p gpurun_out/r5
timeout 300 python tools/tile_debug.py --edges > gpurun_out/r5/tile_debug.txt 2>&1; grep -c "OK " gpurun_out/r5/tile_debug.txt; grep "BAD" gpurun_out/r5/tile_debug.txt | head -12 | cut -c1-250; tail -1 gpurun_out/r5/tile_debug.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_lz4rows.py tests/test_gpu_fullsize.py -x -q -k "tile" > gpurun_out/r5/pytest_tile.txt 2>&1; tail -5 gpurun_out/r5/pytest_tile.txt
FOURMC_LIB=4mc_amd/lib/libhadoop-4mc-tprof.so timeout 600 python tools/tile_prof.py --noalone --full 2048 > gpurun_out/r5/tile_prof.txt 2>&1; cat gpurun_out/r5/tile_prof.txt | cut -c1-330
for nb in 128 512 1024 2048 16384; do FOURMC_DECODE=tile timeout 300 python tools/k1_big.py $nb 2>&1 | grep blocks | cut -c1-100; done
for nb in 128 1024; do FOURMC_TILE_WALK=separate FOURMC_DECODE=tile timeout 300 python tools/k1_big.py $nb 2>&1 | grep blocks | cut -c1-100; done
