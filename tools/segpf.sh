for v in "" segnopf; do lib=4mc_amd/lib/libhadoop-4mc${v:+-$v}.so; echo "== $v"; FOURMC_LIB=$PWD/$lib FOURMC_DECODE=seg timeout 300 python tools/k1_big.py 2048 2>&1 | grep blocks; FOURMC_LIB=$PWD/$lib FOURMC_DECODE=seg timeout 300 python tools/k1_big.py 2>&1 | grep blocks; done
timeout 900 python -m pytest tests/test_gpu_lz4rows.py -x -q -k "seg" 2>&1 | tail -2
