export FOURMC_LIB=$PWD/4mc_amd/lib/libhadoop-4mc-zlprof.so
python tools/zenc_time.py 12 12 2>&1 | grep -E "ZLPROF|blocks" | sort -u | head -30
python tools/zenc_time.py 6 12 2>&1 | grep -E "ZLPROF|blocks" | sort -u | head -30
python tools/zenc_time.py 12 4 logs 2>&1 | grep -E "ZLPROF|blocks" | sort -u | head -10
