# lazy zstd levels: the batched row search (product build) against the one-search-per-trip build (make variant V=nobatch F=zstd_encode D=-DFOURMC_ZLAZY_NOBATCH)
for lib in "" nobatch; do export FOURMC_LIB=$PWD/4mc_amd/lib/libhadoop-4mc${lib:+-$lib}.so; echo "== ${lib:-batch}"
python tools/zenc_time.py 12 2048 | grep level; python tools/zenc_time.py 6 2048 | grep level; python tools/zenc_time.py 12 2048 logs | grep level; python tools/zenc_time.py 12 128 | grep level; done
