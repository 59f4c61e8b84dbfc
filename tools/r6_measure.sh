# round 6 measurements that are not the bench line: the LDS-table variant of the zstd-1 encoder against the product (launch-size sweep),
# the tolerant LZ4 encoder's counters and traffic re-measured         gpurun -- bash tools/r6_measure.sh
mkdir -p gpurun_out/r6e
for n in 128 256 512 2048; do python tools/enc_time.py z1 $n; done 2>&1 | grep blocks > gpurun_out/r6e/z1_product.txt
for n in 128 256 512 2048; do FOURMC_LIB=$PWD/4mc_amd/lib/libhadoop-4mc-z1lds.so python tools/enc_time.py z1 $n; done 2>&1 | grep blocks > gpurun_out/r6e/z1_lds_table.txt
echo "== product"; cat gpurun_out/r6e/z1_product.txt; echo "== 64 KiB hash table in LDS (one block per CU)"; cat gpurun_out/r6e/z1_lds_table.txt
bash tools/k2p_sq.sh > gpurun_out/r6e/k2p_sq.txt 2>&1; tail -40 gpurun_out/r6e/k2p_sq.txt | cut -c1-160
