# LZ4 decode paths by launch size: which path serves which launch best (container corpus)
for nb in 128 256 512 768 1024 1536 2048; do for m in tile seg; do FOURMC_DECODE=$m timeout 300 python tools/k1_big.py $nb 2>&1 | grep blocks | cut -c1-90; done; done
