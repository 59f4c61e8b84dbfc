#!/bin/bash
# time of one encode launch against the number of blocks in it (are the encoders' tables better off in a smaller launch?)
#   tools/enc_sizes.sh            (on the GPU box; writes gpurun_out/enc_sizes.txt)
mkdir -p gpurun_out
for c in hc4 z1 z3 mc; do for n in 128 256 512 1024 2048; do python tools/enc_time.py $c $n; done; done 2>&1 | grep blocks | tee gpurun_out/enc_sizes.txt
for n in 128 256 512 1024 2048; do python tools/zenc_time.py 12 $n; done 2>&1 | grep blocks | tee -a gpurun_out/enc_sizes.txt
