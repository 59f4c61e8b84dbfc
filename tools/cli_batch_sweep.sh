#!/bin/bash
# wall clock of `4mc -d` (streaming path) on an 8 GiB file for several FOURMC_BATCH_BLOCKS; run on the GPU box from the repo root
set -u
d=${1:-/dev/shm}/sweep.$$; mkdir -p $d
python - <<P
import sys; sys.path.insert(0,'tests'); import helpers
B=4<<20; base=helpers.corpus(48*B)
with open("$d/in.bin","wb") as f:
    for k in range(0,2048,48): f.write(base[:min(48,2048-k)*B].tobytes())
P
4mc_amd/bin/4mc ${ZFLAG:-} -f $d/in.bin $d/c.4mc > /dev/null 2>&1
for nb in 64 128 256 512; do
  for rep in 1 2; do
    s=$(date +%s.%N); FOURMC_BATCH_BLOCKS=$nb 4mc_amd/bin/4mc ${ZFLAG:-} -d -f $d/c.4mc $d/back > /dev/null 2>&1; e=$(date +%s.%N)
    python -c "print('batch $nb: %.2f s' % ($e - $s))"; rm -f $d/back
  done
done
s=$(date +%s.%N); FOURMC_MMAP=1 4mc_amd/bin/4mc ${ZFLAG:-} -d -f $d/c.4mc $d/back > /dev/null 2>&1; e=$(date +%s.%N); python -c "print('mapped: %.2f s' % ($e - $s))"
rm -rf $d
