#!/bin/bash
# 4mz decode: kernel times of the entropy and the execute stage against the number of frames in the launch (rocprofv3 --kernel-trace --stats)
#   tools/zdec_sizes.sh [level]      (on the GPU box; writes gpurun_out/zdec_sizes.txt)
export TMPDIR=/tmp; mkdir -p gpurun_out; : > gpurun_out/zdec_sizes.txt
for n in 256 512 1024 2048; do
  ZDEC_SPLIT_ONLY=1 timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/pz_$n -o z -- python tools/zdec_ab.py $n ${1:-1} > /tmp/z_$n.log 2>&1
  grep split /tmp/z_$n.log | tail -1 >> gpurun_out/zdec_sizes.txt
  db=$(find /tmp/pz_$n -name "*_results.db" | head -1)
  python - "$db" $n >> gpurun_out/zdec_sizes.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel" in t]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd[0]})")]
rows = list(cur.execute(f"select name, (end - start) / 1e6 from {kd[0]} where name like '%zstd%' order by start"))
by = {}
for n, d in rows: by.setdefault(n.split('(')[0][-40:], []).append(round(d, 2))
for k, v in by.items(): print(f"  {sys.argv[2]} frames  {k}: {v[-6:]}")
PY
done
cat gpurun_out/zdec_sizes.txt
