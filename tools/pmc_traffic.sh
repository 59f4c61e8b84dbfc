set -u
export TMPDIR=/tmp
tag=r03a; out=gpurun_out/$tag; raw=/tmp/prof_$tag; mkdir -p $out $raw
run() { name=$1; shift; rocprofv3 "$@" -d $raw/$name -o $name -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu > $out/$name.log 2>&1; }
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
for p in fetch write; do db=$(find $raw/$p -name "*_results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/summary_$p.md; done
