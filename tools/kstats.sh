#!/bin/bash
# Per-kernel time of a short train (or cem) bench under rocprofv3:  bash tools/kstats.sh <tag> train|cem [ENV=val ...]
# -> gpurun_out/<tag>/ (kernel stats csv) and the top rows on stdout
tag=$1; w=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 2 --cem-iters 1 --cem-warmup 1 --no-cpu-baseline --no-exact --no-side --no-cem-ra --cem-opt-iter 1 $BENCH_FLAGS > $out/bench.json 2> $out/bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms {tot/1e6:.1f}")
for r in rows[:int(__import__('os').environ.get('KSTATS_ROWS', 28))]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
PY
