#!/bin/bash
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/log.txt
run() { echo "== $*" >> $out/log.txt; env "$@" python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 10 --warmup 3 $FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt; }
for FLAGS in "--sched all" "--deployed" "--deployed --sched all" ""; do
  echo "#### $FLAGS" >> $out/log.txt
  run RAC_MAX_SPLIT_K=8
  run RAC_MAX_SPLIT_K=16
  run RAC_MAX_SPLIT_K=32
done
