#!/bin/bash
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/log.txt
run() { echo "== [$FLAGS] $*" >> $out/log.txt; env "$@" python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 30 --warmup 5 $FLAGS 2>>$out/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['roofline']['achieved'],1), d['time_breakdown_ms'])" >> $out/log.txt; }
FLAGS=""
for i in 1 2 3; do
run X=0
run RAC_STEP_HIGH_PRIORITY=1
done
FLAGS="--deployed"
run X=0
run RAC_STEP_HIGH_PRIORITY=1
FLAGS="--cfg5"
run X=0
run RAC_STEP_HIGH_PRIORITY=1
