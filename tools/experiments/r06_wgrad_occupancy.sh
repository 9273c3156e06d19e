#!/bin/bash
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/log.txt
run() { echo "== [$FLAGS] $*" >> $out/log.txt; env "$@" python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 10 --warmup 3 $FLAGS 2>>$out/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt; }
FLAGS="--sched all"
run X=0
run RAC_WGRAD_LDS_MIN=84000
run RAC_WGRAD_LDS_MIN=84000 RAC_SCHED_FLUSH=4,5
run RAC_WGRAD_LDS_MIN=84000 RAC_SCHED_FLUSH=3,5
run RAC_WGRAD_LDS_MIN=84000 RAC_SCHED_FLUSH=1,2,3,4,5
run RAC_WGRAD_LDS_MIN=84000 RAC_SCHED_FLUSH=2,4,5
FLAGS=""
run X=0
run RAC_WGRAD_LDS_MIN=84000
