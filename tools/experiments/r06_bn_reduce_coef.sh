#!/bin/bash
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/log.txt
run() { echo "== [$FLAGS] $*" >> $out/log.txt; env "$@" python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 10 --warmup 3 $FLAGS 2>>$out/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt; }
FLAGS="--sched all"
for c in 0.365 0.2 0.6 1.0 1.6 0.365; do run RAC_BN_REDUCE_COEF=$c; done
FLAGS=""
for c in 0.365 1.0 0.365; do run RAC_BN_REDUCE_COEF=$c; done
