#!/bin/bash
# sched-sampling step: where the time-batched weight gradients start x the side stream's priority / CU mask
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/log.txt
B="python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 10 --warmup 3 --sched all"
run() { echo "== $*" >> $out/log.txt; env "$@" $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt; }
run X=0
run RAC_SCHED_FLUSH=4,5
run RAC_WGRAD_LOW_PRIORITY=1
run RAC_WGRAD_LOW_PRIORITY=1 RAC_SCHED_FLUSH=4,5
run RAC_WGRAD_LOW_PRIORITY=1 RAC_SCHED_FLUSH=1,2,3,4,5
# CU mask: 24 of every 32 CUs (three quarters of every XCD if bits are XCD-major; otherwise 192 of 256 anyhow)
M=00ffffff00ffffff00ffffff00ffffff00ffffff00ffffff00ffffff00ffffff
run RAC_WGRAD_CU_MASK=$M
run RAC_WGRAD_CU_MASK=$M RAC_SCHED_FLUSH=4,5
run RAC_WGRAD_CU_MASK=$M RAC_SCHED_FLUSH=1,2,3,4,5
python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TF', round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt
RAC_WGRAD_LOW_PRIORITY=1 python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TF lowprio', round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt
RAC_WGRAD_CU_MASK=$M python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TF cumask', round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt
