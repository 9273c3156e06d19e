#!/bin/bash
# same-box A/Bs of round 6's switches on the teacher-forced step, the fed-back step and the deployed model's step
out=gpurun_out/r06_ab; mkdir -p $out; : > $out/log.txt
run() { echo "== [$FLAGS] $*" >> $out/log.txt; env "$@" python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 20 --warmup 5 $FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt; }
FLAGS=""
run X=0
run RAC_BN_REDUCE_OLD=1
run RAC_BN_FUSED_APPLY=0
run RAC_COLSUM_ATOMIC=1
run X=0
FLAGS="--sched all"
run X=0
run RAC_BN_REDUCE_OLD=1
run RAC_BN_FUSED_APPLY=0
run RAC_VGG_WGRAD_BATCH=0
run RAC_SCHED_FLUSH=99
run X=0
FLAGS="--deployed"
run X=0
run RAC_NORM_RECURRENT_CORE=0
run RAC_NORM_RECURRENT_CORE=0 RAC_NORM_CELL_BWD_FUSED=0
run RAC_NORM_RECURRENT_CORE=0 RAC_NORM_CELL_NODE=0
run X=0
