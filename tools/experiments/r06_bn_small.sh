#!/bin/bash
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/log.txt
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "small_bn or vgg_layer" > $out/test.log 2>&1; echo "rc $?" >> $out/test.log
run() { echo "== [$FLAGS] $*" >> $out/log.txt; env "$@" python bench.py --workload train --no-exact --no-cpu-baseline --no-side --steps 10 --warmup 3 $FLAGS 2>>$out/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['time_breakdown_ms'])" >> $out/log.txt; }
FLAGS="--sched all"
run X=0
run RAC_BN_SMALL=0
run RAC_BN_SMALL_BYTES=8388608
run RAC_BN_SMALL_BYTES=2097152
run RAC_BN_SMALL_BYTES=16777216
run X=0
FLAGS="--deployed --sched all"
run X=0
run RAC_BN_SMALL=0
