#!/bin/bash
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/log.txt
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "stream_parts" > $out/test.log 2>&1; echo "rc $?" >> $out/test.log
for n in 1 2 3 4; do
  echo "== RAC_CEM_STREAMS=$n" >> $out/log.txt
  RAC_CEM_STREAMS=$n python bench.py --workload cem --no-exact --no-cpu-baseline --no-side --cem-iters 3 --cem-warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cem', round(d['cem']['s_per_iteration']*1e3,1), 'ra', round(d['cem_ra']['s_per_iteration']*1e3,1), 'get_action', round(d['cem']['get_action']['s_per_call'],3))" >> $out/log.txt
done
