out=gpurun_out/r04t; mkdir -p $out
for r in 1 2; do for v in prev ship; do
  lib=robot_aware_control_amd/variants/librac_$v.so; [ $v = ship ] && lib=robot_aware_control_amd/librac_hip.so
  echo -n "$v cfg5: "; RAC_HIP_LIB=$lib python bench.py --cfg5 --workload train --steps 4 --warmup 2 --no-cpu-baseline --no-exact --no-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))"
done; done > $out/seg.log 2>&1
cat $out/seg.log
