out=gpurun_out/r04t; mkdir -p $out; export RAC_BENCH_SPLIT=1
for r in 1 2 3; do for v in prev ship; do for k in 3 5; do
  lib=robot_aware_control_amd/variants/librac_$v.so; [ $v = ship ] && lib=robot_aware_control_amd/librac_hip.so
  echo -n "$v k=$k M=64000: "; RAC_HIP_LIB=$lib python tools/bench_gemm.py fwd 1000 512 $k 5 2>&1 | grep -i "kernel only" | head -n 1
done; done; done > $out/center.log 2>&1
cat $out/center.log
