out=gpurun_out/r04x; mkdir -p $out; export RAC_BENCH_SPLIT=1
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -n 2
for r in 1 2; do for v in old ship; do
  lib=robot_aware_control_amd/variants/librac_$v.so; [ $v = ship ] && lib=robot_aware_control_amd/librac_hip.so
  for k in 3 5; do echo -n "$v k=$k M=64000: "; RAC_HIP_LIB=$lib python tools/bench_gemm.py fwd 1000 512 $k 5 2>&1 | grep -i "kernel only" | head -n 1; done
  for shp in "1000 64 64 64 64" "1000 64 64 128 64"; do echo -n "$v: "; RAC_HIP_LIB=$lib python tools/bench_rows.py $shp 10 2>&1 | tail -n 1; done
done; done > $out/behind.log 2>&1
cat $out/behind.log
