out=gpurun_out/r04r; mkdir -p $out
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -n 2
for r in 1 2; do for v in nofast ship; do for shp in "1000 64 64 64 64" "1000 64 64 128 64" "1000 16 16 256 256" "1000 32 32 128 128"; do
  lib=robot_aware_control_amd/variants/librac_$v.so; [ $v = ship ] && lib=robot_aware_control_amd/librac_hip.so
  echo -n "$v: "; RAC_HIP_LIB=$lib python tools/bench_rows.py $shp 10 2>&1 | tail -n 1
done; done; done > $out/epi.log 2>&1
cat $out/epi.log
