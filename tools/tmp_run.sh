out=gpurun_out/r04l; mkdir -p $out
for r in 1 2; do for v in ship pin0 pin1 pin2 norefill; do for shp in "1000 16 16 256 256" "1000 32 32 128 128"; do
  lib=robot_aware_control_amd/variants/librac_$v.so; [ $v = ship ] && lib=robot_aware_control_amd/librac_hip.so
  echo -n "$v: "; RAC_HIP_LIB=$lib python tools/bench_rows.py $shp 10 2>&1 | tail -n 1
done; done; done > $out/rows_pin.log 2>&1
cat $out/rows_pin.log
