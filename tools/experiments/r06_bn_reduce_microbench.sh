#!/bin/bash
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/bn.log
for mode in "RAC_BN_REDUCE_OLD=1" "RAC_BN_REDUCE_COEF=0.365" "RAC_BN_REDUCE_COEF=0.8" "RAC_BN_REDUCE_COEF=1.6"; do
  for shape in "327680 64 5 1" "81920 128 5 2" "20480 256 5 3" "5120 512 5 4" "65536 64 1 4" "16384 128 1 6" "4096 256 1 8" "1024 512 1 8"; do
    echo "$mode" >> $out/bn.log
    env $mode python tools/bench_bn_reduce.py $shape 2>&1 | grep -v apply | grep "us" >> $out/bn.log
  done
done
