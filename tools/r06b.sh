#!/bin/bash
# round 6, second GPU pass: determinism floor, BatchNorm reduce block counts, sched / deployed lines
set -o pipefail
out=gpurun_out/r06b; mkdir -p $out
python tools/debug_overlap.py 1 0 > $out/dbg_overlap.log 2>&1
for coef in 0.1 0.2 0.35 0.6 1.0; do
  for shape in "1024 512 1" "5120 512 5" "4096 256 1" "65536 64 1" "327680 64 5" "20480 256 5"; do
    echo "coef $coef" >> $out/bn_reduce.log
    RAC_BN_REDUCE_COEF=$coef python tools/bench_bn_reduce.py $shape 2>&1 | grep "reduce warm" >> $out/bn_reduce.log
  done
done
echo "old" >> $out/bn_reduce.log
for shape in "1024 512 1" "5120 512 5" "4096 256 1" "65536 64 1" "327680 64 5" "20480 256 5"; do
  RAC_BN_REDUCE_OLD=1 python tools/bench_bn_reduce.py $shape 2>&1 | grep "reduce warm" >> $out/bn_reduce.log
done
python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline --no-exact > $out/bench_train.json 2> $out/bench_train.err
