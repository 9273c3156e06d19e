#!/bin/bash
# A/B of the weight-gradient kernels between library builds on one box:  bash tools/ab_wgrad.sh <variant.so> ...
# shapes: the 5x5 / 3x3 gate weights (T = 5 steps of B = 16, presplit operands) and a vgg-stack 3x3 (in-kernel conversion)
export RAC_BENCH_SPLIT=1
for r in 1 2; do
for lib in "" "$@"; do
  for args in "wgrad 16 512 5 20" "wgrad 16 512 3 20"; do
    echo -n "[${lib:-shipped}] T=5 "; RAC_BENCH_T=5 RAC_HIP_LIB=$lib python tools/bench_gemm.py $args 2>/dev/null | tail -1 | cut -c1-120
    echo -n "[${lib:-shipped}] T=1 "; RAC_BENCH_T=1 RAC_HIP_LIB=$lib python tools/bench_gemm.py $args 2>/dev/null | tail -1 | cut -c1-120
  done
done
done
