#!/bin/bash
# The v_pk_fma_f32 experiment (run on the GPU box):  bash tools/repro/run_pk_experiment.sh > gpurun_out/pk_experiment.log
# 1. the standalone instruction-level reproducer: each variant alone, then 4 processes at once
# 2. the real first-layer kernel, shipped library vs a build with packed fp32 ops, 1 and 4 processes
cd "$(dirname "$0")/../.."
R=build/pk_fma_repro
L=${1:-600}
echo "== rocm: $(cat /opt/rocm/.info/version 2>/dev/null); clang: $(/opt/rocm/lib/llvm/bin/clang --version | head -1)"
for v in 0 1 2 3; do
  echo "-- standalone variant $v, 1 process"
  $R $v $L 131072
  echo "-- standalone variant $v, 4 processes"
  for i in 1 2 3 4; do $R $v $L 131072 & done; wait
done
echo "-- first-layer kernel, shipped library (scalar v_fma_f32)"
python tools/repro/pk_stress.py 1 300
python tools/repro/pk_stress.py 4 1000
echo "-- first-layer kernel, packed build (v_pk_fma_f32 with op_sel broadcasts)"
RAC_HIP_LIB=robot_aware_control_amd/variants/librac_packed.so python tools/repro/pk_stress.py 1 300
RAC_HIP_LIB=robot_aware_control_amd/variants/librac_packed.so python tools/repro/pk_stress.py 4 1000
