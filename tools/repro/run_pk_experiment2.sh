#!/bin/bash
# Part 2 of the v_pk_fma_f32 experiment: the same checks while THREE other processes keep the GPU busy with the planner
# benchmark (8 000-workgroup MFMA kernels of ~14 ms): the mixed load under which tests/test_gpu_dist.py fails with the
# packed build of rac_frame.hip.      bash tools/repro/run_pk_experiment2.sh > gpurun_out/pk_experiment2.log 2>&1
cd "$(dirname "$0")/../.."
R=build/pk_fma_repro
load() { python bench.py --workload cem --cem-iters 40 --cem-warmup 0 --no-cpu-baseline --no-exact --no-cem-ra --cem-opt-iter 1 > /dev/null 2>&1; }
load & L1=$!; load & L2=$!; load & L3=$!
sleep 75   # imports + model build of the load processes
echo "== under load (3 x planner benchmark):"
for v in 0 3 1 2; do $R $v 1500 131072; done
echo "-- first-layer kernel, packed build of rac_frame only, 1 process, under load"
RAC_HIP_LIB=robot_aware_control_amd/variants/librac_pk_rac_frame.so python tools/repro/pk_stress.py 1 1500
echo "-- first-layer kernel, shipped library, 1 process, under load"
python tools/repro/pk_stress.py 1 1500
kill $L1 $L2 $L3 2>/dev/null; wait
echo "== load stopped"
