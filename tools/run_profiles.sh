#!/bin/bash
# Profile passes behind profiles/<tag>_*: run on the GPU box from the repo root:  bash tools/run_profiles.sh r02a
# kernel trace / stats and each PMC counter group are collected in SEPARATE rocprofv3 runs (never --pmc with a trace).
# Afterwards (anywhere):  python tools/profile_pack.py gpurun_out/<tag> <tag>   -> profiles/<tag>_*.{md,csv,json}
set -eo pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
# (per-kernel durations and bytes: the optimiser pass in ONE stream, not under the next step's encoder -- see RAC_WGRAD_STREAM below)
export RAC_ADAM_OVERLAP=0
T="--workload train --steps 5 --warmup 2 --no-cpu-baseline --no-exact --no-side"
C="--workload cem --cem-iters 1 --cem-warmup 1 --no-cpu-baseline --no-exact --no-cem-ra --cem-opt-iter 1 --no-side"
echo "[profiles] kernel trace + stats, train" >&2
# (per-kernel durations: weight gradients IN ORDER on the main stream -- on the side stream they overlap other kernels, and
# both sides of an overlap report longer durations than they have alone; the step itself is timed by bench.py with streams on)
RAC_WGRAD_STREAM=0 RAC_SHAPE_LOG=$out/train_shapes.json rocprofv3 --kernel-trace --stats -d "$out/stats_train" -o run --output-format csv -- python3 bench.py $T > "$out/stats_train.json" 2> "$out/stats_train.err"
echo "[profiles] kernel trace + stats, cem" >&2
RAC_SHAPE_LOG=$out/cem_shapes.json rocprofv3 --kernel-trace --stats -d "$out/stats_cem" -o run --output-format csv -- python3 bench.py $C > "$out/stats_cem.json" 2> "$out/stats_cem.err"
G1="python3 tools/bench_gemm.py fwd 16 512 5 3"
G2="python3 tools/bench_gemm.py fwd 1000 512 5 3"
G3="python3 tools/bench_gemm.py wgrad 16 512 5 3"
G4="python3 tools/bench_gemm.py wgrad 16 512 3 3"
export RAC_BENCH_SPLIT=1 RAC_BENCH_T=5
for ctr in FETCH_SIZE WRITE_SIZE; do
  echo "[profiles] pmc $ctr" >&2
  rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_gemm_train" -o run --output-format csv -- $G1 > /dev/null 2> "$out/pmc_${ctr}_1.err"
  rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_gemm_cem" -o run --output-format csv -- $G2 > /dev/null 2> "$out/pmc_${ctr}_2.err"
  rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_wgrad5" -o run --output-format csv -- $G3 > /dev/null 2> "$out/pmc_${ctr}_3.err"
  rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_wgrad3" -o run --output-format csv -- $G4 > /dev/null 2> "$out/pmc_${ctr}_4.err"
  rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_train" -o run --output-format csv -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline --no-exact --no-side > /dev/null 2> "$out/pmc_${ctr}_5.err"
  # (the planner's memory-bound tail: first layer, output head, step tail)
  rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_cem" -o run --output-format csv -- python3 bench.py --workload cem --cem-iters 1 --cem-warmup 0 --no-cpu-baseline --no-exact --no-cem-ra --cem-opt-iter 1 --no-side > /dev/null 2> "$out/pmc_${ctr}_6.err"
done
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SQ2="SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
i=0
for cmd in "$G1" "$G2" "$G3" "$G4" "python3 bench.py --workload cem --cem-iters 1 --cem-warmup 0 --no-cpu-baseline --no-exact --no-cem-ra --cem-opt-iter 1 --no-side"; do
  i=$((i + 1))
  echo "[profiles] SQ counters, run $i" >&2
  rocprofv3 --pmc $SQ1 -d "$out/sq1_$i" -o run --output-format csv -- $cmd > /dev/null 2> "$out/sq1_$i.err"
  rocprofv3 --pmc $SQ2 -d "$out/sq2_$i" -o run --output-format csv -- $cmd > /dev/null 2> "$out/sq2_$i.err" || echo "sq2 $i failed" >&2
done
echo "[profiles] done" >&2
