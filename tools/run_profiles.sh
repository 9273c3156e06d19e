#!/bin/bash
# Profile passes behind profiles/<tag>_*: run on the GPU box from the repo root:  bash tools/run_profiles.sh r01d
# kernel-trace/stats and each PMC counter are collected in SEPARATE rocprofv3 runs of the same bench command.
set -eo pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
T="--workload train --steps 5 --warmup 2 --no-cpu-baseline"
C="--workload cem --cem-iters 1 --cem-warmup 1 --no-cpu-baseline"
echo "[profiles] stats train" >&2
rocprofv3 --kernel-trace --stats -d "$out/stats_train" -o run --output-format csv -- python3 bench.py $T > "$out/stats_train.json" 2> "$out/stats_train.err"
echo "[profiles] stats cem" >&2
rocprofv3 --kernel-trace --stats -d "$out/stats_cem" -o run --output-format csv -- python3 bench.py $C > "$out/stats_cem.json" 2> "$out/stats_cem.err"
for ctr in FETCH_SIZE WRITE_SIZE; do
  echo "[profiles] pmc $ctr train" >&2
  rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_train" -o run --output-format csv -- python3 bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$out/pmc_${ctr}_train.err"
  echo "[profiles] pmc $ctr cem" >&2
  rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_cem" -o run --output-format csv -- python3 bench.py $C > /dev/null 2> "$out/pmc_${ctr}_cem.err"
  echo "[profiles] pmc $ctr gate GEMMs" >&2
  RAC_BENCH_SPLIT=1 rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_gemm_train" -o run --output-format csv -- python3 tools/bench_gemm.py fwd 16 512 5 3 > /dev/null 2> "$out/pmc_${ctr}_gemm_train.err"
  RAC_BENCH_SPLIT=1 rocprofv3 --pmc $ctr -d "$out/pmc_${ctr}_gemm_cem" -o run --output-format csv -- python3 tools/bench_gemm.py fwd 1000 512 5 3 > /dev/null 2> "$out/pmc_${ctr}_gemm_cem.err"
done
echo "[profiles] done" >&2
