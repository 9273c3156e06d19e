#!/bin/bash
# Kernel traces + per-shape tables of the side configurations (scheduled sampling, the deployed GroupNorm model):
#   bash tools/r06_side_profiles.sh <tag>   (on the GPU box, from the repo root)  -> gpurun_out/<tag>/
set -eo pipefail
tag=${1:-r06side}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp RAC_ADAM_OVERLAP=0 RAC_WGRAD_STREAM=0
T="--workload train --steps 5 --warmup 2 --no-cpu-baseline --no-exact --no-side"
C="--workload cem --cem-iters 1 --cem-warmup 1 --no-cpu-baseline --no-exact --no-cem-ra --cem-opt-iter 1 --no-side"
for cfg in "sched:--sched all" "deployed:--deployed" "deployed_sched:--deployed --sched all"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  echo "[side profiles] train $name" >&2
  RAC_SHAPE_LOG=$out/${name}_shapes.json rocprofv3 --kernel-trace --stats -d "$out/$name" -o run --output-format csv -- python3 bench.py $T $flags > "$out/$name.json" 2> "$out/$name.err"
  tr=$(find "$out/$name" -name "*kernel_trace.csv" | head -1)
  python3 tools/shape_profile.py "$tr" "$out/${name}_shapes.json" 5 "train step, $flags" "adam_frag_multi:5" > "$out/${name}_shapes.md" || true
done
echo "[side profiles] cem deployed" >&2
RAC_SHAPE_LOG=$out/deployed_cem_shapes.json rocprofv3 --kernel-trace --stats -d "$out/deployed_cem" -o run --output-format csv -- python3 bench.py $C --deployed > "$out/deployed_cem.json" 2> "$out/deployed_cem.err"
tr=$(find "$out/deployed_cem" -name "*kernel_trace.csv" | head -1)
python3 tools/shape_profile.py "$tr" "$out/deployed_cem_shapes.json" 1 "planner iteration, deployed model (g 256, GroupNorm cells, 48x64)" "cem_step_tail:14" > "$out/deployed_cem_shapes.md" || true
# keep the merged output small: traces are large
find "$out" -name "*kernel_trace.csv" -size +20M -delete || true
echo "[side profiles] done" >&2
