#!/bin/bash
# kernel traces with every stream ON (the configuration bench.py times), for tools/timeline.py:  bash tools/r06_timeline.sh <tag>
set -eo pipefail
tag=${1:-r06tl}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
T="--workload train --steps 4 --warmup 2 --no-cpu-baseline --no-exact --no-side"
for cfg in "tf:" "sched:--sched all" "deployed:--deployed"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  rocprofv3 --kernel-trace -d "$out/$name" -o run --output-format csv -- python3 bench.py $T $flags > "$out/$name.json" 2> "$out/$name.err"
  tr=$(find "$out/$name" -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline.py "$tr" > "$out/${name}_timeline.md" || true
  rm -f "$tr"
done
