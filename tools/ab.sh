#!/bin/bash
# A/B runs on one GPU box, interleaved:  bash tools/ab.sh train|cem ROUNDS "ENV1=a ENV2=b" "ENV1=c" ...
# prints one line per (round, variant): ms/step and frames/s (train) or rollouts/s (cem)
w=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact --no-side --no-cem-ra --cem-opt-iter 1 --workload $w --cem-iters 2 2>/dev/null)
    python - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"[{sys.argv[1] or 'default'}] {d['unit']}: {d['value']:.1f}  ms: {d['ms_per_step']:.2f}  kernel TF: {d['roofline']['achieved']:.0f}", flush=True)
PY
  done
done
