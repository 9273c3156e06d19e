set -eo pipefail
tag=$1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
RAC_SHAPE_LOG=$out/train_shapes.json rocprofv3 --kernel-trace --stats -d "$out/stats_train" -o run --output-format csv -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > "$out/stats_train.json" 2> "$out/stats_train.err"
