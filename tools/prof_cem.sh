set -eo pipefail
out=gpurun_out/p4; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/stats" -o run --output-format csv -- python3 bench.py --workload cem --cem-iters 1 --cem-warmup 1 --no-cpu-baseline > "$out/stats.json" 2> "$out/stats.err"
