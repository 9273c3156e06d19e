# A/B of the gate-GEMM micro-benchmark between library builds:  bash tools/ab_gemm.sh <variant.so> [<variant2.so> ...]
export RAC_BENCH_SPLIT=1
for r in 1 2; do
for lib in "" "$@"; do
  for args in "fwd 1000 512 5 10" "fwd 16 512 5 50" "fwd 1000 512 3 10" "dgrad 16 512 5 50"; do
    echo -n "[${lib:-shipped}] "; RAC_HIP_LIB=$lib python tools/bench_gemm.py $args 2>/dev/null | tail -1 | cut -c1-110
  done
done
done
