out=gpurun_out/r04l; mkdir -p $out; export RAC_BENCH_SPLIT=1
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -n 2
for r in 1 2; do for v in 0 1; do
  echo -n "unroll5=$v k=5 M=64000: "; RAC_TILE_UNROLL5=$v python tools/bench_gemm.py fwd 1000 512 5 5 2>&1 | grep -i "kernel only" | head -n 1
  echo -n "unroll5=$v k=5 M=1024: "; RAC_TILE_UNROLL5=$v python tools/bench_gemm.py fwd 16 512 5 20 2>&1 | grep -i "kernel only" | head -n 1
done; done > $out/u5.log 2>&1
cat $out/u5.log
(bash tools/ab.sh cem 2 "RAC_TILE_UNROLL5=0" "RAC_TILE_UNROLL5=1"; bash tools/ab.sh train 3 "RAC_TILE_UNROLL5=0" "RAC_TILE_UNROLL5=1") > $out/ab_u5.log 2>&1; cat $out/ab_u5.log
