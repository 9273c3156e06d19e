export RAC_BENCH_SPLIT=1
for k in 5 3; do for s in 2 4 5 6 8 10 12 16; do echo -n "k=$k split=$s: "; RAC_SPLIT=$s python tools/bench_gemm.py fwd 16 512 $k 40 2>&1 | grep "kernel only" | sed 's/.*: //'; done; done
for k in 5 3; do for s in 4 8 16; do echo -n "dgrad k=$k split=$s: "; RAC_SPLIT=$s python tools/bench_gemm.py dgrad 16 512 $k 40 2>&1 | grep "dgrad-split" | sed 's/.*: //'; done; done
