out=gpurun_out/r04p; mkdir -p $out; export RAC_BENCH_SPLIT=1 TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -x -q -k "split or conv or lstm or vgg or head or frozen" 2>&1 | tail -n 2
for r in 1 2; do for v in tnorefill ship; do for k in 3 5; do
  lib=robot_aware_control_amd/variants/librac_$v.so; [ $v = ship ] && lib=robot_aware_control_amd/librac_hip.so
  echo -n "$v k=$k M=64000: "; RAC_HIP_LIB=$lib python tools/bench_gemm.py fwd 1000 512 $k 5 2>&1 | grep -i "kernel only" | head -n 1
done; done; done > $out/perm.log 2>&1
cat $out/perm.log
rm -rf $out/pmc; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/pmc -o run --output-format csv -- python3 tools/bench_gemm.py fwd 1000 512 5 2 > $out/pmc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/r04p/pmc/**/*counter_collection.csv', recursive=True)[0]
t=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if 'tile_kernel' in r['Kernel_Name']: t[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(t))
PY
