export TMPDIR=/tmp RAC_BENCH_SPLIT=1
mkdir -p gpurun_out/conf
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/conf/a -o run --output-format csv -- python3 tools/bench_gemm.py fwd 1000 512 5 3 > gpurun_out/conf/a.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/conf/a/**/*counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:50]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'conv16' in k: print(k, {a:f"{b:.3g}" for a,b in v.items()})
PY
