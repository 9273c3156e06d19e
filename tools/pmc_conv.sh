# PMC pass over one conv shape (GPU box): bash tools/pmc_conv.sh "<counters>" <bench_conv args>
set -eo pipefail
ctr=$1; shift
out=gpurun_out/pmc_conv; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp BENCH_DGRAD=0
rocprofv3 --pmc $ctr -d $out -o run --output-format csv -- python3 tools/bench_conv.py "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_conv/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-60:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "conv16" in k:
        print(k, {c: round(v / n[(k, c)], 1) for c, v in d.items()})
PY
