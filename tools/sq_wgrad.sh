#!/bin/bash
# Quick look at the time-batched gate weight gradients (T = 5, B = 16, presplit operands) on the GPU box:
#   bash tools/sq_wgrad.sh <tag> [ENV=val ...]   -> kernel durations (rocprofv3 --kernel-trace --stats) and the SQ counter
# summary (MFMA busy, VALU / LDS / VMEM per MFMA) of wgrad16_kernel for k = 5 and k = 3, separate --pmc passes.
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp RAC_BENCH_SPLIT=1 RAC_BENCH_T=5
for kv in "$@"; do export "$kv"; done
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SQ2="SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
for k in 5 3; do
  rocprofv3 --kernel-trace --stats -d $out/st$k -o run --output-format csv -- python3 tools/bench_gemm.py wgrad 16 512 $k 10 > $out/st$k.log 2>&1
  rocprofv3 --pmc $SQ1 -d $out/sq1_$k -o run --output-format csv -- python3 tools/bench_gemm.py wgrad 16 512 $k 3 > /dev/null 2> $out/sq1_$k.err
  rocprofv3 --pmc $SQ2 -d $out/sq2_$k -o run --output-format csv -- python3 tools/bench_gemm.py wgrad 16 512 $k 3 > /dev/null 2> $out/sq2_$k.err
done
python3 - $out <<'PY'
import csv, glob, os, sys
sys.path.insert(0, "tools")
from profile_pack import counters
out = sys.argv[1]
for k in (5, 3):
    for f in glob.glob(os.path.join(out, f"st{k}", "**", "*kernel_stats.csv"), recursive=True):
        for r in list(csv.DictReader(open(f)))[:6]:
            print(f"k={k} {float(r['AverageNs'])/1e3:9.1f} us x {int(r['Calls']):4d}  {r['Name'][:80]}")
    a, b = counters(os.path.join(out, f"sq1_{k}")), counters(os.path.join(out, f"sq2_{k}"))
    for key, c in a.items():
        if not key[0].startswith("wgrad16"):
            continue
        d = b.get(key, {})
        gui = c.get("GRBM_GUI_ACTIVE", 0) / 8
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        wc = c.get("SQ_WAVE_CYCLES", 1)
        n = mf / 16
        print(f"k={k} {key[0]} ({key[1]} wgs): MFMA busy {100*mf/(1024*gui):.1f} %  active/wait-inst/wait "
              f"{100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:.0f}/{100*c.get('SQ_WAIT_INST_ANY',0)/wc:.0f}/{100*c.get('SQ_WAIT_ANY',0)/wc:.0f} %  "
              f"VALU/MFMA {(d.get('SQ_INSTS_VALU',0)-n)/n:.2f}  LDS/MFMA {d.get('SQ_INSTS_LDS',0)/n:.3f}  VMEM/MFMA {d.get('SQ_INSTS_VMEM_RD',0)/n:.3f}  "
              f"LDS conflicts {c.get('SQ_LDS_BANK_CONFLICT',0):.3g} / {d.get('SQ_LDS_IDX_ACTIVE',0):.3g}")
PY
