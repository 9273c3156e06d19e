#!/bin/bash
# workgroups of the fused BatchNorm apply pass (rac_bn_apply_act) on the teacher-forced step: kernel averages per setting
out=gpurun_out/exp_$(basename $0 .sh); mkdir -p $out; : > $out/log.txt
for nb in 512 1024 2048 4096; do
  echo "== RAC_BN_APPLY_BLOCKS=$nb" >> $out/log.txt
  RAC_BN_APPLY_BLOCKS=$nb KSTATS_ROWS=60 bash tools/kstats.sh exp_bn_apply_$nb train 2>/dev/null | grep -E "total|bn_apply_act|bn_bwd_apply|bn_bwd_reduce" >> $out/log.txt
  python -c "import json; d=json.load(open('gpurun_out/exp_bn_apply_$nb/bench.json')); print('ms_per_step', round(d['ms_per_step'],2))" >> $out/log.txt
  rm -rf gpurun_out/exp_bn_apply_$nb
done
