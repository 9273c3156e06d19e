#!/bin/bash
# SQ-side PMC pass over the gate GEMM micro-benchmark: bash tools/pmc_sq.sh <tag> <B> <k>
set -eo pipefail
tag=${1:-sq}; B=${2:-500}; k=${3:-5}
out=gpurun_out/$tag
mkdir -p "$out"; export TMPDIR=/tmp
RAC_BENCH_SPLIT=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d "$out/sq1" -o run --output-format csv -- python3 tools/bench_gemm.py fwd $B 512 $k 3 > "$out/sq1.log" 2>&1
RAC_BENCH_SPLIT=1 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE -d "$out/sq2" -o run --output-format csv -- python3 tools/bench_gemm.py fwd $B 512 $k 3 > "$out/sq2.log" 2>&1 || echo "sq2 failed" >&2
echo done >&2
