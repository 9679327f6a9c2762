#!/bin/bash
# Runs ON the GPU box: HBM traffic and SQ counters of the bf16 kernels at 608 x 608 (four rocprofv3 --pmc passes of
# tools/run_kernels.py bf16) -> gpurun_out/pmc_bf16.json (copy to profiles/rNN_pmc_bf16.json).   usage: bash tools/collect_bf16_pmc.sh
set -e
R=$PWD; O=$R/gpurun_out/bf16pmc; mkdir -p $O
python3 $R/tools/run_kernels.py bf16 32 3 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o p --output-format csv -- python3 $R/tools/run_kernels.py bf16 32 1 > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o p --output-format csv -- python3 $R/tools/run_kernels.py bf16 32 1 > $O/w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/m -o p --output-format csv -- python3 $R/tools/run_kernels.py bf16 32 1 > $O/m.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/l -o p --output-format csv -- python3 $R/tools/run_kernels.py bf16 32 1 > $O/l.log 2>&1
g() { ls $O/$1/*counter_collection.csv $O/$1/*/*counter_collection.csv 2>/dev/null | head -1; }
python3 $R/tools/pmc_bf16.py $(g f) $(g w) $(g m) $(g l) "$(cat $R/.bench_head 2>/dev/null)" > $R/gpurun_out/pmc_bf16.json
rm -rf $O
