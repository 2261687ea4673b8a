#!/bin/bash
# rocprofv3 passes over the decoder MLP kernels (bench.py --kernels-only): kernel trace + PMC, each its own run.
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_k
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --kernels-only > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python bench.py --kernels-only > $OUT/pmc1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python bench.py --kernels-only > $OUT/pmc2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -- python bench.py --kernels-only > $OUT/pmc3.log 2>&1 || exit 1
echo done
