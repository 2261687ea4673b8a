#!/bin/bash
# extra PMC passes over tools/kbench.py: instruction-issue mix and instruction-cache behaviour
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_x
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d $OUT/pmcA -- python tools/kbench.py > $OUT/pmcA.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_STALL SQ_WAVE_CYCLES --output-format csv -d $OUT/pmcB -- python tools/kbench.py > $OUT/pmcB.log 2>&1 || exit 1
echo done
