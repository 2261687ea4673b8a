#!/bin/bash
# rocprofv3 passes over the objective kernels (tools/objprof.py): kernel trace, then FETCH_SIZE and WRITE_SIZE each in
# a run of its own.  Usage: bash tools/objprof.sh <gas|bern|mb|hep>
set -u
W=${1:-gas}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/objprof_$W
mkdir -p $OUT
timeout -k 10 120 python tools/objprof.py $W 50 > $OUT/events.jsonl 2> $OUT/events.err || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/objprof.py $W 20 > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python tools/objprof.py $W 5 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python tools/objprof.py $W 5 > $OUT/pmc_write.log 2>&1 || exit 1
cat $OUT/events.jsonl
echo done
