#!/bin/bash
# Every trace the profile notes of the step-level numbers are built from, each rocprofv3 run on its own:
# whole training steps (tools/step_trace.sh), one epoch of the YAML runner, and the objective kernels alone.
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2prof
for spec in "32768 eager 30" "512 graph 200" "hepmass graph 100" "miniboone graph 100"; do
  set -- $spec
  bash tools/step_trace.sh $1 $2 $3 > gpurun_out/r2prof/step_$1_$2.md 2>&1 || { echo "step trace $spec failed"; tail -5 gpurun_out/r2prof/step_$1_$2.md; exit 1; }
  cp gpurun_out/steptrace_$1_$2/trace/*/*_kernel_stats.csv gpurun_out/r2prof/step_$1_$2_kernel_stats.csv
  echo "traced $spec"
done
rm -rf gpurun_out/r2prof/runner; mkdir -p gpurun_out/r2prof/runner
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2prof/runner/trace -- python tools/runner_epoch.py 1 > gpurun_out/r2prof/runner/run.log 2>&1 || { echo "runner trace failed"; tail -5 gpurun_out/r2prof/runner/run.log; exit 1; }
cp gpurun_out/r2prof/runner/trace/*/*_kernel_stats.csv gpurun_out/r2prof/runner_epoch_kernel_stats.csv
tail -1 gpurun_out/r2prof/runner/run.log > gpurun_out/r2prof/runner_epoch.json
echo "traced runner epoch"
bash tools/objprof.sh gas > gpurun_out/r2prof/objprof_gas.log 2>&1 || { echo "objprof gas failed"; exit 1; }
bash tools/objprof.sh bern > gpurun_out/r2prof/objprof_bern.log 2>&1 || { echo "objprof bern failed"; exit 1; }
echo "all traces done"
