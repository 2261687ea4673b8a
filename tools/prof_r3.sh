#!/bin/bash
# Round-3 traces, each rocprofv3 run on its own: whole training steps (tools/step_trace.sh), one epoch of the YAML runner
# on the gas-sized table, three epochs of the UCI DeMissVAE config (LAIR update with a prior-replenish proposal, ten
# iterations, final resampling) and the toy DeMissVAE config.  Results under gpurun_out/r3prof; tools/summarize_steps.py r3
# turns them into profiles/r3_training_steps.md.
set -u
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/r3prof
mkdir -p $P
for spec in "32768 eager 30" "512 graph 200" "hepmass graph 100" "miniboone graph 100" "toy graph 100" "uci graph 100" "wide graph 50"; do
  set -- $spec
  bash tools/step_trace.sh $1 $2 $3 > $P/step_$1_$2.md 2>&1 || { echo "step trace $spec failed"; tail -5 $P/step_$1_$2.md; exit 1; }
  cp gpurun_out/steptrace_$1_$2/trace/*/*_kernel_stats.csv $P/step_$1_$2_kernel_stats.csv
  cp gpurun_out/steptrace_$1_$2/trace/*/*_kernel_trace.csv $P/step_$1_$2_kernel_trace.csv
  echo "traced $spec"
done
for spec in "runner_epoch 1" "runner_uci_demiss 3 demiss"; do
  set -- $spec
  name=$1; shift
  rm -rf $P/$name; mkdir -p $P/$name
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$name/trace -- python tools/runner_epoch.py "$@" > $P/$name/run.log 2>&1 || { echo "$name trace failed"; tail -5 $P/$name/run.log; exit 1; }
  cp $P/$name/trace/*/*_kernel_stats.csv $P/${name}_kernel_stats.csv
  echo "traced $name"
done
echo "all traces done"
