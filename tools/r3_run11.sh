set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3h
bash tools/step_trace.sh wide graph 50 > gpurun_out/r3h/wide_pf3.md 2>&1 || { tail gpurun_out/r3h/wide_pf3.md; exit 1; }
cat gpurun_out/r3h/wide_pf3.md
export DMV_WGRAD_BT=4
bash tools/step_trace.sh wide graph 50 > gpurun_out/r3h/wide_pf3_bt4.md 2>&1 || { tail gpurun_out/r3h/wide_pf3_bt4.md; exit 1; }
cat gpurun_out/r3h/wide_pf3_bt4.md
