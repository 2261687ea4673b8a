set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3h
bash tools/step_trace.sh 512 graph 200 > gpurun_out/r3h/s512_a.md 2>&1 || { tail gpurun_out/r3h/s512_a.md; exit 1; }
bash tools/step_trace.sh uci graph 100 > gpurun_out/r3h/uci_a.md 2>&1 || { tail gpurun_out/r3h/uci_a.md; exit 1; }
export DMV_OBJ_SPREAD=1
bash tools/step_trace.sh 512 graph 200 > gpurun_out/r3h/s512_b.md 2>&1 || { tail gpurun_out/r3h/s512_b.md; exit 1; }
bash tools/step_trace.sh uci graph 100 > gpurun_out/r3h/uci_b.md 2>&1 || { tail gpurun_out/r3h/uci_b.md; exit 1; }
grep -h "ms/step\|k_obj_fwd" gpurun_out/r3h/s512_a.md gpurun_out/r3h/uci_a.md gpurun_out/r3h/s512_b.md gpurun_out/r3h/uci_b.md
