set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3d
DMV_FLOOR_FRACTION_MAX=1.0 timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 200 --timeout-method thread > gpurun_out/r3d/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r3d/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r3d/pytest_gpu.log
bash tools/step_trace.sh hepmass graph 100 > gpurun_out/r3d/step_hepmass.md 2>&1 || { tail gpurun_out/r3d/step_hepmass.md; exit 1; }
bash tools/step_trace.sh miniboone graph 100 > gpurun_out/r3d/step_miniboone.md 2>&1 || { tail gpurun_out/r3d/step_miniboone.md; exit 1; }
cat gpurun_out/r3d/step_hepmass.md gpurun_out/r3d/step_miniboone.md
