set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3h
timeout -k 10 600 python -u -m pytest tests/test_gpu_mlp.py tests/test_gpu_objectives.py -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/r3h/pytest_mlp.log 2>&1 || { tail -60 gpurun_out/r3h/pytest_mlp.log; exit 1; }
tail -3 gpurun_out/r3h/pytest_mlp.log
timeout -k 10 300 python tools/wide_point.py > gpurun_out/r3h/wide_point.log 2>&1 || { tail -30 gpurun_out/r3h/wide_point.log; exit 1; }
cat gpurun_out/r3h/wide_point.log
