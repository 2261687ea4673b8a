set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3k
timeout -k 10 600 python -u -m pytest tests/test_gpu_imputation.py tests/test_gpu_models.py tests/test_gpu_runner.py -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/r3k/pytest.log 2>&1 || { tail -60 gpurun_out/r3k/pytest.log; exit 1; }
tail -3 gpurun_out/r3k/pytest.log
