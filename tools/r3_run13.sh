set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3i
timeout -k 10 600 python -u -m pytest tests/test_gpu_imputation.py -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/r3i/pytest_imp.log 2>&1 || { tail -60 gpurun_out/r3i/pytest_imp.log; exit 1; }
tail -3 gpurun_out/r3i/pytest_imp.log
bash tools/step_trace.sh uci graph 100 > gpurun_out/r3i/uci.md 2>&1 || { tail gpurun_out/r3i/uci.md; exit 1; }
cat gpurun_out/r3i/uci.md
