set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3c
timeout -k 10 600 python -u -m pytest tests/test_gpu_objectives.py tests/test_gpu_models.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3c/pytest_obj.log 2>&1 || { tail -40 gpurun_out/r3c/pytest_obj.log; exit 1; }
tail -3 gpurun_out/r3c/pytest_obj.log
bash tools/objprof_strat.sh after || exit 1
