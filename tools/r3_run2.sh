set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r3b/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r3b/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r3b/pytest_gpu.log
bash tools/objprof_strat.sh before || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r3b/bench_default.json 2> gpurun_out/r3b/bench_default.err || { tail -20 gpurun_out/r3b/bench_default.err; exit 1; }
cat gpurun_out/r3b/bench_default.json
