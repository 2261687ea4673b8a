set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3a
timeout -k 10 400 python -u -m pytest tests/test_gpu_runner.py -x -q --timeout 200 --timeout-method thread -k "bench" > gpurun_out/r3a/pytest_bench.log 2>&1 || { tail -30 gpurun_out/r3a/pytest_bench.log; exit 1; }
tail -3 gpurun_out/r3a/pytest_bench.log
timeout -k 10 300 python bench.py > gpurun_out/r3a/bench_default.json 2> gpurun_out/r3a/bench_default.err || { tail -20 gpurun_out/r3a/bench_default.err; exit 1; }
cat gpurun_out/r3a/bench_default.json
bash tools/prof.sh
