set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3j
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/r3j/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/r3j/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r3j/pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err || { tail -30 gpurun_out/r3j/bench.err; exit 1; }
cat gpurun_out/r3j/bench.json
bash tools/prof_r3.sh > gpurun_out/r3j/prof.log 2>&1 || { tail -30 gpurun_out/r3j/prof.log; exit 1; }
tail -12 gpurun_out/r3j/prof.log
timeout -k 10 300 python tools/runner_toy.py > gpurun_out/r3j/runner_toy.log 2>&1 || { tail -20 gpurun_out/r3j/runner_toy.log; exit 1; }
tail -1 gpurun_out/r3j/runner_toy.log
