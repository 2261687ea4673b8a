set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3j
timeout -k 10 600 python tools/bench_points.py gpurun_out/r3j/bench_points.json > gpurun_out/r3j/bench_points.log 2>&1 || { tail -20 gpurun_out/r3j/bench_points.log; exit 1; }
tail -5 gpurun_out/r3j/bench_points.log
