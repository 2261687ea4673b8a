set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3j
bash tools/prof.sh || { tail -5 gpurun_out/prof_k/*.log; exit 1; }
timeout -k 10 600 python bench.py > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err || { tail -30 gpurun_out/r3j/bench.err; exit 1; }
cat gpurun_out/r3j/bench.json
