#!/bin/bash
# Everything the round's numbers come from, on one MI355X (about four minutes): the GPU test suite, the bench line, the
# decoder-kernel profile (tools/prof.sh), the whole-step traces (tools/prof_r3.sh), the bench points and the fuzzers.
# Afterwards, on the build host: python tools/summarize_prof.py r3; python tools/summarize_steps.py r3;
# python tools/parity_margins.py > profiles/parity_margins.md; cp gpurun_out/r3all/bench.json profiles/r3_bench_default.json;
# cp gpurun_out/r3all/bench_points.json profiles/r3_bench_points.json
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3all
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 200 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
cat $O/bench.json
bash tools/prof.sh > $O/prof_k.log 2>&1 || { tail -5 gpurun_out/prof_k/*.log; exit 1; }
bash tools/prof_r3.sh > $O/prof_r3.log 2>&1 || { tail -30 $O/prof_r3.log; exit 1; }
tail -3 $O/prof_r3.log
timeout -k 10 600 python tools/bench_points.py $O/bench_points.json > $O/bench_points.log 2>&1 || { tail -20 $O/bench_points.log; exit 1; }
timeout -k 10 300 python tools/runner_toy.py > $O/runner_toy.log 2>&1 || { tail -20 $O/runner_toy.log; exit 1; }
tail -1 $O/runner_toy.log
timeout -k 10 300 python tools/fuzz_samplers.py 100 11 > $O/fuzz_samplers.log 2>&1 || { grep -v " ok " $O/fuzz_samplers.log | tail; exit 1; }
timeout -k 10 300 python tools/fuzz_one_launch.py 100 12 > $O/fuzz_one_launch.log 2>&1 || { grep -v " ok " $O/fuzz_one_launch.log | tail; exit 1; }
timeout -k 10 300 python tools/fuzz_wide.py 16 13 > $O/fuzz_wide.log 2>&1 || { grep -v " ok " $O/fuzz_wide.log | tail; exit 1; }
for f in $O/fuzz_samplers.log $O/fuzz_one_launch.log $O/fuzz_wide.log; do tail -n 1 $f; done
