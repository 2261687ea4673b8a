set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3f
timeout -k 10 400 python bench.py --no-cpu > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err || { tail -20 gpurun_out/r3f/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3f/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k, v in d["extra"].items(): print(k, v)
PY
timeout -k 10 300 python tools/runner_epoch.py 3 > gpurun_out/r3f/runner_gas.log 2>&1 || { tail -20 gpurun_out/r3f/runner_gas.log; exit 1; }
tail -3 gpurun_out/r3f/runner_gas.log
timeout -k 10 300 python tools/runner_toy.py > gpurun_out/r3f/runner_toy.log 2>&1 || { tail -20 gpurun_out/r3f/runner_toy.log; exit 1; }
tail -3 gpurun_out/r3f/runner_toy.log
timeout -k 10 300 python tools/runner_epoch.py 3 demiss > gpurun_out/r3f/runner_uci_demiss.log 2>&1 || { tail -20 gpurun_out/r3f/runner_uci_demiss.log; exit 1; }
tail -3 gpurun_out/r3f/runner_uci_demiss.log
