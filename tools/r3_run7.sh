set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3g
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/r3g/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/r3g/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r3g/pytest_gpu.log
bash tools/step_trace.sh 512 graph 200 > gpurun_out/r3g/step_512.md 2>&1 || { tail gpurun_out/r3g/step_512.md; exit 1; }
bash tools/step_trace.sh hepmass graph 100 > gpurun_out/r3g/step_hepmass.md 2>&1 || { tail gpurun_out/r3g/step_hepmass.md; exit 1; }
head -3 gpurun_out/r3g/step_512.md; head -1 gpurun_out/r3g/step_hepmass.md
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r3g/uci_demiss_trace
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3g/uci_demiss_trace -- python tools/runner_epoch.py 3 demiss > gpurun_out/r3g/uci_demiss.log 2>&1 || { tail -20 gpurun_out/r3g/uci_demiss.log; exit 1; }
tail -2 gpurun_out/r3g/uci_demiss.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3g/uci_demiss_trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = 600
tot = 0; n = 0
for r in rows:
    per = int(r["Calls"]) / steps
    if per < 0.3: continue
    us = float(r["AverageNs"]) / 1e3
    tot += per * us; n += per
    print(f"| {r['Name'][:70]} | {per:.2f} | {us:.2f} | {per*us:.1f} |")
print("launches per step ~", round(n, 1), " kernel us per step ~", round(tot, 1))
PY
timeout -k 10 300 python tools/runner_toy.py > gpurun_out/r3g/runner_toy.log 2>&1 || { tail -20 gpurun_out/r3g/runner_toy.log; exit 1; }
tail -1 gpurun_out/r3g/runner_toy.log
