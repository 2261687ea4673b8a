#!/bin/bash
# kernel trace of tools/step_trace.py; prints kernels per step and their average durations.  Usage: step_trace.sh B mode steps
set -u
B=${1:-512}; MODE=${2:-graph}; STEPS=${3:-200}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/steptrace_${B}_${MODE}
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/step_trace.py $B $MODE $STEPS > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cat $OUT/run.log | grep "ms/step"
python - "$OUT" "$STEPS" <<'PY'
import csv, glob, sys
out, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + "/trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
lines = []
for r in rows:
    calls = int(r["Calls"])
    per = calls / (steps + 20)
    if per < 0.5:
        continue
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][-60:]
    us = float(r["AverageNs"]) / 1e3
    tot += per * us
    lines.append(f"| {name} | {per:.2f} | {us:.2f} | {per * us:.1f} |")
print("| kernel | launches / step | avg us | us / step |\n|---|---|---|---|")
print("\n".join(lines))
print(f"kernels per step: {sum(int(r['Calls']) for r in rows if int(r['Calls']) / (steps + 20) >= 0.5) / (steps + 20):.1f}; sum of kernel time per step: {tot:.1f} us")
PY
