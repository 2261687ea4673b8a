set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3h
DMV_LIB=demiss_vae_amd/libdmv_trace.so timeout -k 10 200 python tools/ktrace.py > gpurun_out/r3h/ktrace.log 2>&1 || { tail -20 gpurun_out/r3h/ktrace.log; exit 1; }
cat gpurun_out/r3h/ktrace.log
timeout -k 10 400 python bench.py --no-cpu > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err || { tail -20 gpurun_out/r3h/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3h/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k, v in d["extra"].items(): print(k, v)
PY
