#!/bin/bash
# rocprofv3 passes over the small- and mid-batch MLP kernels (tools/kbench.py, gas decoder shapes; rows from the
# arguments, default 2560 and 512): kernel trace and one PMC pass, each a run of its own.
set -u
ROWS="${*:-2560 512}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_s
rm -rf $OUT; mkdir -p $OUT
for rows in $ROWS; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$rows -- python tools/kbench.py $rows > $OUT/trace$rows.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc$rows -- python tools/kbench.py $rows > $OUT/pmc$rows.log 2>&1 || exit 1
done
python - $ROWS <<'PY'
import collections, csv, glob, os, sys
out = ["# rocprofv3 summary: small- and mid-batch MLP kernels (`python tools/kbench.py <rows>`, gas decoder 16-256x4-16)", "",
       "`bash tools/prof_small.sh <rows ...>`: per row count one `rocprofv3 --kernel-trace --stats` run and one `--pmc` run.", ""]
for rows in [int(v) for v in sys.argv[1:]]:
    newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
    out += [f"## {rows} rows", "", "| kernel | calls | avg us | min us |", "|---|---|---|---|"]
    for r in csv.DictReader(open(newest(f"gpurun_out/prof_s/trace{rows}/*/*_kernel_stats.csv"))):
        if r["Name"].startswith(("void k_", "k_")):
            out.append(f"| {r['Name']} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} |")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(f"gpurun_out/prof_s/pmc{rows}/*/*_counter_collection.csv"))):
        if r["Kernel_Name"].startswith(("void k_mlp", "k_mlp", "k_wgrad")):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.append("")
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        out.append(f"- `{k}`: cycles per XCD {cyc:,.0f}; waves x cycles {m['SQ_WAVE_CYCLES']:,.0f}; MFMA busy cycles "
                   f"{m['SQ_VALU_MFMA_BUSY_CYCLES']:,.0f} = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(m['SQ_WAVE_CYCLES'], 1):.2f} of the resident "
                   f"waves' cycles; parked on waitcnt {m['SQ_WAIT_ANY'] / max(m['SQ_WAVE_CYCLES'], 1):.2f}")
    out.append("")
open("gpurun_out/prof_s/summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
