#!/usr/bin/env python
"""Turn the rocprofv3 output of tools/prof.sh (gpurun_out/prof_k) into profiles/<tag>_*.csv + a short .md."""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof_k"
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(f"{src}/trace/vm/*_kernel_stats.csv")
shutil.copy(stats, f"profiles/{tag}_decoder_kernels_kernel_stats.csv")
import datetime
import subprocess
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = "+uncommitted" if subprocess.run(["git", "status", "--porcelain", "demiss_vae_amd", "bench.py"], capture_output=True,
                                         text=True).stdout.strip() else ""
lines = [f"# rocprofv3 summary ({tag}): `python bench.py --kernels-only` (decoder MLP kernels, 163840 rows)", "",
         f"Collected: {datetime.date.today().isoformat()}, tree at {head}{dirty}, one MI355X (tools/prof.sh)", "",
         "## --kernel-trace --stats", "",
         "avg us: over all launches, as rocprofv3 --stats prints it.  steady avg us: the last 20 launches of each kernel (the",
         "timed ones of `bench.py --kernels-only`, which come after 30 untimed launches), from the per-launch kernel trace.", "",
         "| kernel | calls | avg us | steady avg us | min us | max us |", "|---|---|---|---|---|---|"]
trace = newest(f"{src}/trace/vm/*_kernel_trace.csv")
per_launch = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"])):
    per_launch[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
dur = {}
for r in csv.DictReader(open(stats)):
    if r["Name"].startswith(("void k_", "k_")):
        rest = per_launch.get(r["Name"], [])[-20:]
        steady = sum(rest) / len(rest) if rest else float(r["AverageNs"])
        dur[r["Name"]] = steady
        lines.append(f"| {r['Name']} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {steady/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
lines += ["", "## --pmc (separate passes; averages per launch)", ""]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc1", "pmc2", "pmc3"):
    f = newest(f"{src}/{d}/vm/*_counter_collection.csv")
    shutil.copy(f, f"profiles/{tag}_decoder_kernels_{d}.csv")
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(("void k_mlp", "k_mlp", "k_wgrad")):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    lines.append(f"### {k}")
    for c, x in sorted(m.items()):
        lines.append(f"- {c}: {x:,.0f}")
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        if k in dur:
            lines.append(f"- derived: effective shader clock = GRBM_GUI_ACTIVE / 8 / traced duration = {cyc / dur[k]:.3f} GHz "
                         f"(counter pass and trace pass are separate runs)")
        lines.append(f"- derived: cycles per XCD {cyc:,.0f}; MFMA busy / (1024 SIMDs x cycles) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}; "
                     f"waitcnt-parked share SQ_WAIT_ANY/SQ_WAVE_CYCLES = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
    if "FETCH_SIZE" in m:
        lines.append(f"- derived: HBM read bytes = 2 x FETCH_SIZE KiB (gfx950 correction) = {2 * m['FETCH_SIZE'] * 1024 / 1e6:,.0f} MB; "
                     f"written = {m.get('WRITE_SIZE', 0) * 1024 / 1e6:,.0f} MB")
    lines.append("")
open(f"profiles/{tag}_decoder_kernels.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
