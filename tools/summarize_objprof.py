#!/usr/bin/env python
"""Turn the rocprofv3 output of tools/objprof.sh (gpurun_out/objprof_{gas,bern}) into profiles/<tag>_objective_kernels.md
plus the raw kernel-stats CSVs.  Per kernel: average duration from --kernel-trace --stats, algorithmic bytes from the
driver's own JSON lines, achieved GB/s against the 8 TB/s HBM3E peak, and the HBM traffic of the two PMC passes
(FETCH_SIZE doubled: on gfx950 it reports half of a wide coalesced stream; WRITE_SIZE as is; both in KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
KEYS = {"k_latent_sample": "k_latent_sample", "k_obj_fwd": "k_obj_fwd", "k_obj_bwd_dec": "k_obj_bwd_dec", "k_obj_bwd_latent": "k_obj_bwd_latent"}
lines = [f"# Objective kernels under rocprofv3 ({tag})", "",
         "`bash tools/objprof.sh gas` and `bash tools/objprof.sh bern` on one MI355X: `python tools/objprof.py <shape>` under",
         "`rocprofv3 --kernel-trace --stats`, then `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` each in a run of its own.",
         "Algorithmic bytes = every array of the launch read or written once (padding columns a consumer reads count as written).",
         "HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; the factor 2 is the gfx950 correction for wide coalesced reads).",
         "Both working sets (tens to a few hundred MB) partly live in the 256 MB Infinity Cache, so traffic can read below the",
         "algorithmic bytes; achieved GB/s is algorithmic bytes / trace duration.", ""]
for shape, title in (("gas", "UCI-gas MissIWAE: 32768 datapoints x 5 samples = 163 840 rows, D = 8, Zd = 16"),
                     ("bern", "Bernoulli-784 DeMissVAE: 8192 datapoints x 5 imputations = 40 960 rows, D = 784, Zd = 50")):
    src = f"gpurun_out/objprof_{shape}"
    if not os.path.isdir(src):
        continue
    ev = {}
    for l in open(f"{src}/events.jsonl"):
        if l.startswith("{"):
            d = json.loads(l)
            ev[d["kernel"]] = d
    stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0]
    shutil.copy(stats, f"profiles/{tag}_objective_kernels_{shape}_kernel_stats.csv")
    dur = {}
    for r in csv.DictReader(open(stats)):
        n = r["Name"]
        for k, pat in KEYS.items():
            if pat in n and "bwd" not in pat and "bwd" in n:
                continue
            if pat in n:
                dur.setdefault(k, (n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), float(r["AverageNs"]), int(r["Calls"])))
    pmc = collections.defaultdict(dict)
    for cname, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        f = glob.glob(f"{src}/{d}/*/*_counter_collection.csv")[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            for k, pat in KEYS.items():
                n = r["Kernel_Name"]
                if pat in n and not ("bwd" not in pat and "bwd" in n):
                    acc[k].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            pmc[k][cname] = sum(v) / len(v)
    lines += [f"## {title}", "", "| kernel | calls | avg us (trace) | algorithmic MB | GB/s | of 8 TB/s | HBM traffic MB (PMC) | HIP-event us |",
              "|---|---|---|---|---|---|---|---|"]
    for k in KEYS:
        if k not in dur:
            continue
        name, ns, calls = dur[k]
        b = ev[k]["bytes"]
        tr = (2 * pmc[k].get("FETCH_SIZE", 0) + pmc[k].get("WRITE_SIZE", 0)) * 1024 / 1e6
        lines.append(f"| `{name}` | {calls} | {ns / 1e3:.1f} | {b / 1e6:.1f} | {b / ns:.0f} | {b / ns / 8000 * 100:.0f} % | {tr:.1f} | {ev[k]['event_us']} |")
    lines.append("")
open(f"profiles/{tag}_objective_kernels.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
