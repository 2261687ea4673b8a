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

# what the numbers mean for the kernels as they stand (kept with the generator so that a re-run keeps it)
NOTES = """
## Reading the tables

Launch count of the gas training step (B = 32768, K = 5, eager, profiles/r2_training_steps.md): 16 library kernels a step -
sample, one fused objective forward (terms + logsumexp bound + backward weights), the mean / loss kernel, two objective
backward kernels, 2 x (MLP forward, backward chain, weight gradients, reduction), two optimiser kernels.  The first round's
step had `k_obj_terms` + `k_obj_bound` + torch reductions in place of the one forward launch.

Where the forward and backward objective kernels stand against the 60 %-of-8-TB/s mark:
- Bernoulli-784 decoder-side backward: 70 % (HBM-bound as intended: 16-byte accesses, one pass).
- Bernoulli-784 forward: 37 %.  One wave reads a 784-wide row as 16-byte pieces and spends the time in the two softplus
  evaluations per pixel (exp + log on the quarter-rate transcendental unit): about 110 issue cycles per 16 bytes loaded.  It is
  bound by instruction issue, not by memory; the PMC traffic equals the algorithmic bytes (no re-reads).
- gas rows (D = 8, Zd = 16; 150 to 250 bytes per sample row): 19 - 42 %.  A launch moves 25 - 40 MB in 10 - 16 us.  At this
  size the arrays of the previous kernel are still in the 256 MB Infinity Cache (the PMC traffic of the forward is above the
  algorithmic bytes only because its 32-float decoder rows are read with their padding), and the duration is set by the
  dependent chain load -> transcendental math -> cross-lane reduction -> store that every thread runs once, with 2.5
  workgroups per CU: too short a kernel to cover HBM latency with more loads in flight.  What was done about it: thread-per-row
  forward with compile-time row shapes (28.7 -> 15.1 us), LDS table of the per-datapoint q parameters, hardware
  sine / cosine / rcp / sqrt, no 64-bit index math, batched unconditional preloads in the latent backward (17 -> 12 us),
  per-dimension decoder backward (16.6 -> 9.2 us).  The four kernels together are 64 us of the 2.67 ms headline step (2.4 %).
- the latent sampler (13 - 19 %) is bound by Philox + Box-Muller arithmetic: 10 rounds of 32 x 32 -> 64-bit multiplies per four
  normals.
"""
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
lines += NOTES.strip("\n").split("\n")
open(f"profiles/{tag}_objective_kernels.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
