#!/usr/bin/env python
"""Turn the output of tools/prof_r2.sh (gpurun_out/r2prof) into profiles/r2_training_steps.md and copy the raw
rocprofv3 kernel statistics next to it.  `python tools/summarize_steps.py r3` does the same for tools/prof_r3.sh
(see summarize_r3 below)."""
import csv
import json
import shutil
import sys

if len(sys.argv) > 1 and sys.argv[1] == "r3":
    from summarize_steps_r3 import main
    main()
    sys.exit(0)
src = "gpurun_out/r2prof/"
for a, b in (("step_32768_eager", "r2_step_32768"), ("step_512_graph", "r2_step_512"), ("step_hepmass_graph", "r2_step_hepmass"),
             ("step_miniboone_graph", "r2_step_miniboone"), ("runner_epoch", "r2_runner_epoch")):
    shutil.copy(f"{src}{a}_kernel_stats.csv", f"profiles/{b}_kernel_stats.csv")
ep = None
for l in open(src + "runner/run.log"):
    if l.startswith('{"epochs"'):
        ep = json.loads(l)
    if l.startswith('{"steps"'):
        summary = json.loads(l)
steps = ep["steps_per_epoch"]
tab, tot = [], 0.0
for r in csv.DictReader(open(src + "runner_epoch_kernel_stats.csv")):
    per = int(r["Calls"]) / steps
    if per < 0.02:
        continue
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][-60:]
    us = float(r["AverageNs"]) / 1e3
    tot += per * us
    tab.append(f"| {name} | {per:.2f} | {us:.2f} | {per * us:.1f} |")

HEAD = """# Whole training steps under rocprofv3 --kernel-trace --stats (round 2)

`bash tools/prof_r2.sh` (one MI355X), summarised by `python tools/summarize_steps.py`: every trace is a run of its own,
`rocprofv3 --kernel-trace --stats -- python tools/step_trace.py <workload> <eager|graph> <steps>` with one fixed batch size per
trace; 20 warm-up steps are part of every trace (launches / step = calls / (steps + 20)).  Wall time per step is what the
script itself measured under the profiler (profiling slows small steps down; bench.py reports the unprofiled times).  The
raw per-kernel statistics are the `r2_step_*_kernel_stats.csv` and `r2_runner_epoch_kernel_stats.csv` files next to this
note.  `profiles/r1_bench_step_kernel_stats.csv` of the first round mixed batch sizes in one trace and is gone; these traces
replace it.  Up to 49 152 decoder rows the decoder's weight-gradient kernels run on a second stream beside the latent-side
kernel and the encoder's backward pass; kernels that share the chip stretch each other, so in those traces the sum of the
kernel times exceeds the wall time of the step.

What changed in the step since the first round's traces, kernel by kernel (B = 512 replayed step, 33 launches then):
- `k_obj_terms` + `k_obj_bound` + three torch reductions / negations -> one `k_obj_fwd*` launch (terms, logsumexp bound,
  backward weights) plus `k_mean2` (the batch mean and the loss = minus it);
- torch `mul` / `div` / `expand` kernels that scaled the upstream gradient -> folded into `k_obj_bwd_dec*` and
  `k_obj_bwd_latent` (a device scalar and an optional per-datapoint vector);
- torch `zero_` of the encoder-output gradient -> `k_obj_bwd_latent` writes every entry, padding included;
- `k_opt_prep` -> folded into `k_opt_apply` (every workgroup sums the few hundred partial norms itself);
- two `k_pack` launches (re-packing both networks' MFMA fragment images after the optimiser step) -> `k_opt_apply` writes
  each new parameter value straight into the images;
- torch `fill_` of the root gradient (`loss.backward()`) -> the trainer's cached unit gradient;
- torch `add_` on the device-resident Philox counter of a recording -> added on the side by `k_opt_apply`;
- torch `uniform_` + compare + copy kernels that made dropout masks -> keep decisions drawn inside the MLP kernels;
- a replayed step now launches library kernels only.

"""
titles = {"step_32768_eager": "UCI-gas MissIWAE headline step, B = 32768, K = 5, eager (the step bench.py times)",
          "step_512_graph": "the same step at the yaml batch, B = 512, replayed from a HIP graph",
          "step_hepmass_graph": "UCI-hepmass MissSIWAE step, B = 4096, C = 5, replayed from a HIP graph (decoder: 20 480 rows = two "
                                "rounds of four-wave tiles + one round of 16-row tiles)",
          "step_miniboone_graph": "UCI-miniboone MissSIWAE step, B = 8192, C = 5, dropout 0.5 drawn in the kernels, replayed from a HIP "
                                  "graph (decoder: 40 960 rows = one round of one-wave tiles + one round of four-wave tiles)"}
with open("profiles/r2_training_steps.md", "w") as md:
    md.write(HEAD)
    for k, t in titles.items():
        md.write(f"## {t}\n\n" + open(src + k + ".md").read() + "\n")
    md.write(f"""## One epoch of the YAML runner (tools/runner_epoch.py 1: UCI-gas-sized synthetic table, 848 877 kept rows, batch 512, IWAE I = 5
STL, clip 5, cosine schedule, validation pass of 94 310 rows at the end)

{json.dumps(ep)}
{json.dumps(summary)}
launches per training step over the whole epoch (validation included):
| kernel | launches / step | avg us | us / step |
|---|---|---|---|
""" + "\n".join(tab) + f"\nsum of kernel time per training step: {tot:.1f} us\n")
print(open("profiles/r2_training_steps.md").read()[-1500:])
