#!/usr/bin/env python
"""gpurun_out/r3prof (tools/prof_r3.sh) -> profiles/r3_training_steps.md plus the raw rocprofv3 kernel statistics next
to it.  Besides the per-kernel tables it prints, for the replayed steps, the timeline of one step (every launch with
its start, end and hardware queue) from the kernel trace: concurrent kernels stretch each other, which the averages hide."""
import csv
import json
import shutil

SRC = "gpurun_out/r3prof/"
HEAD = """# Whole training steps under rocprofv3 --kernel-trace --stats (round 3)

`bash tools/prof_r3.sh` (one MI355X), summarised by `python tools/summarize_steps.py r3`: every trace is a run of its own,
`rocprofv3 --kernel-trace --stats -- python tools/step_trace.py <workload> <eager|graph> <steps>` with one fixed batch size per
trace; 20 warm-up steps are part of every trace (launches / step = calls / (steps + 20)).  Wall time per step is what the
script itself measured under the profiler (profiling slows small steps down; bench.py reports the unprofiled times).  Raw
per-kernel statistics: the `r3_step_*_kernel_stats.csv`, `r3_runner_*_kernel_stats.csv` files next to this note.

What changed in the steps since `profiles/r2_training_steps.md`, kernel by kernel:
- small batches (at most 4096 sample rows): encoder forward + `k_latent_sample*` + decoder forward -> one launch,
  `k_vae_fwd_rows16` (every 16-row tile of samples recomputes the encoder rows it draws from; encoder outputs and latents
  stay in LDS between the stages);
- `k_obj_bwd_dec*` + decoder input-gradient chain + `k_obj_bwd_latent` + encoder input-gradient chain -> one launch,
  `k_vae_bwd_rows16` (a tile = a few encoder rows and all their samples, so the latent-side sum over a row's samples
  stays in the workgroup); the two networks' weight-gradient GEMMs follow side by side on two streams;
- `k_mean2` -> the last workgroup of the objective forward forms the batch mean and the loss (ticket counter; launches of
  at most 256 workgroups);
- one Gaussian per datapoint (IWAE / VAE): `k_obj_fwd*` -> an epilogue of `k_vae_fwd_rows16` (datapoint-major tiles; the
  bounds are published with atomic exchanges so that the mean needs no device-scope release);
- stratified / imputation mixtures: `k_obj_bwd_latent` -> `k_obj_bwd_latent_mix` (parameters, samples and
  responsibilities of a datapoint in LDS; `profiles/r3_latent_mixture_kernel.md`);
- LAIR update: per iteration encoder + two `cat` + sample + decoder + objective + draw + row map + tail (9 launches), then
  `stack` / `permute` / `gather` and four more launches for the final resampling -> ONE launch for the whole update,
  `k_lair_rows16` (a workgroup walks its datapoints through all iterations; encoder outputs, proposals, decoder outputs,
  log-weights and the proposals kept for the final resampling stay in LDS).  Shapes outside its limits (several importance
  samples per component, more than 16 proposals per datapoint, dropout to draw) take three launches per iteration: the
  one-launch forward with the prior-replenish proposals, the objective forward, and the draws + sampling tail
  (`k_lair_resample_tail`);
- hidden widths 257 - 512 (synthetic pair, not a reference configuration): four-wave 32-row tiles beyond 4096 rows
  (`k_mlp_fwd_coop<16>`, `k_mlp_bwd_coop<16>`: one full layer input + the wave's own slice of the residual stream in
  registers), a last partial round on 16-row tiles.

"""
TITLES = {
    "step_32768_eager": "UCI-gas MissIWAE headline step, B = 32768, K = 5, eager (the step bench.py times)",
    "step_512_graph": "the same step at the yaml batch, B = 512, replayed from a HIP graph",
    "step_hepmass_graph": "UCI-hepmass MissSIWAE step, B = 4096, C = 5, replayed from a HIP graph (decoder: 20 480 rows = two "
                          "rounds of four-wave tiles + one round of 16-row tiles)",
    "step_miniboone_graph": "UCI-miniboone MissSIWAE step, B = 8192, C = 5, dropout 0.5 drawn in the kernels, replayed from a HIP "
                            "graph (decoder: 40 960 rows = one round of one-wave tiles + one round of four-wave tiles)",
    "step_toy_graph": "toy_mog2_large DeMissVAE mvb2 step with its LAIR update (I = 1, no replenish, one iteration), B = 4096, "
                      "K = 5, replayed from a HIP graph",
    "step_uci_graph": "UCI-gas DeMissVAE mvb2 step with its LAIR update (one prior-replenish proposal, ten iterations, final "
                      "resampling), B = 512, K = 5, clip 1, replayed from a HIP graph",
    "step_wide_graph": "synthetic hidden-512 pair with 784-dimensional Bernoulli data (NOT a reference configuration), MissIWAE "
                       "I = 5, B = 4096 (decoder 20 480 rows, hidden 512), replayed from a HIP graph",
}


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][-60:]


def timeline(path, marker="k_opt_apply"):
    """launches between the last two optimiser applications of a trace = one whole step"""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(ends) < 2:
        return ""
    seg = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    out = ["| start us | end us | us | queue | kernel |", "|---|---|---|---|---|"]
    for r in seg:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        out.append(f"| {s / 1e3:.1f} | {e / 1e3:.1f} | {(e - s) / 1e3:.1f} | {r['Queue_Id']} | {short(r['Kernel_Name'])} |")
    return "\n".join(out) + "\n"


def runner_table(stats, steps):
    tab, tot, n = [], 0.0, 0.0
    for r in csv.DictReader(open(stats)):
        per = int(r["Calls"]) / steps
        if per < 0.02:
            continue
        us = float(r["AverageNs"]) / 1e3
        tot += per * us
        n += per
        tab.append(f"| {short(r['Name'])} | {per:.2f} | {us:.2f} | {per * us:.1f} |")
    return "| kernel | launches / step | avg us | us / step |\n|---|---|---|---|\n" + "\n".join(tab) + \
        f"\nlaunches per training step: {n:.1f}; sum of kernel time per training step: {tot:.1f} us\n"


def main():
    for k in TITLES:
        shutil.copy(f"{SRC}{k}_kernel_stats.csv", f"profiles/r3_{k}_kernel_stats.csv")
    for k in ("runner_epoch", "runner_uci_demiss"):
        shutil.copy(f"{SRC}{k}_kernel_stats.csv", f"profiles/r3_{k}_kernel_stats.csv")
    with open("profiles/r3_training_steps.md", "w") as md:
        md.write(HEAD)
        for k, t in TITLES.items():
            md.write(f"## {t}\n\n" + open(SRC + k + ".md").read() + "\n")
            if k != "step_32768_eager":
                md.write("One step of that trace, launch by launch (time from the step's first launch; queue = hardware queue, "
                         "i.e. stream):\n\n" + timeline(f"{SRC}{k}_kernel_trace.csv") + "\n")
        for k, title in (("runner_epoch", "One epoch of the YAML runner (tools/runner_epoch.py 1: UCI-gas-sized synthetic table, "
                                          "848 877 kept rows, batch 512, IWAE I = 5 STL, clip 5, cosine schedule, validation pass "
                                          "at the end)"),
                         ("runner_uci_demiss", "Three epochs of the UCI DeMissVAE config through the YAML runner (tools/runner_epoch.py 3 "
                                               "demiss: 102 400-row table, batch 512, K = 5, LAIR with one prior-replenish proposal, "
                                               "ten iterations and the final resampling, mvb2 objectives, validation each epoch)")):
            ep = summary = None
            for l in open(f"{SRC}{k}/run.log"):
                if l.startswith('{"epochs"'):
                    ep = json.loads(l)
                if l.startswith('{"steps"'):
                    summary = json.loads(l)
            steps = ep["steps_per_epoch"] * ep["epochs"]
            md.write(f"## {title}\n\n{json.dumps(ep)}\n{json.dumps(summary)}\nlaunches per training step over the whole run "
                     f"(validation included):\n" + runner_table(f"{SRC}{k}_kernel_stats.csv", steps) + "\n")
    print(open("profiles/r3_training_steps.md").read()[-3000:])


if __name__ == "__main__":
    main()
