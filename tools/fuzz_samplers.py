#!/usr/bin/env python
"""Random shapes through the one-launch sampler updates against the launches they replace (one MI355X):
dmv_pgibbs_update must give the same bits; dmv_lair_update the same log-weights (rel 5e-5) on every datapoint whose
draws agreed so far and the same imputations on all but a few datapoints.  Prints one line per case and a summary;
exit status 1 on a mismatch.  Usage: python tools/fuzz_samplers.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demiss_vae_amd as dmv  # noqa: E402
from demiss_vae_amd import imputation as imp, rng, synth  # noqa: E402


def case(r):
    K = r.choice([1, 2, 3, 5, 8])
    n_prior = r.choice([0, 0, 1, 2, 3])
    n_prior = min(n_prior, 16 - K)
    T = r.choice([1, 2, 3, 5])
    D = r.choice([3, 5, 8, 13, 20, 33])
    Zd = r.choice([2, 5, 16, 20, 40])
    hidden = r.choice([32, 64, 200, 256])
    nb = r.choice([1, 2, 3])
    B = r.choice([1, 7, 64, 301])
    gen = r.choice(["normal", "normal_with_eps", "bern", "bern_m1_p1"])
    q = r.choice(["normal", "normal_with_eps"])
    concat = r.random() < 0.5
    clip = gen.startswith("normal") and r.random() < 0.4
    final = T > 1 and r.random() < 0.5
    return dict(K=K, n_prior=n_prior, T=T, D=D, Zd=Zd, hidden=hidden, nb=nb, B=B, gen=gen, q=q, concat=concat, clip=clip, final=final)


def build(c, seed):
    D, Zd = c["D"], c["Zd"]
    normal = c["gen"].startswith("normal")
    enc = synth.packed_net(synth.NetShape(2 * D if c["concat"] else D, 2 * Zd, c["nb"], c["hidden"]), seed, "cuda", requires_grad=False)
    dec = synth.packed_net(synth.NetShape(Zd, 2 * D if normal else D, c["nb"], c["hidden"]), seed + 1, "cuda", requires_grad=False)
    dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist=c["gen"], q_dist=c["q"], encoder_use_mis_mask=c["concat"])
    X1, M1 = (t.cuda() for t in synth.batch(seed + 2, c["B"], D))
    g = torch.Generator(device="cuda").manual_seed(seed + 3)
    X = X1[:, None, :].repeat(1, c["K"], 1).contiguous()
    fill = torch.randn(X.shape, device="cuda", generator=g)
    X = torch.where(M1[:, None, :], X, fill)
    if c["gen"] == "bern":
        X = (X > 0.3).float()
    elif c["gen"] == "bern_m1_p1":
        X = (X > 0.3).float() * 2 - 1
    M = M1[:, None, :].repeat(1, c["K"], 1).contiguous()
    return dims, enc, dec, X, M


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = ones = 0
    for i in range(n):
        c = case(r)
        dims, enc, dec, X, M = build(c, 100 + 7 * i)
        D = c["D"]
        ck = dict(clip_min=torch.full((D,), -1.2) if c["clip"] else None, clip_max=torch.full((D,), 1.3) if c["clip"] else None)
        # ---- pseudo-Gibbs: bit-identical
        outs = []
        for one in (True, False):
            imp.ONE_LAUNCH_PGIBBS = one
            rng.set_seed(31 + i)
            outs.append(imp.pseudo_gibbs(dims, enc, dec, X, M, num_iterations=c["T"], **ck))
        imp.ONE_LAUNCH_PGIBBS = True
        pg_ok = torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())
        # ---- LAIR
        kw = dict(num_iterations=c["T"], num_imp_samples=1, num_prior_replenish_proposals=c["n_prior"],
                  resample_final_imps=c["final"], **ck)
        outs, lws = [], []
        for one in (True, False):
            imp.ONE_LAUNCH_LAIR, imp.LAIR_DEBUG = one, {}
            rng.set_seed(77 + i)
            outs.append(imp.lair(dims, enc, dec, X, M, **kw))
            lws.append(imp.LAIR_DEBUG["log_w"])
            if one:
                took_one = bool(imp.LAIR_DEBUG.get("one_launch"))
        imp.ONE_LAUNCH_LAIR, imp.LAIR_DEBUG = True, None
        B, Kp = c["B"], c["K"] + c["n_prior"]
        a, b = (w.view(c["T"], B, Kp) for w in lws)
        same = torch.ones(B, dtype=torch.bool, device="cuda")
        worst = 0.0
        for t in range(c["T"]):
            err = ((a[t] - b[t]).abs() / (1.0 + b[t].abs())).amax(1)
            err = torch.where(torch.isfinite(err), err, torch.full_like(err, 1.0))
            if bool(same.any()):
                worst = max(worst, float(err[same].max()))
            same &= err < 5e-5
        parted = int((~same).sum())
        diff_dp = int((outs[0] != outs[1]).flatten(1).any(1).sum())
        allow = max(1, int(0.04 * B))
        lair_ok = worst < 5e-5 and parted <= allow and diff_dp <= allow + parted and bool(torch.isfinite(outs[0]).all())
        ones += int(took_one)
        print(f"{i:3d} {'ok ' if pg_ok and lair_ok else 'BAD'} one_launch={int(took_one)} pgibbs={'same' if pg_ok else 'DIFF'} lair: worst_lw={worst:.1e} parted={parted} "
              f"diff_dp={diff_dp}/{B}  {c}", flush=True)
        bad += 0 if (pg_ok and lair_ok) else 1
    print(f"{n - bad} of {n} cases agree; {ones} took the one-launch LAIR update")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
