#!/usr/bin/env python
"""Random shapes and objectives through the one-launch encoder-sample-decoder forward and the one-launch backward against
the separate launches (one MI355X): same noise stream, bounds equal to 1e-6, every parameter gradient to 1e-5 of its
tensor's largest entry.  Usage: python tools/fuzz_one_launch.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demiss_vae_amd as dmv  # noqa: E402
from demiss_vae_amd import objectives as O, rng, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for i in range(n):
        kind = r.choice(["iwae", "elbo", "elbo_kl", "miwae", "miwae_looser", "mvae", "cvi", "mvb2", "mvb2_kl", "mvb"])
        B = r.choice([1, 5, 33, 64, 200])
        D = r.choice([3, 5, 8, 21])
        Zd = r.choice([2, 5, 16, 20])
        hidden = r.choice([32, 64, 200, 256])
        nb = r.choice([1, 2, 3])
        nz = r.choice([1, 2])
        I = r.choice([1, 2, 5]) if kind in ("iwae", "miwae", "miwae_looser") else 1
        K = r.choice([1, 2, 3, 5])
        stl, dreg = r.random() < 0.5, False
        if kind == "iwae" and not stl:
            dreg = r.random() < 0.5
        gen = r.choice(["normal", "normal_with_eps", "bern"])
        strat = kind in ("miwae", "miwae_looser", "mvae")
        demiss = kind in ("cvi", "mvb2", "mvb2_kl", "mvb")
        concat = (not demiss) and r.random() < 0.7
        normal = gen.startswith("normal")
        C = K if strat else 1
        enc_out = (2 * Zd + 1) * C if strat else 2 * Zd
        dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist=gen, q_dist="stratified_with_eps" if strat else "normal_with_eps", C=C,
                             encoder_use_mis_mask=concat)
        g = torch.Generator().manual_seed(1000 + i)
        shape = (B, K, D) if demiss else (B, D)
        X = torch.randn(*shape, generator=g)
        if not normal:
            X = (X > 0.3).float()
        X, M = X.cuda(), (torch.rand(*shape, generator=g) < 0.6).cuda()
        desc = dict(kind=kind, B=B, D=D, Zd=Zd, hidden=hidden, nb=nb, nz=nz, I=I, K=K, stl=stl, dreg=dreg, gen=gen, concat=concat)

        def run(fused):
            enc = synth.packed_net(synth.NetShape(2 * D if concat else D, enc_out, nb, hidden), 3 * i + 1, "cuda")
            dec = synth.packed_net(synth.NetShape(Zd, 2 * D if normal else D, nb, hidden), 3 * i + 2, "cuda")
            rng.set_seed(9 + i)
            O.FUSE_SMALL_BACKWARD = O.FUSE_SMALL_FORWARD = fused
            try:
                if kind == "iwae":
                    b, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=nz, num_importance_samples=I, STL=stl, DREG=dreg)
                    (-b).backward()
                elif kind in ("elbo", "elbo_kl"):
                    b, _ = dmv.estimate_elbo(dims, enc, dec, X, M, num_latent_samples=nz, kl_analytic=kind == "elbo_kl")
                    (-b).backward()
                elif kind in ("miwae", "miwae_looser"):
                    b, _ = dmv.estimate_multiple_iwelbo(dims, enc, dec, X, M, num_latent_samples=nz,
                                                        num_importance_samples_for_each_component=I, STL=stl,
                                                        looser=kind == "miwae_looser")
                    (-b).backward()
                elif kind == "mvae":
                    b, _ = dmv.estimate_multiple_elbo(dims, enc, dec, X, M, num_latent_samples=nz)
                    (-b).backward()
                elif kind == "cvi":
                    b = O.cvi_objective(dims, enc, dec, X, M, num_latent_samples=nz, STL=stl).mean()
                    (-b).backward()
                elif kind in ("mvb2", "mvb2_kl"):
                    go, eo = O.mvb2_objectives(dims, enc, dec, X, M, num_latent_samples=nz, kl_analytic=kind == "mvb2_kl",
                                               **({} if kind == "mvb2_kl" else dict(STL=stl)))
                    b = go.mean() + eo.mean()
                    (-go.mean()).backward(retain_graph=True, inputs=dec.params())
                    (-eo.mean()).backward(inputs=enc.params())
                else:
                    b = O.mvb_objective(dims, enc, dec, X, M, num_latent_samples=nz).mean()
                    (-b).backward()
            finally:
                O.FUSE_SMALL_BACKWARD = O.FUSE_SMALL_FORWARD = True
            torch.cuda.synchronize()
            return float(b), [t.grad.clone() for t in enc.params() + dec.params()]

        try:
            (b1, one), (b3, three) = run(True), run(False)
        except Exception as e:              # noqa: BLE001
            print(f"{i:3d} ERR {type(e).__name__}: {e}  {desc}", flush=True)
            bad += 1
            continue
        worst = 0.0
        for a, b in zip(one, three):
            worst = max(worst, float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30))
        fin = all(bool(torch.isfinite(a).all()) for a in one)
        ok = fin and worst <= 1e-5 and abs(b1 - b3) <= 1e-6 * (1 + abs(b3))
        print(f"{i:3d} {'ok ' if ok else 'BAD'} bound {b1:.6f} / {b3:.6f} worst grad diff {worst:.1e}  {desc}", flush=True)
        bad += 0 if ok else 1
    print(f"{n - bad} of {n} cases agree")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
