#!/usr/bin/env python
"""Every bench point of the measurement plan in one run (training steps include Adam; inputs resident on the device):
gas MissIWAE at B 512 / 4096 / 32768, toy DeMissVAE at B 64 / 4096 with K 5 / 25, hepmass and miniboone MissSIWAE at
the configs' batch and 8x it, the K = 50 000 log-likelihood estimator on gas / hepmass / miniboone(C=5) shapes, and the synthetic hidden-512 pair
with 784-dimensional data.
Writes one JSON object.  Usage: python tools/bench_points.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demiss_vae_amd import bench_extra as E  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {"gas_missiwae": [], "toy_demissvae": [], "hepmass_missiwae": [], "miniboone_missiwae": [], "loglik_K50000": []}
    for B in (512, 4096, 32768):
        out["gas_missiwae"].append(E.iwae_step(dev, B))
        print(out["gas_missiwae"][-1], flush=True)
    for B in (512, 4096):       # launch-bound sizes again, replayed from a recording (what the runner does)
        out["gas_missiwae"].append(E.iwae_step(dev, B, graph=True))
        print(out["gas_missiwae"][-1], flush=True)
    # MFMA-sized 784-d layers: NOT a reference configuration (its image models are convolutional)
    out["synthetic_784_h512"] = []
    for B in (200, 4096):
        out["synthetic_784_h512"].append(E.iwae_step(dev, B, name="synthetic 784-d pair, hidden 512 (not a reference config)", D=784,
                                                     Zd=50, enc_shape=(784, 100, 2, 512), dec_shape=(50, 784, 2, 512),
                                                     gen_dist="bern", mask_concat=False))
        print(out["synthetic_784_h512"][-1], flush=True)
    for B in (64, 4096):
        for K in (5, 25):
            out["toy_demissvae"].append(E.demiss_toy_step(dev, B, K))
            print(out["toy_demissvae"][-1], flush=True)
    for B in (512, 4096):
        out["hepmass_missiwae"].append(E.missiwae_step(dev, "UCI-hepmass", B, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256)))
        print(out["hepmass_missiwae"][-1], flush=True)
    for B in (1024, 8192):
        out["miniboone_missiwae"].append(E.missiwae_step(dev, "UCI-miniboone", B, 43, 32, (86, 325, 5, 256), (32, 86, 2, 256),
                                                         dropout=0.5))
        print(out["miniboone_missiwae"][-1], flush=True)
    out["loglik_K50000"].append(E.loglik_eval_rate(dev))
    out["loglik_K50000"].append(E.loglik_eval_rate(dev, name="hepmass", D=21, Zd=16, enc_shape=(42, 32, 2, 256),
                                                   dec_shape=(16, 42, 2, 256)))
    out["loglik_K50000"].append(E.loglik_eval_rate(dev, name="miniboone C=5", D=43, Zd=32, enc_shape=(86, 325, 5, 256),
                                                   dec_shape=(32, 86, 2, 256), C=5))
    for r in out["loglik_K50000"]:
        print(r, flush=True)
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
