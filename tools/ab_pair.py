#!/usr/bin/env python
"""A/B of the two networks' weight gradients after a one-launch backward: one pair of launches on the step's stream,
or side by side on two streams.  Small-batch steps replayed from HIP graphs, three alternations each.
Usage: python tools/ab_pair.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demiss_vae_amd import bench_extra as E, objectives as O  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cases = {
        "gas B=512 graph": lambda: E.iwae_step(dev, 512, graph=True),
        "gas B=512 eager": lambda: E.iwae_step(dev, 512),
        "hepmass B=512": lambda: E.missiwae_step(dev, "UCI-hepmass", 512, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256)),
        "miniboone B=1024": lambda: E.missiwae_step(dev, "UCI-miniboone", 1024, 43, 32, (86, 325, 5, 256), (32, 86, 2, 256),
                                                    dropout=0.5),
        "gas B=1024 graph": lambda: E.iwae_step(dev, 1024, graph=True),
        "hepmass B=1024": lambda: E.missiwae_step(dev, "UCI-hepmass", 1024, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256)),
        "toy B=64 K=5": lambda: E.demiss_toy_step(dev, 64, 5),
        "uci demiss": lambda: E.demiss_toy_step(dev, 512, 5, uci=True),
    }
    if len(sys.argv) > 1:      # quick: the paired launch only, the cases it changes
        for name in ("gas B=512 graph", "hepmass B=512", "miniboone B=1024"):
            for rep in range(3):
                print(f"{name}: pair=True {cases[name]()['ms_per_step']:.4f} ms", flush=True)
        return
    for name, fn in cases.items():
        for rep in range(3):
            for flag in (False, True):
                O.PAIR_WGRAD = flag
                r = fn()
                print(f"{name}: pair={flag} {r['ms_per_step']:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
