#!/usr/bin/env python
"""Steps whose weight gradients use 2x2 blocks (up to 24576 rows), for A/B runs of the small-block kernel and its wave
count through the environment (DMV_WGRAD_BIG, DMV_WGRAD_SLOTS_SMALL).  Usage: python tools/ab_wgrad_small.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demiss_vae_amd import bench_extra as E  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cases = {
        "gas B=512 graph": lambda: E.iwae_step(dev, 512, graph=True),
        "gas B=4096 graph": lambda: E.iwae_step(dev, 4096, graph=True),
        "hepmass B=512": lambda: E.missiwae_step(dev, "UCI-hepmass", 512, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256)),
        "hepmass B=4096": lambda: E.missiwae_step(dev, "UCI-hepmass", 4096, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256)),
        "miniboone B=1024": lambda: E.missiwae_step(dev, "UCI-miniboone", 1024, 43, 32, (86, 325, 5, 256), (32, 86, 2, 256),
                                                    dropout=0.5),
        "toy B=4096 K=5": lambda: E.demiss_toy_step(dev, 4096, 5),
        "uci demiss": lambda: E.demiss_toy_step(dev, 512, 5, uci=True),
    }
    for name, fn in cases.items():
        best = min(fn()["ms_per_step"] for _ in range(3))
        print(f"{name}: {best:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
