#!/usr/bin/env python
"""Up to how many decoder rows the separate-launch backward should end in one pair of weight-gradient launches
(objectives.PAIR_ROWS_MAX) instead of two streams.  Usage: python tools/ab_pair_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demiss_vae_amd import bench_extra as E, objectives as O  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cases = {
        "gas B=1024 graph": lambda: E.iwae_step(dev, 1024, graph=True),
        "gas B=2048 graph": lambda: E.iwae_step(dev, 2048, graph=True),
        "gas B=4096 graph": lambda: E.iwae_step(dev, 4096, graph=True),
        "gas B=8192": lambda: E.iwae_step(dev, 8192),
        "hepmass B=4096": lambda: E.missiwae_step(dev, "UCI-hepmass", 4096, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256)),
    }
    for name, fn in cases.items():
        for rep in range(2):
            for rows in (0, 1 << 20):
                O.PAIR_ROWS_MAX = rows
                print(f"{name}: pair up to {rows} rows {fn()['ms_per_step']:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
