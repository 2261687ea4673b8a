#!/usr/bin/env python
"""The synthetic hidden-512 pair with 784-dimensional data (not a reference configuration) and the mid-size UCI steps that
share the four-wave kernels, one JSON line each.  Usage: python tools/wide_point.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demiss_vae_amd import bench_extra as E  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for B in (200, 4096, 8192):
        r = E.iwae_step(dev, B, name="synthetic 784-d pair, hidden 512 (not a reference config)", D=784, Zd=50,
                        enc_shape=(784, 100, 2, 512), dec_shape=(50, 784, 2, 512), gen_dist="bern", mask_concat=False)
        print(json.dumps(r), flush=True)
    print(json.dumps(E.missiwae_step(dev, "UCI-hepmass", 4096, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256))), flush=True)
    print(json.dumps(E.missiwae_step(dev, "UCI-miniboone", 8192, 43, 32, (86, 325, 5, 256), (32, 86, 2, 256), dropout=0.5)), flush=True)


if __name__ == "__main__":
    main()
