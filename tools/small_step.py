#!/usr/bin/env python
"""The yaml-batch training step (UCI-gas MissIWAE, B = 512, K = 5) on its own: eager and captured in a HIP graph.
python tools/small_step.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demiss_vae_amd as dmv  # noqa: E402
from bench import make_gas  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
K = 5
dims, enc, dec, _, _, X, M = make_gas("cuda", B)
params = enc.params() + dec.params()


def run(fn, label):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    print(f"B={B} {label}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")


opt = torch.optim.Adam(params, lr=1e-3, amsgrad=True, fused=True)


def eager():
    opt.zero_grad(set_to_none=True)
    b, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=1, num_importance_samples=K, STL=True)
    (-b).backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0, foreach=True)
    opt.step()


run(eager, "eager")

from demiss_vae_amd.graph import GraphedStep  # noqa: E402
opt2 = torch.optim.Adam(params, lr=1e-3, amsgrad=True, fused=True, capturable=True)
eps = torch.empty(1, K, B, dims.Zd, device="cuda")


def body():
    opt2.zero_grad(set_to_none=True)
    b, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=1, num_importance_samples=K, STL=True, eps=eps)
    (-b).backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0, foreach=True)
    opt2.step()
    return b


g = GraphedStep(body, refresh=lambda: eps.normal_(), nets=(enc, dec))
run(g, "graph")
print("bound after training:", float(g().detach()))
