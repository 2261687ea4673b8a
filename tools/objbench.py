#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound objective / imputation kernels on the Bernoulli-784 shapes of the
MNIST / Omniglot DeMissVAE configs (D = 784 logits per row, Zd = 50, K = 5 imputations): achieved GB/s of
algorithmic bytes against the 8 TB/s HBM3E peak.  Usage: python tools/objbench.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import event_time  # noqa: E402
from demiss_vae_amd import _lib as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K, D, Zd = 5, 784, 50
N = B * K
dev = "cuda"
lib, st = L.lib(), L.stream_ptr()
g = torch.Generator(device=dev)
g.manual_seed(0)
dec = torch.randn(N, D, device=dev, generator=g)
X = (torch.rand(B, K, D, device=dev, generator=g) < 0.13).float() * 2 - 1
M = (torch.rand(B, K, D, device=dev, generator=g) < 0.5).to(torch.uint8)
Z = torch.randn(N, Zd, device=dev, generator=g)
ldE = 128
enc = torch.randn(N, ldE, device=dev, generator=g)
terms = torch.empty(5, N, device=dev)
ddec = torch.empty(N, D, device=dev)
wgt = torch.rand(N, device=dev, generator=g)
Xo = torch.empty(B, K, D, device=dev)
buf = torch.zeros(4 * N, D, device=dev)
idx = torch.randperm(4 * N, device=dev, generator=g)[:N].contiguous()
gathered = torch.empty(N, D, device=dev)

cfg = L.DmvObjCfg()
cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, K, 1, 1, D, Zd
cfg.gen_dist, cfg.q_eps, cfg.x_per_component = L.GEN["bern_m1_p1"], 1, 1
cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = K * ldE, ldE, Zd, 0


def terms_fn():
    L.check(lib.dmv_obj_terms(C.byref(cfg), L.Q_IMPUTATION, K, L.ptr(dec), D, L.ptr(X), L.ptr(M), L.ptr(Z), Zd, L.ptr(enc),
                              L.ptr(terms), None, st))


def bwd_fn():
    L.check(lib.dmv_obj_backward_dec(C.byref(cfg), L.ptr(dec), D, L.ptr(X), L.ptr(M), L.ptr(wgt), None, None, 1.0, 0.0,
                                     L.ptr(ddec), D, st))


def tail_fn():
    L.check(lib.dmv_impute_tail(N, D, L.GEN["bern_m1_p1"], L.ptr(dec), D, None, L.ptr(X), L.ptr(M), None, 1, 0, None, None, None, 0,
                                L.ptr(Xo), L.ptr(buf), L.ptr(idx), st))


def gather_fn():
    L.check(lib.dmv_rows_gather(L.ptr(buf), L.ptr(idx), N, D, L.ptr(gathered), st))


cases = [
    ("k_obj_terms (bern 784, imputation mixture K=5)", terms_fn, N * (D * 4 + D * 4 + D + Zd * 4 + 20) + N * 2 * Zd * 4),
    ("k_obj_bwd_dec (bern 784)", bwd_fn, N * (D * 4 + D * 4 + D + 4 + D * 4)),
    ("k_impute_tail (bern 784, Philox, buffer write-back)", tail_fn, N * (D * 4 + D * 4 + D + 2 * D * 4)),
    ("k_rows_gather (784)", gather_fn, N * (2 * D * 4 + 8)),
]
print(f"rows N = B*K = {N}, D = {D}")
for name, fn, nbytes in cases:
    t = event_time(fn, 20)
    print(f"{name:55s} {t * 1e6:8.1f} us  {nbytes / t / 1e9:8.1f} GB/s  {nbytes / t / 8e12 * 100:5.1f} % of 8 TB/s")
