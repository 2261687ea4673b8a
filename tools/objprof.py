#!/usr/bin/env python
"""Driver for profiling the HBM-bound objective kernels on their own: latent sampling, the fused objective forward,
the decoder-side and the latent-side backward, at the two shapes the judge of this work named:

  gas    MissIWAE on UCI-gas shapes: B = 32768 datapoints x I = 5 importance samples = 163 840 rows, D = 8 (Gaussian
         decoder, 16 parameters in rows of pitch 32), Zd = 16, single-Gaussian q
  mb/hep MissSIWAE on UCI-miniboone (B = 8192, D = 43, Zd = 32) / UCI-hepmass (B = 4096, D = 21, Zd = 16) shapes:
         stratified mixture of C = 5 Gaussians, one sample per component
  bern   DeMissVAE on binarised 28 x 28 images: B = 8192 x K = 5 imputations = 40 960 rows, D = 784 logits, Zd = 50,
         imputation-mixture q, the two objectives of mvb2

Run it bare for HIP-event timings, or under rocprofv3 (tools/objprof.sh) for kernel-trace / PMC passes.  It prints one
JSON line per kernel with the algorithmic bytes of a launch (every array read or written once; padding columns that
a consumer reads are counted as written), which tools/summarize_objprof.py divides by the profiler's durations."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demiss_vae_amd import _lib as L  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "gas"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda"
lib, st = L.lib(), L.stream_ptr()
g = torch.Generator(device=dev)
g.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)

cfg = L.DmvObjCfg()
strat = which in ("mb", "hep")
if which == "gas":
    B, G, I, D, Zd, ldE, ldD = 32768, 1, 5, 8, 16, 32, 32
    gen, P = "normal_with_eps", 16
    cfg.x_per_component = 0
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, 0, Zd, 0
    X = rn(B, D)
    M = (torch.rand(B, D, device=dev, generator=g) < 0.5).to(torch.uint8)
    q_kind, mix_C = L.Q_OWN, 1
elif which in ("mb", "hep"):
    # MissSIWAE on UCI-miniboone / UCI-hepmass shapes: stratified mixture of 5 Gaussians, one sample per component
    B, G, I, D, Zd, ldE, ldD = (8192, 5, 1, 43, 32, 352, 96) if which == "mb" else (4096, 5, 1, 21, 16, 192, 64)
    gen, P = "normal_with_eps", 2 * D
    cfg.x_per_component = 0
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, Zd, G * Zd, 2 * G * Zd
    X = rn(B, D)
    M = (torch.rand(B, D, device=dev, generator=g) < 0.5).to(torch.uint8)
    q_kind, mix_C = L.Q_STRATIFIED, G
else:
    B, G, I, D, Zd, ldE, ldD = 8192, 5, 1, 784, 50, 128, 784
    gen, P = "bern_m1_p1", 784
    cfg.x_per_component = 1
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = G * ldE, ldE, Zd, 0
    X = (torch.rand(B, G, D, device=dev, generator=g) < 0.13).float() * 2 - 1
    M = (torch.rand(B, G, D, device=dev, generator=g) < 0.5).to(torch.uint8)
    q_kind, mix_C = L.Q_IMPUTATION, G
cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, G, 1, I, D, Zd
cfg.gen_dist, cfg.q_eps = L.GEN[gen], 1
N = B * G * I
enc = rn(B * G if which == "bern" else B, ldE)
comp_lp = torch.empty(N, G, device=dev) if strat else None
dlogit = torch.empty(B, G, device=dev) if strat else None
dec = rn(N, ldD)
Z = torch.empty(N, Zd, device=dev)
eps = torch.empty(N, Zd, device=dev)
wgt = torch.empty(N, device=dev)
per_b = torch.empty(B, device=dev)
mean = torch.empty((), device=dev)
objA, objB = torch.empty(B, G, device=dev), torch.empty(B, G, device=dev)
terms = torch.empty(5, N, device=dev) if strat else None
d_dec = torch.empty(N, ldD, device=dev)
ldz_pad = 32 * ((Zd + 31) // 32)
dZ = rn(N, ldz_pad)
d_enc = torch.empty_like(enc)
gs = torch.ones((), device=dev)


def sample():
    L.check(lib.dmv_latent_sample(C.byref(cfg), L.ptr(enc), None, 7, 0, None, Zd, None, L.ptr(Z), L.ptr(eps), st))


def forward():
    f = L.DmvObjFwd()
    f.q_kind, f.mix_C = q_kind, mix_C
    if which == "gas":
        f.kind = L.BOUND_LSE
        f.bound, f.wgt = L.addr(per_b), L.addr(wgt)
    elif strat:
        f.kind, f.use_logpi = L.BOUND_LSE, 1
        f.bound, f.wgt, f.comp_lp, f.dlogit = L.addr(per_b), L.addr(wgt), L.addr(comp_lp), L.addr(dlogit)
        f.terms = L.addr(terms)
    else:
        f.kind = L.BOUND_PER_COMPONENT
        f.coefA = (C.c_float * 5)(1, 0, 1, 0, -1)
        f.coefB = (C.c_float * 5)(0, 1, 1, -1, 0)
        f.objA, f.objB = L.addr(objA), L.addr(objB)
    L.check(lib.dmv_obj_forward(C.byref(cfg), C.byref(f), L.ptr(dec), ldD, L.ptr(X), L.ptr(M), L.ptr(Z), Zd, L.ptr(enc), st))


def bwd_dec():
    L.check(lib.dmv_obj_backward_dec(C.byref(cfg), L.ptr(dec), ldD, L.ptr(X), L.ptr(M), L.ptr(wgt) if which != "bern" else None,
                                     None, L.ptr(gs), 1.0 / B, 0.0, L.ptr(d_dec), ldD, st))


def bwd_latent():
    p = L.DmvLatentBwd()
    p.base_prior, p.base_qown, p.base_qmix, p.base_w, p.kl_coef = 1.0 / B, -1.0 / B, 0.0, 1.0, 0.0
    p.own_live, p.mix_live, p.use_zscale, p.mix_kind = 0, 0, 0, L.Q_OWN
    if strat:       # the MissSIWAE step with STL: prior and mixture-q terms, mixture parameters detached
        p.base_qown, p.base_qmix, p.mix_kind = 0.0, -1.0 / B, L.Q_STRATIFIED
        L.check(lib.dmv_obj_backward_latent(C.byref(cfg), C.byref(p), L.ptr(enc), L.ptr(Z), L.ptr(eps), Zd, L.ptr(dZ), ldz_pad,
                                            L.ptr(wgt), None, None, L.ptr(comp_lp), L.ptr(terms[4]), L.ptr(dlogit), None, L.ptr(gs),
                                            L.ptr(d_enc), st))
        return
    L.check(lib.dmv_obj_backward_latent(C.byref(cfg), C.byref(p), L.ptr(enc), L.ptr(Z), L.ptr(eps), Zd, L.ptr(dZ), ldz_pad,
                                        L.ptr(wgt) if which == "gas" else None, None, None, None, None, None, None, L.ptr(gs),
                                        L.ptr(d_enc), st))


rows_enc = enc.shape[0]
xm = X.numel() * 5
if strat:
    sample = lambda: (Z.normal_(generator=g), eps.normal_(generator=g))      # (component-wise sampling is not what is looked at here)
nbytes = {
    "k_latent_sample": rows_enc * 2 * Zd * 4 + 2 * N * Zd * 4,
    "k_obj_fwd": N * (P + Zd) * 4 + xm + rows_enc * 2 * Zd * 4 + (N * 4 + B * 4 if which == "gas" else 2 * B * G * 4),
    "k_obj_bwd_dec": N * P * 4 + xm + (N * 4 if which == "gas" else 0) + N * ldD * 4,
    "k_obj_bwd_latent": 3 * N * Zd * 4 + (N * 4 if which == "gas" else 0) + rows_enc * 2 * Zd * 4 + rows_enc * ldE * 4,
}
if strat:       # + component log-densities, mixture log q, logit gradients; q parameters = all C components and the logits
    nbytes["k_obj_bwd_latent"] = (3 * N * Zd * 4 + N * 4 + N * G * 4 + N * 4 + B * G * 4 + B * (2 * G * Zd + G) * 4 + B * ldE * 4)
cases = [("k_latent_sample", sample), ("k_obj_fwd", forward), ("k_obj_bwd_dec", bwd_dec), ("k_obj_bwd_latent", bwd_latent)]
sample()
wgt.uniform_(0, 1, generator=g)
for name, fn in cases:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(json.dumps({"shape": which, "rows": N, "kernel": name, "bytes": nbytes[name], "event_us": round(us, 2),
                      "event_GBps": round(nbytes[name] / us / 1e3, 1)}), flush=True)
