#!/usr/bin/env python
"""Micro-benchmark of the MLP kernels alone (decoder shapes of UCI gas): forward with / without saving
activations, input-gradient chain, weight gradients.  Prints TFLOP/s per kernel."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import event_time, mlp_flops_per_row  # noqa: E402
import demiss_vae_amd as dmv  # noqa: E402
from demiss_vae_amd import _lib as L  # noqa: E402
from demiss_vae_amd.mlp import mlp_forward, pad32  # noqa: E402
from oracle.recipes import NetSpec, make_params  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
spec = NetSpec(*[int(v) for v in sys.argv[2].split(",")]) if len(sys.argv) > 2 else NetSpec(16, 16, 2, 256)
dev = "cuda"
p = make_params(spec, 1)
net = dmv.PackedNet(dmv.NetDims(spec.in_dim, spec.out_dim, spec.nb, spec.hidden),
                    [p[n + ".weight"].to(dev) for n in spec.layer_names()], [p[n + ".bias"].to(dev) for n in spec.layer_names()])
x = torch.randn(rows, spec.in_dim, device=dev)
y, saved = mlp_forward(net, x, spec.in_dim, rows, save=True)
gy = torch.randn_like(y)
gy[:, spec.out_dim:] = 0
lib, st = L.lib(), L.stream_ptr()
nbytes = lib.dmv_mlp_backward_ws_bytes(C.byref(net.c), rows)
ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
dx = torch.empty(rows, pad32(spec.in_dim), device=dev)
gws, gbs = [torch.empty_like(w) for w in net.weights], [torch.empty_like(b) for b in net.biases]
packed = net.packed()


def fwd(sv):
    L.check(lib.dmv_mlp_forward(C.byref(net.c), L.ptr(packed), 0, spec.in_dim, L.ptr(x), spec.in_dim, None, 0, rows,
                                L.ptr(y), y.shape[1], L.ptr(sv), None, None, st))


def bwd(with_w):
    L.check(lib.dmv_mlp_backward(C.byref(net.c), L.ptr(packed), L.ptr(saved), None, L.ptr(gy), gy.shape[1], rows,
                                 L.ptr(dx), dx.shape[1], L.ptr_array(gws) if with_w else None,
                                 L.ptr_array(gbs) if with_w else None, 0.0, L.ptr(ws), nbytes, st))


fl = rows * mlp_flops_per_row(spec)
t_fs = event_time(lambda: fwd(saved), 10)
t_fn = event_time(lambda: fwd(None), 10)
t_c = event_time(lambda: bwd(False), 10)
t_cw = event_time(lambda: bwd(True), 10)
for name, t in (("fwd+save", t_fs), ("fwd", t_fn), ("bwd chain", t_c), ("wgrad", t_cw - t_c)):
    print(f"{name:10s} {t * 1e3:8.3f} ms  {fl / t / 1e12:7.2f} TFLOP/s  {fl / t / 1e12 / 157.3 * 100:5.1f} %")
