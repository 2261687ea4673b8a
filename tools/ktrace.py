#!/usr/bin/env python
"""Phase timing inside the forward MLP kernel (tuning build: make -C demiss_vae_amd/csrc trace, then
DMV_LIB=demiss_vae_amd/libdmv_trace.so python tools/ktrace.py).  Prints, per phase, the median over waves of the
shader-clock ticks spent (the counter runs at 100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demiss_vae_amd as dmv  # noqa: E402
from demiss_vae_amd import _lib as L  # noqa: E402
from demiss_vae_amd.mlp import mlp_forward  # noqa: E402
from oracle.recipes import NetSpec, make_params  # noqa: E402

rows = 163840
spec = NetSpec(16, 16, 2, 256)
p = make_params(spec, 1)
net = dmv.PackedNet(dmv.NetDims(spec.in_dim, spec.out_dim, spec.nb, spec.hidden),
                    [p[n + ".weight"].cuda() for n in spec.layer_names()], [p[n + ".bias"].cuda() for n in spec.layer_names()])
x = torch.randn(rows, spec.in_dim, device="cuda")
from demiss_vae_amd.mlp import pad32  # noqa: E402
y0, saved0 = mlp_forward(net, x, spec.in_dim, rows, save=True)
gy = torch.randn_like(y0)
gy[:, spec.out_dim:] = 0
nbytes = L.lib().dmv_mlp_backward_ws_bytes(C.byref(net.c), rows)
ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device="cuda")
dx = torch.empty(rows, pad32(spec.in_dim), device="cuda")
FWD = {1: "first layer", 2: "act", 3: "gemm l1", 4: "head+act", 5: "gemm l2", 6: "head+act", 7: "gemm l1", 8: "head+act",
       9: "gemm l2", 10: "head+act", 15: "final layer"}
BWD = {1: "mask+gy gemm", 2: "mask", 3: "gemm W2t", 4: "head+mask", 5: "gemm W1t", 6: "mask", 7: "gemm W2t", 8: "head+mask",
       9: "gemm W1t", 10: "mask", 15: "dx tail"}
for save in (True, False, "bwd"):
    for _ in range(3):
        if save == "bwd":
            L.check(L.lib().dmv_mlp_backward(C.byref(net.c), L.ptr(net.packed()), L.ptr(saved0), None, L.ptr(gy), gy.shape[1],
                                             rows, L.ptr(dx), dx.shape[1], None, None, 0.0, L.ptr(ws), nbytes, L.stream_ptr()))
        else:
            mlp_forward(net, x, spec.in_dim, rows, save=save)
    torch.cuda.synchronize()
    buf = np.zeros(16 * 8192, dtype=np.int64)
    raw = C.CDLL(L.lib_path())
    raw.dmv_debug_trace.argtypes = [C.c_void_p, C.c_size_t]
    assert raw.dmv_debug_trace(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(8192, 16)[: rows // 32]
    names = BWD if save == "bwd" else FWD
    print({True: "fwd save", False: "fwd no save", "bwd": "bwd chain"}[save], " kernel span (ticks):", t[:, 15].max() - t[:, 0].min(),
          " per-wave median total:", np.median(t[:, 15] - t[:, 0]))
    prev = 0
    for i in sorted(names):
        d = t[:, i] - t[:, prev]
        print(f"  {names[i]:12s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
        prev = i
    # constant 100 MHz clock: kernel span, per-wave duration, and how much of the span a SIMD slot sat empty
    r0, r1 = t[:, 13], t[:, 14]
    span = (r1.max() - r0.min()) / 100.0
    dur = (r1 - r0) / 100.0
    print(f"  real time: span {span:.1f} us, per-wave median {np.median(dur):.1f} us (x{len(t) / 1024:.0f} rounds = "
          f"{np.median(dur) * len(t) / 1024:.1f} us), shader clock = {np.median((t[:, 15] - t[:, 0]) / dur):.0f} MHz")
    order = np.argsort(r0)
    print("  start times of waves (us after first), every 512th:", np.round((r0[order][::512] - r0.min()) / 100.0, 1))

# weight-gradient kernel: wave durations by (layer, tile block shape)
gws, gbs = [torch.empty_like(w) for w in net.weights], [torch.empty_like(b) for b in net.biases]
for _ in range(3):
    L.check(L.lib().dmv_mlp_backward(C.byref(net.c), L.ptr(net.packed()), L.ptr(saved0), None, L.ptr(gy), gy.shape[1], rows,
                                     L.ptr(dx), dx.shape[1], L.ptr_array(gws), L.ptr_array(gbs), 0.0, L.ptr(ws), nbytes,
                                     L.stream_ptr()))
torch.cuda.synchronize()
buf = np.zeros(16 * 8192, dtype=np.int64)
assert raw.dmv_debug_trace(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 16)
t = t[(t[:, 14] > t[:, 13]) & (t[:, 1] > 0) & (t[:, 1] < 2000)]
r0 = t[:, 13].min()
print(f"wgrad: {len(t)} waves, span {(t[:, 14].max() - r0) / 100.0:.1f} us")
for key in np.unique(t[:, 1]):
    sel = t[t[:, 1] == key]
    d = (sel[:, 14] - sel[:, 13]) / 100.0
    print(f"  layer {key // 100} block {key % 100 // 10}x{key % 10}: {len(sel):4d} waves, duration us p10 {np.percentile(d, 10):7.1f} "
          f"median {np.median(d):7.1f} max {d.max():7.1f}; start max {(sel[:, 13].max() - r0) / 100.0:7.1f} end max {(sel[:, 14].max() - r0) / 100.0:7.1f}")
