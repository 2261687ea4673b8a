#!/usr/bin/env python
"""Phase timing inside the one-launch LAIR update (tuning build: make -C demiss_vae_amd/csrc trace, then
DMV_LIB=demiss_vae_amd/libdmv_trace.so python tools/ktrace_lair.py): median over wavefronts of the shader-clock ticks each
phase of the SECOND iteration takes, UCI-gas DeMissVAE shapes (B = 512, K = 5, one replenish proposal)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demiss_vae_amd as dmv  # noqa: E402
from demiss_vae_amd import _lib as L, imputation as imp, synth  # noqa: E402

B, K, D, Zd = 512, 5, int(os.environ.get("KT_D", 8)), 16
enc = synth.packed_net(synth.NetShape(D, 2 * Zd, 2, 256), 1, "cuda", requires_grad=False)
dec = synth.packed_net(synth.NetShape(Zd, 2 * D, 2, 256), 2, "cuda", requires_grad=False)
if os.environ.get("KT_SAME"):       # (experiment: both passes read the same weight image)
    dec = enc
if os.environ.get("KT_SWAP"):       # (experiment, equal shapes: the images change places)
    enc, dec = dec, enc
if os.environ.get("KT_PAD"):        # (experiment: another allocation between uses, so the images sit elsewhere)
    _pad = torch.empty(int(os.environ["KT_PAD"]), dtype=torch.uint8, device="cuda")
    enc = synth.packed_net(synth.NetShape(D, 2 * Zd, 2, 256), 1, "cuda", requires_grad=False)
    dec = synth.packed_net(synth.NetShape(Zd, 2 * D, 2, 256), 2, "cuda", requires_grad=False)
if os.environ.get("KT_AT"):         # (experiment: the two images at chosen byte offsets of one 64 MB arena)
    oe, od = (int(v, 0) for v in os.environ["KT_AT"].split(","))
    arena = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for net, off in ((enc, oe), (dec, od)):
        net._packed = arena[off: off + 4 * net._n_packed].view(torch.float32)
print("weight images at", hex(enc.packed().data_ptr()), hex(dec.packed().data_ptr()), "bytes", enc.packed().numel() * 4)
dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist="normal_with_eps", q_dist="normal_with_eps", encoder_use_mis_mask=False)
X1, M1 = (t.cuda() for t in synth.batch(3, B, D))
X = X1[:, None, :].repeat(1, K, 1).contiguous()
M = M1[:, None, :].repeat(1, K, 1).contiguous()
for _ in range(3):
    imp.lair(dims, enc, dec, X, M, num_iterations=10, num_imp_samples=1, num_prior_replenish_proposals=1, resample_final_imps=True)
torch.cuda.synchronize()
buf = np.zeros(16 * 8192, dtype=np.int64)
raw = C.CDLL(L.lib_path())
raw.dmv_debug_trace.argtypes = [C.c_void_p, C.c_size_t]
assert raw.dmv_debug_trace(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 16)[: 4 * 256]
names = {1: "encoder chain", 2: "proposals + table", 3: "decoder chain", 4: "log-weight terms (+ noise)", 5: "logsumexp (+ uniforms)",
         6: "draws", 7: "tail"}
print("iteration total, median ticks:", np.median(t[:, 7] - t[:, 0]))
if False:
    print("  experiment: second decoder pass", np.median(t[:, 8] - t[:, 3]), " encoder pass after it", np.median(t[:, 9] - t[:, 8]),
          " third decoder pass", np.median(t[:, 10] - t[:, 9]))
    t[:, 4:8] -= (t[:, 10] - t[:, 3])[:, None]
prev = 0
for i in sorted(names):
    d = t[:, i] - t[:, prev]
    print(f"  {names[i]:28s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
    prev = i
