#!/usr/bin/env python
"""One training step of the UCI-gas MissIWAE model (K = 5, STL, clip + AMSGrad through the library's optimiser) at a
given batch size, eager or replayed from a HIP graph, for rocprofv3 kernel traces:
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/step_trace.py 512 graph 200
Prints the wall time per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demiss_vae_amd as dmv  # noqa: E402
from bench import make_gas  # noqa: E402
from demiss_vae_amd.optim import Adam
from demiss_vae_amd.run import _unit_gradient as unit_gradient  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "512"
mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
K = 5
refresh = None
if what == "toy":
    # toy_mog2_large DeMissVAE: LAIR update + two objectives + two backward passes + Adam, B = 4096, K = 5
    from demiss_vae_amd.bench_extra import demiss_toy_step
    step, refresh, (enc, dec) = demiss_toy_step(torch.device("cuda"), parts=True)
    B = 4096
elif what == "toy64":
    # the same model at the batch size its config trains with
    from demiss_vae_amd.bench_extra import demiss_toy_step
    step, refresh, (enc, dec) = demiss_toy_step(torch.device("cuda"), B=64, K=5, parts=True)
    B = 64
elif what == "uci":
    # UCI-gas DeMissVAE: LAIR update (one prior-replenish proposal, ten iterations, final resampling) + two objectives +
    # two backward passes + clip + Adam, B = 512, K = 5
    from demiss_vae_amd.bench_extra import demiss_toy_step
    step, refresh, (enc, dec) = demiss_toy_step(torch.device("cuda"), B=512, K=5, parts=True, uci=True)
    B = 512
elif what == "wide":
    # synthetic hidden-512 pair with 784-dimensional Bernoulli data (not a reference configuration), MissIWAE I = 5, B = 4096
    from demiss_vae_amd import synth
    B, D, Zd = 4096, 784, 50
    enc = synth.packed_net(synth.NetShape(784, 100, 2, 512), 1, "cuda")
    dec = synth.packed_net(synth.NetShape(50, 784, 2, 512), 2, "cuda")
    dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist="bern", encoder_use_mis_mask=False)
    X, M = (t.cuda() for t in synth.batch(3, B, D))
    X = (X > 1.1).float()
    opt = Adam(enc.params() + dec.params(), lr=1e-3, amsgrad=True).attach_images((enc, dec))

    def step():
        _, _, loss = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_importance_samples=K, STL=True, with_loss=True)
        opt.zero_grad(set_to_none=True)
        loss.backward(unit_gradient(loss))
        opt.step()
elif what in ("hepmass", "miniboone", "hepmass512", "miniboone1024"):
    # MissSIWAE (stratified mixture of 5, one sample per component): configs/uci_hepmass, configs/uci_miniboone shapes
    from demiss_vae_amd import synth
    from demiss_vae_amd.imputation import _keep as keep_for
    B, D, Zd, es, ds, p = (4096, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256), 0.0) if what.startswith("hepmass") else \
        (8192, 43, 32, (86, 325, 5, 256), (32, 86, 2, 256), 0.5)
    if what in ("hepmass512", "miniboone1024"):      # the batch sizes of the configs themselves
        B = int(what[-4:].lstrip("s"))
    enc = synth.packed_net(synth.NetShape(*es, dropout=p), 1, "cuda")
    dec = synth.packed_net(synth.NetShape(*ds, dropout=p), 2, "cuda")
    dims = dmv.ModelDims(D=D, Zd=Zd, q_dist="stratified_with_eps", C=5, encoder_use_mis_mask=True)
    X, M = (t.cuda() for t in synth.batch(3, B, D))
    opt = Adam(enc.params() + dec.params(), lr=1e-3, amsgrad=True).attach_images((enc, dec))

    def step():
        ek, dk = keep_for(enc, B, True, "cuda"), keep_for(dec, B * 5, True, "cuda")      # what the model classes pass
        _, _, loss = dmv.estimate_multiple_iwelbo(dims, enc, dec, X, M, num_importance_samples_for_each_component=1, STL=True,
                                                  enc_keep=ek, dec_keep=dk, with_loss=True)
        opt.zero_grad(set_to_none=True)
        loss.backward(unit_gradient(loss))      # the trainer's cached root gradient: no ones-fill per step
        opt.step()
else:
    B = int(what)
    dims, enc, dec, _, _, X, M = make_gas("cuda", B)
    opt = Adam([{"params": dec.params(), "amsgrad": True, "lr": 1e-3}, {"params": enc.params(), "amsgrad": True, "lr": 1e-3}]).attach_images((enc, dec))

    def step():
        opt.zero_grad(set_to_none=True)
        _, _, loss = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=1, num_importance_samples=K, STL=True,
                                         with_loss=True)
        loss.backward(unit_gradient(loss))      # the trainer's cached root gradient: no ones-fill per step
        opt.clip_on_next_step(5.0)
        opt.step()


fn = dmv.GraphedStep(step, refresh=refresh, nets=(enc, dec)) if mode == "graph" else (step if refresh is None else (lambda: (refresh(), step())))
for _ in range(20):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    fn()
torch.cuda.synchronize()
print(f"B={B} {mode}: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step over {steps} steps")
