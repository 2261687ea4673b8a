"""Secondary workloads reported by bench.py under "extra": the DeMissVAE (toy MoG2-large shapes) training step
with its LAIR imputation update, and the K = 50 000 log-likelihood estimator on UCI-gas shapes."""
import time

import torch

from . import estimate_iwelbo, estimate_multiple_iwelbo, mvb2_objectives, ModelDims, NetDims, PackedNet  # noqa: F401
from .imputation import _keep as keep_for
from .run import _unit_gradient as unit_gradient
from . import imputation as imp
from . import parallel as P
from .optim import Adam
from . import synth
from .synth import NetShape as NetSpec, batch as make_batch


def _sync_time(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _packed(device, spec, seed, requires_grad, dropout=0.0):
    import dataclasses
    return synth.packed_net(dataclasses.replace(spec, dropout=dropout), seed, device, requires_grad)


def missiwae_step(device, name, B, D, Zd, enc_shape, dec_shape, C=5, dropout=0.0):
    """MissSIWAE training step (stratified mixture of C Gaussians as the encoder, one importance sample per
    component, tight bound): configs/uci_hepmass/mis50/miwae_stratified_k5_i1_encm_stl.yaml and the miniboone
    counterpart (dropout 0.5 in both networks).  Fresh dropout masks every step, Adam included."""
    enc_s, dec_s = NetSpec(*enc_shape), NetSpec(*dec_shape)
    enc, dec = _packed(device, enc_s, 1, True, dropout), _packed(device, dec_s, 2, True, dropout)
    dims = ModelDims(D=D, Zd=Zd, q_dist="stratified_with_eps", C=C, encoder_use_mis_mask=True)
    X, M = make_batch(3, B, D)
    X, M = X.to(device), M.to(device)
    opt = Adam(enc.params() + dec.params(), lr=1e-3, amsgrad=True).attach_images((enc, dec))

    def step():
        ek = keep_for(enc, B, True, device)           # as the model classes get them: drawn inside the kernels
        dk = keep_for(dec, B * C, True, device)
        _, _, loss = estimate_multiple_iwelbo(dims, enc, dec, X, M, num_importance_samples_for_each_component=1, STL=True,
                                              enc_keep=ek, dec_keep=dk, with_loss=True)
        opt.zero_grad(set_to_none=True)
        loss.backward(unit_gradient(loss))      # the trainer's cached root gradient: no ones-fill per step
        opt.step()

    t_eager = _sync_time(step, 20, 5)
    # what the trainer runs (trainer.hip_graph, on by default): the step recorded once and replayed; noise and dropout
    # decisions stay fresh through the library's device-resident Philox counter
    from .graph import GraphedStep
    replay = GraphedStep(step, nets=(enc, dec))
    t = _sync_time(replay, 100, 10)
    macs = lambda sp: sum(n * k for n, k in sp.layer_shapes())
    flops = 3 * 2 * B * (macs(enc_s) + C * macs(dec_s))
    out = {"workload": f"{name} MissSIWAE C={C} I_c=1 STL step incl. Adam, B={B}, dropout={dropout}, HIP-graph replay",
           "ms_per_step": round(t * 1e3, 4), "samples_per_s": round(B * C / t, 1), "eager_ms_per_step": round(t_eager * 1e3, 4)}
    if flops:
        out["step_tflops"] = round(flops / t / 1e12, 2)
        out["step_frac_of_fp32_mfma_peak"] = round(flops / t / 1e12 / 157.3, 4)
    return out


def iwae_step(device, B, I=5, *, name="UCI-gas", D=8, Zd=16, enc_shape=(16, 32, 2, 256), dec_shape=(16, 16, 2, 256),
              gen_dist="normal_with_eps", mask_concat=True, graph=False):
    """MissIWAE training step, Adam included; by default on UCI-gas shapes
    (configs/uci_gas/compare/mis50/iwae_i5_encm_stl.yaml)."""
    enc_s, dec_s = NetSpec(*enc_shape), NetSpec(*dec_shape)
    enc, dec = _packed(device, enc_s, 1, True), _packed(device, dec_s, 2, True)
    dims = ModelDims(D=D, Zd=Zd, gen_dist=gen_dist, encoder_use_mis_mask=mask_concat)
    X, M = make_batch(3, B, D)
    if gen_dist.startswith("bern"):                       # about 13 % of the pixels on, as in binarised MNIST
        X = (X > 1.1).float() * 2 - 1 if gen_dist == "bern_m1_p1" else (X > 1.1).float()
    X, M = X.to(device), M.to(device)
    opt = Adam(enc.params() + dec.params(), lr=1e-3, amsgrad=True)

    def step():
        _, _, loss = estimate_iwelbo(dims, enc, dec, X, M, num_importance_samples=I, STL=True, with_loss=True)
        opt.zero_grad(set_to_none=True)
        loss.backward(unit_gradient(loss))      # the trainer's cached root gradient: no ones-fill per step
        opt.step()

    if graph:       # the whole step recorded once and replayed (library noise stays fresh through its device counter)
        from .graph import GraphedStep
        for _ in range(3):
            step()
        replay = GraphedStep(step, nets=(enc, dec))
        t = _sync_time(replay, 200, 20)
        name = name + " (HIP-graph replay)"
    else:
        t = _sync_time(step, 30, 10)
    macs = lambda sp: sum(n * k for n, k in sp.layer_shapes())
    flops = 3 * 2 * B * (macs(enc_s) + I * macs(dec_s))
    return {"workload": f"{name} MissIWAE I={I} STL step incl. Adam, B={B}", "ms_per_step": round(t * 1e3, 4),
            "samples_per_s": round(B * I / t, 1), "step_tflops": round(flops / t / 1e12, 2)}


def demiss_toy_step(device, B=4096, K=5, parts=False, uci=False):
    """toy_mog2_large DeMissVAE (configs/toy_mog2_large/compare/mis50/mvb2vae_lairdmisr0_k5_stl_mlp.yaml shapes:
    D=5, Zd=2, encoder (5,4,3,200), decoder (2,10,3,200), K=5 imputations, LAIR I=1 r=0 T=1, mvb2 + STL):
    gather from the device buffer -> LAIR update with write-back -> two objectives -> two backward calls -> Adam.
    uci: the UCI-gas DeMissVAE config instead (configs/uci_gas/compare/mis50/mvb2vae_lairdmisr1iter10_k5_stl_gradclip1.yaml:
    D=8, Zd=16, encoder (8,32,2,256), decoder (16,16,2,256), normal_with_eps, LAIR with one prior-replenish proposal, ten
    iterations and the final resampling, gradient clipping at norm 1)."""
    if uci:
        enc_s, dec_s, D = NetSpec(8, 32, 2, 256), NetSpec(16, 16, 2, 256), 8
        dims = ModelDims(D=8, Zd=16, gen_dist="normal_with_eps", q_dist="normal_with_eps", encoder_use_mis_mask=False)
        lair_kw = dict(num_iterations=10, num_imp_samples=1, num_prior_replenish_proposals=1, resample_final_imps=True)
        N = 102400
    else:
        enc_s, dec_s, D = NetSpec(5, 4, 3, 200), NetSpec(2, 10, 3, 200), 5
        dims = ModelDims(D=5, Zd=2, gen_dist="normal", q_dist="normal", encoder_use_mis_mask=False)
        lair_kw = dict(num_iterations=1, num_imp_samples=1)
        N = 19366
    enc, dec = _packed(device, enc_s, 1, True), _packed(device, dec_s, 2, True)
    X, M = make_batch(3, N, D)
    buf = imp.ImputationBuffer(X, K, device)
    Mdev = M.to(device)
    params = enc.params() + dec.params()
    opt = Adam(params, lr=1e-3, amsgrad=True).attach_images((enc, dec))
    # batches as the device loader hands them out: index selections of a shuffled split
    order = torch.randperm(N, device=device, generator=torch.Generator(device=device).manual_seed(0))
    index = torch.empty(B, dtype=torch.int64, device=device)
    minus_one = -torch.ones((), device=device)
    state = {"at": 0}

    def next_batch():
        at = state["at"] if state["at"] + B <= N else 0
        index.copy_(order[at:at + B])
        state["at"] = at + B

    def step():
        Xt, idx = buf.gather(index)
        Mk = Mdev[index].unsqueeze(1).expand(B, K, D).contiguous()
        Xt = imp.lair(dims, enc, dec, Xt, Mk, buf=buf, idx_imp=idx, **lair_kw)
        # as the model class steps (vgiwae/models/mvbvae.py, mvb2_manual_optimisation_step): the two objectives as scalar
        # means, ascended through a cached root gradient of -1
        gen_obj, enc_obj = mvb2_objectives(dims, enc, dec, Xt, Mk, num_latent_samples=1, STL=True, reduce_mean=True)
        opt.zero_grad(set_to_none=True)
        gen_obj.backward(minus_one, retain_graph=True, inputs=dec.params())
        enc_obj.backward(minus_one, inputs=enc.params())
        if uci:
            opt.clip_on_next_step(1.0)
        opt.step()

    def eager():
        next_batch()
        step()

    if parts:       # for tracing tools: the step itself
        return step, next_batch, (enc, dec)
    t_eager = _sync_time(eager, 20, 5)
    from .graph import GraphedStep
    replay = GraphedStep(step, refresh=next_batch, nets=(enc, dec))
    t = _sync_time(replay, 100, 10)
    name = "UCI-gas DeMissVAE mvb2 step incl. LAIR update (r=1, 10 iterations, final resampling)" if uci else \
        "toy_mog2_large DeMissVAE mvb2 step incl. LAIR update"
    return {"workload": f"{name}, B={B}, K={K}, HIP-graph replay",
            "ms_per_step": round(t * 1e3, 4), "samples_per_s": round(B * K / t, 1), "eager_ms_per_step": round(t_eager * 1e3, 4)}


def loglik_eval_rate(device, B=50, K=50000, batches=4, name="gas", D=8, Zd=16, enc_shape=(16, 32, 2, 256),
                     dec_shape=(16, 16, 2, 256), C=1, k_shard=None, barrier=None):
    """Refit test step (UCI-gas shapes by default): B=50 test points x K=50 000 importance samples per batch; C > 1:
    a stratified-mixture encoder sampled as a plain mixture.  Under data parallelism every rank takes its own
    batches of test points (no exchange until the final mean); with k_shard = (rank, world) every rank takes the SAME
    points and its contiguous share of the K samples, and the ranks' (max, sum-exp) states are combined by the
    logsumexp all-reduce inside every batch (the returned time is this rank's; the caller takes the maximum)."""
    enc_s, dec_s = NetSpec(*enc_shape), NetSpec(*dec_shape)
    enc, dec = _packed(device, enc_s, 1, False), _packed(device, dec_s, 2, False)
    dims = ModelDims(D=D, Zd=Zd, encoder_use_mis_mask=True, q_dist="stratified_with_eps" if C > 1 else "normal_with_eps", C=C)
    X, _ = make_batch(5 + (P.rank() if k_shard is None else 0), B, D)
    X = X.to(device)
    M = torch.ones_like(X, dtype=torch.bool)

    def run():
        for _ in range(batches):
            imp.estimate_loglik(dims, enc, dec, X, M, num_importance_samples=K, chunk=8192, k_shard=k_shard)

    if k_shard is None:
        t = _sync_time(run, 2, 1)
    else:
        run()
        barrier()
        t = _sync_time(run, 2, 0)
    out = {"workload": f"refit test_step, {name} shapes, B={B} x K={K} per batch", "ms_per_batch": round(t / batches * 1e3, 3),
           "samples_per_s": round(batches * B * K / t, 1)}
    if k_shard is not None:
        out["workload"] += f", the K samples of every point split over {k_shard[1]} ranks (logsumexp all-reduce per batch)"
    return out


def mnist_demiss_step(device, B=200, K=5):
    """DeMissVAE on binarised 28x28 images (configs/mnist_bin/compare/q50/mvb2vae_noencm_pg_k5_stl.yaml: full-size
    convolutional trunks as torch modules, Bernoulli{-1,+1} decoder, mvb2 + STL, one pseudo-Gibbs sweep per step with
    write-back into the device-resident imputation buffer, two backward calls, clip, Adam).  Synthetic images."""
    from vgiwae.models import MVBVAE, MVB_OBJECTIVE
    from vgiwae.shared.vae_enums import DISTRIBUTION
    from vgiwae.shared.vae_resnet import ResNetDecoder, ResNetEncoder
    torch.manual_seed(20220118)
    dec = ResNetDecoder(output_shape=[1, 28, 28], layers=[2, 2, 2, 2], layer_widths=[128, 64, 32, 32],
                        layer_upscale=[7, None, None, None], upscale_factor=7, downscaled_resolution=[4, 4], latent_dim=50,
                        dropout_prob=0.3)
    enc = ResNetEncoder(input_shape=[1, 28, 28], layers=[2, 2, 2, 2], layer_widths=[32, 64, 128, 256], latent_dim=100,
                        dropout_prob=0.3)
    model = MVBVAE(generator_network=dec, generator_distribution=DISTRIBUTION.bern_m1_p1, var_latent_network=enc,
                   var_latent_distribution=DISTRIBUTION.normal_with_eps, num_latent_samples=1, lr_generator=1e-4, lr_latent=1e-4,
                   amsgrad_generator=True, amsgrad_latent=True, var_latent_STL=True, kl_analytic=False,
                   mvb_objective=MVB_OBJECTIVE["mvb2"]).to(device).train()
    N, D = 4000, 784
    g = torch.Generator().manual_seed(1)
    X = (torch.rand(N, D, generator=g) < 0.13).float() * 2 - 1
    M = torch.ones(N, 28, 28, dtype=torch.bool)
    quad = torch.randint(0, 4, (N, 2), generator=g)                 # two (possibly equal) quadrants missing per image
    for q in range(4):
        r, c = (q // 2) * 14, (q % 2) * 14
        sel = (quad == q).any(1)
        M[sel, r:r + 14, c:c + 14] = False
    M = M.reshape(N, D).to(device)
    buf = imp.ImputationBuffer(X, K, device)
    params = list(dec.parameters()) + list(enc.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, amsgrad=True)
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    dims = model.model_dims()

    def step():
        index = torch.randint(0, N, (B,), device=device, generator=gen)
        Xt, idx = buf.gather(index)
        Mk = M[index].unsqueeze(1).expand(B, K, D).contiguous()
        Xt = imp.pseudo_gibbs(dims, model.predict_var_latent_params, dec, Xt, Mk, num_iterations=1, buf=buf, idx_imp=idx)
        gen_obj, enc_obj = model.estimate_mvb2_objectives(Xt, Mk)
        opt.zero_grad(set_to_none=True)
        (-gen_obj.mean()).backward(retain_graph=True, inputs=list(dec.parameters()))
        (-enc_obj.mean()).backward(inputs=list(enc.parameters()))
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()

    t = _sync_time(step, 10, 3)
    return {"workload": f"mnist_bin DeMissVAE mvb2 step (conv trunks in torch/MIOpen), B={B}, K={K}", "ms_per_step": round(t * 1e3, 3),
            "samples_per_s": round(B * K / t, 1)}
