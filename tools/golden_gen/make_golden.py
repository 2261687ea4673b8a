#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: needs the reference checkout (default /root/reference), which is
imported, never copied.  `pytorch_lightning` is not installed, so the plumbing-only
stand-in under tools/golden_gen/pl_stub is put on sys.path; every number written here is
computed by the reference's own functions (vgiwae.models.*, vgiwae.shared.*,
vgiwae.utils.imputation_distribution, vgiwae.data.*_missingness_provider).

Random draws are injected rather than recorded: the three sampling primitives that
torch.distributions uses (`_standard_normal` for rsample, `torch.normal` for sample,
`torch.bernoulli`) and `F.dropout` are replaced for the duration of a case by functions
that consume prepared noise tensors, with the same arithmetic the stock primitive applies
to its own noise (loc + eps*scale, u < p, x*mask/(1-p)).  Categorical draws are taken from
torch itself and recorded.

Usage:  python tools/golden_gen/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import contextlib
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle.recipes import NetSpec, make_params, make_batch, make_noise, make_uniform  # noqa: E402


# ----------------------------------------------------------------------------- noise injection
class Noise:
    """Queues of prepared noise; installed over torch's sampling primitives."""

    def __init__(self):
        self.normal = []      # standard normal tensors, consumed by rsample / sample
        self.uniform = []     # U[0,1) tensors, consumed by torch.bernoulli
        self.dropout = []     # 0/1 keep masks
        self.cat_draws = []   # recorded Categorical.sample outputs
        self.normal_shapes = []

    def _pop_normal(self, shape):
        t = self.normal.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        self.normal_shapes.append(tuple(shape))
        return t


@contextlib.contextmanager
def injected(noise: Noise):
    import torch.distributions.normal as tdn
    import torch.nn.functional as F
    from torch.distributions import Categorical

    orig = (tdn._standard_normal, torch.normal, torch.bernoulli, F.dropout, Categorical.sample)

    def std_normal(shape, dtype, device):
        return noise._pop_normal(shape).to(dtype)

    def normal(mean, std, *a, **k):
        return mean + std * noise._pop_normal(mean.shape).to(mean.dtype)

    def bernoulli(p, *a, **k):
        u = noise.uniform.pop(0)
        assert u.shape == p.shape, (u.shape, p.shape)
        return (u < p).to(p.dtype)

    def dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        m = noise.dropout.pop(0)
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * m.to(x.dtype) / (1.0 - p)

    def cat_sample(self, sample_shape=torch.Size()):
        out = orig[4](self, sample_shape)
        noise.cat_draws.append(out.clone())
        return out

    tdn._standard_normal = std_normal
    torch.normal = normal
    torch.bernoulli = bernoulli
    F.dropout = dropout
    Categorical.sample = cat_sample
    try:
        yield noise
    finally:
        tdn._standard_normal, torch.normal, torch.bernoulli, F.dropout, Categorical.sample = orig


# ----------------------------------------------------------------------------- helpers
def ref_net(spec: NetSpec, params):
    from vgiwae.shared.neural_nets import ResidualFCNetwork
    net = ResidualFCNetwork(spec.in_dim, spec.out_dim, spec.nb, spec.hidden,
                            activation=spec.act, dropout_probability=spec.dropout)
    net.load_state_dict({k: v.clone() for k, v in params.items()})
    return net


def dist_enum(name, C=1):
    from vgiwae.shared.vae_enums import DISTRIBUTION
    if name.startswith("stratified"):
        suffix = "_with_eps" if name.endswith("with_eps") else ""
        return DISTRIBUTION[f"stratified_mixture{C}_normal{suffix}"]
    return DISTRIBUTION[name]


def digest(g: torch.Tensor):
    """Full tensor when small, otherwise row/column sums plus a strided sub-sample."""
    g = g.detach()
    if g.numel() <= 8192:
        return {"full": g.numpy()}
    return {"rowsum": g.sum(1).numpy(), "colsum": g.sum(0).numpy(), "sub": g[::7, ::5].numpy()}


def grads_to_dict(prefix, net):
    out = {}
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        for k, v in digest(p.grad).items():
            out[f"{prefix}.{n}.{k}"] = v
    return out


def params_to_dict(prefix, params):
    return {f"{prefix}.{k}": v.numpy() for k, v in params.items()}


def save(out_dir, name, **arrays):
    path = os.path.join(out_dir, name + ".npz")
    clean = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        clean[k] = np.asarray(v)
    np.savez_compressed(path, **clean)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


def build_model(cls_name, enc_spec, dec_spec, enc_p, dec_p, gen_dist, q_dist, C=1, **kw):
    import vgiwae.models as models
    cls = getattr(models, cls_name)
    model = cls(generator_network=ref_net(dec_spec, dec_p),
                generator_distribution=dist_enum(gen_dist),
                var_latent_network=ref_net(enc_spec, enc_p),
                var_latent_distribution=dist_enum(q_dist, C),
                num_latent_samples=kw.pop("num_latent_samples", 1),
                lr_generator=1e-3, lr_latent=1e-3, **kw)
    model.train()
    return model


def dropout_masks(seed, spec: NetSpec, rows_shape):
    if spec.dropout <= 0:
        return []
    rng = np.random.Generator(np.random.PCG64(seed))
    return [torch.from_numpy((rng.random(tuple(rows_shape) + (spec.hidden,)) >= spec.dropout))
            for _ in range(spec.nb)]


# ----------------------------------------------------------------------------- cases
def relu_kink_margin(spec: NetSpec, params, x):
    """per row, the smallest |pre-activation| over all hidden layers (float64)"""
    p = {k: v.double() for k, v in params.items()}
    a = x.double() @ p["initial_layer.weight"].T + p["initial_layer.bias"]
    closest = a.abs().min(dim=1).values
    a = torch.relu(a)
    for b in range(spec.nb):
        t = a @ p[f"blocks.{b}.linear_layers.0.weight"].T + p[f"blocks.{b}.linear_layers.0.bias"]
        closest = torch.minimum(closest, t.abs().min(dim=1).values)
        u = torch.relu(t) @ p[f"blocks.{b}.linear_layers.1.weight"].T + p[f"blocks.{b}.linear_layers.1.bias"] + a
        closest = torch.minimum(closest, u.abs().min(dim=1).values)
        a = torch.relu(u)
    return closest


def case_mlp(out, name, spec: NetSpec, rows, seed, store_params=True, digest_io=False, kink_margin=None):
    """A2: ResidualFCNetwork forward + backward against a fixed upstream gradient.  digest_io: outputs and input
    gradients of long batches are stored as row / column sums plus a strided sub-sample.  kink_margin: of the `rows`
    drawn rows keep those none of whose hidden pre-activations lies within the margin of zero (among tens of
    thousands of rows a few would otherwise sit within fp32 rounding of a relu kink, where two correct fp32
    implementations pick different derivatives for the row and the weight gradients inherit the difference); the
    kept rows are recorded bit-packed so that the readers rebuild exactly this batch."""
    params = make_params(spec, seed)
    net = ref_net(spec, params)
    net.train()
    x = make_noise(seed + 1, rows, spec.in_dim)
    gy = make_noise(seed + 2, rows, spec.out_dim)
    extra = {}
    if kink_margin is not None:
        assert spec.dropout == 0
        ok = relu_kink_margin(spec, params, x) > kink_margin
        extra = dict(rows_drawn=rows, row_ok=np.packbits(ok.numpy().astype(np.uint8)), kink_margin=kink_margin)
        x, gy, rows = x[ok].contiguous(), gy[ok].contiguous(), int(ok.sum())
    x.requires_grad_(True)
    noise = Noise()
    masks = dropout_masks(seed + 3, spec, (rows,))
    noise.dropout = [m.clone() for m in masks]
    with injected(noise):
        y = net(x)
    y.backward(gy)
    arrays = dict(seed=seed, rows=rows, spec=np.array([spec.in_dim, spec.out_dim, spec.nb, spec.hidden]),
                  dropout=spec.dropout, **extra)
    if digest_io:
        for key, t in (("y", y), ("dx", x.grad)):
            for k, v in digest(t).items():
                arrays[f"{key}.{k}"] = v
    else:
        arrays.update(y=y, dx=x.grad)
    if store_params:
        arrays.update(params_to_dict("p", params))
        arrays.update(x=x, gy=gy)
        for b, m in enumerate(masks):
            arrays[f"mask{b}"] = m
    arrays.update(grads_to_dict("g", net))
    save(out, name, **arrays)


def _iwae_like(out, name, *, D, Zd, enc_spec, dec_spec, B, I, nz, seed, gen_dist, q_dist,
               mask_concat, STL, DREG, binary=False, store_inputs=True):
    from vgiwae.models.iwae import estimate_iwelbo
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    model = build_model("IWAE", enc_spec, dec_spec, enc_p, dec_p, gen_dist, q_dist,
                        encoder_use_mis_mask=mask_concat, var_latent_STL=STL, var_latent_DREG=DREG,
                        num_importance_samples=I, num_latent_samples=nz)
    X, M = make_batch(seed + 2, B, D, binary=binary)
    eps = make_noise(seed + 3, nz, I, B, Zd)
    noise = Noise()
    noise.normal = [eps.clone()]
    noise.dropout = ([m for m in dropout_masks(seed + 4, enc_spec, (B,))]
                     + [m for m in dropout_masks(seed + 5, dec_spec, (nz, I, B))])
    with injected(noise):
        bound = estimate_iwelbo(model, (X, M), var_latent_STL=STL, var_latent_DREG=DREG,
                                num_latent_samples=nz, num_importance_samples=I)
    (-bound).backward()
    arrays = dict(seed=seed, B=B, I=I, nz=nz, D=D, Zd=Zd, STL=STL, DREG=DREG, mask_concat=mask_concat,
                  gen_dist=gen_dist, q_dist=q_dist, binary=binary,
                  enc_spec=np.array([enc_spec.in_dim, enc_spec.out_dim, enc_spec.nb, enc_spec.hidden]),
                  dec_spec=np.array([dec_spec.in_dim, dec_spec.out_dim, dec_spec.nb, dec_spec.hidden]),
                  enc_dropout=enc_spec.dropout, dec_dropout=dec_spec.dropout,
                  bound=bound)
    if store_inputs:
        arrays.update(X=X, M=M, eps=eps)
        arrays.update(params_to_dict("enc", enc_p))
        arrays.update(params_to_dict("dec", dec_p))
    arrays.update(grads_to_dict("genc", model.var_latent_network))
    arrays.update(grads_to_dict("gdec", model.generator_network))
    save(out, name, **arrays)


def cases_iwae(out):
    enc_s = lambda i, o, h=32, nb=2, p=0.0: NetSpec(i, o, nb, h, dropout=p)
    k = 0
    for STL, DREG, tag in [(True, False, "stl"), (False, True, "dreg"), (False, False, "plain")]:
        for q in ["normal", "normal_with_eps"]:
            for mc in [True, False]:
                D, Zd = 8, 16
                _iwae_like(out, f"iwae_{tag}_{q}_{'mc' if mc else 'nomc'}", D=D, Zd=Zd,
                           enc_spec=enc_s(2 * D if mc else D, 2 * Zd), dec_spec=enc_s(Zd, 2 * D),
                           B=16, I=5, nz=2 if tag == "plain" else 1, seed=1000 + 10 * k,
                           gen_dist=q, q_dist=q, mask_concat=mc, STL=STL, DREG=DREG)
                k += 1
    # dropout inside the residual blocks (every MiniBooNE network uses p = 0.5)
    _iwae_like(out, "iwae_stl_dropout", D=8, Zd=16, enc_spec=enc_s(16, 32, p=0.5), dec_spec=enc_s(16, 16, p=0.5),
               B=16, I=5, nz=1, seed=1300, gen_dist="normal_with_eps", q_dist="normal_with_eps",
               mask_concat=True, STL=True, DREG=False)
    # Bernoulli decoders, 784-d
    _iwae_like(out, "iwae_stl_bern_m1_p1_784", D=784, Zd=50, enc_spec=enc_s(784 * 2, 100), dec_spec=enc_s(50, 784),
               B=6, I=5, nz=1, seed=1400, gen_dist="bern_m1_p1", q_dist="normal_with_eps",
               mask_concat=True, STL=True, DREG=False, binary=True, store_inputs=False)
    # UCI-gas shapes at full width (B=512, h=256): inputs rebuilt from the seed recipe
    _iwae_like(out, "iwae_gas_full", D=8, Zd=16, enc_spec=NetSpec(16, 32, 2, 256), dec_spec=NetSpec(16, 16, 2, 256),
               B=512, I=5, nz=1, seed=20220118, gen_dist="normal_with_eps", q_dist="normal_with_eps",
               mask_concat=True, STL=True, DREG=False, store_inputs=False)


def cases_vae(out):
    from vgiwae.models.vae import estimate_elbo
    D, Zd, B = 8, 16, 16
    k = 0
    for kl_analytic, STL, nz, tag in [(True, False, 3, "analytic"), (False, False, 2, "mc"), (False, True, 2, "mc_stl")]:
        seed = 2000 + 10 * k
        k += 1
        enc_spec, dec_spec = NetSpec(2 * D, 2 * Zd, 2, 32), NetSpec(Zd, 2 * D, 2, 32)
        enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
        model = build_model("VAE", enc_spec, dec_spec, enc_p, dec_p, "normal_with_eps", "normal_with_eps",
                            encoder_use_mis_mask=True, var_latent_STL=STL, kl_analytic=kl_analytic,
                            num_latent_samples=nz)
        X, M = make_batch(seed + 2, B, D)
        eps = make_noise(seed + 3, nz, B, Zd)
        noise = Noise()
        noise.normal = [eps.clone()]
        with injected(noise):
            bound = estimate_elbo(model, (X, M), kl_analytic=kl_analytic, var_latent_STL=STL, num_latent_samples=nz)
        (-bound).backward()
        arrays = dict(seed=seed, B=B, nz=nz, D=D, Zd=Zd, STL=STL, kl_analytic=kl_analytic, bound=bound,
                      X=X, M=M, eps=eps, enc_spec=np.array([2 * D, 2 * Zd, 2, 32]), dec_spec=np.array([Zd, 2 * D, 2, 32]))
        arrays.update(params_to_dict("enc", enc_p))
        arrays.update(params_to_dict("dec", dec_p))
        arrays.update(grads_to_dict("genc", model.var_latent_network))
        arrays.update(grads_to_dict("gdec", model.generator_network))
        save(out, f"vae_{tag}", **arrays)


def _multiple(out, name, *, cls, D, Zd, C, enc_spec, dec_spec, B, I, nz, seed, STL, looser=False, store_inputs=True):
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    kw = dict(encoder_use_mis_mask=True, var_latent_STL=STL, num_latent_samples=nz)
    if cls == "MultipleIWAE":
        kw.update(num_importance_samples_for_each_component=I, use_looser_bound=looser)
    else:
        kw.update(kl_analytic=False)
    model = build_model(cls, enc_spec, dec_spec, enc_p, dec_p, "normal_with_eps", "stratified_with_eps", C=C, **kw)
    X, M = make_batch(seed + 2, B, D)
    noise = Noise()
    # dropout inside the residual blocks: the encoder's masks are drawn first, then the decoder's (order of the calls)
    dec_rows = (nz, I, B, C) if cls == "MultipleIWAE" else (nz, B, C)
    noise.dropout = dropout_masks(seed + 4, enc_spec, (B,)) + dropout_masks(seed + 5, dec_spec, dec_rows)
    if cls == "MultipleIWAE":
        from vgiwae.models.multiple_iwae import estimate_multiple_iwelbo
        eps = make_noise(seed + 3, nz, I, B, C, Zd)
        noise.normal = [eps.clone()]
        with injected(noise):
            bound = estimate_multiple_iwelbo(model, (X, M), var_latent_STL=STL, num_latent_samples=nz,
                                             num_importance_samples_for_each_component=I, use_looser_bound=looser)
    else:
        from vgiwae.models.multiple_vae import estimate_multiple_elbo
        eps = make_noise(seed + 3, nz, B, C, Zd)
        noise.normal = [eps.clone()]
        model.hparams["log_var_distribution_mixture_component_weights_epoch_frequency"] = -1
        with injected(noise):
            bound = estimate_multiple_elbo(model, (X, M), kl_analytic=False, var_latent_STL=STL, num_latent_samples=nz)
    (-bound).backward()
    arrays = dict(seed=seed, B=B, I=I, nz=nz, D=D, Zd=Zd, C=C, STL=STL, looser=looser, bound=bound,
                  enc_dropout=enc_spec.dropout, dec_dropout=dec_spec.dropout,
                  enc_spec=np.array([enc_spec.in_dim, enc_spec.out_dim, enc_spec.nb, enc_spec.hidden]),
                  dec_spec=np.array([dec_spec.in_dim, dec_spec.out_dim, dec_spec.nb, dec_spec.hidden]))
    if store_inputs:
        arrays.update(X=X, M=M, eps=eps)
        arrays.update(params_to_dict("enc", enc_p))
        arrays.update(params_to_dict("dec", dec_p))
    arrays.update(grads_to_dict("genc", model.var_latent_network))
    arrays.update(grads_to_dict("gdec", model.generator_network))
    save(out, name, **arrays)


def cases_multiple(out):
    D, Zd = 21, 16
    k = 0
    for C in (1, 5):
        P = 2 * Zd * C + C
        for STL in (True, False):
            for looser in (False, True):
                _multiple(out, f"miwae_C{C}_{'stl' if STL else 'plain'}_{'looser' if looser else 'tight'}",
                          cls="MultipleIWAE", D=D, Zd=Zd, C=C, enc_spec=NetSpec(2 * D, P, 2, 32),
                          dec_spec=NetSpec(Zd, 2 * D, 2, 32), B=12, I=2 if looser else 1, nz=1,
                          seed=3000 + 10 * k, STL=STL, looser=looser)
                k += 1
            _multiple(out, f"mvae_C{C}_{'stl' if STL else 'plain'}", cls="MultipleVAE", D=D, Zd=Zd, C=C,
                      enc_spec=NetSpec(2 * D, P, 2, 32), dec_spec=NetSpec(Zd, 2 * D, 2, 32), B=12, I=1, nz=2,
                      seed=3000 + 10 * k, STL=STL)
            k += 1
    # HEPMASS shapes at full width
    _multiple(out, "miwae_hepmass_full", cls="MultipleIWAE", D=21, Zd=16, C=5,
              enc_spec=NetSpec(42, 165, 2, 256), dec_spec=NetSpec(16, 42, 2, 256), B=512, I=1, nz=1,
              seed=20220119, STL=True, store_inputs=False)
    # MiniBooNE shapes at full width: five residual blocks in the encoder, dropout 0.5 in both networks
    # (configs/uci_miniboone/compare/mis50/miwae_stratified_k5_i1_encm_stl.yaml), yaml batch 1024
    _multiple(out, "miwae_miniboone_full", cls="MultipleIWAE", D=43, Zd=32, C=5,
              enc_spec=NetSpec(86, 325, 5, 256, dropout=0.5), dec_spec=NetSpec(32, 86, 2, 256, dropout=0.5), B=1024, I=1, nz=1,
              seed=20220120, STL=True, store_inputs=False)


def _mvb(out, name, *, D, Zd, K, enc_spec, dec_spec, B, nz, seed, objective, STL, kl_analytic=False,
         gen_dist="normal", q_dist="normal", mask_concat=False, binary=False, store_inputs=True):
    from vgiwae.models.mvbvae import MVB_OBJECTIVE
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    model = build_model("MVBVAE", enc_spec, dec_spec, enc_p, dec_p, gen_dist, q_dist,
                        encoder_use_mis_mask=mask_concat, var_latent_STL=STL, kl_analytic=kl_analytic,
                        num_latent_samples=nz, mvb_objective=MVB_OBJECTIVE[objective])
    X, M = make_batch(seed + 2, B, D, binary=binary)
    # imputed copies: observed entries kept, missing ones filled per imputation
    fill, _ = make_batch(seed + 6, B * K, D, binary=binary)
    Mk = M[:, None, :].expand(B, K, D).contiguous()
    Xt = torch.where(Mk, X[:, None, :].expand(B, K, D), fill.reshape(B, K, D)).contiguous()
    eps = make_noise(seed + 3, nz, B, K, Zd)
    noise = Noise()
    noise.normal = [eps.clone()]
    arrays = dict(seed=seed, B=B, K=K, nz=nz, D=D, Zd=Zd, STL=STL, kl_analytic=kl_analytic, objective=objective,
                  gen_dist=gen_dist, q_dist=q_dist, mask_concat=mask_concat, binary=binary,
                  enc_spec=np.array([enc_spec.in_dim, enc_spec.out_dim, enc_spec.nb, enc_spec.hidden]),
                  dec_spec=np.array([dec_spec.in_dim, dec_spec.out_dim, dec_spec.nb, dec_spec.hidden]))
    enc, dec = model.var_latent_network, model.generator_network
    with injected(noise):
        if objective == "mvb2":
            gen_obj, enc_obj = model.estimate_mvb2_objectives(Xt, Mk)
        else:
            gen_obj = model.estimate_cvi_objective(Xt, Mk)
            enc_obj = None
    arrays["gen_obj"] = gen_obj
    if objective == "mvb2":
        arrays["enc_obj"] = enc_obj
        # the two backward calls of vgiwae/shared/mvb2_manual_optimisation.py:6-8
        (-gen_obj.mean()).backward(retain_graph=True, inputs=list(dec.parameters()))
        (-enc_obj.mean()).backward(inputs=list(enc.parameters()))
    else:
        (-gen_obj.mean()).backward()
    if store_inputs:
        arrays.update(X=X, M=M, Xt=Xt, eps=eps)
        arrays.update(params_to_dict("enc", enc_p))
        arrays.update(params_to_dict("dec", dec_p))
    arrays.update(grads_to_dict("genc", enc))
    arrays.update(grads_to_dict("gdec", dec))
    save(out, name, **arrays)


def cases_mvb(out):
    D, Zd = 5, 2
    k = 0
    for objective in ("mvb2", "cvi"):
        for K in (1, 5):
            for STL in (True, False):
                _mvb(out, f"mvb_{objective}_K{K}_{'stl' if STL else 'plain'}", D=D, Zd=Zd, K=K,
                     enc_spec=NetSpec(D, 2 * Zd, 2, 32), dec_spec=NetSpec(Zd, 2 * D, 2, 32), B=10, nz=2,
                     seed=4000 + 10 * k, objective=objective, STL=STL)
                k += 1
    _mvb(out, "mvb_mvb2_K5_analytic", D=D, Zd=Zd, K=5, enc_spec=NetSpec(D, 2 * Zd, 2, 32),
         dec_spec=NetSpec(Zd, 2 * D, 2, 32), B=10, nz=1, seed=4200, objective="mvb2", STL=False, kl_analytic=True)
    # Bernoulli{-1,+1} decoder with mask-concat encoder (MNIST / Omniglot objective shapes)
    _mvb(out, "mvb_mvb2_K5_bern784", D=784, Zd=50, K=5, enc_spec=NetSpec(2 * 784, 100, 1, 32),
         dec_spec=NetSpec(50, 784, 1, 32), B=4, nz=1, seed=4300, objective="mvb2", STL=True,
         gen_dist="bern_m1_p1", q_dist="normal_with_eps", mask_concat=True, binary=True, store_inputs=False)
    # toy MoG2-large shapes at full width
    _mvb(out, "mvb_toy_full", D=5, Zd=2, K=5, enc_spec=NetSpec(5, 4, 3, 200), dec_spec=NetSpec(2, 10, 3, 200),
         B=64, nz=1, seed=20220120, objective="mvb2", STL=True, store_inputs=False)


def _mvbiwae(out, name, *, objective, enc_w, dec_w, STL=False, DREG=False, nz=2, seed):
    """Importance-weighted DeMissVAE objectives (vgiwae/models/mvbiwae.py) on [nz, I, B, K] samples."""
    from vgiwae.models.mvbiwae import MVB_IWAE_OBJ, IS_WEIGHTS
    D, Zd, K, I, B = 5, 2, 3, 4, 6
    enc_spec, dec_spec = NetSpec(D, 2 * Zd, 1, 16), NetSpec(Zd, 2 * D, 1, 16)
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    model = build_model("MVBIWAE", enc_spec, dec_spec, enc_p, dec_p, "normal", "normal", var_latent_STL=STL,
                        var_latent_DREG=DREG, num_latent_samples=nz, num_importance_samples=I,
                        mvb_objective=MVB_IWAE_OBJ[objective], enc_isweights=IS_WEIGHTS[enc_w], dec_isweights=IS_WEIGHTS[dec_w])
    X, M = make_batch(seed + 2, B, D)
    fill, _ = make_batch(seed + 6, B * K, D)
    Mk = M[:, None, :].expand(B, K, D).contiguous()
    Xt = torch.where(Mk, X[:, None, :].expand(B, K, D), fill.reshape(B, K, D)).contiguous()
    eps = make_noise(seed + 3, nz, I, B, K, Zd)
    noise = Noise()
    noise.normal = [eps.clone()]
    enc, dec = model.var_latent_network, model.generator_network
    arrays = dict(seed=seed, B=B, K=K, I=I, nz=nz, D=D, Zd=Zd, STL=STL, DREG=DREG, objective=objective, enc_w=enc_w, dec_w=dec_w,
                  enc_spec=np.array([enc_spec.in_dim, enc_spec.out_dim, enc_spec.nb, enc_spec.hidden]),
                  dec_spec=np.array([dec_spec.in_dim, dec_spec.out_dim, dec_spec.nb, dec_spec.hidden]), Xt=Xt, M=Mk, eps=eps)
    with injected(noise):
        if objective.startswith("mvb2"):
            gen_obj, enc_obj = model.estimate_mvb2_objectives(Xt, Mk)
        elif objective.startswith("mvb"):
            gen_obj = enc_obj = model.estimate_mvb_objective(Xt, Mk)
        else:
            gen_obj = enc_obj = model.estimate_cvi_objective(Xt, Mk)
    arrays.update(gen_obj=gen_obj, enc_obj=enc_obj)
    if noise.cat_draws:
        arrays["resample_idx"] = noise.cat_draws[0]
    if objective.startswith("mvb2"):
        (-gen_obj.mean()).backward(retain_graph=True, inputs=list(dec.parameters()))
        (-enc_obj.mean()).backward(inputs=list(enc.parameters()))
    else:
        (-gen_obj.mean()).backward()
    arrays.update(params_to_dict("enc", enc_p))
    arrays.update(params_to_dict("dec", dec_p))
    arrays.update(grads_to_dict("genc", enc))
    arrays.update(grads_to_dict("gdec", dec))
    save(out, name, **arrays)


def cases_mvbiwae(out):
    cases = [("mvb_indep", "dmmis", "dmmis", "stl", 2), ("mvb_all", "smis", "smis", "plain", 2),
             ("cvi_indep", "smis", "smis", "dreg", 2), ("cvi_all", "dmmis", "dmmis", "plain", 2),
             ("mvb2_encindep_decall", "smis", "dmmis", "stl", 2), ("mvb2_encindep_decindep", "dmmis", "dmmis", "plain", 2),
             ("mvb2_encall_decall", "dmmis", "smis", "dreg", 2),
             ("mvb2_encallweightedsmis_decall", "dmmis", "dmmis", "stl", 2),
             ("mvb2_encallweighteddmmis_decall", "smis", "dmmis", "dreg", 2),
             ("mvb2_encallweighteddmmis_decall", "dmmis", "dmmis", "plain", 2),
             ("mvb2_all_encweighteddmmis_decweighteddmmis", "dmmis", "dmmis", "dreg", 2),
             ("mvb2_all_encweighteddmmis_decweighteddmmis", "dmmis", "dmmis", "stl", 1),
             ("mvb2_encallweighteddmmis_decallresampl", "dmmis", "dmmis", "stl", 1)]
    for k, (objective, enc_w, dec_w, mode, nz) in enumerate(cases):
        _mvbiwae(out, f"mvbiwae_{objective}_{enc_w}_{dec_w}_{mode}", objective=objective, enc_w=enc_w, dec_w=dec_w,
                 STL=mode == "stl", DREG=mode == "dreg", nz=nz, seed=11000 + 10 * k)


def cases_mog_oracle(out):
    """Oracle sampler of the toy control study (vgiwae/utils/mog_utils.py): interpolated conditional mixtures for every
    start / target the configs name, and one sampling call with injected normal noise and recorded component draws."""
    from vgiwae.utils import mog_utils as ref
    r = np.random.default_rng(77)
    C, D, B, K = 6, 5, 9, 4
    A = r.normal(size=(C, D, D))
    covs = torch.tensor(A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D))
    means = torch.tensor(r.normal(size=(C, D)) * 2)
    probs = r.random(C) + 0.1
    comp_log_probs = torch.tensor(probs / probs.sum()).log()
    X, M = make_batch(771, B, D)
    M[0] = torch.tensor([True, False, False, False, False])
    M[1] = True                                      # fully observed row
    # the reference scales views of `covs` in place for some targets (harmless there: it reloads the parameters at every
    # call), so every call below gets its own copy
    arrays = dict(X=X, M=M, comp_log_probs=comp_log_probs, means=means, covs=covs.clone(), K=K)
    lp, mu, cv = ref.compute_conditional_mog_parameters(X, M, comp_log_probs, means, covs.clone())
    arrays.update(cond_lp=lp, cond_means=mu, cond_covs=cv)
    targets = ["indep_marginal_gaussian", "indep_marginal_mog", "conditional_independent_mog",
               "conditional_independent_mog_created_from_true_indep_mog",
               "conditional_independent_mog_with_var_shrinkage_to_0.01var", "conditional_mog_with_equal_component_probabilities",
               "conditional_mog_with_mode_collapse_to_maxprob_mode", "conditional_mog_with_mode_collapse_to_secondlargestprob_mode",
               "conditional_mog_with_mode_collapse_to_thirdlargestprob_mode",
               "conditional_mog_with_mode_collapse_to_fourthlargestprob_mode",
               "conditional_mog_with_mode_collapse_to_fifthlargestprob_mode", "marginal_gaussian", "marginal_mog",
               "conditional_mog_with_mode_collapse_to_firstnonzero_mode"]
    arrays["targets"] = np.array(targets)
    for i, t in enumerate(targets):
        for start, alpha in ((None, 0.25), ("indep_marginal_mog", 0.5)):
            p = ref.construct_interpolated_conditional_mogs_from_true_mog(
                X, M, start_distribution=start, interpolate_method="linear", interpolate_target=t, alpha=alpha,
                comp_log_probs=comp_log_probs.clone(), means=means.clone(), covs=covs.clone())
            tag = f"t{i}_{'cond' if start is None else 'indep'}"
            arrays.update({f"{tag}_lp": p["comp_log_probs"], f"{tag}_means": p["means"], f"{tag}_covs": p["covs"]})
    # sampling from the last interpolation of the first target
    p = ref.construct_interpolated_conditional_mogs_from_true_mog(
        X, M, start_distribution=None, interpolate_method="linear", interpolate_target=targets[0], alpha=0.5,
        comp_log_probs=comp_log_probs.clone(), means=means.clone(), covs=covs.clone())
    noise = Noise()
    eps = torch.from_numpy(np.random.default_rng(78).normal(size=(B, K, D)))
    noise.normal = [eps.clone()]
    import torch.distributions.multivariate_normal as tdm
    stock = tdm._standard_normal          # MultivariateNormal holds its own reference to the primitive
    tdm._standard_normal = lambda shape, dtype, device: noise._pop_normal(shape).to(dtype)
    try:
        with injected(noise):
            Xs = ref.sample_sparse_mog(K, M, p["comp_log_probs"], p["means"], p["covs"])
    finally:
        tdm._standard_normal = stock
    assert not noise.normal
    arrays.update(sample_eps=eps, sample_comp=noise.cat_draws[0], sample_out=Xs)
    save(out, "mog_oracle", **arrays)


def cases_trajectory(out):
    """A short training run of the reference, step by step: IWAE bound -> backward -> clip_grad_norm_ over all
    parameters -> the optimiser of its own configure_optimizers (Adam, AMSGrad on the generator group, cosine
    schedule) -> scheduler step, with the latent noise of every step taken from the seed recipe."""
    from vgiwae.models.iwae import estimate_iwelbo
    D, Zd, B, I, steps, clip = 8, 16, 32, 5, 25, 0.5
    seed = 12000
    enc_spec, dec_spec = NetSpec(2 * D, 2 * Zd, 2, 32), NetSpec(Zd, 2 * D, 2, 32)
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    model = build_model("IWAE", enc_spec, dec_spec, enc_p, dec_p, "normal_with_eps", "normal_with_eps",
                        encoder_use_mis_mask=True, var_latent_STL=True, num_importance_samples=I,
                        amsgrad_generator=True, amsgrad_latent=False, use_lr_scheduler=True, max_scheduler_steps=steps)
    model.hparams.lr_generator, model.hparams.lr_latent = 3e-3, 1e-3
    conf = model.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    params = [p for g in opt.param_groups for p in g["params"]]
    bounds, norms, lrs = [], [], []
    for t in range(steps):
        X, M = make_batch(seed + 100 + t, B, D)
        noise = Noise()
        noise.normal = [make_noise(seed + 1000 + t, 1, I, B, Zd)]
        with injected(noise):
            bound = estimate_iwelbo(model, (X, M), var_latent_STL=True, num_latent_samples=1, num_importance_samples=I)
        opt.zero_grad()
        (-bound).backward()
        norms.append(torch.nn.utils.clip_grad_norm_(params, clip))
        opt.step()
        sched.step()
        bounds.append(bound.detach())
        lrs.append([g["lr"] for g in opt.param_groups])
    arrays = dict(seed=seed, D=D, Zd=Zd, B=B, I=I, steps=steps, clip=clip, lr_generator=3e-3, lr_latent=1e-3,
                  enc_spec=np.array([enc_spec.in_dim, enc_spec.out_dim, enc_spec.nb, enc_spec.hidden]),
                  dec_spec=np.array([dec_spec.in_dim, dec_spec.out_dim, dec_spec.nb, dec_spec.hidden]),
                  bounds=torch.stack(bounds), grad_norms=torch.stack(norms), lrs=np.array(lrs))
    arrays.update({f"enc_final.{k}": v.detach() for k, v in model.var_latent_network.state_dict().items()})
    arrays.update({f"dec_final.{k}": v.detach() for k, v in model.generator_network.state_dict().items()})
    save(out, "trajectory_iwae", **arrays)


def _reparam(out, name, *, cls, D, Zd, C, B, I, nz, seed, with_eps=True, stl=True):
    """VAE / IWAE whose encoder is a mixture sampled ancestrally with implicit reparameterisation gradients
    (the reparametrised_mixture* families); stl: "sticking the landing" as in every shipped config, else the score of
    q is taken on the live distribution as well."""
    from vgiwae.shared.vae_enums import DISTRIBUTION
    from vgiwae.models.iwae import estimate_iwelbo
    from vgiwae.models.vae import estimate_elbo
    enc_spec, dec_spec = NetSpec(2 * D, 2 * C * Zd + C, 2, 32), NetSpec(Zd, 2 * D, 2, 32)
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    import vgiwae.models as models
    qd = DISTRIBUTION[f"reparametrised_mixture{C}_normal" + ("_with_eps" if with_eps else "")]
    kw = dict(num_importance_samples=I) if cls == "IWAE" else dict(kl_analytic=False)
    model = getattr(models, cls)(generator_network=ref_net(dec_spec, dec_p), generator_distribution=dist_enum("normal_with_eps"),
                                 var_latent_network=ref_net(enc_spec, enc_p), var_latent_distribution=qd,
                                 num_latent_samples=nz, lr_generator=1e-3, lr_latent=1e-3, encoder_use_mis_mask=True,
                                 var_latent_STL=stl, **kw)
    model.train()
    X, M = make_batch(seed + 2, B, D)
    shape = (nz, I, B, C, Zd) if cls == "IWAE" else (nz, B, C, Zd)
    eps = make_noise(seed + 3, *shape)
    noise = Noise()
    noise.normal = [eps.clone()]
    with injected(noise):
        if cls == "IWAE":
            bound = estimate_iwelbo(model, (X, M), var_latent_STL=stl, num_latent_samples=nz, num_importance_samples=I)
        else:
            bound = estimate_elbo(model, (X, M), kl_analytic=False, var_latent_STL=stl, num_latent_samples=nz)
    (-bound).backward()
    arrays = dict(seed=seed, B=B, I=I, nz=nz, D=D, Zd=Zd, C=C, with_eps=with_eps, cls=cls, bound=bound, X=X, M=M, eps=eps,
                  comp_idx=noise.cat_draws[0], STL=stl,
                  enc_spec=np.array([enc_spec.in_dim, enc_spec.out_dim, enc_spec.nb, enc_spec.hidden]),
                  dec_spec=np.array([dec_spec.in_dim, dec_spec.out_dim, dec_spec.nb, dec_spec.hidden]))
    arrays.update(params_to_dict("enc", enc_p))
    arrays.update(params_to_dict("dec", dec_p))
    arrays.update(grads_to_dict("genc", model.var_latent_network))
    arrays.update(grads_to_dict("gdec", model.generator_network))
    save(out, name, **arrays)


def cases_reparam(out):
    _reparam(out, "reparam_iwae_C5", cls="IWAE", D=8, Zd=6, C=5, B=10, I=3, nz=1, seed=8000)
    _reparam(out, "reparam_vae_C5", cls="VAE", D=8, Zd=6, C=5, B=10, I=1, nz=2, seed=8100)
    _reparam(out, "reparam_iwae_C15_noeps", cls="IWAE", D=5, Zd=3, C=15, B=7, I=4, nz=2, seed=8200, with_eps=False)
    _reparam(out, "reparam_iwae_C5_live", cls="IWAE", D=8, Zd=6, C=5, B=10, I=3, nz=1, seed=8300, stl=False)
    _reparam(out, "reparam_vae_C5_live", cls="VAE", D=8, Zd=6, C=5, B=10, I=1, nz=2, seed=8400, stl=False)
    _reparam(out, "reparam_iwae_C15_noeps_live", cls="IWAE", D=5, Zd=3, C=15, B=7, I=4, nz=2, seed=8500, with_eps=False,
             stl=False)


def cases_marginal(out):
    """Grid marginal log-probability diagnostic of the toy configs (vae_marginal_logprob.get_marginal_logprob)."""
    from vgiwae.shared.vae_marginal_logprob import get_marginal_logprob
    for name, D, Zd, steps, lo, hi, gen in (("marginal_2d", 5, 2, 40, None, None, "normal"),
                                           ("marginal_2d_range", 5, 2, 30, -4.0, 4.0, "normal_with_eps"),
                                           ("marginal_1d", 4, 1, 300, None, None, "normal")):
        seed = 9000 + 10 * steps
        enc_spec, dec_spec = NetSpec(D, 2 * Zd, 2, 32), NetSpec(Zd, 2 * D, 3, 32)
        enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
        model = build_model("VAE", enc_spec, dec_spec, enc_p, dec_p, gen, "normal", kl_analytic=True)
        model.eval()
        X, M = make_batch(seed + 2, 12, D)
        with torch.inference_mode():
            obs, full = get_marginal_logprob(model, (X, M), compute_complete=True, grid_min=lo, grid_max=hi, grid_steps=steps)
        arrays = dict(seed=seed, D=D, Zd=Zd, B=12, gen_dist=gen, q_dist="normal", grid_steps=steps,
                      grid_min=np.nan if lo is None else lo, grid_max=np.nan if hi is None else hi, X=X, M=M,
                      marginal_obs=obs, marginal_full=full,
                      enc_spec=np.array([D, 2 * Zd, 2, 32]), dec_spec=np.array([Zd, 2 * D, 3, 32]))
        arrays.update(params_to_dict("enc", enc_p))
        arrays.update(params_to_dict("dec", dec_p))
        save(out, name, **arrays)


def cases_posterior_divs(out):
    """Posterior-divergence diagnostics of the toy study (vae_marginal_logprob.get_posterior_divs)."""
    from vgiwae.shared.vae_marginal_logprob import get_posterior_divs
    names = ("comp_fow", "comp_rev", "comp_jsd", "inc_fow", "inc_rev", "inc_jsd", "imp_fow", "imp_rev", "imp_jsd")
    for name, D, Zd, steps, lo, hi, mask_concat in (("posterior_divs_2d", 5, 2, 60, -6.0, 6.0, False),
                                                   ("posterior_divs_2d_encm", 5, 2, 50, -5.0, 5.0, True),
                                                   ("posterior_divs_1d", 4, 1, 400, -8.0, 8.0, False)):
        seed = 13000 + steps
        B, K = 7, 3
        enc_spec, dec_spec = NetSpec(2 * D if mask_concat else D, 2 * Zd, 2, 32), NetSpec(Zd, 2 * D, 2, 32)
        enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
        model = build_model("VAE", enc_spec, dec_spec, enc_p, dec_p, "normal", "normal", kl_analytic=True,
                            encoder_use_mis_mask=mask_concat)
        model.eval()
        X, M = make_batch(seed + 2, B, D)
        fill, _ = make_batch(seed + 6, B * K, D)
        Mk = M[:, None, :].expand(B, K, D).contiguous()
        Xt = torch.where(Mk, X[:, None, :].expand(B, K, D), fill.reshape(B, K, D)).contiguous()
        with torch.inference_mode():
            res = get_posterior_divs(model, (X, M), (Xt, Mk), grid_min=lo, grid_max=hi, grid_steps=steps)
        arrays = dict(seed=seed, D=D, Zd=Zd, B=B, K=K, gen_dist="normal", q_dist="normal", mask_concat=mask_concat,
                      grid_steps=steps, grid_min=lo, grid_max=hi, X=X, M=M, Xt=Xt, Mk=Mk,
                      enc_spec=np.array([enc_spec.in_dim, 2 * Zd, 2, 32]), dec_spec=np.array([Zd, 2 * D, 2, 32]))
        arrays.update(dict(zip(names, res)))
        arrays.update(params_to_dict("enc", enc_p))
        arrays.update(params_to_dict("dec", dec_p))
        save(out, name, **arrays)


def cases_resnet(out):
    """DeMissVAE (mvb2, STL, pseudo-Gibbs x1) with small convolutional trunks on 28x28 Bernoulli{-1,+1} images: the
    structure of configs/mnist_bin/compare/q50/mvb2vae_noencm_pg_k5_stl.yaml at reduced widths, dropout off."""
    import vgiwae.models as models
    from vgiwae.models.mvbvae import MVB_OBJECTIVE
    from vgiwae.shared.vae_resnet import ResNetEncoder, ResNetDecoder
    from vgiwae.shared.vae_imputation import VAESampler_PseudoGibbs
    seed, B, K, D, Zd = 7000, 6, 3, 784, 6
    for tag, use_mask in (("noencm", False), ("encm", True)):
        torch.manual_seed(seed)
        enc = ResNetEncoder(input_shape=[1, 28, 28], layers=[1, 1], layer_widths=[4, 8], latent_dim=2 * Zd, dropout_prob=0.0,
                            expect_miss_mask_as_extra_chennel=use_mask)
        dec = ResNetDecoder(output_shape=[1, 28, 28], layers=[1, 1], layer_upscale=[None, None], layer_widths=[8, 4],
                            downscaled_resolution=[4, 4], latent_dim=Zd, upscale_factor=2, dropout_prob=0.0)
        enc_sd = {k: v.clone() for k, v in enc.state_dict().items()}
        dec_sd = {k: v.clone() for k, v in dec.state_dict().items()}
        model = models.MVBVAE(generator_network=dec, generator_distribution=dist_enum("bern_m1_p1"), var_latent_network=enc,
                              var_latent_distribution=dist_enum("normal_with_eps"), num_latent_samples=1, lr_generator=1e-4,
                              lr_latent=1e-4, encoder_use_mis_mask=use_mask, var_latent_STL=True, kl_analytic=False,
                              mvb_objective=MVB_OBJECTIVE["mvb2"])
        model.train()
        X, M = make_batch(seed + 2, B, D, binary=True)
        fill, _ = make_batch(seed + 6, B * K, D, binary=True)
        Mk = M[:, None, :].expand(B, K, D).contiguous()
        Xt = torch.where(Mk, X[:, None, :].expand(B, K, D), fill.reshape(B, K, D)).contiguous()
        # one pseudo-Gibbs sweep (no grad), then the two objectives and the two backward calls
        eps_z, u_x = make_noise(seed + 10, B, K, Zd), make_uniform(seed + 20, B, K, D)
        noise = Noise()
        noise.normal, noise.uniform = [eps_z.clone()], [u_x.clone()]
        sampler = VAESampler_PseudoGibbs(num_iterations=1)
        with injected(noise), torch.no_grad():
            X_new = sampler(Xt.clone(), Mk, model=model)
        # the sweep ran the trunks in training mode: batch-norm running statistics moved; the objectives below
        # start from those (kept in the fixture so that the order of operations can be replayed)
        eps = make_noise(seed + 3, 1, B, K, Zd)
        noise = Noise()
        noise.normal = [eps.clone()]
        with injected(noise):
            gen_obj, enc_obj = model.estimate_mvb2_objectives(X_new, Mk)
        (-gen_obj.mean()).backward(retain_graph=True, inputs=list(dec.parameters()))
        (-enc_obj.mean()).backward(inputs=list(enc.parameters()))
        arrays = dict(seed=seed, B=B, K=K, D=D, Zd=Zd, use_mask=use_mask, Xt=Xt, M=Mk, eps_z=eps_z, u_x=u_x, X_new=X_new,
                      eps=eps, gen_obj=gen_obj, enc_obj=enc_obj)
        arrays.update({f"enc.{k}": v.numpy() for k, v in enc_sd.items()})
        arrays.update({f"dec.{k}": v.numpy() for k, v in dec_sd.items()})
        arrays.update(grads_to_dict("genc", enc))
        arrays.update(grads_to_dict("gdec", dec))
        save(out, f"resnet_mvb2_{tag}", **arrays)


class _Duck:
    """What the reference's samplers need from `model` is exactly these five names."""
    pass


def _sampler_model(enc_spec, dec_spec, enc_p, dec_p, gen_dist, q_dist, mask_concat):
    # a real reference VAE object: the samplers call its predict_var_latent_params /
    # create_distribution / generator_network / get_prior / hparams
    return build_model("VAE", enc_spec, dec_spec, enc_p, dec_p, gen_dist, q_dist,
                       encoder_use_mis_mask=mask_concat, kl_analytic=False)


def cases_samplers(out):
    from vgiwae.shared.vae_imputation import VAESampler_PseudoGibbs, VAESampler_LAIR, pseudo_gibbs_iteration
    # ---- pseudo-Gibbs, Gaussian decoder, with clamping and a forced non-finite draw
    D, Zd, B, K = 5, 2, 12, 5
    seed = 5000
    enc_spec, dec_spec = NetSpec(D, 2 * Zd, 2, 32), NetSpec(Zd, 2 * D, 2, 32)
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    model = _sampler_model(enc_spec, dec_spec, enc_p, dec_p, "normal", "normal", False)
    X, M = make_batch(seed + 2, B, D)
    fill, _ = make_batch(seed + 6, B * K, D)
    Mk = M[:, None, :].expand(B, K, D).contiguous()
    Xt = torch.where(Mk, X[:, None, :].expand(B, K, D), fill.reshape(B, K, D)).contiguous()
    for tag, clip in (("noclip", False), ("clip", True)):
        eps_z = make_noise(seed + 3, B, K, Zd)
        eps_x = make_noise(seed + 4, B, K, D)
        if clip:
            eps_x[0, 0, :] = float("inf")          # forces the non-finite fallback
            eps_x[1, 2, 3] = float("nan")
        cmin = torch.full((D,), -0.75)
        cmax = torch.full((D,), 0.5)
        noise = Noise()
        noise.normal = [eps_z.clone(), eps_x.clone()]
        with injected(noise), torch.inference_mode():
            X_new = pseudo_gibbs_iteration(model, Xt.clone(), Mk, clip_values_min=cmin if clip else None,
                                           clip_values_max=cmax if clip else None)
        arrays = dict(seed=seed, D=D, Zd=Zd, B=B, K=K, clip=clip, Xt=Xt, M=Mk, eps_z=eps_z, eps_x=eps_x,
                      clip_min=cmin, clip_max=cmax, X_new=X_new,
                      enc_spec=np.array([D, 2 * Zd, 2, 32]), dec_spec=np.array([Zd, 2 * D, 2, 32]))
        arrays.update(params_to_dict("enc", enc_p))
        arrays.update(params_to_dict("dec", dec_p))
        save(out, f"pgibbs_normal_{tag}", **arrays)

    # ---- pseudo-Gibbs, Bernoulli{-1,+1} decoder with mask-concat encoder, 2 iterations via the sampler class
    D2, Zd2, B2, K2 = 36, 8, 6, 5
    seed = 5100
    enc_spec2, dec_spec2 = NetSpec(2 * D2, 2 * Zd2, 1, 32), NetSpec(Zd2, D2, 1, 32)
    enc_p2, dec_p2 = make_params(enc_spec2, seed), make_params(dec_spec2, seed + 1)
    model2 = _sampler_model(enc_spec2, dec_spec2, enc_p2, dec_p2, "bern_m1_p1", "normal_with_eps", True)
    X2, M2 = make_batch(seed + 2, B2, D2, binary=True)
    fill2, _ = make_batch(seed + 6, B2 * K2, D2, binary=True)
    Mk2 = M2[:, None, :].expand(B2, K2, D2).contiguous()
    Xt2 = torch.where(Mk2, X2[:, None, :].expand(B2, K2, D2), fill2.reshape(B2, K2, D2)).contiguous()
    T = 2
    eps_z = [make_noise(seed + 10 + t, B2, K2, Zd2) for t in range(T)]
    u_x = [make_uniform(seed + 20 + t, B2, K2, D2) for t in range(T)]
    noise = Noise()
    noise.normal = [e.clone() for e in eps_z]
    noise.uniform = [u.clone() for u in u_x]
    sampler = VAESampler_PseudoGibbs(num_iterations=T)
    with injected(noise), torch.inference_mode():
        X_new = sampler(Xt2.clone(), Mk2, model=model2)
    arrays = dict(seed=seed, D=D2, Zd=Zd2, B=B2, K=K2, T=T, Xt=Xt2, M=Mk2, X_new=X_new,
                  eps_z=torch.stack(eps_z), u_x=torch.stack(u_x),
                  enc_spec=np.array([2 * D2, 2 * Zd2, 1, 32]), dec_spec=np.array([Zd2, D2, 1, 32]))
    arrays.update(params_to_dict("enc", enc_p2))
    arrays.update(params_to_dict("dec", dec_p2))
    save(out, "pgibbs_bern_m1_p1", **arrays)

    # ---- LAIR: dmis weights, global multinomial resampling
    for tag, I, r, T, final in (("r0_T1", 1, 0, 1, False), ("r1_T3_final", 2, 1, 3, True), ("r1_T10_final", 1, 1, 10, True)):
        seed = 5200 if r == 0 else 5300 if T == 3 else 5400
        q = "normal" if r == 0 else "normal_with_eps"
        enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
        model = _sampler_model(enc_spec, dec_spec, enc_p, dec_p, q, q, False)
        X, M = make_batch(seed + 2, B, D)
        fill, _ = make_batch(seed + 6, B * K, D)
        Mk = M[:, None, :].expand(B, K, D).contiguous()
        Xt = torch.where(Mk, X[:, None, :].expand(B, K, D), fill.reshape(B, K, D)).contiguous()
        eps_z = [make_noise(seed + 10 + t, I, B, K + r, Zd) for t in range(T)]
        eps_x = [make_noise(seed + 20 + t, B, K, D) for t in range(T + (1 if final else 0))]
        noise = Noise()
        for t in range(T):
            noise.normal += [eps_z[t].clone(), eps_x[t].clone()]
        if final:
            noise.normal.append(eps_x[T].clone())
        sampler = VAESampler_LAIR(num_iterations=T, num_imp_samples=I, num_prior_replenish_proposals=r,
                                  is_weights="dmis", resampling_scheme="global_resampling",
                                  resample_final_imps=final)
        with injected(noise), torch.inference_mode():
            X_new = sampler(Xt.clone(), Mk, model=model)
        arrays = dict(seed=seed, D=D, Zd=Zd, B=B, K=K, I=I, r=r, T=T, final=final, q_dist=q, Xt=Xt, M=Mk,
                      X_new=X_new, eps_z=torch.stack(eps_z), eps_x=torch.stack(eps_x),
                      enc_spec=np.array([D, 2 * Zd, 2, 32]), dec_spec=np.array([Zd, 2 * D, 2, 32]))
        for t, d in enumerate(noise.cat_draws):
            arrays[f"idx{t}"] = d
        arrays.update(params_to_dict("enc", enc_p))
        arrays.update(params_to_dict("dec", dec_p))
        save(out, f"lair_{tag}", **arrays)


def cases_buffer(out):
    """A14: gather / scatter / sub-sampled gather on the reference's ImputationDistribution."""
    from vgiwae.utils.imputation_distribution import ImputationDistribution, IMPUTATION_INIT
    N, D, K, B = 23, 4, 5, 7
    seed = 6000
    X, M = make_batch(seed, N, D)
    index = np.arange(N, dtype=np.int64)
    # the reference's datasets hand numpy arrays to ImputationDistribution
    dist = ImputationDistribution((X.numpy(), M.numpy(), index), target_idx=0, mis_idx=1, index_idx=2,
                                  num_imputations=K, imputation_init=IMPUTATION_INIT("empirical"))
    np.random.seed(seed)
    dist.init_imputations()
    after_init = dist.data_imp.clone()
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    bidx = torch.from_numpy(rng.permutation(N)[:B].astype(np.int64))
    batch = (X[bidx], M[bidx], bidx)
    g = dist.get_imputed_datapoints(batch)
    new_vals = make_noise(seed + 2, B, K, D)
    Mk = g[1]
    X_imp = torch.where(Mk, g[0], new_vals)
    dist.set_imputed_datapoints((X_imp,) + g[1:])
    after_set = dist.data_imp.clone()
    noise_seed = seed + 3
    torch.manual_seed(noise_seed)
    g2 = dist.get_imputed_datapoints(batch, num_imputations=3)
    save(out, "impbuffer", N=N, D=D, K=K, B=B, X=X, M=M, bidx=bidx, after_init=after_init,
         gather_X=g[0], gather_M=g[1], gather_index=g[2], gather_idx_imp=g[3], new_vals=new_vals,
         after_set=after_set, sub_X=g2[0], sub_idx_imp=g2[3], idx_imp_table=dist.idx_imp)


def cases_loglik(out):
    """A15: RefitEncoderUsingIWAE.test_step, standard and stratified (sampled without stratification)."""
    from vgiwae.test.refit_encoder_using_iwae_and_estimate_loglik import RefitEncoderUsingIWAE, RefitEncoderClass
    for tag, K in (("K64", 64), ("K4096", 4096)):
        for kind in ("standard", "stratified"):
            D, Zd, B, C = 8, 16, 10, 5
            seed = 7000 + (0 if kind == "standard" else 100) + K
            if kind == "standard":
                enc_spec, q, Cc = NetSpec(2 * D, 2 * Zd, 2, 32), "normal_with_eps", 1
                cls = "IWAE"
                kw = dict(num_importance_samples=5)
            else:
                enc_spec, q, Cc = NetSpec(2 * D, 2 * Zd * C + C, 2, 32), "stratified_with_eps", C
                cls = "MultipleIWAE"
                kw = dict(num_importance_samples_for_each_component=1)
            dec_spec = NetSpec(Zd, 2 * D, 2, 32)
            enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
            model = build_model(cls, enc_spec, dec_spec, enc_p, dec_p, "normal_with_eps", q, C=Cc,
                                encoder_use_mis_mask=True, **kw)
            model.eval()
            refit = RefitEncoderUsingIWAE(RefitEncoderClass[kind], num_test_latent_samples=1,
                                          num_test_importance_samples=K)
            refit.set_model(model)
            X, _ = make_batch(seed + 2, B, D)
            M = torch.ones(B, D, dtype=torch.bool)      # total_miss_test: 0.0
            logged = {}
            refit.log = lambda name, v, **k: logged.__setitem__(name, v)
            noise = Noise()
            if kind == "standard":
                eps = make_noise(seed + 3, 1, K, B, Zd)
                noise.normal = [eps.clone()]
            else:
                eps_all = make_noise(seed + 3, 1, K, B, C, Zd)
                noise.normal = [eps_all.clone()]
            with injected(noise):
                refit.test_step((X, M), 0)
            arrays = dict(seed=seed, D=D, Zd=Zd, B=B, K=K, C=Cc, kind=kind, X=X, M=M,
                          est_loglik=logged["est_loglik/test"],
                          enc_spec=np.array([enc_spec.in_dim, enc_spec.out_dim, 2, 32]),
                          dec_spec=np.array([Zd, 2 * D, 2, 32]))
            if kind == "standard":
                arrays["eps"] = eps if K <= 64 else np.zeros(0)
            else:
                comp = noise.cat_draws[0]                                   # [1,K,B]
                gathered = torch.gather(eps_all, 3, comp[..., None, None].expand(1, K, B, 1, Zd)).squeeze(3)
                arrays["comp_idx"] = comp
                arrays["eps"] = gathered if K <= 64 else np.zeros(0)
            arrays.update(params_to_dict("enc", enc_p))
            arrays.update(params_to_dict("dec", dec_p))
            save(out, f"loglik_{kind}_{tag}", **arrays)


def cases_loglik_k50000(out):
    """A15 at the configs' K: RefitEncoderUsingIWAE.test_step with 50 000 importance samples on 8 complete test points
    (configs/uci_*/compare/refit_encoder_testcomp: num_test_importance_samples 50000).  Noise is rebuilt from the
    seed; stored are the batch estimate and every point's own estimate (the test step run on that point alone with
    its slice of the same noise)."""
    from vgiwae.test.refit_encoder_using_iwae_and_estimate_loglik import RefitEncoderUsingIWAE, RefitEncoderClass
    D, Zd, B, K, seed = 8, 16, 8, 50000, 7900
    enc_spec, dec_spec = NetSpec(2 * D, 2 * Zd, 2, 32), NetSpec(Zd, 2 * D, 2, 32)
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    model = build_model("IWAE", enc_spec, dec_spec, enc_p, dec_p, "normal_with_eps", "normal_with_eps",
                        encoder_use_mis_mask=True, num_importance_samples=5)
    model.eval()
    refit = RefitEncoderUsingIWAE(RefitEncoderClass.standard, num_test_latent_samples=1, num_test_importance_samples=K)
    refit.set_model(model)
    X, _ = make_batch(seed + 2, B, D)
    M = torch.ones(B, D, dtype=torch.bool)
    eps = make_noise(seed + 3, 1, K, B, Zd)
    logged = {}
    refit.log = lambda name, v, **k: logged.__setitem__(name, v)

    def run(Xb, Mb, eb):
        noise = Noise()
        noise.normal = [eb.clone()]
        with injected(noise):
            refit.test_step((Xb, Mb), 0)
        return float(logged["est_loglik/test"])
    est = run(X, M, eps)
    per_point = np.array([run(X[b:b + 1], M[b:b + 1], eps[:, :, b:b + 1].contiguous()) for b in range(B)], dtype=np.float64)
    arrays = dict(seed=seed, D=D, Zd=Zd, B=B, K=K, C=1, kind="standard", X=X, M=M, est_loglik=est, per_point=per_point,
                  eps=np.zeros(0), enc_spec=np.array([2 * D, 2 * Zd, 2, 32]), dec_spec=np.array([Zd, 2 * D, 2, 32]))
    arrays.update(params_to_dict("enc", enc_p))
    arrays.update(params_to_dict("dec", dec_p))
    save(out, "loglik_standard_K50000", **arrays)


def cases_loglik_gas_full(out):
    """A15 at the shape the headline log-likelihood rate is quoted on: UCI-gas networks (hidden 256, two blocks), a test
    batch of 50 complete points, 50 000 importance samples (configs/uci_gas/compare/refit_encoder_testcomp).  Seed-only:
    weights, data and noise are rebuilt from the recipes; stored are the batch estimate and every point's own estimate
    (the test step on that point alone with its slice of the noise).  Run under no_grad: the stand-in base class has
    no trainer to switch gradients off, and the graph of 2.5 M rows x 256 would not fit."""
    from vgiwae.test.refit_encoder_using_iwae_and_estimate_loglik import RefitEncoderUsingIWAE, RefitEncoderClass
    D, Zd, B, K, seed = 8, 16, 50, 50000, 8100
    enc_spec, dec_spec = NetSpec(2 * D, 2 * Zd, 2, 256), NetSpec(Zd, 2 * D, 2, 256)
    enc_p, dec_p = make_params(enc_spec, seed), make_params(dec_spec, seed + 1)
    model = build_model("IWAE", enc_spec, dec_spec, enc_p, dec_p, "normal_with_eps", "normal_with_eps",
                        encoder_use_mis_mask=True, num_importance_samples=5)
    model.eval()
    refit = RefitEncoderUsingIWAE(RefitEncoderClass.standard, num_test_latent_samples=1, num_test_importance_samples=K)
    refit.set_model(model)
    X, _ = make_batch(seed + 2, B, D)
    M = torch.ones(B, D, dtype=torch.bool)
    eps = make_noise(seed + 3, 1, K, B, Zd)
    logged = {}
    refit.log = lambda name, v, **k: logged.__setitem__(name, v)

    def run(Xb, Mb, eb):
        noise = Noise()
        noise.normal = [eb.clone()]
        with injected(noise), torch.no_grad():
            refit.test_step((Xb, Mb), 0)
        return float(logged["est_loglik/test"])
    est = run(X, M, eps)
    per_point = np.array([run(X[b:b + 1], M[b:b + 1], eps[:, :, b:b + 1].contiguous()) for b in range(B)], dtype=np.float64)
    save(out, "loglik_gas_K50000_full", seed=seed, D=D, Zd=Zd, B=B, K=K, C=1, kind="standard", X=X, M=M, est_loglik=est,
         per_point=per_point, eps=np.zeros(0), enc_spec=np.array([2 * D, 2 * Zd, 2, 256]), dec_spec=np.array([Zd, 2 * D, 2, 256]))


def cases_datamodule(out, ref):
    """(f)2: the reference's MissingDataModule itself (vgiwae/data/missing_data_module.py:171-312) on the toy data it
    ships: split loading, missingness, fully-missing filter, index wrapper and pre-imputation in its own order of
    generator draws.  Stored per configuration: packed masks, kept indices, and of the (pre-imputed) data a row
    sub-sample plus float64 column sums."""
    # version skew: the reference imports corrcoef from numpy.lib.function_base, a module path numpy 2 dropped; the
    # function itself is numpy's own
    if "numpy.lib.function_base" not in sys.modules:
        import types
        shim = types.ModuleType("numpy.lib.function_base")
        shim.corrcoef = np.corrcoef
        sys.modules["numpy.lib.function_base"] = shim
    from vgiwae.data import MissingDataModule
    from vgiwae.data.missing_data_module import DATASET, MISSINGNESS, PRE_IMPUTATION
    cwd = os.getcwd()
    os.chdir(ref)                # the toy loader resolves <data_root>/toy/<file>.mat
    try:
        arrays = {}
        # ('empirical' pre-imputation cannot be pinned: the reference hands a numpy mask to torch.multinomial,
        #  missing_data_module.py:150, and raises TypeError on every dataset it ships)
        configs = (("uniform_oracle", "uniform", 0.5, 0.3, "oracle", False),
                   ("tophalf_mean", "top_half", 0.5, 0.5, "mean", True),
                   ("uniform_zero", "uniform", 0.8, 0.5, "zero", False))
        for tag, miss, mt, mv, pre, pre_val in configs:
            dm = MissingDataModule(dataset=DATASET.toy_mog2_large, batch_size=64, missingness=MISSINGNESS(miss),
                                   total_miss_train=mt, total_miss_val=mv, pre_imputation=PRE_IMPUTATION(pre),
                                   pre_impute_val=pre_val, data_root="./data", setup_seed=20220118, total_miss_test=0.0)
            dm.setup("fit")
            dm.setup("test")
            for split, ds in (("train", dm.train_data), ("val", dm.val_data), ("test", dm.test_data)):
                items = ds[:]
                Xs, Ms, idx = np.asarray(items[0]), np.asarray(items[1]), np.asarray(items[-1])
                arrays[f"{tag}.{split}.n"] = len(ds)
                arrays[f"{tag}.{split}.mask_bits"] = np.packbits(Ms.astype(np.uint8), axis=None)
                arrays[f"{tag}.{split}.index"] = idx.astype(np.int64)
                arrays[f"{tag}.{split}.x_sub"] = Xs[::37].astype(np.float32)
                arrays[f"{tag}.{split}.x_colsum"] = Xs.astype(np.float64).sum(0)
        arrays["configs"] = np.array([",".join(str(v) for v in c) for c in configs])
        save(out, "datamodule", setup_seed=20220118, **arrays)
    finally:
        os.chdir(cwd)


def cases_masks(out, ref):
    """A0: missingness masks, fully-missing filter and index wrapper (loaded by file path because the
    reference's vgiwae.data package __init__ needs torchvision)."""
    def load(fname, mod):
        spec = importlib.util.spec_from_file_location(mod, os.path.join(ref, "vgiwae", "data", fname))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    uni = load("uniform_missingness_provider.py", "ref_uniform")
    quad = load("quadrant_missingness_provider.py", "ref_quadrant")
    filt = load("fully_missing_filter_dataset.py", "ref_filter")
    widx = load("dataset_with_indices.py", "ref_with_idx")

    class Plain(torch.utils.data.Dataset):
        def __init__(self, data):
            self.data = data

        def __getitem__(self, i):
            return self.data[i]

        def __len__(self):
            return len(self.data)

    arrays = {}
    N, D = 37, 5
    data = np.arange(N * D, dtype=np.float32).reshape(N, D)
    for miss in (0.2, 0.5, 0.8):
        g = torch.Generator()
        g.manual_seed(20220118)
        ds = uni.UniformMissingnessDataset(Plain(data), total_miss=miss, rng=g)
        arrays[f"uniform_{int(miss * 100)}"] = ds.miss_mask
        f = filt.FullyMissingDataFilter(ds)
        arrays[f"kept_{int(miss * 100)}"] = np.asarray(f.not_fully_missing_idx)
        w = widx.DatasetWithIndices(f)
        arrays[f"index_{int(miss * 100)}"] = w[:][-1]
    img = np.zeros((19, 28 * 28), dtype=np.float32)
    for miss in (0.25, 0.5, 0.75):
        g = torch.Generator()
        g.manual_seed(20220118)
        ds = quad.QuadrantMissingnessDataset(Plain(img), total_miss=miss, img_dims=[28, 28], rng=g)
        arrays[f"quadrant_{int(miss * 100)}"] = ds.miss_mask
    top = load("tophalf_missingness_provider.py", "ref_tophalf")
    for miss in (0.25, 0.5):
        ds = top.TopHalfMissingnessDataset(Plain(img), total_miss=miss, rng=torch.Generator().manual_seed(20220118))
        arrays[f"tophalf_{int(miss * 100)}"] = ds.miss_mask
    save(out, "masks", N=N, D=D, seed=20220118, **arrays)


def cases_dmis_known_answer(out):
    """The reference's own self-check shapes (vgiwae/shared/iwae.py:135-181): DMIS weights for
    arbitrary Gaussian decoder / encoder parameters, stored so the HIP kernel can be checked on them."""
    from vgiwae.shared.iwae import compute_dmis_log_unnormalised_importance_weights
    B, I, K, D, NZ, DZ = 7, 6, 5, 4, 3, 2
    X = make_noise(8000, B, K, D)
    Z = make_noise(8001, NZ, I, B, K, DZ)
    gen_mean = make_noise(8002, NZ, I, B, K, D)
    gen_scale = (make_noise(8003, NZ, I, B, K, D) * 3) ** 2 + 1e-3
    var_mean = make_noise(8004, B, K, DZ)
    var_scale = (make_noise(8005, B, K, DZ) * 2) ** 2 + 1e-3
    lw = compute_dmis_log_unnormalised_importance_weights(
        X, torch.ones_like(X), Z, var_latent_distr=torch.distributions.Normal(var_mean, var_scale),
        var_comp_neg_idx=-2, prior_distr=torch.distributions.Normal(0., 1.),
        generator_distr=torch.distributions.Normal(gen_mean, gen_scale))
    save(out, "dmis_known_answer", X=X, Z=Z, gen_mean=gen_mean, gen_scale=gen_scale, var_mean=var_mean,
         var_scale=var_scale, log_w=lw)


def cases_resampling(out):
    """Reference outputs of the two resampling helpers its own self-checks exercise (vae_imputation.py:799-876):
    the independent-resampling gather for I == K and I == 2K, and systematic sampling with the uniform offsets
    injected (torch.rand replaced for the duration of the call)."""
    import vgiwae.shared.vae_imputation as vi
    B, P, K = 6, 4, 5
    arrays = {}
    for tag, I in (("ik", K), ("i2k", 2 * K)):
        G = make_noise(8100 + I, I, B, K, P)
        idx = torch.randint(0, I, (B, K), generator=torch.Generator().manual_seed(8200 + I))
        arrays[f"gather_{tag}_params"] = G
        arrays[f"gather_{tag}_idx"] = idx
        arrays[f"gather_{tag}_out"] = vi.select_gen_params_for_ir(G, idx, B, K)
    lw = make_noise(8300, 3, 7, 20) * 1.5
    orig = torch.rand
    for tag, nd, shape in (("all", None, ()), ("five", 5, (4,)), ("one", 1, (4,))):
        u = orig(tuple(shape) + (3, 7, 1), generator=torch.Generator().manual_seed(8400 + len(tag)))
        torch.rand = lambda *a, **k: u.clone()
        try:
            idx = vi._systematic_sample(lw, num_dependent_samples=nd, sample_shape=shape)
        finally:
            torch.rand = orig
        arrays[f"sys_{tag}_u"] = u
        arrays[f"sys_{tag}_idx"] = idx
    save(out, "resampling_helpers", sys_log_w=lw, **arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(HERE, "pl_stub"))
    sys.path.insert(0, args.ref)
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    groups = {
        "mlp": lambda: [case_mlp(args.out, "mlp_h32", NetSpec(16, 32, 2, 32), 40, 100),
                        case_mlp(args.out, "mlp_toy_h200", NetSpec(5, 4, 3, 200), 70, 110, store_params=False),
                        case_mlp(args.out, "mlp_gas_h256", NetSpec(16, 16, 2, 256), 100, 120, store_params=False),
                        case_mlp(args.out, "mlp_dropout", NetSpec(16, 32, 2, 32, dropout=0.5), 40, 130),
                        case_mlp(args.out, "mlp_odd_dims", NetSpec(43, 86, 1, 64), 33, 140, store_params=False),
                        # above every small-batch kernel family's row limit: the kernels the headline benchmark times
                        case_mlp(args.out, "mlp_gas_rows20k", NetSpec(16, 16, 2, 256), 21504, 150, store_params=False,
                                 digest_io=True, kink_margin=1e-5)],
        "iwae": lambda: cases_iwae(args.out),
        "vae": lambda: cases_vae(args.out),
        "multiple": lambda: cases_multiple(args.out),
        "mvb": lambda: cases_mvb(args.out),
        "samplers": lambda: cases_samplers(args.out),
        "buffer": lambda: cases_buffer(args.out),
        "loglik": lambda: cases_loglik(args.out),
        "loglik50k": lambda: cases_loglik_k50000(args.out),
        "loglik_gas_full": lambda: cases_loglik_gas_full(args.out),
        "datamodule": lambda: cases_datamodule(args.out, args.ref),
        "masks": lambda: cases_masks(args.out, args.ref),
        "dmis": lambda: cases_dmis_known_answer(args.out),
        "resnet": lambda: cases_resnet(args.out),
        "reparam": lambda: cases_reparam(args.out),
        "marginal": lambda: cases_marginal(args.out),
        "mvbiwae": lambda: cases_mvbiwae(args.out),
        "mog_oracle": lambda: cases_mog_oracle(args.out),
        "trajectory": lambda: cases_trajectory(args.out),
        "posterior_divs": lambda: cases_posterior_divs(args.out),
        "resampling": lambda: cases_resampling(args.out),
    }
    for name, fn in groups.items():
        if args.only and name not in args.only.split(","):
            continue
        print(f"[{name}]")
        fn()


if __name__ == "__main__":
    main()
