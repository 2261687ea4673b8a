"""CPU restatement of the vgiwae objectives (test infrastructure; see oracle/__init__.py).

Everything is written with plain tensor ops so it runs in fp32 or fp64 and under
autograd.  Noise is always passed in.  Shapes follow the reference:
  X [B, D] float, M [B, D] bool, eps [nz, I, B, Zd] (IWAE), [nz, I, B, C, Zd] (stratified),
  [nz, B, K, Zd] (DeMissVAE).
"""
import math
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F

from .recipes import NetSpec

EPSILON = 1e-5  # vgiwae/shared/vae_enums.py:3
LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))

_ACT = {
    "relu": F.relu,
    "leaky_relu": F.leaky_relu,
    "tanh": torch.tanh,
    "sigmoid": torch.sigmoid,
}


# --------------------------------------------------------------------------------------
# Residual MLP  (vgiwae/shared/neural_nets.py:67-72, 127-133)
# --------------------------------------------------------------------------------------
def mlp_forward(params, x, spec: NetSpec, dropout_masks=None):
    """out = final(act(block_n(... act(block_1(act(initial(x)))))))

    block(a) = a + L1(drop(act(L0(a)))).  `dropout_masks[b]` is the 0/1 keep mask applied
    (with 1/(1-p) scaling) inside block b; None means no dropout.
    """
    act = _ACT[spec.act]
    a = act(F.linear(x, params["initial_layer.weight"], params["initial_layer.bias"]))
    for b in range(spec.nb):
        t = act(F.linear(a, params[f"blocks.{b}.linear_layers.0.weight"],
                         params[f"blocks.{b}.linear_layers.0.bias"]))
        if dropout_masks is not None and spec.dropout > 0:
            t = t * dropout_masks[b].to(t.dtype) / (1.0 - spec.dropout)
        a = act(a + F.linear(t, params[f"blocks.{b}.linear_layers.1.weight"],
                             params[f"blocks.{b}.linear_layers.1.bias"]))
    return F.linear(a, params["final_layer.weight"], params["final_layer.bias"])


# --------------------------------------------------------------------------------------
# Model description
# --------------------------------------------------------------------------------------
@dataclass
class ModelSpec:
    D: int
    Zd: int
    enc: NetSpec
    dec: NetSpec
    gen_dist: str = "normal_with_eps"     # normal | normal_with_eps | bern | bern_m1_p1
    q_dist: str = "normal_with_eps"       # normal | normal_with_eps | stratified | stratified_with_eps
    C: int = 1                            # mixture components for stratified q
    encoder_use_mis_mask: bool = False

    @property
    def q_eps(self):
        return EPSILON if self.q_dist.endswith("with_eps") else 0.0

    @property
    def gen_eps(self):
        return EPSILON if self.gen_dist.endswith("with_eps") else 0.0

    @property
    def q_is_mixture(self):
        return self.q_dist.startswith("stratified")


def encoder_input(spec: ModelSpec, X, M):
    """vgiwae/models/vae.py:80-91: all of X, then all of M, on the last axis."""
    if spec.encoder_use_mis_mask:
        return torch.cat([X, M.to(X.dtype)], dim=-1)
    return X


def q_heads(spec: ModelSpec, p):
    """vgiwae/models/vae.py:113-125, 155-208.  Returns (mu, sigma, logits|None).

    Single Gaussian: mu, sigma [..., Zd].  Stratified: mu, sigma [..., C, Zd], logits [..., C]
    (unnormalised)."""
    if not spec.q_is_mixture:
        mu, raw = p[..., :spec.Zd], p[..., spec.Zd:2 * spec.Zd]
        return mu, F.softplus(raw) + spec.q_eps, None
    C, Zd = spec.C, spec.Zd
    comp, logits = p[..., :-C], p[..., -C:]
    mu = comp[..., :C * Zd].reshape(*comp.shape[:-1], C, Zd)
    raw = comp[..., C * Zd:].reshape(*comp.shape[:-1], C, Zd)
    return mu, F.softplus(raw) + spec.q_eps, logits


def log_normal(x, mu, sigma):
    """torch.distributions.Normal.log_prob restated."""
    return -((x - mu) ** 2) / (2 * sigma ** 2) - torch.log(sigma) - LOG_SQRT_2PI


def log_std_normal(z):
    return -(z ** 2) / 2 - LOG_SQRT_2PI


def log_bern_logits(v01, logits):
    """Bernoulli(logits).log_prob(v) = -BCEWithLogits(logits, v)."""
    return -((1 - v01) * logits - F.logsigmoid(logits))


def gen_logprob(spec: ModelSpec, g, X):
    """Elementwise log p(x_d | z) from decoder output g (vgiwae/models/vae.py:113-129)."""
    D = spec.D
    if spec.gen_dist in ("normal", "normal_with_eps"):
        m, s = g[..., :D], F.softplus(g[..., D:2 * D]) + spec.gen_eps
        return log_normal(X, m, s)
    if spec.gen_dist == "bern":
        return log_bern_logits(X, g[..., :D])
    if spec.gen_dist == "bern_m1_p1":
        # vgiwae/shared/bernoulli_between_m1_and_p1.py:24-26
        return log_bern_logits((X + 1) / 2, g[..., :D])
    raise NotImplementedError(spec.gen_dist)


def gen_sample(spec: ModelSpec, g, noise):
    """Draw x ~ p(.|z): Normal -> m + s*eps (noise = standard normal);
    Bernoulli -> 1[u < sigmoid(logit)] (noise = uniform), mapped to {-1,+1} for bern_m1_p1."""
    D = spec.D
    if spec.gen_dist in ("normal", "normal_with_eps"):
        m, s = g[..., :D], F.softplus(g[..., D:2 * D]) + spec.gen_eps
        return m + noise * s
    b = (noise < torch.sigmoid(g[..., :D])).to(g.dtype)
    if spec.gen_dist == "bern":
        return b
    return b * 2 - 1


# --------------------------------------------------------------------------------------
# Bounds
# --------------------------------------------------------------------------------------
def _encode(spec, enc_params, X_in, M, enc_masks=None):
    return mlp_forward(enc_params, encoder_input(spec, X_in, M), spec.enc, enc_masks)


def estimate_elbo(spec: ModelSpec, enc_params, dec_params, X, M, eps, *, kl_analytic, STL=False,
                  enc_masks=None, dec_masks=None):
    """vgiwae/models/vae.py:393-473.  eps [nz, B, Zd].  Returns (scalar, per-datapoint [B])."""
    Mf = M.to(X.dtype)
    p = _encode(spec, enc_params, X * Mf, M, enc_masks)
    mu, sigma, _ = q_heads(spec, p)
    Z = mu + sigma * eps
    g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
    gen = (gen_logprob(spec, g, X) * Mf).sum(-1)            # [nz, B]
    if kl_analytic:
        kl = 0.5 * (sigma ** 2 + mu ** 2 - 1 - torch.log(sigma ** 2))
        per_b = gen.mean(0) - kl.sum(-1)
    else:
        prior = log_std_normal(Z).sum(-1)
        if STL:
            lq = log_normal(Z, mu.detach(), sigma.detach()).sum(-1)
        else:
            lq = log_normal(Z, mu, sigma).sum(-1)
        per_b = (gen + prior - lq).mean(0)
    return per_b.mean(), per_b


def smis_log_weights(spec: ModelSpec, g, X, M, Z, mu, sigma, logits=None):
    """vgiwae/shared/iwae.py:9-55 with prior N(0,1); q single Gaussian or mixture over the
    components of the same datapoint (MixtureSameFamily.log_prob)."""
    gen = (gen_logprob(spec, g, X) * M.to(X.dtype)).sum(-1)
    prior = log_std_normal(Z).sum(-1)
    if logits is None:
        lq = log_normal(Z, mu, sigma).sum(-1)
    else:
        comp = log_normal(Z.unsqueeze(-2), mu, sigma).sum(-1)          # [..., B, C]
        lq = torch.logsumexp(comp + torch.log_softmax(logits, -1), dim=-1)
    return gen + prior - lq


def estimate_iwelbo(spec: ModelSpec, enc_params, dec_params, X, M, eps, *, STL=False, DREG=False,
                    enc_masks=None, dec_masks=None, return_aux=False):
    """vgiwae/models/iwae.py:70-154.  eps [nz, I, B, Zd].  DREG scales the gradient that
    reaches Z by softmax_i(log_w)."""
    assert not (STL and DREG)
    Mf = M.to(X.dtype)
    p = _encode(spec, enc_params, X * Mf, M, enc_masks)
    mu, sigma, _ = q_heads(spec, p)
    Z = mu + sigma * eps
    hook_w = []
    if DREG and Z.requires_grad:
        Z.register_hook(lambda grad: hook_w[0].unsqueeze(-1) * grad)
    g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
    if STL or DREG:
        log_w = smis_log_weights(spec, g, X, M, Z, mu.detach(), sigma.detach())
    else:
        log_w = smis_log_weights(spec, g, X, M, Z, mu, sigma)
    I = log_w.shape[1]
    if DREG:
        hook_w.append(torch.softmax(log_w.detach(), dim=1))
    per_zb = torch.logsumexp(log_w, dim=1) - math.log(I)
    per_b = per_zb.mean(0)
    if return_aux:
        return per_b.mean(), per_b, dict(log_w=log_w, Z=Z, enc_out=p, dec_out=g)
    return per_b.mean(), per_b


def stratified_mixture_logprob(Z, mu, sigma, logits):
    """vgiwae/shared/iwae.py:106-118: log sum_c' pi_{b,c'} N(z_{..., b, c}; mu_{b,c'}, sigma_{b,c'}).
    Z [..., B, C, Zd]; mu, sigma [B, C, Zd]; logits [B, C] -> [..., B, C]."""
    comp = log_normal(Z.unsqueeze(-2), mu.unsqueeze(-3), sigma.unsqueeze(-3)).sum(-1)  # [...,B,C,C']
    return torch.logsumexp(comp + torch.log_softmax(logits, -1).unsqueeze(-2), dim=-1)


def imputation_mixture_logprob(Z, mu, sigma):
    """vgiwae/shared/iwae.py:119-130: log (1/K) sum_k' prod_j N(z_{..., b, k, j}; mu_{b,k',j}, sigma_{b,k',j}).
    Z [..., B, K, Zd]; mu, sigma [B, K, Zd] -> [..., B, K]."""
    K = Z.shape[-2]
    comp = log_normal(Z.unsqueeze(-2), mu.unsqueeze(-3), sigma.unsqueeze(-3)).sum(-1)  # [...,B,K,K']
    return torch.logsumexp(comp, dim=-1) - math.log(K)


def estimate_multiple_iwelbo(spec: ModelSpec, enc_params, dec_params, X, M, eps, *, STL=False,
                             looser=False, enc_masks=None, dec_masks=None):
    """vgiwae/models/multiple_iwae.py:124-237.  eps [nz, I, B, C, Zd] (one draw per component)."""
    Mf = M.to(X.dtype)
    p = _encode(spec, enc_params, X * Mf, M, enc_masks)
    mu, sigma, logits = q_heads(spec, p)
    Z = mu + sigma * eps                                              # [nz, I, B, C, Zd]
    g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
    gen = (gen_logprob(spec, g, X.unsqueeze(1)) * Mf.unsqueeze(1)).sum(-1)
    prior = log_std_normal(Z).sum(-1)
    if STL:
        lq = stratified_mixture_logprob(Z, mu.detach(), sigma.detach(), logits.detach())
    else:
        lq = stratified_mixture_logprob(Z, mu, sigma, logits)
    log_w = gen + prior - lq                                          # [nz, I, B, C]
    I = log_w.shape[1]
    log_pi = torch.log_softmax(logits, -1)                            # live (not detached)
    if not looser:
        lw = log_w + log_pi - math.log(I)
        per_zb = torch.logsumexp(lw.permute(0, 1, 3, 2).reshape(lw.shape[0], -1, lw.shape[2]), dim=1)
    else:
        per_zb = (torch.logsumexp(log_w - math.log(I), dim=1) * torch.exp(log_pi)).sum(-1)
    per_b = per_zb.mean(0)
    return per_b.mean(), per_b


def estimate_multiple_elbo(spec: ModelSpec, enc_params, dec_params, X, M, eps, *, STL=False,
                           enc_masks=None, dec_masks=None):
    """vgiwae/models/multiple_vae.py:109-195.  eps [nz, B, C, Zd]."""
    Mf = M.to(X.dtype)
    p = _encode(spec, enc_params, X * Mf, M, enc_masks)
    mu, sigma, logits = q_heads(spec, p)
    Z = mu + sigma * eps                                              # [nz, B, C, Zd]
    g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
    gen = (gen_logprob(spec, g, X.unsqueeze(1)) * Mf.unsqueeze(1)).sum(-1)
    prior = log_std_normal(Z).sum(-1)
    if STL:
        lq = stratified_mixture_logprob(Z, mu.detach(), sigma.detach(), logits.detach())
    else:
        lq = stratified_mixture_logprob(Z, mu, sigma, logits)
    ratio = gen + prior - lq
    per_b = (ratio * torch.softmax(logits, -1)).sum(-1).mean(0)
    return per_b.mean(), per_b


def mvb2_objectives(spec: ModelSpec, enc_params, dec_params, Xt, M, eps, *, STL=False,
                    kl_analytic=False, enc_masks=None, dec_masks=None):
    """vgiwae/models/mvbvae.py:67-139.  Xt [B,K,D] imputed data, M [B,K,D], eps [nz,B,K,Zd].
    Returns (generator_obj [B,K], encoder_obj [B,K])."""
    Mf = M.to(Xt.dtype)
    p = _encode(spec, enc_params, Xt, M, enc_masks)                   # encoder sees imputed data
    mu, sigma, _ = q_heads(spec, p)
    Z = mu + sigma * eps
    g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
    lp = gen_logprob(spec, g, Xt)                                     # [nz,B,K,D]
    prior = log_std_normal(Z).sum(-1)
    with torch.no_grad():
        lmix = imputation_mixture_logprob(Z, mu, sigma)
    gen_obj = ((lp * Mf).sum(-1) + prior - lmix).mean(0)
    full = lp.sum(-1).mean(0)
    if kl_analytic:
        kl_neg = -(0.5 * (sigma ** 2 + mu ** 2 - 1 - torch.log(sigma ** 2))).sum(-1)
    else:
        if STL:
            lq = log_normal(Z, mu.detach(), sigma.detach()).sum(-1)
        else:
            lq = log_normal(Z, mu, sigma).sum(-1)
        kl_neg = (prior - lq).mean(0)
    return gen_obj, kl_neg + full


def cvi_objective(spec: ModelSpec, enc_params, dec_params, Xt, M, eps, *, STL=False,
                  enc_masks=None, dec_masks=None):
    """vgiwae/models/mvbvae.py:141-172."""
    Mf = M.to(Xt.dtype)
    p = _encode(spec, enc_params, Xt, M, enc_masks)
    mu, sigma, _ = q_heads(spec, p)
    Z = mu + sigma * eps
    g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
    lp = gen_logprob(spec, g, Xt)
    prior = log_std_normal(Z).sum(-1)
    if STL:
        lmix = imputation_mixture_logprob(Z, mu.detach(), sigma.detach())
    else:
        lmix = imputation_mixture_logprob(Z, mu, sigma)
    return ((lp * Mf).sum(-1) + prior - lmix).mean(0)


def mvb_objective(spec: ModelSpec, enc_params, dec_params, Xt, M, eps, *, STL=False,
                  kl_analytic=False, enc_masks=None, dec_masks=None):
    """Intended semantics of vgiwae/models/mvbvae.py:174-190 (complete-data ELBO per (b,k), M == 1).
    The reference's own function raises AttributeError, so this one has no golden vector:
    parity unpinned."""
    p = _encode(spec, enc_params, Xt, M, enc_masks)
    mu, sigma, _ = q_heads(spec, p)
    Z = mu + sigma * eps
    g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
    full = gen_logprob(spec, g, Xt).sum(-1)
    if kl_analytic:
        return full.mean(0) - (0.5 * (sigma ** 2 + mu ** 2 - 1 - torch.log(sigma ** 2))).sum(-1)
    prior = log_std_normal(Z).sum(-1)
    lq = log_normal(Z, mu.detach() if STL else mu, sigma.detach() if STL else sigma).sum(-1)
    return (full + prior - lq).mean(0)


def mvbiwae_objectives(spec: ModelSpec, enc_params, dec_params, Xt, M, eps, *, objective, enc_isweights="dmmis",
                       dec_isweights="dmmis", STL=False, DREG=False, resample_idx=None, enc_masks=None, dec_masks=None):
    """Importance-weighted DeMissVAE objectives, vgiwae/models/mvbiwae.py:97-650.  Xt, M [B,K,D], eps [nz,I,B,K,Zd].
    Returns per-datapoint (generator objective [B], encoder objective [B]); the single-objective variants
    (mvb_* / cvi_*) return the same tensor twice.

    Importance weights: "smis" uses log q_bk(z), "dmmis" the uniform mixture over the K imputations' posteriors.
    "all" pools the I*K samples of a datapoint in one logsumexp, "indep" takes one logsumexp over I per imputation
    and averages over K.  The weighted variants keep the value logsumexp(a) of the masked ("marginal") weights a but
    take the gradient of the complete-data weights under softmax(a); with DREG they add the score-function
    correction -(1 - w)w d a (mvbiwae.py:382-392).  The weight hook on Z the reference prepares is never
    installed (its log_weights_to_iwelbo is called without var_latent_DREG, models/iwae.py:70-81), so DREG only
    detaches q here."""
    Mf = M.to(Xt.dtype)
    p = _encode(spec, enc_params, Xt, M, enc_masks)
    mu, sigma, _ = q_heads(spec, p)
    Z = mu + sigma * eps                                              # [nz, I, B, K, Zd]
    g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
    lp = gen_logprob(spec, g, Xt)
    gen_marg, gen_full = (lp * Mf).sum(-1), lp.sum(-1)
    prior = log_std_normal(Z).sum(-1)
    mq, sq = (mu.detach(), sigma.detach()) if (STL or DREG) else (mu, sigma)
    lq = {"smis": log_normal(Z, mq, sq).sum(-1), "dmmis": imputation_mixture_logprob(Z, mq, sq)}
    nz, I, B, K = prior.shape

    def pooled(lw):                                                   # [nz, I, B, K] -> [nz, I*K, B]
        return lw.permute(0, 1, 3, 2).reshape(nz, I * K, B)

    def bound(lw, how):
        if how == "all":
            return (torch.logsumexp(pooled(lw), dim=1) - math.log(I * K)).mean(0)
        return (torch.logsumexp(lw, dim=1) - math.log(I)).mean((0, 2))

    if objective in ("mvb_indep", "mvb_all", "cvi_indep", "cvi_all"):
        assert enc_isweights == dec_isweights
        gen = gen_full if objective.startswith("mvb") else gen_marg
        out = bound(gen + prior - lq[enc_isweights], objective.split("_")[1])
        return out, out
    lw_marg = gen_marg + prior - lq[dec_isweights]
    lw_full = gen_full + prior - lq[enc_isweights]
    plain = {"mvb2_encindep_decall": ("all", "indep"), "mvb2_encindep_decindep": ("indep", "indep"),
             "mvb2_encall_decall": ("all", "all")}
    if objective in plain:
        dec_how, enc_how = plain[objective]
        return bound(lw_marg, dec_how), bound(lw_full, enc_how)
    weighted = {"mvb2_encallweightedsmis_decall": "smis", "mvb2_encallweighteddmmis_decall": "dmmis",
                "mvb2_encallweighteddmmis_decallresampl": "dmmis", "mvb2_all_encweighteddmmis_decweighteddmmis": "dmmis"}
    gw = gen_marg + prior - lq[weighted[objective]]                   # weights that re-weight the gradient
    enc_obj = bound(gw.detach() - lw_full.detach() + lw_full, "all")
    if DREG:
        a = pooled(gw)
        w = torch.softmax(a, dim=1)
        v = (((1 - w) * w).detach() * a).sum(1).mean(0)
        enc_obj = enc_obj - v + v.detach()
    if objective == "mvb2_all_encweighteddmmis_decweighteddmmis":
        gen_obj = bound(gw.detach() - lw_full.detach() + lw_full, "all")     # the un-corrected value, both networks
    elif objective.endswith("resampl"):
        flat = lw_marg.permute(0, 2, 1, 3).reshape(nz, B, I * K)             # [nz, B, (i k)]
        assert nz == 1, "the reference's gather only broadcasts for one outer sample"
        picked = torch.gather(flat.expand(resample_idx.shape[0], nz, B, I * K), -1, resample_idx.unsqueeze(-1)).squeeze(-1)
        gen_obj = picked.mean((0, 1))
    else:
        gen_obj = bound(lw_marg, "all")
    return gen_obj, enc_obj


# --------------------------------------------------------------------------------------
# Imputation samplers
# --------------------------------------------------------------------------------------
def pseudo_gibbs_iteration(spec: ModelSpec, enc_params, dec_params, X, M, eps_z, noise_x,
                           clip_min=None, clip_max=None, enc_masks=None, dec_masks=None):
    """vgiwae/shared/vae_imputation.py:26-52.  X [B,K,D] current imputations, M [B,K,D]."""
    with torch.no_grad():
        p = _encode(spec, enc_params, X, M, enc_masks)
        mu, sigma, _ = q_heads(spec, p)
        Z = mu + eps_z * sigma
        g = mlp_forward(dec_params, Z, spec.dec, dec_masks)
        X_m = gen_sample(spec, g, noise_x)
        if clip_min is not None:
            X_m = torch.max(X_m, clip_min)
        if clip_max is not None:
            X_m = torch.min(X_m, clip_max)
        bad = ~torch.isfinite(X_m)
        X_m = torch.where(bad, X, X_m)
        Mf = M.to(X.dtype)
        return X * Mf + X_m * (1 - Mf)


def softplus_inverse(x):
    """vgiwae/utils/softplus_inverse.py:3-7."""
    return x + torch.log(-torch.expm1(-x))


def lair_iteration(spec: ModelSpec, enc_params, dec_params, X, M, eps_z, idx, noise_x, *,
                   num_prior_replenish=0, enc_masks=None, dec_masks=None):
    """One LAIR step with DMIS weights and global multinomial resampling
    (vgiwae/shared/vae_imputation.py:177-201, 401-563).

    X [B,K,D], M [B,K,D]; eps_z [I,B,K+r,Zd]; idx [K+r,B] int64 = the categorical draws over
    the I*(K+r) proposals of each datapoint (proposal index i*(K+r)+k); noise_x [B,K,D].
    Returns dict(X_out, log_w [I,B,K+r], Z [I,B,K+r,Zd], mis_params [B,K,P])."""
    with torch.no_grad():
        B, K, D = X.shape
        r = num_prior_replenish
        p = _encode(spec, enc_params, X, M, enc_masks)                # [B,K,P]
        M1 = M[:, :1]
        Xe = X
        if r > 0:
            Zd = spec.Zd
            rep_mu = torch.zeros(B, r, Zd, dtype=X.dtype)
            rep_raw = softplus_inverse(torch.ones(B, r, Zd, dtype=X.dtype) - spec.q_eps)
            p = torch.cat([p[..., :2 * Zd], torch.cat([rep_mu, rep_raw], -1)], dim=1)
            Xe = torch.cat([X, X[:, :1].expand(B, r, D)], dim=1)
        mu, sigma, _ = q_heads(spec, p)
        Z = mu + eps_z * sigma                                        # [I,B,K+r,Zd]
        g = mlp_forward(dec_params, Z, spec.dec, dec_masks)           # [I,B,K+r,P]
        gen = (gen_logprob(spec, g, Xe) * M1.to(X.dtype)).sum(-1)
        prior = log_std_normal(Z).sum(-1)
        lmix = imputation_mixture_logprob(Z, mu, sigma)
        log_w = gen + prior - lmix                                    # [I,B,K+r]
        I, Kp = log_w.shape[0], log_w.shape[2]
        flat = g.permute(1, 0, 2, 3).reshape(B, I * Kp, -1)           # 'i b k p -> b (i k) p'
        mis = flat[torch.arange(B), idx]                              # [K+r, B, P]
        mis = mis.permute(1, 0, 2)[:, :K]
        X_m = gen_sample(spec, mis, noise_x)
        Mf = M.to(X.dtype)
        X_out = X * Mf + X_m * (1 - Mf)
        return dict(X_out=X_out, log_w=log_w, Z=Z, mis_params=mis)


def inverse_cdf_categorical(log_w, u):
    """Index draw used by the device sampler: smallest j with cumsum(softmax(log_w))[j] > u.
    log_w [B, N], u [B, S] in [0,1) -> idx [B, S] int64.  (The reference draws with
    torch.multinomial, whose random stream cannot be reproduced; the distribution is the same.)"""
    p = torch.softmax(log_w.double(), dim=-1)
    c = torch.cumsum(p, dim=-1)
    idx = torch.searchsorted(c, u.double().clamp(max=1 - 1e-12), right=True)
    return idx.clamp(max=log_w.shape[-1] - 1)


def independent_resampling_gather(gen_params, idx):
    """Proposal selection of the independent-resampling scheme (vgiwae/shared/vae_imputation.py:565-573, used by
    ir_resampling :273-295; no shipped config selects it, restated for the reference's known-answer tests :801-854).
    gen_params [I, B, K, P]; the I K proposals of a datapoint, flattened i-major, are cut into K consecutive groups
    of I and idx [B, K] picks one member of each group -> [B, K, P]."""
    I, B, K, P = gen_params.shape
    flat = gen_params.permute(1, 0, 2, 3).reshape(B, I * K, P)          # j = i K + k
    j = torch.arange(K)[None, :] * I + idx                              # group k2 holds j = k2 I .. k2 I + I - 1
    return flat[torch.arange(B)[:, None], j]


def systematic_sample(log_w, u, num_dependent=None):
    """Systematic resampling (vgiwae/shared/vae_imputation.py:698-728; selectable as resampling_method, unused by the
    shipped configs; restated for the known-answer test :856-876).  log_w [..., n]; u [..., 1] uniform in [0, 1);
    returns [..., num_dependent] indices: sample m is the number of categories whose scaled cumulative
    probability num_dependent * cdf_j + u lies below m + 1."""
    n = log_w.shape[-1]
    nd = n if num_dependent is None else num_dependent
    p = torch.exp(log_w - torch.logsumexp(log_w, dim=-1, keepdim=True))
    p = p / p.sum(dim=-1, keepdim=True)
    level = torch.floor(p.cumsum(-1) * nd + u).clamp(min=0, max=nd)     # [..., n]
    m = torch.arange(nd, dtype=level.dtype)
    return (level.unsqueeze(-1) <= m).sum(dim=-2)


# --------------------------------------------------------------------------------------
# Imputation buffer (vgiwae/utils/imputation_distribution.py:77-79, 139-187)
# --------------------------------------------------------------------------------------
def buffer_init(data, K):
    """data [N,D] -> data_imp [(N K), D], idx_imp [N,K] = n*K+k."""
    N = data.shape[0]
    return data.repeat_interleave(K, dim=0).clone(), torch.arange(N * K).reshape(N, K)


def buffer_gather(data_imp, idx_imp, index, sampled_i=None):
    """Rows of the buffer for dataset indices `index` [B] -> ([B,K',D], idx [B,K'])."""
    idx = idx_imp[index]
    if sampled_i is not None:
        idx = idx[torch.arange(index.shape[0])[:, None], sampled_i]
    return data_imp[idx.reshape(-1)].reshape(idx.shape[0], idx.shape[1], -1), idx


def buffer_scatter(data_imp, idx, X_imp):
    data_imp[idx.reshape(-1)] = X_imp.reshape(-1, X_imp.shape[-1])
    return data_imp


# --------------------------------------------------------------------------------------
# K-sample log-likelihood estimate (vgiwae/test/refit_...py:189-222)
# --------------------------------------------------------------------------------------
def loglik_eval(spec: ModelSpec, enc_params, dec_params, X, M, eps, comp_idx=None):
    """Per-datapoint log (1/K) sum_k w_k.  Standard: eps [1,K,B,Zd].  Stratified q sampled
    without stratification (vgiwae/models/vae.py:298-301): comp_idx [1,K,B] int64 chooses
    the component each z is drawn from, eps [1,K,B,Zd] is that component's noise."""
    with torch.no_grad():
        Mf = M.to(X.dtype)
        p = _encode(spec, enc_params, X * Mf, M)
        mu, sigma, logits = q_heads(spec, p)
        if logits is None:
            Z = mu + sigma * eps
        else:
            B = X.shape[0]
            ar = torch.arange(B)
            Z = mu[ar, comp_idx] + sigma[ar, comp_idx] * eps
        g = mlp_forward(dec_params, Z, spec.dec)
        log_w = smis_log_weights(spec, g, X, M, Z, mu, sigma, logits)
        K = log_w.shape[1]
        per_b = (torch.logsumexp(log_w, dim=1) - math.log(K)).mean(0)
        return per_b.mean(), per_b


# ------------------------------------------------------------------------------------------
# mixture encoder sampled ancestrally, implicit reparameterisation gradients
# (vgiwae/shared/reparametrized_mixture_same_family.py:27-191; vgiwae/models/vae.py:209-256)
# ------------------------------------------------------------------------------------------
def distributional_transform(z, mu, sigma, logits):
    """T_i(z) = sum_k w_ik Phi((z_i - mu_ki)/sigma_ki), w_ik = softmax_k(logit_k + sum_{j<i} log N(z_j; mu_kj, sigma_kj))
    (reparametrized_mixture_same_family.py:119-172).  z [..., Zd]; mu, sigma [..., C, Zd]; logits [..., C]."""
    zc = z.unsqueeze(-2)
    lp = log_normal(zc, mu, sigma)                                  # [..., C, Zd]
    excl = torch.cumsum(lp, dim=-1) - lp                            # exclusive prefix sums
    w = torch.softmax(logits.unsqueeze(-1) + excl, dim=-2)
    cdf = 0.5 * (1.0 + torch.erf((zc - mu) / (sigma * math.sqrt(2.0))))
    return (w * cdf).sum(-2)


def reparam_mixture_rsample(mu, sigma, logits, comp_idx, eps):
    """Value: z = mu_c + sigma_c eps_c for the recorded component draws c (MixtureSameFamily.sample draws every
    component and gathers).  Gradient: z + (s - s.detach()) with s = -J^-1 T(z), J = dT/dz evaluated at z
    (reparametrized_mixture_same_family.py:61-113).  mu, sigma [B, C, Zd]; logits [B, C]; comp_idx [S.., B];
    eps [S.., B, C, Zd]."""
    C, Zd = mu.shape[-2], mu.shape[-1]
    idx = comp_idx[..., None, None].expand(*comp_idx.shape, 1, Zd)
    z_all = mu + sigma * eps
    z = torch.gather(z_all, -2, idx).squeeze(-2).detach()            # [S.., B, Zd]
    if not (mu.requires_grad or logits.requires_grad):
        return z
    lead = z.shape[:-1]
    zf = z.reshape(-1, Zd)
    muf = mu.expand(*lead, C, Zd).reshape(-1, C, Zd)
    sgf = sigma.expand(*lead, C, Zd).reshape(-1, C, Zd)
    lgf = logits.expand(*lead, C).reshape(-1, C)
    T = distributional_transform(zf, muf, sgf, lgf)                 # differentiable w.r.t. the parameters
    with torch.enable_grad():
        zj = zf.clone().requires_grad_(True)
        Tj = distributional_transform(zj, muf.detach(), sgf.detach(), lgf.detach())
        rows = [torch.autograd.grad(Tj[:, i].sum(), zj, retain_graph=True)[0] for i in range(Zd)]
    J = torch.stack(rows, dim=1).detach()                           # [N, Zd(i), Zd(j)], lower triangular
    s_ = -torch.linalg.solve_triangular(J, T.unsqueeze(-1), upper=False).squeeze(-1)
    return (zf + (s_ - s_.detach())).reshape(*lead, Zd)


def mixture_logprob(Z, mu, sigma, logits):
    """MixtureSameFamily.log_prob: logsumexp_c(log pi_c + sum_j log N(z_j; mu_cj, sigma_cj))."""
    lp = log_normal(Z.unsqueeze(-2), mu, sigma).sum(-1)
    return torch.logsumexp(lp + torch.log_softmax(logits, dim=-1), dim=-1)


def estimate_iwelbo_reparam_mixture(spec: ModelSpec, enc_params, dec_params, X, M, eps, comp_idx, *, STL=True):
    """IWAE bound (vgiwae/models/iwae.py:70-154) for a reparametrised-mixture encoder.  eps [nz, I, B, C, Zd],
    comp_idx [nz, I, B].  With a single importance sample this is the Monte-Carlo-KL ELBO of vgiwae/models/vae.py:
    412-445 (pass eps [nz, 1, B, C, Zd])."""
    Mf = M.to(X.dtype)
    p = _encode(spec, enc_params, X * Mf, M, None)
    mu, sigma, logits = q_heads(spec, p)
    Z = reparam_mixture_rsample(mu, sigma, logits, comp_idx, eps)
    g = mlp_forward(dec_params, Z, spec.dec, None)
    gen = (gen_logprob(spec, g, X) * Mf).sum(-1)
    prior = log_std_normal(Z).sum(-1)
    if STL:
        q = mixture_logprob(Z, mu.detach(), sigma.detach(), logits.detach())
    else:
        q = mixture_logprob(Z, mu, sigma, logits)
    log_w = gen + prior - q
    per_zb = torch.logsumexp(log_w, dim=1) - math.log(log_w.shape[1])
    per_b = per_zb.mean(0)
    return per_b.mean(), per_b


# ------------------------------------------------------------------------------------------
# marginal log-probability on a latent grid (vgiwae/shared/vae_marginal_logprob.py:8-51, 66-90, 194-307)
# ------------------------------------------------------------------------------------------
def latent_grid(latent_dim, grid_min=None, grid_max=None, grid_steps=None):
    """(Z [Nz, latent_dim], cell volume).  1-d: linspace(min, max, steps) (end points included) with cell
    (max - min)/steps; 2-d: the lower-left corners min + i (max - min)/steps, i < steps, on both axes."""
    import numpy as np
    if latent_dim == 1:
        lo, hi, steps = -15, 15, 4000
    elif latent_dim == 2:
        lo, hi, steps = -10, 10, 500
    else:
        raise NotImplementedError("numerical integration over more than 2 latent dimensions")
    lo = grid_min if grid_min is not None else lo
    hi = grid_max if grid_max is not None else hi
    steps = grid_steps if grid_steps is not None else steps
    if latent_dim == 1:
        return torch.linspace(lo, hi, steps=steps).reshape(-1, 1), (hi - lo) / steps
    z0, z1 = np.mgrid[lo:hi:(hi - lo) / steps, lo:hi:(hi - lo) / steps]
    Z = torch.tensor(np.stack([z0, z1], -1)).float().reshape(-1, 2)
    return Z, ((hi - lo) / steps) ** 2


def grid_marginal_logprob(spec: ModelSpec, dec_params, X, M, grid_min=None, grid_max=None, grid_steps=None):
    """(log p(x_o), log p(x)) per datapoint by summing over the grid."""
    Z, dz = latent_grid(spec.Zd, grid_min, grid_max, grid_steps)
    prior = log_std_normal(Z).sum(-1, keepdim=True)                       # [Nz, 1]
    g = mlp_forward(dec_params, Z, spec.dec, None).unsqueeze(1)           # [Nz, 1, P]
    comp = gen_logprob(spec, g, X)                                        # [Nz, B, D]
    joint_obs = prior + (comp * M.to(comp.dtype)).sum(-1)
    joint_full = prior + comp.sum(-1)
    ldz = math.log(dz)
    return torch.logsumexp(joint_obs + ldz, dim=0), torch.logsumexp(joint_full + ldz, dim=0)


def posterior_divs(spec: ModelSpec, enc_params, dec_params, X, M, X_imp, M_imp, grid_min=None, grid_max=None, grid_steps=None):
    """vgiwae/shared/vae_marginal_logprob.py:91-192: forward / reverse KL and JSD on the latent grid between the model's
    posterior and the variational distribution, for complete data [B], incomplete data with the mixture of the K
    imputations' variational distributions [B], and every imputed copy [B, K].  X is the complete datapoint."""
    Z, dz = latent_grid(spec.Zd, grid_min, grid_max, grid_steps)
    g = mlp_forward(dec_params, Z, spec.dec)                                     # [Nz, P]
    prior = log_std_normal(Z).sum(-1)                                            # [Nz]

    def posterior(Xr, Mr):
        lp = gen_logprob(spec, g.unsqueeze(1), Xr.unsqueeze(0))                  # [Nz, R, D]
        if Mr is not None:
            lp = lp * Mr.to(lp.dtype).unsqueeze(0)
        joint = prior.unsqueeze(1) + lp.sum(-1)
        return joint - torch.logsumexp(joint + math.log(dz), dim=0)

    def q_logprob(Xr, Mr):
        mu, sigma, _ = q_heads(spec, _encode(spec, enc_params, Xr, Mr))
        return log_normal(Z.unsqueeze(1), mu.unsqueeze(0), sigma.unsqueeze(0)).sum(-1)   # [Nz, R]

    def divs(lp, lq):
        kl_fow = (torch.exp(lp) * (lp - lq) * dz).sum(0)
        kl_rev = (torch.exp(lq) * (lq - lp) * dz).sum(0)
        lm = torch.logsumexp(torch.stack([lq, lp]), dim=0) - math.log(2)
        jsd = 0.5 * ((torch.exp(lq) * (lq - lm) * dz).sum(0) + (torch.exp(lp) * (lp - lm) * dz).sum(0))
        return kl_fow, kl_rev, jsd
    B, K, D = X_imp.shape
    complete = divs(posterior(X, None), q_logprob(X, M))
    lq_imps = q_logprob(X_imp.reshape(B * K, D), M_imp.reshape(B * K, D))        # [Nz, B*K]
    lq_mix = torch.logsumexp(lq_imps.reshape(-1, B, K), dim=2) - math.log(K)
    incomplete = divs(posterior(X, M), lq_mix)
    imputed = tuple(t.reshape(B, K) for t in divs(posterior(X_imp.reshape(B * K, D), None), lq_imps))
    return (*complete, *incomplete, *imputed)
