"""The objectives with caller-supplied networks.

The fused pipelines (objectives.py) run both residual MLPs inside the library.  Models whose encoder / decoder
are other torch modules (the convolutional trunks of the image configs: vgiwae/shared/vae_resnet.py) still get
every other piece from the kernels: latent sampling, the masked log-likelihood / prior / mixture terms, the
bounds and their hand-derived gradients.  Two autograd functions bracket the caller's decoder:

    enc_out = encoder(...)                      # torch module, [rows, P]
    Z       = _Sample(enc_out)                  # z = mu + sigma * eps            (dmv_latent_sample)
    dec_out = decoder(Z)                        # torch module, [N, P_dec]
    out     = _Objective(dec_out, enc_out)      # terms -> bound / objectives     (dmv_obj_terms, dmv_obj_bound, ...)

`_Objective.backward` returns the gradient w.r.t. the decoder output (dmv_obj_backward_dec) and the part of the
encoder-output gradient that does not pass through the decoder (dmv_obj_backward_latent without dZ);
`_Sample.backward` maps the decoder's input gradient through the reparameterisation (the same kernel with only
dZ).  The latent kernel is linear in (dZ, coefficients), so the two parts add up to the fused result."""
import ctypes as C

import torch

from . import _lib as L
from . import rng
from .objectives import ModelDims, _Plan, _bound, _combine, _kl, _latent_params, _u8, _implicit_d_enc, reparam_components


class _State:
    """Buffers shared by the two autograd functions of one evaluation."""


def _setup(dims: ModelDims, plan: _Plan, X, M, enc_out):
    demiss = plan.kind in ("mvb2", "cvi", "mvb")
    c = _State()
    if demiss:
        B, K, D = X.shape
        G, enc_rows = K, B * K
    else:
        B, D = X.shape
        G = dims.C if dims.q_is_mixture else 1
        enc_rows = B
    assert D == dims.D
    assert enc_out.shape[0] == enc_rows and enc_out.dim() == 2
    ldE = enc_out.shape[1]
    nz, I, Zd = plan.nz, plan.I, dims.Zd
    cfg = L.DmvObjCfg()
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, G, nz, I, D, Zd
    cfg.gen_dist, cfg.q_eps = L.GEN[dims.gen_dist], dims.q_eps
    cfg.x_per_component = 1 if demiss else 0
    if demiss:
        cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = K * ldE, ldE, Zd, 0
        q_kind, mix_C = L.Q_IMPUTATION, K
    elif dims.q_is_mixture or dims.q_is_reparam:
        cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, Zd, dims.C * Zd, 2 * dims.C * Zd
        q_kind, mix_C = L.Q_STRATIFIED, dims.C
    else:
        cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, 0, Zd, 0
        q_kind, mix_C = L.Q_OWN, 1
    c.cfg, c.B, c.G, c.N, c.demiss, c.q_kind, c.mix_C = cfg, B, G, nz * I * B * G, demiss, q_kind, mix_C
    c.Xc, c.Mu = X.contiguous().float(), _u8(M)
    c.plan, c.dims = plan, dims
    c.zscale = None
    c.reparam = dims.q_is_reparam
    if c.reparam:
        assert plan.kind == "iwae" and not plan.DREG, "reparametrised mixtures: ELBO (MC KL) and IWAE bounds only"
    return c


class _Sample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, enc_out, eps, comp_idx):
        lib, st = L.lib(), L.stream_ptr()
        Zd, N = c.dims.Zd, c.N
        enc = enc_out.detach().contiguous().float()
        Z = torch.empty((N, Zd), dtype=torch.float32, device=enc.device)
        if c.reparam:
            comp_idx, eps = reparam_components(c.dims, c.cfg, enc, comp_idx, eps)
        if eps is not None:
            used = eps.reshape(N, Zd).contiguous().float()
            L.check(lib.dmv_latent_sample(C.byref(c.cfg), L.ptr(enc), L.ptr(used), 0, 0, None, Zd, L.ptr(comp_idx), L.ptr(Z), None, st))
        else:
            used = torch.empty((N, Zd), dtype=torch.float32, device=enc.device)
            seed, off = rng.next_offset(N * Zd)
            L.check(lib.dmv_latent_sample(C.byref(c.cfg), L.ptr(enc), None, seed, off, rng.base_ptr(), Zd, L.ptr(comp_idx), L.ptr(Z), L.ptr(used), st))
        c.enc_out, c.Z, c.eps = enc, Z, used
        ctx.c = c
        return Z

    @staticmethod
    def backward(ctx, dZ):
        c = ctx.c
        lib, st = L.lib(), L.stream_ptr()
        dZ = dZ.contiguous().float()
        if c.reparam:
            return None, _implicit_d_enc(c, dZ, None, 0.0, 0.0), None, None
        d_enc = torch.zeros_like(c.enc_out)
        p = _latent_params(w=0.0, use_zscale=c.plan.DREG)
        L.check(lib.dmv_obj_backward_latent(C.byref(c.cfg), C.byref(p), L.ptr(c.enc_out), L.ptr(c.Z), L.ptr(c.eps), c.Z.shape[1],
                                            L.ptr(dZ), dZ.shape[1], None, None, L.ptr(c.zscale), None, None, None, None, None,
                                            L.ptr(d_enc), st))
        return None, d_enc, None, None


class _Objective(torch.autograd.Function):
    """outputs: (per-datapoint bound [B],) or (generator objective [B,K], encoder objective [B,K])."""

    @staticmethod
    def forward(ctx, c, dec_out, enc_out):
        lib, st = L.lib(), L.stream_ptr()
        plan = c.plan
        # the inputs are outputs of the caller's torch modules: without gradient mode they do not require grad
        training = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dec = dec_out.detach().contiguous().float()
        assert dec.shape[0] == c.N and dec.shape[1] >= c.dims.dec_params
        terms = torch.empty((5, c.N), dtype=torch.float32, device=dec.device)
        need_comp = training and c.q_kind != L.Q_OWN and plan.kind in ("miwae", "cvi", "mvb2")
        comp_lp = torch.empty((c.N, c.mix_C), dtype=torch.float32, device=dec.device) if need_comp else None
        L.check(lib.dmv_obj_terms(C.byref(c.cfg), c.q_kind, c.mix_C, L.ptr(dec), dec.shape[1], L.ptr(c.Xc), L.ptr(c.Mu),
                                  L.ptr(c.Z), c.Z.shape[1], L.ptr(c.enc_out), L.ptr(terms), L.ptr(comp_lp), st))
        c.dec_out, c.terms, c.comp_lp = dec, terms, comp_lp
        ctx.c = c
        ctx.set_materialize_grads(False)
        k = plan.kind
        if k in ("iwae", "miwae"):
            return (_bound(c, plan, training),)
        if k == "elbo_analytic":
            return (_combine(c, (1, 0, 0, 0, 0)).reshape(c.B) - _kl(c).reshape(c.B),)
        if k == "cvi":
            return (_combine(c, (1, 0, 1, 0, -1)),)
        if k == "mvb":
            if plan.kl_analytic:
                return (_combine(c, (0, 1, 0, 0, 0)) - _kl(c),)
            return (_combine(c, (0, 1, 1, -1, 0)),)
        if k == "mvb2":
            gen_obj = _combine(c, (1, 0, 1, 0, -1))
            enc_obj = (_combine(c, (0, 1, 0, 0, 0)) - _kl(c)) if plan.kl_analytic else _combine(c, (0, 1, 1, -1, 0))
            return gen_obj, enc_obj
        raise NotImplementedError(k)

    @staticmethod
    def backward(ctx, *gouts):
        c = ctx.c
        plan = c.plan
        lib, st = L.lib(), L.stream_ptr()
        nz, S, B, G = plan.nz, plan.nz * plan.I, c.B, c.G
        d_dec = None
        d_enc = None

        def per_sample(g_bg):
            g = g_bg.reshape(1, B, -1).float()
            return (g / nz).expand(S, B, G).contiguous().reshape(-1)

        def dec_part(wt, base_m, base_f):
            nonlocal d_dec
            out = torch.empty_like(c.dec_out)
            L.check(lib.dmv_obj_backward_dec(C.byref(c.cfg), L.ptr(c.dec_out), c.dec_out.shape[1], L.ptr(c.Xc), L.ptr(c.Mu),
                                             L.ptr(wt), None, None, base_m, base_f, L.ptr(out), out.shape[1], st))
            d_dec = out if d_dec is None else d_dec + out

        def enc_part(p, wt, zscale, dlogit, kl_wgt):
            nonlocal d_enc
            out = torch.zeros_like(c.enc_out)
            need_mix = p.base_qmix != 0.0
            L.check(lib.dmv_obj_backward_latent(
                C.byref(c.cfg), C.byref(p), L.ptr(c.enc_out), L.ptr(c.Z), L.ptr(c.eps), c.Z.shape[1], None, 0, L.ptr(wt),
                None, L.ptr(zscale), L.ptr(c.comp_lp) if need_mix else None, L.ptr(c.terms[4]) if need_mix else None,
                L.ptr(dlogit), L.ptr(kl_wgt), None, L.ptr(out), st))
            d_enc = out if d_enc is None else d_enc + out

        k = plan.kind
        if k in ("iwae", "miwae") and gouts[0] is not None:
            g_b = gouts[0].float().contiguous()
            wt = (c.wgt.reshape(S, B, G) * g_b.reshape(1, B, 1)).reshape(-1).contiguous()
            dec_part(wt, 1.0, 0.0)
            if c.reparam:
                return None, d_dec, _implicit_d_enc(c, None, wt, 1.0, -1.0, live=not plan.STL)
            mixture = c.q_kind == L.Q_STRATIFIED
            dl = (c.dlogit * g_b.reshape(B, 1)).contiguous() if mixture else None
            if mixture:
                p = _latent_params(prior=1.0, qmix=-1.0, mix_live=not plan.STL, mix_kind=L.Q_STRATIFIED)
            else:
                p = _latent_params(prior=1.0, qown=-1.0, own_live=not (plan.STL or plan.DREG), use_zscale=plan.DREG)
            enc_part(p, wt, c.zscale, dl, None)
        elif k == "elbo_analytic" and gouts[0] is not None:
            g_b = gouts[0].float().contiguous()
            dec_part(per_sample(g_b), 1.0, 0.0)
            enc_part(_latent_params(kl=-1.0), None, None, None, g_b.reshape(B, 1).contiguous())
        elif k == "cvi" and gouts[0] is not None:
            wt = per_sample(gouts[0])
            dec_part(wt, 1.0, 0.0)
            enc_part(_latent_params(prior=1.0, qmix=-1.0, mix_live=not plan.STL, mix_kind=L.Q_IMPUTATION), wt, None, None, None)
        elif k == "mvb" and gouts[0] is not None:
            wt = per_sample(gouts[0])
            dec_part(wt, 0.0, 1.0)
            if plan.kl_analytic:
                enc_part(_latent_params(kl=-1.0), None, None, None, gouts[0].float().contiguous())
            else:
                enc_part(_latent_params(prior=1.0, qown=-1.0, own_live=not plan.STL), wt, None, None, None)
        elif k == "mvb2":
            # both objectives are ordinary differentiable functions here; which parameters each one updates is the
            # caller's choice of backward(inputs=...) (vgiwae/shared/mvb2_manual_optimisation.py:6-8)
            g_gen, g_enc = gouts
            if g_gen is not None:
                wt = per_sample(g_gen)
                dec_part(wt, 1.0, 0.0)
                enc_part(_latent_params(prior=1.0, qmix=-1.0, mix_live=not plan.STL, mix_kind=L.Q_IMPUTATION), wt, None, None, None)
            if g_enc is not None:
                wt = per_sample(g_enc)
                dec_part(wt, 0.0, 1.0)
                if plan.kl_analytic:
                    enc_part(_latent_params(kl=-1.0), None, None, None, g_enc.float().contiguous())
                else:
                    enc_part(_latent_params(prior=1.0, qown=-1.0, own_live=not plan.STL), wt, None, None, None)
        return None, d_dec, d_enc


def evaluate(dims: ModelDims, plan: _Plan, enc_out, decoder, X, M, eps=None, comp_idx=None):
    """Runs one objective with the caller's networks.  enc_out [rows, P] is the encoder output (rows = B, or B*K
    for the DeMissVAE objectives); decoder maps Z [N, Zd] to [N, P_dec].  Returns the tuple of outputs."""
    c = _setup(dims, plan, X, M, enc_out)
    Z = _Sample.apply(c, enc_out, eps, comp_idx)
    dec_out = decoder(Z)
    return _Objective.apply(c, dec_out.reshape(c.N, -1), enc_out)


# ------------------------------------------------------------------------------------------ estimators
# Same names, keywords and return values as objectives.py; `encode(X2, M2)` maps flattened rows [rows, D] and
# their masks to encoder outputs [rows, P]; `decode(Z)` maps [N, Zd] to decoder outputs.
def _enc_out(dims, plan, encode, X, M):
    D = dims.D
    if plan.kind in ("mvb2", "cvi", "mvb"):
        Xin = X.reshape(-1, D)                      # imputed data goes in as it is
    else:
        Xin = (X * M).reshape(-1, D)                # missing entries zeroed (vgiwae/models/vae.py:285-293)
    return encode(Xin.float(), M.reshape(-1, D)).reshape(Xin.shape[0], -1)


def _ext(dims, plan, encode, decode, X, M, eps, comp_idx=None):
    return evaluate(dims, plan, _enc_out(dims, plan, encode, X, M), decode, X, M, eps, comp_idx)


def estimate_iwelbo(dims, encode, decode, X, M, *, num_latent_samples=1, num_importance_samples=1, STL=False, DREG=False,
                    eps=None, comp_idx=None):
    assert not (STL and DREG), "Cannot use STL and DREG gradients at the same time."
    plan = _Plan("iwae", num_latent_samples, num_importance_samples, STL=STL, DREG=DREG)
    (per_b,) = _ext(dims, plan, encode, decode, X, M, eps, comp_idx)
    return per_b.mean(), per_b


def estimate_elbo(dims, encode, decode, X, M, *, num_latent_samples=1, kl_analytic=True, STL=False, eps=None, comp_idx=None):
    assert not (STL and kl_analytic), '"Sticking the Landing" cannot be used with analytical KL.'
    plan = _Plan("elbo_analytic", num_latent_samples, 1) if kl_analytic else _Plan("iwae", num_latent_samples, 1, STL=STL)
    (per_b,) = _ext(dims, plan, encode, decode, X, M, eps, comp_idx)
    return per_b.mean(), per_b


def estimate_multiple_iwelbo(dims, encode, decode, X, M, *, num_latent_samples=1, num_importance_samples_for_each_component=1,
                             STL=False, looser=False, eps=None):
    plan = _Plan("miwae", num_latent_samples, num_importance_samples_for_each_component, STL=STL, looser=looser)
    (per_b,) = _ext(dims, plan, encode, decode, X, M, eps)
    return per_b.mean(), per_b


def estimate_multiple_elbo(dims, encode, decode, X, M, *, num_latent_samples=1, STL=False, eps=None):
    plan = _Plan("miwae", num_latent_samples, 1, STL=STL, looser=True)
    (per_b,) = _ext(dims, plan, encode, decode, X, M, eps)
    return per_b.mean(), per_b


def mvb2_objectives(dims, encode, decode, Xt, M, *, num_latent_samples=1, STL=False, kl_analytic=False, eps=None):
    plan = _Plan("mvb2", num_latent_samples, 1, STL=STL, kl_analytic=kl_analytic)
    return _ext(dims, plan, encode, decode, Xt, M, eps)


def cvi_objective(dims, encode, decode, Xt, M, *, num_latent_samples=1, STL=False, eps=None):
    (obj,) = _ext(dims, _Plan("cvi", num_latent_samples, 1, STL=STL), encode, decode, Xt, M, eps)
    return obj


def mvb_objective(dims, encode, decode, Xt, M, *, num_latent_samples=1, STL=False, kl_analytic=False, eps=None):
    (obj,) = _ext(dims, _Plan("mvb", num_latent_samples, 1, STL=STL, kl_analytic=kl_analytic), encode, decode, Xt, M, eps)
    return obj
