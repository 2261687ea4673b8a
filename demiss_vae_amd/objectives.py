"""The masked VAE / IWAE / MissSVAE / MissSIWAE / DeMissVAE objectives as fused HIP pipelines.

One call = encoder MLP -> latent sampling -> decoder MLP -> per-sample terms -> bound, all through
libdmv; `_FusedObjective.backward` is the hand-derived backward (decoder-side gradient, decoder
MLP backward, latent-side gradient, encoder MLP backward).  The public functions mirror the
reference's estimators:

  estimate_elbo            vgiwae/models/vae.py:447-473
  estimate_iwelbo          vgiwae/models/iwae.py:125-154
  estimate_multiple_elbo   vgiwae/models/multiple_vae.py:162-195
  estimate_multiple_iwelbo vgiwae/models/multiple_iwae.py:201-237
  mvb2_objectives / cvi_objective / mvb_objective   vgiwae/models/mvbvae.py:67-190

Noise: pass `eps` (standard normal, reference sample shape) for reproducible parity runs, or leave
it None to draw from the on-device Philox stream (demiss_vae_amd.rng).
"""
import ctypes as C
import dataclasses
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from . import rng
from .mlp import (PackedNet, GeneratedDropout, mlp_forward, mlp_backward, mlp_backward_chain, mlp_wgrad, mlp_wgrad_pair, pad32,
                  side_stream, overlap_pays)


@dataclass(frozen=True)
class ModelDims:
    D: int
    Zd: int
    gen_dist: str = "normal_with_eps"    # normal | normal_with_eps | bern | bern_m1_p1
    q_dist: str = "normal_with_eps"      # normal | normal_with_eps | stratified[_with_eps] | reparam[_with_eps]
    C: int = 1
    encoder_use_mis_mask: bool = False

    @property
    def q_eps(self):
        return 1 if self.q_dist.endswith("with_eps") else 0

    @property
    def q_is_mixture(self):
        """stratified mixture: one sample row per component"""
        return self.q_dist.startswith("stratified")

    @property
    def q_is_reparam(self):
        """mixture sampled ancestrally (one row per datapoint) with implicit reparameterisation gradients"""
        return self.q_dist.startswith("reparam")

    @property
    def dec_params(self):
        return 2 * self.D if self.gen_dist in ("normal", "normal_with_eps") else self.D


@dataclass(frozen=True)
class _Plan:
    kind: str            # iwae | elbo_analytic | miwae | mvb2 | cvi | mvb
    nz: int
    I: int
    STL: bool = False
    DREG: bool = False
    looser: bool = False
    kl_analytic: bool = False
    grad: bool = True    # gradient mode at call time (inside an autograd function's forward it always reads "off")
    reduce_mean: bool = False   # mvb2: return the means over (b, k) instead of the [B, K] objectives


def _u8(M: torch.Tensor) -> torch.Tensor:
    M = M.contiguous()
    return M.view(torch.uint8) if M.dtype == torch.bool else M.to(torch.uint8)


class _Ctx:
    pass


FUSE_SMALL_FORWARD = True      # tests switch it off to compare with the three separate launches
FUSE_SMALL_BOUND = True        # ... and this one to compare with forward and objective as two launches
bound_launches = 0             # how many forward passes formed their bound in the same launch (tests look at it)


def fused_forward(cfg, enc: PackedNet, dec: PackedNet, in_mode, X2, M2, eps2, save: bool, keep_eps: bool, n_prior: int = 0,
                  prior_raw: float = 0.0, enc_out=None, Z=None, defer: bool = False):
    """Encoder -> z = mu + sigma eps -> decoder in one launch (dmv_vae_forward) when the shapes allow it, else None.
    X2 / M2 [enc_rows, D]; eps2 [N, Zd] or None (device Philox).  Returns (enc_out, enc_saved, Z, eps_used, dec_out,
    dec_saved) exactly as the separate launches leave them.  n_prior / prior_raw: see dmv_vae_forward (the caller
    passes enc_out [B, G, ld] with the prior rows filled in).  enc_out / Z: buffers to write into.
    defer: do not launch yet; a seventh element `launch(obj)` is returned, to be called once with a DmvObjFwd (the bound
    in the same launch, dmv_vae_forward_bound; returns False if the shapes do not allow that and nothing ran) or None."""
    lib = L.lib()
    if not FUSE_SMALL_FORWARD or not lib.dmv_vae_forward_fits(C.byref(cfg), C.byref(enc.c), C.byref(dec.c)):
        return None
    if save and ((enc.dims.dropout > 0 and enc.dims.nb > 0) or (dec.dims.dropout > 0 and dec.dims.nb > 0)):
        return None
    dev = X2.device
    Zd = cfg.Zd
    enc_rows = cfg.B * ((cfg.G - n_prior) if cfg.x_per_component else 1)
    N = cfg.nz * cfg.I * cfg.B * cfg.G
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    if enc_out is None:
        assert n_prior == 0
        enc_out = new(enc_rows, pad32(enc.dims.out_dim))
    dec_out = new(N, pad32(dec.dims.out_dim))
    if Z is None:
        Z = new(N, Zd)
    enc_saved = new(lib.dmv_mlp_saved_floats(C.byref(enc.c), enc_rows)) if save else None
    dec_saved = new(lib.dmv_mlp_saved_floats(C.byref(dec.c), N)) if save else None
    if eps2 is not None:
        eps_used, seed, off, base = eps2, 0, 0, None
    else:
        eps_used = new(N, Zd) if keep_eps else None
        seed, off = rng.next_offset(N * Zd)
        base = rng.base_ptr()
    def launch(obj=None):
        eo = L.ptr(eps_used) if eps2 is None else None
        if obj is not None:
            rc = lib.dmv_vae_forward_bound(C.byref(cfg), C.byref(enc.c), L.ptr(enc.packed()), in_mode, L.ptr(X2), L.ptr(M2),
                                           C.byref(dec.c), L.ptr(dec.packed()), L.ptr(eps2), seed, off, base, L.ptr(enc_out),
                                           L.ptr(enc_saved), L.ptr(Z), Zd, eo, L.ptr(dec_out), L.ptr(dec_saved), C.byref(obj),
                                           L.stream_ptr())
            if rc == L.E_UNSUPPORTED:
                return False
            L.check(rc)
            return True
        L.check(lib.dmv_vae_forward(C.byref(cfg), C.byref(enc.c), L.ptr(enc.packed()), in_mode, L.ptr(X2), L.ptr(M2), C.byref(dec.c),
                                    L.ptr(dec.packed()), L.ptr(eps2), seed, off, base, n_prior, prior_raw, L.ptr(enc_out),
                                    L.ptr(enc_saved), L.ptr(Z), Zd, eo, L.ptr(dec_out), L.ptr(dec_saved), L.stream_ptr()))
        return True

    if defer:
        return enc_out, enc_saved, Z, eps_used, dec_out, dec_saved, launch
    launch()
    return enc_out, enc_saved, Z, eps_used, dec_out, dec_saved


def _forward_core(dims: ModelDims, plan: _Plan, enc: PackedNet, dec: PackedNet, X, M, eps, enc_keep, dec_keep,
                  training: bool, comp_idx=None, bound_follows: bool = False):
    """Runs the forward pipeline; returns a context holding every buffer the backward needs.
    bound_follows: the caller goes straight on to _objective_forward with a logsumexp bound; the one-launch forward is
    then left pending (c.pending_forward) so that the bound can be formed in the same launch."""
    lib = L.lib()
    st = L.stream_ptr()
    dev = X.device
    demiss = plan.kind in ("mvb2", "cvi", "mvb", "terms")
    c = _Ctx()
    if demiss:
        B, K, D = X.shape
        G, enc_rows = K, B * K
    else:
        B, D = X.shape
        G = dims.C if dims.q_is_mixture else 1
        enc_rows = B
    assert D == dims.D
    Xc = X.contiguous().float()
    Mu = _u8(M)
    assert Mu.shape == Xc.shape
    nz, I, Zd = plan.nz, plan.I, dims.Zd
    N = nz * I * B * G

    # ---- encoder (vgiwae/models/vae.py:80-91; the X*M products of the non-imputing estimators)
    if dims.encoder_use_mis_mask:
        in_mode = L.IN_CONCAT if demiss else L.IN_CONCAT_MASKED
    else:
        in_mode = L.IN_PLAIN if demiss else L.IN_MASKED
    ldE = pad32(enc.dims.out_dim)

    cfg = L.DmvObjCfg()
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, G, nz, I, D, Zd
    cfg.gen_dist, cfg.q_eps = L.GEN[dims.gen_dist], dims.q_eps
    cfg.x_per_component = 1 if demiss else 0
    if demiss:
        cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = K * ldE, ldE, Zd, 0
    elif dims.q_is_mixture or dims.q_is_reparam:
        cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, Zd, dims.C * Zd, 2 * dims.C * Zd
    else:
        cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, 0, Zd, 0
    fused = None
    if not dims.q_is_reparam and comp_idx is None and enc_keep is None and dec_keep is None:
        # small batches: encoder, sampling and decoder in one launch
        defer = bound_follows and FUSE_SMALL_BOUND and not demiss
        fused = fused_forward(cfg, enc, dec, in_mode, Xc.reshape(enc_rows, D), Mu.reshape(enc_rows, D),
                              eps.reshape(N, Zd).contiguous().float() if eps is not None else None, training, training,
                              defer=defer)
    c.pending_forward = None
    if fused is not None:
        enc_out, enc_saved, Z, eps_used, dec_out, dec_saved = fused[:6]
        c.pending_forward = fused[6] if len(fused) > 6 else None
    else:
        enc_out, enc_saved = mlp_forward(enc, Xc, D, enc_rows, in_mode=in_mode, D=D, m=Mu, ldm=D, save=training,
                                         drop_keep=enc_keep)
        if dims.q_is_reparam:
            comp_idx, eps = reparam_components(dims, cfg, enc_out, comp_idx, eps)

        # ---- z = mu + sigma * eps
        Z = torch.empty((N, Zd), dtype=torch.float32, device=dev)
        if eps is not None:
            eps_used = eps.reshape(N, Zd).contiguous().float()
            L.check(lib.dmv_latent_sample(C.byref(cfg), L.ptr(enc_out), L.ptr(eps_used), 0, 0, None, Zd, L.ptr(comp_idx),
                                          L.ptr(Z), None, st))
        else:
            eps_used = torch.empty((N, Zd), dtype=torch.float32, device=dev) if training else None
            seed, off = rng.next_offset(N * Zd)
            L.check(lib.dmv_latent_sample(C.byref(cfg), L.ptr(enc_out), None, seed, off, rng.base_ptr(), Zd, L.ptr(comp_idx),
                                          L.ptr(Z), L.ptr(eps_used), st))

        # ---- decoder
        dec_out, dec_saved = mlp_forward(dec, Z, Zd, N, save=training, drop_keep=dec_keep)

    if demiss:
        q_kind, mix_C = L.Q_IMPUTATION, K
    elif dims.q_is_mixture or dims.q_is_reparam:
        q_kind, mix_C = L.Q_STRATIFIED, dims.C
    else:
        q_kind, mix_C = L.Q_OWN, 1
    c.cfg, c.B, c.G, c.N, c.demiss, c.q_kind, c.mix_C = cfg, B, G, N, demiss, q_kind, mix_C
    c.Xc, c.Mu, c.enc_out, c.enc_saved, c.Z, c.eps, c.dec_out, c.dec_saved = Xc, Mu, enc_out, enc_saved, Z, eps_used, dec_out, dec_saved
    c.terms, c.comp_lp, c.enc_rows = None, None, enc_rows
    c.enc_keep, c.dec_keep = enc_keep, dec_keep
    c.reparam = dims.q_is_reparam
    c.wgt = c.zscale = c.dlogit = None
    return c


def _objective_forward(c, plan: _Plan, training: bool, *, kind, use_logpi=False, coefA=None, coefB=None, want_terms=False,
                       want_means=False):
    """One launch of dmv_obj_forward over the pipeline's decoder outputs: per-sample terms, bound / objectives and the
    buffers the backward reads (softmax weights, DREG scales, mixture-logit gradient, component log-densities).
    Returns (bound [B], mean, -mean) for the bound kinds, (objA, objB, meanA, meanB) for the per-component kind, terms for -1."""
    lib, st = L.lib(), L.stream_ptr()
    dev = c.dec_out.device
    B, G, N = c.B, c.G, c.N
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    f = L.DmvObjFwd()
    f.q_kind, f.mix_C, f.kind, f.use_logpi, f.dreg = c.q_kind, c.mix_C, kind, int(use_logpi), int(plan.DREG)
    # the mixture enters the loss (not stop-gradient) in these plans: their backward needs the component densities
    need_comp = training and c.q_kind != L.Q_OWN and plan.kind in ("miwae", "cvi", "terms") and not c.reparam
    if want_terms or need_comp:
        c.terms = new(5, N)
        f.terms = L.addr(c.terms)
    if need_comp:
        c.comp_lp = new(N, c.mix_C)
        f.comp_lp = L.addr(c.comp_lp)
    out = None
    if kind in (L.BOUND_LSE, L.BOUND_WEIGHTED):
        per_b, mean, neg = new(B), new(), new()
        f.bound, f.mean_out, f.neg_mean_out = L.addr(per_b), L.addr(mean), L.addr(neg)
        if training:
            c.wgt = new(N)
            f.wgt = L.addr(c.wgt)
            if plan.DREG:
                c.zscale = new(N)
                f.zscale = L.addr(c.zscale)
            if c.q_kind == L.Q_STRATIFIED and not c.reparam:
                c.dlogit = new(B, G)
                f.dlogit = L.addr(c.dlogit)
        out = (per_b, mean, neg)
    elif kind == L.BOUND_PER_COMPONENT:
        objA = new(B, G)
        objB = new(B, G) if coefB is not None else None
        mA = new() if want_means else None
        mB = new() if (want_means and coefB is not None) else None
        f.coefA = (C.c_float * 5)(*coefA)
        f.coefB = (C.c_float * 5)(*(coefB if coefB is not None else (0, 0, 0, 0, 0)))
        f.objA, f.objB, f.meanA_out, f.meanB_out = L.addr(objA), L.addr(objB), L.addr(mA), L.addr(mB)
        out = (objA, objB, mA, mB)
    f.ticket = L.ticket_ptr()
    pending, c.pending_forward = getattr(c, "pending_forward", None), None
    if pending is not None:
        # the forward has not run yet: with the bound in its launch if the shapes allow, else by itself
        if kind == L.BOUND_LSE and ((c.q_kind == L.Q_OWN and not use_logpi) or (c.q_kind == L.Q_STRATIFIED and use_logpi)) \
                and pending(f):
            global bound_launches
            bound_launches += 1
            return out
        pending(None)
    rc = lib.dmv_obj_forward(C.byref(c.cfg), C.byref(f), L.ptr(c.dec_out), c.dec_out.shape[1], L.ptr(c.Xc), L.ptr(c.Mu),
                             L.ptr(c.Z), c.Z.shape[1], L.ptr(c.enc_out), st)
    if rc == L.E_UNSUPPORTED and c.cfg.nz * c.cfg.I * c.G > 1024:
        # more samples per datapoint than the one-launch kernel stages in LDS (about 15 000): separate launches for
        # the per-sample terms and for the bound / objectives over them, which loop over the samples
        return _objective_unfused(c, plan, training, kind, coefA, coefB, want_means, need_comp)
    L.check(rc)
    return out if out is not None else c.terms


def _objective_unfused(c, plan: _Plan, training: bool, kind, coefA, coefB, want_means, need_comp):
    """What _objective_forward returns, from dmv_obj_terms followed by dmv_obj_bound / dmv_obj_combine."""
    lib, st = L.lib(), L.stream_ptr()
    dev = c.dec_out.device
    c.terms = torch.empty((5, c.N), dtype=torch.float32, device=dev)
    c.comp_lp = torch.empty((c.N, c.mix_C), dtype=torch.float32, device=dev) if need_comp else None
    L.check(lib.dmv_obj_terms(C.byref(c.cfg), c.q_kind, c.mix_C, L.ptr(c.dec_out), c.dec_out.shape[1], L.ptr(c.Xc), L.ptr(c.Mu),
                              L.ptr(c.Z), c.Z.shape[1], L.ptr(c.enc_out), L.ptr(c.terms), L.ptr(c.comp_lp), st))
    if kind in (L.BOUND_LSE, L.BOUND_WEIGHTED):
        per_b = _bound(c, plan, training)
        mean = per_b.mean()
        return per_b, mean, -mean
    if kind == L.BOUND_PER_COMPONENT:
        objA = _combine(c, coefA)
        objB = _combine(c, coefB) if coefB is not None else None
        mA = objA.mean() if want_means else None
        mB = objB.mean() if (want_means and objB is not None) else None
        return objA, objB, mA, mB
    return c.terms


def reparam_components(dims: ModelDims, cfg, enc_out, comp_idx, eps):
    """Ancestral sampling of a mixture encoder: the component of every sample row (n = s*B + b), drawn from the
    encoder's logits on the device unless given, and the noise of that component when explicit noise
    [S.., B, C, Zd] is given (MixtureSameFamily.sample draws all components, then gathers)."""
    B, S, Cn, Zd = cfg.B, cfg.nz * cfg.I, dims.C, dims.Zd
    dev = enc_out.device
    if comp_idx is None:
        logits = enc_out[:, 2 * Cn * Zd:2 * Cn * Zd + Cn].contiguous()
        draws = torch.empty((B, S), dtype=torch.int64, device=dev)
        seed, off = rng.next_offset(B * S)
        L.check(L.lib().dmv_categorical_draw(B, Cn, S, L.ptr(logits), None, seed, off, rng.base_ptr(), L.ptr(draws), L.stream_ptr()))
        comp_idx = draws.t().contiguous().reshape(-1)
    else:
        comp_idx = comp_idx.reshape(-1).to(torch.int64).contiguous()
    if eps is not None and eps.numel() == S * B * Cn * Zd:
        e = eps.reshape(S * B, Cn, Zd)
        eps = torch.gather(e, 1, comp_idx.view(-1, 1, 1).expand(-1, 1, Zd)).squeeze(1).contiguous()
    return comp_idx, eps


def _bound(c, plan: _Plan, training: bool):
    lib, st = L.lib(), L.stream_ptr()
    dev = c.terms.device
    B, G, N = c.B, c.G, c.N
    per_b = torch.empty(B, dtype=torch.float32, device=dev)
    wgt = torch.empty(N, dtype=torch.float32, device=dev) if training else None
    zscale = torch.empty(N, dtype=torch.float32, device=dev) if (training and plan.DREG) else None
    mixture = c.q_kind == L.Q_STRATIFIED and not getattr(c, "reparam", False)
    dlogit = torch.zeros((B, G), dtype=torch.float32, device=dev) if (training and mixture) else None
    kind = L.BOUND_WEIGHTED if plan.looser else L.BOUND_LSE
    L.check(lib.dmv_obj_bound(C.byref(c.cfg), kind, c.q_kind, 1 if (mixture and not plan.looser) else 0,
                              1 if plan.DREG else 0, L.ptr(c.enc_out), L.ptr(c.terms), None, L.ptr(per_b), None,
                              L.ptr(wgt), L.ptr(zscale), L.ptr(dlogit), st))
    c.wgt, c.zscale, c.dlogit = wgt, zscale, dlogit
    return per_b


def _combine(c, coef):
    lib, st = L.lib(), L.stream_ptr()
    obj = torch.empty((c.B, c.G), dtype=torch.float32, device=c.terms.device)
    arr = (C.c_float * 5)(*coef)
    L.check(lib.dmv_obj_combine(C.byref(c.cfg), arr, L.ptr(c.terms), L.ptr(obj), None, st))
    return obj


def _kl(c):
    lib, st = L.lib(), L.stream_ptr()
    kl = torch.empty((c.B, c.G), dtype=torch.float32, device=c.enc_out.device)
    L.check(lib.dmv_kl_std_normal(C.byref(c.cfg), L.ptr(c.enc_out), L.ptr(kl), st))
    return kl


def _dec_side(c, dec: PackedNet, wgt, base_m, base_f, need_dx, need_param_grads, overlap=None, gb=None, gscale=None,
              complete=False):
    """Decoder-side gradient and the decoder's backward pass.  Row weights = gscale * gb[b] * wgt[n] (each optional).
    overlap: a list; when given (and the batch is small) the decoder's weight-gradient GEMMs go to the side stream
    while the caller carries on upstream, and the list receives what must stay alive until the caller has joined"""
    lib, st = L.lib(), L.stream_ptr()
    d_dec = torch.empty_like(c.dec_out)
    L.check(lib.dmv_obj_backward_dec(C.byref(c.cfg), L.ptr(c.dec_out), c.dec_out.shape[1], L.ptr(c.Xc), L.ptr(c.Mu),
                                     L.ptr(wgt), L.ptr(gb), L.ptr(gscale), base_m, base_f, L.ptr(d_dec), d_dec.shape[1], st))
    side = side_stream(d_dec.device) if (overlap is not None and need_dx and need_param_grads and overlap_pays(c.N)) else None
    if side is not None:
        overlap.append(side)
    return mlp_backward(dec, c.dec_saved, d_dec, c.N, need_dx=need_dx, need_param_grads=need_param_grads,
                        drop_keep=c.dec_keep, wgrad_stream=side, keepalive=overlap, complete=complete)


def _join(overlap):
    """the current stream waits for the side stream's weight gradients; only then may the buffers go"""
    if overlap:
        torch.cuda.current_stream().wait_stream(overlap[0])
        overlap.clear()


def _enc_side(c, enc: PackedNet, p: L.DmvLatentBwd, dZ, wgt, zscale, dlogit, kl_wgt, gb=None, gscale=None):
    lib, st = L.lib(), L.stream_ptr()
    d_enc = torch.empty_like(c.enc_out)          # the kernel writes every entry, padding included
    need_mix = p.base_qmix != 0.0
    L.check(lib.dmv_obj_backward_latent(
        C.byref(c.cfg), C.byref(p), L.ptr(c.enc_out), L.ptr(c.Z), L.ptr(c.eps), c.Z.shape[1],
        L.ptr(dZ), dZ.shape[1] if dZ is not None else 0, L.ptr(wgt), L.ptr(gb), L.ptr(zscale),
        L.ptr(c.comp_lp) if need_mix else None, L.ptr(c.terms[4]) if need_mix else None,
        L.ptr(dlogit), L.ptr(kl_wgt), L.ptr(gscale), L.ptr(d_enc), st))
    _, grads = mlp_backward(enc, c.enc_saved, d_enc, c.enc_rows, need_dx=False, need_param_grads=True,
                            drop_keep=c.enc_keep)
    return grads


FUSE_SMALL_BACKWARD = True     # tests switch it off to compare with the separate launches
PAIR_ROWS_MAX = 8192           # up to this many decoder rows the separate launches also end in one pair of weight-gradient launches


def _paired_backward(c, enc: PackedNet, dec: PackedNet, p: L.DmvLatentBwd, wgt_dec, base_m, base_f, wgt, zscale, dlogit,
                     kl_wgt, gb, gscale, complete):
    """_dec_side + _enc_side with the weight gradients of both networks last, as one pair of launches: on a batch that
    leaves the weight-gradient GEMMs less than one round of waves each, the decoder's - launched beside the latent-side
    kernel and the encoder's chain on a second stream - delay those (their workgroups wait for a free compute unit), and
    two launches queue two rounds.  Returns (decoder gradients, encoder gradients), or None: the caller takes the two
    streams."""
    if not PAIR_WGRAD or not (0 < c.N <= PAIR_ROWS_MAX) or c.enc_rows <= 0:
        return None
    if 2 * (enc.dims.nb + dec.dims.nb) + 4 > 24:
        return None
    # an encoder deeper than the decoder: its long chain on few compute units is the window the decoder's weight
    # gradients fill on the second stream (miniboone, 5 blocks against 2, B = 1024: the pair is 2 % slower; equal depths,
    # gas and hepmass at B = 1024: 4 - 5 % faster)
    if enc.dims.nb > dec.dims.nb:
        return None
    lib, st = L.lib(), L.stream_ptr()
    d_dec = torch.empty_like(c.dec_out)
    L.check(lib.dmv_obj_backward_dec(C.byref(c.cfg), L.ptr(c.dec_out), c.dec_out.shape[1], L.ptr(c.Xc), L.ptr(c.Mu),
                                     L.ptr(wgt_dec), L.ptr(gb), L.ptr(gscale), base_m, base_f, L.ptr(d_dec), d_dec.shape[1], st))
    dZ, ws_dec, nb_dec = mlp_backward_chain(dec, c.dec_saved, d_dec, c.N, need_dx=True, drop_keep=c.dec_keep)
    d_enc = torch.empty_like(c.enc_out)          # the kernel writes every entry, padding included
    need_mix = p.base_qmix != 0.0
    L.check(lib.dmv_obj_backward_latent(
        C.byref(c.cfg), C.byref(p), L.ptr(c.enc_out), L.ptr(c.Z), L.ptr(c.eps), c.Z.shape[1], L.ptr(dZ), dZ.shape[1], L.ptr(wgt),
        L.ptr(gb), L.ptr(zscale), L.ptr(c.comp_lp) if need_mix else None, L.ptr(c.terms[4]) if need_mix else None,
        L.ptr(dlogit), L.ptr(kl_wgt), L.ptr(gscale), L.ptr(d_enc), st))
    _, ws_enc, nb_enc = mlp_backward_chain(enc, c.enc_saved, d_enc, c.enc_rows, need_dx=False, drop_keep=c.enc_keep)
    pair = mlp_wgrad_pair(dec, c.dec_saved, d_dec, c.N, ws_dec, nb_dec, enc, c.enc_saved, d_enc, c.enc_rows, ws_enc, nb_enc,
                          complete_a=complete)
    assert pair is not None
    return pair
PAIR_WGRAD = True              # ... and this one to compare with the two networks' weight gradients on two streams


def _one_launch_backward(c, enc: PackedNet, dec: PackedNet, p: L.DmvLatentBwd, wgt_dec, base_m, base_f, wgt, zscale, kl_wgt,
                         gb_dec, gb_lat, gscale, dec_param_grads: bool, complete: bool = False, dlogit=None):
    """Decoder-side gradient, then the decoder's chain, the latent-side gradient and the encoder's chain in one launch
    (dmv_vae_backward), then the weight gradients of both networks side by side on two streams.  On batches small enough
    for the 16-row kernels whose tiles can hold a unit's samples; returns (decoder parameter gradients or None, encoder
    parameter gradients), or None when the library does not take the case."""
    lib, st = L.lib(), L.stream_ptr()
    if not FUSE_SMALL_BACKWARD or c.reparam:
        return None
    mixture = p.base_qmix != 0.0
    if not lib.dmv_vae_backward_fits(C.byref(c.cfg), C.byref(enc.c), C.byref(dec.c), int(mixture)):
        return None
    dev = c.dec_out.device
    d_dec = torch.empty_like(c.dec_out)
    dg = None
    if d_dec.shape[1] <= 256:        # short decoder rows: the launch forms the decoder-side gradient itself
        dg = L.DmvDecGrad(L.addr(c.dec_out), c.dec_out.shape[1], L.addr(c.Xc), L.addr(c.Mu), L.addr(wgt_dec), L.addr(gb_dec),
                          base_m, base_f)
    else:
        L.check(lib.dmv_obj_backward_dec(C.byref(c.cfg), L.ptr(c.dec_out), c.dec_out.shape[1], L.ptr(c.Xc), L.ptr(c.Mu),
                                         L.ptr(wgt_dec), L.ptr(gb_dec), L.ptr(gscale), base_m, base_f, L.ptr(d_dec), d_dec.shape[1], st))
    d_enc = torch.empty_like(c.enc_out)
    nb_dec = lib.dmv_mlp_backward_ws_bytes(C.byref(dec.c), c.N)
    nb_enc = lib.dmv_mlp_backward_ws_bytes(C.byref(enc.c), c.enc_rows)
    ws_dec = torch.empty((nb_dec + 3) // 4, dtype=torch.float32, device=dev)
    ws_enc = torch.empty((nb_enc + 3) // 4, dtype=torch.float32, device=dev)
    keep = lambda k: L.DROP_GENERATED if isinstance(k, GeneratedDropout) else L.ptr(k)
    L.check(lib.dmv_vae_backward(
        C.byref(c.cfg), C.byref(p), C.byref(dg) if dg is not None else None, C.byref(dec.c), L.ptr(dec.packed()), L.ptr(c.dec_saved), keep(c.dec_keep), L.ptr(d_dec),
        d_dec.shape[1], L.ptr(ws_dec), nb_dec, C.byref(enc.c), L.ptr(enc.packed()), L.ptr(c.enc_saved), keep(c.enc_keep),
        L.ptr(d_enc), L.ptr(ws_enc), nb_enc, L.ptr(c.enc_out), L.ptr(c.Z), L.ptr(c.eps), c.Z.shape[1], L.ptr(wgt), L.ptr(gb_lat),
        L.ptr(zscale), L.ptr(kl_wgt), L.ptr(c.comp_lp) if mixture else None, L.ptr(c.terms[4]) if mixture else None,
        L.ptr(dlogit), L.ptr(gscale), st))
    g_dec = None
    side = None
    if dec_param_grads and PAIR_WGRAD:
        pair = mlp_wgrad_pair(dec, c.dec_saved, d_dec, c.N, ws_dec, nb_dec, enc, c.enc_saved, d_enc, c.enc_rows, ws_enc, nb_enc,
                              complete_a=complete)
        if pair is not None:
            return pair
    if dec_param_grads:
        side = side_stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        g_dec = mlp_wgrad(dec, c.dec_saved, d_dec, c.N, ws_dec, nb_dec, stream=side, complete=complete)
    g_enc = mlp_wgrad(enc, c.enc_saved, d_enc, c.enc_rows, ws_enc, nb_enc)
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
    return g_dec, g_enc


def _implicit_d_enc(c, dZ, wgt, base_prior, base_qmix, live=False):
    """Encoder-output gradient through ancestrally sampled mixture latents (implicit reparameterisation); live: also
    the score of q in its own parameters at fixed z (bounds without "sticking the landing")."""
    lib, st = L.lib(), L.stream_ptr()
    Cn = c.mix_C
    ws = torch.empty(lib.dmv_obj_implicit_ws_floats(C.byref(c.cfg), Cn), dtype=torch.float32, device=c.enc_out.device)
    d_enc = torch.zeros_like(c.enc_out)
    L.check(lib.dmv_obj_backward_latent_implicit(C.byref(c.cfg), Cn, L.ptr(c.enc_out), L.ptr(c.Z), c.Z.shape[1], L.ptr(dZ),
                                                 dZ.shape[1] if dZ is not None else 0, L.ptr(wgt), base_prior, base_qmix,
                                                 int(live), L.ptr(ws), L.ptr(d_enc), st))
    return d_enc


def _latent_params(**kw):
    p = L.DmvLatentBwd()
    p.base_prior = kw.get("prior", 0.0)
    p.base_qown = kw.get("qown", 0.0)
    p.base_qmix = kw.get("qmix", 0.0)
    p.base_w = kw.get("w", 1.0)
    p.kl_coef = kw.get("kl", 0.0)
    p.own_live = int(kw.get("own_live", False))
    p.mix_live = int(kw.get("mix_live", False))
    p.use_zscale = int(kw.get("use_zscale", False))
    p.mix_kind = kw.get("mix_kind", L.Q_OWN)   # no mixture logits unless asked for
    return p


class _FusedObjective(torch.autograd.Function):
    """outputs: bound kinds (iwae, miwae, elbo_analytic): (batch mean, per-datapoint bound [B], minus the batch mean =
    the training loss); DeMissVAE kinds: the objective(s) [B, K], or their means over (b, k) with plan.reduce_mean."""

    @staticmethod
    def forward(ctx, dims, plan, enc, dec, X, M, eps, enc_keep, dec_keep, comp_idx, *params):
        # needs_input_grad mirrors requires_grad of the inputs even when gradients are switched off (validation)
        training = plan.grad and any(ctx.needs_input_grad[10:])
        if dims.q_is_reparam:
            assert plan.kind == "iwae" and not plan.DREG, "reparametrised mixtures: ELBO (MC KL) and IWAE bounds only"
        c = _forward_core(dims, plan, enc, dec, X, M, eps, enc_keep, dec_keep, training, comp_idx=comp_idx,
                          bound_follows=plan.kind in ("iwae", "miwae") and not plan.looser)
        ctx.c, ctx.dims, ctx.plan, ctx.enc, ctx.dec = c, dims, plan, enc, dec
        ctx.set_materialize_grads(False)
        k = plan.kind
        pc = L.BOUND_PER_COMPONENT
        if k in ("iwae", "miwae"):
            mixture = c.q_kind == L.Q_STRATIFIED and not c.reparam
            per_b, mean, neg = _objective_forward(c, plan, training, kind=L.BOUND_WEIGHTED if plan.looser else L.BOUND_LSE,
                                                  use_logpi=mixture and not plan.looser)
            return mean, per_b, neg
        if k == "elbo_analytic":
            gen = _objective_forward(c, plan, training, kind=pc, coefA=(1, 0, 0, 0, 0))[0]
            per_b = gen.reshape(c.B) - _kl(c).reshape(c.B)
            mean = per_b.mean()
            return mean, per_b, -mean
        if k == "cvi":
            return (_objective_forward(c, plan, training, kind=pc, coefA=(1, 0, 1, 0, -1))[0],)
        if k == "mvb":
            if plan.kl_analytic:
                return (_objective_forward(c, plan, training, kind=pc, coefA=(0, 1, 0, 0, 0))[0] - _kl(c),)
            return (_objective_forward(c, plan, training, kind=pc, coefA=(0, 1, 1, -1, 0))[0],)
        if k == "mvb2":
            if plan.kl_analytic:
                gen_obj, enc_obj, mg, _ = _objective_forward(c, plan, training, kind=pc, coefA=(1, 0, 1, 0, -1),
                                                             coefB=(0, 1, 0, 0, 0), want_means=plan.reduce_mean)
                enc_obj = enc_obj - _kl(c)
                return (mg, enc_obj.mean()) if plan.reduce_mean else (gen_obj, enc_obj)
            gen_obj, enc_obj, mg, me = _objective_forward(c, plan, training, kind=pc, coefA=(1, 0, 1, 0, -1),
                                                          coefB=(0, 1, 1, -1, 0), want_means=plan.reduce_mean)
            return (mg, me) if plan.reduce_mean else (gen_obj, enc_obj)
        raise NotImplementedError(k)

    @staticmethod
    def backward(ctx, *gouts):
        c, dims, plan, enc, dec = ctx.c, ctx.dims, ctx.plan, ctx.enc, ctx.dec
        n_enc = 2 * enc.dims.num_layers
        n_dec = 2 * dec.dims.num_layers
        g_enc_params = [None] * n_enc
        g_dec_params = [None] * n_dec
        k = plan.kind
        nz, S = plan.nz, plan.nz * plan.I
        B, G = c.B, c.G

        def per_sample(g_bg):
            """[B] or [B,G] upstream gradient -> per-sample weights [N] (rows (s, b, g)), divided by nz."""
            g = g_bg.reshape(1, B, -1).float()
            return (g / nz).expand(S, B, G).contiguous().reshape(-1)

        def upstream(g_mean, g_per_b, g_neg):
            """Upstream gradient of the per-datapoint bounds as the kernels take it: (gb [B] or None, gscale 0-d or
            None, factor).  The usual loss is (minus) the batch mean: a device scalar and the host factor +-1/B, no
            tensor math."""
            if g_per_b is None and g_mean is None:
                return None, g_neg.float(), -1.0 / B
            if g_per_b is None and g_neg is None:
                return None, g_mean.float(), 1.0 / B
            g_tot = None if g_mean is None else g_mean.float()
            if g_neg is not None:
                g_tot = -g_neg.float() if g_tot is None else g_tot - g_neg.float()
            if g_per_b is None:
                return None, g_tot, 1.0 / B
            gb = g_per_b.float().reshape(B)
            if g_tot is not None:
                gb = gb + g_tot / B
            return gb.contiguous(), None, 1.0

        if k in ("iwae", "miwae") and any(g is not None for g in gouts):
            gb, gs, f = upstream(*gouts)
            if not c.reparam:
                if c.q_kind == L.Q_STRATIFIED:
                    p = _latent_params(prior=f, qmix=-f, w=f, mix_live=not plan.STL, mix_kind=L.Q_STRATIFIED)
                else:
                    p = _latent_params(prior=f, qown=-f, w=f, own_live=not (plan.STL or plan.DREG), use_zscale=plan.DREG)
                one = _one_launch_backward(c, enc, dec, p, c.wgt, f, 0.0, c.wgt, c.zscale, None, gb, gb, gs, True, complete=True,
                                           dlogit=c.dlogit)
                if one is not None:
                    return (None,) * 10 + tuple(one[1]) + tuple(one[0])
            if not c.reparam:
                two = _paired_backward(c, enc, dec, p, c.wgt, f, 0.0, c.wgt, c.zscale, c.dlogit, None, gb, gs, True)
                if two is not None:
                    return (None,) * 10 + tuple(two[1]) + tuple(two[0])
            pending = []
            # (the decoder's gradients are whole here: one backward pass per step for these objectives)
            dZ, g_dec_params = _dec_side(c, dec, c.wgt, f, 0.0, True, True, overlap=pending, gb=gb, gscale=gs, complete=True)
            if c.reparam:
                wt = c.wgt * (gs * f) if gb is None else (c.wgt.reshape(S, B) * gb.reshape(1, B)).reshape(-1)
                d_enc = _implicit_d_enc(c, dZ, wt.contiguous(), 1.0, -1.0, live=not plan.STL)
                _, g_enc_params = mlp_backward(enc, c.enc_saved, d_enc, c.enc_rows, need_dx=False, need_param_grads=True,
                                               drop_keep=c.enc_keep)
                _join(pending)
                return (None,) * 10 + tuple(g_enc_params) + tuple(g_dec_params)
            if c.q_kind == L.Q_STRATIFIED:
                p = _latent_params(prior=f, qmix=-f, w=f, mix_live=not plan.STL, mix_kind=L.Q_STRATIFIED)
            else:
                p = _latent_params(prior=f, qown=-f, w=f, own_live=not (plan.STL or plan.DREG), use_zscale=plan.DREG)
            g_enc_params = _enc_side(c, enc, p, dZ, c.wgt, c.zscale, c.dlogit, None, gb=gb, gscale=gs)
            _join(pending)
        elif k == "elbo_analytic" and any(g is not None for g in gouts):
            gb, gs, f = upstream(*gouts)
            p = _latent_params(kl=-f)
            one = _one_launch_backward(c, enc, dec, p, None, f / nz, 0.0, None, None, gb, gb, None, gs, True, complete=True)
            if one is not None:
                return (None,) * 10 + tuple(one[1]) + tuple(one[0])
            pending = []
            dZ, g_dec_params = _dec_side(c, dec, None, f / nz, 0.0, True, True, overlap=pending, gb=gb, gscale=gs,
                                         complete=True)
            g_enc_params = _enc_side(c, enc, p, dZ, None, None, None, gb.reshape(B, 1) if gb is not None else None, gscale=gs)
            _join(pending)
        elif k == "cvi" and gouts[0] is not None:
            wt = per_sample(gouts[0])
            p = _latent_params(prior=1.0, qmix=-1.0, mix_live=not plan.STL, mix_kind=L.Q_IMPUTATION)
            one = _one_launch_backward(c, enc, dec, p, wt, 1.0, 0.0, wt, None, None, None, None, None, True)
            if one is not None:
                return (None,) * 10 + tuple(one[1]) + tuple(one[0])
            pending = []
            dZ, g_dec_params = _dec_side(c, dec, wt, 1.0, 0.0, True, True, overlap=pending)
            p = _latent_params(prior=1.0, qmix=-1.0, mix_live=not plan.STL, mix_kind=L.Q_IMPUTATION)
            g_enc_params = _enc_side(c, enc, p, dZ, wt, None, None, None)
            _join(pending)
        elif k == "mvb" and gouts[0] is not None:
            wt = per_sample(gouts[0])
            if plan.kl_analytic:
                one = _one_launch_backward(c, enc, dec, _latent_params(kl=-1.0), wt, 0.0, 1.0, wt, None,
                                           gouts[0].float().contiguous(), None, None, None, True)
            else:
                one = _one_launch_backward(c, enc, dec, _latent_params(prior=1.0, qown=-1.0, own_live=not plan.STL), wt, 0.0, 1.0,
                                           wt, None, None, None, None, None, True)
            if one is not None:
                return (None,) * 10 + tuple(one[1]) + tuple(one[0])
            pending = []
            dZ, g_dec_params = _dec_side(c, dec, wt, 0.0, 1.0, True, True, overlap=pending)
            if plan.kl_analytic:
                p = _latent_params(kl=-1.0)
                g_enc_params = _enc_side(c, enc, p, dZ, wt, None, None, gouts[0].float().contiguous())
            else:
                p = _latent_params(prior=1.0, qown=-1.0, own_live=not plan.STL)
                g_enc_params = _enc_side(c, enc, p, dZ, wt, None, None, None)
            _join(pending)
        elif k == "mvb2":
            # vgiwae/shared/mvb2_manual_optimisation.py:6-8: the generator objective updates the decoder
            # only, the encoder objective the encoder only (through the decoder's input gradient).
            g_gen, g_enc = gouts
            if plan.reduce_mean:
                # scalar gradients of means over (b, k): every sample row weighs 1 / (nz B K), times a device scalar
                f = 1.0 / (nz * B * G)
                if g_gen is not None:
                    _, g_dec_params = _dec_side(c, dec, None, f, 0.0, False, True, gscale=g_gen.float())
                if g_enc is not None:
                    gs = g_enc.float()
                    if plan.kl_analytic:
                        p = _latent_params(kl=-1.0 / (B * G))
                    else:
                        p = _latent_params(prior=f, qown=-f, own_live=not plan.STL)
                    one = _one_launch_backward(c, enc, dec, p, None, 0.0, f, None, None, None, None, None, gs, False)
                    if one is not None:
                        g_enc_params = one[1]
                    else:
                        dZ, _ = _dec_side(c, dec, None, 0.0, f, True, False, gscale=gs)
                        g_enc_params = _enc_side(c, enc, p, dZ, None, None, None, None, gscale=gs)
            else:
                if g_gen is not None:
                    _, g_dec_params = _dec_side(c, dec, per_sample(g_gen), 1.0, 0.0, False, True)
                if g_enc is not None:
                    wt = per_sample(g_enc)
                    if plan.kl_analytic:
                        p, klw = _latent_params(kl=-1.0), g_enc.float().contiguous()
                    else:
                        p, klw = _latent_params(prior=1.0, qown=-1.0, own_live=not plan.STL), None
                    one = _one_launch_backward(c, enc, dec, p, wt, 0.0, 1.0, wt, None, klw, None, None, None, False)
                    if one is not None:
                        g_enc_params = one[1]
                    else:
                        dZ, _ = _dec_side(c, dec, wt, 0.0, 1.0, True, False)
                        g_enc_params = _enc_side(c, enc, p, dZ, wt, None, None, klw)
        return (None,) * 10 + tuple(g_enc_params) + tuple(g_dec_params)


class _DemissTerms(torch.autograd.Function):
    """The five per-sample terms [5, nz, I, B, K] as differentiable outputs; the caller composes its objective from
    them with ordinary tensor operations (tiny next to the networks) and the backward turns the terms' cotangents
    into network gradients through the kernels.  split: two copies, the first carrying gradient to the decoder
    parameters only, the second to the encoder parameters only (through the decoder's input gradient)."""

    @staticmethod
    def forward(ctx, dims, nz, I, q_live, split, grad, enc, dec, Xt, M, eps, enc_keep, dec_keep, *params):
        training = grad and any(ctx.needs_input_grad[13:])
        plan = _Plan("terms", nz, I)
        c = _forward_core(dims, plan, enc, dec, Xt, M, eps, enc_keep, dec_keep, training)
        _objective_forward(c, plan, training, kind=L.TERMS_ONLY, want_terms=True)
        ctx.c, ctx.enc, ctx.dec, ctx.q_live, ctx.split = c, enc, dec, q_live, split
        ctx.set_materialize_grads(False)
        shape = (5, nz, I, c.B, c.G)
        if split:
            return c.terms.clone().view(shape), c.terms.view(shape)
        return (c.terms.view(shape),)

    @staticmethod
    def backward(ctx, *gouts):
        c, enc, dec = ctx.c, ctx.enc, ctx.dec
        lib, st = L.lib(), L.stream_ptr()
        g_enc_params = [None] * (2 * enc.dims.num_layers)
        g_dec_params = [None] * (2 * dec.dims.num_layers)

        def dec_cotangent(g):
            d_dec = torch.empty_like(c.dec_out)
            L.check(lib.dmv_obj_backward_dec_terms(C.byref(c.cfg), L.ptr(c.dec_out), c.dec_out.shape[1], L.ptr(c.Xc), L.ptr(c.Mu),
                                                   L.ptr(g[0]), L.ptr(g[1]), L.ptr(d_dec), d_dec.shape[1], st))
            return d_dec

        def encoder_grads(g, dZ):
            d_enc = torch.empty_like(c.enc_out)
            L.check(lib.dmv_obj_backward_latent_terms(
                C.byref(c.cfg), L.Q_IMPUTATION, int(ctx.q_live), L.ptr(c.enc_out), L.ptr(c.Z), L.ptr(c.eps), c.Z.shape[1],
                L.ptr(dZ), dZ.shape[1], L.ptr(g[2]), L.ptr(g[3]), L.ptr(g[4]), None, L.ptr(c.comp_lp),
                L.ptr(c.terms[4]), L.ptr(d_enc), st))
            return mlp_backward(enc, c.enc_saved, d_enc, c.enc_rows, need_dx=False, need_param_grads=True,
                                drop_keep=c.enc_keep)[1]

        flat = [None if g is None else g.reshape(5, -1).float().contiguous() for g in gouts]
        if ctx.split:
            if flat[0] is not None:
                _, g_dec_params = mlp_backward(dec, c.dec_saved, dec_cotangent(flat[0]), c.N, need_dx=False,
                                               need_param_grads=True, drop_keep=c.dec_keep)
            if flat[1] is not None:
                dZ, _ = mlp_backward(dec, c.dec_saved, dec_cotangent(flat[1]), c.N, need_dx=True, need_param_grads=False,
                                     drop_keep=c.dec_keep)
                g_enc_params = encoder_grads(flat[1], dZ)
        elif flat[0] is not None:
            dZ, g_dec_params = mlp_backward(dec, c.dec_saved, dec_cotangent(flat[0]), c.N, need_dx=True, need_param_grads=True,
                                            drop_keep=c.dec_keep)
            g_enc_params = encoder_grads(flat[0], dZ)
        return (None,) * 13 + tuple(g_enc_params) + tuple(g_dec_params)


def demiss_terms(dims, enc, dec, Xt, M, *, num_latent_samples=1, num_importance_samples=1, q_live=True, split=False, eps=None,
                 enc_keep=None, dec_keep=None):
    """Per-sample terms of imputed batches Xt, M [B, K, D] with latent samples of shape [nz, I, B, K] (eps
    [nz, I, B, K, Zd]): a tensor [5, nz, I, B, K] holding sum_d M log p(x_d|z), sum_d log p(x_d|z), log p(z),
    log q_bk(z) and log (1/K) sum_k' q_bk'(z) (vgiwae/shared/iwae.py:17-45, 66-90, 102-132).  q_live False treats q's
    parameters as constants inside the two q terms ("sticking the landing" / doubly reparameterised estimators).
    Returns (terms,) or, with split, (terms_for_decoder, terms_for_encoder)."""
    assert not dims.q_is_mixture and not dims.q_is_reparam
    return _DemissTerms.apply(dims, num_latent_samples, num_importance_samples, q_live, split, torch.is_grad_enabled(), enc, dec, Xt, M, eps,
                              enc_keep, dec_keep, *enc.params(), *dec.params())


def _run(dims, plan, enc, dec, X, M, eps, enc_keep, dec_keep, comp_idx=None):
    plan = dataclasses.replace(plan, grad=torch.is_grad_enabled())
    return _FusedObjective.apply(dims, plan, enc, dec, X, M, eps, enc_keep, dec_keep, comp_idx, *enc.params(), *dec.params())


# ------------------------------------------------------------------------------------------ public API
def _pick(outs, with_loss):
    """(mean bound, per-datapoint bounds) as the reference's estimators return them; with_loss adds minus the mean, the
    loss the training steps minimise, formed by the kernel that forms the mean (differentiable like the mean)."""
    mean, per_b, neg = outs
    return (mean, per_b, neg) if with_loss else (mean, per_b)


def estimate_iwelbo(dims: ModelDims, enc: PackedNet, dec: PackedNet, X, M, *, num_latent_samples=1,
                    num_importance_samples=1, STL=False, DREG=False, eps=None, enc_keep=None, dec_keep=None, comp_idx=None,
                    with_loss=False):
    """IWAE bound averaged over the batch, and the per-datapoint bounds.  eps [nz, I, B, Zd]; for a reparametrised
    mixture encoder eps [nz, I, B, C, Zd] and comp_idx [nz, I, B] (drawn on the device when omitted)."""
    assert not (STL and DREG), "Cannot use STL and DREG gradients at the same time."
    assert not dims.q_is_mixture
    plan = _Plan("iwae", num_latent_samples, num_importance_samples, STL=STL, DREG=DREG)
    return _pick(_run(dims, plan, enc, dec, X, M, eps, enc_keep, dec_keep, comp_idx), with_loss)


def estimate_elbo(dims, enc, dec, X, M, *, num_latent_samples=1, kl_analytic=True, STL=False, eps=None,
                  enc_keep=None, dec_keep=None, comp_idx=None, with_loss=False):
    """VAE ELBO with analytic or Monte-Carlo KL.  eps [nz, B, Zd] (reparametrised mixture: [nz, B, C, Zd] and
    comp_idx [nz, B])."""
    assert not (STL and kl_analytic), '"Sticking the Landing" cannot be used with analytical KL.'
    assert not dims.q_is_mixture
    if kl_analytic:
        plan = _Plan("elbo_analytic", num_latent_samples, 1)
    else:
        plan = _Plan("iwae", num_latent_samples, 1, STL=STL)   # logsumexp over a single sample = identity
    return _pick(_run(dims, plan, enc, dec, X, M, eps, enc_keep, dec_keep, comp_idx), with_loss)


def estimate_multiple_iwelbo(dims, enc, dec, X, M, *, num_latent_samples=1, num_importance_samples_for_each_component=1,
                             STL=False, looser=False, eps=None, enc_keep=None, dec_keep=None, with_loss=False):
    """MissSIWAE bound (stratified mixture encoder).  eps [nz, I, B, C, Zd]."""
    assert dims.q_is_mixture
    plan = _Plan("miwae", num_latent_samples, num_importance_samples_for_each_component, STL=STL, looser=looser)
    return _pick(_run(dims, plan, enc, dec, X, M, eps, enc_keep, dec_keep), with_loss)


def estimate_multiple_elbo(dims, enc, dec, X, M, *, num_latent_samples=1, STL=False, eps=None, enc_keep=None,
                           dec_keep=None, with_loss=False):
    """MissSVAE stratified ELBO = sum_c pi_c * ratio_c.  eps [nz, B, C, Zd]."""
    assert dims.q_is_mixture
    plan = _Plan("miwae", num_latent_samples, 1, STL=STL, looser=True)
    return _pick(_run(dims, plan, enc, dec, X, M, eps, enc_keep, dec_keep), with_loss)


def mvb2_objectives(dims, enc, dec, Xt, M, *, num_latent_samples=1, STL=False, kl_analytic=False, eps=None,
                    enc_keep=None, dec_keep=None, reduce_mean=False):
    """DeMissVAE: (generator objective, encoder objective), each [B, K].  Xt, M [B,K,D]; eps [nz,B,K,Zd].
    Gradients: the first output reaches the decoder parameters only, the second the encoder parameters only.
    reduce_mean: the two means over (b, k) instead (what the training step needs; its backward then forms the
    per-sample weights in one launch instead of three)."""
    assert not dims.q_is_mixture
    plan = _Plan("mvb2", num_latent_samples, 1, STL=STL, kl_analytic=kl_analytic, reduce_mean=reduce_mean)
    return _run(dims, plan, enc, dec, Xt, M, eps, enc_keep, dec_keep)


def cvi_objective(dims, enc, dec, Xt, M, *, num_latent_samples=1, STL=False, eps=None, enc_keep=None, dec_keep=None):
    assert not dims.q_is_mixture
    plan = _Plan("cvi", num_latent_samples, 1, STL=STL)
    return _run(dims, plan, enc, dec, Xt, M, eps, enc_keep, dec_keep)[0]


def mvb_objective(dims, enc, dec, Xt, M, *, num_latent_samples=1, STL=False, kl_analytic=False, eps=None,
                  enc_keep=None, dec_keep=None):
    assert not dims.q_is_mixture
    plan = _Plan("mvb", num_latent_samples, 1, STL=STL, kl_analytic=kl_analytic)
    return _run(dims, plan, enc, dec, Xt, M, eps, enc_keep, dec_keep)[0]
