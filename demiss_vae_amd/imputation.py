"""DeMissVAE imputation update on the device.

  ImputationBuffer      device-resident twin of the reference's ImputationDistribution
                        (vgiwae/utils/imputation_distribution.py:18-187): data_imp [(N K), D] in HBM,
                        idx_imp [N, K] = n*K + k, gather / scatter by dataset index through libdmv.
  pseudo_gibbs          vgiwae/shared/vae_imputation.py:26-82
  lair                  vgiwae/shared/vae_imputation.py:177-201, 390-696 (dmis weights, global
                        multinomial resampling, prior-replenish proposals, resample_final_imps)

Every sampler runs the encoder / decoder through the fused MLP kernels without saving activations
(the reference runs them under inference_mode) and finishes with one fused tail kernel that draws
x_m, clamps, applies the non-finite fallback, blends with the observed entries and (optionally)
writes the rows straight back into the buffer.  Noise is explicit (parity) or Philox (default).
"""
import ctypes as C
import math
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from . import rng
from .mlp import PackedNet, GeneratedDropout, mlp_forward, make_keep_masks, pad32
from .objectives import ModelDims, _u8


def softplus_inverse(x: torch.Tensor) -> torch.Tensor:
    """vgiwae/utils/softplus_inverse.py:3-7"""
    return x + torch.log(-torch.expm1(-x))


# ------------------------------------------------------------------------------------------ buffer
class ImputationBuffer:
    """K imputations per datapoint, kept on the device for the whole run (the reference keeps them on the
    host and copies every step, vgiwae/utils/imputation_distribution.py:160-163, 180)."""

    def __init__(self, data: torch.Tensor, num_imputations: int, device):
        data = torch.as_tensor(data, dtype=torch.float32)
        self.N, self.D = data.shape
        self.K = int(num_imputations)
        self.device = torch.device(device)
        # 'n d -> (n k) d' and '(n k) -> n k'   (imputation_distribution.py:77-79)
        self.data_imp = data.to(self.device).repeat_interleave(self.K, dim=0).contiguous()
        self.idx_imp = torch.arange(self.N * self.K, device=self.device, dtype=torch.int64).reshape(self.N, self.K)

    def gather(self, index: torch.Tensor, sampled_i: Optional[torch.Tensor] = None):
        """Rows of `index` [B] -> (X_imp [B, K', D], idx [B, K']).  sampled_i [B, K'] picks a subset of the K
        imputations (the reference draws it with torch.multinomial without replacement, :146-151)."""
        index = index.to(self.device, torch.int64)
        idx = self.idx_imp[index]
        if sampled_i is not None:
            idx = torch.gather(idx, 1, sampled_i.to(self.device, torch.int64))
        idx = idx.contiguous()
        B, Kp = idx.shape
        out = torch.empty((B, Kp, self.D), dtype=torch.float32, device=self.device)
        L.check(L.lib().dmv_rows_gather(L.ptr(self.data_imp), L.ptr(idx), B * Kp, self.D, L.ptr(out), L.stream_ptr()))
        return out, idx

    def scatter(self, idx: torch.Tensor, X_imp: torch.Tensor):
        idx = idx.to(self.device, torch.int64).contiguous()
        X = X_imp.detach().to(self.device, torch.float32).contiguous()
        L.check(L.lib().dmv_rows_scatter(L.ptr(self.data_imp), L.ptr(idx), idx.numel(), self.D, L.ptr(X), L.stream_ptr()))

    def sample_subset(self, batch_size: int, num_imputations: int, generator=None):
        """num_imputations of the K slots per datapoint, without replacement."""
        w = torch.ones((batch_size, self.K), device=self.device)
        return torch.multinomial(w, num_samples=num_imputations, replacement=False, generator=generator)

    # ---- initial imputations (imputation_distribution.py:274-386); host-side, once per run
    def init_empirical(self, data: np.ndarray, mask: np.ndarray):
        """Missing entries <- draws from the observed values of the same dimension, with numpy's global RNG in
        increasing dimension order exactly as the reference does (:331-338)."""
        data = np.asarray(data)
        mask = np.asarray(mask).astype(bool)
        imp = self.data_imp.cpu()
        mis_imp_not = torch.from_numpy(~np.repeat(mask, self.K, axis=0))
        count = mis_imp_not.sum(0)
        for i in range(data.shape[-1]):
            if count[i] == 0:
                continue
            samples = np.random.choice(data[:, i][mask[:, i]], size=int(count[i]))
            imp[mis_imp_not[:, i], i] = torch.as_tensor(samples, dtype=torch.float32)
        self.data_imp.copy_(imp)

    def init_zero(self, mask):
        m = torch.as_tensor(np.asarray(mask)).to(self.device).repeat_interleave(self.K, dim=0)
        self.data_imp.mul_(m.float())

    def init_mean(self, data, mask):
        data, mask = np.asarray(data, dtype=np.float32), np.asarray(mask).astype(bool)
        means = (data * mask).sum(0) / mask.sum(0)
        m = torch.as_tensor(mask).to(self.device).repeat_interleave(self.K, dim=0)
        self.data_imp.copy_(torch.where(m, self.data_imp, torch.as_tensor(means, device=self.device).expand_as(self.data_imp)))


# ------------------------------------------------------------------------------------------ shared pieces
def _encode_rows(dims: ModelDims, enc, X2, M2, enc_keep):
    if not isinstance(enc, PackedNet):      # caller-supplied torch module (convolutional trunks)
        return enc(X2, M2.bool()).reshape(X2.shape[0], -1).contiguous().float()
    in_mode = L.IN_CONCAT if dims.encoder_use_mis_mask else L.IN_PLAIN
    y, _ = mlp_forward(enc, X2, dims.D, X2.shape[0], in_mode=in_mode, D=dims.D, m=M2, ldm=dims.D, save=False,
                       drop_keep=enc_keep)
    return y


def _sample_cfg(dims: ModelDims, rows: int, I: int, ld_enc: int):
    """Rows (i, r) with r indexing encoder rows: z[i, r] ~ q(. | enc row r)."""
    cfg = L.DmvObjCfg()
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = rows, 1, 1, I, dims.D, dims.Zd
    cfg.gen_dist, cfg.q_eps, cfg.x_per_component = L.GEN[dims.gen_dist], dims.q_eps, 0
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ld_enc, 0, dims.Zd, 0
    return cfg


def _one_launch_pass(dims: ModelDims, enc, dec, X2, M2, rows: int, I: int, eps2, training: bool):
    """Encoder on `rows` rows, I latent draws per row and the decoder on them in one launch when the library takes the
    shapes (objectives.fused_forward), no gradient bookkeeping: (enc_out, Z, cfg, dec_out) or None."""
    from .objectives import fused_forward
    if not (isinstance(enc, PackedNet) and isinstance(dec, PackedNet)):
        return None
    if training and ((enc.dims.dropout > 0 and enc.dims.nb > 0) or (dec.dims.dropout > 0 and dec.dims.nb > 0)):
        return None             # training-mode dropout inside the sampler's passes: separate launches draw the masks
    cfg = _sample_cfg(dims, rows, I, pad32(enc.dims.out_dim))
    in_mode = L.IN_CONCAT if dims.encoder_use_mis_mask else L.IN_PLAIN
    out = fused_forward(cfg, enc, dec, in_mode, X2, M2, eps2, False, False)
    if out is None:
        return None
    enc_out, _, Z, _, dec_out, _ = out
    return enc_out, Z, cfg, dec_out


def _tail(dims, dec_out, row_map, X_old, M, noise, clip_min, clip_max, clip_after_blend, buf, idx_imp):
    N, D = X_old.shape[0] * X_old.shape[1], dims.D
    X_out = torch.empty_like(X_old)
    seed = off = 0
    if noise is None:
        seed, off = rng.next_offset(N * D)
    else:
        noise = noise.reshape(N, D).contiguous().float()
    L.check(L.lib().dmv_impute_tail(N, D, L.GEN[dims.gen_dist], L.ptr(dec_out), dec_out.shape[1], L.ptr(row_map),
                                    L.ptr(X_old), L.ptr(M), L.ptr(noise), seed, off, rng.base_ptr(), L.ptr(clip_min), L.ptr(clip_max),
                                    1 if clip_after_blend else 0, L.ptr(X_out),
                                    L.ptr(buf.data_imp) if buf is not None else None, L.ptr(idx_imp), L.stream_ptr()))
    return X_out


def _keep(net, rows, training, device):
    """what a training-mode call of `net` on `rows` rows passes as drop_keep: for relu networks the kernels draw the
    keep decisions themselves (GeneratedDropout), else a mask tensor from the same Philox stream"""
    if not isinstance(net, PackedNet) or not training or net.dims.dropout <= 0 or net.dims.nb == 0:
        return None
    if GeneratedDropout.supported(net):
        return GeneratedDropout(net, rows)
    n = net.dims.nb * rows * net.dims.hidden
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    seed, off = rng.next_offset(n)
    L.check(L.lib().dmv_dropout_masks(L.ptr(keep), n, net.dims.dropout, seed, off, rng.base_ptr(), L.stream_ptr()))
    return keep


# ------------------------------------------------------------------------------------------ pseudo-Gibbs
# all iterations of the sampler as one launch (dmv_pgibbs_update) where its limits allow
ONE_LAUNCH_PGIBBS = True


@torch.no_grad()
def pseudo_gibbs(dims: ModelDims, enc: PackedNet, dec: PackedNet, X, M, *, num_iterations=1, clip_min=None,
                 clip_max=None, eps_z=None, noise_x=None, training=True, buf: Optional[ImputationBuffer] = None,
                 idx_imp=None):
    """X [B,K,D] current imputations, M [B,K,D] (True = observed).  eps_z [T,B,K,Zd], noise_x [T,B,K,D] optional.
    With `buf`/`idx_imp` the last iteration writes its rows straight into the imputation buffer."""
    B, K, D = X.shape
    dev = X.device
    Xc = X.contiguous().float()
    Mu = _u8(M)
    st = L.stream_ptr()
    lib = L.lib()
    cmin = clip_min.to(dev).float().contiguous() if clip_min is not None else None
    cmax = clip_max.to(dev).float().contiguous() if clip_max is not None else None
    N = B * K
    T = num_iterations
    if (ONE_LAUNCH_PGIBBS and isinstance(enc, PackedNet) and isinstance(dec, PackedNet) and T >= 1 and
            not (training and ((enc.dims.dropout > 0 and enc.dims.nb > 0) or (dec.dims.dropout > 0 and dec.dims.nb > 0))) and
            lib.dmv_pgibbs_update_fits(C.byref(enc.c), C.byref(dec.c), N, T, D, dims.Zd)):
        # every iteration in one launch: a workgroup keeps 16 imputation rows and walks them through all T passes
        Xw = Xc.clone().view(N, D)
        Mw = Mu.reshape(N, D).contiguous()
        e_all = (torch.stack([e.reshape(N, dims.Zd) for e in eps_z[:T]]).contiguous().float().to(dev)
                 if eps_z is not None else None)
        n_all = (torch.stack([x.reshape(N, D) for x in noise_x[:T]]).contiguous().float().to(dev)
                 if noise_x is not None else None)
        off_e, off_t = (C.c_uint64 * T)(), (C.c_uint64 * (T + 1))()
        seed = 0
        for t in range(T):                                  # the ranges of the separate launches, in their order
            if e_all is None:
                seed, off_e[t] = rng.next_offset(N * dims.Zd)
            if n_all is None:
                seed, off_t[t] = rng.next_offset(N * D)
        in_mode = L.IN_CONCAT if dims.encoder_use_mis_mask else L.IN_PLAIN
        L.check(lib.dmv_pgibbs_update(C.byref(enc.c), L.ptr(enc.packed()), in_mode, C.byref(dec.c), L.ptr(dec.packed()), N, T, D,
                                      dims.Zd, L.GEN[dims.gen_dist], 1 if dims.q_eps else 0, L.ptr(Xw), L.ptr(Mw), L.ptr(e_all),
                                      L.ptr(n_all), seed, off_e, off_t, rng.base_ptr(), L.ptr(cmin), L.ptr(cmax),
                                      L.ptr(buf.data_imp) if buf is not None else None,
                                      L.ptr(idx_imp.reshape(-1).contiguous()) if (buf is not None and idx_imp is not None) else None,
                                      st))
        return Xw.view(B, K, D)
    for t in range(num_iterations):
        e = eps_z[t].reshape(N, dims.Zd).contiguous().float() if eps_z is not None else None
        one = _one_launch_pass(dims, enc, dec, Xc.reshape(N, D), Mu.reshape(N, D), N, 1, e, training)
        if one is None:
            enc_out = _encode_rows(dims, enc, Xc.reshape(N, D), Mu.reshape(N, D), _keep(enc, N, training, dev))
            cfg = _sample_cfg(dims, N, 1, enc_out.shape[1])
            Z = torch.empty((N, dims.Zd), dtype=torch.float32, device=dev)
            if e is not None:
                L.check(lib.dmv_latent_sample(C.byref(cfg), L.ptr(enc_out), L.ptr(e), 0, 0, None, dims.Zd, None, L.ptr(Z), None, st))
            else:
                seed, off = rng.next_offset(N * dims.Zd)
                L.check(lib.dmv_latent_sample(C.byref(cfg), L.ptr(enc_out), None, seed, off, rng.base_ptr(), dims.Zd, None, L.ptr(Z), None, st))
            if isinstance(dec, PackedNet):
                dec_out, _ = mlp_forward(dec, Z, dims.Zd, N, save=False, drop_keep=_keep(dec, N, training, dev))
            else:
                dec_out = dec(Z).reshape(N, -1).contiguous().float()
        else:
            dec_out = one[3]
        last = t == num_iterations - 1
        Xc = _tail(dims, dec_out, None, Xc, Mu, noise_x[t] if noise_x is not None else None, cmin, cmax, False,
                   buf if last else None, idx_imp.reshape(-1).contiguous() if (last and idx_imp is not None) else None)
    return Xc


# ------------------------------------------------------------------------------------------ LAIR
# LAIR's whole update as one launch (dmv_lair_update) where its limits allow
ONE_LAUNCH_LAIR = True
# a dict here makes lair() leave every iteration's log-weights in it ("log_w": [T, I B (K + r)]); tests only
LAIR_DEBUG = None
# LAIR's resampling draws and sampling tail as one launch (dmv_lair_resample_tail) where its limits allow
MERGE_RESAMPLE_TAIL = True


@torch.no_grad()
def lair(dims: ModelDims, enc: PackedNet, dec: PackedNet, X, M, *, num_iterations=1, num_imp_samples=1,
         num_prior_replenish_proposals=0, resample_final_imps=False, clip_min=None, clip_max=None,
         eps_z=None, idx=None, noise_x=None, training=True, buf: Optional[ImputationBuffer] = None, idx_imp=None):
    """Latent-adaptive importance resampling with DMIS weights and global multinomial resampling.

    X [B,K,D], M [B,K,D].  Parity inputs (all optional): eps_z [T][I,B,K+r,Zd], idx [T (+1)][K+r, B] categorical
    draws (proposal index i*(K+r)+k; the extra entry is the final resampling's [K, B]), noise_x [T (+1)][B,K,D]."""
    assert not dims.q_is_mixture
    if resample_final_imps:
        assert num_iterations > 1, "Need >1 iterations to resample final imps"
    B, K, D = X.shape
    r, I, Zd, T = num_prior_replenish_proposals, num_imp_samples, dims.Zd, num_iterations
    Kp = K + r
    N = I * B * Kp
    dev = X.device
    lib, st = L.lib(), L.stream_ptr()
    Xc = X.contiguous().float()
    Mu = _u8(M)
    cmin = clip_min.to(dev).float().contiguous() if clip_min is not None else None
    cmax = clip_max.to(dev).float().contiguous() if clip_max is not None else None
    clip = (cmin is not None or cmax is not None)
    # The log-weights sum log p(x_d | z) over OBSERVED entries only, under the mask of a datapoint's first imputation
    # broadcast over the others (vgiwae/shared/vae_imputation.py:407, "M is a repeated tensor"): the observed values are
    # the same in all of a datapoint's imputations, so one [B, D] pair serves every component, prior proposals included
    # (the reference builds a [B, K + r, D] copy for them, :439-441).  Clamping after the blend (:545-561) also clamps
    # observed values that lie outside the range, so with clamping the pair is taken afresh from the current imputations.
    X0, M0 = Xc[:, 0].contiguous(), Mu[:, 0].contiguous()
    lw_dbg = [] if LAIR_DEBUG is not None else None
    ldE = pad32(enc.dims.out_dim) if isinstance(enc, PackedNet) else None
    # prior components: mu = 0, sigma = 1 (:419-441), as rows of the encoder-output array [B, K + r, ld]; constant
    prior_raw = float(softplus_inverse(torch.tensor(1.0 - (1e-5 if dims.q_eps else 0.0), dtype=torch.float32)))
    enc_buf = None

    def enc_buffer(ld):
        buf_ = torch.empty((B, Kp, ld), dtype=torch.float32, device=dev)
        if r > 0:
            buf_[:, K:] = 0.0
            buf_[:, K:, Zd:2 * Zd] = prior_raw
        return buf_

    in_mode = L.IN_CONCAT if dims.encoder_use_mis_mask else L.IN_PLAIN
    # ---- the whole update in one launch where the shapes allow (one importance sample per component, relu networks of
    # equal width, K + r <= 16, no dropout to draw): a workgroup walks its datapoints through all iterations with
    # everything but the imputations in LDS
    if (ONE_LAUNCH_LAIR and I == 1 and isinstance(enc, PackedNet) and isinstance(dec, PackedNet) and
            not (training and ((enc.dims.dropout > 0 and enc.dims.nb > 0) or (dec.dims.dropout > 0 and dec.dims.nb > 0))) and
            lib.dmv_lair_update_fits(C.byref(enc.c), C.byref(dec.c), B, K, r, T, D, Zd, 1 if resample_final_imps else 0)):
        Xw = Xc.clone().view(B * K, D)
        Mw = Mu.reshape(B * K, D).contiguous()
        lw_dbg = torch.empty((T, B * Kp), dtype=torch.float32, device=dev) if LAIR_DEBUG is not None else None
        e_all = (torch.stack([e.reshape(B * Kp, Zd) for e in eps_z[:T]]).contiguous().float().to(dev)
                 if eps_z is not None else None)
        d_all = f_draw = None
        if idx is not None:
            d_all = torch.stack([idx[t].to(dev, torch.int64).t() for t in range(T)]).contiguous()          # [T, B, K+r]
            if resample_final_imps:
                f_draw = idx[T].to(dev, torch.int64).t().contiguous()                                     # [B, K]
        nT = T + (1 if resample_final_imps else 0)
        n_all = (torch.stack([x.reshape(B * K, D) for x in noise_x[:nT]]).contiguous().float().to(dev)
                 if noise_x is not None else None)
        # the Philox ranges of the separate launches, in their order: per iteration proposals, draws, tail; then the final
        # draws and tail
        off_e, off_d, off_t = (C.c_uint64 * T)(), (C.c_uint64 * (T + 1))(), (C.c_uint64 * (T + 1))()
        seed = 0
        for t in range(T):
            if e_all is None:
                seed, off_e[t] = rng.next_offset(B * Kp * Zd)
            if d_all is None:
                seed, off_d[t] = rng.next_offset(B * Kp)
            if n_all is None:
                seed, off_t[t] = rng.next_offset(B * K * D)
        if resample_final_imps:
            if f_draw is None:
                seed, off_d[T] = rng.next_offset(B * K)
            if n_all is None:
                seed, off_t[T] = rng.next_offset(B * K * D)
        L.check(lib.dmv_lair_update(C.byref(enc.c), L.ptr(enc.packed()), in_mode, C.byref(dec.c), L.ptr(dec.packed()),
                                    B, K, r, T, D, Zd, L.GEN[dims.gen_dist], 1 if dims.q_eps else 0, prior_raw,
                                    1 if resample_final_imps else 0, L.ptr(Xw), L.ptr(Mw), L.ptr(e_all), L.ptr(d_all),
                                    L.ptr(f_draw), L.ptr(n_all), seed, off_e, off_d, off_t, rng.base_ptr(),
                                    L.ptr(cmin) if clip else None, L.ptr(cmax) if clip else None,
                                    L.ptr(buf.data_imp) if buf is not None else None,
                                    L.ptr(idx_imp.reshape(-1).contiguous()) if (buf is not None and idx_imp is not None) else None,
                                    L.ptr(lw_dbg), st))
        if LAIR_DEBUG is not None:
            LAIR_DEBUG["log_w"], LAIR_DEBUG["one_launch"] = lw_dbg, True
        return Xw.view(B, K, D)

    # all iterations' proposals and log-weights in place, for the final resampling: [T][(i b k)]
    Z_all = torch.empty((T, N, Zd), dtype=torch.float32, device=dev) if resample_final_imps else None
    lw_all = torch.empty((T, N), dtype=torch.float32, device=dev) if resample_final_imps else None
    cfg = L.DmvObjCfg()
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, Kp, 1, I, D, Zd
    cfg.gen_dist, cfg.q_eps = L.GEN[dims.gen_dist], dims.q_eps
    from .objectives import fused_forward
    one_launch = (isinstance(enc, PackedNet) and isinstance(dec, PackedNet) and
                  not (training and ((enc.dims.dropout > 0 and enc.dims.nb > 0) or (dec.dims.dropout > 0 and dec.dims.nb > 0))))
    for t in range(T):
        if clip and t > 0:
            X0 = Xc[:, 0].contiguous()
        e = eps_z[t].reshape(N, Zd).contiguous().float() if eps_z is not None else None
        Zt = Z_all[t] if resample_final_imps else None
        out = None
        if one_launch:
            # encoder on the B K imputations, proposals z[i, b, k] ~ q(. | component (b, k)) (prior components included)
            # and the decoder on them: one launch
            if enc_buf is None:
                enc_buf = enc_buffer(ldE)
            cfg.x_per_component = 1
            cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = Kp * ldE, ldE, Zd, 0
            out = fused_forward(cfg, enc, dec, in_mode, Xc.reshape(B * K, D), Mu.reshape(B * K, D), e, False, False,
                                n_prior=r, prior_raw=prior_raw, enc_out=enc_buf.view(B * Kp, ldE), Z=Zt)
        if out is not None:
            enc_out, Z, dec_out = out[0], out[2], out[4]
        else:
            rows = _encode_rows(dims, enc, Xc.reshape(B * K, D), Mu.reshape(B * K, D), _keep(enc, B * K, training, dev))
            if enc_buf is None or enc_buf.shape[2] != rows.shape[1]:
                enc_buf = enc_buffer(rows.shape[1])
            ldE = rows.shape[1]
            enc_buf[:, :K] = rows.view(B, K, ldE)
            enc_out = enc_buf.view(B * Kp, ldE)
            scfg = _sample_cfg(dims, B * Kp, I, ldE)
            Z = Zt if Zt is not None else torch.empty((N, Zd), dtype=torch.float32, device=dev)
            if e is not None:
                L.check(lib.dmv_latent_sample(C.byref(scfg), L.ptr(enc_out), L.ptr(e), 0, 0, None, Zd, None, L.ptr(Z), None, st))
            else:
                seed, off = rng.next_offset(N * Zd)
                L.check(lib.dmv_latent_sample(C.byref(scfg), L.ptr(enc_out), None, seed, off, rng.base_ptr(), Zd, None, L.ptr(Z), None, st))
            dec_out, _ = mlp_forward(dec, Z, Zd, N, save=False, drop_keep=_keep(dec, N, training, dev))
        # DMIS log-weights: uniform mixture over the K+r components of the datapoint (shared/iwae.py:57-100):
        # log_w = sum_d M log p(x_d|z) + log p(z) - log q_mix(z), rows (i b k): a by-product of the fused objective forward
        cfg.x_per_component = 0
        cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = Kp * ldE, ldE, Zd, 0
        log_w = lw_all[t] if resample_final_imps else torch.empty(N, dtype=torch.float32, device=dev)
        per_b = torch.empty(B, dtype=torch.float32, device=dev)
        f = L.DmvObjFwd()
        f.q_kind, f.mix_C, f.kind = L.Q_IMPUTATION, Kp, L.BOUND_LSE
        f.log_w, f.bound = L.addr(log_w), L.addr(per_b)
        rc = lib.dmv_obj_forward(C.byref(cfg), C.byref(f), L.ptr(dec_out), dec_out.shape[1], L.ptr(X0), L.ptr(M0),
                                 L.ptr(Z), Zd, L.ptr(enc_out), st)
        if rc == L.E_UNSUPPORTED and I * Kp > 1024:
            # more proposals per datapoint than the one-launch kernel stages in LDS: terms, then the looping bound kernel
            terms = torch.empty((5, N), dtype=torch.float32, device=dev)
            L.check(lib.dmv_obj_terms(C.byref(cfg), L.Q_IMPUTATION, Kp, L.ptr(dec_out), dec_out.shape[1], L.ptr(X0), L.ptr(M0),
                                      L.ptr(Z), Zd, L.ptr(enc_out), L.ptr(terms), None, st))
            L.check(lib.dmv_obj_bound(C.byref(cfg), L.BOUND_LSE, L.Q_IMPUTATION, 0, 0, L.ptr(enc_out), L.ptr(terms),
                                      L.ptr(log_w), L.ptr(per_b), None, None, None, None, st))
        else:
            L.check(rc)
        # resampling: K + r draws per datapoint from Categorical(logits = log_w[b, (i k)]) (:181-186), of which the first
        # K are kept (:521-524); row_map = rows of the chosen proposals in the (i, b, k)-ordered decoder outputs
        if lw_dbg is not None:
            lw_dbg.append(log_w.clone())
        last = (t == T - 1) and not resample_final_imps
        nz_t = noise_x[t] if noise_x is not None else None
        buf_t = buf if last else None
        idx_t = idx_imp.reshape(-1).contiguous() if (last and idx_imp is not None) else None
        if idx is None and MERGE_RESAMPLE_TAIL and I * Kp <= 1024 and Kp <= 64:
            # draws and sampling tail in one launch; the Philox ranges are taken in the order of the two launches below,
            # so both ways produce the same imputations
            seed, off = rng.next_offset(B * Kp)
            X_new = torch.empty_like(Xc)
            seed_t = off_t = 0
            if nz_t is None:
                seed_t, off_t = rng.next_offset(B * K * D)
            else:
                nz_t = nz_t.reshape(B * K, D).contiguous().float()
            L.check(lib.dmv_lair_resample_tail(B, I, Kp, Kp, K, L.ptr(log_w), None, seed, off, D, L.GEN[dims.gen_dist],
                                               L.ptr(dec_out), dec_out.shape[1], L.ptr(Xc), L.ptr(Mu), L.ptr(nz_t), seed_t, off_t,
                                               rng.base_ptr(), L.ptr(cmin) if clip else None, L.ptr(cmax) if clip else None,
                                               L.ptr(X_new), L.ptr(buf_t.data_imp) if buf_t is not None else None, L.ptr(idx_t),
                                               None, st))
            Xc = X_new
            continue
        row_map = torch.empty(B * K, dtype=torch.int64, device=dev)
        if idx is not None:
            draw = idx[t].to(dev, torch.int64).t().contiguous()                 # [B, K+r]
            L.check(lib.dmv_lair_row_map(B, K, Kp, L.ptr(draw), L.ptr(row_map), st))
        else:
            seed, off = rng.next_offset(B * Kp)
            L.check(lib.dmv_categorical_draw_grouped(B, I, Kp, Kp, L.ptr(log_w), None, seed, off, rng.base_ptr(), None,
                                                     L.ptr(row_map), K, st))
        Xc = _tail(dims, dec_out, row_map, Xc, Mu, nz_t, cmin if clip else None, cmax if clip else None, True, buf_t, idx_t)
    if lw_dbg is not None:
        LAIR_DEBUG["log_w"] = torch.stack(lw_dbg)
    if resample_final_imps:
        # K latents per datapoint from the pooled T I (K+r) proposals of all iterations (:671-694), decoded once more
        rows_f = torch.empty(B * K, dtype=torch.int64, device=dev)
        if idx is not None:
            v = idx[T].to(dev, torch.int64).t().contiguous()                    # [B, K] proposal numbers (t i k)
            bcol = torch.arange(B, device=dev, dtype=torch.int64)[:, None]
            rows_f = (((v // Kp) * B + bcol) * Kp + v % Kp).reshape(-1).contiguous()
        else:
            seed, off = rng.next_offset(B * K)
            L.check(lib.dmv_categorical_draw_grouped(B, T * I, Kp, K, L.ptr(lw_all), None, seed, off, rng.base_ptr(), None,
                                                     L.ptr(rows_f), K, st))
        Zf = torch.empty((B * K, Zd), dtype=torch.float32, device=dev)
        L.check(lib.dmv_rows_gather(L.ptr(Z_all), L.ptr(rows_f), B * K, Zd, L.ptr(Zf), st))
        dec_out, _ = mlp_forward(dec, Zf, Zd, B * K, save=False, drop_keep=_keep(dec, B * K, training, dev))
        Xc = _tail(dims, dec_out, None, Xc, Mu, noise_x[T] if noise_x is not None else None, None, None, False,
                   buf, idx_imp.reshape(-1).contiguous() if idx_imp is not None else None)
    return Xc


# ------------------------------------------------------------------------------------------ log-likelihood
@torch.no_grad()
def estimate_loglik(dims: ModelDims, enc: PackedNet, dec: PackedNet, X, M, *, num_importance_samples, chunk=4096,
                    eps=None, comp_idx=None, k_shard=None, k_range=None, return_state=False, group=None):
    """K-sample importance-sampling estimate of log p(x_o) per datapoint
    (vgiwae/test/refit_encoder_using_iwae_and_estimate_loglik.py:189-222), streamed over K in chunks with an
    online (max, sum-exp) so that no [K, B, .] tensor is materialised.

    Mixture encoders are sampled ancestrally: a stratified one thereby "without stratification"
    (vgiwae/models/vae.py:298-301), a reparametrised one as its own rsample does in the forward direction
    (vgiwae/shared/reparametrized_mixture_same_family.py:27-58); both share one parameter layout.  The component of
    each draw comes from comp_idx [K, B] (parity) or from device categorical draws.  eps [K, B, Zd] optional.

    Sharding K over GPUs: k_shard = (rank, world) makes this rank fold only its contiguous share of the K samples and
    combines the ranks' (max, sum-exp) states with two all-reduces of B floats (parallel.sharded_loglik); every rank
    returns the full-K estimate.  Noise is addressed by the global sample number (explicit eps / comp_idx are sliced;
    the Philox counter of latent element (k, b, j) is base + (k B + b) Zd + j whatever the chunking or the sharding),
    so for a single-Gaussian encoder the sharded estimate equals the one-rank estimate up to summation order when the
    ranks share the Philox seed.  k_range = (lo, hi) with return_state=True returns the raw state of those samples."""
    from . import parallel as P
    B, D = X.shape
    K = num_importance_samples
    dev = X.device
    lib, st = L.lib(), L.stream_ptr()
    Xc, Mu = X.contiguous().float(), _u8(M)
    Zd = dims.Zd
    in_mode = L.IN_CONCAT_MASKED if dims.encoder_use_mis_mask else L.IN_MASKED
    enc_out, _ = mlp_forward(enc, Xc, D, B, in_mode=in_mode, D=D, m=Mu, ldm=D, save=False)
    ldE = enc_out.shape[1]
    mixture = dims.q_is_mixture or dims.q_is_reparam
    if mixture:
        Cn = dims.C
        logits = enc_out[:, 2 * Cn * Zd: 2 * Cn * Zd + Cn].contiguous()
    if chunk * B >= 32768:
        # whole rounds of the decoder kernel: 1024 one-wave tiles of 32 rows fill the chip once, and a launch that ends
        # a quarter into a round (4096 x 50 rows = 6.25 rounds) pays for the whole round
        chunk = max(1, (chunk * B // 32768) * 32768 // B)
    # counters for all K samples are reserved at once (every rank reserves the same ranges), a chunk uses its slice
    z_seed, z_off = rng.next_offset(K * B * Zd) if eps is None else (0, 0)
    c_seed, c_off = rng.next_offset(K * B) if (mixture and comp_idx is None) else (0, 0)

    def fold(lo, hi):
        run_max = torch.full((B,), float("-inf"), dtype=torch.float32, device=dev)
        run_sum = torch.zeros(B, dtype=torch.float32, device=dev)
        for k0 in range(lo, hi, chunk):
            Kc = min(chunk, hi - k0)
            N = Kc * B
            cfg = L.DmvObjCfg()
            cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, 1, 1, Kc, D, Zd
            cfg.gen_dist, cfg.q_eps, cfg.x_per_component = L.GEN[dims.gen_dist], dims.q_eps, 0
            comp = None
            if mixture:
                cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, Zd, dims.C * Zd, 2 * dims.C * Zd
                if comp_idx is not None:
                    comp = comp_idx[k0:k0 + Kc].reshape(-1).to(dev, torch.int64).contiguous()
                else:
                    d = torch.empty((B, Kc), dtype=torch.int64, device=dev)
                    L.check(lib.dmv_categorical_draw(B, dims.C, Kc, L.ptr(logits), None, c_seed, c_off + k0 * B, rng.base_ptr(),
                                                     L.ptr(d), st))
                    comp = d.t().contiguous().reshape(-1)
            else:
                cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, 0, Zd, 0
            Z = torch.empty((N, Zd), dtype=torch.float32, device=dev)
            if eps is not None:
                e = eps[k0:k0 + Kc].reshape(N, Zd).contiguous().float()
                L.check(lib.dmv_latent_sample(C.byref(cfg), L.ptr(enc_out), L.ptr(e), 0, 0, None, Zd, L.ptr(comp), L.ptr(Z), None, st))
            else:
                L.check(lib.dmv_latent_sample(C.byref(cfg), L.ptr(enc_out), None, z_seed, z_off + k0 * B * Zd, rng.base_ptr(), Zd,
                                              L.ptr(comp), L.ptr(Z), None, st))
            dec_out, _ = mlp_forward(dec, Z, Zd, N, save=False)
            terms = torch.empty((5, N), dtype=torch.float32, device=dev)
            f = L.DmvObjFwd()
            f.q_kind, f.mix_C, f.kind = (L.Q_STRATIFIED, dims.C, L.TERMS_ONLY) if mixture else (L.Q_OWN, 1, L.TERMS_ONLY)
            f.terms = L.addr(terms)
            L.check(lib.dmv_obj_forward(C.byref(cfg), C.byref(f), L.ptr(dec_out), dec_out.shape[1], L.ptr(Xc), L.ptr(Mu),
                                        L.ptr(Z), Zd, L.ptr(enc_out), st))
            L.check(lib.dmv_lse_update(B, Kc, L.ptr(terms), 4 if mixture else 3, L.ptr(run_max), L.ptr(run_sum), st))
        return run_max, run_sum

    if return_state:
        return fold(*(k_range if k_range is not None else (0, K)))
    assert k_range is None, "k_range goes with return_state=True (the caller combines the states)"
    per_b = P.sharded_loglik(K, fold, shard=k_shard if k_shard is not None else (0, 1), group=group)
    return per_b.mean(), per_b
