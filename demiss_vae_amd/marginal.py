"""Marginal log-probability of (in)complete datapoints by numerical integration over a 1-d / 2-d latent grid: the
evaluation metric of the toy configs (reference: vgiwae/shared/vae_marginal_logprob.py:8-51 create_Z / create_2D_grid,
:66-90 get_marginal_logprob, :194-307).  The decoder runs once over the grid (inference forward kernel), one more
kernel pair integrates every datapoint against it."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from .mlp import PackedNet, mlp_forward
from .objectives import ModelDims, _u8


_GRIDS = {}


def latent_grid(latent_dim: int, device, grid_min=None, grid_max=None, grid_steps=None):
    """cached per (dimension, device, range): the grid is rebuilt on the host otherwise, at every evaluation step"""
    key = (latent_dim, str(torch.device(device)), grid_min, grid_max, grid_steps)
    if key not in _GRIDS:
        _GRIDS[key] = _latent_grid(latent_dim, device, grid_min, grid_max, grid_steps)
    return _GRIDS[key]


def _latent_grid(latent_dim: int, device, grid_min=None, grid_max=None, grid_steps=None):
    """(Z [Nz, latent_dim] on `device`, cell volume) with the reference's conventions: 1-d grids include both end
    points (torch.linspace) yet use (max - min) / steps as the cell; 2-d grids hold the lower-left corners
    min + i (max - min) / steps (numpy.mgrid with a step)."""
    if latent_dim == 1:
        lo, hi, steps = -15, 15, 4000
    elif latent_dim == 2:
        lo, hi, steps = -10, 10, 500
    else:
        raise NotImplementedError("Function not implemented for latent dimensionality greater than 2.")
    lo = grid_min if grid_min is not None else lo
    hi = grid_max if grid_max is not None else hi
    steps = grid_steps if grid_steps is not None else steps
    if latent_dim == 1:
        return torch.linspace(lo, hi, steps=steps).reshape(-1, 1).to(device), (hi - lo) / steps
    step = (hi - lo) / steps
    z0, z1 = np.mgrid[lo:hi:step, lo:hi:step]
    Z = torch.tensor(np.stack([z0, z1], -1)).float().reshape(-1, 2)
    return Z.to(device), step ** 2


@torch.no_grad()
def grid_marginal_logprob(dims: ModelDims, dec: PackedNet, X, M, *, compute_complete=False, grid_min=None, grid_max=None,
                          grid_steps=None):
    """log p(x_o) [B] (and log p(x) [B] when compute_complete) for X, M [B, D]."""
    lib, st = L.lib(), L.stream_ptr()
    dev = X.device
    Z, dz = latent_grid(dims.Zd, dev, grid_min, grid_max, grid_steps)
    Nz = Z.shape[0]
    dec_out, _ = mlp_forward(dec, Z.contiguous(), dims.Zd, Nz, save=False)
    logw = (-0.5 * Z * Z - 0.5 * math.log(2 * math.pi)).sum(-1) + math.log(dz)
    Xc, Mu = X.contiguous().float(), _u8(M)
    B = Xc.shape[0]
    ws = torch.empty(lib.dmv_grid_marginal_ws_floats(Nz, dims.D), dtype=torch.float32, device=dev)
    obs = torch.empty(B, dtype=torch.float32, device=dev)
    full = torch.empty(B, dtype=torch.float32, device=dev) if compute_complete else None
    L.check(lib.dmv_grid_marginal(L.GEN[dims.gen_dist], dims.D, Nz, L.ptr(dec_out), dec_out.shape[1], L.ptr(logw.contiguous()), B,
                                  L.ptr(Xc), L.ptr(Mu), L.ptr(ws), L.ptr(obs), L.ptr(full), st))
    return (obs, full) if compute_complete else obs


@torch.no_grad()
def posterior_divergences(dims: ModelDims, encode, dec: PackedNet, X, M, X_imp, M_imp, *, grid_min=None, grid_max=None,
                          grid_steps=None):
    """The nine posterior-divergence diagnostics of the toy study (vae_marginal_logprob.py:91-192), each per datapoint:
    forward / reverse KL and JSD between
      * p(z | x) and q(z | x)                                    (complete data)            [B]
      * p(z | x_obs) and the mixture (1/K) sum_k q(z | x_imp_k)  (incomplete data)          [B]
      * p(z | x_imp_k) and q(z | x_imp_k)                        (every imputed copy)       [B, K]
    on the latent grid.  X, M [B, D] (X complete, as the synthetic-missingness batches are); X_imp, M_imp [B, K, D];
    encode(X2, M2) -> variational parameters [rows, 2 Zd] (single Gaussians).  Returns the tuple in the reference's
    order: complete (fow, rev, jsd), incomplete (fow, rev, jsd), imputed (fow, rev, jsd)."""
    assert not dims.q_is_mixture and not dims.q_is_reparam, "single-Gaussian variational families only"
    lib, st = L.lib(), L.stream_ptr()
    dev = X.device
    B, K, D, Zd = X_imp.shape[0], X_imp.shape[1], dims.D, dims.Zd
    Z, dz = latent_grid(Zd, dev, grid_min, grid_max, grid_steps)
    Z = Z.contiguous()
    Nz = Z.shape[0]
    dec_out, _ = mlp_forward(dec, Z, Zd, Nz, save=False)
    log_prior = (-0.5 * Z * Z - 0.5 * math.log(2 * math.pi)).sum(-1).contiguous()
    logw = (log_prior + math.log(dz)).contiguous()
    ws = torch.empty(lib.dmv_grid_marginal_ws_floats(Nz, D), dtype=torch.float32, device=dev)
    gen = L.GEN[dims.gen_dist]

    def marginals(Xr, Mr):
        R = Xr.shape[0]
        obs, full = torch.empty(R, device=dev), torch.empty(R, device=dev)
        L.check(lib.dmv_grid_marginal(gen, D, Nz, L.ptr(dec_out), dec_out.shape[1], L.ptr(logw), R, L.ptr(Xr), L.ptr(Mr),
                                      L.ptr(ws), L.ptr(obs), L.ptr(full), st))
        return obs, full

    def q_of(Xr, Mr):
        p = encode(Xr, Mr).float()
        sigma = torch.nn.functional.softplus(p[..., Zd:2 * Zd]) + (1e-5 if dims.q_eps else 0.0)
        return p[..., :Zd].contiguous(), sigma.contiguous()

    def divs(Xr, Mr, log_marg, Kq, mu, sigma):
        R = Xr.shape[0]
        out = torch.empty(3, R, device=dev)
        L.check(lib.dmv_grid_divergences(gen, D, Zd, Nz, L.ptr(ws), L.ptr(log_prior), L.ptr(Z), dz, R, L.ptr(Xr), L.ptr(Mr),
                                         L.ptr(log_marg), Kq, L.ptr(mu), L.ptr(sigma), L.ptr(out), st))
        return out[0], out[1], out[2]

    Xc, Mu = X.contiguous().float(), _u8(M)
    obs, full = marginals(Xc, Mu)
    mu, sigma = q_of(X, M)
    complete = divs(Xc, None, full, 1, mu, sigma)
    Xi = X_imp.reshape(B * K, D).contiguous().float()
    Mi = _u8(M_imp.reshape(B * K, D))
    mu_i, sigma_i = q_of(X_imp.reshape(B * K, D), M_imp.reshape(B * K, D))
    incomplete = divs(Xc, Mu, obs, K, mu_i, sigma_i)                       # rows b, components k: [B][K][Zd]
    _, full_i = marginals(Xi, Mi)
    imputed = tuple(t.reshape(B, K) for t in divs(Xi, None, full_i, 1, mu_i, sigma_i))
    return (*complete, *incomplete, *imputed)
