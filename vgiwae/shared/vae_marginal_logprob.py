"""Rejection sampler for imputations from the model's exact conditional (reference:
vgiwae/shared/vae_marginal_logprob.py:309-411; three toy configs).  z ~ p(z | x_obs) by rejection from a uniform
proposal over the latent box under the envelope max_grid p(x_obs, z), then x_mis ~ p(x | z).

The reference loops over datapoints and over rounds of Z_batchsize proposals; here every round proposes for the whole
batch at once: the decoder runs through the HIP forward kernel on B * Z_batchsize rows, acceptance and bookkeeping
are a few tensor operations.  The proposals come from the device generator, so the draws are not the reference's
(parity unpinned); the tests check the accepted samples against the grid posterior.  The grid marginal
log-probability itself lives in demiss_vae_amd/marginal.py."""
import math

import torch

from demiss_vae_amd.marginal import latent_grid
from demiss_vae_amd.mlp import mlp_forward
from vgiwae.shared.vae_imputation import VAESampler_Base, _engine

_LOG_2PI = math.log(2 * math.pi)


def _decode(dims, dec, Z):
    """(loc, scale) [N, D] of the Gaussian decoder at latents Z [N, Zd]"""
    if dims.gen_dist not in ("normal", "normal_with_eps"):
        raise NotImplementedError("the rejection sampler is used with Gaussian decoders (toy configs) only")
    out, _ = mlp_forward(dec, Z.contiguous().float(), dims.Zd, Z.shape[0], save=False)
    scale = torch.nn.functional.softplus(out[:, dims.D:2 * dims.D]) + (1e-5 if dims.gen_dist.endswith("with_eps") else 0.0)
    return out[:, :dims.D], scale


def _joint_logprob(x, m, Z, loc, scale):
    """log p(x_obs, z): x, m [B, D]; Z, loc, scale [B or 1, n, .] -> [B, n]"""
    lp = -0.5 * ((x.unsqueeze(1) - loc) / scale) ** 2 - torch.log(scale) - 0.5 * _LOG_2PI
    prior = (-0.5 * Z * Z - 0.5 * _LOG_2PI).sum(-1)
    return (lp * m.unsqueeze(1)).sum(-1) + prior


@torch.no_grad()
def grid_envelope(dims, dec, x, m, grid_min, grid_max, grid_steps, chunk=16384):
    """max over the latent grid of log p(x_obs, z) for every datapoint [B] (vae_marginal_logprob.py:389-395)"""
    Z, _ = latent_grid(dims.Zd, x.device, grid_min, grid_max, grid_steps)
    best = torch.full((x.shape[0],), -math.inf, device=x.device)
    for lo in range(0, Z.shape[0], chunk):
        z = Z[lo:lo + chunk]
        loc, scale = _decode(dims, dec, z)
        best = torch.maximum(best, _joint_logprob(x, m, z.unsqueeze(0), loc.unsqueeze(0), scale.unsqueeze(0)).max(1)[0])
    return best


@torch.no_grad()
def sample_pz_given_xo_with_rejection_sampling(dims, dec, x, m, num_samples, envelope, Z_batchsize, Z_min, Z_max,
                                               generator=None, max_rounds=100000):
    """[B, num_samples, Zd] latents from p(z | x_obs) (vae_marginal_logprob.py:309-347).  As in the reference a
    datapoint keeps its first num_samples accepted proposals, in proposal order."""
    B, Zd, dev = x.shape[0], dims.Zd, x.device
    samples = torch.empty(B, num_samples, Zd, device=dev)
    found = torch.zeros(B, dtype=torch.int64, device=dev)
    rows = torch.arange(B, device=dev).unsqueeze(1)
    for _ in range(max_rounds):
        Z = torch.rand(B, Z_batchsize, Zd, device=dev, generator=generator) * (Z_max - Z_min) + Z_min
        loc, scale = _decode(dims, dec, Z.reshape(-1, Zd))
        joint = _joint_logprob(x, m, Z, loc.reshape(B, Z_batchsize, -1), scale.reshape(B, Z_batchsize, -1))
        u = torch.rand(B, Z_batchsize, device=dev, generator=generator) * torch.exp(envelope).unsqueeze(1)
        accepted = u < torch.exp(joint)
        slot = found.unsqueeze(1) + torch.cumsum(accepted, 1) - 1          # where an accepted proposal would go
        take = accepted & (slot < num_samples)
        samples[rows.expand_as(slot)[take], slot[take]] = Z[take]
        found = torch.clamp(found + accepted.sum(1), max=num_samples)
        if bool((found >= num_samples).all()):
            return samples
    raise RuntimeError("rejection sampling did not fill every datapoint; the envelope or the latent box is off")


class VAESampler_Rejection(VAESampler_Base):
    def __init__(self, Z_batchsize, envelope_multiplier=1.0, grid_min=None, grid_max=None, grid_steps=None,
                 clip_imputations=False, clipping_mode=None, data_channel_dim=None):
        assert clip_imputations is False, "Clip imputations is not implemented for rejection sampler."
        self.Z_batchsize, self.envelope_multiplier = Z_batchsize, envelope_multiplier
        self.grid_min, self.grid_max, self.grid_steps = grid_min, grid_max, grid_steps
        self.clip_imputations, self.clipping_mode, self.data_channel_dim = clip_imputations, clipping_mode, data_channel_dim

    @torch.no_grad()
    def __call__(self, X, M, *, model, clip_values_min=None, clip_values_max=None, buffer=None, idx_imp=None, generator=None):
        assert clip_values_min is None and clip_values_max is None, "Clipping not implemented for rejection sampler."
        inner = getattr(model, "model", model)
        if not inner.fused:
            raise NotImplementedError("the rejection sampler needs a residual-MLP decoder (latent dimension 1 or 2)")
        dims, _, dec = _engine(model)
        K = X.shape[1]
        x, m = X[:, 0].float(), M[:, 0]          # observed values repeat across the K copies; old imputations are unused
        envelope = grid_envelope(dims, dec, x, m, self.grid_min, self.grid_max, self.grid_steps)
        envelope = envelope + math.log(self.envelope_multiplier)
        Z = sample_pz_given_xo_with_rejection_sampling(dims, dec, x, m, K, envelope, self.Z_batchsize, self.grid_min,
                                                       self.grid_max, generator=generator)
        loc, scale = _decode(dims, dec, Z.reshape(-1, dims.Zd))
        noise = torch.randn(loc.shape, device=loc.device, generator=generator)
        X_imp = (loc + scale * noise).reshape(X.shape)
        out = torch.where(M, X.float(), X_imp).contiguous()
        if buffer is not None:
            buffer.scatter(idx_imp, out)
        return out
