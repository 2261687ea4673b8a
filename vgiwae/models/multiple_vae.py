"""MissSVAE: stratified ELBO with a finite mixture encoder (reference: vgiwae/models/multiple_vae.py:14-195)."""
from typing import Optional

import torch

import demiss_vae_amd as dmv
from demiss_vae_amd.pl_compat import STEP_OUTPUT
from vgiwae.models.vae import VAE, _inner
from vgiwae.models.multiple_iwae import _MixtureLogitLogging


class MultipleVAE(_MixtureLogitLogging, VAE):
    def __init__(self, *args, kl_analytic: bool = False, log_var_distribution_mixture_component_weights_epoch_frequency: int = -1,
                 **kwargs):
        super().__init__(*args, **kwargs, kl_analytic=kl_analytic)
        self.save_hyperparameters()
        assert not kl_analytic, "Analytical KL is not implemented for a MultipleVAE."
        assert self.hparams.var_latent_distribution.name.startswith("stratified_mixture"), \
            "Only mixture-distributions are supported for encoder in multiple-VAE."

    def _bound(self, batch):
        h = self.hparams
        return estimate_multiple_elbo(self, batch, kl_analytic=False, var_latent_STL=h.var_latent_STL,
                                      num_latent_samples=h.num_latent_samples)

    def training_step(self, batch, batch_idx) -> STEP_OUTPUT:
        elbo = self._bound(batch)
        self.log("elbo/train", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        loss = -elbo
        self.log("loss/train", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx) -> Optional[STEP_OUTPUT]:
        with torch.no_grad():
            elbo = self._bound(batch)
        self.log("elbo/val", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/val", -elbo, on_step=True, on_epoch=True, prog_bar=True, logger=True)


def estimate_multiple_elbo(model, batch, kl_analytic: bool = False, var_latent_STL: bool = False,
                           num_latent_samples: int = 1, eps=None):
    """Batch mean of sum_c pi_c (log p(x_o, z_c) - log q_mix(z_c)) (vgiwae/models/multiple_vae.py:109-195)."""
    if kl_analytic:
        raise NotImplementedError("Analytical KL is not implemented for a MultipleVAE.")
    X, M = batch[:2]
    m = _inner(model)
    dims, enc, dec = m.engine()
    fn, fused = m.estimator("estimate_multiple_elbo")
    kw = {}
    if fused:
        kw["enc_keep"], kw["dec_keep"] = m.dropout_masks(X.shape[0], num_latent_samples * X.shape[0] * dims.C)
    bound, _ = fn(dims, enc, dec, X, M, num_latent_samples=num_latent_samples, STL=var_latent_STL, eps=eps, **kw)
    return bound
