"""MissSIWAE: IWAE with a finite stratified mixture encoder (reference: vgiwae/models/multiple_iwae.py:14-237)."""
import os
from typing import Optional

import numpy as np
import torch

import demiss_vae_amd as dmv
from demiss_vae_amd.pl_compat import STEP_OUTPUT
from vgiwae.models.vae import _bound_and_loss, VAE, _inner


class _MixtureLogitLogging:
    """Optional per-epoch dump of the mixture logits (multiple_iwae.py:86-121)."""

    def on_train_start(self):
        out = super().on_train_start()
        self.mixture_weight_logits = []
        return out

    def on_train_epoch_start(self):
        out = super().on_train_epoch_start()
        self.mixture_weight_logits_epoch = []
        return out

    def _log_due(self):
        f = self.hparams.get("log_var_distribution_mixture_component_weights_epoch_frequency", -1)
        return f != -1 and self.current_epoch % f == 0

    def on_train_epoch_end(self):
        out = super().on_train_epoch_end()
        if self._log_due() and self.mixture_weight_logits_epoch:
            self.mixture_weight_logits.append(np.concatenate(self.mixture_weight_logits_epoch, axis=0))
        return out

    def on_train_end(self):
        out = super().on_train_end()
        if self.hparams.get("log_var_distribution_mixture_component_weights_epoch_frequency", -1) != -1:
            np.savez_compressed(os.path.join(self.logger.experiment.get_logdir(), "mixture_var_weight_logits.npz"),
                                mixture_weight_logits=np.stack(self.mixture_weight_logits, 0))
        return out


class MultipleIWAE(_MixtureLogitLogging, VAE):
    def __init__(self, *args, num_importance_samples_for_each_component: int = 1, kl_analytic: bool = False,
                 var_latent_DREG: bool = False, log_var_distribution_mixture_component_weights_epoch_frequency: int = -1,
                 use_looser_bound: bool = False, **kwargs):
        super().__init__(*args, **kwargs, kl_analytic=kl_analytic)
        self.save_hyperparameters()
        assert not kl_analytic, "Analytical KL is not tractable for a IWAE."
        assert not self.hparams.var_latent_DREG, "DREG gradients are not yet implemented for multiple-IWAE."
        assert self.hparams.var_latent_distribution.name.startswith("stratified_mixture"), \
            "Only mixture-distributions are supported for encoder in multiple-IWAE."

    def estimate_multiple_iwelbo(self, batch, return_loss=False):
        h = self.hparams
        return estimate_multiple_iwelbo(self, batch, var_latent_STL=h.var_latent_STL, var_latent_DREG=h.var_latent_DREG,
                                        num_latent_samples=h.num_latent_samples,
                                        num_importance_samples_for_each_component=h.num_importance_samples_for_each_component,
                                        use_looser_bound=h.use_looser_bound, return_loss=return_loss)

    def training_step(self, batch, batch_idx) -> STEP_OUTPUT:
        elbo, loss = self.estimate_multiple_iwelbo(batch, return_loss=True)     # loss = -elbo
        self.log("elbo/train", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/train", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx) -> Optional[STEP_OUTPUT]:
        with torch.no_grad():
            elbo = self.estimate_multiple_iwelbo(batch)
        self.log("elbo/val", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/val", -elbo, on_step=True, on_epoch=True, prog_bar=True, logger=True)


def estimate_multiple_iwelbo(model, batch, var_latent_STL: bool = False, var_latent_DREG: bool = False,
                             num_latent_samples: int = 1, num_importance_samples_for_each_component: int = 1,
                             use_looser_bound: bool = False, eps=None, return_loss=False):
    """Batch-mean MissSIWAE bound, tight or looser (vgiwae/models/multiple_iwae.py:124-237)."""
    assert not var_latent_DREG
    X, M = batch[:2]
    m = _inner(model)
    dims, enc, dec = m.engine()
    I = num_importance_samples_for_each_component
    fn, fused = m.estimator("estimate_multiple_iwelbo")
    kw = {}
    if fused:
        kw["enc_keep"], kw["dec_keep"] = m.dropout_masks(X.shape[0], num_latent_samples * I * X.shape[0] * dims.C)
    if fused and return_loss:
        kw["with_loss"] = True
    outs = fn(dims, enc, dec, X, M, num_latent_samples=num_latent_samples, num_importance_samples_for_each_component=I,
              STL=var_latent_STL, looser=use_looser_bound, eps=eps, **kw)
    if model.training and getattr(model, "_log_due", lambda: False)() and hasattr(model, "mixture_weight_logits_epoch"):
        # diagnostic dump of the (normalised) mixture logits on logging epochs (multiple_iwae.py:220-226); the pipeline
        # keeps the encoder output to itself, so this is one more encoder pass without gradients
        with torch.no_grad():
            p = m.predict_var_latent_params(X * M, M)
            model.mixture_weight_logits_epoch.append(torch.log_softmax(p[..., -dims.C:], dim=-1).cpu().numpy())
    return _bound_and_loss(outs, return_loss)
