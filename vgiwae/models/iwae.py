"""IWAE for incomplete data (reference: vgiwae/models/iwae.py:11-154): constructor, steps and the module-level
estimator with the reference's signature; the bound, its STL / DREG gradients and the logsumexp over the
importance axis run in the fused HIP pipeline."""
from typing import Optional

import torch

import demiss_vae_amd as dmv
from demiss_vae_amd.pl_compat import STEP_OUTPUT
from vgiwae.models.vae import VAE, _inner, _bound_and_loss


class IWAE(VAE):
    def __init__(self, *args, num_importance_samples: int = 1, kl_analytic: bool = False,
                 var_latent_DREG: bool = False, **kwargs):
        super().__init__(*args, **kwargs, kl_analytic=kl_analytic)
        self.save_hyperparameters()
        assert not kl_analytic, "Analytical KL is not tractable for a IWAE."
        assert not (self.hparams.var_latent_DREG and self.hparams.var_latent_STL), \
            "Cannot use STL and DREG gradients at the same time."

    def _bound(self, batch, return_loss=False):
        h = self.hparams
        return estimate_iwelbo(self, batch, var_latent_STL=h.var_latent_STL, var_latent_DREG=h.var_latent_DREG,
                               num_latent_samples=h.num_latent_samples, num_importance_samples=h.num_importance_samples,
                               return_loss=return_loss)

    def training_step(self, batch, batch_idx) -> STEP_OUTPUT:
        elbo, loss = self._bound(batch, return_loss=True)       # loss = -elbo, formed where the mean is formed
        self.log("elbo/train", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/train", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx) -> Optional[STEP_OUTPUT]:
        with torch.no_grad():
            elbo = self._bound(batch)
        self.log("elbo/val", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/val", -elbo, on_step=True, on_epoch=True, prog_bar=True, logger=True)

    def log_weights_to_iwelbo(self, *args, **kwargs):
        return log_weights_to_iwelbo(*args, **kwargs)


def log_weights_to_iwelbo(log_weights, *, create_Z_hook=None, var_latent_DREG=False):
    """logsumexp over dim 1 minus log I for callers that already hold log-weights [z, i, b, ...]
    (vgiwae/models/iwae.py:70-90).  Small tensor glue; the training path never calls it."""
    iwelbo = torch.logsumexp(log_weights, dim=1) - torch.log(torch.tensor(float(log_weights.shape[1])))
    if var_latent_DREG and create_Z_hook is not None:
        with torch.no_grad():
            create_Z_hook(torch.softmax(log_weights, dim=1))
    return iwelbo


def estimate_iwelbo(model, batch, var_latent_STL: bool = False, var_latent_DREG: bool = False,
                    num_latent_samples: int = 1, num_importance_samples: int = 1, *,
                    sample_stratifieddistr_without_stratification_norsample=False, eps=None, comp_idx=None, k_shard=None,
                    return_loss=False):
    """Batch-mean IWAE bound (vgiwae/models/iwae.py:125-154).

    With gradients enabled this is the fused training pipeline.  The streaming K-sample estimator (no [K, B, .] tensor,
    K folded chunk by chunk into a running logsumexp) takes over only where it computes the same estimator:
      * a stratified encoder that the caller asks to sample as a plain mixture (the flag, vgiwae/models/vae.py:298-301;
        the reference's own call raises for any other family, here the flag is ignored for them);
      * under no_grad with >= 1024 importance samples, for families whose own sampling is ancestral anyway (a single
        Gaussian or a reparametrised mixture).  A stratified encoder without the flag keeps its one-draw-per-component
        sampling, and a network with live dropout keeps its masks: both stay on the non-streaming pipeline.
    k_shard = (rank, world), streaming estimator only: the importance samples are split over the ranks of the process
    group and combined with a logsumexp all-reduce (demiss_vae_amd.imputation.estimate_loglik)."""
    from demiss_vae_amd.imputation import estimate_loglik
    X, M = batch[:2]
    m = _inner(model)
    dims, enc, dec = m.engine()
    unstratified = sample_stratifieddistr_without_stratification_norsample and dims.q_is_mixture
    if unstratified:
        assert not torch.is_grad_enabled(), "Cannot use stratifieddistr_without_stratification_norsample with gradients."
    dropout_live = m.fused and m.training and (enc.dims.dropout > 0 or dec.dims.dropout > 0)
    big = not torch.is_grad_enabled() and num_importance_samples >= 1024 and not dims.q_is_mixture
    if m.fused and num_latent_samples == 1 and not dropout_live and (unstratified or big):
        val, _ = estimate_loglik(dims, enc, dec, X, M, num_importance_samples=num_importance_samples, eps=eps,
                                 comp_idx=comp_idx, k_shard=k_shard)
        return (val, -val) if return_loss else val
    assert k_shard is None or k_shard[1] == 1, "importance samples are sharded by the streaming estimator only"
    if unstratified:
        # all samples in one pass (convolutional trunks; residual MLPs with live dropout): a stratified mixture sampled
        # as a plain mixture is the ancestrally sampled family with the same parameter layout
        import dataclasses
        dims = dataclasses.replace(dims, q_dist=dims.q_dist.replace("stratified", "reparam"))
    fn, fused = m.estimator("estimate_iwelbo")
    kw = {}
    if fused:
        kw["enc_keep"], kw["dec_keep"] = m.dropout_masks(X.shape[0], num_latent_samples * num_importance_samples * X.shape[0])
    if comp_idx is not None:
        kw["comp_idx"] = comp_idx
    if fused and return_loss:
        kw["with_loss"] = True
    return _bound_and_loss(fn(dims, enc, dec, X, M, num_latent_samples=num_latent_samples,
                              num_importance_samples=num_importance_samples, STL=var_latent_STL, DREG=var_latent_DREG, eps=eps,
                              **kw), return_loss)
