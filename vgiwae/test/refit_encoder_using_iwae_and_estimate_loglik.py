"""Refit the encoder of a trained model with an IWAE bound and estimate the test log-likelihood with many
importance samples (reference: vgiwae/test/refit_encoder_using_iwae_and_estimate_loglik.py:20-280)."""
from enum import Enum, auto
from typing import Optional

import torch

from demiss_vae_amd.pl_compat import LightningModule, STEP_OUTPUT
from vgiwae.models.iwae import estimate_iwelbo
from vgiwae.models.multiple_iwae import estimate_multiple_iwelbo
from vgiwae.shared.vae_enums import mixture_components
from vgiwae.utils.estimate_total_train_steps import PLEstimateTotalTrainSteps


class RefitEncoderClass(Enum):
    standard = auto()
    stratified = auto()


def num_importance_samples_for_stratified_mixtures(num_importance_samples, model_var_latent_distribution):
    """Importance samples per component so that components x samples = the requested budget (:265-280)."""
    C = mixture_components(model_var_latent_distribution)
    if C == 0 or not model_var_latent_distribution.name.startswith("stratified"):
        raise NotImplementedError(f"Unknown model_var_latent_distribution={model_var_latent_distribution}")
    return int(num_importance_samples / C)


class RefitEncoderUsingIWAE(LightningModule, PLEstimateTotalTrainSteps):
    def __init__(self, refit_encoder_class: RefitEncoderClass, var_latent_STL: bool = False, var_latent_DREG: bool = False,
                 num_latent_samples: int = 1, num_importance_samples: int = 1, num_test_latent_samples: int = 1,
                 num_test_importance_samples: int = 1, lr_latent: float = 0., amsgrad_latent: bool = False,
                 use_lr_scheduler: bool = False, max_scheduler_steps: int = -1,
                 add_mask_layer_to_existing_encoder: bool = False, use_looser_siwelbo_bound: bool = False):
        super().__init__()
        self.save_hyperparameters()

    def set_model(self, model):
        self.model = model
        if self.hparams.add_mask_layer_to_existing_encoder:
            self.model.var_latent_network.add_mask_inputs_to_existing_network()
            self.model.hparams.encoder_use_mis_mask = True

    def set_datamodule(self, datamodule):
        self.datamodule = datamodule

    def state_dict(self, *args, **kwargs):
        return self.model.state_dict(*args, **kwargs)      # checkpoints hold the wrapped model

    def vae_forward(self, *args, **kwargs):
        return self.model.vae_forward(*args, **kwargs)

    def engine(self):
        return self.model.engine()

    def _refit_bound(self, batch):
        h = self.hparams
        if h.refit_encoder_class is RefitEncoderClass.standard:
            return estimate_iwelbo(self, batch, var_latent_STL=h.var_latent_STL, var_latent_DREG=h.var_latent_DREG,
                                   num_latent_samples=h.num_latent_samples, num_importance_samples=h.num_importance_samples)
        if h.refit_encoder_class is RefitEncoderClass.stratified:
            n = num_importance_samples_for_stratified_mixtures(h.num_importance_samples,
                                                               self.model.hparams.var_latent_distribution)
            return estimate_multiple_iwelbo(self, batch, var_latent_STL=h.var_latent_STL, var_latent_DREG=h.var_latent_DREG,
                                            num_latent_samples=h.num_latent_samples,
                                            num_importance_samples_for_each_component=n,
                                            use_looser_bound=h.use_looser_siwelbo_bound)
        raise NotImplementedError(f"Unknown refit encoder class: {h.refit_encoder_class}")

    def training_step(self, batch, batch_idx) -> STEP_OUTPUT:
        elbo = self._refit_bound(batch)
        self.log("elbo/train", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/train", -elbo, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return -elbo

    def validation_step(self, batch, batch_idx) -> Optional[STEP_OUTPUT]:
        with torch.no_grad():
            elbo = self._refit_bound(batch)
        self.log("elbo/val", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/val", -elbo, on_step=True, on_epoch=True, prog_bar=True, logger=True)

    @property
    def can_shard_samples(self):
        """The K importance samples of a test batch can be split over GPUs: the streaming estimator computes them
        (residual-MLP networks, one outer sample; see vgiwae.models.iwae.estimate_iwelbo)."""
        h, m = self.hparams, self.model
        stratified = h.refit_encoder_class is RefitEncoderClass.stratified
        return bool(m.fused and h.num_test_latent_samples == 1 and not self.training and
                    (stratified or (h.num_test_importance_samples >= 1024 and not m.model_dims().q_is_mixture)))

    def test_step(self, batch, batch_idx, eps=None, comp_idx=None) -> Optional[STEP_OUTPUT]:
        """K-sample estimate of log p(x); a stratified encoder is sampled as a plain mixture (:189-222).  The runner
        sets `loglik_shard = (rank, world)` when the importance samples, not the test points, are split over GPUs."""
        h = self.hparams
        with torch.no_grad():
            elbo = estimate_iwelbo(self, batch, var_latent_STL=h.var_latent_STL, var_latent_DREG=h.var_latent_DREG,
                                   num_latent_samples=h.num_test_latent_samples,
                                   num_importance_samples=h.num_test_importance_samples,
                                   sample_stratifieddistr_without_stratification_norsample=(
                                       h.refit_encoder_class is RefitEncoderClass.stratified),
                                   eps=eps, comp_idx=comp_idx, k_shard=getattr(self, "loglik_shard", None))
        self.log("est_loglik/test", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        return elbo

    def configure_optimizers(self):
        h = self.hparams
        from vgiwae.models.vae import _adam
        opt = _adam([{"params": list(self.model.var_latent_network.parameters()), "amsgrad": h.amsgrad_latent,
                      "lr": h.lr_latent}])
        out = {"optimizer": opt}
        if h.use_lr_scheduler:
            steps = h.max_scheduler_steps if h.max_scheduler_steps != -1 else self.estimated_num_training_steps
            out["lr_scheduler"] = {"scheduler": torch.optim.lr_scheduler.CosineAnnealingLR(opt, steps, eta_min=0),
                                   "interval": "step", "frequency": 1}
        return out

    def configure_gradient_clipping(self, optimizer, optimizer_idx, gradient_clip_val=None, gradient_clip_algorithm=None):
        self.clip_gradients(optimizer, gradient_clip_val=gradient_clip_val, gradient_clip_algorithm=gradient_clip_algorithm)
