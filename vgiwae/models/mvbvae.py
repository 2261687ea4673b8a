"""DeMissVAE: marginalised variational bound with persistent imputations
(reference: vgiwae/models/mvbvae.py:14-275)."""
from enum import Enum, auto
from typing import Optional

import torch

import demiss_vae_amd as dmv
from demiss_vae_amd.pl_compat import STEP_OUTPUT
from vgiwae.models.vae import VAE
from vgiwae.shared.mvb2_manual_optimisation import finish_manual_update, mvb2_manual_optimisation_step


class MVB_OBJECTIVE(Enum):
    mvb = auto()    # MVB objective for theta and phi
    cvi = auto()    # CVI objective for theta and phi
    mvb2 = auto()   # MVB objective for phi, CVI objective for theta


class MVBVAE(VAE):
    def __init__(self, *args, mvb_sample_num_imputations: int = None, mvb_objective: MVB_OBJECTIVE = MVB_OBJECTIVE["mvb"],
                 mvb_update_val_imps: bool = True, mvb_eval_val_loss: bool = False, mvb2_optim_gradclip_val: float = 0,
                 mvb2_optim_gradclip_alg: str = "norm", **kwargs):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()
        if self.hparams.mvb_objective is MVB_OBJECTIVE["mvb2"]:
            self.automatic_optimization = False     # two losses, one manual step

    mvb2_manual_optimisation_step = mvb2_manual_optimisation_step
    finish_manual_update = finish_manual_update

    def on_train_epoch_end(self) -> None:
        super().on_train_epoch_end()
        self.resample_imputations(stage="train")

    def on_validation_epoch_end(self) -> None:
        super().on_validation_epoch_end()
        self.resample_imputations(stage="val")

    # ---- objectives on imputed batches Xt [B,K,D], M [B,K,D]
    def _masks(self, Xt):
        """keep-mask keywords of the fused estimators (the caller-network engine runs torch's own dropout)"""
        if not self.fused:
            return {}
        B, K = Xt.shape[:2]
        ek, dk = self.dropout_masks(B * K, self.hparams.num_latent_samples * B * K)
        return {"enc_keep": ek, "dec_keep": dk}

    def estimate_mvb2_objectives(self, Xt, M, eps=None, reduce_mean=False):
        dims, enc, dec = self.engine()
        extra = {"reduce_mean": True} if (reduce_mean and self.fused) else {}
        out = self.estimator("mvb2_objectives")[0](dims, enc, dec, Xt, M, num_latent_samples=self.hparams.num_latent_samples,
                                                   STL=self.hparams.var_latent_STL, kl_analytic=self.hparams.kl_analytic,
                                                   eps=eps, **self._masks(Xt), **extra)
        return tuple(o.mean() for o in out) if (reduce_mean and not extra) else out

    def estimate_cvi_objective(self, Xt, M, eps=None):
        assert not self.hparams.kl_analytic, "Analytic KL not possible with the CVI objective"
        dims, enc, dec = self.engine()
        return self.estimator("cvi_objective")[0](dims, enc, dec, Xt, M, num_latent_samples=self.hparams.num_latent_samples,
                                                  STL=self.hparams.var_latent_STL, eps=eps, **self._masks(Xt))

    def estimate_mvb_objective(self, Xt, M, eps=None):
        """Complete-data ELBO of every imputed copy.  The reference's method of this name raises AttributeError
        (vgiwae/models/mvbvae.py:186,188), so this follows its documented intent; parity unpinned."""
        dims, enc, dec = self.engine()
        return self.estimator("mvb_objective")[0](dims, enc, dec, Xt, M, num_latent_samples=self.hparams.num_latent_samples,
                                                  STL=self.hparams.var_latent_STL, kl_analytic=self.hparams.kl_analytic,
                                                  eps=eps, **self._masks(Xt))

    def _objective_step(self, Xt, M, stage, optimise):
        kind = self.hparams.mvb_objective
        if kind is MVB_OBJECTIVE["mvb2"]:
            gen_obj, enc_obj = self.estimate_mvb2_objectives(Xt, M, reduce_mean=True)
            if optimise:     # ascent on the objectives = descent on the losses the reference passes (mvbvae.py:118-120)
                self.mvb2_manual_optimisation_step(enc_obj, gen_obj, maximise=True)
            self.log(f"mvb/{stage}", enc_obj, on_step=True, on_epoch=True, prog_bar=False, logger=True)
            self.log(f"cvi/{stage}", gen_obj, on_step=True, on_epoch=True, prog_bar=False, logger=True)
            return -gen_obj.detach()
        if kind is MVB_OBJECTIVE["mvb"]:
            obj = self.estimate_mvb_objective(Xt, M).mean()
            self.log(f"mvb/{stage}", obj, on_step=True, on_epoch=True, prog_bar=False, logger=True)
            return -obj
        if kind is MVB_OBJECTIVE["cvi"]:
            obj = self.estimate_cvi_objective(Xt, M).mean()
            self.log(f"cvi/{stage}", obj, on_step=True, on_epoch=True, prog_bar=False, logger=True)
            return -obj
        raise ValueError(f"Unknown MVB objective: {kind}")

    def training_step(self, batch, batch_idx) -> STEP_OUTPUT:
        dist = self.train_imp_distribution
        imputed = dist.get_imputed_datapoints(batch)                     # gather the K imputations of the batch
        imputed = self.update_imputations(imputed, stage="train")        # sampler + write-back into the buffer
        n_sub = self.hparams.mvb_sample_num_imputations
        if n_sub is not None:
            imputed = dist.get_imputed_datapoints(batch, num_imputations=n_sub)
        Xt, M = imputed[:2]
        loss = self._objective_step(Xt, M, "train", optimise=True)
        self.log("loss/train", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx) -> Optional[STEP_OUTPUT]:
        imputed = None
        if self.hparams.mvb_update_val_imps:
            imputed = self.val_imp_distribution.get_imputed_datapoints(batch)
            imputed = self.update_imputations(imputed, stage="val")
        if self.hparams.mvb_eval_val_loss:
            if imputed is None:
                imputed = self.val_imp_distribution.get_imputed_datapoints(batch)
            with torch.no_grad():
                loss = self._objective_step(imputed[0], imputed[1], "val", optimise=False)
            self.log("loss/val", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
