"""Base class of every model: owns the train / validation imputation buffers and updates them with the
model's sampler (reference: vgiwae/utils/base_imputer_module.py:20-161)."""
import os

import torch

from demiss_vae_amd.pl_compat import LightningModule
from vgiwae.shared.vae_imputation import VAESampler_Base
from vgiwae.utils.imputation_distribution import IMPUTATION_INIT, ImputationDistribution


class ImputerModuleBase(LightningModule):
    def __init__(self, num_imputations_train: int = 1, imputation_init_train: IMPUTATION_INIT = IMPUTATION_INIT("zero"),
                 num_imputations_val: int = 1, imputation_init_val: IMPUTATION_INIT = IMPUTATION_INIT("zero"),
                 init_imputation_distribution: bool = False, mvb_delay_imputation: int = 1,
                 mvb_resample_initial_imputations: bool = False, vae_sampler: VAESampler_Base = None,
                 save_imputations: bool = False):
        super().__init__()
        self.save_hyperparameters()

    def set_datamodule(self, datamodule):
        self.datamodule = datamodule

    def create_imputation_distribution(self, dataset, num_imputations: int, imputation_init: IMPUTATION_INIT):
        return ImputationDistribution(dataset=dataset, target_idx=0, mis_idx=1, index_idx=dataset.index_idx,
                                      num_imputations=num_imputations, imputation_init=imputation_init,
                                      device=self.device)

    def setup(self, stage=None):
        super().setup(stage)
        if stage == "fit" and self.hparams.init_imputation_distribution:
            dm = self.trainer.datamodule if getattr(self, "trainer", None) is not None else self.datamodule
            self.train_imp_distribution = self.create_imputation_distribution(
                dm.train_data, self.hparams.num_imputations_train, self.hparams.imputation_init_train)
            self.train_imp_distribution.init_imputations()
            self.val_imp_distribution = self.create_imputation_distribution(
                dm.val_data, self.hparams.num_imputations_val, self.hparams.imputation_init_val)
            self.val_imp_distribution.init_imputations()

    def on_train_end(self):
        out = super().on_train_end()
        if self.hparams.save_imputations:
            path = os.path.join(self.logger.experiment.get_logdir(), "imputations")
            os.makedirs(path, exist_ok=True)
            for stage in ("train", "val", "test"):
                dist = getattr(self, f"{stage}_imp_distribution", None)
                if dist is not None:
                    dist.save_imputations(os.path.join(path, f"{stage}_imputations_last.npz"))
        return out

    def _clip_values(self, imputed_batch, stage):
        sampler = self.hparams.vae_sampler
        if not getattr(sampler, "clip_imputations", False):
            return None, None
        mode = getattr(sampler, "clipping_mode", None)
        X = imputed_batch[0]
        if mode == "dataset":
            if stage not in ("train", "val"):
                raise NotImplementedError(f"Invalid stage={stage}")
            # built once per device: a host-to-device copy inside the step could not be recorded in a HIP graph
            cached = getattr(self, "_dataset_clip", None)
            if cached is None or cached[0].device != X.device:
                core = self.datamodule.train_data_core
                cached = (torch.as_tensor(core.data_min * 1.5, device=X.device, dtype=torch.float32),
                          torch.as_tensor(core.data_max * 1.5, device=X.device, dtype=torch.float32))
                self._dataset_clip = cached
            return cached
        if mode == "batch":
            # 1.5 x the per-dimension range of the observed values of this batch
            M = imputed_batch[1][:, 0]
            obs = X[:, 0].clone()
            lo = obs.masked_fill(~M, float("inf")).min(0)[0] * 1.5
            hi = obs.masked_fill(~M, float("-inf")).max(0)[0] * 1.5
            return lo, hi
        raise NotImplementedError(f"Invalid clipping_mode={mode}")

    def update_imputations(self, imputed_batch, stage):
        """Run the sampler on (X_imp, M) and write the new imputations back (vgiwae/utils/base_imputer_module.py:104-147).
        The last sampler iteration scatters straight into the device buffer."""
        clip_min, clip_max = self._clip_values(imputed_batch, stage)
        if self.current_epoch >= self.hparams.mvb_delay_imputation:
            if stage not in ("train", "val"):
                raise NotImplementedError()
            dist = self.train_imp_distribution if stage == "train" else self.val_imp_distribution
            X, M = imputed_batch[:2]
            with torch.no_grad():
                X_imp = self.hparams.vae_sampler(X, M, model=self, clip_values_min=clip_min, clip_values_max=clip_max,
                                                 buffer=dist.buffer, idx_imp=imputed_batch[dist.imp_idx])
            imputed_batch = (X_imp,) + tuple(imputed_batch[1:])
        return imputed_batch

    def resample_imputations(self, stage):
        if self.current_epoch < self.hparams.mvb_delay_imputation and self.hparams.mvb_resample_initial_imputations:
            if stage == "train":
                self.train_imp_distribution.init_imputations()
            elif stage == "val":
                self.val_imp_distribution.init_imputations()
            else:
                raise NotImplementedError()
