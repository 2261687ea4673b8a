"""Model classes named by the YAML configs (reference: vgiwae/models/__init__.py:1-12).

The `*_FID`, `*MargLogprob` and `*GradVar` names are the reference's diagnostic wrappers
(models/vae_fid.py:174-180, vae_marglogprob.py:140-146, vae_gradvar.py:51-57).  The `*MargLogprob` classes
compute the grid marginal log-probability of the batch and, when asked for, the nine posterior-divergence
diagnostics on the device (demiss_vae_amd/marginal.py) and log them under the reference's metric names.  The `*_FID`
classes evaluate the Frechet distance between model samples and the data splits in the feature space of a trained
complete-data VAE's encoder every `fid_eval_step_freq` steps (vgiwae/shared/fid.py); without that checkpoint on disk
the metric is skipped with a notice.  The `*GradVar` classes log the per-epoch gradient variance and signal-to-noise
ratio of both networks (vgiwae/shared/vae_grad_var.py)."""
from .vae import VAE
from .iwae import IWAE
from .mvbvae import MVBVAE, MVB_OBJECTIVE
from .mvbiwae import MVBIWAE, MVB_IWAE_OBJ, IS_WEIGHTS
from .multiple_vae import MultipleVAE
from .multiple_iwae import MultipleIWAE

# keyword arguments of the reference's diagnostic wrappers and their defaults
# (models/vae_fid.py:20-26, models/vae_marglogprob.py:20-22; the GradVar wrapper adds none)
_DIAGNOSTIC_KWARGS = {
    "_FID": dict(fid_eval_step_freq=1, inception_model_type=None, inception_model_path=None, reference_samples=None,
                 num_samples_per_batch=None),
    "MargLogprob": dict(marginal_eval_batchsize=-1, marginal_eval_freq=1, marginal_eval_train=False, latent_grid_min=None,
                        latent_grid_max=None, latent_grid_steps=None, compute_complete_var_distribution_kl=False),
    "GradVar": dict(),
}


def _marginal_logprob(self, batch, stage):
    """marginal_logprob/<stage> and complete_marginal_logprob/<stage> (models/vae_marglogprob.py:38-52, 88-100)"""
    import torch
    from demiss_vae_amd.marginal import grid_marginal_logprob
    h = self.hparams
    last = getattr(getattr(self, "trainer", None), "max_epochs", None)
    if not (self.current_epoch % h.marginal_eval_freq == 0 or (last is not None and self.current_epoch == last - 1)):
        return
    with torch.no_grad():
        obs, full = grid_marginal_logprob(self.model_dims(), self.generator_network.packed_net(), batch[0], batch[1],
                                          compute_complete=True, grid_min=h.latent_grid_min, grid_max=h.latent_grid_max,
                                          grid_steps=h.latent_grid_steps)
    self.log(f"marginal_logprob/{stage}", obs.mean(), on_step=False, on_epoch=True, prog_bar=True, logger=True)
    self.log(f"complete_marginal_logprob/{stage}", full.mean(), on_step=False, on_epoch=True, prog_bar=True, logger=True)
    if h.compute_complete_var_distribution_kl:
        # vae_marglogprob.py:54-78, 98-122: divergences between the model's posteriors and the variational ones
        from demiss_vae_amd.marginal import posterior_divergences
        imputed = getattr(self, f"{stage}_imp_distribution").get_imputed_datapoints(batch)
        with torch.no_grad():
            divs = posterior_divergences(self.model_dims(), self.predict_var_latent_params, self.generator_network.packed_net(),
                                         batch[0], batch[1], imputed[0], imputed[1], grid_min=h.latent_grid_min,
                                         grid_max=h.latent_grid_max, grid_steps=h.latent_grid_steps)
        names = [f"{a}_posterior_{b}" for a in ("complete", "incomplete", "complete_imps") for b in ("kl_fow", "kl_rev", "jsd")]
        for name, value in zip(names, divs):
            self.log(f"{name}/{stage}", value.mean(), on_step=False, on_epoch=True, prog_bar=True, logger=True)


def _fid_members(base):
    """hooks of the FID wrapper (models/vae_fid.py:42-158)"""
    import os
    import torch
    from vgiwae.shared.fid import compute_inception_gaussian, gaussian_fid_score

    def _inception(self):
        """encoder of the trained VAE named by the hyper-parameters, or None; kept out of the module tree so that it
        never reaches state_dict() or the optimiser (vae_fid.py:93-100)"""
        if "_fid_inception" not in self.__dict__:
            h, net = self.hparams, None
            if h.inception_model_type is not None:
                if h.inception_model_type != "VAE_encoder":
                    raise NotImplementedError(f"Unknown inception model type: {h.inception_model_type}")
                if h.inception_model_path and os.path.exists(h.inception_model_path):
                    from demiss_vae_amd.run import load_model
                    net = load_model(h.inception_model_path, "VAE").var_latent_network.to(self.device).eval()
                    for p in net.parameters():
                        p.requires_grad_(False)
                else:
                    print(f"FID skipped: no inception model at {h.inception_model_path}")
            self.__dict__["_fid_inception"] = net
        return self.__dict__["_fid_inception"]

    def on_train_start(self):
        base.on_train_start(self)
        self.fid_ref_dataset_sizes, self.fid_ref_gaussians = [], []
        net = _inception(self)
        if net is None:
            return
        dm = self.trainer.datamodule
        batch = min(self.hparams.num_samples_per_batch or 5000, 5000)
        for ref in self.hparams.reference_samples or []:
            if ref not in ("dataset_test", "dataset_val", "dataset_train"):
                raise NotImplementedError(f"Unknown ref={ref}")
            split = ref[len("dataset_"):]
            if not hasattr(dm, f"{split}_data_core"):
                dm.setup(stage="test" if split == "test" else "fit")
            data = getattr(dm, f"{split}_data_core")[:]
            data = torch.as_tensor(data[0] if isinstance(data, tuple) else data).to(self.device)
            self.fid_ref_dataset_sizes.append(data.shape[0])
            self.fid_ref_gaussians.append(compute_inception_gaussian(data, inception_model=net, batch_size=batch))

    def eval_fid_score(self, ref_id):
        n = min(self.fid_ref_dataset_sizes[ref_id], 20000)
        per = self.hparams.num_samples_per_batch or n
        with torch.no_grad():
            samples = torch.cat([self.sample_vae(min(per, n - b)) for b in range(0, n, per)], dim=0)
            mean, cov = compute_inception_gaussian(samples, inception_model=_inception(self), batch_size=min(per, 5000))
        return gaussian_fid_score(*self.fid_ref_gaussians[ref_id], mean, cov)

    def on_train_batch_end(self, outputs, batch, batch_idx):
        step = self.trainer.global_step
        if not self.fid_ref_gaussians or not (step % self.hparams.fid_eval_step_freq == 0 or step == self.trainer.max_steps):
            return
        for ref_id, ref in enumerate(self.hparams.reference_samples):
            self.log(f"fid_score/{ref[len('dataset_'):]}", eval_fid_score(self, ref_id), on_step=True, on_epoch=False)
    return dict(on_train_start=on_train_start, eval_fid_score=eval_fid_score, on_train_batch_end=on_train_batch_end)


def _with_ignored_diagnostics(base, suffix):
    defaults = _DIAGNOSTIC_KWARGS[suffix]

    def __init__(self, *args, **kwargs):
        extra = {k: kwargs.pop(k, v) for k, v in defaults.items()}
        base.__init__(self, *args, **kwargs)
        self.hparams.update(extra)
        if suffix == "MargLogprob" and extra["latent_grid_min"] is not None and extra["latent_grid_max"] is not None:
            assert extra["latent_grid_min"] < extra["latent_grid_max"]
    name = base.__name__ + suffix
    members = {"__init__": __init__, "__doc__": f"{base.__name__} with the {suffix} wrapper's keyword arguments",
               "__module__": __name__}
    if suffix == "MargLogprob":
        def training_step(self, batch, batch_idx):
            out = base.training_step(self, batch, batch_idx)
            if self.hparams.marginal_eval_train:
                _marginal_logprob(self, batch, "train")
            return out

        def validation_step(self, batch, batch_idx):
            out = base.validation_step(self, batch, batch_idx)
            _marginal_logprob(self, batch, "val")
            return out
        members.update(training_step=training_step, validation_step=validation_step)
    if suffix == "_FID":
        members.update(_fid_members(base))
    if suffix == "GradVar":
        from vgiwae.shared import vae_grad_var as gv

        def on_train_epoch_start(self):
            base.on_train_epoch_start(self)
            gv.gradvar_on_train_epoch_start(self)

        def on_train_epoch_end(self):
            base.on_train_epoch_end(self)
            gv.gradvar_training_epoch_end(self)

        def on_before_optimizer_step(self, optimizer, optimizer_idx):
            base.on_before_optimizer_step(self, optimizer, optimizer_idx)
            gv.gradvar_on_before_optimizer_step(self, optimizer, optimizer_idx)
        members.update(on_train_epoch_start=on_train_epoch_start, on_train_epoch_end=on_train_epoch_end,
                       on_before_optimizer_step=on_before_optimizer_step)
    return type(name, (base,), members)


for _base in (VAE, IWAE, MVBVAE, MVBIWAE, MultipleIWAE, MultipleVAE):
    for _suffix in _DIAGNOSTIC_KWARGS:
        _cls = _with_ignored_diagnostics(_base, _suffix)
        globals()[_cls.__name__] = _cls
del _base, _suffix, _cls
