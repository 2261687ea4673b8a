"""Model classes named by the YAML configs (reference: vgiwae/models/__init__.py:1-12).

The `*_FID`, `*MargLogprob` and `*GradVar` names are the reference's diagnostic wrappers
(models/vae_fid.py:174-180, vae_marglogprob.py:140-146, vae_gradvar.py:51-57).  The `*MargLogprob` classes
compute the grid marginal log-probability of the batch and, when asked for, the nine posterior-divergence
diagnostics on the device (demiss_vae_amd/marginal.py) and log them under the reference's metric names; FID (needs a
pre-trained encoder checkpoint) and the gradient-variance probe are accepted as keyword arguments, recorded, and
skipped."""
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
    return type(name, (base,), members)


for _base in (VAE, IWAE, MVBVAE, MVBIWAE, MultipleIWAE, MultipleVAE):
    for _suffix in _DIAGNOSTIC_KWARGS:
        _cls = _with_ignored_diagnostics(_base, _suffix)
        globals()[_cls.__name__] = _cls
del _base, _suffix, _cls
