"""Model classes named by the YAML configs (reference: vgiwae/models/__init__.py:1-12).

The `*_FID`, `*MargLogprob` and `*GradVar` names are the reference's diagnostic wrappers
(models/vae_fid.py:174-180, vae_marglogprob.py:140-146, vae_gradvar.py:51-57).  Their diagnostics
(FID against a pre-trained encoder checkpoint, grid marginal log-probability for 2-d latents, gradient
variance) are outside the accelerated path: the classes here accept and record the same keyword
arguments so configs instantiate, train and evaluate exactly like their base class, and skip the diagnostic."""
from .vae import VAE
from .iwae import IWAE
from .mvbvae import MVBVAE, MVB_OBJECTIVE
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


def _with_ignored_diagnostics(base, suffix):
    defaults = _DIAGNOSTIC_KWARGS[suffix]

    def __init__(self, *args, **kwargs):
        extra = {k: kwargs.pop(k, v) for k, v in defaults.items()}
        base.__init__(self, *args, **kwargs)
        self.hparams.update(extra)
    name = base.__name__ + suffix
    return type(name, (base,), {"__init__": __init__, "__doc__": f"{base.__name__} (the {suffix} diagnostic is skipped)",
                                "__module__": __name__})


for _base in (VAE, IWAE, MVBVAE, MultipleIWAE, MultipleVAE):
    for _suffix in _DIAGNOSTIC_KWARGS:
        _cls = _with_ignored_diagnostics(_base, _suffix)
        globals()[_cls.__name__] = _cls
del _base, _suffix, _cls
