"""Imputation samplers with the reference's constructor signatures and call convention
(vgiwae/shared/vae_imputation.py:21-23, 54-82, 575-696), running on the device kernels.

`sampler(X, M, model=model, clip_values_min=..., clip_values_max=...)` takes the current imputations
X [B,K,D] and masks M [B,K,D] and returns the updated imputations.  Schemes no shipped config uses
(systematic / local / independent resampling, smis weights, prior_mixture_probability, batchsize > 0)
are rejected at construction."""
from demiss_vae_amd import imputation as _imp


def _engine(model):
    inner = getattr(model, "model", model)     # RefitEncoderUsingIWAE wraps the trained model
    return inner.engine()


class VAESampler_Base:
    def __call__(self, X, M, *, clip_values_min=None, clip_values_max=None):
        raise NotImplementedError()


class VAESampler_PseudoGibbs(VAESampler_Base):
    def __init__(self, num_iterations: int, *, batchsize: int = -1, clip_imputations: bool = False,
                 data_channel_dim: int = None):
        if batchsize > 0:
            raise NotImplementedError("batchsize > 0 is broken in the reference (vae_imputation.py:73-75) and unused")
        if data_channel_dim is not None:
            raise NotImplementedError("data_channel_dim (image-shaped batches) is handled by flattening before the call")
        self.num_iterations, self.batchsize = num_iterations, batchsize
        self.clip_imputations, self.data_channel_dim = clip_imputations, data_channel_dim
        self.clipping_mode = None

    def __call__(self, X, M, *, model, clip_values_min=None, clip_values_max=None, buffer=None, idx_imp=None, **noise):
        if not self.clip_imputations:
            clip_values_min = clip_values_max = None
        dims, enc, dec = _engine(model)
        return _imp.pseudo_gibbs(dims, enc, dec, X, M, num_iterations=self.num_iterations, clip_min=clip_values_min,
                                 clip_max=clip_values_max, training=model.training, buf=buffer, idx_imp=idx_imp, **noise)


class VAESampler_LAIR(VAESampler_Base):
    def __init__(self, num_iterations, num_imp_samples, num_prior_replenish_proposals, is_weights, resampling_scheme, *,
                 prior_mixture_probability=0.0, batchsize=-1, resample_final_imps=False, resampling_mode="multinomial",
                 glr_cluster_size=None, clip_imputations=False, clipping_mode=None, data_channel_dim=None):
        if is_weights != "dmis" or resampling_scheme != "global_resampling" or resampling_mode != "multinomial":
            raise NotImplementedError("only is_weights='dmis', resampling_scheme='global_resampling', "
                                      "resampling_mode='multinomial' (what every shipped config uses) have kernels")
        if prior_mixture_probability > 0 or batchsize > 0 or data_channel_dim is not None:
            raise NotImplementedError("prior_mixture_probability / batchsize / data_channel_dim are not used by any config")
        if resample_final_imps:
            assert num_iterations > 1, "Need >1 iterations to resample final imps"
        self.num_iterations, self.num_imp_samples = num_iterations, num_imp_samples
        self.num_prior_replenish_proposals = num_prior_replenish_proposals
        self.prior_mixture_probability, self.is_weights = prior_mixture_probability, is_weights
        self.resampling_scheme, self.glr_cluster_size, self.batchsize = resampling_scheme, glr_cluster_size, batchsize
        self.resample_final_imps, self.resampling_mode = resample_final_imps, resampling_mode
        self.clip_imputations, self.clipping_mode, self.data_channel_dim = clip_imputations, clipping_mode, data_channel_dim

    def __call__(self, X, M, *, model, clip_values_min=None, clip_values_max=None, buffer=None, idx_imp=None, **noise):
        if not self.clip_imputations:
            clip_values_min = clip_values_max = None
        dims, enc, dec = _engine(model)
        return _imp.lair(dims, enc, dec, X, M, num_iterations=self.num_iterations, num_imp_samples=self.num_imp_samples,
                         num_prior_replenish_proposals=self.num_prior_replenish_proposals,
                         resample_final_imps=self.resample_final_imps, clip_min=clip_values_min, clip_max=clip_values_max,
                         training=model.training, buf=buffer, idx_imp=idx_imp, **noise)
