"""VAE for incomplete data with the reference's constructor, hooks and helper methods
(vgiwae/models/vae.py:27-473); the ELBO and its gradients run as one fused HIP pipeline.

`engine()` hands the kernels what they need (model dims + the two packed networks).  `vae_forward` /
`create_distribution` are kept for callers that want torch.distributions objects (diagnostics); they
call the same MLP kernels through autograd but are not used by the training / evaluation steps."""
from collections import namedtuple
from typing import Optional

import torch
from einops import rearrange

import demiss_vae_amd as dmv
from demiss_vae_amd.pl_compat import STEP_OUTPUT
from vgiwae.shared.neural_nets import ResidualFCNetwork
from vgiwae.shared.vae_enums import DISTRIBUTION, PRIOR_DISTRIBUTION, EPSILON, mixture_components, engine_name
from vgiwae.utils.base_imputer_module import ImputerModuleBase
from vgiwae.utils.estimate_total_train_steps import PLEstimateTotalTrainSteps

VAEForwardOutputs = namedtuple("VAEForwardOutputs", ["var_latent_distr", "var_latent_distr_detached", "Z",
                                                     "prior_distr", "generator_distr"])


class VAE(ImputerModuleBase, PLEstimateTotalTrainSteps):
    def __init__(self, generator_network, generator_distribution: DISTRIBUTION, var_latent_network,
                 var_latent_distribution: DISTRIBUTION, num_latent_samples: int, lr_generator: float, lr_latent: float,
                 *args, prior_distribution: PRIOR_DISTRIBUTION = PRIOR_DISTRIBUTION.std_normal,
                 amsgrad_generator: bool = False, amsgrad_latent: bool = False, use_lr_scheduler: bool = False,
                 max_scheduler_steps: int = -1, encoder_use_mis_mask: bool = False, var_latent_STL: bool = False,
                 kl_analytic: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()
        assert not (var_latent_STL and kl_analytic), '"Sticking the Landing" cannot be used with analytical KL.'
        self.generator_network = generator_network
        self.var_latent_network = var_latent_network

    # ------------------------------------------------------------------ bridge to the kernels
    @property
    def latent_dims(self):
        net = self.generator_network
        if isinstance(net, ResidualFCNetwork):
            return net.input_dim
        if hasattr(net, "latent_dim"):
            return net.latent_dim
        raise NotImplementedError(f"latent_dims for {type(net)}")

    def model_dims(self) -> dmv.ModelDims:
        h = self.hparams
        gen = h.generator_distribution
        if gen not in (DISTRIBUTION.normal, DISTRIBUTION.normal_with_eps, DISTRIBUTION.bern, DISTRIBUTION.bern_m1_p1):
            raise NotImplementedError(f"generator distribution {gen} has no HIP kernel")
        q = h.var_latent_distribution
        if q.name.startswith("cont_bern") or q.name.startswith("bern"):
            raise NotImplementedError(f"variational distribution {q} has no HIP kernel")
        net = self.generator_network
        out = net.output_dim if hasattr(net, "output_dim") else int(torch.tensor(net.output_shape).prod())
        D = out // 2 if gen in (DISTRIBUTION.normal, DISTRIBUTION.normal_with_eps) else out
        return dmv.ModelDims(D=D, Zd=self.latent_dims, gen_dist=gen.name, q_dist=engine_name(q),
                             C=max(1, mixture_components(q)), encoder_use_mis_mask=bool(h.encoder_use_mis_mask))

    @property
    def fused(self) -> bool:
        """Both networks are residual MLPs: the whole step runs inside the library.  Otherwise (convolutional
        trunks) the networks stay torch modules and the kernels take their outputs (demiss_vae_amd/external.py)."""
        return isinstance(self.generator_network, ResidualFCNetwork) and isinstance(self.var_latent_network, ResidualFCNetwork)

    def engine(self):
        """(ModelDims, encoder, decoder): packed networks for the fused pipelines, else two callables
        encode(X2 [rows, D], M2 [rows, D]) and decode(Z [N, Zd])."""
        if self.fused:
            return self.model_dims(), self.var_latent_network.packed_net(), self.generator_network.packed_net()
        return self.model_dims(), self.predict_var_latent_params, self.generator_network

    def dropout_masks(self, enc_rows: int, dec_rows: int):
        return self.var_latent_network.keep_masks(enc_rows), self.generator_network.keep_masks(dec_rows)

    def estimator(self, name: str):
        """The estimator `name` of the fused or of the caller-network engine, with keep-masks bound in."""
        if self.fused:
            return getattr(dmv, name), True
        from demiss_vae_amd import external
        return getattr(external, name), False

    # ------------------------------------------------------------------ reference helper surface
    def predict_var_latent_params(self, X, M):
        if self.hparams.encoder_use_mis_mask:
            from vgiwae.shared.vae_resnet import ResNetEncoder
            if isinstance(self.var_latent_network, ResNetEncoder):
                return self.var_latent_network(X, M.to(X.dtype))                    # mask as an extra image channel
            return self.var_latent_network(torch.cat([X, M.to(X.dtype)], dim=-1))   # all of X, then all of M
        return self.var_latent_network(X)

    def get_prior(self):
        if self.hparams.prior_distribution is PRIOR_DISTRIBUTION.std_normal:
            return torch.distributions.Normal(loc=0., scale=1.)
        raise NotImplementedError(f"prior {self.hparams.prior_distribution}")

    def split_params(self, params, num_params):
        return rearrange(params, "... (params d) -> params ... d", params=num_params)

    def create_distribution(self, params, distribution: DISTRIBUTION, *, validate_args=None):
        from vgiwae.shared.stratified_mixture_same_family import StratifiedMixtureSameFamily
        from vgiwae.shared.bernoulli_between_m1_and_p1 import Bernoulli_m1_p1
        td = torch.distributions
        name = distribution.name
        eps = EPSILON if name.endswith("with_eps") else 0.0
        if name in ("normal", "normal_with_eps"):
            loc, raw = self.split_params(params, 2)
            return td.Normal(loc, torch.nn.functional.softplus(raw) + eps, validate_args=validate_args)
        if name == "bern":
            return td.Bernoulli(logits=params, validate_args=validate_args)
        if name == "bern_m1_p1":
            return Bernoulli_m1_p1(logits=params, validate_args=validate_args)
        if name.startswith("stratified_mixture"):
            C = mixture_components(distribution)
            comp, logits = params[..., :-C], params[..., -C:]
            loc, raw = self.split_params(comp, 2)
            loc = rearrange(loc, "... (k z) -> ... k z", k=C)
            scale = rearrange(torch.nn.functional.softplus(raw) + eps, "... (k z) -> ... k z", k=C)
            return StratifiedMixtureSameFamily(td.Categorical(logits=logits),
                                               td.Independent(td.Normal(loc, scale, validate_args=validate_args), 1))
        raise NotImplementedError(f"create_distribution({distribution})")

    def sample_vae(self, num_samples):
        Z = self.get_prior().sample((num_samples, self.latent_dims)).to(self.device)
        return self.create_distribution(self.generator_network(Z), self.hparams.generator_distribution).sample()

    def vae_forward(self, X, M, *, return_detached_latent_distr, Z_sample_shape=None,
                    sample_stratifieddistr_without_stratification_norsample=False):
        if Z_sample_shape is None:
            Z_sample_shape = (self.hparams.num_latent_samples,)
        p = self.predict_var_latent_params(X, M)
        q = self.create_distribution(p, self.hparams.var_latent_distribution)
        q_det = self.create_distribution(p.detach(), self.hparams.var_latent_distribution) if return_detached_latent_distr else None
        if sample_stratifieddistr_without_stratification_norsample:
            assert not p.requires_grad
            q.set_sample_not_stratified()
            Z = q.sample(sample_shape=Z_sample_shape)
        else:
            Z = q.rsample(sample_shape=Z_sample_shape)
        gen = self.create_distribution(self.generator_network(Z), self.hparams.generator_distribution)
        return VAEForwardOutputs(q, q_det, Z, self.get_prior(), gen)

    # ------------------------------------------------------------------ steps
    def training_step(self, batch, batch_idx) -> STEP_OUTPUT:
        elbo, loss = estimate_elbo(self, batch, kl_analytic=self.hparams.kl_analytic, var_latent_STL=self.hparams.var_latent_STL,
                                   num_latent_samples=self.hparams.num_latent_samples, return_loss=True)
        self.log("elbo/train", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/train", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx) -> Optional[STEP_OUTPUT]:
        with torch.no_grad():
            elbo = estimate_elbo(self, batch, kl_analytic=self.hparams.kl_analytic, var_latent_STL=self.hparams.var_latent_STL,
                                 num_latent_samples=self.hparams.num_latent_samples)
        self.log("elbo/val", elbo, on_step=True, on_epoch=True, prog_bar=False, logger=True)
        self.log("loss/val", -elbo, on_step=True, on_epoch=True, prog_bar=True, logger=True)

    def configure_optimizers(self):
        """Adam with one parameter group per network and an optional per-step cosine schedule (vae.py:349-374)."""
        h = self.hparams
        groups = [{"params": list(self.generator_network.parameters()), "amsgrad": h.amsgrad_generator, "lr": h.lr_generator},
                  {"params": list(self.var_latent_network.parameters()), "amsgrad": h.amsgrad_latent, "lr": h.lr_latent}]
        opt = _adam(groups)
        out = {"optimizer": opt}
        if h.use_lr_scheduler:
            steps = h.max_scheduler_steps if h.max_scheduler_steps != -1 else self.estimated_num_training_steps
            out["lr_scheduler"] = {"scheduler": torch.optim.lr_scheduler.CosineAnnealingLR(opt, steps, eta_min=0),
                                   "interval": "step", "frequency": 1}
        return out

    def configure_gradient_clipping(self, optimizer, optimizer_idx, gradient_clip_val=None, gradient_clip_algorithm=None):
        self.clip_gradients(optimizer, gradient_clip_val=gradient_clip_val, gradient_clip_algorithm=gradient_clip_algorithm)


def _adam(groups):
    """The library's Adam (same update rule; clipping and both groups in two launches) for fp32 parameters on the
    GPU, torch's otherwise."""
    params = [p for g in groups for p in g["params"]]
    if params and all(p.is_cuda and p.dtype == torch.float32 for p in params) and len(params) <= 72:
        from demiss_vae_amd.optim import Adam
        return Adam(groups)
    return torch.optim.Adam(groups)


def _bound_and_loss(outs, return_loss):
    """What an estimator of this package returns: the batch-mean bound, or (bound, loss = -bound).  The fused engine
    hands the loss over as a third output (negated by the kernel that forms the mean); otherwise it is negated here."""
    if not return_loss:
        return outs[0]
    return (outs[0], outs[2]) if len(outs) > 2 else (outs[0], -outs[0])


def _inner(model):
    """The object that owns the networks (RefitEncoderUsingIWAE passes itself and wraps the trained model)."""
    return getattr(model, "model", model)


def estimate_elbo(model, batch, kl_analytic: bool = False, var_latent_STL: bool = False, num_latent_samples: int = 1,
                  eps=None, comp_idx=None, return_loss=False):
    """Batch-mean ELBO (vgiwae/models/vae.py:447-473) through the fused pipeline."""
    X, M = batch[:2]
    m = _inner(model)
    dims, enc, dec = m.engine()
    fn, fused = m.estimator("estimate_elbo")
    kw = {}
    if fused:
        kw["enc_keep"], kw["dec_keep"] = m.dropout_masks(X.shape[0], num_latent_samples * X.shape[0])
    if comp_idx is not None:
        kw["comp_idx"] = comp_idx
    if fused and return_loss:
        kw["with_loss"] = True
    return _bound_and_loss(fn(dims, enc, dec, X, M, num_latent_samples=num_latent_samples, kl_analytic=kl_analytic,
                              STL=var_latent_STL, eps=eps, **kw), return_loss)
