"""Oracle imputation sampler of the toy control study (reference: vgiwae/utils/mog_utils.py:10-108, 216-234, 277-554).

The toy data come from a known mixture of Gaussians, so the exact conditional p(x_mis | x_obs) is itself a mixture
with closed-form parameters.  `VAESampler_MoG_Oracle` imputes from it, or from a deliberately degraded version:
the true conditional (or another start distribution) is interpolated linearly towards a target (independent
dimensions, collapsed modes, shrunk variances, ...) with weight `interpolate_alpha` on the start.  This is
B x C x D x D dense linear algebra with D = 5: batched torch.linalg calls on the device, no custom kernel.
Parameters are kept in the precision the data file stores them in (float64), as the reference does.

Divergence estimates between mixtures and the Wasserstein interpolation are evaluation tools no config uses; they
are not provided."""
import math

import torch

from vgiwae.shared.vae_imputation import VAESampler_Base


def compute_conditional_mog_parameters(X, M, comp_log_probs, means, covs):
    """Parameters of p(x_mis | x_obs) = sum_c pi_c|o N(x_mis; mu_c|o, Sigma_c|o) for every row of X [B, D] with
    mask M (True = observed) under the mixture (comp_log_probs [C], means [C, D], covs [C, D, D]):
    (log pi_c|o [B, C], mu_c|o [B, C, D], Sigma_c|o [B, C, D, D]); entries of observed dimensions are zero.
    (mog_utils.py:10-61)"""
    obs = M.unsqueeze(1)                                        # [B, 1, D]
    mis = ~obs
    S = covs.unsqueeze(0)                                       # [1, C, D, D]
    mu = means.unsqueeze(0)
    S_mm = S * mis.unsqueeze(-1) * mis.unsqueeze(-2)
    S_mo = S * mis.unsqueeze(-1) * obs.unsqueeze(-2)
    S_oo = S * obs.unsqueeze(-1) * obs.unsqueeze(-2)
    # identity on the missing diagonal makes the observed block invertible inside the full-size matrix
    S_oo = S_oo + torch.eye(X.shape[-1], device=X.device, dtype=S.dtype) * mis.unsqueeze(-1)
    resid = (X.unsqueeze(1) - mu) * obs                         # [B, C, D]
    sol = torch.linalg.solve(S_oo, resid.unsqueeze(-1)).squeeze(-1)
    cond_means = mu * mis + (S_mo @ sol.unsqueeze(-1)).squeeze(-1)
    cond_covs = S_mm - S_mo @ torch.linalg.solve(S_oo, S_mo.transpose(-1, -2))
    n_obs = obs.sum(-1)
    log_p_obs = 0.5 * (-n_obs * math.log(2 * math.pi) - torch.logdet(S_oo) - (resid * sol).sum(-1))
    joint = comp_log_probs + log_p_obs
    return joint - torch.logsumexp(joint, dim=-1, keepdim=True), cond_means, cond_covs


def sample_sparse_mog(num_samples, M, comp_log_probs, cond_means, cond_covs, *, generator=None):
    """num_samples draws per row from the conditional mixtures; observed dimensions come back as zero.  [B, K, D]
    float32 (mog_utils.py:63-108)."""
    B, D = M.shape
    eye_obs = torch.eye(D, device=M.device, dtype=cond_covs.dtype) * M.unsqueeze(1).unsqueeze(-1)
    Lc, _ = torch.linalg.cholesky_ex(cond_covs + eye_obs)      # ones on the observed diagonal keep it factorisable
    probs = torch.exp(comp_log_probs)
    comp = torch.multinomial(probs / probs.sum(-1, keepdim=True), num_samples, replacement=True, generator=generator)
    rows = torch.arange(B, device=M.device)[:, None]
    L_sel, mean_sel = Lc[rows, comp], cond_means.expand(B, -1, -1)[rows, comp]
    noise = torch.randn(mean_sel.shape, device=M.device, dtype=mean_sel.dtype, generator=generator)
    return sample_given_draws(M, L_sel, mean_sel, noise)


def sample_given_draws(M, L_sel, mean_sel, noise):
    x = mean_sel + (L_sel @ noise.unsqueeze(-1)).squeeze(-1)
    return (x * (~M).unsqueeze(1)).float()


def linear_interpolate_sparse_mogs(params1, params2, *, alpha: float):
    """alpha * mixture 1 + (1 - alpha) * mixture 2, component by component (mog_utils.py:216-234)"""
    assert 0 <= alpha <= 1, "Linear interpolation coefficient must be between 0 and 1."
    lp1, lp2 = params1["comp_log_probs"], params2["comp_log_probs"]
    log_a, log_b = torch.log(torch.tensor(alpha)), torch.log(torch.tensor(1 - alpha))    # single precision, as there
    return {"comp_log_probs": torch.logsumexp(torch.stack(torch.broadcast_tensors(log_a + lp1, log_b + lp2)), dim=0),
            "means": alpha * params1["means"] + (1 - alpha) * params2["means"],
            "covs": alpha * params1["covs"] + (1 - alpha) * params2["covs"]}


def _moment_matched_gaussian(comp_log_probs, means, covs):
    """Mean and covariance of the single Gaussian the reference fits to the mixture.  Its helper receives the
    component probabilities where it expects their logs and exponentiates once more (mog_utils.py:277-291,
    298, 315), so the components are weighted by exp(pi_c), unnormalised; kept, since the control study's
    targets are defined by it."""
    w = torch.exp(torch.exp(comp_log_probs))
    mean = (means * w[:, None]).sum(0)
    dif = means - mean
    cov = ((covs + dif[..., None] @ dif[:, None, :]) * w[:, None, None]).sum(0)
    return mean, cov


def _kth_largest_mode(cond_log_probs, k):
    idx = torch.topk(torch.exp(cond_log_probs), k=k, dim=-1)[1][:, k - 1]
    return _one_mode(cond_log_probs, idx)


def _one_mode(cond_log_probs, idx):
    out = torch.full_like(cond_log_probs, -math.inf)
    out[torch.arange(out.shape[0], device=out.device), idx] = 0.0
    return out


_COLLAPSE = {"maxprob": 1, "secondlargestprob": 2, "thirdlargestprob": 3, "fourthlargestprob": 4, "fifthlargestprob": 5}


def construct_target_distribution(target, X, M, *, comp_log_probs, means, covs, comp_log_probs_given_o, means_m_given_o,
                                  covs_m_given_o, batch_size, device):
    """(log pi [B, C], means [B or 1, C, D], covs [B or 1, C, D, D]) of a named distribution over the missing values
    (mog_utils.py:293-431)."""
    C, D = means.shape
    eye = torch.eye(D, device=covs.device, dtype=covs.dtype)
    if target in ("indep_marginal_gaussian", "marginal_gaussian"):
        mean, cov = _moment_matched_gaussian(comp_log_probs, means, covs)
        if target.startswith("indep"):
            cov = cov * eye
        return comp_log_probs_given_o, mean.expand(1, C, D).to(device), cov.expand(1, C, D, D).to(device)
    if target in ("indep_marginal_mog", "marginal_mog"):
        cv = covs * eye if target.startswith("indep") else covs
        return comp_log_probs.expand(batch_size, C), means.unsqueeze(0), cv.unsqueeze(0)
    prefix = "conditional_mog_with_mode_collapse_to_"
    if target.startswith(prefix):
        which = target[len(prefix):-len("_mode")]
        if which in _COLLAPSE:
            lp = _kth_largest_mode(comp_log_probs_given_o, _COLLAPSE[which])
        elif which == "firstnonzero":
            nonzero = torch.exp(comp_log_probs_given_o) > 0
            lp = _one_mode(comp_log_probs_given_o, ((nonzero.cumsum(-1) == 1) & nonzero).max(-1)[1])
        else:
            raise NotImplementedError(f"target = {target!r} not implemented.")
        return lp, means_m_given_o, covs_m_given_o
    if target == "conditional_mog_with_equal_component_probabilities":
        nonzero = torch.exp(comp_log_probs_given_o) > 0
        lp = (nonzero / nonzero.sum(-1, keepdim=True)).to(comp_log_probs_given_o.dtype).log()
        return lp, means_m_given_o, covs_m_given_o
    if target == "conditional_independent_mog_with_var_shrinkage_to_0.01var":
        return comp_log_probs_given_o, means_m_given_o, covs_m_given_o * (0.01 * eye)
    if target == "conditional_independent_mog":
        return comp_log_probs_given_o, means_m_given_o, covs_m_given_o * eye
    if target == "conditional_independent_mog_created_from_true_indep_mog":
        return compute_conditional_mog_parameters(X, M, comp_log_probs, means, covs * eye)
    raise NotImplementedError(f"target = {target!r} not implemented.")


def construct_interpolated_conditional_mogs_from_true_mog(X, M, *, start_distribution, interpolate_method,
                                                          interpolate_target, alpha: float, comp_log_probs, means, covs):
    """Per-row mixture over the missing values: alpha * start + (1 - alpha) * target (mog_utils.py:433-501)."""
    if interpolate_method != "linear":
        raise NotImplementedError(f"interpolate_method = {interpolate_method!r}: every config interpolates linearly")
    cond = compute_conditional_mog_parameters(X, M, comp_log_probs, means, covs)
    kw = dict(comp_log_probs=comp_log_probs, means=means, covs=covs, comp_log_probs_given_o=cond[0],
              means_m_given_o=cond[1], covs_m_given_o=cond[2], batch_size=X.shape[0], device=X.device)
    mis = (~M).unsqueeze(1)

    def only_missing(lp, mu, cv):                               # observed dimensions carry no mass in these mixtures
        return {"comp_log_probs": lp, "means": mu * mis, "covs": cv * mis.unsqueeze(-1) * mis.unsqueeze(-2)}
    target = only_missing(*construct_target_distribution(interpolate_target, X, M, **kw))
    if start_distribution is None or start_distribution == "true_conditional":
        start = {"comp_log_probs": cond[0], "means": cond[1], "covs": cond[2]}
    else:
        start = only_missing(*construct_target_distribution(start_distribution, X, M, **kw))
    return linear_interpolate_sparse_mogs(start, target, alpha=alpha)


class VAESampler_MoG_Oracle(VAESampler_Base):
    def __init__(self, start_distribution: str = "true_conditional", interpolate_method: str = "linear",
                 interpolate_target: str = "indep_marginal_gaussian", interpolate_alpha: float = 1.0,
                 clip_imputations=False, clipping_mode=None, data_channel_dim=None):
        assert 0 <= interpolate_alpha <= 1, "Interpolation coefficient must be between 0 and 1."
        assert clip_imputations is False, "Clip imputations is not implemented for MoG oracle sampler."
        self.start_distribution, self.interpolate_method = start_distribution, interpolate_method
        self.interpolate_target, self.interpolate_alpha = interpolate_target, interpolate_alpha
        self.clip_imputations, self.clipping_mode, self.data_channel_dim = clip_imputations, clipping_mode, data_channel_dim
        self._mog = None

    def _true_mog(self, model, device):
        if self._mog is None or self._mog[0].device != device:
            f = model.datamodule.train_data_core.data_file
            self._mog = (torch.as_tensor(f["comp_probs"]).log().squeeze(0).to(device), torch.as_tensor(f["means"]).to(device),
                         torch.as_tensor(f["covs"]).to(device))
        return self._mog

    def __call__(self, X, M, *, model, clip_values_min=None, clip_values_max=None, buffer=None, idx_imp=None, generator=None):
        """X, M [B, K, D] -> new imputations [B, K, D]: K fresh draws of the missing values of every datapoint (the
        previous imputations are not used); written back into `buffer` at rows idx_imp when given."""
        assert clip_values_min is None and clip_values_max is None, "Clipping not implemented for MoG oracle sampler."
        K = X.shape[1]
        x, m = X[:, 0], M[:, 0]
        comp_log_probs, means, covs = self._true_mog(model, X.device)
        params = construct_interpolated_conditional_mogs_from_true_mog(
            x, m, start_distribution=self.start_distribution, interpolate_method=self.interpolate_method,
            interpolate_target=self.interpolate_target, alpha=self.interpolate_alpha, comp_log_probs=comp_log_probs,
            means=means, covs=covs)
        X_imp = sample_sparse_mog(K, m, params["comp_log_probs"], params["means"], params["covs"], generator=generator)
        out = (x.unsqueeze(1) * m.unsqueeze(1) + X_imp * (~m).unsqueeze(1)).contiguous()
        if buffer is not None:
            buffer.scatter(idx_imp, out)
        return out
