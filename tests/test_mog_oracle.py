"""Oracle sampler of the toy control study against vectors computed by the reference's own functions
(vgiwae/utils/mog_utils.py; tools/golden_gen/make_golden.py, group mog_oracle).  Pure tensor code: runs on the CPU."""
import numpy as np
import pytest
import torch

import golden_util as gu
from vgiwae.utils import mog_utils as mu


@pytest.fixture(scope="module")
def g():
    return {k: (torch.from_numpy(v) if v.dtype.kind in "fbi" else v) for k, v in gu.load("mog_oracle").items()}


def close(a, b):
    # float64 throughout, but log-probabilities near zero are differences of large numbers: two hosts' libm / BLAS
    # builds differ there by a few 1e-8 absolute
    torch.testing.assert_close(a.expand(b.shape) if a.dim() == b.dim() else a, b, rtol=1e-7, atol=1e-7)


def test_conditional_parameters(g):
    lp, means, covs = mu.compute_conditional_mog_parameters(g["X"], g["M"], g["comp_log_probs"], g["means"], g["covs"])
    close(lp, g["cond_lp"]), close(means, g["cond_means"]), close(covs, g["cond_covs"])
    M = g["M"]
    assert (means[M.unsqueeze(1).expand_as(means)] == 0).all()                  # nothing on observed dimensions
    torch.testing.assert_close(torch.exp(lp).sum(-1), torch.ones(len(M), dtype=lp.dtype))
    # fully observed row: responsibilities of the complete datapoint
    full = torch.distributions.MultivariateNormal(g["means"], g["covs"]).log_prob(g["X"][1].double()) + g["comp_log_probs"]
    torch.testing.assert_close(lp[1], full - torch.logsumexp(full, 0))


def test_every_target_and_start(g):
    for i, t in enumerate(g["targets"]):
        for start, alpha, tag in ((None, 0.25, "cond"), ("indep_marginal_mog", 0.5, "indep")):
            p = mu.construct_interpolated_conditional_mogs_from_true_mog(
                g["X"], g["M"], start_distribution=start, interpolate_method="linear", interpolate_target=str(t), alpha=alpha,
                comp_log_probs=g["comp_log_probs"], means=g["means"], covs=g["covs"])
            close(p["comp_log_probs"], g[f"t{i}_{tag}_lp"])
            close(p["means"], g[f"t{i}_{tag}_means"])
            close(p["covs"], g[f"t{i}_{tag}_covs"])
    with pytest.raises(NotImplementedError):
        mu.construct_interpolated_conditional_mogs_from_true_mog(
            g["X"], g["M"], start_distribution=None, interpolate_method="wasserstein", interpolate_target="marginal_mog",
            alpha=0.5, comp_log_probs=g["comp_log_probs"], means=g["means"], covs=g["covs"])


def test_sampling_with_the_reference_draws(g):
    X, M, K = g["X"], g["M"], int(g["K"])
    p = mu.construct_interpolated_conditional_mogs_from_true_mog(
        X, M, start_distribution=None, interpolate_method="linear", interpolate_target=str(g["targets"][0]), alpha=0.5,
        comp_log_probs=g["comp_log_probs"], means=g["means"], covs=g["covs"])
    D = X.shape[1]
    Lc, _ = torch.linalg.cholesky_ex(p["covs"] + torch.eye(D, dtype=torch.float64) * M.unsqueeze(1).unsqueeze(-1))
    comp = g["sample_comp"].T                                              # recorded draws [K, B] -> [B, K]
    rows = torch.arange(len(M))[:, None]
    out = mu.sample_given_draws(M, Lc[rows, comp], p["means"].expand(len(M), -1, -1)[rows, comp], g["sample_eps"])
    torch.testing.assert_close(out, g["sample_out"], rtol=1e-6, atol=1e-6)
    # own draws: right shape, zero on observed entries, moments of the missing entries near the mixture's
    gen = torch.Generator().manual_seed(0)
    big = mu.sample_sparse_mog(4000, M, p["comp_log_probs"], p["means"], p["covs"], generator=gen)
    assert big.shape == (len(M), 4000, D) and big.dtype == torch.float32
    assert (big[M.unsqueeze(1).expand_as(big)] == 0).all()
    want = (torch.exp(p["comp_log_probs"]).unsqueeze(-1) * p["means"]).sum(1)
    err = (big.double().mean(1) - want).abs()
    assert (err <= 5 * big.double().std(1) / 4000 ** 0.5 + 1e-12).all()


def test_sampler_call_keeps_observed_values(g):
    from types import SimpleNamespace
    f = {"comp_probs": np.exp(g["comp_log_probs"].numpy())[None], "means": g["means"].numpy(), "covs": g["covs"].numpy()}
    model = SimpleNamespace(datamodule=SimpleNamespace(train_data_core=SimpleNamespace(data_file=f)))
    sampler = mu.VAESampler_MoG_Oracle(interpolate_target="conditional_independent_mog", interpolate_alpha=0.5)
    K = 3
    X = g["X"].unsqueeze(1).repeat(1, K, 1)
    M = g["M"].unsqueeze(1).repeat(1, K, 1)
    out = sampler(X, M, model=model, generator=torch.Generator().manual_seed(1))
    assert out.shape == X.shape and out.dtype == torch.float32
    assert torch.equal(out[M], X[M]) and not torch.equal(out[~M], X[~M])
    assert not torch.equal(out[:, 0][~g["M"]], out[:, 1][~g["M"]])           # the K imputations are separate draws
    with pytest.raises(AssertionError):
        mu.VAESampler_MoG_Oracle(clip_imputations=True)


def test_gaussian_fid_score():
    """Frechet distance between Gaussians: closed form for commuting covariances, zero for equal ones"""
    from vgiwae.shared.fid import gaussian_fid_score
    m1, m2 = torch.tensor([0.0, 1.0, -1.0]), torch.tensor([0.5, 1.0, 1.0])
    d1, d2 = torch.tensor([1.0, 4.0, 0.25]), torch.tensor([2.25, 1.0, 0.25])
    got = gaussian_fid_score(m1, torch.diag(d1), m2, torch.diag(d2))
    want = ((m1 - m2) ** 2).sum() + ((d1.sqrt() - d2.sqrt()) ** 2).sum()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    A = torch.randn(4, 4, generator=torch.Generator().manual_seed(0))
    C = A @ A.T + torch.eye(4)
    assert abs(float(gaussian_fid_score(m1[:1].repeat(4), C, m1[:1].repeat(4), C))) < 1e-4
