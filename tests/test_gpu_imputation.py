"""Imputation update (buffer, pseudo-Gibbs, LAIR) and K-sample log-likelihood on the GPU vs the golden
vectors produced by the reference's samplers / test_step.  Through the C ABI."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import core
from oracle.recipes import make_noise

pytestmark = pytest.mark.gpu


def _nets(spec, enc_p, dec_p):
    import gpu_util as G
    return (G.model_dims(spec), G.packed_net(spec.enc, enc_p, requires_grad=False),
            G.packed_net(spec.dec, dec_p, requires_grad=False))


@pytest.fixture(params=[True, False], ids=["one_launch", "separate_launches"])
def pgibbs_path(request):
    """the sampler's iterations as one launch (dmv_pgibbs_update) or as a forward launch and a tail launch each"""
    from demiss_vae_amd import imputation as imp
    imp.ONE_LAUNCH_PGIBBS = request.param
    yield request.param
    imp.ONE_LAUNCH_PGIBBS = True


@pytest.mark.parametrize("name", gu.names("pgibbs_normal"))
def test_pseudo_gibbs_normal(name, pgibbs_path):
    from demiss_vae_amd import imputation as imp
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(dict(g, gen_dist="normal", q_dist="normal", mask_concat=False))
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    clip = bool(g["clip"])
    Xt, M = torch.from_numpy(g["Xt"]).cuda(), torch.from_numpy(g["M"]).cuda()
    out = imp.pseudo_gibbs(dims, enc, dec, Xt, M, num_iterations=1,
                           clip_min=torch.from_numpy(g["clip_min"]) if clip else None,
                           clip_max=torch.from_numpy(g["clip_max"]) if clip else None,
                           eps_z=torch.from_numpy(g["eps_z"])[None].cuda(), noise_x=torch.from_numpy(g["eps_x"])[None].cuda())
    torch.testing.assert_close(out.cpu(), torch.from_numpy(g["X_new"]), rtol=1e-5, atol=2e-6)
    assert torch.equal(out[M], Xt[M])      # observed entries bit-exact


@pytest.mark.parametrize("case", ["toy_clip", "gas", "gas_bern"])
def test_pseudo_gibbs_one_launch_equals_separate_launches(case):
    """dmv_pgibbs_update walks the same tiles through the same chains and takes the same Philox ranges as a
    dmv_vae_forward + dmv_impute_tail pair per iteration: identical imputations, bit for bit, and identical rows in the
    imputation buffer."""
    from demiss_vae_amd import imputation as imp, rng, synth
    from demiss_vae_amd.imputation import ImputationBuffer
    import demiss_vae_amd as dmv
    if case == "toy_clip":
        g = gu.load(gu.names("pgibbs_normal")[0])
        spec, enc_p, dec_p = gu.model_case(dict(g, gen_dist="normal", q_dist="normal", mask_concat=False))
        dims, enc, dec = _nets(spec, enc_p, dec_p)
        X, M = torch.from_numpy(g["Xt"]).cuda(), torch.from_numpy(g["M"]).cuda()
        D = X.shape[-1]
        kw = dict(num_iterations=4, clip_min=torch.full((D,), -0.8), clip_max=torch.full((D,), 0.9))
    else:
        B, K, D, Zd = 203, 5, 8, 16
        bern = case == "gas_bern"
        enc = synth.packed_net(synth.NetShape(2 * D, 2 * Zd, 2, 256), 21, "cuda", requires_grad=False)
        dec = synth.packed_net(synth.NetShape(Zd, D if bern else 2 * D, 2, 256), 22, "cuda", requires_grad=False)
        dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist="bern" if bern else "normal_with_eps", q_dist="normal_with_eps",
                             encoder_use_mis_mask=True)
        X1, M1 = (t.cuda() for t in synth.batch(23, B, D))
        X = X1[:, None, :].repeat(1, K, 1).contiguous()
        X = torch.where(M1[:, None, :], X, torch.randn_like(X))
        if bern:
            X = (X > 0.3).float()
        M = M1[:, None, :].repeat(1, K, 1).contiguous()
        kw = dict(num_iterations=3)
    B, K = X.shape[0], X.shape[1]
    outs, bufs = [], []
    for one in (True, False):
        imp.ONE_LAUNCH_PGIBBS = one
        try:
            rng.set_seed(555)
            buf = ImputationBuffer(X[:, 0].cpu(), K, "cuda")
            idx_imp = buf.idx_imp[:B].clone()
            outs.append(imp.pseudo_gibbs(dims, enc, dec, X, M, buf=buf, idx_imp=idx_imp, **kw))
            bufs.append(buf.data_imp.clone())
        finally:
            imp.ONE_LAUNCH_PGIBBS = True
    assert torch.equal(outs[0], outs[1]) and torch.equal(bufs[0], bufs[1])
    assert not torch.equal(outs[0], X)
    if case != "toy_clip":
        assert torch.equal(outs[0][M], X[M])


def test_pseudo_gibbs_bernoulli(pgibbs_path):
    from demiss_vae_amd import imputation as imp
    g = gu.load("pgibbs_bern_m1_p1")
    spec, enc_p, dec_p = gu.model_case(dict(g, gen_dist="bern_m1_p1", q_dist="normal_with_eps", mask_concat=True))
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    out = imp.pseudo_gibbs(dims, enc, dec, torch.from_numpy(g["Xt"]).cuda(), torch.from_numpy(g["M"]).cuda(),
                           num_iterations=int(g["T"]), eps_z=torch.from_numpy(g["eps_z"]).cuda(),
                           noise_x=torch.from_numpy(g["u_x"]).cuda())
    ref = torch.from_numpy(g["X_new"])
    # a draw flips only if u is within rounding distance of sigmoid(logit): allow a vanishing fraction
    assert (out.cpu() != ref).float().mean().item() < 2e-3


@pytest.mark.parametrize("one_launch", [True, False])
@pytest.mark.parametrize("name", gu.names("lair_"))
def test_lair(name, one_launch):
    """The reference's LAIR update with its recorded noise and draws, through the one-launch update (one importance
    sample per component) and through the separate launches."""
    from demiss_vae_amd import imputation as imp
    g = gu.load(name)
    q = str(g["q_dist"])
    spec, enc_p, dec_p = gu.model_case(dict(g, gen_dist=q, q_dist=q, mask_concat=False))
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    T, r, final = int(g["T"]), int(g["r"]), bool(g["final"])
    idx = [torch.from_numpy(g[f"idx{t}"]) for t in range(T + (1 if final else 0))]
    imp.ONE_LAUNCH_LAIR = one_launch
    imp.LAIR_DEBUG = {}
    try:
        out = imp.lair(dims, enc, dec, torch.from_numpy(g["Xt"]).cuda(), torch.from_numpy(g["M"]).cuda(),
                       num_iterations=T, num_imp_samples=int(g["I"]), num_prior_replenish_proposals=r,
                       resample_final_imps=final, eps_z=[torch.from_numpy(e).cuda() for e in g["eps_z"]],
                       idx=idx, noise_x=[torch.from_numpy(e).cuda() for e in g["eps_x"]])
        seen = "log_w" in imp.LAIR_DEBUG
    finally:
        imp.ONE_LAUNCH_LAIR, imp.LAIR_DEBUG = True, None
    assert seen
    torch.testing.assert_close(out.cpu(), torch.from_numpy(g["X_new"]), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("case", ["small_r1_final_clip", "small_r0", "gas_r1_final", "gas_bern_r2"])
def test_lair_one_launch_update_equals_separate_launches(case):
    """dmv_lair_update against the launches it replaces, both fed from the same Philox stream: every iteration's
    log-weights agree to rounding on the datapoints whose draws agreed so far (a draw flips only when its uniform sits
    within rounding of a bin edge, after which that datapoint's trajectories part), the imputations are identical on
    all but a handful of datapoints, and observed entries come back untouched."""
    from demiss_vae_amd import imputation as imp, rng, synth
    import demiss_vae_amd as dmv
    if case.startswith("small"):
        name = [n for n in gu.names("lair_")][0]
        g = gu.load(name)
        q = str(g["q_dist"])
        spec, enc_p, dec_p = gu.model_case(dict(g, gen_dist=q, q_dist=q, mask_concat=False))
        dims, enc, dec = _nets(spec, enc_p, dec_p)
        X, M = torch.from_numpy(g["Xt"]).cuda(), torch.from_numpy(g["M"]).cuda()
        D = X.shape[-1]
        final, r, clip = case == "small_r1_final_clip", (1 if case == "small_r1_final_clip" else 0), case == "small_r1_final_clip"
        T = 3
    else:
        # UCI-gas shapes (hidden 256), a batch that ends inside a tile
        B, K, D, Zd = 301, 5, 8, 16
        bern = case == "gas_bern_r2"
        gen = "bern" if bern else "normal_with_eps"
        enc = synth.packed_net(synth.NetShape(2 * D if bern else D, 2 * Zd, 2, 256), 11, "cuda", requires_grad=False)
        dec = synth.packed_net(synth.NetShape(Zd, D if bern else 2 * D, 2, 256), 12, "cuda", requires_grad=False)
        dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist=gen, q_dist="normal_with_eps", encoder_use_mis_mask=bern)
        X1, M1 = (t.cuda() for t in synth.batch(13, B, D))
        if bern:
            X1 = (X1 > 0.3).float()
        X = X1[:, None, :].repeat(1, K, 1).contiguous()
        X = torch.where(M1[:, None, :], X, (torch.randn_like(X) > 0.5).float() if bern else torch.randn_like(X))
        M = M1[:, None, :].repeat(1, K, 1).contiguous()
        final, r, clip, T = not bern, (2 if bern else 1), False, 4
    kw = dict(num_iterations=T, num_imp_samples=1, num_prior_replenish_proposals=r, resample_final_imps=final,
              clip_min=torch.full((D,), -1.5) if clip else None, clip_max=torch.full((D,), 1.5) if clip else None)
    outs, lws = [], []
    for one in (True, False):
        imp.ONE_LAUNCH_LAIR, imp.LAIR_DEBUG = one, {}
        try:
            rng.set_seed(977)
            outs.append(imp.lair(dims, enc, dec, X, M, **kw))
            lws.append(imp.LAIR_DEBUG["log_w"])
        finally:
            imp.ONE_LAUNCH_LAIR, imp.LAIR_DEBUG = True, None
    B, K = X.shape[0], X.shape[1]
    Kp = K + r
    a, b = (w.view(T, B, Kp) for w in lws)
    same = torch.ones(B, dtype=torch.bool, device=a.device)          # datapoints whose trajectories have not parted
    for t in range(T):
        err = ((a[t] - b[t]).abs() / (1.0 + b[t].abs())).amax(1)
        assert float(err[same].max()) < 2e-5, (t, float(err[same].max()))
        same &= err < 2e-5
        # (a parted trajectory shows in the next iteration's log-weights)
    assert float(same.float().mean()) > 0.97
    diff_dp = (outs[0] != outs[1]).flatten(1).any(1)
    assert float(diff_dp.float().mean()) < 0.03, float(diff_dp.float().mean())
    obs = X[M].clamp(-1.5, 1.5) if clip else X[M]
    assert torch.equal(outs[0][M], obs) and not torch.equal(outs[0], X)


@pytest.mark.parametrize("final,clip", [(False, False), (True, True)])
def test_lair_one_launch_resampling_equals_two_launches(final, clip):
    """The merged resampling + sampling-tail launch takes the same Philox ranges and does the same arithmetic as the
    draw launch followed by the tail launch: identical imputations, bit for bit, from the same generator state."""
    from demiss_vae_amd import imputation as imp, rng
    name = [n for n in gu.names("lair_")][0]
    g = gu.load(name)
    q = str(g["q_dist"])
    spec, enc_p, dec_p = gu.model_case(dict(g, gen_dist=q, q_dist=q, mask_concat=False))
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    X, M = torch.from_numpy(g["Xt"]).cuda(), torch.from_numpy(g["M"]).cuda()
    D = X.shape[-1]
    kw = dict(num_iterations=3, num_imp_samples=2, num_prior_replenish_proposals=1, resample_final_imps=final,
              clip_min=torch.full((D,), -1.5) if clip else None, clip_max=torch.full((D,), 1.5) if clip else None)
    outs = []
    for merged in (True, False):
        imp.MERGE_RESAMPLE_TAIL = merged
        try:
            rng.set_seed(4321)
            outs.append(imp.lair(dims, enc, dec, X, M, **kw))
        finally:
            imp.MERGE_RESAMPLE_TAIL = True
    assert torch.equal(outs[0], outs[1])
    # observed entries come back unchanged (clamped with everything else where the reference clamps the blended tensor)
    obs = X[M].clamp(-1.5, 1.5) if clip else X[M]
    assert torch.equal(outs[0][M], obs) and not torch.equal(outs[0], X)


def test_lair_device_draws_match_inverse_cdf():
    """Device categorical draws = inverse CDF of softmax(logits) at the given uniforms (oracle restatement)."""
    from demiss_vae_amd import _lib as L
    B, P, K = 37, 53, 9
    logits = make_noise(5, B, P) * 3
    u = torch.from_numpy(np.random.Generator(np.random.PCG64(6)).random((B, K)).astype(np.float32))
    idx = torch.empty((B, K), dtype=torch.int64, device="cuda")
    lg, ug = logits.cuda(), u.cuda()
    L.check(L.lib().dmv_categorical_draw(B, P, K, L.ptr(lg), L.ptr(ug), 0, 0, None, L.ptr(idx), L.stream_ptr()))
    ref = core.inverse_cdf_categorical(logits, u)
    # fp32 vs fp64 cumulative sums can disagree only when u sits on a bin edge
    assert (idx.cpu() != ref).float().mean().item() < 5e-3
    # rows longer than one wave (several strides of the staging loop) and longer than the LDS stage (fallback path)
    for B3, P3, K3 in ((9, 700, 70), (5, 1500, 12)):
        lg3 = make_noise(8, B3, P3) * 2
        u3 = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).random((B3, K3)).astype(np.float32))
        d3 = torch.empty((B3, K3), dtype=torch.int64, device="cuda")
        lg3g, u3g = lg3.cuda(), u3.cuda()
        L.check(L.lib().dmv_categorical_draw(B3, P3, K3, L.ptr(lg3g), L.ptr(u3g), 0, 0, None, L.ptr(d3), L.stream_ptr()))
        ref3 = core.inverse_cdf_categorical(lg3, u3)
        assert ((d3.cpu() - ref3).abs() > 1).float().mean().item() == 0.0       # at most the neighbouring bin, on an edge
        assert (d3.cpu() != ref3).float().mean().item() < 2e-2
    # Philox path: empirical frequencies follow softmax(logits)
    B2, P2, K2 = 4, 6, 64
    lg2 = (make_noise(7, 1, P2) * 1.5).expand(B2, P2).contiguous().cuda()
    counts = torch.zeros(P2)
    for rep in range(200):
        d = torch.empty((B2, K2), dtype=torch.int64, device="cuda")
        L.check(L.lib().dmv_categorical_draw(B2, P2, K2, L.ptr(lg2), None, 42, rep * B2 * K2, None, L.ptr(d), L.stream_ptr()))
        counts += torch.bincount(d.cpu().reshape(-1), minlength=P2).float()
    freq = counts / counts.sum()
    torch.testing.assert_close(freq, torch.softmax(lg2[0].cpu(), 0), atol=0.01, rtol=0)


def test_imputation_buffer():
    from demiss_vae_amd.imputation import ImputationBuffer
    g = gu.load("impbuffer")
    K = int(g["K"])
    buf = ImputationBuffer(torch.from_numpy(g["X"]), K, "cuda")
    assert torch.equal(buf.idx_imp.cpu(), torch.from_numpy(g["idx_imp_table"]))
    np.random.seed(int(6000))
    buf.init_empirical(g["X"], g["M"])
    assert torch.equal(buf.data_imp.cpu(), torch.from_numpy(g["after_init"]))      # same numpy draws, bit-exact
    bidx = torch.from_numpy(g["bidx"]).cuda()
    gx, gi = buf.gather(bidx)
    assert torch.equal(gx.cpu(), torch.from_numpy(g["gather_X"]))
    assert torch.equal(gi.cpu(), torch.from_numpy(g["gather_idx_imp"]))
    X_imp = torch.where(torch.from_numpy(g["gather_M"]).cuda(), gx, torch.from_numpy(g["new_vals"]).cuda())
    buf.scatter(gi, X_imp)
    assert torch.equal(buf.data_imp.cpu(), torch.from_numpy(g["after_set"]))
    # sub-sampled gather with the reference's slot draws
    sub_idx = torch.from_numpy(g["sub_idx_imp"])
    slots = (sub_idx - torch.from_numpy(g["bidx"])[:, None] * K).cuda()
    sx, si = buf.gather(bidx, sampled_i=slots)
    assert torch.equal(si.cpu(), sub_idx) and torch.equal(sx.cpu(), torch.from_numpy(g["sub_X"]))
    # the device's own subset draws: K' distinct slots per datapoint
    s = buf.sample_subset(7, 3)
    assert s.shape == (7, 3) and all(len(set(row.tolist())) == 3 for row in s.cpu())


@pytest.mark.parametrize("name", gu.names("loglik_"))
def test_loglik(name):
    from demiss_vae_amd import imputation as imp
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(g)
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    X, M = torch.from_numpy(g["X"]).cuda(), torch.from_numpy(g["M"]).cuda()
    B, K, C = int(g["B"]), int(g["K"]), int(g["C"])
    seed = int(g["seed"])
    if str(g["kind"]) == "standard":
        eps = gu.eps_of(g, 1, K, B, spec.Zd)[0].cuda()
        val, per_b = imp.estimate_loglik(dims, enc, dec, X, M, num_importance_samples=K, chunk=1000, eps=eps)
    else:
        comp = torch.from_numpy(g["comp_idx"])
        if g["eps"].size:
            eps = torch.from_numpy(g["eps"])
        else:
            eps_all = make_noise(seed + 3, 1, K, B, C, spec.Zd)
            eps = torch.gather(eps_all, 3, comp[..., None, None].expand(1, K, B, 1, spec.Zd)).squeeze(3)
        val, per_b = imp.estimate_loglik(dims, enc, dec, X, M, num_importance_samples=K, chunk=1000, eps=eps[0].cuda(),
                                         comp_idx=comp[0].cuda())
    # north-star tolerance for final log-likelihoods: 0.01 nats; the kernels are far inside it
    assert abs(val.item() - float(g["est_loglik"])) < 1e-4 * max(1.0, abs(float(g["est_loglik"])))
    if "per_point" in g:            # the K = 50 000 case of the test step also pins every test point's estimate
        torch.testing.assert_close(per_b.cpu().double(), torch.from_numpy(g["per_point"]).double(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", [n for n in gu.names("reparam_") if str(gu.load(n)["cls"]) != "VAE" and int(gu.load(n)["nz"]) == 1])
def test_loglik_reparametrised_mixture_encoder(name):
    """The streaming estimator with a reparametrised-mixture encoder: the reference's `standard` refit class rsamples
    the real mixture (test/refit_encoder_using_iwae_and_estimate_loglik.py:193-198), which in the forward direction is
    ancestral sampling.  With the reference's recorded component draws and noise the streamed value must equal the
    reference's IWAE bound of the same case (a proposal built from component 0 alone would be off by whole nats)."""
    import dataclasses
    import gpu_util as G
    from demiss_vae_amd import imputation as imp
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(g)
    q = "reparam_with_eps" if bool(g["with_eps"]) else "reparam"
    dims = dataclasses.replace(G.model_dims(spec), q_dist=q)
    enc = G.packed_net(spec.enc, enc_p, requires_grad=False)
    dec = G.packed_net(spec.dec, dec_p, requires_grad=False)
    X, M = (t.cuda() for t in gu.batch_of(g))
    eps, comp = torch.from_numpy(g["eps"]), torch.from_numpy(g["comp_idx"])          # [1, I, B, C, Zd], [1, I, B]
    I = int(g["I"])
    own = torch.gather(eps, 3, comp[..., None, None].expand(*comp.shape, 1, spec.Zd)).squeeze(3)
    val, _ = imp.estimate_loglik(dims, enc, dec, X, M, num_importance_samples=I, chunk=3, eps=own[0].cuda(),
                                 comp_idx=comp[0].cuda())
    np.testing.assert_allclose(val.item(), float(g["bound"]), rtol=1e-5)


def test_loglik_philox_large_K_consistency():
    """K = 50 000 with device noise: two disjoint halves of the samples combine (max / sum-exp) to the estimate
    over all of them, and agree with each other within Monte-Carlo error."""
    import math
    from demiss_vae_amd import imputation as imp, rng
    g = gu.load("loglik_standard_K64")
    spec, enc_p, dec_p = gu.model_case(g)
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    X, M = torch.from_numpy(g["X"]).cuda(), torch.from_numpy(g["M"]).cuda()
    rng.set_seed(7)
    K = 50000
    m1, s1 = imp.estimate_loglik(dims, enc, dec, X, M, num_importance_samples=K // 2, return_state=True)
    m2, s2 = imp.estimate_loglik(dims, enc, dec, X, M, num_importance_samples=K // 2, return_state=True)
    m = torch.maximum(m1, m2)
    tot = m + torch.log(s1 * torch.exp(m1 - m) + s2 * torch.exp(m2 - m)) - math.log(K)
    a = m1 + torch.log(s1) - math.log(K // 2)
    b = m2 + torch.log(s2) - math.log(K // 2)
    assert torch.allclose(tot, torch.logaddexp(a, b) - math.log(2.0), atol=1e-5)
    # untrained networks give heavy-tailed weights, so two independent 25k-sample estimates agree only loosely
    assert (a - b).abs().mean().item() < 1.0
    assert tot.mean().item() > float(g["est_loglik"]) - 0.5            # more samples: a tighter bound in expectation


def test_lair_row_map_bit_exact():
    """rows of the resampled proposals: the index arithmetic of the reference's gather (vae_imputation.py:193-198 after
    :521-524), bit for bit against the tensor formula"""
    from demiss_vae_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    for B, K, r, I in ((7, 5, 0, 1), (64, 5, 1, 3), (300, 3, 2, 2)):
        Kp = K + r
        draws = torch.randint(0, I * Kp, (B, Kp), generator=g).cuda()
        out = torch.empty(B * K, dtype=torch.int64, device="cuda")
        L.check(L.lib().dmv_lair_row_map(B, K, Kp, L.ptr(draws), L.ptr(out), L.stream_ptr()))
        d = draws[:, :K]
        ref = (((d // Kp) * B + torch.arange(B, device="cuda")[:, None]) * Kp + d % Kp).reshape(-1)
        assert torch.equal(out, ref)
