"""HIP objective pipelines vs the golden vectors produced by the reference (bounds rel 1e-5, gradients
rel 1e-4 with a norm-wise floor), through the C ABI."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

RT_B, RT_G, AT_G = 1e-5, 1e-4, 2e-6


def _nets(spec, enc_p, dec_p):
    import gpu_util as G
    return G.model_dims(spec), G.packed_net(spec.enc, enc_p), G.packed_net(spec.dec, dec_p)


def _check(g, spec, enc, dec):
    import gpu_util as G
    assert gu.check_grads(g, "genc", G.grads_of(enc, spec.enc), RT_G, AT_G) > 0
    assert gu.check_grads(g, "gdec", G.grads_of(dec, spec.dec), RT_G, AT_G) > 0


@pytest.mark.parametrize("name", gu.names("iwae_"))
def test_iwae(name):
    import demiss_vae_amd as dmv
    import gpu_util as G
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(g)
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    X, M = gu.batch_of(g)
    B, I, nz = int(g["B"]), int(g["I"]), int(g["nz"])
    eps = gu.eps_of(g, nz, I, B, spec.Zd)
    seed = int(g["seed"])
    ek = G.keep_to_dev(gu.dropout_masks(seed + 4, spec.enc, (B,)))
    dk = G.keep_to_dev(gu.dropout_masks(seed + 5, spec.dec, (nz, I, B)))
    bound, per_b = dmv.estimate_iwelbo(dims, enc, dec, X.cuda(), M.cuda(), num_latent_samples=nz,
                                       num_importance_samples=I, STL=bool(g["STL"]), DREG=bool(g["DREG"]),
                                       eps=eps.cuda(), enc_keep=ek, dec_keep=dk)
    (-bound).backward()
    np.testing.assert_allclose(bound.item(), float(g["bound"]), rtol=RT_B)
    _check(g, spec, enc, dec)


@pytest.mark.parametrize("name", gu.names("vae_"))
def test_vae(name):
    import demiss_vae_amd as dmv
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(g)
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    X, M = gu.batch_of(g)
    bound, _ = dmv.estimate_elbo(dims, enc, dec, X.cuda(), M.cuda(), num_latent_samples=int(g["nz"]),
                                 kl_analytic=bool(g["kl_analytic"]), STL=bool(g["STL"]), eps=gu.eps_of(g).cuda())
    (-bound).backward()
    np.testing.assert_allclose(bound.item(), float(g["bound"]), rtol=RT_B)
    _check(g, spec, enc, dec)


@pytest.mark.parametrize("name", gu.names("miwae_") + gu.names("mvae_"))
def test_multiple(name):
    import demiss_vae_amd as dmv
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(g)
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    X, M = gu.batch_of(g)
    B, I, nz, C = int(g["B"]), int(g["I"]), int(g["nz"]), int(g["C"])
    if name.startswith("miwae"):
        import gpu_util as G
        seed = int(g["seed"])
        eps = gu.eps_of(g, nz, I, B, C, spec.Zd)
        ek = G.keep_to_dev(gu.dropout_masks(seed + 4, spec.enc, (B,)) or None)      # miniboone: dropout 0.5 in both networks
        dk = G.keep_to_dev(gu.dropout_masks(seed + 5, spec.dec, (nz, I, B, C)) or None)
        bound, _ = dmv.estimate_multiple_iwelbo(dims, enc, dec, X.cuda(), M.cuda(), num_latent_samples=nz,
                                                num_importance_samples_for_each_component=I, STL=bool(g["STL"]),
                                                looser=bool(g["looser"]), eps=eps.cuda(), enc_keep=ek, dec_keep=dk)
    else:
        eps = gu.eps_of(g, nz, B, C, spec.Zd)
        bound, _ = dmv.estimate_multiple_elbo(dims, enc, dec, X.cuda(), M.cuda(), num_latent_samples=nz,
                                              STL=bool(g["STL"]), eps=eps.cuda())
    (-bound).backward()
    np.testing.assert_allclose(bound.item(), float(g["bound"]), rtol=RT_B)
    _check(g, spec, enc, dec)


@pytest.mark.parametrize("name", gu.names("mvb_"))
def test_mvb(name):
    import demiss_vae_amd as dmv
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(g)
    dims, enc, dec = _nets(spec, enc_p, dec_p)
    Xt, Mk = gu.imputed_batch(g)
    B, K, nz = int(g["B"]), int(g["K"]), int(g["nz"])
    eps = gu.eps_of(g, nz, B, K, spec.Zd).cuda()
    STL = bool(g["STL"])
    if str(g["objective"]) == "mvb2":
        gen_obj, enc_obj = dmv.mvb2_objectives(dims, enc, dec, Xt.cuda(), Mk.cuda(), num_latent_samples=nz, STL=STL,
                                               kl_analytic=bool(g["kl_analytic"]), eps=eps)
        torch.testing.assert_close(enc_obj.detach().cpu(), torch.from_numpy(g["enc_obj"]), rtol=RT_B, atol=1e-5)
        # the two backward calls of the manual optimisation step, with retain_graph on the first
        (-gen_obj.mean()).backward(retain_graph=True, inputs=dec.params())
        (-enc_obj.mean()).backward(inputs=enc.params())
    else:
        gen_obj = dmv.cvi_objective(dims, enc, dec, Xt.cuda(), Mk.cuda(), num_latent_samples=nz, STL=STL, eps=eps)
        (-gen_obj.mean()).backward()
    torch.testing.assert_close(gen_obj.detach().cpu(), torch.from_numpy(g["gen_obj"]), rtol=RT_B, atol=1e-5)
    _check(g, spec, enc, dec)


def test_philox_noise_statistics():
    """Without explicit eps the latent noise comes from the device Philox stream: check its moments and that
    the bound it gives agrees with the oracle evaluated on the noise the kernel reports having used."""
    import ctypes as C
    import demiss_vae_amd as dmv
    from demiss_vae_amd import _lib as L, rng
    rng.set_seed(1234)
    cfg = L.DmvObjCfg()
    N, Zd = 1 << 16, 16
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = N, 1, 1, 1, 1, Zd
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off = 2 * Zd, 0, Zd
    enc = torch.zeros(N, 2 * Zd, device="cuda")
    Z = torch.empty(N, Zd, device="cuda")
    eps = torch.empty(N, Zd, device="cuda")
    L.check(L.lib().dmv_latent_sample(C.byref(cfg), L.ptr(enc), None, 99, 0, None, Zd, None, L.ptr(Z), L.ptr(eps), L.stream_ptr()))
    e = eps.double().cpu()
    assert abs(e.mean().item()) < 5e-3 and abs(e.std().item() - 1) < 5e-3
    assert abs((e ** 4).mean().item() - 3) < 0.1            # kurtosis of a standard normal
    assert abs(torch.corrcoef(torch.stack([e[:-1, 0], e[1:, 0]]))[0, 1].item()) < 0.02
    sp0 = float(torch.nn.functional.softplus(torch.zeros(())))
    torch.testing.assert_close(Z.cpu(), (eps * sp0).cpu(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("engine", ["fused", "caller_networks"])
@pytest.mark.parametrize("name", gu.names("reparam_"))
def test_reparametrised_mixture(name, engine):
    """Mixture encoder sampled ancestrally with implicit reparameterisation gradients (the reparametrised_mixture*
    families; vgiwae/shared/reparametrized_mixture_same_family.py), with STL and (the *_live fixtures) without, through
    both engines: the fused residual
    MLP pipeline and the caller-network one (here the same networks as plain torch functions).  Encoder-side
    tolerance as in the oracle test: the reference's fp32 triangular solve is itself only good to ~1e-3."""
    import dataclasses
    import demiss_vae_amd as dmv
    import gpu_util as G
    from demiss_vae_amd import external
    from oracle import core
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(g)
    q = "reparam_with_eps" if bool(g["with_eps"]) else "reparam"
    dims = dataclasses.replace(G.model_dims(spec), q_dist=q)
    X, M = (t.cuda() for t in gu.batch_of(g))
    eps, comp = torch.from_numpy(g["eps"]).cuda(), torch.from_numpy(g["comp_idx"]).cuda()
    nz, I, is_vae = int(g["nz"]), int(g["I"]), str(g["cls"]) == "VAE"
    stl = bool(g.get("STL", True))
    if engine == "fused":
        enc, dec = G.packed_net(spec.enc, enc_p), G.packed_net(spec.dec, dec_p)
        fn, a, b = dmv, enc, dec
    else:
        ep = {k: v.cuda().requires_grad_(True) for k, v in enc_p.items()}
        dp = {k: v.cuda().requires_grad_(True) for k, v in dec_p.items()}
        fn = external
        a = lambda X2, M2: core.mlp_forward(ep, torch.cat([X2, M2.to(X2.dtype)], -1), spec.enc)
        b = lambda Z: core.mlp_forward(dp, Z, spec.dec)
    if is_vae:
        bound, _ = fn.estimate_elbo(dims, a, b, X, M, num_latent_samples=nz, kl_analytic=False, STL=stl, eps=eps, comp_idx=comp)
    else:
        bound, _ = fn.estimate_iwelbo(dims, a, b, X, M, num_latent_samples=nz, num_importance_samples=I, STL=stl, eps=eps,
                                      comp_idx=comp)
    (-bound).backward()
    np.testing.assert_allclose(bound.item(), float(g["bound"]), rtol=RT_B)
    if engine == "fused":
        genc, gdec = G.grads_of(enc, spec.enc), G.grads_of(dec, spec.dec)
    else:
        genc, gdec = {k: v.grad for k, v in ep.items()}, {k: v.grad for k, v in dp.items()}
    assert gu.check_grads(g, "genc", genc, 2e-3, AT_G, floor=2e-4) > 0
    assert gu.check_grads(g, "gdec", gdec, RT_G, AT_G) > 0
    # device-drawn components and noise: finite and in the neighbourhood of the explicit-noise bound
    if engine == "fused":
        b2, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=4, num_importance_samples=8, STL=True)
        assert torch.isfinite(b2) and abs(b2.item() - float(g["bound"])) < 0.5 * abs(float(g["bound"])) + 5


def test_latent_sample_vector_and_scalar_kernels_agree():
    """the 16-byte path of dmv_latent_sample (aligned buffers) draws the same values as the scalar path (forced here by a
    misaligned noise output buffer), with and without explicit noise and for mixture components picked per row"""
    import ctypes as C
    from demiss_vae_amd import _lib as L
    B, G, S, Zd = 37, 5, 3, 8
    N = S * B * G
    cfg = L.DmvObjCfg()
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, G, 1, S, 1, Zd
    ldE = 2 * G * Zd + G
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, Zd, G * Zd, 2 * G * Zd
    enc = torch.randn(B, ldE, device="cuda")
    outs = []
    for shift in (0, 1):
        Z = torch.empty(N, Zd, device="cuda")
        buf = torch.empty(N * Zd + 4, device="cuda")
        used = buf[shift:shift + N * Zd]
        L.check(L.lib().dmv_latent_sample(C.byref(cfg), L.ptr(enc), None, 5, 64, None, Zd, None, L.ptr(Z), L.ptr(used), L.stream_ptr()))
        outs.append((Z.clone(), used.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    eps = outs[0][1].reshape(N, Zd).contiguous()
    Z2 = torch.empty(N, Zd, device="cuda")
    L.check(L.lib().dmv_latent_sample(C.byref(cfg), L.ptr(enc), L.ptr(eps), 0, 0, None, Zd, None, L.ptr(Z2), None, L.stream_ptr()))
    assert torch.equal(Z2, outs[0][0])
    mu = enc[:, :G * Zd].reshape(B, G, Zd)
    sg = torch.nn.functional.softplus(enc[:, G * Zd:2 * G * Zd]).reshape(B, G, Zd)
    want = mu.unsqueeze(0) + sg.unsqueeze(0) * eps.reshape(S, B, G, Zd)
    torch.testing.assert_close(Z2.reshape(S, B, G, Zd), want, rtol=1e-5, atol=1e-6)


def test_no_grad_evaluation_saves_nothing_and_accepts_dropout_networks():
    """Under torch.no_grad() (validation) the pipelines run in inference mode: no saved activations, and networks with
    dropout need no masks (the parameters still have requires_grad=True, which autograd reports to the function)."""
    import demiss_vae_amd as dmv
    from oracle.recipes import NetSpec, make_batch, make_params
    import gpu_util as G
    enc_s, dec_s = NetSpec(16, 32, 1, 64, dropout=0.5), NetSpec(16, 16, 1, 64, dropout=0.5)
    enc, dec = G.packed_net(enc_s, make_params(enc_s, 1)), G.packed_net(dec_s, make_params(dec_s, 2))
    dims = dmv.ModelDims(D=8, Zd=16, encoder_use_mis_mask=True)
    X, M = (t.cuda() for t in make_batch(3, 40, 8))
    eps = torch.randn(1, 5, 40, 16, device="cuda")
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        b0, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_importance_samples=5, eps=eps)
    assert not b0.requires_grad and torch.isfinite(b0)
    with pytest.raises(RuntimeError):            # training mode without masks is an error, not a silent no-dropout pass
        dmv.estimate_iwelbo(dims, enc, dec, X, M, num_importance_samples=5, eps=eps)
    plain_enc = G.packed_net(NetSpec(16, 32, 1, 64), make_params(enc_s, 1))
    plain_dec = G.packed_net(NetSpec(16, 16, 1, 64), make_params(dec_s, 2))
    b1, _ = dmv.estimate_iwelbo(dims, plain_enc, plain_dec, X, M, num_importance_samples=5, eps=eps)
    torch.testing.assert_close(b0, b1.detach())   # inference with dropout networks = the same networks without dropout


def test_gradients_in_permanent_slabs_equal_fresh_ones():
    """PackedNet.use_flat_grads: the backward pass writes the parameter gradients into one permanent buffer per network
    and autograd keeps those views as .grad (what the data-parallel exchange reduces in place).  Same values as with
    fresh tensors; a second backward pass without zero_grad adds to them instead of aliasing the buffer."""
    import demiss_vae_amd as dmv
    g = gu.load("iwae_stl_normal_with_eps_mc")
    spec, enc_p, dec_p = gu.model_case(g)
    X, M = gu.batch_of(g)
    B, I, nz = int(g["B"]), int(g["I"]), int(g["nz"])
    eps = gu.eps_of(g, nz, I, B, spec.Zd).cuda()

    def run(flat, passes=1):
        dims, enc, dec = _nets(spec, enc_p, dec_p)
        ready = []
        if flat:
            enc.use_flat_grads()
            dec.use_flat_grads()
            dec.on_grads_ready = lambda net: ready.append(net)
        for _ in range(passes):
            bound, _ = dmv.estimate_iwelbo(dims, enc, dec, X.cuda(), M.cuda(), num_latent_samples=nz, num_importance_samples=I,
                                           STL=True, eps=eps)
            (-bound).backward()
        return enc, dec, ready

    e0, d0, _ = run(False)
    e1, d1, ready = run(True)
    assert e1.grads_live_in_slab() and d1.grads_live_in_slab() and ready == [d1]
    for a, b in zip(e0.params() + d0.params(), e1.params() + d1.params()):
        assert torch.equal(a.grad, b.grad)
    _check(g, spec, e1, d1)
    # the next step (after zero_grad) lands in the same memory
    where = [p.grad.data_ptr() for p in d1.params()]
    for p in e1.params() + d1.params():
        p.grad = None
    bound, _ = dmv.estimate_iwelbo(_nets(spec, enc_p, dec_p)[0], e1, d1, X.cuda(), M.cuda(), num_latent_samples=nz,
                                   num_importance_samples=I, STL=True, eps=eps)
    (-bound).backward()
    assert [p.grad.data_ptr() for p in d1.params()] == where and d1.grads_live_in_slab()
    # accumulation over two passes: twice the gradient
    e2, d2, _ = run(True, passes=2)
    for a, b in zip(e0.params() + d0.params(), e2.params() + d2.params()):
        torch.testing.assert_close(b.grad, 2 * a.grad, rtol=1e-6, atol=1e-7)


def _latent_mix_case(seed, B, G, S, Zd, strat, own_live, mix_live, with_opt):
    """Random inputs for dmv_obj_backward_latent with a mixture term, and the float64 autograd gradient of
         L = sum_n dZ_n . z_n + w_n (a_p log p(z_n) + a_o log q_own(z_n)) + a_m w_n log q_mix(z_n) + a_w sum dlogit . logits
    with z_n = mu + sigma eps (zscale multiplies the gradient through z), q parameters inside log q detached unless live."""
    import ctypes as C
    from demiss_vae_amd import _lib as L
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
    N = S * B * G
    if strat:
        ld = 2 * G * Zd + G + 3
        enc = rn(B, ld) * 0.7
        sb, sg_, raw_off, logit_off = ld, Zd, G * Zd, 2 * G * Zd
        mu_of = lambda e: e[:, :G * Zd].reshape(B, G, Zd)
        raw_of = lambda e: e[:, G * Zd:2 * G * Zd].reshape(B, G, Zd)
    else:
        ldE = 2 * Zd + 5
        enc = rn(B * G, ldE) * 0.7
        sb, sg_, raw_off, logit_off = G * ldE, ldE, Zd, 0
        mu_of = lambda e: e[:, :Zd].reshape(B, G, Zd)
        raw_of = lambda e: e[:, Zd:2 * Zd].reshape(B, G, Zd)
    eps = rn(S, B, G, Zd)
    dZ = rn(S, B, G, Zd) * 0.3
    w = torch.rand(S, B, G, generator=gen, dtype=torch.float64) + 0.2
    zs = torch.rand(S, B, G, generator=gen, dtype=torch.float64) + 0.5
    gb = torch.rand(B, generator=gen, dtype=torch.float64) + 0.5
    dlogit = rn(B, G)
    a_p, a_o, a_m, a_w, gsc = 0.8, -0.6, -1.3, 0.9, 0.7
    QE = 1e-5

    e = enc.clone().requires_grad_(True)
    mu, sig = mu_of(e), torch.nn.functional.softplus(raw_of(e)) + QE
    z_path = mu + sig * eps                                   # [S,B,G,Zd]
    z = z_path.detach()
    logn = lambda zz, m, s: (-(zz - m) ** 2 / (2 * s * s) - torch.log(s) - 0.5 * np.log(2 * np.pi)).sum(-1)
    mu_o, sig_o = (mu, sig) if own_live else (mu.detach(), sig.detach())
    mu_m, sig_m = (mu, sig) if mix_live else (mu.detach(), sig.detach())
    if strat:
        logpi = torch.log_softmax(e[:, logit_off:logit_off + G], -1)
        logpi_m = logpi if mix_live else logpi.detach()
    else:
        logpi = None
        logpi_m = torch.full((B, G), -np.log(G), dtype=torch.float64)
    # the density terms as functions of a free z, to take d/dz at the sample and (when live) d/dparams
    zf = z.clone().requires_grad_(True)
    comp = logn(zf[:, :, :, None, :], mu_m[None, :, None], sig_m[None, :, None])       # [S,B,G,C]
    qmix = torch.logsumexp(logpi_m[None, :, None, :] + comp, -1)
    prior = (-0.5 * zf * zf - 0.5 * np.log(2 * np.pi)).sum(-1)
    qown = logn(zf, mu_o[None], sig_o[None])
    wg = w * gb[None, :, None] * gsc if with_opt else w * gsc
    dens = (wg * (a_p * prior + a_o * qown + a_m * qmix)).sum()
    gz, = torch.autograd.grad(dens, zf, retain_graph=True)
    through_z = (dZ + gz.detach()) * (zs[..., None] if with_opt else 1.0)
    total = dens + (through_z * z_path).sum()
    if strat:      # dlogit is a gradient with respect to the raw logits
        total = total + a_w * gsc * ((gb[:, None] if with_opt else 1.0) * dlogit * e[:, logit_off:logit_off + G]).sum()
    total.backward()
    ref = e.grad.clone()

    dev = "cuda"
    f32 = lambda t: t.to(torch.float32).to(dev).contiguous()
    cfg = L.DmvObjCfg()
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, G, 1, S, 4, Zd
    cfg.gen_dist, cfg.q_eps, cfg.x_per_component = L.GEN["normal"], 1, 0
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = sb, sg_, raw_off, logit_off
    p = L.DmvLatentBwd()
    p.base_prior, p.base_qown, p.base_qmix, p.base_w, p.kl_coef = a_p, a_o, a_m, a_w, 0.0
    p.own_live, p.mix_live, p.use_zscale = int(own_live), int(mix_live), int(with_opt)
    p.mix_kind = L.Q_STRATIFIED if strat else L.Q_IMPUTATION
    enc_d, Z_d, eps_d, dZ_d = f32(enc), f32(z.reshape(N, Zd)), f32(eps.reshape(N, Zd)), f32(dZ.reshape(N, Zd))
    w_d, zs_d, gb_d, dl_d = f32(w.reshape(N)), f32(zs.reshape(N)), f32(gb), f32(dlogit)
    comp_d, qmix_d = f32(comp.detach().reshape(N, G)), f32(qmix.detach().reshape(N))
    gs_d = torch.tensor(gsc, device=dev)
    out = torch.full_like(enc_d, float("nan"))
    lib, st = L.lib(), L.stream_ptr()
    L.check(lib.dmv_obj_backward_latent(C.byref(cfg), C.byref(p), L.ptr(enc_d), L.ptr(Z_d), L.ptr(eps_d), Zd, L.ptr(dZ_d), Zd,
                                        L.ptr(w_d), L.ptr(gb_d) if with_opt else None, L.ptr(zs_d) if with_opt else None,
                                        L.ptr(comp_d), L.ptr(qmix_d), L.ptr(dl_d) if strat else None, None, L.ptr(gs_d),
                                        L.ptr(out), st))
    torch.cuda.synchronize()
    return out.cpu().double(), ref


@pytest.mark.parametrize("B,G,S,Zd,strat", [
    (37, 5, 1, 16, True),        # hepmass-like: three datapoints per workgroup, ragged last workgroup
    (9, 5, 3, 32, True),         # several sample groups per component
    (50, 5, 2, 2, False),        # toy imputation mixture: more responsibilities than threads per datapoint
    (7, 25, 1, 2, False),        # 25 imputations
    (5, 5, 1, 50, False),        # image latent width
    (3, 64, 1, 3, True),         # most components the library takes
    (3, 8, 2, 130, False),       # G * Zd above 1024: the thread-per-entry kernel
])
@pytest.mark.parametrize("own_live,mix_live,with_opt", [(False, False, True), (True, True, False)])
def test_latent_backward_with_mixture_against_autograd(B, G, S, Zd, strat, own_live, mix_live, with_opt):
    got, ref = _latent_mix_case(B * 131 + G, B, G, S, Zd, strat, own_live, mix_live, with_opt)
    assert torch.isfinite(got).all()                     # every entry written (padding zeroed)
    tol = 1e-4 * ref.abs() + 2e-5 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bad.any(), (int(bad.sum()), float((got - ref).abs().max()), float(ref.abs().max()))
    assert (got[ref == 0] == 0).all()


def test_dmis_known_answer_through_the_kernels():
    """The reference's own known-answer case for the DMIS log-weights (vgiwae/shared/iwae.py:135-181: NZ, I, B, K, D, Zd =
    3, 6, 7, 5, 4, 2, arbitrary decoder and encoder distribution parameters) through dmv_obj_terms with the imputation
    mixture.  The fixture holds scales; the kernels take pre-softplus values, inverted here in float64."""
    import ctypes as C
    from demiss_vae_amd import _lib as L
    g = gu.load("dmis_known_answer")
    NZ, I, B, K, Zd = g["Z"].shape
    D = g["X"].shape[-1]
    N = NZ * I * B * K
    inv_softplus = lambda s: np.log(np.expm1(s.astype(np.float64)))
    dev = "cuda"
    ldD, ldE = 2 * D + 3, 2 * Zd + 1
    dec = torch.zeros(N, ldD)
    dec[:, :D] = torch.from_numpy(g["gen_mean"]).reshape(N, D)
    dec[:, D:2 * D] = torch.from_numpy(inv_softplus(g["gen_scale"])).float().reshape(N, D)
    enc = torch.zeros(B * K, ldE)
    enc[:, :Zd] = torch.from_numpy(g["var_mean"]).reshape(B * K, Zd)
    enc[:, Zd:2 * Zd] = torch.from_numpy(inv_softplus(g["var_scale"])).float().reshape(B * K, Zd)
    X = torch.from_numpy(g["X"]).contiguous()
    M = torch.ones(B, K, D, dtype=torch.uint8)
    cfg = L.DmvObjCfg()
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, K, NZ, I, D, Zd
    cfg.gen_dist, cfg.q_eps, cfg.x_per_component = L.GEN["normal"], 0, 1
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = K * ldE, ldE, Zd, 0
    dec_d, enc_d, X_d, M_d = dec.to(dev), enc.to(dev), X.to(dev), M.to(dev)
    Z_d = torch.from_numpy(g["Z"]).reshape(N, Zd).contiguous().to(dev)
    lib, st = L.lib(), L.stream_ptr()
    terms = torch.empty(5, N, device=dev)
    L.check(lib.dmv_obj_terms(C.byref(cfg), L.Q_IMPUTATION, K, L.ptr(dec_d), ldD, L.ptr(X_d), L.ptr(M_d), L.ptr(Z_d), Zd,
                              L.ptr(enc_d), L.ptr(terms), None, st))
    want = torch.from_numpy(g["log_w"]).reshape(N)
    got = (terms[0] + terms[2] - terms[4]).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # the one-launch forward forms the same log-weights on its way to the bound
    log_w, per_b = torch.empty(N, device=dev), torch.empty(B, device=dev)
    f = L.DmvObjFwd()
    f.q_kind, f.mix_C, f.kind = L.Q_IMPUTATION, K, L.BOUND_LSE
    f.log_w, f.bound = L.addr(log_w), L.addr(per_b)
    L.check(lib.dmv_obj_forward(C.byref(cfg), C.byref(f), L.ptr(dec_d), ldD, L.ptr(X_d), L.ptr(M_d), L.ptr(Z_d), Zd,
                                L.ptr(enc_d), st))
    torch.testing.assert_close(log_w.cpu(), want, rtol=1e-5, atol=1e-5)


def test_more_samples_per_datapoint_than_the_one_launch_forward_holds():
    """nz * I just above what dmv_obj_forward stages in LDS: the pipeline falls back to terms + bound launches; bound and
    gradients against the oracle on the same noise."""
    import demiss_vae_amd as dmv
    import gpu_util as G
    from oracle import core
    from oracle.recipes import NetSpec, make_params, make_batch, make_noise
    enc_s, dec_s = NetSpec(12, 8, 1, 32), NetSpec(4, 12, 1, 32)
    spec = core.ModelSpec(D=6, Zd=4, enc=enc_s, dec=dec_s, encoder_use_mis_mask=True)
    B, I = 3, 16001
    ep = {k: v.clone().requires_grad_(True) for k, v in make_params(enc_s, 21).items()}
    dp = {k: v.clone().requires_grad_(True) for k, v in make_params(dec_s, 22).items()}
    X, M = make_batch(23, B, 6)
    eps = make_noise(24, 1, I, B, 4)
    ref, ref_b = core.estimate_iwelbo(spec, ep, dp, X, M, eps, STL=True)
    (-ref).backward()
    dims, enc, dec = _nets(spec, {k: v.detach() for k, v in ep.items()}, {k: v.detach() for k, v in dp.items()})
    got, per_b = dmv.estimate_iwelbo(dims, enc, dec, X.cuda(), M.cuda(), num_latent_samples=1, num_importance_samples=I,
                                     STL=True, eps=eps.cuda())
    (-got).backward()
    np.testing.assert_allclose(got.item(), ref.item(), rtol=RT_B)
    torch.testing.assert_close(per_b.cpu(), ref_b.detach(), rtol=1e-5, atol=1e-5)
    for s_, net, ref_p in ((enc_s, enc, ep), (dec_s, dec, dp)):
        for n, w, b in zip(s_.layer_names(), net.weights, net.biases):
            for name, t in ((n + ".weight", w), (n + ".bias", b)):
                g_ref = ref_p[name].grad
                tol = 1e-4 * g_ref.abs() + 2e-5 * g_ref.abs().max()
                assert bool(((t.grad.cpu() - g_ref).abs() <= tol).all()), name
    # no-grad evaluation of the same shape (the validation path)
    with torch.no_grad():
        v, _ = dmv.estimate_iwelbo(dims, enc, dec, X.cuda(), M.cuda(), num_latent_samples=1, num_importance_samples=I,
                                   STL=True, eps=eps.cuda())
    np.testing.assert_allclose(v.item(), ref.item(), rtol=RT_B)


def _forward_buffers(dims, plan, enc, dec, X, M, eps, training):
    from demiss_vae_amd import objectives as O
    from demiss_vae_amd import rng
    rng.set_seed(77)
    c = O._forward_core(dims, plan, enc, dec, X, M, eps, None, None, training)
    torch.cuda.synchronize()
    return c


@pytest.mark.parametrize("case", ["iwae_gas", "iwae_ragged", "miwae_hepmass", "demiss_toy", "demiss_mask_concat", "iwae_h64"])
@pytest.mark.parametrize("training,explicit", [(True, True), (True, False), (False, False)])
def test_one_launch_forward_equals_three_launches(case, training, explicit):
    """dmv_vae_forward (encoder, sampling, decoder in one launch) leaves every buffer - encoder outputs, latents, noise,
    decoder outputs, both networks' saved activations - bit for bit as dmv_mlp_forward + dmv_latent_sample +
    dmv_mlp_forward do, with explicit noise and with the device Philox stream."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd import objectives as O
    from demiss_vae_amd import _lib as L
    from demiss_vae_amd import synth
    import ctypes as C
    shapes = {
        #                      B    D  Zd  I  K/C  enc net              dec net              q, demiss, mask-concat
        "iwae_gas":           (512, 8, 16, 5, 1, (16, 32, 2, 256), (16, 16, 2, 256), "normal_with_eps", False, True),
        "iwae_ragged":        (37, 8, 16, 3, 1, (16, 32, 2, 256), (16, 16, 2, 256), "normal", False, True),
        "miwae_hepmass":      (100, 21, 16, 2, 5, (42, 165, 2, 256), (16, 42, 2, 256), "stratified_with_eps", False, True),
        "demiss_toy":         (64, 5, 2, 1, 5, (5, 4, 3, 200), (2, 10, 3, 200), "normal", True, False),
        "demiss_mask_concat": (21, 6, 3, 2, 7, (12, 6, 1, 96), (3, 12, 1, 96), "normal_with_eps", True, True),
        "iwae_h64":           (50, 8, 4, 4, 1, (8, 8, 2, 64), (4, 16, 2, 64), "normal_with_eps", False, False),
    }
    B, D, Zd, I, G, es, ds, q, demiss, mc = shapes[case]
    enc = synth.packed_net(synth.NetShape(*es), 1, "cuda")
    dec = synth.packed_net(synth.NetShape(*ds), 2, "cuda")
    dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist="normal", q_dist=q, C=G if q.startswith("strat") else 1, encoder_use_mis_mask=mc)
    gen = torch.Generator().manual_seed(5)
    if demiss:
        X = torch.randn(B, G, D, generator=gen).cuda()
        M = (torch.rand(B, G, D, generator=gen) < 0.5).cuda()
        plan = O._Plan("mvb2", 1, I)
    else:
        X = torch.randn(B, D, generator=gen).cuda()
        M = (torch.rand(B, D, generator=gen) < 0.5).cuda()
        plan = O._Plan("miwae" if q.startswith("strat") else "iwae", 1, I)
    N = I * B * G
    eps = torch.randn(N, Zd, generator=gen).cuda() if explicit else None
    cfg_probe = _forward_buffers(dims, plan, enc, dec, X, M, eps, training).cfg
    assert L.lib().dmv_vae_forward_fits(C.byref(cfg_probe), C.byref(enc.c), C.byref(dec.c)) == 1
    assert O.FUSE_SMALL_FORWARD
    one = _forward_buffers(dims, plan, enc, dec, X, M, eps, training)
    O.FUSE_SMALL_FORWARD = False
    try:
        three = _forward_buffers(dims, plan, enc, dec, X, M, eps, training)
    finally:
        O.FUSE_SMALL_FORWARD = True
    for name in ("enc_out", "Z", "dec_out") + (("eps", "enc_saved", "dec_saved") if training else ()):
        a, b = getattr(one, name), getattr(three, name)
        assert a.shape == b.shape, name
        if name.endswith("saved"):
            # rows past the batch in the last 32-row tile of a saved buffer are scratch in both paths: compare the rest
            rows = one.enc_rows if name == "enc_saved" else N
            net = enc if name == "enc_saved" else dec
            a, b = _saved_live(net, a, rows), _saved_live(net, b, rows)
        assert torch.equal(a, b), name


def _saved_live(net, saved, rows):
    """the entries of a saved-activation buffer that a forward pass over `rows` rows defines, as integers (sign-bit words
    are not numbers): it holds whole 32-row tiles, the first-layer input is written up to a multiple of 8 columns, and
    a row's sign bits are one halfword per 32 hidden units"""
    from demiss_vae_amd.mlp import pad32
    d = net.dims
    rp = (rows + 31) // 32 * 32
    Ip, Hp = pad32(d.in_dim), pad32(d.hidden)
    nl = 2 * d.nb + 1
    NT = Hp // 32
    mw = 2 * ((NT + 1) // 2)
    bits = saved.view(torch.int32)
    xs = bits[:rp * Ip].view(rp, Ip)[:rows, :8 * ((d.in_dim + 7) // 8)]
    acts = bits[rp * Ip:rp * Ip + nl * rp * Hp].view(nl, rp, Hp)[:, :rows]
    masks = bits[rp * Ip + nl * rp * Hp:].view(torch.int16).view(nl, rp, 2, mw)[:, :rows, :, :NT]
    return torch.cat([xs.reshape(-1), acts.reshape(-1), masks.reshape(-1).to(torch.int32)])


@pytest.mark.parametrize("case", [
    "iwae_stl", "iwae_dreg", "iwae_plain", "iwae_ragged_nz2", "elbo_analytic", "elbo_mc", "mvb2", "mvb2_mean", "mvb2_kl", "mvb",
    "mvb_kl", "iwae_dropout", "miwae_stl", "miwae_plain_i2", "miwae_looser", "mvae_plain", "cvi", "cvi_stl_nz2"])
def test_one_launch_backward_equals_separate_launches(case):
    """dmv_vae_backward (decoder chain, latent-side gradient, encoder chain in one launch) against the three separate
    launches on the same forward buffers: every parameter gradient to 1e-5 of its tensor's largest entry (same terms in
    the same order; only the compiler's choice of fused multiply-adds may differ)."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd import objectives as O
    from demiss_vae_amd import rng
    from demiss_vae_amd import synth
    gas = ((16, 32, 2, 256), (16, 16, 2, 256))
    toy = ((5, 4, 3, 200), (2, 10, 3, 200))
    hep = ((42, 165, 2, 256), (16, 42, 2, 256))          # stratified mixture of 5 in 16 latent dimensions
    shapes = {
        "iwae_stl": (512, 8, 16, 1, 5, 1, gas, dict(STL=True)),
        "iwae_dreg": (64, 8, 16, 1, 5, 1, gas, dict(DREG=True)),
        "iwae_plain": (64, 8, 16, 1, 16, 1, gas, dict()),
        "iwae_ragged_nz2": (37, 8, 16, 2, 3, 1, gas, dict(STL=True)),
        "iwae_dropout": (50, 8, 16, 1, 5, 1, gas, dict(STL=True)),
        "elbo_analytic": (100, 8, 16, 2, 1, 1, gas, dict()),
        "elbo_mc": (100, 8, 16, 1, 1, 1, gas, dict()),
        "mvb2": (64, 5, 2, 1, 1, 5, toy, dict(STL=True)),
        "mvb2_mean": (64, 5, 2, 1, 1, 5, toy, dict(STL=True)),
        "mvb2_kl": (20, 5, 2, 2, 1, 3, toy, dict()),
        "mvb": (33, 5, 2, 1, 1, 5, toy, dict()),
        "mvb_kl": (33, 5, 2, 1, 1, 5, toy, dict()),
        "miwae_stl": (512, 21, 16, 1, 1, 5, hep, dict(STL=True)),
        "miwae_plain_i2": (41, 21, 16, 1, 2, 5, hep, dict()),
        "miwae_looser": (41, 21, 16, 1, 1, 5, hep, dict(looser=True)),
        "mvae_plain": (41, 21, 16, 2, 1, 5, hep, dict()),
        "cvi": (33, 5, 2, 1, 1, 5, toy, dict()),
        "cvi_stl_nz2": (20, 5, 2, 2, 1, 3, toy, dict(STL=True)),
    }
    B, D, Zd, nz, I, K, (es, ds), kw = shapes[case]
    drop = 0.5 if case == "iwae_dropout" else 0.0
    demiss = case.startswith(("mvb", "cvi"))
    strat = case.startswith(("miwae", "mvae"))
    mc = not demiss
    dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist="normal_with_eps", q_dist="stratified_with_eps" if strat else "normal_with_eps",
                         C=K if strat else 1, encoder_use_mis_mask=mc)
    gen = torch.Generator().manual_seed(11)
    if demiss:
        X = torch.randn(B, K, D, generator=gen).cuda()
        M = (torch.rand(B, K, D, generator=gen) < 0.5).cuda()
    else:
        X = torch.randn(B, D, generator=gen).cuda()
        M = (torch.rand(B, D, generator=gen) < 0.5).cuda()

    def run(fused):
        enc = synth.packed_net(synth.NetShape(*es, dropout=drop), 1, "cuda")
        dec = synth.packed_net(synth.NetShape(*ds, dropout=drop), 2, "cuda")
        rng.set_seed(5)
        O.FUSE_SMALL_BACKWARD = fused
        try:
            ek = dk = None
            if drop:
                from demiss_vae_amd.imputation import _keep
                ek, dk = _keep(enc, B, True, "cuda"), _keep(dec, nz * I * B, True, "cuda")
            if case.startswith("iwae"):
                bound, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=nz, num_importance_samples=I,
                                               enc_keep=ek, dec_keep=dk, **kw)
                (-bound).backward()
            elif case.startswith("elbo"):
                bound, _ = dmv.estimate_elbo(dims, enc, dec, X, M, num_latent_samples=nz, kl_analytic=case == "elbo_analytic")
                (-bound).backward()
            elif case.startswith("miwae"):
                bound, _ = dmv.estimate_multiple_iwelbo(dims, enc, dec, X, M, num_latent_samples=nz,
                                                        num_importance_samples_for_each_component=I, **kw)
                (-bound).backward()
            elif case.startswith("mvae"):
                bound, _ = dmv.estimate_multiple_elbo(dims, enc, dec, X, M, num_latent_samples=nz)
                (-bound).backward()
            elif case.startswith("cvi"):
                obj = O.cvi_objective(dims, enc, dec, X, M, num_latent_samples=nz, **kw)
                (-obj.mean()).backward()
            elif case.startswith("mvb2"):
                gobj, eobj = O.mvb2_objectives(dims, enc, dec, X, M, num_latent_samples=nz, kl_analytic=case == "mvb2_kl",
                                               reduce_mean=case == "mvb2_mean", **kw)
                (-gobj.mean()).backward(retain_graph=True, inputs=dec.params())
                (-eobj.mean()).backward(inputs=enc.params())
            else:
                obj = O.mvb_objective(dims, enc, dec, X, M, num_latent_samples=nz, kl_analytic=case == "mvb_kl")
                (-obj.mean()).backward()
        finally:
            O.FUSE_SMALL_BACKWARD = True
        torch.cuda.synchronize()
        return [t.grad.clone() for t in enc.params() + dec.params()]

    one, three = run(True), run(False)
    assert all(g is not None for g in one)
    for a, b in zip(one, three):
        assert torch.isfinite(a).all()
        tol = 1e-5 * float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= tol, (case, float((a - b).abs().max()), float(b.abs().max()))


def test_batch_means_formed_by_the_last_workgroup():
    """With a ticket counter dmv_obj_forward forms the batch mean (and the loss, minus it) in the same launch: equal to
    the mean of the per-datapoint bounds and to the two-launch path, call after call, the counter back at zero."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd import _lib as L
    from demiss_vae_amd import synth
    enc = synth.packed_net(synth.NetShape(16, 32, 2, 64), 1, "cuda")
    dec = synth.packed_net(synth.NetShape(16, 16, 2, 64), 2, "cuda")
    dims = dmv.ModelDims(D=8, Zd=16, encoder_use_mis_mask=True)
    gen = torch.Generator().manual_seed(3)
    for B in (1, 37, 2000):
        X = torch.randn(B, 8, generator=gen).cuda()
        M = (torch.rand(B, 8, generator=gen) < 0.5).cuda()
        eps = torch.randn(1, 5, B, 16, generator=gen).cuda()
        assert L.ticket_ptr() is not None
        outs = []
        for _ in range(2):
            mean, per_b, neg = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_importance_samples=5, STL=True, eps=eps, with_loss=True)
            outs.append((mean.item(), neg.item()))
            np.testing.assert_allclose(mean.item(), per_b.double().mean().item(), rtol=2e-6)
            assert neg.item() == -mean.item()
        assert outs[0] == outs[1]
        saved, L.ticket_ptr = L.ticket_ptr, lambda: None
        try:
            mean2, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_importance_samples=5, STL=True, eps=eps)
        finally:
            L.ticket_ptr = saved
        np.testing.assert_allclose(outs[0][0], mean2.item(), rtol=2e-6)
    torch.cuda.synchronize()
    slots, _ = L._TICKETS[torch.cuda.current_device()]
    assert int(slots.abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["iwae_stl", "iwae_dreg", "iwae_plain_nz2", "vae_mc", "iwae_I16", "iwae_bern", "iwae_eval",
                                  "iwae_ragged", "miwae_stl", "miwae_plain_i2", "miwae_nz2_c3", "miwae_eval", "miwae_bern_c16"])
def test_forward_with_bound_equals_two_launches(case):
    """dmv_vae_forward_bound (encoder, sampling, decoder and the logsumexp bound in one launch, datapoint-major tiles)
    against dmv_vae_forward + dmv_obj_forward on the same noise: bounds and their mean to 2e-6 (other summation order
    over a row's dimensions), parameter gradients to 1e-5 of each tensor's largest entry.  The miwae cases: a stratified
    mixture posterior, one sample row per component, the tight MissSIWAE bound (with and without "sticking the landing":
    without it the backward also reads the component densities and the mixture term the launch leaves)."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd import objectives as O
    from demiss_vae_amd import rng, synth
    shapes = {          # B, D, Zd, nz, I, hidden, gen, kwargs, training
        "iwae_stl": (512, 8, 16, 1, 5, 256, "normal_with_eps", dict(STL=True), True),
        "iwae_dreg": (100, 8, 16, 1, 5, 256, "normal_with_eps", dict(DREG=True), True),
        "iwae_plain_nz2": (37, 5, 2, 2, 3, 200, "normal", dict(), True),
        "vae_mc": (64, 8, 16, 3, 1, 64, "normal_with_eps", dict(), True),
        "iwae_I16": (33, 8, 16, 1, 16, 64, "normal", dict(STL=True), True),
        "iwae_bern": (50, 20, 5, 1, 4, 96, "bern", dict(STL=True), True),
        "iwae_eval": (200, 8, 16, 1, 5, 256, "normal_with_eps", dict(), False),
        "iwae_ragged": (1, 3, 2, 1, 7, 32, "normal", dict(STL=True), True),
        "miwae_stl": (512, 21, 16, 1, 1, 256, "normal_with_eps", dict(STL=True), True),
        "miwae_plain_i2": (41, 21, 16, 1, 2, 64, "normal_with_eps", dict(), True),
        "miwae_nz2_c3": (37, 5, 2, 2, 2, 200, "normal", dict(STL=True), True),
        "miwae_eval": (200, 21, 16, 1, 1, 256, "normal_with_eps", dict(), False),
        "miwae_bern_c16": (9, 20, 5, 1, 1, 96, "bern", dict(STL=True), True),
    }
    B, D, Zd, nz, I, hidden, gen, kw, training = shapes[case]
    normal = gen.startswith("normal")
    Cn = 0 if not case.startswith("miwae") else 3 if case == "miwae_nz2_c3" else 16 if case == "miwae_bern_c16" else 5
    dims = dmv.ModelDims(D=D, Zd=Zd, gen_dist=gen, q_dist="stratified_with_eps" if Cn else "normal_with_eps", C=max(Cn, 1),
                         encoder_use_mis_mask=True)
    g = torch.Generator().manual_seed(17)
    X = torch.randn(B, D, generator=g)
    if not normal:
        X = (X > 0.2).float()
    X, M = X.cuda(), (torch.rand(B, D, generator=g) < 0.6).cuda()

    def run(fused):
        enc = synth.packed_net(synth.NetShape(2 * D, 2 * Cn * Zd + Cn if Cn else 2 * Zd, 2, hidden), 1, "cuda",
                               requires_grad=training)
        dec = synth.packed_net(synth.NetShape(Zd, 2 * D if normal else D, 2, hidden), 2, "cuda", requires_grad=training)
        rng.set_seed(23)
        O.FUSE_SMALL_BOUND, before = fused, O.bound_launches
        try:
            with torch.set_grad_enabled(training):
                if Cn:
                    mean, per_b, neg = dmv.estimate_multiple_iwelbo(dims, enc, dec, X, M, num_latent_samples=nz,
                                                                    num_importance_samples_for_each_component=I,
                                                                    with_loss=True, **kw)
                else:
                    mean, per_b, neg = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=nz,
                                                           num_importance_samples=I, with_loss=True, **kw)
                if training:
                    neg.backward()
        finally:
            O.FUSE_SMALL_BOUND = True
        torch.cuda.synchronize()
        assert (O.bound_launches - before == 1) == fused
        grads = [t.grad.clone() for t in enc.params() + dec.params()] if training else []
        return mean.detach().clone(), per_b.detach().clone(), neg.detach().clone(), grads

    one, two = run(True), run(False)
    torch.testing.assert_close(one[1], two[1], rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(one[0], two[0], rtol=2e-6, atol=2e-6)
    assert float(one[2]) == -float(one[0])
    # the mean is formed by whichever workgroup finishes last, from bounds the others published: again and again
    for _ in range(30 if case in ("iwae_eval", "iwae_stl", "miwae_stl") else 2):
        mean, per_b, _, _ = run(True)
        np.testing.assert_allclose(mean.item(), per_b.double().mean().item(), rtol=3e-6)
    for a, b in zip(one[3], two[3]):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, (case, float((a - b).abs().max()), float(b.abs().max()))


def test_step_with_paired_weight_gradients_equals_two_streams():
    """After the one-launch backward the weight gradients of encoder and decoder are formed by one pair of launches
    (dmv_mlp_backward_wgrad_pair) instead of side by side on two streams: equal to rounding (the pair cuts the rows into
    fewer slices), with and without a permanent gradient buffer.  The same for a batch just beyond the one-launch kernels
    (separate launches for the chains, then the pair), with a mixture posterior and dropout drawn in the kernels."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd import objectives as O, rng, synth
    from demiss_vae_amd.imputation import _keep
    gen = torch.Generator().manual_seed(21)

    def run(pair, flat, B, mixture):
        D, Zd, Cn = (21, 16, 5) if mixture else (8, 16, 1)
        drop = 0.5 if mixture else 0.0
        dims = dmv.ModelDims(D=D, Zd=Zd, q_dist="stratified_with_eps" if mixture else "normal_with_eps", C=Cn,
                             encoder_use_mis_mask=True)
        g2 = torch.Generator().manual_seed(22)
        X = torch.randn(B, D, generator=g2).cuda()
        M = (torch.rand(B, D, generator=g2) < 0.5).cuda()
        enc = synth.packed_net(synth.NetShape(2 * D, 2 * Cn * Zd + Cn if mixture else 2 * Zd, 2, 256, dropout=drop), 1, "cuda")
        dec = synth.packed_net(synth.NetShape(Zd, 2 * D, 2, 256, dropout=drop), 2, "cuda")
        if flat:
            enc.use_flat_grads()
            dec.use_flat_grads()
        rng.set_seed(9)
        O.PAIR_WGRAD = pair
        try:
            if mixture:
                ek, dk = _keep(enc, B, True, "cuda"), _keep(dec, B * Cn, True, "cuda")
                bound, _ = dmv.estimate_multiple_iwelbo(dims, enc, dec, X, M, num_importance_samples_for_each_component=1,
                                                        STL=True, enc_keep=ek, dec_keep=dk)
            else:
                bound, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=1, num_importance_samples=5, STL=True)
            (-bound).backward()
        finally:
            O.PAIR_WGRAD = True
        torch.cuda.synchronize()
        return [t.grad.clone() for t in enc.params() + dec.params()]

    for flat, B, mixture in ((False, 512, False), (True, 512, False), (False, 1200, False), (True, 1024, True)):
        a, b = run(True, flat, B, mixture), run(False, flat, B, mixture)
        for x, y in zip(a, b):
            assert float(y.abs().max()) > 0 and float((x - y).abs().max()) <= 1e-5 * float(y.abs().max())
