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
    families; vgiwae/shared/reparametrized_mixture_same_family.py), STL, through both engines: the fused residual
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
        bound, _ = fn.estimate_elbo(dims, a, b, X, M, num_latent_samples=nz, kl_analytic=False, STL=True, eps=eps, comp_idx=comp)
    else:
        bound, _ = fn.estimate_iwelbo(dims, a, b, X, M, num_latent_samples=nz, num_importance_samples=I, STL=True, eps=eps,
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
