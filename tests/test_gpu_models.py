"""The drop-in `vgiwae` model classes on the GPU: the class-level steps reproduce the reference's numbers
(golden vectors), reference-format state_dicts load, and a short training run on the toy data learns."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _build(cls_name, spec, enc_p, dec_p, **kw):
    import vgiwae.models as models
    from vgiwae.shared.neural_nets import ResidualFCNetwork
    from vgiwae.shared.vae_enums import DISTRIBUTION

    def net(s, p):
        n = ResidualFCNetwork(s.in_dim, s.out_dim, s.nb, s.hidden, dropout_probability=s.dropout)
        n.load_state_dict(p)                       # the reference's state_dict layout
        return n

    def dist(name, C):
        if name.startswith("stratified"):
            return DISTRIBUTION[f"stratified_mixture{C}_normal" + ("_with_eps" if name.endswith("with_eps") else "")]
        return DISTRIBUTION[name]
    model = getattr(models, cls_name)(generator_network=net(spec.dec, dec_p), generator_distribution=dist(spec.gen_dist, 1),
                                      var_latent_network=net(spec.enc, enc_p),
                                      var_latent_distribution=dist(spec.q_dist, spec.C), lr_generator=1e-3, lr_latent=1e-3,
                                      encoder_use_mis_mask=spec.encoder_use_mis_mask, **kw)
    return model.cuda().train()


def _grads(net, spec):
    sd = dict(net.named_parameters())
    return {k: v.grad for k, v in sd.items() if v.grad is not None}


def test_iwae_class_matches_reference():
    from vgiwae.models.iwae import estimate_iwelbo
    g = gu.load("iwae_gas_full")
    spec, enc_p, dec_p = gu.model_case(g)
    model = _build("IWAE_FID", spec, enc_p, dec_p, num_latent_samples=1, num_importance_samples=5, var_latent_STL=True,
                   inception_model_path="missing.ckpt", fid_eval_step_freq=5000)
    X, M = gu.batch_of(g)
    eps = gu.eps_of(g, 1, 5, int(g["B"]), spec.Zd)
    bound = estimate_iwelbo(model, (X.cuda(), M.cuda()), var_latent_STL=True, num_latent_samples=1,
                            num_importance_samples=5, eps=eps.cuda())
    (-bound).backward()
    np.testing.assert_allclose(bound.item(), float(g["bound"]), rtol=1e-5)
    assert gu.check_grads(g, "genc", _grads(model.var_latent_network, spec.enc), 1e-4, 2e-6) > 0
    assert gu.check_grads(g, "gdec", _grads(model.generator_network, spec.dec), 1e-4, 2e-6) > 0
    # training_step with device (Philox) noise: finite, close to the explicit-noise bound
    loss = model.training_step((X.cuda(), M.cuda(), torch.arange(X.shape[0]).cuda()), 0)
    assert torch.isfinite(loss) and abs(loss.item() + float(g["bound"])) < 0.5


def test_mvb2_manual_step_matches_reference_grads():
    g = gu.load("mvb_toy_full")
    spec, enc_p, dec_p = gu.model_case(g)
    from vgiwae.models import MVB_OBJECTIVE
    model = _build("MVBVAEMargLogprob", spec, enc_p, dec_p, num_latent_samples=1, var_latent_STL=True, kl_analytic=False,
                   mvb_objective=MVB_OBJECTIVE["mvb2"], marginal_eval_freq=5, latent_grid_steps=500)
    assert model.automatic_optimization is False
    Xt, Mk = gu.imputed_batch(g)
    eps = gu.eps_of(g, 1, int(g["B"]), int(g["K"]), spec.Zd)
    gen_obj, enc_obj = model.estimate_mvb2_objectives(Xt.cuda(), Mk.cuda(), eps=eps.cuda())
    opt = model.configure_optimizers()["optimizer"]
    model._optimizers = opt
    before = [p.detach().clone() for p in model.parameters()]
    model.mvb2_manual_optimisation_step(-enc_obj.mean(), -gen_obj.mean())
    assert gu.check_grads(g, "genc", _grads(model.var_latent_network, spec.enc), 1e-4, 2e-6) > 0
    assert gu.check_grads(g, "gdec", _grads(model.generator_network, spec.dec), 1e-4, 2e-6) > 0
    assert all(not torch.equal(a, b) for a, b in zip(before, model.parameters()))       # the step moved every tensor
    # the packed weights follow the optimiser update: a second evaluation differs from the first
    gen2, _ = model.estimate_mvb2_objectives(Xt.cuda(), Mk.cuda(), eps=eps.cuda())
    assert not torch.allclose(gen2, gen_obj)
    # the training step's route: scalar means out of the pipeline, ascent on them (no negations); same gradients
    model2 = _build("MVBVAEMargLogprob", spec, enc_p, dec_p, num_latent_samples=1, var_latent_STL=True, kl_analytic=False,
                    mvb_objective=MVB_OBJECTIVE["mvb2"], marginal_eval_freq=5, latent_grid_steps=500)
    gen_m, enc_m = model2.estimate_mvb2_objectives(Xt.cuda(), Mk.cuda(), eps=eps.cuda(), reduce_mean=True)
    assert gen_m.dim() == 0 and enc_m.dim() == 0
    torch.testing.assert_close(gen_m, gen_obj.mean().detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(enc_m, enc_obj.mean().detach(), rtol=1e-6, atol=1e-6)
    model2._optimizers = model2.configure_optimizers()["optimizer"]
    model2.mvb2_manual_optimisation_step(enc_m, gen_m, maximise=True)
    assert gu.check_grads(g, "genc", _grads(model2.var_latent_network, spec.enc), 1e-4, 2e-6) > 0
    assert gu.check_grads(g, "gdec", _grads(model2.generator_network, spec.dec), 1e-4, 2e-6) > 0


def test_refit_test_step_matches_reference():
    from vgiwae.test.refit_encoder_using_iwae_and_estimate_loglik import RefitEncoderUsingIWAE, RefitEncoderClass
    for name, cls, kind in (("loglik_standard_K4096", "IWAE", "standard"), ("loglik_stratified_K64", "MultipleIWAE", "stratified")):
        g = gu.load(name)
        spec, enc_p, dec_p = gu.model_case(g)
        kw = dict(num_importance_samples=5) if cls == "IWAE" else dict(num_importance_samples_for_each_component=1)
        model = _build(cls, spec, enc_p, dec_p, num_latent_samples=1, **kw).eval()
        refit = RefitEncoderUsingIWAE(RefitEncoderClass[kind], num_test_latent_samples=1,
                                      num_test_importance_samples=int(g["K"]))
        refit.set_model(model)
        X, M = torch.from_numpy(g["X"]).cuda(), torch.from_numpy(g["M"]).cuda()
        if kind == "standard":
            eps = gu.eps_of(g, 1, int(g["K"]), int(g["B"]), spec.Zd)[0].cuda()
            val = refit.test_step((X, M), 0, eps=eps)
        else:
            val = refit.test_step((X, M), 0, eps=torch.from_numpy(g["eps"])[0].cuda(),
                                  comp_idx=torch.from_numpy(g["comp_idx"])[0].cuda())
        assert abs(val.item() - float(g["est_loglik"])) < 0.01          # north-star: within 0.01 nats


@pytest.mark.parametrize("tag", ["noencm", "encm"])
def test_conv_trunk_demissvae_matches_reference(tag):
    """DeMissVAE with convolutional trunks (the MNIST / Omniglot configs at reduced widths): the trunks are torch
    modules with the reference's state_dict, everything around them runs in the kernels.  One pseudo-Gibbs sweep,
    the two mvb2 objectives and the two backward calls against the reference's numbers.  Tolerances are wider than
    for the MLP path: the convolutions run in MIOpen here and in the CPU backend in the fixture."""
    import vgiwae.models as models
    from vgiwae.models import MVB_OBJECTIVE
    from vgiwae.shared.vae_enums import DISTRIBUTION
    from vgiwae.shared.vae_imputation import VAESampler_PseudoGibbs
    from vgiwae.shared.vae_resnet import ResNetDecoder, ResNetEncoder
    g = gu.load(f"resnet_mvb2_{tag}")
    B, K, D, Zd, use_mask = int(g["B"]), int(g["K"]), int(g["D"]), int(g["Zd"]), bool(g["use_mask"])
    enc = ResNetEncoder(input_shape=[1, 28, 28], layers=[1, 1], layer_widths=[4, 8], latent_dim=2 * Zd, dropout_prob=0.0,
                        expect_miss_mask_as_extra_chennel=use_mask)
    dec = ResNetDecoder(output_shape=[1, 28, 28], layers=[1, 1], layer_upscale=[None, None], layer_widths=[8, 4],
                        downscaled_resolution=[4, 4], latent_dim=Zd, upscale_factor=2, dropout_prob=0.0)
    enc.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.keys() if k.startswith("enc.")})
    dec.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.keys() if k.startswith("dec.")})
    model = models.MVBVAE(generator_network=dec, generator_distribution=DISTRIBUTION.bern_m1_p1, var_latent_network=enc,
                          var_latent_distribution=DISTRIBUTION.normal_with_eps, num_latent_samples=1, lr_generator=1e-4,
                          lr_latent=1e-4, encoder_use_mis_mask=use_mask, var_latent_STL=True, kl_analytic=False,
                          mvb_objective=MVB_OBJECTIVE["mvb2"]).cuda().train()
    assert not model.fused
    Xt, Mk = torch.from_numpy(g["Xt"]).cuda(), torch.from_numpy(g["M"]).cuda()
    sampler = VAESampler_PseudoGibbs(num_iterations=1)
    X_new = sampler(Xt.clone(), Mk, model=model, eps_z=torch.from_numpy(g["eps_z"])[None].cuda(),
                    noise_x=torch.from_numpy(g["u_x"])[None].cuda())
    ref_new = torch.from_numpy(g["X_new"]).cuda()
    assert torch.equal(X_new[Mk], ref_new[Mk])                               # observed entries untouched, bit-exact
    flips = (X_new != ref_new).float().mean().item()                          # a draw flips only where u ~ sigmoid(logit)
    assert flips < 2e-3, flips
    gen_obj, enc_obj = model.estimate_mvb2_objectives(ref_new, Mk, eps=torch.from_numpy(g["eps"]).cuda())
    torch.testing.assert_close(gen_obj.cpu(), torch.from_numpy(g["gen_obj"]), rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(enc_obj.cpu(), torch.from_numpy(g["enc_obj"]), rtol=2e-4, atol=2e-3)
    (-gen_obj.mean()).backward(retain_graph=True, inputs=list(dec.parameters()))
    (-enc_obj.mean()).backward(inputs=list(enc.parameters()))
    # Conditioning of the two fixtures, measured with the reference itself on the CPU (float32 against float64, same
    # weights and noise; objectives agree to 6e-8 in both): the gradients of the `encm` case differ norm-wise by
    # 4e-3 .. 6e-3 in every encoder tensor and by 3e-4 .. 5e-3 in the decoder's (binary images and masks through
    # 3x3 convolutions give exact ties at the relu / max-pool kinks, which rounding breaks either way); those of
    # `noencm` by 1e-5.  So `encm` pins an fp32 implementation to about 1 % of a tensor's largest entry and no better
    # (elementwise floor 1e-2, most entries lean on it, plus a norm-wise bound); `noencm` keeps the tight floor.
    floor, share = (1e-2, 1.0) if use_mask else (2e-3, gu.FLOOR_FRACTION_MAX)
    for prefix, net in (("genc", enc), ("gdec", dec)):
        grads = {k: v.grad for k, v in net.named_parameters() if v.grad is not None}
        assert gu.check_grads(g, prefix, grads, 5e-3, 1e-5, floor=floor, floor_fraction_max=share) > 5
        for k, v in grads.items():
            if f"{prefix}.{k}.full" in g:
                ref = torch.from_numpy(g[f"{prefix}.{k}.full"]).double()
                assert float((v.cpu().double() - ref).norm() / ref.norm()) < (2e-2 if use_mask else 5e-4), (prefix, k)


@pytest.mark.parametrize("name", gu.names("marginal_"))
def test_grid_marginal_logprob_matches_reference(name):
    """The toy configs' evaluation metric: log p(x_o) and log p(x) by integrating over the latent grid."""
    import gpu_util as G
    from demiss_vae_amd.marginal import grid_marginal_logprob
    g = gu.load(name)
    spec, _, dec_p = gu.model_case(g)
    X, M = gu.batch_of(g)
    lo = None if np.isnan(g["grid_min"]) else float(g["grid_min"])
    hi = None if np.isnan(g["grid_max"]) else float(g["grid_max"])
    obs, full = grid_marginal_logprob(G.model_dims(spec), G.packed_net(spec.dec, dec_p, requires_grad=False), X.cuda(), M.cuda(),
                                      compute_complete=True, grid_min=lo, grid_max=hi, grid_steps=int(g["grid_steps"]))
    np.testing.assert_allclose(obs.cpu().numpy(), g["marginal_obs"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(full.cpu().numpy(), g["marginal_full"], rtol=1e-5, atol=2e-5)


def test_marglogprob_class_logs_the_metric():
    g = gu.load("mvb_toy_full")
    spec, enc_p, dec_p = gu.model_case(g)
    from vgiwae.models import MVB_OBJECTIVE
    model = _build("MVBVAEMargLogprob", spec, enc_p, dec_p, num_latent_samples=1, var_latent_STL=True, kl_analytic=False,
                   mvb_objective=MVB_OBJECTIVE["mvb2"], marginal_eval_freq=5, latent_grid_steps=100, mvb_update_val_imps=False)
    logged = {}
    model.log = lambda k, v, **kw: logged.__setitem__(k, float(v))
    X, M = gu.batch_of(g)
    model.eval()
    model.validation_step((X.cuda(), M.cuda(), torch.arange(X.shape[0]).cuda()), 0)
    assert np.isfinite(logged["marginal_logprob/val"]) and logged["complete_marginal_logprob/val"] < logged["marginal_logprob/val"]


def test_graphed_step_matches_eager_steps():
    """One training step recorded in a HIP graph and replayed = the same step run eagerly (same noise, same Adam):
    bounds and parameters agree after three eager + three replayed steps vs six eager ones; the fragment images are
    re-packed inside the recording."""
    import demiss_vae_amd as dmv
    import gpu_util as G
    g = gu.load("iwae_stl_normal_with_eps_mc")
    spec, enc_p, dec_p = gu.model_case(g)
    X, M = (t.cuda() for t in gu.batch_of(g))
    B, I = int(g["B"]), int(g["I"])
    noise = [torch.randn(1, I, B, spec.Zd, generator=torch.Generator().manual_seed(s)).cuda() for s in range(6)]
    results = []
    for graphed in (False, True):
        dims, enc, dec = G.model_dims(spec), G.packed_net(spec.enc, enc_p), G.packed_net(spec.dec, dec_p)
        params = enc.params() + dec.params()
        opt = torch.optim.Adam(params, lr=1e-3, amsgrad=True, capturable=True)
        eps = torch.zeros_like(noise[0])

        def body():
            opt.zero_grad(set_to_none=True)
            b, _ = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=1, num_importance_samples=I, STL=True, eps=eps)
            (-b).backward()
            torch.nn.utils.clip_grad_norm_(params, 5.0)
            opt.step()
            return b
        bounds = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for e in noise[:3]:                              # eager steps (they also create the optimiser state)
                eps.copy_(e)
                bounds.append(body().detach().clone())
        torch.cuda.current_stream().wait_stream(side)
        step = dmv.GraphedStep(body, nets=(enc, dec), warmup=0) if graphed else body
        for e in noise[3:]:
            eps.copy_(e)
            bounds.append(step().detach().clone())
        results.append((torch.stack(bounds), [p.detach().clone() for p in params]))
    (b0, p0), (b1, p1) = results
    torch.testing.assert_close(b0, b1, rtol=1e-6, atol=1e-6)
    for a_, b_ in zip(p0, p1):
        torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-7)


def test_short_training_run_learns():
    """A few hundred Adam steps of MissIWAE on synthetic correlated data: the bound improves markedly."""
    from oracle.recipes import NetSpec, make_params, make_mask
    from oracle.core import ModelSpec
    rng = np.random.Generator(np.random.PCG64(0))
    N, D, Zd = 4096, 8, 4
    A = rng.standard_normal((3, D)).astype(np.float32)
    Xn = (rng.standard_normal((N, 3)).astype(np.float32) @ A + 0.1 * rng.standard_normal((N, D)).astype(np.float32))
    X, M = torch.from_numpy(Xn).cuda(), torch.from_numpy(make_mask(rng, N, D)).cuda()
    enc_s, dec_s = NetSpec(2 * D, 2 * Zd, 2, 64), NetSpec(Zd, 2 * D, 2, 64)
    spec = ModelSpec(D=D, Zd=Zd, enc=enc_s, dec=dec_s, encoder_use_mis_mask=True)
    model = _build("IWAE", spec, make_params(enc_s, 1), make_params(dec_s, 2), num_latent_samples=1,
                   num_importance_samples=5, var_latent_STL=True, amsgrad_generator=True, amsgrad_latent=True)
    opt = model.configure_optimizers()["optimizer"]
    first = last = None
    for step in range(300):
        idx = torch.randint(0, N, (512,), device="cuda")
        loss = model.training_step((X[idx], M[idx], idx), step)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        if step < 10:
            first = loss.item() if first is None else 0.9 * first + 0.1 * loss.item()
        last = loss.item() if last is None else 0.9 * last + 0.1 * loss.item()
    assert last < first - 1.0, (first, last)


@pytest.mark.parametrize("name", gu.names("mvbiwae_"))
def test_mvbiwae_class_matches_reference(name):
    """every objective variant of the importance-weighted DeMissVAE: values and both networks' gradients"""
    from vgiwae.models import IS_WEIGHTS, MVB_IWAE_OBJ
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(dict(g, gen_dist="normal", q_dist="normal", mask_concat=False))
    objective = str(g["objective"])
    model = _build("MVBIWAE", spec, enc_p, dec_p, num_latent_samples=int(g["nz"]), num_importance_samples=int(g["I"]),
                   var_latent_STL=bool(g["STL"]), var_latent_DREG=bool(g["DREG"]), mvb_objective=MVB_IWAE_OBJ[objective],
                   enc_isweights=IS_WEIGHTS[str(g["enc_w"])], dec_isweights=IS_WEIGHTS[str(g["dec_w"])])
    assert model.automatic_optimization is not objective.startswith("mvb2")
    Xt, Mk, eps = (torch.from_numpy(g[k]).cuda() for k in ("Xt", "M", "eps"))
    if objective.startswith("mvb2"):
        ridx = torch.from_numpy(g["resample_idx"]).cuda() if "resample_idx" in g else None
        gen_obj, enc_obj = model.estimate_mvb2_objectives(Xt, Mk, eps=eps, resample_idx=ridx)
        (-gen_obj.mean()).backward(retain_graph=True, inputs=list(model.generator_network.parameters()))
        (-enc_obj.mean()).backward(inputs=list(model.var_latent_network.parameters()))
    else:
        fn = model.estimate_mvb_objective if objective.startswith("mvb") else model.estimate_cvi_objective
        gen_obj = enc_obj = fn(Xt, Mk, eps=eps)
        (-gen_obj.mean()).backward()
    torch.testing.assert_close(gen_obj.detach().cpu(), torch.from_numpy(g["gen_obj"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(enc_obj.detach().cpu(), torch.from_numpy(g["enc_obj"]), rtol=1e-5, atol=1e-5)
    assert gu.check_grads(g, "genc", _grads(model.var_latent_network, spec.enc), 1e-4, 2e-6) > 0
    assert gu.check_grads(g, "gdec", _grads(model.generator_network, spec.dec), 1e-4, 2e-6) > 0


def test_mvbiwae_training_steps_run_and_learn():
    """training_step / validation_step of the importance-weighted DeMissVAE with device noise and the imputation
    buffers: a manual-optimisation variant and a single-objective one"""
    from types import SimpleNamespace
    from oracle.recipes import NetSpec, make_batch, make_params
    from oracle.core import ModelSpec
    from vgiwae.data import DatasetWithIndices
    from vgiwae.models import IS_WEIGHTS, MVB_IWAE_OBJ
    from vgiwae.shared.vae_imputation import VAESampler_PseudoGibbs
    from vgiwae.utils.imputation_distribution import IMPUTATION_INIT
    spec = ModelSpec(D=5, Zd=2, enc=NetSpec(5, 4, 1, 32), dec=NetSpec(2, 10, 1, 32), gen_dist="normal", q_dist="normal")
    X, M = make_batch(5, 256, 5)

    class Data(torch.utils.data.Dataset):
        def __getitem__(self, i):
            return X.numpy()[i], M.numpy()[i]

        def __len__(self):
            return len(X)
    for objective in ("mvb2_encallweighteddmmis_decall", "cvi_all"):
        torch.manual_seed(0)
        model = _build("MVBIWAE", spec, make_params(spec.enc, 1), make_params(spec.dec, 2), num_latent_samples=1,
                       num_importance_samples=3, var_latent_STL=True, mvb_objective=MVB_IWAE_OBJ[objective],
                       enc_isweights=IS_WEIGHTS.dmmis, dec_isweights=IS_WEIGHTS.dmmis, num_imputations_train=4,
                       num_imputations_val=4, imputation_init_train=IMPUTATION_INIT.empirical,
                       imputation_init_val=IMPUTATION_INIT.empirical, init_imputation_distribution=True, mvb_eval_val_loss=True,
                       mvb_delay_imputation=0, vae_sampler=VAESampler_PseudoGibbs(num_iterations=1))
        data = DatasetWithIndices(Data())
        model.trainer = SimpleNamespace(datamodule=SimpleNamespace(train_data=data, val_data=data), max_epochs=1, max_steps=-1)
        model.setup("fit")
        opt = model.configure_optimizers()["optimizer"]
        model._optimizers = opt
        batch = (X.cuda(), M.cuda(), torch.arange(256).cuda())
        losses = []
        for step in range(40):
            model._logged = {}
            loss = model.training_step(batch, step)
            if model.automatic_optimization:
                opt.zero_grad()
                loss.backward()
                opt.step()
            losses.append(float(loss.detach()))
        assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])
        assert ("cvi/train" in model._logged) and "loss/train" in model._logged
        model._logged = {}
        model.validation_step(batch, 0)
        assert np.isfinite(float(model._logged["loss/val"]))


def test_rejection_sampler_draws_from_the_grid_posterior():
    """VAESampler_Rejection: accepted latents follow p(z | x_obs) as tabulated on the latent grid; imputations keep the
    observed values and are written back into the buffer.  (Own device draws: checked statistically.)"""
    import math
    from oracle.recipes import NetSpec, make_batch, make_params
    from oracle.core import ModelSpec
    from demiss_vae_amd.imputation import ImputationBuffer
    from vgiwae.shared import vae_marginal_logprob as R
    spec = ModelSpec(D=5, Zd=2, enc=NetSpec(5, 4, 1, 32), dec=NetSpec(2, 10, 2, 32), gen_dist="normal", q_dist="normal")
    model = _build("VAE", spec, make_params(spec.enc, 1), make_params(spec.dec, 2), num_latent_samples=1).eval()
    X, M = make_batch(9, 6, 5)
    x, m = X.cuda(), M.cuda()
    dims, _, dec = model.engine()
    lo, hi, steps, K = -4.0, 4.0, 160, 3000
    env = R.grid_envelope(dims, dec, x, m, lo, hi, steps)
    # the same joint on the grid, by the oracle-side formulas on the CPU
    from oracle import core
    Z, _ = R.latent_grid(2, "cpu", lo, hi, steps)
    g = core.mlp_forward(make_params(spec.dec, 2), Z, spec.dec)
    joint = (core.gen_logprob(spec, g.unsqueeze(0), X.unsqueeze(1)) * M.unsqueeze(1)).sum(-1) + core.log_std_normal(Z).sum(-1)
    torch.testing.assert_close(env.cpu(), joint.max(1)[0], rtol=1e-4, atol=1e-4)
    gen = torch.Generator(device="cuda").manual_seed(3)
    Zs = R.sample_pz_given_xo_with_rejection_sampling(dims, dec, x, m, K, env + math.log(1.05), 8192, lo, hi, generator=gen)
    assert Zs.shape == (6, K, 2) and bool(((Zs >= lo) & (Zs <= hi)).all())
    post = torch.softmax(joint.double(), 1)                                         # grid posterior [B, Nz]
    mean = post @ Z.double()
    var = post @ (Z.double() ** 2) - mean ** 2
    err = (Zs.double().mean(1).cpu() - mean).abs()
    assert bool((err < 5 * (var / K).sqrt() + 2 * (hi - lo) / steps).all()), (err, var)   # Monte-Carlo error + one grid cell
    # the sampler call: K copies, observed entries kept, missing ones fresh per copy, buffer updated
    sampler = R.VAESampler_Rejection(Z_batchsize=4096, envelope_multiplier=1.05, grid_min=lo, grid_max=hi, grid_steps=steps)
    buf = ImputationBuffer(X, 4, "cuda")
    Xt, idx = buf.gather(torch.arange(6).cuda())
    Mk = m.unsqueeze(1).expand(6, 4, 5).contiguous()
    out = sampler(Xt, Mk, model=model, buffer=buf, idx_imp=idx, generator=gen)
    assert out.shape == (6, 4, 5) and torch.equal(out[Mk], Xt[Mk])
    assert not torch.equal(out[:, 0][~m], out[:, 1][~m])
    assert torch.equal(buf.gather(torch.arange(6).cuda())[0], out)


def test_mog_oracle_sampler_on_the_device():
    """VAESampler_MoG_Oracle inside update_imputations' calling convention (device tensors, buffer write-back)"""
    from types import SimpleNamespace
    from demiss_vae_amd.imputation import ImputationBuffer
    from vgiwae.utils.mog_utils import VAESampler_MoG_Oracle
    g = gu.load("mog_oracle")
    f = {"comp_probs": np.exp(g["comp_log_probs"])[None], "means": g["means"], "covs": g["covs"]}
    model = SimpleNamespace(datamodule=SimpleNamespace(train_data_core=SimpleNamespace(data_file=f)))
    X, M = torch.from_numpy(g["X"]), torch.from_numpy(g["M"])
    buf = ImputationBuffer(X, 5, "cuda")
    index = torch.arange(len(X)).cuda()
    Xt, idx = buf.gather(index)
    Mk = M.cuda().unsqueeze(1).expand(-1, 5, -1).contiguous()
    sampler = VAESampler_MoG_Oracle(start_distribution="indep_marginal_mog", interpolate_target="conditional_independent_mog",
                                    interpolate_alpha=0.25)
    out = sampler(Xt, Mk, model=model, buffer=buf, idx_imp=idx, generator=torch.Generator(device="cuda").manual_seed(0))
    assert out.is_cuda and out.dtype == torch.float32 and torch.isfinite(out).all()
    assert torch.equal(out[Mk], Xt[Mk]) and torch.equal(buf.gather(index)[0], out)


def test_graph_replays_draw_fresh_library_noise():
    """A recorded step whose noise comes from the library's Philox stream: every replay advances the device-resident
    counter, so replays see new draws (all standard normal, no repeats), from a stream of their own."""
    import ctypes as C
    import demiss_vae_amd as dmv
    import gpu_util as G
    from demiss_vae_amd import _lib as L, rng
    g = gu.load("iwae_stl_normal_with_eps_mc")
    spec, enc_p, dec_p = gu.model_case(g)
    dims, enc, dec = G.model_dims(spec), G.packed_net(spec.enc, enc_p, requires_grad=False), \
        G.packed_net(spec.dec, dec_p, requires_grad=False)
    X, M = (t.cuda() for t in gu.batch_of(g))
    rng.set_seed(7)
    cfg = L.DmvObjCfg()
    N, Zd = 4096, 8
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = N, 1, 1, 1, 1, Zd
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off = 2 * Zd, 0, Zd
    enc_out = torch.zeros(N, 2 * Zd, device="cuda")
    Z, used = torch.empty(N, Zd, device="cuda"), torch.empty(N, Zd, device="cuda")

    def body():
        with torch.no_grad():
            seed, off = rng.next_offset(N * Zd)
            L.check(L.lib().dmv_latent_sample(C.byref(cfg), L.ptr(enc_out), None, seed, off, rng.base_ptr(), Zd, None, L.ptr(Z),
                                              L.ptr(used), L.stream_ptr()))
            return dmv.estimate_iwelbo(dims, enc, dec, X, M, num_importance_samples=5)[1]
    before = rng.get_state()
    step = dmv.GraphedStep(body, warmup=1)
    host_after_capture = rng.get_state()
    draws, bounds = [], []
    for _ in range(5):
        bounds.append(step().clone())
        draws.append(used.clone())
    assert rng.get_state() == host_after_capture                       # replays leave the eager stream alone ...
    assert host_after_capture[1] > before[1]                           # ... which only the eager warm-up advanced
    per_step = N * Zd + ((5 * X.shape[0] * spec.Zd + 3) & ~3)
    assert int(step.noise.counter) == 5 * per_step
    flat = torch.stack(draws).reshape(5, -1)
    assert len({tuple(d[:8].tolist()) for d in flat}) == 5             # different on every replay
    assert torch.unique(flat).numel() > 0.99 * flat.numel()            # no range reused (fp32 birthday collisions aside)
    e = flat.double()
    assert abs(e.mean().item()) < 0.01 and abs(e.std().item() - 1) < 0.01
    assert len({round(float(b.mean()), 5) for b in bounds}) == 5       # the bound is re-estimated with fresh noise
    eager = torch.stack([body().mean() for _ in range(20)]).mean()
    assert abs(float(torch.stack(bounds).mean()) - float(eager)) < 0.5


def test_fully_connected_networks_run_through_the_caller_network_engine():
    """FullyConnectedNetwork trunks (torch modules) with the kernels at their outputs: IWAE bound and gradients vs the
    same computation in float64 tensor operations on the CPU."""
    import copy
    from oracle import core
    from oracle.recipes import NetSpec, make_batch
    from vgiwae.models import IWAE
    from vgiwae.models.iwae import estimate_iwelbo
    from vgiwae.shared.neural_nets import FullyConnectedNetwork
    from vgiwae.shared.vae_enums import DISTRIBUTION
    torch.manual_seed(3)
    D, Zd, B, I = 6, 3, 9, 4
    dec, enc = FullyConnectedNetwork([Zd, 24, 2 * D]), FullyConnectedNetwork([2 * D, 24, 24, 2 * Zd], activation=torch.tanh)
    ref_dec, ref_enc = copy.deepcopy(dec).double(), copy.deepcopy(enc).double()
    model = IWAE(generator_network=dec, generator_distribution=DISTRIBUTION.normal_with_eps, var_latent_network=enc,
                 var_latent_distribution=DISTRIBUTION.normal_with_eps, num_latent_samples=1, num_importance_samples=I,
                 lr_generator=1e-3, lr_latent=1e-3, encoder_use_mis_mask=True, var_latent_STL=True).cuda().train()
    assert not model.fused and model.latent_dims == Zd
    X, M = make_batch(21, B, D)
    eps = torch.randn(1, I, B, Zd)
    bound = estimate_iwelbo(model, (X.cuda(), M.cuda()), var_latent_STL=True, num_importance_samples=I, eps=eps.cuda())
    (-bound).backward()
    spec = core.ModelSpec(D=D, Zd=Zd, enc=NetSpec(2 * D, 2 * Zd, 0, 1), dec=NetSpec(Zd, 2 * D, 0, 1), encoder_use_mis_mask=True)
    Xd, Md = X.double(), M
    mu, sigma, _ = core.q_heads(spec, ref_enc(core.encoder_input(spec, Xd * Md, Md.double())))
    Z = mu + sigma * eps.double()
    log_w = core.smis_log_weights(spec, ref_dec(Z), Xd, Md, Z, mu.detach(), sigma.detach())
    want = (torch.logsumexp(log_w, 1) - np.log(I)).mean(0).mean()
    (-want).backward()
    np.testing.assert_allclose(bound.item(), want.item(), rtol=2e-5)
    for mine, ref in ((dec, ref_dec), (enc, ref_enc)):
        for (k, p), q in zip(mine.named_parameters(), ref.parameters()):
            torch.testing.assert_close(p.grad.cpu().double(), q.grad, rtol=2e-3, atol=2e-5, msg=k)


def test_training_trajectory_matches_reference():
    """25 optimisation steps through the model class, the library's clip + Adam / AMSGrad step and a cosine schedule,
    on the reference's own trajectory: bound and gradient norm at every step, parameters at the end."""
    from oracle.recipes import make_batch, make_noise
    from vgiwae.models.iwae import estimate_iwelbo
    from demiss_vae_amd.optim import Adam
    g = gu.load("trajectory_iwae")
    spec, enc_p, dec_p = gu.model_case(dict(g, gen_dist="normal_with_eps", q_dist="normal_with_eps", mask_concat=True))
    seed, steps, B, I = int(g["seed"]), int(g["steps"]), int(g["B"]), int(g["I"])
    model = _build("IWAE", spec, enc_p, dec_p, num_latent_samples=1, num_importance_samples=I, var_latent_STL=True,
                   amsgrad_generator=True, amsgrad_latent=False, use_lr_scheduler=True, max_scheduler_steps=steps)
    model.hparams.lr_generator, model.hparams.lr_latent = float(g["lr_generator"]), float(g["lr_latent"])
    conf = model.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    assert isinstance(opt, Adam)
    for t in range(steps):
        X, M = make_batch(seed + 100 + t, B, spec.D)
        eps = make_noise(seed + 1000 + t, 1, I, B, spec.Zd)
        bound = estimate_iwelbo(model, (X.cuda(), M.cuda()), var_latent_STL=True, num_importance_samples=I, eps=eps.cuda())
        opt.zero_grad()
        (-bound).backward()
        model.configure_gradient_clipping(opt, 0, gradient_clip_val=float(g["clip"]), gradient_clip_algorithm="norm")
        opt.step()
        sched.step()
        np.testing.assert_allclose(bound.item(), g["bounds"][t], rtol=2e-4)
        np.testing.assert_allclose(opt.grad_norm.item(), g["grad_norms"][t], rtol=2e-3)
        np.testing.assert_allclose([grp["lr"] for grp in opt.param_groups], g["lrs"][t], rtol=1e-6)
    for name, net in (("enc", model.var_latent_network), ("dec", model.generator_network)):
        for k, v in net.state_dict().items():
            torch.testing.assert_close(v.cpu(), torch.from_numpy(g[f"{name}_final.{k}"]), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name", gu.names("posterior_divs_"))
def test_posterior_divergences_match_reference(name):
    from demiss_vae_amd.marginal import posterior_divergences
    g = gu.load(name)
    spec, enc_p, dec_p = gu.model_case(g)
    model = _build("VAE", spec, enc_p, dec_p, num_latent_samples=1).eval()
    X, M, Xt, Mk = (torch.from_numpy(g[k]).cuda() for k in ("X", "M", "Xt", "Mk"))
    out = posterior_divergences(model.model_dims(), model.predict_var_latent_params, model.generator_network.packed_net(),
                                X, M, Xt, Mk, grid_min=float(g["grid_min"]), grid_max=float(g["grid_max"]),
                                grid_steps=int(g["grid_steps"]))
    names = ("comp_fow", "comp_rev", "comp_jsd", "inc_fow", "inc_rev", "inc_jsd", "imp_fow", "imp_rev", "imp_jsd")
    for k, v in zip(names, out):
        torch.testing.assert_close(v.cpu(), torch.from_numpy(g[k]), rtol=2e-4, atol=2e-5, msg=k)


def test_refit_test_step_with_torch_trunks_matches_the_streaming_estimator():
    """RefitEncoderUsingIWAE.test_step for networks that are not residual MLPs (the image refit configs) against the
    streaming estimator of the same networks: a stratified-mixture encoder sampled as a plain mixture, same noise and
    component draws."""
    from vgiwae.test.refit_encoder_using_iwae_and_estimate_loglik import RefitEncoderUsingIWAE, RefitEncoderClass
    g = gu.load("loglik_stratified_K64")
    spec, enc_p, dec_p = gu.model_case(g)
    X, M = (t.cuda() for t in gu.batch_of(g))
    K, B = int(g["K"]), int(g["B"])
    eps, comp = torch.from_numpy(g["eps"]).cuda(), torch.from_numpy(g["comp_idx"]).cuda()

    class Trunk(torch.nn.Module):                      # hides the residual MLP from the fused engine
        def __init__(self, inner):
            super().__init__()
            self.inner, self.latent_dim, self.output_dim = inner, inner.input_dim, inner.output_dim

        def forward(self, x):
            return self.inner(x)
    values = []
    for wrap in (False, True):
        model = _build("MultipleIWAE", spec, enc_p, dec_p, num_latent_samples=1, num_importance_samples_for_each_component=1,
                       var_latent_STL=True).eval()
        if wrap:
            model.generator_network, model.var_latent_network = Trunk(model.generator_network), Trunk(model.var_latent_network)
            assert not model.fused
        refit = RefitEncoderUsingIWAE(RefitEncoderClass.stratified, var_latent_STL=True, num_test_importance_samples=K)
        refit.set_model(model)
        with torch.no_grad():
            values.append(float(refit.test_step((X, M), 0, eps=eps, comp_idx=comp)))
    np.testing.assert_allclose(values[0], float(g["est_loglik"]), atol=1e-4)
    np.testing.assert_allclose(values[1], values[0], rtol=1e-5)
