"""CPU tests of the host side: the batch contract (masks / filter / index, bit-exact against the reference's
outputs), the C ABI surface, and the Philox generator shared by the device kernels."""
import ctypes
import os
import re

import numpy as np
import torch

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Plain(torch.utils.data.Dataset):
    def __init__(self, data):
        self.data = data

    def __getitem__(self, i):
        return self.data[i]

    def __len__(self):
        return len(self.data)


def test_masks_filter_index_bit_exact():
    from vgiwae.data import (UniformMissingnessDataset, QuadrantMissingnessDataset, FullyMissingDataFilter,
                             DatasetWithIndices)
    g = gu.load("masks")
    N, D = int(g["N"]), int(g["D"])
    data = np.arange(N * D, dtype=np.float32).reshape(N, D)
    for miss in (20, 50, 80):
        rng = torch.Generator()
        rng.manual_seed(int(g["seed"]))
        ds = UniformMissingnessDataset(_Plain(data), total_miss=miss / 100, rng=rng)
        assert ds.miss_mask.dtype == np.bool_ and np.array_equal(ds.miss_mask, g[f"uniform_{miss}"])
        assert (~ds.miss_mask).sum() == int(miss / 100 * N * D)
        f = FullyMissingDataFilter(ds)
        assert np.array_equal(np.asarray(f.not_fully_missing_idx), g[f"kept_{miss}"])
        w = DatasetWithIndices(f)
        assert w.index_idx == 2
        X, M, idx = w[:]
        assert idx.dtype == np.int64 and np.array_equal(idx, g[f"index_{miss}"])
        assert M.any(axis=1).all() and np.array_equal(X, data[g[f"kept_{miss}"]])
    img = np.zeros((19, 784), dtype=np.float32)
    for miss in (25, 50, 75):
        rng = torch.Generator()
        rng.manual_seed(int(g["seed"]))
        ds = QuadrantMissingnessDataset(_Plain(img), total_miss=miss / 100, img_dims=[28, 28], rng=rng)
        assert np.array_equal(ds.miss_mask, g[f"quadrant_{miss}"])
        assert ((~ds.miss_mask).sum(1) == 784 * miss // 100).all()
    from vgiwae.data.tophalf_missingness_provider import TopHalfMissingnessDataset
    for miss in (25, 50):
        rng = torch.Generator()
        rng.manual_seed(int(g["seed"]))
        ds = TopHalfMissingnessDataset(_Plain(img), total_miss=miss / 100, rng=rng)
        assert np.array_equal(ds.miss_mask, g[f"tophalf_{miss}"])


def check_data_module_against_fixture():
    """The data module end to end (split loading, missingness, fully-missing filter, index wrapper, pre-imputation, in
    the reference's order of generator draws: vgiwae/data/missing_data_module.py:171-312) against what the reference's
    own module produced on the toy mixture data it ships (tests/golden/toy/ holds that data file).  Masks and indices
    bit-exact; data values exact for oracle / zero pre-imputation, to float32 rounding for column means."""
    from vgiwae.data import MissingDataModule
    from vgiwae.data.missing_data_module import DATASET, MISSINGNESS, PRE_IMPUTATION
    g = gu.load("datamodule")
    checked = 0
    for line in g["configs"]:
        tag, miss, mt, mv, pre, pre_val = str(line).split(",")
        dm = MissingDataModule(dataset=DATASET.toy_mog2_large, batch_size=64, missingness=MISSINGNESS(miss),
                               total_miss_train=float(mt), total_miss_val=float(mv), pre_imputation=PRE_IMPUTATION(pre),
                               pre_impute_val=pre_val == "True", data_root=gu.GOLDEN, setup_seed=int(g["setup_seed"]),
                               total_miss_test=0.0)
        dm.setup("fit")
        dm.setup("test")
        for split, ds in (("train", dm.train_data), ("val", dm.val_data), ("test", dm.test_data)):
            items = ds[:]
            Xs, Ms, idx = np.asarray(items[0]), np.asarray(items[1]), np.asarray(items[-1])
            k = f"{tag}.{split}"
            assert len(ds) == int(g[k + ".n"])
            assert Ms.dtype == np.bool_ and np.array_equal(np.packbits(Ms.astype(np.uint8), axis=None), g[k + ".mask_bits"])
            assert idx.dtype == np.int64 and np.array_equal(idx, g[k + ".index"])
            if pre == "mean":
                np.testing.assert_allclose(Xs[::37], g[k + ".x_sub"], rtol=1e-6, atol=1e-7)
            else:
                assert np.array_equal(Xs[::37].astype(np.float32), g[k + ".x_sub"])
            np.testing.assert_allclose(Xs.astype(np.float64).sum(0), g[k + ".x_colsum"], rtol=1e-6)
            checked += 1
    assert checked == 9


def test_data_module_matches_reference_fixture():
    check_data_module_against_fixture()


def test_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/dmv.h declares."""
    header = open(os.path.join(ROOT, "include", "dmv.h")).read()
    declared = set(re.findall(r"\b(dmv_[a-z0-9_]+)\s*\(", header))
    from demiss_vae_amd import _lib
    lib = ctypes.CDLL(_lib.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/dmv.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    l = _lib.lib()
    assert l.dmv_version() == 100 and l.dmv_pad32(33) == 64
    assert b"workspace" in l.dmv_strerror(-3)
    # pure host entry points: sizes follow the documented layout
    net = _lib.DmvNet(16, 16, 256, 2, 0, 0.0)
    # forward + transposed fragment images, then one 64-lane bias fragment per 32-wide output tile
    per_hidden = 2 * (8 * 32 * 256) + 8 * 64
    assert l.dmv_mlp_packed_floats(ctypes.byref(net)) == (2 * 8 * 2 * 256 + 8 * 64) + 4 * per_hidden + (2 * 32 * 256 + 64)
    assert l.dmv_mlp_saved_floats(ctypes.byref(net), 100) == 128 * 32 + 5 * 128 * (256 + 8)  # rows padded to 32; 8 sign-bit words per row and layer
    assert l.dmv_mlp_packed_floats(ctypes.byref(_lib.DmvNet(16, 16, 512, 2, 0, 0.0))) > 0    # 257..512: relu only
    assert l.dmv_mlp_packed_floats(ctypes.byref(_lib.DmvNet(16, 16, 512, 2, 0, 0.5))) > 0
    assert l.dmv_mlp_packed_floats(ctypes.byref(_lib.DmvNet(16, 16, 512, 2, 2, 0.0))) == 0
    bad = _lib.DmvNet(16, 16, 513, 2, 0, 0.0)
    assert l.dmv_mlp_packed_floats(ctypes.byref(bad)) == 0          # hidden width outside the kernels' range


def test_enum_names_match_config_surface():
    from vgiwae.shared.vae_enums import DISTRIBUTION, mixture_components, engine_name
    assert DISTRIBUTION["normal_with_eps"].value == 7 and DISTRIBUTION["bern_m1_p1"].value == 14
    assert mixture_components(DISTRIBUTION["stratified_mixture25_normal_with_eps"]) == 25
    assert engine_name(DISTRIBUTION["stratified_mixture5_normal"]) == "stratified"
    assert len(DISTRIBUTION) == 22


def test_model_classes_construct_on_cpu_and_forward_is_gpu_only():
    import pytest
    import vgiwae.models as models
    from vgiwae.shared.neural_nets import ResidualFCNetwork
    from vgiwae.shared.vae_enums import DISTRIBUTION
    from vgiwae.shared.vae_imputation import VAESampler_LAIR, VAESampler_PseudoGibbs
    enc, dec = ResidualFCNetwork(5, 4, 3, 200), ResidualFCNetwork(2, 10, 3, 200)
    assert [k for k in enc.state_dict()][:2] == ["initial_layer.weight", "initial_layer.bias"]
    assert "blocks.2.linear_layers.1.bias" in enc.state_dict() and "final_layer.weight" in enc.state_dict()
    sampler = VAESampler_LAIR(num_iterations=1, num_imp_samples=1, num_prior_replenish_proposals=0, is_weights="dmis",
                              resampling_scheme="global_resampling")
    m = models.MVBVAEMargLogprob(generator_network=dec, generator_distribution=DISTRIBUTION.normal, var_latent_network=enc,
                                 var_latent_distribution=DISTRIBUTION.normal, num_latent_samples=1, lr_generator=1e-3,
                                 lr_latent=1e-3, var_latent_STL=True, kl_analytic=False, num_imputations_train=5,
                                 mvb_objective=models.MVB_OBJECTIVE["mvb2"], vae_sampler=sampler, latent_grid_steps=500)
    assert m.hparams.num_imputations_train == 5 and m.hparams.latent_grid_steps == 500 and not m.automatic_optimization
    assert m.model_dims().D == 5 and m.model_dims().Zd == 2
    with pytest.raises(RuntimeError):
        enc(torch.zeros(3, 5))                         # no CPU compute path
    with pytest.raises(NotImplementedError):
        VAESampler_LAIR(1, 1, 0, "smis", "global_resampling")
    assert VAESampler_PseudoGibbs(num_iterations=1).num_iterations == 1


def test_conv_trunks_load_reference_state_dicts():
    """The convolutional trunks take the reference's state_dict keys and shapes unchanged (strict load), and map
    [.., 784] <-> latent parameters with the reference's shape conventions."""
    from vgiwae.shared.vae_resnet import ResNetDecoder, ResNetEncoder
    for tag, use_mask in (("noencm", False), ("encm", True)):
        g = gu.load(f"resnet_mvb2_{tag}")
        Zd = int(g["Zd"])
        enc = ResNetEncoder(input_shape=[1, 28, 28], layers=[1, 1], layer_widths=[4, 8], latent_dim=2 * Zd,
                            expect_miss_mask_as_extra_chennel=use_mask)
        dec = ResNetDecoder(output_shape=[1, 28, 28], layers=[1, 1], layer_upscale=[None, None], layer_widths=[8, 4],
                            downscaled_resolution=[4, 4], latent_dim=Zd, upscale_factor=2)
        enc.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("enc.")}, strict=True)
        dec.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("dec.")}, strict=True)
        X = torch.from_numpy(g["Xt"])
        M = torch.from_numpy(g["M"]).float()
        with torch.no_grad():
            p = enc(X, M) if use_mask else enc(X)
            assert p.shape == (X.shape[0], X.shape[1], 2 * Zd)
            assert dec(p[..., :Zd]).shape == X.shape
    # the full-size trunks of configs/mnist_bin/compare/q50/mvb2vae_noencm_pg_k5_stl.yaml
    dec = ResNetDecoder(output_shape=[1, 28, 28], layers=[2, 2, 2, 2], layer_widths=[128, 64, 32, 32],
                        layer_upscale=[7, None, None, None], upscale_factor=7, downscaled_resolution=[4, 4], latent_dim=50,
                        dropout_prob=0.3).eval()
    with torch.no_grad():
        assert dec(torch.zeros(2, 50)).shape == (2, 784)
    assert "layers.0.0.upsample.0.1.weight" in dec.state_dict() and "layers.3.0.upsample.0.weight" not in dec.state_dict()
