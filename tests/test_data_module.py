"""Data module, device loader and config instantiation on the CPU (reference: vgiwae/data/missing_data_module.py:171-312,
train.py:40-78).  The reference's data package does not import here (it needs torchvision and matplotlib), so the
composition below is checked against the already pinned parts (masks, filter, index wrapper) and by properties;
pre-imputation and the composition order are parity unpinned."""
import numpy as np
import pytest
import torch

from demiss_vae_amd import run
from demiss_vae_amd.loader import DeviceLoader
from vgiwae.data import (DATASET, MISSINGNESS, PRE_IMPUTATION, DatasetWithIndices, FullyMissingDataFilter, MissingDataModule,
                         UniformMissingnessDataset)
from vgiwae.data.datasets import ToyDataset, UCITable


@pytest.fixture
def data_root(tmp_path):
    import scipy.io as sio
    r = np.random.default_rng(0)
    (tmp_path / "toy").mkdir()
    sio.savemat(tmp_path / "toy" / "data_mog2_large.mat",
                {s: r.normal(size=(n, 5)).astype(np.float32) for s, n in (("train", 403), ("val", 57), ("test", 61))})
    (tmp_path / "gas").mkdir()
    for s, n in (("train", 300), ("val", 40), ("test", 50)):
        np.savez(tmp_path / "gas" / f"{s}.npz", r.normal(size=(n, 8)).astype(np.float32))
    return str(tmp_path)


def module(root, dataset="toy_mog2_large", pre="oracle", pre_val=False, **kw):
    dm = MissingDataModule(dataset=DATASET[dataset], batch_size=64, missingness=MISSINGNESS.uniform, total_miss_train=0.5,
                           total_miss_val=0.5, pre_imputation=PRE_IMPUTATION(pre), pre_impute_val=pre_val, data_root=root,
                           setup_seed=20220118, **kw)
    dm.device = torch.device("cpu")
    return dm


def test_fit_setup_matches_the_pinned_parts(data_root):
    dm = module(data_root)
    dm.setup("fit")
    # the same chain built by hand from the mask / filter / index classes, sharing one generator train -> val
    rng = torch.Generator().manual_seed(20220118)
    for split, got in (("train", dm.train_data), ("val", dm.val_data)):
        core = ToyDataset(data_root, filename="data_mog2_large", split=split)
        want = DatasetWithIndices(FullyMissingDataFilter(UniformMissingnessDataset(core, target_idx=0, total_miss=0.5, rng=rng),
                                                         miss_mask_idx=1))
        for a, b in zip(got[:], want[:]):
            assert np.array_equal(a, b)
        X, M, I = got[:]
        assert X.dtype == np.float32 and M.dtype == np.bool_ and I.dtype == np.int64
        assert M.any(axis=1).all() and np.array_equal(I, np.arange(len(I)))
        assert abs(M.mean() - 0.5) < 0.05
    assert dm.train_data_core.data_min.shape == (5,)


def test_uci_table_and_test_stage(data_root):
    dm = module(data_root, dataset="uci_gas", total_miss_test=0.0, test_batch_size=7, use_test_instead_train=True)
    dm.setup("fit")
    assert len(dm.train_data) == 50 and len(dm.val_data) <= 40          # the test split stands in for the training data
    assert np.array_equal(dm.train_data_core.data, UCITable(data_root, "gas", "train").data)
    X, M, I = dm.test_data[:]
    assert M.all() and X.shape == (50, 8)
    batches = list(dm.test_dataloader())
    assert len(batches) == 8 and batches[-1][0].shape == (1, 8)
    with pytest.raises(NotImplementedError):
        dm.setup("predict")


@pytest.mark.parametrize("pre", ["mean", "zero", "empirical"])
def test_pre_imputation(data_root, pre):
    ref = module(data_root)
    ref.setup("fit")
    X0, M0, _ = (np.array(a) for a in ref.train_data[:])
    V0 = np.array(ref.val_data[:][0])
    dm = module(data_root, pre=pre)
    dm.setup("fit")
    X, M, _ = dm.train_data[:]
    assert np.array_equal(M, M0) and np.array_equal(X[M], X0[M0])       # masks and observed values are untouched
    assert np.array_equal(dm.val_data[:][0], V0)                       # pre_impute_val is off
    for d in range(X.shape[1]):
        miss, obs = X[~M[:, d], d], X0[M0[:, d], d]
        if pre == "mean":
            np.testing.assert_allclose(miss, obs.mean(), rtol=1e-5)
        elif pre == "zero":
            assert (miss == 0).all()
        else:
            # draws are without replacement (torch.multinomial's default, as in the reference), so only the first
            # len(obs) of them are guaranteed to be observed values
            assert np.isin(miss, obs).sum() >= min(len(miss), len(obs)) and len(np.unique(miss)) > 10
    if pre == "empirical":                                              # seeded: the same draws again
        again = module(data_root, pre=pre)
        again.setup("fit")
        assert np.array_equal(again.train_data[:][0], X)


def test_device_loader_layout_and_epochs(data_root):
    dm = module(data_root)
    dm.setup("fit")
    n = len(dm.train_data)
    loader = dm.train_dataloader()
    assert len(loader) == -(-n // 64)
    X, M, I = dm.train_data[:]
    seen = []
    for b in loader:
        assert [t.dtype for t in b] == [torch.float32, torch.bool, torch.int64]
        assert np.array_equal(b[0].numpy(), X[b[2].numpy()]) and np.array_equal(b[1].numpy(), M[b[2].numpy()])
        seen.append(b[2])
    assert seen[-1].numel() == n - 64 * (len(loader) - 1)
    order = torch.cat(seen)
    assert np.array_equal(np.sort(order.numpy()), np.arange(n)) and not np.array_equal(order.numpy(), np.arange(n))
    assert not torch.equal(torch.cat([b[2] for b in loader]), order)                      # a new permutation per epoch
    val = torch.cat([b[2] for b in dm.val_dataloader()])
    assert torch.equal(val, torch.arange(len(dm.val_data)))
    assert len(DeviceLoader(dm.train_data, 64, False, "cpu", drop_last=True)) == n // 64


def test_device_loader_shards_are_disjoint_equal_and_fixed(data_root):
    dm = module(data_root)
    dm.setup("fit")
    n = len(dm.train_data)
    shards = [DeviceLoader(dm.train_data, 16, True, "cpu", shard=(r, 3)) for r in range(3)]
    ids = [set(torch.cat([b[2] for b in s]).tolist()) for s in shards]
    assert all(len(i) == n // 3 for i in ids) and len(set.union(*ids)) == 3 * (n // 3)
    assert len({len(s) for s in shards}) == 1
    assert ids[1] == set(torch.cat([b[2] for b in shards[1]]).tolist())                   # same rows next epoch


def test_image_datasets_and_block_missingness(tmp_path):
    """Omniglot / MNIST files in the reference's npz layout (synthetic images): resize, binarisation, quadrant and
    top-half masks through the data module"""
    r = np.random.default_rng(0)
    (tmp_path / "omniglot").mkdir()
    (tmp_path / "MNIST").mkdir()
    for split, n in (("train", 40), ("val", 10), ("test", 12)):
        img = np.clip(r.random((n, 105, 105)).astype(np.float32) * 1.4 - 0.2, 0, 1)
        img[:, :, :50] = 1.0                                                      # white left half
        np.savez(tmp_path / "omniglot" / f"{split}.npz", data=img, targets=np.arange(n))
        np.savez(tmp_path / "MNIST" / f"{split}.npz", data=r.integers(0, 256, (n, 28, 28), dtype=np.uint8), targets=np.arange(n) % 10)
    kw = dict(batch_size=8, total_miss_train=0.5, total_miss_val=0.5, pre_imputation=PRE_IMPUTATION.oracle, pre_impute_val=False,
              data_root=str(tmp_path), setup_seed=20220118, img_dims=[28, 28])
    dm = MissingDataModule(dataset=DATASET.omniglot_28x28_bin, missingness=MISSINGNESS.quadrants, **kw)
    dm.device = torch.device("cpu")
    dm.setup("fit")
    X, M, T, I = dm.train_data[:]
    assert X.shape == (40, 784) and set(np.unique(X)) == {-1.0, 1.0} and dm.train_data_core.img_shape == (28, 28)
    assert (X.reshape(40, 28, 28)[:, :, :12] == 1).all()                          # the white half survives resize + binarisation
    quads = M.reshape(40, 2, 14, 2, 14).transpose(0, 1, 3, 2, 4).reshape(40, 4, 196)
    assert ((quads.all(-1) | (~quads).all(-1)).all()) and ((~quads).all(-1).sum(1) == 2).all()   # two whole quadrants missing
    again = MissingDataModule(dataset=DATASET.omniglot_28x28_bin, missingness=MISSINGNESS.quadrants, **kw)
    again.setup("fit")
    assert np.array_equal(again.train_data[:][0], X) and np.array_equal(again.train_data[:][1], M)
    batch = next(iter(dm.train_dataloader()))
    assert [t.shape for t in batch] == [(8, 784), (8, 784), (8,), (8,)]
    small = MissingDataModule(dataset=DATASET.mnist_14x14, missingness=MISSINGNESS.top_half, **kw)
    small.setup("fit")
    Xs, Ms = small.train_data[:][:2]
    assert Xs.shape == (40, 196) and 0 <= Xs.min() and Xs.max() < 1
    assert not Ms[:, :98].any() and Ms[:, 98:].all()


CONFIG = """
seed_everything: 20220118
experiment_subdir_base: logs/toy/mvb2
data:
  setup_seed: 20220118
  dataset: toy_mog2_large
  batch_size: 64
  missingness: 'uniform'
  total_miss_train: 0.5
  total_miss_val: 0.5
  pre_imputation: 'oracle'
  pre_impute_val: false
  data_root: ./data
model:
  class_path: vgiwae.models.MVBVAEMargLogprob
  init_args:
    mvb_objective: mvb2
    mvb2_optim_gradclip_val: 5
    var_latent_STL: True
    kl_analytic: False
    num_latent_samples: 2
    lr_generator: 1.0e-3
    lr_latent: 1.0e-3
    use_lr_scheduler: True
    num_imputations_train: 5
    num_imputations_val: 5
    imputation_init_train: empirical
    imputation_init_val: empirical
    init_imputation_distribution: True
    save_imputations: True
    latent_grid_steps: 50
    compute_complete_var_distribution_kl: True
    vae_sampler:
      class_path: vgiwae.shared.vae_imputation.VAESampler_PseudoGibbs
      init_args:
        num_iterations: 3
        clip_imputations: False
    generator_network:
      class_path: vgiwae.shared.neural_nets.ResidualFCNetwork
      init_args: {input_dim: 2, output_dim: 10, num_residual_blocks: 1, residual_block_dim: 32, dropout_probability: 0.0}
    generator_distribution: normal_with_eps
    var_latent_network:
      class_path: vgiwae.shared.neural_nets.ResidualFCNetwork
      init_args: {input_dim: 10, output_dim: 4, num_residual_blocks: 1, residual_block_dim: 32}
    var_latent_distribution: normal_with_eps
    encoder_use_mis_mask: True
trainer:
  max_epochs: 2
  gradient_clip_val: 5
  gradient_clip_algorithm: norm
"""


def test_config_instantiation(tmp_path):
    from vgiwae.models import MVB_OBJECTIVE
    from vgiwae.shared.vae_enums import DISTRIBUTION
    from vgiwae.shared.vae_imputation import VAESampler_PseudoGibbs
    from vgiwae.utils.imputation_distribution import IMPUTATION_INIT
    path = tmp_path / "cfg.yaml"
    path.write_text(CONFIG)
    cfg = run.load_config(str(path), [("data.data_root", "/somewhere"), ("trainer.max_epochs", "3"),
                                      ("model.init_args.lr_latent", "2e-4")])
    assert cfg["data"]["data_root"] == "/somewhere" and cfg["trainer"]["max_epochs"] == 3
    dm = run.construct(MissingDataModule, cfg["data"])
    assert dm.hparams.dataset is DATASET.toy_mog2_large and dm.hparams.missingness is MISSINGNESS.uniform
    assert dm.hparams.pre_imputation is PRE_IMPUTATION.oracle
    m = run.instantiate(cfg["model"])
    h = m.hparams
    assert type(m).__name__ == "MVBVAEMargLogprob" and not m.automatic_optimization
    assert h.mvb_objective is MVB_OBJECTIVE.mvb2 and h.generator_distribution is DISTRIBUTION.normal_with_eps
    assert h.imputation_init_train is IMPUTATION_INIT.empirical and isinstance(h.vae_sampler, VAESampler_PseudoGibbs)
    assert h.lr_latent == 2e-4 and h.latent_grid_steps == 50 and m.generator_network.output_dim == 10
    assert run._parse_overrides(["--a.b", "1", "--c=2"]) == [("a.b", "1"), ("c", "2")]


def test_checkpoint_round_trip_and_path_resolution(tmp_path):
    path = tmp_path / "cfg.yaml"
    path.write_text(CONFIG)
    cfg = run.load_config(str(path))
    m = run.instantiate(cfg["model"])
    exp = str(tmp_path / "logs/toy/mvb2/seed_m20220118_d20220118")
    for want in ("version_0", "version_1"):
        out = run.next_version_dir(exp)
        assert out.endswith(want)
        run.save_checkpoint(m, out + "/checkpoints/last.ckpt", cfg["model"], {"epoch": 2})
    template = str(tmp_path / "logs/toy/mvb2/seed_{}/lightning_logs/version_{}/checkpoints/last.ckpt")
    found = run.resolve_model_path(template, "./logs/toy/refit/seed_m20220118_d20220118")
    assert found == exp + "/lightning_logs/version_1/checkpoints/last.ckpt"
    back = run.load_model(found, "MVBVAE")
    assert type(back).__name__ == "MVBVAE"
    from vgiwae.models import MVBVAE_FID
    assert type(MVBVAE_FID.load_from_checkpoint(found)).__name__ == "MVBVAE_FID"
    for (k, a), (k2, b) in zip(m.state_dict().items(), back.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    ck = torch.load(found, weights_only=True)
    assert ck["hyper_parameters"]["mvb_objective"] == "mvb2" and ck["epoch"] == 2


def test_gradient_variance_probe_matches_the_stacked_gradient_formula(tmp_path):
    """*GradVar classes: running sums reproduce std / mean over the stacked per-step gradients
    (vgiwae/shared/vae_grad_var.py:20-45), dead coordinates left out of the signal-to-noise sum"""
    path = tmp_path / "cfg.yaml"
    path.write_text(CONFIG.replace("MVBVAEMargLogprob", "MVBVAEGradVar").replace("    latent_grid_steps: 50\n", "")
                    .replace("    compute_complete_var_distribution_kl: True\n", ""))
    m = run.instantiate(run.load_config(str(path))["model"])
    gen = torch.Generator().manual_seed(0)
    for epoch in range(2):
        m.on_train_epoch_start()
        stacks = {"generator": [], "encoder": []}
        for step in range(7):
            for name, net in (("generator", m.generator_network), ("encoder", m.var_latent_network)):
                for i, p in enumerate(net.parameters()):
                    p.grad = torch.randn(p.shape, generator=gen) * (0.0 if i == 1 else 1.0 + step)     # one dead tensor
                stacks[name].append(torch.cat([p.grad.flatten() for p in net.parameters()]))
            m.on_before_optimizer_step(None, 0)
        m._logged = {}
        m.train_imp_distribution = m.val_imp_distribution = None       # the DeMissVAE epoch-end hook is not under test
        m.resample_imputations = lambda stage: None
        m.on_train_epoch_end()
        for name in ("generator", "encoder"):
            g = torch.stack(stacks[name]).double()
            std = g.std(0, unbiased=True)
            snr = (g.mean(0) / std).abs()
            np.testing.assert_allclose(float(m._logged[f"{name}_grad_var/train"]), float((std ** 2).sum()), rtol=1e-5)
            np.testing.assert_allclose(float(m._logged[f"{name}_grad_snr/train"]), float(snr[torch.isfinite(snr)].sum()), rtol=1e-5)
