"""The YAML runner end to end on the device: a DeMissVAE config (manual optimisation, imputation buffers, grid
marginal metric), an IWAE config (automatic optimisation, cosine schedule, clipping), then the refit-and-test config
over the IWAE run's checkpoint (reference entry points: train.py:40-113,
refit_encoder_using_iwae_and_estimate_loglik.py:46-138).  Data are synthetic files in a temporary directory."""
import json
import math
import os

import numpy as np
import pytest
import torch

from test_data_module import CONFIG

pytestmark = pytest.mark.gpu

IWAE_CONFIG = """
seed_everything: 20220118
experiment_subdir_base: logs/toy/iwae_i5
add_checkpoint_callback: True
save_custom_epochs: [1]
save_custom_steps: [5]
data:
  setup_seed: 20220118
  dataset: toy_mog2_large
  batch_size: 128
  missingness: 'uniform'
  total_miss_train: 0.5
  total_miss_val: 0.5
  pre_imputation: 'oracle'
  pre_impute_val: false
model:
  class_path: vgiwae.models.IWAE
  init_args:
    var_latent_STL: True
    num_latent_samples: 1
    num_importance_samples: 5
    lr_generator: 1e-3
    lr_latent: 1e-3
    use_lr_scheduler: True
    generator_network:
      class_path: vgiwae.shared.neural_nets.ResidualFCNetwork
      init_args: {input_dim: 2, output_dim: 10, num_residual_blocks: 1, residual_block_dim: 64}
    generator_distribution: normal_with_eps
    var_latent_network:
      class_path: vgiwae.shared.neural_nets.ResidualFCNetwork
      init_args: {input_dim: 10, output_dim: 4, num_residual_blocks: 1, residual_block_dim: 64}
    var_latent_distribution: normal_with_eps
    encoder_use_mis_mask: True
trainer:
  max_epochs: 6
  gradient_clip_val: 5
  gradient_clip_algorithm: norm
"""

REFIT_CONFIG = """
seed_everything: 20220118
experiment_subdir_base: logs/toy/refit/iwae_i5
data:
  setup_seed: 20220118
  dataset: toy_mog2_large
  use_test_instead_train: True
  batch_size: 128
  test_batch_size: 50
  missingness: 'uniform'
  total_miss_train: 0.5
  total_miss_val: 0.5
  total_miss_test: 0.0
  pre_imputation: 'oracle'
  pre_impute_val: false
refit_model:
  refit_encoder_class: standard
  var_latent_STL: True
  num_latent_samples: 1
  num_importance_samples: 50
  num_test_importance_samples: 4096
  lr_latent: 1e-4
  amsgrad_latent: True
  use_lr_scheduler: True
model_class: IWAE
load_best_model: True
model_path: ROOT/logs/toy/iwae_i5/seed_{}/lightning_logs/version_{}/checkpoints/last.ckpt
trainer:
  gradient_clip_val: 5
  gradient_clip_algorithm: norm
  max_steps: 30
"""


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    import scipy.io as sio
    root = tmp_path_factory.mktemp("runner")
    r = np.random.default_rng(1)
    (root / "data" / "toy").mkdir(parents=True)

    # a three-component mixture of Gaussians with its parameters in the file, like the reference's toy data
    C, D = 3, 5
    A = r.normal(size=(C, D, D))
    covs = A @ A.transpose(0, 2, 1) / D + 0.2 * np.eye(D)
    means = r.normal(size=(C, D)) * 2
    probs = np.array([[0.5, 0.3, 0.2]])

    def draw(n):
        c = r.choice(C, size=n, p=probs[0])
        return (means[c] + np.einsum("nij,nj->ni", np.linalg.cholesky(covs)[c], r.normal(size=(n, D)))).astype(np.float32)
    sio.savemat(root / "data" / "toy" / "data_mog2_large.mat", {"train": draw(2000), "val": draw(300), "test": draw(200),
                                                                "comp_probs": probs, "means": means, "covs": covs})
    return root


def launch(workdir, name, text, *extra, recordable=True, min_replays=lambda steps: steps - 8):
    """min_replays: every (batch shape, imputing or not) kind of step runs eagerly twice before it is recorded"""
    from demiss_vae_amd import run
    path = workdir / f"{name}.yaml"
    path.write_text(text.replace("ROOT", str(workdir)))
    lines = []
    history, results = run.main(["--config", str(path), "--data.data_root", str(workdir / "data"),
                                 "--trainer.default_root_dir", str(workdir), *extra], log=lines.append)
    summary = [json.loads(l) for l in lines if str(l).startswith('{"steps"')][0]
    if recordable is None:      # networks that are not residual MLPs: no recording is attempted
        assert summary["hip_graph_replays"] == 0
    elif recordable:      # every batch shape runs eagerly twice, then as a replay of its recording
        assert not any("not recordable" in str(l) for l in lines)
        assert summary["hip_graph_replays"] >= min_replays(summary["steps"]), summary
    else:               # a step that synchronises with the host falls back to eager execution, loudly
        assert any("not recordable" in str(l) for l in lines) and summary["hip_graph_replays"] < summary["steps"]
    return history, results, lines


def test_demissvae_config_trains(workdir):
    history, _, lines = launch(workdir, "mvb2", CONFIG)
    assert len(history) == 2 and history[-1]["step"] == 2 * math.ceil(history[-1]["step"] / 2)
    for row in history:
        assert {"mvb/train", "cvi/train", "loss/train", "marginal_logprob/val", "complete_marginal_logprob/val",
                "complete_posterior_kl_fow/val", "incomplete_posterior_jsd/val", "complete_imps_posterior_kl_rev/val"} <= set(row)
        assert min(v for k, v in row.items() if "posterior" in k) > -1e-3
        assert all(math.isfinite(v) for v in row.values())
    assert history[1]["loss/train"] < history[0]["loss/train"]
    assert history[1]["marginal_logprob/val"] > history[0]["marginal_logprob/val"]
    out = workdir / "logs/toy/mvb2/seed_m20220118_d20220118/lightning_logs/version_0"
    ck = torch.load(out / "checkpoints" / "last.ckpt", weights_only=True)
    assert ck["epoch"] == 2 and any(k.startswith("generator_network.") for k in ck["state_dict"])
    assert [json.loads(l)["epoch"] for l in (out / "metrics.jsonl").read_text().splitlines()] == [0, 1]
    imps = np.load(out / "imputations" / "train_imputations_last.npz")                     # save_imputations: True
    assert imps["data_imps"].shape == (1936 * 5, 5) and int(imps["num_imputations"]) == 5


def test_control_study_config_with_oracle_imputations(workdir):
    """the same model with imputations drawn from a degraded version of the true conditional mixture"""
    text = CONFIG.replace("logs/toy/mvb2", "logs/toy/oracle").replace(
        """      class_path: vgiwae.shared.vae_imputation.VAESampler_PseudoGibbs
      init_args:
        num_iterations: 3
        clip_imputations: False
""", """      class_path: vgiwae.utils.mog_utils.VAESampler_MoG_Oracle
      init_args:
        start_distribution: indep_marginal_mog
        interpolate_method: linear
        interpolate_target: conditional_independent_mog
        interpolate_alpha: 0.5
""")
    assert "VAESampler_MoG_Oracle" in text
    # the oracle sampler's batched factorisations synchronise with the host, so its steps are not recorded
    history, _, _ = launch(workdir, "oracle", text, recordable=False)
    assert len(history) == 2 and all(math.isfinite(v) for row in history for v in row.values())
    assert history[1]["marginal_logprob/val"] > history[0]["marginal_logprob/val"]


def test_iwae_config_then_refit_and_test_loglik(workdir):
    history, _, _ = launch(workdir, "iwae", IWAE_CONFIG)
    assert len(history) == 6 and history[-1]["step"] == 6 * 16
    assert history[-1]["elbo/val"] > history[0]["elbo/val"] and history[-1]["elbo/train"] > history[0]["elbo/train"]

    ckpts = sorted(os.listdir(workdir / "logs/toy/iwae_i5/seed_m20220118_d20220118/lightning_logs/version_0/checkpoints"))
    best = [c for c in ckpts if c.startswith("epoch=")]
    assert len(best) == 1 and set(ckpts) == {best[0], "last.ckpt", "custom_ckpt_by_epoch_1.ckpt", "custom_ckpt_by_step_5.ckpt"}
    for c in ("custom_ckpt_by_epoch_1.ckpt", "custom_ckpt_by_step_5.ckpt"):       # the refit looks for one file besides last
        os.remove(workdir / "logs/toy/iwae_i5/seed_m20220118_d20220118/lightning_logs/version_0/checkpoints" / c)

    refit_history, results, lines = launch(workdir, "refit", REFIT_CONFIG)
    assert any("Using pre-trained model from" in l and "version_0" in l and best[0] in l for l in lines)
    assert refit_history[-1]["step"] == 30                      # max_steps ends the run inside an epoch
    ll = results["est_loglik/test"]
    # complete test data, 4096 importance samples: above the 5-sample bound on half-missing validation data per
    # dimension is not comparable, so compare with the complete-data 50-sample bound of the refit's training rows
    assert math.isfinite(ll) and ll > refit_history[-1]["elbo/train"] - 0.05
    # the generator is frozen during the refit: only the encoder moved
    before = torch.load(workdir / "logs/toy/iwae_i5/seed_m20220118_d20220118/lightning_logs/version_0/checkpoints" / best[0],
                        weights_only=True)["state_dict"]
    after = torch.load(workdir / "logs/toy/refit/iwae_i5/seed_m20220118_d20220118/lightning_logs/version_0/checkpoints/last.ckpt",
                       weights_only=True)["state_dict"]
    moved = {k.split(".")[0] for k in before if not torch.equal(before[k], after[k])}
    assert moved == {"var_latent_network"}


def test_two_rank_sharded_loglik_equals_one_rank(workdir):
    """The K-sample test log-likelihood under torch.distributed (two ranks sharing the GPU, gloo): test points dealt to
    the ranks, or the importance samples of every batch split over the ranks and combined with the logsumexp
    all-reduce, give the one-rank value (device noise is addressed by batch and global sample number, so the draws
    are the same in every layout); with the reference's recorded noise the K-sharded estimator reproduces the
    reference's est_loglik of the golden cases."""
    import subprocess
    import sys
    ckdir = workdir / "logs/toy/iwae_i5/seed_m20220118_d20220118/lightning_logs/version_0/checkpoints"
    if not (ckdir / "last.ckpt").exists():
        launch(workdir, "iwae", IWAE_CONFIG)
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loglik_shard_worker.py")
    args = [worker, str(ckdir / "last.ckpt"), str(workdir / "data"), "4096"]
    env = dict(os.environ, DMV_DIST_BACKEND="gloo", PYTHONPATH=os.pathsep.join(filter(None, [os.getcwd(), os.environ.get("PYTHONPATH")])))

    def run_world(n):
        cmd = [sys.executable] + args if n == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
             "--master-port", "29541"] + args
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=110, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        return [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][0]

    one, two = run_world(1), run_world(2)
    assert one["world"] == 1 and two["world"] == 2
    ref = one["points"]
    assert math.isfinite(ref)
    for mode in ("points", "samples", "auto"):
        assert abs(one[mode] - ref) <= 1e-5 * abs(ref), (mode, one)
        assert abs(two[mode] - ref) <= 1e-5 * abs(ref), (mode, two, ref)
    for name in ("loglik_standard_K64", "loglik_stratified_K64", "loglik_standard_K4096"):
        for res in (one, two):
            got, want = res[name]
            assert abs(got - want) < 1e-4 * max(1.0, abs(want)), (name, res["world"], got, want)


def test_fid_metric_against_a_trained_encoder(workdir):
    """the *_FID classes: train a complete-data VAE, then a model whose FID is measured in its encoder's feature space"""
    vae = IWAE_CONFIG.replace("logs/toy/iwae_i5", "logs/toy/vae_complete").replace("vgiwae.models.IWAE", "vgiwae.models.VAE") \
        .replace("    num_importance_samples: 5\n", "    kl_analytic: False\n").replace("    encoder_use_mis_mask: True\n", "") \
        .replace("{input_dim: 10, output_dim: 4,", "{input_dim: 5, output_dim: 4,") \
        .replace("total_miss_train: 0.5", "total_miss_train: 0.0").replace("max_epochs: 6", "max_epochs: 2")
    launch(workdir, "vae_complete", vae)
    ckpt = workdir / "logs/toy/vae_complete/seed_m20220118_d20220118/lightning_logs/version_0/checkpoints/last.ckpt"
    assert ckpt.exists()
    fid = IWAE_CONFIG.replace("logs/toy/iwae_i5", "logs/toy/iwae_fid").replace("vgiwae.models.IWAE", "vgiwae.models.IWAE_FID") \
        .replace("    num_importance_samples: 5\n", f"""    num_importance_samples: 5
    fid_eval_step_freq: 8
    inception_model_type: VAE_encoder
    inception_model_path: {ckpt}
    reference_samples: [ dataset_train, dataset_val, dataset_test ]
    num_samples_per_batch: 500
""").replace("max_epochs: 6", "max_epochs: 3")
    history, _, _ = launch(workdir, "iwae_fid", fid)
    for row in history:
        assert {"fid_score/train", "fid_score/val", "fid_score/test"} <= set(row)
        assert all(math.isfinite(row[k]) and row[k] > -1e-3 for k in row if k.startswith("fid_score"))
    assert history[-1]["fid_score/train"] < history[0]["fid_score/train"]          # samples approach the data
    sd = torch.load(workdir / "logs/toy/iwae_fid/seed_m20220118_d20220118/lightning_logs/version_0/checkpoints/last.ckpt",
                    weights_only=True)["state_dict"]
    assert not any("inception" in k for k in sd)


def test_gradvar_class_logs_through_recorded_steps(workdir):
    """IWAEGradVar through the runner: the probe's accumulation is part of the recorded step"""
    text = IWAE_CONFIG.replace("logs/toy/iwae_i5", "logs/toy/iwae_gradvar").replace("vgiwae.models.IWAE", "vgiwae.models.IWAEGradVar") \
        .replace("max_epochs: 6", "max_epochs: 2")
    history, _, _ = launch(workdir, "iwae_gradvar", text)
    for row in history:
        for k in ("generator_grad_var/train", "encoder_grad_var/train", "generator_grad_snr/train", "encoder_grad_snr/train"):
            assert math.isfinite(row[k]) and row[k] > 0, (k, row)


def test_two_rank_data_parallel_run_keeps_replicas_identical(workdir):
    """torchrun with two ranks sharing the GPU (gloo collectives staged through the host): every rank trains on its fixed
    half of the split with half the batch, gradients are averaged before the clip, replicas stay bit-identical; the
    forward and backward passes of a step replay from a HIP graph, the collective stays outside it"""
    import subprocess
    import sys
    path = workdir / "iwae_dp.yaml"
    path.write_text(IWAE_CONFIG.replace("logs/toy/iwae_i5", "logs/toy/iwae_dp").replace("max_epochs: 6", "max_epochs: 3"))
    env = dict(os.environ, DMV_DIST_BACKEND="gloo", PYTHONPATH=os.pathsep.join(filter(None, [os.getcwd(), os.environ.get("PYTHONPATH")])))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29533", "-m", "demiss_vae_amd.run", "--config", str(path),
                          "--data.data_root", str(workdir / "data"), "--trainer.default_root_dir", str(workdir)],
                         capture_output=True, text=True, timeout=110, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    epochs, summary = [r for r in rows if "epoch" in r], [r for r in rows if "replica_spread" in r][0]
    assert summary["world"] == 2 and summary["replica_spread"] == 0.0
    # forward + backward replay from a recording; exchange, clipping and optimiser follow eagerly after each replay
    assert summary["hip_graph_replays"] >= summary["steps"] - 4
    assert [r["step"] for r in epochs] == [16, 32, 48]                  # 968 rows per rank, 64 per rank and step
    assert epochs[-1]["elbo/val"] > epochs[0]["elbo/val"]


def test_two_rank_data_parallel_demissvae_replays(workdir):
    """the two-loss DeMissVAE step under data parallelism: imputation update and both backward passes replay from the
    recording, the step is completed outside it (exchange, clipping, optimiser); replicas stay bit-identical"""
    import subprocess
    import sys
    path = workdir / "mvb2_dp.yaml"
    path.write_text(CONFIG.replace("logs/toy/mvb2", "logs/toy/mvb2_dp").replace("max_epochs: 2", "max_epochs: 3")
                    .replace("    save_imputations: True\n", ""))
    env = dict(os.environ, DMV_DIST_BACKEND="gloo", PYTHONPATH=os.pathsep.join(filter(None, [os.getcwd(), os.environ.get("PYTHONPATH")])))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29534", "-m", "demiss_vae_amd.run", "--config", str(path),
                          "--data.data_root", str(workdir / "data"), "--trainer.default_root_dir", str(workdir)],
                         capture_output=True, text=True, timeout=110, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    epochs, summary = [r for r in rows if "epoch" in r], [r for r in rows if "replica_spread" in r][0]
    assert summary["world"] == 2 and summary["replica_spread"] == 0.0
    assert summary["hip_graph_replays"] >= summary["steps"] - 8, summary
    assert all(math.isfinite(r["mvb/train"]) and math.isfinite(r["cvi/train"]) for r in epochs)
    assert epochs[-1]["cvi/train"] > epochs[0]["cvi/train"]


def test_stratified_mixture_config_with_dropout_replays(workdir):
    """MissSIWAE (stratified mixture encoder, dropout in both networks: the miniboone configs' structure) and its refit
    with the stratified refit class: recorded steps, dropout masks fresh on every replay, K-sample test estimate"""
    text = IWAE_CONFIG.replace("logs/toy/iwae_i5", "logs/toy/miwae").replace("vgiwae.models.IWAE", "vgiwae.models.MultipleIWAE") \
        .replace("    num_importance_samples: 5\n", "    num_importance_samples_for_each_component: 1\n") \
        .replace("var_latent_distribution: normal_with_eps", "var_latent_distribution: stratified_mixture5_normal_with_eps") \
        .replace("{input_dim: 10, output_dim: 4, num_residual_blocks: 1, residual_block_dim: 64}",
                 "{input_dim: 10, output_dim: 25, num_residual_blocks: 1, residual_block_dim: 64, dropout_probability: 0.3}") \
        .replace("{input_dim: 2, output_dim: 10, num_residual_blocks: 1, residual_block_dim: 64}",
                 "{input_dim: 2, output_dim: 10, num_residual_blocks: 1, residual_block_dim: 64, dropout_probability: 0.3}") \
        .replace("save_custom_epochs: [1]\n", "").replace("save_custom_steps: [5]\n", "").replace("max_epochs: 6", "max_epochs: 4")
    assert "stratified_mixture5" in text and text.count("dropout_probability: 0.3") == 2
    text = text.replace("    num_importance_samples_for_each_component: 1\n",
                        "    num_importance_samples_for_each_component: 1\n"
                        "    log_var_distribution_mixture_component_weights_epoch_frequency: 2\n")
    history, _, _ = launch(workdir, "miwae", text, min_replays=lambda steps: steps // 2 - 8)   # epochs 1 and 3 replay
    assert history[-1]["elbo/val"] > history[0]["elbo/val"]
    dump = np.load(workdir / "logs/toy/miwae/seed_m20220118_d20220118/lightning_logs/version_0/mixture_var_weight_logits.npz")
    assert dump["mixture_weight_logits"].shape[0] == 2 and dump["mixture_weight_logits"].shape[-1] == 5    # epochs 0 and 2
    refit = REFIT_CONFIG.replace("logs/toy/refit/iwae_i5", "logs/toy/refit/miwae").replace("logs/toy/iwae_i5", "logs/toy/miwae") \
        .replace("refit_encoder_class: standard", "refit_encoder_class: stratified").replace("model_class: IWAE", "model_class: MultipleIWAE")
    _, results, _ = launch(workdir, "refit_miwae", refit)
    assert math.isfinite(results["est_loglik/test"])


IMAGE_CONFIG = """
seed_everything: 20220118
experiment_subdir_base: logs/mnist_bin/mvb2vae_pg
add_checkpoint_callback: True
data:
  setup_seed: 20220118
  dataset: mnist_bin
  batch_size: 32
  missingness: quadrants
  img_dims: [ 28, 28 ]
  total_miss_train: 0.50
  total_miss_val: 0.50
  pre_imputation: 'oracle'
  pre_impute_val: false
model:
  class_path: vgiwae.models.MVBVAE_FID
  init_args:
    fid_eval_step_freq: 5000
    inception_model_type: VAE_encoder
    inception_model_path: ./nowhere/last.ckpt
    reference_samples: [ dataset_train, dataset_val, dataset_test ]
    mvb_sample_num_imputations: null
    mvb_objective: mvb2
    mvb_eval_val_loss: True
    mvb_delay_imputation: 1
    vae_sampler:
      class_path: vgiwae.shared.vae_imputation.VAESampler_PseudoGibbs
      init_args:
        num_iterations: 1
        clip_imputations: False
    generator_network:
      class_path: vgiwae.shared.vae_resnet.ResNetDecoder
      init_args:
        output_shape: [1, 28, 28]
        layers: [1, 1]
        layer_widths: [8, 4]
        layer_upscale: [null, null]
        upscale_factor: 2
        downscaled_resolution: [4, 4]
        latent_dim: 6
        dropout_prob: 0.1
    generator_distribution: bern_m1_p1
    prior_distribution: std_normal
    var_latent_network:
      class_path: vgiwae.shared.vae_resnet.ResNetEncoder
      init_args:
        input_shape: [1, 28, 28]
        layers: [1, 1]
        layer_widths: [4, 8]
        latent_dim: 12
        dropout_prob: 0.1
        expect_miss_mask_as_extra_chennel: False
    var_latent_distribution: normal_with_eps
    encoder_use_mis_mask: False
    kl_analytic: False
    var_latent_STL: True
    num_latent_samples: 1
    lr_generator: 1e-3
    lr_latent: 1e-3
    amsgrad_generator: True
    amsgrad_latent: True
    use_lr_scheduler: True
    init_imputation_distribution: True
    num_imputations_train: 3
    num_imputations_val: 3
    imputation_init_train: empirical
    imputation_init_val: empirical
    mvb2_optim_gradclip_val: 5
    mvb2_optim_gradclip_alg: norm
trainer:
  max_epochs: 3
"""


def test_image_config_with_convolutional_trunks(workdir):
    """the MNIST / Omniglot config family: binarised images, quadrant masks, convolutional trunks as torch modules with
    the kernels at their outputs, pseudo-Gibbs imputation from the second epoch on, validation loss, best checkpoint;
    the FID wrapper without its inception checkpoint skips the metric"""
    r = np.random.default_rng(3)
    (workdir / "data" / "MNIST").mkdir(exist_ok=True)
    proto = (r.random((4, 28, 28)) < 0.2)                                        # four "digits"
    for split, n in (("train", 192), ("val", 48), ("test", 48)):
        labels = r.integers(0, 4, n)
        img = np.where(proto[labels] ^ (r.random((n, 28, 28)) < 0.03), 230, 10).astype(np.uint8)
        np.savez(workdir / "data" / "MNIST" / f"{split}.npz", data=img, targets=labels)
    history, _, _ = launch(workdir, "mnist", IMAGE_CONFIG, recordable=None)
    assert len(history) == 3
    for row in history:
        assert {"mvb/train", "cvi/train", "loss/train", "loss/val", "mvb/val", "cvi/val"} <= set(row)
        assert all(math.isfinite(v) for v in row.values())
    assert history[-1]["loss/train"] < history[0]["loss/train"]
    ckpts = os.listdir(workdir / "logs/mnist_bin/mvb2vae_pg/seed_m20220118_d20220118/lightning_logs/version_0/checkpoints")
    assert "last.ckpt" in ckpts and sum(c.startswith("epoch=") for c in ckpts) == 1
    # refit the convolutional encoder and estimate the test log-likelihood (the image refit configs: 5 x 1000 samples)
    refit = REFIT_CONFIG.replace("logs/toy/refit/iwae_i5", "logs/mnist_bin/refit").replace("logs/toy/iwae_i5", "logs/mnist_bin/mvb2vae_pg") \
        .replace("dataset: toy_mog2_large", "dataset: mnist_bin\n  img_dims: [ 28, 28 ]").replace("missingness: 'uniform'", "missingness: quadrants") \
        .replace("batch_size: 128", "batch_size: 32").replace("test_batch_size: 50", "test_batch_size: 5") \
        .replace("num_importance_samples: 50\n", "num_importance_samples: 10\n").replace("num_test_importance_samples: 4096", "num_test_importance_samples: 200") \
        .replace("model_class: IWAE", "model_class: MVBVAE_FID").replace("max_steps: 30", "max_steps: 6")
    _, results, _ = launch(workdir, "mnist_refit", refit, recordable=None)
    assert math.isfinite(results["est_loglik/test"]) and results["est_loglik/test"] < 0


def _variant(name, cls, extra, q="normal_with_eps", enc_out=4, sampler=None, epochs=3):
    """a small config of another model family on the toy table, derived from the IWAE one"""
    text = IWAE_CONFIG.replace("logs/toy/iwae_i5", f"logs/toy/{name}").replace("vgiwae.models.IWAE", f"vgiwae.models.{cls}") \
        .replace("    num_importance_samples: 5\n", extra) \
        .replace("var_latent_distribution: normal_with_eps", f"var_latent_distribution: {q}") \
        .replace("{input_dim: 10, output_dim: 4,", "{input_dim: 10, output_dim: %d," % enc_out) \
        .replace("save_custom_epochs: [1]\n", "").replace("save_custom_steps: [5]\n", "").replace("max_epochs: 6", f"max_epochs: {epochs}")
    if sampler:
        text = text.replace("    generator_network:\n", sampler + "    generator_network:\n", 1)
    return text


LAIR = """    mvb_objective: mvb2
    mvb_eval_val_loss: True
    mvb_delay_imputation: 1
    mvb2_optim_gradclip_val: 1
    mvb2_optim_gradclip_alg: norm
    init_imputation_distribution: True
    num_imputations_train: 5
    num_imputations_val: 5
    imputation_init_train: empirical
    imputation_init_val: empirical
    vae_sampler:
      class_path: vgiwae.shared.vae_imputation.VAESampler_LAIR
      init_args:
        num_iterations: 4
        num_imp_samples: 1
        num_prior_replenish_proposals: 1
        resample_final_imps: True
        is_weights: dmis
        resampling_scheme: global_resampling
"""

FAMILIES = {
    # UCI DeMissVAE configs: LAIR with a prior-replenish proposal, several iterations, final resampling
    "uci_demissvae": ("MVBVAE", "    kl_analytic: False\n", "normal_with_eps", 4, LAIR, "loss/val"),
    "vae_analytic": ("VAE", "    kl_analytic: True\n", "normal_with_eps", 4, None, "elbo/val"),
    "missvae_stratified": ("MultipleVAE", "", "stratified_mixture5_normal_with_eps", 25, None, "elbo/val"),
    "reparam_iwae": ("IWAE", "    num_importance_samples: 5\n", "reparametrised_mixture5_normal_with_eps", 25, None, "elbo/val"),
    "mvbiwae": ("MVBIWAE", "    num_importance_samples: 3\n    mvb_objective: mvb2_encallweighteddmmis_decall\n" + LAIR.split("    mvb2_optim")[0].replace("    mvb_objective: mvb2\n", "")
                + LAIR[LAIR.index("    init_imputation"):], "normal_with_eps", 4, None, "loss/val"),
}


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_other_model_families_through_the_runner(workdir, family):
    cls, extra, q, enc_out, sampler, metric = FAMILIES[family]
    text = _variant(family, cls, extra, q=q, enc_out=enc_out, sampler=sampler)
    if family == "vae_analytic":
        text = text.replace("    var_latent_STL: True\n", "    var_latent_STL: False\n")
    if family == "uci_demissvae":
        text = text.replace("    encoder_use_mis_mask: True\n", "").replace("{input_dim: 10, output_dim: 4,", "{input_dim: 5, output_dim: 4,")
    history, _, _ = launch(workdir, family, text)
    assert len(history) == 3 and all(math.isfinite(v) for row in history for v in row.values())
    better = (lambda a, b: a < b) if metric.startswith("loss") else (lambda a, b: a > b)
    assert better(history[-1][metric], history[0][metric]), [r[metric] for r in history]


def test_validation_steps_replay_and_agree_with_eager_validation(workdir):
    """Validation steps of the plain bound classes are replayed from recordings (every batch shape runs eagerly
    once first); the validation bound of such a run agrees with a run whose steps all execute eagerly (other noise
    streams, same model and data: agreement to Monte-Carlo and trajectory noise)."""
    rows = {}
    for tag, extra in (("replayed", ()), ("eager", ("--trainer.hip_graph", "false"))):
        text = IWAE_CONFIG.replace("logs/toy/iwae_i5", f"logs/toy/iwae_val_{tag}")
        history, _, lines = launch(workdir, f"iwae_val_{tag}", text, *extra, min_replays=lambda steps: 0)
        summary = [json.loads(l) for l in lines if str(l).startswith('{"steps"')][0]
        rows[tag] = (history, summary)
    assert rows["eager"][1]["validation_replays"] == 0 and rows["eager"][1]["hip_graph_replays"] == 0
    epochs = len(rows["replayed"][0])
    val_batches = math.ceil(300 / 128)              # (fewer rows after the fully-missing filter never change the count here)
    assert rows["replayed"][1]["validation_replays"] >= (epochs - 1) * (val_batches - 1)
    a, b = rows["replayed"][0][-1]["elbo/val"], rows["eager"][0][-1]["elbo/val"]
    print("validation bound, replayed vs eager:", a, b)
    assert math.isfinite(a) and abs(a - b) < 0.3, (a, b)


def _bench(args, env_extra, launcher=()):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, *launcher, os.path.join(root, "bench.py"), *args], capture_output=True, text=True,
                         timeout=170, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE json line, from rank 0
    return json.loads(lines[0])


def test_bench_line_under_torchrun_with_two_ranks():
    """`bench.py --gpus 2` launched as the driver launches it (torchrun, one rank per GPU; here both ranks on the one
    GPU with gloo collectives): one JSON line, n_gpus = 2, whole-job value = 2 x batch x K / step time"""
    row = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "4096", "--no-cpu"],
                 {"DMV_BENCH_BACKEND": "gloo"},
                 launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29537"))
    assert row["n_gpus"] == 2 and row["steps"] == 5 and row["scaling"] == "weak" and row["config"]["parallelism"] == "dp2"
    assert abs(row["value"] - 2 * 4096 * 5 / (row["ms_per_step"] * 1e-3)) < 1e-3 * row["value"]
    assert row["extra"]["loglik_eval_K50000"]["samples_per_s"] > 0


def test_bench_starts_its_own_ranks_for_gpus_2():
    """`python bench.py --gpus 2` with NO launcher around it: the script starts two ranks itself (a torch.distributed.run
    child, before it touches the GPU; here both ranks on the one GPU with gloo collectives) and relays rank 0's one
    line, which carries every record of the one-GPU line plus the rank count the collective library saw and the
    log-likelihood rate with the K samples split over the ranks"""
    row = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "4096", "--cpu-budget", "2"],
                 {"DMV_BENCH_BACKEND": "gloo"})
    assert row["n_gpus"] == 2 and row["rccl_ranks"] == 2 and row["config"]["parallelism"] == "dp2"
    assert abs(row["value"] - 2 * 4096 * 5 / (row["ms_per_step"] * 1e-3)) < 1e-3 * row["value"]
    r = row["roofline"]
    assert r["bound"] == "mfma" and 0.05 < r["frac"] < 1.0 and r["frac"] == r["frac_in_run"] and "frac_from_profile" in r
    assert 1.0 < r["shader_clock_ghz"] < 2.6, r["shader_clock_ghz"]
    assert row["cpu_baseline"]["value"] > 0 and row["cpu_baseline"]["kind"] == "port" and "B=4096" in row["cpu_baseline"]["sample"]
    assert row["extra"]["loglik_eval_K50000"]["samples_per_s"] > 0
    assert row["extra"]["loglik_eval_K50000_samples_sharded"]["samples_per_s"] > 0


def test_bench_refuses_a_world_that_is_not_gpus():
    """a launcher that started another number of ranks than --gpus names: non-zero exit, no line"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                         text=True, timeout=60, env=env, cwd=root)
    assert out.returncode != 0 and not out.stdout.strip() and "--gpus 2" in out.stderr


def test_bench_step_through_rccl_with_one_rank():
    """the data-parallel step as it runs on several GPUs - gradient slabs, the decoder's all-reduce launched
    asynchronously from inside the backward pass, RCCL's averaging reduction - rehearsed with a world of one rank"""
    row = _bench(["--steps", "5", "--warmup", "2", "--batch", "4096", "--no-cpu", "--no-extra"],
                 {"DMV_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29539", "RANK": "0", "WORLD_SIZE": "1",
                  "LOCAL_RANK": "0"})
    assert row["n_gpus"] == 1 and row["value"] > 0
