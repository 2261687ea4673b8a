#!/usr/bin/env python
"""End-to-end epoch rate of the YAML runner on a synthetic table of UCI-gas size (852k x 8 training rows, batch 512,
the configs' IWAE I=5 STL model): data loader, step, clipping, optimiser, scheduler, logging, validation pass.
Usage: python tools/runner_epoch.py [epochs] [demiss]   (demiss: the UCI DeMissVAE config on a 102k-row table)"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIG = """
seed_everything: 20220118
experiment_subdir_base: logs/uci_gas/iwae_i5_encm_stl
data:
  setup_seed: 20220118
  dataset: uci_gas
  batch_size: 512
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
      init_args: {input_dim: 16, output_dim: 16, num_residual_blocks: 2, residual_block_dim: 256}
    generator_distribution: normal_with_eps
    var_latent_network:
      class_path: vgiwae.shared.neural_nets.ResidualFCNetwork
      init_args: {input_dim: 16, output_dim: 32, num_residual_blocks: 2, residual_block_dim: 256}
    var_latent_distribution: normal_with_eps
    encoder_use_mis_mask: True
trainer:
  gradient_clip_val: 5
  gradient_clip_algorithm: norm
"""

# the UCI DeMissVAE configs (configs/uci_gas/compare/mis50/mvb2vae_lairdmisr1iter10_k5_stl_gradclip1.yaml): two-loss manual
# step, K = 5 imputations per datapoint updated by ten LAIR iterations with one prior-replenish proposal and a final
# resampling over all iterations' proposals
DEMISS = CONFIG.replace("logs/uci_gas/iwae_i5_encm_stl", "logs/uci_gas/mvb2vae_lair").replace("vgiwae.models.IWAE", "vgiwae.models.MVBVAE") \
    .replace("    num_importance_samples: 5\n", """    kl_analytic: False
    mvb_objective: mvb2
    mvb_eval_val_loss: True
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
        num_iterations: 10
        num_imp_samples: 1
        num_prior_replenish_proposals: 1
        resample_final_imps: True
        is_weights: dmis
        resampling_scheme: global_resampling
""").replace("{input_dim: 16, output_dim: 32,", "{input_dim: 8, output_dim: 32,").replace("encoder_use_mis_mask: True", "encoder_use_mis_mask: False")


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    demiss = len(sys.argv) > 2 and sys.argv[2] == "demiss"
    from demiss_vae_amd import run
    with tempfile.TemporaryDirectory() as root:
        r = np.random.default_rng(0)
        os.makedirs(os.path.join(root, "data", "gas"))
        A = r.normal(size=(3, 8)).astype(np.float32)
        for split, n in (("train", 102400 if demiss else 852174), ("val", 10240 if demiss else 94685), ("test", 1024 if demiss else 105206)):
            z = r.normal(size=(n, 3)).astype(np.float32)
            x = np.tanh(z @ A) + 0.1 * r.normal(size=(n, 8)).astype(np.float32)
            np.savez(os.path.join(root, "data", "gas", f"{split}.npz"), x.astype(np.float32))
        cfg = os.path.join(root, "cfg.yaml")
        with open(cfg, "w") as f:
            f.write(DEMISS if demiss else CONFIG)
        t0 = time.time()
        if os.environ.get("RUNNER_PROFILE"):
            import cProfile
            import pstats
            prof = cProfile.Profile()
            prof.enable()
        history, _ = run.main(["--config", cfg, "--data.data_root", os.path.join(root, "data"),
                               "--trainer.default_root_dir", root, "--trainer.max_epochs", str(epochs)])
        total = time.time() - t0
        if os.environ.get("RUNNER_PROFILE"):
            prof.disable()
            pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
    per = [history[0]["time_s"]] + [b["time_s"] - a["time_s"] for a, b in zip(history, history[1:])]
    steps = history[0]["step"]
    print(json.dumps({"epochs": epochs, "steps_per_epoch": steps, "epoch_s": per, "wall_s": round(total, 2),
                      "train_rows_per_s_last_epoch": round(852174 * 0.998 / per[-1]),
                      "ms_per_step_incl_validation": round(1e3 * per[-1] / steps, 3)}))


if __name__ == "__main__":
    main()
