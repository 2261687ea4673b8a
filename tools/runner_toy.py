#!/usr/bin/env python
"""Epoch rate of the YAML runner on the toy DeMissVAE configuration (19 366 x 5 training rows, batch 64, K = 5 imputations,
LAIR sampler, mvb2 objective with STL, networks (5,4,3,200) / (2,10,3,200)) on synthetic mixture data.
Usage: python tools/runner_toy.py [epochs] [hip_graph true|false]"""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIG = """
seed_everything: 20220118
experiment_subdir_base: logs/toy/mvb2vae_lair
data:
  setup_seed: 20220118
  dataset: toy_mog2_large
  batch_size: 64
  missingness: 'uniform'
  total_miss_train: 0.5
  total_miss_val: 0.5
  pre_imputation: 'oracle'
  pre_impute_val: false
model:
  class_path: vgiwae.models.MVBVAE
  init_args:
    mvb_objective: mvb2
    mvb2_optim_gradclip_val: 5
    var_latent_STL: True
    kl_analytic: False
    num_latent_samples: 1
    lr_generator: 1e-3
    lr_latent: 1e-3
    amsgrad_generator: True
    amsgrad_latent: True
    num_imputations_train: 5
    num_imputations_val: 5
    imputation_init_train: empirical
    imputation_init_val: empirical
    init_imputation_distribution: True
    mvb_delay_imputation: 0
    vae_sampler:
      class_path: vgiwae.shared.vae_imputation.VAESampler_LAIR
      init_args:
        num_iterations: 1
        num_imp_samples: 1
        num_prior_replenish_proposals: 0
        is_weights: dmis
        resampling_scheme: global_resampling
    generator_network:
      class_path: vgiwae.shared.neural_nets.ResidualFCNetwork
      init_args: {input_dim: 2, output_dim: 10, num_residual_blocks: 3, residual_block_dim: 200}
    generator_distribution: normal
    var_latent_network:
      class_path: vgiwae.shared.neural_nets.ResidualFCNetwork
      init_args: {input_dim: 5, output_dim: 4, num_residual_blocks: 3, residual_block_dim: 200}
    var_latent_distribution: normal
trainer:
  max_epochs: 3
"""


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    graph = sys.argv[2] if len(sys.argv) > 2 else "true"
    import scipy.io as sio
    from demiss_vae_amd import run
    with tempfile.TemporaryDirectory() as root:
        r = np.random.default_rng(0)
        os.makedirs(os.path.join(root, "data", "toy"))
        means = r.normal(size=(4, 5)) * 2

        def draw(n):
            return (means[r.integers(0, 4, n)] + 0.5 * r.normal(size=(n, 5))).astype(np.float32)
        sio.savemat(os.path.join(root, "data", "toy", "data_mog2_large.mat"), {"train": draw(19366), "val": draw(2000), "test": draw(2000)})
        cfg = os.path.join(root, "cfg.yaml")
        with open(cfg, "w") as f:
            f.write(CONFIG)
        history, _ = run.main(["--config", cfg, "--data.data_root", os.path.join(root, "data"), "--trainer.default_root_dir", root,
                               "--trainer.max_epochs", str(epochs), "--trainer.hip_graph", graph])
    per = [history[0]["time_s"]] + [b["time_s"] - a["time_s"] for a, b in zip(history, history[1:])]
    steps = history[0]["step"]
    print(json.dumps({"hip_graph": graph, "steps_per_epoch": steps, "epoch_s": per,
                      "ms_per_step_incl_validation": round(1e3 * per[-1] / steps, 3)}))


if __name__ == "__main__":
    main()
