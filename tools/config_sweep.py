#!/usr/bin/env python
"""A few training steps and a validation pass of one config per family, on synthetic data files of the right shapes.
python tools/config_sweep.py <directory of the reference's configs> [name filter ...]
A family = one combination of model class, network class, distributions, gradient estimator, objective, sampler
(class and main switches), mask-concatenation, dataset and missingness; refit configs (they need a trained
checkpoint) and the control-study evaluation configs without a model are left out.  Needs the GPU."""
import glob, json, os, sys, tempfile, time, traceback
import numpy as np
import scipy.io as sio
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demiss_vae_amd import run

root = tempfile.mkdtemp(prefix="sweep_")
r = np.random.default_rng(1)
os.makedirs(f"{root}/data/toy")
C, D = 3, 5
A = r.normal(size=(C, D, D)); covs = A @ A.transpose(0, 2, 1) / D + 0.2 * np.eye(D)
means = r.normal(size=(C, D)) * 2; probs = np.array([[0.5, 0.3, 0.2]])
def draw(n):
    c = r.choice(C, size=n, p=probs[0])
    return (means[c] + np.einsum("nij,nj->ni", np.linalg.cholesky(covs)[c], r.normal(size=(n, D)))).astype(np.float32)
sio.savemat(f"{root}/data/toy/data_mog2_large.mat", {"train": draw(1500), "val": draw(200), "test": draw(200),
                                                    "comp_probs": probs, "means": means, "covs": covs})
for name, d in (("gas", 8), ("power", 6), ("hepmass", 21), ("miniboone", 43)):
    os.makedirs(f"{root}/data/{name}")
    for split, n in (("train", 3000), ("val", 300), ("test", 300)):
        np.savez(f"{root}/data/{name}/{split}.npz", r.normal(size=(n, d)).astype(np.float32))
os.makedirs(f"{root}/data/MNIST"); os.makedirs(f"{root}/data/omniglot")
for split, n in (("train", 700), ("val", 100), ("test", 100)):
    np.savez(f"{root}/data/MNIST/{split}.npz", data=(r.random((n, 28, 28)) > 0.87).astype(np.uint8) * 255, targets=r.integers(0, 10, n))
    np.savez(f"{root}/data/omniglot/{split}.npz", data=(r.random((n, 105, 105)) > 0.87).astype(np.float32), targets=r.integers(0, 10, n))

def families(cfg_dir):
    import yaml
    short = lambda cp: cp.split(".")[-1] if isinstance(cp, str) else cp
    seen = {}
    for f in sorted(glob.glob(os.path.join(cfg_dir, "**", "*.yaml"), recursive=True)):
        try:
            c = yaml.safe_load(open(f))
        except Exception:
            continue
        if not isinstance(c, dict) or not isinstance(c.get("model"), dict) or "class_path" not in c["model"]:
            continue
        a = c["model"].get("init_args", {})
        s = a.get("vae_sampler") or {}
        sa = s.get("init_args", {}) if isinstance(s, dict) else {}
        d = c.get("data", {})
        sig = (short(c["model"]["class_path"]), short(a.get("generator_network", {}).get("class_path")),
               a.get("generator_distribution"), str(a.get("var_latent_distribution")), bool(a.get("var_latent_STL")),
               bool(a.get("var_latent_DREG")), a.get("kl_analytic"), a.get("mvb_objective") or a.get("objective"),
               short(s.get("class_path")) if isinstance(s, dict) else None, sa.get("num_prior_replenish_proposals"),
               sa.get("resample_final_imps"), sa.get("clip_imputations"), a.get("encoder_use_mis_mask"), d.get("dataset"),
               d.get("missingness"), str(sa.get("interpolate_target"))[:20], str(sa.get("start_distribution"))[:20])
        seen.setdefault(sig, f)
    return list(seen.values())


results = {}
cfg_dir, only = sys.argv[1], sys.argv[2:]
for path in families(cfg_dir):
    name = os.path.relpath(path, cfg_dir)
    if only and not any(o in name for o in only):
        continue
    lines = []
    t0 = time.time()
    try:
        run.main(["--config", path, "--data.data_root", f"{root}/data", "--trainer.default_root_dir", f"{root}/{name[:-5].replace(os.sep, '__')}",
                  "--trainer.max_epochs", "1", "--trainer.max_steps", "6"], log=lines.append)
        torch.cuda.synchronize()
        results[name] = "ok"
    except Exception as e:
        results[name] = f"{type(e).__name__}: {str(e)[:300]} @ {traceback.format_exc().strip().splitlines()[-3][:160]}"
    print(f"{name}: {results[name]}  ({time.time() - t0:.1f}s)", flush=True)
bad = {k: v for k, v in results.items() if v != "ok"}
print(json.dumps({"n": len(results), "failed": len(bad)}))
