"""Frechet distance between model samples and data in the feature space of a trained encoder (reference:
vgiwae/shared/fid.py:9-64; the `*_FID` model classes evaluate it every few thousand steps).  Features come from the
encoder's forward pass (HIP kernels for residual MLPs); the 2 Zd x 2 Zd matrix square root runs in scipy on the host,
as in the reference."""
import numpy as np
import scipy.linalg
import torch


@torch.no_grad()
def compute_inception_gaussian(samples, *, inception_model, batch_size=5000):
    """(mean [F], covariance [F, F]) of the features of `samples` [N, D]"""
    feats = torch.cat([inception_model(samples[b:b + batch_size]) for b in range(0, samples.shape[0], batch_size)], dim=0)
    return feats.mean(dim=0), torch.cov(feats.T)


def gaussian_fid_score(ref_mean, ref_cov, samples_mean, samples_cov):
    """squared 2-Wasserstein distance between two Gaussians: |m1 - m2|^2 + tr(C1 + C2 - 2 (C2 C1)^(1/2))"""
    ref_mean, ref_cov, samples_mean, samples_cov = (t.detach().cpu() for t in (ref_mean, ref_cov, samples_mean, samples_cov))
    root = scipy.linalg.sqrtm((samples_cov @ ref_cov).numpy())
    if np.iscomplexobj(root):
        root = root.real
    root = torch.as_tensor(root, dtype=ref_cov.dtype)
    return torch.norm(ref_mean - samples_mean, p=2) ** 2 + torch.trace(ref_cov + samples_cov - 2 * root)


def compute_fid_score(refs, samples, *, inception_model):
    return gaussian_fid_score(*compute_inception_gaussian(refs, inception_model=inception_model),
                              *compute_inception_gaussian(samples, inception_model=inception_model))
