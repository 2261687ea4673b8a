"""GPU-box run of the batch contract (masks / filter / index / pre-imputation against the reference's outputs, the device
loader's hand-over) and of the corner the contract leaves to the kernels: a non-finite value at a MISSING entry."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def test_masks_filter_index_and_data_module_on_the_gpu_box():
    import test_host_logic as H
    H.test_masks_filter_index_bit_exact()
    H.check_data_module_against_fixture()


def test_device_loader_hands_over_the_batch_contract():
    """(X float32, M bool, index int64) on the device, every kept row exactly once per epoch, last short batch kept
    (the reference's DataLoader default drop_last=False, vgiwae/data/missing_data_module.py:292-312)."""
    from vgiwae.data import MissingDataModule
    from vgiwae.data.missing_data_module import DATASET, MISSINGNESS, PRE_IMPUTATION
    dm = MissingDataModule(dataset=DATASET.toy_mog2_large, batch_size=512, missingness=MISSINGNESS.uniform,
                           total_miss_train=0.5, total_miss_val=0.3, pre_imputation=PRE_IMPUTATION.oracle,
                           pre_impute_val=False, data_root=gu.GOLDEN, setup_seed=20220118)
    dm.setup("fit")
    g = gu.load("datamodule")
    n = int(g["uniform_oracle.train.n"])
    seen = []
    host_X, host_M, host_idx = dm.train_data[:]
    for X, M, idx in dm.train_dataloader():
        assert X.is_cuda and X.dtype == torch.float32 and M.dtype == torch.bool and idx.dtype == torch.int64
        assert M.any(dim=1).all()
        rows = idx.cpu().numpy()
        pos = np.searchsorted(host_idx, rows)
        assert np.array_equal(X.cpu().numpy(), np.asarray(host_X, dtype=np.float32)[pos])
        assert np.array_equal(M.cpu().numpy(), host_M[pos])
        seen.append(rows)
    seen = np.concatenate(seen)
    assert len(seen) == n and np.array_equal(np.sort(seen), g["uniform_oracle.train.index"])


def _log_normal(x, mu, s):
    return -0.5 * ((x - mu) / s) ** 2 - torch.log(s) - 0.5 * math.log(2 * math.pi)


@pytest.mark.parametrize("shape", ["normal_rows4", "normal_generic", "bern_vec4"])
def test_non_finite_value_at_a_missing_entry_makes_the_bound_nan(shape):
    """The reference multiplies the per-dimension log-probabilities by the float mask (vgiwae/shared/iwae.py:20), so a
    NaN (or an infinite log-probability) under a zero of the mask still poisons that datapoint's bound: 0 * NaN.  The
    kernels form the same product; the other datapoints of the batch are untouched."""
    from demiss_vae_amd import _lib as L
    lib, st = L.lib(), L.stream_ptr()
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    if shape == "bern_vec4":
        B, I, D, Zd, gen = 24, 3, 784, 12, "bern"
    elif shape == "normal_rows4":
        B, I, D, Zd, gen = 96, 5, 8, 16, "normal_with_eps"
    else:
        B, I, D, Zd, gen = 40, 5, 7, 10, "normal"
    normal = gen.startswith("normal")
    P = 2 * D if normal else D
    ld = 32 * ((P + 31) // 32) if normal else P
    N = B * I
    cfg = L.DmvObjCfg()
    cfg.B, cfg.G, cfg.nz, cfg.I, cfg.D, cfg.Zd = B, 1, 1, I, D, Zd
    cfg.gen_dist, cfg.q_eps, cfg.x_per_component = L.GEN[gen], 1, 0
    ldE = 32 * ((2 * Zd + 31) // 32)
    cfg.enc_sb, cfg.enc_sg, cfg.raw_off, cfg.logit_off = ldE, 0, Zd, 0
    enc = torch.randn(B, ldE, device="cuda", generator=g)
    dec = torch.randn(N, ld, device="cuda", generator=g)
    Z = torch.randn(N, Zd, device="cuda", generator=g)
    if normal:
        X = torch.randn(B, D, device="cuda", generator=g)
    else:
        X = (torch.rand(B, D, device="cuda", generator=g) < 0.3).float()
    M = torch.rand(B, D, device="cuda", generator=g) < 0.6
    M[:, 0] = True
    M[3, 5], M[7, 2], M[9, 1] = False, False, True
    X[3, 5] = float("nan")                 # NaN under a missing entry
    if normal:
        X[7, 2] = float("inf")             # log-probability -inf under a missing entry: 0 * -inf
    else:
        X[7, 2] = float("nan")
    Mu8 = M.to(torch.uint8).contiguous()
    per_b = torch.empty(B, device="cuda")
    wgt = torch.empty(N, device="cuda")
    f = L.DmvObjFwd()
    f.q_kind, f.mix_C, f.kind = L.Q_OWN, 1, L.BOUND_LSE
    f.bound, f.wgt = L.addr(per_b), L.addr(wgt)
    L.check(lib.dmv_obj_forward(C.byref(cfg), C.byref(f), L.ptr(dec), ld, L.ptr(X), L.ptr(Mu8), L.ptr(Z), Zd, L.ptr(enc), st))
    torch.cuda.synchronize()

    # the same bound in plain torch (float64), rows ordered n = i * B + b
    Xd, Md = X.double(), M.double()
    decd = dec.double().reshape(I, B, ld)
    if normal:
        s = torch.nn.functional.softplus(decd[..., D:2 * D]) + (1e-5 if gen == "normal_with_eps" else 0.0)
        lp = _log_normal(Xd, decd[..., :D], s)
    else:
        lp = Xd * torch.nn.functional.logsigmoid(decd) + (1 - Xd) * torch.nn.functional.logsigmoid(-decd)
    gen_lp = (lp * Md).sum(-1)
    Zd_ = Z.double().reshape(I, B, Zd)
    mu = enc.double()[:, :Zd]
    sig = torch.nn.functional.softplus(enc.double()[:, Zd:2 * Zd]) + 1e-5
    prior = _log_normal(Zd_, torch.zeros((), device="cuda", dtype=torch.float64), torch.ones((), device="cuda", dtype=torch.float64)).sum(-1)
    lq = _log_normal(Zd_, mu, sig).sum(-1)
    ref = torch.logsumexp(gen_lp + prior - lq, dim=0) - math.log(I)
    bad = torch.isnan(ref)
    assert bad.sum().item() == 2 and bad[3] and bad[7] and not bad[9]
    assert torch.equal(torch.isnan(per_b), bad)
    torch.testing.assert_close(per_b[~bad].double(), ref[~bad], rtol=2e-5, atol=2e-5)
