"""The N > 1 path on CPU: two processes over gloo.  The collectives (flat gradient all-reduce, max/sum
log-sum-exp combine, shard arithmetic) are exercised with the CPU oracle standing in for the per-rank compute."""
import math
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import core
from oracle.recipes import NetSpec, make_params, make_batch, make_noise


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spec():
    enc, dec = NetSpec(16, 32, 1, 32), NetSpec(16, 16, 1, 32)
    return core.ModelSpec(D=8, Zd=16, enc=enc, dec=dec, encoder_use_mis_mask=True)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from demiss_vae_amd import parallel as P
    torch.set_num_threads(1)
    spec = _spec()
    B, I = 32, 5
    X, M = make_batch(3, B, 8)
    eps = make_noise(4, 1, I, B, 16)
    lo, hi = P.shard_slice(B)
    ep = {k: v.clone().requires_grad_(True) for k, v in make_params(spec.enc, 1).items()}
    dp = {k: v.clone().requires_grad_(True) for k, v in make_params(spec.dec, 2).items()}
    bound, _ = core.estimate_iwelbo(spec, ep, dp, X[lo:hi], M[lo:hi], eps[:, :, lo:hi], STL=True)
    (-bound).backward()
    params = list(ep.values()) + list(dp.values())
    P.allreduce_grads(params)
    # K-sharded log-likelihood: each rank folds its half of the importance samples
    K = 64
    epsK = make_noise(5, 1, K, B, 16)
    klo, khi = P.shard_slice(K)
    with torch.no_grad():
        Mf = M.float()
        p = core.mlp_forward(make_params(spec.enc, 1), core.encoder_input(spec, X * Mf, M), spec.enc)
        mu, sg, _ = core.q_heads(spec, p)
        Z = mu + sg * epsK[:, klo:khi]
        g = core.mlp_forward(make_params(spec.dec, 2), Z, spec.dec)
        lw = core.smis_log_weights(spec, g, X, M, Z, mu, sg)[0]          # [K/2, B]
        m_r = lw.max(0)[0]
        s_r = torch.exp(lw - m_r).sum(0)
    m, s = P.lse_allreduce(m_r, s_r)
    ll = m + torch.log(s) - math.log(K)
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in params], "ll": ll, "shards": [P.shard_slice(10, r, 3) for r in range(3)]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_and_lse_combine(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    spec = _spec()
    B, I, K = 32, 5, 64
    X, M = make_batch(3, B, 8)
    ep = {k: v.clone().requires_grad_(True) for k, v in make_params(spec.enc, 1).items()}
    dp = {k: v.clone().requires_grad_(True) for k, v in make_params(spec.dec, 2).items()}
    bound, _ = core.estimate_iwelbo(spec, ep, dp, X, M, make_noise(4, 1, I, B, 16), STL=True)
    (-bound).backward()
    for a, b in zip(got["grads"], [p.grad for p in list(ep.values()) + list(dp.values())]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)       # mean of shard gradients = full-batch gradient
    _, per_b = core.loglik_eval(spec, make_params(spec.enc, 1), make_params(spec.dec, 2), X, M, make_noise(5, 1, K, B, 16))
    torch.testing.assert_close(got["ll"], per_b, rtol=1e-5, atol=1e-5)
    assert got["shards"] == [(0, 4), (4, 8), (8, 10)]
