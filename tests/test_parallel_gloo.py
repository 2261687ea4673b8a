"""The N > 1 path on CPU: two processes over gloo.  The collectives (flat gradient all-reduce, max/sum
log-sum-exp combine, shard arithmetic) and the sharded test pass of the runner are the product's own code; the CPU
oracle stands in for the per-rank kernels only."""
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
    # K-sharded log-likelihood through the product's own orchestration (parallel.sharded_loglik: this rank's share of
    # the samples, logsumexp all-reduce, - log K); the CPU oracle stands in for the kernels inside `fold` only
    K = 64
    epsK = make_noise(5, 1, K, B, 16)

    def fold_for(Xb, Mb, eps_b):
        def fold(klo, khi):
            with torch.no_grad():
                p = core.mlp_forward(make_params(spec.enc, 1), core.encoder_input(spec, Xb * Mb.float(), Mb), spec.enc)
                mu, sg, _ = core.q_heads(spec, p)
                Z = mu + sg * eps_b[:, klo:khi]
                g = core.mlp_forward(make_params(spec.dec, 2), Z, spec.dec)
                lw = core.smis_log_weights(spec, g, Xb, Mb, Z, mu, sg)[0]          # [share of K, B]
                m_r = lw.max(0)[0]
                return m_r, torch.exp(lw - m_r).sum(0)
        return fold
    ll = P.sharded_loglik(K, fold_for(X, M, epsK))

    # the runner's test pass (demiss_vae_amd.run.test) in both layouts, over three ragged batches
    from demiss_vae_amd import run

    class _Data:
        def setup(self, stage):
            assert stage == "test"

        def test_dataloader(self):
            return [(X[:12], M[:12], 0), (X[12:24], M[12:24], 12), (X[24:], M[24:], 24)]

    class _Model:
        can_shard_samples = True

        def eval(self):
            pass

        def test_step(self, batch, i):
            Xb, Mb, b0 = batch
            per_b = P.sharded_loglik(K, fold_for(Xb, Mb, epsK[:, :, b0:b0 + Xb.shape[0]]), shard=getattr(self, "loglik_shard", (0, 1)))
            self._logged = {"est_loglik/test": per_b.mean(), "points": float(Xb.shape[0])}

    runs = {mode: run.test(_Model(), _Data(), log=lambda *_: None, shard=mode) for mode in ("points", "samples")}

    # gradients that live in their network's permanent slab: the overlapped exchange (launched by the hook the backward
    # pass calls, here by hand), the blocking one of a second network, and loose gradients beside them in one call
    import demiss_vae_amd as dmv
    slab_report = {}
    nets = []
    for tag, seed in (("overlapped", 6), ("blocking", 7)):
        ns = NetSpec(6, 5, 1, 32)
        pr = make_params(ns, seed)
        ws = [pr[n + ".weight"].clone().requires_grad_(True) for n in ns.layer_names()]
        bs = [pr[n + ".bias"].clone().requires_grad_(True) for n in ns.layer_names()]
        nets.append(dmv.PackedNet(dmv.NetDims(6, 5, 1, 32), ws, bs))
    P.overlap_grad_exchange(nets[:1])
    nets[1].use_flat_grads(True)
    for net in nets:
        for i, (t, v) in enumerate(zip(net.params(), net.grad_views())):
            v.copy_(torch.full_like(t, float(i + 1)) * (rank + 1) + t.detach())
            t.grad = v
        assert net.grads_live_in_slab() and net.grad_views() is None      # a second pass would not alias the slab
    nets[0].on_grads_ready(nets[0])
    assert nets[0].pending_exchange is not None and nets[1].pending_exchange is None
    # a second backward pass while the exchange is in flight is refused (mlp_backward asks before it writes gradients)
    try:
        nets[0].refuse_pass_during_exchange()
        raise AssertionError("second pass accepted")
    except RuntimeError as e:
        assert "one backward" in str(e)
    nets[1].refuse_pass_during_exchange()
    loose = torch.nn.Parameter(torch.zeros(7))
    loose.grad = torch.full((7,), float(rank))
    P.allreduce_grads(nets[0].params() + nets[1].params() + [loose])
    for tag, net in zip(("overlapped", "blocking"), nets):
        assert net.pending_exchange is None and net.grads_live_in_slab()
        slab_report[tag] = [(t.grad - t.detach()).clone() for t in net.params()]
    slab_report["loose"] = loose.grad.clone()
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in params], "ll": ll, "shards": [P.shard_slice(10, r, 3) for r in range(3)],
                    "runs": runs, "slabs": slab_report}, out)
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
    # run.test: the mean over test points whichever way the work was split (batch-size-weighted over ragged batches)
    for mode in ("points", "samples"):
        assert abs(got["runs"][mode]["est_loglik/test"] - per_b.mean().item()) < 1e-5 * abs(per_b.mean().item()), mode
        assert abs(got["runs"][mode]["points"] - (12 * 12 + 12 * 12 + 8 * 8) / 32) < 1e-9
    # slab-resident gradients: rank r held (i + 1) * (r + 1) in tensor i -> the mean over two ranks is 1.5 * (i + 1)
    for tag in ("overlapped", "blocking"):
        for i, g in enumerate(got["slabs"][tag]):
            torch.testing.assert_close(g, torch.full_like(g, 1.5 * (i + 1)), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got["slabs"]["loose"], torch.full((7,), 0.5))
