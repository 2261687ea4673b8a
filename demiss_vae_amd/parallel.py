"""Multi-GPU pieces of the hot path: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).

Training is pure data parallelism: datapoints are independent and every loss is a mean over the batch,
so each rank runs the fused step on its own shard and the only exchange is one all-reduce per network of
its parameter gradients (UCI models: 1-2 MB each, latency-bound, so whole-network messages beat per-tensor
buckets), done before gradient clipping so the clip norm is that of the averaged gradient.  The gradients
live permanently in one buffer per network (no gather copy), and the decoder's exchange starts as soon as
the decoder's backward pass has written them, under the encoder's backward pass.

The K-sample log-likelihood shards either test points (no exchange until the final sum) or importance
samples of the same points; the latter combines per-rank (max, sum-exp) pairs with a max all-reduce
followed by a sum all-reduce of the rescaled sums (two messages of B floats).
"""
import math
from typing import Sequence

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def all_reduce(t: torch.Tensor, op=None, group=None):
    """dist.all_reduce; under the gloo backend (CPU tests, rehearsals with several ranks on one GPU) device tensors
    are staged through host memory"""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def _average(flat: torch.Tensor, group=None):
    if dist.get_backend(group) == "nccl":
        all_reduce(flat, dist.ReduceOp.AVG, group)          # RCCL averages inside the reduction
    else:
        all_reduce(flat, dist.ReduceOp.SUM, group)
        flat.div_(dist.get_world_size(group))


def overlap_grad_exchange(nets, group=None, force=False):
    """Start the exchange of a network's gradients the moment its backward pass has written them, while the rest of
    the backward pass (the encoder's, after the decoder's) is still running: `nets` get use_flat_grads() and a hook
    that launches an asynchronous all-reduce of their gradient slab; allreduce_grads() at the end of the step waits
    for it instead of reducing again.  One backward pass per network and step (the hook fires only where an objective
    marks the gradients complete); steps recorded into a HIP graph and gloo exchanges of device tensors (staged through
    the host) fall back to the blocking exchange."""
    for net in nets:
        net.use_flat_grads(True)
        net.on_grads_ready = None
    if not (dist.is_available() and dist.is_initialized()) or (world() == 1 and not force):
        return

    def launch(net):
        slab = net.grad_slab()
        if torch.cuda.is_available() and slab.is_cuda and torch.cuda.is_current_stream_capturing():
            return
        nccl = dist.get_backend(group) == "nccl"
        if slab.is_cuda and not nccl:
            return
        if net.pending_exchange is not None:
            raise RuntimeError("a second backward pass before the gradients of the first were exchanged: "
                               "overlap_grad_exchange needs one backward pass per step")
        work = dist.all_reduce(slab, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, group=group, async_op=True)
        net.pending_exchange = (work, nccl)

    for net in nets:
        net.on_grads_ready = launch


def allreduce_grads(params: Sequence[torch.Tensor], group=None, force=False):
    """Average .grad over ranks (equal shard sizes assumed).  Gradients that live in their network's permanent slab
    (PackedNet.use_flat_grads) are reduced where they are, one collective per network, or were already sent off by
    overlap_grad_exchange and are only waited for; any others go through one flat gather + all-reduce.  force: run the
    collectives even in a world of one rank (rehearsals of the RCCL path on a single GPU)."""
    if world() == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return
    from .mlp import slab_owner
    with_grad = [p for p in params if p.grad is not None]
    if not with_grad:
        return
    asked = {id(p) for p in with_grad}
    nets, loose = {}, []
    for p in with_grad:
        net = slab_owner(p.grad)
        if net is None:
            loose.append(p)
        else:
            nets.setdefault(id(net), (net, []))[1].append(p)
    for net, ps in nets.values():
        whole = net.grads_live_in_slab() and all(id(t) in asked for t in net.params())
        pending, net.pending_exchange = net.pending_exchange, None
        if pending is not None:
            work, averaged = pending
            work.wait()
            if not whole:
                raise RuntimeError("gradients left their slab after the overlapped exchange had started")
            if not averaged:
                net.grad_slab().div_(dist.get_world_size(group))
        elif whole:
            _average(net.grad_slab(), group)
        else:
            loose += ps
    if not loose:
        return
    grads = [p.grad for p in loose]
    flat = _flatten_dense_tensors(grads)
    _average(flat, group)
    # the averaged gradients stay where they are: every .grad becomes a view of the flat buffer (one gather kernel and
    # the collective per step, no copies back)
    for p, g in zip(loose, _unflatten_dense_tensors(flat, grads)):
        p.grad = g


def lse_allreduce(run_max: torch.Tensor, run_sum: torch.Tensor, group=None):
    """Combine per-rank streaming-logsumexp states (max m_r, s_r = sum exp(x - m_r)) into the global one:
    m = max_r m_r (all-reduce MAX), s = sum_r s_r exp(m_r - m) (all-reduce SUM).  Returns (m, s)."""
    if world() == 1:
        return run_max, run_sum
    m = run_max.clone()
    all_reduce(m, dist.ReduceOp.MAX, group)
    s = run_sum * torch.exp(run_max - m)
    s = torch.where(torch.isfinite(run_max), s, torch.zeros_like(s))
    all_reduce(s, dist.ReduceOp.SUM, group)
    return m, s


def sharded_loglik(K: int, fold, shard=None, group=None):
    """Per-datapoint log (1/K) sum_k exp(log_w_k) with the K importance samples split over the ranks
    (vgiwae/test/refit_encoder_using_iwae_and_estimate_loglik.py:189-222 computes it on one device).

    fold(lo, hi) -> (run_max, run_sum): the streaming-logsumexp state of the samples lo <= k < hi, this rank's
    contiguous share of the K (an empty share gives (-inf, 0)).  The states are combined with lse_allreduce; every
    rank returns the same [B] tensor.  shard = (rank, world) overrides the process group's."""
    r, w = (rank(), world()) if shard is None else shard
    lo, hi = shard_slice(K, r, w)
    m, s = fold(lo, hi)
    if w > 1:
        m, s = lse_allreduce(m, s, group)
    return m + torch.log(s) - math.log(K)


def allreduce_weighted_means(sums: dict, weights: dict, device=None, group=None):
    """Means of quantities accumulated as (sum of weight * value, sum of weights) per key on every rank: one SUM
    all-reduce of all of them (ranks that saw no batch contribute zeros; the key sets are united first)."""
    if world() == 1:
        return {k: float(sums[k]) / float(weights[k]) for k in sorted(sums)}
    mine = sorted(sums)
    every = [None] * world()
    dist.all_gather_object(every, mine, group=group)
    keys = sorted(set().union(*every))
    if device is None and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())      # RCCL reduces device tensors only
    vec = torch.tensor([float(sums.get(k, 0.0)) for k in keys] + [float(weights.get(k, 0.0)) for k in keys],
                       dtype=torch.float64, device=device)
    all_reduce(vec, dist.ReduceOp.SUM, group)
    vec = vec.cpu()
    n = len(keys)
    return {k: float(vec[i]) / float(vec[n + i]) for i, k in enumerate(keys)}


def shard_slice(n: int, r: int = None, w: int = None):
    """Contiguous, near-equal shard [lo, hi) of n items for rank r of w."""
    r = rank() if r is None else r
    w = world() if w is None else w
    per = math.ceil(n / w)
    return min(n, r * per), min(n, (r + 1) * per)
