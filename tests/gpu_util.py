"""Helpers shared by the GPU parity tests: build the HIP-side objects from an oracle-side case."""
import torch

import demiss_vae_amd as dmv


def net_dims(spec):
    return dmv.NetDims(spec.in_dim, spec.out_dim, spec.nb, spec.hidden, spec.act, spec.dropout)


def packed_net(spec, params, device="cuda", requires_grad=True):
    ws, bs = [], []
    for n in spec.layer_names():
        ws.append(params[n + ".weight"].to(device).clone().requires_grad_(requires_grad))
        bs.append(params[n + ".bias"].to(device).clone().requires_grad_(requires_grad))
    return dmv.PackedNet(net_dims(spec), ws, bs)


def model_dims(spec):
    return dmv.ModelDims(D=spec.D, Zd=spec.Zd, gen_dist=spec.gen_dist, q_dist=spec.q_dist, C=spec.C,
                         encoder_use_mis_mask=spec.encoder_use_mis_mask)


def grads_of(net, spec):
    out = {}
    for n, w, b in zip(spec.layer_names(), net.weights, net.biases):
        if w.grad is not None:
            out[n + ".weight"] = w.grad
        if b.grad is not None:
            out[n + ".bias"] = b.grad
    return out


def keep_to_dev(masks, device="cuda"):
    """list of nb bool masks [..., hidden] -> uint8 [nb, rows, hidden]"""
    if masks is None:
        return None
    return torch.stack([m.reshape(-1, m.shape[-1]) for m in masks]).to(torch.uint8).to(device).contiguous()
