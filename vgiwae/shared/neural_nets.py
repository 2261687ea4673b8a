"""Residual fully-connected network with the reference's constructor and state_dict layout
(vgiwae/shared/neural_nets.py:49-133), evaluated by the fused fp32-MFMA kernels of libdmv.

Parameter names (`initial_layer`, `blocks.N.linear_layers.{0,1}`, `final_layer`) and shapes are the
reference's, so its checkpoints load with load_state_dict.  forward() has no CPU implementation: the
HIP library is the only compute path."""
from typing import Callable, List, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

import demiss_vae_amd as dmv
from demiss_vae_amd.imputation import _keep

ACTIVATIONS = {"relu": F.relu, "leaky_relu": F.leaky_relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}
_ACT_NAMES = {F.relu: "relu", F.leaky_relu: "leaky_relu", torch.tanh: "tanh", torch.sigmoid: "sigmoid",
              F.sigmoid: "sigmoid"}


class ResidualBlock(nn.Module):
    """Parameter container of one block: two Linear(features, features) layers and the dropout rate."""

    def __init__(self, features: int, *, activation: Callable = F.relu, dropout_probability: float = 0.0):
        super().__init__()
        self.activation = activation
        self.linear_layers = nn.ModuleList([nn.Linear(features, features), nn.Linear(features, features)])
        self.dropout = nn.Dropout(p=dropout_probability)

    def reset_params(self):
        for layer in self.linear_layers:
            layer.reset_parameters()


class ResidualFCNetwork(nn.Module):
    """Linear -> act -> [a + Linear(drop(act(Linear(a)))) -> act] x num_residual_blocks -> Linear."""

    def __init__(self, input_dim: int, output_dim: int, num_residual_blocks: int, residual_block_dim: int, *,
                 activation: Union[Callable, str] = F.relu, dropout_probability: float = 0.0):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.num_residual_blocks, self.residual_block_dim = num_residual_blocks, residual_block_dim
        if isinstance(activation, str):
            self.activation_name = activation
            self.activation = ACTIVATIONS[activation]
        elif activation in _ACT_NAMES:
            self.activation_name = _ACT_NAMES[activation]
            self.activation = activation
        else:
            raise NotImplementedError("only relu / leaky_relu / tanh / sigmoid have HIP kernels")
        self.dropout_probability = dropout_probability
        self.initial_layer = nn.Linear(input_dim, residual_block_dim)
        self.blocks = nn.ModuleList([ResidualBlock(residual_block_dim, activation=self.activation,
                                                   dropout_probability=dropout_probability)
                                     for _ in range(num_residual_blocks)])
        self.final_layer = nn.Linear(residual_block_dim, output_dim)
        self._packed = None

    # ---- bridge to the kernels
    def net_dims(self) -> dmv.NetDims:
        return dmv.NetDims(self.input_dim, self.output_dim, self.num_residual_blocks, self.residual_block_dim,
                           self.activation_name, float(self.dropout_probability))

    def _linears(self) -> List[nn.Linear]:
        out = [self.initial_layer]
        for b in self.blocks:
            out += [b.linear_layers[0], b.linear_layers[1]]
        return out + [self.final_layer]

    def packed_net(self) -> dmv.PackedNet:
        """Fragment-ordered image of the current parameters (rebuilt if the parameters were replaced, e.g.
        by .to(device); refreshed in place when an optimiser step changed them)."""
        lin = self.__dict__.get("_layer_cache")          # the layer objects never change; their parameters may
        if lin is None:
            lin = self.__dict__["_layer_cache"] = self._linears()
        ws, bs = [l._parameters["weight"] for l in lin], [l._parameters["bias"] for l in lin]
        p = self._packed
        if p is None or any(a is not b for a, b in zip(p.weights + p.biases, ws + bs)):
            p = self._packed = dmv.PackedNet(self.net_dims(), ws, bs)
        return p

    def keep_masks(self, rows: int):
        """Device dropout masks for one call in training mode (None when dropout is off)."""
        return _keep(self.packed_net(), rows, self.training, self.initial_layer.weight.device)

    def forward(self, inputs: torch.Tensor):
        if not inputs.is_cuda:
            raise RuntimeError("ResidualFCNetwork runs on the MI355X kernels only; move the model and inputs to cuda")
        net = self.packed_net()
        rows = inputs.numel() // self.input_dim
        return dmv.ResidualMLPFunction.apply(net, torch.is_grad_enabled(), self.keep_masks(rows), inputs, *net.params())


class FullyConnectedNetwork(nn.Module):
    """Plain stack of linear layers with an activation between them (vgiwae/shared/neural_nets.py:15-46); same
    constructor and state_dict keys (`linear_layers.N.*`).  No shipped config uses it; it stays a torch module
    (rocBLAS GEMMs) and the kernels take over at its outputs, like the convolutional trunks."""

    def __init__(self, layer_dims: List[int], *, activation: Callable = F.relu):
        super().__init__()
        self.layer_dims, self.activation = layer_dims, activation
        self.linear_layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(layer_dims[:-1], layer_dims[1:]))

    @property
    def latent_dim(self):          # what the model classes ask a decoder for
        return self.layer_dims[0]

    @property
    def output_dim(self):
        return self.layer_dims[-1]

    def forward(self, inputs: torch.Tensor):
        out = inputs
        for layer in self.linear_layers[:-1]:
            out = self.activation(layer(out))
        return self.linear_layers[-1](out)
