"""Convolutional trunks of the image configs (reference: vgiwae/shared/vae_resnet.py:45-304): a ResNet encoder and
its mirror-image decoder built from resize-convolutions.  These stay stock torch modules (MIOpen convolutions);
the kernels of this package take over at their outputs (demiss_vae_amd/external.py).  Constructor arguments,
attribute names and therefore state_dict keys follow the reference so its checkpoints load."""
from math import prod
from typing import List, Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F


class Interpolate(nn.Module):
    """F.interpolate (nearest) as a module, to a fixed size or by a scale factor."""

    def __init__(self, size=None, scale_factor=None):
        super().__init__()
        self.size, self.scale_factor = size, scale_factor

    def forward(self, x):
        return F.interpolate(x, size=self.size, scale_factor=self.scale_factor)


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


def _resize_conv(cin, cout, k, scale=1):
    """nearest-neighbour upsampling followed by a convolution (no checkerboard artefacts)"""
    return _conv(cin, cout, k) if scale == 1 else nn.Sequential(Interpolate(scale_factor=scale), _conv(cin, cout, k))


class _ResidualBlock(nn.Module):
    """out = relu(dropout(bn2(conv2(relu(bn1(conv1(x)))))) + skip(x)); `skip_name` is the attribute (and
    state_dict prefix) the projection of the skip path lives under."""
    expansion = 1
    skip_name = "skip"

    def _init(self, conv1, width1, conv2, planes, skip, dropout_prob):
        self.conv1, self.bn1 = conv1, nn.BatchNorm2d(width1)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = conv2, nn.BatchNorm2d(planes)
        setattr(self, self.skip_name, skip)
        self.dropout = nn.Dropout2d(p=dropout_prob)

    def forward(self, x):
        skip = getattr(self, self.skip_name)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.dropout(self.bn2(self.conv2(out)))
        out = out + (x if skip is None else skip(x))
        return self.relu(out)


class EncoderBlock(_ResidualBlock):
    """basic ResNet block: the stride and the channel change sit in the first convolution"""
    skip_name = "downsample"

    def __init__(self, inplanes, planes, stride=1, downsample=None, dropout_prob=0.):
        super().__init__()
        self._init(_conv(inplanes, planes, 3, stride), planes, _conv(planes, planes, 3), planes, downsample, dropout_prob)


class DecoderBlock(_ResidualBlock):
    """the block in reverse: resize-convolutions, upscaling and channel change in the second one"""
    skip_name = "upsample"

    def __init__(self, inplanes, planes, scale=1, upsample=None, dropout_prob=0.):
        super().__init__()
        self._init(_resize_conv(inplanes, inplanes, 3), inplanes, _resize_conv(inplanes, planes, 3, scale), planes, upsample,
                   dropout_prob)


def _flat_leading(x_shape, item_numel):
    """leading (batch) dims of a tensor whose trailing dims hold one item of item_numel elements"""
    n = 1
    for i in range(len(x_shape) - 1, -1, -1):
        n *= x_shape[i]
        if n == item_numel:
            return tuple(x_shape[:i])
    raise ValueError(f"shape {tuple(x_shape)} does not end in an item of {item_numel} elements")


class ResNetEncoder(nn.Module):
    def __init__(self, input_shape: Tuple[int, int, int], layers: List[int], layer_widths: List[int], latent_dim: int,
                 first_conv: bool = False, maxpool1: bool = False, dropout_prob: float = 0.,
                 expect_miss_mask_as_extra_chennel: bool = False):
        super().__init__()
        assert len(layers) == len(layer_widths) and len(layers) > 0
        self.input_shape, self.latent_dim = input_shape, latent_dim
        self.first_conv, self.maxpool1 = first_conv, maxpool1
        self.inplanes = layer_widths[0]
        cin = input_shape[0] + (1 if expect_miss_mask_as_extra_chennel else 0)
        self.conv1 = self._stem(cin, self.inplanes)
        self.bn1 = nn.BatchNorm2d(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1) if maxpool1 else nn.MaxPool2d(kernel_size=1, stride=1)
        self.layers = nn.ModuleList(
            self._make_layer(EncoderBlock, w, n, stride=1 if i == 0 else 2, dropout_prob=dropout_prob)
            for i, (n, w) in enumerate(zip(layers, layer_widths)))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.final = nn.Linear(in_features=layer_widths[-1], out_features=latent_dim)

    def _stem(self, cin, cout):
        if self.first_conv:
            return nn.Conv2d(cin, cout, kernel_size=7, stride=2, padding=3, bias=False)
        return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=False)

    def add_mask_inputs_to_existing_network(self):
        """one more input channel for the missingness mask, keeping the trained filters of the data channels"""
        old = self.conv1
        new = self._stem(old.in_channels + 1, old.out_channels)
        new.weight.data[:, :old.in_channels] = old.weight.data
        self.conv1 = new

    def _make_layer(self, block, planes, blocks, stride=1, dropout_prob=0.):
        proj = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            proj = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, stride),
                                 nn.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, proj, dropout_prob=dropout_prob)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, dropout_prob=dropout_prob) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x, m=None):
        lead = _flat_leading(x.shape, prod(self.input_shape))
        x = x.reshape(-1, *self.input_shape)
        if m is not None:
            x = torch.cat((x, m.reshape(-1, 1, *self.input_shape[1:])), dim=-3)   # mask as the last channel
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        for layer in self.layers:
            x = layer(x)
        x = torch.flatten(self.avgpool(x), 1)
        return self.final(x.reshape(*lead, -1))


class ResNetDecoder(nn.Module):
    def __init__(self, output_shape: Tuple[int, int, int], layers: List[int], layer_upscale: List[Optional[int]],
                 layer_widths: List[int], downscaled_resolution: Tuple[int, int], latent_dim: int, upscale_factor: float,
                 first_conv: bool = False, maxpool1: bool = False, dropout_prob: float = 0.):
        super().__init__()
        assert len(layers) == len(layer_widths) and len(layers) > 0
        self.output_shape, self.layer_widths = output_shape, layer_widths
        self.expansion = DecoderBlock.expansion
        self.first_conv, self.maxpool1 = first_conv, maxpool1
        self.input_height = output_shape[-2]
        self.downscaled_resolution, self.latent_dim = downscaled_resolution, latent_dim
        self.upscale_factor = upscale_factor
        self.inplanes = layer_widths[0] * 2 * self.expansion
        self.linear = nn.Linear(latent_dim, self.inplanes * prod(downscaled_resolution))

        self.layers = nn.ModuleList()
        last = len(layers) - 1
        for i, (n, w) in enumerate(zip(layers, layer_widths)):
            doubles = i < last or self.maxpool1          # every stage but the last doubles the resolution
            if i == last and self.maxpool1:
                self.upscale_factor *= 2
            self.layers.append(self._make_layer(DecoderBlock, w, n, scale=2 if doubles else 1, dropout_prob=dropout_prob))
        if self.first_conv:
            self.upscale = Interpolate(scale_factor=2)
            self.upscale_factor *= 2
        else:
            self.upscale = Interpolate(scale_factor=1)
        self.upscales = nn.ModuleList([Interpolate(size=up) if up is not None else None for up in layer_upscale])
        self.upscale1 = Interpolate(size=int(self.input_height // self.upscale_factor))   # right after the linear layer
        self.conv1 = nn.Conv2d(layer_widths[-1] * self.expansion, output_shape[0], kernel_size=3, stride=1, padding=1,
                               bias=False)

    @property
    def input_dim(self):
        return self.linear.in_features

    def _make_layer(self, block, planes, blocks, scale=1, dropout_prob=0.):
        proj = None
        if scale != 1 or self.inplanes != planes * block.expansion:
            proj = nn.Sequential(_resize_conv(self.inplanes, planes * block.expansion, 1, scale),
                                 nn.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, scale, proj, dropout_prob=dropout_prob)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, dropout_prob=dropout_prob) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        lead = x.shape[:-1]
        x = self.linear(x.reshape(-1, x.shape[-1]))
        x = self.upscale1(x.reshape(x.size(0), self.layer_widths[0] * 2 * self.expansion, *self.downscaled_resolution))
        for layer, up in zip(self.layers, self.upscales):
            x = layer(x)
            if up is not None:
                x = up(x)
        x = self.conv1(self.upscale(x))
        return x.reshape(*lead, -1)
