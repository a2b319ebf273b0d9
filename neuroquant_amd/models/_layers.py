"""Building blocks of the NeRV / HNeRV video INRs (reference models/_layers.py), parameter names kept so the
reference's checkpoints load unchanged (e.g. `decoder.3.conv.0.weight`, `encoder.stages.1.0.pwconv1.weight`).

The decoder block is the hot path: once wrapped by QuantModel it runs on the fused HIP conv+PixelShuffle+GELU
kernel.  The ConvNeXt encoder is a one-time precompute and stays plain PyTorch.
"""
from math import ceil, pi

import torch
import torch.nn as nn
import torch.nn.functional as F


def OutImg(x, out_bias='tanh'):
    if out_bias == 'sigmoid':
        return torch.sigmoid(x)
    if out_bias == 'tanh':
        return torch.tanh(x) * 0.5 + 0.5
    return x + float(out_bias)


class Sin(nn.Module):
    def forward(self, input):
        return torch.sin(input)


_ACTS = {
    'relu': lambda: nn.ReLU(True),
    'leaky': lambda: nn.LeakyReLU(inplace=True),
    'leaky01': lambda: nn.LeakyReLU(negative_slope=0.1, inplace=True),
    'relu6': lambda: nn.ReLU6(inplace=True),
    'gelu': lambda: nn.GELU(),
    'sin': lambda: Sin(),
    'swish': lambda: nn.SiLU(inplace=True),
    'softplus': lambda: nn.Softplus(),
    'hardswish': lambda: nn.Hardswish(inplace=True),
}


def ActivationLayer(act_type):
    if act_type not in _ACTS:
        raise KeyError(f"Unknown activation function {act_type}.")
    return _ACTS[act_type]()


def NormLayer(norm_type, ch_width):
    if norm_type == 'none':
        return nn.Identity()
    if norm_type == 'batch':
        return nn.BatchNorm2d(num_features=ch_width, track_running_stats=False)
    if norm_type == 'instance':
        return nn.InstanceNorm2d(num_features=ch_width)
    raise NotImplementedError


class NeRVBlock(nn.Module):
    """conv (stride 1, 'same') -> PixelShuffle(stride) -> norm -> act   (reference _layers.py:20-36)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride, bias, norm, act):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(in_channel, out_channel * stride * stride, kernel_size, stride=1,
                      padding=ceil((kernel_size - 1) // 2), bias=bias),
            nn.PixelShuffle(stride) if stride != 1 else nn.Identity(),
        )
        self.norm = NormLayer(norm, out_channel)
        self.act = ActivationLayer(act)

    def forward(self, x):
        return self.act(self.norm(self.conv(x)))


class PositionEncoding(nn.Module):
    """NeRV frame-index embedding: [sin(b^i * pi * t), cos(b^i * pi * t)] (reference _layers.py:77-86)."""

    def __init__(self, base, level):
        super().__init__()
        self.pe_bases = base ** torch.arange(int(level)) * pi

    def forward(self, pos):
        value_list = pos * self.pe_bases.to(pos.device)
        pe_embed = torch.cat([torch.sin(value_list), torch.cos(value_list)], dim=-1)
        return pe_embed.view(pos.size(0), -1, 1, 1)


class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.data_format = data_format
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_last":
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class Block(nn.Module):
    """ConvNeXt block: 7x7 depthwise conv -> LN -> 1x1 (4x) -> GELU -> 1x1 -> layer scale -> residual."""

    def __init__(self, dim, drop_path=0., layer_scale_init_value=1e-6):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)),
                                  requires_grad=True) if layer_scale_init_value > 0 else None
        if drop_path > 0.:
            raise NotImplementedError("stochastic depth is not used by the shipped configs (drop_path_rate=0)")
        self.drop_path = nn.Identity()

    def forward(self, x):
        y = self.dwconv(x).permute(0, 2, 3, 1)
        y = self.pwconv2(self.act(self.pwconv1(self.norm(y))))
        if self.gamma is not None:
            y = self.gamma * y
        return x + y.permute(0, 3, 1, 2)


class ConvNeXt(nn.Module):
    """HNeRV frame encoder (reference _layers.py:134-193): per stage a strided patchify conv + `stage_blocks` Blocks."""

    def __init__(self, stage_blocks=0, strds=[2, 2, 2, 2], dims=[96, 192, 384, 768], in_chans=3, drop_path_rate=0.,
                 layer_scale_init_value=1e-6):
        super().__init__()
        self.downsample_layers = nn.ModuleList()
        self.stages = nn.ModuleList()
        self.stage_num = len(dims)
        for i in range(self.stage_num):
            if i > 0:
                down = nn.Sequential(LayerNorm(dims[i - 1], eps=1e-6, data_format="channels_first"),
                                     nn.Conv2d(dims[i - 1], dims[i], kernel_size=strds[i], stride=strds[i]))
            else:
                down = nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=strds[i], stride=strds[i]),
                                     LayerNorm(dims[0], eps=1e-6, data_format="channels_first"))
            self.downsample_layers.append(down)
            self.stages.append(nn.Sequential(*[Block(dim=dims[i], layer_scale_init_value=layer_scale_init_value)
                                               for _ in range(stage_blocks)]))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=.02)
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        for down, stage in zip(self.downsample_layers, self.stages):
            x = stage(down(x))
        return x
