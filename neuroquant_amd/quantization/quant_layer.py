"""QuantModule: a Conv2d with fake-quantised weight/bias (reference quantization/quant_layer.py), running on the
HIP implicit-GEMM convolution instead of F.conv2d."""
import math
from typing import Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .quantizer import StraightThrough, UniformAffineQuantizer


def _next_power_of_two(n: int):
    return 1 if n == 0 else 2 ** math.ceil(math.log2(n))


def hadamard_along_channel_weight(x: torch.Tensor, normalize: bool = True):
    """Orthonormal Walsh-Hadamard transform along C_in of a (C_out, C_in, KH, KW) weight, C_in a power of two
    (reference quant_layer.py:16-22); its own inverse."""
    return ops.hadamard_along_channel_weight(x)


class QuantModule(nn.Module):
    """Convert a Conv2d to its fake-quantised counterpart.  Attribute names follow the reference
    (quant_layer.py:24-89): weight (shares the conv's Parameter), org_weight, hadamard_weight, C, bias, org_bias,
    use_weight_quant, weight_quantizer, bias_quantizer, fwd_kwargs, fwd_func."""

    def __init__(self, org_module: Union[nn.Conv2d,], hadamard: bool = True, weight_quant_params: dict = {}):
        super().__init__()
        if not isinstance(org_module, nn.Conv2d):
            raise ValueError('Not supported modules: {}'.format(org_module))
        self.fwd_kwargs = dict(stride=org_module.stride, padding=org_module.padding,
                               dilation=org_module.dilation, groups=org_module.groups)
        self.fwd_func = F.conv2d  # kept for interface parity; the HIP convolution below is what runs
        k = org_module.kernel_size[0]
        self._hip_ok = (org_module.kernel_size[0] == org_module.kernel_size[1] and k in (1, 3, 5)
                        and tuple(org_module.stride) == (1, 1) and tuple(org_module.padding) == (k // 2, k // 2)
                        and tuple(org_module.dilation) == (1, 1) and org_module.groups == 1)

        self.weight = org_module.weight
        self.org_weight = org_module.weight.data.clone()
        self.hadamard = hadamard
        if self.hadamard:
            self.C = self.weight.shape[1]
            # zero-pad C_in to a power of two, then transform (quant_layer.py:45-49); needs the GPU
            self.hadamard_weight = ops.hadamard_weight_of(org_module.weight.data)
        if org_module.bias is not None:
            self.bias = org_module.bias
            self.org_bias = org_module.bias.data.clone()
        else:
            self.bias = None
            self.org_bias = None

        self.use_weight_quant = False
        self.weight_quantizer = UniformAffineQuantizer(**weight_quant_params)
        self.bias_quantizer = UniformAffineQuantizer(**weight_quant_params)
        self._wb_override = None  # (W_hat, b_hat) injected by the calibration engine for one iteration
        self.extra_repr = org_module.extra_repr

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        for name in ('org_weight', 'hadamard_weight', 'org_bias'):
            t = getattr(self, name, None)
            if isinstance(t, torch.Tensor):
                setattr(self, name, fn(t))
        return self

    # ---- weights -------------------------------------------------------------------------------
    def quantized_params(self):
        """(W_hat, b_hat) through the current quantisers (quant_layer.py:69-74)."""
        if self.hadamard:
            weight = ops.hadamard_along_channel_weight(self.weight_quantizer(self.hadamard_weight), n_out=self.C)
        else:
            weight = self.weight_quantizer(self.weight)
        bias = self.bias_quantizer(self.bias) if self.bias is not None else None
        return weight, bias

    def _current_params(self):
        if self._wb_override is not None:
            return self._wb_override
        if self.use_weight_quant:
            return self.quantized_params()
        return self.org_weight, self.org_bias

    # ---- forward -------------------------------------------------------------------------------
    def forward_fused(self, input: torch.Tensor, epilogue: int, r: int = 1):
        if not self._hip_ok:
            raise NotImplementedError('only stride-1 "same" convolutions with k in {1,3,5} are built (SURVEY §8a-8)')
        weight, bias = self._current_params()
        return ops.conv2d_fused(input, weight, bias, epilogue, r)

    def forward(self, input: torch.Tensor):
        return self.forward_fused(input, ops.EPI_PLAIN)

    def forward_out_img(self, input: torch.Tensor):
        """head conv fused with OutImg 'tanh' (models/_layers.py:10-16)."""
        return self.forward_fused(input, ops.EPI_TANH)

    def set_quant_state(self, weight_quant: bool = False):
        self.use_weight_quant = weight_quant

    def get_weight_perturbation(self):
        return self.org_weight - self.weight_quantizer(self.weight)
