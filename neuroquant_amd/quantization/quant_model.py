"""QuantModel wrapper (reference quantization/quant_model.py): swaps decoder convs / NeRV blocks for their
quantised counterparts (anything under a child named '*encoder*' is left alone) and keeps the state toggles."""
from typing import Union

import torch.nn as nn

from ..models import HNeRV, NeRV
from .quant_block import BaseQuantBlock, specials
from .quant_layer import QuantModule
from .quantizer import StraightThrough


class QuantModel(nn.Module):
    def __init__(self, model: Union[NeRV, HNeRV], hadamard: bool = True, weight_quant_params: dict = {}):
        super().__init__()
        self.model = model
        self.hadamard = hadamard
        self.quant_module_refactor(self.model, weight_quant_params)

    def quant_module_refactor(self, module: nn.Module, weight_quant_params: dict = {}):
        for name, child in module.named_children():
            if 'encoder' in name:
                continue
            if type(child) in specials:
                setattr(module, name, specials[type(child)](child, self.hadamard, weight_quant_params))
            elif isinstance(child, nn.Conv2d):
                setattr(module, name, QuantModule(child, self.hadamard, weight_quant_params))
            elif isinstance(child, StraightThrough):
                continue
            else:
                self.quant_module_refactor(child, weight_quant_params)

    def quant_modules(self):
        return [m for m in self.model.modules() if isinstance(m, QuantModule)]

    def set_quant_state(self, weight_quant: bool = False):
        for m in self.model.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.set_quant_state(weight_quant)

    def encode(self, input):
        return self.model.encode(input)

    def decode(self, input):
        return self.model.decode(input)

    def forward(self, input):
        return self.model.decode(input)

    def set_bitwidth(self, bit, init=False):
        """per-layer bit-widths in modules() order; returns the parameter-weighted average (quant_model.py:58-72)."""
        bits, num_param = 0., 0.
        for count, m in enumerate(self.quant_modules()):
            for q in (m.weight_quantizer, m.bias_quantizer):
                q.bitwidth_refactor(bit[count])
                q.inited = init
            bits += m.weight_quantizer.n_bits * m.weight.numel() + m.bias_quantizer.n_bits * m.bias.numel()
            num_param += m.weight.numel() + m.bias.numel()
        return bits / num_param

    def get_quantized_param(self):
        out = []
        for m in self.quant_modules():
            out += [m.weight_quantizer.x_quant, m.bias_quantizer.x_quant]
        return out

    def get_perturbation(self):
        return [m.get_weight_perturbation() for m in self.quant_modules()]
