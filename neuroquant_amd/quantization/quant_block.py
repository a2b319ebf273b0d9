"""Quantised NeRV block (reference quantization/quant_block.py): conv -> PixelShuffle -> act, fused into one
HIP launch when the block is the shipped configuration (PixelShuffle + exact GELU)."""
import torch.nn as nn

from .. import ops
from ..models._layers import NeRVBlock
from .quant_layer import QuantModule


class BaseQuantBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.use_weight_quant = False
        self.trained = False
        self.ignore_reconstruction = False

    def set_quant_state(self, weight_quant: bool = False):
        self.use_weight_quant = weight_quant
        for m in self.modules():
            if isinstance(m, QuantModule):
                m.set_quant_state(weight_quant)


class QuantNeRVBlock(BaseQuantBlock):
    def __init__(self, basic_block: NeRVBlock, hadamard: bool = True, weight_quant_params: dict = {}):
        super().__init__()
        self.conv = QuantModule(basic_block.conv[0], hadamard, weight_quant_params)
        self.pixelshuffle = basic_block.conv[1]
        self.act = basic_block.act  # NB: like the reference, the block's norm layer is not carried over
        if isinstance(self.pixelshuffle, nn.PixelShuffle):
            self._r = self.pixelshuffle.upscale_factor
        elif isinstance(self.pixelshuffle, nn.Identity):
            self._r = 1
        else:
            self._r = None
        self._fusable = (self._r is not None and isinstance(self.act, nn.GELU)
                         and getattr(self.act, 'approximate', 'none') == 'none')

    def forward(self, x):
        if self._fusable:
            return self.conv.forward_fused(x, ops.EPI_PS_GELU, self._r)
        return self.act(self.pixelshuffle(self.conv(x)))


specials = {
    NeRVBlock: QuantNeRVBlock,
}
