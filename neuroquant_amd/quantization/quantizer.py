"""Quantisers with the reference's interface (quantization/quantizer.py) on HIP kernels.

Only what the network-wise calibration path reaches is implemented: UniformAffineQuantizer with
scale_method 'max' and AdaRoundQuantizer with round_mode 'learned_hard_sigmoid' (SURVEY.md §2 row 5).
"""
import logging
import time

import torch
import torch.nn as nn

from .. import ops


class StraightThrough(nn.Module):
    def __init__(self, channel_num: int = 1):
        super().__init__()

    def forward(self, input):
        return input


def lp_loss(pred, tgt, p=2.0, reduction='none'):
    """reference quantizer.py:66-73.  p=2 / reduction='none' (the calibration loss) runs in one HIP kernel."""
    if reduction == 'none' and p == 2.0 and pred.is_cuda:
        return ops.l2_loss(pred, tgt)
    if reduction == 'none':
        return (pred - tgt).abs().pow(p).sum(1).mean()
    return (pred - tgt).abs().pow(p).mean()


class UniformAffineQuantizer(nn.Module):
    """Asymmetric uniform quantiser, lazily initialised at the first forward (reference quantizer.py:76-243).

    Attributes kept from the reference: n_bits, n_levels, delta (nn.Parameter once inited), zero_point, inited,
    channel_wise, scale_method, sym, prob, is_training.
    """

    def __init__(self, n_bits: int = 8, symmetric: bool = False, channel_wise: bool = False,
                 scale_method: str = 'max', prob: float = 1.0):
        super().__init__()
        self.sym = symmetric
        assert 2 <= n_bits <= 8, 'bitwidth not supported'
        self.n_bits = n_bits
        self.n_levels = 2 ** self.n_bits
        self.delta = None
        self.zero_point = None
        self.inited = False
        self.channel_wise = channel_wise
        self.scale_method = scale_method
        self.prob = prob
        self.is_training = False

    def forward(self, x: torch.Tensor):
        if self.inited is False:
            delta, self.zero_point = self.init_quantization_scale(x, self.channel_wise)
            self.delta = nn.Parameter(delta)
            self.inited = True
        x_ans = ops.uaq_fake_quant(x, self.delta, self.zero_point, self.n_levels)
        if self.is_training and self.prob < 1.0:
            x_ans = torch.where(torch.rand_like(x) < self.prob, x_ans, x)
        return x_ans

    def init_quantization_scale(self, x: torch.Tensor, channel_wise: bool = False):
        if 'max' not in self.scale_method:
            raise NotImplementedError(f"scale_method {self.scale_method!r}: only 'max' is on the calibration path")
        if self.sym:
            raise NotImplementedError('symmetric quantisation is not on the calibration path')
        return ops.scale_init_max(x, self.n_levels, channel_wise)

    def bitwidth_refactor(self, refactored_bit: int):
        assert 2 <= refactored_bit <= 8, 'bitwidth not supported'
        self.n_bits = refactored_bit
        self.n_levels = 2 ** self.n_bits

    def extra_repr(self):
        return (f'bit={self.n_bits}, scale_method={self.scale_method}, symmetric={self.sym}, '
                f'channel_wise={self.channel_wise},')


class AdaRoundQuantizer(nn.Module):
    """Adaptive-rounding quantiser (reference quantizer.py:247-323), round_mode 'learned_hard_sigmoid'."""

    def __init__(self, uaq: UniformAffineQuantizer, weight_tensor: torch.Tensor, round_mode='learned_round_sigmoid'):
        super().__init__()
        self.n_bits = uaq.n_bits
        self.sym = uaq.sym
        self.n_levels = uaq.n_levels
        self.round_mode = round_mode
        self.alpha = None
        self.soft_targets = False
        self.x_quant = None
        self.gamma, self.zeta = -0.1, 1.1
        self.beta = 2 / 3
        if round_mode != 'learned_hard_sigmoid':
            raise NotImplementedError
        logging.info('Init alpha to be FP32')
        t0 = time.time()
        # delta / zero-point go through fp16 (quantizer.py:264-265); alpha = logit of the rounding residue
        delta, self.zero_point, alpha = ops.adaround_init(weight_tensor, uaq.delta, uaq.zero_point)
        self.alpha = nn.Parameter(alpha)
        self.delta = nn.Parameter(delta)
        logging.info('init time: {}'.format(time.time() - t0))

    def forward(self, x):
        if self.round_mode != 'learned_hard_sigmoid':
            raise ValueError('Wrong rounding mode')
        y, self.x_quant = ops.adaround_fake_quant(x, self.alpha, self.delta, self.zero_point, self.n_levels,
                                                  bool(self.soft_targets))
        return y

    def get_soft_targets(self):
        return torch.clamp(torch.sigmoid(self.alpha) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def extra_repr(self):
        return 'bit={}'.format(self.n_bits)
