"""ORACLE -- test infrastructure, NOT product code.

CPU (PyTorch-CPU, fp32) restatement of the reference's network-wise calibration hot path
(SURVEY.md §8a).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this file; the product (`neuroquant_amd/`) never does and fails loudly when
its HIP library is missing.

Pinning: every function below is checked in `tests/test_oracle_golden.py` against vectors the
real reference produced in the build container (`tests/golden/*.npz`, generator
`tests/golden/make_golden.py`).  One boundary is "parity unpinned": the Walsh-Hadamard
butterflies live in the un-vendored, un-versioned pip package `hadamard_transform`
(reference quantization/quant_layer.py:7,19); `fwht` here restates the published orthonormal
Sylvester transform and the Hadamard goldens were captured with an equivalent stand-in, so they
pin everything around the transform but not its internal summation order.

Written functionally (plain tensors + small records) on purpose: it is a restatement of the
algorithm, not a copy of the reference's module tree.  file:line citations are relative to
/root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

GAMMA, ZETA = -0.1, 1.1  # quantizer.py:274 (rectified-sigmoid stretch)


# ----------------------------------------------------------------------------------------------
# Uniform affine quantiser (quantizer.py:76-168), 'max' scale method only (north-star configs)
# ----------------------------------------------------------------------------------------------
def _scale_init_flat(x: torch.Tensor, n_levels: int):
    """quantizer.py:156-168 -- min/max with 0 included, division in Python double, result fp32."""
    x_min = min(x.min().item(), 0.0)
    x_max = max(x.max().item(), 0.0)
    delta = torch.tensor((x_max - x_min) / (n_levels - 1), dtype=torch.float32)
    delta = torch.maximum(delta, torch.tensor(1e-8, dtype=torch.float32))
    zp = (torch.tensor(-x_min, dtype=torch.float32) / delta).round()
    return delta, zp


def scale_init_max(x: torch.Tensor, n_levels: int, channel_wise: bool):
    """delta / zero-point initialisation (quantizer.py:127-168).

    channel_wise + 4-D weight -> one pair per output channel, shape (C_out,1,1,1);
    channel_wise + 1-D bias   -> ONE scalar pair, shape (1,)  (quantizer.py:144-152);
    otherwise                 -> 0-d scalars.
    """
    x = x.detach()
    if channel_wise:
        if x.dim() == 4:
            pairs = [_scale_init_flat(x[c], n_levels) for c in range(x.shape[0])]
            d = torch.stack([p[0] for p in pairs]).view(-1, 1, 1, 1)
            z = torch.stack([p[1] for p in pairs]).view(-1, 1, 1, 1)
            return d, z
        if x.dim() == 1:
            d, z = _scale_init_flat(x, n_levels)
            return d.view(-1), z.view(-1)
        raise ValueError("channel-wise init expects a 4-D weight or 1-D bias")
    return _scale_init_flat(x, n_levels)


def _round_ste(u: torch.Tensor) -> torch.Tensor:
    return (u.round() - u).detach() + u  # quantizer.py:53-57 (half-to-even, straight-through)


def uaq_fake_quant(x, delta, zp, n_levels: int):
    """quantizer.py:117-119.  Differentiable w.r.t. delta (and x) through autograd."""
    x_int = _round_ste(x / delta) + zp
    x_q = torch.clamp(x_int, 0, n_levels - 1)
    return (x_q - zp) * delta


def uaq_ddelta(x, gy, delta, zp, n_levels: int):
    """Closed form of d(sum(gy*y))/d(delta), reduced like delta's shape."""
    u = x / delta
    x_int = u.round() + zp
    inside = ((x_int >= 0) & (x_int <= n_levels - 1)).to(x.dtype)
    x_q = torch.clamp(x_int, 0, n_levels - 1)
    t = gy * ((x_q - zp) - inside * u)
    if delta.dim() == 4:
        return t.sum(dim=(1, 2, 3), keepdim=True)
    return t.sum().view(delta.shape)


# ----------------------------------------------------------------------------------------------
# AdaRound quantiser, 'learned_hard_sigmoid' (quantizer.py:247-319)
# ----------------------------------------------------------------------------------------------
def adaround_init(x, delta_uaq, zp_uaq):
    """quantizer.py:264-265 (fp16 round trip of delta / zp) and 305-314 (alpha so that h(alpha)=frac)."""
    delta = delta_uaq.detach().half().float()
    zp = zp_uaq.detach().half().float()
    u = x.detach() / delta
    rest = u - torch.floor(u)
    alpha = -torch.log((ZETA - GAMMA) / (rest - GAMMA) - 1)
    return delta, zp, alpha


def soft_targets(alpha):
    return torch.clamp(torch.sigmoid(alpha) * (ZETA - GAMMA) + GAMMA, 0, 1)  # quantizer.py:302-303


def adaround_fake_quant(x, alpha, delta, zp, n_levels: int, soft: bool):
    """quantizer.py:288-300.  Returns (dequantised value, integer grid value x_quant)."""
    x_floor = torch.floor(x / delta)
    x_int = x_floor + (soft_targets(alpha) if soft else (alpha >= 0).float())
    x_q = torch.clamp(x_int + zp, 0, n_levels - 1)
    return (x_q - zp) * delta, x_q


def adaround_dalpha(x, gy, alpha, delta, zp, n_levels: int):
    """Closed form of d(sum(gy*y))/d(alpha) for the soft forward."""
    s = torch.sigmoid(alpha)
    lin = s * (ZETA - GAMMA) + GAMMA
    h = torch.clamp(lin, 0, 1)
    x_int = torch.floor(x / delta) + h + zp
    inside = ((x_int >= 0) & (x_int <= n_levels - 1)).to(x.dtype)
    hprime = (ZETA - GAMMA) * s * (1 - s) * ((lin >= 0) & (lin <= 1)).to(x.dtype)
    return gy * delta * inside * hprime


def round_regulariser(alphas: Sequence[torch.Tensor], b: float, weight: float):
    """calib_model.py:39-47 -- sum over WEIGHT quantisers only of weight * sum(1 - |2h-1|^b)."""
    total = 0
    for a in alphas:
        rv = soft_targets(a)
        total = total + weight * (1 - ((rv - 0.5).abs() * 2).pow(b)).sum()
    return total


def temp_decay(t: int, t_max: int, rel_start_decay: float, start_b: float, end_b: float) -> float:
    """data_utils.py:24-41 (linear, despite the docstring there)."""
    start = rel_start_decay * t_max
    if t < start:
        return start_b
    rel_t = (t - start) / (t_max - start)
    return end_b + (start_b - end_b) * max(0.0, 1 - rel_t)


def lp_loss(pred, tgt, p: float = 2.0):
    return (pred - tgt).abs().pow(p).sum(1).mean()  # quantizer.py:66-71, reduction='none' branch


def psnr_per_frame(out, gt):
    mse = F.mse_loss(out, gt, reduction="none").flatten(1).mean(1)  # utils.py:148-151
    return -10 * torch.log10(mse + 1e-9)


# ----------------------------------------------------------------------------------------------
# Walsh-Hadamard transform along C_in (quant_layer.py:13-22, 44-49, 70-71)  [parity unpinned]
# ----------------------------------------------------------------------------------------------
def next_pow2(n: int) -> int:
    return 1 if n == 0 else 2 ** math.ceil(math.log2(n))


def fwht(x: torch.Tensor) -> torch.Tensor:
    """Orthonormal Sylvester-ordered WHT along the last dim (length must be 2^k)."""
    n = x.shape[-1]
    assert n > 0 and n & (n - 1) == 0
    lead = x.shape[:-1]
    y, h = x, 1
    while h < n:
        y = y.reshape(*lead, n // (2 * h), 2, h)
        y = torch.stack((y[..., 0, :] + y[..., 1, :], y[..., 0, :] - y[..., 1, :]), dim=-2).reshape(*lead, n)
        h *= 2
    return y / math.sqrt(n)


def hadamard_along_cin(w: torch.Tensor) -> torch.Tensor:
    """(C_out, C_pad, KH, KW) -> same shape, WHT along dim 1 (quant_layer.py:16-22)."""
    return fwht(w.permute(0, 2, 3, 1)).permute(0, 3, 1, 2).contiguous()


def hadamard_weight_of(w: torch.Tensor) -> torch.Tensor:
    """zero-pad C_in to 2^k FIRST, then transform (quant_layer.py:45-49)."""
    pad = next_pow2(w.shape[1]) - w.shape[1]
    return hadamard_along_cin(F.pad(w, (0, 0, 0, 0, 0, pad)))


# ----------------------------------------------------------------------------------------------
# Decoder (HNeRV.py:29-71, NeRV.py:24-65, _layers.py:10-36)
# ----------------------------------------------------------------------------------------------
@dataclass
class ConvLayer:
    w: torch.Tensor            # (C_out, C_in, k, k) full-precision weight
    b: torch.Tensor            # (C_out,)
    shuffle: int = 1           # PixelShuffle factor applied after the conv (1 = none)
    gelu: bool = False         # exact-erf GELU after the shuffle
    # quantiser state (filled by QuantStack)
    n_bits: int = 8
    wd: Optional[torch.Tensor] = None
    wz: Optional[torch.Tensor] = None
    bd: Optional[torch.Tensor] = None
    bz: Optional[torch.Tensor] = None
    wa: Optional[torch.Tensor] = None   # alpha (weight), shape of (Hadamard-domain) weight
    ba: Optional[torch.Tensor] = None   # alpha (bias)
    hw: Optional[torch.Tensor] = None   # Hadamard-domain padded weight
    w_soft: bool = True
    b_soft: bool = True

    @property
    def pad(self):
        return self.w.shape[-1] // 2


@dataclass
class Decoder:
    arch: str                  # 'hnerv' | 'nerv'
    layers: List[ConvLayer]
    fc_hw: tuple = (1, 1)      # channel->space factors applied to layer-0 output (NeRV.py:49-51)
    out_bias: str = "tanh"
    # Convolution used by forward(); tests swap it to probe the sensitivity of a calibration trajectory to the fp32
    # summation order of the convolutions (float64 accumulation, flipped spatial order): tests/golden/make_sensitivity.py
    conv_fn: object = F.conv2d

    @staticmethod
    def from_state_dict(sd: dict, arch: str, dec_strides: Sequence[int], fc_hw=(1, 1), out_bias="tanh") -> "Decoder":
        """Keys follow the reference checkpoints: decoder.0.{weight,bias}, decoder.N.conv.0.{weight,bias},
        head_layer.{weight,bias} (HNeRV.py:31-42)."""
        g = lambda k: torch.as_tensor(sd[k]).float().clone()
        layers = [ConvLayer(g("decoder.0.weight"), g("decoder.0.bias"))]
        for i, s in enumerate(dec_strides, start=1):
            layers.append(ConvLayer(g(f"decoder.{i}.conv.0.weight"), g(f"decoder.{i}.conv.0.bias"), shuffle=s, gelu=True))
        layers.append(ConvLayer(g("head_layer.weight"), g("head_layer.bias")))
        return Decoder(arch, layers, tuple(fc_hw), out_bias)

    def forward(self, emb: torch.Tensor, weights=None):
        """weights: optional list of (W, b) overriding the FP parameters (fake-quantised ones)."""
        x = emb
        for i, L in enumerate(self.layers):
            W, b = (L.w, L.b) if weights is None else weights[i]
            x = self.conv_fn(x, W, b, stride=1, padding=L.pad)
            if i == 0:
                n, c, h, w = x.shape
                fh, fw = self.fc_hw
                x = x.view(n, -1, fh, fw, h, w).permute(0, 1, 4, 2, 5, 3).reshape(n, -1, fh * h, fw * w)
            if L.shuffle != 1:
                x = F.pixel_shuffle(x, L.shuffle)
            if L.gelu:
                x = F.gelu(x)
        if self.out_bias == "tanh":
            return torch.tanh(x) * 0.5 + 0.5      # _layers.py:10-16
        if self.out_bias == "sigmoid":
            return torch.sigmoid(x)
        return x + float(self.out_bias)


# ----------------------------------------------------------------------------------------------
# Quantised stack + the two-phase calibration loop (quant_model.py, calib_model.py:92-240)
# ----------------------------------------------------------------------------------------------
class QuantStack:
    """Holds the quantiser state of all layers of a Decoder and produces fake-quantised weights."""

    def __init__(self, dec: Decoder, bits: Sequence[int], hadamard: bool, channel_wise: bool = True):
        assert len(bits) == len(dec.layers)
        self.dec, self.hadamard, self.channel_wise = dec, hadamard, channel_wise
        self.phase = "uaq"          # 'uaq' (phase 1 / before) or 'ada' (phase 2 / after)
        for L, nb in zip(dec.layers, bits):
            assert 2 <= nb <= 8, "bitwidth not supported"
            L.n_bits = nb
            L.hw = hadamard_weight_of(L.w) if hadamard else None
            src = L.hw if hadamard else L.w
            L.wd, L.wz = scale_init_max(src, 2 ** nb, channel_wise)   # lazily at first forward in the
            L.bd, L.bz = scale_init_max(L.b, 2 ** nb, channel_wise)   # reference (quantizer.py:112-115)
            L.wa = L.ba = None

    def avg_bits(self) -> float:
        """quant_model.py:58-72."""
        bits = sum(L.n_bits * L.w.numel() + L.n_bits * L.b.numel() for L in self.dec.layers)
        return bits / sum(L.w.numel() + L.b.numel() for L in self.dec.layers)

    def to_adaround(self):
        """calib_model.py:170-191."""
        for L in self.dec.layers:
            src = L.hw if self.hadamard else L.w
            L.wd, L.wz, L.wa = adaround_init(src, L.wd, L.wz)
            L.bd, L.bz, L.ba = adaround_init(L.b, L.bd, L.bz)
            L.w_soft = L.b_soft = True
        self.phase = "ada"

    def fake_quant_weights(self):
        out = []
        for L in self.dec.layers:
            nl = 2 ** L.n_bits
            src = L.hw if self.hadamard else L.w
            if self.phase == "uaq":
                Wq = uaq_fake_quant(src, L.wd, L.wz, nl)
                bq = uaq_fake_quant(L.b, L.bd, L.bz, nl)
            else:
                Wq, _ = adaround_fake_quant(src, L.wa, L.wd, L.wz, nl, L.w_soft)
                bq, _ = adaround_fake_quant(L.b, L.ba, L.bd, L.bz, nl, L.b_soft)
            if self.hadamard:
                Wq = hadamard_along_cin(Wq)[:, : L.w.shape[1]]       # quant_layer.py:70-71
            out.append((Wq, bq))
        return out

    def forward(self, emb):
        return self.dec.forward(emb, self.fake_quant_weights())


def calibrate(qs: QuantStack, cali_data: torch.Tensor, frames: torch.Tensor, order, iters: int,
              weight: float = 0.01, b_range=(20, 2), warmup: float = 0.2, p: float = 2.0, lr: float = 0.003,
              max_steps: Optional[int] = None, on_step=None, probe=None):
    """model_reconstruction (calib_model.py:92-240) with `gt` replaced by frames + a recorded batch order.

    order: int array (n_epochs, batches_per_epoch, B) -- frame indices per iteration.
    Returns the per-iteration log [(total, round, b, count)], counts restarting per phase like the reference.
    max_steps truncates the run (for timing a bounded sample) without changing the schedule; on_step(done) is called
    before every iteration (bench.py stamps the clock there); probe(phase, qs, fq_weights) is called after backward and
    before the optimiser step with the fake-quantised (W, b) pairs whose .grad hold dL/dW^, dL/db^ (gradient parity tests).
    """
    n_batches = order.shape[1]
    log, done = [], 0

    def run(params, opt_lr, epochs, ep0, round_on, max_count):
        nonlocal done
        for q in params:
            q.requires_grad_(True)
        opt = torch.optim.Adam(params, lr=opt_lr)
        loss_start = max_count * warmup
        count = 0
        for ep in range(ep0, ep0 + epochs):
            for it in range(n_batches):
                if on_step is not None:
                    on_step(done)
                if max_steps is not None and done >= max_steps:
                    return
                idx = torch.as_tensor(order[ep][it], dtype=torch.int64)
                if probe is not None:
                    fq = qs.fake_quant_weights()
                    for W_, b_ in fq:
                        W_.retain_grad()
                        b_.retain_grad()
                    out = qs.dec.forward(cali_data[idx], fq)
                else:
                    out = qs.forward(cali_data[idx])
                opt.zero_grad()
                count += 1
                rec = lp_loss(out, frames[idx], p)
                b = temp_decay(count, max_count, warmup, b_range[0], b_range[1])
                if count < loss_start or not round_on:
                    b, rl = 0, 0
                else:
                    rl = round_regulariser([L.wa for L in qs.dec.layers], b, weight)
                total = rl + rec
                total.backward()
                if probe is not None:
                    probe("ada" if round_on else "uaq", qs, fq)
                opt.step()
                log.append((float(total.detach()), float(rl.detach()) if torch.is_tensor(rl) else float(rl), float(b), count))
                done += 1
        for q in params:
            q.requires_grad_(False)

    # phase 1: learn delta (weight AND bias), Adam lr 1e-3, max_count 2100 hard-coded (calib_model.py:134-142)
    ep1 = int(0.05 * iters / n_batches)
    params = []
    for L in qs.dec.layers:
        params += [L.wd, L.bd]
    run(params, 0.001, ep1, 0, False, 2100)
    # phase 2: learn alpha (calib_model.py:170-226)
    qs.to_adaround()
    params = []
    for L in qs.dec.layers:
        params += [L.wa, L.ba]
    run(params, lr, int(iters / n_batches) - ep1, ep1, True, iters)
    # finish: weights go hard, biases stay soft (calib_model.py:231-240)
    for L in qs.dec.layers:
        L.w_soft = False
        for t in (L.wa, L.ba, L.wd, L.bd):
            t.requires_grad_(False)
    return log


# ----------------------------------------------------------------------------------------------
# Sensitivity criteria of the bit-allocation sweep (methods/bit_assign.py:57-118, 120-168, 171-217)
# ----------------------------------------------------------------------------------------------
def weight_perturbation(qs: QuantStack):
    """quant_layer.py:86-89 / quant_model.py:82-87: org_weight - weight_quantizer(weight), one tensor per layer.  The
    quantiser sees the raw (un-transformed) weight there, also when Hadamard is on."""
    out = []
    for L in qs.dec.layers:
        nl = 2 ** L.n_bits
        if qs.phase == "uaq":
            Wq = uaq_fake_quant(L.w, L.wd, L.wz, nl)
        else:
            Wq, _ = adaround_fake_quant(L.w, L.wa, L.wd, L.wz, nl, L.w_soft)
        out.append((L.w - Wq).detach())
    return out


def sensitivity(dec: Decoder, vec: Sequence[torch.Tensor], batches, mode: str, max_batches: int = 10):
    """mode 'omega': sum_l <v_l, (H v)_l>, H = Hessian of nn.MSELoss (mean over ALL elements) w.r.t. the decoder conv
    weights, accumulated (summed) over the first `max_batches` batches (bit_assign.py:57-118, 171-199).
    mode 'fisher_diag': sum_l <v_l^2, g_l^2>, g = dL/dW accumulated over the same batches (120-168, 200-211).
    batches: iterable of (emb, img).  Returns (total, per-layer list, per-layer accumulated gradient tensors)."""
    ws = [L.w.clone().requires_grad_(True) for L in dec.layers]
    bs = [L.b.clone().requires_grad_(True) for L in dec.layers]
    acc = [torch.zeros_like(w) for w in ws]
    for i, (emb, img) in enumerate(batches):
        if i >= max_batches:
            break
        out = dec.forward(emb, list(zip(ws, bs)))
        loss = F.mse_loss(out, img)
        if mode == "omega":
            grad_f = torch.autograd.grad(loss, ws, create_graph=True)
            prod = sum((g * v).sum() for g, v in zip(grad_f, vec))
            hv = torch.autograd.grad(prod, ws)
        elif mode == "fisher_diag":
            hv = torch.autograd.grad(loss, ws)
        else:
            raise ValueError('Not implemented sensitivity criteria: {}'.format(mode))
        for a, h in zip(acc, hv):
            a += h
    if mode == "omega":
        per = [(g * v).sum() for g, v in zip(acc, vec)]
    else:
        per = [(v.pow(2) * g.pow(2)).sum() for g, v in zip(acc, vec)]
    return sum(per), per, acc
