"""Quantised-model export: the integer grid tensors a downstream entropy coder needs.

The reference stops at `torch.save(qnn)` (methods/calibrate_network.py:304-308) and `QuantModel.get_quantized_param`
(quant_model.py:74-80); it computes only the average bit-width, never a bit-rate ("parity unpinned", build-defined).
Here: per layer the clamped integer levels x_quant (uint8, exactly what the calibrated forward uses: hard rounding for
weights), delta, zero_point, bit-width; plus two size estimates: the nominal bits/param (= set_bitwidth's average)
and the zeroth-order empirical entropy of the level histograms (what an ideal static entropy coder would reach).

The bias quirk (calib_model.py:231-240): after calibration the reference switches only the WEIGHT quantisers to hard
rounding; the bias quantisers stay soft, so the model whose PSNR the reference (and this repo's `evaluate`) reports uses
biases (floor(b/delta) + h(alpha) + zp - zp) * delta that are NOT on the integer grid.  The export therefore carries
both: `b{i}_levels` = the hard decision (what an entropy coder can store) and `b{i}_soft` = the fp32 bias the evaluated
model really used (`bias_soft: true` in the JSON).  `dequantize(path, bias=...)` rebuilds either model; the parity
test decodes both with the oracle and pins the evaluated PSNR to the soft one and the (small) gap of the hard one.
"""
import json
import math
import os

import numpy as np
import torch

from . import ops
from .quantization.quant_layer import QuantModule
from .quantization.quantizer import AdaRoundQuantizer


@torch.no_grad()
def _levels(q, x):
    """integer levels of tensor x under quantiser q (hard decision)."""
    if isinstance(q, AdaRoundQuantizer):
        _, xq = ops.adaround_forward(x, q.alpha.data, q.delta.data, q.zero_point, q.n_levels, False, want_xq=True)
    else:
        d, z = q.delta.data, q.zero_point
        xq = torch.clamp(torch.round(x / d) + z, 0, q.n_levels - 1)
    return xq.round().to(torch.uint8)


def _entropy_bits(levels: torch.Tensor, n_levels: int) -> float:
    hist = torch.bincount(levels.flatten().long(), minlength=n_levels).double()
    p = hist[hist > 0] / hist.sum()
    return float(-(p * torch.log2(p)).sum() * levels.numel())


@torch.no_grad()
def export_quantized(qnn, path, frames=None, height=None, width=None):
    """Write `<path>.npz` (levels, delta, zero_point per layer) and `<path>.json` (sizes / bpp).  Returns the summary.

    frames/height/width (optional) give bits per pixel = total bits / (frames * H * W)."""
    arrays, layers = {}, []
    nominal = entropy = n_param = 0
    for i, m in enumerate(mm for mm in qnn.model.modules() if isinstance(mm, QuantModule)):
        src = m.hadamard_weight if m.hadamard else m.org_weight
        wq, bq = m.weight_quantizer, m.bias_quantizer
        wl, bl = _levels(wq, src), _levels(bq, m.org_bias)
        arrays[f"w{i}_levels"], arrays[f"b{i}_levels"] = wl.cpu().numpy(), bl.cpu().numpy()
        # the bias the evaluated model uses (still soft after calibration, calib_model.py:231-240)
        arrays[f"b{i}_soft"] = bq(m.org_bias).detach().float().cpu().numpy()
        arrays[f"w{i}_cin"] = np.array(m.weight.shape[1])
        for tag, q in (("w", wq), ("b", bq)):
            arrays[f"{tag}{i}_delta"] = q.delta.detach().float().cpu().numpy()
            arrays[f"{tag}{i}_zero_point"] = q.zero_point.detach().float().cpu().numpy()
        e = _entropy_bits(wl, wq.n_levels) + _entropy_bits(bl, bq.n_levels)
        nb = wq.n_bits * m.weight.numel() + bq.n_bits * m.bias.numel()   # quant_model.py:68 counts the unpadded weight
        layers.append(dict(layer=i, shape=list(m.weight.shape), hadamard=bool(m.hadamard), n_bits=wq.n_bits,
                           bias_soft=bool(getattr(bq, "soft_targets", False)),
                           levels_stored=int(wl.numel() + bl.numel()), nominal_bits=int(nb), entropy_bits=round(e, 1)))
        nominal += nb
        entropy += e
        n_param += m.weight.numel() + m.bias.numel()
    summary = dict(layers=layers, params=int(n_param), avg_bits_nominal=nominal / n_param,
                   avg_bits_entropy=entropy / n_param, total_bytes_nominal=math.ceil(nominal / 8),
                   total_bytes_entropy=math.ceil(entropy / 8))
    if frames and height and width:
        px = frames * height * width
        summary.update(bpp_nominal=nominal / px, bpp_entropy=entropy / px)
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    np.savez_compressed(path + ".npz", **arrays)
    with open(path + ".json", "w") as f:
        json.dump(summary, f, indent=1)
    return summary


def dequantize(path, bias="soft"):
    """[(W, b)] per layer (numpy fp32) from `<path>.npz`: W = (levels - zero_point) * delta (Hadamard-domain levels are
    returned as stored; the caller applies H and the [:, :C_in] slice), b = the exported soft bias (`bias='soft'`, the
    evaluated model) or its hard integer decision (`bias='hard'`, what a bit stream would carry)."""
    z = np.load(path + ".npz")
    out, i = [], 0
    while f"w{i}_levels" in z.files:
        W = (z[f"w{i}_levels"].astype(np.float32) - z[f"w{i}_zero_point"]) * z[f"w{i}_delta"]
        if bias == "soft":
            b = z[f"b{i}_soft"]
        else:
            b = (z[f"b{i}_levels"].astype(np.float32) - z[f"b{i}_zero_point"]) * z[f"b{i}_delta"]
        out.append((W.astype(np.float32), b.astype(np.float32)))
        i += 1
    return out
