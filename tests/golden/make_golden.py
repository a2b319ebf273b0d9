#!/usr/bin/env python3
"""Golden-vector generator (build container only; NOT run by tests, bench or the product).

Imports the real reference from /root/reference read-only (python -B, no bytecode written),
with the four missing pip packages replaced by `_ref_stubs.py`, runs the reference's own
classes/functions for the calibration hot path on seeded inputs and writes the inputs and
outputs as small .npz/.json fixtures into this directory.  Only DATA is written; no reference
source or bytecode is copied.

    cd /root/repo && python3 -B tests/golden/make_golden.py [--only NAME ...]

Fixtures (what each one pins; reference file:line):
  uaq.npz            UniformAffineQuantizer init 'max' + forward + d(delta)   quantizer.py:111-168
  adaround.npz       AdaRoundQuantizer init_alpha / soft+hard forward / grads quantizer.py:259-319
  zero_channel.npz   the all-zero-output-channel quirk (delta 1e-8 -> fp16 0 -> NaN alpha / NaN forward)  quantizer.py:163-165, 264-265
  roundloss.npz      rounding regulariser value + grad                        calib_model.py:39-47
  tempdecay.json     LinearTempDecay table                                    data_utils.py:24-41
  quantmodule.npz    QuantModule.forward, Hadamard off/on (on = FWHT stub)    quant_layer.py:24-89
  decode.npz         tiny HNeRV / NeRV decode (FP + quantised), avg bit-width HNeRV.py:49-71, NeRV.py:44-65,
                                                                              quant_model.py:58-72
  frames_320x640.npz 8 Bunny frames, 2x area-downsampled + center-cropped     (data)
  bunny8_640x1280.npz the first 8 Bunny frames, center-cropped to 640x1280 as the reference's loader does
                     (data; the operating point of the reference's logged runs)       videosets/datasets.py:19-28
  traj_hnerv.npz     tiny HNeRV: checkpoint, embeddings, batch order, per-iteration losses of the
                     real model_reconstruction, final alpha/delta/x_quant, PSNRs    calib_model.py:92-240
  traj_nerv_had.npz  tiny NeRV + Hadamard: same (FWHT via stub -> "parity unpinned" at the transform)
  config1_hnerv3m.npz  BASELINE configs[0] at FULL size with the real reference: HNeRV-3M (trained checkpoint
                     hnerv3m_bunny8_f16.npz), 8 frames of 640x1280, 48 iterations: loss log + PSNRs     calib_model.py:92-240
  config1_hnerv3m_long.npz  the real reference for 3000 iterations at full size (148 phase-1 + 2852 phase-2), every loss, final
                     PSNRs / scales / masks                                                               calib_model.py:92-240
  config2_nerv3m_hadamard.npz   BASELINE configs[2] at FULL size with the real reference: NeRV-3M + --hadamard (trained
                     checkpoint nerv3m_bunny8real_f16.npz), the 8 real 640x1280 crops, 48 iterations            quant_layer.py:44-49, 70-71
  config1_hnerv3m_hadamard.npz  the same for HNeRV-3M + --hadamard on hnerv3m_bunny8real_f16.npz                 calib_model.py:170-191
  omega.npz          tiny HNeRV (checkpoint of traj_hnerv.npz): get_perturbation() and the reference's own
                     sensitivity_criterion ('omega' = v'Hv by double backward, 'fisher_diag') for the two toy
                     candidates of bit_assign.py:27-30                          bit_assign.py:57-217
"""
import argparse
import json
import math
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import _ref_stubs

_ref_stubs.install()

from quantization.quantizer import UniformAffineQuantizer, AdaRoundQuantizer, lp_loss  # noqa: E402
from quantization.quant_layer import QuantModule  # noqa: E402
from quantization.quant_model import QuantModel  # noqa: E402
import quantization.calib_model as ref_calib  # noqa: E402
from quantization.data_utils import LinearTempDecay  # noqa: E402
from models import HNeRV, NeRV  # noqa: E402

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


# ----------------------------------------------------------------------------- uaq
def edge_weight(g, shape):
    """rows: random, all-positive, all-negative, constant, zeros, random*20, tiny, random..."""
    w = torch.randn(shape, generator=g)
    w[1] = w[1].abs() + 0.01
    w[2] = -w[2].abs() - 0.01
    w[3] = 0.37
    w[4] = 0.0
    w[5] = w[5] * 20
    if shape[0] > 6:
        w[6] = w[6] * 1e-3
    return w


def gen_uaq():
    g = torch.Generator().manual_seed(1)
    out = {}
    for nb in range(2, 9):
        # channel-wise weight
        w = edge_weight(g, (8, 5, 3, 3))
        q = UniformAffineQuantizer(n_bits=8, channel_wise=True, scale_method="max")
        q.bitwidth_refactor(nb)
        y = q(w)
        go = torch.randn(w.shape, generator=g)
        (y * go).sum().backward()
        out[f"w{nb}_x"] = npy(w)
        out[f"w{nb}_delta"] = npy(q.delta)
        out[f"w{nb}_zp"] = npy(q.zero_point)
        out[f"w{nb}_y"] = npy(y)
        out[f"w{nb}_go"] = npy(go)
        out[f"w{nb}_ddelta"] = npy(q.delta.grad)
        # second forward after a delta perturbation (phase-1 state: delta no longer the init value)
        with torch.no_grad():
            q.delta.mul_(1.0 + 0.05 * torch.randn(q.delta.shape, generator=g))
        q.delta.grad = None
        y2 = q(w)
        (y2 * go).sum().backward()
        out[f"w{nb}_delta2"] = npy(q.delta)
        out[f"w{nb}_y2"] = npy(y2)
        out[f"w{nb}_ddelta2"] = npy(q.delta.grad)
        # bias (1-D, one scalar delta/zp even when channel_wise)
        b = torch.randn(11, generator=g) * 0.3
        qb = UniformAffineQuantizer(n_bits=8, channel_wise=True, scale_method="max")
        qb.bitwidth_refactor(nb)
        yb = qb(b)
        gb = torch.randn(b.shape, generator=g)
        (yb * gb).sum().backward()
        out[f"b{nb}_x"] = npy(b)
        out[f"b{nb}_delta"] = npy(qb.delta)
        out[f"b{nb}_zp"] = npy(qb.zero_point)
        out[f"b{nb}_y"] = npy(yb)
        out[f"b{nb}_go"] = npy(gb)
        out[f"b{nb}_ddelta"] = npy(qb.delta.grad)
    # layer-wise (channel_wise=False) weight
    w = torch.randn(4, 3, 3, 3, generator=g)
    q = UniformAffineQuantizer(n_bits=5, channel_wise=False, scale_method="max")
    y = q(w)
    go = torch.randn(w.shape, generator=g)
    (y * go).sum().backward()
    out["lw_x"], out["lw_delta"], out["lw_zp"], out["lw_y"] = npy(w), npy(q.delta), npy(q.zero_point), npy(y)
    out["lw_go"], out["lw_ddelta"] = npy(go), npy(q.delta.grad)
    save("uaq.npz", **out)


# ----------------------------------------------------------------------------- adaround
def gen_adaround():
    g = torch.Generator().manual_seed(2)
    out = {}
    for nb in (2, 3, 4, 6, 8):
        w = edge_weight(g, (8, 5, 3, 3))
        w[4] = torch.randn(5, 3, 3, generator=g) * 0.05  # an all-zero row gives delta=1e-8 -> fp16 0 -> NaN; keep finite here
        q = UniformAffineQuantizer(n_bits=8, channel_wise=True, scale_method="max")
        q.bitwidth_refactor(nb)
        q(w)
        out[f"w{nb}_x"] = npy(w)
        out[f"w{nb}_uaq_delta"] = npy(q.delta)
        out[f"w{nb}_uaq_zp"] = npy(q.zero_point)
        a = AdaRoundQuantizer(uaq=q, round_mode="learned_hard_sigmoid", weight_tensor=w)
        out[f"w{nb}_delta"] = npy(a.delta)
        out[f"w{nb}_zp"] = npy(a.zero_point)
        out[f"w{nb}_alpha0"] = npy(a.alpha)
        # move alpha away from init so that some h(alpha) saturate at 0 / 1
        with torch.no_grad():
            a.alpha.add_(3.0 * torch.randn(a.alpha.shape, generator=g))
            a.alpha.view(-1)[:4] = torch.tensor([0.0, -0.0, 2.3978953, -2.3978953])  # sign + near clamp edge
        out[f"w{nb}_alpha"] = npy(a.alpha)
        a.soft_targets = True
        ys = a(w)
        go = torch.randn(w.shape, generator=g)
        (ys * go).sum().backward()
        out[f"w{nb}_ysoft"] = npy(ys)
        out[f"w{nb}_xq_soft"] = npy(a.x_quant)
        out[f"w{nb}_go"] = npy(go)
        out[f"w{nb}_dalpha"] = npy(a.alpha.grad)
        out[f"w{nb}_ddelta"] = npy(a.delta.grad)
        a.soft_targets = False
        yh = a(w)
        out[f"w{nb}_yhard"] = npy(yh)
        out[f"w{nb}_xq_hard"] = npy(a.x_quant)
        # bias
        b = torch.randn(13, generator=g) * 0.2
        qb = UniformAffineQuantizer(n_bits=8, channel_wise=True, scale_method="max")
        qb.bitwidth_refactor(nb)
        qb(b)
        ab = AdaRoundQuantizer(uaq=qb, round_mode="learned_hard_sigmoid", weight_tensor=b)
        ab.soft_targets = True
        yb = ab(b)
        gb = torch.randn(b.shape, generator=g)
        (yb * gb).sum().backward()
        out[f"b{nb}_x"], out[f"b{nb}_uaq_delta"], out[f"b{nb}_uaq_zp"] = npy(b), npy(qb.delta), npy(qb.zero_point)
        out[f"b{nb}_delta"], out[f"b{nb}_zp"], out[f"b{nb}_alpha0"] = npy(ab.delta), npy(ab.zero_point), npy(ab.alpha)
        out[f"b{nb}_ysoft"], out[f"b{nb}_go"], out[f"b{nb}_dalpha"] = npy(yb), npy(gb), npy(ab.alpha.grad)
    save("adaround.npz", **out)


# ----------------------------------------------------------------------------- all-zero output channel
def gen_zero_channel():
    """an all-zero output channel gets delta = 1e-8 from the 'max' init (quantizer.py:163-165), which the
    fp16 round trip of AdaRoundQuantizer.__init__ (quantizer.py:264-265) flushes to 0 -> x/delta = 0/0 -> alpha = NaN and
    every forward of that channel is NaN in the reference.  Recorded here exactly as the reference produces it."""
    g = torch.Generator().manual_seed(12)
    w = torch.randn(4, 5, 3, 3, generator=g) * 0.3
    w[1] = 0.0
    q = UniformAffineQuantizer(n_bits=8, channel_wise=True, scale_method="max")
    q.bitwidth_refactor(4)
    y_uaq = q(w)
    out = {"x": npy(w), "uaq_delta": npy(q.delta), "uaq_zp": npy(q.zero_point), "y_uaq": npy(y_uaq)}
    a = AdaRoundQuantizer(uaq=q, round_mode="learned_hard_sigmoid", weight_tensor=w)
    out["delta"], out["zp"], out["alpha0"] = npy(a.delta), npy(a.zero_point), npy(a.alpha)
    a.soft_targets = True
    ys = a(w)
    go = torch.randn(w.shape, generator=g)
    (ys * go).sum().backward()
    out["ysoft"], out["go"], out["dalpha"] = npy(ys), npy(go), npy(a.alpha.grad)
    a.soft_targets = False
    out["yhard"] = npy(a(w))
    print("  zero channel: uaq delta", out["uaq_delta"].ravel(), "-> fp16 delta", out["delta"].ravel(),
          "| NaN alpha rows", np.isnan(out["alpha0"]).reshape(4, -1).all(1), "| NaN ysoft rows", np.isnan(out["ysoft"]).reshape(4, -1).all(1))
    save("zero_channel.npz", **out)


# ----------------------------------------------------------------------------- round loss
def gen_roundloss():
    g = torch.Generator().manual_seed(3)
    out = {}
    conv = nn.Conv2d(5, 8, 3, 1, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
    qm = QuantModule(conv, hadamard=False, weight_quant_params=dict(n_bits=4, channel_wise=True, scale_method="max"))
    qm.set_quant_state(True)
    qm(torch.randn(1, 5, 4, 4, generator=g))
    qm.weight_quantizer = AdaRoundQuantizer(uaq=qm.weight_quantizer, round_mode="learned_hard_sigmoid",
                                            weight_tensor=qm.org_weight.data)
    with torch.no_grad():
        qm.weight_quantizer.alpha.add_(2.0 * torch.randn(qm.weight_quantizer.alpha.shape, generator=g))
    out["alpha"] = npy(qm.weight_quantizer.alpha)
    holder = nn.Sequential(qm)
    for b in (20, 7.3, 2):
        lf = ref_calib.LossFunction(holder, round_loss="relaxation", weight=0.01, max_count=100,
                                    b_range=(20, 2), warmup=0.0)
        lf.round_loss = 0
        qm.weight_quantizer.alpha.grad = None
        lf.collect_round_loss(holder, b)
        lf.round_loss.backward()
        tag = str(b).replace(".", "p")
        out[f"loss_b{tag}"] = npy(lf.round_loss)
        out[f"dalpha_b{tag}"] = npy(qm.weight_quantizer.alpha.grad)
    save("roundloss.npz", **out)


# ----------------------------------------------------------------------------- temperature schedule
def gen_tempdecay():
    res = {}
    for t_max, rel in ((21000, 0.2), (2100, 0.2), (400, 0.2), (50, 0.2), (1000, 0.0)):
        sched = LinearTempDecay(t_max, rel_start_decay=rel, start_b=20, end_b=2)
        ts = sorted(set([1, 2, 5, 9, 10, 11, 40, 79, 80, 81, 100, 399, 400, 401, 4199, 4200, 4201, 4500, 10000,
                         19500, 19998, 21000, 22000]))
        res[f"{t_max}_{rel}"] = [[t, float(sched(t))] for t in ts]
    with open(os.path.join(HERE, "tempdecay.json"), "w") as f:
        json.dump(res, f, indent=0)
    print("wrote tempdecay.json")


# ----------------------------------------------------------------------------- QuantModule
def gen_quantmodule():
    g = torch.Generator().manual_seed(5)
    out = {}
    for (co, ci, k) in ((8, 5, 3), (12, 16, 1), (6, 37, 5)):
        for had in (False, True):
            tag = f"c{co}_{ci}_{k}_{'h' if had else 'n'}"
            conv = nn.Conv2d(ci, co, k, 1, k // 2)
            with torch.no_grad():
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
                conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.1)
            qm = QuantModule(conv, hadamard=had, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
            qm.weight_quantizer.bitwidth_refactor(4)
            qm.bias_quantizer.bitwidth_refactor(4)
            x = torch.randn(2, ci, 6, 7, generator=g)
            out[f"{tag}_w"], out[f"{tag}_b"], out[f"{tag}_x"] = npy(conv.weight), npy(conv.bias), npy(x)
            qm.set_quant_state(False)
            out[f"{tag}_y_fp"] = npy(qm(x))
            qm.set_quant_state(True)
            y = qm(x)
            out[f"{tag}_y_uaq"] = npy(y)
            out[f"{tag}_wdelta"] = npy(qm.weight_quantizer.delta)
            out[f"{tag}_wzp"] = npy(qm.weight_quantizer.zero_point)
            out[f"{tag}_bdelta"] = npy(qm.bias_quantizer.delta)
            out[f"{tag}_bzp"] = npy(qm.bias_quantizer.zero_point)
            go = torch.randn(y.shape, generator=g)
            (y * go).sum().backward()
            out[f"{tag}_go"] = npy(go)
            out[f"{tag}_dwdelta"] = npy(qm.weight_quantizer.delta.grad)
            out[f"{tag}_dbdelta"] = npy(qm.bias_quantizer.delta.grad)
            if had:
                out[f"{tag}_hw"] = npy(qm.hadamard_weight)
            # phase-2 state
            wt = qm.hadamard_weight.data if had else qm.org_weight.data
            qm.weight_quantizer = AdaRoundQuantizer(uaq=qm.weight_quantizer, round_mode="learned_hard_sigmoid", weight_tensor=wt)
            qm.bias_quantizer = AdaRoundQuantizer(uaq=qm.bias_quantizer, round_mode="learned_hard_sigmoid", weight_tensor=qm.bias.data)
            qm.weight_quantizer.soft_targets = True
            qm.bias_quantizer.soft_targets = True
            with torch.no_grad():
                qm.weight_quantizer.alpha.add_(1.5 * torch.randn(qm.weight_quantizer.alpha.shape, generator=g))
                qm.bias_quantizer.alpha.add_(1.5 * torch.randn(qm.bias_quantizer.alpha.shape, generator=g))
            out[f"{tag}_walpha"], out[f"{tag}_balpha"] = npy(qm.weight_quantizer.alpha), npy(qm.bias_quantizer.alpha)
            xin = x.clone().requires_grad_(True)
            y = qm(xin)
            (y * go).sum().backward()
            out[f"{tag}_y_ada"] = npy(y)
            out[f"{tag}_dwalpha"] = npy(qm.weight_quantizer.alpha.grad)
            out[f"{tag}_dbalpha"] = npy(qm.bias_quantizer.alpha.grad)
            out[f"{tag}_dx"] = npy(xin.grad)
            qm.weight_quantizer.soft_targets = False
            out[f"{tag}_y_hard"] = npy(qm(x))  # NB bias quantizer stays soft (calib_model.py:231-240)
    save("quantmodule.npz", **out)


# ----------------------------------------------------------------------------- tiny configs
TINY_HNERV = dict(crop_h=320, crop_w=640, diff_enc=False, stage_block=1, enc_strides=[5, 4, 4, 2, 2],
                  enc_channel=[16, 16, 16, 16, 8], channel_reduce=1.2, channel_lbound=6, dec_in_channel=12,
                  dec_kernels=[1, 3, 5, 5, 5], dec_strides=[5, 4, 4, 2, 2], dec_norm="none", dec_acts="gelu",
                  out_bias="tanh")
TINY_NERV = dict(crop_h=320, crop_w=640, diff_enc=False, base=1.25, level=20, channel_reduce=2, channel_lbound=6,
                 dec_in_channel=20, dec_kernels=[3, 3, 3, 3, 3], dec_strides=[5, 4, 4, 2, 2], dec_norm="none",
                 dec_acts="gelu", out_bias="tanh")
BITS = [6, 5, 4, 5, 5, 6, 6]


def seeded_init(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(fan_in))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)


def sd_np(model, prefix=""):
    return {prefix + k.replace(".", "/"): npy(v) for k, v in model.state_dict().items()}


def gen_decode():
    out = {}
    g = torch.Generator().manual_seed(7)
    for name, cls, cfg in (("hnerv", HNeRV, TINY_HNERV), ("nerv", NeRV, TINY_NERV)):
        torch.manual_seed(11)
        model = cls(cfg)
        seeded_init(model, 17)
        model.eval()
        out.update(sd_np(model, f"{name}_sd:"))
        if name == "hnerv":
            emb = torch.randn(2, cfg["enc_channel"][-1], 1, 2, generator=g) * 0.5
        else:
            emb = model.encode(torch.tensor([0.25, 0.625]))
            out["nerv_norm_idx"] = np.array([0.25, 0.625], dtype=np.float32)
        out[f"{name}_emb"] = npy(emb)
        with torch.no_grad():
            y, elist, _ = model.decode(emb)
        out[f"{name}_y_fp_sub"] = npy(y[..., ::7, ::7])  # strided subsample keeps the fixture small
        out[f"{name}_y_fp_sum"] = np.array(y.double().sum().item())
        for i, e in enumerate(elist):
            out[f"{name}_embed_shape{i}"] = np.array(e.shape)
            if e.numel() <= 20000:
                out[f"{name}_embed_list{i}"] = npy(e)
        for had in (False, True):
            import copy
            m2 = copy.deepcopy(model)
            qnn = QuantModel(m2, hadamard=had, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
            avg = qnn.set_bitwidth(BITS)
            qnn.eval()
            qnn.set_quant_state(True)
            with torch.no_grad():
                yq, _, _ = qnn(emb)
            tag = f"{name}_{'h' if had else 'n'}"
            out[f"{tag}_avgbits"] = np.array(avg, dtype=np.float64)
            out[f"{tag}_y_q_sub"] = npy(yq[..., ::7, ::7])
            out[f"{tag}_y_q_sum"] = np.array(yq.double().sum().item())
    save("decode.npz", **out)


# ----------------------------------------------------------------------------- frames
def gen_frames():
    from PIL import Image
    idxs = [1, 17, 33, 49, 65, 81, 97, 113]
    frames = []
    for i in idxs:
        im = Image.open(f"{REF}/bunny/{i:04d}.png").convert("RGB")
        a = np.asarray(im).astype(np.float32)  # 720 x 1280 x 3
        a = a.reshape(360, 2, 640, 2, 3).mean(axis=(1, 3))  # 2x area downsample -> 360 x 640
        a = a[20:340]  # center crop to 320 x 640
        frames.append(np.clip(np.rint(a), 0, 255).astype(np.uint8).transpose(2, 0, 1))
    frames = np.stack(frames)
    save("frames_320x640.npz", frames=frames, src_index=np.array(idxs))
    return frames


def gen_bunny8_real():
    """bunny8_640x1280.npz: the first 8 Bunny frames exactly as the reference's loader hands them over -- rows 40:680 of
    the 720x1280 PNGs = center_crop(img, (640, 1280)) (videosets/datasets.py:19-28), uint8, no resampling.
    Data only: the PNG pixels, cropped."""
    from PIL import Image
    frames = []
    for i in range(1, 9):
        a = np.asarray(Image.open(f"{REF}/bunny/{i:04d}.png").convert("RGB"))
        assert a.shape == (720, 1280, 3)
        frames.append(a[40:680].transpose(2, 0, 1).copy())
    save("bunny8_640x1280.npz", frames=np.stack(frames), src_index=np.arange(1, 9))


def load_frames():
    p = os.path.join(HERE, "frames_320x640.npz")
    if not os.path.exists(p):
        gen_frames()
    return torch.from_numpy(np.load(p)["frames"]).float() / 255.0


def psnr_frames(out, gt):
    mse = F.mse_loss(out, gt, reduction="none").flatten(1).mean(1)
    return -10 * torch.log10(mse + 1e-9)


class ReplayLoader:
    """gt stand-in for model_reconstruction: len() + iteration over a recorded list of index batches."""

    def __init__(self, frames, order, n_frames):
        self.frames, self.order, self.pos, self.n = frames, order, 0, n_frames

    def __len__(self):
        return self.order.shape[1]

    def __iter__(self):
        ep = self.order[self.pos]
        self.pos += 1
        for idx in ep:
            idx_t = torch.as_tensor(idx, dtype=torch.int64)
            yield {"img": self.frames[idx_t], "idx": idx_t, "norm_idx": idx_t.float() / self.n}


def train_fp(model, frames, arch, epochs, lr, seed):
    """Plain FP32 fit (NOT the reference trainer, only a way to get a non-trivial checkpoint)."""
    torch.manual_seed(seed)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    n = frames.shape[0]
    total = epochs * n
    it = 0
    model.train()
    for ep in range(epochs):
        perm = torch.randperm(n)
        for i in perm.tolist():
            cur_lr = lr * 0.5 * (1 + math.cos(math.pi * it / total)) if it > 0.1 * total else lr * it / (0.1 * total) + 1e-6
            for gr in opt.param_groups:
                gr["lr"] = cur_lr
            img = frames[i:i + 1]
            inp = img if arch == "hnerv" else torch.tensor([i / n])
            out, _, _ = model(inp)
            loss = F.mse_loss(out, img)
            opt.zero_grad()
            loss.backward()
            opt.step()
            it += 1
        if ep % 20 == 0 or ep == epochs - 1:
            print(f"  train ep {ep} loss {loss.item():.5f}", flush=True)
    model.eval()


def gen_traj(name, arch, cls, cfg, hadamard, iters, train_epochs, lr_train, seed):
    frames = load_frames()
    n, B = frames.shape[0], 2
    torch.manual_seed(seed)
    model = cls(cfg)
    t0 = time.time()
    train_fp(model, frames, arch, train_epochs, lr_train, seed)
    print(f"  trained in {time.time() - t0:.1f}s")
    out = {}
    out.update(sd_np(model, "sd:"))
    with torch.no_grad():
        if arch == "hnerv":
            emb = torch.cat([model.encode(frames[i:i + 1]) for i in range(n)], 0)
        else:
            emb = torch.cat([model.encode(torch.tensor([i / n])) for i in range(n)], 0)
        y_fp = torch.cat([model.decode(emb[i:i + 1])[0] for i in range(n)], 0)
    out["emb"] = npy(emb)
    out["psnr_fp"] = npy(psnr_frames(y_fp, frames))
    print("  FP psnr", out["psnr_fp"].mean())

    qnn = QuantModel(model=model, hadamard=hadamard, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    out["avgbits"] = np.array(qnn.set_bitwidth(BITS), dtype=np.float64)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:B])
        y_q0 = torch.cat([qnn(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_q_noopt"] = npy(psnr_frames(y_q0, frames))
    print("  quant w/o opt psnr", out["psnr_q_noopt"].mean())
    qms = [m for m in qnn.model.modules() if isinstance(m, QuantModule)]
    for li, m in enumerate(qms):
        out[f"init_wdelta{li}"] = npy(m.weight_quantizer.delta)
        out[f"init_wzp{li}"] = npy(m.weight_quantizer.zero_point)
        out[f"init_bdelta{li}"] = npy(m.bias_quantizer.delta)
        out[f"init_bzp{li}"] = npy(m.bias_quantizer.zero_point)

    # recorded batch order: one permutation per epoch (shuffle=True, drop_last=True; calibrate_network.py:162-165)
    g = torch.Generator().manual_seed(903)
    n_ep = iters // (n // B)
    order = torch.stack([torch.randperm(n, generator=g)[: (n // B) * B].view(n // B, B) for _ in range(n_ep)]).numpy()
    out["order"] = order
    loader = ReplayLoader(frames, order, n)

    log = []
    orig_call = ref_calib.LossFunction.__call__

    def recording_call(self, pred, tgt, grad=None):
        total = orig_call(self, pred, tgt, grad)
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round == "none":
            b = 0
        log.append((float(total), float(self.round_loss), float(b), self.count))
        return total

    ref_calib.LossFunction.__call__ = recording_call
    t0 = time.time()
    try:
        ref_calib.model_reconstruction(qnn, cali_data=emb, gt=loader, arch=arch, batch_size=B, iters=iters,
                                       weight=0.01, opt_mode="mse", hadamard=hadamard, b_range=(20, 2),
                                       warmup=0.2, p=2.0, lr=0.003)
    finally:
        ref_calib.LossFunction.__call__ = orig_call
    print(f"  model_reconstruction: {len(log)} iterations in {time.time() - t0:.1f}s")
    out["loss_log"] = np.array(log, dtype=np.float64)  # total, round, b, count
    out["iters"] = np.array(iters)

    qnn.set_quant_state(True)
    with torch.no_grad():
        y_q1 = torch.cat([qnn(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_q_opt"] = npy(psnr_frames(y_q1, frames))
    print("  quant w/ opt psnr", out["psnr_q_opt"].mean())
    for li, m in enumerate(qms):
        out[f"fin_walpha{li}"] = npy(m.weight_quantizer.alpha)
        out[f"fin_balpha{li}"] = npy(m.bias_quantizer.alpha)
        out[f"fin_wdelta{li}"] = npy(m.weight_quantizer.delta)
        out[f"fin_wzp{li}"] = npy(m.weight_quantizer.zero_point)
        out[f"fin_bdelta{li}"] = npy(m.bias_quantizer.delta)
        out[f"fin_bzp{li}"] = npy(m.bias_quantizer.zero_point)
        out[f"fin_wxq{li}"] = npy(m.weight_quantizer.x_quant).astype(np.int16)
        out[f"fin_bxq{li}"] = npy(m.bias_quantizer.x_quant)
    save(name, **out)


def gen_omega():
    """Runs methods/bit_assign.py:sensitivity_criterion itself.  The only accommodation: run_hessian_vector_product
    calls `.cuda()` unconditionally (bit_assign.py:110) and this container has no GPU, so Tensor.cuda is an identity
    for the duration of the call (arithmetic untouched, everything stays fp32 on the CPU)."""
    import copy
    import methods.bit_assign as ref_ba
    z = np.load(os.path.join(HERE, "traj_hnerv.npz"))
    model = HNeRV(TINY_HNERV)
    model.load_state_dict({k[3:].replace("/", "."): torch.from_numpy(z[k]) for k in z.files if k.startswith("sd:")}, strict=True)
    model.eval()
    frames = load_frames()
    n, B = frames.shape[0], 2
    batches = [dict(img=frames[i:i + B], idx=torch.arange(i, i + B), norm_idx=torch.arange(i, i + B).float() / n)
               for i in range(0, n, B)]
    with torch.no_grad():
        emb = torch.cat([model.encode(frames[i:i + 1]) for i in range(n)], 0)
    out = {"bits": np.array(list(ref_ba.hnerv_candidate.values()))}
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for ci, (cand, bits) in enumerate(ref_ba.hnerv_candidate.items()):
            for mode in ("omega", "fisher_diag"):
                # same sequence as bit_assign.py:343-359
                qnn = QuantModel(model=copy.deepcopy(model), hadamard=False,
                                 weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
                qnn.eval()
                out[f"avgbits{ci}"] = np.array(qnn.set_bitwidth(bits), dtype=np.float64)
                qnn.set_quant_state(True)
                with torch.no_grad():
                    qnn(emb[:B])
                net = copy.deepcopy(model)
                score = ref_ba.sensitivity_criterion(mode, "hnerv", net, qnn, batches, use_cuda=False)
                vec = qnn.get_perturbation()
                acc = ref_ba.gradtensor_to_vec(net)
                out[f"{mode}{ci}"] = np.array(float(score), dtype=np.float64)
                if mode == "omega":
                    out[f"{mode}{ci}_layers"] = np.array([float((g * v).sum()) for g, v in zip(acc, vec)])
                else:
                    out[f"{mode}{ci}_layers"] = np.array([float((v.pow(2) * g.pow(2)).sum()) for g, v in zip(acc, vec)])
                print(f"  {cand} {bits} {mode}: {float(score):.6e}")
                if ci == 1:
                    for li, (g, v) in enumerate(zip(acc, vec)):
                        out[f"{mode}_acc{li}"] = npy(g)
                        out[f"vec{li}"] = npy(v)
    finally:
        torch.Tensor.cuda = orig_cuda
    save("omega.npz", **out)


HNERV_3M = dict(crop_h=640, crop_w=1280, diff_enc=False, stage_block=1, enc_strides=[5, 4, 4, 2, 2],
                enc_channel=[64, 64, 64, 64, 16], channel_reduce=1.2, channel_lbound=12, dec_in_channel=92,
                dec_kernels=[1, 3, 5, 5, 5], dec_strides=[5, 4, 4, 2, 2], dec_norm="none", dec_acts="gelu",
                out_bias="tanh")


def gen_config1():
    """BASELINE configs[0] with the REAL reference at full size: HNeRV Bunny_1280x640_3M, the trained
    checkpoint of hnerv3m_bunny8_f16.npz (make_ckpt_fixture.py), 8 Bunny-derived 640x1280 frames (frames_320x640.npz with
    every pixel repeated 2x2), --precision 6 5 4 5 5 6 6, batch 2, iters_w = 50 -> 0 phase-1 epochs + 12 phase-2 epochs =
    48 iterations of the reference's own model_reconstruction.  Stored: the recorded batch order, the 48-entry loss log,
    per-frame PSNRs (FP / quant w/o opt / quant w/ opt), initial scales.  ~4 min of CPU."""
    ck = np.load(os.path.join(HERE, "hnerv3m_bunny8_f16.npz"))
    small = np.load(os.path.join(HERE, "frames_320x640.npz"))["frames"]
    frames = torch.from_numpy(np.repeat(np.repeat(small, 2, axis=2), 2, axis=3).copy()).float() / 255.0
    n, B, iters = frames.shape[0], 2, 50
    torch.manual_seed(1)
    model = HNeRV(HNERV_3M)
    sd = {k[3:].replace("/", "."): torch.from_numpy(ck[k].astype(np.float32)) for k in ck.files if k.startswith("sd:")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("encoder") for m in missing), (missing, unexpected)
    model.eval()
    emb = torch.from_numpy(ck["emb"])
    out = {}
    with torch.no_grad():
        y_fp = torch.cat([model.decode(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_fp"] = npy(psnr_frames(y_fp, frames))
    print("  FP psnr", out["psnr_fp"].mean(), flush=True)
    qnn = QuantModel(model=model, hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    out["avgbits"] = np.array(qnn.set_bitwidth(BITS), dtype=np.float64)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:B])
        y_q0 = torch.cat([qnn(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_q_noopt"] = npy(psnr_frames(y_q0, frames))
    print("  quant w/o opt psnr", out["psnr_q_noopt"].mean(), flush=True)
    qms = [m for m in qnn.model.modules() if isinstance(m, QuantModule)]
    for li, m in enumerate(qms):
        out[f"init_wdelta{li}"] = npy(m.weight_quantizer.delta)
        out[f"init_bdelta{li}"] = npy(m.bias_quantizer.delta)
    g = torch.Generator().manual_seed(903)
    n_ep = iters // (n // B)
    order = torch.stack([torch.randperm(n, generator=g)[: (n // B) * B].view(n // B, B) for _ in range(n_ep)]).numpy()
    out["order"] = order
    loader = ReplayLoader(frames, order, n)
    log = []
    orig_call = ref_calib.LossFunction.__call__

    def recording_call(self, pred, tgt, grad=None):
        total = orig_call(self, pred, tgt, grad)
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round == "none":
            b = 0
        log.append((float(total), float(self.round_loss), float(b), self.count))
        return total

    ref_calib.LossFunction.__call__ = recording_call
    t0 = time.time()
    try:
        ref_calib.model_reconstruction(qnn, cali_data=emb, gt=loader, arch="hnerv", batch_size=B, iters=iters, weight=0.01,
                                       opt_mode="mse", hadamard=False, b_range=(20, 2), warmup=0.2, p=2.0, lr=0.003)
    finally:
        ref_calib.LossFunction.__call__ = orig_call
    print(f"  model_reconstruction: {len(log)} iterations in {time.time() - t0:.1f}s", flush=True)
    out["loss_log"] = np.array(log, dtype=np.float64)
    out["iters"] = np.array(iters)
    out["seconds_cpu8"] = np.array(time.time() - t0)
    qnn.set_quant_state(True)
    with torch.no_grad():
        y_q1 = torch.cat([qnn(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_q_opt"] = npy(psnr_frames(y_q1, frames))
    print("  quant w/ opt psnr", out["psnr_q_opt"].mean(), flush=True)
    save("config1_hnerv3m.npz", **out)


NERV_3M = dict(crop_h=640, crop_w=1280, diff_enc=False, base=1.25, level=80, channel_reduce=2, channel_lbound=24,
               dec_in_channel=145, dec_kernels=[3, 3, 3, 3, 3], dec_strides=[5, 4, 4, 2, 2], dec_norm="none",
               dec_acts="gelu", out_bias="tanh")   # configs/NeRV/Bunny_1280x640_3M.yaml:3-20


def gen_fullsize_hadamard(name, arch, ckpt_name):
    """Round 4: the Hadamard path at FULL size against the REAL reference.  `arch` 'nerv': BASELINE
    configs[2] = NeRV Bunny_1280x640_3M with --hadamard on the checkpoint fixture nerv3m_bunny8real_f16.npz (this repo's
    trainer on one MI355X, the eight real Bunny crops, FP 33.4 dB; reference log: FP 33.25 dB); 'hnerv': HNeRV-3M with
    --hadamard on hnerv3m_bunny8real_f16.npz (FP 38.07 dB).  Frames = bunny8_640x1280.npz (the reference's center crop),
    --precision 6 5 4 5 5 6 6, batch 2, iters_w = 50 -> 48 phase-2 iterations of the reference's own model_reconstruction
    (quant_layer.py:44-49, 70-71: pad C_in to 2^k, transform, quantise in the transform domain, transform back, slice;
    calib_model.py:170-191: alpha on every one of the C_pad coefficients).  Stored: batch order, 48-entry loss log, PSNRs
    (FP / w/o opt / w/ opt), the initial scales on the padded transform-domain weights, the final hard-rounding masks.
    The butterflies run in _ref_stubs._fwht_normalized (pip hadamard_transform is absent): everything AROUND the transform
    is the reference's, its fp32 summation order is the stand-in's ("parity unpinned" at that one boundary)."""
    ck = np.load(os.path.join(HERE, ckpt_name))
    frames = torch.from_numpy(np.load(os.path.join(HERE, "bunny8_640x1280.npz"))["frames"].copy()).float() / 255.0
    n, B, iters = frames.shape[0], 2, 50
    torch.manual_seed(1)
    model = HNeRV(HNERV_3M) if arch == "hnerv" else NeRV(NERV_3M)
    sd = {k[3:].replace("/", "."): torch.from_numpy(ck[k].astype(np.float32)) for k in ck.files if k.startswith("sd:")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("encoder") for m in missing), (missing, unexpected)
    model.eval()
    emb = torch.from_numpy(ck["emb"].astype(np.float32))
    out = {}
    with torch.no_grad():
        y_fp = torch.cat([model.decode(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_fp"] = npy(psnr_frames(y_fp, frames))
    print("  FP psnr", out["psnr_fp"].mean(), flush=True)
    qnn = QuantModel(model=model, hadamard=True, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    out["avgbits"] = np.array(qnn.set_bitwidth(BITS), dtype=np.float64)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:B])
        y_q0 = torch.cat([qnn(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_q_noopt"] = npy(psnr_frames(y_q0, frames))
    print("  quant w/o opt psnr", out["psnr_q_noopt"].mean(), flush=True)
    qms = [m for m in qnn.model.modules() if isinstance(m, QuantModule)]
    for li, m in enumerate(qms):
        out[f"init_wdelta{li}"] = npy(m.weight_quantizer.delta)
        out[f"init_wzp{li}"] = npy(m.weight_quantizer.zero_point)
        out[f"init_bdelta{li}"] = npy(m.bias_quantizer.delta)
        out[f"cpad{li}"] = np.array(m.hadamard_weight.shape[1])
    g = torch.Generator().manual_seed(903)
    n_ep = iters // (n // B)
    order = torch.stack([torch.randperm(n, generator=g)[: (n // B) * B].view(n // B, B) for _ in range(n_ep)]).numpy()
    out["order"] = order
    loader = ReplayLoader(frames, order, n)
    log = []
    orig_call = ref_calib.LossFunction.__call__

    def recording_call(self, pred, tgt, grad=None):
        total = orig_call(self, pred, tgt, grad)
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round == "none":
            b = 0
        log.append((float(total), float(self.round_loss), float(b), self.count))
        return total

    ref_calib.LossFunction.__call__ = recording_call
    t0 = time.time()
    try:
        ref_calib.model_reconstruction(qnn, cali_data=emb, gt=loader, arch=arch, batch_size=B, iters=iters, weight=0.01,
                                       opt_mode="mse", hadamard=True, b_range=(20, 2), warmup=0.2, p=2.0, lr=0.003)
    finally:
        ref_calib.LossFunction.__call__ = orig_call
    print(f"  model_reconstruction: {len(log)} iterations in {time.time() - t0:.1f}s", flush=True)
    out["loss_log"] = np.array(log, dtype=np.float64)
    out["iters"] = np.array(iters)
    out["seconds_cpu8"] = np.array(time.time() - t0)
    for li, m in enumerate(qms):   # final hard decisions, bit-packed (alpha >= 0 on all C_pad coefficients), and final scales
        out[f"mask{li}"] = np.packbits((m.weight_quantizer.alpha.detach() >= 0).numpy().reshape(-1))
        out[f"final_wdelta{li}"] = npy(m.weight_quantizer.delta)
    qnn.set_quant_state(True)
    with torch.no_grad():
        y_q1 = torch.cat([qnn(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_q_opt"] = npy(psnr_frames(y_q1, frames))
    print("  quant w/ opt psnr", out["psnr_q_opt"].mean(), flush=True)
    save(name, **out)


def gen_config1_long(iters=3000):
    """Round 4: the REAL reference over thousands of iterations at full size, phase 1 included (the
    full-size reference evidence stopped at 48 phase-2 iterations).  HNeRV Bunny_1280x640_3M, checkpoint
    hnerv3m_bunny8real_f16.npz (FP 38.07 dB), the eight real 640x1280 crops, --precision 6 5 4 5 5 6 6, batch 2, iters_w = 3000
    -> int(0.05*3000/4) = 37 phase-1 epochs (148 iterations) + 713 phase-2 epochs (2852 iterations) of the reference's own
    model_reconstruction.  Stored: batch order, every iteration's (total, round, b, count), PSNRs FP / w/o opt / w/ opt, the
    scales after phase 1 (= final: phase 2 does not step them) and the final hard-rounding masks.  ~2 h on 6 threads."""
    torch.set_num_threads(6)
    ck = np.load(os.path.join(HERE, "hnerv3m_bunny8real_f16.npz"))
    frames = torch.from_numpy(np.load(os.path.join(HERE, "bunny8_640x1280.npz"))["frames"].copy()).float() / 255.0
    n, B = frames.shape[0], 2
    torch.manual_seed(1)
    model = HNeRV(HNERV_3M)
    sd = {k[3:].replace("/", "."): torch.from_numpy(ck[k].astype(np.float32)) for k in ck.files if k.startswith("sd:")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("encoder") for m in missing), (missing, unexpected)
    model.eval()
    emb = torch.from_numpy(ck["emb"].astype(np.float32))
    out = {}
    with torch.no_grad():
        y_fp = torch.cat([model.decode(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_fp"] = npy(psnr_frames(y_fp, frames))
    qnn = QuantModel(model=model, hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    out["avgbits"] = np.array(qnn.set_bitwidth(BITS), dtype=np.float64)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:B])
        y_q0 = torch.cat([qnn(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_q_noopt"] = npy(psnr_frames(y_q0, frames))
    print("  FP / w/o opt psnr", out["psnr_fp"].mean(), out["psnr_q_noopt"].mean(), flush=True)
    g = torch.Generator().manual_seed(903)
    n_ep = iters // (n // B)
    order = torch.stack([torch.randperm(n, generator=g)[: (n // B) * B].view(n // B, B) for _ in range(n_ep)]).numpy()
    out["order"] = order.astype(np.int16)
    loader = ReplayLoader(frames, order, n)
    log = []
    orig_call = ref_calib.LossFunction.__call__

    def recording_call(self, pred, tgt, grad=None):
        total = orig_call(self, pred, tgt, grad)
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round == "none":
            b = 0
        log.append((float(total), float(self.round_loss), float(b), self.count))
        if len(log) % 100 == 0:
            print(f"    iteration {len(log)}: total {log[-1][0]:.5f}, {time.time() - t0:.0f}s", flush=True)
        return total

    ref_calib.LossFunction.__call__ = recording_call
    t0 = time.time()
    try:
        ref_calib.model_reconstruction(qnn, cali_data=emb, gt=loader, arch="hnerv", batch_size=B, iters=iters, weight=0.01,
                                       opt_mode="mse", hadamard=False, b_range=(20, 2), warmup=0.2, p=2.0, lr=0.003)
    finally:
        ref_calib.LossFunction.__call__ = orig_call
    print(f"  model_reconstruction: {len(log)} iterations in {time.time() - t0:.1f}s", flush=True)
    out["loss_log"] = np.array(log, dtype=np.float64)
    out["iters"] = np.array(iters)
    out["seconds_cpu6"] = np.array(time.time() - t0)
    qms = [m for m in qnn.model.modules() if isinstance(m, QuantModule)]
    for li, m in enumerate(qms):
        out[f"mask{li}"] = np.packbits((m.weight_quantizer.alpha.detach() >= 0).numpy().reshape(-1))
        out[f"final_wdelta{li}"] = npy(m.weight_quantizer.delta)
        out[f"final_bdelta{li}"] = npy(m.bias_quantizer.delta)
    qnn.set_quant_state(True)
    with torch.no_grad():
        y_q1 = torch.cat([qnn(emb[i:i + 1])[0] for i in range(n)], 0)
    out["psnr_q_opt"] = npy(psnr_frames(y_q1, frames))
    print("  quant w/ opt psnr", out["psnr_q_opt"].mean(), flush=True)
    save("config1_hnerv3m_long.npz", **out)


GENS = {
    "uaq": gen_uaq,
    "adaround": gen_adaround,
    "zero_channel": gen_zero_channel,
    "roundloss": gen_roundloss,
    "tempdecay": gen_tempdecay,
    "quantmodule": gen_quantmodule,
    "decode": gen_decode,
    "frames": gen_frames,
    "bunny8_real": gen_bunny8_real,
    "traj_hnerv": lambda: gen_traj("traj_hnerv.npz", "hnerv", HNeRV, TINY_HNERV, False, 400, 150, 2e-3, 903),
    "traj_nerv_had": lambda: gen_traj("traj_nerv_had.npz", "nerv", NeRV, TINY_NERV, True, 200, 150, 2e-3, 904),
    "omega": gen_omega,
    "config1": gen_config1,
    "config1_long": gen_config1_long,
    "config2": lambda: gen_fullsize_hadamard("config2_nerv3m_hadamard.npz", "nerv", "nerv3m_bunny8real_f16.npz"),
    "config1_hadamard": lambda: gen_fullsize_hadamard("config1_hnerv3m_hadamard.npz", "hnerv", "hnerv3m_bunny8real_f16.npz"),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    for k, fn in GENS.items():
        if a.only and k not in a.only:
            continue
        print(f"== {k}")
        fn()
