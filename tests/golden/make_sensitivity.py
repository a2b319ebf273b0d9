#!/usr/bin/env python3
"""Sensitivity of the golden calibration trajectories to the fp32 SUMMATION ORDER of the convolutions (CPU only).

Why: the reference's network-wise calibration is chaotic at the bit level -- phase 1 moves every scale by lr = 1e-3
(5-10 % of delta) per Adam step and `round()` flips follow (calib_model.py:134-165, quantizer.py:53-57), phase 2 snaps soft
roundings to hard ones.  An implementation with a different (equally valid) summation order inside the convolutions
therefore cannot follow the reference's per-iteration losses / final scales / rounding masks to fp32 precision for hundreds
of iterations; how far it drifts is a property of the ALGORITHM, measured here with the oracle itself: the same oracle
(which reproduces the reference golden to 2e-4 per iteration when run the same way) is re-run with its convolutions
perturbed at the last-bit level only:

    threads1    torch.set_num_threads(1): oneDNN partitions the reductions differently
    conv_f64    every convolution accumulated in float64, rounded to fp32 once (the "most exact" fp32 result)
    conv_perm   input channels of every convolution visited in a fixed random order (same sum, different order)

and the spread w.r.t. the reference's recorded trajectory is stored in traj_sensitivity.json.  The GPU trajectory tests
(tests/test_hip_parity.py) bound the HIP path by a stated multiple of this spread instead of by hand-loosened numbers.

    python3 tests/golden/make_sensitivity.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import BITS, T, load_npz, state_dict_from_npz  # noqa: E402
from oracle import nq_oracle as O  # noqa: E402


def conv_f64(x, W, b, **kw):
    return F.conv2d(x.double(), W.double(), b.double(), **kw).float()


def make_conv_perm(seed=5):
    perms = {}

    def conv_perm(x, W, b, **kw):
        c = x.shape[1]
        if c not in perms:
            perms[c] = torch.randperm(c, generator=torch.Generator().manual_seed(seed + c))
        p = perms[c]
        return F.conv2d(x[:, p].contiguous(), W[:, p].contiguous(), b, **kw)

    return conv_perm


def run_variant(name, arch, had, strides, fc_hw, z, frames, conv_fn=None, threads=8):
    torch.set_num_threads(threads)
    dec = O.Decoder.from_state_dict(state_dict_from_npz(z, "sd:"), arch, strides, fc_hw)
    if conv_fn is not None:
        dec.conv_fn = conv_fn
    qs = O.QuantStack(dec, BITS, hadamard=had)
    emb = T(z["emb"])
    log = np.array(O.calibrate(qs, emb, frames, z["order"], int(z["iters"])))
    ref = z["loss_log"]
    rel = np.abs(log[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
    n_b = z["order"].shape[1]
    ep1 = int(0.05 * int(z["iters"]) / n_b) * n_b
    with torch.no_grad():
        psnr = float(O.psnr_per_frame(qs.forward(emb), frames).mean())
    same = tot = 0
    d_rel = []
    for li, L in enumerate(qs.dec.layers):
        same += int(((L.wa >= 0).numpy() == (z[f"fin_walpha{li}"] >= 0)).sum())
        tot += L.wa.numel()
        d_rel.append(np.abs(L.wd.detach().numpy() - z[f"fin_wdelta{li}"]).reshape(-1) / np.abs(z[f"fin_wdelta{li}"]).reshape(-1))
    d_rel = np.concatenate(d_rel)
    out = dict(loss_rel_first3=float(rel[:3].max()), loss_rel_phase1=float(rel[:max(ep1, 1)].max()),
               loss_rel_iter20=float(rel[:20].max()), loss_rel_all=float(rel.max()), loss_rel_median=float(np.median(rel)),
               final_delta_rel_max=float(d_rel.max()), final_delta_rel_median=float(np.median(d_rel)),
               mask_agreement=same / tot, psnr=psnr, psnr_ref=float(z["psnr_q_opt"].mean()),
               dpsnr_dB=abs(psnr - float(z["psnr_q_opt"].mean())))
    print(name, json.dumps(out), flush=True)
    return out


def main():
    frames = T(load_npz("frames_320x640.npz")["frames"]).float() / 255.0
    res = {}
    for tag, fname, arch, had, fc_hw in (("hnerv", "traj_hnerv.npz", "hnerv", False, (1, 1)),
                                          ("nerv_had", "traj_nerv_had.npz", "nerv", True, (1, 2))):
        z = load_npz(fname)
        r = {}
        r["same_order"] = run_variant(f"{tag}/same_order", arch, had, [5, 4, 4, 2, 2], fc_hw, z, frames)
        r["threads1"] = run_variant(f"{tag}/threads1", arch, had, [5, 4, 4, 2, 2], fc_hw, z, frames, threads=1)
        r["conv_f64"] = run_variant(f"{tag}/conv_f64", arch, had, [5, 4, 4, 2, 2], fc_hw, z, frames, conv_fn=conv_f64)
        r["conv_perm"] = run_variant(f"{tag}/conv_perm", arch, had, [5, 4, 4, 2, 2], fc_hw, z, frames, conv_fn=make_conv_perm())
        pert = [r[k] for k in ("threads1", "conv_f64", "conv_perm")]
        r["spread"] = {k: (min(p[k] for p in pert) if k == "mask_agreement" else max(p[k] for p in pert))
                       for k in ("loss_rel_first3", "loss_rel_phase1", "loss_rel_iter20", "loss_rel_all", "final_delta_rel_max",
                                 "final_delta_rel_median", "mask_agreement", "dpsnr_dB")}
        res[tag] = r
    with open(os.path.join(HERE, "traj_sensitivity.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote traj_sensitivity.json")


if __name__ == "__main__":
    main()
