#!/usr/bin/env python3
"""Precision / PSNR gate at the headline size (HNeRV Bunny_1280x640_3M, bits 6 5 4 5 5 6 6, B = 2).

BASELINE.json's metric is "calibration iters/sec + final PSNR vs ref"; the default convolution path is bf16x3 (fp32
operands split into bf16 hi+lo on the BF16 matrix pipe), so the throughput number is only worth something if that path
ends at the same PSNR as exact fp32.  This tool establishes it on a TRAINED 3M model ("sensitivity grows
with PSNR ... repeat on the real model"):

  1. frames: the 8 Bunny-derived fixture frames (tests/golden/frames_320x640.npz) upsampled 2x (nearest) to 640x1280
     -- real video content, bit-identical on every machine -- or `--frames synthetic` (SURVEY §8d frames);
  2. FP32 fit of HNeRV-3M (encoder + decoder) with the repo's own trainer path (fused HIP decoder, torch Adam) to a real
     operating point (>= 30 dB);
  3. the SAME calibration (same recorded batch order; phase 1 = scales AND phase 2 = AdaRound; reference flow
     methods/calibrate_network.py:229-298, quantization/calib_model.py:134-226) under exact-fp32 MFMA, under exact fp32
     with the two frames of every batch swapped (same mathematics, different fp32 summation order: what fp32 does to
     ITSELF) and under bf16x3, for several batch orders -> the mean final PSNRs must agree within 0.02 dB (north-star
     bar) and no bf16x3 run may sit further from its fp32 twin than max(0.02 dB, 2x the fp32 self-spread);
  4. GPU exact-fp32 vs the CPU oracle for `--oracle-iters` iterations (chosen so that int(0.05*iters/len(gt)) >= 1, i.e.
     phase 1 runs) on the same checkpoint / frames / order -> per-iteration loss agreement and final PSNR.

    python tools/precision_gate.py --train-steps 3000 --iters 2000 --oracle-iters 200 --out gpurun_out/gate.json
    python tools/precision_gate.py --iters 21000 --frames-n 8 ...        (the full-length schedule)
"""
import argparse
import copy
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import BITS, HNERV_3M, NERV_3M  # noqa: E402

FLAGS = dict(weight=0.01, b_range=(20, 2), warmup=0.2, lr=0.003)


def bunny_frames_640(dev, n=8):
    """uint8 (n,3,640,1280): the committed 320x640 Bunny-derived frames, every pixel repeated 2x2 (integer-exact)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "frames_320x640.npz"))["frames"][:n]
    f = torch.from_numpy(z.copy())
    return f.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3).contiguous().to(dev)


def bunny_real_640(dev, n=8):
    """uint8 (n,3,640,1280): the first frames of the Bunny sequence, center-cropped as the reference's loader does
    (videosets/datasets.py:19-28; fixture tests/golden/bunny8_640x1280.npz) -- the reference's own operating point."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "bunny8_640x1280.npz"))["frames"][:n]
    return torch.from_numpy(z.copy()).contiguous().to(dev)


def make_order(n, B, iters, seed=903):
    """(epochs, batches, B) frame indices: one seeded permutation per epoch (shuffle=True, drop_last=True loader,
    reference calibrate_network.py:162-165), recorded so that every engine replays the same batches."""
    g = torch.Generator().manual_seed(seed)
    n_ep = max(int(iters / (n // B)), 1)
    return torch.stack([torch.randperm(n, generator=g)[: (n // B) * B].view(n // B, B) for _ in range(n_ep)]).numpy()


def load_fixture_checkpoint(name, dev, arch="hnerv"):
    """(model, emb, fp_psnr) from a committed decoder checkpoint fixture (tests/golden/<name>: fp16-grid decoder weights +
    embeddings, made by tests/golden/make_ckpt_fixture.py); the encoder stays at its initial values -- calibration and
    evaluation only consume the decoder and the embeddings."""
    from neuroquant_amd.models import HNeRV, NeRV
    z = np.load(os.path.join(ROOT, "tests", "golden", name))
    torch.manual_seed(903)
    model = HNeRV(HNERV_3M) if arch == "hnerv" else NeRV(NERV_3M)
    sd = {k[3:].replace("/", "."): torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("sd:")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("encoder") for m in missing), (missing, unexpected)
    return model.to(dev).eval(), torch.from_numpy(z["emb"].astype(np.float32)).to(dev), float(z["fp_psnr_trainer"])


def train_checkpoint(frames_u8, steps, dev, seed=903, lr=1e-3, log=print, arch="hnerv"):
    """FP32 fit of HNeRV-3M (arch "nerv": NeRV-3M, input = frame index / n) on `frames_u8` through the repo's trainer path
    (methods/regress.py: fused HIP decoder node, encoder + Adam in PyTorch).  -> (model in eval mode on dev, embeddings
    (n,16,2,4) / (n,160,1,1), FP PSNR)."""
    from neuroquant_amd import ops
    from neuroquant_amd.models import HNeRV, NeRV
    from neuroquant_amd.utils import CacheLoader, FrameCache
    torch.manual_seed(seed)
    model = (HNeRV(HNERV_3M) if arch == "hnerv" else NeRV(NERV_3M)).to(dev)
    cache = FrameCache(frames_u8)
    n = len(cache)
    B = 2
    per_epoch = n // B
    loader = CacheLoader(cache, list(range(n)), B, seed=seed, epoch_batches=per_epoch)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    model.train()
    t0, it = time.time(), 0
    while it < steps:
        for sample in loader:
            if it >= steps:
                break
            cur = lr * it / (0.1 * steps) + 1e-6 if it < 0.1 * steps else lr * 0.5 * (1 + math.cos(math.pi * (it - 0.1 * steps) / (0.9 * steps)))
            for gr in opt.param_groups:
                gr["lr"] = cur
            img = sample["img"]
            out, _, _ = model(img if arch == "hnerv" else sample["idx"].to(dev).float() / n)
            loss = ops.l2_loss(out, img) / img.shape[1]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            it += 1
            if it % 2000 == 0:
                log(f"  fit step {it}: batch PSNR {-10 * math.log10(float(loss) + 1e-12):.2f} dB, {time.time() - t0:.0f}s")
    torch.cuda.synchronize()
    model.eval()
    with torch.no_grad():
        idx = torch.arange(n, device=dev)
        emb = torch.cat([model.encode(cache.batch(idx[i:i + 1])) for i in range(n)]) if arch == "hnerv" \
            else model.encode(idx.float() / n)
        psnr = float(torch.cat([ops.frame_psnr(model.decode(emb[i:i + 1])[0], cache.batch(idx[i:i + 1]))
                                for i in range(n)]).mean())
    log(f"FP32 fit: {steps} steps in {time.time() - t0:.1f}s, FP PSNR {psnr:.3f} dB")
    return model, emb, psnr


def eval_psnr(qnn, emb, frames, precision="fp32"):
    """mean per-frame PSNR of the (quantised) model, decoded with the given convolution precision."""
    from neuroquant_amd import ops
    prev = ops._PRECISION
    ops.set_conv_precision(precision)
    try:
        with torch.no_grad():
            return float(torch.cat([ops.frame_psnr(qnn(emb[i:i + 1])[0], frames[i:i + 1])
                                    for i in range(emb.shape[0])]).double().mean())
    finally:
        ops.set_conv_precision(prev)


def calibrate_gpu(model, frames_u8, emb, order, iters, precision, record=True, flags=FLAGS, bits=BITS, arch="hnerv"):
    """One calibration of a deep copy of `model` under `precision`.  -> dict(psnr..., log (iters,4) or None, qnn)."""
    from neuroquant_amd import ops
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache
    frames = frames_u8.float() / 255.0
    ops.set_conv_precision(precision)
    try:
        had = arch == "nerv"   # the reference calibrates NeRV with --hadamard (BASELINE configs[2])
        qnn = QuantModel(copy.deepcopy(model), hadamard=had,
                         weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
        qnn.set_bitwidth(bits)
        qnn.eval()
        qnn.set_quant_state(True)
        with torch.no_grad():
            qnn(emb[:2])       # lazy scale init (calibrate_network.py:235-238)
        res = {"q_noopt": eval_psnr(qnn, emb, frames)}
        rec = [] if record else None
        loader = CacheLoader(FrameCache(frames_u8), list(range(frames_u8.shape[0])), order.shape[2], order=order)
        torch.cuda.synchronize()
        t0 = time.time()
        model_reconstruction(qnn, cali_data=emb, gt=loader, arch=arch, batch_size=order.shape[2], iters=iters,
                             hadamard=had, recorder=rec, **flags)
        torch.cuda.synchronize()
        res["seconds"] = time.time() - t0
        qnn.set_quant_state(True)
        res["q_opt"] = eval_psnr(qnn, emb, frames)                       # evaluated with exact-fp32 convolutions
        res["q_opt_eval_bf16x3"] = eval_psnr(qnn, emb, frames, "bf16x3")  # ... and with the default kernels
        res["iterations"] = len(rec) if rec is not None else None
        return res, (np.array(rec) if rec else None), qnn
    finally:
        ops.set_conv_precision(None)


def calibrate_oracle(sd, frames_u8, emb, order, iters, threads, flags=FLAGS, bits=BITS, on_step=None):
    from oracle import nq_oracle as O
    torch.set_num_threads(threads)
    dec = O.Decoder.from_state_dict({k: v for k, v in sd.items() if not k.startswith("encoder")}, "hnerv",
                                    HNERV_3M["dec_strides"])
    fr, em = frames_u8.cpu().float() / 255.0, emb.cpu()
    qs = O.QuantStack(dec, bits, hadamard=False)
    res = {}
    with torch.no_grad():
        res["q_noopt"] = float(O.psnr_per_frame(qs.forward(em), fr).double().mean())
    t0 = time.time()
    log = np.array(O.calibrate(qs, em, fr, order, iters, on_step=on_step, **flags))
    res["seconds"] = time.time() - t0
    with torch.no_grad():
        res["q_opt"] = float(O.psnr_per_frame(qs.forward(em), fr).double().mean())
    res["iterations"] = len(log)
    return res, log, qs


def mask_agreement(qnn_a, qnn_b):
    same = tot = 0
    for ma, mb in zip(qnn_a.quant_modules(), qnn_b.quant_modules()):
        same += int(((ma.weight_quantizer.alpha >= 0) == (mb.weight_quantizer.alpha >= 0)).sum())
        tot += ma.weight_quantizer.alpha.numel()
    return same / tot


def run(args, log=print):
    dev = torch.device("cuda", 0)
    if not hasattr(args, "arch"):   # callers that build their own namespace (tests/test_full_size.py)
        args.arch = "hnerv"
    n, B = args.frames_n, 2
    if args.frames == "bunny":
        frames_u8 = bunny_frames_640(dev, n)
    elif args.frames == "bunny_real":
        frames_u8 = bunny_real_640(dev, n)
    else:
        from neuroquant_amd.utils import synthetic_frames
        frames_u8 = synthetic_frames(n, 640, 1280, seed=903, device=dev)
    n = frames_u8.shape[0]
    if args.ckpt and args.ckpt.endswith(".npz"):
        model, emb, fp_psnr = load_fixture_checkpoint(os.path.basename(args.ckpt), dev, arch=args.arch)
        from neuroquant_amd import ops
        with torch.no_grad():   # what THIS decoder (fp16-grid weights) decodes to
            fp_psnr = float(torch.cat([ops.frame_psnr(model.decode(emb[i:i + 1])[0], frames_u8[i:i + 1].float() / 255.0)
                                       for i in range(n)]).double().mean())
        log(f"fixture checkpoint {args.ckpt}: FP PSNR {fp_psnr:.3f} dB")
    elif args.ckpt and os.path.exists(args.ckpt):
        from neuroquant_amd.models import HNeRV
        blob = torch.load(args.ckpt, map_location="cpu")
        model = HNeRV(HNERV_3M)
        model.load_state_dict(blob["sd"])
        model = model.to(dev).eval()
        emb, fp_psnr = blob["emb"].to(dev), blob["fp_psnr"]
    else:
        model, emb, fp_psnr = train_checkpoint(frames_u8, args.train_steps, dev, log=log, arch=args.arch)
        if args.save_ckpt:
            os.makedirs(os.path.dirname(os.path.abspath(args.save_ckpt)), exist_ok=True)
            torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "emb": emb.cpu(), "fp_psnr": fp_psnr},
                       args.save_ckpt)
    res = {"config": f"{'HNeRV' if args.arch == 'hnerv' else 'NeRV + Hadamard'} Bunny_1280x640_3M, {n} frames ({args.frames}), B={B}, bits {BITS}, iters_w={args.iters}",
           "fp_psnr": fp_psnr, "train_steps": args.train_steps}

    # ---- fp32 MFMA vs bf16x3, same recorded order, phase 1 + phase 2 ----
    # The calibration is chaotic at the bit level (tests/golden/make_sensitivity.py: the reference's own algorithm moves
    # its final PSNR by ~0.015 dB under a last-bit change of summation order on the tiny model), so "bf16x3 == fp32"
    # can only be judged against what exact fp32 does to ITSELF: every order is also run with the two frames of each
    # batch swapped -- the same mathematics (the loss is a mean over the batch), a different fp32 summation order in the
    # weight gradients and the loss reduction.  Several batch orders (seeds) give the spread of all three.
    runs = []
    keep = {}
    for sd_ in args.seeds:
        order = make_order(n, B, args.iters, seed=sd_)
        row = {"order_seed": sd_}
        swapped = np.ascontiguousarray(order[..., ::-1])
        for tag, prec, od in (("fp32", "fp32", order), ("fp32_swapped", "fp32", swapped),
                              ("bf16x3", "bf16x3", order), ("bf16x3_swapped", "bf16x3", swapped)):
            r, lg, qnn = calibrate_gpu(model, frames_u8, emb, od, args.iters, prec, record=args.record, arch=args.arch)
            row[tag] = r
            keep[tag] = (lg, qnn)
            log(f"seed {sd_} {tag}: {r['seconds']:.1f}s, PSNR w/o opt {r['q_noopt']:.4f} -> w/ opt {r['q_opt']:.4f} dB")
        row["mask_agreement_fp32_vs_bf16x3"] = mask_agreement(keep["fp32"][1], keep["bf16x3"][1])
        row["mask_agreement_fp32_vs_fp32_swapped"] = mask_agreement(keep["fp32"][1], keep["fp32_swapped"][1])
        if args.record:
            a, b = keep["fp32"][0], keep["bf16x3"][0]
            rel = np.abs(a[:, 0] - b[:, 0]) / np.abs(a[:, 0])
            ep1 = int(0.05 * args.iters / (n // B)) * (n // B)
            row["loss_rel_diff_fp32_vs_bf16x3"] = {"first3": float(rel[:3].max()), "phase1_max": float(rel[:max(ep1, 1)].max()),
                                                   "max": float(rel.max()), "median": float(np.median(rel)),
                                                   "phase1_iterations": ep1}
        runs.append(row)
        keep.clear()
    res["runs"] = runs
    summarise(res)
    log(f"fp32 vs itself (swapped batch halves): <= {res['fp32_self_spread_dB']:.4f} dB; bf16x3 vs fp32: <= "
        f"{res['dpsnr_fp32_vs_bf16x3_dB']:.4f} dB; means {res['mean_q_opt']['fp32']:.4f} / {res['mean_q_opt']['bf16x3']:.4f}")

    # ---- GPU exact fp32 vs the CPU oracle ----
    if args.oracle_iters:
        assert args.arch == "hnerv", "the oracle leg of this tool is wired for HNeRV (--oracle-iters 0 with --arch nerv)"
        it_o = args.oracle_iters
        assert int(0.05 * it_o / (n // B)) >= 1, "choose --oracle-iters so that phase 1 runs (int(0.05*iters/len(gt)) >= 1)"
        order_o = make_order(n, B, it_o, seed=904)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        g, glog, _ = calibrate_gpu(model, frames_u8, emb, order_o, it_o, "fp32", record=True)
        g3, glog3, _ = calibrate_gpu(model, frames_u8, emb, order_o, it_o, "bf16x3", record=True)
        c, clog, _ = calibrate_oracle(sd, frames_u8, emb, order_o, it_o, args.cpu_threads)
        rel = np.abs(glog[:, 0] - clog[:, 0]) / np.abs(clog[:, 0])
        rel3 = np.abs(glog3[:, 0] - clog[:, 0]) / np.abs(clog[:, 0])
        ep1 = int(0.05 * it_o / (n // B)) * (n // B)
        res["oracle"] = {"iters_w": it_o, "iterations": int(len(clog)), "phase1_iterations": ep1, "cpu": c, "gpu_fp32": g,
                         "gpu_bf16x3": g3,
                         "loss_rel_diff_fp32": {"first": float(rel[0]), "first3": float(rel[:3].max()),
                                                "phase1_max": float(rel[:ep1].max()), "max": float(rel.max())},
                         "loss_rel_diff_bf16x3": {"first": float(rel3[0]), "first3": float(rel3[:3].max()),
                                                  "phase1_max": float(rel3[:ep1].max()), "max": float(rel3.max())},
                         "dpsnr_fp32_dB": abs(g["q_opt"] - c["q_opt"]), "dpsnr_bf16x3_dB": abs(g3["q_opt"] - c["q_opt"]),
                         "cpu_threads": args.cpu_threads}
        log(f"oracle {it_o} iters: CPU {c['q_opt']:.4f} dB ({c['seconds']:.0f}s), GPU fp32 {g['q_opt']:.4f}, bf16x3 {g3['q_opt']:.4f}; "
            f"loss rel diff fp32 max {rel.max():.2e}, bf16x3 max {rel3.max():.2e}")
    return res


def summarise(res):
    """Population statistics over res["runs"] (also used to merge the per-seed records of several GPU calls)."""
    runs = res["runs"]
    res["fp32"], res["bf16x3"] = runs[0]["fp32"], runs[0]["bf16x3"]
    f32 = [r[t]["q_opt"] for r in runs for t in ("fp32", "fp32_swapped")]
    b3 = [r[t]["q_opt"] for r in runs for t in ("bf16x3", "bf16x3_swapped")]
    res["q_opt_fp32_runs"], res["q_opt_bf16x3_runs"] = f32, b3
    res["fp32_self_spread_dB"] = max(abs(r["fp32"]["q_opt"] - r["fp32_swapped"]["q_opt"]) for r in runs)
    res["bf16x3_self_spread_dB"] = max(abs(r["bf16x3"]["q_opt"] - r["bf16x3_swapped"]["q_opt"]) for r in runs)
    res["dpsnr_fp32_vs_bf16x3_dB"] = max(abs(r["fp32"]["q_opt"] - r["bf16x3"]["q_opt"]) for r in runs)
    res["mean_q_opt"] = {"fp32": float(np.mean(f32)), "bf16x3": float(np.mean(b3))}
    res["dmean_dB"] = abs(res["mean_q_opt"]["fp32"] - res["mean_q_opt"]["bf16x3"])
    # Welch statistic of the two populations (same-size samples of the two precisions)
    se = math.sqrt(np.var(f32, ddof=1) / len(f32) + np.var(b3, ddof=1) / len(b3)) if len(f32) > 1 else float("nan")
    res["welch_t"] = float((np.mean(b3) - np.mean(f32)) / se) if se and se == se else None
    # the round-3 bar: |mean difference| < 0.02 dB AND every bf16x3 run inside the range the exact-fp32
    # runs span, widened by 0.02 dB
    lo, hi = min(f32), max(f32)
    res["fp32_range_dB"] = [lo, hi]
    res["bf16x3_outside_fp32_range_dB"] = max(max(lo - v, v - hi, 0.0) for v in b3)
    res["strict_ok"] = bool(res["dmean_dB"] < 0.02 and res["bf16x3_outside_fp32_range_dB"] <= 0.02)
    return res


# Spread of the final PSNR of EXACT fp32 under last-bit perturbations of its own arithmetic (batch halves swapped, another
# batch order, a re-ordered bias-gradient sum in another build) on the trained 3M model: 20 runs at iters_w = 2000 over
# three builds sit between 31.80 and 31.98 dB per checkpoint (population standard deviation 0.035-0.04 dB, pair
# differences up to 0.09 dB, profiles/r02_precision_gate_2000.json); at iters_w = 21000 the runs collapse to within
# 0.01 dB (profiles/r02_precision_gate_21000.json).  A two-run estimate of that spread is itself noisy (0.012 and 0.087 dB
# were both observed), so the yardstick has a floor per operating point.
SPREAD_FLOOR_DB = {"short": 0.08, "full": 0.02}


def gate_ok(res):
    """bf16x3 must be indistinguishable from exact fp32.  The yardstick is what exact fp32 does to ITSELF when only its
    summation order changes: S = max(measured fp32_self_spread_dB, the floor for the schedule length above) -- i.e. the
    north-star's 0.02 dB applies to the full-length schedule, where the algorithm's own run-to-run noise is below it, and
    a short schedule is judged against its (larger) noise.  The mean final PSNRs of the two precisions must agree within
    S and no bf16x3 run may sit further from its fp32 twin than 2 S."""
    full = "iters_w=2" in res["config"] and int(res["config"].rsplit("iters_w=", 1)[1]) >= 20000
    S = max(res["fp32_self_spread_dB"], SPREAD_FLOOR_DB["full" if full else "short"])
    return res["dmean_dB"] <= S and res["dpsnr_fp32_vs_bf16x3_dB"] <= 2 * S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", choices=("hnerv", "nerv"), default="hnerv", help="nerv: NeRV-3M calibrated with the Hadamard transform")
    ap.add_argument("--train-steps", type=int, default=3000)
    ap.add_argument("--iters", type=int, default=2000, help="iters_w of the fp32-vs-bf16x3 calibration (21000 = full length)")
    ap.add_argument("--oracle-iters", type=int, default=200, help="iters_w of the GPU-vs-CPU-oracle calibration; 0 = skip")
    ap.add_argument("--frames", choices=("bunny", "bunny_real", "synthetic"), default="bunny")
    ap.add_argument("--frames-n", type=int, default=8)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-record", dest="record", action="store_false", help="no per-iteration loss log (no host sync per step)")
    ap.add_argument("--seeds", type=int, nargs="+", default=[903, 904], help="batch-order seeds of the fp32-vs-bf16x3 runs")
    ap.add_argument("--ckpt", default=None, help="load this checkpoint instead of training")
    ap.add_argument("--save-ckpt", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--merge", nargs="+", default=None, help="merge the per-seed records of several runs of this tool "
                    "(same checkpoint / frames / iters) into one record with the population statistics; no GPU needed")
    args = ap.parse_args()
    if args.merge:
        parts = [json.load(open(p)) for p in args.merge]
        assert len({p["config"] for p in parts}) == 1 and len({round(p["fp_psnr"], 4) for p in parts}) == 1, "different runs"
        res = {k: parts[0][k] for k in ("config", "fp_psnr", "train_steps")}
        res["merged_from"] = [os.path.basename(p) for p in args.merge]
        res["runs"] = [r for p in parts for r in p["runs"]]
        assert len({r["order_seed"] for r in res["runs"]}) == len(res["runs"]), "a seed appears twice"
        summarise(res)
        txt = json.dumps(res, indent=1)
        print(txt)
        if args.out:
            open(args.out, "w").write(txt)
        sys.exit(0 if res["strict_ok"] else 1)
    res = run(args)
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt)
    ok = gate_ok(res) and res["fp_psnr"] >= 30.0
    if "oracle" in res:
        ok = ok and res["oracle"]["dpsnr_fp32_dB"] < 0.02 and res["oracle"]["dpsnr_bf16x3_dB"] < 0.02
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
