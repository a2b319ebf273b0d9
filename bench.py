#!/usr/bin/env python3
"""bench.py -- calibration iterations/s of the network-wise calibration hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one phase-2 calibration iteration (AdaRound alpha + regulariser, the 95 % of a 21k run) of HNeRV
Bunny_1280x640_3M: uint8 frame gather, fake-quant of all 7 layers, decoder forward, L2 loss, full backward,
d(alpha) + regulariser gradient, Adam.  Per-GPU batch = 2 frames (BASELINE config 1); with N ranks the global batch
is 2N frames sharded by rank with one RCCL all-reduce over the conv weight gradients (config 3 at N=8) -> weak
scaling, value = B=2-equivalent iterations/s summed over ranks.  Inputs are synthetic and resident in HBM before the
timed region; weights are seeded random-init (iteration cost is value-independent).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside every 5th timed step) and
`cpu_baseline` (the oracle = CPU restatement of the reference path, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HNERV_3M = dict(crop_h=640, crop_w=1280, diff_enc=False, stage_block=1, enc_strides=[5, 4, 4, 2, 2],
                enc_channel=[64, 64, 64, 64, 16], channel_reduce=1.2, channel_lbound=12, dec_in_channel=92,
                dec_kernels=[1, 3, 5, 5, 5], dec_strides=[5, 4, 4, 2, 2], dec_norm="none", dec_acts="gelu",
                out_bias="tanh")
BITS = [6, 5, 4, 5, 5, 6, 6]
# --workload nerv: BASELINE configs[2] (NeRV Bunny_1280x640_3M + Hadamard); not the headline line, a secondary check
NERV_3M = dict(crop_h=640, crop_w=1280, diff_enc=False, base=1.25, level=80, channel_reduce=2, channel_lbound=24,
               dec_in_channel=145, dec_kernels=[3, 3, 3, 3, 3], dec_strides=[5, 4, 4, 2, 2], dec_norm="none",
               dec_acts="gelu", out_bias="tanh")
# /opt/skills/guides/MI355X_MICROARCH.md, Matrix cores: fp32-input MFMA 157.3 TF; dense BF16 MFMA ~2.5 PF.  The bf16x3
# kernels spend three BF16 MFMAs per fp32-equivalent product (hi*hi + hi*lo + lo*hi), so their ceiling in ALGORITHMIC
# (fp32-equivalent) FLOP/s is 2500/3.
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
FLAGS = dict(weight=0.01, b_range=(20, 2), lr=0.003)


def build_model(seed=903, workload="hnerv"):
    from neuroquant_amd.models import HNeRV, NeRV
    torch.manual_seed(seed)
    model = HNeRV(HNERV_3M) if workload == "hnerv" else NeRV(NERV_3M)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():  # decoder: variance-preserving init so activations / gradients have trained-model scale
        for name, p in model.named_parameters():
            if name.startswith("encoder"):
                continue
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / p[0].numel()) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return model.eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=132)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--workload", choices=("hnerv", "nerv"), default="hnerv",
                    help="hnerv = the headline config (default); nerv = NeRV-3M + Hadamard (BASELINE configs[2]), no cpu baseline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("NQ_DP_REHEARSAL"))
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from neuroquant_amd import ops
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache, synthetic_frames

    B = 2                       # frames per GPU per iteration (BASELINE config 1)
    gB = B * world
    K, W = args.steps, args.warmup
    n_frames = max(args.frames // gB * gB, gB)

    # ---- workload, resident in HBM ----
    nerv = args.workload == "nerv"
    model = build_model(workload=args.workload)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("encoder")}
    model = model.to(dev)
    frames_u8 = synthetic_frames(n_frames, 640, 1280, seed=903, device=dev)
    cache = FrameCache(frames_u8)
    with torch.no_grad():
        if nerv:
            emb = model.encode(torch.arange(n_frames, device=dev).float() / n_frames)
        else:
            emb = torch.cat([model.encode(cache.batch(torch.arange(i, min(i + 4, n_frames), device=dev)))
                             for i in range(0, n_frames, 4)])
    qnn = QuantModel(model, hadamard=nerv, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    avg_bits = qnn.set_bitwidth(BITS)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:B])            # lazy scale init (calibrate_network.py:235-238)

    steps_total = W + K
    # one "epoch" of W+K+1 batches (shuffled passes over the frames chained): iters = len(loader) gives
    # int(0.05*iters/len) = 0 phase-1 epochs and exactly one phase-2 epoch, for any K
    loader = CacheLoader(cache, list(range(n_frames)), gB, seed=903, rank=rank, world=world, epoch_batches=steps_total + 1)
    iters = len(loader)

    t = {}

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    PROF_EVERY = 5   # HIP events around the conv launches of every 5th timed step (the records themselves cost time)

    def hook(done):
        if W < done < steps_total:
            ops.profile_sample((done - W) % PROF_EVERY == 0)
        if done == W:
            sync()
            if not os.environ.get("NQ_BENCH_NOPROF"):
                ops.profile_start()
            t["t0"] = time.perf_counter()
        elif done == steps_total:
            t["t_enq"] = time.perf_counter()   # host finished enqueueing the timed steps (before the device drains)
            sync()
            t["t1"] = time.perf_counter()
            t["prof"] = ops.profile_stop() if not os.environ.get("NQ_BENCH_NOPROF") else {}

    model_reconstruction(qnn, cali_data=emb, gt=loader, arch=args.workload, batch_size=gB, iters=iters, hadamard=nerv,
                         warmup=0.0, max_steps=steps_total, step_hook=hook, **FLAGS)
    if "t1" not in t:
        hook(steps_total)
    elapsed = t["t1"] - t["t0"]
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded around every conv launch of the timed steps) ----
    prof = t["prof"]
    rows = []
    for key, (cnt, ms) in prof.items():
        kind, k, cin, cout, H, Wd, Bk, epi = key
        flops = 2.0 * Bk * cout * cin * k * k * H * Wd
        rows.append(dict(kernel=kind, k=k, cin=cin, cout=cout, H=H, W=Wd, launches=cnt, avg_ms=ms / cnt, total_ms=ms,
                         gflop_per_launch=flops / 1e9, tflops=flops / (ms / cnt * 1e-3) / 1e12))
    rows.sort(key=lambda r: -r["total_ms"])
    conv_ms = sum(r["total_ms"] for r in rows)
    conv_flops = sum(r["gflop_per_launch"] * r["launches"] for r in rows) * 1e9
    if not rows:   # NQ_BENCH_NOPROF=1: throughput only (measures the cost of the event records themselves)
        print(json.dumps({"value": round(K * world / elapsed, 3), "ms_per_step": round(elapsed / K * 1e3, 3), "note": "no profiling"}))
        return
    dom = rows[0]
    prof_steps = len([d for d in range(W, steps_total) if (d - W) % PROF_EVERY == 0])
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f'{dom["kernel"]}_k{dom["k"]}_{dom["cin"]}_{dom["cout"]}', {}).get("bytes")
        except Exception:
            traffic = None
    is3 = dom["kernel"].endswith("3")
    peak = PEAK_BF16X3_TFLOPS if is3 else PEAK_F32_MFMA_TFLOPS
    roofline = dict(bound="mfma", kernel=f'{dom["kernel"]} k{dom["k"]} {dom["cin"]}->{dom["cout"]} {dom["H"]}x{dom["W"]}',
                    achieved=round(dom["tflops"], 2), peak=round(peak, 1), unit="TFLOP/s",
                    frac=round(dom["tflops"] / peak, 4), traffic=traffic,
                    peak_basis=("dense BF16 MFMA 2500 TF / 3 products per fp32-equivalent FLOP (bf16x3)" if is3
                                else "fp32-input MFMA 157.3 TF"),
                    avg_launch_ms=round(dom["avg_ms"], 4), gflop_per_launch=round(dom["gflop_per_launch"], 2),
                    all_conv_tflops=round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                    conv_share_of_step=round(conv_ms / prof_steps / (elapsed / K * 1e3), 3), profiled_steps=prof_steps,
                    host_enqueue_ms_per_step=round((t["t_enq"] - t["t0"]) / K * 1e3, 3))
    try:  # per-kernel table for DESIGN.md / profiles (scratch; not part of the contract line)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_n{world}.json"), "w") as f:
            json.dump(rows, f, indent=1)
    except OSError:
        pass

    # ---- CPU baseline: the oracle (restatement of the reference's PyTorch-CPU path) on this box's host cores ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline and not nerv:
        cpu = cpu_baseline(sd_cpu, frames_u8[:8], emb[:8], args.cpu_iters)

    value = K * world / elapsed
    out = {
        "metric": "calibration iters/sec (HNeRV Bunny 1280x640, 21k iters) + final PSNR vs ref",
        "value": round(value, 3), "unit": "it/s (B=2 frames per iteration-equivalent, summed over GPUs)",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 operands split into bf16 hi+lo, 3 BF16 MFMAs per product, fp32 accumulate (bf16x3); fp32 MFMA on small layers",
        "data": "synthetic",
        "config": {"workload": ("NeRV Bunny_1280x640_3M + Hadamard" if nerv else "HNeRV Bunny_1280x640_3M")
                   + ", channel_wise, bits 6 5 4 5 5 6 6, phase-2 (AdaRound) iteration",
                   "per_gpu_batch": B, "global_batch": gB, "frames": n_frames, "avg_bits": avg_bits,
                   "parallelism": f"dp{world}"},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(sd, frames_u8, emb, n_iters):
    """Time the oracle's phase-2 iteration on the same HNeRV-3M weights / first 8 frames, all host cores."""
    import numpy as np
    from oracle import nq_oracle as O
    # the box's CPU share for one GPU is 16 cores (os.cpu_count() reports the whole 256-thread host; oversubscribing
    # it made the oracle 10x slower), so use the affinity mask capped at 16
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    dec = O.Decoder.from_state_dict(sd, "hnerv", HNERV_3M["dec_strides"])
    qs = O.QuantStack(dec, BITS, hadamard=False)
    frames = frames_u8.cpu().float() / 255.0
    order = np.array([[[0, 1], [2, 3], [4, 5], [6, 7]]] * 4)
    # iters=50 -> 0 phase-1 epochs, phase-2 iterations only (BASELINE config 0 shape); 1 untimed + n timed
    t0 = time.perf_counter()
    O.calibrate(qs, emb.cpu(), frames, order, 50, warmup=0.0, max_steps=1, **FLAGS)
    t1 = time.perf_counter()
    qs2 = O.QuantStack(O.Decoder.from_state_dict(sd, "hnerv", HNERV_3M["dec_strides"]), BITS, hadamard=False)
    t2 = time.perf_counter()
    O.calibrate(qs2, emb.cpu(), frames, order, 50, warmup=0.0, max_steps=1 + n_iters, **FLAGS)
    t3 = time.perf_counter()
    per_iter = ((t3 - t2) - (t1 - t0)) / n_iters
    return {"value": round(1.0 / per_iter, 4), "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"{n_iters} phase-2 iterations (B=2) of the same HNeRV-3M workload after 1 untimed, oracle on "
                      f"torch-CPU with {cores} threads, {per_iter:.2f} s/iter"}


if __name__ == "__main__":
    main()
