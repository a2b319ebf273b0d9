#!/usr/bin/env python3
"""bench.py -- calibration iterations/s of the network-wise calibration hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--global-batch 16]

A step = one phase-2 calibration iteration (AdaRound alpha + regulariser, the 95 % of a 21k run) of HNeRV
Bunny_1280x640_3M: uint8 frame gather, fake-quant of all 7 layers, decoder forward, L2 loss, full backward,
d(alpha) + regulariser gradient, Adam.  Inputs are synthetic and resident in HBM before the timed region; weights of the
timed (headline) runs are seeded random-init.  The iteration executes the same instructions whatever the values, but its
DURATION is not value-independent on this chip: the clock the matrix pipe holds depends on how many operand bits toggle
(MI355X_MICROARCH.md, DVFS), so the same kernels are also timed on real data -- the `trained` object below.

Scaling modes:
  default            weak: per-GPU batch = 2 frames (BASELINE configs[1]; global batch 2N = configs[3] at N = 8), one RCCL
                     all-reduce over the conv weight gradients; value = B=2-equivalent iterations/s summed over ranks.
  --global-batch G   strong: the global batch is fixed (configs[3]: G = 16), per-GPU batch = G/N; value = global-batch-G
                     iterations/s.

Prints ONE JSON line (rank 0) with
  `roofline`      dominant kernel of the timed (default-precision, bf16x3) run, HIP-event timed inside every 10th timed step (the 6th, 16th, ...); the
                  bound (mfma | hbm) is chosen per kernel from its arithmetic intensity against the ridge of its matrix pipe;
  `repeats`       it/s of `--repeats` independent timed runs of the same K steps (`value` = the first);
  `fp32`          the same workload re-timed with exact-fp32 MFMA convolutions (NQ_CONV_PRECISION=fp32) and its roofline;
  `phase1`        the same workload's PHASE-1 iteration (scale learning, 5 % of a run) timed the same way;
  `nerv`          BASELINE configs[2] (NeRV Bunny_1280x640_3M + Hadamard) timed the same way, with its own roofline object;
  `trained`       the headline iteration timed the same way on the COMMITTED trained checkpoint (38 dB) and the eight real
                  Bunny crops (tests/golden/hnerv3m_bunny8real_f16.npz, bunny8_640x1280.npz) instead of seeded weights and
                  synthetic frames: it/s and the dominant kernel's time on the real workload;
  `uvg`           BASELINE configs[4] shape (HNeRV UVG 960x1920, ~12 M decoder parameters, tools/hnerv_uvg_12m.yaml, synthetic
                  frames): phase-2 it/s with its own roofline object + the Omega bit-allocation sweep timed per candidate;
  `psnr`          BASELINE configs[0] (the first 8 Bunny frames at 640x1280, iters_w = 50) on the committed trained
                  checkpoint: final PSNR of the CPU oracle, the GPU with exact fp32 and with bf16x3 (bar: within 0.02 dB);
  `cpu_baseline`  the oracle (CPU restatement of the reference path) timed on this box's host cores on that same run
                  (48 iterations, first 4 discarded).
"""
import argparse
import gc
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
PEAK_HBM_BPS = 8.0e12
FLAGS = dict(weight=0.01, b_range=(20, 2), lr=0.003)
PROF_EVERY = 10  # HIP events around the conv launches of every 10th timed step: those steps are launched eagerly (~1.2 ms of
                 # host work each), all others are hipGraph replays
PROF_PHASE = 5   # ... the 6th, 16th, ... timed step: by then the host is several replays (>= 10 ms of GPU work) ahead of the
                 # device, so the eager enqueue of a profiled step is hidden even on a box with a slow host (profiling the
                 # FIRST timed step, right behind the synchronisation that opens the region, exposed it: one box with 9 ms
                 # per eager step read 340 instead of 450 it/s)


def uvg_12m_config():
    from neuroquant_amd.utils import get_config
    return get_config(os.path.join(ROOT, "tools", "hnerv_uvg_12m.yaml"))


def build_model(seed=903, workload="hnerv"):
    from neuroquant_amd.models import HNeRV, NeRV
    torch.manual_seed(seed)
    model = NeRV(NERV_3M) if workload == "nerv" else HNeRV(uvg_12m_config() if workload == "uvg12m" else HNERV_3M)
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


def roofline_of(prof, elapsed, K, t_enq, precision, prof_steps):
    """-> (roofline dict, per-kernel rows) from ops.profile_stop()'s {key: (launches, total_ms)}."""
    rows = []
    for key, (cnt, ms) in prof.items():
        kind, k, cin, cout, H, Wd, Bk, epi = key
        flops = 2.0 * Bk * cout * cin * k * k * H * Wd
        # algorithmic bytes of the launch (every tensor read / written the minimum number of times): forward /
        # data gradient read x, write y -- twice for the PixelShuffle+GELU epilogue (a and gelu'), plus one read of gelu'
        # for the data-gradient epilogue; the weight gradient reads x and dY.  Weights are negligible.
        px = 4.0 * Bk * H * Wd
        if kind.startswith("conv_wgrad"):
            nbytes = px * (cin + cout)
        else:
            nbytes = px * (cin + cout * (2 if epi == 1 else 1) + (cout if epi == 4 else 0))
        rows.append(dict(kernel=kind, k=k, cin=cin, cout=cout, H=H, W=Wd, B=Bk, epi=epi, launches=cnt, avg_ms=ms / cnt,
                         total_ms=ms, gflop_per_launch=flops / 1e9, tflops=flops / (ms / cnt * 1e-3) / 1e12,
                         mbytes_per_launch=nbytes / 1e6, gbps=nbytes / (ms / cnt * 1e-3) / 1e9))
    if not rows:
        return None, rows
    rows.sort(key=lambda r: -r["total_ms"])
    conv_ms = sum(r["total_ms"] for r in rows)
    conv_flops = sum(r["gflop_per_launch"] * r["launches"] for r in rows) * 1e9
    dom = rows[0]
    is3 = dom["kernel"].endswith("3")
    peak = PEAK_BF16X3_TFLOPS if is3 else PEAK_F32_MFMA_TFLOPS
    # HBM bytes per launch of that kernel: PMC passes of the SAME build (profiles/hbm_traffic.json, collected with
    # rocprofv3 --pmc as profiles/README.md describes).  Only reported while the profiled duration still matches the live
    # one (a stale table must not outlive a kernel change) and for the per-GPU batch it was taken at; else null.
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            ent = json.load(open(tpath)).get(f'{dom["kernel"]}_k{dom["k"]}_{dom["cin"]}_{dom["cout"]}')
            if ent and ent.get("batch", 2) == dom["B"] and abs(ent["dur_us"] * 1e-3 - dom["avg_ms"]) <= 0.2 * dom["avg_ms"]:
                traffic, traffic_src = ent.get("bytes"), ent.get("source")
        except Exception:
            traffic = None
    # which roof bounds the dominant kernel: arithmetic intensity (algorithmic flops / algorithmic bytes) against the
    # ridge of ITS matrix pipe (833 TF / 8 TB/s = 104 flop/B for bf16x3, 157.3 / 8 = 19.7 for fp32 MFMA)
    ai = dom["gflop_per_launch"] * 1e3 / dom["mbytes_per_launch"]
    ridge = peak * 1e12 / PEAK_HBM_BPS
    name = f'{dom["kernel"]} k{dom["k"]} {dom["cin"]}->{dom["cout"]} {dom["H"]}x{dom["W"]} B{dom["B"]}'
    basis = ("dense BF16 MFMA 2500 TF / 3 products per fp32-equivalent FLOP (bf16x3)" if is3 else "fp32-input MFMA 157.3 TF")
    if ai < ridge:
        head = dict(bound="hbm", kernel=name, achieved=round(dom["gbps"], 1), peak=PEAK_HBM_BPS / 1e9, unit="GB/s",
                    frac=round(dom["gbps"] * 1e9 / PEAK_HBM_BPS, 4), traffic=traffic, traffic_source=traffic_src,
                    peak_basis="HBM3E 8 TB/s (MI355X_MICROARCH.md; 6.29 TB/s measured copy)",
                    mfma_tflops=round(dom["tflops"], 2), mfma_frac=round(dom["tflops"] / peak, 4), mfma_peak_basis=basis)
    else:
        head = dict(bound="mfma", kernel=name, achieved=round(dom["tflops"], 2), peak=round(peak, 1), unit="TFLOP/s",
                    frac=round(dom["tflops"] / peak, 4), traffic=traffic, traffic_source=traffic_src, peak_basis=basis)
    roofline = dict(head, arithmetic_intensity=round(ai, 1), ridge=round(ridge, 1),
                    algorithmic_mbytes_per_launch=round(dom["mbytes_per_launch"], 1),
                    avg_launch_ms=round(dom["avg_ms"], 4), gflop_per_launch=round(dom["gflop_per_launch"], 2),
                    all_conv_tflops=round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                    conv_share_of_step=round(conv_ms / max(prof_steps, 1) / (elapsed / K * 1e3), 3),
                    profiled_steps=prof_steps, host_enqueue_ms_per_step=round(t_enq / K * 1e3, 3))
    return roofline, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=132)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: fixed global batch (BASELINE configs[3]: 16), per-GPU batch = G/N; default 0 = weak "
                         "scaling with 2 frames per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the psnr / cpu_baseline leg (CPU oracle, ~1 min)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the exact-fp32 re-timing")
    ap.add_argument("--fp32-steps", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=3, help="timed runs of K steps; value = the first, all go into `repeats`")
    ap.add_argument("--no-phase1", action="store_true", help="skip the phase-1 (scale learning) timing")
    ap.add_argument("--no-nerv", action="store_true", help="skip the NeRV-3M + Hadamard object (BASELINE configs[2])")
    ap.add_argument("--no-trained", action="store_true", help="skip the `trained` object (headline iteration on the committed "
                    "checkpoint and the real Bunny crops)")
    ap.add_argument("--no-uvg", action="store_true", help="skip the `uvg` object (BASELINE configs[4] shape + Omega sweep timing)")
    ap.add_argument("--uvg-steps", type=int, default=10)
    ap.add_argument("--precision", choices=("bf16x3", "fp32"), default="bf16x3",
                    help="convolution precision of the HEADLINE runs (default bf16x3; fp32 = exact fp32-input MFMA: used to "
                         "profile that path -- the default line already carries it as the `fp32` object)")
    ap.add_argument("--workload", choices=("hnerv", "nerv", "uvg12m"), default="hnerv",
                    help="hnerv = the headline config (default); nerv = NeRV-3M + Hadamard (BASELINE configs[2]); uvg12m = HNeRV UVG "
                         "960x1920 ~12M (BASELINE configs[4] shape) as the headline line of this run; neither has a cpu baseline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # before the first call that initialises HIP/HSA (the runtime reads its environment once): dmabuf IPC for RCCL
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("NQ_DP_REHEARSAL"))
    if use_dist:
        # NQ_DIST_BACKEND: "nccl" (= RCCL, one rank per GPU: the driver's launch) or "gloo" (rehearsals of the N > 1 command
        # with several ranks on ONE card -- RCCL refuses two ranks on a device; tests/test_dp_gpu.py)
        backend = os.environ.get("NQ_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from neuroquant_amd import ops
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache, synthetic_frames

    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} ranks")
        gB = args.global_batch
        B = gB // world
    else:
        B = 2                   # frames per GPU per iteration (BASELINE configs[1])
        gB = B * world
    K, W = args.steps, args.warmup
    n_frames = max(args.frames // gB * gB, gB)
    # ---- workloads, resident in HBM (built on first use) ----
    wl_box = {}

    def workload_of(name):
        """-> dict(arch, hadamard, cache, n, emb, make): frames (uint8 cache), decoder inputs and a factory of fresh models.
        hnerv / nerv: seeded weights + 132 synthetic 640x1280 frames (the headline); trained: the committed 38 dB HNeRV-3M
        checkpoint + the eight real Bunny crops; uvg12m: seeded HNeRV UVG-12M + 8 synthetic 960x1920 frames."""
        if name in wl_box:
            return wl_box[name]
        if name == "trained":
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import precision_gate as pg
            n = max(8 // gB * gB, gB)
            assert n <= 8, "the real-crop fixture holds 8 frames"
            cache = FrameCache(pg.bunny_real_640(dev, n))
            _, emb, _ = pg.load_fixture_checkpoint("hnerv3m_bunny8real_f16.npz", dev)
            wl = dict(arch="hnerv", hadamard=False, cache=cache, n=n, emb=emb[:n].contiguous(),
                      make=lambda: pg.load_fixture_checkpoint("hnerv3m_bunny8real_f16.npz", dev)[0])
        else:
            hw = (960, 1920) if name == "uvg12m" else (640, 1280)
            n = max(8 // gB * gB, gB) if name == "uvg12m" else n_frames
            key = ("frames", hw, n)
            if key not in wl_box:
                wl_box[key] = FrameCache(synthetic_frames(n, hw[0], hw[1], seed=903, device=dev))
            cache = wl_box[key]
            model = build_model(workload=name).to(dev)
            with torch.no_grad():
                if name == "nerv":
                    emb = model.encode(torch.arange(n, device=dev).float() / n)
                else:
                    emb = torch.cat([model.encode(cache.batch(torch.arange(i, min(i + 4, n), device=dev)))
                                     for i in range(0, n, 4)])
            del model
            wl = dict(arch="nerv" if name == "nerv" else "hnerv", hadamard=name == "nerv", cache=cache, n=n, emb=emb,
                      make=lambda: build_model(workload=name).to(dev))
        wl_box[name] = wl
        return wl

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(precision, K, W, workload=args.workload, phase1=False):
        """Fresh QuantModel on the seeded weights; W untimed + K timed phase-2 iterations under `precision` (phase1: the
        same count of PHASE-1 iterations -- scales learned through the UAQ fake-quant, reference calib_model.py:119-165)."""
        wl = workload_of(workload)
        nerv, arch, cache, emb = wl["hadamard"], wl["arch"], wl["cache"], wl["emb"]
        ops.set_conv_precision(precision)
        model = wl["make"]()
        qnn = QuantModel(model, hadamard=nerv, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
        avg_bits = qnn.set_bitwidth(BITS)
        qnn.eval()
        qnn.set_quant_state(True)
        with torch.no_grad():
            qnn(emb[:B])            # lazy scale init (calibrate_network.py:235-238)
        steps_total = W + K
        # one "epoch" of W+K+1 batches (shuffled passes over the frames chained): iters = len(loader) gives
        # int(0.05*iters/len) = 0 phase-1 epochs and exactly one phase-2 epoch, for any K
        loader = CacheLoader(cache, list(range(wl["n"])), gB, seed=903, rank=rank, world=world, epoch_batches=steps_total + 1)
        t = {}
        phase = PROF_PHASE if K > PROF_PHASE else 0

        def hook(done):
            if os.environ.get("NQ_BENCH_STEPLOG"):   # debugging: host time at every iteration boundary -> stderr
                t.setdefault("log", []).append((done, time.perf_counter()))
            if W < done < steps_total:
                ops.profile_sample((done - W) % PROF_EVERY == phase)
            if done == W:
                # the two profiled iterations of the timed region run eagerly (hundreds of Python calls each): a generation-2
                # garbage collection landing in one of them costs ~10 ms of host time = 0.9 ms/step on the 20-step average
                # (seen as a 310 it/s outlier among the repeats).  Collect before, not during.
                gc.collect()
                gc.disable()
                sync()
                if not os.environ.get("NQ_BENCH_NOPROF"):
                    ops.profile_start()
                    ops.profile_sample(phase == 0)
                t["t0"] = time.perf_counter()
            elif done == steps_total and "t1" not in t:
                # (first arrival only: after a phase-1 run model_reconstruction goes on to set phase 2 up and its first
                # iteration reports the same count again -- that set-up is not part of the timed steps)
                t["t_enq"] = time.perf_counter()   # host finished enqueueing the timed steps (before the device drains)
                sync()
                t["t1"] = time.perf_counter()
                gc.enable()
                t["prof"] = ops.profile_stop() if not os.environ.get("NQ_BENCH_NOPROF") else {}

        # iters = 20 * len(loader): int(0.05 * iters / len) = 1 phase-1 epoch of W+K+1 iterations, cut after W+K by max_steps
        model_reconstruction(qnn, cali_data=emb, gt=loader, arch=arch, batch_size=gB,
                             iters=(20 if phase1 else 1) * len(loader),
                             hadamard=nerv, warmup=0.0, max_steps=steps_total, step_hook=hook, **FLAGS)
        if "t1" not in t:
            hook(steps_total)
        ops.set_conv_precision(None)
        if os.environ.get("NQ_BENCH_DUMP_PROF") and t.get("prof"):   # debugging: per-kernel averages of every timed run
            print("prof", workload, precision, sorted(((k[0], k[2], k[3], round(ms / c * 1e3, 1)) for k, (c, ms) in t["prof"].items()),
                                                      key=lambda r: -r[3])[:12], file=sys.stderr)
        if "log" in t:
            print("steplog", workload, "phase1" if phase1 else "phase2", precision,
                  [(d, round((b - a) * 1e3, 2)) for (_, a), (d, b) in zip(t["log"], t["log"][1:])], file=sys.stderr)
        elapsed = t["t1"] - t["t0"]
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        prof_steps = len([d for d in range(W, steps_total) if (d - W) % PROF_EVERY == phase])
        del qnn, model
        torch.cuda.empty_cache()
        return dict(elapsed=float(tmax.item()), t_enq=t["t_enq"] - t["t0"], prof=t["prof"], avg_bits=avg_bits,
                    prof_steps=prof_steps)

    nerv = args.workload == "nerv"
    main_run = timed_run(args.precision, K, W)
    # the same K timed steps again (fresh model, same warm-up): the spread the 47 ms timed region has on this box goes into
    # the line (`repeats`); `value` stays the first, contract run
    more_runs = [timed_run(args.precision, K, W) for _ in range(max(args.repeats - 1, 0))]
    # the secondary objects (exact fp32, phase 1, NeRV) describe the single-GPU kernels: N = 1 only -- at N > 1 the line is
    # the data-parallel headline and its repeats, nothing else rides on the collectives
    if world > 1:
        args.no_fp32 = args.no_phase1 = args.no_nerv = args.no_trained = args.no_uvg = True
    fp32_run = None
    if not args.no_fp32:
        Kf = max(1, min(K, args.fp32_steps))
        fp32_run = (timed_run("fp32", Kf, W), Kf)
    # phase 1 (5 % of a run: 990 of the 20 988 iterations) timed the same way
    p1_run = timed_run("bf16x3", K, W, phase1=True) if not args.no_phase1 else None
    # BASELINE configs[2] (NeRV Bunny_1280x640_3M + Hadamard) as a measured object of the same line
    nerv_run = None
    if args.workload == "hnerv" and not args.no_nerv:
        nerv_run = timed_run("bf16x3", K, W, workload="nerv")
    # the headline iteration on REAL data: the committed 38 dB checkpoint and the eight real Bunny crops
    trained_run = None
    if args.workload == "hnerv" and not args.no_trained and gB <= 8:
        trained_run = timed_run("bf16x3", K, W, workload="trained")
    # BASELINE configs[4] shape on one GPU + the Omega sweep, timed per candidate
    uvg_run = omega = None
    if args.workload == "hnerv" and not args.no_uvg and not strong:
        Ku = max(1, min(K, args.uvg_steps))
        uvg_run = (timed_run("bf16x3", Ku, min(W, 3), workload="uvg12m"), Ku)
        omega = omega_sweep_timing(workload_of("uvg12m"), dev, B)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    elapsed = main_run["elapsed"]
    per_step_units = 1 if strong else world     # strong: one global-batch iteration per step; weak: `world` B=2-equivalents
    if os.environ.get("NQ_BENCH_NOPROF"):   # throughput only (measures the cost of the event records themselves)
        print(json.dumps({"value": round(K * per_step_units / elapsed, 3), "ms_per_step": round(elapsed / K * 1e3, 3),
                          "note": "no profiling"}))
        return
    roofline, rows = roofline_of(main_run["prof"], elapsed, K, main_run["t_enq"], args.precision, main_run["prof_steps"])
    try:  # per-kernel table for DESIGN.md / profiles (scratch; not part of the contract line)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_n{world}.json"), "w") as f:
            json.dump(rows, f, indent=1)
    except OSError:
        pass

    fp32 = None
    if fp32_run is not None:
        r, Kf = fp32_run
        rl, rows32 = roofline_of(r["prof"], r["elapsed"], Kf, r["t_enq"], "fp32", r["prof_steps"])
        fp32 = {"value": round(Kf * per_step_units / r["elapsed"], 3), "ms_per_step": round(r["elapsed"] / Kf * 1e3, 3),
                "steps": Kf, "dtype": "f32 (exact fp32-input MFMA, NQ_CONV_PRECISION=fp32)", "roofline": rl}
        try:
            with open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_fp32_n{world}.json"), "w") as f:
                json.dump(rows32, f, indent=1)
        except OSError:
            pass

    nerv_obj = None
    if nerv_run is not None:
        rl, rows_n = roofline_of(nerv_run["prof"], nerv_run["elapsed"], K, nerv_run["t_enq"], "bf16x3", nerv_run["prof_steps"])
        nerv_obj = {"value": round(K * per_step_units / nerv_run["elapsed"], 3),
                    "ms_per_step": round(nerv_run["elapsed"] / K * 1e3, 3), "steps": K, "unit": "it/s",
                    "config": {"workload": "NeRV Bunny_1280x640_3M + Hadamard (BASELINE configs[2]), channel_wise, bits 6 5 4 5 5 6 6, "
                                           "phase-2 (AdaRound) iteration", "per_gpu_batch": B, "avg_bits": nerv_run["avg_bits"]},
                    "roofline": rl,
                    # whole step against both roofs (76.0 GFLOP and 1 531 + 259 MB per B = 2 iteration)
                    "step_tflops": round(76.0e9 * (B / 2) / (nerv_run["elapsed"] / K) / 1e12, 2),
                    "step_gbps": round(1.79e9 * (B / 2) / (nerv_run["elapsed"] / K) / 1e9, 1)}
        try:
            with open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_nerv_n{world}.json"), "w") as f:
                json.dump(rows_n, f, indent=1)
        except OSError:
            pass

    trained_obj = None
    if trained_run is not None:
        rl, rows_t = roofline_of(trained_run["prof"], trained_run["elapsed"], K, trained_run["t_enq"], "bf16x3", trained_run["prof_steps"])
        trained_obj = {"value": round(K * per_step_units / trained_run["elapsed"], 3),
                       "ms_per_step": round(trained_run["elapsed"] / K * 1e3, 3), "steps": K, "unit": "it/s",
                       "data": "real: tests/golden/hnerv3m_bunny8real_f16.npz (trained, FP 38.07 dB) on tests/golden/bunny8_640x1280.npz",
                       "dominant_kernel": rl and rl["kernel"], "dominant_kernel_ms": rl and rl["avg_launch_ms"],
                       "synthetic_dominant_kernel_ms": roofline and roofline["avg_launch_ms"],
                       "roofline": rl,
                       "note": "same kernels, same shapes, same K steps as `value`; only the operand VALUES differ (trained "
                               "weights, real frames): what the clock-vs-toggle-rate dependence of this chip does to the number"}
        try:
            with open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_trained_n{world}.json"), "w") as f:
                json.dump(rows_t, f, indent=1)
        except OSError:
            pass

    uvg_obj = None
    if uvg_run is not None:
        r, Ku = uvg_run
        rl, rows_u = roofline_of(r["prof"], r["elapsed"], Ku, r["t_enq"], "bf16x3", r["prof_steps"])
        conv_gflop = sum(x["gflop_per_launch"] * x["launches"] for x in rows_u) / max(r["prof_steps"], 1)
        conv_mb = sum(x["mbytes_per_launch"] * x["launches"] for x in rows_u) / max(r["prof_steps"], 1)
        uvg_obj = {"value": round(Ku * per_step_units / r["elapsed"], 3), "ms_per_step": round(r["elapsed"] / Ku * 1e3, 3),
                   "steps": Ku, "unit": "it/s",
                   "config": {"workload": "HNeRV UVG 960x1920, 11.84 M decoder parameters (tools/hnerv_uvg_12m.yaml = the reference's "
                                          "UVG_1920x960_3M.yaml with dec_in_channel 185; BASELINE configs[4] shape), channel_wise, "
                                          "bits 6 5 4 5 5 6 6, phase-2 (AdaRound) iteration, synthetic frames",
                              "per_gpu_batch": B, "avg_bits": r["avg_bits"]},
                   "roofline": rl,
                   "conv_gflop_per_step": round(conv_gflop, 1), "conv_mbytes_per_step": round(conv_mb, 1),
                   "step_tflops": round(conv_gflop * 1e9 / (r["elapsed"] / Ku) / 1e12, 2),
                   "step_mfma_frac": round(conv_gflop * 1e9 / (r["elapsed"] / Ku) / 1e12 / PEAK_BF16X3_TFLOPS, 4),
                   "omega_sweep": omega}
        try:
            with open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_uvg_n{world}.json"), "w") as f:
                json.dump(rows_u, f, indent=1)
        except OSError:
            pass

    # ---- PSNR vs the CPU oracle + CPU baseline timing (BASELINE configs[0]); single GPU, headline workload only ----
    psnr = cpu = None
    if world == 1 and not args.no_cpu_baseline and args.workload == "hnerv":
        psnr, cpu = psnr_and_cpu_baseline(dev)

    value = K * per_step_units / elapsed
    out = {
        "metric": "calibration iters/sec (HNeRV Bunny 1280x640, 21k iters) + final PSNR vs ref",
        "value": round(value, 3),
        "unit": (f"it/s (global batch {gB} frames per iteration, sharded over the GPUs)" if strong
                 else "it/s (B=2 frames per iteration-equivalent, summed over GPUs)"),
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": ("f32 operands split into bf16 hi+lo, 3 BF16 MFMAs per product, fp32 accumulate (bf16x3); fp32 MFMA on small layers"
                  if args.precision == "bf16x3" else "f32 (exact fp32-input MFMA, --precision fp32)"),
        "data": "synthetic",
        "config": {"workload": {"nerv": "NeRV Bunny_1280x640_3M + Hadamard", "hnerv": "HNeRV Bunny_1280x640_3M",
                                "uvg12m": "HNeRV UVG 960x1920 ~12M (tools/hnerv_uvg_12m.yaml)"}[args.workload]
                   + ", channel_wise, bits 6 5 4 5 5 6 6, phase-2 (AdaRound) iteration",
                   "per_gpu_batch": B, "global_batch": gB, "frames": workload_of(args.workload)["n"], "avg_bits": main_run["avg_bits"],
                   "parallelism": f"dp{world}"},
        "roofline": roofline,
        "repeats": {"values": [round(K * per_step_units / r["elapsed"], 3) for r in [main_run] + more_runs],
                    "note": "it/s of independent timed runs of the same K steps on this box; `value` is the first"},
        "fp32": fp32,
        "phase1": None if p1_run is None else {
            "value": round(K * per_step_units / p1_run["elapsed"], 3), "ms_per_step": round(p1_run["elapsed"] / K * 1e3, 3),
            "steps": K, "note": "phase-1 iterations (UAQ fake-quant, d(delta), Adam on the scales) of the same workload; a 21k "
                                "run is 990 of these + 19 998 phase-2 iterations (`value`)"},
        "nerv": nerv_obj,
        "trained": trained_obj,
        "uvg": uvg_obj,
        "psnr": psnr,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def omega_sweep_timing(wl, dev, B):
    """The Omega bit-allocation sweep (reference methods/bit_assign.py:57-118, 171-217, 343-372) on the UVG-12M workload, timed
    per candidate: v'Hv of the two toy candidates over the workload's frames in batches of B (the reference uses the first
    10 batches; the 8 synthetic frames give 4), double backward through all 7 layers on the HIP kernels."""
    import copy
    from neuroquant_amd.methods import bit_assign
    from neuroquant_amd.quantization import QuantModel
    model = wl["make"]().eval()
    cache, n, emb = wl["cache"], wl["n"], wl["emb"]
    batches = []
    for i in range(0, n, B):
        idx = torch.arange(i, min(i + B, n), device=dev)
        batches.append(dict(img=cache.batch(idx), idx=idx, norm_idx=idx.float() / n))
    out = {"batches_per_candidate": len(batches), "batch": B, "candidates": {}}
    for name, bits in bit_assign.hnerv_candidate.items():
        qn = QuantModel(copy.deepcopy(model), hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
        qn.eval()
        qn.set_bitwidth(bits)
        qn.set_quant_state(True)
        with torch.no_grad():
            qn(emb[:B])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        score = float(bit_assign.sensitivity_criterion("omega", "hnerv", copy.deepcopy(model), qn, batches))
        torch.cuda.synchronize()
        out["candidates"][name] = {"bits": bits, "score": score, "seconds": round(time.perf_counter() - t0, 3)}
        del qn
    out["seconds_per_candidate"] = round(sum(c["seconds"] for c in out["candidates"].values()) / len(out["candidates"]), 3)
    del model
    torch.cuda.empty_cache()
    return out


def psnr_and_cpu_baseline(dev):
    """BASELINE configs[0]: HNeRV-3M, 8 frames, --precision 6 5 4 5 5 6 6, iters_w = 50 (0 phase-1 epochs + 12 phase-2
    epochs = 48 iterations) on the COMMITTED trained checkpoint at the reference's operating point
    (tests/golden/hnerv3m_bunny8real_f16.npz: FP 38.07 dB on the first 8 Bunny frames cropped to 640x1280 as the reference's
    loader does, tests/golden/bunny8_640x1280.npz; reference log: FP 37.57 dB): GPU exact fp32, GPU bf16x3 and the CPU oracle
    on the same checkpoint / frames / recorded batch order.  The oracle run doubles as the CPU baseline: iterations 5..48
    timed, all host cores of this box's share.  48 iterations move the model little (the full-length
    fp32-vs-bf16x3 comparison is profiles/r03_precision_gate_21000.json); this leg pins the forward chain, the first 48 Adam
    steps and the CPU rate."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import precision_gate as pg
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the box's CPU share for one GPU is 16 cores (os.cpu_count() reports the whole host; oversubscribing it made the
    # oracle 10x slower), so use the affinity mask capped at 16
    cores = max(1, min(avail, 16))
    n, B, iters = 8, 2, 50
    frames_u8 = pg.bunny_real_640(dev, n)
    model, emb, _ = pg.load_fixture_checkpoint("hnerv3m_bunny8real_f16.npz", dev)
    from neuroquant_amd import ops
    with torch.no_grad():
        fp_psnr = float(torch.cat([ops.frame_psnr(model.decode(emb[i:i + 1])[0], frames_u8[i:i + 1].float() / 255.0)
                                   for i in range(n)]).double().mean())
    order = pg.make_order(n, B, iters)
    g32, _, _ = pg.calibrate_gpu(model, frames_u8, emb, order, iters, "fp32", record=False)
    g3, _, _ = pg.calibrate_gpu(model, frames_u8, emb, order, iters, "bf16x3", record=False)
    stamps = {}

    def on_step(done):
        stamps[done] = time.perf_counter()

    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    c, clog, _ = pg.calibrate_oracle(sd, frames_u8, emb, order, iters, cores, on_step=on_step)
    t_end = time.perf_counter()
    n_it = len(clog)
    skip = 4
    per_iter = (t_end - stamps[skip]) / (n_it - skip)
    psnr = {"config": f"BASELINE configs[0]: HNeRV-3M, first {n} Bunny frames cropped to 640x1280, B={B}, iters_w={iters} ({n_it} "
                      f"phase-2 iterations), committed checkpoint tests/golden/hnerv3m_bunny8real_f16.npz",
            "fp_model": round(fp_psnr, 4), "q_noopt": round(c["q_noopt"], 4),
            "oracle": round(c["q_opt"], 4), "fp32": round(g32["q_opt"], 4), "bf16x3": round(g3["q_opt"], 4),
            "max_abs_diff_dB": round(max(abs(g32["q_opt"] - c["q_opt"]), abs(g3["q_opt"] - c["q_opt"])), 5),
            "tolerance_dB": 0.02}
    cpu = {"value": round(1.0 / per_iter, 4), "unit": "it/s", "cores": cores, "kind": "port",
           "sample": f"configs[0]: {n_it} phase-2 iterations (B=2) of HNeRV-3M at 640x1280, first {skip} discarded, oracle on "
                     f"torch-CPU with {cores} threads, {per_iter:.2f} s/iter; final PSNR compared in `psnr`"}
    return psnr, cpu


if __name__ == "__main__":
    main()
