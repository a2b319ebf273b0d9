#!/usr/bin/env python3
"""Per-kernel micro-benchmark of the hot convolutions at the shapes of HNeRV Bunny_1280x640_3M (B = 2): interleaved rounds
in ONE process (guide §5.4 rule 24), random data, median over rounds; algorithmic TFLOP/s and fraction of the bf16x3
ceiling (2500/3 TF).  Variants of a kernel selected by environment switches are timed side by side.

    python tools/bench_kernels.py [--rounds 7] [--only wgrad] [--check]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuroquant_amd import ops  # noqa: E402

PEAK3 = 2500.0 / 3.0
# (name, B, Cin, H, W, Cout, k, r) of the forward convolution of each big layer
LAYERS = [("dec5", 2, 44, 320, 640, 148, 5, 2), ("dec4", 2, 53, 160, 320, 176, 5, 2), ("dec3", 2, 64, 40, 80, 848, 5, 4),
          ("dec2", 2, 77, 10, 20, 1024, 3, 4)]


def timeit(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--check", action="store_true", help="compare variants / fp32 kernels (max error relative to the output scale)")
    args = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    cases = []
    for name, B, cin, H, W, cout, k, r in LAYERS:
        x = torch.randn(B, cin, H, W, generator=g).cuda()
        w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda()
        b = (torch.randn(cout, generator=g) * 0.1).cuda()
        dy = torch.randn(B, cout, H, W, generator=g).cuda()
        zprev = torch.rand(B, cin, H, W, generator=g).cuda()
        flops = 2.0 * B * cout * cin * k * k * H * W
        if ops.conv3_supported(B, cin, H, W, cout, k):
            wt3 = ops.weight_layout3(w)
            cases.append((f"{name} fwd   igemm3 {cin}->{cout}", flops, {"": lambda x=x, wt3=wt3, b=b, cout=cout, k=k, r=r:
                                                                          ops.conv3_forward_raw(x, wt3, b, cout, k, ops.EPI_PS_GELU, r)}))
        if ops.conv3_supported(B, cout, H, W, cin, k):
            wt3t = ops.weight_layout3(w, transposed=True)
            rp = {"dec5": 2, "dec4": 4, "dec3": 4, "dec2": 5}[name]   # shuffle factor of the layer BELOW (its un-shuffle)
            if H % rp == 0 and W % rp == 0:
                cases.append((f"{name} dgrad igemm3 {cout}->{cin}", flops, {"": lambda dy=dy, wt3t=wt3t, cin=cin, k=k, rp=rp, zprev=zprev:
                                                                            ops.conv3_forward_raw(dy, wt3t, None, cin, k, ops.EPI_DGRAD_GELU, rp, zprev=zprev)}))
        if ops.conv_wgrad3_supported(B, cin, H, W, cout, k):
            def wg(pc, x=x, dy=dy, cout=cout, k=k):
                os.environ["NQ_WGRAD3_PC"] = pc
                return ops.conv_wgrad3_raw(x, dy, cout, k, True)
            def wgs1(wg=wg):
                os.environ["NQ_WGRAD3_SS"] = "1"
                r = wg("1")
                os.environ["NQ_WGRAD3_SS"] = "0"
                return r
            var = {"4wave": lambda wg=wg: wg("0"), "pc seg32": wgs1, "prod/cons": lambda wg=wg: wg("1")}
            cases.append((f"{name} wgrad3 {cin}x{cout}", flops, var))
    # head layer 37 -> 3, k3, 640x1280 (streaming VALU kernels)
    hx = torch.randn(2, 37, 640, 1280, generator=g).cuda()
    hw = (torch.randn(3, 37, 3, 3, generator=g) / math.sqrt(37 * 9)).cuda()
    hb = torch.zeros(3).cuda()
    hdy = torch.randn(2, 3, 640, 1280, generator=g).cuda()
    hz = torch.rand(2, 37, 640, 1280, generator=g).cuda()
    hwt, hdims, hwb, hdims_b = ops.weight_layouts(hw, True)
    hfl = 2.0 * 2 * 3 * 37 * 9 * 640 * 1280
    cases.append(("head fwd 37->3", hfl, {"": lambda: ops.conv_forward_raw(hx, hwt, hdims, hb, 3, 3, ops.EPI_TANH, 1)}))
    cases.append(("head dgrad 3->37", hfl, {"": lambda: ops.conv_forward_raw(hdy, hwb, hdims_b, None, 37, 3, ops.EPI_DGRAD_GELU, 2, zprev=hz)}))
    cases.append(("head wgrad 37x3", hfl, {"": lambda: ops.conv_wgrad_swapped3(hx, hdy, 3, 3, True)}))
    cases = [c for c in cases if args.only in c[0]]
    for name, flops, variants in cases:
        for fn in variants.values():
            fn()
        torch.cuda.synchronize()
        if args.check and len(variants) > 1:
            outs = {k: fn() for k, fn in variants.items()}
            keys = list(outs)
            ref = outs[keys[0]]
            for kk in keys[1:]:
                for a_, b_ in zip(ref, outs[kk]):
                    if a_ is not None:
                        print(f"   {name}: {kk} vs {keys[0]} max |diff| / max|ref| = {float((a_ - b_).abs().max() / a_.abs().max()):.2e}")
        times = {k: [] for k in variants}
        for _ in range(args.rounds):
            for k, fn in variants.items():
                times[k].append(timeit(fn, args.reps))
        for k, ts in times.items():
            ts.sort()
            med = ts[len(ts) // 2]
            tf = flops / med / 1e9
            print(f"{name:34s} {k:10s} median {med * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us  {tf:7.1f} TF  frac {tf / PEAK3:.3f}", flush=True)


if __name__ == "__main__":
    main()
