#!/usr/bin/env python3
"""Micro-benchmark of the convolutions of NeRV-3M's last two blocks (B = 2; 24->96 at 160x320 and 24->96 at 320x640, both
k3 + PixelShuffle 2 + GELU): forward, data gradient and weight gradient, median of interleaved rounds, algorithmic bytes per second.

    python tools/bench_nerv_tail.py           # NQ_IG3_OCC3_STEPS=<n> selects the 3-waves-per-SIMD build for <= n k-steps
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuroquant_amd import ops  # noqa: E402


def timeit(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    cases = []
    for name, cin, H, W, cout, rp in (("blk5", 24, 320, 640, 96, 2), ("blk4", 24, 160, 320, 96, 4)):
        B, k, r = 2, 3, 2
        x = torch.randn(B, cin, H, W, generator=g).cuda()
        w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda()
        b = (torch.randn(cout, generator=g) * 0.1).cuda()
        dy = torch.randn(B, cout, H, W, generator=g).cuda()
        zprev = torch.rand(B, cin, H, W, generator=g).cuda()
        wt3, wt3t = ops.weight_layout3(w), ops.weight_layout3(w, transposed=True)
        by_f = 4.0 * B * H * W * (cin + 2 * cout)
        by_d = 4.0 * B * H * W * (cout + 2 * cin)
        cases.append((f"{name} fwd   {cin}->{cout} {H}x{W}", by_f,
                      lambda x=x, wt3=wt3, b=b, cout=cout: ops.conv3_forward_raw(x, wt3, b, cout, 3, ops.EPI_PS_GELU, 2)))
        cases.append((f"{name} dgrad {cout}->{cin} {H}x{W}", by_d,
                      lambda dy=dy, wt3t=wt3t, cin=cin, rp=rp, zprev=zprev:
                      ops.conv3_forward_raw(dy, wt3t, None, cin, 3, ops.EPI_DGRAD_GELU, rp, zprev=zprev)))
        by_w = 4.0 * B * H * W * (cin + cout)
        cases.append((f"{name} wgrad {cin}x{cout} {H}x{W}", by_w,
                      lambda x=x, dy=dy, cout=cout: ops.conv_wgrad3_raw(x, dy, cout, 3, True)))
    ts = {n: [] for n, _, _ in cases}
    for _ in range(2):
        for n, _, fn in cases:
            timeit(fn, 2)
    for _ in range(7):
        for n, _, fn in cases:
            ts[n].append(timeit(fn))
    print("NQ_IG3_OCC3_STEPS =", os.environ.get("NQ_IG3_OCC3_STEPS"))
    for n, by, _ in cases:
        t = sorted(ts[n])[len(ts[n]) // 2]
        print(f"  {n:32s} {t:8.1f} us  {by / t / 1e6:7.2f} TB/s")


if __name__ == "__main__":
    main()
