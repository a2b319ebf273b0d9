#!/usr/bin/env python3
"""How does a conv kernel's time depend on the NUMBER of workgroups (partial last round on 256 CUs x 2 resident
workgroups)?  Times the bf16x3 data-gradient kernels of dec5 / dec4 at several image heights (same width, same channels)
and prints time, workgroups, rounds of 512 and time per workgroup-round.

    python tools/bench_tail.py
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuroquant_amd import ops  # noqa: E402


def timeit(fn, n=5, rounds=5):
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    g = torch.Generator().manual_seed(0)
    for name, cin, cout, W, rp, heights, th, tw in (
            ("dec5 dgrad 148->44", 44, 148, 640, 2, (192, 200, 208, 256, 296, 304, 312, 320, 328), 8, 32),
            ("dec4 dgrad 176->53", 53, 176, 320, 4, (128, 160, 200, 208, 256, 264), 8, 32),
            ("dec5 fwd 44->148", 44, 148, 640, 2, (200, 208, 256, 304, 312, 320), 8, 32)):
        w = (torch.randn(cout, cin, 5, 5, generator=g) / math.sqrt(cin * 25)).cuda()
        fwd = "fwd" in name
        wt3 = ops.weight_layout3(w, transposed=not fwd)
        b = torch.zeros(cout).cuda()
        for H in heights:
            if fwd:
                x = torch.randn(2, cin, H, W, generator=g).cuda()
                fn = lambda: ops.conv3_forward_raw(x, wt3, b, cout, 5, ops.EPI_PS_GELU, 2)
                co_tiles = 2
            else:
                dy = torch.randn(2, cout, H, W, generator=g).cuda()
                z = torch.rand(2, cin, H, W, generator=g).cuda()
                fn = lambda: ops.conv3_forward_raw(dy, wt3, None, cin, 5, ops.EPI_DGRAD_GELU, rp, zprev=z)
                co_tiles = 1
            fn()
            torch.cuda.synchronize()
            t = timeit(fn) * 1e3
            wgs = 2 * ((H + th - 1) // th) * ((W + tw - 1) // tw) * co_tiles
            print(f"{name:20s} H={H:4d}  {wgs:5d} workgroups = {wgs / 512:5.2f} rounds of 512   {t:7.1f} us   "
                  f"{t / wgs * 512:6.1f} us per full round   {t / math.ceil(wgs / 512):6.1f} us per started round", flush=True)


if __name__ == "__main__":
    main()
