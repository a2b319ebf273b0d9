#!/usr/bin/env python3
"""Race screen for the LDS-DMA weight ring of conv_igemm3: the kernels are deterministic, so every
repetition of a launch must reproduce the first one BIT FOR BIT -- also while a second stream keeps HBM and the L2s busy
(uneven load shifts when a DMA piece lands relative to the barrier that publishes it).  An early fragment read shows up as a
handful of differing output values in some repetition.

    python tools/race_screen.py [--reps 60]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuroquant_amd import ops  # noqa: E402

# (name, B, Cin, H, W, Cout, k, r, r_below): forward conv of the layer; its data gradient runs Cout -> Cin
SHAPES = [("hnerv dec5", 2, 44, 320, 640, 148, 5, 2, 2), ("hnerv dec4", 2, 53, 160, 320, 176, 5, 2, 4),
          ("hnerv dec3", 2, 64, 40, 80, 848, 5, 4, 4), ("hnerv dec2", 2, 77, 10, 20, 1024, 3, 4, 5),
          ("nerv blk5", 2, 24, 320, 640, 96, 3, 2, 2), ("nerv blk4", 2, 48, 160, 320, 96, 3, 2, 2),
          ("ragged", 3, 37, 67, 131, 52, 5, 2, 1)]


def screen(reps, only=None, log=print):
    """-> number of output words that differed from the first launch over all repetitions of all cases."""
    g = torch.Generator().manual_seed(7)
    side = torch.cuda.Stream()
    junk_a = torch.randn(64 << 20, device="cuda")     # 256 MB
    junk_b = torch.empty_like(junk_a)
    bad = 0
    for name, B, cin, H, W, cout, k, r, rb in SHAPES:
        if only is not None and name not in only:
            continue
        x = torch.randn(B, cin, H, W, generator=g).cuda()
        w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda()
        b = (torch.randn(cout, generator=g) * 0.1).cuda()
        dy = torch.randn(B, cout, H, W, generator=g).cuda()
        zprev = torch.rand(B, cin, H, W, generator=g).cuda()
        cases = []
        if ops.conv3_supported(B, cin, H, W, cout, k):
            wt3 = ops.weight_layout3(w)
            cases.append(("fwd", lambda: ops.conv3_forward_raw(x, wt3, b, cout, k, ops.EPI_PS_GELU, r)))
        if ops.conv3_supported(B, cout, H, W, cin, k) and H % rb == 0 and W % rb == 0:
            wt3t = ops.weight_layout3(w, transposed=True)
            cases.append(("dgrad", lambda: ops.conv3_forward_raw(dy, wt3t, None, cin, k, ops.EPI_DGRAD_GELU, rb, zprev=zprev)))
        for tag, fn in cases:
            ref = [t.clone() for t in fn() if t is not None]
            torch.cuda.synchronize()
            ndiff = 0
            for i in range(reps):
                if i % 2:   # every other repetition under load from a second stream
                    with torch.cuda.stream(side):
                        junk_b.copy_(junk_a)
                        junk_a.mul_(1.0)
                out = [t for t in fn() if t is not None]
                for o, rf in zip(out, ref):
                    ndiff += int((o.view(torch.int32) != rf.view(torch.int32)).sum())
            torch.cuda.synchronize()
            log(f"{name:12s} {tag:6s} {cin:4d}->{cout:4d} {H}x{W}: {reps} repetitions, {ndiff} differing words")
            bad += ndiff
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=60)
    args = ap.parse_args()
    bad = screen(args.reps)
    print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} words)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
