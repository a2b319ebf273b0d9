#!/usr/bin/env python3
"""Micro-benchmark: fp32-MFMA implicit GEMM vs the bf16x3 kernel on the decoder's big layers (forward + data gradient),
with the error of each against a float64 CPU reference on a slice."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuroquant_amd import ops

def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

g = torch.Generator().manual_seed(0)
for (B, cin, H, W, cout, k, epi, r) in [(2, 44, 320, 640, 148, 5, ops.EPI_PS_GELU, 2), (2, 148, 320, 640, 44, 5, ops.EPI_PLAIN, 1),
                                         (2, 53, 160, 320, 176, 5, ops.EPI_PS_GELU, 2), (2, 176, 160, 320, 53, 5, ops.EPI_PLAIN, 1),
                                         (2, 64, 40, 80, 848, 5, ops.EPI_PS_GELU, 4)]:
    x = torch.randn(B, cin, H, W, generator=g).cuda()
    w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    wt, dims, _, _ = ops.weight_layouts(w, False)
    wt3 = ops.weight_layout3(w)
    f32 = lambda: ops.conv_forward_raw(x, wt, dims, b, cout, k, epi, r)
    b3 = lambda: ops.conv3_forward_raw(x, wt3, b, cout, k, epi, r)
    y32, z32 = f32(); y3, z3 = b3()
    ref32 = z32 if z32 is not None else y32
    ref3 = z3 if z3 is not None else y3
    # float64 reference on a crop (first frame, rows 0..23)
    xs = x[:1, :, :24 + k, :].double().cpu(); yd = torch.nn.functional.conv2d(xs, w.double().cpu(), b.double().cpu(), padding=k // 2)[:, :, :24]
    if epi in (ops.EPI_PS_GELU,): yd = torch.nn.functional.pixel_shuffle(yd, r)
    rows = yd.shape[2]
    e32 = (ref32[:1, :, :rows].double().cpu() - yd).abs().max().item(); e3 = (ref3[:1, :, :rows].double().cpu() - yd).abs().max().item()
    flops = 2.0 * B * cout * cin * k * k * H * W
    t32, t3 = t(f32), t(b3)
    print(f"{cin:4d}->{cout:4d} k{k} {H}x{W}: fp32 {t32:.3f} ms ({flops/t32/1e9:.1f} TF, err {e32:.2e}) | bf16x3 {t3:.3f} ms ({flops/t3/1e9:.1f} TF, err {e3:.2e}) | speedup {t32/t3:.2f}x  scale {yd.abs().max().item():.2f}")
