#!/usr/bin/env python3
"""Micro-benchmark of the head convolution (37 -> 3, k3, 640x1280): streaming VALU kernels vs the bf16x3 implicit GEMM,
forward (+tanh) and data gradient (+gelu', un-shuffle)."""
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
B, cin, H, W, cout, k = 2, 37, 640, 1280, 3, 3
x = torch.randn(B, cin, H, W, generator=g).cuda()
w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda()
b = (torch.randn(cout, generator=g) * 0.1).cuda()
dy = torch.randn(B, cout, H, W, generator=g).cuda()
z = torch.randn(B, cin, H, W, generator=g).cuda()
wt, dims, wb, dims_b = ops.weight_layouts(w, True)
wt3, wt3t = ops.weight_layout3(w), ops.weight_layout3(w, transposed=True)
f_old = lambda: ops.conv_forward_raw(x, wt, dims, b, cout, k, ops.EPI_TANH, 1)
f_new = lambda: ops.conv3_forward_raw(x, wt3, b, cout, k, ops.EPI_TANH, 1)
d_old = lambda: ops.conv_forward_raw(dy, wb, dims_b, None, cin, k, ops.EPI_DGRAD_GELU, 2, zprev=z)
d_new = lambda: ops.conv3_forward_raw(dy, wt3t, None, cin, k, ops.EPI_DGRAD_GELU, 2, zprev=z)
yo, yn = f_old()[0], f_new()[0]
do, dn = d_old()[0], d_new()[0]
print("fwd  max diff", (yo - yn).abs().max().item(), " dgrad max diff", (do - dn).abs().max().item(), "scale", do.abs().max().item())
print(f"head fwd  : VALU {t(f_old):.3f} ms | bf16x3 {t(f_new):.3f} ms")
print(f"head dgrad: VALU {t(d_old):.3f} ms | bf16x3 {t(d_new):.3f} ms")
