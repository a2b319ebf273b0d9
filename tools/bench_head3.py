#!/usr/bin/env python3
"""Head data gradient (3 -> 37, k3, x gelu', PixelUnshuffle(2), 640x1280, B=2): LDS-staged kernel (NQ_HEAD_DGRAD=1) vs the
register-streaming kernel (default, round 3), interleaved rounds in one process + difference of the results."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuroquant_amd import ops

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

g = torch.Generator().manual_seed(0)
B, cin, H, W, cout, k = 2, 37, 640, 1280, 3, 3
w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda()
dy = torch.randn(B, cout, H, W, generator=g).cuda()
z = torch.randn(B, cin, H, W, generator=g).cuda()
_, _, wb, dims_b = ops.weight_layouts(w, True)
f = lambda: ops.conv_forward_raw(dy, wb, dims_b, None, cin, k, ops.EPI_DGRAD_GELU, 2, zprev=z)[0]
os.environ["NQ_HEAD_DGRAD"] = "1"; y_old = f().clone()
os.environ["NQ_HEAD_DGRAD"] = "0"
ref = torch.nn.functional.pixel_unshuffle(torch.nn.functional.conv_transpose2d(dy.double(), w.double(), padding=1) * z.double(), 2)
print("old vs f64", (y_old - ref).abs().max().item(), "scale", ref.abs().max().item())
for rd in (sys.argv[1:] or ["2"]):
    os.environ["NQ_HEAD_DG_R"] = rd
    y_new = f().clone()
    print(f"R={rd}: new vs f64 {(y_new - ref).abs().max().item():.3e}  old vs new {(y_old - y_new).abs().max().item():.3e}")
for rnd in range(3):
    os.environ["NQ_HEAD_DGRAD"] = "1"; a = t(f)
    os.environ["NQ_HEAD_DGRAD"] = "0"
    row = []
    for rd in (sys.argv[1:] or ["2"]):
        os.environ["NQ_HEAD_DG_R"] = rd
        c = t(f)
        row.append(f"R={rd}: {c:.1f} us ({505.0e6 / c / 1e6:.2f} TB/s)")
    print(f"round {rnd}: LDS-staged {a:.1f} us | streaming " + " | ".join(row))
