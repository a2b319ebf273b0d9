#!/usr/bin/env python3
"""Head forward (37 -> 3, k3, tanh, 640x1280, B=2): LDS-staged kernel (NQ_HEAD_FWD=1) vs the register-streaming kernel
(default, round 3), interleaved rounds in one process + difference of the results."""
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
x = torch.randn(B, cin, H, W, generator=g).cuda()
w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda()
b = (torch.randn(cout, generator=g) * 0.1).cuda()
wt, dims, _, _ = ops.weight_layouts(w, False)
f = lambda: ops.conv_forward_raw(x, wt, dims, b, cout, k, ops.EPI_TANH, 1)[0]
os.environ["NQ_HEAD_FWD"] = "1"; y_old = f().clone()
os.environ["NQ_HEAD_FWD"] = "0"; y_new = f().clone()
ref = torch.tanh(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)) * 0.5 + 0.5
print("old vs f64", (y_old - ref).abs().max().item(), "new vs f64", (y_new - ref).abs().max().item(), "old vs new", (y_old - y_new).abs().max().item())
for rnd in range(3):
    os.environ["NQ_HEAD_FWD"] = "1"; a = t(f)
    os.environ["NQ_HEAD_FWD"] = "0"
    row = []
    for rd in (sys.argv[1:] or ["5,8"]):
        os.environ["NQ_HEAD_RD"] = rd
        c = t(f)
        row.append(f"{rd}: {c:.1f} us ({262.1e6 / c / 1e6:.2f} TB/s)")
    print(f"round {rnd}: LDS-staged {a:.1f} us | streaming " + " | ".join(row))
