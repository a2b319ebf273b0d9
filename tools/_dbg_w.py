import os, sys, torch
sys.path.insert(0, '/root/repo')
from neuroquant_amd import ops
g = torch.Generator().manual_seed(0)
B, cin, H, W, cout, k = 2, 44, 48, 224, 148, 5
x = torch.randn(B, cin, H, W, generator=g).cuda(); dy = torch.randn(B, cout, H, W, generator=g).cuda()
os.environ["NQ_WGRAD3_PC"] = "0"; a, _ = ops.conv_wgrad3_raw(x, dy, cout, k, True)
os.environ["NQ_WGRAD3_PC"] = "1"; b, _ = ops.conv_wgrad3_raw(x, dy, cout, k, True)
d = (a - b).abs()
print("max", float(d.max()), "scale", float(a.abs().max()))
print("by kh,kw:\n", d.amax((0, 1)))
print("by ci:", d.amax((0, 2, 3)))
# isolate: x = delta at one position
for (yy, xx) in ((0, 0), (0, 100), (20, 0), (20, 223), (47, 100), (20, 100), (20,32),(20,31),(20,33)):
    x2 = torch.zeros_like(x); x2[0, 3, yy, xx] = 1.0
    os.environ["NQ_WGRAD3_PC"] = "0"; a, _ = ops.conv_wgrad3_raw(x2, dy, cout, k, True)
    os.environ["NQ_WGRAD3_PC"] = "1"; b, _ = ops.conv_wgrad3_raw(x2, dy, cout, k, True)
    print((yy, xx), float((a - b).abs().max()), float(a.abs().max()))
