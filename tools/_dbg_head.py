import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuroquant_amd import ops
torch.manual_seed(0)
def run(x, w, b):
    wt, dims, _, _ = ops.weight_layouts(w, False)
    os.environ["NQ_HEAD_FWD"] = "0"
    return ops.conv_forward_raw(x, wt, dims, b, 3, 3, ops.EPI_PLAIN, 1)[0]
B, cin, H, W = 1, 2, 12, 512
for (ci, yy, xx) in [(0, 5, 100), (1, 0, 0), (0, 11, 255), (0, 6, 256), (1, 3, 511)]:
    x = torch.zeros(B, cin, H, W, device="cuda"); x[0, ci, yy, xx] = 1.0
    w = torch.arange(3 * cin * 9, dtype=torch.float32, device="cuda").view(3, cin, 3, 3) + 1
    b = torch.zeros(3, device="cuda")
    y = run(x, w, b)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    d = (y - ref).abs()
    print("impulse", (ci, yy, xx), "maxdiff", d.max().item())
    nz = (y != 0).nonzero()[:12].tolist(); nr = (ref != 0).nonzero()[:12].tolist()
    print("  got nz", [(t, round(y[tuple(t)].item(), 1)) for t in nz][:9])
    print("  ref nz", [(t, round(ref[tuple(t)].item(), 1)) for t in nr][:9])
x = torch.randn(1, 3, 12, 512, device="cuda"); w = torch.randn(3, 3, 3, 3, device="cuda"); b = torch.randn(3, device="cuda")
y = run(x, w, b); ref = torch.nn.functional.conv2d(x, w, b, padding=1)
d = (y - ref).abs(); print("random maxdiff", d.max().item(), "rows with err", (d.amax(dim=(0, 1, 3)) > 1e-4).nonzero().flatten().tolist(), "cols", (d.amax(dim=(0, 1, 2)) > 1e-4).nonzero().flatten().tolist()[:20])
