"""Torch-facing wrappers of the libnqhip C ABI (include/nq_hip.h).

PyTorch is plumbing here: device memory, the current HIP stream and the autograd tape.  All arithmetic on
the hot path runs in the hand-written gfx950 kernels; there is NO CPU / eager fallback -- every op raises if
its tensors are not fp32 HIP tensors or if libnqhip.so is missing.
"""
import contextlib
import ctypes
import functools
import math
import os
import threading

import torch
from torch.autograd import Function

from . import _lib as L

_QCACHE = {}


def _q(name, *args):
    """Pure shape queries of the library (supported / workspace size / operand size), memoised: ~40 ctypes round trips
    per iteration otherwise.  The two environment switches that change a plan are part of the key (tools/bench_kernels.py
    flips them between launches of one process)."""
    key = (name, args, os.environ.get("NQ_WGRAD3_PC"), os.environ.get("NQ_WGRAD3_SS"))
    v = _QCACHE.get(key)
    if v is None:
        v = _QCACHE[key] = getattr(L.lib(), name)(*args)
    return v

EPI_PLAIN, EPI_PS_GELU, EPI_TANH, EPI_PS, EPI_DGRAD_GELU = (L.EPI_PLAIN, L.EPI_PS_GELU, L.EPI_TANH, L.EPI_PS,
                                                                L.EPI_DGRAD_GELU)
# split {hi | lo} word interchange between the bf16x3 kernels (include/nq_hip.h: NQ_EPI_X_SPLIT / NQ_EPI_Y_SPLIT)
EPI_X_SPLIT, EPI_Y_SPLIT = 0x100, 0x200


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- optional per-launch timing with HIP events on the launch stream (bench.py's roofline object) ----
_PROF = None
_PROF_ON = True


def profile_start():
    global _PROF, _PROF_ON
    _PROF, _PROF_ON = {}, True


def profile_sample(on: bool):
    """Pause / resume recording while a profile is open (bench.py times every N-th step only: each event record is a
    marker packet in the queue, 50 of them per step cost ~4 % throughput)."""
    global _PROF_ON
    _PROF_ON = bool(on)


def profile_stop():
    """-> {key: (launches, total_ms)}; synchronises."""
    global _PROF
    prof, _PROF = _PROF, None
    torch.cuda.synchronize()
    return {k: (len(v), sum(s.elapsed_time(e) for s, e in v)) for k, v in (prof or {}).items()}


def profiling_active():
    """True while per-launch HIP events are being recorded (such steps are launched eagerly, never replayed from a graph)."""
    return _PROF is not None and _PROF_ON


def _timed(key, fn):
    if _PROF is None or not _PROF_ON:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = fn()
    e.record()
    _PROF.setdefault(key, []).append((s, e))
    return out


def _dev(t: torch.Tensor, name="tensor") -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"neuroquant_amd: {name} must live on the GPU (no CPU fallback exists)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"neuroquant_amd: {name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _rows(x: torch.Tensor, scale: torch.Tensor):
    """(rows, row_len, per_row) of the quantiser kernels for tensor x and its delta/zp tensor."""
    if scale.numel() > 1:
        rows = x.shape[0]
        if scale.numel() != rows:
            raise RuntimeError("delta/zero_point must hold one value per output channel or a single value")
        return rows, x.numel() // rows, 1
    return 1, x.numel(), 0


# ------------------------------------------------------------------------------------------ quantisers
def scale_init_max(x: torch.Tensor, n_levels: int, channel_wise: bool):
    """UniformAffineQuantizer.init_quantization_scale 'max' (reference quantizer.py:127-168)."""
    x = _dev(x.detach(), "x")
    if channel_wise:
        if x.dim() == 4:
            rows, shape = x.shape[0], (-1, 1, 1, 1)
        elif x.dim() == 1:
            rows, shape = 1, (-1,)
        else:
            raise ValueError
    else:
        rows, shape = 1, ()
    delta = torch.empty(rows, device=x.device, dtype=torch.float32)
    zp = torch.empty_like(delta)
    L.check(L.lib().nq_scale_init_max(_p(x), rows, x.numel() // rows, n_levels, _p(delta), _p(zp), _stream()), "scale_init")
    return delta.view(shape), zp.view(shape)


def uaq_forward(x, delta, zp, n_levels, out=None):
    x, delta, zp = _dev(x), _dev(delta), _dev(zp)
    rows, rl, per_row = _rows(x, delta)
    y = torch.empty_like(x) if out is None else out
    L.check(L.lib().nq_uaq_forward(_p(x), _p(delta), _p(zp), _p(y), rows, rl, per_row, n_levels, _stream()), "uaq_forward")
    return y


def uaq_backward(x, gy, delta, zp, n_levels, want_dx=False):
    """d(delta) of the UAQ fake-quant; want_dx -> (d(delta), dx) with dx the straight-through gradient w.r.t. x."""
    x, gy, delta, zp = _dev(x), _dev(gy), _dev(delta), _dev(zp)
    rows, rl, per_row = _rows(x, delta)
    dd = torch.empty_like(delta)
    dx = torch.empty_like(x) if want_dx else None
    L.check(L.lib().nq_uaq_backward(_p(x), _p(gy), _p(delta), _p(zp), _p(dd), _p(dx), rows, rl, per_row, n_levels, _stream()),
            "uaq_backward")
    return (dd, dx) if want_dx else dd


def adaround_init(x, delta_uaq, zp_uaq):
    """-> (delta, zp) after the fp16 round trip, alpha  (reference quantizer.py:264-265, 305-314)."""
    x, d0, z0 = _dev(x.detach()), _dev(delta_uaq.detach()), _dev(zp_uaq.detach())
    rows, rl, per_row = _rows(x, d0)
    d, z, a = torch.empty_like(d0), torch.empty_like(z0), torch.empty_like(x)
    L.check(L.lib().nq_adaround_init(_p(x), _p(d0), _p(z0), _p(d), _p(z), _p(a), rows, rl, per_row, _stream()),
            "adaround_init")
    return d, z, a


def adaround_forward(x, alpha, delta, zp, n_levels, soft, want_xq=False, out=None):
    x, alpha, delta, zp = _dev(x), _dev(alpha), _dev(delta), _dev(zp)
    rows, rl, per_row = _rows(x, delta)
    y = torch.empty_like(x) if out is None else out
    xq = torch.empty_like(x) if want_xq else None
    L.check(L.lib().nq_adaround_forward(_p(x), _p(alpha), _p(delta), _p(zp), _p(y), _p(xq), rows, rl, per_row, n_levels,
                                        1 if soft else 0, _stream()), "adaround_forward")
    return (y, xq) if want_xq else y


def adaround_backward(x, gy, alpha, delta, zp, n_levels, reg_weight=0.0, reg_b=0.0, out=None):
    x, gy, alpha, delta, zp = _dev(x), _dev(gy), _dev(alpha), _dev(delta), _dev(zp)
    rows, rl, per_row = _rows(x, delta)
    da = torch.empty_like(alpha) if out is None else out
    L.check(L.lib().nq_adaround_backward(_p(x), _p(gy), _p(alpha), _p(delta), _p(zp), _p(da), rows, rl, per_row, n_levels,
                                         float(reg_weight), float(reg_b), _stream()), "adaround_backward")
    return da


def round_loss(alpha, b, weight, out=None, accumulate=False):
    alpha = _dev(alpha)
    n = alpha.numel()
    ws = torch.empty(_q("nq_reduce_ws_floats", n), device=alpha.device, dtype=torch.float32)
    if out is None:
        out = torch.zeros((), device=alpha.device, dtype=torch.float32)
    L.check(L.lib().nq_round_loss(_p(alpha), n, float(b), float(weight), _p(ws), _p(out), 1 if accumulate else 0, _stream()),
            "round_loss")
    return out


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.Adam update of tensor p in place (step counts from 1)."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    L.check(L.lib().nq_adam_step(_p(p), _p(_dev(g)), _p(m), _p(v), p.numel(), lr / bc1, beta1, beta2, eps, bc2 ** 0.5,
                                 _stream()), "adam_step")


def adaround_forward_multi(items):
    """items: [(x, alpha, delta, zp, n_levels, soft)] -> [fake-quantised tensors], ONE launch for all of them
    (same arithmetic as adaround_forward, bit for bit)."""
    segs = (L.AdaSeg * len(items))()
    outs = []
    for sg, (x, alpha, delta, zp, n_levels, soft) in zip(segs, items):
        x, alpha, delta, zp = _dev(x), _dev(alpha), _dev(delta), _dev(zp)
        rows, rl, per_row = _rows(x, delta)
        y = torch.empty_like(x)
        outs.append(y)
        sg.x, sg.gy, sg.alpha, sg.delta, sg.zp, sg.out = _p(x), None, _p(alpha), _p(delta), _p(zp), _p(y)
        sg.rows, sg.row_len, sg.per_row, sg.n_levels, sg.soft, sg.reg_weight = rows, rl, per_row, n_levels, int(bool(soft)), 0.0
    L.check(L.lib().nq_adaround_forward_multi(segs, len(items), _stream()), "adaround_forward_multi")
    return outs


def uaq_forward_multi(items):
    """items: [(x, delta, zp, n_levels)] -> [UAQ fake-quantised tensors], ONE launch (same arithmetic as uaq_forward)."""
    segs = (L.AdaSeg * len(items))()
    outs = []
    for sg, (x, delta, zp, n_levels) in zip(segs, items):
        x, delta, zp = _dev(x), _dev(delta), _dev(zp)
        rows, rl, per_row = _rows(x, delta)
        y = torch.empty_like(x)
        outs.append(y)
        sg.x, sg.gy, sg.alpha, sg.delta, sg.zp, sg.out = _p(x), None, None, _p(delta), _p(zp), _p(y)
        sg.rows, sg.row_len, sg.per_row, sg.n_levels, sg.soft, sg.reg_weight = rows, rl, per_row, n_levels, 0, 0.0
    L.check(L.lib().nq_uaq_forward_multi(segs, len(items), _stream()), "uaq_forward_multi")
    return outs


def uaq_backward_multi(items):
    """items: [(x, gy, delta, zp, n_levels)] -> [d(delta)], ONE launch (same arithmetic and summation order as uaq_backward)."""
    segs = (L.AdaSeg * len(items))()
    outs, keep = [], []
    for sg, (x, gy, delta, zp, n_levels) in zip(segs, items):
        x, gy, delta, zp = _dev(x), _dev(gy), _dev(delta), _dev(zp)
        rows, rl, per_row = _rows(x, delta)
        dd = torch.empty_like(delta)
        outs.append(dd)
        keep.append((x, gy))
        sg.x, sg.gy, sg.alpha, sg.delta, sg.zp, sg.out = _p(x), _p(gy), None, _p(delta), _p(zp), _p(dd)
        sg.rows, sg.row_len, sg.per_row, sg.n_levels, sg.soft, sg.reg_weight = rows, rl, per_row, n_levels, 0, 0.0
    L.check(L.lib().nq_uaq_backward_multi(segs, len(items), _stream()), "uaq_backward_multi")
    return outs


def step_prologue(order_tab, scal_tab, step_ctr, cur_idx, cur_scal):
    """nq_step_prologue: cur_idx <- order_tab[*step], cur_scal <- scal_tab[*step], *step += 1 (captured iterations)."""
    L.check(L.lib().nq_step_prologue(_p(order_tab), _p(scal_tab), _p(step_ctr), _p(cur_idx), _p(cur_scal), cur_idx.numel(),
                                     cur_scal.numel(), _stream()), "step_prologue")


def step_prologue_gather(order_tab, scal_tab, step_ctr, cur_idx, cur_scal, table, out):
    """nq_step_prologue_gather: step_prologue + out <- table[order_tab[*step]] in ONE launch (step_ctr: two int32 {counter,
    ticket}, both zero at the start of an epoch)."""
    table = _dev(table)
    if step_ctr.numel() < 2 or out.shape[0] != cur_idx.numel() or out[0].numel() != table[0].numel() or not out.is_contiguous():
        raise RuntimeError("step_prologue_gather: step_ctr needs two ints, out must be (B,) + table.shape[1:], contiguous")
    L.check(L.lib().nq_step_prologue_gather(_p(order_tab), _p(scal_tab), _p(step_ctr), _p(cur_idx), _p(cur_scal), cur_idx.numel(),
                                            cur_scal.numel(), _p(table), table.shape[0], table[0].numel(), _p(out), _stream()),
            "step_prologue_gather")


def adaround_backward_multi(items, reg_b=0.0, dyn=None):
    """items: [(x, gy, alpha, delta, zp, n_levels, reg_weight)] -> [d(alpha)] (+ regulariser gradient where
    reg_weight != 0), ONE launch.  dyn (device floats {reg_b, gate, ...} of the current step) replaces the host reg_b and
    gates reg_weight (captured iterations; same arithmetic)."""
    segs = (L.AdaSeg * len(items))()
    outs = []
    for sg, (x, gy, alpha, delta, zp, n_levels, reg_weight) in zip(segs, items):
        x, gy, alpha, delta, zp = _dev(x), _dev(gy), _dev(alpha), _dev(delta), _dev(zp)
        rows, rl, per_row = _rows(x, delta)
        da = torch.empty_like(alpha)
        outs.append(da)
        sg.x, sg.gy, sg.alpha, sg.delta, sg.zp, sg.out = _p(x), _p(gy), _p(alpha), _p(delta), _p(zp), _p(da)
        sg.rows, sg.row_len, sg.per_row, sg.n_levels, sg.soft, sg.reg_weight = rows, rl, per_row, n_levels, 1, float(reg_weight)
    if dyn is not None:
        L.check(L.lib().nq_adaround_backward_multi_dyn(segs, len(items), _p(dyn), _stream()), "adaround_backward_multi_dyn")
    else:
        L.check(L.lib().nq_adaround_backward_multi(segs, len(items), float(reg_b), _stream()), "adaround_backward_multi")
    return outs


def adaround_adam_multi(items, opt, reg_b=0.0, dyn=None, beta1=0.9, beta2=0.999, eps=1e-8):
    """items: [(x, gy, alpha, delta, zp, n_levels, reg_weight)] in the order of opt.params (= the alphas): d(alpha) (+ the
    regulariser gradient) and opt's Adam step in ONE launch; advances opt.t.  Bit-identical to adaround_backward_multi
    followed by opt.step (tested); d(alpha) itself is not materialised."""
    assert len(items) == len(opt.params)
    opt.t += 1
    segs = (L.AdaAdamSeg * len(items))()
    keep = []
    for i, (sg, (x, gy, alpha, delta, zp, n_levels, reg_weight)) in enumerate(zip(segs, items)):
        x, gy, alpha, delta, zp = _dev(x), _dev(gy), _dev(alpha), _dev(delta), _dev(zp)
        if alpha.data_ptr() != opt.params[i].data_ptr():
            raise RuntimeError("adaround_adam_multi: items must be in the optimiser's parameter order")
        rows, rl, per_row = _rows(x, delta)
        keep.append((x, gy))
        sg.x, sg.gy, sg.alpha, sg.delta, sg.zp, sg.m, sg.v = _p(x), _p(gy), _p(alpha), _p(delta), _p(zp), _p(opt.m[i]), _p(opt.v[i])
        sg.rows, sg.row_len, sg.per_row, sg.n_levels, sg.reg_weight = rows, rl, per_row, n_levels, float(reg_weight)
    step_size, bc2_sqrt = opt.scalars(opt.t, beta1, beta2)
    L.check(L.lib().nq_adaround_adam_multi(segs, len(items), float(reg_b), step_size, beta1, beta2, eps, bc2_sqrt, _p(dyn),
                                           _stream()), "adaround_adam_multi")


class FusedAdam:
    """torch.optim.Adam(params, lr) with default betas/eps; all tensors updated by ONE nq_adam_step_multi launch."""

    def __init__(self, params, lr):
        self.params = [p for p in params]
        self.lr = lr
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def scalars(self, t, beta1=0.9, beta2=0.999):
        """(lr/(1-beta1^t), sqrt(1-beta2^t)) of step t, as the eager path passes them (host doubles -> fp32)."""
        return self.lr / (1 - beta1 ** t), (1 - beta2 ** t) ** 0.5

    def step(self, grads=None, beta1=0.9, beta2=0.999, eps=1e-8, dyn=None):
        """dyn: device floats whose [2], [3] hold scalars(t) of this step (captured iterations)."""
        self.t += 1
        todo = [(p, (p.grad if grads is None else grads[i]), self.m[i], self.v[i]) for i, p in enumerate(self.params)]
        todo = [t for t in todo if t[1] is not None]
        if not todo:
            return
        segs = (L.AdamSeg * len(todo))()
        keep = []
        for sg, (p, g, m, v) in zip(segs, todo):
            g = _dev(g).contiguous()
            keep.append(g)
            sg.p, sg.g, sg.m, sg.v, sg.n = _p(p.data), _p(g), _p(m), _p(v), p.numel()
        if dyn is not None:
            L.check(L.lib().nq_adam_step_multi_dyn(segs, len(todo), _p(dyn), beta1, beta2, eps, _stream()), "adam_step_multi_dyn")
            return
        bc1, bc2 = 1 - beta1 ** self.t, 1 - beta2 ** self.t
        L.check(L.lib().nq_adam_step_multi(segs, len(todo), self.lr / bc1, beta1, beta2, eps, bc2 ** 0.5, _stream()),
                "adam_step_multi")


class _UAQFn(Function):
    @staticmethod
    def forward(ctx, x, delta, zp, n_levels):
        ctx.save_for_backward(x, delta, zp)
        ctx.n_levels = n_levels
        return uaq_forward(x, delta, zp, n_levels)

    @staticmethod
    def backward(ctx, gy):
        x, delta, zp = ctx.saved_tensors
        # round_ste (quantizer.py:53-57): d/dx = gy inside the clamp range.  Only computed when somebody asks for it (a
        # trainable quantiser input); the calibration engine never does -- the FP weight is not optimised
        if ctx.needs_input_grad[0]:
            dd, dx = uaq_backward(x, gy, delta, zp, ctx.n_levels, want_dx=True)
            return dx, dd.view_as(delta), None, None
        return None, uaq_backward(x, gy, delta, zp, ctx.n_levels).view_as(delta), None, None


class _AdaRoundFn(Function):
    @staticmethod
    def forward(ctx, x, alpha, delta, zp, n_levels, soft):
        y, xq = adaround_forward(x, alpha, delta, zp, n_levels, soft, want_xq=True)
        ctx.save_for_backward(x, alpha, delta, zp)
        ctx.n_levels, ctx.soft = n_levels, soft
        ctx.mark_non_differentiable(xq)
        return y, xq

    @staticmethod
    def backward(ctx, gy, _gxq):
        x, alpha, delta, zp = ctx.saved_tensors
        # x enters through floor(x/delta) (quantizer.py:290), whose gradient is zero: None IS the reference's gradient
        # for x.  d(delta) exists in the reference's graph but no optimiser ever steps it in phase 2 (calib_model.py:
        # 186-195 collects alpha only); it is not computed here -- ask loudly rather than return silent zeros
        if ctx.needs_input_grad[2] and os.environ.get("NQ_STRICT_GRADS"):
            raise NotImplementedError("d(delta) of the AdaRound fake-quant is not built (never stepped by the reference)")
        da = adaround_backward(x, gy, alpha, delta, zp, ctx.n_levels) if ctx.soft else None
        return None, da, None, None, None, None


class _RoundLossFn(Function):
    @staticmethod
    def forward(ctx, alpha, b, weight):
        ctx.save_for_backward(alpha)
        ctx.b, ctx.weight = float(b), float(weight)
        return round_loss(alpha, b, weight)

    @staticmethod
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        da = torch.empty_like(alpha)
        L.check(L.lib().nq_round_loss_backward(_p(alpha), alpha.numel(), ctx.b, ctx.weight, _p(_dev(g)), _p(da), 0, _stream()),
                "round_loss_backward")
        return da, None, None


def uaq_fake_quant(x, delta, zp, n_levels):
    return _UAQFn.apply(x, delta, zp, n_levels)


def adaround_fake_quant(x, alpha, delta, zp, n_levels, soft):
    return _AdaRoundFn.apply(x, alpha, delta, zp, n_levels, soft)


def round_regulariser(alpha, b, weight):
    return _RoundLossFn.apply(alpha, b, weight)


# ------------------------------------------------------------------------------------------ Hadamard
def next_pow2(n: int) -> int:
    return 1 if n == 0 else 2 ** math.ceil(math.log2(n))


def fwht_channels(w: torch.Tensor, n: int, n_out: int) -> torch.Tensor:
    """WHT of length n along dim 1 of (C_out, C, KH, KW): input zero-padded from C to n, first n_out kept."""
    w = _dev(w)
    co, c, kh, kw = w.shape
    y = torch.empty((co, n_out, kh, kw), device=w.device, dtype=torch.float32)
    L.check(L.lib().nq_fwht(_p(w), _p(y), co, n, kh * kw, c, n_out, _stream()), "fwht")
    return y


def fwht_channels_multi(items):
    """items: [(w, n, n_out)] -> [fwht_channels(w, n, n_out)], ONE launch for all tensors (bit-identical results)."""
    segs = (L.FwhtSeg * len(items))()
    outs = []
    for sg, (w, n, n_out) in zip(segs, items):
        w = _dev(w)
        co, c, kh, kw = w.shape
        y = torch.empty((co, n_out, kh, kw), device=w.device, dtype=torch.float32)
        sg.x, sg.y, sg.outer, sg.inner, sg.n, sg.n_in, sg.n_out = _p(w), _p(y), co, kh * kw, n, c, n_out
        outs.append(y)
    if items:
        L.check(L.lib().nq_fwht_multi(segs, len(items), _stream()), "fwht_multi")
    return outs


def _fq_fwht_segs(items, bwd):
    """items: [(x, alpha, delta, zp, n_levels, soft, n, c_in, reg_weight, gy, m, v)] -> (FqFwhtSeg array, outputs, keep-alive).
    n == 0: a plain tensor (bias); else x / alpha are (C_out, n, kh, kw) and the spatial side is (C_out, c_in, kh, kw)."""
    segs = (L.FqFwhtSeg * len(items))()
    outs, keep = [], []
    for sg, (x, alpha, delta, zp, n_levels, soft, n, c_in, reg_weight, gy, m, v) in zip(segs, items):
        x, alpha, delta, zp = _dev(x), _dev(alpha), _dev(delta), _dev(zp)
        sg.x, sg.alpha, sg.delta, sg.zp = _p(x), _p(alpha), _p(delta), _p(zp)
        sg.n_levels, sg.soft, sg.reg_weight = n_levels, int(bool(soft)), float(reg_weight)
        if n:
            co, npad, kh, kw = x.shape
            if npad != n or delta.numel() not in (1, co):
                raise RuntimeError("fused fake-quant + FWHT: x must be (C_out, n, kh, kw) with per-channel or scalar delta")
            sg.outer, sg.inner, sg.n, sg.c_in, sg.per_row = co, kh * kw, n, c_in, int(delta.numel() > 1)
            spatial = (co, c_in, kh, kw)
        else:
            if delta.numel() != 1:
                raise RuntimeError("fused fake-quant + FWHT: a plain tensor takes one scalar delta / zero point")
            sg.outer, sg.inner, sg.n, sg.c_in, sg.per_row = x.numel(), 1, 0, 0, 0
            spatial = tuple(x.shape)
        if bwd:
            gy = _dev(gy)
            if tuple(gy.shape) != spatial:
                raise RuntimeError(f"fused FWHT + d(alpha) + Adam: gradient shape {tuple(gy.shape)} != {spatial}")
            sg.gy, sg.m, sg.v = _p(gy), _p(m), _p(v)
            keep.append((x, gy))
        else:
            y = torch.empty(spatial, device=x.device, dtype=torch.float32)
            sg.y = _p(y)
            outs.append(y)
            keep.append(x)
    return segs, outs, keep


def adaround_fwht_multi(items):
    """items: [(x, alpha, delta, zp, n_levels, soft, n, c_in)] -> [H(Q(x))[:, :c_in]] (n == 0: Q(x) of a bias), ONE launch
    (nq_adaround_fwht_multi): bit-identical to adaround_forward_multi followed by fwht_channels_multi."""
    segs, outs, keep = _fq_fwht_segs([it + (0.0, None, None, None) for it in items], bwd=False)
    L.check(L.lib().nq_adaround_fwht_multi(segs, len(items), _stream()), "adaround_fwht_multi")
    return outs


def fwht_adaround_adam_multi(items, opt, reg_b=0.0, dyn=None, beta1=0.9, beta2=0.999, eps=1e-8):
    """items: [(x, gy, alpha, delta, zp, n_levels, reg_weight, n, c_in)] in the order of opt.params (= the alphas); gy is the
    SPATIAL-domain gradient d(loss)/d(W^) (C_out, c_in, kh, kw) (n == 0: the bias gradient).  H on the zero-padded gradient,
    d(alpha) (+ regulariser gradient) and opt's Adam step in ONE launch; advances opt.t.  Bit-identical to
    fwht_channels_multi + adaround_adam_multi."""
    assert len(items) == len(opt.params)
    opt.t += 1
    full = []
    for i, (x, gy, alpha, delta, zp, n_levels, reg_weight, n, c_in) in enumerate(items):
        if alpha.data_ptr() != opt.params[i].data_ptr():
            raise RuntimeError("fwht_adaround_adam_multi: items must be in the optimiser's parameter order")
        full.append((x, alpha, delta, zp, n_levels, True, n, c_in, reg_weight, gy, opt.m[i], opt.v[i]))
    segs, _, keep = _fq_fwht_segs(full, bwd=True)
    step_size, bc2_sqrt = opt.scalars(opt.t, beta1, beta2)
    L.check(L.lib().nq_fwht_adaround_adam_multi(segs, len(items), float(reg_b), step_size, beta1, beta2, eps, bc2_sqrt, _p(dyn),
                                                _stream()), "fwht_adaround_adam_multi")


def fq_fwht_fusable(n, inner):
    """rows short enough for the fused launches' LDS tile (every layer of the shipped models: n <= 256, 3 x 3 kernels)"""
    return n * inner <= 8192 and n <= 1024


class _HadamardFn(Function):
    @staticmethod
    def forward(ctx, w, n, n_out):
        ctx.n, ctx.c_in = n, w.shape[1]
        return fwht_channels(w, n, n_out)

    @staticmethod
    def backward(ctx, g):
        return fwht_channels(g, ctx.n, ctx.c_in), None, None


def hadamard_along_channel_weight(x: torch.Tensor, n_out=None) -> torch.Tensor:
    """reference quant_layer.py:16-22 (differentiable; n_out folds the [:, :C] slice of :71 into the store)."""
    n = next_pow2(x.shape[1])
    if n != x.shape[1] and n_out is None:
        raise ValueError("channel count must be a power of two (pad first, quant_layer.py:45-49)")
    return _HadamardFn.apply(x, n, x.shape[1] if n_out is None else n_out)


def hadamard_weight_of(w: torch.Tensor) -> torch.Tensor:
    """zero-pad C_in to 2^k, then transform (reference quant_layer.py:45-49)."""
    n = next_pow2(w.shape[1])
    return fwht_channels(w.detach(), n, n)


# ------------------------------------------------------------------------------------------ convolution
@functools.lru_cache(maxsize=None)
def conv_operand_dims(cin, cout, k):
    kr, ld = ctypes.c_int(0), ctypes.c_int(0)
    L.check(L.lib().nq_conv_operand_dims(cin, cout, k, ctypes.byref(kr), ctypes.byref(ld)), "conv_operand_dims")
    return kr.value, ld.value


def weight_layouts(w, need_bwd):
    """-> (wt_fwd, dims_fwd, wt_bwd | None, dims_bwd | None) GEMM operands of an OIHW weight."""
    w = _dev(w)
    cout, cin, k, _ = w.shape
    kf, lf = conv_operand_dims(cin, cout, k)
    wt = torch.empty(kf * lf, device=w.device, dtype=torch.float32)
    wb, kb, lb = None, 0, 0
    if need_bwd:
        kb, lb = conv_operand_dims(cout, cin, k)
        wb = torch.empty(kb * lb, device=w.device, dtype=torch.float32)
    L.check(L.lib().nq_weight_layouts(_p(w), _p(wt), _p(wb), cout, cin, k, kf, lf, kb, lb, _stream()), "weight_layouts")
    return wt, (kf, lf), wb, (kb, lb)


def weight_layouts_multi(items):
    """items: [(w, need_fwd, need_bwd)] -> [(wt_fwd | None, dims_fwd, wt_bwd | None, dims_bwd)] as weight_layouts, ONE launch."""
    segs = (L.WLSeg * len(items))()
    outs = []
    for sg, (w, need_fwd, need_bwd) in zip(segs, items):
        w = _dev(w)
        cout, cin, k, _ = w.shape
        wt = wb = None
        kf = lf = kb = lb = 0
        if need_fwd:
            kf, lf = conv_operand_dims(cin, cout, k)
            wt = torch.empty(kf * lf, device=w.device, dtype=torch.float32)
        if need_bwd:
            kb, lb = conv_operand_dims(cout, cin, k)
            wb = torch.empty(kb * lb, device=w.device, dtype=torch.float32)
        sg.w, sg.wt_fwd, sg.wt_bwd, sg.Cout, sg.Cin, sg.k = _p(w), _p(wt), _p(wb), cout, cin, k
        sg.krows_fwd, sg.ld_fwd, sg.krows_bwd, sg.ld_bwd = kf, lf, kb, lb
        outs.append((wt, (kf, lf), wb, (kb, lb)))
    if items:
        L.check(L.lib().nq_weight_layouts_multi(segs, len(items), _stream()), "weight_layouts_multi")
    return outs


def weight_layouts_all(items3, items_f):
    """weight_layout3_multi(items3) and weight_layouts_multi(items_f) in ONE launch (nq_weight_layouts_all; same bytes)."""
    if not items3 or not items_f or os.environ.get("NQ_LAYOUTS_ALL", "1") == "0":
        return weight_layout3_multi(items3), weight_layouts_multi(items_f)
    segs3 = (L.WL3Seg * len(items3))()
    outs3 = []
    for sg, (w, transposed) in zip(segs3, items3):
        w = _dev(w)
        cw_out, cw_in, k, _ = w.shape
        cin, cout = (cw_out, cw_in) if transposed else (cw_in, cw_out)
        buf = torch.empty(_q("nq_conv3_weight_bytes", cin, cout, k), device=w.device, dtype=torch.uint8)
        outs3.append(buf)
        sg.w, sg.wt3, sg.Cin, sg.Cout, sg.k, sg.transposed = _p(w), _p(buf), cin, cout, k, int(bool(transposed))
    segsf = (L.WLSeg * len(items_f))()
    outsf = []
    for sg, (w, need_fwd, need_bwd) in zip(segsf, items_f):
        w = _dev(w)
        cout, cin, k, _ = w.shape
        wt = wb = None
        kf = lf = kb = lb = 0
        if need_fwd:
            kf, lf = conv_operand_dims(cin, cout, k)
            wt = torch.empty(kf * lf, device=w.device, dtype=torch.float32)
        if need_bwd:
            kb, lb = conv_operand_dims(cout, cin, k)
            wb = torch.empty(kb * lb, device=w.device, dtype=torch.float32)
        sg.w, sg.wt_fwd, sg.wt_bwd, sg.Cout, sg.Cin, sg.k = _p(w), _p(wt), _p(wb), cout, cin, k
        sg.krows_fwd, sg.ld_fwd, sg.krows_bwd, sg.ld_bwd = kf, lf, kb, lb
        outsf.append((wt, (kf, lf), wb, (kb, lb)))
    L.check(L.lib().nq_weight_layouts_all(segs3, len(items3), segsf, len(items_f), _stream()), "weight_layouts_all")
    return outs3, outsf


def conv_forward_raw(x, wt, dims, bias, cout, k, epilogue, r, in_gelu=False, zprev=None, fmt=0):
    """One nq_conv_forward launch.  Returns (y, z): z = shuffled pre-activation for the PixelShuffle epilogues
    (y is None for EPI_PS), y = un-shuffled gradient for EPI_DGRAD_GELU.  fmt = EPI_Y_SPLIT: y as split {hi | lo} words
    (conv_split_out says where)."""
    B, cin, H, W = x.shape
    y = z = None
    if epilogue in (EPI_PS_GELU, EPI_PS):
        z = torch.empty((B, cout // (r * r), H * r, W * r), device=x.device, dtype=torch.float32)
        if epilogue == EPI_PS_GELU:
            y = torch.empty_like(z)
    elif epilogue == EPI_DGRAD_GELU:
        y = torch.empty((B, cout * r * r, H // r, W // r), device=x.device, dtype=torch.float32)
    else:
        y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    nws = _q("nq_conv_forward_ws_floats", B, cin, H, W, cout, k)
    ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
    _timed(("conv_igemm", k, cin, cout, H, W, B, epilogue),
           lambda: L.check(L.lib().nq_conv_forward(_p(x), _p(wt), _p(bias), _p(y), _p(z), _p(ws), B, cin, H, W, cout, k,
                                                   dims[0], dims[1], r, epilogue | fmt, 1 if in_gelu else 0, _p(zprev),
                                                   _stream()), "conv_forward"))
    return y, z


def conv3_supported(B, cin, H, W, cout, k):
    return bool(_q("nq_conv3_supported", B, cin, H, W, cout, k))


def conv3_split_io(B, cin, H, W, cout, k):
    """EPI_X_SPLIT | EPI_Y_SPLIT bits: the sides of conv3_forward_raw that may travel as split {hi | lo} words for this shape"""
    return int(_q("nq_conv3_split_io", B, cin, H, W, cout, k))


def conv_wgrad3_split_io(B, cin, H, W, cout, k):
    """bit 0: x, bit 1: dy of conv_wgrad3_raw may be split words for this shape"""
    return int(_q("nq_conv_wgrad3_split_io", B, cin, H, W, cout, k))


def conv_split_out(B, cin, H, W, cout, k, r, epilogue, has_bias=False):
    """True when conv_forward_raw can write y as split words for this call (the streaming head data gradient)"""
    return bool(_q("nq_conv_split_out", B, cin, H, W, cout, k, r, epilogue, 0, 1 if has_bias else 0))


def split_words(x):
    """float tensor -> the same shape holding split {hi | lo} words (bit patterns in a float32 tensor)"""
    x = _dev(x).contiguous()
    y = torch.empty_like(x)
    L.check(L.lib().nq_split_words(_p(x), _p(y), x.numel(), _stream()), "split_words")
    return y


def weight_layout3(w, transposed=False):
    """pre-split (bf16 hi/lo) operand of the bf16x3 conv kernels; transposed=True -> operand of the data gradient."""
    w = _dev(w)
    cw_out, cw_in, k, _ = w.shape
    cin, cout = (cw_out, cw_in) if transposed else (cw_in, cw_out)
    buf = torch.empty(_q("nq_conv3_weight_bytes", cin, cout, k), device=w.device, dtype=torch.uint8)
    L.check(L.lib().nq_weight_layout3(_p(w), _p(buf), cin, cout, k, 1 if transposed else 0, _stream()), "weight_layout3")
    return buf


def weight_layout3_multi(items):
    """items: [(w, transposed)] -> [operand buffers], ONE launch (same bytes as weight_layout3 per item)."""
    segs = (L.WL3Seg * len(items))()
    outs = []
    for sg, (w, transposed) in zip(segs, items):
        w = _dev(w)
        cw_out, cw_in, k, _ = w.shape
        cin, cout = (cw_out, cw_in) if transposed else (cw_in, cw_out)
        buf = torch.empty(_q("nq_conv3_weight_bytes", cin, cout, k), device=w.device, dtype=torch.uint8)
        outs.append(buf)
        sg.w, sg.wt3, sg.Cin, sg.Cout, sg.k, sg.transposed = _p(w), _p(buf), cin, cout, k, int(bool(transposed))
    if items:
        L.check(L.lib().nq_weight_layout3_multi(segs, len(items), _stream()), "weight_layout3_multi")
    return outs


def conv3_forward_raw(x, wt3, bias, cout, k, epilogue, r, zprev=None, fmt=0):
    """bf16x3 counterpart of conv_forward_raw (same outputs).  fmt: EPI_X_SPLIT (x holds split words) | EPI_Y_SPLIT (y is
    written as split words), where conv3_split_io allows."""
    B, cin, H, W = x.shape
    y = z = None
    if epilogue in (EPI_PS_GELU, EPI_PS):
        z = torch.empty((B, cout // (r * r), H * r, W * r), device=x.device, dtype=torch.float32)
        if epilogue == EPI_PS_GELU:
            y = torch.empty_like(z)
    elif epilogue == EPI_DGRAD_GELU:
        y = torch.empty((B, cout * r * r, H // r, W // r), device=x.device, dtype=torch.float32)
    else:
        y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    nws = _q("nq_conv_forward3_ws_floats", B, cin, H, W, cout, k)
    ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
    _timed(("conv_igemm3", k, cin, cout, H, W, B, epilogue),
           lambda: L.check(L.lib().nq_conv_forward3(_p(x), _p(wt3), _p(bias), _p(y), _p(z), _p(zprev), _p(ws), B, cin, H, W,
                                                    cout, k, r, epilogue | fmt, _stream()), "conv_forward3"))
    return y, z


def conv_wgrad3_supported(B, cin, H, W, cout, k):
    return bool(_q("nq_conv_wgrad3_supported", B, cin, H, W, cout, k))


class PendingReductions:
    """Slab reductions of several weight-gradient launches, performed by ONE nq_wgrad_reduce_multi launch at `flush()`
    (bit-identical to the per-launch reductions; the slabs and outputs stay referenced until then)."""

    def __init__(self):
        self.segs, self.keep = [], []

    def add(self, seg, *tensors):
        self.segs.append(seg)
        self.keep.append(tensors)

    def flush(self):
        if self.segs:
            arr = (L.WgrSeg * len(self.segs))(*self.segs)
            L.check(L.lib().nq_wgrad_reduce_multi(arr, len(self.segs), _stream()), "wgrad_reduce_multi")
        self.segs, self.keep = [], []


def conv_wgrad3_raw(x, dy, cout, k, want_db, out=None, defer=None, fmt=0):
    """bf16x3 counterpart of conv_wgrad_raw.  out = (dw, db) pre-allocated contiguous outputs (views of a flat
    gradient arena) or None.  defer (PendingReductions): only the split kernel runs now, dw / db are valid after
    defer.flush().  fmt bit 0 / 1: x / dy hold split {hi | lo} words (conv_wgrad3_split_io)."""
    B, cin, H, W = x.shape
    ws = torch.empty(_q("nq_conv_wgrad3_ws_floats", B, cin, H, W, cout, k), device=x.device, dtype=torch.float32)
    dw = torch.empty((cout, cin, k, k), device=x.device, dtype=torch.float32) if out is None else out[0]
    db = (torch.empty(cout, device=x.device, dtype=torch.float32) if out is None else out[1]) if want_db else None
    if defer is not None:
        seg = L.WgrSeg()
        _timed(("conv_wgrad3", k, cin, cout, H, W, B, 0),
               lambda: L.check(L.lib().nq_conv_wgrad3_slabs_fmt(_p(x), _p(dy), _p(dw), _p(db), _p(ws), B, cin, H, W, cout, k,
                                                                ctypes.byref(seg), fmt, _stream()), "conv_wgrad3_slabs"))
        defer.add(seg, ws, dw, db)
        return dw, db
    _timed(("conv_wgrad3", k, cin, cout, H, W, B, 0),
           lambda: L.check(L.lib().nq_conv_wgrad3_fmt(_p(x), _p(dy), _p(dw), _p(db), _p(ws), B, cin, H, W, cout, k, fmt,
                                                      _stream()), "conv_wgrad3"))
    return dw, db


def channel_sum(x):
    x = _dev(x)
    B, C = x.shape[0], x.shape[1]
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    ws = torch.empty(512 * C, device=x.device, dtype=torch.float32)
    L.check(L.lib().nq_channel_sum(_p(x), _p(out), _p(ws), B, C, x.numel() // (B * C), _stream()), "channel_sum")
    return out


def conv_wgrad_swapped3(x, dy, cout, k, want_db, out=None, defer=None):
    """Weight gradient of a conv with very few OUTPUT channels (the 3-channel head) by swapping operand roles:
    R[ci][(co,kh,kw)] = sum_p x[ci][p] * dy[co][p + tap]  is the weight gradient of the conv  dy -> x-channels, and
    dW[co][ci][kh][kw] = R[ci][co][K-1-kh][K-1-kw].  The big tensor (x, 242 MB) is then the un-shifted GEMM operand
    read exactly once, only the 3-channel dy needs halo rows, and the MFMA tile is 37(->48) x 27(->64) instead of
    3(->16) x 333(->384)."""
    B, cin, H, W = x.shape
    ws = torch.empty(_q("nq_conv_wgrad3_ws_floats", B, cout, H, W, cin, k), device=x.device, dtype=torch.float32)
    dw = torch.empty((cout, cin, k, k), device=x.device, dtype=torch.float32) if out is None else out[0]
    # the slab reduction writes dW[co][ci][K-1-kh][K-1-kw] directly (no permute / flip / copy passes)
    if defer is not None:
        seg = L.WgrSeg()
        _timed(("conv_wgrad3", k, cout, cin, H, W, B, 0),
               lambda: L.check(L.lib().nq_conv_wgrad3_swapped_slabs(_p(x), _p(dy), _p(dw), _p(ws), B, cin, H, W, cout, k,
                                                                    ctypes.byref(seg), _stream()), "conv_wgrad3_swapped_slabs"))
        defer.add(seg, ws, dw)
    else:
        _timed(("conv_wgrad3", k, cout, cin, H, W, B, 0),
               lambda: L.check(L.lib().nq_conv_wgrad3_swapped(_p(x), _p(dy), _p(dw), _p(ws), B, cin, H, W, cout, k, _stream()),
                               "conv_wgrad3_swapped"))
    db = None
    if want_db:
        db = channel_sum(dy)
        if out is not None:
            out[1].copy_(db)
            db = out[1]
    return dw, db


def conv_wgrad_raw(x, dy, cout, k, want_db, x_gelu=False, out=None, defer=None):
    B, cin, H, W = x.shape
    ws = torch.empty(_q("nq_conv_wgrad_ws_floats", B, cin, H, W, cout, k), device=x.device, dtype=torch.float32)
    dw = torch.empty((cout, cin, k, k), device=x.device, dtype=torch.float32) if out is None else out[0]
    db = (torch.empty(cout, device=x.device, dtype=torch.float32) if out is None else out[1]) if want_db else None
    if defer is not None:
        seg = L.WgrSeg()
        _timed(("conv_wgrad", k, cin, cout, H, W, B, 0),
               lambda: L.check(L.lib().nq_conv_wgrad_slabs(_p(x), _p(dy), _p(dw), _p(db), _p(ws), B, cin, H, W, cout, k,
                                                           1 if x_gelu else 0, ctypes.byref(seg), _stream()), "conv_wgrad_slabs"))
        defer.add(seg, ws, dw, db)
        return dw, db
    _timed(("conv_wgrad", k, cin, cout, H, W, B, 0),
           lambda: L.check(L.lib().nq_conv_wgrad(_p(x), _p(dy), _p(dw), _p(db), _p(ws), B, cin, H, W, cout, k,
                                                 1 if x_gelu else 0, _stream()), "conv_wgrad"))
    return dw, db


class _ConvFn(Function):
    """stride-1 'same' conv (+bias) fused with PixelShuffle+GELU or tanh*0.5+0.5 (reference quant_layer.py:80,
    quant_block.py:31-35, models/_layers.py:10-36)."""

    @staticmethod
    def forward(ctx, x, w, b, epilogue, r):
        x, w = _dev(x, "x"), _dev(w, "weight")
        b = _dev(b, "bias") if b is not None else None
        cout, cin, k, k2 = w.shape
        if k != k2 or x.shape[1] != cin:
            raise ValueError(f"conv shape mismatch: x {tuple(x.shape)} w {tuple(w.shape)}")
        need_dx = ctx.needs_input_grad[0]
        wt, dims, wb, dims_b = weight_layouts(w, need_dx)
        y, z = conv_forward_raw(x, wt, dims, b, cout, k, epilogue, r)
        ctx.save_for_backward(x, wb, z if epilogue == EPI_PS_GELU else (y if epilogue == EPI_TANH else None))
        ctx.meta = (cout, cin, k, epilogue, r, dims_b, b is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wb, aux = ctx.saved_tensors
        cout, cin, k, epilogue, r, dims_b, has_b = ctx.meta
        B, _, H, W = x.shape
        gy = _dev(gy, "grad")
        if epilogue == EPI_PS_GELU:
            dconv = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
            L.check(L.lib().nq_ps_gelu_backward(_p(gy), _p(aux), _p(dconv), B, cout // (r * r), H, W, r, _stream()),
                    "ps_gelu_backward")
        elif epilogue == EPI_TANH:
            dconv = torch.empty_like(gy)
            L.check(L.lib().nq_tanh_out_backward(_p(gy), _p(aux), _p(dconv), gy.numel(), _stream()), "tanh_backward")
        else:
            dconv = gy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx, _ = conv_forward_raw(dconv, wb, dims_b, None, cin, k, EPI_PLAIN, 1)
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            dw, db = conv_wgrad_raw(x, dconv, cout, k, has_b)
        return dx, dw, db, None, None


# ---- twice-differentiable convolution (Hessian-vector products of the Omega bit-allocation criterion) ----
# A stride-1 'same' convolution is bilinear in (x, w), so F = conv(x, w), D = dgrad(gy, w) and Wg = wgrad(x, gy) are
# closed under differentiation:  dF = (D(gy,w), Wg(x,gy));  dD = (F(g,w), Wg(g,gy));  dWg = (D(gy,G), F(x,G)).
# Each is an autograd Function whose backward is built from the other two, so create_graph=True works to any order
# on the same HIP kernels as the calibration loop (bf16x3 where the grid fills the chip, exact fp32 MFMA otherwise).
_PRECISION = None   # set_conv_precision() override; None -> NQ_CONV_PRECISION (default 'bf16x3')


def set_conv_precision(precision):
    """'bf16x3' | 'fp32' | None (= environment NQ_CONV_PRECISION, default 'bf16x3'): which MFMA pipe the large
    convolutions run on from now on.  bench.py / the precision gate switch it between runs of one process."""
    global _PRECISION
    if precision not in (None, "fp32", "bf16x3"):
        raise ValueError(f"unknown conv precision {precision!r}")
    _PRECISION = precision


def get_conv_precision():
    return _PRECISION or os.environ.get("NQ_CONV_PRECISION", "bf16x3")


def _use3(precision):
    return (precision or get_conv_precision()) == "bf16x3"


def _conv_plain(x, w, precision=None):
    cout, cin, k, _ = w.shape
    B, _, H, W = x.shape
    if _use3(precision) and conv3_supported(B, cin, H, W, cout, k):
        return conv3_forward_raw(x, weight_layout3(w), None, cout, k, EPI_PLAIN, 1)[0]
    wt, dims, _, _ = weight_layouts(w, need_bwd=False)
    return conv_forward_raw(x, wt, dims, None, cout, k, EPI_PLAIN, 1)[0]


def _dgrad_plain(gy, w, precision=None):
    cout, cin, k, _ = w.shape
    B, _, H, W = gy.shape
    if _use3(precision) and conv3_supported(B, cout, H, W, cin, k):
        return conv3_forward_raw(gy, weight_layout3(w, transposed=True), None, cin, k, EPI_PLAIN, 1)[0]
    _, _, wb, dims_b = weight_layouts(w, need_bwd=True)
    return conv_forward_raw(gy, wb, dims_b, None, cin, k, EPI_PLAIN, 1)[0]


def _wgrad_plain(x, gy, k, precision=None):
    B, cin, H, W = x.shape
    cout = gy.shape[1]
    if _use3(precision) and conv_wgrad3_supported(B, cin, H, W, cout, k):
        return conv_wgrad3_raw(x, gy, cout, k, False)[0]
    if _use3(precision) and cout <= 4 and cin > 4 and cout * k * k <= 64 and conv_wgrad3_supported(B, cout, H, W, cin, k):
        return conv_wgrad_swapped3(x, gy, cout, k, False)[0]
    return conv_wgrad_raw(x, gy, cout, k, False)[0]


class _ConvF(Function):
    @staticmethod
    def forward(ctx, x, w):
        x, w = _dev(x, "x").contiguous(), _dev(w, "weight").contiguous()
        ctx.save_for_backward(x, w)
        return _conv_plain(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        return (_ConvD.apply(gy, w) if ctx.needs_input_grad[0] else None,
                _ConvWg.apply(x, gy, w.shape[-1]) if ctx.needs_input_grad[1] else None)


class _ConvD(Function):
    @staticmethod
    def forward(ctx, gy, w):
        gy, w = _dev(gy, "gy").contiguous(), _dev(w, "weight").contiguous()
        ctx.save_for_backward(gy, w)
        return _dgrad_plain(gy, w)

    @staticmethod
    def backward(ctx, g):
        gy, w = ctx.saved_tensors
        g = g.contiguous()
        return (_ConvF.apply(g, w) if ctx.needs_input_grad[0] else None,
                _ConvWg.apply(g, gy, w.shape[-1]) if ctx.needs_input_grad[1] else None)


class _ConvWg(Function):
    @staticmethod
    def forward(ctx, x, gy, k):
        x, gy = _dev(x, "x").contiguous(), _dev(gy, "gy").contiguous()
        ctx.save_for_backward(x, gy)
        return _wgrad_plain(x, gy, k)

    @staticmethod
    def backward(ctx, G):
        x, gy = ctx.saved_tensors
        G = G.contiguous()
        return (_ConvD.apply(gy, G) if ctx.needs_input_grad[0] else None,
                _ConvF.apply(x, G) if ctx.needs_input_grad[1] else None, None)


# ---- the elementwise pieces of the decoder, twice differentiable on this repo's own kernels (round 4: they went through
#      ATen before).  Each Function's backward is built from Functions of the same family, so
#      torch.autograd.grad(create_graph=True) followed by .backward() -- the reference's Hessian-vector recipe,
#      methods/bit_assign.py:88-114 -- never leaves libnqhip. ----
_ACT_BASE = {"gelu": 0, "tanh": 3}


def act_dd_raw(x, kind, order, g=None, g2=None):
    """f^(order)(x) * g * g2 (nq_act_dd): kind 'gelu' = exact-erf GELU (nn.GELU(), _layers.py:104-105), 'tanh' = OutImg's
    tanh(x)*0.5+0.5 (_layers.py:10-16); order 0 / 1 / 2 = value, first and second derivative."""
    x = _dev(x, "x")
    g = _dev(g, "g") if g is not None else None
    g2 = _dev(g2, "g2") if g2 is not None else None
    y = torch.empty_like(x)
    L.check(L.lib().nq_act_dd(_p(x), _p(g), _p(g2), _p(y), x.numel(), _ACT_BASE[kind] + order, _stream()), "act_dd")
    return y


class _ActDD(Function):
    """y = f^(order)(x) * g  (g may be None).  d/dx = gy * g * f^(order+1)(x), d/dg = gy * f^(order)(x): the same Function one
    order up / without g, so the graph of a backward pass is itself differentiable (up to the second derivative: the
    Hessian-vector products need no more; a third is refused loudly)."""

    @staticmethod
    def forward(ctx, x, g, kind, order):
        ctx.save_for_backward(x, g)
        ctx.kind, ctx.order = kind, order
        return act_dd_raw(x, kind, order, g)

    @staticmethod
    def backward(ctx, gy):
        x, g = ctx.saved_tensors
        gy = gy.contiguous()
        dx = dg = None
        if ctx.needs_input_grad[0]:
            if ctx.order >= 2:
                raise NotImplementedError("third derivative of the activation is not built (Hessian-vector products need two)")
            dx = _ActDD2.apply(x, gy, g, ctx.kind, ctx.order + 1)
        if g is not None and ctx.needs_input_grad[1]:
            dg = _ActDD.apply(x, gy, ctx.kind, ctx.order)
        return dx, dg, None, None


class _ActDD2(Function):
    """y = f^(order)(x) * g * h (h may be None): the d/dx of _ActDD with its two multipliers kept apart (no elementwise product
    outside the kernel).  Differentiable w.r.t. g and h (what the second backward pass asks for) and, while order < 2, x."""

    @staticmethod
    def forward(ctx, x, g, h, kind, order):
        ctx.save_for_backward(x, g, h)
        ctx.kind, ctx.order = kind, order
        return act_dd_raw(x, kind, order, g, h)

    @staticmethod
    def backward(ctx, gy):
        x, g, h = ctx.saved_tensors
        gy = gy.contiguous()
        dx = dg = dh = None
        if ctx.needs_input_grad[0]:
            if ctx.order >= 2 or h is not None:
                raise NotImplementedError("third derivative of the activation is not built (Hessian-vector products need two)")
            dx = _ActDD2.apply(x, gy, g, ctx.kind, ctx.order + 1)
        if ctx.needs_input_grad[1]:
            dg = _ActDD2.apply(x, gy, h, ctx.kind, ctx.order) if h is not None else _ActDD.apply(x, gy, ctx.kind, ctx.order)
        if h is not None and ctx.needs_input_grad[2]:
            dh = _ActDD2.apply(x, gy, g, ctx.kind, ctx.order)
        return dx, dg, dh, None, None


def gelu_dd(x):
    """exact-erf GELU, twice differentiable (reference nn.GELU(), models/_layers.py:104-105)."""
    return _ActDD.apply(x.contiguous(), None, "gelu", 0)


def tanh_out_dd(x):
    """OutImg: tanh(x) * 0.5 + 0.5, twice differentiable (reference models/_layers.py:10-16)."""
    return _ActDD.apply(x.contiguous(), None, "tanh", 0)


def pixel_shuffle_raw(x, r, inverse=False):
    x = _dev(x, "x")
    B, Cx, Hx, Wx = x.shape
    if inverse:
        if Hx % r or Wx % r:
            raise ValueError("pixel_unshuffle: spatial size not divisible by r")
        C, H, W = Cx, Hx // r, Wx // r
        y = torch.empty((B, C * r * r, H, W), device=x.device, dtype=torch.float32)
    else:
        if Cx % (r * r):
            raise ValueError("pixel_shuffle: channels not divisible by r*r")
        C, H, W = Cx // (r * r), Hx, Wx
        y = torch.empty((B, C, H * r, W * r), device=x.device, dtype=torch.float32)
    L.check(L.lib().nq_pixel_shuffle(_p(x), _p(y), B, C, H, W, r, 1 if inverse else 0, _stream()), "pixel_shuffle")
    return y


class _PixelShuffleDD(Function):
    """PixelShuffle(r) / its inverse: a permutation, so the backward is the inverse permutation of the same family (any order)."""

    @staticmethod
    def forward(ctx, x, r, inverse):
        ctx.r, ctx.inverse = r, inverse
        return pixel_shuffle_raw(x.contiguous(), r, inverse)

    @staticmethod
    def backward(ctx, gy):
        return _PixelShuffleDD.apply(gy.contiguous(), ctx.r, not ctx.inverse), None, None


def pixel_shuffle_dd(x, r):
    """torch.nn.PixelShuffle(r) (reference models/_layers.py:20-36), differentiable to any order."""
    return _PixelShuffleDD.apply(x, r, False)


class _BiasAddDD(Function):
    """y = x + bias[None, :, None, None].  d/dx = gy, d/dbias = channel sums of gy -- a Function whose backward is the
    broadcast again: linear, differentiable to any order."""

    @staticmethod
    def forward(ctx, x, bias):
        x, bias = _dev(x, "x").contiguous(), _dev(bias, "bias").contiguous()
        y = torch.empty_like(x)
        B, C = x.shape[0], x.shape[1]
        L.check(L.lib().nq_bias_add(_p(x), _p(bias), _p(y), B, C, x.numel() // (B * C), _stream()), "bias_add")
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        return (gy if ctx.needs_input_grad[0] else None), (_ChannelSumDD.apply(gy) if ctx.needs_input_grad[1] else None)


class _ChannelSumDD(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return channel_sum(x.contiguous())

    @staticmethod
    def backward(ctx, g):   # broadcast of g over (B, :, H, W)
        g = _dev(g, "g").contiguous()
        B, C = ctx.shape[0], ctx.shape[1]
        y = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        L.check(L.lib().nq_bias_add(None, _p(g), _p(y), B, C, y.numel() // (B * C), _stream()), "bias_add")
        return y


def conv2d_dd(x, w, b=None):
    """stride-1 'same' convolution (+bias), differentiable to any order (reference F.conv2d, quant_layer.py:80, as used
    under torch.autograd.grad(create_graph=True) by methods/bit_assign.py:88-114)."""
    y = _ConvF.apply(x, w)
    return y if b is None else _BiasAddDD.apply(y, b)


def decoder_stack_dd(emb, spec, weights):
    """Decoder forward (reference HNeRV.py:49-71 / NeRV.py:44-65), twice differentiable, every operator on this repo's HIP
    kernels: the bilinear convolution family (_ConvF / _ConvD / _ConvWg), the bias broadcast, PixelShuffle, the exact-erf
    GELU and OutImg's tanh with their first and second derivatives (nq_act_dd).  Only the NeRV channel -> space view of
    layer 0 (NeRV.py:49-51: a reshape / permute, no arithmetic) is a PyTorch tensor view."""
    x = emb
    for l, ((k, r, act), (W, b)) in enumerate(zip(spec.layers, weights)):
        x = conv2d_dd(x, W, b)
        if l == 0 and spec.fc_hw != (1, 1):
            x = _space_from_channels(x, *spec.fc_hw)
        if r > 1:
            x = pixel_shuffle_dd(x, r)
        if act:
            x = gelu_dd(x)
    return tanh_out_dd(x) if spec.tanh_out else x


def conv2d_fused(x, w, b, epilogue=EPI_PLAIN, r=1):
    if not torch.is_grad_enabled() or not (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        # inference (the per-frame evaluation decodes, calibrate_network.py:82-145): no graph to record, and the big
        # layers can take the bf16x3 kernel like the calibration loop does
        x, w = _dev(x, "x"), _dev(w, "weight")
        b = _dev(b, "bias") if b is not None else None
        cout, cin, k, k2 = w.shape
        if k != k2 or x.shape[1] != cin:
            raise ValueError(f"conv shape mismatch: x {tuple(x.shape)} w {tuple(w.shape)}")
        B, _, H, W = x.shape
        if _use3(None) and conv3_supported(B, cin, H, W, cout, k):
            return conv3_forward_raw(x, weight_layout3(w), b, cout, k, epilogue, r)[0]
        wt, dims, _, _ = weight_layouts(w, False)
        return conv_forward_raw(x, wt, dims, b, cout, k, epilogue, r)[0]
    return _ConvFn.apply(x, w, b, epilogue, r)


class DecoderSpec:
    """Static description of a conv decoder stack for `decoder_stack`: per layer (k, shuffle r, gelu after),
    the (fc_h, fc_w) channel->space reshape after layer 0 and whether OutImg is tanh*0.5+0.5."""

    def __init__(self, layers, fc_hw=(1, 1), tanh_out=True, precision=None):
        self.layers = [tuple(l) for l in layers]
        self.fc_hw = tuple(fc_hw)
        self.tanh_out = tanh_out
        # 'bf16x3': convolutions whose grid fills the chip run on the BF16 matrix pipe with split fp32 operands
        # (hi*hi + hi*lo + lo*hi, fp32 accumulate; see conv_igemm3_impl.h); 'fp32': exact fp32 MFMA everywhere.
        self.precision = precision or get_conv_precision()
        if self.precision not in ("fp32", "bf16x3"):
            raise ValueError(f"unknown conv precision {self.precision!r}")
        # NQ_WGRAD_STREAM=1: weight gradients on a second HIP stream, concurrent with the data-gradient chain (measured
        # +3 % it/s on HNeRV-3M; off by default so that per-kernel durations in profiles stay separable)
        self.overlap_wgrad = os.environ.get("NQ_WGRAD_STREAM", "0") == "1"


def _space_from_channels(x, fh, fw):
    n, c, h, w = x.shape  # reference NeRV.py:49-51
    return x.view(n, -1, fh, fw, h, w).permute(0, 1, 4, 2, 5, 3).reshape(n, -1, fh * h, fw * w)


def _channels_from_space(g, fh, fw):
    n, c, hh, ww = g.shape
    return g.view(n, c, hh // fh, fh, ww // fw, fw).permute(0, 1, 3, 5, 2, 4).reshape(n, c * fh * fw, hh // fh, ww // fw)


# Data-parallel hook: when set, the decoder node writes ALL conv weight/bias gradients into one flat arena
# and calls the hook on it (e.g. an in-place RCCL all-reduce with ReduceOp.AVG): no flatten / un-flatten copies around the
# collective.  The hook is per-thread state that every decoder node CAPTURES in its forward (ctx.nq_arena), so the
# backward -- which PyTorch runs on its own autograd thread -- uses what was installed when the forward ran, two decoders
# in one process do not see each other's hook, and whether the exchange happened is recorded on the node itself
# (`img.grad_fn.nq_arena_reduced`), not in a module global.
_ARENA_TLS = threading.local()


def _arena_state():
    return getattr(_ARENA_TLS, "state", (None, False))


def set_grad_arena_hook(fn, two_phase=False):
    """fn(arena) is called once per backward with the flat gradient arena.  two_phase=True: the node first runs ALL data
    gradients, then the weight gradients of the deep layers (most of the parameters, little work), calls
    fn(arena[:split], False), runs the weight gradients of the last layers (most of the work, few parameters) and calls
    fn(arena[split:], True): an asynchronous collective started by the first call overlaps the expensive kernels.
    Applies to decoder nodes whose FORWARD runs on this thread from now on; prefer the `grad_arena_hook` context manager."""
    _ARENA_TLS.state = (fn, bool(two_phase) and fn is not None)


@contextlib.contextmanager
def grad_arena_hook(fn, two_phase=False):
    """`with ops.grad_arena_hook(fn, two_phase):` installs the hook for the body and restores the previous one on the way
    out, exceptions included."""
    prev = _arena_state()
    set_grad_arena_hook(fn, two_phase)
    try:
        yield
    finally:
        _ARENA_TLS.state = prev


def arena_reduced(img):
    """True when the decoder node behind `img` handed its gradient arena to a hook during backward (the caller then
    skips its own gradient exchange)."""
    return bool(getattr(getattr(img, "grad_fn", None), "nq_arena_reduced", False))


# Fused loss tail (round 4): inside `with ops.fused_head_loss(...)` a tanh-headed decoder node whose head is the 3-channel 3x3
# convolution runs nq_head_forward_loss -- the head AND lp_loss / tanh backward / bias gradient in one pass over the image --
# and leaves the result in its hand-over slot; l2_loss_head_grad then returns it without launching anything.  Per-thread
# state like the arena hook; a decoder that cannot use it simply ignores the request.
_LOSS_TLS = threading.local()


@contextlib.contextmanager
def fused_head_loss(tgt=None, cache_u8=None, idx=None):
    """Request the fused loss tail for decoder forwards in the body: target = float frames `tgt` (B,3,H,W) or frames
    cache_u8[idx] / 255 of the uint8 frame cache."""
    prev = getattr(_LOSS_TLS, "req", None)
    _LOSS_TLS.req = None if os.environ.get("NQ_FUSED_HEAD_LOSS", "1") == "0" else (tgt, cache_u8, idx)
    try:
        yield
    finally:
        _LOSS_TLS.req = prev


def head_forward_loss_raw(x, wt, dims, bias, cout, k, tgt=None, cache_u8=None, idx=None):
    """nq_head_forward_loss: -> (img, loss, dconv, db) or None when the fused kernel does not apply."""
    B, cin, H, W = x.shape
    if cout != 3 or k != 3 or (W & 3) or (dims[1] & 3) or os.environ.get("NQ_HEAD_FWD") == "1":
        return None
    if cache_u8 is not None:
        if not cache_u8.is_cuda or cache_u8.dtype != torch.uint8 or not cache_u8.is_contiguous() \
                or tuple(cache_u8.shape[1:]) != (cout, H, W) or idx is None or idx.numel() != B:
            return None
        idx = idx.to(device=x.device, dtype=torch.int64).contiguous()
        tgt = None
    elif tgt is not None:
        if not tgt.is_cuda or tgt.dtype != torch.float32 or tuple(tgt.shape) != (B, cout, H, W):
            return None
        tgt = tgt.contiguous()
    else:
        return None
    y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    dconv = torch.empty_like(y)
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    db = torch.empty(cout, device=x.device, dtype=torch.float32)
    ws = torch.empty(_q("nq_head_forward_loss_ws_floats", B, H, W), device=x.device, dtype=torch.float32)
    rc = _timed(("conv_igemm", k, cin, cout, H, W, B, EPI_TANH),
                lambda: L.lib().nq_head_forward_loss(_p(x), _p(wt), dims[1], _p(bias), _p(y), _p(tgt), _p(cache_u8), _p(idx), _p(loss),
                                                     _p(dconv), _p(db), _p(ws), B, cin, H, W, B * H * W, 1.0, _stream()))
    if rc == -2:     # NQ_ERR_UNSUPPORTED: the caller runs the separate kernels
        return None
    L.check(rc, "head_forward_loss")
    return y, loss, dconv, db


_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _DecoderStackFn(Function):
    """Whole decoder (reference HNeRV.py:49-71 / NeRV.py:44-65) as ONE autograd node with an explicit schedule.

    Every block's conv epilogue writes a = gelu(PixelShuffle(conv+bias)) and d = gelu'(...) from one erf (EPI_PS_GELU);
    the data-gradient kernel of the layer above multiplies by d and un-shuffles in its epilogue (EPI_DGRAD_GELU).  So
    per block: 1 launch forward, 2 backward (+ the small split-K reductions), no elementwise passes over activations.
    """

    @staticmethod
    def forward(ctx, emb, spec, *wb):
        x = _dev(emb, "embedding")
        n = len(spec.layers)
        saved_in, saved_z, metas = [], [], []
        in_gelu = False      # (kept in the metas for the generic kernels' GELU-on-load option; unused by this schedule)
        zprev = None         # gelu'(pre-activation) behind x, saved by the producing epilogue
        # pre-pass over the static shapes: which layers run on the bf16x3 kernels (forward / data gradient); all their
        # pre-split operands are then built by ONE launch before the first convolution
        Bx, Hx, Wx = x.shape[0], x.shape[2], x.shape[3]
        plan, items = [], []
        fio, bio, wio = [0] * n, [0] * n, [0] * n   # split {hi | lo} word capabilities: forward / data gradient / weight gradient
        for l, (k, r, act) in enumerate(spec.layers):
            W = _dev(wb[2 * l], "weight")
            cout, cin = W.shape[0], W.shape[1]
            use3 = spec.precision == "bf16x3" and conv3_supported(Bx, cin, Hx, Wx, cout, k)
            use3_bwd = spec.precision == "bf16x3" and l > 0 and conv3_supported(Bx, cout, Hx, Wx, cin, k)
            plan.append((use3, use3_bwd, len(items) if use3 else -1, len(items) + int(use3) if use3_bwd else -1))
            if use3:
                items.append((W, False))
                fio[l] = conv3_split_io(Bx, cin, Hx, Wx, cout, k)
            if use3_bwd:
                items.append((W, True))
                bio[l] = conv3_split_io(Bx, cout, Hx, Wx, cin, k)
            if spec.precision == "bf16x3" and conv_wgrad3_supported(Bx, cin, Hx, Wx, cout, k):
                wio[l] = conv_wgrad3_split_io(Bx, cin, Hx, Wx, cout, k)
            if l == n - 1 and l > 0 and not use3_bwd:   # the head's data gradient (streaming vector kernel) can write split words
                kp, rp, actp = spec.layers[l - 1]
                if actp and conv_split_out(Bx, cout, Hx, Wx, cin, k, rp, EPI_DGRAD_GELU):
                    bio[l] = EPI_Y_SPLIT
            if l == 0:
                Hx, Wx = Hx * spec.fc_hw[0], Wx * spec.fc_hw[1]
            Hx, Wx = Hx * r, Wx * r
        # Which tensors travel as split {hi | lo} words (round 4; NQ_SPLIT_IO=0: none): the input x_l of layer l (l >= 2: written by
        # the GELU epilogue of layer l - 1, read by layer l's forward patch staging and by its weight gradient) and the conv-output
        # gradient g_l (written by the data gradient of layer l + 1, read by layer l's weight gradient and data gradient) --
        # where EVERY kernel on both sides takes / writes that form.  Same values in the matrix pipe: identical results.
        split_on = spec.precision == "bf16x3" and os.environ.get("NQ_SPLIT_IO", "1") != "0"
        xsp, gsp = [False] * n, [False] * n
        if split_on:
            for l in range(2, n):
                xsp[l] = bool(spec.layers[l - 1][2] and (fio[l - 1] & EPI_Y_SPLIT) and (fio[l] & EPI_X_SPLIT) and (wio[l] & 1))
            for l in range(1, n - 1):
                gsp[l] = bool((bio[l + 1] & EPI_Y_SPLIT) and (bio[l] & EPI_X_SPLIT) and (wio[l] & 2) and spec.layers[l - 1][2]
                              and not (l == 1 and spec.fc_hw != (1, 1)))
        # the fp32 operands (layers / directions that stay on the fp32 kernels) come from the SAME launch (round 4).
        # layer 0's data-gradient operand is only needed when the embedding itself is trained (FP32 trainer: the
        # ConvNeXt encoder sits below it, reference regress.py:259-266)
        need_bs = [(l > 0 and not plan[l][1]) or (l == 0 and ctx.needs_input_grad[0]) for l in range(n)]
        fp32_items = [(_dev(wb[2 * l], "weight"), not plan[l][0], need_bs[l]) for l in range(n)
                      if (not plan[l][0]) or need_bs[l]]
        operands, fp32_list = weight_layouts_all(items, fp32_items)
        fp32_ops = iter(fp32_list)
        for l, (k, r, act) in enumerate(spec.layers):
            W = _dev(wb[2 * l], "weight")
            b = _dev(wb[2 * l + 1], "bias") if wb[2 * l + 1] is not None else None
            cout, cin = W.shape[0], W.shape[1]
            use3, use3_bwd, i_f, i_b = plan[l]
            need_b = need_bs[l]
            wt = dims = wbk = dims_b = None
            if use3:
                wt3 = operands[i_f]
            if (not use3) or need_b:
                wt, dims, wbk, dims_b = next(fp32_ops)
                if not need_b:
                    wbk = dims_b = None
            last = l == n - 1
            if last:
                epi = EPI_TANH if spec.tanh_out else EPI_PLAIN
            elif act:
                epi = EPI_PS_GELU
            elif r > 1:
                epi = EPI_PS
            else:
                epi = EPI_PLAIN
            fused_loss = None
            if last and epi == EPI_TANH and not use3 and getattr(_LOSS_TLS, "req", None) is not None:
                tgt_, cache_, idx_ = _LOSS_TLS.req
                fused_loss = head_forward_loss_raw(x, wt, dims, b, cout, k, tgt_, cache_, idx_)
            if fused_loss is not None:
                y, z = fused_loss[0], None
            elif use3:
                y, z = conv3_forward_raw(x, wt3, b, cout, k, epi, r,
                                         fmt=(EPI_X_SPLIT if xsp[l] else 0) | (EPI_Y_SPLIT if l + 1 < n and xsp[l + 1] else 0))
            else:
                y, z = conv_forward_raw(x, wt, dims, b, cout, k, epi, r, in_gelu=in_gelu)
            saved_in.append(x)
            saved_z.append(zprev)
            metas.append((k, r, act, cout, cin, in_gelu, wbk, dims_b, b is not None, operands[i_b] if use3_bwd else None,
                          xsp[l], gsp[l]))
            if epi == EPI_PS_GELU:
                x, zprev, in_gelu = y, z, False
            elif epi == EPI_PS:
                x, zprev, in_gelu = z, None, False
            else:
                x, zprev, in_gelu = y, None, False
            if l == 0 and spec.fc_hw != (1, 1):
                x = _space_from_channels(x, *spec.fc_hw).contiguous()
        ctx.spec, ctx.metas, ctx.n = spec, metas, n
        ctx.save_for_backward(x, *saved_in, *saved_z)
        # per-node state (reachable from outside as img.grad_fn.<name>): the data-parallel hook installed on this thread
        # when the forward ran, whether backward handed the arena to it, and the hand-over slot of the fused loss tail
        ctx.nq_arena, ctx.nq_arena_reduced = _arena_state(), False
        ctx.nq_head = _HeadHandoff() if spec.tanh_out else None
        if fused_loss is not None:   # the loss tail already ran behind the head: l2_loss_head_grad hands it out
            ctx.nq_head.loss, ctx.nq_head.dconv, ctx.nq_head.db = fused_loss[1], fused_loss[2], fused_loss[3]
            ctx.nq_head.version = fused_loss[2]._version
        return x

    @staticmethod
    def backward(ctx, g_img):
        hook = ctx.nq_arena[0]
        steps = _decoder_backward_steps(ctx, g_img)
        while True:
            try:
                part, last = next(steps)
            except StopIteration as done:
                return done.value
            if last is None:
                hook(part)
            else:
                hook(part, last)


def _decoder_backward_steps(ctx, g_img):
    """The backward pass of a decoder node as a GENERATOR: it yields (arena part, last) wherever a part of the flat gradient
    arena is complete and must be exchanged between data-parallel ranks (last = None: the whole arena at once), and
    returns the gradient tuple of _DecoderStackFn.backward.  The autograd node drives it and calls the installed hook at
    every yield; the captured data-parallel iteration (quantization/calib_model.py) drives it stage by stage, with the
    collectives launched eagerly between three replayed graphs."""
    spec, metas, n = ctx.spec, ctx.metas, ctx.n
    saved = ctx.saved_tensors
    img, xs, zs = saved[0], saved[1:1 + n], saved[1 + n:]
    d_emb = None
    g = _dev(g_img, "grad")
    head_db = None
    head = ctx.nq_head
    if head is not None and head.loss is not None:
        # the fused loss tail ran behind the head but nobody took its result (l2_loss_head_grad was not called): the incoming
        # gradient is an ordinary d(loss)/d(image)
        head.dconv = head.db = head.loss = None
    if head is not None and head.dconv is not None:
        # l2_loss_head_grad handed over the gradient at the head conv's OUTPUT (tanh backward applied) and the head's
        # bias gradient.  Anything but that very tensor, unmodified, cannot be continued from: say so loudly.
        if head.dconv.data_ptr() != g.data_ptr() or g.shape != head.dconv.shape:
            raise RuntimeError("neuroquant_amd: the gradient returned by ops.l2_loss_head_grad must be passed to "
                               "backward() as it is (it already contains the tanh backward of this decoder)")
        dconv = g
        # scaled / edited in place since the hand-over: the pre-summed bias gradient no longer matches -> re-sum it
        head_db = head.db if g._version == head.version else None
        head.dconv = head.db = None
    elif spec.tanh_out:
        dconv = torch.empty_like(g)
        L.check(L.lib().nq_tanh_out_backward(_p(g), _p(img), _p(dconv), g.numel(), _stream()), "tanh_backward")
    else:
        dconv = g
    grads = [None] * (2 * n)
    # The weight gradients are off the critical path (only the data gradients chain): they run on a second HIP
    # stream so that their workgroups fill the partial last rounds ("tails") of the data-gradient kernels and the
    # launch gaps of the small deep layers.  Every dconv stays referenced until the streams are joined.
    main = torch.cuda.current_stream()
    side = _side_stream(g.device) if spec.overlap_wgrad else None
    keep = []

    # one flat arena for all weight/bias gradients when a data-parallel hook is installed (in-place collective)
    arena = views = None
    arena_hook, arena_two_phase = ctx.nq_arena
    if arena_hook is not None:
        sizes = []
        for l in range(n):
            k, r, act, cout, cin = metas[l][:5]
            sizes += [cout * cin * k * k, cout if metas[l][8] else 0]
        arena = torch.empty(sum(sizes), device=g.device, dtype=torch.float32)
        views, off = [], 0
        for l in range(n):
            k, r, act, cout, cin = metas[l][:5]
            wv = arena[off:off + sizes[2 * l]].view(cout, cin, k, k)
            off += sizes[2 * l]
            bv = arena[off:off + sizes[2 * l + 1]] if sizes[2 * l + 1] else None
            off += sizes[2 * l + 1]
            views.append((wv, bv))

    # the split-K weight-gradient kernels leave slabs; their fixed-order reductions run in ONE launch per group of
    # layers (`pending.flush()`: at the end of the backward pass, or before each part of the arena goes to the
    # data-parallel hook) instead of one ~10 us launch per layer.  Same sums in the same order: bit-identical.
    pending = PendingReductions() if side is None and os.environ.get("NQ_DEFER_REDUCE", "1") != "0" else None

    def wgrad(l, dconv):
        k, r, act, cout, cin, in_gelu, wbk, dims_b, has_b, W3, x_split, g_split = metas[l]
        x_in = xs[l]
        Bx, _, Hx, Wx = x_in.shape
        out = views[l] if views is not None else None
        if spec.precision == "bf16x3" and not in_gelu and conv_wgrad3_supported(Bx, cin, Hx, Wx, cout, k):
            return conv_wgrad3_raw(x_in, dconv, cout, k, has_b, out=out, defer=pending,
                                   fmt=(1 if x_split else 0) | (2 if g_split else 0))
        if spec.precision == "bf16x3" and not in_gelu and cout <= 4 and cin > 4 and cout * k * k <= 64 \
                and conv_wgrad3_supported(Bx, cout, Hx, Wx, cin, k):
            if l == n - 1 and has_b and head_db is not None:   # bias gradient handed over by l2_loss_head_grad
                dw, _ = conv_wgrad_swapped3(x_in, dconv, cout, k, False, out=out, defer=pending)
                if out is not None:
                    out[1].copy_(head_db)
                    return dw, out[1]
                return dw, head_db
            return conv_wgrad_swapped3(x_in, dconv, cout, k, has_b, out=out, defer=pending)
        return conv_wgrad_raw(x_in, dconv, cout, k, has_b, x_gelu=in_gelu, out=out, defer=pending)

    def dgrad(l, dconv):
        """conv-output gradient of layer l -> conv-output gradient of layer l - 1 (l >= 1)"""
        k, r, act, cout, cin, in_gelu, wbk, dims_b, has_b, W3, x_split, g_split = metas[l]
        kp, rp, actp = spec.layers[l - 1]
        epi_b, r_b, zp = (EPI_DGRAD_GELU, rp, zs[l]) if actp else (EPI_PLAIN, 1, None)
        if not actp and rp != 1:
            raise NotImplementedError("PixelShuffle without activation between decoder layers")
        out_split = EPI_Y_SPLIT if metas[l - 1][11] else 0   # the gradient this call produces is layer l - 1's g
        # the layer below ends in GELU: d(pre-activation) = dgrad * gelu'(z), stored as ITS conv-output gradient
        if W3 is not None:   # W3 = pre-built transposed operand
            d, _ = conv3_forward_raw(dconv, W3, None, cin, k, epi_b, r_b, zprev=zp, fmt=(EPI_X_SPLIT if g_split else 0) | out_split)
        else:
            d, _ = conv_forward_raw(dconv, wbk, dims_b, None, cin, k, epi_b, r_b, zprev=zp, fmt=out_split)
        if not actp and l == 1 and spec.fc_hw != (1, 1):
            d = _channels_from_space(d, *spec.fc_hw).contiguous()
        return d

    def emb_grad(dconv):   # d(embedding): plain data gradient through layer 0 (no activation below it)
        k, r, act, cout, cin, in_gelu, wbk, dims_b, has_b, W3 = metas[0][:10]
        return conv_forward_raw(dconv, wbk, dims_b, None, cin, k, EPI_PLAIN, 1)[0]

    if arena is not None and arena_two_phase and n > 1:
        # Data-parallel schedule: the data-gradient chain first, then the weight gradients of the deep layers (most
        # of the parameters, a few per cent of the work) whose part of the arena is handed to the hook at once, then
        # the last layers' weight gradients (>= 80 % of the weight-gradient flops) while that collective is in
        # flight, then the rest of the arena.  Same kernels on the same operands as the single-GPU order: identical bits.
        dcs = [None] * n
        dcs[n - 1] = dconv
        for l in range(n - 1, 0, -1):
            dcs[l - 1] = dgrad(l, dcs[l])
        if ctx.needs_input_grad[0]:
            d_emb = emb_grad(dcs[0])
        flops = [metas[l][3] * metas[l][4] * metas[l][0] ** 2 * xs[l].shape[2] * xs[l].shape[3] for l in range(n)]
        split, late = n - 1, flops[n - 1]
        while split > 1 and late < 0.8 * sum(flops):
            split -= 1
            late += flops[split]
        for l in range(split):
            grads[2 * l], grads[2 * l + 1] = wgrad(l, dcs[l])
            dcs[l] = None
        off = sum(sizes[:2 * split])
        if pending is not None:
            pending.flush()
        yield arena[:off], False
        for l in range(split, n):
            grads[2 * l], grads[2 * l + 1] = wgrad(l, dcs[l])
            dcs[l] = None
        if pending is not None:
            pending.flush()
        yield arena[off:], True
        ctx.nq_arena_reduced = True
        return (d_emb, None) + tuple(grads)

    for l in range(n - 1, -1, -1):
        if side is not None:
            keep.append(dconv)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dw, db = wgrad(l, dconv)
            for t in (dw, db):
                if t is not None:
                    t.record_stream(main)
        else:
            dw, db = wgrad(l, dconv)
        grads[2 * l], grads[2 * l + 1] = dw, db
        if l == 0:
            if ctx.needs_input_grad[0]:
                d_emb = emb_grad(dconv)
            break
        dconv = dgrad(l, dconv)
    if side is not None:
        main.wait_stream(side)
        keep.clear()
    if pending is not None:
        pending.flush()
    if arena is not None:
        yield arena, None
        ctx.nq_arena_reduced = True
    return (d_emb, None) + tuple(grads)




class _ManualNode:
    """Stand-in for the autograd context of _DecoderStackFn when the decoder is driven WITHOUT autograd (captured
    data-parallel iterations): forward fills it, decoder_backward_steps consumes it."""

    def __init__(self, n_inputs, want_emb_grad=False):
        self.needs_input_grad = (bool(want_emb_grad), False) + (True,) * n_inputs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def decoder_forward_manual(emb, spec: DecoderSpec, weights, two_phase=True):
    """decoder_stack without autograd: -> (image, node).  The node always collects its weight gradients in ONE flat arena
    and decoder_backward_steps(node, g) yields its parts for the caller to exchange (it calls no hook itself)."""
    flat = []
    for W, b in weights:
        flat += [W, b]
    node = _ManualNode(len(flat))
    with torch.no_grad():
        img = _DecoderStackFn.forward(node, emb, spec, *flat)
    node.nq_arena = ("manual", bool(two_phase))
    return img, node


def decoder_backward_steps(node, g_img):
    """Generator over the backward pass of a decoder_forward_manual node: yields (arena part, last | None) at the exchange
    points; StopIteration.value = (d_emb, None, dW0, db0, dW1, ...) as _DecoderStackFn.backward returns them."""
    return _decoder_backward_steps(node, g_img)


def decoder_stack(emb, spec: DecoderSpec, weights):
    """weights: [(W, b), ...] per layer (already fake-quantised).  Returns the output image."""
    flat = []
    for W, b in weights:
        flat += [W, b]
    return _DecoderStackFn.apply(emb, spec, *flat)


# ------------------------------------------------------------------------------------------ loss / metrics / frames
class _L2LossFn(Function):
    """lp_loss(pred, tgt, p=2): sum over channels, mean over batch*H*W (reference quantizer.py:66-71)."""

    @staticmethod
    def forward(ctx, pred, tgt):
        pred, tgt = _dev(pred, "pred"), _dev(tgt, "tgt")
        n = pred.numel()
        mean_count = n // pred.shape[1]
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        dpred = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        ws = torch.empty(_q("nq_reduce_ws_floats", n), device=pred.device, dtype=torch.float32)
        L.check(L.lib().nq_l2_loss(_p(pred), _p(tgt), _p(loss), _p(dpred), _p(ws), n, mean_count, 1.0, _stream()), "l2_loss")
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None


def l2_loss(pred, tgt):
    return _L2LossFn.apply(pred, tgt)


def l2_loss_and_grad(pred, tgt):
    """(loss, d loss / d pred) from ONE kernel, outside autograd: `pred.backward(dpred)` then feeds the decoder node
    directly (no ones-tensor, no elementwise multiply by the upstream gradient)."""
    pred_d, tgt = _dev(pred.detach(), "pred"), _dev(tgt, "tgt")
    n = pred_d.numel()
    loss = torch.empty((), device=pred_d.device, dtype=torch.float32)
    dpred = torch.empty_like(pred_d)
    ws = torch.empty(_q("nq_reduce_ws_floats", n), device=pred_d.device, dtype=torch.float32)
    L.check(L.lib().nq_l2_loss(_p(pred_d), _p(tgt), _p(loss), _p(dpred), _p(ws), n, n // pred_d.shape[1], 1.0, _stream()),
            "l2_loss")
    return loss, dpred


class _HeadHandoff:
    """Hand-over slot of the fused loss tail, owned by ONE decoder node (ctx.nq_head, reachable as img.grad_fn.nq_head):
    l2_loss_head_grad fills it with the gradient at the head conv's output and the head's bias gradient, the node's
    backward takes them when the incoming gradient is that very tensor (`version` = its in-place edit counter at the
    hand-over: an edited gradient is still continued from, but the bias gradient is summed again)."""
    __slots__ = ("dconv", "db", "version", "loss")

    def __init__(self):
        self.dconv = self.db = self.version = self.loss = None


def l2_loss_tanh_head_raw(pred, tgt=None, cache_u8=None, idx=None):
    """One nq_l2_loss_tanh_head launch on an image `pred` = tanh(conv)*0.5+0.5: -> (loss, dconv, db) with lp_loss(pred,
    tgt, p=2), the gradient at the conv OUTPUT (tanh backward applied) and its per-channel sums (the head's bias
    gradient); the target is `tgt` or frames `cache_u8[idx]/255` read straight from the uint8 cache.  None when the
    kernel's tiling does not apply (H*W % 4096 != 0)."""
    pred_d = _dev(pred.detach(), "pred")
    if pred_d.dim() != 4:
        return None
    B, C, H, W = pred_d.shape
    if (H * W) % 4096 != 0 or C > 1024:
        return None
    if cache_u8 is not None:
        if not cache_u8.is_cuda or cache_u8.dtype != torch.uint8 or not cache_u8.is_contiguous() \
                or tuple(cache_u8.shape[1:]) != (C, H, W) or idx.numel() != B:
            raise RuntimeError("frame cache must be a contiguous uint8 GPU tensor of the image's shape")
        idx = idx.to(device=pred_d.device, dtype=torch.int64).contiguous()
    else:
        tgt = _dev(tgt, "tgt")
    n = pred_d.numel()
    loss = torch.empty((), device=pred_d.device, dtype=torch.float32)
    dconv = torch.empty_like(pred_d)
    db = torch.empty(C, device=pred_d.device, dtype=torch.float32)
    ws = torch.empty(2 * _q("nq_reduce_ws_floats", n), device=pred_d.device, dtype=torch.float32)
    L.check(L.lib().nq_l2_loss_tanh_head(_p(pred_d), _p(tgt) if cache_u8 is None else None,
                                         _p(cache_u8) if cache_u8 is not None else None,
                                         _p(idx) if cache_u8 is not None else None, _p(loss), _p(dconv), _p(db), _p(ws),
                                         B, C, H * W, n // C, 1.0, _stream()), "l2_loss_tanh_head")
    return loss, dconv, db


def l2_loss_head_grad(pred, tgt=None, cache_u8=None, idx=None, node=None):
    """lp_loss(pred, tgt, p=2) for `pred` = the image a tanh-headed `decoder_stack` just returned, fused with what its
    backward does first: returns (loss, g) where `pred.backward(g)` continues at the head convolution -- g is the
    gradient at the head conv's OUTPUT (tanh backward applied) and the head's bias gradient is handed over with it
    (one pass over the image instead of loss + tanh backward + channel sums; the target can be read straight from the
    uint8 frame cache).  The hand-over lives on pred's own autograd node, so g belongs to THIS decoder call only and
    must reach backward() as returned.  Returns None when `pred` is not such an image (no graph recorded, another
    producer) or the fused kernel does not apply; the caller then uses l2_loss_and_grad."""
    # only the tensor a tanh-headed decoder node returned carries a hand-over slot (node: the manual node of
    # decoder_forward_manual, whose image has no autograd history)
    head = getattr(node if node is not None else pred.grad_fn, "nq_head", None)
    if not isinstance(head, _HeadHandoff) or os.environ.get("NQ_FUSED_LOSS", "1") == "0":
        return None
    if head.loss is not None:   # computed behind the head convolution (ops.fused_head_loss): nothing to launch
        loss, head.loss = head.loss, None
        return loss, head.dconv
    out = l2_loss_tanh_head_raw(pred, tgt, cache_u8, idx)
    if out is None:
        return None
    loss, head.dconv, head.db = out
    head.version = head.dconv._version
    return loss, head.dconv


def frame_psnr(out, gt):
    """per-frame PSNR (reference utils.py:148-151): -10*log10(mean((out-gt)^2) + 1e-9)."""
    out, gt = _dev(out.detach()), _dev(gt.detach())
    frames = out.shape[0]
    flen = out.numel() // frames
    sse = torch.empty(frames, device=out.device, dtype=torch.float32)
    L.check(L.lib().nq_frame_sse(_p(out), _p(gt), _p(sse), frames, flen, _stream()), "frame_sse")
    return -10 * torch.log10(sse / flen + 1e-9)


def gather_frames_u8(frames_u8: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """float frames[idx] / 255 from a GPU-resident uint8 cache (reference videosets/datasets.py:19-24)."""
    if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or not frames_u8.is_contiguous():
        raise RuntimeError("frame cache must be a contiguous uint8 GPU tensor")
    idx = idx.to(device=frames_u8.device, dtype=torch.int64).contiguous()
    n = idx.numel()
    flen = frames_u8[0].numel()
    out = torch.empty((n,) + tuple(frames_u8.shape[1:]), device=frames_u8.device, dtype=torch.float32)
    L.check(L.lib().nq_gather_frames_u8(_p(frames_u8), _p(idx), _p(out), n, flen, _stream()), "gather_frames")
    return out
