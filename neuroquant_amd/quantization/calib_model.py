"""Network-wise calibration (reference quantization/calib_model.py): learn the per-channel scales, then the
AdaRound rounding variables, against the reconstruction loss of the WHOLE decoder.

`model_reconstruction` keeps the reference's signature and in-place semantics but runs each iteration as an
explicit schedule on HIP kernels:

    fake-quant (+Hadamard) of all layers          nq_uaq_forward / nq_adaround_forward / nq_fwht
    decoder forward + backward                    nq_conv_forward / nq_conv_wgrad / nq_ps_gelu_backward ...
    reconstruction loss + its gradient            nq_l2_loss
    d(delta) or d(alpha) + regulariser gradient   nq_uaq_backward / nq_adaround_backward (fused)
    Adam                                          nq_adam_step

Autograd only spans the convolution stack; the parameter side is explicit, so no per-layer graph nodes, no
host synchronisation per iteration (the reference's decode sync and 500-step float() logging are the only
syncs it had; the log sync is kept, the decode sync is not).
"""
import contextlib
import logging
import os

import torch
import torch.nn as nn

from .. import ops
from ..utils import allreduce_mean_
from .data_utils import LinearTempDecay
from .quant_layer import QuantModule
from .quant_model import QuantModel
from .quantizer import AdaRoundQuantizer, lp_loss


def _quant_modules(module: nn.Module, out=None):
    """QuantModules in the order of the reference's recursion (children order, '*encoder*' skipped)."""
    out = [] if out is None else out
    for name, child in module.named_children():
        if 'encoder' in name:
            continue
        if isinstance(child, QuantModule):
            out.append(child)
        else:
            _quant_modules(child, out)
    return out


class LossFunction:
    """rec loss + rounding regulariser with a linear temperature schedule (reference calib_model.py:16-89).
    Generic autograd form for callers that drive their own loop; `model_reconstruction` uses the fused kernels."""

    def __init__(self, model: nn.Module, round_loss: str = 'relaxation', weight: float = 1., rec_loss: str = 'mse',
                 max_count: int = 2000, b_range: tuple = (10, 2), decay_start: float = 0.0, warmup: float = 0.0,
                 p: float = 2.):
        self.model = model
        self.round = round_loss
        self.weight = weight
        self.rec = rec_loss
        self.loss_start = max_count * warmup
        self.p = p
        self.temp_decay = LinearTempDecay(max_count, rel_start_decay=warmup + (1 - warmup) * decay_start,
                                          start_b=b_range[0], end_b=b_range[1])
        self.count = 0
        self.round_loss = 0

    def collect_round_loss(self, module, b):
        for m in _quant_modules(module):
            self.round_loss = self.round_loss + ops.round_regulariser(m.weight_quantizer.alpha, b, self.weight)

    def __call__(self, pred, tgt, grad=None):
        self.count += 1
        if self.rec != 'mse':
            raise NotImplementedError(f'rec_loss {self.rec!r}: the drivers hard-code opt_mode="mse"')
        rec_loss = lp_loss(pred, tgt, p=self.p)
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round == 'none':
            b = self.round_loss = 0
        elif self.round == 'relaxation':
            self.round_loss = 0
            self.collect_round_loss(self.model, b)
        else:
            raise NotImplementedError
        total_loss = self.round_loss + rec_loss
        if self.count % 500 == 0:
            logging.info('Total loss:\t{:.4f} (rec:{:.4f}, round:{:.4f})\tb={:.2f}\tcount={}'.format(
                float(total_loss), float(rec_loss), float(self.round_loss), b, self.count))
        return total_loss


class _Layer:
    """Per-QuantModule view used by the explicit schedule."""

    def __init__(self, m: QuantModule, hadamard: bool):
        self.m = m
        self.hadamard = hadamard
        self.src = m.hadamard_weight if hadamard else m.weight.data
        self.n = self.src.shape[1]           # transform length (C_pad) when hadamard
        self.c_in = m.weight.shape[1]
        self.bias = m.bias.data
        self.W = self.b = None

    def forward_uaq(self):
        wq, bq = self.m.weight_quantizer, self.m.bias_quantizer
        Wq = ops.uaq_forward(self.src, wq.delta.data, wq.zero_point, wq.n_levels)
        self._finish(Wq, ops.uaq_forward(self.bias, bq.delta.data, bq.zero_point, bq.n_levels))

    def forward_ada(self):
        wq, bq = self.m.weight_quantizer, self.m.bias_quantizer
        Wq = ops.adaround_forward(self.src, wq.alpha.data, wq.delta.data, wq.zero_point, wq.n_levels, wq.soft_targets)
        self._finish(Wq, ops.adaround_forward(self.bias, bq.alpha.data, bq.delta.data, bq.zero_point, bq.n_levels,
                                              bq.soft_targets))

    def uaq_items(self):
        wq, bq = self.m.weight_quantizer, self.m.bias_quantizer
        return [(self.src, wq.delta.data, wq.zero_point, wq.n_levels), (self.bias, bq.delta.data, bq.zero_point, bq.n_levels)]

    def ada_items(self):
        wq, bq = self.m.weight_quantizer, self.m.bias_quantizer
        return [(self.src, wq.alpha.data, wq.delta.data, wq.zero_point, wq.n_levels, wq.soft_targets),
                (self.bias, bq.alpha.data, bq.delta.data, bq.zero_point, bq.n_levels, bq.soft_targets)]

    def _finish(self, Wq, b, transformed=False):
        self.W = (ops.fwht_channels(Wq, self.n, self.c_in) if self.hadamard and not transformed else Wq).requires_grad_(True)
        self.b = b.requires_grad_(True)
        self.m._wb_override = (self.W, self.b)

    def grads(self):
        gW = self.W.grad
        if self.hadamard:
            gW = ops.fwht_channels(gW, self.n, self.n)   # H on the zero-padded gradient
        return gW, self.b.grad

    def release(self):
        self.m._wb_override = None
        self.W = self.b = None


_AVG_OK = {}    # backend name -> the backend reduces with ReduceOp.AVG (RCCL does, this image's gloo too; older gloo refuses it)


def _avg_unsupported(err):
    """True for the error a backend raises when it has no ReduceOp.AVG ("AVG is only available with the NCCL backend",
    "Cannot use ReduceOp.AVG with Gloo", "unsupported reduce op" ...).  Anything else -- a communicator fault, a timeout,
    a wrong device -- is NOT swallowed: with RCCL an asynchronous collective can also raise RuntimeError for those."""
    msg = str(err).lower()
    return "avg" in msg or "reduceop" in msg or "reduce op" in msg or "unsupported reduction" in msg


def _mean_all_reduce(t):
    """Asynchronous in-place mean of `t` over the ranks; -> a function that completes it (the compute stream waits for the
    collective -- no host synchronisation with RCCL).  ReduceOp.AVG where the backend has it (RCCL: the division happens
    inside the collective, one pass over the arena), else SUM and a scale after the wait (a backend that refuses AVG)."""
    import torch.distributed as dist
    be = dist.get_backend()
    if _AVG_OK.get(be, True):
        try:
            w = dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)
            _AVG_OK[be] = True
            return w.wait
        except (RuntimeError, ValueError, NotImplementedError) as err:
            if not _avg_unsupported(err):
                raise
            _AVG_OK[be] = False
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    scale = 1.0 / dist.get_world_size()

    def fin():
        w.wait()
        if scale != 1.0:
            t.mul_(scale)
    return fin


def model_reconstruction(model: QuantModel, cali_data: torch.Tensor, gt, arch: str = 'hnerv', batch_size: int = 8,
                         iters: int = 20000, weight: float = 0.01, opt_mode: str = 'mse', hadamard: bool = True,
                         b_range: tuple = (20, 2), warmup: float = 0.0, p: float = 2.0, lr: float = 0.0015,
                         recorder: list = None, max_steps: int = None, step_hook=None, probe=None):
    """Network-wise calibration, in place (reference calib_model.py:92-240).

    gt: any sized iterable of dict batches {'img' (B,3,H,W), 'idx' (B,), 'norm_idx'}; cali_data[idx] feeds the
    decoder.  recorder (optional list) receives (total, round, b, count) per iteration -- this forces one host
    sync per iteration and is meant for parity tests.  max_steps truncates the run and step_hook(done) is called
    before every iteration (bench.py uses both to time exactly K steps).  probe(phase, layers, grads) is called after
    the backward pass and before the optimiser step with phase 'uaq' | 'ada', the engine's per-layer records (L.W.grad /
    L.b.grad = dL/dW^, dL/db^) and the list of parameter gradients in optimiser order (gradient parity tests).
    """
    if arch not in ('hnerv', 'nerv'):
        raise ValueError
    if opt_mode != 'mse' or p != 2.0:
        raise NotImplementedError('only opt_mode="mse", p=2 is on the calibration path')
    model.set_quant_state(True)
    device = next(model.parameters()).device
    layers = [_Layer(m, hadamard) for m in _quant_modules(model)]
    for L in layers:  # scales are initialised lazily at the first quantiser call (quantizer.py:112-115)
        if not L.m.weight_quantizer.inited:
            L.m.weight_quantizer(L.src)
        if not L.m.bias_quantizer.inited:
            L.m.bias_quantizer(L.bias)
    done = 0
    dp = torch.distributed.is_available() and torch.distributed.is_initialized() and \
        (torch.distributed.get_world_size() > 1 or bool(os.environ.get("NQ_DP_REHEARSAL")))

    hook_scope = contextlib.nullcontext()
    if dp and device.type == 'cuda':
        # fused decoder node: gradients land in one arena that is all-reduced in place (RCCL, mean over ranks), in two
        # asynchronous pieces: the deep layers' gradients (85 % of the bytes of an HNeRV-3M, ready after a few per cent of
        # the weight-gradient work) travel while the last layers' weight gradients are computed (ops.grad_arena_hook).
        # The hook is scoped to this call: removed on the way out, exceptions included.
        import torch.distributed as dist
        pending = []

        def reduce_part(part, last=True):
            pending.append(_mean_all_reduce(part))
            if last:   # the compute stream waits for both collectives (no host sync)
                for fin in pending:
                    fin()
                pending.clear()

        hook_scope = ops.grad_arena_hook(reduce_part, two_phase=os.environ.get("NQ_DP_OVERLAP", "1") != "0")

    # Captured iterations (hipGraph): when `gt` can hand over a whole epoch of frame indices
    # (utils.CacheLoader.epoch_indices) the iteration is captured ONCE per phase with torch.cuda.graph and replayed: the
    # ~50 launches + ~60 allocations + the autograd walk of one iteration become one graph launch.  What changes between
    # iterations -- frame indices, temperature / regulariser gate, Adam's bias corrections -- sits in per-epoch device
    # tables; nq_step_prologue copies the current row into fixed slots that the kernels read (same values and arithmetic
    # as the host-argument path: replays are bit-identical to eager launches, tested).  Iterations that log (every 500,
    # the reference's float() sync), that are HIP-event profiled, or the first three of a phase run eagerly through the
    # SAME body.  Off for: recorder / probe (tests that inspect every iteration), generic `gt` iterables, NQ_GRAPH=0.
    # Data-parallel runs replay the staged form below (three graphs with the collectives between them) when NQ_DP_GRAPH=1
    # and run eagerly otherwise.
    use_graph = (os.environ.get("NQ_GRAPH", "1") != "0" and device.type == 'cuda' and recorder is None
                 and probe is None and hasattr(gt, 'epoch_indices') and hasattr(gt, 'cache'))
    # Data-parallel runs (round 3): the iteration is captured as up to THREE graphs -- everything up to the first complete
    # part of the gradient arena | the last layers' weight gradients | the parameter side -- and the in-place all-reduces
    # are launched EAGERLY between their replays (asynchronous, on RCCL's stream): the collectives never sit inside a
    # captured region, the ~1.2 ms of host work per eager iteration becomes three graph launches + two collective
    # enqueues.  The decoder then runs without autograd (ops.decoder_forward_manual / decoder_backward_steps: the same
    # code as the autograd node, driven stage by stage).  Opt-in (NQ_DP_GRAPH=1): on one MI355X with a 1-rank RCCL group the
    # headline workload is GPU-bound either way (HNeRV-3M: 435.5 it/s eager, 1.30 ms of host work per 2.30 ms step, vs 432.8
    # captured), while NeRV-3M + Hadamard is host-bound when eager (648 -> 883 it/s captured); and a capture next to a live
    # process group has one more failure mode than eager launches (RCCL's watchdog thread polls its events: the captures
    # below therefore use the thread-local capture mode), so the driver's multi-GPU runs keep the path proven in round 2
    # unless asked otherwise.
    dp_graph = use_graph and dp and os.environ.get("NQ_DP_GRAPH", "0") == "1"
    if dp and not dp_graph:
        use_graph = False
    fused_stack = None
    if dp_graph:
        from ..models._decode import _fused_stack
        fused_stack = _fused_stack(model.model) if hasattr(model, 'model') else None
        if fused_stack is None:   # decoder not fusable into one node: eager data-parallel iterations
            dp_graph = use_graph = False

    def _capture_staged(run_body):
        """Capture one data-parallel iteration as consecutive graphs cut at every complete part of the gradient arena:
        -> ([graphs], [arena parts]) with len(graphs) == len(parts) + 1.  One memory pool, one capture stream."""
        graphs, parts = [torch.cuda.CUDAGraph()], []
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        # thread-local capture mode: every launch of the staged body comes from THIS thread (no autograd worker), and the
        # process group's watchdog thread must stay free to query its events while we capture (in the default global mode
        # its hipEventQuery aborts the process: "operation not permitted when stream is capturing")
        with torch.cuda.stream(cap):
            graphs[0].capture_begin(pool=pool, capture_error_mode="thread_local")

            def cut(k, part):
                graphs[-1].capture_end()
                parts.append(part)
                graphs.append(torch.cuda.CUDAGraph())
                graphs[-1].capture_begin(pool=pool, capture_error_mode="thread_local")

            run_body(cut)
            graphs[-1].capture_end()
        torch.cuda.current_stream().wait_stream(cap)
        return graphs, parts

    def _replay_staged(staged_graph):
        """graph 0 | all-reduce(part 0) | graph 1 | all-reduce(part 1) | ... | wait for the collectives | last graph."""
        import torch.distributed as dist
        graphs, parts = staged_graph
        works = []
        for g, part in zip(graphs[:-1], parts):
            g.replay()
            works.append(_mean_all_reduce(part))
        for fin in works:   # the compute stream waits for them (no host sync)
            fin()
        graphs[-1].replay()

    # Hadamard runs: the fake-quant and the transform share one launch per direction (round 4) when every layer's rows fit the
    # fused kernels' tile (all shipped models); NQ_FUSED_FWHT=0 keeps the separate launches (bit-identical either way, tested)
    fuse_had = bool(hadamard) and os.environ.get("NQ_FUSED_FWHT", "1") != "0" and \
        all(ops.fq_fwht_fusable(L.n, L.src.shape[2] * L.src.shape[3]) for L in layers)

    def run(epochs, params, opt_lr, max_count, ada):
        nonlocal done
        opt = ops.FusedAdam(params, lr=opt_lr)
        loss_start = max_count * warmup
        temp = LinearTempDecay(max_count, rel_start_decay=warmup, start_b=b_range[0], end_b=b_range[1])
        count = 0

        def sched(c):
            """(b, regulariser on) of the iteration with counter c (calib_model.py:62, 76-81)."""
            b = temp(c)
            reg_on = ada and not (c < loss_start)
            return (b if reg_on else 0), reg_on

        def body(get_batch, b, reg_on, dyn, want_log, staged=None):
            """One iteration.  dyn = None: scalars travel as host arguments (b, reg_on, Adam's t); else they are read from
            the device slots `dyn` = {b, gate, lr/(1-beta1^t), sqrt(1-beta2^t)} filled by nq_step_prologue.
            staged(k, part): data-parallel captured iterations -- the decoder runs without autograd and `staged` is called
            with every complete part of the gradient arena (k = 0, 1), where the caller cuts the graph / reduces the part."""
            img, inputs = get_batch()
            if ada and fuse_had:   # fake-quant + H of all layers (and the fake-quant of the biases) in ONE launch
                items = []
                for L in layers:
                    (xw, aw, dw_, zw, nlw, sw), (xb, ab, db_, zb, nlb, sb) = L.ada_items()
                    items += [(xw, aw, dw_, zw, nlw, sw, L.n, L.c_in), (xb, ab, db_, zb, nlb, sb, 0, 0)]
                fq = ops.adaround_fwht_multi(items)
                for i, L in enumerate(layers):
                    L._finish(fq[2 * i], fq[2 * i + 1], transformed=True)
            elif ada:   # all 14 fake-quantised tensors in one launch
                fq = ops.adaround_forward_multi(
                    [it for L in layers for it in L.ada_items()])
                if hadamard:   # H(Q(H w)) of every layer in one launch (quant_layer.py:70-71)
                    had = ops.fwht_channels_multi([(fq[2 * i], L.n, L.c_in) for i, L in enumerate(layers)])
                    for i, L in enumerate(layers):
                        L._finish(had[i], fq[2 * i + 1], transformed=True)
                else:
                    for i, L in enumerate(layers):
                        L._finish(fq[2 * i], fq[2 * i + 1])
            else:   # phase 1: the UAQ fake-quant of all 14 tensors in one launch (+ one FWHT launch with Hadamard)
                fq = ops.uaq_forward_multi([it for L in layers for it in L.uaq_items()])
                if hadamard:
                    had = ops.fwht_channels_multi([(fq[2 * i], L.n, L.c_in) for i, L in enumerate(layers)])
                    for i, L in enumerate(layers):
                        L._finish(had[i], fq[2 * i + 1], transformed=True)
                else:
                    for i, L in enumerate(layers):
                        L._finish(fq[2 * i], fq[2 * i + 1])
            node = None
            # (the loss tail runs behind the head convolution when the decoder is the fused stack: ops.fused_head_loss)
            loss_req = ops.fused_head_loss(cache_u8=img[0], idx=img[1]) if isinstance(img, tuple) else ops.fused_head_loss(tgt=img)
            with loss_req:
                if staged is None:
                    img_out, _, _ = model(inputs)
                else:   # the same decoder node, driven by hand (weights as model._wb_override set them, in layer order)
                    spec, provs = fused_stack
                    img_out, node = ops.decoder_forward_manual(inputs, spec, [p() for p in provs],
                                                               two_phase=os.environ.get("NQ_DP_OVERLAP", "1") != "0")
            # lp_loss p=2 (quantizer.py:66-71) and its gradient; behind a tanh-headed fused decoder the loss kernel also
            # applies the tanh backward and sums the head's bias gradient (ops.l2_loss_head_grad), reading the target
            # straight from the uint8 frame cache when the batch came as (frames_u8, indices)
            fused = ops.l2_loss_head_grad(img_out, cache_u8=img[0], idx=img[1], node=node) if isinstance(img, tuple) \
                else ops.l2_loss_head_grad(img_out, tgt=img, node=node)
            if fused is None:
                if isinstance(img, tuple):
                    img = ops.gather_frames_u8(img[0], img[1])
                fused = ops.l2_loss_and_grad(img_out, img)
            rec, dimg = fused
            if staged is None:
                img_out.backward(dimg)
            else:
                steps, k = ops.decoder_backward_steps(node, dimg), 0
                while True:
                    try:
                        part, _last = next(steps)
                    except StopIteration as fin:
                        res = fin.value
                        break
                    staged(k, part)
                    k += 1
                for i, L in enumerate(layers):   # what autograd would have left in .grad
                    L.W.grad, L.b.grad = res[2 + 2 * i], res[3 + 2 * i]
            if dp and staged is None and not ops.arena_reduced(img_out):
                # generic path (decoder not fused into one node): flatten, all-reduce, un-flatten
                allreduce_mean_([t.grad for L in layers for t in (L.W, L.b)])
            # (the logged rounding loss is the forward value: alpha BEFORE this iteration's update -- the fused kernel below
            # updates alpha in place)
            if want_log:
                rl = torch.zeros((), device=device)
                if reg_on:
                    for L in layers:
                        ops.round_loss(L.m.weight_quantizer.alpha.data, b, weight, out=rl, accumulate=True)
                total, rl_f = float(rec) + float(rl), float(rl)
                if recorder is not None:
                    recorder.append((total, rl_f, float(b), count))
                if count % 500 == 0:
                    logging.info('Total loss:\t{:.4f} (rec:{:.4f}, round:{:.4f})\tb={:.2f}\tcount={}'.format(
                        total, float(rec), rl_f, b, count))
            grads = []
            if ada and fuse_had and probe is None and os.environ.get("NQ_FUSED_ADAM", "1") != "0":
                # H on the zero-padded gradients + d(alpha) + regulariser gradient + Adam of all layers in ONE launch: the
                # transform-domain gradient never goes to memory
                items = []
                for L in layers:
                    wq, bq = L.m.weight_quantizer, L.m.bias_quantizer
                    items.append((L.src, L.W.grad, wq.alpha.data, wq.delta.data, wq.zero_point, wq.n_levels,
                                  weight if (reg_on or dyn is not None) else 0.0, L.n, L.c_in))
                    items.append((L.bias, L.b.grad, bq.alpha.data, bq.delta.data, bq.zero_point, bq.n_levels, 0.0, 0, 0))
                ops.fwht_adaround_adam_multi(items, opt, b, dyn=dyn)
                grads = None
            elif ada:   # d(alpha) of all 14 tensors (+ regulariser gradient on the weights) in one launch
                items = []
                # H on the zero-padded gradients of all layers in one launch
                gWs = ops.fwht_channels_multi([(L.W.grad, L.n, L.n) for L in layers]) if hadamard else None
                for i, L in enumerate(layers):
                    gW, gb = (gWs[i], L.b.grad) if hadamard else L.grads()
                    wq, bq = L.m.weight_quantizer, L.m.bias_quantizer
                    # with dyn the regulariser weight is always passed and gated on the device by dyn[1]
                    items.append((L.src, gW, wq.alpha.data, wq.delta.data, wq.zero_point, wq.n_levels,
                                  weight if (reg_on or dyn is not None) else 0.0))
                    items.append((L.bias, gb, bq.alpha.data, bq.delta.data, bq.zero_point, bq.n_levels, 0.0))
                if probe is None and os.environ.get("NQ_FUSED_ADAM", "1") != "0":
                    # nobody looks at d(alpha): backward of the fake-quant + regulariser gradient + Adam in ONE pass
                    ops.adaround_adam_multi(items, opt, b, dyn=dyn)
                    grads = None
                else:
                    grads = ops.adaround_backward_multi(items, b, dyn=dyn)
            else:   # d(delta) of all 14 tensors in one launch
                gWs = ops.fwht_channels_multi([(L.W.grad, L.n, L.n) for L in layers]) if hadamard else None
                items = []
                for i, L in enumerate(layers):
                    gW, gb = (gWs[i], L.b.grad) if hadamard else L.grads()
                    wq, bq = L.m.weight_quantizer, L.m.bias_quantizer
                    items.append((L.src, gW, wq.delta.data, wq.zero_point, wq.n_levels))
                    items.append((L.bias, gb, bq.delta.data, bq.zero_point, bq.n_levels))
                grads = ops.uaq_backward_multi(items)
            if probe is not None:
                probe('ada' if ada else 'uaq', layers, grads)
            if grads is not None:
                opt.step(grads, dyn=dyn)
            for L in layers:
                L.release()

        if not use_graph:
            for _ in range(epochs):
                model.train()
                for sample in gt:
                    if step_hook is not None:
                        step_hook(done)
                    if max_steps is not None and done >= max_steps:
                        return
                    count += 1
                    b, reg_on = sched(count)
                    img = sample['img'].to(device, non_blocking=True)
                    inputs = cali_data[sample['idx'].to(device, non_blocking=True)]
                    body(lambda: (img, inputs), b, reg_on, None, recorder is not None or count % 500 == 0)
                    done += 1
            return

        st = None       # static device state of this phase's captured iteration
        graph, warm = None, 0
        for _ in range(epochs):
            model.train()
            idx_tab = gt.epoch_indices()                     # (batches, frames per batch on this rank), device int64
            nb, per = idx_tab.shape
            if st is None or st['order'].shape[0] < nb or st['cur_idx'].numel() != per:
                st = dict(order=torch.zeros((nb, per), device=device, dtype=torch.int64),
                          scal=torch.zeros((nb, 4), device=device, dtype=torch.float32),
                          step=torch.zeros(2, device=device, dtype=torch.int32),     # {step counter, arrival ticket}
                          emb=torch.zeros((per,) + tuple(cali_data.shape[1:]), device=device, dtype=torch.float32),
                          cur_idx=torch.zeros(per, device=device, dtype=torch.int64),
                          cur_scal=torch.zeros(4, device=device, dtype=torch.float32))
                graph, warm = None, 0
            rows = []
            for c in range(count + 1, count + nb + 1):
                b, reg_on = sched(c)
                rows.append((float(b), 1.0 if reg_on else 0.0) + opt.scalars(opt.t + (c - count)))
            st['scal'][:nb].copy_(torch.tensor(rows, dtype=torch.float32), non_blocking=True)
            st['order'][:nb].copy_(idx_tab)
            st['step'].zero_()

            def static_batch():
                # current batch indices + scalars into their fixed slots and the batch's decoder inputs cali_data[idx], one launch
                ops.step_prologue_gather(st['order'], st['scal'], st['step'], st['cur_idx'], st['cur_scal'], cali_data, st['emb'])
                return (gt.cache.frames, st['cur_idx']), st['emb']

            for i in range(nb):
                if step_hook is not None:
                    step_hook(done)
                if max_steps is not None and done >= max_steps:
                    return
                count += 1
                b, reg_on = sched(count)
                want_log = count % 500 == 0
                if want_log or warm < 3 or ops.profiling_active():
                    body(static_batch, b, reg_on, st['cur_scal'], want_log)
                    warm += 1
                elif graph is None and dp_graph:
                    graph = _capture_staged(lambda cut: body(static_batch, b, reg_on, st['cur_scal'], False, staged=cut))
                    _replay_staged(graph)
                elif graph is None:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        body(static_batch, b, reg_on, st['cur_scal'], False)   # recorded, not executed (opt.t advanced)
                    graph.replay()
                elif dp_graph:
                    _replay_staged(graph)
                    opt.t += 1
                else:
                    graph.replay()
                    opt.t += 1
                done += 1

    with hook_scope:
        # ---- phase 1: scales (calib_model.py:119-165; lr and max_count are hard-coded there) ----
        params = []
        for L in layers:
            params += [L.m.weight_quantizer.delta, L.m.bias_quantizer.delta]
        epochs1 = int(0.05 * iters / len(gt))
        run(epochs1, params, 0.001, 2100, ada=False)
        torch.cuda.empty_cache()

        # ---- phase 2: rounding variables (calib_model.py:170-226) ----
        params = []
        for L in layers:
            m = L.m
            m.weight_quantizer = AdaRoundQuantizer(uaq=m.weight_quantizer, round_mode='learned_hard_sigmoid',
                                                   weight_tensor=(m.hadamard_weight if hadamard else m.org_weight).data)
            m.bias_quantizer = AdaRoundQuantizer(uaq=m.bias_quantizer, round_mode='learned_hard_sigmoid',
                                                 weight_tensor=m.bias.data)
            m.weight_quantizer.soft_targets = True
            m.bias_quantizer.soft_targets = True
            params += [m.weight_quantizer.alpha, m.bias_quantizer.alpha]
        run(int(iters / len(gt)) - epochs1, params, lr, iters, ada=True)
        torch.cuda.empty_cache()

    # ---- finish: weights go hard; the bias quantisers stay soft, as in the reference (calib_model.py:231-240) ----
    for L in layers:
        L.m.weight_quantizer.soft_targets = False
