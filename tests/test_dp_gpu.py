"""Data-parallel path on the GPU with a 1-rank RCCL group (NQ_DP_REHEARSAL=1): the gradient-arena hook + in-place
all_reduce(ReduceOp.AVG) of model_reconstruction and the driver's process-group / sharded-loader wiring
(methods/calibrate_network.py:dist_setup) run inside the GPU suite, and must be bit-identical to the single-process run
(AVG over one rank is the identity).  The N > 1 arithmetic is covered on CPU by tests/test_dp_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import ROOT, T, BITS, TINY_HNERV, state_dict_from_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture
def rehearsal_env(monkeypatch):
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"),
                 ("MASTER_PORT", str(_free_port())), ("NQ_DP_REHEARSAL", "1"), ("HSA_ENABLE_IPC_MODE_LEGACY", "0")):
        monkeypatch.setenv(k, v)
    yield
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


class _Replay:
    def __init__(self, frames, order):
        self.frames, self.order, self.pos = frames, order, 0

    def __len__(self):
        return self.order.shape[1]

    def __iter__(self):
        ep = self.order[self.pos]
        self.pos += 1
        for idx in ep:
            t = torch.as_tensor(idx, dtype=torch.int64, device=DEV)
            yield {"img": self.frames[t], "idx": t, "norm_idx": t.float() / 8}


def _calibrate(golden, iters=80):
    from neuroquant_amd.models import HNeRV
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    z = golden("traj_hnerv.npz")
    frames = (T(golden("frames_320x640.npz")["frames"]).float() / 255.0).to(DEV)
    model = HNeRV(TINY_HNERV)
    model.load_state_dict(state_dict_from_npz(z, "sd:"))
    model = model.to(DEV).eval()
    emb = T(z["emb"]).to(DEV)
    qnn = QuantModel(model, hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    qnn.set_bitwidth(BITS)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:2])
    rec = []
    model_reconstruction(qnn, cali_data=emb, gt=_Replay(frames, z["order"]), arch="hnerv", batch_size=2, iters=iters,
                         weight=0.01, hadamard=False, b_range=(20, 2), warmup=0.2, lr=0.003, recorder=rec)
    return np.array(rec), [m.weight_quantizer.alpha.detach().clone() for m in qnn.quant_modules()], \
        [m.weight_quantizer.delta.detach().clone() for m in qnn.quant_modules()]


def test_rccl_one_rank_group_is_bit_identical(golden, rehearsal_env, monkeypatch):
    """80 iterations (4 phase-1 + 76 phase-2) of the tiny HNeRV: plain run vs the run with the RCCL hook installed (the
    data-parallel schedule: all data gradients first, two asynchronous in-place all-reduces)."""
    import torch.distributed as dist
    from neuroquant_amd import ops
    monkeypatch.delenv("NQ_DP_REHEARSAL")
    log0, a0, d0 = _calibrate(golden)
    monkeypatch.setenv("NQ_DP_REHEARSAL", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    calls = []
    orig = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.numel())
        return orig(t, *a, **k)

    monkeypatch.setattr(dist, "all_reduce", counting)
    log1, a1, d1 = _calibrate(golden)
    assert ops._arena_state() == (None, False)                # removed again at the end of model_reconstruction
    n_arena = sum(m.numel() for m in a0) + sum(int(a.shape[0]) for a in a0)   # all conv weights + biases
    # TWO collectives per iteration that together cover the whole arena: the deep layers' part first (it travels while
    # the last layers' weight gradients are computed), then the rest
    assert len(calls) == 160 and all(a + b == n_arena and a > 0 and b > 0 for a, b in zip(calls[0::2], calls[1::2]))
    np.testing.assert_array_equal(log0, log1)
    for x, y in zip(a0 + d0, a1 + d1):
        assert torch.equal(x, y)


def test_driver_joins_the_group_and_shards_the_loader(rehearsal_env, tmp_path, monkeypatch):
    """methods/calibrate_network.calibrate under the torchrun environment: binds the GPU, initialises RCCL before any
    other GPU call, builds the rank-sharded loader, logs / saves on rank 0 -- same PSNRs and parameters as the plain run."""
    import torch.distributed as dist
    from neuroquant_amd.methods import calibrate_network as cn

    def run():
        args = cn.parse_args(["--arch", "hnerv", "--synthetic", "8", "--batch_size", "2", "--channel_wise", "--init", "max",
                              "--iters_w", "80", "--weight", "0.01", "--b_start", "20", "--b_end", "2", "--warmup", "0.2",
                              "--lr", "0.003", "--precision", "6", "5", "4", "5", "5", "6", "6"])
        args.outf = str(tmp_path / f"o{len(os.listdir(tmp_path))}")
        cn.seed_all(903)
        res = cn.calibrate(args, dict(TINY_HNERV))
        return [float(r) for r in res], [m.weight_quantizer.alpha.detach().clone() for m in args.qnn.quant_modules()]

    monkeypatch.delenv("RANK")
    plain, a0 = run()
    assert not dist.is_initialized()
    monkeypatch.setenv("RANK", "0")
    dp, a1 = run()
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    assert plain == dp
    for x, y in zip(a0, a1):
        assert torch.equal(x, y)


# ------------------------------------------------------------------------------------------------------------------
# world = 2 emulated in ONE process: the arena / ReduceOp.AVG path under a reduction that is NOT the identity
# ------------------------------------------------------------------------------------------------------------------
class _DoneWork:
    def wait(self):
        return True


def _order4(epochs=80, seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randperm(8, generator=g).view(2, 4) for _ in range(epochs)]).numpy()


def _tiny_qnn(golden):
    from neuroquant_amd.models import HNeRV
    from neuroquant_amd.quantization import QuantModel
    z = golden("traj_hnerv.npz")
    model = HNeRV(TINY_HNERV)
    model.load_state_dict(state_dict_from_npz(z, "sd:"))
    model = model.to(DEV).eval()
    emb = T(z["emb"]).to(DEV)
    qnn = QuantModel(model, hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    qnn.set_bitwidth(BITS)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:2])
    return qnn, emb, z


_DP_FLAGS = dict(arch="hnerv", batch_size=4, iters=40, weight=0.01, hadamard=False, b_range=(20, 2), warmup=0.0, lr=0.003)
_DP_STEPS = 5      # len(gt) = 2, iters = 40 -> 1 phase-1 epoch (2 iterations), then 3 phase-2 iterations (regulariser on)


def _engine_run(golden, monkeypatch, rank, world, other, steps=_DP_STEPS, iters=None, inspect=True):
    """One `model_reconstruction` of `_DP_STEPS` iterations on rank `rank` of `world`; world = 2 replaces dist.all_reduce by
    an average with the OTHER rank's recorded arena part of the same iteration (`other[step][offset]`, or the rank's own
    data where no record exists yet).  -> dict(parts, grads, arena, log, delta, alpha)."""
    import torch.distributed as dist
    from neuroquant_amd.quantization import model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache
    qnn, emb, _ = _tiny_qnn(golden)
    frames_u8 = T(golden("frames_320x640.npz")["frames"]).to(DEV)
    loader = CacheLoader(FrameCache(frames_u8), list(range(8)), 4, rank=rank, world=world, order=_order4(80))
    out = dict(parts=[[] for _ in range(steps)], grads=[], arena=[], phases=[])
    cur = {"step": -1}

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        assert op == dist.ReduceOp.AVG and async_op, "the arena is averaged in place, asynchronously"
        assert t.is_contiguous() and t.dtype == torch.float32
        st, off = cur["step"], t.storage_offset()
        mine = t.detach().clone()
        out["parts"][st].append((off, mine, t.untyped_storage().data_ptr()))
        theirs = mine
        if other is not None and off in other[st]:
            theirs = other[st][off]
            assert theirs.shape == mine.shape, "both ranks cut the arena at the same place"
        t.copy_((mine + theirs) * 0.5)          # ReduceOp.AVG over two ranks (commutative: replicas stay bit-identical)
        return _DoneWork()

    if world > 1:
        monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)

    def probe(phase, layers, grads):
        out["phases"].append(phase)
        out["arena"].append([t.grad.clone() for L in layers for t in (L.W, L.b)])     # what the parameter side consumes
        out["grads"].append([g.clone() for g in grads])

    rec = []
    flags = dict(_DP_FLAGS, iters=iters or _DP_FLAGS["iters"])
    # inspect = False: no recorder / probe -> the engine may replay captured iterations (three graphs, collectives between)
    model_reconstruction(qnn, cali_data=emb, gt=loader, recorder=rec if inspect else None, max_steps=steps,
                         probe=probe if inspect else None,
                         step_hook=lambda done: cur.__setitem__("step", min(done, steps - 1)), **flags)
    out["log"] = np.array(rec)
    out["delta"] = [q.delta.detach().clone() for m in qnn.quant_modules() for q in (m.weight_quantizer, m.bias_quantizer)]
    out["alpha"] = [q.alpha.detach().clone() for m in qnn.quant_modules() for q in (m.weight_quantizer, m.bias_quantizer)]
    return out


def _records(run):
    return [{off: t for off, t, _ in step} for step in run["parts"]]


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_emulated_world2_arena_reduction(golden, rehearsal_env, monkeypatch, overlap):
    """BASELINE configs[3]'s exchange on one GPU: global batch 4 = 2 ranks x 2 frames.  Each rank runs the PRODUCT path
    (`model_reconstruction` with torch.distributed up -> ops.grad_arena_hook -> asynchronous in-place all_reduce(AVG) on
    views of one arena) with the collective replaced by the average with the other rank's recorded arena part.  Replicas
    hold identical parameters, so rank r's arena of iteration t only depends on the exchanges of iterations < t: sweep k
    replays both ranks against the other's records of sweep k-1 and is exact for iterations <= k; the last two sweeps
    must agree bit for bit (fixed point = lock-step execution).  Checked on the exact sweep:
      * every arena element is reduced exactly once per iteration (the parts tile [0, #conv parameters));
      * both replicas end bit-identical (parameter gradients of every iteration, final delta / alpha);
      * the reduced arena, d(delta), d(alpha) -- regulariser on, computed locally, counted once -- and the losses equal
        the single-process global-batch-4 run to summation-order tolerance, and the CPU oracle at B = 4."""
    import torch.distributed as dist
    from oracle import nq_oracle as O
    monkeypatch.setenv("NQ_DP_OVERLAP", overlap)
    monkeypatch.delenv("NQ_DP_REHEARSAL")
    dist.init_process_group("gloo", rank=0, world_size=1)
    single = _engine_run(golden, monkeypatch, 0, 1, None)           # the whole batch of 4 on one device, no exchange
    assert single["phases"] == ["uaq", "uaq", "ada", "ada", "ada"]
    monkeypatch.setenv("NQ_DP_REHEARSAL", "1")                      # world_size 1 group, data-parallel path on

    recs, runs = [None, None], None
    for sweep in range(_DP_STEPS + 1):
        prev = runs
        runs = [_engine_run(golden, monkeypatch, r, 2, recs[1 - r]) for r in (0, 1)]
        recs = [_records(runs[0]), _records(runs[1])]
    for r in (0, 1):                                                # fixed point: the records the last sweep consumed are
        for a, b in zip(prev[r]["parts"], runs[r]["parts"]):        # the records it produced
            assert len(a) == len(b) and all(x[0] == y[0] and torch.equal(x[1], y[1]) for x, y in zip(a, b))

    n_arena = sum(t.numel() for t in single["arena"][0])
    for r in (0, 1):
        for step in runs[r]["parts"]:
            assert len(step) == (2 if overlap == "1" else 1)
            assert len({base for _, _, base in step}) == 1                          # views of ONE arena
            pos = 0
            for off, t, _ in sorted(step, key=lambda p: p[0]):
                assert off == pos and t.numel() > 0                                 # disjoint, no gap
                pos += t.numel()
            assert pos == n_arena                                                   # ... and complete
    # the two shards' own gradients really differ (the reduction is not the identity)
    a0 = torch.cat([t for _, t, _ in sorted(runs[0]["parts"][0], key=lambda p: p[0])])
    a1 = torch.cat([t for _, t, _ in sorted(runs[1]["parts"][0], key=lambda p: p[0])])
    assert float((a0 - a1).abs().max()) > 1e-3 * float(a0.abs().max())

    # replicas: bit-identical everywhere
    for key in ("arena", "grads"):
        for s0, s1 in zip(runs[0][key], runs[1][key]):
            assert all(torch.equal(x, y) for x, y in zip(s0, s1))
    for key in ("delta", "alpha"):
        assert all(torch.equal(x, y) for x, y in zip(runs[0][key], runs[1][key]))

    def rel(a, b):
        return float((a.double() - b.double()).abs().max()) / (float(b.double().abs().max()) + 1e-30)

    # vs the single-process global batch: same mathematics (mean over 4 frames = mean of two means over 2), another
    # summation order.  Bounds: 1e-4 of each tensor's largest entry on the arena (measured 3e-5 under bf16x3: 2^-17 per
    # product, tests/test_full_size.py uses the same bound); d(delta) 1e-3 (a sum of cancelling terms,
    # tests/test_full_size.py); d(alpha) 1e-4.  From the second iteration on the two runs no longer hold the same
    # parameters bit for bit (Adam turns a last-bit gradient difference into a +-lr step where the gradient is ~0; measured
    # 6e-4 on the arena of the first phase-2 iteration, after two such steps of delta), so only the very first iteration is
    # held to the tight max-norm bound.  Later, a channel whose delta took another +-lr step rounds differently
    # (floor(x / delta)) and single elements of d(alpha) differ completely (measured: 0.2 of the largest entry on one
    # element, 6e-4 on the arena), so those iterations are held element-wise instead: fewer than 2 % of a tensor's entries
    # may differ by more than 1e-3 of its largest one -- a regulariser gradient counted twice (or halved, or averaged
    # with a stale one) changes nearly ALL entries of every d(alpha) by far more: it dominates d(alpha) at b = 20.
    def frac_off(a, b):
        """share of entries further apart than 1e-3 of the largest one; only tensors of >= 1000 entries are judged this way
        (a 12-entry bias gradient has no statistics: one re-rounded channel is 8 % of it)"""
        if a.numel() < 1000:
            return 0.0 if bool(torch.isfinite(a).all()) else 1.0
        return float(((a.double() - b.double()).abs() > 1e-3 * float(b.double().abs().max()) + 1e-30).double().mean())

    for st, phase in enumerate(single["phases"]):
        for x, y in zip(runs[0]["arena"][st], single["arena"][st]):
            if st == 0:
                assert rel(x, y) < 1e-4, (st, rel(x, y))
            else:
                assert frac_off(x, y) < 0.02, (st, frac_off(x, y))
        for x, y in zip(runs[0]["grads"][st], single["grads"][st]):
            if st == 0:
                assert rel(x, y) < 1e-3, (st, phase, rel(x, y))
            else:
                assert frac_off(x, y) < 0.02, (st, phase, frac_off(x, y))
    # the check has teeth: doubling the regulariser's share of d(alpha) would be seen (reg = d(alpha) - rec part is not
    # available separately, so scale the whole gradient of one weight tensor by the factor a double count would give it)
    g_w = single["grads"][2][8]                      # d(alpha) of a mid-size weight tensor, first phase-2 iteration
    assert frac_off(g_w * 1.5, g_w) > 0.2
    # losses: global reconstruction loss = mean of the two ranks' local means; the regulariser is the same number everywhere
    rec_dp = 0.5 * ((runs[0]["log"][:, 0] - runs[0]["log"][:, 1]) + (runs[1]["log"][:, 0] - runs[1]["log"][:, 1]))
    np.testing.assert_allclose(rec_dp, single["log"][:, 0] - single["log"][:, 1], rtol=2e-4)
    np.testing.assert_array_equal(runs[0]["log"][:, 1:], runs[1]["log"][:, 1:])
    np.testing.assert_allclose(runs[0]["log"][:, 1], single["log"][:, 1], rtol=2e-4)
    assert float(single["log"][2:, 1].min()) > 0, "regulariser on in the phase-2 iterations"
    for x, y in zip(runs[0]["delta"] + runs[0]["alpha"], single["delta"] + single["alpha"]):
        # parameters after 2 + 3 Adam steps: equal up to the rare +-lr flips described above (measured: 1 of a 50-row delta)
        assert float(((x - y).abs() > 1e-4 * float(y.abs().max())).float().mean()) < 0.05

    # vs the CPU oracle at B = 4 (first iteration of each phase: gradients; every iteration: losses)
    z = golden("traj_hnerv.npz")
    dec = O.Decoder.from_state_dict({k: v for k, v in state_dict_from_npz(z, "sd:").items() if not k.startswith("encoder")},
                                    "hnerv", TINY_HNERV["dec_strides"])
    qs = O.QuantStack(dec, BITS, hadamard=False)
    frames = T(golden("frames_320x640.npz")["frames"]).float() / 255.0
    ref = dict(arena=[], grads=[])

    def oprobe(phase, qs_, fq):
        ref["arena"].append([t.grad.clone() for W, b in fq for t in (W, b)])
        ref["grads"].append([t.grad.clone() for L in qs_.dec.layers
                             for t in ((L.wd, L.bd) if phase == "uaq" else (L.wa, L.ba))])

    flags = {k: _DP_FLAGS[k] for k in ("weight", "b_range", "warmup", "lr")}
    olog = np.array(O.calibrate(qs, T(z["emb"]), frames, _order4(), _DP_FLAGS["iters"], max_steps=_DP_STEPS, probe=oprobe,
                                **flags))
    np.testing.assert_allclose(0.5 * (runs[0]["log"][:, 0] + runs[1]["log"][:, 0]), olog[:, 0], rtol=1e-3)
    for st in (0, 2):
        for x, y in zip(runs[0]["arena"][st], ref["arena"][st]):
            if st == 0:
                assert rel(x.cpu(), y) < 1e-4, (st, rel(x.cpu(), y))
            else:
                assert frac_off(x.cpu(), y) < 0.02, (st, frac_off(x.cpu(), y))
    for x, y in zip(runs[0]["grads"][0], ref["grads"][0]):
        assert rel(x.cpu().reshape(-1), y.reshape(-1)) < 1e-3


def test_captured_dp_iterations_equal_eager(golden, rehearsal_env, monkeypatch):
    """Data-parallel iterations replayed from graphs (round 3): with torch.distributed up the engine
    captures an iteration as up to three graphs -- forward + data gradients + the deep layers' weight gradients | the last
    layers' weight gradients | d(alpha) + Adam -- and launches the two asynchronous all_reduce(AVG) calls eagerly between
    their replays.  (a) on a 1-rank RCCL group, 100 iterations (20 phase-1 + 80 phase-2): final delta / alpha bit-identical
    to eager data-parallel iterations (NQ_DP_GRAPH=0) and to the run without torch.distributed, two collectives per
    iteration in both modes; (b) world = 2 emulated as in test_emulated_world2_arena_reduction: the captured path fed
    with the other rank's recorded arena parts reproduces the eager path's records and parameters bit for bit."""
    import torch.distributed as dist
    from neuroquant_amd.quantization import model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache
    frames_u8 = T(golden("frames_320x640.npz")["frames"]).to(DEV)

    def run(dp_graph):
        monkeypatch.setenv("NQ_DP_GRAPH", dp_graph)
        qnn, emb, _ = _tiny_qnn(golden)
        loader = CacheLoader(FrameCache(frames_u8), list(range(8)), 2, order=golden("traj_hnerv.npz")["order"])
        model_reconstruction(qnn, cali_data=emb, gt=loader, arch="hnerv", batch_size=2, iters=440, weight=0.01, hadamard=False,
                             b_range=(20, 2), warmup=0.2, lr=0.003, max_steps=100)
        torch.cuda.synchronize()
        return [q.alpha.detach().clone() for m in qnn.quant_modules() for q in (m.weight_quantizer, m.bias_quantizer)] + \
            [q.delta.detach().clone() for m in qnn.quant_modules() for q in (m.weight_quantizer, m.bias_quantizer)]

    monkeypatch.delenv("NQ_DP_REHEARSAL")
    plain = run("0")                                   # no process group: the single captured graph
    monkeypatch.setenv("NQ_DP_REHEARSAL", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    calls = []
    orig = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.numel())
        return orig(t, *a, **k)

    monkeypatch.setattr(dist, "all_reduce", counting)
    eager = run("0")
    n_eager = len(calls)
    calls.clear()
    staged = run("1")
    assert n_eager == 200 and len(calls) == 200        # two collectives per iteration, captured or not
    for a, b, c in zip(plain, eager, staged):
        assert torch.equal(a, b) and torch.equal(a, c)
    dist.destroy_process_group()

    # (b) emulated world = 2: eager sweeps to the lock-step fixed point, then ONE captured run per rank against those records
    monkeypatch.setenv("NQ_DP_OVERLAP", "1")
    monkeypatch.setenv("NQ_DP_GRAPH", "0")
    dist.init_process_group("gloo", rank=0, world_size=1)
    steps, iters = 12, 120                             # len(gt) = 2: 3 phase-1 epochs = 6 iterations, then 6 phase-2 iterations
    recs, runs = [None, None], None
    for sweep in range(steps + 1):
        runs = [_engine_run(golden, monkeypatch, r, 2, recs[1 - r], steps=steps, iters=iters, inspect=False) for r in (0, 1)]
        recs = [_records(runs[0]), _records(runs[1])]
    monkeypatch.setenv("NQ_DP_GRAPH", "1")
    cap = [_engine_run(golden, monkeypatch, r, 2, recs[1 - r], steps=steps, iters=iters, inspect=False) for r in (0, 1)]
    for r in (0, 1):
        for a, b in zip(runs[r]["parts"], cap[r]["parts"]):
            assert len(a) == len(b) == 2 and all(x[0] == y[0] and torch.equal(x[1], y[1]) for x, y in zip(a, b))
        for key in ("delta", "alpha"):
            assert all(torch.equal(x, y) for x, y in zip(runs[r][key], cap[r][key]))
    assert all(torch.equal(x, y) for x, y in zip(cap[0]["alpha"] + cap[0]["delta"], cap[1]["alpha"] + cap[1]["delta"]))


def test_two_processes_gloo_on_gpu(golden, tmp_path, monkeypatch):
    """A REAL world of two: two processes on this GPU, different frames on each, a genuine collective between them (gloo:
    RCCL refuses two ranks on one device) through the product path -- sharded CacheLoader, ops.grad_arena_hook, the
    asynchronous in-place mean all-reduce of the two arena parts.  Both replicas
    must end bit-identical, and equal the single-process run of the global batch to summation-order tolerance."""
    import subprocess
    import sys
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NQ_DP_REHEARSAL"):
        env.pop(k, None)
    outs = [str(tmp_path / f"rank{r}.npz") for r in (0, 1)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), outs[r]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in (0, 1)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    r0, r1 = np.load(outs[0]), np.load(outs[1])
    keys = [k for k in r0.files if k != "log"]
    assert keys and all(np.array_equal(r0[k], r1[k]) for k in keys), "replicas diverged"
    # each rank logs ITS shard's reconstruction loss: they differ (different frames), their mean is the global-batch loss
    assert r0["log"].shape == r1["log"].shape == (5, 4)
    assert np.abs(r0["log"][:, 0] - r1["log"][:, 0]).max() > 0

    single = _engine_run(golden, monkeypatch, 0, 1, None)          # the whole batch of 4 in this process, no exchange
    rec_mean = 0.5 * ((r0["log"][:, 0] - r0["log"][:, 1]) + (r1["log"][:, 0] - r1["log"][:, 1]))
    rec_single = single["log"][:, 0] - single["log"][:, 1]
    assert np.abs(rec_mean - rec_single).max() <= 2e-4 * np.abs(rec_single).max(), (rec_mean, rec_single)
    qi = 0
    for i in range(len(single["delta"]) // 2):
        for tag in ("w", "b"):
            d, a = single["delta"][qi].cpu().numpy(), single["alpha"][qi].cpu().numpy()
            qi += 1
            assert np.abs(r0[f"delta_{tag}{i}"] - d).max() <= 1e-3 * np.abs(d).max()
            if a.size >= 1000:   # alpha moves by lr per step: a few elements flip sign of a near-zero gradient
                assert (np.abs(r0[f"alpha_{tag}{i}"] - a) > 0.05).mean() < 0.01


@pytest.mark.parametrize("extra,want_gb,want_scaling", [((), 4, "weak"), (("--global-batch", "16"), 16, "strong")])
def test_bench_command_at_world_two(extra, want_gb, want_scaling):
    """Rehearsal of the driver's SCALE command before hardware sees it: `bench.py --gpus 2 --steps 4 --warmup 2` as two fresh
    processes under the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), both on this card, process group
    gloo (NQ_DIST_BACKEND; RCCL refuses two ranks on one device).  Both ranks must exit 0, rank 0 alone prints ONE JSON line
    with the data-parallel headline (n_gpus 2, dp2, the global batch, a roofline object) and the single-GPU secondary
    objects suppressed."""
    import json
    import subprocess
    import sys
    port = _free_port()
    bench = os.path.join(ROOT, "bench.py")
    procs = []
    for r in (0, 1):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NQ_DIST_BACKEND="gloo", RANK=str(r), LOCAL_RANK="0",
                   WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("NQ_DP_REHEARSAL", None)
        procs.append(subprocess.Popen([sys.executable, bench, "--gpus", "2", "--steps", "4", "--warmup", "2", "--repeats", "1",
                                       "--frames", "32", "--no-cpu-baseline", *extra], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((o.decode(errors="replace"), e.decode(errors="replace")))
    assert all(p.returncode == 0 for p in procs), "\n".join(e[-3000:] for _, e in outs)
    lines0 = [ln for ln in outs[0][0].splitlines() if ln.strip().startswith("{")]
    lines1 = [ln for ln in outs[1][0].splitlines() if ln.strip().startswith("{")]
    assert len(lines0) == 1 and not lines1, (outs[0][0][-500:], outs[1][0][-500:])
    line = json.loads(lines0[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2
    assert line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == want_gb
    assert line["config"]["per_gpu_batch"] == want_gb // 2 and line["scaling"] == want_scaling
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["higher_is_better"] is True
    # weak: `value` counts B=2-equivalents of BOTH ranks; strong: global-batch iterations
    units = 2 if want_scaling == "weak" else 1
    assert abs(line["value"] - units * 1e3 / line["ms_per_step"]) <= 1e-2 * line["value"]
    rl = line["roofline"]
    assert rl and rl["frac"] > 0 and rl["kernel"].startswith("conv_") and f' B{want_gb // 2}' in rl["kernel"]
    for k in ("fp32", "phase1", "nerv", "trained", "uvg", "psnr", "cpu_baseline"):
        assert line[k] is None, k
