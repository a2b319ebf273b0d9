"""Data-parallel path on CPU: world_size 2, gloo.  Covers the pieces the 8-GPU run relies on:
identical global batch order on every rank, disjoint rank shards, ONE all-reduce(sum)/world over the flattened
conv-gradient list keeping replicas bit-identical, and 'mean of local means == global mean' for equal shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeCache:
    """FrameCache stand-in without a GPU: `frames` carries the device, batch(idx) returns deterministic frames."""

    def __init__(self, n):
        self.frames = torch.zeros(n, 1)
        self.n = n

    def __len__(self):
        return self.n

    def batch(self, idx):
        return idx.float().view(-1, 1, 1, 1).expand(-1, 3, 2, 2).clone()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neuroquant_amd.utils import CacheLoader, allreduce_mean_
        cache = _FakeCache(16)
        loader = CacheLoader(cache, list(range(16)), batch_size=4, seed=903, rank=rank, world=world)
        assert len(loader) == 4
        seen = []
        for ep in range(2):
            for s in loader:
                assert s["img"].shape == (2, 3, 2, 2) and s["idx"].shape == (2,)
                assert torch.equal(s["img"][:, 0, 0, 0], s["idx"].float())
                seen.append(s["idx"].clone())
        seen = torch.stack(seen)                                   # (8 iterations, 2 local frames)
        gathered = [torch.zeros_like(seen) for _ in range(world)]
        dist.all_gather(gathered, seen)
        both = torch.cat(gathered, dim=1)                          # global batch per iteration
        for ep in range(2):
            frames = both[ep * 4:(ep + 1) * 4].reshape(-1)
            assert sorted(frames.tolist()) == list(range(16))      # drop_last over a full permutation, no overlap
        # gradient exchange: mean over ranks, identical everywhere, shapes preserved
        g = torch.Generator().manual_seed(rank)
        grads = [torch.randn(3, 2, 1, 1, generator=g), torch.randn(3, generator=g)]
        local = [t.clone() for t in grads]
        allreduce_mean_(grads)
        g0, g1 = torch.Generator().manual_seed(0), torch.Generator().manual_seed(1)
        want = [(torch.randn(3, 2, 1, 1, generator=g0) + torch.randn(3, 2, 1, 1, generator=g1)) / 2,
                (torch.randn(3, generator=g0) + torch.randn(3, generator=g1)) / 2]
        for a, b, l in zip(grads, want, local):
            assert a.shape == l.shape and torch.allclose(a, b, atol=1e-7)
        # mean of local means == global mean: DP gradient of lp_loss on a toy linear "decoder"
        w = torch.ones(4, requires_grad=True)
        xg = torch.arange(32, dtype=torch.float32).view(8, 4) / 10
        loss_global = ((xg * w).sum(1) - 1).pow(2).mean()
        (gw_global,) = torch.autograd.grad(loss_global, w)
        xl = xg[rank * 4:(rank + 1) * 4]
        (gw_local,) = torch.autograd.grad(((xl * w).sum(1) - 1).pow(2).mean(), w)
        gl = [gw_local.clone()]
        allreduce_mean_(gl)
        assert torch.allclose(gl[0], gw_global, rtol=1e-6)
        # the engine's arena exchange (calib_model._mean_all_reduce): asynchronous in-place mean of two views of one arena
        # (ReduceOp.AVG where the backend takes it, else SUM and a scale after the wait: decided once per backend)
        from neuroquant_amd.quantization import calib_model as cm
        arena = torch.arange(10, dtype=torch.float32) * (rank + 1)
        fins = [cm._mean_all_reduce(arena[:6]), cm._mean_all_reduce(arena[6:])]
        for fin in fins:
            fin()
        assert torch.equal(arena, torch.arange(10, dtype=torch.float32) * 1.5), arena
        assert "gloo" in cm._AVG_OK      # decided (this gloo may or may not reduce with AVG; the mean is the same)
        # the SUM-and-scale branch (a backend without ReduceOp.AVG), forced: same mean, in place, asynchronous
        cm._AVG_OK["gloo"] = False
        arena = torch.arange(10, dtype=torch.float32) * (rank + 1)
        fins = [cm._mean_all_reduce(arena[:6]), cm._mean_all_reduce(arena[6:])]
        for fin in fins:
            fin()
        assert torch.equal(arena, torch.arange(10, dtype=torch.float32) * 1.5), arena
        # a backend that REFUSES AVG at the call: remembered, and the same call falls through to SUM-and-scale ...
        real_all_reduce, calls = dist.all_reduce, []

        def refusing(t, op=dist.ReduceOp.SUM, **kw):
            calls.append(op)
            if op == dist.ReduceOp.AVG:
                raise RuntimeError("Cannot use ReduceOp.AVG with Gloo")
            return real_all_reduce(t, op=op, **kw)

        cm._AVG_OK.pop("gloo")
        dist.all_reduce = refusing
        try:
            arena = torch.full((4,), float(rank + 1))
            cm._mean_all_reduce(arena)()
            assert torch.equal(arena, torch.full((4,), 1.5)) and cm._AVG_OK["gloo"] is False
            assert calls == [dist.ReduceOp.AVG, dist.ReduceOp.SUM]
            cm._mean_all_reduce(arena)()
            assert calls[2:] == [dist.ReduceOp.SUM]          # decided once per backend
            # ... but any OTHER error of the collective (a communicator fault) is not swallowed
            cm._AVG_OK.pop("gloo")

            def broken(t, op=dist.ReduceOp.SUM, **kw):
                raise RuntimeError("NCCL communicator was aborted on rank 0")

            dist.all_reduce = broken
            try:
                cm._mean_all_reduce(arena)
                raise AssertionError("a communicator error must propagate")
            except RuntimeError as e:
                assert "aborted" in str(e) and "gloo" not in cm._AVG_OK
        finally:
            dist.all_reduce = real_all_reduce
        # bit-allocation sweep: candidates are dealt round-robin to the ranks, scores gathered everywhere
        from neuroquant_amd.methods import bit_assign as ba
        cands = {f"candidate{i + 1}": [2 + (i + j) % 6 for j in range(7)] for i in range(5)}
        mine = ba.owned_candidates(list(cands), rank, world)
        assert mine == [n for i, n in enumerate(cands) if i % world == rank]
        scores = ba.gather_scores({n: float(sum(cands[n])) + 0.25 * rank for n in mine}, world)
        assert set(scores) == set(cands)
        best, bits, sc = ba.pick_best(cands, scores)
        assert sc == min(scores.values()) and bits == cands[best]
        picks = [None] * world
        dist.all_gather_object(picks, best)
        assert len(set(picks)) == 1                                 # every rank reports the same winner
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_sharding_and_allreduce_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_allreduce_is_noop_without_process_group():
    from neuroquant_amd.utils import allreduce_mean_
    t = [torch.ones(3)]
    assert allreduce_mean_(t)[0].equal(torch.ones(3))


def test_full_loader_covers_every_frame_once_without_dropping():
    """bit_assign's FullLoader = the reference's full_dataloader (shuffle=True, drop_last=False)."""
    from neuroquant_amd.methods.bit_assign import FullLoader, owned_candidates, pick_best
    ld = FullLoader(_FakeCache(11), batch_size=4, seed=903)
    assert len(ld) == 3
    idx = torch.cat([b["idx"] for b in ld])
    assert sorted(idx.tolist()) == list(range(11))
    sizes = [b["idx"].numel() for b in FullLoader(_FakeCache(11), 4)]
    assert sizes == [4, 4, 3]
    assert owned_candidates(["a", "b", "c"], 0, 1) == ["a", "b", "c"]
    assert pick_best({"a": [2], "b": [3]}, {"a": 1.0, "b": 1.0})[0] == "a"      # ties go to the earlier candidate
