"""Data-parallel path on the GPU with a 1-rank RCCL group (NQ_DP_REHEARSAL=1): the gradient-arena hook + in-place
all_reduce(ReduceOp.AVG) of model_reconstruction and the driver's process-group / sharded-loader wiring
(methods/calibrate_network.py:dist_setup) run inside the GPU suite, and must be bit-identical to the single-process run
(AVG over one rank is the identity).  The N > 1 arithmetic is covered on CPU by tests/test_dp_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import T, BITS, TINY_HNERV, state_dict_from_npz

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
    assert ops._GRAD_ARENA_HOOK is None                       # removed again at the end of model_reconstruction
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
