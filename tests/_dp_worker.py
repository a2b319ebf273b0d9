"""One rank of a REAL two-process data-parallel calibration on one GPU (tests/test_dp_gpu.py::test_two_processes_gloo_on_gpu):
both ranks use cuda:0, the process group is gloo (RCCL does not run two ranks on one device), so the product path
(model_reconstruction -> ops.grad_arena_hook -> asynchronous in-place mean all-reduce of the two arena parts) runs with a
genuine collective between two processes holding different frames.  argv: rank world port out.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main(rank, world, port, out):
    import torch.distributed as dist
    from conftest import T, BITS, TINY_HNERV, state_dict_from_npz
    from neuroquant_amd.models import HNeRV
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache
    os.environ["NQ_GRAPH"] = "0"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = "cuda"
    gdir = os.path.join(HERE, "golden")
    z = np.load(os.path.join(gdir, "traj_hnerv.npz"))
    model = HNeRV(TINY_HNERV)
    model.load_state_dict(state_dict_from_npz(z, "sd:"))
    model = model.to(dev).eval()
    emb = T(z["emb"]).to(dev)
    qnn = QuantModel(model, hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    qnn.set_bitwidth(BITS)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:2])
    frames_u8 = T(np.load(os.path.join(gdir, "frames_320x640.npz"))["frames"]).to(dev)
    g = torch.Generator().manual_seed(5)
    order = torch.stack([torch.randperm(8, generator=g).view(2, 4) for _ in range(80)]).numpy()
    loader = CacheLoader(FrameCache(frames_u8), list(range(8)), 4, rank=rank, world=world, order=order)
    rec = []
    model_reconstruction(qnn, cali_data=emb, gt=loader, recorder=rec, max_steps=5, arch="hnerv", batch_size=4, iters=40,
                         weight=0.01, hadamard=False, b_range=(20, 2), warmup=0.0, lr=0.003)
    torch.cuda.synchronize()
    res = {"log": np.array(rec)}
    for i, m in enumerate(qnn.quant_modules()):
        for tag, q in (("w", m.weight_quantizer), ("b", m.bias_quantizer)):
            res[f"delta_{tag}{i}"] = q.delta.detach().cpu().numpy()
            res[f"alpha_{tag}{i}"] = q.alpha.detach().cpu().numpy()
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
