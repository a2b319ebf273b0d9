"""GPU tests of the rows either side of the hot loop at the bar of the loop itself:
  f-2  frame ingestion + evaluation: `load_frames` (PNG -> center crop, videosets/datasets.py:8-30), `CacheLoader`,
       `evaluate` (calibrate_network.py:82-145) against the oracle's decode + PSNR;
  f-1 / BASELINE configs[4]: the UVG 960x1920 ~12M-parameter HNeRV (tools/hnerv_uvg_12m.yaml, PixelShuffle 3): two
       calibration iterations and a one-batch Omega score of both toy candidates against the oracle AT THAT SIZE;
  f-4  the FP32 trainer's learning-rate schedule (`adjust_lr`, reference utils.py:77-97).
"""
import copy
import math
import os
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, T, BITS, TINY_HNERV, state_dict_from_npz
from oracle import nq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(a, b, rtol=0.0, atol=0.0):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------ f-2
def test_load_frames_cacheloader_evaluate(golden, tmp_path):
    from PIL import Image
    from neuroquant_amd.methods import calibrate_network as cn
    from neuroquant_amd.models import HNeRV
    from neuroquant_amd.quantization import QuantModel
    from neuroquant_amd.utils import CacheLoader, FrameCache
    small = golden("frames_320x640.npz")["frames"]                       # (8,3,320,640) uint8
    # PNGs larger than the crop (odd margins): 361 x 685, the fixture frame in the centre-crop window
    top, left = int(round((361 - 320) / 2.0)), int(round((685 - 640) / 2.0))   # torchvision center_crop rounding
    rng = np.random.default_rng(0)
    for i, f in enumerate(small):
        big = rng.integers(0, 256, size=(361, 685, 3), dtype=np.uint8)
        big[top:top + 320, left:left + 640] = f.transpose(1, 2, 0)
        Image.fromarray(big).save(tmp_path / f"{i:04d}.png")
    args = types.SimpleNamespace(synthetic=0, data_path=str(tmp_path), seed=903, arch="hnerv", print_freq=50, val_ind_list=[])
    frames = cn.load_frames(args, dict(TINY_HNERV), DEV)
    assert frames.dtype == torch.uint8 and frames.is_cuda and tuple(frames.shape) == (8, 3, 320, 640)
    assert torch.equal(frames.cpu(), torch.from_numpy(small))                # center crop, sorted order: bit-exact

    cache = FrameCache(frames)
    loader = CacheLoader(cache, list(range(8)), 2, seed=5)
    assert len(loader) == 4
    for ep in range(2):
        seen = []
        for s in loader:
            assert s["img"].shape == (2, 3, 320, 640) and s["img"].dtype == torch.float32 and s["idx"].dtype == torch.int64
            assert torch.equal(s["img"].cpu(), frames[s["idx"]].cpu().float() / 255.0)   # img / 255 (datasets.py:23), bit-exact (CPU true division)
            _close(s["norm_idx"], s["idx"].float() / 8)                          # float(idx) / len(video) (datasets.py:50)
            seen += s["idx"].tolist()
        assert sorted(seen) == list(range(8))                                    # shuffle=True, drop_last=True, one pass

    # evaluate(): per-frame encode -> decode -> PSNR, FP and quantised, vs the oracle on the same checkpoint
    z = golden("traj_hnerv.npz")
    sd = state_dict_from_npz(z, "sd:")
    model = HNeRV(TINY_HNERV)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    res, embeds = cn.evaluate(model, cache, args, dict(TINY_HNERV))
    assert len(embeds) == 8 and tuple(embeds[0].shape) == (1, 8, 1, 2)
    _close(torch.cat(embeds), z["emb"], rtol=1e-3, atol=1e-4)
    fr = torch.from_numpy(small).float() / 255.0
    dec = O.Decoder.from_state_dict(sd, "hnerv", [5, 4, 4, 2, 2])
    with torch.no_grad():
        ref_fp = float(O.psnr_per_frame(dec.forward(T(z["emb"])), fr).mean())
    assert abs(float(res[0]) - ref_fp) < 2e-3 and float(res[1]) == 0.0           # all frames "seen" (data_split 1_1_1)
    _close(float(res[0]), float(z["psnr_fp"].mean()), atol=2e-3)                # and the reference's own number
    qnn = QuantModel(copy.deepcopy(model), hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    qnn.set_bitwidth(BITS)
    qnn.eval()
    qnn.set_quant_state(True)
    res_q, _ = cn.evaluate(qnn, cache, args, dict(TINY_HNERV))
    _close(float(res_q[0]), float(z["psnr_q_noopt"].mean()), atol=2e-3)          # reference: quantised, before calibration


# ------------------------------------------------------------------------------------------ configs[4]
def test_uvg_12m_calibration_and_omega_vs_oracle():
    """BASELINE configs[4] shape on ONE GPU: HNeRV UVG 960x1920, 11.8 M decoder parameters, strides 5,4,4,3,2 (PixelShuffle 3
    between dec3 and dec4).  Two phase-2 calibration iterations (losses vs the oracle) and the Omega score v'Hv of both toy
    candidates on one batch (double backward through all 7 layers; vs the oracle's CPU double backward)."""
    from neuroquant_amd.methods import bit_assign
    from neuroquant_amd.models import HNeRV
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache, get_config, synthetic_frames
    cfg = get_config(os.path.join(ROOT, "tools", "hnerv_uvg_12m.yaml"))
    torch.manual_seed(7)
    model = HNeRV(cfg)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not name.startswith("encoder"):
                p.copy_(torch.randn(p.shape, generator=g) * ((1.5 / p[0].numel()) ** 0.5 if p.dim() > 1 else 0.02))
    assert 11.5e6 < sum(p.numel() for n, p in model.named_parameters() if not n.startswith("encoder")) < 12.5e6
    sd = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("encoder")}
    model = model.to(DEV).eval()
    frames_u8 = synthetic_frames(2, 960, 1920, seed=5, device=DEV)
    frames = frames_u8.float() / 255.0
    with torch.no_grad():
        emb = model.encode(frames)
    torch.set_num_threads(16)

    # ---- two calibration iterations ----
    qnn = QuantModel(copy.deepcopy(model), hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    qnn.set_bitwidth(BITS)
    qnn.eval()
    qnn.set_quant_state(True)
    with torch.no_grad():
        qnn(emb[:2])
    order = np.array([[[0, 1]]] * 10)
    rec = []
    model_reconstruction(qnn, cali_data=emb, gt=CacheLoader(FrameCache(frames_u8), [0, 1], 2, order=order), arch="hnerv",
                         batch_size=2, iters=10, weight=0.01, hadamard=False, b_range=(20, 2), warmup=0.0, lr=0.003,
                         recorder=rec, max_steps=2)
    dec = O.Decoder.from_state_dict(sd, "hnerv", cfg["dec_strides"])
    qs = O.QuantStack(dec, BITS, hadamard=False)
    ref = np.array(O.calibrate(qs, emb.cpu(), frames.cpu(), order, 10, weight=0.01, b_range=(20, 2), warmup=0.0, lr=0.003,
                               max_steps=2))
    log = np.array(rec)
    np.testing.assert_array_equal(log[:, 2:], ref[:, 2:])
    np.testing.assert_allclose(log[:, :2], ref[:, :2], rtol=1e-4)

    # ---- Omega of the two toy candidates, one batch ----
    batch = [dict(img=frames, idx=torch.arange(2, device=DEV), norm_idx=torch.arange(2, device=DEV).float() / 2)]
    scores, refs = [], []
    for bits in bit_assign.hnerv_candidate.values():
        qn = QuantModel(copy.deepcopy(model), hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
        qn.eval()
        qn.set_bitwidth(bits)
        qn.set_quant_state(True)
        with torch.no_grad():
            qn(emb[:2])
        scores.append(float(bit_assign.sensitivity_criterion("omega", "hnerv", copy.deepcopy(model), qn, batch)))
        d2 = O.Decoder.from_state_dict(sd, "hnerv", cfg["dec_strides"])
        q2 = O.QuantStack(d2, bits, hadamard=False)
        vec = O.weight_perturbation(q2)
        for v, vg in zip(vec, qn.get_perturbation()):
            assert torch.equal(v, vg.cpu())                                      # perturbations bit-exact
        total, _, _ = O.sensitivity(d2, vec, [(emb.cpu(), frames.cpu())], "omega", max_batches=1)
        refs.append(float(total))
    print("UVG-12M omega: GPU", scores, "oracle", refs)
    for s, r in zip(scores, refs):
        assert abs(s - r) <= 5e-3 * abs(r), (s, r)
    assert (scores[0] < scores[1]) == (refs[0] < refs[1])


# ------------------------------------------------------------------------------------------ f-4
def test_adjust_lr_matches_reference_formula():
    """methods/regress.py::adjust_lr vs the closed form of the reference's utils.py:77-97, both schedule types."""
    from neuroquant_amd.methods.regress import adjust_lr

    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0}]

    for lr_type in ("cosine_0.1_1_0.1", "hybrid_0.2_2_1.5_0.05_0.01"):
        args = types.SimpleNamespace(lr_type=lr_type, lr=5e-4)
        for e in (0.0, 0.03, 0.1, 0.25, 0.5, 0.9, 0.999):
            got = adjust_lr(Opt, e, args)
            parts = [float(x) for x in lr_type.split("_")[1:]]
            if lr_type.startswith("cosine"):
                up, pw, mn = parts
                want = mn + (1 - mn) * (e / up) ** pw if e < up else max(0.5 * (math.cos(math.pi * (e - up) / (1 - up)) + 1.0), 0.05)
            else:
                up, pw, dpw, mn, fin = parts
                want = mn + (1 - mn) * (e / up) ** pw if e < up else 1 - (1 - fin) * ((e - up) / (1 - up)) ** dpw
            assert got == args.lr * want and all(g["lr"] == got for g in Opt.param_groups)
