"""GPU parity at the sizes bench.py times (BASELINE configs[1] / configs[2]: 640x1280, ~3M-parameter decoders, B = 2).

  * single-step gradient parity: ONE phase-1 (scales) and ONE phase-2 (AdaRound) iteration of the HIP engine against the
    CPU oracle's autograd on identical weights / frames, every entry of the conv-gradient arena (14 dW^ / db^), every
    d(delta) and every d(alpha) (regulariser term included), for HNeRV-3M and NeRV-3M + Hadamard, under exact-fp32 MFMA and
    under the default bf16x3 kernels.  This is what exercises conv_wgrad3<5,6> / <4,7>, the split-K paths, head_dgrad, the
    r = 5 / 4 / 2 epilogues and the 256-long FWHT at the shapes of the headline benchmark;
  * the precision / PSNR gate on a TRAINED HNeRV-3M (tools/precision_gate.py): fp32 vs bf16x3 over phase 1 + phase 2,
    and GPU vs the CPU oracle with phase 1 running.
Tolerances are written next to each check.
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, BITS
from oracle import nq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(arch, B=2):
    import bench
    from neuroquant_amd.utils import synthetic_frames
    model = bench.build_model(workload=arch)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("encoder")}
    frames_u8 = synthetic_frames(B, 640, 1280, seed=11, device=DEV)
    model = model.to(DEV)
    with torch.no_grad():
        if arch == "hnerv":
            emb = torch.cat([model.encode(frames_u8[i:i + 2].float() / 255.0) for i in range(0, B, 2)])
        else:
            emb = model.encode((torch.arange(B, device=DEV).float() + 0.5) / B)
    return model, sd, frames_u8, emb


class _OneBatch:
    def __init__(self, frames, n):
        self.frames, self.n = frames, n

    def __len__(self):
        return 1

    def __iter__(self):
        idx = torch.arange(self.n, device=DEV)
        yield {"img": self.frames, "idx": idx, "norm_idx": idx.float() / self.n}


def _gpu_step(model, frames, emb, had, precision, phase):
    """-> (arena gradients [(dW^, db^)], parameter gradients in optimiser order) of ONE iteration of `phase`."""
    import copy
    from neuroquant_amd import ops
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    ops.set_conv_precision(precision)
    try:
        qnn = QuantModel(copy.deepcopy(model), hadamard=had,
                         weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
        qnn.set_bitwidth(BITS)
        qnn.eval()
        qnn.set_quant_state(True)
        with torch.no_grad():
            qnn(emb[:2])
        got = {}

        def probe(ph, layers, grads):
            if ph == phase and not got:
                got["arena"] = [(L.W.grad.clone(), L.b.grad.clone()) for L in layers]
                got["params"] = [g.clone() for g in grads]

        # len(gt) = 1: iters = 20 -> int(0.05*20/1) = 1 phase-1 epoch; iters = 10 -> none (straight into phase 2)
        iters = 20 if phase == "uaq" else 10
        B = frames.shape[0]
        model_reconstruction(qnn, cali_data=emb, gt=_OneBatch(frames, B), arch="hnerv" if not had else "nerv", batch_size=B,
                             iters=iters, weight=0.01, hadamard=had, b_range=(20, 2), warmup=0.0, lr=0.003, max_steps=1,
                             probe=probe)
        return got["arena"], got["params"]
    finally:
        ops.set_conv_precision(None)


def _cpu_step(sd, arch, fc_hw, frames, emb, had, phase):
    torch.set_num_threads(16)
    dec = O.Decoder.from_state_dict(sd, arch, [5, 4, 4, 2, 2], fc_hw)
    qs = O.QuantStack(dec, BITS, hadamard=had)
    got = {}

    def probe(ph, qs_, fq):
        if ph == phase and not got:
            got["arena"] = [(W.grad.clone(), b.grad.clone()) for W, b in fq]
            if ph == "uaq":
                got["params"] = [t.grad.clone() for L in qs_.dec.layers for t in (L.wd, L.bd)]
            else:
                got["params"] = [t.grad.clone() for L in qs_.dec.layers for t in (L.wa, L.ba)]
                # elements whose clamp indicator sits on the edge: x_int + zp within 1e-3 of 0 or L-1.  There the
                # indicator 1{0 <= x_int+zp <= L-1} (quantizer.py:294) flips with the last bit of sigmoid/log (libm vs
                # OCML), and the whole element gradient with it; they are excluded from the d(alpha) comparison
                edge = []
                for L in qs_.dec.layers:
                    for x, a, d, z in ((L.hw if had else L.w, L.wa, L.wd, L.wz), (L.b, L.ba, L.bd, L.bz)):
                        xi = torch.floor(x / d) + O.soft_targets(a.detach()) + z
                        qmax = 2 ** L.n_bits - 1
                        edge.append(((xi.abs() < 1e-3) | ((xi - qmax).abs() < 1e-3)))
                got["edge"] = edge

    order = np.array([[list(range(frames.shape[0]))]] * 20)
    iters = 20 if phase == "uaq" else 10
    O.calibrate(qs, emb.cpu(), frames.cpu(), order, iters, weight=0.01, b_range=(20, 2), warmup=0.0, lr=0.003, max_steps=1,
                probe=probe)
    return got


def _cmp(name, a, b, tol, mask=None, report=None):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    scale = float(b.abs().max()) + 1e-30
    err = (a - b).abs() / scale
    if mask is not None:
        err = err[~mask.reshape(-1)]
    worst = float(err.max()) if err.numel() else 0.0
    if report is not None:
        report.append((name, worst))
    assert worst <= tol, f"{name}: max |diff| / max|ref| = {worst:.3e} > {tol:.1e}"


# (arch, hadamard, batch): BASELINE configs[1] and configs[2] at the bench's per-GPU batch of 2, and configs[3]'s GLOBAL
# batch of 16 on one GPU (grid sizes, split plans and 32-bit buffer offsets at 1.9 GB tensors: the dec5 data gradient's
# input is 16 x 148 x 320 x 640 x 4 B = 1.94 GB, just below the 2 GiB where signed offsets would wrap)
# ("hnerv", True, 2): the reference also calibrates HNeRV with --hadamard (34.96 -> 37.19 dB); its C_in are not
# powers of two (92, 77, 53, 44, 37: zero-padded to 128 / 64)
@pytest.mark.parametrize("arch,had,B", [("hnerv", False, 2), ("nerv", True, 2), ("hnerv", False, 16), ("hnerv", True, 2)])
def test_full_size_single_step_gradients(arch, had, B):
    model, sd, frames_u8, emb = _setup(arch, B)
    frames = frames_u8.float() / 255.0
    fc_hw = (model.fc_h, model.fc_w)
    for phase in ("uaq", "ada"):
        ref = _cpu_step(sd, arch, fc_hw, frames, emb, had, phase)
        for precision in ("fp32", "bf16x3"):
            arena, params = _gpu_step(model, frames, emb, had, precision, phase)
            rep = []
            # Bound: 1e-4 of each tensor's largest entry.  Measured: fp32 MFMA <= ~1e-5 (summation order only); bf16x3
            # <= ~4e-5 (2^-17 per product on top), both printed below.
            tol = 1e-4
            for l, ((gW, gb), (rW, rb)) in enumerate(zip(arena, ref["arena"])):
                _cmp(f"{arch} {phase} {precision} dW^[{l}]", gW, rW, tol, report=rep)
                _cmp(f"{arch} {phase} {precision} db^[{l}]", gb, rb, tol, report=rep)
            for i, (g, r) in enumerate(zip(params, ref["params"])):
                kind = ("ddelta" if phase == "uaq" else "dalpha") + ("_w" if i % 2 == 0 else "_b") + f"[{i // 2}]"
                mask = ref["edge"][i] if phase == "ada" else None
                if mask is not None:
                    assert float(mask.double().mean()) < 5e-3, "clamp-edge elements must stay a small minority"
                # d(delta) is a per-channel SUM of (gradient x rounding residue) terms of random sign: the conv error of
                # every term survives while the terms cancel, so its bound is looser by the cancellation factor
                ptol = 1e-3 if phase == "uaq" else tol
                _cmp(f"{arch} {phase} {precision} {kind}", g, r, ptol, mask=mask, report=rep)
            worst = max(rep, key=lambda t: t[1])
            print(f"[{arch} had={had} B={B} {phase} {precision}] worst {worst[0]}: {worst[1]:.2e}")


def test_conv3_operands_between_2_and_4_gib():
    """32-bit buffer offsets of the bf16x3 kernels on operands past 2 GiB: `conv_igemm3` takes tensors below
    4 GiB with UNSIGNED offsets (the sign of the one halo quad in front of the tensor comes from the 64-bit value), the
    producer/consumer weight gradient is not offered from 2 GiB on (tests/test_cabi_cpu.py) and the 4-wave kernel with
    64-bit pointers runs instead.  No CPU oracle at this size: the exact-fp32 kernels (64-bit pointers, another tiling)
    are the reference, bound 1e-4 of the result's largest entry as everywhere else; the comparison covers every output
    pixel, i.e. also the rows whose input lies beyond the 2 GiB mark."""
    from neuroquant_amd import ops
    g = torch.Generator().manual_seed(3)
    B, cin, cout, H, W, k = 1, 20, 8, 5200, 5200, 5           # x: 20 * 5200^2 * 4 B = 2.16 GB; k = 5, 16 + 4 channels (tail chunk)
    x = torch.empty(B, cin, H, W, device=DEV)
    for c in range(cin):
        x[0, c] = torch.randn(H, W, generator=g).to(DEV)
    assert x.numel() * 4 > 2 ** 31
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(DEV)
    assert ops.conv3_supported(B, cin, H, W, cout, k)
    y3 = ops._conv_plain(x, w, precision="bf16x3")
    y32 = ops._conv_plain(x, w, precision="fp32")
    scale = float(y32.abs().max())
    assert float((y3 - y32).abs().max()) <= 1e-4 * scale
    assert float((y3[:, :, -64:] - y32[:, :, -64:]).abs().max()) <= 1e-4 * scale and float(y3[:, :, -64:].abs().max()) > 0.1 * scale
    # weight gradient with x beyond 2 GiB: 48 output channels would select the 8-wave kernel for a smaller tensor
    del y3
    dy = torch.empty(B, 48, H, W, device=DEV)
    for c in range(48):
        dy[0, c] = torch.randn(H, W, generator=g).to(DEV)
    assert ops.conv_wgrad3_supported(B, cin, H, W, 48, k)
    dw3 = ops._wgrad_plain(x, dy, k, precision="bf16x3")
    dw32 = ops._wgrad_plain(x, dy, k, precision="fp32")
    assert float((dw3 - dw32).abs().max()) <= 1e-4 * float(dw32.abs().max())


@pytest.mark.parametrize("prec", ("fp32", "bf16x3"))
def test_config1_reference_fixture(golden, prec):
    """BASELINE configs[0] against the REAL reference at full size (tests/golden/config1_hnerv3m.npz, produced by
    make_golden.py::gen_config1 running /root/reference's own model_reconstruction on the trained checkpoint fixture
    hnerv3m_bunny8_f16.npz): HNeRV-3M, 8 frames of 640x1280, bits 6 5 4 5 5 6 6, B = 2, iters_w = 50 -> 48 phase-2
    iterations in the recorded batch order.  Checked: initial scales bit-exact, average bit-width, PSNR FP / quantised
    w/o opt / w/ opt, and EVERY one of the 48 (total loss, round loss, b, count) log entries."""
    import tools_path  # noqa: F401  (adds tools/ to sys.path)
    import precision_gate as pg
    from neuroquant_amd import ops
    from neuroquant_amd.models import HNeRV
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache
    import bench
    z, ck = golden("config1_hnerv3m.npz"), golden("hnerv3m_bunny8_f16.npz")
    frames_u8 = pg.bunny_frames_640(DEV, 8)
    frames = frames_u8.float() / 255.0
    model = HNeRV(bench.HNERV_3M)
    sd = {k[3:].replace("/", "."): torch.from_numpy(v.astype(np.float32)) for k, v in ck.items() if k.startswith("sd:")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("encoder") for m in missing)
    model = model.to(DEV).eval()
    emb = torch.from_numpy(ck["emb"]).to(DEV)
    ops.set_conv_precision(prec)
    try:
        qnn = QuantModel(model, hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
        assert qnn.set_bitwidth(BITS) == float(z["avgbits"]) == 4.79399210722922     # reference log ...052303.log:233
        qnn.eval()

        def psnr():
            with torch.no_grad():
                return torch.cat([ops.frame_psnr(qnn(emb[i:i + 1])[0], frames[i:i + 1]) for i in range(8)]).cpu().numpy()

        qnn.set_quant_state(False)
        np.testing.assert_allclose(psnr(), z["psnr_fp"], atol=2e-3)           # dB per frame; conv summation order only
        qnn.set_quant_state(True)
        with torch.no_grad():
            qnn(emb[:2])
        for li, m in enumerate(qnn.quant_modules()):
            assert np.array_equal(m.weight_quantizer.delta.detach().cpu().numpy(), z[f"init_wdelta{li}"])   # bit-exact
            assert np.array_equal(m.bias_quantizer.delta.detach().cpu().numpy(), z[f"init_bdelta{li}"])
        np.testing.assert_allclose(psnr(), z["psnr_q_noopt"], atol=2e-3)
        rec = []
        loader = CacheLoader(FrameCache(frames_u8), list(range(8)), 2, order=z["order"])
        model_reconstruction(qnn, cali_data=emb, gt=loader, arch="hnerv", batch_size=2, iters=int(z["iters"]), weight=0.01,
                             hadamard=False, b_range=(20, 2), warmup=0.2, lr=0.003, recorder=rec)
        log, ref = np.array(rec), z["loss_log"]
        assert log.shape == ref.shape == (48, 4)
        np.testing.assert_array_equal(log[:, 2:], ref[:, 2:])                 # temperature, counter
        rel = np.abs(log[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
        print(f"config1 vs reference [{prec}]: loss rel diff max {rel.max():.2e}; PSNR w/ opt {psnr().mean():.4f} vs {z['psnr_q_opt'].mean():.4f}")
        # phase 2 starts from alpha with h(alpha) = frac(w/delta) (soft weights == FP weights) and moves smoothly: no
        # rounding flips inside 48 iterations, so the losses follow the reference to conv-rounding precision
        np.testing.assert_allclose(log[:, 0], ref[:, 0], rtol=5e-5)
        np.testing.assert_allclose(log[:, 1], ref[:, 1], rtol=5e-5, atol=1e-6)
        qnn.set_quant_state(True)
        np.testing.assert_allclose(psnr(), z["psnr_q_opt"], atol=5e-3)
        assert abs(float(psnr().mean()) - float(z["psnr_q_opt"].mean())) < 2e-3     # north-star bar is 0.02 dB
    finally:
        ops.set_conv_precision(None)


@pytest.mark.parametrize("prec", ("fp32", "bf16x3"))
@pytest.mark.parametrize("fixture,arch,ckpt", [("config2_nerv3m_hadamard.npz", "nerv", "nerv3m_bunny8real_f16.npz"),
                                               ("config1_hnerv3m_hadamard.npz", "hnerv", "hnerv3m_bunny8real_f16.npz")])
def test_hadamard_reference_fixture(golden, fixture, arch, ckpt, prec):
    """The Hadamard path at FULL size against the REAL reference (round 4; tests/golden/make_golden.py::gen_fullsize_hadamard ran
    /root/reference's own QuantModel(hadamard=True) + model_reconstruction): BASELINE configs[2] = NeRV Bunny_1280x640_3M with
    --hadamard (C_in padded to 256 / 256 / 128 / 64 / 32 / 32 / 32) and HNeRV-3M with --hadamard (16 / 128 / 128 / 64 / 64 / 64 /
    64), trained checkpoints, the eight real 640x1280 Bunny crops, bits 6 5 4 5 5 6 6, B = 2, iters_w = 50 -> 48 phase-2
    iterations in the recorded order.  Checked: the initial scales and zero points on the PADDED TRANSFORM-DOMAIN weights
    bit-exact (quant_layer.py:44-49), the average bit-width, PSNR FP / w/o opt / w/ opt to 2e-3 dB, every one of the 48
    (total, round, b, count) log entries to 5e-5 -- the regulariser runs over all C_pad coefficients (calib_model.py:170-191)
    -- and the final hard-rounding decisions.  (The butterflies of the fixture are the stand-in's: parity at the transform's own
    fp32 summation order stays unpinned, tests/golden/_ref_stubs.py.)"""
    import tools_path  # noqa: F401
    import precision_gate as pg
    from neuroquant_amd import ops
    from neuroquant_amd.models import HNeRV, NeRV
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache
    import bench
    z, ck = golden(fixture), golden(ckpt)
    frames_u8 = pg.bunny_real_640(DEV, 8)
    frames = frames_u8.float() / 255.0
    torch.manual_seed(1)
    model = HNeRV(bench.HNERV_3M) if arch == "hnerv" else NeRV(bench.NERV_3M)
    sd = {k[3:].replace("/", "."): torch.from_numpy(v.astype(np.float32)) for k, v in ck.items() if k.startswith("sd:")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("encoder") for m in missing)
    model = model.to(DEV).eval()
    emb = torch.from_numpy(ck["emb"].astype(np.float32)).to(DEV)
    ops.set_conv_precision(prec)
    try:
        qnn = QuantModel(model, hadamard=True, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
        assert qnn.set_bitwidth(BITS) == float(z["avgbits"])
        if arch == "nerv":
            assert float(z["avgbits"]) == 4.946213722986429              # reference log ...080342.log:143
        qnn.eval()

        def psnr():
            with torch.no_grad():
                return torch.cat([ops.frame_psnr(qnn(emb[i:i + 1])[0], frames[i:i + 1]) for i in range(8)]).cpu().numpy()

        qnn.set_quant_state(False)
        np.testing.assert_allclose(psnr(), z["psnr_fp"], atol=2e-3)
        qnn.set_quant_state(True)
        with torch.no_grad():
            qnn(emb[:2])
        mods = qnn.quant_modules()
        for li, m in enumerate(mods):
            assert m.hadamard_weight.shape[1] == int(z[f"cpad{li}"])                                       # padded length
            assert np.array_equal(m.weight_quantizer.delta.detach().cpu().numpy(), z[f"init_wdelta{li}"]), li   # bit-exact
            assert np.array_equal(m.weight_quantizer.zero_point.detach().cpu().numpy(), z[f"init_wzp{li}"]), li
            assert np.array_equal(m.bias_quantizer.delta.detach().cpu().numpy(), z[f"init_bdelta{li}"]), li
        np.testing.assert_allclose(psnr(), z["psnr_q_noopt"], atol=2e-3)
        rec = []
        loader = CacheLoader(FrameCache(frames_u8), list(range(8)), 2, order=z["order"])
        model_reconstruction(qnn, cali_data=emb, gt=loader, arch=arch, batch_size=2, iters=int(z["iters"]), weight=0.01,
                             hadamard=True, b_range=(20, 2), warmup=0.2, lr=0.003, recorder=rec)
        log, ref = np.array(rec), z["loss_log"]
        assert log.shape == ref.shape == (48, 4)
        np.testing.assert_array_equal(log[:, 2:], ref[:, 2:])                 # temperature, counter
        rel = np.abs(log[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
        qnn.set_quant_state(True)
        got = psnr()
        print(f"{fixture} vs reference [{prec}]: loss rel diff max {rel.max():.2e}; PSNR w/ opt {got.mean():.4f} vs {z['psnr_q_opt'].mean():.4f}")
        np.testing.assert_allclose(log[:, 0], ref[:, 0], rtol=5e-5)
        np.testing.assert_allclose(log[:, 1], ref[:, 1], rtol=5e-5, atol=1e-6)
        np.testing.assert_allclose(got, z["psnr_q_opt"], atol=5e-3)
        assert abs(float(got.mean()) - float(z["psnr_q_opt"].mean())) < 2e-3     # north-star bar is 0.02 dB
        # final hard decisions on all C_pad coefficients: alpha moved by <= 48 * lr from where h(alpha) = frac(w / delta), so only
        # coefficients that start within that distance of the threshold can end on the other side of it
        same = tot = 0
        for li, m in enumerate(mods):
            mask = (m.weight_quantizer.alpha.detach() >= 0).cpu().numpy().reshape(-1)
            want = np.unpackbits(z[f"mask{li}"])[: mask.size].astype(bool)
            same += int((mask == want).sum())
            tot += mask.size
            np.testing.assert_allclose(m.weight_quantizer.delta.detach().cpu().numpy(), z[f"final_wdelta{li}"], rtol=0, atol=0)
        print(f"  hard-rounding masks equal to the reference: {same / tot:.6f}")
        assert same / tot > 0.999, same / tot
    finally:
        ops.set_conv_precision(None)


@pytest.mark.parametrize("prec", ("fp32", "bf16x3"))
def test_config1_long_reference_fixture(golden, prec):
    """The REAL reference over 3000 iterations at full size, phase 1 included (tests/golden/config1_hnerv3m_long.npz,
    make_golden.py::gen_config1_long: /root/reference's own model_reconstruction, HNeRV-3M at 38.07 dB on the eight real
    crops, iters_w = 3000 -> 148 phase-1 + 2852 phase-2 iterations, 2.8 h of CPU).  The calibration is chaotic at the bit
    level (tests/golden/make_sensitivity.py; at this length exact fp32 on the GPU differs from ITSELF by 0.09 dB when the two
    frames of every batch are swapped: 37.3891 vs 37.3005 dB, profiles/r04_precision_gate_3000.json), so what can be held to
    the reference is: the schedule (b, count) of all 3000 iterations exactly; the loss of the first iteration (identical
    parameters: conv rounding only) tightly and the losses of phase 1 / of the last 500 iterations as populations; the
    regulariser column all the way; and the final PSNR within the spread the algorithm has at this length.  Measured on the
    round-4 build: reference 37.2821 dB, GPU fp32 37.3890, bf16x3 37.3318 (the reference sits 0.05 - 0.11 dB from both, as far
    as exact fp32 sits from itself under a swap of the batch order); the final hard-rounding masks agree to 60 % only --
    individual rounding decisions are NOT reproducible at this length by anyone, the objective is."""
    import tools_path  # noqa: F401
    import precision_gate as pg
    from neuroquant_amd import ops
    from neuroquant_amd.quantization import QuantModel, model_reconstruction
    from neuroquant_amd.utils import CacheLoader, FrameCache
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "config1_hnerv3m_long.npz")):
        pytest.skip("fixture not generated yet (make_golden.py --only config1_long: 2.8 h of CPU)")
    z = golden("config1_hnerv3m_long.npz")
    frames_u8 = pg.bunny_real_640(DEV, 8)
    frames = frames_u8.float() / 255.0
    model, emb, _ = pg.load_fixture_checkpoint("hnerv3m_bunny8real_f16.npz", DEV)
    order, iters = z["order"].astype(np.int64), int(z["iters"])
    assert np.array_equal(order, pg.make_order(8, 2, iters, seed=903))
    ops.set_conv_precision(prec)
    try:
        qnn = QuantModel(model, hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
        assert qnn.set_bitwidth(BITS) == float(z["avgbits"])
        qnn.eval()
        qnn.set_quant_state(True)
        with torch.no_grad():
            qnn(emb[:2])

        def psnr():
            with torch.no_grad():
                return torch.cat([ops.frame_psnr(qnn(emb[i:i + 1])[0], frames[i:i + 1]) for i in range(8)]).cpu().numpy()

        np.testing.assert_allclose(psnr(), z["psnr_q_noopt"], atol=2e-3)
        rec = []
        loader = CacheLoader(FrameCache(frames_u8), list(range(8)), 2, order=order)
        model_reconstruction(qnn, cali_data=emb, gt=loader, arch="hnerv", batch_size=2, iters=iters, weight=0.01,
                             hadamard=False, b_range=(20, 2), warmup=0.2, lr=0.003, recorder=rec)
        log, ref = np.array(rec), z["loss_log"]
        assert log.shape == ref.shape == (iters, 4)
        np.testing.assert_array_equal(log[:, 2:], ref[:, 2:])                 # temperature and counter, every iteration
        p1 = 148                                                               # phase-1 iterations: int(0.05*3000/4) epochs of 4
        assert ref[p1 - 1, 3] == p1 and ref[p1, 3] == 1                        # the counter restarts with phase 2
        rel = np.abs(log[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
        assert rel[0] < 2e-5, rel[:4]                                          # iteration 1: identical parameters
        # reconstruction losses (total - round) as populations: phase 1 and the last 500 iterations of phase 2
        rec_g, rec_r = log[:, 0] - log[:, 1], ref[:, 0] - ref[:, 1]
        # (measured on the round-4 build: phase 1 0.8 % fp32 / 2.0 % bf16x3, last 500 iterations 2.3 % / 1.9 % -- the latter IS the
        # final PSNR difference, 10 log10(1.023) = 0.10 dB, so its bound is the PSNR bound of 0.2 dB = 4.7 %)
        pop = []
        for (lo, hi), bound in (((0, p1), 0.03), ((iters - 500, iters), 0.047)):
            pop.append(abs(rec_g[lo:hi].mean() - rec_r[lo:hi].mean()) / rec_r[lo:hi].mean())
            assert pop[-1] <= bound, (lo, hi, pop)
        # the regulariser is a sum over 2.6 M rounding variables: smooth, follows the reference closely all the way
        # (measured: 0.11 % of its maximum, both precisions)
        on = ref[:, 1] > 0
        reg = np.abs(log[on, 1] - ref[on, 1]).max() / ref[on, 1].max()
        assert reg <= 0.005, reg
        print(f"config1_long [{prec}]: reconstruction-loss means, phase 1 / last 500: {pop[0]:.2e} / {pop[1]:.2e} relative; "
              f"regulariser max deviation {reg:.2e} of its maximum; phase-1 losses max rel {rel[:p1].max():.2e}")
        qnn.set_quant_state(True)
        got, want = float(psnr().mean()), float(z["psnr_q_opt"].mean())
        print(f"config1_long vs reference [{prec}]: first-iteration rel diff {rel[0]:.1e}; final PSNR {got:.4f} vs {want:.4f} dB; "
              f"masks equal {_mask_agreement(qnn, z):.4f}")
        assert abs(got - want) <= 0.2, (got, want)                             # 2 x the algorithm's own spread at 3000 iterations
        assert got >= float(z["psnr_q_noopt"].mean()) + 1.0                    # and the calibration did its work
    finally:
        ops.set_conv_precision(None)


def _mask_agreement(qnn, z):
    same = tot = 0
    for li, m in enumerate(qnn.quant_modules()):
        mask = (m.weight_quantizer.alpha.detach() >= 0).cpu().numpy().reshape(-1)
        want = np.unpackbits(z[f"mask{li}"])[: mask.size].astype(bool)
        same += int((mask == want).sum())
        tot += mask.size
    return same / tot


def test_precision_gate_trained_hnerv_3m():
    """tools/precision_gate.py at reduced length: HNeRV-3M fitted to >= 30 dB on the 8 Bunny-derived frames, then
    (a) 2000-iteration calibrations (100 phase-1 + 1900 phase-2 iterations) for two recorded batch orders under exact
        fp32, exact fp32 with swapped batch halves (same maths, other summation order) and bf16x3 (both ways too): mean
        final PSNR of bf16x3 within max(0.02 dB, S) of fp32's, single runs within max(0.02 dB, 2 S), S = what exact fp32
        differs from itself;
    (b) GPU (both precisions) vs the CPU oracle over a calibration whose phase 1 runs (NQ_GATE_ORACLE_ITERS, default 120
        -> 4 phase-1 + 116 phase-2 iterations; the tool's 200-iteration record is profiles/r02_precision_gate.json):
        final PSNR within the short-schedule spread floor (0.08 dB), first iteration of the loss within 1e-5 (later iterations: sanity bound)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import precision_gate as pg
    o_it = int(os.environ.get("NQ_GATE_ORACLE_ITERS", "120"))
    args = types.SimpleNamespace(train_steps=int(os.environ.get("NQ_GATE_TRAIN_STEPS", "3000")), iters=2000,
                                 oracle_iters=o_it, frames="bunny", frames_n=8, cpu_threads=16, record=False, ckpt=None,
                                 save_ckpt=None, seeds=[903])   # (one batch order x four arms: the full-length pair below
                                                                # carries the precision claim now)
    res = pg.run(args)
    print({k: v for k, v in res.items() if k != "config"})
    assert res["fp_psnr"] >= 30.0, res["fp_psnr"]
    assert res["fp32"]["q_opt"] > res["fp32"]["q_noopt"] + 0.1          # the calibration does move the model
    # bf16x3 vs exact fp32, judged against what exact fp32 does to itself under a re-ordered summation (the same maths with
    # the two frames of every batch swapped, another batch order or a re-ordered bias-gradient sum moves the final PSNR of
    # this 2000-iteration schedule by up to 0.09 dB, profiles/r02_precision_gate_2000.json: the calibration is chaotic,
    # tests/golden/make_sensitivity.py; the full 21 000-iteration schedule collapses to < 0.01 dB and is held to 0.02 dB,
    # profiles/r02_precision_gate_21000.json): population means within S, single runs within 2 S, S = max(measured
    # self-spread, 0.08 dB) -- precision_gate.gate_ok
    # Round 4: a SANITY bound only.  The precision claim (0.02 dB) is carried by the full-length pair below
    # (test_full_length_calibration_fp32_vs_bf16x3), where the schedule has settled; a 2000-iteration run has not, and
    # exact fp32 differs from itself by up to 0.09 dB here.
    assert res["dpsnr_fp32_vs_bf16x3_dB"] <= 0.2 and res["dmean_dB"] <= 0.2, {
        k: res[k] for k in ("q_opt_fp32_runs", "q_opt_bf16x3_runs", "fp32_self_spread_dB", "dmean_dB", "dpsnr_fp32_vs_bf16x3_dB")}
    o = res["oracle"]
    assert o["phase1_iterations"] >= 1 and o["iterations"] == o_it // 4 * 4
    # 120 iterations of a chaotic recursion on two machines with different summation orders (CPU oracle vs GPU): the final
    # PSNRs differed by 0.0002 ... 0.033 dB over the boxes / checkpoints met (the FP32 fit itself varies with the box:
    # the ConvNeXt encoder runs on MIOpen's per-box kernel choice) -- bounded by the short-schedule spread floor
    assert o["dpsnr_fp32_dB"] < pg.SPREAD_FLOOR_DB["short"] and o["dpsnr_bf16x3_dB"] < pg.SPREAD_FLOOR_DB["short"], o
    # iteration 0 (identical parameters on both sides): conv rounding only.  From iteration 1 on phase 1 has moved every
    # scale by lr = 1e-3 and the trajectories drift apart as in tests/golden/traj_sensitivity.json (<= ~1e-2)
    assert o["loss_rel_diff_fp32"]["first"] < 1e-5 and o["loss_rel_diff_bf16x3"]["first"] < 1e-5, o
    # Later iterations are a sanity bound only: Adam's first steps move every scale by +-lr whatever the size of its
    # gradient, so a near-zero gradient component whose SIGN differs between two summation orders splits the trajectories at
    # once (observed: batch losses 1e-3 ... 1e-1 apart at single iterations, final PSNR 0.0002 ... 0.033 dB apart).  The
    # per-iteration parity evidence is the reference fixture above (48 logged iterations to 5e-5) and the single-step
    # gradient test, both deterministic.
    assert o["loss_rel_diff_fp32"]["max"] < 0.3 and o["loss_rel_diff_bf16x3"]["max"] < 0.3, o


def test_full_length_calibration_fp32_vs_bf16x3(caplog):
    """The long-run precision claim under the driver's eyes: ONE full-length calibration pair (iters_w = 21000 -> 1048
    phase-1 + 19 952 phase-2 iterations on 8 frames; reference schedule calib_model.py:144, 205) with exact-fp32 and with
    bf16x3 convolutions, batch-order seed 903, on the committed operating-point fixtures (hnerv3m_bunny8real_f16.npz: FP
    38.07 dB on the eight real Bunny crops, bunny8_640x1280.npz).  North-star bar: |PSNR_bf16x3 - PSNR_fp32| <= 0.02 dB; both
    must recover >= 1.5 dB over the un-optimised quantised model; the temperature the run logs at counts 4500 / 19500 is the
    reference log's (b = 19.68 / 3.61, results/...052303.log:273, :303)."""
    import logging
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import precision_gate as pg
    n, B, iters = 8, 2, 21000
    frames_u8 = pg.bunny_real_640(DEV, n)
    model, emb, _ = pg.load_fixture_checkpoint("hnerv3m_bunny8real_f16.npz", DEV)
    order = pg.make_order(n, B, iters, seed=903)
    assert order.shape == (5250, 4, 2)
    res = {}
    with caplog.at_level(logging.INFO):
        for prec in ("fp32", "bf16x3"):
            caplog.clear()
            r, _, _ = pg.calibrate_gpu(model, frames_u8, emb, order, iters, prec, record=False)
            res[prec] = r
            # the engine logs every 500th count of each phase like the reference (calib_model.py:86-88); phase 2 is the later one
            b_at = {}
            for rec in caplog.records:
                m = re.search(r"b=([0-9.]+)\s+count=(\d+)", rec.getMessage())
                if m:
                    b_at[int(m.group(2))] = m.group(1)       # phase 2 overwrites phase 1's entries of the same count
            assert b_at.get(4500) == "19.68" and b_at.get(19500) == "3.61", {k: b_at.get(k) for k in (500, 4500, 19500)}
            print(f"[{prec}] 21k: {r['seconds']:.1f} s, PSNR w/o opt {r['q_noopt']:.4f} -> w/ opt {r['q_opt']:.4f} dB")
    f32, b3 = res["fp32"], res["bf16x3"]
    assert f32["q_noopt"] == b3["q_noopt"]                       # the same starting point (evaluated with exact fp32)
    assert abs(b3["q_opt"] - f32["q_opt"]) <= 0.02, (f32["q_opt"], b3["q_opt"])
    assert f32["q_opt"] >= f32["q_noopt"] + 1.5 and b3["q_opt"] >= b3["q_noopt"] + 1.5, (f32, b3)
    # the default kernels evaluate the model they calibrated to the same number
    assert abs(b3["q_opt_eval_bf16x3"] - b3["q_opt"]) <= 2e-3
