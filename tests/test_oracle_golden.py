"""Pins the oracle (oracle/nq_oracle.py) against vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import T, GOLDEN, BITS, state_dict_from_npz
from oracle import nq_oracle as O

torch.set_num_threads(8)


def eq(a, b, rtol=0.0, atol=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_zero_channel_quirk_is_the_references(golden):
    """An all-zero output channel: 'max' init gives delta = 1e-8, AdaRound's fp16 round trip makes it 0, alpha and every
    forward of that channel become NaN (reference quantizer.py:163-165, 264-265, 305-314; recorded by make_golden.py from
    the reference itself).  The oracle restates exactly that; the HIP path deviates on purpose (the channel
    stays exactly 0), see tests/test_hip_parity.py::test_zero_channel_is_guarded."""
    z = golden("zero_channel.npz")
    x = T(z["x"])
    d, zp = O.scale_init_max(x, 16, True)
    eq(d, z["uaq_delta"]); eq(zp, z["uaq_zp"])
    assert float(d[1]) == np.float32(1e-8)
    eq(O.uaq_fake_quant(x, d, zp, 16), z["y_uaq"])
    d2, zp2, a0 = O.adaround_init(x, d, zp)
    eq(d2, z["delta"]); eq(zp2, z["zp"])
    assert float(d2[1]) == 0.0
    assert np.array_equal(np.isnan(a0.numpy()), np.isnan(z["alpha0"])) and np.isnan(a0[1].numpy()).all()
    ys, _ = O.adaround_fake_quant(x, a0, d2, zp2, 16, soft=True)
    assert np.array_equal(np.isnan(ys.numpy()), np.isnan(z["ysoft"])) and np.isnan(z["ysoft"][1]).all()
    ok = ~np.isnan(z["ysoft"])
    eq(ys.numpy()[ok], z["ysoft"][ok], rtol=1e-6, atol=1e-7)
    yh, _ = O.adaround_fake_quant(x, a0, d2, zp2, 16, soft=False)
    assert np.array_equal(np.isnan(yh.numpy()), np.isnan(z["yhard"]))


# ------------------------------------------------------------------ UAQ
@pytest.mark.parametrize("nb", range(2, 9))
def test_uaq_weight(golden, nb):
    z = golden("uaq.npz")
    x, go = T(z[f"w{nb}_x"]), T(z[f"w{nb}_go"])
    d, zp = O.scale_init_max(x, 2 ** nb, True)
    eq(d, z[f"w{nb}_delta"])          # bit-exact: same double division, same fp32 cast
    eq(zp, z[f"w{nb}_zp"])
    d = d.clone().requires_grad_(True)
    y = O.uaq_fake_quant(x, d, zp, 2 ** nb)
    eq(y.detach(), z[f"w{nb}_y"])
    (y * go).sum().backward()
    eq(d.grad, z[f"w{nb}_ddelta"], rtol=1e-6, atol=1e-6)
    eq(O.uaq_ddelta(x, go, d.detach(), zp, 2 ** nb), z[f"w{nb}_ddelta"], rtol=2e-5, atol=2e-4)
    # perturbed delta
    d2 = T(z[f"w{nb}_delta2"]).requires_grad_(True)
    y2 = O.uaq_fake_quant(x, d2, zp, 2 ** nb)
    eq(y2.detach(), z[f"w{nb}_y2"])
    (y2 * go).sum().backward()
    eq(d2.grad, z[f"w{nb}_ddelta2"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("nb", range(2, 9))
def test_uaq_bias(golden, nb):
    z = golden("uaq.npz")
    x, go = T(z[f"b{nb}_x"]), T(z[f"b{nb}_go"])
    d, zp = O.scale_init_max(x, 2 ** nb, True)
    assert d.shape == (1,) and zp.shape == (1,)
    eq(d, z[f"b{nb}_delta"])
    eq(zp, z[f"b{nb}_zp"])
    d = d.clone().requires_grad_(True)
    y = O.uaq_fake_quant(x, d, zp, 2 ** nb)
    eq(y.detach(), z[f"b{nb}_y"])
    (y * go).sum().backward()
    eq(d.grad, z[f"b{nb}_ddelta"], rtol=1e-6, atol=1e-6)


def test_uaq_layerwise(golden):
    z = golden("uaq.npz")
    x = T(z["lw_x"])
    d, zp = O.scale_init_max(x, 32, False)
    eq(d, z["lw_delta"])
    eq(zp, z["lw_zp"])
    eq(O.uaq_fake_quant(x, d, zp, 32), z["lw_y"])
    eq(O.uaq_ddelta(x, T(z["lw_go"]), d, zp, 32), z["lw_ddelta"], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ AdaRound
@pytest.mark.parametrize("nb", (2, 3, 4, 6, 8))
def test_adaround(golden, nb):
    z = golden("adaround.npz")
    nl = 2 ** nb
    for kind in ("w", "b"):
        x = T(z[f"{kind}{nb}_x"])
        d, zp, a0 = O.adaround_init(x, T(z[f"{kind}{nb}_uaq_delta"]), T(z[f"{kind}{nb}_uaq_zp"]))
        eq(d, z[f"{kind}{nb}_delta"])
        eq(zp, z[f"{kind}{nb}_zp"])
        eq(a0, z[f"{kind}{nb}_alpha0"])
        go = T(z[f"{kind}{nb}_go"])
        alpha = T(z[f"{kind}{nb}_alpha"]) if kind == "w" else a0.clone()
        alpha.requires_grad_(True)
        y, xq = O.adaround_fake_quant(x, alpha, d, zp, nl, True)
        eq(y.detach(), z[f"{kind}{nb}_ysoft"])
        (y * go).sum().backward()
        eq(alpha.grad, z[f"{kind}{nb}_dalpha"], rtol=1e-6, atol=1e-9)
        eq(O.adaround_dalpha(x, go, alpha.detach(), d, zp, nl), z[f"{kind}{nb}_dalpha"], rtol=1e-5, atol=1e-8)
        if kind == "w":
            eq(xq.detach(), z[f"w{nb}_xq_soft"])
            yh, xqh = O.adaround_fake_quant(x, alpha.detach(), d, zp, nl, False)
            eq(yh, z[f"w{nb}_yhard"])
            eq(xqh, z[f"w{nb}_xq_hard"])


def test_round_regulariser(golden):
    z = golden("roundloss.npz")
    for b in (20, 7.3, 2):
        tag = str(b).replace(".", "p")
        a = T(z["alpha"]).requires_grad_(True)
        loss = O.round_regulariser([a], b, 0.01)
        eq(loss.detach(), z[f"loss_b{tag}"])
        loss.backward()
        eq(a.grad, z[f"dalpha_b{tag}"])


def test_temp_decay():
    with open(os.path.join(GOLDEN, "tempdecay.json")) as f:
        tab = json.load(f)
    for key, rows in tab.items():
        t_max, rel = key.split("_")
        for t, want in rows:
            assert O.temp_decay(t, int(t_max), float(rel), 20, 2) == want
    # known answers from the reference's own logs: b=19.68 @4500, 3.61 @19500 of 21000
    assert round(O.temp_decay(4500, 21000, 0.2, 20, 2), 2) == 19.68
    assert round(O.temp_decay(19500, 21000, 0.2, 20, 2), 2) == 3.61


def test_schedule_of_the_long_reference_run(golden):
    """The (b, count) columns and the regulariser's warm-up of the REAL reference over 3000 full-size iterations
    (config1_hnerv3m_long.npz, make_golden.py::gen_config1_long) against the oracle's restatement of the two-phase schedule
    (calib_model.py:134-142, 170-226; data_utils.py:24-41): phase 1 = int(0.05 * iters / n_batches) epochs without the
    regulariser, the counter restarts, phase 2 switches the regulariser on at count = warmup * iters."""
    z = golden("config1_hnerv3m_long.npz")
    ref, iters, n_batches = z["loss_log"], int(z["iters"]), 4
    ep1 = int(0.05 * iters / n_batches)
    want = [(0.0, c) for c in range(1, ep1 * n_batches + 1)]
    for c in range(1, (int(iters / n_batches) - ep1) * n_batches + 1):
        want.append((O.temp_decay(c, iters, 0.2, 20, 2) if c >= iters * 0.2 else 0.0, c))
    want = np.array(want)
    assert ref.shape == (len(want), 4) and len(want) == 3000
    np.testing.assert_array_equal(ref[:, 3], want[:, 1])
    np.testing.assert_allclose(ref[:, 2], want[:, 0], rtol=1e-7, atol=0)   # (the fixture stores the log as float32)
    on = want[:, 0] > 0
    assert (ref[~on, 1] == 0).all() and (ref[on, 1] > 0).all()             # regulariser exactly where b is defined


# ------------------------------------------------------------------ QuantModule composition
@pytest.mark.parametrize("shape", ((8, 5, 3), (12, 16, 1), (6, 37, 5)))
@pytest.mark.parametrize("had", (False, True))
def test_quantmodule(golden, shape, had):
    z = golden("quantmodule.npz")
    co, ci, k = shape
    tag = f"c{co}_{ci}_{k}_{'h' if had else 'n'}"
    w, b, x, go = (T(z[f"{tag}_{n}"]) for n in ("w", "b", "x", "go"))
    L = O.ConvLayer(w, b)
    dec = O.Decoder("hnerv", [L])
    dec.out_bias = "0"
    qs = O.QuantStack(dec, [4], had)
    eq(L.wd, z[f"{tag}_wdelta"]); eq(L.wz, z[f"{tag}_wzp"]); eq(L.bd, z[f"{tag}_bdelta"]); eq(L.bz, z[f"{tag}_bzp"])
    if had:
        eq(L.hw, z[f"{tag}_hw"], atol=1e-6)
    tol = dict(rtol=1e-5, atol=2e-5) if had else dict(rtol=0, atol=0)
    eq(dec.forward(x), z[f"{tag}_y_fp"])
    L.wd.requires_grad_(True); L.bd.requires_grad_(True)
    y = qs.forward(x)
    eq(y.detach(), z[f"{tag}_y_uaq"], **tol)
    (y * go).sum().backward()
    eq(L.wd.grad, z[f"{tag}_dwdelta"], rtol=1e-4, atol=1e-4)
    eq(L.bd.grad, z[f"{tag}_dbdelta"], rtol=1e-4, atol=1e-4)
    L.wd = L.wd.detach(); L.bd = L.bd.detach()
    qs.to_adaround()
    L.wa, L.ba = T(z[f"{tag}_walpha"]).requires_grad_(True), T(z[f"{tag}_balpha"]).requires_grad_(True)
    xin = x.clone().requires_grad_(True)
    y = qs.forward(xin)
    eq(y.detach(), z[f"{tag}_y_ada"], **tol)
    (y * go).sum().backward()
    eq(L.wa.grad, z[f"{tag}_dwalpha"], rtol=1e-4, atol=1e-6)
    eq(L.ba.grad, z[f"{tag}_dbalpha"], rtol=1e-4, atol=1e-6)
    eq(xin.grad, z[f"{tag}_dx"], rtol=1e-4, atol=1e-5)
    L.w_soft = False
    eq(qs.forward(x).detach(), z[f"{tag}_y_hard"], **tol)


def test_fwht_properties():
    g = torch.Generator().manual_seed(0)
    for n in (1, 2, 16, 64, 256):
        x = torch.randn(7, n, generator=g)
        y = O.fwht(x)
        eq(O.fwht(y), x, atol=1e-5)                                   # involution (quant_layer.py:93-100)
        eq((y * y).sum(-1), (x * x).sum(-1), rtol=1e-5)               # Parseval
    h4 = O.fwht(torch.eye(4)) * 2
    eq(h4, [[1, 1, 1, 1], [1, -1, 1, -1], [1, 1, -1, -1], [1, -1, -1, 1]])


# ------------------------------------------------------------------ decode
@pytest.mark.parametrize("arch", ("hnerv", "nerv"))
def test_decode(golden, arch):
    z = golden("decode.npz")
    sd = state_dict_from_npz(z, f"{arch}_sd:")
    fc = (1, 1) if arch == "hnerv" else (1, 2)
    dec = O.Decoder.from_state_dict(sd, arch, [5, 4, 4, 2, 2], fc)
    emb = T(z[f"{arch}_emb"])
    y = dec.forward(emb)
    eq(y[..., ::7, ::7], z[f"{arch}_y_fp_sub"])
    eq(y.double().sum(), z[f"{arch}_y_fp_sum"], rtol=1e-12)
    for had in (False, True):
        tag = f"{arch}_{'h' if had else 'n'}"
        qs = O.QuantStack(O.Decoder.from_state_dict(sd, arch, [5, 4, 4, 2, 2], fc), BITS, had)
        assert qs.avg_bits() == float(z[f"{tag}_avgbits"])
        yq = qs.forward(emb)
        eq(yq[..., ::7, ::7], z[f"{tag}_y_q_sub"], atol=(2e-5 if had else 0))


# ------------------------------------------------------------------ trajectories (real model_reconstruction)
def _traj(golden, name, arch, had):
    z = golden(name)
    sd = state_dict_from_npz(z, "sd:")
    frames = T(golden("frames_320x640.npz")["frames"]).float() / 255.0
    dec = O.Decoder.from_state_dict(sd, arch, [5, 4, 4, 2, 2], (1, 1) if arch == "hnerv" else (1, 2))
    emb = T(z["emb"])
    eq(O.psnr_per_frame(dec.forward(emb), frames), z["psnr_fp"], atol=1e-4)
    qs = O.QuantStack(dec, BITS, had)
    assert qs.avg_bits() == float(z["avgbits"])
    for li, L in enumerate(dec.layers):
        eq(L.wd, z[f"init_wdelta{li}"]); eq(L.wz, z[f"init_wzp{li}"])
        eq(L.bd, z[f"init_bdelta{li}"]); eq(L.bz, z[f"init_bzp{li}"])
    with torch.no_grad():
        eq(O.psnr_per_frame(qs.forward(emb), frames), z["psnr_q_noopt"], atol=1e-3)
    log = np.array(O.calibrate(qs, emb, frames, z["order"], int(z["iters"])))
    ref = z["loss_log"]
    assert log.shape == ref.shape
    eq(log[:, 2:], ref[:, 2:])                       # temperature b and counters: exact
    return z, qs, emb, frames, log, ref


def test_trajectory_hnerv(golden):
    z, qs, emb, frames, log, ref = _traj(golden, "traj_hnerv.npz", "hnerv", False)
    # same ops, same thread count, same machine class: trajectories coincide to fp32 noise
    eq(log[:, 0], ref[:, 0], rtol=2e-4)
    eq(log[:, 1], ref[:, 1], rtol=2e-4, atol=1e-6)
    with torch.no_grad():
        p = O.psnr_per_frame(qs.forward(emb), frames)
    assert abs(float(p.mean()) - float(z["psnr_q_opt"].mean())) < 0.02      # north-star bar
    for li, L in enumerate(qs.dec.layers):
        eq(L.wd, z[f"fin_wdelta{li}"], rtol=1e-4)
        agree = ((L.wa >= 0).numpy() == (z[f"fin_walpha{li}"] >= 0)).mean()
        assert agree > 0.995, (li, agree)


def test_trajectory_nerv_hadamard(golden):
    z, qs, emb, frames, log, ref = _traj(golden, "traj_nerv_had.npz", "nerv", True)
    eq(log[:, 0], ref[:, 0], rtol=1e-3)
    with torch.no_grad():
        p = O.psnr_per_frame(qs.forward(emb), frames)
    assert abs(float(p.mean()) - float(z["psnr_q_opt"].mean())) < 0.02


# ------------------------------------------------------------------------------------------ Omega bit allocation
def _omega_setup(golden):
    zt, zo = golden("traj_hnerv.npz"), golden("omega.npz")
    sd = state_dict_from_npz(zt, "sd:")
    frames = T(golden("frames_320x640.npz")["frames"]).float() / 255.0
    emb = T(zt["emb"])
    batches = [(emb[i:i + 2], frames[i:i + 2]) for i in range(0, frames.shape[0], 2)]
    return zo, sd, batches


@pytest.mark.parametrize("ci", (0, 1))
def test_sensitivity_criteria_match_reference(golden, ci):
    """oracle sensitivity() vs the reference's own sensitivity_criterion (bit_assign.py:171-217) on the tiny HNeRV
    checkpoint: perturbations bit-exact, Omega = v'Hv and the diagonal-Fisher score to fp32 summation noise."""
    zo, sd, batches = _omega_setup(golden)
    bits = [int(b) for b in zo["bits"][ci]]
    dec = O.Decoder.from_state_dict(sd, "hnerv", [5, 4, 4, 2, 2])
    qs = O.QuantStack(dec, bits, False)
    assert qs.avg_bits() == float(zo[f"avgbits{ci}"])
    vec = O.weight_perturbation(qs)
    if ci == 1:
        for li, v in enumerate(vec):
            eq(v, zo[f"vec{li}"])
    for mode in ("omega", "fisher_diag"):
        total, per, acc = O.sensitivity(dec, vec, batches, mode)
        eq(np.array([float(p) for p in per]), zo[f"{mode}{ci}_layers"], rtol=2e-4, atol=1e-7 * abs(float(zo[f"{mode}{ci}"])))
        assert abs(float(total) - float(zo[f"{mode}{ci}"])) <= 2e-4 * abs(float(zo[f"{mode}{ci}"]))
        if ci == 1:
            for li, g in enumerate(acc):
                ref = zo[f"{mode}_acc{li}"]
                eq(g, ref, rtol=1e-3, atol=2e-5 * float(np.abs(ref).max()))


def test_omega_ranks_candidates_like_reference(golden):
    zo, sd, batches = _omega_setup(golden)
    scores = []
    for ci in (0, 1):
        dec = O.Decoder.from_state_dict(sd, "hnerv", [5, 4, 4, 2, 2])
        qs = O.QuantStack(dec, [int(b) for b in zo["bits"][ci]], False)
        scores.append(float(O.sensitivity(dec, O.weight_perturbation(qs), batches, "omega")[0]))
    assert (scores[0] < scores[1]) == (float(zo["omega0"]) < float(zo["omega1"]))


# ------------------------------------------------------------------ sensitivity of the trajectories to summation order
def test_trajectory_sensitivity_fixture(golden):
    """tests/golden/traj_sensitivity.json (made by make_sensitivity.py with this oracle): run the same way, the oracle
    follows the reference's trajectory bit for bit ('same_order'); with its convolutions perturbed at the last bit it
    decorrelates -- per-iteration loss ~1e-2, final scales a few %, masks ~80 % -- while the PSNR stays inside 0.02 dB.
    Re-checks one perturbed variant live (first 60 iterations) so the fixture cannot go stale silently."""
    with open(os.path.join(GOLDEN, "traj_sensitivity.json")) as f:
        sens = json.load(f)
    for tag in ("hnerv", "nerv_had"):
        so, sp = sens[tag]["same_order"], sens[tag]["spread"]
        assert so["loss_rel_all"] <= 2e-4 and so["dpsnr_dB"] < 1e-3 and so["mask_agreement"] > 0.999
        assert 1e-3 < sp["loss_rel_all"] < 5e-2           # the chaos is real ...
        assert sp["mask_agreement"] < 0.95
        assert sp["dpsnr_dB"] < 0.02                      # ... and the PSNR bar still holds for the reference's own noise
    import sys
    sys.path.insert(0, GOLDEN)
    import make_sensitivity as ms
    z = golden("traj_hnerv.npz")
    frames = T(golden("frames_320x640.npz")["frames"]).float() / 255.0
    dec = O.Decoder.from_state_dict(state_dict_from_npz(z, "sd:"), "hnerv", [5, 4, 4, 2, 2])
    dec.conv_fn = ms.make_conv_perm()
    qs = O.QuantStack(dec, BITS, hadamard=False)
    log = np.array(O.calibrate(qs, T(z["emb"]), frames, z["order"], int(z["iters"]), max_steps=60))
    rel = np.abs(log[:, 0] - z["loss_log"][:60, 0]) / np.abs(z["loss_log"][:60, 0])
    assert rel[:3].max() <= 2e-4
    assert 1e-4 < rel.max() <= 3 * sens["hnerv"]["spread"]["loss_rel_all"], rel.max()


def test_config1_reference_fixture_oracle(golden):
    """The oracle against the full-size reference fixture (config1_hnerv3m.npz: the real reference's model_reconstruction on
    HNeRV-3M at 640x1280): initial scales bit-exact, PSNR of the quantised model, and the first 3 of its 48 logged
    iterations (the CPU suite stays short; the GPU suite checks all 48 against the same fixture)."""
    z, ck = golden("config1_hnerv3m.npz"), golden("hnerv3m_bunny8_f16.npz")
    small = golden("frames_320x640.npz")["frames"]
    frames = torch.from_numpy(np.repeat(np.repeat(small, 2, axis=2), 2, axis=3).copy()).float() / 255.0
    sd = {k[3:].replace("/", "."): torch.from_numpy(v.astype(np.float32)) for k, v in ck.items() if k.startswith("sd:")}
    dec = O.Decoder.from_state_dict(sd, "hnerv", [5, 4, 4, 2, 2])
    qs = O.QuantStack(dec, BITS, hadamard=False)
    assert qs.avg_bits() == float(z["avgbits"])
    for li, L in enumerate(dec.layers):
        eq(L.wd, z[f"init_wdelta{li}"]); eq(L.bd.view(-1), z[f"init_bdelta{li}"].reshape(-1))
    emb = T(ck["emb"])
    with torch.no_grad():
        eq(O.psnr_per_frame(qs.forward(emb[:2]), frames[:2]), z["psnr_q_noopt"][:2], atol=1e-3)
    log = np.array(O.calibrate(qs, emb, frames, z["order"], int(z["iters"]), max_steps=3))
    eq(log[:, 2:], z["loss_log"][:3, 2:])
    eq(log[:, 0], z["loss_log"][:3, 0], rtol=2e-5)
    eq(log[:, 1], z["loss_log"][:3, 1], rtol=2e-5)
