"""CPU-side checks of the boundary: libnqhip.so loads, exports every symbol include/nq_hip.h declares, rejects bad
arguments before touching the device, and the host-side mirror of the reference API behaves like the reference
(known answers from the reference's own logs)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, HNERV_3M, NERV_3M, TINY_HNERV, BITS


def test_header_symbols_exported():
    from neuroquant_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nq_hip.h")).read()
    declared = set(re.findall(r"\b(nq_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in nq_hip.h but not exported"
    assert declared == set(_lib.EXPORTS)
    assert lib.nq_abi_version() == 5 == _lib.ABI_VERSION
    assert lib.nq_error_string(-1) == b"invalid argument"
    # ... and nothing else: the library is built with -fvisibility=hidden, only NQ_API declarations are dynamic symbols
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in ("T", "t")}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_invalid_arguments_are_rejected_without_a_gpu():
    from neuroquant_amd import _lib
    lib = _lib.lib()
    n = None
    assert lib.nq_uaq_forward(n, n, n, n, 4, 4, 1, 16, n) == -1
    assert lib.nq_fwht(ctypes.c_void_p(16), ctypes.c_void_p(32), 1, 12, 1, 12, 12, n) == -1      # not a power of two
    assert lib.nq_fwht(ctypes.c_void_p(16), ctypes.c_void_p(32), 1, 2048, 1, 8, 8, n) == -2      # too long
    assert lib.nq_conv_forward(ctypes.c_void_p(16), ctypes.c_void_p(16), n, ctypes.c_void_p(16), n, n, 1, 4, 8, 8, 4, 7, 196,
                               16, 1, 0, 0, n, n) == -2                                       # k=7 not built
    assert lib.nq_conv_forward_ws_floats(2, 44, 320, 640, 148, 5) == 0                        # enough tiles: no split-K
    assert lib.nq_conv_forward_ws_floats(2, 848, 40, 80, 64, 5) > 0                           # dec3 data gradient: split-K
    kr, ld = ctypes.c_int(), ctypes.c_int()
    assert lib.nq_conv_operand_dims(44, 148, 5, ctypes.byref(kr), ctypes.byref(ld)) == 0
    assert kr.value == 44 * 25 and ld.value == 160
    assert lib.nq_conv_operand_dims(53, 176, 5, ctypes.byref(kr), ctypes.byref(ld)) == 0
    assert kr.value == 56 * 25 and ld.value == 176
    assert lib.nq_conv_wgrad_ws_floats(2, 44, 320, 640, 148, 5) > 0
    assert lib.nq_reduce_ws_floats(4096 * 3 + 1) == 4


def test_wgrad3_plan_keeps_big_operands_off_the_32bit_offset_kernel():
    """The producer/consumer weight-gradient kernel addresses x through signed and dy through unsigned 32-bit byte offsets
    (raw buffer loads); an operand of 2 GiB or more must take the 4-wave kernel (64-bit pointers) -- ADVICE r2.  Host-only:
    the plan is a pure function of the shape."""
    from neuroquant_amd import _lib
    lib = _lib.lib()

    def plan(B, cin, H, W, cout, k):
        v = [ctypes.c_int() for _ in range(4)]
        assert lib.nq_conv_wgrad3_plan(B, cin, H, W, cout, k, *[ctypes.byref(t) for t in v]) == 0
        return tuple(t.value for t in v)      # (mi, ni, nsplit, pc)

    assert plan(2, 44, 320, 640, 148, 5) == (5, 6, 42, 14)        # dec5 at the bench's size: 8 waves, 4-row segments
    assert plan(16, 44, 320, 640, 148, 5)[3] == 14                # per-GPU B = 16 (dY 1.94 GB): still below 2 GiB
    assert plan(18, 44, 320, 640, 148, 5)[3] == 0                 # dY = 18*148*320*640*4 = 2.18 GB
    assert plan(2, 148, 1600, 1280, 44, 5)[3] == 0                # x  = 2.42 GB
    assert plan(2, 3, 640, 1280, 37, 3)[3] == 4                   # the role-swapped head problem: 128-pixel segments
    assert plan(24, 3, 640, 1280, 37, 3)[3] == 0                  # ... with a 2.9 GB "dy" (the 37-channel activation)
    assert lib.nq_conv_wgrad3_plan(2, 44, 320, 640, 148, 7, *[ctypes.byref(ctypes.c_int()) for _ in range(4)]) == -1


def test_few_pixel_layers_are_routed_to_the_flat_kernels():
    """Round 4 launch plans, host-only (pure functions of the shape): the deep layers of both 3M models go to the few-pixel
    kernels (conv_flat3.hip: no slabs for a forward; conv_wgrad_flat3.hip: no slabs at all), toy convolutions stay on the
    exact-fp32 kernels, the big layers keep the tiled kernels, and a 3-chunk K loop is no longer split."""
    from neuroquant_amd import _lib
    lib = _lib.lib()
    for B, cin, H, W, cout, k in [(2, 77, 10, 20, 1024, 3), (2, 145, 2, 4, 1800, 3), (2, 72, 10, 20, 576, 3)]:
        assert lib.nq_conv3_supported(B, cin, H, W, cout, k) == 1
        assert lib.nq_conv_forward3_ws_floats(B, cin, H, W, cout, k) == 0          # forward: no slabs, no finish launch
        assert lib.nq_conv_wgrad3_supported(B, cin, H, W, cout, k) == 1
        assert lib.nq_conv_wgrad3_ws_floats(B, cin, H, W, cout, k) == 4            # a token workspace
    assert lib.nq_conv3_supported(2, 5, 6, 7, 8, 3) == 0                             # a toy layer: exact fp32
    assert lib.nq_conv_wgrad3_supported(2, 5, 6, 8, 8, 3) == 0
    assert lib.nq_conv_forward3_ws_floats(2, 44, 320, 640, 148, 5) == 0              # dec5: enough tiles
    assert lib.nq_conv_wgrad3_ws_floats(2, 44, 320, 640, 148, 5) > 1 << 20           # ... and split-K slabs for its weight gradient
    assert lib.nq_conv_forward3_ws_floats(2, 36, 40, 80, 384, 3) == 0                # NeRV dec3 forward: 3 chunks, not split (was 2 slabs)
    assert lib.nq_conv_forward3_ws_floats(2, 848, 40, 80, 64, 5) > 0                 # HNeRV dec3 data gradient: 53 chunks, split
    assert lib.nq_head_forward_loss_ws_floats(2, 640, 1280) == 2 * 5 * 160 * 4       # 4 floats per (frame, 256-column strip, 4-row block)


def test_split_word_interchange_plan():
    """ABI v5, host-only: which side of which launch offers the split {hi | lo} word form (include/nq_hip.h NQ_EPI_X_SPLIT /
    NQ_EPI_Y_SPLIT).  The tiled bf16x3 kernels of the big layers take and write it, the row-segment weight gradient takes both
    operands, the streaming head data gradient writes it; split-K launches do not write it, the few-pixel kernels, the 4-wave
    weight gradient and everything that is not bf16x3 know nothing of it."""
    from neuroquant_amd import _lib
    lib = _lib.lib()
    X, Y = 0x100, 0x200
    for shape in [(2, 44, 320, 640, 148, 5), (2, 148, 320, 640, 44, 5), (2, 53, 160, 320, 176, 5), (2, 176, 160, 320, 53, 5), (2, 64, 40, 80, 848, 5),
                  (2, 24, 320, 640, 96, 3), (2, 96, 320, 640, 24, 3), (2, 24, 160, 320, 96, 3), (2, 89, 480, 960, 296, 5)]:
        assert lib.nq_conv3_split_io(*shape) == X | Y, shape
    assert lib.nq_conv3_split_io(2, 848, 40, 80, 64, 5) == X            # HNeRV dec3 data gradient: split-K, the finish kernel writes floats
    for shape in [(2, 77, 10, 20, 1024, 3), (2, 145, 2, 4, 1800, 3)]:
        assert lib.nq_conv3_split_io(*shape) == Y, shape                 # few-pixel forward without slabs: writes, never reads
    assert lib.nq_conv3_split_io(2, 1024, 10, 20, 77, 3) == X           # HNeRV dec2 data gradient: tiled, split-K
    for shape in [(2, 5, 6, 7, 8, 3), (2, 44, 320, 640, 148, 7)]:
        assert lib.nq_conv3_split_io(*shape) == 0, shape                 # toy layer, unsupported k
    for shape in [(2, 44, 320, 640, 148, 5), (2, 53, 160, 320, 176, 5), (2, 64, 40, 80, 848, 5), (2, 24, 320, 640, 96, 3), (2, 24, 160, 320, 96, 3)]:
        assert lib.nq_conv_wgrad3_split_io(*shape) == 3, shape
    assert lib.nq_conv_wgrad3_split_io(2, 36, 40, 80, 384, 3) == 3      # NeRV dec3: 40 splits instead of 42 keep it on that kernel
    for shape in [(2, 77, 10, 20, 1024, 3), (2, 36, 20, 40, 384, 3), (2, 3, 640, 1280, 37, 3)]:
        assert lib.nq_conv_wgrad3_split_io(*shape) == 0, shape           # few-pixel, 4-wave and the role-swapped head kernels
    # nq_conv_split_out(B, Cin, H, W, Cout, k, r, epilogue, in_gelu, has_bias): the head's data gradient 3 -> 37 / 3 -> 24, un-shuffle 2
    assert lib.nq_conv_split_out(2, 3, 640, 1280, 37, 3, 2, 4, 0, 0) == 1
    assert lib.nq_conv_split_out(2, 3, 640, 1280, 24, 3, 2, 4, 0, 0) == 1
    assert lib.nq_conv_split_out(2, 3, 640, 1280, 37, 3, 2, 4, 0, 1) == 0   # (a bias: not the data gradient)
    assert lib.nq_conv_split_out(2, 37, 640, 1280, 3, 3, 1, 2, 0, 1) == 0   # the head forward
    assert lib.nq_conv_split_out(2, 44, 320, 640, 148, 5, 2, 1, 0, 1) == 0  # an fp32 matrix-pipe layer


def test_ops_refuse_cpu_tensors():
    from neuroquant_amd import ops
    with pytest.raises(RuntimeError):
        ops.uaq_forward(torch.zeros(2, 2), torch.ones(1), torch.zeros(1), 4)
    with pytest.raises(RuntimeError):
        ops.l2_loss(torch.zeros(1, 3, 2, 2), torch.zeros(1, 3, 2, 2))


@pytest.mark.parametrize("arch,cfg,bits,want", [
    ("hnerv", HNERV_3M, [6, 5, 4, 5, 5, 6, 6], 4.79399210722922),    # results/...052303.log:233
    ("hnerv", HNERV_3M, [2, 3, 4, 6, 4, 4, 2], 4.956511535893288),   # results/...132138.log:233
    ("nerv", NERV_3M, [6, 5, 4, 5, 5, 6, 6], 4.946213722986429),     # results/...080342.log:143
])
def test_average_bitwidth_known_answers(arch, cfg, bits, want):
    from neuroquant_amd.models import HNeRV, NeRV
    from neuroquant_amd.quantization import QuantModel, QuantModule, QuantNeRVBlock
    model = (HNeRV if arch == "hnerv" else NeRV)(cfg)
    dec = sum(p.numel() for p in model.decoder.parameters()) + sum(p.numel() for p in model.head_layer.parameters())
    qnn = QuantModel(model, hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True, scale_method="max"))
    assert qnn.set_bitwidth(bits) == want
    mods = qnn.quant_modules()
    assert len(mods) == 7 and isinstance(qnn.model.decoder[1], QuantNeRVBlock)
    assert isinstance(qnn.model.decoder[0], QuantModule) and isinstance(qnn.model.head_layer, QuantModule)
    assert sum(m.weight.numel() + m.bias.numel() for m in mods) == dec
    if arch == "hnerv":
        assert not any(isinstance(m, QuantModule) for m in qnn.model.encoder.modules())   # encoder left alone
        assert dec == 2646219                                                              # SURVEY §8 layer table
    else:
        assert dec == 3078499


def test_bitwidth_bounds_and_state_toggles():
    from neuroquant_amd.models import HNeRV
    from neuroquant_amd.quantization import QuantModel, UniformAffineQuantizer
    with pytest.raises(AssertionError):
        UniformAffineQuantizer(n_bits=9)
    q = UniformAffineQuantizer(n_bits=4)
    with pytest.raises(AssertionError):
        q.bitwidth_refactor(1)
    qnn = QuantModel(HNeRV(TINY_HNERV), hadamard=False, weight_quant_params=dict(n_bits=8, channel_wise=True))
    qnn.set_quant_state(True)
    assert all(m.use_weight_quant for m in qnn.quant_modules())
    qnn.set_quant_state(False)
    assert not any(m.use_weight_quant for m in qnn.quant_modules())
    qnn.set_bitwidth(BITS, init=False)
    assert [m.weight_quantizer.n_bits for m in qnn.quant_modules()] == BITS
    assert [m.bias_quantizer.n_levels for m in qnn.quant_modules()] == [2 ** b for b in BITS]


def test_state_dict_keys_match_reference_checkpoints(golden):
    from conftest import state_dict_from_npz
    from neuroquant_amd.models import HNeRV
    sd = state_dict_from_npz(golden("traj_hnerv.npz"), "sd:")
    model = HNeRV(TINY_HNERV)
    assert set(model.state_dict().keys()) == set(sd.keys())
    model.load_state_dict(sd, strict=True)


def test_temperature_schedule_matches_golden():
    import json
    from neuroquant_amd.quantization import LinearTempDecay
    tab = json.load(open(os.path.join(ROOT, "tests", "golden", "tempdecay.json")))
    for key, rows in tab.items():
        t_max, rel = key.split("_")
        sched = LinearTempDecay(int(t_max), rel_start_decay=float(rel), start_b=20, end_b=2)
        for t, want in rows:
            assert sched(t) == want
