"""ctypes binding of libnqhip.so (include/nq_hip.h).  There is no fallback: if the HIP library is
missing the import of any op raises, loudly."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p, POINTER, Structure

_HERE = os.path.dirname(os.path.abspath(__file__))
# NQ_LIB points the loader at another build of the SAME library (kernel A/B runs, tools/bench_kernels.py); default in-tree
LIB_PATH = os.environ.get("NQ_LIB") or os.path.join(_HERE, "libnqhip.so")

NQ_OK = 0
ABI_VERSION = 5   # nq_abi_version() of the library this binding was written against (include/nq_hip.h)
EPI_PLAIN, EPI_PS_GELU, EPI_TANH, EPI_PS, EPI_DGRAD_GELU = 0, 1, 2, 3, 4


class NQLibraryError(RuntimeError):
    pass


class AdaSeg(Structure):
    """nq_ada_seg (include/nq_hip.h): one tensor of a multi-tensor AdaRound launch."""
    _fields_ = [("x", c_void_p), ("gy", c_void_p), ("alpha", c_void_p), ("delta", c_void_p), ("zp", c_void_p),
                ("out", c_void_p), ("rows", c_int64), ("row_len", c_int64), ("per_row", c_int), ("n_levels", c_int),
                ("soft", c_int), ("reg_weight", c_float)]


class WL3Seg(Structure):
    """nq_wl3_seg (include/nq_hip.h)."""
    _fields_ = [("w", c_void_p), ("wt3", c_void_p), ("Cin", c_int), ("Cout", c_int), ("k", c_int), ("transposed", c_int)]


class WLSeg(Structure):
    """nq_wl_seg (include/nq_hip.h)."""
    _fields_ = [("w", c_void_p), ("wt_fwd", c_void_p), ("wt_bwd", c_void_p), ("Cout", c_int), ("Cin", c_int), ("k", c_int),
                ("krows_fwd", c_int), ("ld_fwd", c_int), ("krows_bwd", c_int), ("ld_bwd", c_int)]


class FwhtSeg(Structure):
    """nq_fwht_seg (include/nq_hip.h)."""
    _fields_ = [("x", c_void_p), ("y", c_void_p), ("outer", c_int64), ("inner", c_int64), ("n", c_int), ("n_in", c_int),
                ("n_out", c_int)]


class AdaAdamSeg(Structure):
    """nq_ada_adam_seg (include/nq_hip.h)."""
    _fields_ = [("x", c_void_p), ("gy", c_void_p), ("alpha", c_void_p), ("delta", c_void_p), ("zp", c_void_p), ("m", c_void_p),
                ("v", c_void_p), ("rows", c_int64), ("row_len", c_int64), ("per_row", c_int), ("n_levels", c_int),
                ("reg_weight", c_float)]


class WgrSeg(Structure):
    """nq_wgr_seg (include/nq_hip.h): one pending slab reduction."""
    _fields_ = [("slab", c_void_p), ("slab_db", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("Cout", c_int), ("N", c_int),
                ("co_pad", c_int), ("n_pad", c_int), ("nsplit", c_int), ("swap_kk", c_int), ("sg", c_int)]


class FqFwhtSeg(Structure):
    """nq_fq_fwht_seg (include/nq_hip.h): one tensor of a launch that fuses the AdaRound fake-quant with the Hadamard transform."""
    _fields_ = [("x", c_void_p), ("alpha", c_void_p), ("delta", c_void_p), ("zp", c_void_p), ("m", c_void_p), ("v", c_void_p),
                ("gy", c_void_p), ("y", c_void_p), ("outer", c_int64), ("inner", c_int64), ("n", c_int), ("c_in", c_int),
                ("per_row", c_int), ("n_levels", c_int), ("soft", c_int), ("reg_weight", c_float)]


class AdamSeg(Structure):
    """nq_adam_seg (include/nq_hip.h)."""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int64)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise NQLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C neuroquant_amd/csrc`).  neuroquant_amd has no CPU/eager fallback.")
    # ONE HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and libnqhip.so is linked against the system one
    # (same SONAME).  Whichever is loaded first serves both -- so torch goes first: with the system runtime loaded first,
    # torch later brings its own, this library's kernels are registered with the other one, and every launch fails
    # (seen with __graft_entry__.build() followed by smoke() in one process).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    P, I, L, F = c_void_p, c_int, c_int64, c_float

    def sig(name, res, *args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("nq_abi_version", I)
    # a stale / foreign build (NQ_LIB) would be called with shifted arguments: wild device writes instead of an error
    got = lib.nq_abi_version()
    if got != ABI_VERSION:
        raise NQLibraryError(f"{LIB_PATH} has ABI version {got}, this package binds version {ABI_VERSION}: rebuild it "
                             "(`make -C neuroquant_amd/csrc`)")
    sig("nq_error_string", c_char_p, I)
    sig("nq_scale_init_max", I, P, L, L, I, P, P, P)
    sig("nq_uaq_forward", I, P, P, P, P, L, L, I, I, P)
    sig("nq_uaq_backward", I, P, P, P, P, P, P, L, L, I, I, P)
    sig("nq_adaround_init", I, P, P, P, P, P, P, L, L, I, P)
    sig("nq_adaround_forward", I, P, P, P, P, P, P, L, L, I, I, I, P)
    sig("nq_adaround_backward", I, P, P, P, P, P, P, L, L, I, I, F, F, P)
    sig("nq_adaround_adam_multi", I, POINTER(AdaAdamSeg), I, F, F, F, F, F, F, P, P)
    sig("nq_adaround_forward_multi", I, POINTER(AdaSeg), I, P)
    sig("nq_uaq_forward_multi", I, POINTER(AdaSeg), I, P)
    sig("nq_uaq_backward_multi", I, POINTER(AdaSeg), I, P)
    sig("nq_adaround_backward_multi", I, POINTER(AdaSeg), I, F, P)
    sig("nq_adam_step_multi", I, POINTER(AdamSeg), I, F, F, F, F, F, P)
    sig("nq_step_prologue", I, P, P, P, P, P, I, I, P)
    sig("nq_step_prologue_gather", I, P, P, P, P, P, I, I, P, L, L, P, P)
    sig("nq_adaround_backward_multi_dyn", I, POINTER(AdaSeg), I, P, P)
    sig("nq_adam_step_multi_dyn", I, POINTER(AdamSeg), I, P, F, F, F, P)
    sig("nq_reduce_ws_floats", L, L)
    sig("nq_round_loss", I, P, L, F, F, P, P, I, P)
    sig("nq_round_loss_backward", I, P, L, F, F, P, P, I, P)
    sig("nq_adam_step", I, P, P, P, P, L, F, F, F, F, F, P)
    sig("nq_fwht", I, P, P, L, I, L, I, I, P)
    sig("nq_fwht_multi", I, POINTER(FwhtSeg), I, P)
    sig("nq_weight_layouts", I, P, P, P, I, I, I, I, I, I, I, P)
    sig("nq_conv_operand_dims", I, I, I, I, POINTER(c_int), POINTER(c_int))
    sig("nq_conv_forward_ws_floats", L, I, I, I, I, I, I)
    sig("nq_conv_forward", I, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P)
    sig("nq_conv3_supported", I, I, I, I, I, I, I)
    sig("nq_conv3_split_io", I, I, I, I, I, I, I)
    sig("nq_conv_split_out", I, I, I, I, I, I, I, I, I, I, I)
    sig("nq_split_words", I, P, P, L, P)
    sig("nq_conv_wgrad3_split_io", I, I, I, I, I, I, I)
    sig("nq_conv_wgrad3_fmt", I, P, P, P, P, P, I, I, I, I, I, I, I, P)
    sig("nq_conv_wgrad3_slabs_fmt", I, P, P, P, P, P, I, I, I, I, I, I, POINTER(WgrSeg), I, P)
    sig("nq_conv3_weight_bytes", L, I, I, I)
    sig("nq_weight_layout3", I, P, P, I, I, I, I, P)
    sig("nq_weight_layout3_multi", I, POINTER(WL3Seg), I, P)
    sig("nq_weight_layouts_multi", I, POINTER(WLSeg), I, P)
    sig("nq_weight_layouts_all", I, POINTER(WL3Seg), I, POINTER(WLSeg), I, P)
    sig("nq_conv_forward3_ws_floats", L, I, I, I, I, I, I)
    sig("nq_conv_forward3", I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P)
    sig("nq_conv_wgrad3_supported", I, I, I, I, I, I, I)
    sig("nq_conv_wgrad3_ws_floats", L, I, I, I, I, I, I)
    sig("nq_conv_wgrad3_plan", I, I, I, I, I, I, I, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int))
    sig("nq_conv_wgrad3", I, P, P, P, P, P, I, I, I, I, I, I, P)
    sig("nq_conv_wgrad3_swapped", I, P, P, P, P, I, I, I, I, I, I, P)
    sig("nq_conv_wgrad3_slabs", I, P, P, P, P, P, I, I, I, I, I, I, POINTER(WgrSeg), P)
    sig("nq_conv_wgrad3_swapped_slabs", I, P, P, P, P, I, I, I, I, I, I, POINTER(WgrSeg), P)
    sig("nq_conv_wgrad_slabs", I, P, P, P, P, P, I, I, I, I, I, I, I, POINTER(WgrSeg), P)
    sig("nq_wgrad_reduce_multi", I, POINTER(WgrSeg), I, P)
    sig("nq_conv_wgrad_ws_floats", L, I, I, I, I, I, I)
    sig("nq_conv_wgrad", I, P, P, P, P, P, I, I, I, I, I, I, I, P)
    sig("nq_ps_gelu_backward", I, P, P, P, I, I, I, I, I, P)
    sig("nq_tanh_out_backward", I, P, P, P, L, P)
    sig("nq_l2_loss", I, P, P, P, P, P, L, L, F, P)
    sig("nq_channel_sum", I, P, P, P, I, I, L, P)
    sig("nq_l2_loss_tanh_head", I, P, P, P, P, P, P, P, P, I, I, L, L, F, P)
    sig("nq_frame_sse", I, P, P, P, L, L, P)
    sig("nq_gather_frames_u8", I, P, P, P, L, L, P)
    sig("nq_adaround_fwht_multi", I, POINTER(FqFwhtSeg), I, P)
    sig("nq_fwht_adaround_adam_multi", I, POINTER(FqFwhtSeg), I, F, F, F, F, F, F, P, P)
    sig("nq_head_forward_loss_ws_floats", L, I, I, I)
    sig("nq_head_forward_loss", I, P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, L, F, P)
    sig("nq_act_dd", I, P, P, P, P, L, I, P)
    sig("nq_pixel_shuffle", I, P, P, I, I, I, I, I, I, P)
    sig("nq_bias_add", I, P, P, P, I, I, L, P)
    return lib


EXPORTS = (
    "nq_abi_version", "nq_error_string", "nq_scale_init_max", "nq_uaq_forward", "nq_uaq_backward",
    "nq_adaround_init", "nq_adaround_forward", "nq_adaround_backward", "nq_reduce_ws_floats", "nq_round_loss", "nq_round_loss_backward",
    "nq_adam_step", "nq_adaround_adam_multi", "nq_adaround_forward_multi", "nq_uaq_forward_multi", "nq_uaq_backward_multi", "nq_adaround_backward_multi", "nq_adam_step_multi", "nq_step_prologue", "nq_step_prologue_gather", "nq_adaround_backward_multi_dyn", "nq_adam_step_multi_dyn", "nq_fwht", "nq_fwht_multi", "nq_weight_layouts", "nq_conv_operand_dims", "nq_conv_forward_ws_floats", "nq_conv_forward",
    "nq_conv3_supported", "nq_conv3_weight_bytes", "nq_weight_layout3", "nq_weight_layout3_multi", "nq_weight_layouts_multi", "nq_conv_forward3_ws_floats", "nq_conv_forward3",
    "nq_conv_wgrad3_supported", "nq_conv_wgrad3_ws_floats", "nq_conv_wgrad3_plan", "nq_conv_wgrad3", "nq_conv_wgrad3_swapped",
    "nq_conv_wgrad3_slabs", "nq_conv_wgrad3_swapped_slabs", "nq_conv_wgrad_slabs", "nq_wgrad_reduce_multi",
    "nq_conv_wgrad_ws_floats", "nq_conv_wgrad", "nq_ps_gelu_backward", "nq_tanh_out_backward", "nq_l2_loss",
    "nq_channel_sum", "nq_l2_loss_tanh_head", "nq_frame_sse", "nq_gather_frames_u8",
    "nq_act_dd", "nq_pixel_shuffle", "nq_bias_add", "nq_conv3_split_io", "nq_conv_split_out", "nq_split_words",
    "nq_conv_wgrad3_split_io", "nq_conv_wgrad3_fmt", "nq_conv_wgrad3_slabs_fmt", "nq_head_forward_loss_ws_floats", "nq_head_forward_loss",
    "nq_adaround_fwht_multi", "nq_fwht_adaround_adam_multi", "nq_weight_layouts_all",
)

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def check(rc, what=""):
    if rc != NQ_OK:
        msg = lib().nq_error_string(rc).decode()
        raise NQLibraryError(f"libnqhip {what}: {msg} ({rc})")
