"""neuroquant_amd -- MI355X-native network-wise calibration engine with NeuroQuant's Python API.

    from neuroquant_amd.quantization import QuantModel, QuantModule, model_reconstruction
    from neuroquant_amd.models import HNeRV, NeRV

The quantised decoder (fake-quant, Hadamard, conv+PixelShuffle+GELU, loss, Adam) runs in hand-written
gfx950 kernels behind libnqhip.so (include/nq_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
