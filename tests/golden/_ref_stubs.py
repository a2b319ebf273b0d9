"""In-process stand-ins for the four pip packages the reference imports but this
image lacks.  Used ONLY by make_golden.py, in the build container,
to import /root/reference read-only and capture golden vectors.  Never imported by
the product, the oracle, or any test.

  hadamard_transform  -> quantization/quant_layer.py:7   (FWHT: arithmetic ON the path;
                         the pip package's source is not in the reference tree and no
                         version is pinned, so Hadamard goldens are "parity unpinned":
                         they pin everything around the transform, not its summation order)
  timm.models.layers  -> models/_layers.py:6             (init helper + DropPath; encoder only)
  pytorch_msssim      -> utils.py:12                     (MS-SSIM, not judged, out of scope)
  torchvision         -> videosets/datasets.py:2-3, methods/calibrate_network.py:14
"""
import sys
import types
import math

import numpy as np
import torch


def _fwht_normalized(x: torch.Tensor) -> torch.Tensor:
    """Orthonormal Sylvester-ordered Walsh-Hadamard transform along the last dim."""
    n = x.shape[-1]
    assert n & (n - 1) == 0 and n > 0, "length must be a power of two"
    y = x.clone()
    h = 1
    lead = x.shape[:-1]
    while h < n:
        y = y.reshape(*lead, n // (2 * h), 2, h)
        a = y[..., 0, :]
        b = y[..., 1, :]
        y = torch.stack((a + b, a - b), dim=-2).reshape(*lead, n)
        h *= 2
    return y / math.sqrt(n)


def _pad_to_power_of_2(x: torch.Tensor) -> torch.Tensor:
    n = x.shape[-1]
    p = 1 if n == 0 else 2 ** math.ceil(math.log2(n))
    return torch.nn.functional.pad(x, (0, p - n))


def install():
    if "hadamard_transform" not in sys.modules:
        m = types.ModuleType("hadamard_transform")
        m.hadamard_transform = _fwht_normalized
        m.pad_to_power_of_2 = _pad_to_power_of_2
        sys.modules["hadamard_transform"] = m

    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")

        class DropPath(torch.nn.Module):
            def __init__(self, drop_prob=0.0):
                super().__init__()
                self.drop_prob = drop_prob

            def forward(self, x):
                return x

        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        layers.DropPath = DropPath
        timm.models = models
        models.layers = layers
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = models
        sys.modules["timm.models.layers"] = layers

    if "pytorch_msssim" not in sys.modules:
        m = types.ModuleType("pytorch_msssim")

        def _unavailable(*a, **k):
            raise RuntimeError("pytorch_msssim is not available in this image")

        m.ms_ssim = _unavailable
        m.ssim = _unavailable
        sys.modules["pytorch_msssim"] = m

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        io = types.ModuleType("torchvision.io")
        tr = types.ModuleType("torchvision.transforms")
        trf = types.ModuleType("torchvision.transforms.functional")
        ut = types.ModuleType("torchvision.utils")

        def read_image(path):
            from PIL import Image
            a = np.asarray(Image.open(path).convert("RGB"))
            return torch.from_numpy(a.copy()).permute(2, 0, 1)

        def center_crop(img, size):
            th, tw = size
            h, w = img.shape[-2:]
            top = int(round((h - th) / 2.0))
            left = int(round((w - tw) / 2.0))
            return img[..., top:top + th, left:left + tw]

        def save_image(*a, **k):
            raise RuntimeError("torchvision.utils.save_image is not available in this image")

        io.read_image = read_image
        trf.center_crop = center_crop
        ut.save_image = save_image
        tv.io, tv.transforms, tv.utils = io, tr, ut
        tr.functional = trf
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.io"] = io
        sys.modules["torchvision.transforms"] = tr
        sys.modules["torchvision.transforms.functional"] = trf
        sys.modules["torchvision.utils"] = ut
