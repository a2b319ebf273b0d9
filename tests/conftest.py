import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.asarray(a).copy())


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_npz(name)
        return cache[name]

    return get


def state_dict_from_npz(z, prefix):
    """keys were stored as '<prefix>a/b/c' -> 'a.b.c'"""
    return {k[len(prefix):].replace("/", "."): T(v) for k, v in z.items() if k.startswith(prefix)}


TINY_HNERV = dict(crop_h=320, crop_w=640, diff_enc=False, stage_block=1, enc_strides=[5, 4, 4, 2, 2],
                  enc_channel=[16, 16, 16, 16, 8], channel_reduce=1.2, channel_lbound=6, dec_in_channel=12,
                  dec_kernels=[1, 3, 5, 5, 5], dec_strides=[5, 4, 4, 2, 2], dec_norm="none", dec_acts="gelu",
                  out_bias="tanh")
TINY_NERV = dict(crop_h=320, crop_w=640, diff_enc=False, base=1.25, level=20, channel_reduce=2, channel_lbound=6,
                 dec_in_channel=20, dec_kernels=[3, 3, 3, 3, 3], dec_strides=[5, 4, 4, 2, 2], dec_norm="none",
                 dec_acts="gelu", out_bias="tanh")
HNERV_3M = dict(crop_h=640, crop_w=1280, diff_enc=False, stage_block=1, enc_strides=[5, 4, 4, 2, 2],
                enc_channel=[64, 64, 64, 64, 16], channel_reduce=1.2, channel_lbound=12, dec_in_channel=92,
                dec_kernels=[1, 3, 5, 5, 5], dec_strides=[5, 4, 4, 2, 2], dec_norm="none", dec_acts="gelu",
                out_bias="tanh")
NERV_3M = dict(crop_h=640, crop_w=1280, diff_enc=False, base=1.25, level=80, channel_reduce=2, channel_lbound=24,
               dec_in_channel=145, dec_kernels=[3, 3, 3, 3, 3], dec_strides=[5, 4, 4, 2, 2], dec_norm="none",
               dec_acts="gelu", out_bias="tanh")
BITS = [6, 5, 4, 5, 5, 6, 6]
