"""NeRV (reference models/NeRV.py): positional encoding of the frame index -> 1x1 conv -> channel->space
reshape -> 5 NeRV blocks -> 3x3 head -> tanh."""
import time

import numpy as np
import torch
import torch.nn as nn

from ._layers import NeRVBlock, OutImg, PositionEncoding
from ._decode import run_decoder


class NeRV(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc_h = int(cfg['crop_h'] // np.prod(cfg['dec_strides']))
        self.fc_w = int(cfg['crop_w'] // np.prod(cfg['dec_strides']))
        base, level = cfg['base'], cfg['level']
        self.encoder = PositionEncoding(base, level)
        in_channel = cfg['dec_in_channel']
        dec_layers = [nn.Conv2d(int(level * 2), in_channel * self.fc_h * self.fc_w, 1, 1, 0)]
        for ks, stride in zip(cfg['dec_kernels'], cfg['dec_strides']):
            out_channel = int(max(round(in_channel / cfg['channel_reduce']), cfg['channel_lbound']))
            dec_layers.append(NeRVBlock(in_channel, out_channel, ks, stride, bias=True, norm=cfg['dec_norm'],
                                        act=cfg['dec_acts']))
            in_channel = out_channel
        self.decoder = nn.ModuleList(dec_layers)
        self.head_layer = nn.Conv2d(in_channel, 3, 3, 1, 1)
        self.out_bias = cfg['out_bias']
        self.sync_decode = None

    def encode(self, img):
        return self.encoder(img[:, None]).float()

    def decode(self, img_embed):
        dec_start = time.time()
        img_out, embed_list = run_decoder(self, img_embed, embed_after_reshape=True)
        sync = (not torch.is_grad_enabled()) if self.sync_decode is None else self.sync_decode
        if sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        return img_out, embed_list, time.time() - dec_start

    def forward(self, input):
        return self.decode(self.encode(input))
