"""HNeRV (reference models/HNeRV.py): ConvNeXt encoder -> 1x1 conv -> 5 NeRV blocks -> 3x3 head -> tanh."""
import time

import numpy as np
import torch
import torch.nn as nn

from ._layers import ConvNeXt, NeRVBlock, OutImg
from ._decode import run_decoder


class HNeRV(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        assert cfg['enc_strides'] == cfg['dec_strides']
        self.fc_h = int(np.prod(cfg['enc_strides']) // np.prod(cfg['dec_strides']))
        self.fc_w = int(np.prod(cfg['enc_strides']) // np.prod(cfg['dec_strides']))
        self.encoder = ConvNeXt(stage_blocks=cfg['stage_block'], strds=cfg['enc_strides'], dims=cfg['enc_channel'],
                                drop_path_rate=0)
        in_channel = cfg['dec_in_channel']
        dec_layers = [nn.Conv2d(cfg['enc_channel'][-1], in_channel, 1, 1, 0)]
        for ks, stride in zip(cfg['dec_kernels'], cfg['dec_strides']):
            out_channel = int(max(round(in_channel / cfg['channel_reduce']), cfg['channel_lbound']))
            dec_layers.append(NeRVBlock(in_channel, out_channel, ks, stride, bias=True, norm=cfg['dec_norm'],
                                        act=cfg['dec_acts']))
            in_channel = out_channel
        self.decoder = nn.ModuleList(dec_layers)
        self.head_layer = nn.Conv2d(in_channel, 3, 3, 1, 1)
        self.out_bias = cfg['out_bias']
        # the reference synchronises the device inside every decode (HNeRV.py:67-68), stalling the launch queue
        # once per calibration iteration; here only eval-mode decodes (where dec_time is reported) synchronise
        self.sync_decode = None

    def encode(self, img):
        return self.encoder(img)

    def decode(self, img_embed):
        dec_start = time.time()
        img_out, embed_list = run_decoder(self, img_embed, embed_after_reshape=False)
        sync = (not torch.is_grad_enabled()) if self.sync_decode is None else self.sync_decode
        if sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        return img_out, embed_list, time.time() - dec_start

    def forward(self, input):
        return self.decode(self.encode(input))
