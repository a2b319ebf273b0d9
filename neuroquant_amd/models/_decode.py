"""Shared decode schedule of HNeRV / NeRV (reference HNeRV.py:49-71, NeRV.py:44-65)."""
import torch

from ._layers import OutImg


def run_decoder(model, img_embed, embed_after_reshape: bool):
    """layer 0 -> (fc_h, fc_w) channel->space reshape -> blocks -> head -> OutImg.

    The head conv and OutImg('tanh') fuse into one HIP launch when the head has been wrapped by QuantModel
    (QuantModule.forward_out_img); otherwise they run as in the reference.
    """
    embed_list = [img_embed]
    output = model.decoder[0](img_embed)
    if not embed_after_reshape:
        embed_list.append(output)
    n, c, h, w = output.shape
    if model.fc_h != 1 or model.fc_w != 1:
        output = output.view(n, -1, model.fc_h, model.fc_w, h, w).permute(0, 1, 4, 2, 5, 3) \
                       .reshape(n, -1, model.fc_h * h, model.fc_w * w)
    if embed_after_reshape:
        embed_list.append(output)
    for layer in model.decoder[1:]:
        output = layer(output)
        embed_list.append(output)
    fused_head = getattr(model.head_layer, 'forward_out_img', None)
    if fused_head is not None and model.out_bias == 'tanh' and output.is_cuda:
        img_out = fused_head(output)
    else:
        img_out = OutImg(model.head_layer(output), model.out_bias)
    return img_out, embed_list
