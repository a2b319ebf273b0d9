"""Shared decode schedule of HNeRV / NeRV (reference HNeRV.py:49-71, NeRV.py:44-65)."""
import torch

from ._layers import OutImg


def _fused_stack(model):
    """(DecoderSpec, [QuantModule...]) when the whole decoder has been wrapped by QuantModel and is the shipped
    shape (conv -> PixelShuffle -> exact GELU blocks, tanh head); None otherwise."""
    from .. import ops
    from ..quantization.quant_block import QuantNeRVBlock
    from ..quantization.quant_layer import QuantModule
    if model.out_bias != 'tanh':
        return None
    first, head = model.decoder[0], model.head_layer
    if not (isinstance(first, QuantModule) and isinstance(head, QuantModule) and first._hip_ok and head._hip_ok):
        return None
    mods, layers = [first], [(first.weight.shape[-1], 1, False)]
    for blk in model.decoder[1:]:
        if not (isinstance(blk, QuantNeRVBlock) and blk._fusable and blk.conv._hip_ok):
            return None
        mods.append(blk.conv)
        layers.append((blk.conv.weight.shape[-1], blk._r, True))
    mods.append(head)
    layers.append((head.weight.shape[-1], 1, False))
    return ops.DecoderSpec(layers, (model.fc_h, model.fc_w), True), mods


def run_decoder(model, img_embed, embed_after_reshape: bool):
    """layer 0 -> (fc_h, fc_w) channel->space reshape -> blocks -> head -> OutImg.

    With autograd enabled and a fully wrapped decoder the whole stack runs as ONE fused autograd node
    (ops.decoder_stack): activations are never materialised, so embed_list then only carries the input embedding.
    Without autograd (evaluation) every layer runs on its own fused kernel and embed_list is complete, as in the
    reference.
    """
    if torch.is_grad_enabled() and img_embed.is_cuda:
        fused = _fused_stack(model)
        if fused is not None:
            from .. import ops
            spec, mods = fused
            img_out = ops.decoder_stack(img_embed, spec, [m._current_params() for m in mods])
            return img_out, [img_embed]
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
