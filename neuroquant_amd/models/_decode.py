"""Shared decode schedule of HNeRV / NeRV (reference HNeRV.py:49-71, NeRV.py:44-65)."""
import torch
import torch.nn as nn

from ._layers import OutImg


def _plain_conv_ok(conv):
    k = conv.kernel_size[0]
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size[0] == conv.kernel_size[1] and k in (1, 3, 5)
            and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (k // 2, k // 2)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is not None)


def _fused_stack(model):
    """(DecoderSpec, [(W, b) providers]) when the whole decoder is the shipped shape (conv -> PixelShuffle -> exact
    GELU blocks, tanh head), either wrapped by QuantModel (QuantModule / QuantNeRVBlock) or plain FP32 modules
    (nn.Conv2d / NeRVBlock: the FP32 trainer); None otherwise."""
    from .. import ops
    from ..quantization.quant_block import QuantNeRVBlock
    from ..quantization.quant_layer import QuantModule
    from ._layers import NeRVBlock
    if model.out_bias != 'tanh':
        return None

    def conv_of(m):
        """-> (callable returning (W, b), kernel size) for a quantised or plain conv, or None."""
        if isinstance(m, QuantModule):
            return (m._current_params, m.weight.shape[-1]) if m._hip_ok else None
        if isinstance(m, nn.Conv2d) and _plain_conv_ok(m):
            return (lambda m=m: (m.weight, m.bias)), m.kernel_size[0]
        return None

    first, head = conv_of(model.decoder[0]), conv_of(model.head_layer)
    if first is None or head is None:
        return None
    provs, layers = [first[0]], [(first[1], 1, False)]
    for blk in model.decoder[1:]:
        if isinstance(blk, QuantNeRVBlock) and blk._fusable:
            c, r = conv_of(blk.conv), blk._r
        elif isinstance(blk, NeRVBlock) and isinstance(blk.norm, nn.Identity) and isinstance(blk.act, nn.GELU) \
                and getattr(blk.act, 'approximate', 'none') == 'none' \
                and isinstance(blk.conv[1], (nn.PixelShuffle, nn.Identity)):
            c = conv_of(blk.conv[0])
            r = blk.conv[1].upscale_factor if isinstance(blk.conv[1], nn.PixelShuffle) else 1
        else:
            return None
        if c is None:
            return None
        provs.append(c[0])
        layers.append((c[1], r, True))
    provs.append(head[0])
    layers.append((head[1], 1, False))
    return ops.DecoderSpec(layers, (model.fc_h, model.fc_w), True), provs


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
            spec, provs = fused
            img_out = ops.decoder_stack(img_embed, spec, [p() for p in provs])
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
