"""FP32 INR trainer (counterpart of the reference's methods/regress.py:151-322; SURVEY §8f-4, a "next" row).

Same flags and schedule (Adam, weight_decay 0, `--lr_type cosine_0.1_1_0.1` / hybrid, L2 loss, per-epoch
`model_latest.pth`, final `epoch{N}.pth` state_dicts that `calibrate_network` loads).  The decoder runs as the fused
HIP stack (`ops.decoder_stack`: implicit-GEMM convs, fused PixelShuffle/GELU/tanh, data/weight gradients), frames come
from the GPU-resident cache; the ConvNeXt encoder and Adam stay in PyTorch.  MS-SSIM / tensorboard are not produced.

    python -m neuroquant_amd.methods.regress --arch hnerv --config cfg.yaml --data_path bunny/ --vid Bunny
"""
import argparse
import logging
import math
import os
import sys
import time
from datetime import datetime

import torch

from ..models import HNeRV, NeRV
from ..utils import CacheLoader, FrameCache, RoundTensor, data_split, get_config, setup_logger
from .. import ops
from .calibrate_network import evaluate, load_frames, seed_all


def parse_args(argv):
    p = argparse.ArgumentParser(description='running parameters', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--seed', default=903, type=int)
    p.add_argument('--outf', default='unify')
    p.add_argument('--config', type=str)
    p.add_argument('--arch', type=str)
    p.add_argument('--data_path', type=str)
    p.add_argument('--synthetic', type=int, default=0)
    p.add_argument('--vid', type=str, default='Bunny')
    p.add_argument('--data_split', type=str, default='1_1_1')
    p.add_argument('-p', '--print-freq', default=50, type=int)
    p.add_argument('--lr_type', type=str, default='cosine_0.1_1_0.1')
    p.add_argument('--weight', default='None', type=str)
    p.add_argument('--eval_only', action='store_true', default=False)
    return p.parse_args(argv)


def adjust_lr(optimizer, cur_epoch, args, eta_min=0.05):
    """reference utils.py:77-97."""
    if 'hybrid' in args.lr_type:
        up_ratio, up_pow, down_pow, min_lr, final_lr = [float(x) for x in args.lr_type.split('_')[1:]]
        if cur_epoch < up_ratio:
            lr_mult = min_lr + (1. - min_lr) * (cur_epoch / up_ratio) ** up_pow
        else:
            lr_mult = 1 - (1 - final_lr) * ((cur_epoch - up_ratio) / (1. - up_ratio)) ** down_pow
    elif 'cosine' in args.lr_type:
        up_ratio, up_pow, min_lr = [float(x) for x in args.lr_type.split('_')[1:]]
        if cur_epoch < up_ratio:
            lr_mult = min_lr + (1. - min_lr) * (cur_epoch / up_ratio) ** up_pow
        else:
            lr_mult = max(0.5 * (math.cos(math.pi * (cur_epoch - up_ratio) / (1 - up_ratio)) + 1.0), eta_min)
    else:
        raise NotImplementedError
    for g in optimizer.param_groups:
        g['lr'] = args.lr * lr_mult
    return args.lr * lr_mult


def train(args, cfg):
    if not torch.cuda.is_available():
        raise RuntimeError('neuroquant_amd needs an AMD GPU')
    if cfg.get('loss', 'l2') != 'l2':
        raise NotImplementedError('only the l2 loss of the shipped configs is built')
    device = 'cuda'
    cache = FrameCache(load_frames(args, cfg, device))
    n = len(cache)
    train_ind, args.val_ind_list = data_split(list(range(n)), [int(x) for x in args.data_split.split('_')], False, 0)
    loader = CacheLoader(cache, train_ind, cfg['batch_size'], seed=args.seed)
    model = (HNeRV if args.arch == 'hnerv' else NeRV)(cfg).to(device)
    os.makedirs(args.outf, exist_ok=True)
    setup_logger(os.path.join(args.outf, time.strftime('%Y%m%d_%H%M%S') + '.log'))
    if args.weight != 'None':
        model.load_state_dict(torch.load(args.weight, map_location='cpu'), strict=False)
    if args.eval_only:
        res, _ = evaluate(model, cache, args, cfg)
        logging.info(f'best_pred_seen_psnr: {RoundTensor(res[0], 2)}')
        return res
    optimizer = torch.optim.Adam(model.parameters(), weight_decay=0.)
    args.lr = cfg['learning_rate']
    start = datetime.now()
    for epoch in range(cfg['epoch']):
        model.train()
        psnrs = []
        for i, sample in enumerate(loader):
            lr = adjust_lr(optimizer, (epoch + float(i) / len(loader)) / cfg['epoch'], args)
            img = sample['img']
            img_out, _, _ = model(img if args.arch == 'hnerv' else sample['norm_idx'])
            loss = ops.l2_loss(img_out, img) / img.shape[1]     # F.mse_loss(...).flatten(1).mean(1).mean()
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            optimizer.step()
            psnrs.append(ops.frame_psnr(img_out, img))
            if i % args.print_freq == 0 or i == len(loader) - 1:
                logging.info('[{}], Epoch[{}/{}], Step [{}/{}], lr:{:.2e} pred_PSNR: {}'.format(
                    datetime.now().strftime("%Y/%m/%d %H:%M:%S"), epoch + 1, cfg['epoch'], i + 1, len(loader), lr,
                    RoundTensor(torch.cat(psnrs).mean().cpu(), 2)))
        if (epoch + 1) % cfg.get('eval_freq', 30) == 0 or (cfg['epoch'] - epoch) in [1, 3, 5]:
            res, _ = evaluate(model, cache, args, cfg)
            logging.info(f'Eval at epoch {epoch + 1}: pred_seen_psnr: {RoundTensor(res[0], 2)}')
        torch.save(model.state_dict(), '{}/model_latest.pth'.format(args.outf))
        if (epoch + 1) % cfg['epoch'] == 0:
            torch.save(model.state_dict(), f'{args.outf}/epoch{epoch + 1}.pth')
    logging.info(f"Training complete in: {str(datetime.now() - start)}")
    return model


def main(argv):
    args = parse_args(argv)
    cfg = get_config(args.config)
    seed_all(args.seed)
    exp_id = f"{args.vid}_e{cfg['epoch']}_b{cfg['batch_size']}_lr{cfg['learning_rate']}_{cfg['loss']}"
    args.outf = os.path.join('results', args.outf, exp_id)
    return train(args, cfg)


if __name__ == '__main__':
    main(sys.argv[1:])
