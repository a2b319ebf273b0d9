"""Network-wise calibration driver (counterpart of the reference's methods/calibrate_network.py).

Same flags, same sequence (FP eval -> QuantModel -> set_bitwidth -> lazy scale init -> eval quant off / on ->
model_reconstruction -> final eval -> torch.save), with the reference's PNG DataLoader replaced by a GPU-resident
uint8 frame cache (frames are decoded ONCE; the reference re-decodes every frame every epoch in 4 worker processes,
which alone caps it at ~55 it/s) and PSNR computed by the HIP reduction kernel.  MS-SSIM is not
computed (third-party pytorch_msssim, out of scope).

    python -m neuroquant_amd.methods.calibrate_network --arch hnerv --config cfg.yaml --data_path bunny/ --vid Bunny \
        --ckpt epoch300.pth --batch_size 2 --channel_wise --init max --iters_w 21000 --weight 0.01 --b_start 20 \
        --b_end 2 --warmup 0.2 --lr 0.003 --precision 6 5 4 5 5 6 6
    (--synthetic N uses N synthetic Bunny-shaped frames instead of --data_path)
"""
import argparse
import logging
import os
import random
import sys
import time
from datetime import datetime

import numpy as np
import torch

from ..models import HNeRV, NeRV
from ..quantization import QuantModel, model_reconstruction
from ..utils import (CacheLoader, FrameCache, RoundTensor, data_split, get_config, setup_logger, synthetic_frames)
from .. import ops


def parse_args(argv):
    p = argparse.ArgumentParser(description='running parameters', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--seed', default=903, type=int)
    p.add_argument('--outf', default='unify')
    p.add_argument('--config', type=str)
    p.add_argument('--arch', type=str)
    p.add_argument('-p', '--print-freq', default=50, type=int)
    p.add_argument('--data_path', type=str)
    p.add_argument('--synthetic', type=int, default=0, help='use N synthetic frames instead of --data_path')
    p.add_argument('--vid', type=str, default='Bunny')
    p.add_argument('--data_split', type=str, default='1_1_1')
    p.add_argument('--batch_size', default=12, type=int)
    p.add_argument('--precision', type=int, nargs='+', default=[8] * 7)
    p.add_argument('--channel_wise', action='store_true')
    p.add_argument('--hadamard', action='store_true')
    p.add_argument('--iters_w', default=20000, type=int)
    p.add_argument('--weight', default=0.01, type=float)
    p.add_argument('--b_start', default=20, type=int)
    p.add_argument('--b_end', default=2, type=int)
    p.add_argument('--warmup', default=0.2, type=float)
    p.add_argument('--input_prob', default=1.0, type=float)
    p.add_argument('--lr', default=0.0015, type=float)
    p.add_argument('--norm_p', default=2.0, type=float)
    p.add_argument('--init', default='max', type=str, choices=['max', 'mse', 'gaussian', 'l1', 'l2'])
    p.add_argument('--opt_mode', default='mse', type=str, choices=['mse', 'fisher_diag', 'fisher_full', 'lp_norm'])
    p.add_argument('--ckpt', default='None', type=str)
    p.add_argument('--dump_vis', action='store_true', default=False)
    return p.parse_args(argv)


def seed_all(seed=903):
    random.seed(seed)
    np.random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def dist_setup():
    """Data-parallel launch: under `python -m torch.distributed.run --nproc-per-node N ... -m
    neuroquant_amd.methods.calibrate_network ...` every rank binds its GPU and joins the RCCL group BEFORE any other GPU call.
    -> (rank, world, device).  Single process: (0, 1, 'cuda').  NQ_DP_REHEARSAL=1 runs the collective path on a 1-rank
    group (one GPU)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if 'RANK' not in os.environ or (world == 1 and not os.environ.get('NQ_DP_REHEARSAL')):
        return 0, 1, 'cuda'
    import torch.distributed as dist
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    # the HSA runtime reads its environment when it initialises, i.e. at the first GPU call below: the variable that makes
    # RCCL's cross-process buffer sharing use dmabuf (the only IPC mode this host driver supports) must be set BEFORE it
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    return rank, world, f'cuda:{local}'


def load_frames(args, cfg, device):
    """-> uint8 (N,3,crop_h,crop_w) on `device`: sorted PNGs, center crop (reference videosets/datasets.py:8-30)."""
    h, w = cfg['crop_h'], cfg['crop_w']
    if args.synthetic:
        return synthetic_frames(args.synthetic, h, w, seed=args.seed, device=device)
    from PIL import Image
    files = [os.path.join(args.data_path, x) for x in sorted(os.listdir(args.data_path))]
    out = torch.empty((len(files), 3, h, w), dtype=torch.uint8)
    for i, f in enumerate(files):
        a = torch.from_numpy(np.asarray(Image.open(f).convert('RGB')).copy()).permute(2, 0, 1)
        top, left = int(round((a.shape[1] - h) / 2.0)), int(round((a.shape[2] - w) / 2.0))
        out[i] = a[:, top:top + h, left:left + w]
    return out.to(device)


@torch.no_grad()
def evaluate(model, cache: FrameCache, args, cfg):
    """Per-frame decode + PSNR (reference calibrate_network.py:82-145); returns ([seen_psnr, unseen_psnr], embeddings)."""
    model.eval()
    n = len(cache)
    dev = cache.frames.device
    psnr, embeds, dec_times = [], [], []
    for i in range(n):
        idx = torch.tensor([i], device=dev)
        img = cache.batch(idx)
        emb = model.encode(img) if args.arch == 'hnerv' else model.encode(idx.float() / n)
        out, embed_list, dec_time = model.decode(emb)
        embeds.append(embed_list[0])
        dec_times.append(dec_time)
        psnr.append(ops.frame_psnr(out, img))
        if i % args.print_freq == 0 or i == n - 1:
            logging.info('[{}], Eval at Step [{}/{}], FPS {}, PSNR {}'.format(
                datetime.now().strftime("%Y/%m/%d %H:%M:%S"), i + 1, n,
                round(cfg.get('batch_size', 1) / (sum(dec_times) / len(dec_times)), 1),
                RoundTensor(torch.cat(psnr).mean().cpu(), 2)))
    psnr = torch.cat(psnr).cpu()
    seen = [i for i in range(n) if i not in args.val_ind_list]
    res = [psnr[seen].mean() if seen else torch.zeros(()),
           psnr[args.val_ind_list].mean() if args.val_ind_list else torch.zeros(())]
    model.train()
    return res, embeds


def calibrate(args, cfg):
    if not torch.cuda.is_available():
        raise RuntimeError('neuroquant_amd needs an AMD GPU (no CPU path); the reference CPU path lives in oracle/ for tests')
    # one process per GPU: --batch_size is the GLOBAL batch, each rank takes batch_size/world frames of every batch; all
    # ranks hold all frames / embeddings / parameters, evaluate identically (replicas are bit-identical), rank 0 logs + saves
    rank, world, device = dist_setup()
    if args.batch_size % world:
        raise ValueError(f'--batch_size {args.batch_size} must divide over {world} ranks')
    frames = load_frames(args, cfg, device)
    cache = FrameCache(frames)
    n = len(cache)
    split = [int(x) for x in args.data_split.split('_')]
    train_ind, args.val_ind_list = data_split(list(range(n)), split, False, 0)
    train_loader = CacheLoader(cache, train_ind, args.batch_size, seed=args.seed, rank=rank, world=world)

    model = (HNeRV if args.arch == 'hnerv' else NeRV)(cfg).to(device)
    dec_param = sum(p.numel() for p in model.decoder.parameters()) / 1e6
    if rank == 0:
        os.makedirs(args.outf, exist_ok=True)
        setup_logger(os.path.join(args.outf, time.strftime('%Y%m%d_%H%M%S') + '.log'))
    else:
        logging.getLogger().setLevel(logging.WARNING)
    logging.info(f'Decoder_{round(dec_param, 2)}M' + (f' | data-parallel over {world} GPUs, {args.batch_size // world} frames per GPU' if world > 1 else ''))
    if args.ckpt != 'None':
        logging.info("=> loading checkpoint '{}'".format(args.ckpt))
        model.load_state_dict(torch.load(args.ckpt, map_location='cpu'), strict=False)
    else:
        logging.info('no --ckpt: random-initialised weights (throughput runs only)')
    model.to(device)

    def report(tag, res):
        logging.info(f'{tag}: best_pred_seen_psnr: {RoundTensor(res[0], 2)} | best_pred_unseen_psnr: {RoundTensor(res[1], 2)}')

    logging.info('=======================Full-precision model========================')
    res, embedding_list = evaluate(model, cache, args, cfg)
    report('FP', res)

    wq_params = {'n_bits': 8, 'channel_wise': args.channel_wise, 'scale_method': args.init}
    qnn = QuantModel(model=model, hadamard=args.hadamard, weight_quant_params=wq_params).to(device)
    args.qbits = qnn.set_bitwidth(args.precision)
    qnn.eval()
    cali_data = torch.cat(embedding_list, dim=0)
    logging.info('input embedding shape: {}'.format(cali_data.shape))

    qnn.set_quant_state(True)
    t0 = time.time()
    with torch.no_grad():
        qnn(cali_data[:args.batch_size])
    torch.cuda.synchronize()
    logging.info('Init time: {}'.format(time.time() - t0))

    qnn.set_quant_state(False)
    report('Close quantization', evaluate(qnn, cache, args, cfg)[0])
    qnn.set_quant_state(True)
    report('Weight quantization w/o opt', evaluate(qnn, cache, args, cfg)[0])

    logging.info('average bit-width: {}'.format(args.qbits))
    start = datetime.now()
    qnn.set_quant_state(weight_quant=True)
    model_reconstruction(qnn, cali_data=cali_data, gt=train_loader, arch=args.arch, batch_size=args.batch_size,
                         iters=args.iters_w, weight=args.weight, opt_mode='mse', hadamard=args.hadamard,
                         b_range=(args.b_start, args.b_end), warmup=args.warmup, p=args.norm_p, lr=args.lr)
    torch.cuda.synchronize()
    logging.info(f"Training complete in: {str(datetime.now() - start)}")

    qnn.set_quant_state(weight_quant=True)
    res = evaluate(qnn, cache, args, cfg)[0]
    report('Weight quantization w/ opt', res)
    tag = 'CW' if args.channel_wise else 'LW'
    if rank == 0:
        torch.save(qnn, "{}/{}_W{}_prob{}_{}-init_{}.pth".format(args.outf, args.arch, args.qbits, args.input_prob, args.init, tag))
    args.qnn = qnn
    return res


def main(argv):
    args = parse_args(argv)
    cfg = get_config(args.config)
    seed_all(args.seed)   # NB the reference defines seed_all but never calls it (calibrate_network.py:68-78, 311-324)
    exp_id = f"{args.vid}_e{cfg.get('epoch')}_b{cfg.get('batch_size')}_lr{cfg.get('learning_rate')}_{cfg.get('loss')}"
    args.outf = os.path.join('results', args.outf, exp_id,
                             "network-wise_calib/hadamard-{}_{}-init_batch{}_CW_weight{}_brange{}-{}_warmup{}_lr{}".format(
                                 args.hadamard, args.init, args.batch_size, args.weight, args.b_start, args.b_end,
                                 args.warmup, args.lr))
    try:
        return calibrate(args, cfg)
    finally:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1:])
