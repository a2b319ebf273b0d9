"""Sensitivity-guided mixed-precision bit allocation (counterpart of the reference's methods/bit_assign.py).

Same entry points and flow as the reference (file:line are the reference's):
  gradtensor_to_vec             bit_assign.py:36-55
  run_hessian_vector_product    bit_assign.py:57-118   H·v of nn.MSELoss w.r.t. the decoder conv weights, summed over the
                                                       first 10 batches, by double backward (autograd.grad(create_graph=True)
                                                       then prod.backward()) -- here through ops.decoder_stack_dd, i.e.
                                                       the HIP convolution kernels composed as twice-differentiable
                                                       autograd Functions (conv is bilinear: F, dgrad, wgrad are closed
                                                       under differentiation)
  run_approx_param_fisher       bit_assign.py:120-168  accumulated dL/dW
  sensitivity_criterion         bit_assign.py:171-217  'omega' = sum_l <v_l,(Hv)_l>,  'fisher_diag' = sum_l <v_l^2, g_l^2>
  assign / main                 bit_assign.py:276-446  FP eval -> per candidate: QuantModel, set_bitwidth, lazy scale init,
                                                       score; the lowest score wins

    python -m neuroquant_amd.methods.bit_assign --arch hnerv --config cfg.yaml --data_path bunny/ --vid Bunny \
        --ckpt epoch300.pth --batch_size 2 --channel_wise --init max --mode omega
    (--synthetic N uses N synthetic frames; --candidates "2 3 4 6 4 4 2" "6 5 4 5 5 6 6" overrides the toy candidates)

Frames come from the GPU-resident uint8 cache instead of the PNG DataLoader; batches are drawn shuffle=True,
drop_last=False like the reference's full_dataloader (bit_assign.py:281-283).  Candidates are independent, so on a
multi-GPU node rank r scores candidates r, r+world, ... and the scores are all-gathered (replicas, no data-path
collective).
"""
import argparse
import copy
import logging
import os
import random
import sys
import time
from datetime import datetime

import numpy as np
import torch
import torch.nn as nn

from ..models import HNeRV, NeRV
from ..models._decode import _fused_stack
from ..quantization import QuantModel
from ..utils import FrameCache, RoundTensor, data_split, get_config, setup_logger
from .. import ops
from .calibrate_network import evaluate, load_frames

# toy example (bit_assign.py:26-35)
hnerv_candidate = {
    'candidate1': [2, 3, 4, 6, 4, 4, 2],  # 4.96 bit
    'candidate2': [6, 5, 4, 5, 5, 6, 6],  # 4.79 bit
}
nerv_candidate = {
    'candidate1': [5, 6, 3, 4, 5, 4, 3],  # 5.47 bit
    'candidate2': [6, 5, 5, 6, 7, 6, 7],  # 5.12 bit
}
MAX_BATCHES = 10  # bit_assign.py:116-118


class FullLoader:
    """The reference's `full_dataloader` (bit_assign.py:281-283: shuffle=True, drop_last=False) over a FrameCache."""

    def __init__(self, cache: FrameCache, batch_size, seed=903):
        self.cache, self.bs = cache, batch_size
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (len(self.cache) + self.bs - 1) // self.bs

    def __iter__(self):
        n = len(self.cache)
        perm = torch.randperm(n, generator=self.gen).to(self.cache.frames.device)
        for i in range(0, n, self.bs):
            idx = perm[i:i + self.bs]
            yield {'img': self.cache.batch(idx), 'idx': idx, 'norm_idx': idx.float() / n}


def gradtensor_to_vec(net):
    """Gradients of the decoder's weight tensors, in named_parameters order (bit_assign.py:36-55)."""
    return [v.grad.data for k, v in net.named_parameters() if 'encoder' not in k and 'weight' in k]


def _decode_dd(arch, net, sample):
    """Twice-differentiable forward of `net` on one batch.  The encoder output does not depend on the decoder weights,
    so it is computed without a graph (the reference builds one through the encoder and never uses it)."""
    spec_provs = _fused_stack(net)
    if spec_provs is None:
        raise NotImplementedError('bit allocation needs the shipped decoder shape (conv/PixelShuffle/GELU blocks, tanh head)')
    spec, provs = spec_provs
    img = sample['img']
    with torch.no_grad():
        emb = net.encode(img) if arch == 'hnerv' else net.encode(sample['norm_idx'])
    return ops.decoder_stack_dd(emb, spec, [p() for p in provs]), img


def run_hessian_vector_product(arch, vec, params, net, criterion, dataloader, use_cuda=True):
    """H·vec accumulated into the .grad of `params` (bit_assign.py:57-118)."""
    if not use_cuda:
        raise RuntimeError('neuroquant_amd has no CPU path (the CPU restatement lives in oracle/ for tests)')
    net.cuda()
    vec = [v.detach().cuda() for v in vec]
    net.eval()
    net.zero_grad()
    for count, sample in enumerate(dataloader):
        if count >= MAX_BATCHES:
            break
        img_out, img = _decode_dd(arch, net, sample)
        loss = criterion(img_out, img)
        grad_f = torch.autograd.grad(loss, inputs=params, create_graph=True, allow_unused=True)
        prod = sum((g * v).sum() for g, v in zip(grad_f, vec) if g is not None)
        prod.backward()


def run_approx_param_fisher(arch, vec, params, net, criterion, dataloader, use_cuda=True):
    """dL/dW accumulated into the .grad of `params` over the first 10 batches (bit_assign.py:120-168)."""
    if not use_cuda:
        raise RuntimeError('neuroquant_amd has no CPU path (the CPU restatement lives in oracle/ for tests)')
    net.cuda()
    net.eval()
    net.zero_grad()
    for count, sample in enumerate(dataloader):
        if count >= MAX_BATCHES:
            break
        img_out, img = _decode_dd(arch, net, sample)
        criterion(img_out, img).backward()


def sensitivity_criterion(mode, arch, net, qnn, dataloader, use_cuda=True):
    """Sensitivity score of one mixed-precision candidate (bit_assign.py:171-217): net = FP model, qnn = the
    QuantModel whose `get_perturbation()` gives v_l = W_l - Q(W_l)."""
    params = []
    for k, v in net.named_parameters():
        if 'encoder' in k:
            continue
        if 'weight' in k:
            v.requires_grad = True
            params.append(v)
    with torch.no_grad():
        vec = [v.detach() for v in qnn.get_perturbation()]
    criterion = nn.MSELoss()
    if mode == 'omega':
        run_hessian_vector_product(arch, vec, params, net, criterion, dataloader, use_cuda)
        total = 0.
        for count, (g, v) in enumerate(zip(gradtensor_to_vec(net), vec)):
            cur = (g * v).sum()
            total = total + cur
            logging.info(f"[{count:d}-th layer] {float(cur):.3e}")
        return total
    if mode == 'fisher_diag':
        run_approx_param_fisher(arch, vec, params, net, criterion, dataloader, use_cuda)
        total = 0.
        for count, (g, v) in enumerate(zip(gradtensor_to_vec(net), vec)):
            cur = (v.pow(2) * g.pow(2)).sum()
            total = total + cur
            logging.info(f"[{count:d}-th layer] {float(cur):.3e}")
        return total
    raise ValueError('Not implemented sensitivity criteria: {}'.format(mode))


def owned_candidates(names, rank=0, world=1):
    """Candidates scored by `rank`: round-robin in dict order (candidates are independent -> replicas, no collective
    on the data path)."""
    return [n for i, n in enumerate(names) if i % world == rank]


def gather_scores(scores, world):
    """Union of the per-rank {name: score} dicts on every rank (torch.distributed.all_gather_object)."""
    if world <= 1:
        return dict(scores)
    import torch.distributed as dist
    gathered = [None] * world
    dist.all_gather_object(gathered, scores)
    return {k: v for d in gathered for k, v in d.items()}


def pick_best(candidate_dict, scores):
    """Lowest score wins; ties go to the earlier candidate (bit_assign.py:361-365 uses a strict '<')."""
    names = list(candidate_dict)
    best = min(names, key=lambda c: (scores[c], names.index(c)))
    return best, candidate_dict[best], scores[best]


def score_candidates(model, candidate_dict, cali_data, cache, args, rank=0, world=1):
    """-> {name: score} for the candidates this rank owns."""
    device = next(model.parameters()).device
    scores = {}
    mine = set(owned_candidates(list(candidate_dict), rank, world))
    for candidate, bits in candidate_dict.items():
        if candidate not in mine:
            continue
        wq_params = {'n_bits': 8, 'channel_wise': args.channel_wise, 'scale_method': args.init}
        qnn = QuantModel(model=copy.deepcopy(model), hadamard=args.hadamard, weight_quant_params=wq_params).to(device)
        qnn.eval()
        avg_bits = float(qnn.set_bitwidth(bits))
        qnn.set_quant_state(True)
        with torch.no_grad():
            qnn(cali_data[:args.batch_size].to(device))      # lazy scale init (bit_assign.py:353-355)
        logging.info(f"[{candidate}: {bits}] Average Quantization Bit-Width:\t{avg_bits:.4f}")
        loader = FullLoader(cache, args.batch_size, seed=args.seed)   # same batches for every candidate
        score = float(sensitivity_criterion(args.mode, args.arch, copy.deepcopy(model), qnn, loader, use_cuda=True))
        logging.info(f"[{candidate}: {bits}] The {args.mode} sensitivity score =\t{score:.3e}")
        scores[candidate] = score
    return scores


def assign(args, cfg):
    if not torch.cuda.is_available():
        raise RuntimeError('neuroquant_amd needs an AMD GPU (no CPU path); the reference CPU path lives in oracle/ for tests')
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    device = 'cuda'
    frames = load_frames(args, cfg, device)
    cache = FrameCache(frames)
    n = len(cache)
    split = [int(x) for x in args.data_split.split('_')]
    _, args.val_ind_list = data_split(list(range(n)), split, False, 0)

    model = (HNeRV if args.arch == 'hnerv' else NeRV)(cfg).to(device)
    os.makedirs(args.outf, exist_ok=True)
    setup_logger(os.path.join(args.outf, time.strftime('%Y%m%d_%H%M%S') + f'_r{rank}.log'))
    if args.ckpt != 'None':
        logging.info("=> loading checkpoint '{}'".format(args.ckpt))
        model.load_state_dict(torch.load(args.ckpt, map_location='cpu'), strict=False)
    else:
        logging.info('no --ckpt: random-initialised weights (throughput runs only)')
    model.to(device)

    logging.info('=======================Full-precision model========================')
    res, embedding_list = evaluate(model, cache, args, cfg)
    logging.info(f'FP: best_pred_seen_psnr: {RoundTensor(res[0], 2)} | best_pred_unseen_psnr: {RoundTensor(res[1], 2)}')
    cali_data = torch.cat(embedding_list, dim=0)

    if args.candidates:
        candidate_dict = {f'candidate{i + 1}': [int(b) for b in c.split()] for i, c in enumerate(args.candidates)}
    else:
        candidate_dict = hnerv_candidate if args.arch == 'hnerv' else nerv_candidate
    scores = gather_scores(score_candidates(model, candidate_dict, cali_data, cache, args, rank, world), world)
    best_candidate, best_bits, best_score = pick_best(candidate_dict, scores)
    logging.info("=" * 60)
    logging.info(f"Best Candidate: {best_candidate}")
    logging.info(f"Bit Configuration: {best_bits}")
    logging.info(f"Minimum Score: {best_score:.4e}")
    logging.info("=" * 60)
    return best_candidate, best_bits, best_score


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
    p.add_argument('--hadamard', action='store_true')
    p.add_argument('--channel_wise', action='store_true')
    p.add_argument('--init', default='max', type=str, choices=['max', 'mse', 'gaussian', 'l1', 'l2'])
    p.add_argument('--mode', default='omega', type=str, choices=['omega', 'fisher_diag'])
    p.add_argument('--candidates', type=str, nargs='*', default=None,
                   help='bit lists, one quoted string per candidate; default: the reference\'s toy candidates')
    p.add_argument('--ckpt', default='None', type=str)
    return p.parse_args(argv)


def seed_all(seed=903):
    random.seed(seed)
    np.random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def main(argv):
    args = parse_args(argv)
    seed_all(args.seed)
    cfg = get_config(args.config)
    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    exp_id = f"{args.vid}_e{cfg.get('epoch')}_b{cfg.get('batch_size')}_lr{cfg.get('learning_rate')}_{cfg.get('loss')}"
    args.outf = os.path.join('results', args.outf, exp_id,
                             "sensitivity-{}_{}-init_batch{}_CW".format(args.mode, args.init, args.batch_size))
    return assign(args, cfg)


if __name__ == '__main__':
    main(sys.argv[1:])
