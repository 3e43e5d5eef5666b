"""Evaluation driver with the reference's CLI surface (tools/test.py:20-60: --cfg_file --batch_size --ckpt --launcher
--set ...): builds the detector through the registry, loads a checkpoint and runs eval_one_epoch over the test split of a
directory laid out like ONCE (pcdet.datasets.build_dataloader), or with `--synthetic` over deterministic ONCE-shape frame
pairs with synthetic labels."""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pcdet.config import cfg, cfg_from_list, cfg_from_yaml_file, log_config_to_file  # noqa: E402
from pcdet.datasets import build_dataloader  # noqa: E402
from pcdet.models import build_network  # noqa: E402
from pcdet.utils import common_utils  # noqa: E402
from tmae_amd.eval import eval_one_epoch  # noqa: E402
from tmae_amd.train import SyntheticEvalLoader, SyntheticTemporalDataset  # noqa: E402


def main():
    p = argparse.ArgumentParser(description='T-MAE evaluation on MI355X')
    p.add_argument('--cfg_file', type=str, required=True)
    p.add_argument('--batch_size', type=int, default=None)
    p.add_argument('--extra_tag', type=str, default='default')
    p.add_argument('--ckpt', type=str, default=None)
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    p.add_argument('--amp', action='store_true')
    p.add_argument('--set', dest='set_cfgs', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--synthetic_points', type=int, default=120000)
    p.add_argument('--synthetic_samples', type=int, default=32)
    p.add_argument('--output_dir', type=str, default=None)
    p.add_argument('--data_path', type=str, default=None, help='ONCE root (default: DATA_CONFIG.DATA_PATH)')
    # the rest of the reference's command line (tools/test.py:27-86, tools/scripts/once_test.sh): accepted as they are
    p.add_argument('--workers', type=int, default=4)
    p.add_argument('--tcp_port', type=int, default=18888)
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--max_waiting_mins', type=int, default=1)
    p.add_argument('--start_epoch', type=int, default=0)
    p.add_argument('--eval_tag', type=str, default='default')
    p.add_argument('--eval_all', action='store_true', help='evaluate every checkpoint of --ckpt_dir')
    p.add_argument('--ckpt_dir', type=str, default=None)
    p.add_argument('--save_to_file', action='store_true')
    p.add_argument('--fuse_conv_bn', action='store_true')
    p.add_argument('--fixed_gap_eval', type=int, default=-1)
    p.add_argument('--random_init', action='store_true',
                   help='evaluate the randomly initialised model (smoke runs only: the AP figures mean nothing)')
    args = p.parse_args()
    cfg_from_yaml_file(args.cfg_file, cfg)
    cfg.TAG = Path(args.cfg_file).stem
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs, cfg)
    if args.launcher == 'pytorch':
        world, rank = common_utils.init_dist_pytorch(backend='nccl')
    else:
        world, rank = 1, 0
        torch.cuda.set_device(0)
    bs = args.batch_size or cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU
    out = Path(args.output_dir or (Path(cfg.ROOT_DIR) / 'output' / cfg.TAG / args.extra_tag / 'eval'))
    out.mkdir(parents=True, exist_ok=True)
    logger = common_utils.create_logger(out / f'log_eval_{time.strftime("%Y%m%d-%H%M%S")}.txt', rank=rank)
    log_config_to_file(cfg, logger=logger)
    if args.eval_all:                                    # repeat_eval_ckpt (tools/test.py:101-170) without the waiting loop
        ckpt_dir = Path(args.ckpt_dir or out.parent / 'ckpt')

        def epoch_of(p_):
            try:
                return int(p_.stem.rsplit('_', 1)[1])
            except (IndexError, ValueError):
                return -1
        # ascending epoch number; files of one epoch number cannot coexist, so this is also the reference's mtime order
        # for the checkpoints of ONE run (tools/test.py:101-170)
        ckpts = [str(p_) for p_ in sorted(ckpt_dir.glob('checkpoint_epoch_*.pth'), key=epoch_of)
                 if epoch_of(p_) >= args.start_epoch]
        if not ckpts:
            raise FileNotFoundError(f'--eval_all: no checkpoint_epoch_*.pth (epoch >= {args.start_epoch}) in {ckpt_dir}')
    elif args.ckpt:
        if not Path(args.ckpt).is_file():
            raise FileNotFoundError(f'--ckpt {args.ckpt} does not exist')
        ckpts = [args.ckpt]
    elif args.random_init:
        logger.warning('evaluating RANDOMLY INITIALISED weights (--random_init): the figures below are not results')
        ckpts = [None]
    else:                                                # the reference requires a checkpoint (tools/test.py:53-58,195)
        raise ValueError('give --ckpt <file>, or --eval_all with --ckpt_dir, or --random_init for a smoke run')
    real_loader = None
    if args.synthetic:
        ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=args.synthetic_points, batch_size=bs, rank=rank)
    else:
        if args.fixed_gap_eval >= 0:
            cfg.DATA_CONFIG.FIXED_GAP = args.fixed_gap_eval
        ds, real_loader, _ = build_dataloader(cfg.DATA_CONFIG, cfg.CLASS_NAMES, bs, dist=world > 1, root_path=args.data_path,
                                              workers=args.workers, logger=logger, training=False)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds, logger).cuda()
    cfg.LOCAL_RANK = int(os.environ.get('LOCAL_RANK', 0))
    for ck in ckpts:
        if ck:
            model.load_params_from_file(ck, logger=logger)
        loader = real_loader if real_loader is not None else SyntheticEvalLoader(ds, args.synthetic_samples, bs, rank=rank, world=world)
        tag = Path(ck).stem if ck else 'random_init'
        ret = eval_one_epoch(cfg, model, loader, tag, logger, dist_test=world > 1, result_dir=out / args.eval_tag / tag,
                             save_to_file=args.save_to_file, amp_dtype=torch.bfloat16 if args.amp else None)
        if rank == 0:
            logger.info({k: float(v) for k, v in ret.items()})
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
