"""Training driver with the reference's CLI surface (tools/train.py:35-149: --cfg_file --batch_size --epochs
--launcher --amp --set --ckpt --extra_tag ...) for MODEL.NAME == TMAE (pre-training) and CenterPoint (fine-tuning).
One process per GPU; `--launcher pytorch` reads the torchrun environment and uses RCCL ('nccl') for the gradient
all-reduce.  Data: a directory laid out like ONCE under DATA_CONFIG.DATA_PATH (or --data_path) through
pcdet.datasets.build_dataloader (tmae_amd.data: .bin reader threads + the on-device two-frame pipeline), or
`--synthetic`: deterministic ONCE-shape frame pairs generated on the fly."""
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
from pcdet.models import build_network, model_fn_decorator  # noqa: E402
from pcdet.utils import common_utils  # noqa: E402
from tmae_amd.train import (SyntheticTemporalDataset, build_optimizer, build_scheduler, train_one_step,  # noqa: E402
                            wrap_ddp)


def parse_config():
    p = argparse.ArgumentParser(description='T-MAE pre-training on MI355X')
    p.add_argument('--cfg_file', type=str, required=True)
    p.add_argument('--batch_size', type=int, default=None, help='total batch size (split over GPUs, train.py:163-169)')
    p.add_argument('--epochs', type=int, default=None)
    p.add_argument('--workers', type=int, default=4)
    p.add_argument('--extra_tag', type=str, default='default')
    p.add_argument('--ckpt', type=str, default=None)
    p.add_argument('--pretrained_model', type=str, default=None)
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--amp', action='store_true', help='bf16 autocast (the reference: fp16 + GradScaler)')
    p.add_argument('--fix_random_seed', action='store_true')
    p.add_argument('--ckpt_save_interval', type=int, default=1)
    p.add_argument('--max_ckpt_save_num', type=int, default=30)
    p.add_argument('--set', dest='set_cfgs', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--synthetic_points', type=int, default=120000)
    p.add_argument('--iters_per_epoch', type=int, default=100)
    p.add_argument('--output_dir', type=str, default=None)
    p.add_argument('--data_path', type=str, default=None, help='ONCE root (default: DATA_CONFIG.DATA_PATH)')
    p.add_argument('--reference_rng_order', action='store_true',
                   help='np.random stream of a single-process reference run (one host sync per sample, --workers 0)')
    # the rest of the reference's command line (tools/train.py:37-131, tools/scripts/once_train.sh): accepted so that its
    # launch lines run unchanged; what has no meaning here says so when it is used
    p.add_argument('--tcp_port', type=int, default=18888, help='unused: the rendezvous comes from the launcher environment')
    p.add_argument('--sync_bn', action='store_true', help='batch statistics over all ranks (torch.nn.SyncBatchNorm semantics)')
    p.add_argument('--merge_all_iters_to_one_epoch', action='store_true')
    p.add_argument('--max_waiting_mins', type=int, default=1)
    p.add_argument('--start_epoch', type=int, default=0)
    p.add_argument('--num_epochs_to_eval', type=int, default=0, help='evaluate the last N checkpoints after training (fine-tune configs)')
    p.add_argument('--save_to_file', action='store_true')
    p.add_argument('--fuse_conv_bn', action='store_true')
    p.add_argument('--wandb', action='store_true', help='ignored (no network)')
    p.add_argument('--wandb_proj_name', type=str, default='t-mae-0.05')
    p.add_argument('--fixed_gap_eval', type=int, default=None)
    args = p.parse_args()
    if args.merge_all_iters_to_one_epoch:
        raise NotImplementedError('--merge_all_iters_to_one_epoch is not used by the T-MAE recipes')
    cfg_from_yaml_file(args.cfg_file, cfg)
    cfg.TAG = Path(args.cfg_file).stem
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs, cfg)
    return args, cfg


def save_checkpoint(model, optimizer, epoch, it, path):
    """{'epoch','it','model_state','optimizer_state','scaler','version'} (train_utils.py:245-270); `epoch` = the number
    of epochs trained so far, as the reference stores it (train_utils.py:217-232)."""
    m = model.module if hasattr(model, 'module') else model
    state = {k: v.cpu() for k, v in m.state_dict().items()}
    tmp = Path(str(path) + '.tmp')                       # readers never see a partly written file
    torch.save({'epoch': epoch, 'it': it, 'model_state': state, 'optimizer_state': optimizer.state_dict(),
                'scaler': None, 'version': 'tmae_amd'}, tmp)
    os.replace(tmp, path)


def _ckpt_epoch(p):
    try:
        return int(p.stem.rsplit('_', 1)[1])
    except (IndexError, ValueError):
        return -1


def list_checkpoints(ckpt_dir):
    """checkpoint_epoch_<N>.pth of `ckpt_dir`, ascending in N (epoch numbers, not mtimes: every rank and every file
    system orders them the same way).  Used for evaluation lists; pruning goes by age (prune_checkpoints)."""
    return sorted((p for p in Path(ckpt_dir).glob('checkpoint_epoch_*.pth') if _ckpt_epoch(p) >= 0), key=_ckpt_epoch)


def prune_checkpoints(ckpt_dir, max_ckpt_save_num):
    """Make room BEFORE the next save, oldest files (mtime) first, so that at most max_ckpt_save_num remain afterwards --
    the reference's rule (train_utils.py:219-228).  The file about to be written is never a candidate, whatever stale
    higher-epoch files an earlier run left in the directory."""
    files = sorted(Path(ckpt_dir).glob('checkpoint_epoch_*.pth'), key=os.path.getmtime)
    if len(files) >= max_ckpt_save_num:
        for old in files[:len(files) - max_ckpt_save_num + 1]:
            old.unlink()


def main():
    args, cfg = parse_config()
    pin = None
    if args.launcher == 'pytorch':
        # this rank's host cores (a disjoint slice per rank of the node, next to its GPU where sysfs shows that): set before the
        # first GPU call, so that the HIP runtime's threads and the .bin reader threads inherit it (tmae_amd/train/affinity.py)
        from tmae_amd.train.affinity import format_cpulist, pin_rank
        pin = pin_rank(int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', 1))))
        world, rank = common_utils.init_dist_pytorch(backend='nccl')
        local_rank = int(os.environ.get('LOCAL_RANK', 0))
    else:
        world, rank, local_rank = 1, 0, 0
        torch.cuda.set_device(0)
    bs = cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU if args.batch_size is None else args.batch_size // world
    epochs = cfg.OPTIMIZATION.NUM_EPOCHS if args.epochs is None else args.epochs
    if args.fix_random_seed:
        common_utils.set_random_seed(666 + rank)
    out = Path(args.output_dir or (Path(cfg.ROOT_DIR) / 'output' / cfg.TAG / args.extra_tag))
    (out / 'ckpt').mkdir(parents=True, exist_ok=True)
    logger = common_utils.create_logger(out / f'log_train_{time.strftime("%Y%m%d-%H%M%S")}.txt', rank=rank)
    log_config_to_file(cfg, logger=logger)
    if pin is not None:
        logger.info(f'rank {rank}: host cores {format_cpulist(pin["cores"])} ({pin["source"]}, {pin["allowed"]} allowed)')
    train_loader = train_sampler = None
    if args.synthetic:
        ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=args.synthetic_points, batch_size=bs, rank=rank,
                                      n_boxes=40 if cfg.MODEL.get('DENSE_HEAD', None) is not None else 0)   # labels for the fine-tune config
        iters_per_epoch = args.iters_per_epoch
    else:
        ds, train_loader, train_sampler = build_dataloader(
            cfg.DATA_CONFIG, cfg.CLASS_NAMES, bs, dist=world > 1, root_path=args.data_path, workers=args.workers, logger=logger,
            training=True, total_epochs=epochs, device=torch.device('cuda', torch.cuda.current_device()))
        train_loader.pipeline.reference_rng_order = bool(args.reference_rng_order)
        iters_per_epoch = len(train_loader)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds, logger)
    if args.sync_bn:
        # the reference's conversion (tools/train.py:244-245); the fused BatchNorm kernels read the module type and merge
        # their statistics and backward sums over the ranks of the process group (tmae_amd.ops._BatchNormReLU)
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model = model.cuda()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    start_epoch = it = 0
    if args.pretrained_model:
        model.load_params_from_file(args.pretrained_model, logger=logger)
    ckpt_dir = out / 'ckpt'
    resume = args.ckpt
    if resume is None:
        # resume from the newest checkpoint of the output directory, as the reference does (tools/train.py:268-280)
        found = sorted(ckpt_dir.glob('*checkpoint_epoch_*.pth'), key=os.path.getmtime)
        if found:
            resume = str(found[-1])
            logger.info(f'resuming from {resume} (newest checkpoint in {ckpt_dir})')
    if resume:
        # the stored 'epoch' counts trained epochs (the reference saves cur_epoch + 1): resume AT it
        it, start_epoch = model.load_params_with_optimizer(resume, optimizer=opt, logger=logger)
        start_epoch = max(int(start_epoch), 0)
    stale = [p_.name for p_ in list_checkpoints(ckpt_dir) if _ckpt_epoch(p_) > start_epoch]
    if stale and rank == 0:
        logger.warning(f'{ckpt_dir} holds checkpoints of later epochs than the starting epoch {start_epoch}: {stale}; '
                       f'they count towards --max_ckpt_save_num and are pruned by age like any other')
    model.train()
    ddp = wrap_ddp(model, local_rank)
    sched, _ = build_scheduler(opt, iters_per_epoch, epochs, -1, cfg.OPTIMIZATION)
    model_func = model_fn_decorator()
    amp = torch.bfloat16 if args.amp else None
    for epoch in range(start_epoch, epochs):
        t0 = time.time()
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
            ds.set_epoch(epoch)
        batches = (ds.batch(epoch * iters_per_epoch + i) for i in range(iters_per_epoch)) if train_loader is None else train_loader
        # one batch of lookahead: the step is told which batch comes next and voxelises it between its forward and backward
        # (TemporalDynVFE.prefetch), so that no step begins by waiting for its voxel counts
        feed = iter(batches)
        batch = next(feed, None)
        i = -1
        while batch is not None:
            i += 1
            nxt = next(feed, None)
            loss, tb, _ = train_one_step(ddp, opt, sched, batch, it, model_func, amp_dtype=amp,
                                         grad_norm_clip=cfg.OPTIMIZATION.get('GRAD_NORM_CLIP', None),
                                         next_batch=nxt if isinstance(nxt, dict) and torch.is_tensor(nxt.get('points', None)) else None)
            batch = nxt
            it += 1
            if rank == 0 and (i % 10 == 0 or i == iters_per_epoch - 1):
                logger.info(f'epoch {epoch} it {i}/{iters_per_epoch} loss {float(loss):.5f} lr {opt.lr:.2e}')
        if rank == 0:
            logger.info(f'epoch {epoch} done in {time.time() - t0:.1f} s '
                        f'({bs * world * iters_per_epoch / (time.time() - t0):.1f} frame-pairs/s incl. the data path)')
            if (epoch + 1) % args.ckpt_save_interval == 0:
                prune_checkpoints(ckpt_dir, args.max_ckpt_save_num)           # before the save, by age: never the new file
                save_checkpoint(ddp, opt, epoch + 1, it, ckpt_dir / f'checkpoint_epoch_{epoch + 1}.pth')
    # --num_epochs_to_eval N (tools/train.py:335-372 -> repeat_eval_ckpt): evaluate the last N checkpoints of a detector
    if args.num_epochs_to_eval > 0 and cfg.MODEL.get('DENSE_HEAD', None) is not None:
        from tmae_amd.eval import eval_one_epoch
        from tmae_amd.train import SyntheticEvalLoader
        test_loader = None
        if not args.synthetic:
            _, test_loader, _ = build_dataloader(cfg.DATA_CONFIG, cfg.CLASS_NAMES, bs, dist=world > 1, root_path=args.data_path,
                                                 workers=args.workers, logger=logger, training=False)
        cfg.LOCAL_RANK = local_rank
        # rank 0 has finished writing / pruning before anybody lists the directory, and every rank evaluates the list rank 0
        # saw (a rank that globbed by itself could see a file about to be pruned: mismatched all_gather_object calls)
        names = [[str(p_) for p_ in list_checkpoints(out / 'ckpt')[-args.num_epochs_to_eval:]]] if rank == 0 else [None]
        if world > 1:
            dist.barrier()
            dist.broadcast_object_list(names, src=0)
        for ck in map(Path, names[0]):
            model.load_params_from_file(str(ck), logger=logger)
            loader = test_loader if test_loader is not None else SyntheticEvalLoader(ds, 4 * bs, bs, rank=rank, world=world)
            ret = eval_one_epoch(cfg, model, loader, ck.stem, logger, dist_test=world > 1, result_dir=out / 'eval' / ck.stem,
                                 amp_dtype=amp)
            if rank == 0:
                logger.info({k: float(v) for k, v in ret.items()})
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
