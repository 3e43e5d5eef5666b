"""Benchmark of the T-MAE pre-training hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full pre-training iteration (VFE -> Siamese SST encoder -> masking -> WCA -> dense decoder ->
Chamfer -> backward -> Adam one-cycle) over one batch of synthetic 120k-point ONCE-shape frame pairs, inputs
resident in HBM.  value = frame-pairs/s of the whole job (all ranks).  Rank 0 prints ONE JSON line carrying the
`roofline` of the dominant hand-written kernel (timed live with HIP events on the launch stream) and the
`cpu_baseline` (the oracle timed on this box's host cores on one frame pair).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))

# MIOpen's default find benchmarks every applicable solver for the 4 dense decoder convolutions, including the
# reference 'naive' solvers that take ~90 s on these shapes; they are never the winner, so leave them out of the search
for _k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD',
           'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
    os.environ.setdefault(_k, '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16
VALU_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-per-gpu', type=int, default=8)        # OPTIMIZATION.BATCH_SIZE_PER_GPU
    ap.add_argument('--points', type=int, default=120000)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--task', default='pretrain', choices=['pretrain', 'finetune'],
                    help='pretrain = the metric (t_mae_ssl.yaml); finetune = BASELINE configs[4]: CenterPoint head on the '
                         'two-frame encoder (t_mae.yaml, synthetic labels; informational, batch 6 = its recipe)')
    ap.add_argument('--shape', default='once', choices=['once', 'waymo'],
                    help='once = BASELINE configs[1] (the metric); waymo = configs[3] shape: 5 point features, z in [-2,4), '
                         '6 m pillars (use with --points 180000)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-points', type=int, default=120000)
    ap.add_argument('--probe-only', action='store_true', help='only time the roofline kernel (for PMC runs)')
    return ap.parse_args()


def _pmc(key):
    f = os.path.join(ROOT, 'profiles', 'round1_pmc.json')       # rocprofv3 --pmc passes of --probe-only
    if not os.path.exists(f):
        return None
    return json.load(open(f)).get(key, {}).get('traffic_bytes_per_op')


_STAGE2_TOKENS = []


def _stage2_tokens(model, batch):
    """Stage-2 token count (previous + current frame) of `batch`: one no-grad forward pass, once per process."""
    if not _STAGE2_TOKENS:
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            model(dict(batch))
        _STAGE2_TOKENS.append(int(model.backbone_3d.last_pair_tokens[1]))
    return _STAGE2_TOKENS[0]


def token_gemm_roofline(model, batch, amp_dtype, iters=20):
    """`roofline`: the dominant hand-written kernel family of the step by GPU time (profiles/round1_g_kernel_stats.md):
    the token GEMM of csrc/token_gemm.hip, here its W-resident persistent kernel token_gemm_res_kernel<256,4> on its
    heaviest frequent shape -- the stage-2 FFN / q,k in-projection  Y[m,512] = X[m,256] W^T + b  over the token list of
    both frames (bench batch).  One launch per op, timed with HIP events on the launch stream.  Algorithmic bytes: X read
    once, Y written once, W and b read once.  The probe's launches are the last 23 launches of that
    kernel in the process (3 warm-up + 20 timed): that is how the profile summaries tell them from the forward pass
    that measures m."""
    from tmae_amd._lib import lib, check
    m = _stage2_tokens(model, batch)
    n, k = 512, 256
    dev = next(model.parameters()).device
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    b = torch.randn(n, device=dev).bfloat16()
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        check(lib.tmae_token_gemm(x.data_ptr(), k, m, k, w.data_ptr(), n, b.data_ptr(), y.data_ptr(), n, st),
              'tmae_token_gemm')
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_alg = m * (n + k) * 2 + (n * k + n) * 2
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    return {'kernel': 'token_gemm_res_kernel<256,4> (W-resident persistent token GEMM Y[m,512] = X[m,256] W^T + b of the stage-2 '
                      'token list; one launch per op)', 'bound': 'hbm', 'achieved': round(achieved, 2),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5),
            'traffic': _pmc('token_gemm'),
            'traffic_source': 'profiles/round1_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, bytes per op)',
            'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(bytes_alg), 'tokens': m, 'n': n, 'k': k}


def wgrad_roofline(model, batch, amp_dtype, iters=20):
    """`roofline_wgrad` (the step's dominant kernel until its loads were made branch-free): the token-split weight
    gradient of the d = 256 stages, wgrad256_kernel (csrc/wgrad.hip), on its most frequent heavy shape -- the
    stage-2 FFN / in-projection  dW[512,256] = dY^T X  over the token list of both frames (bench batch).  The op = that
    kernel + its slab-reduction launch, timed with HIP events on the launch stream.  Algorithmic bytes: dY and
    X read once, dW and db written once; the fp32 slabs are overhead and show up in `traffic`."""
    from tmae_amd import ops
    m = _stage2_tokens(model, batch)
    n, k = 512, 256
    dev = next(model.parameters()).device
    dy = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    for _ in range(3):
        ops.linear_wgrad(dy, x, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.linear_wgrad(dy, x, True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_alg = m * (n + k) * 2 + (n * k + n) * 4
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    return {'kernel': 'wgrad256_kernel (token-split weight gradient dW[512,256] = dY^T X of the stage-2 token list; the '
                      'op = the kernel + its slab-reduction launch)', 'bound': 'hbm',
            'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': _pmc('wgrad'),
            'traffic_source': 'profiles/round1_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, bytes per op)',
            'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(bytes_alg), 'tokens': m, 'n': n, 'k': k}


def attention_roofline(model, batch, amp_dtype, iters=20):
    """Times the dominant hand-written kernel of the step -- the ragged window attention backward
    (`win_attn_bwd_mfma_kernel<16,{1,2,4}>`: one launch per tile class) on the stage-1 previous-frame tensors --
    through the C ABI with HIP events on torch's current stream (the stream the ABI launches on), nothing else in
    between.  Priced by its ALGORITHMIC HBM bytes: q,k,v,out,dout read once, dq,dk,dv written
    once, + lse + the dense index grid."""
    from tmae_amd import ops
    from tmae_amd._lib import lib, check
    with torch.no_grad():
        bd = model.vfe(dict(batch))
        ind = bd['voxel_coords_prev'][:, [0, 2, 3]].int().contiguous()
        bs = int(bd['batch_size'])
        grid = ops.index_grid(ind, bs, 468, 468)
        wl = ops.window_worklist(grid, grid, bs, 468, 468, False)
    m, d, H, dh = ind.shape[0], 128, 8, 16
    dt = amp_dtype or torch.float32
    es, code = (2, 1) if dt == torch.bfloat16 else (4, 0)
    dev = ind.device
    qk = torch.randn(m, 2 * d, device=dev).to(dt)
    v = torch.randn(m, d, device=dev).to(dt)
    dout = torch.randn(m, d, device=dev).to(dt)
    out, dqk, dv = torch.empty_like(v), torch.empty_like(qk), torch.empty_like(v)
    lse = torch.empty(m, H, device=dev)
    tau = torch.ones(1, device=dev)
    nblk = lib.tmae_win_attn_num_blocks(bs, 468, 468, H, dh)
    part = torch.zeros(nblk, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    wlp = wl.data_ptr() if code == 1 else None
    q_, k_ = qk.data_ptr(), qk.data_ptr() + d * es
    check(lib.tmae_win_attn_fwd(q_, 2 * d, k_, 2 * d, v.data_ptr(), d, code, m, m, H, dh, grid.data_ptr(),
                                grid.data_ptr(), bs, 468, 468, 0, tau.data_ptr(), 0.01, out.data_ptr(), d,
                                lse.data_ptr(), wlp, st), 'fwd')

    def bwd():
        check(lib.tmae_win_attn_bwd(q_, 2 * d, k_, 2 * d, v.data_ptr(), d, out.data_ptr(), d, dout.data_ptr(), d,
                                    lse.data_ptr(), code, m, m, H, dh, grid.data_ptr(), grid.data_ptr(), bs, 468, 468,
                                    0, tau.data_ptr(), 0.01, dqk.data_ptr(), 2 * d, dqk.data_ptr() + d * es, 2 * d,
                                    dv.data_ptr(), d, part.data_ptr(), wlp, st), 'bwd')
    for _ in range(3):
        bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    # algorithmic bytes of the op: read q,k,v,dout (+ out on the fp32 path; the bf16 kernel rebuilds D from P.dP),
    # write dq,dk,dv; + the lse rows and one pass over the row-index grid
    units = 7 if code == 1 else 8
    bytes_alg = m * d * es * units + m * H * 4 + bs * 468 * 468 * 4
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    traffic = _pmc('attention') if code == 1 else None
    return {'kernel': 'win_attn_bwd_mfma_kernel<16,NT> (stage-1 self-attention backward, previous frame; the op = its 3 '
                      'tile-class launches NT=1,2,4)', 'bound': 'hbm', 'achieved': round(achieved, 2),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
            'traffic_source': 'profiles/round1_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, bytes per op)',
            'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(bytes_alg), 'tokens': int(m)}


def cpu_baseline(n_points):
    """The oracle (oracle/tmae_oracle.py, a CPU restatement pinned against the reference) timed on this box's
    host cores: forward + backward of ONE frame pair, fp32, all cores."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import tmae_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # the 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    cfg = O.default_model_cfg(3)
    P = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=0).items()}
    pts, prv = O.synth_frame_pair(n_points, 1, seed=0)
    vox = O.voxelize(pts, cfg['point_cloud_range'], cfg['voxel_size'], cfg['grid_size'])
    noise = np.random.default_rng(0).random(vox['voxel_coords'].shape[0]).astype(np.float32)
    t0 = time.perf_counter()
    loss = O.forward_loss(P, pts, prv, noise, 1, cfg)
    loss.backward()
    dt = time.perf_counter() - t0
    return {'value': round(1.0 / dt, 5), 'unit': 'frame-pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'1 frame pair of {n_points} pts/frame, forward+backward (no optimizer), fp32, {dt:.1f} s'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl')          # RCCL over xGMI
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import model_fn_decorator
    from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler,
                                train_one_step, wrap_ddp)
    yaml_name = 't_mae_ssl.yaml' if args.task == 'pretrain' else 't_mae.yaml'
    cfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', yaml_name), EasyDict())
    npf = 5
    if args.shape == 'waymo':
        cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [-74.88, -74.88, -2.0, 74.88, 74.88, 4.0]
        for p_ in cfg.DATA_CONFIG.DATA_PROCESSOR:
            if p_.NAME == 'calculate_grid_size':
                p_.VOXEL_SIZE = [0.32, 0.32, 6.0]
        npf = 6
    ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=args.points,
                                  batch_size=args.batch_per_gpu, rank=rank, num_point_features=npf,
                                  n_boxes=40 if args.task == 'finetune' else 0)
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg, ds).to(dev)
    model.train()
    ddp = wrap_ddp(model, local_rank)
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 1000, cfg.OPTIMIZATION.NUM_EPOCHS, -1, cfg.OPTIMIZATION)
    model_func = model_fn_decorator()
    amp = torch.bfloat16 if args.dtype == 'bf16' else None

    # synthetic batches, uploaded before the timed region (distinct scans, cycled)
    nb = min(args.steps + args.warmup, 4)
    batches = []
    for i in range(nb):
        b = ds.batch(i)
        batches.append({'points': torch.from_numpy(b['points']).to(dev), 'points_prev': torch.from_numpy(b['points_prev']).to(dev),
                        'batch_size': b['batch_size']})
        if 'gt_boxes' in b:
            batches[-1]['gt_boxes'] = torch.from_numpy(b['gt_boxes']).to(dev)

    def step(i):
        return train_one_step(ddp, opt, sched, dict(batches[i % nb]), i, model_func, amp_dtype=amp)[0]

    if args.probe_only:
        print(json.dumps({'roofline': token_gemm_roofline(model, dict(batches[0]), amp),
                          'roofline_wgrad': wgrad_roofline(model, dict(batches[0]), amp),
                          'roofline_attention': attention_roofline(model, dict(batches[0]), amp)}), flush=True)
        return

    def log(msg):
        if rank == 0:
            print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)

    log(f'model on {dev}, {nb} synthetic batches resident ({int(batches[0]["points"].shape[0])} + '
        f'{int(batches[0]["points_prev"].shape[0])} points each); warm-up ...')
    for i in range(args.warmup):
        tw = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        log(f'warm-up step {i}: {time.perf_counter() - tw:.2f} s')
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss_val = float(loss.detach())
    assert np.isfinite(loss_val), 'non-finite loss in the timed region'

    if rank == 0:
        pairs = args.batch_per_gpu * world * args.steps
        line = {
            'metric': ('frame-pairs/sec T-MAE pretrain, 120k-pt ONCE scans' if args.task == 'pretrain' else
                       'frame-pairs/sec T-MAE fine-tune (CenterPoint head), 120k-pt ONCE scans'),
            'value': round(pairs / elapsed, 4),
            'unit': 'frame-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'configs[{1 if args.shape == "once" else 3}]: {"ONCE" if args.shape == "once" else "Waymo"}-shape synthetic {args.points}-pt frame-pairs, full 3-stage SST '
                                   f'encoder + temporal cross-attn + decoder + Chamfer, fwd+bwd+Adam one-cycle',
                       'batch_per_gpu': args.batch_per_gpu, 'global_batch': args.batch_per_gpu * world,
                       'parallelism': f'dp{world}', 'grid': '468x468x1', 'final_loss': round(loss_val, 5),
                       'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)},
        }
        log(f'timed region done: {1e3 * elapsed / args.steps:.1f} ms/step; timing the dominant kernel ...')
        line['roofline'] = token_gemm_roofline(model, dict(batches[0]), amp)
        # round-1 history: the two kernels that led the profile before this one, still priced the same way
        line['roofline_wgrad'] = wgrad_roofline(model, dict(batches[0]), amp)
        line['roofline_attention'] = attention_roofline(model, dict(batches[0]), amp)
        if world == 1 and not args.no_cpu_baseline:
            log('timing the CPU oracle on one frame pair (cpu_baseline) ...')
            line['cpu_baseline'] = cpu_baseline(args.cpu_points)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
