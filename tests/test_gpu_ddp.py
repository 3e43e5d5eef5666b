"""GPU (-m gpu): two ranks on the one GPU of the box (gloo transport, CUDA tensors): the full TMAE model under
DistributedDataParallel -- custom autograd Functions + Siamese weight reuse + DDP hooks -- one optimizer step;
ranks must end with identical weights and the averaged gradient of the two shards."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, fl

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, shipped=False):
    for p in (os.path.join(ROOT, 't-mae_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import tmae_oracle as O
    from conftest import build_product_model
    from pcdet.models import model_fn_decorator
    from tmae_amd.train import AdamOneCycle, OneCycle, train_one_step, wrap_ddp
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    P = O.init_params(O.default_model_cfg(1), seed=3, pred_scale=0.1)
    model, cfg, _ = build_product_model(1, params=P, device=dev)
    model.train()
    # shipped = the path bench.py / tools/train.py take: wrap_ddp (gradient_as_bucket_view, 25 MB buckets, buffer
    # broadcast), the reference's optimizer grouping, bf16 autocast + refresh_param_copies, two steps
    ddp = wrap_ddp(model, 0) if shipped else torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    opt = AdamOneCycle(model if shipped else model.parameters())
    sch = OneCycle(opt, 10, 3e-3, [0.95, 0.85], 10, 0.4)
    pts, prv = O.synth_frame_pair(3000, 2, seed=50 + rank)                 # rank-specific shard
    noise = torch.rand(100000, generator=torch.Generator().manual_seed(rank))
    batch = {'points': torch.from_numpy(pts).to(dev), 'points_prev': torch.from_numpy(prv).to(dev), 'batch_size': 2}
    vox = O.voxelize(pts, [-74.88, -74.88, -5, 74.88, 74.88, 3], [0.32, 0.32, 8], [468, 468, 1])
    batch['mae_noise'] = noise[:vox['voxel_coords'].shape[0]].to(dev)
    amp = torch.bfloat16 if shipped else None
    loss, _, _ = train_one_step(ddp, opt, sch, dict(batch), 0, model_fn_decorator(), amp_dtype=amp)
    if shipped:
        loss, _, _ = train_one_step(ddp, opt, sch, dict(batch), 1, model_fn_decorator(), amp_dtype=amp)
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
    wts = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    gl = [torch.zeros_like(g) for _ in range(world)]
    wl = [torch.zeros_like(wts) for _ in range(world)]
    dist.all_gather(gl, g)
    dist.all_gather(wl, wts)
    q.put((rank, fl(loss), bool(torch.equal(gl[0], gl[1])), bool(torch.equal(wl[0], wl[1])),
           bool(torch.isfinite(g).all()), float(g.norm())))
    dist.destroy_process_group()


@pytest.mark.parametrize('shipped', [False, True])
def test_two_ranks_full_model_ddp_step(shipped):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000 + (7 if shipped else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, shipped)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=500) for _ in range(2))
    [p.join(120) for p in procs]
    assert res[0][1] != res[1][1]                        # different shards -> different local losses
    for rank, loss, same_g, same_w, finite, gn in res:
        assert finite and gn > 0
        assert same_g, 'DDP did not leave the same (averaged) gradient on both ranks'
        assert same_w, 'weights diverged after the optimizer step'


def _rccl_worker(port, q):
    """One rank on the `nccl` backend (= RCCL): process-group creation, DistributedDataParallel with bucket views over
    RCCL's all-reduce (one rank: the collective runs, the result is the local gradient), the shipped step twice."""
    for p in (os.path.join(ROOT, 't-mae_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    import tmae_oracle as O
    from conftest import build_product_model
    from pcdet.models import model_fn_decorator
    from tmae_amd.train import AdamOneCycle, OneCycle, train_one_step
    dev = torch.device('cuda:0')
    P = O.init_params(O.default_model_cfg(1), seed=3, pred_scale=0.1)
    pts, prv = O.synth_frame_pair(3000, 2, seed=50)
    vox = O.voxelize(pts, [-74.88, -74.88, -5, 74.88, 74.88, 3], [0.32, 0.32, 8], [468, 468, 1])
    noise = torch.rand(100000, generator=torch.Generator().manual_seed(0))[:vox['voxel_coords'].shape[0]]
    out = []
    for use_ddp in (False, True):
        model, cfg, _ = build_product_model(1, params=P, device=dev)
        model.train()
        net = model
        if use_ddp:               # wrap_ddp's arguments (it returns the bare model for a world of one)
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=True,
                                                            gradient_as_bucket_view=True, bucket_cap_mb=25)
        opt = AdamOneCycle(model)
        sch = OneCycle(opt, 10, 3e-3, [0.95, 0.85], 10, 0.4)
        batch = {'points': torch.from_numpy(pts).to(dev), 'points_prev': torch.from_numpy(prv).to(dev), 'batch_size': 2,
                 'mae_noise': noise.to(dev)}
        losses = []
        for it in range(2):
            loss, _, _ = train_one_step(net, opt, sch, dict(batch), it, model_fn_decorator(), amp_dtype=torch.bfloat16)
            losses.append(fl(loss))
        wts = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
        out.append((losses, wts))
    t = torch.ones(1024, device=dev)
    dist.all_reduce(t)
    ok = bool((t == 1).all())
    dist.barrier()
    dist.destroy_process_group()
    (l0, w0), (l1, w1) = out
    q.put((l0, l1, float((w0 - w1).abs().max()), float(w0.abs().max()), ok))


def test_single_rank_rccl_backend_runs_the_shipped_step():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(31700 + os.getpid() % 2000, q))
    p.start()
    l0, l1, dw, wmax, ok = q.get(timeout=500)
    p.join(120)
    assert ok, 'RCCL all_reduce on one rank changed the data'
    assert all(abs(a - b) <= 2e-3 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)
    assert dw <= 2e-2 * wmax, (dw, wmax)          # same two steps with and without DistributedDataParallel


def test_bench_two_rank_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` end to end -- its own torch.distributed.run launch, rank bookkeeping, barriers, the
    MAX-over-ranks clock, the per-rank skew fields and rank 0's ONE JSON line -- with the test-only switch
    TMAE_BENCH_SHARED_GPU=1 (both ranks on GPU 0, gloo instead of RCCL: two ranks cannot share a GPU under RCCL).  The
    8-GPU run is the driver's; this is the part of it that must not fail on its first execution."""
    import json
    import subprocess
    env = dict(os.environ, TMAE_BENCH_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
           '--no-secondary', '--batch-per-gpu', '4', '--points', '40000']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, lines                                     # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks'] == 2 and d['config']['global_batch'] == 8 and d['config']['parallelism'] == 'dp2'
    assert d['steps'] == 3 and d['scaling'] == 'weak' and 'REHEARSAL' in d['collective_backend']
    assert 0 < d['rank_ms_per_step']['min'] <= d['rank_ms_per_step']['max'] <= d['ms_per_step'] * 1.001
    assert abs(d['value'] - 8 * 3 / (d['ms_per_step'] * 3e-3)) <= 1e-2 * d['value']
    for key in ('roofline', 'roofline_wgrad', 'roofline_attention', 'roofline_step', 'box_peaks'):
        assert key in d, key
    assert 'GELU2' in d['roofline']['kernel'] and 'roofline_token_gemm_plain' in d     # the priced instance = the dual-store kernel
    assert d['roofline']['peak_measured'] == d['box_peaks']['hbm_copy_gbs'] > 1000
    assert d['roofline_step']['peak_measured'] == d['box_peaks']['mfma_bf16_tflops'] > 100
    # each rank's launcher (and the threads it starts) is pinned to its own cores before its first GPU call (train/affinity.py)
    print('rank cores:', d['rank_cores'])
    assert len(d['rank_cores']['per_rank']) == 2 and d['rank_cores']['disjoint'] and d['rank_cores']['source'] != 'unpinned'


def _syncbn_worker(rank, world, port, q):
    for p in (os.path.join(ROOT, 't-mae_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tmae_amd import ops
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    out = {}
    g = torch.Generator().manual_seed(7)
    # ---- row BatchNorm (+ ReLU), two row groups (the two frames of the pair encoder), different row counts per rank
    c = 128
    sizes = [[3000, 1700], [2100, 2600]]
    xs = [torch.randn(sum(sz), c, generator=g) * 2 + 0.5 for sz in sizes]
    dys = [torch.randn(sum(sz), c, generator=g) for sz in sizes]
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    bn = torch.nn.SyncBatchNorm(c, eps=1e-3, momentum=0.01).to(dev)
    with torch.no_grad():
        bn.weight.copy_(gamma), bn.bias.copy_(beta)
    bn.train()
    x = xs[rank].to(dev).requires_grad_(True)
    y = ops.batch_norm_relu(x, bn, relu=True, groups=sizes[rank])
    y.backward(dys[rank].to(dev))
    out['bn'] = (y.detach().cpu(), x.grad.cpu(), bn.weight.grad.cpu(), bn.bias.grad.cpu(), bn.running_mean.cpu(),
                 bn.running_var.cpu(), int(bn.num_batches_tracked))
    # the same union of rows through ONE plain BatchNorm per group on this rank alone = the definition of SyncBatchNorm
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev)
    with torch.no_grad():
        ref.weight.copy_(gamma), ref.bias.copy_(beta)
    ref.train()
    ys, gxs = [], []
    for grp in range(2):
        parts = []
        for r in range(world):
            o = sum(sizes[r][:grp])
            parts.append((xs[r][o:o + sizes[r][grp]], dys[r][o:o + sizes[r][grp]]))
        xu = torch.cat([p_[0] for p_ in parts]).to(dev).requires_grad_(True)
        yu = torch.relu(ref(xu))
        yu.backward(torch.cat([p_[1] for p_ in parts]).to(dev))
        lo = sum(sizes[r][grp] for r in range(rank))
        ys.append(yu.detach()[lo:lo + sizes[rank][grp]].cpu())
        gxs.append(xu.grad[lo:lo + sizes[rank][grp]].cpu())
    out['bn_ref'] = (torch.cat(ys), torch.cat(gxs), ref.weight.grad.cpu(), ref.bias.grad.cpu(), ref.running_mean.cpu(),
                     ref.running_var.cpu())
    # gamma / beta gradients are the rank's own sums (DDP averages them later): their sum over ranks = the union's
    gsum = torch.stack([bn.weight.grad, bn.bias.grad]).clone()
    dist.all_reduce(gsum)
    out['bn_gsum'] = gsum.cpu()
    # numpy, not tensors: a tensor travels through the queue as a file descriptor of a process that may be gone
    out = {k: tuple(v.numpy() if torch.is_tensor(v) else v for v in val) if isinstance(val, tuple) else val.numpy()
           for k, val in out.items()}
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equals_the_union_batch():
    """tools/train.py --sync_bn (the reference: convert_sync_batchnorm, tools/train.py:244-245): a SyncBatchNorm module
    makes ops.batch_norm_relu merge its statistics and backward sums over the ranks.  Two ranks (gloo, one GPU) with
    different row counts against ONE BatchNorm over the union of their rows: outputs, input gradients, the summed
    gamma / beta gradients and the running statistics."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 30900 + os.getpid() % 2000
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=300) for _ in range(2))
    [p.join(60) for p in procs]
    for rank in range(2):
        y, gx, gw, gb, rm, rv, nbt = (torch.from_numpy(v) if isinstance(v, np.ndarray) else v for v in res[rank]['bn'])
        y0, gx0, gw0, gb0, rm0, rv0 = (torch.from_numpy(v) for v in res[rank]['bn_ref'])
        gsum = torch.from_numpy(res[rank]['bn_gsum'])
        assert nbt == 2                                                       # two groups = two BatchNorm calls
        torch.testing.assert_close(y, y0, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(gx, gx0, rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(rm, rm0, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv, rv0, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(gsum[0], gw0, rtol=2e-4, atol=2e-3)
        torch.testing.assert_close(gsum[1], gb0, rtol=2e-4, atol=2e-3)
