"""CPU, world_size 2, gloo: the N>1 path of the step -- frame-pair data parallel: per-rank shards,
DDP gradient averaging, identical weights after AdamOneCycle on every rank, max-over-ranks timing."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tmae_amd.train import AdamOneCycle, OneCycle, wrap_ddp
    from tmae_amd.train.engine import train_one_step
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 1))
    ddp = wrap_ddp(model, rank)
    opt = AdamOneCycle(ddp.parameters(), wd=0.01)
    sch = OneCycle(opt, 10, 3e-3, [0.95, 0.85], 10, 0.4)
    x = torch.randn(4, 8, generator=torch.Generator().manual_seed(100 + rank))     # rank-specific shard

    def model_func(m, batch):
        return m(batch).pow(2).mean(), {}, {}

    # expected averaged gradient, computed without DDP
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 1))
    ref.load_state_dict(model.state_dict())
    grads = []
    for r in range(world):
        ref.zero_grad()
        xr = torch.randn(4, 8, generator=torch.Generator().manual_seed(100 + r))
        ref(xr).pow(2).mean().backward()
        grads.append([p.grad.clone() for p in ref.parameters()])
    expect = [sum(g[i] for g in grads) / world for i in range(len(grads[0]))]

    opt.zero_grad()
    with torch.autocast('cpu', enabled=False):
        loss = model_func(ddp, x)[0]
    loss.backward()
    ok_grad = all(torch.allclose(p.grad, e, atol=1e-6) for p, e in zip(model.parameters(), expect))
    loss2, _, _ = train_one_step(ddp, opt, sch, x, 0, model_func, amp_dtype=None)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                         # bench.py's time reduction
    q.put((rank, ok_grad, same, float(t)))
    dist.destroy_process_group()


def test_two_rank_data_parallel_step():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in range(2)]
    [p.join(60) for p in procs]
    for rank, ok_grad, same, tmax in res:
        assert ok_grad, f'rank {rank}: DDP gradient is not the mean over ranks'
        assert same, f'rank {rank}: weights diverged across ranks after the optimizer step'
        assert tmax == 2.0
