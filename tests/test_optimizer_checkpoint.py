"""A15 (optimizer) and 8f-3 (checkpoints) against fixtures produced by the reference's OWN classes
(oracle/gen_golden_optim.py: OptimWrapper + OneCycle + checkpoint_state, run in the build container):
parameter groups, the 20-step trajectory, and a checkpoint written by the reference loop."""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import build_finetune_model, build_product_model, golden, load_cfg, fl


# the fixture's tiny model (oracle/gen_golden_optim.py), restated
class TinyOwner(nn.Module):
    def __init__(self):
        super().__init__()
        self.own = nn.Parameter(torch.zeros(4))
        self.proj = nn.Linear(5, 3)


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(6, 5)
        self.bn = nn.BatchNorm1d(5)
        self.owner = TinyOwner()
        self.fc2 = nn.Linear(3, 2, bias=False)


def _optim_cfg():
    from pcdet.config import EasyDict
    return EasyDict(dict(OPTIMIZER='adam_onecycle', LR=0.003, WEIGHT_DECAY=0.01, MOMS=[0.95, 0.85], PCT_START=0.4,
                         DIV_FACTOR=10))


def _group_names(model, opt):
    names = {id(p): n for n, p in model.named_parameters()}
    return [[names[id(p)] for p in g['params']] for g in opt.param_groups]


def test_param_groups_equal_the_reference_optimizer():
    """Order and membership of the two Adam param groups (non-BatchNorm / BatchNorm leaves) and the parameters the
    reference never trains (in_proj_weight / in_proj_bias / tau of every attention module)."""
    from tmae_amd.train import build_optimizer
    g = golden('O1_optimizer')
    for tag, model in (('pre', build_product_model(3)[0]), ('ft', build_finetune_model()[0])):
        opt = build_optimizer(model, _optim_cfg())
        groups = _group_names(model, opt)
        assert groups[0] == [str(n) for n in g[f'{tag}_group0']], tag
        assert groups[1] == [str(n) for n in g[f'{tag}_group1']], tag
        names = {id(p): n for n, p in model.named_parameters()}
        assert [names[id(p)] for p in opt.unoptimized] == [str(n) for n in g[f'{tag}_unoptimized']]
    pre = build_product_model(3)[0]
    opt = build_optimizer(pre, _optim_cfg())
    assert sum(p.numel() for p in opt.unoptimized) == 2665746
    cfg = _optim_cfg()
    cfg.TRAIN_NONLEAF_PARAMS = True                                          # opt-in: train them too
    assert not build_optimizer(pre, cfg).unoptimized


def test_trajectory_equals_the_reference_optimizer():
    """20 steps of build_optimizer + build_scheduler in the reference loop's order on the fixture's gradients."""
    from tmae_amd.train import build_optimizer, build_scheduler
    g = golden('O1_optimizer')
    names = [str(n) for n in g['tiny_names']]
    model = Tiny()
    assert [n for n, _ in model.named_parameters()] == names
    with torch.no_grad():
        for i, (n, p) in enumerate(model.named_parameters()):
            p.copy_(torch.from_numpy(g[f'tiny_init_{i}']))
    opt = build_optimizer(model, _optim_cfg())
    sched, _ = build_scheduler(opt, 10, 2, -1, _optim_cfg())
    groups = _group_names(model, opt)
    assert groups[0] == [str(n) for n in g['tiny_group0']] and groups[1] == [str(n) for n in g['tiny_group1']]
    for it in range(20):
        sched.step(it)
        assert abs(opt.lr - float(g['tiny_lr'][it])) <= 1e-12 and abs(opt.mom - float(g['tiny_mom'][it])) <= 1e-12
        opt.zero_grad()
        for i, (n, p) in enumerate(model.named_parameters()):
            p.grad = None if (it == 7 and n == 'fc2.weight') else torch.from_numpy(g[f'tiny_grad_{i}'][it].copy())
        opt.step()
        for i, (n, p) in enumerate(model.named_parameters()):
            np.testing.assert_allclose(p.detach().numpy(), g[f'tiny_traj_{i}'][it], rtol=1e-6, atol=1e-7,
                                       err_msg=f'{n} step {it}')


@pytest.mark.gpu
def test_trajectory_equals_the_reference_optimizer_gpu():
    """The same 20 steps on the GPU, all on the one-launch multi-tensor step (tmae_adam_step): step 7 has a parameter
    without a gradient (decay only), and from step 8 on that tensor's step count lags the others' -- every table entry
    carries its own step number, as torch.optim.Adam keeps one per parameter."""
    from tmae_amd.train import build_optimizer, build_scheduler
    g = golden('O1_optimizer')
    dev = torch.device('cuda', 0)
    model = Tiny().to(dev)
    with torch.no_grad():
        for i, (n, p) in enumerate(model.named_parameters()):
            p.copy_(torch.from_numpy(g[f'tiny_init_{i}']))
    opt = build_optimizer(model, _optim_cfg())
    assert opt._native
    sched, _ = build_scheduler(opt, 10, 2, -1, _optim_cfg())
    for it in range(20):
        sched.step(it)
        opt.zero_grad()
        for i, (n, p) in enumerate(model.named_parameters()):
            p.grad = None if (it == 7 and n == 'fc2.weight') else torch.from_numpy(g[f'tiny_grad_{i}'][it].copy()).to(dev)
        assert opt._native_step() is True                   # (== opt.step(); mixed step counts stay on the one-launch path)
        for i, (n, p) in enumerate(model.named_parameters()):
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f'tiny_traj_{i}'][it], rtol=2e-6, atol=2e-7,
                                       err_msg=f'{n} step {it}')
    # the state is torch.optim.Adam's: step tensors on the device, equal to the number of gradients each tensor received
    owned = {id(p) for p in opt.decayed}
    for n, p in model.named_parameters():
        if id(p) in owned:
            st = opt.opt.state[p]
            assert float(st['step']) == (19.0 if n == 'fc2.weight' else 20.0), (n, float(st['step']))


@pytest.mark.gpu
def test_multi_tensor_adam_step_equals_torch_adam():
    """tmae_adam_step vs torch.optim.Adam (single-tensor formulas) + the decoupled decay on 40 tensors of odd sizes,
    some gradients 4-byte aligned views into a flat buffer (DDP's gradient_as_bucket_view), over 5 steps with changing lr
    and beta1."""
    from tmae_amd.train.optim import AdamOneCycle
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    sizes = [(1,), (3,), (4097,), (128, 33), (256, 9, 17), (8191,), (4096,), (5, 7, 11)] * 5
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in sizes]
    ref = [p.detach().clone().double() for p in ps]
    m = [torch.zeros_like(r) for r in ref]
    v = [torch.zeros_like(r) for r in ref]
    opt = AdamOneCycle(ps, lr=3e-3, wd=0.01, betas=(0.9, 0.99))
    assert opt._native
    flat = torch.zeros(sum(p.numel() for p in ps) + 1, device=dev)
    for it in range(5):
        versions = [p._version for p in ps]
        opt.lr, opt.mom = 3e-3 * (1 + it), 0.95 - 0.02 * it
        off = 1                                              # +1: the views start 4 bytes off a 16-byte boundary
        for p in ps:
            gview = flat[off:off + p.numel()].view_as(p)
            gview.copy_(torch.randn_like(p) * (10.0 ** (it - 2)))
            p.grad = gview if p.numel() % 2 else gview.clone()
            off += p.numel()
        opt.step(copy_dtype=torch.bfloat16)            # also refreshes the bf16 copies the autocast path reads
        from tmae_amd import ops
        for p in ps:
            c = p._tmae_copy
            assert c[0] == p._version and torch.equal(c[2], p.detach().bfloat16())
            assert ops.cast_param(p, torch.bfloat16) is c[2]
        b1, b2, lr, eps = opt.mom, 0.99, opt.lr, 1e-8
        assert all(p._version > v0 for p, v0 in zip(ps, versions))    # caches keyed on the version see the update
        for k, p in enumerate(ps):
            gd = p.grad.double()
            ref[k] *= 1 - 0.01 * lr
            m[k] = m[k] + (gd - m[k]) * (1 - b1)
            v[k] = b2 * v[k] + (1 - b2) * gd * gd
            bc1, bc2 = 1 - b1 ** (it + 1), 1 - b2 ** (it + 1)
            ref[k] -= lr / bc1 * m[k] / (v[k].sqrt() / bc2 ** 0.5 + eps)
            err = (p.detach().double() - ref[k]).abs().max().item()
            assert err <= 2e-6 * max(1.0, ref[k].abs().max().item()), (it, k, err)
            st = opt.opt.state[p]
            assert float(st['step']) == it + 1
            assert (st['exp_avg'].double() - m[k]).abs().max().item() <= 5e-6 * max(1e-3, m[k].abs().max().item())
    assert opt._table is not None


@pytest.mark.gpu
def test_optimizer_step_does_not_wait_for_the_gpu():
    """The one-launch step must ENQUEUE and return: its pointer table travels from pinned memory with an asynchronous
    copy.  (A pageable torch.tensor(..., device=) copy made the host wait for everything queued before -- the whole
    backward pass -- and the next forward then started on an empty queue.)  With ~100 ms of matmuls queued, step() has
    to come back long before they finish."""
    import time
    from tmae_amd.train.optim import AdamOneCycle
    dev = torch.device('cuda', 0)
    ps = [torch.nn.Parameter(torch.randn(64, 64, device=dev)) for _ in range(50)]
    opt = AdamOneCycle(ps, lr=1e-3, wd=0.01)
    for it in range(2):                                   # first steps: state / table creation
        for p in ps:
            p.grad = torch.randn_like(p)
        opt.step(copy_dtype=torch.bfloat16)
    a = torch.randn(8192, 8192, device=dev)
    b = (a @ a) * 1e-4                                    # library warm-up outside the measurement
    torch.cuda.synchronize()
    for _ in range(16):
        b = (b @ a) * 1e-4
    for p in ps:
        p.grad = torch.randn_like(p)                      # fresh gradient tensors: the table has to be re-sent
    t1 = time.perf_counter()
    opt.step(copy_dtype=torch.bfloat16)
    t_step = time.perf_counter() - t1
    torch.cuda.synchronize()
    t_rest = time.perf_counter() - t1 - t_step
    assert t_rest > 0.02, (t_step, t_rest)                # the GPU still had >= 20 ms of queued work when step() returned
    assert t_step < 0.02, (t_step, t_rest)                # ... and step() did not wait for it


@pytest.mark.gpu
def test_optimizer_steps_queued_far_ahead_of_the_gpu_read_their_own_tables():
    """Four optimizer steps (and four BatchNorm running-statistics flushes) enqueued back to back behind ~100 ms of
    queued matmuls, no synchronisation in between: every step's pointer table travels through a pinned staging buffer,
    and a buffer may be rewritten only after the copy last queued out of it has run (_lib.PinnedStager) -- otherwise an
    earlier launch would read a later step's gradient / exp_avg pointers.  Result vs the same steps on torch.optim.Adam."""
    from tmae_amd import ops
    from tmae_amd.train.optim import AdamOneCycle
    dev = torch.device('cuda', 0)
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(97, 33, device=dev)) for _ in range(30)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    grads = [[torch.randn_like(p) for p in ps] for _ in range(5)]
    opt = AdamOneCycle(ps, lr=2e-3, wd=0.01)
    ropt = torch.optim.Adam(ref, lr=2e-3, betas=(0.9, 0.99))
    bns = [torch.nn.BatchNorm1d(16).to(dev) for _ in range(3)]
    rbn = [(b.running_mean.clone(), b.running_var.clone()) for b in bns]
    stats = [[(torch.randn(16, device=dev), torch.rand(16, device=dev) + 0.5) for _ in bns] for _ in range(5)]

    def one(it):
        for p, gr in zip(ps, grads[it]):
            p.grad = gr.clone()                               # fresh allocations: the table changes every step
        opt.step()
        with ops.defer_bn_updates():
            for b, (mu, var) in zip(bns, stats[it]):
                ops._bn_running_update(b, mu, var, 100.0)
    one(0)                                                    # state / table creation
    a = torch.randn(8192, 8192, device=dev)
    b = (a @ a) * 1e-4
    torch.cuda.synchronize()
    for _ in range(16):
        b = (b @ a) * 1e-4                                    # ~100 ms of queued work in front of the steps
    for it in range(1, 5):
        one(it)
    torch.cuda.synchronize()
    for it in range(5):
        with torch.no_grad():
            torch._foreach_mul_(ref, 1.0 - 0.01 * 2e-3)
        for p, gr in zip(ref, grads[it]):
            p.grad = gr
        ropt.step()
        for k, (mu, var) in enumerate(stats[it]):
            rbn[k][0].mul_(0.9).add_(mu, alpha=0.1)
            rbn[k][1].mul_(0.9).add_(var, alpha=0.1 * 100.0 / 99.0)
    for p, r in zip(ps, ref):
        assert (p - r).abs().max().item() <= 2e-6 * max(1.0, r.abs().max().item())
    for b, (rm, rv) in zip(bns, rbn):
        torch.testing.assert_close(b.running_mean, rm, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(b.running_var, rv, rtol=1e-6, atol=1e-6)
        assert int(b.num_batches_tracked) == 5


def build_tiny_tmae(device='cpu'):
    """The small 1-stage model of the O2 fixture through the product's registry path (YAML edits only)."""
    from pcdet.models import build_network
    from tmae_amd.train import SyntheticTemporalDataset
    g = golden('O2_checkpoint')
    t = json.loads(str(g['tiny_model']))
    cfg = load_cfg(1)
    cfg.MODEL.VFE.MLPS = t['vfe_mlps']
    enc = cfg.MODEL.BACKBONE_3D.SST_BLOCK_LIST[0].ENCODER
    enc.D_MODEL, enc.NHEAD, enc.DIM_FEEDFORWARD, enc.NUM_BLOCKS = t['d_model'], t['nhead'], t['dff'], t['num_blocks']
    fl = cfg.MODEL.BACKBONE_3D.FUSE_LAYER['x_conv1']
    fl.NUM_FILTER = fl.NUM_UPSAMPLE_FILTER = t['d_model']
    ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=1000, batch_size=2)
    return build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds).to(device), g


def reference_checkpoint(g, spconv1_layout=False, prefix='state'):
    """The dict the reference's checkpoint_state returned, rebuilt from the fixture's arrays."""
    names = [str(n) for n in g['state_names']]
    ms = {n: torch.from_numpy(np.array(g[f'{prefix}_{i}'])) for i, n in enumerate(names)}
    if spconv1_layout:          # what spconv 1.x holds: (k1, k2, c_in, c_out) (detector3d_template.py:373-383)
        for n in names:
            if n.endswith('conv_out.0.weight') and ('sst_blocks' in n or 'wca_blocks' in n):
                ms[n] = ms[n].permute(1, 2, 3, 0).contiguous()
    pgs = json.loads(str(g['opt_param_groups']))
    for pg in pgs:
        pg['betas'] = tuple(pg['betas'])
    state = {}
    for idx in g['opt_state_ids']:
        idx = int(idx)
        state[idx] = {'step': torch.tensor(float(g[f'opt_{idx}_step'])),
                      'exp_avg': torch.from_numpy(np.array(g[f'opt_{idx}_exp_avg'])),
                      'exp_avg_sq': torch.from_numpy(np.array(g[f'opt_{idx}_exp_avg_sq']))}
    return {'epoch': int(g['epoch']), 'it': int(g['it']), 'model_state': ms,
            'optimizer_state': {'state': state, 'param_groups': pgs}, 'scaler': None, 'version': str(g['version'])}


@pytest.mark.parametrize('spconv1', [False, True])
def test_reference_checkpoint_loads_with_optimizer(tmp_path, spconv1):
    """load_params_with_optimizer on a checkpoint of the reference loop: every tensor, (it, epoch), the Adam moments
    in the reference's group layout -- proven by taking the NEXT optimizer step and comparing with the reference's."""
    from tmae_amd.train import build_optimizer, build_scheduler
    model, g = build_tiny_tmae()
    f = tmp_path / 'checkpoint_epoch_1.pth'
    torch.save(reference_checkpoint(g, spconv1), f)
    opt = build_optimizer(model, _optim_cfg())
    groups = _group_names(model, opt)
    assert groups[0] == [str(n) for n in g['group0']] and groups[1] == [str(n) for n in g['group1']]
    it, epoch = model.load_params_with_optimizer(str(f), to_cpu=True, optimizer=opt)
    assert (it, epoch) == (2, 1)
    sd = model.state_dict()
    for i, n in enumerate(str(x) for x in g['state_names']):
        assert torch.equal(sd[n], torch.from_numpy(np.array(g[f'state_{i}']))), n
    sched, _ = build_scheduler(opt, 5, 2, -1, _optim_cfg())
    sched.step(it)
    assert abs(opt.lr - float(g['next_lr'])) < 1e-12 and abs(opt.mom - float(g['next_mom'])) < 1e-12
    opt.zero_grad()
    for p in model.parameters():
        p.grad = 0.01 * p.detach() + 0.001
    opt.step()
    for n, s, q in zip(g['next_names'], g['next_sum'], g['next_sumsq']):
        p = dict(model.named_parameters())[str(n)].detach().double()
        assert abs(float(p.sum()) - s) <= 1e-6 * max(1.0, abs(s)) + 2e-6 * p.numel() ** 0.5, n
        assert abs(float((p ** 2).sum()) - q) <= 2e-6 * max(1.0, q), n
    # round trip: our own checkpoint carries the same optimizer layout
    sd2 = opt.state_dict()
    assert [len(pg['params']) for pg in sd2['param_groups']] == [len(g['group0']), len(g['group1'])]


def test_foreign_optimizer_state_is_skipped_with_a_warning(tmp_path, capsys):
    from tmae_amd.train import build_optimizer
    model, g = build_tiny_tmae()
    ck = reference_checkpoint(g)
    ck['optimizer_state']['param_groups'] = ck['optimizer_state']['param_groups'][:1]      # e.g. a plain torch Adam
    f = tmp_path / 'c.pth'
    torch.save(ck, f)
    opt = build_optimizer(model, _optim_cfg())
    it, epoch = model.load_params_with_optimizer(str(f), to_cpu=True, optimizer=opt)
    assert (it, epoch) == (2, 1) and 'not loaded' in capsys.readouterr().out


@pytest.mark.gpu
@pytest.mark.parametrize('spconv1', [False, True])
def test_reference_checkpoint_forward_on_gpu(tmp_path, spconv1):
    """Weights of the reference checkpoint through load_params_from_file, then the reference's loss on the fixture's
    third batch (train mode, same masking noise): 1e-4."""
    model, g = build_tiny_tmae('cuda')
    f = tmp_path / 'c.pth'
    torch.save(reference_checkpoint(g, spconv1), f)
    model.load_params_from_file(str(f))
    model.train()
    dev = torch.device('cuda')
    bd = {'points': torch.from_numpy(g['points']).to(dev), 'points_prev': torch.from_numpy(g['points_prev']).to(dev),
          'batch_size': 2, 'mae_noise': torch.from_numpy(g['noise']).to(dev)}
    ret, _, _ = model(bd)
    assert abs(fl(ret['loss']) - float(g['loss'])) < 1e-4, (fl(ret['loss']), float(g['loss']))


@pytest.mark.gpu
def test_two_training_steps_follow_the_reference_loop():
    """The reference's first two training steps (forward, backward, OptimWrapper.step under OneCycle) replayed from
    its initial weights on the same batches and masking noise: both losses, and the weights after step 2 (Adam's
    first updates are +-lr * sign-like, so single elements with near-zero gradients may differ by up to 2 lr)."""
    from pcdet.models import model_fn_decorator
    from tmae_amd.train import build_optimizer, build_scheduler, train_one_step
    model, g = build_tiny_tmae('cuda')
    init = reference_checkpoint(g, prefix='init')['model_state']
    res = model.load_state_dict(init, strict=False)
    assert not res.unexpected_keys and not res.missing_keys
    model.train()
    opt = build_optimizer(model, _optim_cfg())
    sched, _ = build_scheduler(opt, 5, 2, -1, _optim_cfg())
    dev = torch.device('cuda')
    lrs = []
    for it in range(2):
        sched.step(it)
        lrs.append(opt.lr)
        bd = {'points': torch.from_numpy(g[f'b{it}_points']).to(dev), 'batch_size': 2,
              'points_prev': torch.from_numpy(g[f'b{it}_points_prev']).to(dev),
              'mae_noise': torch.from_numpy(g[f'b{it}_noise']).to(dev)}
        loss, _, _ = train_one_step(model, opt, sched, bd, it, model_fn_decorator(), amp_dtype=None)
        assert abs(fl(loss) - float(g['step_losses'][it])) < 1e-4, (it, fl(loss), float(g['step_losses'][it]))
    sd = model.state_dict()
    init = {str(n): torch.from_numpy(np.array(g[f'init_{i}'])) for i, n in enumerate(g['state_names'])}
    cosines, moved = {}, set(str(n) for n in g['group0']) | set(str(n) for n in g['group1'])
    for i, n in enumerate(str(x) for x in g['state_names']):
        ref = torch.from_numpy(np.array(g[f'state_{i}']))
        mine = sd[n].cpu()
        if not ref.is_floating_point():
            assert torch.equal(mine, ref), n                                  # num_batches_tracked, global_step
            continue
        d = (mine - ref).abs()
        if 'running_var' in n or 'running_mean' in n:
            # fp32 batch variances of features that carry absolute coordinates (|x| ~ 75 m) differ at ~1e-4 relative
            # between the CPU's and the GPU's summation orders
            assert d.max() <= 2e-3 * max(1.0, float(ref.abs().max())), n
        elif n in moved:
            # Adam's first two updates are ~ lr_t * sign(g) (the second up to ~1.4 lr_t): an element whose gradient
            # is rounding noise may flip in both (2 |update| each); everything else must follow the reference's update
            assert d.max() <= 2 * (lrs[0] + 1.5 * lrs[1]), (n, float(d.max()))
            du, dr = (mine - init[n]).flatten().double(), (ref - init[n]).flatten().double()
            cosines[n] = float(torch.nn.functional.cosine_similarity(du, dr, dim=0))
        else:
            assert torch.equal(mine, ref) and torch.equal(ref, init[n]), n     # in_proj / tau: never updated
    vals = sorted(cosines.values())
    worst = min(cosines, key=cosines.get)
    assert vals[len(vals) // 2] > 0.99 and vals[0] > 0.8, (worst, vals[0], vals[len(vals) // 2])


def test_skip_unread_gradients_leaves_the_trained_weights_unchanged():
    """OPTIMIZATION.SKIP_UNREAD_GRADIENTS (opt-in): parameters in no optimizer group stop receiving gradients; the
    owned parameters follow the same trajectory (no clipping, as in the reference's AMP branch)."""
    import copy
    import torch.nn as nn
    from tmae_amd.train.optim import AdamOneCycle

    class Attn(nn.Module):                       # a parameter next to a child module: owned by no leaf
        def __init__(self):
            super().__init__()
            self.in_proj = nn.Parameter(torch.randn(6, 6) * 0.3)
            self.out = nn.Linear(6, 6)

        def forward(self, x):
            return self.out(torch.tanh(x @ self.in_proj.t()))

    torch.manual_seed(0)
    a = nn.Sequential(Attn(), nn.BatchNorm1d(6), nn.Linear(6, 3))
    b = copy.deepcopy(a)
    oa, ob = AdamOneCycle(a, lr=1e-2), AdamOneCycle(b, lr=1e-2)
    assert len(oa.unoptimized) == 1
    ob.skip_unread_gradients()
    assert not b[0].in_proj.requires_grad and all(p.requires_grad for p in ob.params)
    x = torch.randn(16, 6)
    for _ in range(4):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m(x).square().mean().backward()
            o.step()
    assert a[0].in_proj.grad is not None and b[0].in_proj.grad is None
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
