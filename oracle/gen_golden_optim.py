"""Pins the optimizer and the checkpoint format against the UNMODIFIED reference
classes and writes the data fixtures tests/golden/O1_optimizer.npz and O2_checkpoint.npz.
TEST INFRASTRUCTURE, build container only (/root/reference does not travel).

Reference code run here on the CPU, loaded by path:
  tools/train_utils/optimization/fastai_optim.py        OptimWrapper (true_wd, bn_wd), split_bn_bias
  tools/train_utils/optimization/learning_schedules_fastai.py   OneCycle
  tools/train_utils/optimization/__init__.py            build_optimizer / build_scheduler (adam_onecycle branch)
  tools/train_utils/train_utils.py                      checkpoint_state
  pcdet/models/...                                      TemporalDynVFE, SiamWCA_MAE (oracle/ref_import.py),
                                                        SiamWCA, SSTBEVBackbone, CenterHead (gen_golden_finetune.py)

O1: (a) the names of the parameters in the reference optimizer's two param groups -- and of those in neither --
        for the 3-stage pre-training model and for the fine-tune model;
    (b) a 20-step parameter trajectory of a tiny model (Linear, BatchNorm1d, a module that owns parameters AND has a
        child, like nn.MultiheadAttention) under build_optimizer + build_scheduler in the order of train_one_epoch
        (train_utils.py:59-100: lr_scheduler.step(it); zero_grad; backward; step), with gradients given by the fixture.
O2: a small 1-stage T-MAE (d_model 64, 4 heads) trained for 2 real steps by the reference loop, then the dict
    that the reference's checkpoint_state returns (model_state, optimizer_state, epoch, it), the reference's loss
    on a third batch with the checkpoint's weights, and the per-tensor checksums of the parameters after one more
    optimizer step with the gradients g = 0.01 p + 0.001.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tmae_oracle as O           # noqa: E402
import ref_import as R            # noqa: E402
from gen_golden import save       # noqa: E402

TOOLS = '/root/reference/tools/train_utils'


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_training():
    R.load_reference()
    pkg = types.ModuleType('train_utils')
    pkg.__path__ = [TOOLS]
    sys.modules['train_utils'] = pkg
    sub = types.ModuleType('train_utils.optimization')
    sub.__path__ = [TOOLS + '/optimization']
    sys.modules['train_utils.optimization'] = sub
    _load('train_utils.optimization.fastai_optim', TOOLS + '/optimization/fastai_optim.py')
    _load('train_utils.optimization.learning_schedules_fastai', TOOLS + '/optimization/learning_schedules_fastai.py')
    opt = _load('train_utils.optimization', TOOLS + '/optimization/__init__.py')
    for absent in ('tqdm', 'tensorboardX', 'wandb'):          # imported at the top of train_utils.py, unused here
        if absent not in sys.modules:
            try:
                __import__(absent)
            except ImportError:
                m = types.ModuleType(absent)
                m.tqdm = m.trange = None
                sys.modules[absent] = m
    tu = _load('train_utils.train_utils', TOOLS + '/train_utils.py')
    return opt, tu


class Detector(nn.Module):
    """The part of Detector3DTemplate that shapes the state_dict / module order: children in module_topology order
    and the `global_step` buffer (detector3d_template.py:23,25-59)."""

    def __init__(self, **mods):
        super().__init__()
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        for k, v in mods.items():
            self.add_module(k, v)


def group_names(model, optimizer):
    names = {id(p): n for n, p in model.named_parameters()}
    groups = [[names[id(p)] for p in g['params']] for g in optimizer.opt.param_groups]
    owned = {n for g in groups for n in g}
    rest = [n for n, p in model.named_parameters() if n not in owned and p.requires_grad]
    return groups, rest


class TinyOwner(nn.Module):
    """Owns parameters and has a child, like nn.MultiheadAttention (in_proj_weight / out_proj)."""

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


def tiny_cfg():
    return R.AttrDict(dict(OPTIMIZER='adam_onecycle', LR=0.003, WEIGHT_DECAY=0.01, MOMENTUM=0.9, MOMS=[0.95, 0.85],
                           PCT_START=0.4, DIV_FACTOR=10, DECAY_STEP_LIST=[35, 45], LR_DECAY=0.1, LR_CLIP=1e-7,
                           LR_WARMUP=False, WARMUP_EPOCH=1, GRAD_NORM_CLIP=10))


def o1(optm):
    out = {}
    # ---- (a) param groups of the real models
    V, B, _ = R.build_reference_model(3)
    model = Detector(vfe=V, backbone_3d=B)
    opt = optm.build_optimizer(model, tiny_cfg())
    groups, rest = group_names(model, opt)
    assert len(groups) == 2
    out.update(pre_group0=np.array(groups[0]), pre_group1=np.array(groups[1]), pre_unoptimized=np.array(rest))
    print('pre-train model: groups', [len(g) for g in groups], 'unoptimized', len(rest),
          sum(p.numel() for n, p in model.named_parameters() if n in set(rest)))
    import gen_golden_finetune as GF
    Vf, B3, B2, H, _ = GF.build_finetune_reference(3)
    ft = Detector(vfe=Vf, backbone_3d=B3, backbone_2d=B2, dense_head=H)
    opt = optm.build_optimizer(ft, tiny_cfg())
    groups, rest = group_names(ft, opt)
    out.update(ft_group0=np.array(groups[0]), ft_group1=np.array(groups[1]), ft_unoptimized=np.array(rest))
    print('fine-tune model: groups', [len(g) for g in groups], 'unoptimized', len(rest))

    # ---- (b) tiny-model trajectory
    rng = np.random.default_rng(42)
    tiny = Tiny()
    init = {}
    with torch.no_grad():
        for n, p in tiny.named_parameters():
            v = rng.normal(0, 0.5, tuple(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(v))
            init[n] = v
    names = [n for n, _ in tiny.named_parameters()]
    total, steps = 20, 20
    cfg = tiny_cfg()
    opt = optm.build_optimizer(tiny, cfg)
    sched, _ = optm.build_scheduler(opt, total_iters_each_epoch=10, total_epochs=2, last_epoch=-1, optim_cfg=cfg)
    groups, rest = group_names(tiny, opt)
    assert rest == ['owner.own'], rest
    grads = {n: rng.normal(0, 1.0, (steps,) + tuple(init[n].shape)).astype(np.float32) for n in names}
    # step 7: fc2 gets no gradient (a parameter whose grad is None is still decayed, fastai_optim.py:139-150)
    traj = {n: [] for n in names}
    lrs, moms = [], []
    for it in range(steps):
        sched.step(it)
        lrs.append(float(opt.lr))
        moms.append(float(opt.mom))
        opt.zero_grad()
        for n, p in tiny.named_parameters():
            p.grad = None if (it == 7 and n == 'fc2.weight') else torch.from_numpy(grads[n][it].copy())
        opt.step()
        for n, p in tiny.named_parameters():
            traj[n].append(p.detach().numpy().copy())
    assert np.array_equal(traj['owner.own'][-1], init['owner.own'])           # never touched by the reference
    out.update(tiny_names=np.array(names), tiny_group0=np.array(groups[0]), tiny_group1=np.array(groups[1]),
               tiny_lr=np.array(lrs), tiny_mom=np.array(moms), tiny_total=total)
    for i, n in enumerate(names):
        out[f'tiny_init_{i}'] = init[n]
        out[f'tiny_grad_{i}'] = grads[n]
        out[f'tiny_traj_{i}'] = np.stack(traj[n])
    save('O1_optimizer', **out)


TINY_MODEL = dict(vfe_mlps=[[32, 64]], d_model=64, nhead=4, dff=128, num_blocks=1)


def tiny_tmae():
    ref = R.load_reference()
    cfg = R.reference_cfg(1)
    cfg.MODEL.VFE.MLPS = TINY_MODEL['vfe_mlps']
    b = cfg.MODEL.BACKBONE_3D
    enc = b.SST_BLOCK_LIST[0].ENCODER
    enc.D_MODEL, enc.NHEAD, enc.DIM_FEEDFORWARD, enc.NUM_BLOCKS = (TINY_MODEL['d_model'], TINY_MODEL['nhead'],
                                                                   TINY_MODEL['dff'], TINY_MODEL['num_blocks'])
    fl = b.FUSE_LAYER['x_conv1']
    fl.NUM_FILTER = fl.NUM_UPSAMPLE_FILTER = TINY_MODEL['d_model']
    torch.manual_seed(3)
    pcr = np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, dtype=np.float32)
    vs, grid = [0.32, 0.32, 8.0], np.array([468, 468, 1])
    V = ref['vfe'].TemporalDynVFE(cfg.MODEL.VFE, num_point_features=5, voxel_size=vs, point_cloud_range=pcr,
                                  grid_size=grid)
    B = ref['mae'].SiamWCA_MAE(b, input_channels=V.get_output_feature_dim(), grid_size=grid, voxel_size=vs,
                               point_cloud_range=pcr)
    model = Detector(vfe=V, backbone_3d=B)
    with torch.no_grad():                     # a conditioned head (as F10/F11) and a live temperature
        B.decoder_pred.weight.mul_(0.1)
        B.decoder_pred.bias.mul_(0.1)
        for n, p in model.named_parameters():
            if n.endswith('tau'):
                p.fill_(0.3)
    return model, cfg


def run_reference_step(model, pts, prv, bs, seed):
    bd = dict(points=torch.from_numpy(pts), points_prev=torch.from_numpy(prv), batch_size=bs)
    bd = model.vfe(bd)
    vc = bd['voxel_coords'].numpy()
    torch.manual_seed(seed)                   # random_masking draws torch.rand(1, L) per sample (common_utils.py:49-63)
    noise = np.concatenate([torch.rand(1, int((vc[:, 0] == b).sum())).numpy()[0] for b in range(bs)])
    torch.manual_seed(seed)
    bd = model.backbone_3d(bd)
    loss, _ = model.backbone_3d.get_loss()
    return loss, noise


def o2(optm, tu):
    model, _ = tiny_tmae()
    model.train()
    cfg = tiny_cfg()
    opt = optm.build_optimizer(model, cfg)
    sched, _ = optm.build_scheduler(opt, total_iters_each_epoch=5, total_epochs=2, last_epoch=-1, optim_cfg=cfg)
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    bs = 2
    it, losses, batches = 0, [], []
    for step in range(2):
        pts, prv = O.synth_frame_pair(2500, bs, seed=300 + step)
        sched.step(it)
        opt.zero_grad()
        loss, noise = run_reference_step(model, pts, prv, bs, seed=11 + step)
        loss.backward()
        # the AMP branch of the reference loop does not clip (train_utils.py:88-97); neither do we
        opt.step()
        model.global_step += 1
        it += 1
        losses.append(float(loss))
        batches.append((pts, prv, noise))
    import copy
    ckpt = copy.deepcopy(tu.checkpoint_state(model, opt, epoch=1, it=it))   # = what torch.save would freeze here
    assert set(ckpt.keys()) == {'epoch', 'it', 'model_state', 'optimizer_state', 'scaler', 'version'}
    out = dict(tiny_model=json.dumps(TINY_MODEL), epoch=ckpt['epoch'], it=ckpt['it'], version=str(ckpt['version']),
               step_losses=np.array(losses))
    names = list(ckpt['model_state'].keys())
    out['state_names'] = np.array(names)
    for i, n in enumerate(names):
        out[f'state_{i}'] = ckpt['model_state'][n].detach().cpu().numpy()
        out[f'init_{i}'] = init_state[n].numpy()
    osd = ckpt['optimizer_state']
    pgs = []
    for g in osd['param_groups']:
        pgs.append({k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in g.items()})
    out['opt_param_groups'] = json.dumps(pgs)
    out['opt_state_ids'] = np.array(sorted(osd['state'].keys()))
    for idx, st in osd['state'].items():
        out[f'opt_{idx}_step'] = np.asarray(float(st['step']))
        out[f'opt_{idx}_exp_avg'] = st['exp_avg'].numpy()
        out[f'opt_{idx}_exp_avg_sq'] = st['exp_avg_sq'].numpy()
    groups, rest = group_names(model, opt)
    out.update(group0=np.array(groups[0]), group1=np.array(groups[1]), unoptimized=np.array(rest))
    for i, (pts, prv, noise) in enumerate(batches):
        out[f'b{i}_points'], out[f'b{i}_points_prev'], out[f'b{i}_noise'] = pts, prv, noise
    # the loss of a third batch with the checkpoint's weights (train mode: batch statistics)
    pts, prv = O.synth_frame_pair(2500, bs, seed=302)
    loss3, noise3 = run_reference_step(model, pts, prv, bs, seed=13)
    out.update(points=pts, points_prev=prv, noise=noise3, loss=float(loss3))
    # one more optimizer step of the reference with synthetic gradients g = 0.01 p + 0.001
    sched.step(it)
    opt.zero_grad()
    for p in model.parameters():
        p.grad = (0.01 * p.detach() + 0.001)
    opt.step()
    pn = [n for n, _ in model.named_parameters()]
    out['next_names'] = np.array(pn)
    out['next_sum'] = np.array([float(p.detach().double().sum()) for _, p in model.named_parameters()])
    out['next_sumsq'] = np.array([float((p.detach().double() ** 2).sum()) for _, p in model.named_parameters()])
    out['next_lr'], out['next_mom'] = float(opt.lr), float(opt.mom)
    save('O2_checkpoint', **out)
    print('O2: losses', losses, 'loss3', float(loss3), 'params', sum(p.numel() for p in model.parameters()))


def main():
    optm, tu = load_reference_training()
    o1(optm)
    o2(optm, tu)


if __name__ == '__main__':
    main()
