"""Shapes and durations of the weight-gradient launches of one training step (bench.py's model and batch).
usage (GPU box): python profiles/scripts/op_shapes.py"""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
from pcdet.config import EasyDict, cfg_from_yaml_file
from pcdet.models import model_fn_decorator
from tmae_amd import ops
from tmae_amd._lib import lib
from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler,
                            train_one_step, wrap_ddp)

dev = torch.device('cuda', 0)
cfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml'), EasyDict())
ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=120000, batch_size=8, rank=0)
torch.manual_seed(0)
model = build_model_from_cfg(cfg, ds).to(dev).train()
ddp = wrap_ddp(model, 0)
opt = build_optimizer(model, cfg.OPTIMIZATION)
sched, _ = build_scheduler(opt, 1000, cfg.OPTIMIZATION.NUM_EPOCHS, -1, cfg.OPTIMIZATION)
b = ds.batch(0)
batch = {'points': torch.from_numpy(b['points']).to(dev), 'points_prev': torch.from_numpy(b['points_prev']).to(dev),
         'batch_size': b['batch_size']}
fn = model_fn_decorator()
for i in range(2):
    train_one_step(ddp, opt, sched, dict(batch), i, fn, amp_dtype=torch.bfloat16)
torch.cuda.synchronize()

log = []
orig_lw, orig_sw = ops.linear_wgrad, lib.tmae_spconv_wgrad
def lw(dy, x, want_bias=True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_lw(dy, x, want_bias); e1.record()
    log.append(('linear', dy.shape[0], dy.shape[1], x.shape[1], dy.stride(0), x.stride(0), e0, e1))
    return r
def sw(dy, ldy, f, ldf, nbr, m_out, cout, cin, dw, ws, wsb, st):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_sw(dy, ldy, f, ldf, nbr, m_out, cout, cin, dw, ws, wsb, st); e1.record()
    log.append(('spconv', m_out, cout, 9 * cin, ldy, ldf, e0, e1))
    return r
ops.linear_wgrad = lw
lib.tmae_spconv_wgrad = sw
train_one_step(ddp, opt, sched, dict(batch), 2, fn, amp_dtype=torch.bfloat16)
torch.cuda.synchronize()
tot = 0
for kind, m, n, k, ldy, ldx, e0, e1 in log:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    byt = m * (n + k) * 2
    print(f'{kind:7s} m={m:7d} n={n:4d} k={k:5d} ldy={ldy:5d} ldx={ldx:5d} {us:8.1f} us {byt / us / 1e3:7.0f} GB/s')
print('total', tot / 1e3, 'ms over', len(log), 'launches')
