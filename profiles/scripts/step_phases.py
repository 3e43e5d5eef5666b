"""Host-side timeline of the training loop, steady state (no sync between steps, 4 batches cycled like bench.py): per step the
wall time the host spends in forward / backward / optimizer, and how far the GPU is behind the host when each phase has been
enqueued.  python3 profiles/scripts/step_phases.py [steps]"""
import os, sys, time, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
from pcdet.config import EasyDict, cfg_from_yaml_file
from pcdet.models import model_fn_decorator
from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler)
from tmae_amd import ops
dev = torch.device('cuda', 0)
cfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml'), EasyDict())
ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=120000, batch_size=8, rank=0)
torch.manual_seed(0)
model = build_model_from_cfg(cfg, ds).to(dev).train()
opt = build_optimizer(model, cfg.OPTIMIZATION)
sched, _ = build_scheduler(opt, 1000, cfg.OPTIMIZATION.NUM_EPOCHS, -1, cfg.OPTIMIZATION)
batches = []
for i in range(4):
    b = ds.batch(i)
    batches.append({'points': torch.from_numpy(b['points']).to(dev), 'points_prev': torch.from_numpy(b['points_prev']).to(dev), 'batch_size': b['batch_size']})
fn = model_fn_decorator()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25
rec = []


def one(i, timed):
    t0 = time.perf_counter()
    sched.step(i)
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss, _, _ = fn(model, dict(batches[i % 4]))
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step(copy_dtype=torch.bfloat16)
    ops.refresh_param_copies(opt.params, torch.bfloat16)
    t3 = time.perf_counter()
    if timed:
        rec.append((t1 - t0, t2 - t1, t3 - t2, t0))
    return loss if os.environ.get('HOLD_LOSS') else None


for i in range(int(os.environ.get('SP_WARM', '5'))):
    one(i, False)
    if os.environ.get('SP_SYNC'):
        torch.cuda.synchronize()
torch.cuda.synchronize()
ta = time.perf_counter()
for i in range(N):
    held = one(5 + i, True)
tb = time.perf_counter()
torch.cuda.synchronize()
tc = time.perf_counter()
f = sum(r[0] for r in rec) / N * 1e3
b = sum(r[1] for r in rec) / N * 1e3
o = sum(r[2] for r in rec) / N * 1e3
print('host step durations ms:', [round((b_[3] - a_[3]) * 1e3, 1) for a_, b_ in zip(rec, rec[1:])])
print(f'{(tc - ta) / N * 1e3:.1f} ms/step; host wall per step: forward {f:.1f} ms (incl. its two host syncs), backward {b:.1f} ms, '
      f'optimizer {o:.1f} ms; GPU behind the host at the end of the loop: {(tc - tb) * 1e3:.1f} ms')
