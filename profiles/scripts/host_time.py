import os, sys, time, torch
ROOT=os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
from pcdet.config import EasyDict, cfg_from_yaml_file
from pcdet.models import model_fn_decorator
from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler, train_one_step, wrap_ddp)
dev = torch.device('cuda', 0)
cfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml'), EasyDict())
ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=120000, batch_size=8, rank=0)
torch.manual_seed(0)
model = build_model_from_cfg(cfg, ds).to(dev).train()
opt = build_optimizer(model, cfg.OPTIMIZATION)
sched, _ = build_scheduler(opt, 1000, cfg.OPTIMIZATION.NUM_EPOCHS, -1, cfg.OPTIMIZATION)
b = ds.batch(0)
batch = {'points': torch.from_numpy(b['points']).to(dev), 'points_prev': torch.from_numpy(b['points_prev']).to(dev), 'batch_size': b['batch_size']}
fn = model_fn_decorator()
for i in range(3):
    train_one_step(model, opt, sched, dict(batch), i, fn, amp_dtype=torch.bfloat16)
torch.cuda.synchronize()
# wall per step with full sync vs host enqueue time
for rep in range(3):
    c0 = time.process_time(); th0 = time.thread_time()
    t0 = time.perf_counter()
    train_one_step(model, opt, sched, dict(batch), 3, fn, amp_dtype=torch.bfloat16)
    t1 = time.perf_counter()
    print(f'cpu time: process {1e3*(time.process_time()-c0):.1f} ms, main thread {1e3*(time.thread_time()-th0):.1f} ms')
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host enqueue {1e3*(t1-t0):.1f} ms, +wait {1e3*(t2-t1):.1f} ms, total {1e3*(t2-t0):.1f} ms')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
train_one_step(model, opt, sched, dict(batch), 3, fn, amp_dtype=torch.bfloat16)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
