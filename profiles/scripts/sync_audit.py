"""Which torch calls of one training step synchronise the host with the GPU?  torch.cuda.set_sync_debug_mode('warn') for
one step of the pre-training (default) or fine-tune ('finetune') configuration; prints the distinct call sites.
usage (GPU box): python profiles/scripts/sync_audit.py [finetune]"""
import os, sys, warnings, collections, traceback
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
import torch
from pcdet.config import EasyDict, cfg_from_yaml_file
from pcdet.models import model_fn_decorator
from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler, train_one_step)
FT = len(sys.argv) > 1 and sys.argv[1] == 'finetune'
dev = torch.device('cuda', 0)
cfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae.yaml' if FT else 't_mae_ssl.yaml'), EasyDict())
ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=120000, batch_size=8, rank=0, n_boxes=40 if FT else 0)
torch.manual_seed(0)
model = build_model_from_cfg(cfg, ds).to(dev).train()
opt = build_optimizer(model, cfg.OPTIMIZATION)
sched, _ = build_scheduler(opt, 1000, cfg.OPTIMIZATION.NUM_EPOCHS, -1, cfg.OPTIMIZATION)
b = ds.batch(0)
batch = {'points': torch.from_numpy(b['points']).to(dev), 'points_prev': torch.from_numpy(b['points_prev']).to(dev), 'batch_size': b['batch_size']}
if 'gt_boxes' in b:
    batch['gt_boxes'] = torch.from_numpy(b['gt_boxes']).to(dev)
fn = model_fn_decorator()
for i in range(3):
    train_one_step(model, opt, sched, dict(batch), i, fn, amp_dtype=torch.bfloat16)
torch.cuda.synchronize()
sites = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' in str(message).lower():
        st = [f for f in traceback.extract_stack() if ('tmae_amd' in f.filename or 'pcdet' in f.filename) and 'sync_audit' not in f.filename]
        full = traceback.extract_stack()
        where = (f'{os.path.relpath(st[-1].filename, ROOT)}:{st[-1].lineno} {st[-1].line}' if st else
                 'no product frame: ' + ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in full[-6:-1]))
        sites[where] += 1
warnings.showwarning = hook
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
train_one_step(model, opt, sched, dict(batch), 3, fn, amp_dtype=torch.bfloat16)
torch.cuda.set_sync_debug_mode('default')
torch.cuda.synchronize()
print(f'{sum(sites.values())} synchronising calls in one {"fine-tune" if FT else "pre-training"} step:')
for k, v in sites.most_common():
    print(f'{v:3d}  {k}')
