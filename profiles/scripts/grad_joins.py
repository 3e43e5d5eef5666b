"""Where the autograd engine itself adds gradients: graph nodes whose output feeds >= 2 consumers get an accumulation add
([m, d] elementwise pass, no Python frame in any profile).  Walks the graph of one forward and prints those joins with the
gradient's shape and the consumer nodes.   python3 profiles/scripts/grad_joins.py [finetune] > gpurun_out/joins.txt"""
import collections, os, sys, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
from pcdet.config import EasyDict, cfg_from_yaml_file
from pcdet.models import model_fn_decorator
from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler, train_one_step)
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
FT = len(sys.argv) > 1 and sys.argv[1] == 'finetune'
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
for i in range(2):
    train_one_step(model, opt, sched, dict(batch), i, fn, amp_dtype=torch.bfloat16)
model.train()
with torch.autocast('cuda', dtype=torch.bfloat16):
    loss = fn(model, dict(batch)).loss
seen, edges, stack = set(), collections.defaultdict(list), [loss.grad_fn]
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    for (nxt, idx) in n.next_functions:
        if nxt is not None:
            edges[(nxt, idx)].append(type(n).__name__)
            stack.append(nxt)
rows = []
for (n, idx), users in edges.items():
    if len(users) < 2 or type(n).__name__ == 'AccumulateGrad':
        continue
    try:
        shp = tuple(n._input_metadata[idx].shape)
    except Exception:
        shp = ()
    numel = 1
    for s in shp:
        numel *= int(s)
    rows.append((numel, shp, type(n).__name__, idx, users))
for numel, shp, name, idx, users in sorted(rows, key=lambda r: -r[0]):
    print(f'{str(shp):22s} output {idx} of {name:40s} <- {users}')
