"""Which aten ops (with input shapes) one training step launches: torch.profiler, one step, grouped by op + shapes.
scratch [finetune] > out.txt"""
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
for i in range(3):
    train_one_step(model, opt, sched, dict(batch), i, fn, amp_dtype=torch.bfloat16)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_one_step(model, opt, sched, dict(batch), 3, fn, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize()
rows = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    own = getattr(ev, 'self_device_time_total', 0)
    if own <= 0 or not ev.name.startswith('aten::'):
        continue
    key = (ev.name, str(ev.input_shapes)[:100])
    rows[key] += 1; dur[key] += own
print(f'{sum(rows.values())} framework operators with device work, {sum(dur.values())/1e3:.2f} ms')
for key, n in sorted(rows.items(), key=lambda kv: -dur[kv[0]])[:60]:
    print(f'{n:4d} x {dur[key]/n:8.1f} us  {key[0]:28s} {key[1]}')
