"""Which Python call sites launch the small aten kernels of one training step (fills, copies, adds, ...): a TorchDispatchMode
records every aten op with the innermost repository frame of the calling thread (forward and the Python parts of backward).
python3 profiles/scripts/small_ops_census.py [finetune] > gpurun_out/census.txt"""
import collections, os, sys, traceback, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
from torch.utils._python_dispatch import TorchDispatchMode
from pcdet.config import EasyDict, cfg_from_yaml_file
from pcdet.models import model_fn_decorator
from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler, train_one_step)
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
SKIP = ('aten::view', 'aten::_unsafe_view', 'aten::empty', 'aten::as_strided', 'aten::detach', 'aten::alias', 'aten::t', 'aten::transpose',
        'aten::permute', 'aten::slice', 'aten::select', 'aten::unsqueeze', 'aten::squeeze', 'aten::expand', 'aten::reshape', 'aten::_reshape_alias',
        'aten::empty_like', 'aten::empty_strided', 'aten::new_empty', 'aten::split', 'aten::unbind', 'aten::narrow', 'aten::lift_fresh', 'aten::result_type')
cnt = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if not name.startswith(SKIP):
            where = '?'
            for fr in reversed(traceback.extract_stack()[:-1]):
                if ('tmae_amd' in fr.filename or 'pcdet' in fr.filename) and 'small_ops_census' not in fr.filename:
                    where = f'{fr.filename.replace(ROOT, "")}:{fr.lineno} {fr.name}'
                    break
            shp = ','.join(str(tuple(a.shape)) for a in args if isinstance(a, torch.Tensor))[:48]
            dt = next((str(a.dtype).replace('torch.', '') for a in args if isinstance(a, torch.Tensor)), '')
            cnt[(name, where, shp, dt)] += 1
        return func(*args, **(kwargs or {}))


with Rec():
    train_one_step(model, opt, sched, dict(batch), 3, fn, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize()
for (name, where, shp, dt), c in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0]))[:220]:
    print(f'{c:4d}  {name:28s} {dt:9s} {shp:50s} {where}')
