"""Which aten ops (with input shapes) one training step launches: torch.profiler, one step, grouped by op + shapes.
python3 profiles/scripts/torch_ops_census.py [finetune] > out.txt"""
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    train_one_step(model, opt, sched, dict(batch), 3, fn, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize()
LAUNCHING = ('aten::fill_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::mul_', 'aten::sum', 'aten::cat',
             'aten::index', 'aten::gelu', 'aten::mm', 'aten::addmm', 'aten::addmm_', 'aten::neg', 'aten::div', 'aten::sub',
             'aten::where', 'aten::gt', 'aten::lt', 'aten::ge', 'aten::eq', 'aten::flip', 'aten::cumsum', 'aten::sort',
             'aten::index_select', 'aten::gather', 'aten::scatter_', 'aten::clamp', 'aten::clamp_min', 'aten::sqrt',
             'aten::_foreach_copy_', 'aten::_foreach_mul_', 'aten::_foreach_add_', 'aten::_fused_adam_', 'aten::mean',
             'aten::div_', 'aten::sub_', 'aten::zero_', 'aten::bitwise_and', 'aten::any', 'aten::all', 'aten::max', 'aten::min',
             'aten::nonzero', 'aten::masked_fill_', 'aten::reciprocal', 'aten::rsqrt', 'aten::exp', 'aten::abs', 'aten::linalg_vector_norm',
             'aten::_local_scalar_dense', 'aten::item', 'aten::arange', 'aten::stack', 'aten::bmm', 'aten::matmul', 'aten::linear')
cnt = collections.Counter()
for ev in prof.events():
    if ev.name not in LAUNCHING:
        continue
    where = 'autograd/other'
    for fr in ev.stack:
        if ('tmae_amd' in fr or 'pcdet' in fr or 'bench.py' in fr) and 'profiler' not in fr:
            where = fr.replace(ROOT, '')
            break
    shp = str(ev.input_shapes)[:60]
    cnt[(ev.name, where, shp)] += 1
for (name, where, shp), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:150]:
    print(f'{c:4d}  {name:24s} {where:90s} {shp}')
