"""Which tensors of ONE fp32 forward + backward differ between two runs from identical state?  (answers: which ops are not
run-to-run reproducible on the fp32 path)"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
import numpy as np, torch
from pcdet.config import EasyDict, cfg_from_yaml_file
from pcdet.models import model_fn_decorator
from tmae_amd.train import SyntheticTemporalDataset, build_model_from_cfg
dev = torch.device('cuda', 0)
cfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml'), EasyDict())
ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=20000, batch_size=2)
b = ds.batch(0)
batch = {'points': torch.from_numpy(b['points']).to(dev), 'points_prev': torch.from_numpy(b['points_prev']).to(dev), 'batch_size': b['batch_size']}
torch.manual_seed(1234)
model = build_model_from_cfg(cfg, ds).to(dev).train()
mf = model_fn_decorator()
for amp in (None, torch.bfloat16):
    runs = []
    for r in range(3):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(99)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp is not None):
            ret, tb, disp = mf(model, dict(batch))
        loss = ret.loss if hasattr(ret, 'loss') else ret[0] if isinstance(ret, (tuple, list)) else ret
        loss = loss.mean()
        loss.backward()
        runs.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    print('dtype', 'fp32' if amp is None else 'bf16', 'losses', [f'{l:.9f}' for l, _ in runs])
    bad = [(n, float((runs[0][1][n] - runs[k][1][n]).abs().max() / (runs[0][1][n].abs().max() + 1e-30))) for n in runs[0][1] for k in (1, 2)
           if not torch.equal(runs[0][1][n], runs[k][1][n])]
    names = sorted({n for n, _ in bad})
    print(len(names), 'of', len(runs[0][1]), 'gradients differ between runs')
    for n in names[:60]:
        print('   ', n, max(v for m, v in bad if m == n))
