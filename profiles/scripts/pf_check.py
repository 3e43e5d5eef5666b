import os, sys, numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for p in ('t-mae_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import tmae_oracle as O
from conftest import build_product_model
import tmae_amd.modules.sst as sst
dev = torch.device('cuda:0')
pts, prv = O.synth_frame_pair(120000, 1, seed=0)
cfg = O.default_model_cfg(3)
vox = O.voxelize(pts, cfg['point_cloud_range'], cfg['voxel_size'], cfg['grid_size'])
noise = np.random.default_rng(0).random(vox['voxel_coords'].shape[0]).astype(np.float32)
P = O.init_params(cfg, seed=0, pred_scale=0.1)
model, _, _ = build_product_model(3, params=P, device=dev, batch_size=1)
model.train()
def cu(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def run(amp, fold):
    sst._POS_FOLD = fold
    bd = {'points': cu(pts), 'points_prev': cu(prv), 'batch_size': 1, 'mae_noise': cu(noise)}
    model.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        ret, _, _ = model(bd)
    ret['loss'].backward()
    return float(ret['loss']), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}
l32, g32 = run(False, True)
l16f, g16f = run(True, True)
l16o, g16o = run(True, False)
print('loss fp32', l32, 'bf16 fold', l16f, 'bf16 nofold', l16o)
def cmp(a, b):
    a, b = a.reshape(-1), b.reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))
for n in g32:
    if 'in_proj' in n or n.endswith('linear2.weight') and 'sst_blocks.2.encoder_blocks.1.encoder_list.1' in n:
        print(n.replace('backbone_3d.', ''), 'fold-vs-fp32 cos/rel %.4f %.4f' % cmp(g16f[n], g32[n]), '| nofold-vs-fp32 %.4f %.4f' % cmp(g16o[n], g32[n]),
              '| fold-vs-nofold %.4f %.4f' % cmp(g16f[n], g16o[n]))
