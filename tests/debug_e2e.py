"""Diagnostic (not a test): per-stage differences between the HIP product path and the CPU oracle."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import build_product_model, golden
import tmae_oracle as O

name, nst = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('F11_e2e_1stage', 1)
g = golden(name)
cfg = O.default_model_cfg(nst)
P = O.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']),
                       pred_scale=float(g['pred_scale']) if 'pred_scale' in g.files else 1.0)
bs = int(g['batch_size'])
dev = torch.device('cuda:0')
model, _, _ = build_product_model(nst, params=P, device=dev)
model.train()
cap = {}
lo = O.forward_loss(P, g['points'], g['points_prev'], g['noise'], bs, cfg, cap)
bd = {'points': torch.from_numpy(g['points']).to(dev), 'points_prev': torch.from_numpy(g['points_prev']).to(dev),
      'batch_size': bs, 'mae_noise': torch.from_numpy(g['noise']).to(dev)}
with torch.no_grad():
    bd = model.vfe(bd)
    d = lambda a, b: float((a.detach().float().cpu() - b.detach().float()).abs().max())
    print('vfe cur', d(bd['voxel_features'], cap['vfe_cur']['voxel_features']), 'prev', d(bd['voxel_features_prev'], cap['vfe_prv']['voxel_features']))
    bb = model.backbone_3d
    fp, sp_ = bb.sparse_encode(bd['voxel_features_prev'], bd['voxel_coords_prev'], bs, True)
    for i in range(nst):
        print('prev stage', i, d(fp[f'x_conv{i+1}'].features, cap[f'prev_stage{i}']['features']),
              'idx equal', np.array_equal(fp[f'x_conv{i+1}'].indices.cpu().numpy(), cap[f'prev_stage{i}']['indices']))
    vf, vc, mask = bb.mask_voxels(bd['voxel_features'], bd['voxel_coords'], bs, bd['voxels_per_sample'], bd['mae_noise'])
    print('mask equal', np.array_equal(mask.cpu().numpy(), cap['mask']))
    fc, st = bb.sparse_encode(vf, vc, bs)
    for i in range(nst):
        print('cur stage', i, d(fc[f'x_conv{i+1}'].features, cap[f'cur_stage{i}']['features']))
    fc = bb.sparse_cross_attn(fc, fp)
    for i in range(nst):
        print('wca stage', i, d(fc[f'x_conv{i+1}'].features, cap[f'wca_stage{i}']['features']))
    spatial, _ = bb.dense_conv(fc, st)
    print('spatial', d(spatial, cap['spatial_features']), 'max', float(cap['spatial_features'].abs().max()))
    # decoder pieces separately (fp32 conv algorithms)
    x = fc['x_conv1'].dense()
    xo = O.to_dense(cap['wca_stage0']['features'], cap['cur_stage0']['indices'], cap['cur_stage0']['shape'], bs)
    print('dense()', d(x, xo))
    y = bb.decoder_deblocks[0][0](x)
    yo = torch.nn.functional.conv_transpose2d(xo, P['backbone_3d.decoder_deblocks.0.0.weight'], stride=1)
    print('deconv', d(y, yo), 'max', float(yo.abs().max()))
