"""Pins the oracle's TOKEN-DROPPING path against the reference and writes tests/golden/F13_e2e_dropping.npz.

The shipped YAMLs never drop a token (8 x 8 windows, top level max_tokens 64: SURVEY A-6); DROP_INFO with smaller
max_tokens is a pure YAML edit the reference accepts: voxels whose in-window rank reaches their level's max_tokens skip
the encoder of an SST block (they keep the residual, spt_backbone.py:47-135,342-353) and, per shift, the window
cross-attention (SiamWCA.py:65-215).  Here the UNMODIFIED reference VFE + SiamWCA_MAE (2 stages) run on CPU with
DROP_INFO = {0: 4 tokens for windows of < 4 voxels, 1: 8 for < 8, 2: 12 for the rest} on a dense small cloud (so that many
windows lose tokens), with the deterministic stable-rank stand-in for the CUDA-only `get_inner_win_inds` (canonical
form, SURVEY A-5); the oracle must reproduce loss, mask, predictions and gradient norms."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R                  # noqa: E402
import tmae_oracle as O                 # noqa: E402
from gen_golden import save, check      # noqa: E402

DROP = {0: dict(max_tokens=4, drop_range=(0, 4)), 1: dict(max_tokens=8, drop_range=(4, 8)),
        2: dict(max_tokens=12, drop_range=(8, 100000))}


def main():
    nst, npts, bs = 2, 9000, 2
    ref = R.load_reference()
    cfg = R.reference_cfg(nst)
    for blk in cfg.MODEL.BACKBONE_3D.SST_BLOCK_LIST:
        for mode in ('train', 'test'):
            blk.PREPROCESS.DROP_INFO[mode] = R.AttrDict({str(k): {'max_tokens': v['max_tokens'], 'drop_range': list(v['drop_range'])}
                                                         for k, v in DROP.items()})
    torch.manual_seed(0)
    pcr = np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, dtype=np.float32)
    vs, grid = [0.32, 0.32, 8.0], np.array([468, 468, 1])
    Vn = ref['vfe'].TemporalDynVFE(cfg.MODEL.VFE, num_point_features=5, voxel_size=vs, point_cloud_range=pcr, grid_size=grid)
    Bn = ref['mae'].SiamWCA_MAE(cfg.MODEL.BACKBONE_3D, input_channels=Vn.get_output_feature_dim(), grid_size=grid,
                                voxel_size=vs, point_cloud_range=pcr)
    c1 = O.default_model_cfg(nst)
    c1['drop_info'] = dict(DROP)
    P = O.init_params(c1, seed=9, tau=0.2, pred_scale=0.1)
    res = Vn.load_state_dict({k[4:]: v for k, v in P.items() if k.startswith('vfe.')}, strict=False)
    assert not res.unexpected_keys
    res = Bn.load_state_dict({k[12:]: v for k, v in P.items() if k.startswith('backbone_3d.')}, strict=False)
    assert not res.unexpected_keys and all('running' in k or 'num_batches' in k for k in res.missing_keys), res
    Vn.train(), Bn.train()
    # a dense cloud: every point within ~25 m of the sensor, so that windows hold up to ~40 voxels
    rng = np.random.default_rng(4)
    cur, prv = [], []
    for b in range(bs):
        r = rng.uniform(2.5, 25.0, npts)
        th = rng.uniform(0, 2 * np.pi, npts)
        pts = np.stack([np.full(npts, b), r * np.cos(th), r * np.sin(th), rng.normal(-1.6, 0.3, npts), rng.uniform(0, 1, npts)],
                       1).astype(np.float32)
        cur.append(pts)
        q = pts[rng.random(npts) < 0.9].copy()
        q[:, 1] += np.float32(0.4)
        q[:, 2] -= np.float32(0.15)
        prv.append(q)
    pts, pts_prev = np.concatenate(cur), np.concatenate(prv)
    bd = dict(points=torch.from_numpy(pts), points_prev=torch.from_numpy(pts_prev), batch_size=bs)
    bd = Vn(bd)
    vc = bd['voxel_coords'].numpy()
    torch.manual_seed(5)
    noise = np.concatenate([torch.rand(1, int((vc[:, 0] == b).sum())).numpy()[0] for b in range(bs)])
    torch.manual_seed(5)
    bd = Bn(bd)
    loss, _ = Bn.get_loss()
    loss.backward()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    cap = {}
    ol = O.forward_loss(Pg, pts, pts_prev, noise, bs, c1, cap)
    ol.backward()
    check('loss', ol, loss, 1e-5)
    check('mask', cap['mask'], bd['voxel_mae_mask'])
    check('pred', cap['pred_points'], Bn.forward_ret_dict['pred_points'], 1e-4)
    gn = {}
    for mod, pre in ((Vn, 'vfe.'), (Bn, 'backbone_3d.')):
        for n_, p_ in mod.named_parameters():
            g_ref, g_or = p_.grad, Pg[pre + n_].grad
            assert g_ref is not None and g_or is not None, n_
            check('grad ' + n_, g_or, g_ref, 2e-3 * max(1.0, float(g_ref.abs().max())))
            gn[pre + n_] = float(g_ref.norm())
    # how much was dropped (the fixture is only worth something if tokens really go)
    info = O.sst_input_layer(cap['prev_stage0']['coords4'] if 'coords4' in cap.get('prev_stage0', {}) else
                             np.concatenate([vc[:, :1], np.zeros((len(vc), 1), np.int64), vc[:, 2:]], 1), (468, 468, 1), c1)
    dropped = len(vc) - len(info['voxel_keep_inds'])
    print(f'stage-1 encoder of the (unmasked) current frame would drop {dropped} of {len(vc)} voxels')
    assert dropped > 0.05 * len(vc)
    save('F13_e2e_dropping', n_points=npts, batch_size=bs, num_stages=nst, param_seed=9, tau=np.float32(0.2),
         pred_scale=np.float32(0.1), drop_max_tokens=np.array([v['max_tokens'] for v in DROP.values()]),
         drop_lower=np.array([v['drop_range'][0] for v in DROP.values()]),
         drop_upper=np.array([v['drop_range'][1] for v in DROP.values()]),
         points=pts, points_prev=pts_prev, noise=noise, loss=loss.detach(), mask=bd['voxel_mae_mask'],
         pred_points=Bn.forward_ret_dict['pred_points'].detach(), grad_names=np.array(list(gn.keys())),
         grad_norms=np.array(list(gn.values())), dropped_stage1_unmasked=np.int64(dropped),
         spatial_checksum=bd['spatial_features'].detach().double().sum(),
         spatial_abs_checksum=bd['spatial_features'].detach().double().abs().sum())
    print('dropping fixture written; the oracle equals the reference with tokens dropped')


if __name__ == '__main__':
    main()
