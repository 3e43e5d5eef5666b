"""Pins oracle/finetune_oracle.py against the UNMODIFIED reference modules of the fine-tune path and writes the
data fixtures tests/golden/G*.npz (build container only; /root/reference does not travel).

Reference modules run here on CPU: CenterHead (assign_targets, forward, get_loss), SSTBEVBackbone, SiamWCA,
TemporalDynVFE, loss_utils.{FocalLossCenterNet, RegLossCenterNet}, centernet_utils.{gaussian_radius,
draw_gaussian_to_heatmap}.  Stand-ins only for what is absent in the image: numba (decorator), the CUDA extension
packages iou3d_nms / roiaware_pool3d (never called on the training path) and Tensor.cuda (the head's constructor
moves a class-id table to the GPU, center_head.py:62-66).
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tmae_oracle as O           # noqa: E402
import finetune_oracle as FO      # noqa: E402
import ref_import as R            # noqa: E402
from gen_golden import save, check  # noqa: E402


def load_finetune_reference():
    ref = R.load_reference()
    for name, sub in [('pcdet.models.backbones_2d', '/models/backbones_2d'), ('pcdet.models.dense_heads', '/models/dense_heads'),
                      ('pcdet.ops.iou3d_nms', '/ops/iou3d_nms'), ('pcdet.ops.roiaware_pool3d', '/ops/roiaware_pool3d')]:
        m = types.ModuleType(name)
        m.__path__ = [R.REF + sub]
        sys.modules[name] = m
    nb = types.ModuleType('numba')
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules['numba'] = nb
    sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_utils'] = types.ModuleType('pcdet.ops.iou3d_nms.iou3d_nms_utils')
    sys.modules['pcdet.ops.iou3d_nms'].iou3d_nms_utils = sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_utils']
    sys.modules['pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils'] = types.ModuleType('pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils')
    sys.modules['pcdet.ops.roiaware_pool3d'].roiaware_pool3d_utils = sys.modules['pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils']
    ref.update(
        bev=importlib.import_module('pcdet.models.backbones_2d.sst_bev_backbone'),
        head=importlib.import_module('pcdet.models.dense_heads.center_head'),
        loss_utils=importlib.import_module('pcdet.utils.loss_utils'),
        centernet_utils=importlib.import_module('pcdet.models.model_utils.centernet_utils'),
    )
    return ref


def build_finetune_reference(num_stages=3, seed=0):
    ref = load_finetune_reference()
    with open('/root/reference/tools/cfgs/once_models/t_mae.yaml') as f:
        cfg = R.AttrDict(yaml.safe_load(f))
    if num_stages < 3:
        b = cfg.MODEL.BACKBONE_3D
        b.SST_BLOCK_LIST = b.SST_BLOCK_LIST[:num_stages]
        b.FEATURES_SOURCE = b.FEATURES_SOURCE[:num_stages]
    torch.manual_seed(seed)
    pcr = np.array([-74.88, -74.88, -5.0, 74.88, 74.88, 3.0], dtype=np.float32)
    vs = [0.32, 0.32, 8.0]
    grid = np.array([468, 468, 1])
    V = ref['vfe'].TemporalDynVFE(cfg.MODEL.VFE, num_point_features=5, voxel_size=vs, point_cloud_range=pcr, grid_size=grid)
    B3 = ref['siam'].SiamWCA(cfg.MODEL.BACKBONE_3D, input_channels=V.get_output_feature_dim(), grid_size=grid,
                             voxel_size=vs, point_cloud_range=pcr)
    B2 = ref['bev'].SSTBEVBackbone(cfg.MODEL.BACKBONE_2D, input_channels=B3.num_point_features)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        H = ref['head'].CenterHead(cfg.MODEL.DENSE_HEAD, input_channels=B2.num_bev_features, num_class=5,
                                   class_names=FO.CLASS_NAMES, grid_size=grid, point_cloud_range=pcr, voxel_size=vs,
                                   predict_boxes_when_training=False)
    finally:
        torch.Tensor.cuda = orig_cuda
    return V, B3, B2, H, cfg


def load_into(mod, P, prefix):
    sd = {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}
    res = mod.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all('running' in k or 'num_batches' in k for k in res.missing_keys), res.missing_keys


def main():
    ref = load_finetune_reference()
    cu = ref['centernet_utils']
    cfg = FO.default_finetune_cfg(3)

    # ---- G1: gaussian radius + heatmap drawing + target assignment, incl. boxes on the border / degenerate / padded
    rng = np.random.default_rng(4)
    gt = FO.synth_gt_boxes(3, 40, seed=2)
    gt[0, 0, :2] = [74.87, -74.87]             # corner cell
    gt[0, 1, :2] = [80.0, 0.0]                 # outside the range: clamped to the last cell
    gt[1, 2, 3] = 0.0                          # dx = 0: skipped
    gt[2, 3, :2] = [-74.88, 10.0]              # exactly on the range minimum
    V, B3, B2, H, rcfg = build_finetune_reference(3, seed=0)
    gtt = torch.from_numpy(gt)
    tgt_ref = H.assign_targets(gtt.clone(), feature_map_size=torch.Size([468, 468]))
    tgt = FO.assign_targets(gtt.clone(), (468, 468), cfg)
    for k in ('heatmaps', 'target_boxes', 'inds', 'masks'):
        check('targets ' + k, tgt[k][0], tgt_ref[k][0], 0.0)
    h, w = torch.rand(50) * 30 + 0.1, torch.rand(50) * 30 + 0.1
    check('gaussian_radius', FO.gaussian_radius(h, w, 0.1), cu.gaussian_radius(h, w, min_overlap=0.1), 0.0)
    hm = tgt['heatmaps'][0]
    nz = hm.nonzero()
    save('G1_centerhead_targets', gt_boxes=gt, heat_nz_index=nz.numpy().astype(np.int32),
         heat_nz_value=hm[nz[:, 0], nz[:, 1], nz[:, 2], nz[:, 3]].numpy(), heat_shape=np.array(hm.shape),
         target_boxes=tgt['target_boxes'][0].numpy(), inds=tgt['inds'][0].numpy(), masks=tgt['masks'][0].numpy())

    # ---- G2: focal / regression losses on random predictions with these targets (+ gradients)
    gen = torch.Generator().manual_seed(123)       # tests regenerate the same predictions from this seed
    pred_hm = torch.rand(3, 5, 468, 468, generator=gen) * 0.98 + 0.01
    pred_hm.requires_grad_(True)
    pred_box = torch.randn(3, 8, 468, 468, generator=gen).requires_grad_(True)
    l_ref = ref['loss_utils'].FocalLossCenterNet()(pred_hm, tgt_ref['heatmaps'][0])
    r_ref = ref['loss_utils'].RegLossCenterNet()(pred_box, tgt_ref['masks'][0], tgt_ref['inds'][0], tgt_ref['target_boxes'][0])
    (l_ref + r_ref.sum()).backward()
    g_hm_ref, g_box_ref = pred_hm.grad.clone(), pred_box.grad.clone()
    pred_hm.grad = pred_box.grad = None
    l_or = FO.focal_loss_centernet(pred_hm, tgt['heatmaps'][0])
    r_or = FO.reg_loss_centernet(pred_box, tgt['masks'][0], tgt['inds'][0], tgt['target_boxes'][0])
    (l_or + r_or.sum()).backward()
    check('focal', l_or, l_ref, 1e-5 * float(l_ref.abs()))
    check('reg', r_or, r_ref, 1e-6)
    check('focal grad', pred_hm.grad, g_hm_ref, 1e-6)
    check('reg grad', pred_box.grad, g_box_ref, 1e-7)
    idx = torch.from_numpy(rng.integers(0, 468 * 468 * 15, 4000))
    save('G2_centerhead_losses', pred_seed=123, focal=l_ref.detach().numpy(), reg=r_ref.detach().numpy(),
         grad_probe_index=idx.numpy(), grad_probe_hm=g_hm_ref.flatten()[idx].numpy(),
         grad_box_nz=g_box_ref.flatten()[g_box_ref.flatten().nonzero()[:, 0]].numpy(),
         grad_box_nz_index=g_box_ref.flatten().nonzero()[:, 0].numpy())

    # ---- G4: box decoding (pure torch in the reference: pinned) on random head outputs of a small map
    gen = torch.Generator().manual_seed(77)
    B, C, Hh, Ww, K = 2, 5, 96, 80, 60
    hm = torch.rand(B, C, Hh, Ww, generator=gen) ** 6
    center, cz = torch.rand(B, 2, Hh, Ww, generator=gen), torch.randn(B, 1, Hh, Ww, generator=gen) - 1
    dim = torch.rand(B, 3, Hh, Ww, generator=gen) * 3 + 0.5
    rc, rs = torch.randn(B, 1, Hh, Ww, generator=gen), torch.randn(B, 1, Hh, Ww, generator=gen)
    pcr = np.array([-74.88, -74.88, -5.0, 74.88, 74.88, 3.0], dtype=np.float32)
    lim = torch.tensor([-74.88, -74.88, -3.0, 74.88, -50.0, 3.0])            # a limit that actually removes boxes
    ref_d = cu.decode_bbox_from_heatmap(heatmap=hm, rot_cos=rc, rot_sin=rs, center=center, center_z=cz, dim=dim, vel=None,
                                        iou=torch.ones_like(hm[:, 0:1]), point_cloud_range=pcr, voxel_size=[0.32, 0.32, 8.0],
                                        feature_map_stride=1, K=K, circle_nms=False, score_thresh=0.3,
                                        post_center_limit_range=lim)
    or_d = FO.decode_bbox_from_heatmap(hm, rc, rs, center, cz, dim, pcr, [0.32, 0.32, 8.0], 1, K, 0.3, lim)
    arrs = {}
    for k in range(B):
        for key in ('pred_boxes', 'pred_scores', 'pred_labels'):
            check(f'decode {key} {k}', or_d[k][key], ref_d[k][key], 0.0)
            arrs[f'{key}_{k}'] = ref_d[k][key].numpy()
        assert 0 < len(ref_d[k]['pred_scores']) < K
    save('G4_decode', seed=77, shape=np.array([B, C, Hh, Ww, K]), score_thresh=np.float32(0.3), limit=lim.numpy(), **arrs)

    # ---- G3: end-to-end fine-tune step (VFE -> SiamWCA -> SSTBEVBackbone -> CenterHead loss), small clouds
    for tag, nst, npts, bs in (('G3_finetune_e2e_3stage', 3, 4000, 2),):
        print(tag)
        c = FO.default_finetune_cfg(nst)
        V, B3, B2, H, rcfg = build_finetune_reference(nst, seed=0)
        P = FO.init_finetune_params(c, seed=11, tau=0.25)
        load_into(V, P, 'vfe.'), load_into(B3, P, 'backbone_3d.'), load_into(B2, P, 'backbone_2d.'), load_into(H, P, 'dense_head.')
        for m in (V, B3, B2, H):
            m.train()
        pts, prv = O.synth_frame_pair(npts, bs, seed=33)
        gtb = FO.synth_gt_boxes(bs, 25, seed=8)
        bd = dict(points=torch.from_numpy(pts), points_prev=torch.from_numpy(prv), batch_size=bs, gt_boxes=torch.from_numpy(gtb))
        bd = H(B2(B3(V(bd))))
        loss, tb = H.get_loss()
        loss.backward()
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        cap = {}
        ol = FO.finetune_loss(Pg, pts, prv, gtb, bs, c, cap)
        ol.backward()
        check('loss', ol, loss, 2e-5 * max(1.0, float(loss.abs())))
        check('spatial_features_2d', cap['spatial_features_2d'], bd['spatial_features_2d'], 2e-4)
        gn = {}
        for mod, pre in ((V, 'vfe.'), (B3, 'backbone_3d.'), (B2, 'backbone_2d.'), (H, 'dense_head.')):
            for n_, p_ in mod.named_parameters():
                g_ref, g_or = p_.grad, Pg[pre + n_].grad
                assert g_ref is not None and g_or is not None, pre + n_
                check('grad ' + pre + n_, g_or, g_ref, 3e-3 * max(1.0, float(g_ref.abs().max())))
                gn[pre + n_] = float(g_ref.norm())
        names = [k for k in P.keys()]
        ref_names = [pre + k for mod, pre in ((V, 'vfe.'), (B3, 'backbone_3d.'), (B2, 'backbone_2d.'), (H, 'dense_head.'))
                     for k in mod.state_dict().keys()]
        save(tag, n_points=npts, batch_size=bs, param_seed=11, tau=np.float32(0.25), hm_scale=np.float32(0.2), points=pts, points_prev=prv, gt_boxes=gtb,
             loss=loss.detach().numpy(), hm_loss=np.float32(tb['hm_loss_head_0']), loc_loss=np.float32(tb['loc_loss_head_0']),
             grad_names=np.array(list(gn.keys())), grad_norms=np.array(list(gn.values())),
             state_names=np.array(ref_names), state_shapes=np.array([str(tuple(v.shape)) for mod in (V, B3, B2, H) for v in mod.state_dict().values()]),
             x2d_checksum=bd['spatial_features_2d'].detach().double().sum().numpy(),
             x2d_abs_checksum=bd['spatial_features_2d'].detach().double().abs().sum().numpy())
        missing = set(n for n in names) - set(ref_names)
        assert not missing, missing
    print('fine-tune fixtures written; oracle pinned against the reference on every one of them')


if __name__ == '__main__':
    main()
