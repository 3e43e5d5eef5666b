"""G6: the CenterHead with an IoU branch and multi-class NMS (center_head.py:49-394 with 'iou' in HEAD_DICT,
loss_utils.IoULossCenterNet :399-420, model_nms_utils.multi_class_agnostic_nms :28-46, centernet_utils.decode_bbox_from_heatmap
:154-220) -- run from the UNMODIFIED reference classes on a small feature map.  TEST INFRASTRUCTURE, build container only.

Stand-ins (the CUDA extension iou3d_nms is absent): `boxes_iou3d_gpu` and `nms_gpu` of iou3d_nms_utils are served by
oracle/finetune_oracle.py's float64 restatements (PARITY UNPINNED for the rotated-IoU arithmetic itself; everything around it --
target slots, the gather of decoded boxes, the loss normalisation, score rectification, per-class top-K / thresholds / caps --
is the reference's own code)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import finetune_oracle as FO      # noqa: E402
import ref_import as R            # noqa: E402
import gen_golden_finetune as GF  # noqa: E402
from gen_golden import save       # noqa: E402

CLASSES = ['Car', 'Bus', 'Truck', 'Pedestrian', 'Cyclist']
PCR = [-15.36, -15.36, -5.0, 15.36, 15.36, 3.0]
HEAD_CFG = dict(
    NAME='CenterHead', CLASS_AGNOSTIC=False, CLASS_NAMES_EACH_HEAD=[['Car', 'Bus', 'Truck'], ['Pedestrian', 'Cyclist']],
    SHARED_CONV_CHANNEL=16, USE_BIAS_BEFORE_NORM=True, NUM_HM_CONV=2,
    SEPARATE_HEAD_CFG=dict(HEAD_ORDER=['center', 'center_z', 'dim', 'rot'],
                           HEAD_DICT={'center': {'out_channels': 2, 'num_conv': 2}, 'center_z': {'out_channels': 1, 'num_conv': 2},
                                      'dim': {'out_channels': 3, 'num_conv': 2}, 'rot': {'out_channels': 2, 'num_conv': 2},
                                      'iou': {'out_channels': 1, 'num_conv': 2}}),
    TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=1, NUM_MAX_OBJS=60, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
    LOSS_CONFIG=dict(LOSS_WEIGHTS={'cls_weight': 1.0, 'loc_weight': 2.0, 'iou_weight': 1.0,
                                   'code_weights': [1.0] * 8}),
    POST_PROCESSING=dict(SCORE_THRESH=0.05, POST_CENTER_LIMIT_RANGE=PCR, MAX_OBJ_PER_SAMPLE=80,
                         NMS_CONFIG=dict(NMS_TYPE='multi_class_nms', NMS_THRESH=[0.5, 0.5, 0.5, 0.3, 0.3],
                                         NMS_PRE_MAXSIZE=[60, 60, 60, 40, 40], NMS_POST_MAXSIZE=[20, 20, 20, 10, 10],
                                         IOU_RECTIFIER=[0.68, 0.68, 0.68, 0.71, 0.65])))


def main():
    ref = GF.load_finetune_reference()
    iou_mod = sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_utils']
    iou_mod.boxes_iou3d_gpu = lambda a, b: torch.from_numpy(FO.iou3d(a.detach().numpy(), b.detach().numpy())).float()
    iou_mod.nms_gpu = lambda boxes, scores, thresh, pre_maxsize=None, **kw: (
        torch.from_numpy(FO.nms_bev(boxes.detach().numpy(), scores.detach().numpy(), thresh, pre_maxsize)).long(), None)
    nms_utils = __import__('importlib').import_module('pcdet.models.model_utils.model_nms_utils')
    nms_utils.iou3d_nms_utils = iou_mod
    ref['loss_utils'].iou3d_nms_utils = iou_mod
    ref['head'].model_nms_utils = nms_utils
    cfg = R.AttrDict(HEAD_CFG)
    torch.manual_seed(5)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        H = ref['head'].CenterHead(cfg, input_channels=16, num_class=5, class_names=CLASSES, grid_size=np.array([96, 96, 1]),
                                   point_cloud_range=np.array(PCR, np.float32), voxel_size=[0.32, 0.32, 8.0],
                                   predict_boxes_when_training=False)
        with torch.no_grad():                     # an untrained head decodes nothing: give the branches some signal
            for n, p in H.named_parameters():
                if n.endswith('hm.1.bias'):
                    p.fill_(-1.2)
                elif 'iou' in n and n.endswith('weight') and p.dim() == 4:
                    p.mul_(0.2)
        rng = np.random.default_rng(9)
        B = 2
        gt = np.zeros((B, 14, 8), np.float32)
        sizes = np.array([[4.4, 1.9, 1.6], [8.0, 2.6, 3.0], [6.5, 2.4, 2.8], [0.8, 0.8, 1.75], [2.0, 0.8, 1.6]], np.float32)
        for b in range(B):
            n = 12 - 3 * b
            cls = rng.integers(1, 6, n)
            gt[b, :n, 0:2] = rng.uniform(-14, 14, (n, 2))
            gt[b, :n, 2] = rng.normal(-1.0, 0.3, n)
            gt[b, :n, 3:6] = sizes[cls - 1] * rng.uniform(0.9, 1.1, (n, 3))
            gt[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
            gt[b, :n, 7] = cls
        x = torch.randn(B, 16, 96, 96)
        H.train()
        dd = H({'spatial_features_2d': x, 'gt_boxes': torch.from_numpy(gt.copy()), 'batch_size': B})   # the reference relabels gt_boxes IN PLACE (center_head.py:199-203)
        loss, tb = H.get_loss()
        loss.backward()
        gnames = [n for n, p in H.named_parameters() if p.grad is not None]
        gnorms = [float(p.grad.norm()) for n, p in H.named_parameters() if p.grad is not None]
        tgt = H.forward_ret_dict['target_dicts']
        out = dict(head_cfg=__import__('json').dumps(HEAD_CFG), pc_range=np.array(PCR, np.float32), x=x.numpy(), gt_boxes=gt,
                   loss=float(loss), tb_names=np.array(sorted(tb)), tb_values=np.array([float(tb[k]) for k in sorted(tb)]),
                   grad_names=np.array(gnames), grad_norms=np.array(gnorms),
                   iou_boxes_0=tgt['iou_boxes'][0].numpy(), iou_boxes_1=tgt['iou_boxes'][1].numpy(),
                   masks_0=tgt['masks'][0].numpy(), masks_1=tgt['masks'][1].numpy())
        sd = H.state_dict()
        out['state_names'] = np.array(list(sd.keys()))
        for i, (k, v) in enumerate(sd.items()):
            out[f'state_{i}'] = v.detach().numpy()
        H.zero_grad()
        H.eval()
        with torch.no_grad():
            dd = H({'spatial_features_2d': x, 'batch_size': B})
        for k, fd in enumerate(dd['final_box_dicts']):
            out[f'det_boxes_{k}'] = fd['pred_boxes'].numpy()
            out[f'det_scores_{k}'] = fd['pred_scores'].numpy()
            out[f'det_labels_{k}'] = fd['pred_labels'].numpy()
            print('sample', k, 'detections', len(fd['pred_scores']), 'labels', np.bincount(fd['pred_labels'].numpy(), minlength=6))
    finally:
        torch.Tensor.cuda = orig_cuda
    print('loss', float(loss), {k: float(v) for k, v in tb.items()})
    save('G6_iou_head', **out)


if __name__ == '__main__':
    main()
