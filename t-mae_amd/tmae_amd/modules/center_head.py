"""CenterHead (pcdet/models/dense_heads/center_head.py:11-394), training path: shared conv, per-class-group separate
heads, target assignment and losses.  Same constructor kwargs, forward contract (`data_dict -> data_dict`,
`forward_ret_dict`, `get_loss() -> (loss, tb_dict)`) and state_dict names as the reference.  Targets are assigned on the
device by one HIP launch per head (the reference loops over samples and boxes on the CPU) and the heat-map focal loss is
one fused kernel; evaluation decodes the boxes with torch top-K / gathers and runs the rotated NMS of csrc/iou3d_nms.hip."""
import copy

import torch
import torch.nn as nn
from torch.nn.init import kaiming_normal_

from .. import ops
from .bev_backbone import conv_bn_relu_nhwc


def _conv3(cin, cout, bias):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=bias)


class SeparateHead(nn.Module):
    """One branch per entry of `sep_head_dict` = {name: {out_channels, num_conv}}: (num_conv - 1) x [conv3x3 + BN + ReLU]
    and a final biased conv3x3, registered under the entry's name (state_dict names `heads_list.{i}.{name}.{k}...`,
    center_head.py:11-45).  Init: heat-map branches get the focal-loss prior on their last bias, every other branch
    Kaiming-normal weights and zero biases."""

    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for name, spec in sep_head_dict.items():
            stem = [nn.Sequential(_conv3(input_channels, input_channels, use_bias), nn.BatchNorm2d(input_channels),
                                  nn.ReLU(inplace=True)) for _ in range(spec['num_conv'] - 1)]
            branch = nn.Sequential(*stem, _conv3(input_channels, spec['out_channels'], True))
            if 'hm' in name:
                branch[-1].bias.data.fill_(init_bias)
            else:
                for conv in (m for m in branch.modules() if isinstance(m, nn.Conv2d)):
                    kaiming_normal_(conv.weight.data)
                    if conv.bias is not None:
                        nn.init.constant_(conv.bias, 0)
            self.add_module(name, branch)

    def forward(self, x):
        ret = {}
        src = x          # what the next branch reads: x, or x's alias out of the previous branch's stem conv (its input gradient
        for name in self.sep_head_dict:      # then accumulates into the later branches' instead of meeting them in autograd adds)
            y = src
            for j, layer in enumerate(getattr(self, name)):
                if j == 0 and isinstance(layer, nn.Sequential) and self.training:
                    y, src = conv_bn_relu_nhwc(layer, y, chain=True)
                else:
                    y = conv_bn_relu_nhwc(layer, y) if isinstance(layer, nn.Sequential) else ops.conv3x3_channel_bias(y, layer)
            ret[name] = y
        return ret


def _reg_loss(pred, target, mask):
    """loss_utils._reg_loss (loss_utils.py:321-352): per-code L1 sums over the assigned slots / number of objects."""
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float() * (~torch.isnan(target)).float()
    loss = torch.abs(pred * m - target * m).sum(dim=(0, 1))
    return loss / torch.clamp_min(num, min=1.0)


class CenterHead(nn.Module):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size = grid_size
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.voxel_size = [float(v) for v in voxel_size]
        self.feature_map_stride = self.model_cfg.TARGET_ASSIGNER_CONFIG.get('FEATURE_MAP_STRIDE', None)
        self.class_names = list(class_names)
        self.class_names_each_head = []
        for cur in self.model_cfg.CLASS_NAMES_EACH_HEAD:
            self.class_names_each_head.append([x for x in cur if x in self.class_names])
        total = sum(len(x) for x in self.class_names_each_head)
        assert total == len(self.class_names), f'class_names_each_head={self.class_names_each_head}'
        # global class id (1-based, 0 = padding) -> index inside the head, per head (center_head.py:195-206)
        for hi, names in enumerate(self.class_names_each_head):
            cmap = torch.full((len(self.class_names) + 1,), -1, dtype=torch.int32)
            for gi, n in enumerate(self.class_names):
                if n in names:
                    cmap[gi + 1] = names.index(n)
            self.register_buffer(f'_cls_map_{hi}', cmap, persistent=False)
            self.register_buffer(f'_cls_id_{hi}', torch.tensor([self.class_names.index(n) for n in names]), persistent=False)
        use_bias = self.model_cfg.get('USE_BIAS_BEFORE_NORM', False)
        self.shared_conv = nn.Sequential(
            nn.Conv2d(input_channels, self.model_cfg.SHARED_CONV_CHANNEL, 3, stride=1, padding=1, bias=use_bias),
            nn.BatchNorm2d(self.model_cfg.SHARED_CONV_CHANNEL),
            nn.ReLU(inplace=True))
        self.heads_list = nn.ModuleList()
        self.separate_head_cfg = self.model_cfg.SEPARATE_HEAD_CFG
        for names in self.class_names_each_head:
            hd = copy.deepcopy(dict(self.separate_head_cfg.HEAD_DICT))
            hd = {k: dict(v) for k, v in hd.items()}
            hd['hm'] = dict(out_channels=len(names), num_conv=self.model_cfg.NUM_HM_CONV)
            self.heads_list.append(SeparateHead(self.model_cfg.SHARED_CONV_CHANNEL, hd, init_bias=-2.19, use_bias=use_bias))
        self.with_iou = 'iou' in self.separate_head_cfg.HEAD_DICT
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}

    def assign_targets(self, gt_boxes, feature_map_size=None, **kwargs):
        """center_head.py:168-231.  gt_boxes [B, M, 8]; feature_map_size (H, W)."""
        cfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        ret = {'heatmaps': [], 'target_boxes': [], 'iou_boxes': [], 'inds': [], 'masks': []}
        for hi, names in enumerate(self.class_names_each_head):
            cmap = getattr(self, f'_cls_map_{hi}')
            heat, tb, inds, mask = ops.centerhead_targets(
                gt_boxes, cmap, len(names), feature_map_size, self.point_cloud_range,
                self.voxel_size, cfg.FEATURE_MAP_STRIDE, cfg.NUM_MAX_OBJS, cfg.GAUSSIAN_OVERLAP, cfg.MIN_RADIUS)
            ret['heatmaps'].append(heat), ret['target_boxes'].append(tb), ret['inds'].append(inds), ret['masks'].append(mask)
            if self.with_iou:
                ret['iou_boxes'].append(self._iou_boxes(gt_boxes, cmap, cfg.NUM_MAX_OBJS, mask))
        return ret

    @staticmethod
    def _iou_boxes(gt_boxes, cmap, nmax, mask):
        """center_head.py:122,163: the raw (x, y, z, dx, dy, dz, heading) of the ground-truth box in every assigned slot
        (slot k = k-th box of the head's classes in the sample: a stable compaction), zeros elsewhere."""
        B, M, _ = gt_boxes.shape
        mine = cmap[gt_boxes[..., -1].long().clamp(0, cmap.numel() - 1)] >= 0                    # [B, M]
        order = torch.sort((~mine).int(), dim=1, stable=True)[1]                                 # the head's boxes first
        comp = torch.gather(gt_boxes[..., :7], 1, order.unsqueeze(2).expand(B, M, 7))
        comp = comp * torch.gather(mine, 1, order).unsqueeze(2)
        out = gt_boxes.new_zeros((B, nmax, 7))
        k = min(M, nmax)
        out[:, :k] = comp[:, :k]
        return out * mask.unsqueeze(2).to(out.dtype)

    def _code_weights(self, like, values):
        """LOSS_WEIGHTS.code_weights as a tensor on `like`'s device, cached: new_tensor(list) is a pageable, synchronous
        host-to-device copy -- once per head and step the host waited there for the whole forward pass."""
        key = (like.device, like.dtype, tuple(float(v) for v in values))
        cache = self.__dict__.setdefault('_cw_cache', {})
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.tensor(key[2], dtype=like.dtype, device=like.device)
        return t

    def get_loss(self):
        """center_head.py:237-262 (+ loss_utils FocalLossCenterNet / RegLossCenterNet)."""
        pred_dicts = self.forward_ret_dict['pred_dicts']
        target_dicts = self.forward_ret_dict['target_dicts']
        w = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        tb_dict = {}
        loss = 0
        for idx, pd in enumerate(pred_dicts):
            hm_loss = ops.focal_loss_centernet(pd['hm'], target_dicts['heatmaps'][idx]) * w['cls_weight']
            pred_boxes = torch.cat([pd[n] for n in self.separate_head_cfg.HEAD_ORDER], dim=1).float()
            B, C = pred_boxes.shape[0], pred_boxes.shape[1]
            ind = target_dicts['inds'][idx]
            feat = pred_boxes.permute(0, 2, 3, 1).reshape(B, -1, C)
            pred = feat.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))
            reg = _reg_loss(pred, target_dicts['target_boxes'][idx], target_dicts['masks'][idx])
            loc_loss = (reg * self._code_weights(reg, w['code_weights'])).sum() * w['loc_weight']
            loss = loss + hm_loss + loc_loss
            tb_dict['hm_loss_head_%d' % idx] = hm_loss.detach()        # tensors: no host sync (the reference calls .item())
            tb_dict['loc_loss_head_%d' % idx] = loc_loss.detach()
            if self.with_iou:
                iou_loss = self._iou_loss(pd, ind, target_dicts['masks'][idx], target_dicts['iou_boxes'][idx]) * w['iou_weight']
                loss = loss + iou_loss
                tb_dict['iou_loss_head_%d' % idx] = iou_loss.detach()
        return loss, tb_dict

    def _iou_loss(self, pd, ind, mask, iou_boxes):
        """center_head.py:254-276 + IoULossCenterNet (loss_utils.py:399-420): L1 between the IoU branch at the assigned cells
        and 2 IoU3D(decoded box, ground truth) - 1, summed / (number of objects + 1e-4).  The reference selects the
        assigned slots with a boolean mask (a host sync) and takes the diagonal of the full IoU matrix; here every
        slot goes through the paired IoU kernel and the mask enters as a weight."""
        B, _, H, W = pd['dim'].shape
        ys, xs = torch.meshgrid(torch.arange(H, device=ind.device), torch.arange(W, device=ind.device), indexing='ij')
        xs = (xs.view(1, 1, H, W) + pd['center'][:, 0:1].float()) * self.feature_map_stride * self.voxel_size[0] + self.point_cloud_range[0]
        ys = (ys.view(1, 1, H, W) + pd['center'][:, 1:2].float()) * self.feature_map_stride * self.voxel_size[1] + self.point_cloud_range[1]
        rot = torch.atan2(pd['rot'][:, 1:2].float(), pd['rot'][:, 0:1].float())
        boxes = torch.cat([xs, ys, pd['center_z'].float(), pd['dim'].float().exp(), rot], dim=1).detach()      # [B, 7, H, W]

        def at(feat):
            f = feat.permute(0, 2, 3, 1).reshape(B, H * W, feat.shape[1])
            return f.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], feat.shape[1]))
        m = mask.bool()
        pred_box = torch.where(m.unsqueeze(2), at(boxes), iou_boxes.new_ones(()))     # unassigned slots: harmless unit boxes
        gt_box = torch.where(m.unsqueeze(2), iou_boxes, iou_boxes.new_ones(()))
        target = 2 * ops.boxes_iou3d_paired(pred_box.reshape(-1, 7), gt_box.reshape(-1, 7)).view(B, -1) - 1
        pred = at(pd['iou'].float()).squeeze(2)
        return (torch.abs(pred - target) * m.float()).sum() / (m.float().sum() + 1e-4)

    @staticmethod
    def _decode(heatmap, rot_cos, rot_sin, center, center_z, dim, pc_range, voxel_size, stride, K, score_thresh, limit,
                iou=None):
        """centernet_utils.decode_bbox_from_heatmap / _topk (centernet_utils.py:131-220): two-level top-K over the class
        maps, gather the regression maps at the peaks, boxes (x, y, z, dx, dy, dz, heading), centre-range + score mask."""
        B, C, H, W = heatmap.shape
        ts, ti = torch.topk(heatmap.flatten(2, 3), K)
        ti = ti % (H * W)
        ys = torch.div(ti, W, rounding_mode='floor').float()
        xs = (ti % W).int().float()
        score, tind = torch.topk(ts.view(B, -1), K)
        cls = torch.div(tind, K, rounding_mode='floor').int()
        inds = ti.view(B, -1).gather(1, tind)
        ys, xs = ys.view(B, -1).gather(1, tind), xs.view(B, -1).gather(1, tind)

        def tg(feat):
            f = feat.permute(0, 2, 3, 1).reshape(B, -1, feat.shape[1])
            return f.gather(1, inds.unsqueeze(2).expand(B, K, feat.shape[1]))

        c, rs, rc, cz, dm = tg(center), tg(rot_sin), tg(rot_cos), tg(center_z), tg(dim)
        ious = tg(iou).squeeze(2) if iou is not None else score.new_ones(score.shape)
        angle = torch.atan2(rs, rc)
        x = (xs.view(B, K, 1) + c[:, :, 0:1]) * stride * voxel_size[0] + pc_range[0]
        y = (ys.view(B, K, 1) + c[:, :, 1:2]) * stride * voxel_size[1] + pc_range[1]
        boxes = torch.cat([x, y, cz, dm, angle], dim=-1)
        mask = (boxes[..., :3] >= limit[:3]).all(2) & (boxes[..., :3] <= limit[3:]).all(2)
        if score_thresh is not None:
            mask &= score > score_thresh
        return [dict(pred_boxes=boxes[k, mask[k]], pred_scores=score[k, mask[k]], pred_ious=ious[k, mask[k]],
                     pred_labels=cls[k, mask[k]]) for k in range(B)]

    def generate_predicted_boxes(self, batch_size, pred_dicts):
        """center_head.py:264-334: decode every head, class-agnostic rotated NMS (model_nms_utils.py:6-27)."""
        pp = self.model_cfg.POST_PROCESSING
        ret = [{'pred_boxes': [], 'pred_scores': [], 'pred_labels': []} for _ in range(batch_size)]
        for idx, pd in enumerate(pred_dicts):
            pd = {k: v.float() for k, v in pd.items()}
            limit = pd['hm'].new_tensor(pp.POST_CENTER_LIMIT_RANGE)
            iou = torch.clamp((pd['iou'] + 1) * 0.5, min=0, max=1) if 'iou' in pd else None      # center_head.py:283-286
            dec = self._decode(pd['hm'].sigmoid(), pd['rot'][:, 0:1], pd['rot'][:, 1:2], pd['center'], pd['center_z'],
                               pd['dim'].exp(), self.point_cloud_range, self.voxel_size, self.feature_map_stride,
                               pp.MAX_OBJ_PER_SAMPLE, pp.SCORE_THRESH, limit, iou=iou)
            nms = pp.NMS_CONFIG
            if nms.NMS_TYPE not in ('nms_gpu', 'multi_class_nms'):
                raise NotImplementedError(f'NMS_TYPE {nms.NMS_TYPE} (the reference supports nms_gpu / multi_class_nms here)')
            for k, fd in enumerate(dec):
                fd['pred_labels'] = getattr(self, f'_cls_id_{idx}')[fd['pred_labels'].long()]
                scores, boxes, labels = fd['pred_scores'], fd['pred_boxes'], fd['pred_labels']
                if nms.NMS_TYPE == 'nms_gpu':                       # class-agnostic (model_nms_utils.py:6-25)
                    sel = scores.new_zeros((0,), dtype=torch.long)
                    if scores.shape[0] > 0:
                        top, indices = torch.topk(scores, k=min(nms.NMS_PRE_MAXSIZE, scores.shape[0]))
                        keep, _ = ops.nms_gpu(boxes[indices][:, 0:7], top, nms.NMS_THRESH)
                        sel = indices[keep[:nms.NMS_POST_MAXSIZE]]
                    out_scores = scores[sel]
                else:                                               # per class, IoU-rectified scores (:28-46)
                    sel, out_scores = self._multi_class_nms(scores, fd['pred_ious'], labels, boxes, nms)
                ret[k]['pred_boxes'].append(boxes[sel])
                ret[k]['pred_scores'].append(out_scores)
                ret[k]['pred_labels'].append(labels[sel])
        for k in range(batch_size):
            ret[k]['pred_boxes'] = torch.cat(ret[k]['pred_boxes'], dim=0)
            ret[k]['pred_scores'] = torch.cat(ret[k]['pred_scores'], dim=0)
            ret[k]['pred_labels'] = torch.cat(ret[k]['pred_labels'], dim=0) + 1
        return ret

    @staticmethod
    def _multi_class_nms(scores, ious, labels, boxes, nms):
        """model_nms_utils.multi_class_agnostic_nms (model_nms_utils.py:28-46): score^(1-r_c) * iou^r_c, then per class
        top-K, rotated NMS with the class's threshold, post-NMS cap."""
        rect = scores.new_tensor(nms.IOU_RECTIFIER)[labels]
        rs = torch.pow(scores, 1 - rect) * torch.pow(ious, rect)
        selected = []
        for c in range(len(nms.NMS_THRESH)):
            src = (labels == c).nonzero(as_tuple=True)[0]
            if src.numel() == 0:
                continue
            top, indices = torch.topk(rs[src], k=min(nms.NMS_PRE_MAXSIZE[c], src.numel()))
            keep, _ = ops.nms_gpu(boxes[src][indices][:, 0:7], top, nms.NMS_THRESH[c])
            selected.append(src[indices[keep[:nms.NMS_POST_MAXSIZE[c]]]])
        sel = torch.cat(selected, dim=0) if selected else scores.new_zeros((0,), dtype=torch.long)
        return sel, rs[sel]

    def forward(self, data_dict):
        x2d = data_dict['spatial_features_2d']
        x = conv_bn_relu_nhwc(self.shared_conv, x2d)
        pred_dicts = [head(x) for head in self.heads_list]
        if self.training:
            self.forward_ret_dict['target_dicts'] = self.assign_targets(
                data_dict['gt_boxes'], feature_map_size=x2d.size()[2:],
                feature_map_stride=data_dict.get('spatial_features_2d_strides', None))
        self.forward_ret_dict['pred_dicts'] = pred_dicts
        if not self.training or self.predict_boxes_when_training:
            data_dict['final_box_dicts'] = self.generate_predicted_boxes(data_dict['batch_size'], pred_dicts)
        return data_dict
