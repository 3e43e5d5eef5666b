"""CPU restatement of the FINE-TUNE training step of T-MAE (BASELINE configs[4]):
TemporalDynVFE -> SiamWCA (two-frame SST encoder + window cross-attention, no masking) -> SSTBEVBackbone ->
CenterHead targets + losses, as configured by tools/cfgs/once_models/t_mae.yaml.

TEST INFRASTRUCTURE, like tmae_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.  Every function cites the reference lines it follows; oracle/gen_golden_finetune.py pins it against the
unmodified reference modules (CenterHead, SSTBEVBackbone, SiamWCA, loss_utils, centernet_utils) and writes
tests/golden/G*.npz.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

import tmae_oracle as O

CLASS_NAMES = ['Car', 'Bus', 'Truck', 'Pedestrian', 'Cyclist']


def default_finetune_cfg(num_stages=3):
    """t_mae.yaml:57-239 as a plain dict (model part)."""
    cfg = O.default_model_cfg(num_stages)
    cfg['drop_info'] = {0: dict(max_tokens=8, drop_range=(0, 8)), 1: dict(max_tokens=16, drop_range=(8, 16)),
                        2: dict(max_tokens=32, drop_range=(16, 32)), 3: dict(max_tokens=48, drop_range=(32, 48)),
                        4: dict(max_tokens=64, drop_range=(48, 100000))}
    cfg.update(
        bev_convs=[dict(out_channels=128, dilation=1, padding=1), dict(out_channels=128, dilation=1, padding=1),
                   dict(out_channels=128, dilation=2, padding=2), dict(out_channels=128, dilation=1, padding=1)],
        bev_shortcut=(0, 1, 2),
        class_names=list(CLASS_NAMES), class_names_each_head=[list(CLASS_NAMES)],
        shared_conv_channel=64, use_bias_before_norm=True, num_hm_conv=2,
        head_order=['center', 'center_z', 'dim', 'rot'],
        head_dict=OrderedDict(center=dict(out_channels=2, num_conv=2), center_z=dict(out_channels=1, num_conv=2),
                              dim=dict(out_channels=3, num_conv=2), rot=dict(out_channels=2, num_conv=2)),
        feature_map_stride=1, num_max_objs=500, gaussian_overlap=0.1, min_radius=2,
        cls_weight=1.0, loc_weight=2.0, code_weights=[1.0] * 8,
    )
    return cfg


# --------------------------------------------------------------------------- targets
def gaussian_radius(height, width, min_overlap):
    """centernet_utils.gaussian_radius (centernet_utils.py:9-36), fp32 torch like the reference."""
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + (b1 ** 2 - 4 * c1).sqrt()) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + (b2 ** 2 - 16 * c2).sqrt()) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + (b3 ** 2 - 4 * a3 * c3).sqrt()) / 2
    return torch.min(torch.min(r1, r2), r3)


def gaussian2d(radius):
    """centernet_utils.gaussian2D with sigma = diameter / 6 (centernet_utils.py:39-45,48-50): float64 numpy."""
    d = 2 * radius + 1
    sigma = d / 6
    m = (d - 1.) / 2.
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def assign_targets_single(gt, num_classes, fm_size_xy, cfg):
    """CenterHead.assign_target_of_single_head (center_head.py:107-166).  gt [n, 8] (x,y,z,dx,dy,dz,heading,cls 1-based)
    fp32 tensor.  Returns heatmap [C,H,W], ret_boxes [500,8], inds [500] i64, mask [500] i64."""
    W, H = fm_size_xy
    nmax = cfg['num_max_objs']
    pcr, vs, stride = cfg['point_cloud_range'], cfg['voxel_size'], cfg['feature_map_stride']
    heatmap = gt.new_zeros(num_classes, H, W)
    ret_boxes = gt.new_zeros((nmax, gt.shape[-1]))
    inds = torch.zeros(nmax, dtype=torch.long)
    mask = torch.zeros(nmax, dtype=torch.long)
    x, y, z = gt[:, 0], gt[:, 1], gt[:, 2]
    cx = torch.clamp((x - pcr[0]) / vs[0] / stride, min=0, max=W - 0.5)
    cy = torch.clamp((y - pcr[1]) / vs[1] / stride, min=0, max=H - 0.5)
    center = torch.stack([cx, cy], dim=-1)
    ci = center.int()
    dx = gt[:, 3] / vs[0] / stride
    dy = gt[:, 4] / vs[1] / stride
    radius = torch.clamp_min(gaussian_radius(dx, dy, cfg['gaussian_overlap']).int(), min=cfg['min_radius'])
    for k in range(min(nmax, gt.shape[0])):
        if dx[k] <= 0 or dy[k] <= 0:
            continue
        if not (0 <= ci[k][0] <= W and 0 <= ci[k][1] <= H):
            continue
        cls = int(gt[k, -1] - 1)
        r = int(radius[k])
        g = gaussian2d(r)
        xi, yi = int(center[k][0]), int(center[k][1])
        left, right = min(xi, r), min(W - xi, r + 1)
        top, bottom = min(yi, r), min(H - yi, r + 1)
        mh = heatmap[cls, yi - top:yi + bottom, xi - left:xi + right]
        mg = torch.from_numpy(g[r - top:r + bottom, r - left:r + right]).float()
        if min(mg.shape) > 0 and min(mh.shape) > 0:
            torch.max(mh, mg, out=mh)
        inds[k] = ci[k, 1] * W + ci[k, 0]
        mask[k] = 1
        ret_boxes[k, 0:2] = center[k] - ci[k].float()
        ret_boxes[k, 2] = z[k]
        ret_boxes[k, 3:6] = gt[k, 3:6].log()
        ret_boxes[k, 6] = torch.cos(gt[k, 6])
        ret_boxes[k, 7] = torch.sin(gt[k, 6])
    return heatmap, ret_boxes, inds, mask


def assign_targets(gt_boxes, fm_size_hw, cfg):
    """CenterHead.assign_targets (center_head.py:168-231): per head, per sample; boxes of classes outside the head are
    skipped, the class id becomes the index inside the head + 1.  gt_boxes [B, M, 8] (zero rows = padding, class 0)."""
    W, H = fm_size_hw[1], fm_size_hw[0]
    all_names = ['bg'] + cfg['class_names']
    out = dict(heatmaps=[], target_boxes=[], inds=[], masks=[])
    for names in cfg['class_names_each_head']:
        hs, bs, is_, ms = [], [], [], []
        for b in range(gt_boxes.shape[0]):
            cur = gt_boxes[b]
            rows = []
            for i in range(cur.shape[0]):
                nm = all_names[int(cur[i, -1])]
                if nm not in names:
                    continue
                t = cur[i].clone()
                t[-1] = names.index(nm) + 1
                rows.append(t[None])
            g = torch.cat(rows, 0) if rows else cur[:0]
            h, rb, ii, mm = assign_targets_single(g, len(names), (W, H), cfg)
            hs.append(h), bs.append(rb), is_.append(ii), ms.append(mm)
        out['heatmaps'].append(torch.stack(hs)), out['target_boxes'].append(torch.stack(bs))
        out['inds'].append(torch.stack(is_)), out['masks'].append(torch.stack(ms))
    return out


# --------------------------------------------------------------------------- losses
def focal_loss_centernet(pred, gt):
    """loss_utils.neg_loss_cornernet (loss_utils.py:273-309), mask=None."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum()
    neg_loss = (torch.log(1 - pred) * torch.pow(pred, 2) * torch.pow(1 - gt, 4) * neg).sum()
    num_pos = pos.sum()
    if num_pos == 0:
        return -neg_loss
    return -(pos_loss + neg_loss) / num_pos


def reg_loss_centernet(output, mask, ind, target):
    """RegLossCenterNet (loss_utils.py:355-380) = gather at ind + _reg_loss (:321-352): per-code L1 sums / num."""
    B, C = output.shape[0], output.shape[1]
    feat = output.permute(0, 2, 3, 1).reshape(B, -1, C)
    pred = feat.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float() * (~torch.isnan(target)).float()
    loss = torch.abs(pred * m - target * m).sum(dim=(0, 1))
    return loss / torch.clamp_min(num, min=1.0)


# --------------------------------------------------------------------------- network
def bn2d(x, p, name, eps):
    return O.batch_norm_train(x, p[name + '.weight'], p[name + '.bias'], eps)


def bev_backbone(x, p, prefix, cfg):
    """SSTBEVBackbone.forward (sst_bev_backbone.py:26-43): conv3x3(+dilation) + BN(eps 1e-3) + ReLU, residual where
    the shape is kept and the index is in CONV_SHORTCUT."""
    for i, kw in enumerate(cfg['bev_convs']):
        t = F.conv2d(x, p[f'{prefix}conv_layer.{i}.0.weight'], padding=kw['padding'], dilation=kw['dilation'])
        t = F.relu(bn2d(t, p, f'{prefix}conv_layer.{i}.1', 1e-3))
        x = t + x if (t.shape == x.shape and i in cfg['bev_shortcut']) else t
    return x


def center_head(x, p, prefix, cfg):
    """CenterHead.forward network part (center_head.py:76-83,361-368) + SeparateHead (:11-45): default BN eps 1e-5."""
    bias = p.get(prefix + 'shared_conv.0.bias')
    x = F.relu(bn2d(F.conv2d(x, p[prefix + 'shared_conv.0.weight'], bias, padding=1), p, prefix + 'shared_conv.1', 1e-5))
    preds = []
    for hi, names in enumerate(cfg['class_names_each_head']):
        hd = OrderedDict(cfg['head_dict'])
        hd['hm'] = dict(out_channels=len(names), num_conv=cfg['num_hm_conv'])
        out = {}
        for name, spec in hd.items():
            pre = f'{prefix}heads_list.{hi}.{name}.'
            y = x
            for k in range(spec['num_conv'] - 1):
                y = F.conv2d(y, p[f'{pre}{k}.0.weight'], p.get(f'{pre}{k}.0.bias'), padding=1)
                y = F.relu(bn2d(y, p, f'{pre}{k}.1', 1e-5))
            k = spec['num_conv'] - 1
            out[name] = F.conv2d(y, p[f'{pre}{k}.weight'], p[f'{pre}{k}.bias'], padding=1)
        preds.append(out)
    return preds


def center_head_loss(preds, targets, cfg):
    """CenterHead.get_loss (center_head.py:237-262), no IoU head."""
    loss = 0
    parts = {}
    for idx, pd in enumerate(preds):
        hm = torch.clamp(pd['hm'].sigmoid(), min=1e-4, max=1 - 1e-4)
        hm_loss = focal_loss_centernet(hm, targets['heatmaps'][idx]) * cfg['cls_weight']
        pred_boxes = torch.cat([pd[n] for n in cfg['head_order']], dim=1)
        reg = reg_loss_centernet(pred_boxes, targets['masks'][idx], targets['inds'][idx], targets['target_boxes'][idx])
        loc_loss = (reg * reg.new_tensor(cfg['code_weights'])).sum() * cfg['loc_weight']
        loss = loss + hm_loss + loc_loss
        parts[f'hm_loss_head_{idx}'], parts[f'loc_loss_head_{idx}'] = hm_loss, loc_loss
    return loss, parts


def siamwca_forward(params, cur, prv, batch_size, cfg, capture=None):
    """SiamWCA.forward (SiamWCA.py:621-667): both frames through the Siamese SST blocks, window cross-attention per
    scale, dense_conv (:594-619, modules `deblocks` / `conv_out`)."""
    cap = capture if capture is not None else {}
    bp = 'backbone_3d.'
    gx, gy, _ = cfg['grid_size']

    def encode(feat, vcoords):
        ind = np.asarray(vcoords)[:, [0, 2, 3]]
        shape = (gy, gx)
        outs = []
        for si, stage in enumerate(cfg['stages']):
            feat, ind, shape = O.sst_block(feat, ind, shape, params, f'{bp}sst_blocks.{si}.', stage, cfg, {})
            outs.append((feat, ind, shape))
        return outs

    ms_prev = encode(prv['voxel_features'], prv['voxel_coords'])
    ms_cur = encode(cur['voxel_features'], cur['voxel_coords'])
    ms = []
    for si, stage in enumerate(cfg['stages']):
        f, ind, shape = ms_cur[si]
        fp, indp, _ = ms_prev[si]
        f = O.wca_block(f, ind, fp, indp, shape, params, f'{bp}wca_blocks.{si}.', stage, cfg, {})
        ms.append((f, ind, shape))
    spatial = O.dense_decoder(ms, batch_size, params, bp, cfg, deblocks='deblocks', conv_out='conv_out')
    cap['spatial_features'] = spatial
    return spatial


def finetune_loss(params, points, points_prev, gt_boxes, batch_size, cfg, capture=None):
    """CenterPoint.forward + get_training_loss (centerpoint.py:9-33) on the t_mae.yaml module list."""
    cap = capture if capture is not None else {}
    cur = O.vfe_forward(params, 'vfe.', points, cfg)
    prv = O.vfe_forward(params, 'vfe.', points_prev, cfg)
    spatial = siamwca_forward(params, cur, prv, batch_size, cfg, cap)
    x2d = bev_backbone(spatial, params, 'backbone_2d.', cfg)
    cap['spatial_features_2d'] = x2d
    preds = center_head(x2d, params, 'dense_head.', cfg)
    targets = assign_targets(torch.as_tensor(gt_boxes, dtype=torch.float32), tuple(x2d.shape[2:]), cfg)
    loss, parts = center_head_loss(preds, targets, cfg)
    cap.update(preds=preds, targets=targets, parts=parts)
    return loss


# --------------------------------------------------------------------------- init / synthetic labels
def init_finetune_params(cfg, seed=0, num_point_features=4, tau=None, hm_scale=0.2):
    """State dict with the reference's names (CenterPoint of t_mae.yaml): vfe.*, backbone_3d.{sst_blocks,wca_blocks,
    deblocks,conv_out}.*, backbone_2d.conv_layer.*, dense_head.{shared_conv,heads_list}.*.  `hm_scale` scales the last
    heat-map conv (an untrained head with unit-scale logits gives a focal loss of several hundred whose gradient through
    the seven BatchNorm layers below it is ill-conditioned: 1 % run-to-run in fp32)."""
    base = O.init_params(cfg, seed=seed, num_point_features=num_point_features, tau=tau)
    P = OrderedDict()
    for k, v in base.items():
        if 'decoder_pred' in k:
            continue
        P[k.replace('decoder_deblocks', 'deblocks').replace('decoder_conv_out', 'conv_out')] = v
    g = torch.Generator().manual_seed(seed + 1000)

    def conv(name, cout, cin, bias):
        P[name + '.weight'] = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
        if bias:
            P[name + '.bias'] = 0.05 * torch.randn(cout, generator=g)

    def norm(name, c):
        P[name + '.weight'] = 1.0 + 0.1 * torch.randn(c, generator=g)
        P[name + '.bias'] = 0.1 * torch.randn(c, generator=g)

    cin = sum(f['cout'] for f in cfg['fuse']) // len(cfg['fuse'])
    for i, kw in enumerate(cfg['bev_convs']):
        conv(f'backbone_2d.conv_layer.{i}.0', kw['out_channels'], cin, False)
        norm(f'backbone_2d.conv_layer.{i}.1', kw['out_channels'])
        cin = kw['out_channels']
    sc = cfg['shared_conv_channel']
    conv('dense_head.shared_conv.0', sc, cin, cfg['use_bias_before_norm'])
    norm('dense_head.shared_conv.1', sc)
    for hi, names in enumerate(cfg['class_names_each_head']):
        hd = OrderedDict(cfg['head_dict'])
        hd['hm'] = dict(out_channels=len(names), num_conv=cfg['num_hm_conv'])
        for name, spec in hd.items():
            pre = f'dense_head.heads_list.{hi}.{name}.'
            for k in range(spec['num_conv'] - 1):
                conv(f'{pre}{k}.0', sc, sc, cfg['use_bias_before_norm'])
                norm(f'{pre}{k}.1', sc)
            conv(f'{pre}{spec["num_conv"] - 1}', spec['out_channels'], sc, True)
            if name == 'hm':
                P[f'{pre}{spec["num_conv"] - 1}.bias'].fill_(-2.19)
                P[f'{pre}{spec["num_conv"] - 1}.weight'] *= hm_scale
    return P


def synth_gt_boxes(batch_size, n_boxes, seed, limit=74.0):
    """Synthetic ONCE-style labels [B, n_boxes, 8]: x, y, z, dx, dy, dz, heading, class (1..5; 0 = padding row)."""
    rng = np.random.default_rng(seed)
    sizes = np.array([[4.4, 1.9, 1.6], [11.0, 2.9, 3.4], [7.5, 2.6, 3.0], [0.8, 0.8, 1.75], [2.0, 0.8, 1.6]], np.float32)
    out = np.zeros((batch_size, n_boxes, 8), np.float32)
    for b in range(batch_size):
        n = int(rng.integers(n_boxes // 2, n_boxes + 1))
        cls = rng.integers(1, 6, n)
        out[b, :n, 0] = rng.uniform(-limit, limit, n)
        out[b, :n, 1] = rng.uniform(-limit, limit, n)
        out[b, :n, 2] = rng.normal(-1.0, 0.4, n)
        out[b, :n, 3:6] = sizes[cls - 1] * rng.uniform(0.85, 1.15, (n, 3)).astype(np.float32)
        out[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
        out[b, :n, 7] = cls
    return out


# --------------------------------------------------------------------------- evaluation: decode + rotated NMS
def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, pc_range, voxel_size, stride, K,
                             score_thresh, post_center_limit_range):
    """centernet_utils.decode_bbox_from_heatmap (centernet_utils.py:154-220, _topk :131-151): two-level top-K over the
    class heat maps, gather the regression maps at the peaks, build (x, y, z, dx, dy, dz, heading), keep boxes inside
    the centre range with score > thresh.  Inputs [B, c, H, W] tensors (heatmap already sigmoid, dim already exp)."""
    B, C, H, W = heatmap.shape
    ts, ti = torch.topk(heatmap.flatten(2, 3), K)
    ti = ti % (H * W)
    ys = torch.div(ti, W, rounding_mode='floor').float()
    xs = (ti % W).int().float()
    score, tind = torch.topk(ts.view(B, -1), K)
    cls = torch.div(tind, K, rounding_mode='floor').int()
    g1 = lambda t: t.view(B, -1, 1).gather(1, tind.unsqueeze(2)).view(B, K)
    inds, ys, xs = g1(ti), g1(ys), g1(xs)

    def tg(feat):
        f = feat.permute(0, 2, 3, 1).contiguous().view(B, -1, feat.shape[1])
        return f.gather(1, inds.unsqueeze(2).expand(B, K, feat.shape[1]))

    c, rs, rc, cz, dm = tg(center), tg(rot_sin), tg(rot_cos), tg(center_z), tg(dim)
    angle = torch.atan2(rs, rc)
    x = (xs.view(B, K, 1) + c[:, :, 0:1]) * stride * voxel_size[0] + pc_range[0]
    y = (ys.view(B, K, 1) + c[:, :, 1:2]) * stride * voxel_size[1] + pc_range[1]
    boxes = torch.cat([x, y, cz, dm, angle], dim=-1)
    lim = torch.as_tensor(post_center_limit_range, dtype=torch.float32)
    mask = (boxes[..., :3] >= lim[:3]).all(2) & (boxes[..., :3] <= lim[3:]).all(2)
    if score_thresh is not None:
        mask &= score > score_thresh
    return [dict(pred_boxes=boxes[k, mask[k]], pred_scores=score[k, mask[k]], pred_labels=cls[k, mask[k]]) for k in range(B)]


def _rect_corners(b):
    x, y, dx, dy, a = float(b[0]), float(b[1]), float(b[3]), float(b[4]), float(b[6])
    c, s = math.cos(a), math.sin(a)
    pts = [(-dx / 2, -dy / 2), (dx / 2, -dy / 2), (dx / 2, dy / 2), (-dx / 2, dy / 2)]      # counter-clockwise
    return [(x + px * c - py * s, y + px * s + py * c) for px, py in pts]


def _clip(poly, a, b):
    """Sutherland-Hodgman: keep the part of `poly` on the left of the directed edge a -> b."""
    out = []
    n = len(poly)
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def overlap_bev(box_a, box_b):
    """Area of the intersection of two rotated BEV rectangles (x, y, z, dx, dy, dz, heading), float64.  Restates
    what iou3d_nms `box_overlap` computes (iou3d_nms_kernel.cu:113-207: corner-in-box tests + edge intersections +
    polygon area) by convex clipping.  PARITY UNPINNED: the reference's implementation is CUDA-only here
    (iou3d_cpu.cpp includes cuda.h: unbuildable in this image) and ships no test vectors."""
    poly = _rect_corners(box_a)
    cb = _rect_corners(box_b)
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, cb[i], cb[(i + 1) % 4])
    if len(poly) < 3:
        return 0.0
    area = 0.0
    for i in range(len(poly)):
        p, q = poly[i], poly[(i + 1) % len(poly)]
        area += p[0] * q[1] - q[0] * p[1]
    return abs(area) / 2


def iou_bev(box_a, box_b):
    """iou3d_nms `iou_bev` (iou3d_nms_kernel.cu:209-217): overlap / max(Sa + Sb - overlap, EPS=1e-8)."""
    sa, sb = float(box_a[3]) * float(box_a[4]), float(box_b[3]) * float(box_b[4])
    o = overlap_bev(box_a, box_b)
    return o / max(sa + sb - o, 1e-8)


def nms_bev(boxes, scores, thresh, pre_maxsize=None):
    """iou3d_nms_utils.nms_gpu (iou3d_nms_utils.py:84-99) + nms_kernel / CPU suppression loop (iou3d_nms.cpp:104-150):
    sort by score (descending), keep a box unless a kept higher-score box overlaps it with BEV IoU > thresh."""
    boxes = np.asarray(boxes, np.float64)
    order = np.argsort(-np.asarray(scores, np.float64), kind='stable')
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = []
    for i in order:
        if all(iou_bev(boxes[i], boxes[j]) <= thresh for j in keep):
            keep.append(int(i))
    return np.array(keep, np.int64)


def iou3d(boxes_a, boxes_b):
    """iou3d_nms_utils.boxes_iou3d_gpu (iou3d_nms_utils.py:48-81): BEV overlap x height overlap / union volume."""
    A, Bn = np.asarray(boxes_a, np.float64), np.asarray(boxes_b, np.float64)
    out = np.zeros((len(A), len(Bn)))
    for i, a in enumerate(A):
        for j, b in enumerate(Bn):
            h = max(min(a[2] + a[5] / 2, b[2] + b[5] / 2) - max(a[2] - a[5] / 2, b[2] - b[5] / 2), 0.0)
            o = overlap_bev(a, b) * h
            out[i, j] = o / max(a[3] * a[4] * a[5] + b[3] * b[4] * b[5] - o, 1e-6)
    return out
