"""ONCE average precision (pcdet/datasets/once_temporal/once_eval/evaluation.py:26-157 and the helpers it calls,
eval_utils.py:3-48) for the fine-tune path's evaluation.

Same definitions, defaults (super-classes, 'Overall&Distance', AP with heading, 50 recall points) and result keys as the
reference; what differs is the execution:
  * the rotated BEV intersections come from the HIP kernel of csrc/iou3d_nms.hip (`ops.boxes_overlap_bev`; the
    reference launches a numba.cuda kernel, iou_utils.py:277-343), all samples of a part in one call, and the 3-D IoU with
    the heading filter (evaluation.py:360-396) is finished on the device;
  * the greedy matching (accumulate_scores :183-217, compute_statistics :219-269 -- numba-compiled loops over
    (threshold, ground truth, prediction) in the reference) runs once per ground-truth box for ALL score thresholds
    at a time: the scan over predictions has a closed form (the first maximum-IoU accepted prediction, else the first
    ignored one), so the result is identical, not approximately equal.
"""
import numpy as np
import torch

SUPERCLASS_IOU_THRESHOLDS = {'Vehicle': 0.7, 'Pedestrian': 0.3, 'Cyclist': 0.5}
DIFFICULTIES = ('overall', '0-30m', '30-50m', '50m-inf')


def compute_split_parts(num_samples, num_parts):
    part, rem = num_samples // num_parts, num_samples % num_parts
    if part == 0:
        return [num_samples]
    return [part] * num_parts + ([rem] if rem else [])


def iou3d_with_heading(gt_boxes, pred_boxes, device=None):
    """[N,7] x [M,7] (x, y, z, dx, dy, dz, rot) -> IoU [N,M] (numpy float64), evaluation.py:360-396.  The eval
    convention turns a POSITIVE angle CLOCKWISE (iou_utils.py:218-242); the detection kernels turn counter-clockwise
    (iou3d_nms), hence the negated headings handed to the kernel."""
    from .. import ops
    device = device or torch.device('cuda', torch.cuda.current_device())
    g = torch.as_tensor(np.asarray(gt_boxes, np.float64).reshape(-1, 7), device=device)
    p = torch.as_tensor(np.asarray(pred_boxes, np.float64).reshape(-1, 7), device=device)
    if g.shape[0] == 0 or p.shape[0] == 0:
        return np.zeros((g.shape[0], p.shape[0]))
    gk, pk = g.float().clone(), p.float().clone()
    gk[:, 6], pk[:, 6] = -gk[:, 6], -pk[:, 6]
    inter2d = ops.boxes_overlap_bev(gk, pk).double()
    gmax, gmin = g[:, 2:3] + g[:, 5:6] * 0.5, g[:, 2:3] - g[:, 5:6] * 0.5
    pmax, pmin = (p[:, 2:3] + p[:, 5:6] * 0.5).t(), (p[:, 2:3] - p[:, 5:6] * 0.5).t()
    inter_h = (torch.minimum(gmax, pmax) - torch.maximum(gmin, pmin)).clamp(min=0)
    inter3d = inter2d * inter_h
    union = (g[:, 3:4] * g[:, 4:5] * g[:, 5:6]) + (p[:, 3:4] * p[:, 4:5] * p[:, 5:6]).t() - inter3d
    iou = inter3d / union
    diff = (g[:, 6:7] - p[:, 6:7].t()).abs()
    diff = torch.where(diff >= np.pi, 2 * np.pi - diff, diff)
    iou = torch.where(diff > np.pi / 2, torch.zeros_like(iou), iou)
    return iou.cpu().numpy()


def compute_iou3d(gt_annos, pred_annos, split_parts):
    ious, idx = [], 0
    for n in split_parts:
        gpart, ppart = gt_annos[idx:idx + n], pred_annos[idx:idx + n]
        gb = np.concatenate([np.asarray(a['boxes_3d'], np.float64).reshape(-1, 7) for a in gpart], 0)
        pb = np.concatenate([np.asarray(a['boxes_3d'], np.float64).reshape(-1, 7) for a in ppart], 0)
        full = iou3d_with_heading(gb, pb)
        gi = pi = 0
        for g, p in zip(gpart, ppart):
            ng, npd = len(g['name']), len(p['name'])
            ious.append(full[gi:gi + ng, pi:pi + npd])
            gi += ng
            pi += npd
        idx += n
    return ious


def overall_distance_filter(boxes, level):
    dist = np.sqrt(np.sum(boxes[:, 0:3] * boxes[:, 0:3], axis=1))
    keep = [np.ones(len(boxes), bool), dist < 30, (dist >= 30) & (dist < 50), dist >= 50][level]
    return ~keep


def filter_data(gt_anno, pred_anno, difficulty_level, class_name):
    """Flags 0 = accepted, 1 = same class but outside this difficulty (ignored), -1 = other class."""
    def flags(anno):
        names = np.asarray(anno['name'])
        f = np.zeros(len(names), dtype=np.int64)
        if class_name == 'Vehicle':
            f[np.logical_or(names == 'Pedestrian', names == 'Cyclist')] = -1
        else:
            f[names != class_name] = -1
        f[overall_distance_filter(np.asarray(anno['boxes_3d'], np.float64).reshape(-1, 7), difficulty_level)] = 1
        return f
    return flags(gt_anno), flags(pred_anno)


def get_thresholds(scores, num_gt, num_pr_points):
    eps = 1e-6
    scores = np.sort(np.asarray(scores, np.float64))[::-1]
    recall_level, thresholds, n = 0, [], len(scores)
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < n - 1 else l_recall
        if (r_recall + l_recall < 2 * recall_level) and i < n - 1:
            continue
        thresholds.append(score)
        recall_level += 1 / num_pr_points
        while r_recall + l_recall + eps > 2 * recall_level:
            thresholds.append(score)
            recall_level += 1 / num_pr_points
    return thresholds


def accumulate_scores(iou, pred_scores, gt_flag, pred_flag, iou_threshold):
    """Scores of the true positives when every prediction counts (no score threshold)."""
    assigned = np.zeros(iou.shape[1], bool)
    usable = pred_flag != -1
    acc = []
    for i in np.nonzero(gt_flag != -1)[0]:
        cand = usable & ~assigned & (iou[i] > iou_threshold)
        if not cand.any():
            continue
        det = int(np.argmax(np.where(cand, pred_scores, -np.inf)))        # first maximum, as the reference's strict '>'
        assigned[det] = True
        if not (gt_flag[i] == 1 or pred_flag[det] == 1):
            acc.append(pred_scores[det])
    return np.asarray(acc, np.float64)


def compute_statistics_all(iou, pred_scores, gt_flag, pred_flag, thresholds, iou_threshold):
    """(tp, fp, fn) [T,3] for every score threshold at once.  The reference's scan over the predictions of one ground
    truth box ends on: the first maximum-IoU prediction among the accepted ones (flag 0) if there is any, else the
    first ignored (flag 1) one."""
    T, P = len(thresholds), iou.shape[1]
    th = np.asarray(thresholds, np.float64)[:, None]
    under = pred_scores[None, :] < th                                       # [T,P]
    assigned = np.zeros((T, P), bool)
    out = np.zeros((T, 3))
    if P == 0:                                                              # nothing predicted: every valid box is missed
        out[:, 2] = int(np.sum(gt_flag == 0))
        return out
    rows = np.arange(T)
    for i in np.nonzero(gt_flag != -1)[0]:
        cand = ~assigned & ~under & (pred_flag != -1)[None, :] & (iou[i] > iou_threshold)[None, :]
        c0 = cand & (pred_flag == 0)[None, :]
        c1 = cand & (pred_flag == 1)[None, :]
        has0, has1 = c0.any(1), c1.any(1)
        det = np.where(has0, np.argmax(np.where(c0, iou[i][None, :], -np.inf), axis=1), np.argmax(c1, axis=1))
        detected = has0 | has1
        ignore = detected & ((gt_flag[i] == 1) | (~has0))                   # ~has0 & detected: matched an ignored pred
        tp = detected & ~ignore
        out[:, 0] += tp
        out[:, 2] += (~detected) & (gt_flag[i] == 0)
        sel = rows[detected]
        assigned[sel, det[detected]] = True
    out[:, 1] = (~(assigned | (pred_flag == -1)[None, :] | (pred_flag == 1)[None, :] | under)).sum(1)
    return out


def get_evaluation_results(gt_annos, pred_annos, classes, num_pr_points=50, num_parts=100, ious=None):
    """-> (result string, {'AP_<class>/<difficulty>': value, ..., 'AP_mean/<difficulty>': value})."""
    assert len(gt_annos) == len(pred_annos), 'the number of GT must match predictions'
    classes = list(classes)
    if any(c in classes for c in ('Car', 'Bus', 'Truck')):
        assert all(c in classes for c in ('Car', 'Bus', 'Truck')), 'Car/Bus/Truck must all exist for vehicle detection'
    classes = ['Vehicle'] + [c for c in classes if c not in ('Car', 'Bus', 'Truck')]
    n = len(gt_annos)
    if ious is None:
        ious = compute_iou3d(gt_annos, pred_annos, compute_split_parts(n, num_parts))
    scores = [np.asarray(p['score'], np.float64) for p in pred_annos]
    precision = np.zeros([len(classes), 4, num_pr_points + 1])
    recall = np.zeros_like(precision)
    with np.errstate(invalid='ignore', divide='ignore'):
        for ci, cur in enumerate(classes):
            thr = SUPERCLASS_IOU_THRESHOLDS[cur]
            for di in range(4):
                flags = [filter_data(gt_annos[s], pred_annos[s], di, cur) for s in range(n)]
                num_valid = sum(int(np.sum(gf == 0)) for gf, _ in flags)
                acc = [accumulate_scores(ious[s], scores[s], flags[s][0], flags[s][1], thr) for s in range(n)]
                ths = get_thresholds(np.concatenate(acc, 0), num_valid, num_pr_points)
                cm = np.zeros([len(ths), 3])
                for s in range(n):
                    if len(ths):
                        cm += compute_statistics_all(ious[s], scores[s], flags[s][0], flags[s][1], ths, thr)
                nt = len(ths)
                recall[ci, di, :nt] = cm[:, 0] / (cm[:, 0] + cm[:, 2])
                precision[ci, di, :nt] = cm[:, 0] / (cm[:, 0] + cm[:, 1])
                for ti in range(nt):
                    precision[ci, di, ti] = np.max(precision[ci, di, ti:], axis=-1)
                    recall[ci, di, ti] = np.max(recall[ci, di, ti:], axis=-1)
    AP = 0
    for i in range(1, precision.shape[-1]):            # the reference's summation order (evaluation.py:125-128)
        AP = AP + precision[..., i]
    AP = AP / num_pr_points * 100
    ret_dict = {}
    ret_str = '\n|AP@%-9s|' % str(num_pr_points) + ''.join('%-12s|' % d for d in DIFFICULTIES) + '\n'
    for ci, cur in enumerate(classes):
        ret_str += '|%-12s|' % cur
        for di, d in enumerate(DIFFICULTIES):
            ret_dict['AP_' + cur + '/' + d] = AP[ci, di]
            ret_str += '%-12.2f|' % AP[ci, di]
        ret_str += '\n'
    mAP = np.mean(AP, axis=0)
    ret_str += '|%-12s|' % 'mAP'
    for di, d in enumerate(DIFFICULTIES):
        ret_dict['AP_mean/' + d] = mAP[di]
        ret_str += '%-12.2f|' % mAP[di]
    return ret_str + '\n', ret_dict
