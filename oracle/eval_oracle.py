"""CPU restatement of the ONCE evaluation the fine-tune path ends in.  TEST INFRASTRUCTURE: only
tests/ import this file.

Follows pcdet/datasets/once_temporal/once_eval/evaluation.py:
  get_evaluation_results :26-157   get_thresholds :159-181   accumulate_scores :183-217   compute_statistics :219-269
  filter_data :271-330             iou3d_kernel_with_heading :360-396   compute_iou3d :398-419
and once_eval/eval_utils.py (compute_split_parts :3-11, overall_distance_filter :32-48).

Pinned against the reference's own functions run in the build container (oracle/gen_golden_eval.py -> tests/golden/
G5_once_eval.npz): everything above is the reference's pure-Python / numpy logic and is reproduced exactly.
PARITY UNPINNED for one piece: the BEV rectangle intersection `rotate_iou_gpu_eval` (once_eval/iou_utils.py:309-343) is
a numba.cuda kernel -- no CUDA and no numba in the image -- so `bev_intersection` restates its published definition
(clockwise-positive angle, iou_utils.py:218-242; intersection polygon area, :245-259) by float64 convex clipping, and
the pinning run feeds THIS function to the reference in the kernel's place.
"""
import math

import numpy as np

IOU_THRESHOLDS = {'Car': 0.7, 'Bus': 0.7, 'Truck': 0.7, 'Pedestrian': 0.3, 'Cyclist': 0.5}
SUPERCLASS_IOU_THRESHOLDS = {'Vehicle': 0.7, 'Pedestrian': 0.3, 'Cyclist': 0.5}


def _corners_clockwise(b):
    """rbbox_to_corners (iou_utils.py:218-242): [x, y, dx, dy, angle], rotated CLOCKWISE for a positive angle."""
    x, y, dx, dy, a = (float(v) for v in b)
    c, s = math.cos(a), math.sin(a)
    pts = [(-dx / 2, -dy / 2), (-dx / 2, dy / 2), (dx / 2, dy / 2), (dx / 2, -dy / 2)]
    return [(c * px + s * py + x, -s * px + c * py + y) for px, py in pts]


def _clip(poly, a, b, sign):
    out = []
    n = len(poly)
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        sp = sign * ((b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]))
        sq = sign * ((b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0]))
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def _area2(poly):
    s = 0.0
    for i in range(len(poly)):
        p, q = poly[i], poly[(i + 1) % len(poly)]
        s += p[0] * q[1] - q[0] * p[1]
    return s


def bev_intersection(boxes, query_boxes):
    """rotate_iou_gpu_eval(boxes [N,5], query_boxes [K,5], criterion=2): intersection AREAS [N,K] (float64)."""
    boxes, query_boxes = np.asarray(boxes, np.float64), np.asarray(query_boxes, np.float64)
    out = np.zeros((len(boxes), len(query_boxes)))
    qc = [_corners_clockwise(b) for b in query_boxes]
    for i, a in enumerate(boxes):
        pa = _corners_clockwise(a)
        for j, cb in enumerate(qc):
            if math.hypot(a[0] - query_boxes[j][0], a[1] - query_boxes[j][1]) > \
                    (math.hypot(a[2], a[3]) + math.hypot(query_boxes[j][2], query_boxes[j][3])) / 2:
                continue
            sign = 1.0 if _area2(cb) > 0 else -1.0
            poly = list(pa)
            for e in range(4):
                if not poly:
                    break
                poly = _clip(poly, cb[e], cb[(e + 1) % 4], sign)
            if len(poly) >= 3:
                out[i, j] = abs(_area2(poly)) / 2
    return out


def iou3d_with_heading(gt_boxes, pred_boxes, bev_fn=bev_intersection):
    """iou3d_kernel_with_heading (evaluation.py:360-396): [x, y, z, dx, dy, dz, rot] boxes."""
    gt_boxes, pred_boxes = np.asarray(gt_boxes, np.float64), np.asarray(pred_boxes, np.float64)
    inter2d = bev_fn(gt_boxes[:, [0, 1, 3, 4, 6]], pred_boxes[:, [0, 1, 3, 4, 6]])
    gt_max, gt_min = gt_boxes[:, [2]] + gt_boxes[:, [5]] * 0.5, gt_boxes[:, [2]] - gt_boxes[:, [5]] * 0.5
    pr_max, pr_min = pred_boxes[:, [2]] + pred_boxes[:, [5]] * 0.5, pred_boxes[:, [2]] - pred_boxes[:, [5]] * 0.5
    inter_h = np.minimum(gt_max, pr_max.T) - np.maximum(gt_min, pr_min.T)
    inter_h[inter_h <= 0] = 0
    inter3d = inter2d * inter_h
    gt_vol = gt_boxes[:, [3]] * gt_boxes[:, [4]] * gt_boxes[:, [5]]
    pr_vol = pred_boxes[:, [3]] * pred_boxes[:, [4]] * pred_boxes[:, [5]]
    iou = inter3d / (gt_vol + pr_vol.T - inter3d)
    diff = np.abs(gt_boxes[:, [6]] - pred_boxes[:, [6]].T)
    rev = 2 * np.pi - diff
    diff[diff >= np.pi] = rev[diff >= np.pi]
    iou[diff > np.pi / 2] = 0
    return iou


def compute_split_parts(num_samples, num_parts):
    part, rem = num_samples // num_parts, num_samples % num_parts
    if part == 0:
        return [num_samples]
    return [part] * num_parts + ([rem] if rem else [])


def overall_distance_filter(boxes, level):
    ignore = np.ones(boxes.shape[0], dtype=bool)
    dist = np.sqrt(np.sum(boxes[:, 0:3] * boxes[:, 0:3], axis=1))
    if level == 0:
        flag = np.ones(boxes.shape[0], dtype=bool)
    elif level == 1:
        flag = dist < 30
    elif level == 2:
        flag = (dist >= 30) & (dist < 50)
    elif level == 3:
        flag = dist >= 50
    else:
        raise AssertionError(level)
    ignore[flag] = False
    return ignore


def filter_data(gt_anno, pred_anno, difficulty_level, class_name, use_superclass=True):
    def flags(anno):
        names = np.asarray(anno['name'])
        f = np.zeros(len(names), dtype=np.int64)
        if use_superclass and class_name == 'Vehicle':
            rej = np.logical_or(names == 'Pedestrian', names == 'Cyclist')
        else:
            rej = names != class_name
        f[rej] = -1
        f[overall_distance_filter(np.asarray(anno['boxes_3d'], np.float64).reshape(-1, 7), difficulty_level)] = 1
        return f
    return flags(gt_anno), flags(pred_anno)


def get_thresholds(scores, num_gt, num_pr_points):
    eps = 1e-6
    scores = np.sort(np.asarray(scores, np.float64))[::-1]
    recall_level = 0
    thresholds = []
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < len(scores) - 1 else l_recall
        if (r_recall + l_recall < 2 * recall_level) and i < len(scores) - 1:
            continue
        thresholds.append(score)
        recall_level += 1 / num_pr_points
        while r_recall + l_recall + eps > 2 * recall_level:
            thresholds.append(score)
            recall_level += 1 / num_pr_points
    return thresholds


def accumulate_scores(iou, pred_scores, gt_flag, pred_flag, iou_threshold):
    num_gt, num_pred = iou.shape
    assigned = np.zeros(num_pred, bool)
    acc = []
    for i in range(num_gt):
        if gt_flag[i] == -1:
            continue
        det_idx, detected_score = -1, -1
        for j in range(num_pred):
            if pred_flag[j] == -1 or assigned[j]:
                continue
            if iou[i, j] > iou_threshold and pred_scores[j] > detected_score:
                det_idx, detected_score = j, pred_scores[j]
        if detected_score == -1 and gt_flag[i] == 0:
            pass
        elif detected_score != -1 and (gt_flag[i] == 1 or pred_flag[det_idx] == 1):
            assigned[det_idx] = True
        elif detected_score != -1:
            acc.append(pred_scores[det_idx])
            assigned[det_idx] = True
    return np.asarray(acc, np.float64)


def compute_statistics(iou, pred_scores, gt_flag, pred_flag, score_threshold, iou_threshold):
    num_gt, num_pred = iou.shape
    assigned = np.zeros(num_pred, bool)
    under = pred_scores < score_threshold
    tp = fp = fn = 0
    for i in range(num_gt):
        if gt_flag[i] == -1:
            continue
        det_idx, detected, best, to_ignore = -1, False, 0, False
        for j in range(num_pred):
            if pred_flag[j] == -1 or assigned[j] or under[j]:
                continue
            v = iou[i, j]
            if v > iou_threshold and (v > best or to_ignore) and pred_flag[j] == 0:
                best, det_idx, detected, to_ignore = v, j, True, False
            elif v > iou_threshold and not detected and pred_flag[j] == 1:
                det_idx, detected, to_ignore = j, True, True
        if not detected and gt_flag[i] == 0:
            fn += 1
        elif detected and (gt_flag[i] == 1 or pred_flag[det_idx] == 1):
            assigned[det_idx] = True
        elif detected:
            tp += 1
            assigned[det_idx] = True
    for j in range(num_pred):
        if not (assigned[j] or pred_flag[j] == -1 or pred_flag[j] == 1 or under[j]):
            fp += 1
    return tp, fp, fn


def compute_iou3d(gt_annos, pred_annos, split_parts, bev_fn=bev_intersection):
    ious, idx = [], 0
    for n in split_parts:
        gpart, ppart = gt_annos[idx:idx + n], pred_annos[idx:idx + n]
        gb = np.concatenate([np.asarray(a['boxes_3d'], np.float64).reshape(-1, 7) for a in gpart], 0)
        pb = np.concatenate([np.asarray(a['boxes_3d'], np.float64).reshape(-1, 7) for a in ppart], 0)
        full = iou3d_with_heading(gb, pb, bev_fn)
        gi = pi = 0
        for g, p in zip(gpart, ppart):
            ng, npd = len(g['name']), len(p['name'])
            ious.append(full[gi:gi + ng, pi:pi + npd])
            gi += ng
            pi += npd
        idx += n
    return ious


def get_evaluation_results(gt_annos, pred_annos, classes, num_pr_points=50, num_parts=100, ious=None,
                           bev_fn=bev_intersection):
    """The reference's defaults: use_superclass, 'Overall&Distance', ap_with_heading.  Returns (ret_dict, AP array,
    ious)."""
    assert len(gt_annos) == len(pred_annos)
    classes = list(classes)
    if any(c in classes for c in ('Car', 'Bus', 'Truck')):
        assert all(c in classes for c in ('Car', 'Bus', 'Truck'))
    classes = ['Vehicle'] + [c for c in classes if c not in ('Car', 'Bus', 'Truck')]
    n = len(gt_annos)
    if ious is None:
        ious = compute_iou3d(gt_annos, pred_annos, compute_split_parts(n, num_parts), bev_fn)
    diffs = ['overall', '0-30m', '30-50m', '50m-inf']
    precision = np.zeros([len(classes), 4, num_pr_points + 1])
    recall = np.zeros_like(precision)
    with np.errstate(invalid='ignore', divide='ignore'):
        for ci, cur in enumerate(classes):
            thr = SUPERCLASS_IOU_THRESHOLDS[cur]
            for di in range(4):
                all_scores, gflags, pflags, num_valid = [], [], [], 0
                for s in range(n):
                    gf, pf = filter_data(gt_annos[s], pred_annos[s], di, cur)
                    gflags.append(gf)
                    pflags.append(pf)
                    num_valid += int(np.sum(gf == 0))
                    all_scores.append(accumulate_scores(ious[s], np.asarray(pred_annos[s]['score'], np.float64), gf, pf, thr))
                ths = get_thresholds(np.concatenate(all_scores, 0), num_valid, num_pr_points)
                cm = np.zeros([len(ths), 3])
                for s in range(n):
                    sc = np.asarray(pred_annos[s]['score'], np.float64)
                    for ti, t in enumerate(ths):
                        cm[ti] += compute_statistics(ious[s], sc, gflags[s], pflags[s], t, thr)
                for ti in range(len(ths)):
                    recall[ci, di, ti] = cm[ti, 0] / (cm[ti, 0] + cm[ti, 2])
                    precision[ci, di, ti] = cm[ti, 0] / (cm[ti, 0] + cm[ti, 1])
                for ti in range(len(ths)):
                    precision[ci, di, ti] = np.max(precision[ci, di, ti:], axis=-1)
                    recall[ci, di, ti] = np.max(recall[ci, di, ti:], axis=-1)
    AP = 0
    for i in range(1, precision.shape[-1]):
        AP = AP + precision[..., i]
    AP = AP / num_pr_points * 100
    ret = {}
    for ci, cur in enumerate(classes):
        for di, d in enumerate(diffs):
            ret['AP_' + cur + '/' + d] = AP[ci, di]
    mAP = np.mean(AP, axis=0)
    for di, d in enumerate(diffs):
        ret['AP_mean/' + d] = mAP[di]
    return ret, AP, ious


def synth_annos(num_samples, seed, class_names=('Car', 'Bus', 'Truck', 'Pedestrian', 'Cyclist')):
    """Synthetic ground truth + detections with every case of the matching logic: jittered true positives, duplicates
    (lower score on the same object), wrong-class and flipped-heading boxes, far-range objects, misses, clutter, an
    empty-prediction and an empty-ground-truth sample."""
    rng = np.random.default_rng(seed)
    sizes = {'Car': (4.4, 1.9, 1.6), 'Bus': (11.0, 2.9, 3.4), 'Truck': (7.5, 2.6, 3.0), 'Pedestrian': (0.8, 0.8, 1.75),
             'Cyclist': (2.0, 0.8, 1.6)}
    gts, preds = [], []
    for s in range(num_samples):
        ng = 0 if s == 1 else int(rng.integers(6, 14))
        names = rng.choice(class_names, ng)
        boxes = np.zeros((ng, 7))
        r = rng.uniform(3, 72, ng)
        th = rng.uniform(0, 2 * np.pi, ng)
        boxes[:, 0], boxes[:, 1], boxes[:, 2] = r * np.cos(th), r * np.sin(th), rng.normal(-1.0, 0.3, ng)
        for i, nme in enumerate(names):
            boxes[i, 3:6] = np.array(sizes[nme]) * rng.uniform(0.9, 1.1, 3)
        boxes[:, 6] = rng.uniform(-np.pi, np.pi, ng)
        gts.append({'name': np.array(names, dtype='<U10'), 'boxes_3d': boxes})
        pb, pn, ps = [], [], []
        if s != 2:
            for i in range(ng):
                u = rng.uniform()
                if u < 0.15:
                    continue                                               # missed
                b = boxes[i].copy()
                b[:3] += rng.normal(0, 0.12, 3) * (1 + (names[i] in ('Bus', 'Truck')))
                b[3:6] *= rng.uniform(0.93, 1.07, 3)
                b[6] += rng.normal(0, 0.06)
                nme = names[i]
                if u < 0.25:
                    b[6] += np.pi                                          # flipped heading: no match
                elif u < 0.33:
                    nme = rng.choice([c for c in class_names if c != names[i]])
                pb.append(b), pn.append(nme), ps.append(rng.uniform(0.2, 0.98))
                if u > 0.85:                                               # duplicate detection of the same object
                    b2 = b.copy()
                    b2[:2] += rng.normal(0, 0.1, 2)
                    pb.append(b2), pn.append(nme), ps.append(rng.uniform(0.1, 0.6))
            for _ in range(int(rng.integers(2, 7))):                      # clutter
                nme = rng.choice(class_names)
                b = np.zeros(7)
                rr, tt = rng.uniform(3, 72), rng.uniform(0, 2 * np.pi)
                b[:3] = rr * np.cos(tt), rr * np.sin(tt), rng.normal(-1.0, 0.3)
                b[3:6] = np.array(sizes[nme]) * rng.uniform(0.9, 1.1, 3)
                b[6] = rng.uniform(-np.pi, np.pi)
                pb.append(b), pn.append(nme), ps.append(rng.uniform(0.05, 0.5))
        preds.append({'name': np.array(pn, dtype='<U10') if pn else np.zeros(0, dtype='<U10'),
                      'score': np.round(np.array(ps, np.float64), 3),       # rounded: ties between scores occur
                      'boxes_3d': np.array(pb).reshape(-1, 7)})
    return gts, preds
