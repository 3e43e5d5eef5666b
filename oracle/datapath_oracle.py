"""CPU restatement of the two-frame DATA PATH in front of the T-MAE step: what
ONCETemporalDataset.__getitem__ / prepare_data do to one (current, previous) scan pair before collate, for the
pre-training config (t_mae_ssl.yaml: random_world_flip / rotation / scaling, range crop, shuffle).

TEST INFRASTRUCTURE (see tmae_oracle.py).  Random draws are explicit arguments (`params`), produced by
`draw_params` with the reference's own call order on `np.random`, so a seeded run of the reference is reproduced
bit for bit.  Pinned against the reference functions by oracle/gen_golden_datapath.py (fixture tests/golden/D1)."""
import numpy as np
import torch


def remove_ego_points(points, center_radius):
    """once_temporal_dataset.remove_ego_points (once_temporal_dataset.py:43-45)."""
    mask = ~((np.abs(points[:, 0]) < center_radius) & (np.abs(points[:, 1]) < center_radius))
    return points[mask]


def quat_to_matrix(q):
    """scipy Rotation.from_quat(q).as_matrix() for a scalar-last quaternion (x, y, z, w), float64."""
    x, y, z, w = (np.asarray(q, np.float64) / np.linalg.norm(np.asarray(q, np.float64)))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def prev_to_cur_transform(pose_prv, pose_cur):
    """The two steps of once_utils.convert_prv_frame_to_cur (once_utils.py:4-29) as matrices: returns
    (R1, t1, M2) with p_global = p R1^T + t1 (or identity when pose_prv is all zeros ... see the quirk below) and
    p_cur = [p_global, 1] M2^T (M2 = inverse of the current pose, or None).  Quirk kept: `np.any(pose) == 0` is
    True only when EVERY entry is zero."""
    pose_prv, pose_cur = np.asarray(pose_prv, np.float64), np.asarray(pose_cur, np.float64)
    step1 = None if (np.any(pose_prv) == 0) else (quat_to_matrix(pose_prv[:4]), pose_prv[4:].copy())
    M2 = None
    if not (np.any(pose_cur) == 0):
        M = np.zeros((4, 4))
        M[:3, :3] = quat_to_matrix(pose_cur[:4])
        M[:3, 3] = pose_cur[4:]
        M[3, 3] = 1
        M2 = np.linalg.inv(M)
    return step1, M2


def convert_prv_frame_to_cur(pc_prv, pose_prv, pose_cur):
    step1, M2 = prev_to_cur_transform(pose_prv, pose_cur)
    g = pc_prv[:, :3] if step1 is None else np.dot(pc_prv[:, :3], step1[0].T) + step1[1]
    if M2 is not None:
        g = np.dot(np.concatenate([g, np.ones((g.shape[0], 1))], axis=-1), M2.T)[:, :3]
    return np.concatenate([g[:, :3], pc_prv[:, 3:]], axis=-1)


def draw_params(cfg_aug, n_total_after_crop=None):
    """The np.random calls of DataAugmentor.random_world_{flip,rotation,scaling} in the reference's order
    (data_augmentor.py:55-142).  Call with the global np.random seeded like the reference run."""
    flips = []
    for axis in cfg_aug['flip_axes']:
        if np.random.choice([False, True], replace=False, p=[1 - cfg_aug['flip_prob'], cfg_aug['flip_prob']]):
            flips.append(axis)
    en = np.random.choice([False, True], replace=False, p=[1 - cfg_aug['rot_prob'], cfg_aug['rot_prob']])
    rr = cfg_aug['rot_range'] if en else [0.0, 0.0]
    rot = np.random.uniform(rr[0], rr[1])
    en = np.random.choice([False, True], replace=False, p=[1 - cfg_aug['scale_prob'], cfg_aug['scale_prob']])
    sr = cfg_aug['scale_range'] if en else [1.0, 1.0]
    scale = np.random.uniform(sr[0], sr[1])
    return dict(flips=flips, rot=rot, scale=scale)


def augment(points, params):
    """points [n, 3+C] float64/float32 -> float32: flips (data_augmentor.py:68-82), rotation about z in fp32
    (common_utils.rotate_points_along_z, common_utils.py:99-121: fp32 matmul with (cos, sin, -sin, cos)), scaling."""
    p = np.array(points, copy=True)
    for axis in params['flips']:
        if axis == 'x':
            p[:, 1] = -p[:, 1]
        else:
            p[:, 0] = -p[:, 0]
    t = torch.from_numpy(p).float()
    ang = torch.from_numpy(np.array([params['rot']])).float()
    c, s = torch.cos(ang), torch.sin(ang)
    z, o = ang.new_zeros(1), ang.new_ones(1)
    rm = torch.stack((c, s, z, -s, c, z, z, z, o), dim=1).view(-1, 3, 3).float()
    rot = torch.matmul(t[None, :, 0:3], rm)[0]
    p = torch.cat((rot, t[:, 3:]), dim=-1).numpy()
    p[:, :3] *= params['scale']
    return p


def prepare_pair(points, points_prev, pose_cur, pose_prev, params, perm, pc_range, ego_radius=2.0, align=True,
                 augment_points=True):
    """One sample: remove ego points, align the previous frame, joint augmentation, range crop on x / y
    (common_utils.mask_points_by_range), shuffle of the combined array with `perm` (an index permutation of the kept
    points, previous frame first), split.  Returns (points_prev [n0,4], points [n1,4]) float32."""
    cur = remove_ego_points(points, ego_radius)
    prv = remove_ego_points(points_prev, ego_radius)
    if align:
        prv = convert_prv_frame_to_cur(prv, pose_prev, pose_cur)
    both = np.vstack((np.hstack((prv, np.zeros((prv.shape[0], 1)))), np.hstack((cur, np.ones((cur.shape[0], 1))))))
    if augment_points:             # training; the test-mode path (_combine_two_pcs, once_temporal_dataset.py:214-218) has none
        both = augment(both, params)                                   # the group-id column rides along untouched
    m = (both[:, 0] >= pc_range[0]) & (both[:, 0] <= pc_range[3]) & (both[:, 1] >= pc_range[1]) & (both[:, 1] <= pc_range[4])
    both = both[m]
    if perm is not None:
        both = both[perm]
    return both[both[:, -1] == 0, :-1], both[both[:, -1] == 1, :-1]


def collate(samples):
    """DatasetTemplate.collate_batch for points / points_prev (dataset.py:190-207): prepend the sample index."""
    out = {}
    for key in ('points_prev', 'points'):
        out[key] = np.concatenate([np.pad(s[key], ((0, 0), (1, 0)), mode='constant', constant_values=i)
                                   for i, s in enumerate(samples)], axis=0)
    out['batch_size'] = len(samples)
    return out


# ----------------------------------------------------------------------------------------------- labels (fine-tune)

def limit_period(val, offset=0.5, period=np.pi):
    """common_utils.limit_period (common_utils.py:85-88): fp32 (check_numpy_to_torch casts to float)."""
    v = torch.from_numpy(np.asarray(val)).float()
    return (v - torch.floor(v / period + offset) * period).numpy()


def rotate_z_fp32(xyz, angle):
    """common_utils.rotate_points_along_z on one batch of [n, 3] rows: fp32 matmul with (cos, sin; -sin, cos)."""
    t = torch.from_numpy(np.asarray(xyz)).float()
    ang = torch.from_numpy(np.array([angle])).float()
    c, s = torch.cos(ang), torch.sin(ang)
    z, o = ang.new_zeros(1), ang.new_ones(1)
    rm = torch.stack((c, s, z, -s, c, z, z, z, o), dim=1).view(-1, 3, 3).float()
    return torch.matmul(t[None, :, 0:3], rm)[0].numpy()


def augment_boxes(gt_boxes, params):
    """gt_boxes [n, 7] (x, y, z, dx, dy, dz, heading) through random_world_flip / rotation / scaling
    (data_augmentor.py:55-142), in the array's own dtype (the reference edits annos['boxes_3d'] in place; the rotated
    centres come back from an fp32 matmul)."""
    b = np.array(gt_boxes, copy=True)
    for axis in params['flips']:
        if axis == 'x':
            b[:, 1] = -b[:, 1]
            b[:, 6] = -b[:, 6]
        else:
            b[:, 0] = -b[:, 0]
            b[:, 6] = -(b[:, 6] + np.pi)
    b[:, 0:3] = rotate_z_fp32(b[:, 0:3], params['rot'])
    b[:, 6] += params['rot']
    b[:, :6] *= params['scale']
    return b


def boxes_to_corners_3d(boxes):
    """box_utils.boxes_to_corners_3d (box_utils.py:28-53), fp32."""
    b = torch.from_numpy(np.asarray(boxes)).float()
    template = b.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                             [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    corners = b[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    c, s = torch.cos(b[:, 6]), torch.sin(b[:, 6])
    z, o = b.new_zeros(b.shape[0]), b.new_ones(b.shape[0])
    rm = torch.stack((c, s, z, -s, c, z, z, z, o), dim=1).view(-1, 3, 3)
    corners = torch.matmul(corners, rm) + b[:, None, 0:3]
    return corners.numpy()


def mask_boxes_outside_range(boxes, limit_range, min_num_corners=1):
    """box_utils.mask_boxes_outside_range_numpy (box_utils.py:56-72)."""
    limit_range = np.asarray(limit_range, dtype=np.float32)
    corners = boxes_to_corners_3d(np.asarray(boxes)[:, 0:7])
    mask = ((corners >= limit_range[0:3]) & (corners <= limit_range[3:6])).all(axis=2)
    return mask.sum(axis=1) >= min_num_corners


def prepare_labels(gt_boxes, gt_names, class_names, params, pc_range, training=True, remove_outside=True):
    """The label side of ONCETemporalDataset.prepare_data (once_temporal_dataset.py:246-330) without gt_sampling:
    world augmentation of the boxes (training), heading wrapped to [-pi, pi) (DataAugmentor.forward,
    data_augmentor.py:243-246), boxes of classes outside `class_names` dropped (gt_boxes_mask / keep_arrays_by_name),
    class index appended as an 8th column, boxes with no corner inside the range removed (training,
    data_processor.py:85-89).  Returns gt_boxes [k, 8] (None: the sample has no box left and the reference draws
    another index, once_temporal_dataset.py:199-202)."""
    b = np.array(gt_boxes, copy=True)
    names = np.asarray(gt_names)
    if training:
        keep = np.array([n in class_names for n in names], dtype=np.bool_)
        b = augment_boxes(b, params)
        b[:, 6] = limit_period(b[:, 6], offset=0.5, period=2 * np.pi)
        b, names = b[keep], names[keep]
    sel = np.array([i for i, n in enumerate(names) if n in class_names], dtype=np.int64)
    b, names = b[sel], names[sel]
    cls = np.array([class_names.index(n) + 1 for n in names], dtype=np.int32)
    b = np.concatenate((b, cls.reshape(-1, 1).astype(np.float32)), axis=1)
    if training and remove_outside:
        b = b[mask_boxes_outside_range(b, pc_range, 1)] if len(b) else b
    if training and len(b) == 0:
        return None
    return b


def collate_boxes(box_list):
    """DatasetTemplate.collate_batch for gt_boxes (dataset.py:208-213): zero-padded [B, max_n, 8] float32."""
    mx = max(len(x) for x in box_list)
    out = np.zeros((len(box_list), mx, box_list[0].shape[-1]), dtype=np.float32)
    for k, x in enumerate(box_list):
        out[k, :len(x), :] = x
    return out


# ----------------------------------------------------------------------------------------------- sample index logic

def generate_intervals(start_id, end_id, max_interval):
    """DatasetTemplate._generate_intervals (dataset.py:240-252): (first, last-exclusive) per frame of a sequence."""
    return [(max(start_id, last - max_interval), last) for last in range(start_id + 1, end_id + 1)]


def build_intervals(infos, scan_window, split):
    """ONCETemporalDataset.include_once_data (once_temporal_dataset.py:72-108), including its boundary behaviour: a
    sequence's intervals are emitted when the NEXT sequence starts (or at the last info), the frame at index i then
    opens the next run, and the very last info closes the last run one frame early."""
    intervals, seq_id, start_id = [], '', 0
    for i, info in enumerate(infos):
        if seq_id != info['sequence_id'] or i == len(infos) - 1:
            seq_id = info['sequence_id']
            intervals.extend(generate_intervals(start_id, i, scan_window))
            start_id = i
    if split in ('train', 'val'):
        intervals = [iv for iv in intervals if 'annos' in infos[iv[1] - 1]]
    return intervals


def pick_pair(interval, scan_window, fixed_gap=-1):
    """(idx, idx_prev) of ONCETemporalDataset.__getitem__ (once_temporal_dataset.py:142-156); draws from np.random
    exactly when the reference does."""
    num_frames = interval[1] - interval[0]
    idx = interval[1] - 1
    sampling_window = int(np.floor(scan_window / 3))
    if fixed_gap == -1:
        if num_frames == 1:
            idx_prev = idx
        else:
            idx_prev = np.random.choice(np.arange(interval[0], interval[0] + sampling_window), 1)[0]
    else:
        idx_prev = max(interval[0], idx - fixed_gap)
    return idx, int(idx_prev)


# ----------------------------------------------------------------------------------------------- gt_sampling
# DataBaseSampler (pcdet/datasets/augmentor/database_sampler.py) in front of the world augmentations of the fine-tune
# config.  The Python logic is the reference's (pinned by tests/golden/D3_gt_sampling.npz, produced by the reference's own
# DataBaseSampler / ONCETemporalDataset run with stand-ins for its two COMPILED geometry helpers); those two helpers --
# iou3d_nms_cuda.boxes_iou_bev_cpu and roiaware_pool3d_cuda.points_in_boxes_cpu -- are restated here from their sources
# (roiaware_pool3d.cpp:119-140 is in the reference tree; iou3d_cpu.cpp includes cuda.h and cannot be built here):
# PARITY UNPINNED for the two helpers themselves.  The sampler only asks whether two boxes overlap at all (iou == 0).

def points_in_boxes_cpu(points_xyz, boxes):
    """roiaware_pool3d.cpp:119-168 (check_pt_in_box3d_cpu): [nb, n] int mask.  fp32 point / box values; cos / sin of the
    double angle rounded to fp32; the products in fp32; the three comparisons in double, MARGIN = (float)1e-2."""
    p = np.asarray(points_xyz, np.float32)
    b = np.asarray(boxes, np.float32)
    out = np.zeros((b.shape[0], p.shape[0]), np.int32)
    margin = np.float64(np.float32(1e-2))
    for i in range(b.shape[0]):
        cx, cy, cz, dx, dy, dz, rz = (b[i, k] for k in range(7))
        zin = ~(np.abs(p[:, 2] - cz).astype(np.float64) > np.float64(dz) / 2.0)
        ca, sa = np.float32(np.cos(-np.float64(rz))), np.float32(np.sin(-np.float64(rz)))
        sx, sy = p[:, 0] - cx, p[:, 1] - cy
        lx = sx * ca + sy * (-sa)
        ly = sx * sa + sy * ca
        inside = (np.abs(lx).astype(np.float64) < np.float64(dx) / 2.0 + margin) & \
                 (np.abs(ly).astype(np.float64) < np.float64(dy) / 2.0 + margin)
        out[i] = (zin & inside).astype(np.int32)
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """[na, nb] bool: do the two rotated BEV rectangles overlap with positive area (separating-axis test, float64)?
    Stands where boxes_bev_iou_cpu(...) != 0 stands in DataBaseSampler.__call__ (database_sampler.py:240-244)."""
    def corners(b):
        b = np.asarray(b, np.float64)
        c, s = np.cos(b[:, 6]), np.sin(b[:, 6])
        hx, hy = b[:, 3] / 2, b[:, 4] / 2
        loc = np.array([[1, 1], [1, -1], [-1, -1], [-1, 1]], np.float64)
        x = b[:, None, 0] + loc[None, :, 0] * hx[:, None] * c[:, None] - loc[None, :, 1] * hy[:, None] * s[:, None]
        y = b[:, None, 1] + loc[None, :, 0] * hx[:, None] * s[:, None] + loc[None, :, 1] * hy[:, None] * c[:, None]
        axes = np.stack([np.stack([c, s], -1), np.stack([-s, c], -1)], 1)          # [n, 2 axes, 2]
        return np.stack([x, y], -1), axes
    ca, axa = corners(boxes_a)
    cb, axb = corners(boxes_b)
    na, nb = ca.shape[0], cb.shape[0]
    over = np.ones((na, nb), bool)
    for axes, own_a in ((axa, True), (axb, False)):
        for k in range(2):
            ax = axes[:, k]                                                         # [n, 2]
            if own_a:
                pa = np.einsum('ipc,ic->ip', ca, ax)[:, None, :].repeat(nb, 1)      # [na, nb, 4]
                pb = np.einsum('jpc,ic->ijp', cb, ax)
            else:
                pa = np.einsum('ipc,jc->ijp', ca, ax)
                pb = np.einsum('jpc,jc->jp', cb, ax)[None, :, :].repeat(na, 0)
            over &= (pa.max(-1) > pb.min(-1) + 1e-9) & (pb.max(-1) > pa.min(-1) + 1e-9)
    return over


class DataBaseSamplerOracle:
    """DataBaseSampler.__init__ / sample_with_fixed_number / __call__ / add_sampled_boxes_to_scene
    (database_sampler.py:13-259) without shared memory, road planes or fake-lidar boxes (the ONCE config has none)."""

    def __init__(self, db_infos, cfg, class_names, read_points):
        self.class_names = list(class_names)
        self.cfg = cfg
        self.read_points = read_points                       # info -> [n, NUM_POINT_FEATURES] float32 (the .bin crop)
        self.db_infos = {c: list(db_infos.get(c, [])) for c in class_names}
        for name_num in cfg.get('filter_by_min_points', []):
            name, mn = name_num.split(':')
            if int(mn) > 0 and name in self.db_infos:
                self.db_infos[name] = [i for i in self.db_infos[name] if i['num_points_in_gt'] >= int(mn)]
        self.limit_whole_scene = bool(cfg.get('limit_whole_scene', False))
        self.fade_epoch = int(cfg.get('fade_epoch', 0))
        self.sample_groups, self.sample_class_num = {}, {}
        for x in cfg['sample_groups']:
            name, num = x.split(':')
            if name not in class_names:
                continue
            self.sample_class_num[name] = num
            self.sample_groups[name] = {'sample_num': num, 'pointer': len(self.db_infos[name]),
                                        'indices': np.arange(len(self.db_infos[name]))}

    def _sample(self, name, grp):
        num, pointer, indices = int(grp['sample_num']), grp['pointer'], grp['indices']
        if pointer >= len(self.db_infos[name]):
            indices = np.random.permutation(len(self.db_infos[name]))
            pointer = 0
        out = [self.db_infos[name][i] for i in indices[pointer:pointer + num]]
        grp['pointer'], grp['indices'] = pointer + num, indices
        return out

    def __call__(self, gt_boxes, gt_names, cur_epoch=0, total_epochs=1):
        """Returns (sampled_boxes [k, 7] float32, sampled infos) -- what __call__ appends to the scene."""
        if total_epochs < self.fade_epoch + cur_epoch + 1:
            return np.zeros((0, 7), np.float32), []
        existed = np.asarray(gt_boxes)
        names = np.asarray(gt_names).astype(str)
        n0 = existed.shape[0]
        chosen = []
        for name, grp in self.sample_groups.items():
            if self.limit_whole_scene:
                grp['sample_num'] = str(int(self.sample_class_num[name]) - int(np.sum(name == names)))
            if int(grp['sample_num']) > 0:
                cand = self._sample(name, grp)
                sb = np.stack([c['box3d_lidar'] for c in cand], 0).astype(np.float32)
                o1 = boxes_overlap_bev(sb[:, :7], existed[:, :7]) if existed.shape[0] > 0 else None
                o2 = boxes_overlap_bev(sb[:, :7], sb[:, :7])
                o2[np.arange(len(sb)), np.arange(len(sb))] = False
                if o1 is None:
                    o1 = o2
                valid = np.nonzero(~(o1.any(1) | o2.any(1)))[0]
                chosen += [cand[i] for i in valid]
                existed = np.concatenate((existed, sb[valid]), axis=0)
        return existed[n0:, :], chosen

    def object_points(self, infos):
        pts = []
        for info in infos:
            p = np.array(self.read_points(info), np.float32, copy=True)
            p[:, :3] += info['box3d_lidar'][:3]
            pts.append(p)
        return np.concatenate(pts, 0) if pts else None


def prepare_pair_sampled(points, points_prev, pose_cur, pose_prev, gt_boxes, gt_names, class_names, sampler, draw, pc_range,
                         extra_width=(0.0, 0.0, 0.0), ego_radius=2.0, align=True, shuffle=True):
    """One training sample with gt_sampling at the head of the augmentor queue (prepare_data, once_temporal_dataset.py:246-330;
    DataAugmentor.forward; _combine_two_pcs_with_delimiter / _attach_group_ids :225-244): returns (points_prev, points,
    gt_boxes [k, 8]) or None (no box left).  `draw()` = the world-augmentation draws, called AFTER the sampler's own
    np.random use, `shuffle`: np.random.permutation over the kept rows."""
    cur = remove_ego_points(points, ego_radius)
    prv = remove_ego_points(points_prev, ego_radius)
    if align:
        prv = convert_prv_frame_to_cur(prv, pose_prev, pose_cur)
    boxes = np.array(gt_boxes, copy=True)
    names = np.asarray(gt_names)
    mask = np.array([n in class_names for n in names], dtype=np.bool_)
    sboxes, infos = sampler(boxes, names)
    added = None
    if len(infos) > 0:
        obj = sampler.object_points(infos)
        large = np.array(sboxes[:, 0:7], np.float32, copy=True)
        large[:, 3:6] += np.asarray(extra_width, np.float32)[None, :]
        keep_p = points_in_boxes_cpu(np.asarray(prv, np.float32)[:, :3], large).sum(0) == 0
        keep_c = points_in_boxes_cpu(np.asarray(cur, np.float32)[:, :3], large).sum(0) == 0
        prv, cur = np.asarray(prv, np.float32)[keep_p], np.asarray(cur, np.float32)[keep_c]     # remove_points_in_boxes3d: fp32 from here
        added = obj
        boxes = np.concatenate([boxes[mask], sboxes], 0)
        names = np.concatenate([names[mask], np.array([i['name'] for i in infos])], 0)
        mask = np.ones(len(boxes), np.bool_)
    params = draw()
    # combined order after _attach_group_ids: [added (group 1), added (group 0), prev (0), cur (1)]
    parts = []
    if added is not None:
        parts += [np.hstack((added, np.ones((len(added), 1)))), np.hstack((added, np.zeros((len(added), 1))))]
    parts += [np.hstack((prv, np.zeros((prv.shape[0], 1)))), np.hstack((cur, np.ones((cur.shape[0], 1))))]
    both = augment(np.vstack(parts), params)
    m = (both[:, 0] >= pc_range[0]) & (both[:, 0] <= pc_range[3]) & (both[:, 1] >= pc_range[1]) & (both[:, 1] <= pc_range[4])
    both = both[m]
    # labels: everything already class-filtered when boxes were pasted (gt_boxes_mask was consumed by the sampler)
    b = augment_boxes(boxes, params)
    b[:, 6] = limit_period(b[:, 6], offset=0.5, period=2 * np.pi)
    if len(infos) == 0:
        b, names = b[mask], names[mask]
    sel = np.array([i for i, n in enumerate(names) if n in class_names], dtype=np.int64)
    b, names = b[sel], names[sel]
    cls = np.array([class_names.index(n) + 1 for n in names], dtype=np.int32)
    b = np.concatenate((b, cls.reshape(-1, 1).astype(np.float32)), axis=1)
    b = b[mask_boxes_outside_range(b, pc_range, 1)] if len(b) else b
    if shuffle:
        both = both[np.random.permutation(both.shape[0])]
    if len(b) == 0:
        return None
    return both[both[:, -1] == 0, :-1], both[both[:, -1] == 1, :-1], b
