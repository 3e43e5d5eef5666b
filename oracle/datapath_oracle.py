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
