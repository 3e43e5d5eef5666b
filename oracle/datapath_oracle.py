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


def prepare_pair(points, points_prev, pose_cur, pose_prev, params, perm, pc_range, ego_radius=2.0, align=True):
    """One sample: remove ego points, align the previous frame, joint augmentation, range crop on x / y
    (common_utils.mask_points_by_range), shuffle of the combined array with `perm` (an index permutation of the kept
    points, previous frame first), split.  Returns (points_prev [n0,4], points [n1,4]) float32."""
    cur = remove_ego_points(points, ego_radius)
    prv = remove_ego_points(points_prev, ego_radius)
    if align:
        prv = convert_prv_frame_to_cur(prv, pose_prev, pose_cur)
    both = np.vstack((np.hstack((prv, np.zeros((prv.shape[0], 1)))), np.hstack((cur, np.ones((cur.shape[0], 1))))))
    both = augment(both, params)                                       # the group-id column rides along untouched
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
