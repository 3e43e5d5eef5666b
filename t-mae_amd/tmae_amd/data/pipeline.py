"""On-device two-frame data path: what ONCETemporalDataset.__getitem__ / prepare_data and
DatasetTemplate.collate_batch do to a batch of (current, previous) scans before the model sees them
(once_temporal_dataset.py:139-330, data_augmentor.py:55-142, data_processor.py:77-102, dataset.py:190-239) --
ego-point removal, pose alignment, joint flip / rotation / scaling, range crop, shuffle, sample-index column -- as
HIP launches on the raw scans instead of numpy on dataloader workers.

The random draws are made on the host with `np.random` in the reference's call order (flip per axis, rotation enable
+ angle, scaling enable + factor, one permutation of the kept points of both frames), so a run seeded like the
reference reproduces its batches (tests/golden/D1).  One host sync per batch (the kept-point counts)."""

import numpy as np
import torch

from .. import ops


def quat_to_matrix(q):
    """Rotation matrix of a scalar-last quaternion (x, y, z, w), float64 (scipy Rotation.from_quat(q).as_matrix())."""
    x, y, z, w = (np.asarray(q, np.float64) / np.linalg.norm(np.asarray(q, np.float64)))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def prev_to_cur_transform(pose_prv, pose_cur):
    """once_utils.convert_prv_frame_to_cur (once_utils.py:4-29) as its two affine steps: (r1t1 [12] | None, m2 [12] |
    None).  ONCE poses are (qx, qy, qz, qw, tx, ty, tz); an all-zero pose means "static": that step is skipped."""
    pose_prv, pose_cur = np.asarray(pose_prv, np.float64), np.asarray(pose_cur, np.float64)
    r1t1 = None
    if np.any(pose_prv):
        r1t1 = np.concatenate([quat_to_matrix(pose_prv[:4]).reshape(-1), pose_prv[4:]])
    m2 = None
    if np.any(pose_cur):
        M = np.zeros((4, 4))
        M[:3, :3] = quat_to_matrix(pose_cur[:4])
        M[:3, 3] = pose_cur[4:]
        M[3, 3] = 1
        m2 = np.linalg.inv(M)[:3].reshape(-1).copy()
    return r1t1, m2


class TemporalPairPipeline:
    def __init__(self, dataset_cfg, training=True, ego_radius=2.0):
        self.pc_range = [float(v) for v in dataset_cfg.POINT_CLOUD_RANGE]
        self.align = bool(dataset_cfg.get('ALIGN_TWO_FRAMES', True))
        self.training = training
        self.ego_radius = float(ego_radius)
        self.flip_axes, self.flip_prob = [], 0.0
        self.rot_prob, self.rot_range = 0.0, [0.0, 0.0]
        self.scale_prob, self.scale_range = 0.0, [1.0, 1.0]
        aug = dataset_cfg.get('DATA_AUGMENTOR', None)
        disabled = set(aug.get('DISABLE_AUG_LIST', [])) if aug is not None else set()
        self.aug_order = []
        for c in (aug.AUG_CONFIG_LIST if (aug is not None and training) else []):
            if c.NAME in disabled:
                continue
            if c.NAME == 'random_world_flip':
                self.flip_axes, self.flip_prob = list(c.ALONG_AXIS_LIST), float(c.PROBABILITY)
            elif c.NAME == 'random_world_rotation':
                self.rot_prob, self.rot_range = float(c.PROBABILITY), [float(v) for v in c.WORLD_ROT_ANGLE]
            elif c.NAME == 'random_world_scaling':
                self.scale_prob, self.scale_range = float(c.PROBABILITY), [float(v) for v in c.WORLD_SCALE_RANGE]
            elif c.NAME == 'gt_sampling':
                continue                                  # label-database pasting: fine-tune only, not built
            else:
                raise NotImplementedError(f'augmentation {c.NAME}')
            self.aug_order.append(c.NAME)
        self.shuffle = False
        for p in dataset_cfg.DATA_PROCESSOR:
            if p.NAME == 'shuffle_points':
                self.shuffle = bool(p.SHUFFLE_ENABLED['train' if training else 'test'])

    def draw(self):
        """np.random calls of the three world augmentations in the reference's order (only those configured)."""
        flips, rot, scale = [], 0.0, 1.0
        for name in self.aug_order:
            if name == 'random_world_flip':
                for axis in self.flip_axes:
                    if np.random.choice([False, True], replace=False, p=[1 - self.flip_prob, self.flip_prob]):
                        flips.append(axis)
            elif name == 'random_world_rotation':
                en = np.random.choice([False, True], replace=False, p=[1 - self.rot_prob, self.rot_prob])
                rr = self.rot_range if en else [0.0, 0.0]
                rot = np.random.uniform(rr[0], rr[1])
            elif name == 'random_world_scaling':
                en = np.random.choice([False, True], replace=False, p=[1 - self.scale_prob, self.scale_prob])
                sr = self.scale_range if en else [1.0, 1.0]
                scale = np.random.uniform(sr[0], sr[1])
        return dict(flips=flips, rot=float(rot), scale=float(scale))

    def __call__(self, samples, device, params=None, perms=None):
        """samples: list of dict(points [n,4], points_prev [n,4] (numpy or tensors), pose, pose_prev (7 floats each,
        optional)).  Returns the collated batch_dict {'points', 'points_prev' [N, 5] on `device`, 'batch_size'}.
        `params` / `perms` override the random draws (parity tests)."""
        launched = []
        for b, s in enumerate(samples):
            par = params[b] if params is not None else self.draw()
            ang = torch.tensor([par['rot']], dtype=torch.float64).float()             # rotate_points_along_z: fp32 angle
            cosa, sina = float(torch.cos(ang)), float(torch.sin(ang))
            r1t1 = m2 = None
            if self.align and 'pose' in s and 'pose_prev' in s:
                r1t1, m2 = prev_to_cur_transform(s['pose_prev'], s['pose'])
            frames = []
            for key, xf in (('points_prev', (r1t1, m2)), ('points', (None, None))):
                pts = torch.as_tensor(s[key], dtype=torch.float32).to(device, non_blocking=True)
                frames.append(ops.frame_prepare(pts, xf[0], xf[1], self.ego_radius, 'x' in par['flips'], 'y' in par['flips'],
                                                cosa, sina, np.float32(par['scale']), self.pc_range, b))
            launched.append(frames)
        counts = torch.stack([f[1] for fr in launched for f in fr]).cpu().view(-1, 2).tolist()     # the one host sync
        outs = {'points_prev': [], 'points': []}
        for b, (fr, (n0, n1)) in enumerate(zip(launched, counts)):
            prv, cur = fr[0][0][:n0], fr[1][0][:n1]
            if self.shuffle:
                perm = perms[b] if perms is not None else np.random.permutation(n0 + n1)   # data_processor.py:92-96
                perm = torch.as_tensor(perm, dtype=torch.long, device=device)
                sel_prv, sel_cur = perm[perm < n0], perm[perm >= n0] - n0                 # order inside the shuffled array
                prv, cur = prv[sel_prv], cur[sel_cur]
            outs['points_prev'].append(prv)
            outs['points'].append(cur)
        return {'points': torch.cat(outs['points'], 0), 'points_prev': torch.cat(outs['points_prev'], 0),
                'batch_size': len(samples)}
