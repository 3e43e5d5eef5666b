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


def _rotate_z_fp32(xyz, angle):
    """[n, 3] rows about z by `angle`, as the reference rotates box centres and corners: an fp32 matmul on the host
    (common_utils.rotate_points_along_z, common_utils.py:99-121).  angle: a float or an [n] array (one per row)."""
    t = torch.as_tensor(np.asarray(xyz)).float()
    ang = torch.as_tensor(np.atleast_1d(np.asarray(angle))).float()
    c, s_ = torch.cos(ang), torch.sin(ang)
    z, o = torch.zeros_like(ang), torch.ones_like(ang)
    rm = torch.stack((c, s_, z, -s_, c, z, z, z, o), dim=1).view(-1, 3, 3)
    if rm.shape[0] == 1:
        return torch.matmul(t[None, :, 0:3], rm)[0].numpy()
    return torch.matmul(t.view(rm.shape[0], -1, 3), rm).numpy()


class TemporalPairPipeline:
    """points / points_prev of a batch on the device, gt_boxes (a few dozen rows per sample) on the host.

    reference_rng_order=True keeps the np.random stream of a single-process run of the reference (per sample: previous-
    frame pick, augmentation draws, shuffle permutation) at the price of one host sync per SAMPLE (the permutation's
    length is the number of points that survive the crop); the default draws all augmentations first and the
    permutations after the batch's one sync."""

    def __init__(self, dataset_cfg, training=True, ego_radius=2.0, class_names=None, logger=None,
                 reference_rng_order=False, root_path=None):
        self.pc_range = [float(v) for v in dataset_cfg.POINT_CLOUD_RANGE]
        self.align = bool(dataset_cfg.get('ALIGN_TWO_FRAMES', True))
        self.training = training
        self.class_names = list(class_names) if class_names is not None else None
        self.reference_rng_order = bool(reference_rng_order)
        self.ego_radius = float(ego_radius)
        self.flip_axes, self.flip_prob = [], 0.0
        self.rot_prob, self.rot_range = 0.0, [0.0, 0.0]
        self.scale_prob, self.scale_range = 0.0, [1.0, 1.0]
        aug = dataset_cfg.get('DATA_AUGMENTOR', None)
        disabled = set(aug.get('DISABLE_AUG_LIST', [])) if aug is not None else set()
        self.aug_order = []
        self.sampler, self.cur_epoch, self.total_epochs = None, 0, 1 << 30
        for pos_, c in enumerate(aug.AUG_CONFIG_LIST if (aug is not None and training) else []):
            if c.NAME in disabled:
                continue
            if c.NAME == 'random_world_flip':
                self.flip_axes, self.flip_prob = list(c.ALONG_AXIS_LIST), float(c.PROBABILITY)
            elif c.NAME == 'random_world_rotation':
                self.rot_prob, self.rot_range = float(c.PROBABILITY), [float(v) for v in c.WORLD_ROT_ANGLE]
            elif c.NAME == 'random_world_scaling':
                self.scale_prob, self.scale_range = float(c.PROBABILITY), [float(v) for v in c.WORLD_SCALE_RANGE]
            elif c.NAME == 'gt_sampling':
                # label-database pasting (database_sampler.py), the head of the fine-tune recipe's augmentor queue; it needs
                # the data root (database infos + object crops): without one -- synthetic scans -- it is skipped, loudly
                if self.aug_order:
                    raise NotImplementedError('gt_sampling behind another augmentation (the recipes put it first)')
                if root_path is None or self.class_names is None:
                    if logger is not None:
                        logger.warning('DATA_AUGMENTOR gt_sampling needs a data root with its label database: skipped')
                    continue
                from .database_sampler import DataBaseSampler
                self.sampler = DataBaseSampler(root_path, c, self.class_names, logger)
                continue
            else:
                raise NotImplementedError(f'augmentation {c.NAME}')
            self.aug_order.append(c.NAME)
        self.shuffle, self.remove_outside, self.min_corners = False, False, 1
        for p in dataset_cfg.DATA_PROCESSOR:
            if p.NAME == 'shuffle_points':
                self.shuffle = bool(p.SHUFFLE_ENABLED['train' if training else 'test'])
            elif p.NAME == 'mask_points_and_boxes_outside_range':
                self.remove_outside = bool(p.REMOVE_OUTSIDE_BOXES) and training        # data_processor.py:85
                self.min_corners = int(p.get('min_num_corners', 1))

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

    # ------------------------------------------------------------------ labels (host: a few dozen boxes per sample)
    def prepare_labels(self, gt_boxes, gt_names, par):
        """The label side of prepare_data (once_temporal_dataset.py:246-330; gt_sampling excepted): in training the
        boxes follow the points through flip / rotation / scaling (data_augmentor.py:55-142: y-flip negates the
        heading, x-flip maps it to -(h + pi); centres rotate in fp32; the first six columns scale), headings are
        wrapped to [-pi, pi) in fp32 (data_augmentor.py:243-246); boxes of other classes are dropped, the class index
        (1-based) becomes column 8, and in training boxes without a corner inside POINT_CLOUD_RANGE go
        (data_processor.py:85-89, box_utils.py:56-72).  None = training sample with no box left."""
        if self.class_names is None:
            raise ValueError('TemporalPairPipeline needs class_names to prepare gt_boxes')
        b = np.array(gt_boxes, copy=True)
        names = [str(n) for n in np.asarray(gt_names)]
        known = np.array([n in self.class_names for n in names], dtype=np.bool_)
        if self.training:
            for axis in par['flips']:
                if axis == 'x':
                    b[:, 1], b[:, 6] = -b[:, 1], -b[:, 6]
                else:
                    b[:, 0], b[:, 6] = -b[:, 0], -(b[:, 6] + np.pi)
            b[:, 0:3] = _rotate_z_fp32(b[:, 0:3], par['rot'])
            b[:, 6] += par['rot']
            b[:, :6] *= par['scale']
            h = torch.as_tensor(b[:, 6]).float()
            b[:, 6] = (h - torch.floor(h / (2 * np.pi) + 0.5) * (2 * np.pi)).numpy()
        b = b[known]
        cls = np.array([self.class_names.index(n) + 1 for n, k in zip(names, known) if k], dtype=np.float32)
        b = np.concatenate((b, cls.reshape(-1, 1)), axis=1)
        if self.remove_outside and len(b):
            lim = np.asarray(self.pc_range, dtype=np.float32)
            half = torch.tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1],
                                 [-1, 1, 1]), dtype=torch.float32) / 2
            b32 = torch.as_tensor(b[:, :7]).float()
            corners = (b32[:, None, 3:6].repeat(1, 8, 1) * half[None]).numpy()
            corners = _rotate_z_fp32(corners.reshape(-1, 3), b32[:, 6].numpy()) + b32[:, None, 0:3].numpy()
            inside = ((corners >= lim[0:3]) & (corners <= lim[3:6])).all(axis=2)
            b = b[inside.sum(axis=1) >= self.min_corners]
        if self.training and len(b) == 0:
            return None
        return b

    # ------------------------------------------------------------------ one batch
    def _launch(self, s, par, b, device, paste=None):
        """Frames of one sample on the device: [previous, current(, pasted object points)] as (rows, count) pairs.
        paste = (object points [n, 4] host, removal table [nb, 8] float64 host or None): gt_sampling."""
        ang = torch.tensor([par['rot']], dtype=torch.float64).float()             # rotate_points_along_z: fp32 angle
        cosa, sina = float(torch.cos(ang)), float(torch.sin(ang))
        r1t1 = m2 = None
        if self.align and 'pose' in s and 'pose_prev' in s:
            r1t1, m2 = prev_to_cur_transform(s['pose_prev'], s['pose'])
        rb = None
        if paste is not None and paste[1] is not None:
            rb = torch.from_numpy(paste[1]).to(device, non_blocking=True)
        frames = []
        for key, xf in (('points_prev', (r1t1, m2)), ('points', (None, None))):
            pts = torch.as_tensor(s[key], dtype=torch.float32).to(device, non_blocking=True)
            frames.append(ops.frame_prepare(pts, xf[0], xf[1], self.ego_radius, 'x' in par['flips'], 'y' in par['flips'],
                                            cosa, sina, np.float32(par['scale']), self.pc_range, b, remove_boxes=rb))
        if paste is not None:
            obj = torch.as_tensor(paste[0], dtype=torch.float32).to(device, non_blocking=True)
            frames.append(ops.frame_prepare(obj, None, None, 0.0, 'x' in par['flips'], 'y' in par['flips'], cosa, sina,
                                            np.float32(par['scale']), self.pc_range, b))
        return frames

    def __call__(self, samples, device, params=None, perms=None, resample=None):
        """samples: list of dict(points [n,4], points_prev [n,4] (numpy or tensors), pose, pose_prev (7 floats each;
        absent = no alignment), optionally gt_boxes [k,7] + gt_names [k], frame_id) -- or of callables returning one
        (read lazily, in order).  Returns the collated batch_dict {'points', 'points_prev' [N, 5] on `device`,
        'batch_size'} (+ 'gt_boxes' [B, max_k, 8] float32 numpy, 'frame_id').  `params` / `perms` override the random
        draws (parity tests); `resample()` supplies a replacement for a training sample that lost all its boxes (the
        reference draws a new random index, once_temporal_dataset.py:199-202)."""
        launched, boxes_out, frame_ids, strict_counts = [], [], [], []
        for b, s in enumerate(samples):
            while True:
                if callable(s):
                    s = s()
                gt_boxes, gt_names, paste = s.get('gt_boxes', None), s.get('gt_names', None), None
                if self.sampler is not None and gt_boxes is not None:
                    # gt_sampling first (its np.random use precedes the world-augmentation draws): candidates that collide
                    # with no box of the scene, their points in front of both frames, the scene points inside them removed
                    sboxes, infos = self.sampler.sample(gt_boxes, gt_names, self.cur_epoch, self.total_epochs)
                    if infos:
                        known = np.array([str(n) in self.class_names for n in np.asarray(gt_names)], dtype=np.bool_)
                        gt_boxes = np.concatenate([np.asarray(gt_boxes)[known][:, :7], sboxes], 0)
                        gt_names = np.concatenate([np.asarray(gt_names)[known], np.array([i['name'] for i in infos])], 0)
                        from .database_sampler import removal_table
                        paste = (self.sampler.object_points(infos),
                                 removal_table(sboxes, self.sampler.extra_width) if self.sampler.remove_points else None)
                par = params[b] if params is not None else self.draw()
                boxes = None
                if gt_boxes is not None:
                    boxes = self.prepare_labels(gt_boxes, gt_names, par)
                frames = self._launch(s, par, b, device, paste)
                if self.reference_rng_order:                     # per-sample sync: the permutation is this sample's last draw
                    cnt = [int(v) for v in torch.stack([f[1] for f in frames]).cpu().view(-1).tolist()]
                    n0, n1, na = cnt[0], cnt[1], (cnt[2] if len(cnt) > 2 else 0)
                    perm = None
                    if self.shuffle:
                        perm = perms[b] if perms is not None else np.random.permutation(n0 + n1 + 2 * na)
                    strict_counts.append((n0, n1, na, perm))
                if gt_boxes is not None and boxes is None:
                    if resample is None:
                        raise ValueError(f'sample {b} has no gt box left after class / range filtering and no resample() was given')
                    if self.reference_rng_order:
                        strict_counts.pop()
                    s = resample()
                    continue
                break
            launched.append(frames)
            if boxes is not None:
                boxes_out.append(boxes)
            if 'frame_id' in s:
                frame_ids.append(s['frame_id'])
        if self.reference_rng_order:
            counts = [(n0, n1, na) for n0, n1, na, _ in strict_counts]
        else:
            zero = torch.zeros((1,), dtype=torch.int32, device=device)
            flat = torch.stack([(fr[i][1] if i < len(fr) else zero) for fr in launched for i in range(3)])
            counts = flat.cpu().view(-1, 3).tolist()                                          # the one host sync
        outs = {'points_prev': [], 'points': []}
        for b, (fr, (n0, n1, na)) in enumerate(zip(launched, counts)):
            prv, cur = fr[0][0][:n0], fr[1][0][:n1]
            if na > 0:
                # the pasted points stand in front of BOTH frames (_attach_group_ids): combined order of the reference's
                # array = [pasted (current copy), pasted (previous copy), previous, current]
                obj = fr[2][0][:na]
                prv, cur = torch.cat([obj, prv], 0), torch.cat([obj, cur], 0)
            if self.shuffle:
                if self.reference_rng_order:
                    perm = strict_counts[b][3]
                else:
                    perm = perms[b] if perms is not None else np.random.permutation(n0 + n1 + 2 * na)   # data_processor.py:92-96
                perm = torch.as_tensor(perm, dtype=torch.long, device=device)
                in_prev = (perm >= na) & (perm < 2 * na + n0)
                sel_prv = perm[in_prev] - na                                                  # -> [pasted | previous]
                pc = perm[~in_prev]
                sel_cur = torch.where(pc < na, pc, pc - (na + n0))                            # -> [pasted | current]
                prv, cur = prv[sel_prv], cur[sel_cur]
            outs['points_prev'].append(prv)
            outs['points'].append(cur)
        batch = {'points': torch.cat(outs['points'], 0), 'points_prev': torch.cat(outs['points_prev'], 0),
                 'batch_size': len(launched)}
        if boxes_out:
            if len(boxes_out) != len(launched):
                raise ValueError('either every sample of a batch carries gt_boxes or none does')
            mx = max(len(x) for x in boxes_out)                                           # dataset.py:208-213
            gt = np.zeros((len(boxes_out), mx, boxes_out[0].shape[-1]), dtype=np.float32)
            for k, x in enumerate(boxes_out):
                gt[k, :len(x), :] = x
            batch['gt_boxes'] = gt
        if frame_ids:
            batch['frame_id'] = np.array(frame_ids)
        return batch
