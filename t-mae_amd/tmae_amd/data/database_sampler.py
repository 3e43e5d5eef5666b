"""gt_sampling: label-database pasting in front of the world augmentations of the fine-tune recipe
(pcdet/datasets/augmentor/database_sampler.py:13-259; t_mae.yaml DATA_AUGMENTOR.AUG_CONFIG_LIST[0]).

Host side (a dozen boxes per sample): the database infos (`DB_INFO_PATH` pickles: per class a list of
{name, path, box3d_lidar, num_points_in_gt, ...}), `PREPARE.filter_by_min_points` / `filter_by_difficulty`, the per-class
sample groups with their pointer / permutation (`np.random.permutation` exactly when the reference draws), `LIMIT_WHOLE_SCENE`,
`FADE_EPOCH`, and the collision rule -- a candidate is pasted only if its BEV rectangle overlaps neither a box of the scene nor
another candidate of its class batch (the reference asks `boxes_bev_iou_cpu(...) == 0`; here a separating-axis test, which answers
the same question without the intersection polygon).  The object crops (`gt_database/*.bin`, NUM_POINT_FEATURES floats per point,
stored relative to the box centre) are read on the host and shifted to the box.

Device side (`TemporalPairPipeline`): the scene points of BOTH frames that fall into a pasted box (enlarged by
`REMOVE_EXTRA_WIDTH`) are dropped inside `tmae_frame_prepare_boxes`, after the alignment and before the augmentation, with the
arithmetic of the reference's `points_in_boxes_cpu` (roiaware_pool3d.cpp:119-140); the pasted points go in front of both frames
(`_attach_group_ids`, once_temporal_dataset.py:225-244) and through the same flip / rotation / scaling / crop.

Not built: `USE_ROAD_PLANE`, `DATABASE_WITH_FAKELIDAR` (KITTI only), `USE_SHARED_MEMORY` (a host-memory optimisation of the
reference's multi-process loader)."""
import pickle
from pathlib import Path

import numpy as np


def bev_rectangles_overlap(boxes_a, boxes_b):
    """[na, nb] bool: positive-area overlap of rotated BEV rectangles (x, y, dx, dy, heading = columns 0, 1, 3, 4, 6),
    separating-axis test in float64."""
    a, b = np.asarray(boxes_a, np.float64), np.asarray(boxes_b, np.float64)
    na, nb = a.shape[0], b.shape[0]
    if na == 0 or nb == 0:
        return np.zeros((na, nb), bool)

    def geom(x):
        c, s = np.cos(x[:, 6]), np.sin(x[:, 6])
        ux, uy = np.stack([c, s], 1), np.stack([-s, c], 1)                       # the rectangle's own axes
        return x[:, 0:2], ux, uy, x[:, 3] / 2, x[:, 4] / 2
    ca, uxa, uya, hxa, hya = geom(a)
    cb, uxb, uyb, hxb, hyb = geom(b)
    d = cb[None, :, :] - ca[:, None, :]                                           # [na, nb, 2]
    sep = np.zeros((na, nb), bool)
    for own_a, axes in ((True, (uxa, uya)), (False, (uxb, uyb))):
        for ax in axes:
            axis = ax[:, None, :] if own_a else ax[None, :, :]                    # broadcast to [na, nb, 2]
            dist = np.abs((d * axis).sum(-1))
            ra = hxa[:, None] * np.abs((uxa[:, None, :] * axis).sum(-1)) + hya[:, None] * np.abs((uya[:, None, :] * axis).sum(-1))
            rb = hxb[None, :] * np.abs((uxb[None, :, :] * axis).sum(-1)) + hyb[None, :] * np.abs((uyb[None, :, :] * axis).sum(-1))
            sep |= dist >= ra + rb - 1e-9
    return ~sep


def removal_table(boxes, extra_width):
    """[nb, 8] float64 for tmae_frame_prepare_boxes from sampled boxes [nb, 7] (enlarge_box3d + the constants of
    check_pt_in_box3d_cpu): cx, cy, cz, (float)cos(-h), (float)sin(-h), dz / 2, dx / 2 + 1e-2f, dy / 2 + 1e-2f."""
    b = np.array(boxes[:, 0:7], np.float32, copy=True)
    b[:, 3:6] += np.asarray(extra_width, np.float32)[None, :]
    t = np.zeros((b.shape[0], 8), np.float64)
    t[:, 0:3] = b[:, 0:3]
    t[:, 3] = np.cos(-b[:, 6].astype(np.float64)).astype(np.float32)
    t[:, 4] = np.sin(-b[:, 6].astype(np.float64)).astype(np.float32)
    margin = np.float64(np.float32(1e-2))
    t[:, 5] = b[:, 5].astype(np.float64) / 2.0
    t[:, 6] = b[:, 3].astype(np.float64) / 2.0 + margin
    t[:, 7] = b[:, 4].astype(np.float64) / 2.0 + margin
    return t


class DataBaseSampler:
    def __init__(self, root_path, sampler_cfg, class_names, logger=None):
        self.root_path, self.cfg, self.class_names, self.logger = Path(root_path), sampler_cfg, list(class_names), logger
        for opt in ('USE_ROAD_PLANE', 'DATABASE_WITH_FAKELIDAR', 'USE_SHARED_MEMORY'):
            if sampler_cfg.get(opt, False):
                raise NotImplementedError(f'gt_sampling {opt} is not used by the ONCE recipes')
        self.db_infos = {c: [] for c in self.class_names}
        for rel in sampler_cfg.DB_INFO_PATH:
            with open(self.root_path.resolve() / rel, 'rb') as f:
                infos = pickle.load(f)
            for c in self.class_names:
                self.db_infos[c].extend(infos.get(c, []))
        for func, val in sampler_cfg.PREPARE.items():
            if func == 'filter_by_min_points':
                self._filter_min_points(val)
            elif func == 'filter_by_difficulty':
                self.db_infos = {k: [i for i in v if i['difficulty'] not in val] for k, v in self.db_infos.items()}
            else:
                raise NotImplementedError(f'gt_sampling PREPARE.{func}')
        self.fade_epoch = int(sampler_cfg.get('FADE_EPOCH', 0))
        self.limit_whole_scene = bool(sampler_cfg.get('LIMIT_WHOLE_SCENE', False))
        self.remove_points = bool(sampler_cfg.get('REMOVE_POINTS', True))
        self.extra_width = [float(v) for v in sampler_cfg.REMOVE_EXTRA_WIDTH]
        self.num_point_features = int(sampler_cfg.NUM_POINT_FEATURES)
        self.sample_groups, self.sample_class_num = {}, {}
        for x in sampler_cfg.SAMPLE_GROUPS:
            name, num = x.split(':')
            if name not in self.class_names:
                continue
            self.sample_class_num[name] = num
            # pointer at the end: the first use draws a permutation (database_sampler.py:47-51,126-129)
            self.sample_groups[name] = {'sample_num': num, 'pointer': len(self.db_infos[name]),
                                        'indices': np.arange(len(self.db_infos[name]))}

    def _filter_min_points(self, spec):
        for name_num in spec:
            name, mn = name_num.split(':')
            if int(mn) > 0 and name in self.db_infos:
                kept = [i for i in self.db_infos[name] if i['num_points_in_gt'] >= int(mn)]
                if self.logger is not None:
                    self.logger.info('Database filter by min points %s: %d => %d' % (name, len(self.db_infos[name]), len(kept)))
                self.db_infos[name] = kept

    def _take(self, name, grp):
        num, pointer, indices = int(grp['sample_num']), grp['pointer'], grp['indices']
        if pointer >= len(self.db_infos[name]):
            indices = np.random.permutation(len(self.db_infos[name]))
            pointer = 0
        picked = [self.db_infos[name][i] for i in indices[pointer:pointer + num]]
        grp['pointer'], grp['indices'] = pointer + num, indices
        return picked

    def sample(self, gt_boxes, gt_names, cur_epoch=0, total_epochs=1):
        """The candidates that survive the collision rule for a scene with `gt_boxes` [n, 7+] (ALL its boxes, also those of
        classes that are filtered out later: they block pasting too): (boxes [k, 7] float32, infos)."""
        if total_epochs < self.fade_epoch + cur_epoch + 1:
            return np.zeros((0, 7), np.float32), []
        existed = np.asarray(gt_boxes)
        names = np.asarray(gt_names).astype(str)
        n0 = existed.shape[0]
        chosen = []
        for name, grp in self.sample_groups.items():
            if self.limit_whole_scene:
                grp['sample_num'] = str(int(self.sample_class_num[name]) - int(np.sum(name == names)))
            if int(grp['sample_num']) <= 0:
                continue
            cand = self._take(name, grp)
            if not cand:
                continue
            sb = np.stack([c['box3d_lidar'] for c in cand], 0).astype(np.float32)
            among = bev_rectangles_overlap(sb[:, :7], sb[:, :7])
            among[np.arange(len(sb)), np.arange(len(sb))] = False
            with_scene = bev_rectangles_overlap(sb[:, :7], existed[:, :7]) if existed.shape[0] > 0 else among
            valid = np.nonzero(~(with_scene.any(1) | among.any(1)))[0]
            chosen += [cand[i] for i in valid]
            existed = np.concatenate((existed, sb[valid].astype(existed.dtype)), axis=0)
        return np.asarray(existed[n0:, :7], np.float32), chosen

    def object_points(self, infos):
        """[n, NUM_POINT_FEATURES] float32: the crops of `infos`, shifted to their boxes, in paste order."""
        out = []
        for info in infos:
            p = np.fromfile(str(self.root_path / info['path']), dtype=np.float32).reshape(-1, self.num_point_features)
            p[:, :3] += np.asarray(info['box3d_lidar'][:3])          # float32 += float64: summed in double, rounded once (as the reference)
            out.append(p)
        return np.concatenate(out, 0)
