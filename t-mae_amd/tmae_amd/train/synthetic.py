"""Deterministic synthetic ONCE-shape two-frame scans: log-uniform radius, uniform azimuth,
ground-hugging z, uniform intensity; cropped to |x|,|y| <= 74.88 like mask_points_by_range
(pcdet/utils/common_utils.py:124-127); previous frame = current translated by (0.5, 0.1) m and re-cropped.
Batches follow the collate layout of the reference (pcdet/datasets/dataset.py:203-208): rows
[batch_idx, x, y, z, intensity] float32 under the keys points / points_prev, plus batch_size."""
import numpy as np


def synth_frame_pair(n_points, batch_size, seed, shift=(0.5, 0.1), limit=74.88, extra_features=0):
    cur, prv = [], []
    for b in range(batch_size):
        rng = np.random.default_rng(seed * 64 + b)
        r = np.exp(rng.uniform(np.log(2.0), np.log(105.0), n_points))
        th = rng.uniform(0, 2 * np.pi, n_points)
        x, y = r * np.cos(th), r * np.sin(th)
        z = rng.normal(-1.7, 0.3, n_points) + 3 * rng.uniform(0, 1, n_points) ** 4
        it = rng.uniform(0, 1, n_points)
        cols = [np.full(n_points, b), x, y, z, it] + [rng.uniform(0, 1, n_points) for _ in range(extra_features)]
        pts = np.stack(cols, axis=1).astype(np.float32)          # Waymo-shape scans carry elongation as a 5th feature
        keep = (np.abs(pts[:, 1]) <= limit) & (np.abs(pts[:, 2]) <= limit)
        cur.append(pts[keep])
        q = pts.copy()
        q[:, 1] += np.float32(shift[0])
        q[:, 2] += np.float32(shift[1])
        keep = (np.abs(q[:, 1]) <= limit) & (np.abs(q[:, 2]) <= limit)
        prv.append(q[keep])
    return np.concatenate(cur), np.concatenate(prv)


class _PointFeatureEncoder:
    def __init__(self, n):
        self.num_point_features = n   # ONCE: x, y, z, intensity, group_id = 5 (once_temporal_dataset.yaml); Waymo: 6


def synth_gt_boxes(batch_size, n_boxes, seed, limit=74.0):
    """Synthetic ONCE-style labels [B, n_boxes, 8]: x, y, z, dx, dy, dz, heading, class (1..5; zero rows = padding),
    the layout of `gt_boxes` after collate (dataset.py:203-221)."""
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


class SyntheticTemporalDataset:
    """Stands where ONCETemporalDataset stands for build_network: exposes class_names,
    point_feature_encoder.num_point_features, grid_size, point_cloud_range, voxel_size
    (detector3d_template.py:22,46-53) and yields collated batches."""

    def __init__(self, dataset_cfg, class_names, n_points=120000, batch_size=8, rank=0, length=1 << 30,
                 num_point_features=5, n_boxes=0):
        self.n_boxes = n_boxes                          # > 0: also yield synthetic `gt_boxes` (fine-tune path)
        self.class_names = class_names
        self.point_feature_encoder = _PointFeatureEncoder(num_point_features)
        self.point_cloud_range = np.array(dataset_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        vs = None
        for p in dataset_cfg.DATA_PROCESSOR:
            if p.NAME == 'calculate_grid_size':
                vs = p.VOXEL_SIZE
        self.voxel_size = list(vs)
        g = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(vs)
        self.grid_size = np.round(g).astype(np.int64)          # data_processor.py:166-171
        self.n_points, self.batch_size, self.rank, self.length = n_points, batch_size, rank, length

    def __len__(self):
        return self.length

    def batch(self, iteration):
        pts, prv = synth_frame_pair(self.n_points, self.batch_size, 1000 * self.rank + iteration,
                                    limit=float(self.point_cloud_range[3]),
                                    extra_features=self.point_feature_encoder.num_point_features - 5)
        out = {'points': pts, 'points_prev': prv, 'batch_size': self.batch_size}
        if self.n_boxes > 0:
            out['gt_boxes'] = synth_gt_boxes(self.batch_size, self.n_boxes, 7919 * self.rank + iteration,
                                             limit=float(self.point_cloud_range[3]) - 1.0)
        return out


class SyntheticEvalLoader:
    """Evaluation split of the synthetic data with the dataset / dataloader surface eval_one_epoch needs
    (tools/eval_utils/eval_utils.py:24-161; once_temporal_dataset.py:552-600): `.dataset` (class_names, __len__,
    generate_prediction_dicts, evaluation) and iteration over collated batches carrying `frame_id` and `gt_boxes`.
    Sample s is generated from seed s alone; rank r of W iterates over the samples r, r + W, ... (DistributedSampler
    order without shuffling, datasets/__init__.py:22-42)."""

    def __init__(self, base: SyntheticTemporalDataset, num_samples, batch_size, rank=0, world=1, n_boxes=12):
        self.base, self.num_samples, self.batch_size, self.rank, self.world = base, num_samples, batch_size, rank, world
        self.n_boxes = n_boxes
        self.class_names = base.class_names
        self.dataset = self

    def __len__(self):
        return self.num_samples

    def _sample(self, s):
        pts, prv = synth_frame_pair(self.base.n_points, 1, 100000 + s, limit=float(self.base.point_cloud_range[3]),
                                    extra_features=self.base.point_feature_encoder.num_point_features - 5)
        gt = synth_gt_boxes(1, self.n_boxes, 200000 + s, limit=float(self.base.point_cloud_range[3]) - 1.0)[0]
        return pts, prv, gt

    def gt_anno(self, s):
        gt = self._sample(s)[2]
        gt = gt[np.abs(gt).sum(1) != 0]
        return {'name': np.array(self.class_names)[gt[:, 7].astype(np.int64) - 1], 'boxes_3d': gt[:, :7].astype(np.float64)}

    def __iter__(self):
        mine = list(range(self.rank, self.num_samples, self.world))
        for b0 in range(0, len(mine), self.batch_size):
            ids = mine[b0:b0 + self.batch_size]
            cur, prv, gts = [], [], []
            for k, s in enumerate(ids):
                p, q, g = self._sample(s)
                p[:, 0], q[:, 0] = k, k
                cur.append(p), prv.append(q), gts.append(g)
            yield {'points': np.concatenate(cur), 'points_prev': np.concatenate(prv), 'batch_size': len(ids),
                   'gt_boxes': np.stack(gts), 'frame_id': np.array([str(s) for s in ids])}

    @staticmethod
    def generate_prediction_dicts(batch_dict, pred_dicts, class_names, output_path=None):
        """once_temporal_dataset.py:552-585: name / score / boxes_3d / frame_id per sample."""
        if output_path is not None:
            raise NotImplementedError('the reference does not write ONCE result files either (:583-584)')
        annos = []
        for index, box_dict in enumerate(pred_dicts):
            scores = box_dict['pred_scores'].detach().cpu().numpy()
            boxes = box_dict['pred_boxes'].detach().cpu().numpy()
            labels = box_dict['pred_labels'].detach().cpu().numpy()
            if scores.shape[0] == 0:
                anno = {'name': np.zeros(0), 'score': np.zeros(0), 'boxes_3d': np.zeros((0, 7))}
            else:
                anno = {'name': np.array(class_names)[labels - 1], 'score': scores, 'boxes_3d': boxes}
            anno['frame_id'] = batch_dict['frame_id'][index]
            annos.append(anno)
        return annos

    def evaluation(self, det_annos, class_names, **kwargs):
        """once_temporal_dataset.py:587-600: ONCE AP of the detections against the split's annotations."""
        from ..eval import get_evaluation_results
        gts = [self.gt_anno(int(a['frame_id'])) for a in det_annos]
        return get_evaluation_results(gts, [dict(a) for a in det_annos], list(class_names))
