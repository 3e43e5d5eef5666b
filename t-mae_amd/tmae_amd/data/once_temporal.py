"""ONCE two-frame dataset and batch loader in front of the on-device data path.

`ONCETemporalDataset` has the constructor and the surface of the reference's class
(pcdet/datasets/once_temporal/once_temporal_dataset.py:13-137: ImageSets/<split>.txt, INFO_PATH pickles under the data
root, `data/<sequence>/lidar_roof/<frame>.bin` = float32 x 4 per point, interval list per sequence, annotation
filter for the train / val splits) but hands out RAW samples: the two scans as they are on disk, the two poses and
the frame's annotations.  Everything `__getitem__` / `prepare_data` / `collate_batch` then do to them (ego removal,
alignment, joint augmentation, crop, shuffle, collate; boxes through the same flips / rotation / scaling, class filter,
outside-range filter) is `TemporalPairPipeline` -- points on the device in one launch group per frame, the few boxes
of a sample on the host.  `DeviceBatchLoader` stands where the reference's `DataLoader` stands: reader threads
prefetch the `.bin` files of the next batches, the main thread runs the pipeline and yields collated batch_dicts whose
point tensors are already resident in HBM.

The sample-index logic (intervals, previous-frame pick with `np.random.choice`, resampling of samples that end up
without boxes) follows the reference draw for draw; tests/golden/D2_once_dataset.npz holds what the reference's own
class returns on a tiny ONCE-layout directory."""
import pickle
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from .pipeline import TemporalPairPipeline

_SPLITS = ('train', 'val', 'test', 'raw_small', 'raw_medium', 'raw_large')


class _PointFeatureEncoder:
    """num_point_features of POINT_FEATURE_ENCODING (point_feature_encoder.py:13-15,44-47: the used feature list)."""

    def __init__(self, cfg):
        self.used_feature_list = list(cfg.used_feature_list)
        self.src_feature_list = list(cfg.src_feature_list)
        if self.src_feature_list[:3] != ['x', 'y', 'z']:
            raise ValueError('POINT_FEATURE_ENCODING.src_feature_list must start with x, y, z')
        if self.used_feature_list != self.src_feature_list:
            raise NotImplementedError('POINT_FEATURE_ENCODING: used_feature_list != src_feature_list (column selection) '
                                      'is not used by the ONCE configs')
        self.num_point_features = len(self.used_feature_list)


def interval_list(start_id, end_id, max_interval):
    """(first, last-exclusive) frame index range ending at every frame of one sequence run (dataset.py:240-252)."""
    return [(max(start_id, last - max_interval), last) for last in range(start_id + 1, end_id + 1)]


class ONCETemporalDataset:
    def __init__(self, dataset_cfg, class_names, training=True, root_path=None, logger=None):
        self.dataset_cfg, self.class_names, self.training, self.logger = dataset_cfg, list(class_names), training, logger
        self.root_path = Path(root_path if root_path is not None else dataset_cfg.DATA_PATH)
        self.split = dataset_cfg.DATA_SPLIT['train' if training else 'test']
        if self.split not in _SPLITS:
            raise ValueError(f'DATA_SPLIT {self.split!r}: expected one of {_SPLITS}')
        self.point_cloud_range = np.array(dataset_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        self.point_feature_encoder = _PointFeatureEncoder(dataset_cfg.POINT_FEATURE_ENCODING)
        self.voxel_size = None
        for p in dataset_cfg.DATA_PROCESSOR:
            if p.NAME in ('calculate_grid_size', 'transform_points_to_voxels'):
                self.voxel_size = list(p.VOXEL_SIZE)
        if self.voxel_size is None:
            raise ValueError('DATA_PROCESSOR needs calculate_grid_size (VOXEL_SIZE)')
        g = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(self.voxel_size)
        self.grid_size = np.round(g).astype(np.int64)                       # data_processor.py:166-171
        self.align_two_frames = bool(dataset_cfg.get('ALIGN_TWO_FRAMES', False))
        self.scan_window = int(dataset_cfg.get('SCAN_WINDOW', 1))
        self.sampling_window = int(np.floor(self.scan_window / 3))
        self.fixed_gap = int(dataset_cfg.get('FIXED_GAP', -1))
        seq_file = self.root_path / 'ImageSets' / (self.split + '.txt')
        self.sample_seq_list = [x.strip() for x in open(seq_file).readlines()]
        self.once_infos, self.once_intervals = [], []
        self.total_epochs, self.cur_epoch = 0, 0
        self.include_once_data(self.split)

    # ------------------------------------------------------------------ index
    def include_once_data(self, split):
        """once_temporal_dataset.py:72-108, boundary behaviour included: a run's intervals are emitted when the next
        sequence starts (or at the last info); the frame at that index opens the next run."""
        infos = []
        for info_path in self.dataset_cfg.INFO_PATH[split]:
            info_path = self.root_path / info_path
            if not info_path.exists():
                continue
            with open(info_path, 'rb') as f:
                infos.extend(pickle.load(f))
        intervals, seq_id, start_id = [], '', 0
        for i, info in enumerate(infos):
            if seq_id != info['sequence_id'] or i == len(infos) - 1:
                seq_id = info['sequence_id']
                intervals.extend(interval_list(start_id, i, self.scan_window))
                start_id = i
        if self.split in ('train', 'val'):
            intervals = [iv for iv in intervals if 'annos' in infos[iv[1] - 1]]
        self.once_infos.extend(infos)
        self.once_intervals.extend(intervals)
        if self.logger is not None:
            self.logger.info('Total samples for ONCE dataset: %d' % len(intervals))

    def set_epoch(self, epoch):
        self.cur_epoch = epoch

    def __len__(self):
        return len(self.once_intervals)

    def pick(self, index):
        """(idx, idx_prev) into once_infos for sample `index`; draws from np.random exactly when the reference does
        (once_temporal_dataset.py:142-156)."""
        first, last = self.once_intervals[index]
        idx = last - 1
        if self.fixed_gap == -1:
            if last - first == 1:
                idx_prev = idx
            else:
                idx_prev = int(np.random.choice(np.arange(first, first + self.sampling_window), 1)[0])
        else:
            idx_prev = max(first, idx - self.fixed_gap)
        if not (idx_prev <= idx < last):
            raise ValueError(f'sample {index}: previous frame {idx_prev} outside its interval {first, last} '
                             f'(SCAN_WINDOW {self.scan_window} < 3 gives an empty sampling window)')
        return idx, idx_prev

    # ------------------------------------------------------------------ IO
    def get_lidar(self, sequence_id, frame_id):
        path = self.root_path / 'data' / sequence_id / 'lidar_roof' / ('%s.bin' % frame_id)
        return np.fromfile(str(path), dtype=np.float32).reshape(-1, 4)

    def read_pair(self, idx, idx_prev):
        """Raw sample for TemporalPairPipeline: scans as stored, poses only when the alignment applies (two different
        frames and ALIGN_TWO_FRAMES, once_temporal_dataset.py:170-173), annotations of the current frame."""
        info, info_prev = self.once_infos[idx], self.once_infos[idx_prev]
        s = {'points': self.get_lidar(info['sequence_id'], info['frame_id']),
             'points_prev': self.get_lidar(info['sequence_id'], info_prev['frame_id']), 'frame_id': info['frame_id']}
        if self.align_two_frames and info['frame_id'] != info_prev['frame_id']:
            s['pose'], s['pose_prev'] = np.asarray(info['pose'], np.float64), np.asarray(info_prev['pose'], np.float64)
        if 'annos' in info:
            s['gt_names'] = np.asarray(info['annos']['name'])
            s['gt_boxes'] = np.array(info['annos']['boxes_3d'], copy=True)
        return s

    def raw_sample(self, index):
        return self.read_pair(*self.pick(index))

    # ------------------------------------------------------------------ evaluation surface (eval_utils.py:24-161)
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
        """once_temporal_dataset.py:587-600: ONCE AP of the detections against the annotations of the sampled frames."""
        from ..eval import get_evaluation_results
        by_frame = {info['frame_id']: info['annos'] for info in self.once_infos if 'annos' in info}
        gts = [{'name': np.asarray(by_frame[str(a['frame_id'])]['name']),
                'boxes_3d': np.asarray(by_frame[str(a['frame_id'])]['boxes_3d'], np.float64)} for a in det_annos]
        return get_evaluation_results(gts, [dict(a) for a in det_annos], list(class_names))


class EpochSampler:
    """Sample order of one rank: torch's DistributedSampler rule (a permutation seeded with the epoch, padded to a
    multiple of the world size, every world-th index from `rank`) for training, the plain strided order for
    evaluation (pcdet/datasets/__init__.py:22-42)."""

    def __init__(self, n, rank=0, world=1, shuffle=True, seed=0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0
        self.num_samples = (n + world - 1) // world
        self.total_size = self.num_samples * world

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            indices = torch.randperm(self.n, generator=g).tolist()
        else:
            indices = list(range(self.n))
        while len(indices) < self.total_size:
            indices += indices[:self.total_size - len(indices)]
        return iter(indices[self.rank:self.total_size:self.world])


class DeviceBatchLoader:
    """for batch_dict in loader: ...  -- the reference's DataLoader surface (.dataset, .sampler, len()) over
    ONCETemporalDataset + TemporalPairPipeline.  `workers` reader threads fetch the scans of up to `prefetch` batches
    ahead (file IO releases the GIL); the previous-frame picks are drawn in sample order when a batch is scheduled, the
    augmentation draws and permutations when it is processed.  workers = 0 reads synchronously; with
    `pipeline.reference_rng_order` that reproduces the np.random stream of a single-process run of the reference."""

    def __init__(self, dataset, batch_size, sampler, pipeline, device, workers=4, drop_last=False, prefetch=2):
        self.dataset, self.batch_size, self.sampler, self.pipeline = dataset, int(batch_size), sampler, pipeline
        self.device, self.workers, self.drop_last, self.prefetch = device, int(workers), drop_last, max(int(prefetch), 1)

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _resample(self):
        """A sample whose boxes all vanished: the reference draws a new random index (once_temporal_dataset.py:199-202)."""
        return self.dataset.raw_sample(int(np.random.randint(len(self.dataset))))

    def __iter__(self):
        order = list(self.sampler)
        self.pipeline.cur_epoch = self.dataset.cur_epoch                 # gt_sampling's FADE_EPOCH rule
        chunks = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
            chunks.pop()
        ds = self.dataset
        if self.workers <= 0:
            for ch in chunks:
                if self.pipeline.reference_rng_order:
                    yield self.pipeline([lambda i=i: ds.raw_sample(i) for i in ch], self.device, resample=self._resample)
                else:
                    yield self.pipeline([ds.raw_sample(i) for i in ch], self.device, resample=self._resample)
            return
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            pending = []

            def schedule(ch):
                picks = [ds.pick(i) for i in ch]                          # np.random on the main thread, in sample order
                pending.append([pool.submit(ds.read_pair, *pk) for pk in picks])
            nxt = 0
            while nxt < len(chunks) and len(pending) < self.prefetch:
                schedule(chunks[nxt])
                nxt += 1
            while pending:
                futs = pending.pop(0)
                if nxt < len(chunks):
                    schedule(chunks[nxt])
                    nxt += 1
                yield self.pipeline([f.result() for f in futs], self.device, resample=self._resample)


def build_dataloader(dataset_cfg, class_names, batch_size, dist, root_path=None, workers=4, logger=None, training=True,
                     merge_all_iters_to_one_epoch=False, total_epochs=0, drop_last=False, device=None, seed=0):
    """pcdet.datasets.build_dataloader (pcdet/datasets/__init__.py:45-91): (dataset, dataloader, sampler)."""
    if merge_all_iters_to_one_epoch:
        raise NotImplementedError('merge_all_iters_to_one_epoch is not used by the T-MAE recipes')
    if dataset_cfg.DATASET != 'ONCETemporalDataset':
        raise NotImplementedError(f'DATASET {dataset_cfg.DATASET}: only ONCETemporalDataset is built (SURVEY 8f-2)')
    dataset = ONCETemporalDataset(dataset_cfg, class_names, training=training, root_path=root_path, logger=logger)
    dataset.total_epochs = total_epochs
    rank, world = 0, 1
    if dist:
        import torch.distributed as td
        rank, world = td.get_rank(), td.get_world_size()
    sampler = EpochSampler(len(dataset), rank, world, shuffle=training, seed=seed)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    pipeline = TemporalPairPipeline(dataset_cfg, training=training, class_names=class_names, logger=logger,
                                    root_path=dataset.root_path)
    pipeline.total_epochs = total_epochs if total_epochs else pipeline.total_epochs
    loader = DeviceBatchLoader(dataset, batch_size, sampler, pipeline, device, workers=workers, drop_last=drop_last)
    return dataset, loader, sampler
