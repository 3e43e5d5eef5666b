"""Pins the label / sample-index side of oracle/datapath_oracle.py against the reference's own dataset class and
writes tests/golden/D2_once_dataset.npz.

The UNMODIFIED reference `ONCETemporalDataset` (pcdet/datasets/once_temporal/once_temporal_dataset.py) is
instantiated on a tiny directory laid out like ONCE (ImageSets/<split>.txt, once_infos_<split>.pkl,
data/<seq>/lidar_roof/<frame>.bin written here from seeded random scans) with the fine-tune DATA_CONFIG of
tools/cfgs/once_models/t_mae.yaml (gt_sampling disabled through DISABLE_AUG_LIST: label-database pasting is not
built) and its `__getitem__` / `collate_batch` run with seeded `np.random`.  Captured: the interval list, the
(idx, idx_prev) picks, every sample's points / points_prev / gt_boxes and the collated batch; the oracle
(generate/build_intervals, pick_pair, prepare_pair, prepare_labels, collate, collate_boxes) must reproduce all of it bit
for bit from the raw scans, poses, annotations and the same random draws.  The fixture stores the RAW inputs (scans,
poses, boxes, names) and the expected outputs: data only."""
import os
import pickle
import sys
import tempfile
from pathlib import Path

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R                          # noqa: E402
import datapath_oracle as D                     # noqa: E402
from gen_golden import save, check              # noqa: E402
from gen_golden_datapath import load_datapath_reference   # noqa: E402

CLASSES = ['Car', 'Bus', 'Truck', 'Pedestrian', 'Cyclist']
SIZES = {'Car': [4.4, 1.9, 1.6], 'Bus': [11.0, 2.9, 3.4], 'Truck': [7.5, 2.6, 3.0], 'Pedestrian': [0.8, 0.8, 1.75],
         'Cyclist': [2.0, 0.8, 1.6], 'Tricycle': [2.5, 1.2, 1.7]}


def write_tiny_once(root, rng):
    """3 sequences x (4, 3, 2) frames; annotations on some frames only (as in ONCE, where every 2nd-5th frame is
    labelled), one static (all-zero) pose, one class outside CLASS_NAMES, one box far outside the range."""
    infos, raw = [], {}
    seqs = [('000076', 4), ('000080', 3), ('000092', 2)]
    (root / 'ImageSets').mkdir(parents=True)
    for split in ('train', 'val'):
        (root / 'ImageSets' / f'{split}.txt').write_text('\n'.join(s for s, _ in seqs) + '\n')
    t = 1616100800000
    for seq, nf in seqs:
        d = root / 'data' / seq / 'lidar_roof'
        d.mkdir(parents=True)
        for f in range(nf):
            fid = str(t)
            t += 500
            n = int(rng.integers(1800, 2600))
            pts = np.concatenate([rng.uniform(-85, 85, (n, 2)), rng.normal(-1.5, 1.0, (n, 1)), rng.uniform(0, 1, (n, 1))],
                                 axis=1).astype(np.float32)
            pts[:40, :2] = rng.uniform(-2.5, 2.5, (40, 2))
            pts.tofile(d / f'{fid}.bin')
            q = rng.normal(0, 0.02, 4)
            q[3] = 1.0
            pose = np.concatenate([q / np.linalg.norm(q), [0.8 * f + rng.normal(0, 0.05), rng.normal(0, 0.05), 0.0]])
            if seq == '000080' and f == 1:
                pose = np.zeros(7)                                     # "static" pose: the alignment step is skipped
            info = {'sequence_id': seq, 'frame_id': fid, 'timestamp': int(fid), 'pose': pose}
            if not (seq == '000076' and f == 1):                       # an unlabelled frame: its interval is filtered
                k = int(rng.integers(4, 9))
                names = rng.choice(CLASSES + ['Tricycle'], k)
                boxes = np.zeros((k, 7))
                boxes[:, 0:2] = rng.uniform(-70, 70, (k, 2))
                boxes[:, 2] = rng.normal(-1.0, 0.3, k)
                boxes[:, 3:6] = np.array([SIZES[n_] for n_ in names]) * rng.uniform(0.9, 1.1, (k, 3))
                boxes[:, 6] = rng.uniform(-np.pi, np.pi, k)
                boxes[0, 0:2] = [120.0, -130.0]                       # no corner inside the range after any augmentation
                info['annos'] = {'name': names, 'boxes_3d': boxes}
            infos.append(info)
            raw[fid] = pts
    for split in ('train', 'val'):
        with open(root / f'once_infos_{split}.pkl', 'wb') as fh:
            pickle.dump(infos, fh)
    return infos, raw


def main():
    ref = load_datapath_reference()
    DS = ref['once'].ONCETemporalDataset
    ycfg = yaml.safe_load(open(os.path.join(R.REF, '..', 'tools', 'cfgs', 'once_models', 't_mae.yaml')))['DATA_CONFIG']
    base = yaml.safe_load(open(os.path.join(R.REF, '..', 'tools', 'cfgs', 'dataset_configs', 'once_temporal_dataset.yaml')))
    base.update({k: v for k, v in ycfg.items() if k != '_BASE_CONFIG_'})
    base['DATA_AUGMENTOR']['DISABLE_AUG_LIST'] = ['gt_sampling']
    cfg = R.AttrDict(base)
    rng = np.random.default_rng(21)
    store = {}
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp) / 'once'
        infos, raw = write_tiny_once(root, rng)
        ds = DS(dataset_cfg=cfg, class_names=CLASSES, training=True, root_path=root, logger=None)
        # ---- sample index logic
        want_iv = [tuple(int(v) for v in iv) for iv in ds.once_intervals]
        got_iv = D.build_intervals(infos, int(cfg.SCAN_WINDOW), 'train')
        assert got_iv == want_iv, (got_iv, want_iv)
        assert D.generate_intervals(0, 6, 3) == [(0, 1), (0, 2), (0, 3), (1, 4), (2, 5), (3, 6)]       # dataset.py:241-245
        assert D.generate_intervals(0, 6, 2) == [(0, 1), (0, 2), (1, 3), (2, 4), (3, 5), (4, 6)]
        pcr = np.array(cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        cfg_aug = dict(flip_axes=['x', 'y'], flip_prob=0.5, rot_prob=1.0, rot_range=[-0.78539816, 0.78539816],
                       scale_prob=1.0, scale_range=[0.95, 1.05])
        samples_ref, samples_or, picks = [], [], []
        for index in range(len(ds)):
            np.random.seed(500 + index)
            sample = ds[index]                                                   # the reference, end to end
            # ---- the oracle on the raw inputs, with the same draws in the same order
            np.random.seed(500 + index)
            idx, idx_prev = D.pick_pair(want_iv[index], int(cfg.SCAN_WINDOW), int(cfg.get('FIXED_GAP', -1)))
            info, info_prev = infos[idx], infos[idx_prev]
            params = D.draw_params(cfg_aug)
            pts, prv = raw[info['frame_id']], raw[info_prev['frame_id']]
            align = info['frame_id'] != info_prev['frame_id']
            # number of points that survive ego removal + crop = length of the shuffle permutation
            o_prev, o_cur = D.prepare_pair(pts, prv, info['pose'], info_prev['pose'], params, None, pcr, align=align)
            perm = np.random.permutation(len(o_prev) + len(o_cur))
            o_prev, o_cur = D.prepare_pair(pts, prv, info['pose'], info_prev['pose'], params, perm, pcr, align=align)
            boxes = D.prepare_labels(info['annos']['boxes_3d'], info['annos']['name'], CLASSES, params, pcr)
            assert boxes is not None, 'fixture samples keep at least one box (the resample rule is tested separately)'
            assert str(sample['frame_id']) == info['frame_id']
            check(f'sample {index} points_prev', o_prev, sample['points_prev'].astype(np.float32), 0.0)
            check(f'sample {index} points', o_cur, sample['points'].astype(np.float32), 0.0)
            check(f'sample {index} gt_boxes', boxes.astype(np.float64), np.asarray(sample['gt_boxes'], np.float64), 0.0)
            samples_ref.append(sample)
            samples_or.append({'points_prev': o_prev, 'points': o_cur, 'gt_boxes': boxes})
            picks.append((idx, idx_prev))
            store.update({f'flip_x_{index}': np.int32('x' in params['flips']), f'flip_y_{index}': np.int32('y' in params['flips']),
                          f'rot_{index}': np.float64(params['rot']), f'scale_{index}': np.float64(params['scale']),
                          f'perm_{index}': perm, f'gt_boxes_{index}': np.asarray(sample['gt_boxes'], np.float64)})
        c_ref = DS.collate_batch(samples_ref)
        c_or = D.collate(samples_or)
        check('collate points', c_or['points'], c_ref['points'], 0.0)
        check('collate points_prev', c_or['points_prev'], c_ref['points_prev'], 0.0)
        check('collate gt_boxes', D.collate_boxes([s['gt_boxes'] for s in samples_or]), c_ref['gt_boxes'], 0.0)
        # ---- test mode (no augmentation, no shuffle, no outside-box removal): the first two samples
        ds_t = DS(dataset_cfg=cfg, class_names=CLASSES, training=False, root_path=root, logger=None)     # split 'val' = same infos
        assert [tuple(int(v) for v in iv) for iv in ds_t.once_intervals] == want_iv
        np.random.seed(77)
        st = ds_t[2]
        np.random.seed(77)
        idx, idx_prev = D.pick_pair(want_iv[2], int(cfg.SCAN_WINDOW), -1)
        t_prev, t_cur = D.prepare_pair(raw[infos[idx]['frame_id']], raw[infos[idx_prev]['frame_id']], infos[idx]['pose'],
                                       infos[idx_prev]['pose'], dict(flips=[], rot=0.0, scale=1.0), None, pcr,
                                       align=idx != idx_prev, augment_points=False)
        t_boxes = D.prepare_labels(infos[idx]['annos']['boxes_3d'], infos[idx]['annos']['name'], CLASSES, None, pcr, training=False)
        check('test-mode points', t_cur.astype(np.float32), st['points'].astype(np.float32), 0.0)
        check('test-mode points_prev', t_prev.astype(np.float32), st['points_prev'].astype(np.float32), 0.0)
        check('test-mode gt_boxes', t_boxes.astype(np.float64), np.asarray(st['gt_boxes'], np.float64), 0.0)
        # raw inputs of the fixture
        for i, info in enumerate(infos):
            store[f'info_seq_{i}'] = np.array(info['sequence_id'])
            store[f'info_frame_{i}'] = np.array(info['frame_id'])
            store[f'info_pose_{i}'] = np.asarray(info['pose'], np.float64)
            store[f'scan_{i}'] = raw[info['frame_id']]
            store[f'has_annos_{i}'] = np.int32('annos' in info)
            if 'annos' in info:
                store[f'names_{i}'] = np.array([str(n) for n in info['annos']['name']])
                store[f'boxes_{i}'] = np.asarray(info['annos']['boxes_3d'], np.float64)
        save('D2_once_dataset', n_infos=len(infos), n_samples=len(want_iv), intervals=np.array(want_iv, np.int64),
             picks=np.array(picks, np.int64), points=c_ref['points'].astype(np.float32),
             points_prev=c_ref['points_prev'].astype(np.float32), gt_boxes=c_ref['gt_boxes'].astype(np.float32),
             test_points=st['points'].astype(np.float32), test_points_prev=st['points_prev'].astype(np.float32),
             test_gt_boxes=np.asarray(st['gt_boxes'], np.float64), **store)
    print('dataset fixture written; label and index logic pinned against the reference ONCETemporalDataset')


if __name__ == '__main__':
    main()
