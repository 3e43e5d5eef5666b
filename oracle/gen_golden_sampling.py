"""Pins the gt_sampling restatement of oracle/datapath_oracle.py against the reference and writes
tests/golden/D3_gt_sampling.npz.

The UNMODIFIED reference `ONCETemporalDataset` + `DataAugmentor` + `DataBaseSampler` (database_sampler.py) run on the tiny
ONCE-layout directory of gen_golden_dataset.py, extended by a label database (once_dbinfos_train.pkl + gt_database/*.bin object
crops written here), with the fine-tune DATA_CONFIG of tools/cfgs/once_models/t_mae.yaml AS IT IS (gt_sampling first in the
augmentor queue, LIMIT_WHOLE_SCENE, SAMPLE_GROUPS Car:1 Bus:4 Truck:3 Pedestrian:2 Cyclist:2).  The sampler's two COMPILED geometry
helpers are absent here (iou3d_nms_cuda.boxes_iou_bev_cpu, roiaware_pool3d_cuda.points_in_boxes_cpu: CUDA extension modules) and
are replaced by the oracle's restatements (datapath_oracle.boxes_overlap_bev / points_in_boxes_cpu): the fixture therefore pins
every line of the reference's PYTHON logic -- sampling order and pointers, LIMIT_WHOLE_SCENE, the collision rule, where the
pasted points go (_combine_two_pcs_with_delimiter / _attach_group_ids: in front of BOTH frames), the box list -- while the two
helpers stay parity unpinned (SURVEY 8c rule for absent compiled dependencies)."""
import os
import pickle
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R                          # noqa: E402
import datapath_oracle as D                     # noqa: E402
from gen_golden import save, check              # noqa: E402
from gen_golden_datapath import load_datapath_reference   # noqa: E402
from gen_golden_dataset import write_tiny_once, CLASSES, SIZES   # noqa: E402


def write_database(root, rng, per_class=7):
    db, crops = {c: [] for c in CLASSES}, {}
    (root / 'gt_database').mkdir(parents=True, exist_ok=True)
    k = 0
    for c in CLASSES:
        for _ in range(per_class):
            box = np.zeros(7)
            box[0:2] = rng.uniform(-60, 60, 2)
            box[2] = rng.normal(-1.0, 0.2)
            box[3:6] = np.array(SIZES[c]) * rng.uniform(0.9, 1.1, 3)
            box[6] = rng.uniform(-np.pi, np.pi)
            n = int(rng.integers(3, 40))                       # some below filter_by_min_points (5)
            pts = np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)) * box[3:6], rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
            rel = f'gt_database/{c}_{k}.bin'
            pts.tofile(root / rel)
            db[c].append({'name': c, 'path': rel, 'box3d_lidar': box, 'num_points_in_gt': n, 'difficulty': 0})
            crops[rel] = pts
            k += 1
    with open(root / 'once_dbinfos_train.pkl', 'wb') as fh:
        pickle.dump(db, fh)
    return db, crops


def main():
    ref = load_datapath_reference()
    DS = ref['once'].ONCETemporalDataset
    iou_mod = sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_utils']
    roi_mod = sys.modules['pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils']
    iou_mod.boxes_bev_iou_cpu = lambda a, b: D.boxes_overlap_bev(np.asarray(a), np.asarray(b)).astype(np.float32)
    roi_mod.points_in_boxes_cpu = lambda pts, boxes: torch.from_numpy(D.points_in_boxes_cpu(pts.numpy(), boxes.numpy()))
    ycfg = yaml.safe_load(open(os.path.join(R.REF, '..', 'tools', 'cfgs', 'once_models', 't_mae.yaml')))['DATA_CONFIG']
    base = yaml.safe_load(open(os.path.join(R.REF, '..', 'tools', 'cfgs', 'dataset_configs', 'once_temporal_dataset.yaml')))
    base.update({k: v for k, v in ycfg.items() if k != '_BASE_CONFIG_'})
    cfg = R.AttrDict(base)
    gs = [c for c in cfg.DATA_AUGMENTOR.AUG_CONFIG_LIST if c.NAME == 'gt_sampling'][0]
    rng = np.random.default_rng(33)
    store = {}
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp) / 'once'
        infos, raw = write_tiny_once(root, rng)
        db, crops = write_database(root, rng)
        ds = DS(dataset_cfg=cfg, class_names=CLASSES, training=True, root_path=root, logger=None)
        ds.total_epochs, ds.cur_epoch = 1, 0
        iv = [tuple(int(v) for v in x) for x in ds.once_intervals]
        pcr = np.array(cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        cfg_aug = dict(flip_axes=['x', 'y'], flip_prob=0.5, rot_prob=1.0, rot_range=[-0.78539816, 0.78539816],
                       scale_prob=1.0, scale_range=[0.95, 1.05])
        osampler = D.DataBaseSamplerOracle(
            db, dict(filter_by_min_points=list(gs.PREPARE['filter_by_min_points']), sample_groups=list(gs.SAMPLE_GROUPS),
                     limit_whole_scene=bool(gs.LIMIT_WHOLE_SCENE)), CLASSES, lambda info: crops[info['path']])
        n_pasted = 0
        samples_ref, samples_or = [], []
        for index in range(len(ds)):
            np.random.seed(900 + index)
            sample = ds[index]
            np.random.seed(900 + index)
            idx, idx_prev = D.pick_pair(iv[index], int(cfg.SCAN_WINDOW), int(cfg.get('FIXED_GAP', -1)))
            info, info_prev = infos[idx], infos[idx_prev]
            out = D.prepare_pair_sampled(raw[info['frame_id']], raw[info_prev['frame_id']], info['pose'], info_prev['pose'],
                                         info['annos']['boxes_3d'], info['annos']['name'], CLASSES, osampler,
                                         lambda: D.draw_params(cfg_aug), pcr, extra_width=tuple(gs.REMOVE_EXTRA_WIDTH),
                                         align=info['frame_id'] != info_prev['frame_id'])
            assert out is not None
            o_prev, o_cur, o_boxes = out
            check(f'sample {index} points_prev', o_prev.astype(np.float32), sample['points_prev'].astype(np.float32), 0.0)
            check(f'sample {index} points', o_cur.astype(np.float32), sample['points'].astype(np.float32), 0.0)
            check(f'sample {index} gt_boxes', o_boxes.astype(np.float64), np.asarray(sample['gt_boxes'], np.float64), 0.0)
            n_pasted += len(o_boxes)
            samples_ref.append(sample)
            samples_or.append({'points_prev': o_prev.astype(np.float32), 'points': o_cur.astype(np.float32), 'gt_boxes': o_boxes})
            store[f'gt_boxes_{index}'] = np.asarray(sample['gt_boxes'], np.float64)
            store[f'n_prev_{index}'], store[f'n_cur_{index}'] = np.int64(len(o_prev)), np.int64(len(o_cur))
        c_ref = DS.collate_batch(samples_ref)
        c_or = D.collate(samples_or)
        check('collate points', c_or['points'], c_ref['points'], 0.0)
        check('collate points_prev', c_or['points_prev'], c_ref['points_prev'], 0.0)
        check('collate gt_boxes', D.collate_boxes([s['gt_boxes'] for s in samples_or]), c_ref['gt_boxes'], 0.0)
        n_own = sum(int(sum(n in CLASSES for n in infos[i[1] - 1]['annos']['name'])) for i in iv)
        print(f'{n_pasted} boxes in the batch, {n_own} own boxes of the scenes before the range filter: the rest were pasted')
        assert n_pasted > n_own
        for i, info in enumerate(infos):
            store[f'info_seq_{i}'] = np.array(info['sequence_id'])
            store[f'info_frame_{i}'] = np.array(info['frame_id'])
            store[f'info_pose_{i}'] = np.asarray(info['pose'], np.float64)
            store[f'scan_{i}'] = raw[info['frame_id']]
            store[f'has_annos_{i}'] = np.int32('annos' in info)
            if 'annos' in info:
                store[f'names_{i}'] = np.array([str(n) for n in info['annos']['name']])
                store[f'boxes_{i}'] = np.asarray(info['annos']['boxes_3d'], np.float64)
        k = 0
        for c in CLASSES:
            for e in db[c]:
                store[f'db_name_{k}'], store[f'db_path_{k}'] = np.array(e['name']), np.array(e['path'])
                store[f'db_box_{k}'], store[f'db_npts_{k}'] = np.asarray(e['box3d_lidar'], np.float64), np.int64(e['num_points_in_gt'])
                store[f'db_crop_{k}'] = crops[e['path']]
                k += 1
        save('D3_gt_sampling', n_infos=len(infos), n_samples=len(iv), n_db=k, seed_base=np.int64(900),
             points=c_ref['points'].astype(np.float32), points_prev=c_ref['points_prev'].astype(np.float32),
             gt_boxes=c_ref['gt_boxes'].astype(np.float32), **store)
    print('gt_sampling fixture written; the Python logic of DataBaseSampler is pinned against the reference')


if __name__ == '__main__':
    main()
