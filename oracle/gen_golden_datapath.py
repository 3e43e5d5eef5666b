"""Pins oracle/datapath_oracle.py against the reference's own data-path functions (run here on CPU) and writes
tests/golden/D1_datapath.npz: once_utils.convert_prv_frame_to_cur, once_temporal_dataset.remove_ego_points and the
ONCETemporalDataset helpers (_combine_two_pcs_with_delimiter, _attach_group_ids, _split_two_pcs), DataAugmentor.
random_world_{flip,rotation,scaling}, DataProcessor.{mask_points_and_boxes_outside_range, shuffle_points},
DatasetTemplate.collate_batch -- in the order ONCETemporalDataset.__getitem__ / prepare_data call them."""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R                # noqa: E402
import gen_golden_finetune as GF      # noqa: E402
import datapath_oracle as D           # noqa: E402
from gen_golden import save, check    # noqa: E402


def load_datapath_reference():
    R.load_reference()
    GF.load_finetune_reference()
    for name, sub in [('pcdet.datasets', '/datasets'), ('pcdet.datasets.augmentor', '/datasets/augmentor'),
                      ('pcdet.datasets.processor', '/datasets/processor'), ('pcdet.datasets.once', '/datasets/once'),
                      ('pcdet.datasets.once_temporal', '/datasets/once_temporal'),
                      ('pcdet.datasets.once_temporal.once_eval', '/datasets/once_temporal/once_eval')]:
        m = types.ModuleType(name)
        m.__path__ = [R.REF + sub]
        sys.modules[name] = m
    for stub in ('cv2', 'skimage', 'skimage.transform', 'tools', 'tools.visual_utils', 'tools.visual_utils.open3d_vis_utils'):
        if stub not in sys.modules:
            try:
                importlib.import_module(stub)
            except Exception:
                sys.modules[stub] = types.ModuleType(stub)
    vis = sys.modules['tools.visual_utils.open3d_vis_utils']
    vis.draw_scenes = vis.draw_scenes_with_2pcs = lambda *a, **k: None
    return dict(once_utils=importlib.import_module('pcdet.datasets.once_temporal.once_eval.once_utils'),
                aug=importlib.import_module('pcdet.datasets.augmentor.data_augmentor'),
                proc=importlib.import_module('pcdet.datasets.processor.data_processor'),
                dataset=importlib.import_module('pcdet.datasets.dataset'),
                once=importlib.import_module('pcdet.datasets.once_temporal.once_temporal_dataset'))


def main():
    ref = load_datapath_reference()
    DS = ref['once'].ONCETemporalDataset
    AUG = ref['aug'].DataAugmentor
    PROC = ref['proc'].DataProcessor
    pcr = np.array([-74.88, -74.88, -5.0, 74.88, 74.88, 3.0], dtype=np.float32)
    cfg_aug = dict(flip_axes=['x', 'y'], flip_prob=0.5, rot_prob=1.0, rot_range=[-0.78539816, 0.78539816],
                   scale_prob=1.0, scale_range=[0.95, 1.05])
    fl = R.AttrDict(dict(NAME='random_world_flip', PROBABILITY=0.5, ALONG_AXIS_LIST=['x', 'y']))
    ro = R.AttrDict(dict(NAME='random_world_rotation', PROBABILITY=1.0, WORLD_ROT_ANGLE=[-0.78539816, 0.78539816]))
    sc = R.AttrDict(dict(NAME='random_world_scaling', PROBABILITY=1.0, WORLD_SCALE_RANGE=[0.95, 1.05]))
    proc_self = types.SimpleNamespace(point_cloud_range=pcr, training=True, mode='train')
    mcfg = R.AttrDict(dict(NAME='mask_points_and_boxes_outside_range', REMOVE_OUTSIDE_BOXES=True))
    scfg = R.AttrDict(dict(NAME='shuffle_points', SHUFFLE_ENABLED={'train': True, 'test': False}))
    rng = np.random.default_rng(12)
    samples_ref, samples_or, store = [], [], {}
    poses = [([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], [0.01, -0.02, 0.03, 0.999, 1.5, -0.4, 0.05]),    # identity-like prev
             ([0.02, 0.01, -0.3, 0.95, 12.0, -3.0, 0.2], [0.0] * 7),                                 # static current pose: skipped
             ([0.0] * 7, [0.0] * 7)]                                                                  # both static
    for si, (pose_prv, pose_cur) in enumerate(poses):
        n1, n0 = 5000 + 300 * si, 4700 + 200 * si
        mk = lambda n: np.concatenate([rng.uniform(-90, 90, (n, 2)), rng.normal(-1.5, 1.0, (n, 1)), rng.uniform(0, 1, (n, 1))],
                                      axis=1).astype(np.float32)
        pts, prv = mk(n1), mk(n0)
        pts[:50, :2] = rng.uniform(-2.5, 2.5, (50, 2))                         # ego-vehicle returns
        prv[:50, :2] = rng.uniform(-2.5, 2.5, (50, 2))
        pts[50, :2] = [74.88, -74.88]                                          # exactly on the crop boundary (before aug)
        pose_prv, pose_cur = np.array(pose_prv), np.array(pose_cur)
        # ---- the reference, in __getitem__ / prepare_data order (once_temporal_dataset.py:139-212, 246-330)
        np.random.seed(100 + si)
        p1 = ref['once'].remove_ego_points(pts.copy(), 2)
        p0 = ref['once'].remove_ego_points(prv.copy(), 2)
        p0 = ref['once_utils'].convert_prv_frame_to_cur(p0, pose_prv, pose_cur)
        dd = {'points': DS._combine_two_pcs_with_delimiter(None, p0, p1, delimiter=-np.inf),
              'transformation_3d_list': [], 'transformation_3d_params': {}}
        dd = AUG.random_world_flip(None, dd, config=fl)
        dd = AUG.random_world_rotation(None, dd, config=ro)
        dd = AUG.random_world_scaling(None, dd, config=sc)
        dd['points'] = DS._attach_group_ids(None, dd['points'])
        dd = PROC.mask_points_and_boxes_outside_range(proc_self, dd, config=mcfg)
        n_kept = dd['points'].shape[0]
        state = np.random.get_state()
        dd = PROC.shuffle_points(proc_self, dd, config=scfg)
        r_prev, r_cur = DS._split_two_pcs(None, dd['points'])
        # ---- the oracle with the same draws
        np.random.seed(100 + si)
        params = D.draw_params(cfg_aug)
        np.random.set_state(state)
        perm = np.random.permutation(n_kept)
        o_prev, o_cur = D.prepare_pair(pts, prv, pose_cur, pose_prv, params, perm, pcr)
        check(f'sample {si} prev', o_prev, r_prev.astype(np.float32), 0.0)
        check(f'sample {si} cur', o_cur, r_cur.astype(np.float32), 0.0)
        assert params['flips'] == dd['transformation_3d_params']['random_world_flip']
        samples_ref.append({'points_prev': r_prev.astype(np.float32), 'points': r_cur.astype(np.float32)})
        samples_or.append({'points_prev': o_prev, 'points': o_cur})
        store.update({f'pts_{si}': pts, f'prv_{si}': prv, f'pose_prv_{si}': pose_prv, f'pose_cur_{si}': pose_cur,
                      f'flip_x_{si}': np.int32('x' in params['flips']), f'flip_y_{si}': np.int32('y' in params['flips']),
                      f'rot_{si}': np.float64(params['rot']), f'scale_{si}': np.float64(params['scale']), f'perm_{si}': perm})
    c_ref = ref['dataset'].DatasetTemplate.collate_batch(samples_ref)
    c_or = D.collate(samples_or)
    check('collate points', c_or['points'], c_ref['points'], 0.0)
    check('collate points_prev', c_or['points_prev'], c_ref['points_prev'], 0.0)
    save('D1_datapath', n_samples=len(poses), points=c_ref['points'].astype(np.float32),
         points_prev=c_ref['points_prev'].astype(np.float32), **store)
    print('data-path fixture written; oracle pinned against the reference functions')


if __name__ == '__main__':
    main()
