"""CPU: the data-path oracle (oracle/datapath_oracle.py) against the batch captured from the reference's dataset
functions (tests/golden/D1_datapath.npz, written by oracle/gen_golden_datapath.py)."""
import numpy as np

from conftest import golden

PCR = [-74.88, -74.88, -5.0, 74.88, 74.88, 3.0]


def test_d1_datapath(dp_oracle):
    g = golden('D1_datapath')
    samples = []
    for i in range(int(g['n_samples'])):
        params = dict(flips=(['x'] if int(g[f'flip_x_{i}']) else []) + (['y'] if int(g[f'flip_y_{i}']) else []),
                      rot=float(g[f'rot_{i}']), scale=float(g[f'scale_{i}']))
        prv, cur = dp_oracle.prepare_pair(g[f'pts_{i}'], g[f'prv_{i}'], g[f'pose_cur_{i}'], g[f'pose_prv_{i}'], params,
                                          g[f'perm_{i}'], PCR)
        samples.append({'points_prev': prv, 'points': cur})
    c = dp_oracle.collate(samples)
    assert np.array_equal(c['points'], g['points']) and np.array_equal(c['points_prev'], g['points_prev'])
    assert c['batch_size'] == 3


def test_pose_quirks_and_draw_order(dp_oracle):
    # an all-zero pose means "static": that step is skipped (once_utils.py:9-10,17-18)
    p = np.array([[1.0, 2.0, 3.0, 0.5]], np.float32)
    same = dp_oracle.convert_prv_frame_to_cur(p, np.zeros(7), np.zeros(7))
    assert np.array_equal(same, p)
    moved = dp_oracle.convert_prv_frame_to_cur(p, np.array([0, 0, 0, 1.0, 1.0, 0, 0]), np.zeros(7))
    assert np.allclose(moved[0, :3], [2.0, 2.0, 3.0]) and moved[0, 3] == 0.5
    back = dp_oracle.convert_prv_frame_to_cur(p, np.zeros(7), np.array([0, 0, 0, 1.0, 1.0, 0, 0]))
    assert np.allclose(back[0, :3], [0.0, 2.0, 3.0])
    cfg = dict(flip_axes=['x', 'y'], flip_prob=0.5, rot_prob=1.0, rot_range=[-0.78539816, 0.78539816], scale_prob=1.0,
               scale_range=[0.95, 1.05])
    np.random.seed(3)
    a = dp_oracle.draw_params(cfg)
    np.random.seed(3)
    b = dp_oracle.draw_params(cfg)
    assert a == b and -0.78539816 <= a['rot'] <= 0.78539816 and 0.95 <= a['scale'] <= 1.05


def _d2_params(g, i):
    return dict(flips=(['x'] if int(g[f'flip_x_{i}']) else []) + (['y'] if int(g[f'flip_y_{i}']) else []),
                rot=float(g[f'rot_{i}']), scale=float(g[f'scale_{i}']))


def test_d2_oracle_labels_and_index_logic(dp_oracle):
    """The oracle against what the reference's own ONCETemporalDataset returned on the tiny ONCE-layout directory (D2):
    interval list, (idx, idx_prev) picks, points of both frames, gt_boxes through the joint augmentation / class filter
    / outside-range filter, the collated batch -- bit for bit from the raw scans, poses and annotations."""
    import tempfile
    from conftest import write_once_directory
    g = golden('D2_once_dataset')
    with tempfile.TemporaryDirectory() as tmp:
        infos = write_once_directory(tmp, g)
    iv = dp_oracle.build_intervals(infos, 3, 'train')
    assert iv == [tuple(int(v) for v in r) for r in g['intervals']]
    classes = ['Car', 'Bus', 'Truck', 'Pedestrian', 'Cyclist']
    pcr = np.array(PCR, dtype=np.float32)
    samples = []
    for i in range(int(g['n_samples'])):
        np.random.seed(500 + i)
        idx, idx_prev = dp_oracle.pick_pair(iv[i], 3, -1)
        assert (idx, idx_prev) == tuple(int(v) for v in g['picks'][i])
        par = _d2_params(g, i)
        prv, cur = dp_oracle.prepare_pair(np.array(g[f'scan_{idx}']), np.array(g[f'scan_{idx_prev}']), infos[idx]['pose'],
                                          infos[idx_prev]['pose'], par, g[f'perm_{i}'], pcr, align=idx != idx_prev)
        boxes = dp_oracle.prepare_labels(infos[idx]['annos']['boxes_3d'], infos[idx]['annos']['name'], classes, par, pcr)
        assert np.array_equal(boxes, g[f'gt_boxes_{i}'])
        samples.append({'points_prev': prv, 'points': cur, 'gt_boxes': boxes})
    c = dp_oracle.collate(samples)
    assert np.array_equal(c['points'], g['points']) and np.array_equal(c['points_prev'], g['points_prev'])
    assert np.array_equal(dp_oracle.collate_boxes([s['gt_boxes'] for s in samples]), g['gt_boxes'])


def test_d2_product_labels_and_index_logic_host_side():
    """The product's host-side label preparation and sample-index logic (tmae_amd.data: ONCETemporalDataset,
    TemporalPairPipeline.prepare_labels, EpochSampler) against the same fixture -- no GPU involved: boxes are a few dozen
    rows per sample and stay on the host, as do the random draws."""
    import tempfile
    import torch
    from conftest import write_once_directory, finetune_data_cfg
    from tmae_amd.data import ONCETemporalDataset, TemporalPairPipeline, EpochSampler
    g = golden('D2_once_dataset')
    cfg = finetune_data_cfg()
    with tempfile.TemporaryDirectory() as tmp:
        write_once_directory(tmp, g)
        ds = ONCETemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True, root_path=tmp)
        assert [tuple(iv) for iv in ds.once_intervals] == [tuple(int(v) for v in r) for r in g['intervals']]
        assert ds.grid_size.tolist() == [468, 468, 1] and ds.point_feature_encoder.num_point_features == 5
        pipe = TemporalPairPipeline(cfg.DATA_CONFIG, training=True, class_names=cfg.CLASS_NAMES)
        for i in range(len(ds)):
            np.random.seed(500 + i)
            pk = ds.pick(i)
            assert pk == tuple(int(v) for v in g['picks'][i])
            par = pipe.draw()                                          # same np.random stream as the reference's sample
            assert par == _d2_params(g, i)
            s = ds.read_pair(*pk)
            assert ('pose' in s) == (pk[0] != pk[1])
            boxes = pipe.prepare_labels(s['gt_boxes'], s['gt_names'], par)
            assert np.array_equal(boxes, g[f'gt_boxes_{i}']), i
        # evaluation mode: no augmentation, classes filtered, outside boxes kept (data_processor.py:85)
        ds_t = ONCETemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=False, root_path=tmp)
        pipe_t = TemporalPairPipeline(cfg.DATA_CONFIG, training=False, class_names=cfg.CLASS_NAMES)
        np.random.seed(77)
        s = ds_t.raw_sample(2)
        assert np.array_equal(pipe_t.prepare_labels(s['gt_boxes'], s['gt_names'], pipe_t.draw()), g['test_gt_boxes'])
        # a sample whose boxes all fall outside the classes: training -> None (the loader then draws another index)
        assert pipe.prepare_labels(np.array([[1., 2., 0., 2., 1., 1., 0.3]]), np.array(['Tricycle']), _d2_params(g, 0)) is None
    # sampler = torch's DistributedSampler rule
    from torch.utils.data import DistributedSampler
    for world in (1, 3):
        for rank in range(world):
            ref = DistributedSampler(range(10), num_replicas=world, rank=rank, shuffle=True, seed=0)
            ref.set_epoch(4)
            mine = EpochSampler(10, rank, world, shuffle=True, seed=0)
            mine.set_epoch(4)
            assert list(mine) == list(ref) and len(mine) == len(ref)
    assert list(EpochSampler(7, 1, 2, shuffle=False)) == [1, 3, 5, 0]


def test_d3_gt_sampling_host_logic():
    """gt_sampling (label-database pasting, database_sampler.py) of the product -- sampling order / pointers / permutations,
    LIMIT_WHOLE_SCENE, the collision rule, the box list after the world augmentation -- against what the reference's own
    DataBaseSampler + ONCETemporalDataset returned (fixture D3; the two compiled geometry helpers of the reference were
    stand-ins there: parity unpinned for those).  Host side only: the pasted / removed POINTS are the GPU test's."""
    import tempfile
    from conftest import write_once_directory, finetune_data_cfg
    from tmae_amd.data import ONCETemporalDataset, TemporalPairPipeline
    g = golden('D3_gt_sampling')
    cfg = finetune_data_cfg(gt_sampling=True)
    with tempfile.TemporaryDirectory() as tmp:
        write_once_directory(tmp, g)
        ds = ONCETemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True, root_path=tmp)
        pipe = TemporalPairPipeline(cfg.DATA_CONFIG, training=True, class_names=cfg.CLASS_NAMES, root_path=tmp)
        assert pipe.sampler is not None and pipe.aug_order == ['random_world_flip', 'random_world_rotation', 'random_world_scaling']
        pipe.total_epochs = 1
        n_own = n_all = 0
        for i in range(len(ds)):
            np.random.seed(int(g['seed_base']) + i)
            s = ds.raw_sample(i)
            sboxes, infos = pipe.sampler.sample(s['gt_boxes'], s['gt_names'], 0, 1)
            known = np.array([str(n) in cfg.CLASS_NAMES for n in s['gt_names']])
            boxes = np.concatenate([s['gt_boxes'][known][:, :7], sboxes], 0) if infos else s['gt_boxes']
            names = np.concatenate([s['gt_names'][known], np.array([x['name'] for x in infos])], 0) if infos else s['gt_names']
            got = pipe.prepare_labels(boxes, names, pipe.draw())
            assert np.array_equal(got, g[f'gt_boxes_{i}']), i
            n_own += int(known.sum())
            n_all += len(got)
        assert n_all > n_own                                       # boxes were really pasted
        # the fade-out rule (database_sampler.py:218-219): no pasting once total_epochs < FADE_EPOCH + cur_epoch + 1
        assert pipe.sampler.sample(s['gt_boxes'], s['gt_names'], cur_epoch=1, total_epochs=1)[1] == []
    # the overlap test behind the collision rule: touching is not overlapping, rotation counts
    from tmae_amd.data.database_sampler import bev_rectangles_overlap
    a = np.array([[0, 0, 0, 4, 2, 1, 0.0]])
    assert bev_rectangles_overlap(a, np.array([[3.9, 0, 0, 4, 2, 1, 0.0]]))[0, 0]
    assert not bev_rectangles_overlap(a, np.array([[4.0, 0, 0, 4, 2, 1, 0.0]]))[0, 0]
    assert not bev_rectangles_overlap(a, np.array([[3.3, 0.0, 0, 2, 2, 1, 0.0]]))[0, 0]            # x in [2.3, 4.3]
    assert bev_rectangles_overlap(a, np.array([[3.3, 0.0, 0, 2, 2, 1, np.pi / 4]]))[0, 0]          # its corner reaches x = 1.886
    assert not bev_rectangles_overlap(a, np.array([[3.2, 1.8, 0, 2, 2, 1, np.pi / 4]]))[0, 0]      # diamond edge x + y = 3.586 > 3
