import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 't-mae_amd'), os.path.join(ROOT, 'oracle'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
CFG_YAML = os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def fl(x):
    """float() of a scalar that may still require grad (float(tensor) warns then)."""
    return float(x.detach()) if hasattr(x, 'detach') else float(x)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def load_cfg(num_stages=3):
    from pcdet.config import EasyDict, cfg_from_yaml_file
    cfg = cfg_from_yaml_file(CFG_YAML, EasyDict())
    if num_stages < 3:
        b = cfg.MODEL.BACKBONE_3D
        b.SST_BLOCK_LIST = b.SST_BLOCK_LIST[:num_stages]
        b.FEATURES_SOURCE = b.FEATURES_SOURCE[:num_stages]
    return cfg


def build_finetune_model(params=None, device='cpu', n_points=1000, batch_size=2):
    """CenterPoint (t_mae.yaml: TemporalDynVFE + SiamWCA + SSTBEVBackbone + CenterHead) through the registry path."""
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import build_network
    from tmae_amd.train import SyntheticTemporalDataset
    cfg = cfg_from_yaml_file(os.path.join(os.path.dirname(CFG_YAML), 't_mae.yaml'), EasyDict())
    ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=n_points, batch_size=batch_size, n_boxes=20)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    if params is not None:
        res = model.load_state_dict(params, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        assert all('running_' in k or 'num_batches' in k or k == 'global_step' for k in res.missing_keys), res.missing_keys
    return model.to(device), cfg, ds


def build_product_model(num_stages=3, params=None, device='cpu', n_points=1000, batch_size=2, partial=False,
                        waymo_shape=False, cfg_edit=None):
    """TMAE through the pcdet registry path; `params` = oracle-style state dict (reference key names); cfg_edit(cfg): changes
    to the YAML's config before the model is built (options the shipped YAMLs leave off)."""
    from pcdet.models import build_network
    from tmae_amd.train import SyntheticTemporalDataset
    cfg = load_cfg(num_stages)
    if cfg_edit is not None:
        cfg_edit(cfg)
    npf = 5
    if waymo_shape:      # BASELINE configs[3]: 5 point features (x,y,z,intensity,elongation), z range [-2,4), 6 m pillars
        cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [-74.88, -74.88, -2.0, 74.88, 74.88, 4.0]
        for p in cfg.DATA_CONFIG.DATA_PROCESSOR:
            if p.NAME == 'calculate_grid_size':
                p.VOXEL_SIZE = [0.32, 0.32, 6.0]
        npf = 6
    ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=n_points, batch_size=batch_size,
                                  num_point_features=npf)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    if params is not None:
        res = model.load_state_dict(params, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        assert partial or all('running_' in k or 'num_batches' in k or k == 'global_step' for k in res.missing_keys), \
            res.missing_keys
    return model.to(device), cfg, ds


@pytest.fixture(scope='session')
def oracle():
    import tmae_oracle
    return tmae_oracle


@pytest.fixture(scope='session')
def ft_oracle():
    import finetune_oracle
    return finetune_oracle


@pytest.fixture(scope='session')
def dp_oracle():
    import datapath_oracle
    return datapath_oracle


def write_once_directory(root, g):
    """A directory laid out like ONCE (ImageSets/<split>.txt, once_infos_<split>.pkl, data/<seq>/lidar_roof/<frame>.bin)
    rebuilt from the raw inputs stored in the D2 fixture.  Returns the info list."""
    import pickle
    from pathlib import Path
    root = Path(root)
    infos = []
    for i in range(int(g['n_infos'])):
        info = {'sequence_id': str(g[f'info_seq_{i}']), 'frame_id': str(g[f'info_frame_{i}']), 'pose': np.array(g[f'info_pose_{i}'])}
        if int(g[f'has_annos_{i}']):
            info['annos'] = {'name': np.array([str(n) for n in g[f'names_{i}']]), 'boxes_3d': np.array(g[f'boxes_{i}'])}
        d = root / 'data' / info['sequence_id'] / 'lidar_roof'
        d.mkdir(parents=True, exist_ok=True)
        np.asarray(g[f'scan_{i}'], np.float32).tofile(d / f"{info['frame_id']}.bin")
        infos.append(info)
    if 'n_db' in g.files:                    # gt_sampling's label database: once_dbinfos_train.pkl + gt_database/*.bin crops
        db = {}
        for k in range(int(g['n_db'])):
            rel = str(g[f'db_path_{k}'])
            (root / rel).parent.mkdir(parents=True, exist_ok=True)
            np.asarray(g[f'db_crop_{k}'], np.float32).tofile(root / rel)
            db.setdefault(str(g[f'db_name_{k}']), []).append(
                {'name': str(g[f'db_name_{k}']), 'path': rel, 'box3d_lidar': np.array(g[f'db_box_{k}']),
                 'num_points_in_gt': int(g[f'db_npts_{k}']), 'difficulty': 0})
        with open(root / 'once_dbinfos_train.pkl', 'wb') as fh:
            pickle.dump(db, fh)
    (root / 'ImageSets').mkdir(parents=True, exist_ok=True)
    seqs = list(dict.fromkeys(info['sequence_id'] for info in infos))
    for split in ('train', 'val'):
        (root / 'ImageSets' / f'{split}.txt').write_text('\n'.join(seqs) + '\n')
        with open(root / f'once_infos_{split}.pkl', 'wb') as fh:
            pickle.dump(infos, fh)
    return infos


def finetune_data_cfg(gt_sampling=False):
    """t_mae.yaml (its DATA_CONFIG carries the reference recipe's augmentor queue, gt_sampling first); gt_sampling=False:
    with the label-database pasting disabled through DISABLE_AUG_LIST (the D2 fixture's configuration)."""
    from pcdet.config import EasyDict, cfg_from_yaml_file
    cfg = cfg_from_yaml_file(os.path.join(os.path.dirname(CFG_YAML), 't_mae.yaml'), EasyDict())
    assert cfg.DATA_CONFIG.DATA_AUGMENTOR.AUG_CONFIG_LIST[0].NAME == 'gt_sampling'
    if not gt_sampling:
        cfg.DATA_CONFIG.DATA_AUGMENTOR.DISABLE_AUG_LIST = ['gt_sampling']
    return cfg
