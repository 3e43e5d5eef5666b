"""Build-container-only loader for the reference's Python hot path.

TEST INFRASTRUCTURE.  Imports the *unmodified* reference modules from /root/reference by
path (no package __init__ runs) with CPU stand-ins -- built from this repo's own oracle
restatements -- for the native / third-party dependencies that are absent here:
``sst_ops_cuda``, ``torch_scatter``, ``spconv``, ``pytorch3d``, ``SharedArray``.
Used only by ``oracle/gen_golden.py`` to produce ``tests/golden/*.npz``; /root/reference
does not exist on the GPU box, so nothing at test/bench time imports this file.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tmae_oracle as O  # noqa: E402

REF = '/root/reference/pcdet'


class AttrDict(dict):
    """Tiny easydict replacement (easydict is not installed here)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(x) for x in v]
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


# ----------------------------------------------------------------------------- stand-ins

def _make_sst_ops():
    m = types.ModuleType('pcdet.ops.sst_ops.sst_ops_cuda')

    def ingroup_inds_wrapper(group_inds, out_inds):
        out_inds.copy_(torch.from_numpy(O.stable_ingroup_rank(group_inds.numpy())))
        return 1

    def group_inner_inds_wrapper(inverse_inds, group_inds):
        M, K = group_inds.shape
        group_inds.copy_(torch.from_numpy(O.group_inner_inds(inverse_inds.numpy(), M, K)))
        return 1

    m.ingroup_inds_wrapper = ingroup_inds_wrapper
    m.group_inner_inds_wrapper = group_inner_inds_wrapper
    return m


def _make_torch_scatter():
    m = types.ModuleType('torch_scatter')

    def scatter(src, index, dim=0, reduce='mean'):
        assert dim == 0 and reduce == 'mean'
        return O.segment_mean(src, index, int(index.max()) + 1)

    def scatter_max(src, index, dim=0):
        assert dim == 0
        return O.segment_max(src, index, int(index.max()) + 1), None

    m.scatter, m.scatter_max = scatter, scatter_max
    return m


def _make_spconv():
    sp = types.ModuleType('spconv')
    spt = types.ModuleType('spconv.pytorch')
    conv_mod = types.ModuleType('spconv.pytorch.conv')

    class SparseConvTensor:
        def __init__(self, features, indices, spatial_shape, batch_size):
            self.features, self.indices = features, indices
            self.spatial_shape, self.batch_size = list(spatial_shape), batch_size

        def replace_feature(self, f):
            return SparseConvTensor(f, self.indices, self.spatial_shape, self.batch_size)

        def dense(self):
            return O.to_dense(self.features, self.indices.numpy().astype(np.int64),
                              tuple(int(s) for s in self.spatial_shape), self.batch_size)

    class SparseModule(nn.Module):
        pass

    class SparseConvolution(SparseModule):
        def __init__(self, cin, cout, k, stride=1, padding=0, bias=False, indice_key=None, subm=False):
            super().__init__()
            assert k == 3 and not bias
            self.subm, self.stride = subm, stride
            self.weight = nn.Parameter(torch.randn(cout, 3, 3, cin) * (2.0 / (9 * cin)) ** 0.5)

        def forward(self, x):
            kind = 'subm' if self.subm else 'down'
            out_ind, out_shape, pairs = O.sparse_rulebook(x.indices.numpy().astype(np.int64),
                                                          tuple(int(s) for s in x.spatial_shape), kind)
            y = O.sparse_conv(x.features, self.weight, pairs, out_ind.shape[0])
            return SparseConvTensor(y, torch.from_numpy(out_ind).int(), out_shape, x.batch_size)

    class SubMConv2d(SparseConvolution):
        def __init__(self, cin, cout, k, bias=False, indice_key=None, **kw):
            super().__init__(cin, cout, k, bias=bias, indice_key=indice_key, subm=True)

    class SparseConv2d(SparseConvolution):
        def __init__(self, cin, cout, k, stride=1, padding=0, bias=False, indice_key=None, **kw):
            assert stride == 2 and padding == 1
            super().__init__(cin, cout, k, stride=stride, padding=padding, bias=bias, indice_key=indice_key)

    class SparseSequential(SparseModule):
        def __init__(self, *mods):
            super().__init__()
            for i, m in enumerate(mods):
                self.add_module(str(i), m)

        def forward(self, x):
            for m in self._modules.values():
                if isinstance(m, SparseModule):
                    x = m(x)
                else:
                    x = x.replace_feature(m(x.features))
            return x

    for mod in (sp, spt):
        mod.SparseConvTensor = SparseConvTensor
        mod.SparseModule = SparseModule
        mod.SparseSequential = SparseSequential
        mod.SubMConv2d, mod.SparseConv2d = SubMConv2d, SparseConv2d
        mod.conv = conv_mod
    conv_mod.SparseConvolution = SparseConvolution
    sp.pytorch = spt
    return sp, spt, conv_mod


def _make_pytorch3d():
    p3 = types.ModuleType('pytorch3d')
    loss = types.ModuleType('pytorch3d.loss')

    def chamfer_distance(x, y, weights=None):
        return O.chamfer_distance(x, y, weights), None

    loss.chamfer_distance = chamfer_distance
    p3.loss = loss
    return p3, loss


_loaded = {}


def load_reference():
    """Returns dict(vfe=module, mae=module, sst_utils=..., spt=..., siam=..., cosine_msa=..., common_utils=...)."""
    if _loaded:
        return _loaded
    for name, sub in [('pcdet', ''), ('pcdet.models', '/models'),
                      ('pcdet.models.backbones_3d', '/models/backbones_3d'),
                      ('pcdet.models.backbones_3d.vfe', '/models/backbones_3d/vfe'),
                      ('pcdet.models.model_utils', '/models/model_utils'),
                      ('pcdet.utils', '/utils'), ('pcdet.ops', '/ops'), ('pcdet.ops.sst_ops', '/ops/sst_ops')]:
        m = types.ModuleType(name)
        m.__path__ = [REF + sub]
        sys.modules[name] = m
    sys.modules['pcdet.ops.sst_ops.sst_ops_cuda'] = _make_sst_ops()
    sys.modules['torch_scatter'] = _make_torch_scatter()
    sys.modules['SharedArray'] = types.ModuleType('SharedArray')
    sp, spt, conv_mod = _make_spconv()
    sys.modules['spconv'], sys.modules['spconv.pytorch'], sys.modules['spconv.pytorch.conv'] = sp, spt, conv_mod
    p3, loss = _make_pytorch3d()
    sys.modules['pytorch3d'], sys.modules['pytorch3d.loss'] = p3, loss
    _loaded.update(
        vfe=importlib.import_module('pcdet.models.backbones_3d.vfe.temporal_dyn_vfe'),
        mae=importlib.import_module('pcdet.models.backbones_3d.SiamWCA_MAE'),
        spt=importlib.import_module('pcdet.models.backbones_3d.spt_backbone'),
        siam=importlib.import_module('pcdet.models.backbones_3d.SiamWCA'),
        sst_utils=importlib.import_module('pcdet.models.model_utils.sst_utils'),
        cosine_msa=importlib.import_module('pcdet.models.model_utils.cosine_msa'),
        common_utils=importlib.import_module('pcdet.utils.common_utils'),
    )
    return _loaded


def reference_cfg(num_stages=3):
    with open('/root/reference/tools/cfgs/once_models/t_mae_ssl.yaml') as f:
        cfg = AttrDict(yaml.safe_load(f))
    if num_stages < 3:
        b = cfg.MODEL.BACKBONE_3D
        b.SST_BLOCK_LIST = b.SST_BLOCK_LIST[:num_stages]
        b.FEATURES_SOURCE = b.FEATURES_SOURCE[:num_stages]
    return cfg


def build_reference_model(num_stages=3, seed=0):
    """Instantiate the reference TemporalDynVFE + SiamWCA_MAE on CPU."""
    ref = load_reference()
    cfg = reference_cfg(num_stages)
    torch.manual_seed(seed)
    pcr = np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, dtype=np.float32)
    vs = [0.32, 0.32, 8.0]
    grid = np.array([468, 468, 1])
    V = ref['vfe'].TemporalDynVFE(cfg.MODEL.VFE, num_point_features=5, voxel_size=vs,
                                  point_cloud_range=pcr, grid_size=grid)
    B = ref['mae'].SiamWCA_MAE(cfg.MODEL.BACKBONE_3D, input_channels=V.get_output_feature_dim(),
                               grid_size=grid, voxel_size=vs, point_cloud_range=pcr)
    return V, B, cfg
