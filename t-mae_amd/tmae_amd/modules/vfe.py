"""TemporalDynVFE: dynamic pillar voxelisation + point MLP + max pooling for the two frames.

Host-side mirror of pcdet/models/backbones_3d/vfe/temporal_dyn_vfe.py (same constructor kwargs, same
batch_dict keys, same parameter names ``dvfe_mlps.0.{0,1,3,4}``); the irregular work runs in HIP kernels
(ops.voxelize / segment_csr / vfe_point_features / scatter_max).  Both frames are voxelised before the
single host sync that fetches their point / voxel counts.
"""
import torch
import torch.nn as nn

from .. import ops


def make_fc_layers(fc_cfg, input_channels):
    """Linear(no bias) + BatchNorm1d + ReLU per entry (pcdet/models/model_utils/network_utils.py:25-40)."""
    layers, c_in = [], input_channels
    for c in fc_cfg:
        layers += [nn.Linear(c_in, c, bias=False), nn.BatchNorm1d(c), nn.ReLU(inplace=True)]
        c_in = c
    return nn.Sequential(*layers)


class VFETemplate(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg

    def get_output_feature_dim(self):
        raise NotImplementedError


class TemporalDynVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, grid_size, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.sample_type = model_cfg.get('TYPE', 'mean')
        if self.sample_type != 'mean':
            raise NotImplementedError('TemporalDynVFE: only TYPE mean is on the T-MAE path')
        num_point_features -= 1                       # group_id column (temporal_dyn_vfe.py:16)
        mlps = model_cfg.get('MLPS', None)
        if mlps is None or len(mlps) != 1:
            raise NotImplementedError('TemporalDynVFE: exactly one MLPS entry (t_mae_ssl.yaml:54)')
        if model_cfg.WITH_DISTANCE or not model_cfg.USE_ABSLOTE_XYZ or not model_cfg.USE_CLUSTER_XYZ:
            raise NotImplementedError('TemporalDynVFE: USE_ABSLOTE_XYZ + USE_CLUSTER_XYZ, no distance')
        if model_cfg.get('AGGREGATION_MLPS', None) is not None:
            raise NotImplementedError('TemporalDynVFE: AGGREGATION_MLPS is unused by the T-MAE configs')
        input_channels = num_point_features + 6
        self.dvfe_mlps = nn.ModuleList([make_fc_layers(mlps[0], input_channels)])
        self.finetuning = model_cfg.get('FT', False)
        self.num_point_features = mlps[0][-1]
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.grid_size = [int(v) for v in grid_size]

    def get_output_feature_dim(self):
        return self.num_point_features

    def _features(self, vox):
        """Point features -> MLP -> voxel max (temporal_dyn_vfe.py:85-119)."""
        m = vox['voxel_coords'].shape[0]
        perm, offsets = ops.segment_csr(vox['inverse'], m)
        vox['perm'], vox['offsets'] = perm, offsets
        layers = list(self.dvfe_mlps[0])
        first = layers[0]
        inv_rows, perm_rows = vox['inverse'], perm
        if ops.compute_dtype(vox['points']) == torch.bfloat16 and first.in_features <= 16:
            # bf16 autocast: the first Linear sees ABSOLUTE coordinates (up to 75 m), which bf16 would round to
            # 0.25-0.5 m steps -- coarser than the pillar.  The features arrive as hi + lo bf16 pairs instead
            # (~16 mantissa bits; the reference's fp16 AMP keeps 11) and the Linear contracts over both halves.
            # rows sorted by voxel from here on (the MLP and the norms are row-wise / order-free): the voxel max below and its
            # backward walk consecutive rows instead of gathering 256-byte rows all over the point list
            _, x2, inv_rows = ops.vfe_point_features_bf16x2(vox['points'], vox['point_coords'], vox['inverse'], perm, offsets,
                                                            m, self.point_cloud_range, self.voxel_size, csr_order=True)
            perm_rows = None
            x = ops.linear_split_input(x2, first.weight)
            layers = layers[1:]
        else:
            _, x = ops.vfe_point_features(vox['points'], vox['point_coords'], vox['inverse'], perm, offsets, m,
                                          self.point_cloud_range, self.voxel_size)
        norms = [i for i, layer in enumerate(layers) if isinstance(layer, (nn.BatchNorm1d, nn.SyncBatchNorm))]
        last_norm = norms[-1] if norms and not any(isinstance(layer, nn.Linear) for layer in layers[norms[-1]:]) else -1
        for i, layer in enumerate(layers):
            if isinstance(layer, nn.Linear):
                w = layer.weight
                if x.shape[1] % 8:           # 10 (11) point features: zero-pad the contraction to 16 so that the
                    pad = 8 - x.shape[1] % 8  # weight gradient runs in the token-split kernel (k must be 8-aligned)
                    x = torch.nn.functional.pad(x, (0, pad))
                    w = torch.nn.functional.pad(w, (0, pad))
                x = ops.linear(x, w, None)
            elif isinstance(layer, (nn.BatchNorm1d, nn.SyncBatchNorm)):
                if i == last_norm:
                    # the MLP's last norm + ReLU are applied by the voxel max as it reads the rows: no normalised [points, c] tensor
                    x_max, _ = ops.bn_relu_scatter_max(x, layer, inv_rows, perm_rows, offsets, m)
                    return x_max
                x = ops.batch_norm_relu(x, layer, relu=True)         # the ReLU that follows is fused
        x_max, _ = ops.scatter_max(x, inv_rows, perm_rows, offsets, m)
        return x_max

    _PREFETCH = '_vfe_prefetch'

    def prefetch(self, batch_dict):
        """Voxelise both frames of a batch that a LATER forward() will consume and start the copy of their counts to the host
        (train_one_step(next_batch=...): enqueued between the current step's forward and backward).  forward() then finds the
        counts on the host instead of stalling for them at the top of the step -- the launching thread was parked there until the
        GPU had finished the whole previous step, and the small index kernels behind the wait arrived one by one on an empty
        queue (0.67 ms of idle GPU per step, profiles/round6_i_kernel_stats.md).  Index work only: nothing here needs gradients."""
        pts = batch_dict.get('points', None)
        if self._PREFETCH in batch_dict or not (torch.is_tensor(pts) and pts.is_cuda):
            return
        bs = int(batch_dict['batch_size'])
        with torch.no_grad():
            launched = [ops.voxelize_launch(batch_dict[k], bs, self.point_cloud_range, self.voxel_size, self.grid_size)
                        for k in ('points', 'points_prev')]
            batch_dict[self._PREFETCH] = (launched, ops.HostCopy(torch.stack([o['counts'] for o in launched]), 'vfe_counts'))

    def forward(self, batch_dict, **kwargs):
        bs = int(batch_dict['batch_size'])
        pre = batch_dict.pop(self._PREFETCH, None)
        if pre is not None:
            launched, counts = pre[0], pre[1].get()
        else:
            launched = [ops.voxelize_launch(batch_dict[k], bs, self.point_cloud_range, self.voxel_size, self.grid_size)
                        for k in ('points', 'points_prev')]
            counts = ops.to_host(torch.stack([o['counts'] for o in launched]))     # the one host sync of the VFE
        for suffix, out, cnt in (('', launched[0], counts[0]), ('_prev', launched[1], counts[1])):
            vox = ops.voxelize_finish(out, cnt)
            x = self._features(vox)
            batch_dict['points' + suffix] = vox['points']
            batch_dict['point_coords' + suffix] = vox['point_coords']
            batch_dict['point_inverse_indices' + suffix] = vox['inverse']
            batch_dict['voxel_coords' + suffix] = vox['voxel_coords']
            batch_dict['voxel_features' + suffix] = x
            # extras the backbone reuses instead of recomputing (not in the reference's dict)
            batch_dict['voxels_per_sample' + suffix] = vox['voxels_per_sample']
            batch_dict['point_csr' + suffix] = (vox['perm'], vox['offsets'])
        if self.finetuning:
            for k in ('points', 'point_coords', 'point_inverse_indices'):
                batch_dict.pop(k), batch_dict.pop(k + '_prev')
        return batch_dict
