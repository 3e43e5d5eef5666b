"""SiamWCA_MAE: Siamese SST encoder + window cross-attention + dense BEV decoder + Chamfer loss.

Host-side mirror of pcdet/models/backbones_3d/SiamWCA_MAE.py (same constructor kwargs, batch_dict keys,
``forward_ret_dict`` and parameter names) over the HIP operators of tmae_amd.ops.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .sparse import SparseConvTensor, prefetch_down
from .sst import SSTBlockV1, WCABlock


_SHIFTS = {}
_BYX = {}


def _byx(coords):
    """Columns (b, y, x) of [b, z, y, x] voxel coordinates as int32.  (`coords[:, [0, 2, 3]]` builds its index tensor
    from the Python list with a pageable, SYNCHRONOUS host-to-device copy: near the end of forward() the host waited
    ~10 ms there for the whole queued forward pass; the index is cached per device instead.)"""
    idx = _BYX.get(coords.device)
    if idx is None:
        idx = _BYX[coords.device] = torch.tensor([0, 2, 3], dtype=torch.long, device=coords.device)
    return coords.index_select(1, idx).int()



def _sample_shift(batch, device):
    """[batch, 0, 0] int32 on `device` (the current frame's samples follow the previous frame's in the joint token list)."""
    key = (int(batch), device)
    t = _SHIFTS.get(key)
    if t is None:
        t = _SHIFTS[key] = torch.tensor([int(batch), 0, 0], dtype=torch.int32, device=device)
    return t


class SiamWCA_MAE(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, voxel_size, point_cloud_range, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.grid_size = [int(v) for v in grid_size]
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.sparse_shape = [self.grid_size[1], self.grid_size[0]]           # (ny, nx)

        self.mask_cfg = model_cfg.get('MASK_CONFIG', None)
        self.mask_ratio = self.mask_cfg.RATIO if self.mask_cfg is not None else 0.0
        if model_cfg.get('ASYMMETRIC', False) and model_cfg.ASYMMETRIC.get('ENABLED', False):
            raise NotImplementedError('ASYMMETRIC encoders are not used by the shipped T-MAE configs')

        in_channels = input_channels
        self.sst_blocks = nn.ModuleList()
        for cfg in model_cfg.SST_BLOCK_LIST:
            self.sst_blocks.append(SSTBlockV1(cfg, in_channels, cfg.NAME))
            in_channels = cfg.ENCODER.D_MODEL
        for blk, nxt in zip(list(self.sst_blocks)[:-1], list(self.sst_blocks)[1:]):      # consecutive strided stages share one sync
            if blk.conv_down is not None and nxt.conv_down is not None:
                getattr(blk.conv_down, '0').lookahead = True
        self.wca_blocks = nn.ModuleList()
        for cfg in model_cfg.SST_BLOCK_LIST:
            self.wca_blocks.append(WCABlock(cfg, cfg.ENCODER.D_MODEL, cfg.NAME))

        in_channels = 0
        self.decoder_deblocks = nn.ModuleList()
        self.deblock_strides = []
        for src in model_cfg.FEATURES_SOURCE:
            c = model_cfg.FUSE_LAYER[src]
            self.decoder_deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c.NUM_FILTER, c.NUM_UPSAMPLE_FILTER, c.UPSAMPLE_STRIDE, stride=c.UPSAMPLE_STRIDE,
                                   bias=False),
                nn.BatchNorm2d(c.NUM_UPSAMPLE_FILTER, eps=1e-3, momentum=0.01),
                nn.ReLU(inplace=True)))
            in_channels += c.NUM_UPSAMPLE_FILTER
            self.deblock_strides.append(c.UPSAMPLE_STRIDE)
        n_src = len(self.decoder_deblocks)
        self.decoder_conv_out = nn.Sequential(
            nn.Conv2d(in_channels, in_channels // n_src, 3, padding=1, bias=False),
            nn.BatchNorm2d(in_channels // n_src, eps=1e-3, momentum=0.01),
            nn.ReLU(inplace=True))
        in_channels = in_channels // n_src
        self.decoder_pred = nn.Linear(in_channels, self.mask_cfg.NUM_PRD_POINTS * 3, bias=True)
        self.forward_ret_dict = {}
        self.num_point_features = in_channels
        self.last_pair_tokens = []     # rows per stage of the last sparse_encode_pair call (both frames)
        self.last_stage_indices = []   # per stage (prev indices, cur indices, (ny, nx)) of the last forward (bench FLOP model)
        self.pair_encode = True        # both frames through the Siamese encoder as one token list (sparse_encode_pair)

    # ------------------------------------------------------------------ masking (SiamWCA_MAE.py:166-182)
    def mask_voxels(self, all_voxel_features, all_voxel_coords, batch_size, voxels_per_sample, noise=None):
        m = all_voxel_coords.shape[0]
        dev = all_voxel_coords.device
        if noise is None:
            noise = torch.rand(m, device=dev)
        offs = np.concatenate([[0], np.cumsum(voxels_per_sample)]).astype(np.int32)
        keep_frac = 1 - self.mask_ratio
        n_vis = int(sum(int(L * keep_frac) for L in voxels_per_sample))        # known on the host: no sync
        offs_dev = torch.from_numpy(offs).pin_memory().to(dev, non_blocking=True)      # (a pageable copy is synchronous)
        mask, vis_index, _ = ops.random_mask(noise, offs_dev, batch_size, keep_frac)
        vis = vis_index[:n_vis].long()
        return ops.gather_rows(all_voxel_features, vis), all_voxel_coords[vis], mask       # vis: distinct rows

    # ------------------------------------------------------------------ encoder (SiamWCA_MAE.py:184-218)
    def sparse_encode(self, voxel_features, voxel_coords, batch_size, previous_sstblock=False):
        x = SparseConvTensor(voxel_features, _byx(voxel_coords), self.sparse_shape,
                             batch_size)
        feats, strides = {}, {}
        for i, blk in enumerate(self.sst_blocks):                   # Siamese: the same weights for both frames
            x = blk(x)
            feats[f'x_conv{i + 1}'] = x
            strides[f'x_conv{i + 1}'] = self.sparse_shape[0] // x.spatial_shape[0]
        return feats, strides

    def sparse_encode_pair(self, feats_prev, coords_prev, feats_cur, coords_cur, batch_size):
        """Both frames through the Siamese encoder as ONE token list (samples 0..B-1 = previous frame, B..2B-1 =
        current frame): every kernel runs once on the larger list and every shared weight gets one gradient.
        Attention and the sparse convs never mix samples, and BatchNorm keeps per-frame statistics (`groups`), so
        the results are those of the reference's two calls (SiamWCA_MAE.py:262-263, :289)."""
        B = batch_size
        ind_p = _byx(coords_prev)
        ind_c = _byx(coords_cur)
        shift = _sample_shift(B, ind_c.device)       # cached: a torch.tensor(..., device=) per step is a synchronous copy
        ind_c = ind_c + shift
        cdt = ops.compute_dtype(feats_prev)
        x = SparseConvTensor(torch.cat([feats_prev.to(cdt), feats_cur.to(cdt)], 0), torch.cat([ind_p, ind_c], 0),
                             self.sparse_shape, 2 * B, groups=((ind_p.shape[0], B), (ind_c.shape[0], B)))
        out_p, out_c, strides = {}, {}, {}
        self.last_pair_tokens, self.last_stage_indices = [], []
        prefetch_down(self.sst_blocks, x)
        for i, blk in enumerate(self.sst_blocks):
            x = blk(x)
            key = f'x_conv{i + 1}'
            m0 = x.groups[0][0]
            self.last_pair_tokens.append(int(x.features.shape[0]))
            ny, nx = x.spatial_shape
            # the two frames' rows: outputs of the block's last norm itself when it forked them (their gradients then meet the
            # next stage's inside the BatchNorm backward), else views whose backward is one concatenation
            f_p, f_c = x.frame_halves if x.frame_halves is not None else ops.split_rows(x.features, m0)
            # previous frame: its rows come first, so the first B samples of the row-index grid are already its grid
            out_p[key] = SparseConvTensor(f_p, x.indices[:m0], x.spatial_shape, B, grid=x.grid[:B * ny * nx])
            out_c[key] = SparseConvTensor(f_c, x.indices[m0:] - shift, x.spatial_shape, B)
            strides[key] = self.sparse_shape[0] // ny
            self.last_stage_indices.append((out_p[key].indices, out_c[key].indices, (ny, nx)))
        return out_p, out_c, strides

    def sparse_cross_attn(self, feats, feats_prev, dtime=0):
        for i, blk in enumerate(self.wca_blocks):
            feats[f'x_conv{i + 1}'] = blk(feats[f'x_conv{i + 1}'], feats_prev[f'x_conv{i + 1}'], dtime)
        return feats

    # ------------------------------------------------------------------ decoder (SiamWCA_MAE.py:231-253)
    def dense_conv(self, feats, strides, gather=None):
        """gather = (rowmap, indices) of the sites whose decoder feature the caller wants (forward(): every current voxel): returns
        (spatial, stride, rows) with rows = spatial at those sites, taken inside the last norm's autograd node
        (ops.batch_norm_relu_gather) when that norm runs on the fused row kernels."""
        out_strides, sources = [], []
        for i, src in enumerate(self.model_cfg.FEATURES_SOURCE):
            sp = feats[src]
            blk = self.decoder_deblocks[i]
            sources.append((sp.features, sp.grid, sp.indices, sp.spatial_shape, blk[0], blk[1]))
            out_strides.append(strides[src] // self.model_cfg.FUSE_LAYER[src].UPSAMPLE_STRIDE)
        if ops.deblocks_fusable(sources, self.training):
            # one write of the concat buffer instead of dense() + deconv + BN + ReLU + cat over dense tensors
            cat = ops.deblocks_to_dense(sources, feats[self.model_cfg.FEATURES_SOURCE[0]].batch_size,
                                        self.sparse_shape[0], self.sparse_shape[1]).permute(0, 3, 1, 2)
        else:
            cat = torch.cat([self.decoder_deblocks[i](feats[src].dense())
                             for i, src in enumerate(self.model_cfg.FEATURES_SOURCE)], dim=1)
        conv, bn = self.decoder_conv_out[0], self.decoder_conv_out[1]
        nhwc = cat.permute(0, 2, 3, 1)
        moments = None
        if (self.training and nhwc.is_contiguous() and ops.dense_conv3x3_ok(nhwc, conv)
                and os.environ.get('TMAE_DENSE_CONV', 'native') != 'miopen'):
            # (the conv's epilogue also returns the column sums of y and y^2: the norm below needs no statistics pass over y)
            y, moments = ops.dense_conv3x3(nhwc, conv.weight, moments=True)
            y = y.permute(0, 3, 1, 2)
        else:
            y = conv(cat)
        if (self.training and y.is_cuda and y.is_contiguous(memory_format=torch.channels_last)
                and y.shape[1] in (64, 128, 256) and isinstance(self.decoder_conv_out[2], nn.ReLU)):
            # BatchNorm2d + ReLU over a channels-last tensor = the row kernels over [B*Y*X, C]
            b, c, ny, nx = y.shape
            if gather is not None:
                rows, picked = ops.batch_norm_relu_gather(y.permute(0, 2, 3, 1).reshape(b * ny * nx, c), bn, True, gather[0], gather[1],
                                                          b, ny, nx, moments=moments)
                return rows.view(b, ny, nx, c).permute(0, 3, 1, 2), out_strides[0], picked
            rows = ops.batch_norm_relu(y.permute(0, 2, 3, 1).reshape(b * ny * nx, c), bn, relu=True)
            spatial = rows.view(b, ny, nx, c).permute(0, 3, 1, 2)
        else:
            spatial = self.decoder_conv_out[2](bn(y))
        if gather is not None:
            return spatial, out_strides[0], ops.dense_gather(spatial.permute(0, 2, 3, 1), gather[0], gather[1])
        return spatial, out_strides[0]

    # ------------------------------------------------------------------ targets (SiamWCA_MAE.py:124-152)
    def target_assigner(self, batch_dict):
        voxel_features = batch_dict['voxel_features']
        voxel_coords = batch_dict['voxel_coords']
        perm, offsets = batch_dict['point_csr']
        _, gt = ops.group_points(batch_dict['points'], voxel_coords, perm, offsets, self.mask_cfg.NUM_GT_POINTS,
                                 self.point_cloud_range, self.voxel_size, want_inds=False)
        pred = ops.linear(voxel_features, self.decoder_pred.weight, self.decoder_pred.bias).view(voxel_features.shape[0], -1, 3)
        return {'pred_points': pred, 'gt_points': gt, 'mask': batch_dict['voxel_mae_mask']}

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        r = self.forward_ret_dict
        loss, _ = ops.chamfer_distance(r['pred_points'].float(), r['gt_points'], weights=r['mask'])
        return loss, tb_dict

    # ------------------------------------------------------------------ forward (SiamWCA_MAE.py:255-322)
    def forward(self, batch_dict):
        bs = int(batch_dict['batch_size'])
        all_feats, all_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
        vis_feats, vis_coords, mask = self.mask_voxels(all_feats, all_coords, bs, batch_dict['voxels_per_sample'],
                                                      batch_dict.get('mae_noise', None))
        batch_dict['voxel_mae_mask'] = mask
        if self.pair_encode and batch_dict['voxel_coords_prev'].shape[0] > 1 and vis_coords.shape[0] > 1:
            feats_prev, feats, strides = self.sparse_encode_pair(
                batch_dict['voxel_features_prev'], batch_dict['voxel_coords_prev'], vis_feats, vis_coords, bs)
        else:                                                       # the reference's two calls
            feats_prev, _ = self.sparse_encode(batch_dict['voxel_features_prev'], batch_dict['voxel_coords_prev'], bs,
                                               previous_sstblock=True)
            feats, strides = self.sparse_encode(vis_feats, vis_coords, bs)
            self.last_stage_indices = [(feats_prev[k].indices, feats[k].indices, tuple(feats[k].spatial_shape))
                                       for k in feats]
        feats = self.sparse_cross_attn(feats, feats_prev, dtime=batch_dict.get('dt', 0))
        all_ind = _byx(all_coords)
        grid_all = ops.index_grid(all_ind, bs, self.sparse_shape[0], self.sparse_shape[1])
        spatial, spatial_stride, pyramid = self.dense_conv(feats, strides, gather=(grid_all, all_ind))   # decoder feature at EVERY current voxel
        batch_dict['multi_scale_3d_features'] = feats
        batch_dict['multi_scale_3d_strides'] = strides
        batch_dict['spatial_features'] = spatial
        batch_dict['spatial_features_stride'] = spatial_stride
        assert spatial.shape[0] == bs and spatial.shape[2] == self.grid_size[1] and spatial.shape[3] == self.grid_size[0]

        batch_dict.update({
            'voxel_features': pyramid, 'voxel_coords': all_coords,
            'voxel_shuffle_inds': torch.arange(all_coords.shape[0], device=all_coords.device, dtype=torch.long)})
        self.forward_ret_dict = self.target_assigner(batch_dict)
        return batch_dict
