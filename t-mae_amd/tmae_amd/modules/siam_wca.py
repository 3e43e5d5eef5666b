"""SiamWCA: the fine-tune backbone (pcdet/models/backbones_3d/SiamWCA.py:450-667) -- the pre-training encoder
without masking: both frames through the Siamese SST blocks, window cross-attention per scale, dense BEV fusion.
Same kernels and the same module tree as SiamWCA_MAE; the fusion modules are called `deblocks` / `conv_out` here
(SiamWCA.py:517-548), so pre-trained `sst_blocks.*` / `wca_blocks.*` weights load by name."""
import torch.nn as nn

from .siam_wca_mae import SiamWCA_MAE
from .sst import SSTBlockV1, WCABlock


class SiamWCA(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, voxel_size, point_cloud_range, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.grid_size = [int(v) for v in grid_size]
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.sparse_shape = [self.grid_size[1], self.grid_size[0]]
        asym = model_cfg.get('ASYMMETRIC', None)
        if asym is not None and asym.get('ENABLED', False):
            raise NotImplementedError('ASYMMETRIC encoders are not used by the shipped T-MAE configs')
        in_channels = input_channels
        self.sst_blocks = nn.ModuleList()
        for c in model_cfg.SST_BLOCK_LIST:
            self.sst_blocks.append(SSTBlockV1(c, in_channels, c.NAME))
            in_channels = c.ENCODER.D_MODEL
        for blk, nxt in zip(list(self.sst_blocks)[:-1], list(self.sst_blocks)[1:]):   # consecutive strided stages: one sync
            if blk.conv_down is not None and nxt.conv_down is not None:
                getattr(blk.conv_down, '0').lookahead = True
        self.wca_blocks = nn.ModuleList([WCABlock(c, c.ENCODER.D_MODEL, c.NAME) for c in model_cfg.SST_BLOCK_LIST])
        in_channels = 0
        self.deblocks = nn.ModuleList()
        for src in model_cfg.FEATURES_SOURCE:
            c = model_cfg.FUSE_LAYER[src]
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c.NUM_FILTER, c.NUM_UPSAMPLE_FILTER, c.UPSAMPLE_STRIDE, stride=c.UPSAMPLE_STRIDE,
                                   bias=False),
                nn.BatchNorm2d(c.NUM_UPSAMPLE_FILTER, eps=1e-3, momentum=0.01),
                nn.ReLU(inplace=True)))
            in_channels += c.NUM_UPSAMPLE_FILTER
        n_src = len(self.deblocks)
        self.conv_out = nn.Sequential(
            nn.Conv2d(in_channels, in_channels // n_src, 3, padding=1, bias=False),
            nn.BatchNorm2d(in_channels // n_src, eps=1e-3, momentum=0.01),
            nn.ReLU(inplace=True))
        self.num_point_features = in_channels // n_src
        self.last_pair_tokens = []
        self.pair_encode = True

    # the encoder / cross-attention / fusion code is SiamWCA_MAE's (same attribute names where it matters)
    sparse_encode = SiamWCA_MAE.sparse_encode
    sparse_encode_pair = SiamWCA_MAE.sparse_encode_pair
    sparse_cross_attn = SiamWCA_MAE.sparse_cross_attn

    @property
    def decoder_deblocks(self):
        return self.deblocks

    @property
    def decoder_conv_out(self):
        return self.conv_out

    dense_conv = SiamWCA_MAE.dense_conv

    def forward(self, batch_dict):
        bs = int(batch_dict['batch_size'])
        cur_f, cur_c = batch_dict['voxel_features'], batch_dict['voxel_coords']
        prv_f, prv_c = batch_dict['voxel_features_prev'], batch_dict['voxel_coords_prev']
        if self.pair_encode and self.training and prv_c.shape[0] > 1 and cur_c.shape[0] > 1:
            feats_prev, feats, strides = self.sparse_encode_pair(prv_f, prv_c, cur_f, cur_c, bs)
        else:
            feats_prev, _ = self.sparse_encode(prv_f, prv_c, bs, previous_sstblock=True)
            feats, strides = self.sparse_encode(cur_f, cur_c, bs)
        strides = {k: 2 ** (i + 1) for i, k in enumerate(feats)}            # SiamWCA.py:577-580
        feats = self.sparse_cross_attn(feats, feats_prev, dtime=batch_dict.get('dt', 0))
        spatial, spatial_stride = self.dense_conv(feats, strides)
        batch_dict['multi_scale_3d_features'] = feats
        batch_dict['multi_scale_3d_strides'] = strides
        batch_dict['spatial_features'] = spatial
        batch_dict['spatial_features_stride'] = spatial_stride
        return batch_dict
