"""SSTBEVBackbone (pcdet/models/backbones_2d/sst_bev_backbone.py:6-43): a stack of 3x3 Conv2d (+dilation) +
BatchNorm2d(eps 1e-3, momentum 0.01) + ReLU with residual shortcuts.  In bf16 training the 128 -> 128
convolutions (dilation 1 and 2) run on the halo-tiled implicit GEMM of csrc/spconv_igemm.hip (forward and input gradient)
and the token-split weight-gradient kernel (ops.dense_conv3x3; TMAE_DENSE_CONV=miopen: the library's), everything in fp32
on MIOpen (channels-last); the norm + ReLU run on the row kernels of csrc/batchnorm.hip over the
[B*Y*X, C] view."""
import os

import torch
import torch.nn as nn

from .. import ops


def conv_bn_relu_nhwc(seq, x, shortcut=False, chain=False):
    """seq = Sequential(Conv2d, BatchNorm2d, ReLU) on a channels-last tensor; fused norm+ReLU in training.
    chain: returns (result, what the next module that reads the same x should read instead) -- an alias of x through which the
    readers' input gradients accumulate (ops._Conv3x3C64) where the 64 -> 64 kernel takes the conv, x itself otherwise.
    shortcut: returns seq(x) + x (sst_bev_backbone.py:35-41).  On the halo-conv path the shortcut is added inside the norm's apply
    kernel forward and inside the conv's input-gradient kernel backward (TMAE_BEV_SHORTCUT=add: as separate elementwise passes)."""
    if chain:
        assert not shortcut
        res = _conv_bn_relu_nhwc(seq, x, False, True)
        return res if isinstance(res, tuple) else (res, x)
    return _conv_bn_relu_nhwc(seq, x, shortcut, False)


def _conv_bn_relu_nhwc(seq, x, shortcut, chain):
    conv, bn = seq[0], seq[1]
    chained = None
    nhwc = x.permute(0, 2, 3, 1)
    if (bn.training and os.environ.get('TMAE_DENSE_CONV', 'halo') != 'miopen' and nhwc.is_contiguous()
            and ops.dense_conv3x3_ok(nhwc, conv) and conv.out_channels % 128 == 0):
        if (shortcut and conv.out_channels == conv.in_channels and conv.out_channels in (64, 128, 256) and len(seq) == 3
                and isinstance(seq[2], nn.ReLU) and os.environ.get('TMAE_BEV_SHORTCUT', 'fused') == 'fused'):
            y, res = ops.dense_conv3x3(nhwc, conv.weight, conv.dilation[0], fork=True)
            b, ny, nx, c = y.shape
            rows = ops.batch_norm_relu(y.reshape(b * ny * nx, c), bn, relu=True, pre_bias=conv.bias,
                                       post=res.reshape(b * ny * nx, c))
            return rows.view(b, ny, nx, c).permute(0, 3, 1, 2)
        y = ops.dense_conv3x3(nhwc, conv.weight, conv.dilation[0]).permute(0, 3, 1, 2)
        fused = True
    elif bn.training and not shortcut and ops.conv3x3_c64_ok(nhwc, conv):
        # CenterHead's 64 -> 64 stems (center_head.py:28-31): csrc/headconv.hip (a bias is folded into the norm below, as in the
        # library branch)
        if chain:
            y, nxt = ops.conv3x3_c64(nhwc, conv.weight, chain=True)
            y, chained = y.permute(0, 3, 1, 2), nxt.permute(0, 3, 1, 2)
        else:
            y = ops.conv3x3_c64(nhwc, conv.weight).permute(0, 3, 1, 2)
        fused = True
    elif bn.training and not shortcut and ops.conv3x3_c128to64_ok(nhwc, conv):
        y = ops.conv3x3_c128to64(nhwc, conv.weight).permute(0, 3, 1, 2)         # CenterHead's shared conv (center_head.py:85-89)
        fused = True
    else:
        fused = (bn.training and x.is_cuda and conv.out_channels in (64, 128, 256) and len(seq) == 3
                 and isinstance(seq[2], nn.ReLU))
        # a bias in front of the training-mode norm (USE_BIAS_BEFORE_NORM) is folded away: see ops.batch_norm_relu
        y = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups) \
            if (fused and conv.bias is not None) else conv(x)
    if (fused and y.is_contiguous(memory_format=torch.channels_last)
            and y.shape[1] in (64, 128, 256) and len(seq) == 3 and isinstance(seq[2], nn.ReLU)):
        b, c, ny, nx = y.shape
        rows = ops.batch_norm_relu(y.permute(0, 2, 3, 1).reshape(b * ny * nx, c), bn, relu=True, pre_bias=conv.bias)
        out = rows.view(b, ny, nx, c).permute(0, 3, 1, 2)
        if chained is not None:
            return out, chained
        return out + x if shortcut else out
    if fused and conv.bias is not None:
        y = y + conv.bias.view(1, -1, 1, 1).to(y.dtype)
    y = bn(y)
    y = seq[2](y) if len(seq) > 2 else y
    return y + x if shortcut else y


class SSTBEVBackbone(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        input_channels = model_cfg.NUM_FILTER
        self.conv_shortcut = list(model_cfg.CONV_SHORTCUT)
        layers = []
        for kw in model_cfg.CONV_KWARGS:
            kw = dict(kw)
            layers.append(nn.Sequential(
                nn.Conv2d(input_channels, **kw, bias=False),
                nn.BatchNorm2d(kw['out_channels'], eps=1e-3, momentum=0.01),
                nn.ReLU(inplace=True)))
            input_channels = kw['out_channels']
        self.conv_layer = nn.ModuleList(layers)
        self.num_bev_features = input_channels

    def forward(self, data_dict):
        out = data_dict['spatial_features']
        for i, conv in enumerate(self.conv_layer):
            sc = i in self.conv_shortcut and conv[0].out_channels == out.shape[1] and conv[0].stride == (1, 1)
            out = conv_bn_relu_nhwc(conv, out, shortcut=sc)
        data_dict['spatial_features_2d'] = out
        return data_dict
