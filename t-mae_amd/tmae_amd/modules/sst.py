"""SST encoder stage (SSTBlockV1) and window cross-attention stage (WCABlock) on ragged window attention.

Host-side mirror of pcdet/models/backbones_3d/spt_backbone.py:11-353, SiamWCA.py:21-447,
model_utils/{sst_basic_block,wca_block,cosine_msa}.py: same module tree and parameter names
(``encoder_blocks.{j}.encoder_list.{k}.win_attn.self_attn.{in_proj_weight,in_proj_bias,tau,out_proj.*}`` ...),
so reference checkpoints load unchanged.  What differs is the execution: there is no SSTInputLayer
bucketing into padded [windows, T, C] tensors, no per-drop-level Python loop and no host sync -- one
dense index grid per sparse tensor and one ragged attention launch per layer.
"""
import os
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .sparse import SparseConvTensor, post_act_block


def pos_embed_table(feat_dim, window_shape, pos_temperature, normalize_pos=False):
    """Sin/cos embedding of every in-window position, [wy*wx, feat_dim] float32 (cell = y*wx + x).
    Formula of SSTInputLayer.get_pos_embed (spt_backbone.py:186-224)."""
    wx, wy = int(window_shape[0]), int(window_shape[1])
    ys, xs = torch.meshgrid(torch.arange(wy), torch.arange(wx), indexing='ij')
    y = ys.reshape(-1) - wy / 2
    x = xs.reshape(-1) - wx / 2
    if normalize_pos:
        x = x / wx * 2 * 3.1415
        y = y / wy * 2 * 3.1415
    pos_length = feat_dim // 2
    inv_freq = torch.arange(pos_length, dtype=torch.float32)
    inv_freq = pos_temperature ** (2 * torch.div(inv_freq, 2, rounding_mode='floor') / pos_length)
    ex = x[:, None] / inv_freq[None, :]
    ey = y[:, None] / inv_freq[None, :]
    ex = torch.stack([ex[:, ::2].sin(), ex[:, 1::2].cos()], dim=-1).flatten(1)
    ey = torch.stack([ey[:, ::2].sin(), ey[:, 1::2].cos()], dim=-1).flatten(1)
    table = torch.cat([ex, ey], dim=-1).float().contiguous()
    assert table.shape == (wx * wy, feat_dim)
    return table


class CosineMultiheadAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (cosine_msa.py:441-458): packed in-proj,
    out_proj, learnable temperature tau clamped at tau_min: (1,1,1) shared by the heads, or (1, num_heads, 1, 1) with
    non_shared_tau (cosine_msa.py:453-456)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, tau_min=0.01, cosine=True, non_shared_tau=False):
        super().__init__()
        if dropout != 0.0 or not cosine:
            raise NotImplementedError('T-MAE configs use cosine attention, dropout 0')
        self.embed_dim, self.num_heads, self.tau_min = embed_dim, num_heads, tau_min
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.tau = nn.Parameter(torch.ones(1, num_heads, 1, 1) if non_shared_tau else torch.ones(1, 1, 1))
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class WindowPlan:
    """Everything a stage's attention layers need about one sparse index set: indices, dense grid, shape, and --
    built once, shared by all layers of the stage -- the per-shift window work lists (region batching as index
    lists; `other` = the key frame of the cross-attention stage)."""

    def __init__(self, sp: SparseConvTensor, other: 'WindowPlan' = None):
        self.indices, self.grid = sp.indices, sp.grid
        self.batch, (self.ny, self.nx) = sp.batch_size, sp.spatial_shape
        self.key_grid = self.grid if other is None else other.grid
        self._wl = {}
        self._cells = {}
        self._shift_grids = {}           # shift -> (query grid, key grid) with dropped tokens masked out (token dropping)

    def set_shift_grids(self, shift, grid_q, grid_k):
        """Attention of `shift` sees these grids instead of the stage's: a token whose cell reads -1 does not exist for
        it (DROP_INFO with max_tokens below the window's voxel count: per-shift keep sets, SiamWCA.py:141-215)."""
        self._shift_grids[bool(shift)] = (grid_q, grid_k)
        self._wl.pop(bool(shift), None)

    def grid_q(self, shift):
        return self._shift_grids.get(bool(shift), (self.grid, self.key_grid))[0]

    def grid_k(self, shift):
        return self._shift_grids.get(bool(shift), (self.grid, self.key_grid))[1]

    def covered(self, shift):
        """every token of the stage (and of the key frame) lies in a window of this shift's grids: true unless token dropping
        masked some out (set_shift_grids) -- ops.win_attn may then zero only the orphan rows instead of whole outputs"""
        return bool(shift) not in self._shift_grids

    def worklist(self, shift):
        shift = bool(shift)
        if shift not in self._wl:
            self._wl[shift] = ops.window_worklist(self.grid_q(shift), self.grid_k(shift), self.batch, self.ny, self.nx, shift)
        return self._wl[shift]

    def cells(self, shift, window_shape):
        """cell bytes of the tokens for this shift (ops.window_cells), shared by the layers of the stage."""
        shift = bool(shift)
        if shift not in self._cells:
            self._cells[shift] = ops.window_cells(self.indices, window_shape, shift)
        return self._cells[shift]


# TMAE_POS_FOLD=0: the in-projections run on a materialised x + pos (two GEMMs) instead of the position-folded GEMM
_POS_FOLD = os.environ.get('TMAE_POS_FOLD', '1') != '0'
# TMAE_FFN_GELU=pass: linear1 and the GELU as two launches (A/B of the dual-store epilogue, profiles/scripts/ab_gelu.sh)
_FFN_GELU_FUSED = os.environ.get('TMAE_FFN_GELU', 'fused') != 'pass'
# TMAE_FFN_RESIDUAL=add: `src + linear2(act)` as the first summand pair of norm2 (two tensors read, the sum written by the norm)
# instead of out of linear2's GEMM (ops.gelu_linear(residual=...), tmae_token_gemm_res)
_FFN_RESIDUAL_FUSED = os.environ.get('TMAE_FFN_RESIDUAL', 'fused') != 'add'


class WindowAttention(nn.Module):
    def __init__(self, d_model, nhead, dropout, layer_cfg):
        super().__init__()
        self.nhead = nhead
        if not layer_cfg.get('cosine', False):
            raise NotImplementedError('non-cosine window attention is not on the T-MAE path')
        self.self_attn = CosineMultiheadAttention(d_model, nhead, dropout=dropout,
                                                  tau_min=layer_cfg.get('tau_min', 0.01),
                                                  non_shared_tau=layer_cfg.get('non_shared_tau', False))

    def forward(self, x, plan, pos_table, window_shape, shift):
        """q = k = x + pos, v = x (sst_basic_block.py:22-54), windows of `shift` read from plan.grid.
        Returns (attention output, x's alias for the residual branch): see ops.proj_fork."""
        a = self.self_attn
        d = a.embed_dim
        if _POS_FOLD and ops._pos_proj_ok(x, d, 3 * d):
            qkv, x_res = ops.pos_proj(x, a.in_proj_weight, a.in_proj_bias, 0, 3 * d, 0, 2 * d,
                                      plan.cells(shift, window_shape), ops.pos_axes(pos_table, window_shape), fork=True,
                                      inplace_dx=True)
            o = ops.win_attn(qkv, None, None, a.tau, plan.grid, plan.grid, self.nhead, plan.batch, plan.ny, plan.nx,
                             shift, a.tau_min, worklist=plan.worklist(shift))
            return ops.linear(o, a.out_proj.weight, a.out_proj.bias), x_res
        qk, v, x_res = ops.proj_fork(x, a.in_proj_weight, a.in_proj_bias, ((0, 2 * d, True), (2 * d, 3 * d, False)),
                                     pos=(plan.indices, pos_table, window_shape, shift), fork=True,
                                     inplace_dx=True)
        o = ops.win_attn(qk, v, None, a.tau, plan.grid, plan.grid, self.nhead, plan.batch, plan.ny, plan.nx,
                         shift, a.tau_min, worklist=plan.worklist(shift))
        return ops.linear(o, a.out_proj.weight, a.out_proj.bias), x_res


class WindowCrossAttention(nn.Module):
    def __init__(self, d_model, nhead, dropout, layer_cfg):
        super().__init__()
        self.nhead = nhead
        if not layer_cfg.get('cosine', False):
            raise NotImplementedError('non-cosine window attention is not on the T-MAE path')
        self.cross_attn = CosineMultiheadAttention(d_model, nhead, dropout=dropout,
                                                   tau_min=layer_cfg.get('tau_min', 0.01),
                                                   non_shared_tau=layer_cfg.get('non_shared_tau', False))

    def forward(self, x, plan, x_prv, plan_prv, pos_table, window_shape, shift, prv_alias=False):
        """q = cur + pos, k = prev + pos_prev, v = prev (wca_block.py:26-67).  Query tokens whose window is
        empty in the previous frame come back as zero rows (= not in keep_inds, wca_block.py:93-96).
        prv_alias: a third result, what the next cross layer should take as ITS x_prv (ops._PosProjCross kv_alias; x_prv itself
        where that node is not used)."""
        a = self.cross_attn
        d = a.embed_dim
        w, b = a.in_proj_weight, a.in_proj_bias
        if _POS_FOLD and ops._pos_proj_ok(x, d, d) and ops._pos_proj_ok(x_prv, d, 2 * d):
            E = ops.pos_axes(pos_table, window_shape)
            res = ops.pos_proj_cross(x, x_prv, w, b, plan.cells(shift, window_shape),
                                     plan_prv.cells(shift, window_shape), E, inplace_dx=True, kv_alias=prv_alias)
            q, kv, x_res = res[:3]
            o = ops.win_attn(q, kv, 'kv', a.tau, plan.grid_q(shift), plan.grid_k(shift), self.nhead, plan.batch, plan.ny,
                             plan.nx, shift, a.tau_min, worklist=plan.worklist(shift), covered=plan.covered(shift))
            return (o, x_res, res[3]) if prv_alias else (o, x_res)
        q, x_res = ops.proj_fork(x, w, b, ((0, d, True),), pos=(plan.indices, pos_table, window_shape, shift), fork=True,
                                 inplace_dx=True)
        k, v = ops.proj_fork(x_prv, w, b, ((d, 2 * d, True), (2 * d, 3 * d, False)),
                             pos=(plan_prv.indices, pos_table, window_shape, shift))
        o = ops.win_attn(q, k, v, a.tau, plan.grid_q(shift), plan.grid_k(shift), self.nhead, plan.batch, plan.ny, plan.nx,
                         shift, a.tau_min, worklist=plan.worklist(shift), covered=plan.covered(shift))
        return (o, x_res, x_prv) if prv_alias else (o, x_res)


def _activation(name):
    if name == 'gelu':
        return F.gelu
    if name == 'relu':
        return F.relu
    raise RuntimeError(f'activation should be relu/gelu, not {name}.')


class _EncoderTail(nn.Module):
    """linear1/linear2/norm1/norm2 shared by the self and cross layers (post-norm, dropout 0).  Children are
    registered in the reference's order -- win_attn, linear1, linear2, norm1, norm2 (sst_basic_block.py:64-73) --
    because the optimizer's parameter groups follow the module order (train/optim.py)."""

    def __init__(self, win_attn, d_model, dim_feedforward, activation):
        super().__init__()
        self.win_attn = win_attn
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.activation = _activation(activation)

    def tail(self, src, attn, bmask=None, passthrough=False, post=None):
        """src = LN1(src + attn); src = LN2(src + linear2(act(linear1(src)))) -- both adds fused into the norms.
        bmask: 0/1 row weights of attn (cross layers).  passthrough / post: the block residual (`x + encoder(x)`,
        spt_backbone.py:342-353) -- the block's first layer hands out an alias of its input (passthrough: returns
        (out, alias)), the last layer adds it to its output inside norm2 (post); see ops._AddLayerNorm."""
        alias = None
        if passthrough:
            src, alias = ops.add_layer_norm(src, attn, self.norm1.weight, self.norm1.bias, self.norm1.eps, bmask=bmask,
                                            passthrough=True)
        else:
            src = ops.add_layer_norm(src, attn, self.norm1.weight, self.norm1.bias, self.norm1.eps, bmask=bmask)
        if self.activation is F.gelu and _FFN_GELU_FUSED:
            # linear1 and the GELU in one launch (the activation is a second store of the GEMM's epilogue), the GELU
            # backward fused into the dX GEMM of linear2
            h_pre, h_act, src_res = ops.proj_fork(src, self.linear1.weight, self.linear1.bias,
                                                  ((0, self.linear1.out_features, False),), fork=True, inplace_dx=True, gelu=True)
            if _FFN_RESIDUAL_FUSED:
                # ... and `src + linear2(act)` out of linear2's GEMM (the residual tile rides in its LDS ring): norm2 reads one tensor
                xs = ops.gelu_linear(h_pre, self.linear2.weight, self.linear2.bias, h=h_act, residual=src_res)
                out = ops.add_layer_norm(xs, None, self.norm2.weight, self.norm2.bias, self.norm2.eps, post=post)
                return (out, alias) if passthrough else out
            src2 = ops.gelu_linear(h_pre, self.linear2.weight, self.linear2.bias, h=h_act)
        else:
            h_pre, src_res = ops.proj_fork(src, self.linear1.weight, self.linear1.bias,
                                           ((0, self.linear1.out_features, False),), fork=True, inplace_dx=True)
            if self.activation is F.gelu:
                src2 = ops.gelu_linear(h_pre, self.linear2.weight, self.linear2.bias)
            else:
                src2 = ops.linear(self.activation(h_pre), self.linear2.weight, self.linear2.bias)
        out = ops.add_layer_norm(src_res, src2, self.norm2.weight, self.norm2.bias, self.norm2.eps, post=post)
        return (out, alias) if passthrough else out


class EncoderLayer(_EncoderTail):
    """sst_basic_block.EncoderLayer (sst_basic_block.py:57-84)."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout, activation, layer_cfg):
        super().__init__(WindowAttention(d_model, nhead, dropout, layer_cfg), d_model, dim_feedforward, activation)

    def forward(self, src, plan, pos_table, window_shape, shift, passthrough=False, post=None):
        attn, src_res = self.win_attn(src, plan, pos_table, window_shape, shift)
        return self.tail(src_res, attn, passthrough=passthrough, post=post)


class BasicShiftBlockV2(nn.Module):
    """Two encoder layers: shift 0 then shift 1 (sst_basic_block.py:87-114)."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout, activation, layer_cfg):
        super().__init__()
        self.encoder_list = nn.ModuleList([EncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, layer_cfg)
                                           for _ in range(2)])

    def forward(self, src, plan, pos_table, window_shape, tap=False, post=None):
        """tap: the first layer also returns an alias of the block input (-> (out, alias)); post: added to the last
        layer's output (the residual of the enclosing SSTBlockV1)."""
        alias = None
        # the block residual is folded into the norms of the FIRST (tap) and LAST (post) layer: they must be two layers
        assert len(self.encoder_list) >= 2 or (not tap and post is None), 'residual folding needs >= 2 encoder layers'
        for i, layer in enumerate(self.encoder_list):
            last = i == len(self.encoder_list) - 1
            if tap and i == 0:
                src, alias = layer(src, plan, pos_table, window_shape, i == 1, passthrough=True)
            else:
                src = layer(src, plan, pos_table, window_shape, i == 1, post=post if last else None)
        return (src, alias) if tap else src


class WCAEncoderLayer(_EncoderTail):
    """wca_block.EncoderLayer (wca_block.py:70-103)."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout, activation, layer_cfg):
        super().__init__(WindowCrossAttention(d_model, nhead, dropout, layer_cfg), d_model, dim_feedforward, activation)

    def forward(self, src, plan, src_prv, plan_prv, pos_table, window_shape, shift, kept, passthrough=False, post=None,
                prv_alias=False):
        """prv_alias: the result becomes (result, what the next cross layer takes as src_prv) -- see WindowCrossAttention."""
        a = self.win_attn.cross_attn
        o, src_res, *nxt_prv = self.win_attn(src, plan, src_prv, plan_prv, pos_table, window_shape, shift, prv_alias=prv_alias)
        # src[keep] += out_proj(attn): kept = query rows whose window also holds previous-frame tokens (the out-proj
        # bias must not reach the other rows): a 0/1 row weight of the update inside the fused add + norm
        upd = ops.linear(o, a.out_proj.weight, a.out_proj.bias)
        out = self.tail(src_res, upd, bmask=kept, passthrough=passthrough, post=post)
        return (out, nxt_prv[0]) if prv_alias else out


class BasicShiftBlock_WCA(nn.Module):
    """Two cross layers, shift 0 then shift 1 (wca_block.py:106-145)."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout, activation, layer_cfg):
        super().__init__()
        self.encoder_list = nn.ModuleList([WCAEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation,
                                                           layer_cfg) for _ in range(2)])

    def forward(self, src, plan, src_prv, plan_prv, pos_table, window_shape, kept_list, residual=False):
        """residual: returns src + layers(src) (WCABlock.forward's `x + res`), the sum taken inside the last norm."""
        alias = None
        assert len(self.encoder_list) >= 2 or not residual, 'residual folding needs >= 2 cross layers'
        for i, layer in enumerate(self.encoder_list):
            last = i == len(self.encoder_list) - 1
            # every layer but the last hands the previous frame's rows on through its in-projection node: the layers' gradients for
            # them then meet inside the earlier layer's input-gradient GEMM instead of in an autograd add (ops._PosProjCross)
            if residual and i == 0:
                (src, alias), src_prv = layer(src, plan, src_prv, plan_prv, pos_table, window_shape, i == 1, kept_list[i],
                                              passthrough=True, prv_alias=True)
            elif not last:
                src, src_prv = layer(src, plan, src_prv, plan_prv, pos_table, window_shape, i == 1, kept_list[i], prv_alias=True)
            else:
                src = layer(src, plan, src_prv, plan_prv, pos_table, window_shape, i == 1, kept_list[i],
                            post=alias if (residual and last) else None)
        return src


def _check_preprocess(pre):
    ws = pre.WINDOW_SHAPE
    if list(ws) != [8, 8, 1]:
        raise NotImplementedError('window attention kernels are built for 8x8x1 windows (t_mae_ssl.yaml:61)')
    if pre.get('SHUFFLE_VOXELS', False):
        raise NotImplementedError('SHUFFLE_VOXELS is False in the T-MAE configs')
    if max(int(v['max_tokens']) for v in pre.DROP_INFO['train'].values()) > ws[0] * ws[1]:
        raise NotImplementedError('DROP_INFO max_tokens above the 64 cells of a window')


def _drop_info(pre):
    """(drop_info dict, can_drop): DROP_INFO['train'] -- the reference reads `self.training` at construction time, when it
    is still True (spt_backbone.py:32, SURVEY A-6) -- and whether any window can lose tokens at all: only if some level
    admits more voxels than its max_tokens.  With the shipped levels (16 / 32 / 64 for < 16 / < 32 / the rest, 64 cells
    per window) nobody is ever dropped and the token-dropping code below is never entered."""
    di = {int(k): dict(v) for k, v in pre.DROP_INFO['train'].items()}
    cells = int(pre.WINDOW_SHAPE[0]) * int(pre.WINDOW_SHAPE[1])
    can = any(min(int(v['drop_range'][1]) - 1, cells) > int(v['max_tokens']) for v in di.values())
    return di, can


def _mask_grid(plan, keep):
    """The plan's dense row-index grid with the cells of the tokens with keep == 0 set to -1 (no host sync)."""
    ind = plan.indices.long()
    cell = (ind[:, 0] * plan.ny + ind[:, 1]) * plan.nx + ind[:, 2]
    rows = torch.arange(ind.shape[0], device=ind.device, dtype=torch.int32)
    g = torch.full_like(plan.grid, -1)
    return g.scatter_(0, cell, torch.where(keep.bool(), rows, torch.full_like(rows, -1)))


class SSTBlockV1(nn.Module):
    """conv_down -> NUM_BLOCKS x BasicShiftBlockV2 -> residual -> conv_out (spt_backbone.py:267-353)."""

    def __init__(self, model_cfg, input_channels, indice_key, **kwargs):
        super().__init__()
        if kwargs.get('half_channels', False):
            raise NotImplementedError('ASYMMETRIC/HALF_CHANNELS is not used by the shipped T-MAE configs')
        self.model_cfg = model_cfg
        enc = model_cfg.ENCODER
        d_model, stride = enc.D_MODEL, enc.STRIDE
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.conv_down = None
        if stride > 1:
            self.conv_down = post_act_block(input_channels, d_model, 3, norm_fn=norm_fn, stride=stride, padding=1,
                                            indice_key=f'{indice_key}_spconv', conv_type='spconv', dim=2)
        _check_preprocess(model_cfg.PREPROCESS)
        self.window_shape = list(model_cfg.PREPROCESS.WINDOW_SHAPE)
        self.drop_info, self.can_drop = _drop_info(model_cfg.PREPROCESS)
        self.encoder_blocks = nn.ModuleList([
            BasicShiftBlockV2(d_model, enc.NHEAD, enc.DIM_FEEDFORWARD, enc.DROPOUT, enc.ACTIVATION, enc.LAYER_CFG)
            for _ in range(enc.NUM_BLOCKS)])
        self.conv_out = post_act_block(d_model, d_model, 3, norm_fn=norm_fn, indice_key=f'{indice_key}_subm', dim=2)
        self.register_buffer('pos_table', pos_embed_table(d_model, self.window_shape, model_cfg.PREPROCESS.POS_TEMPERATURE,
                                                          bool(model_cfg.PREPROCESS.get('NORMALIZE_POS', False))), persistent=False)

    def kept_rows(self, sp: SparseConvTensor):
        """Rows that survive SSTInputLayer.drop_voxel (spt_backbone.py:73-135): in shift 0 a voxel whose in-window rank
        (ascending row index = cell order, SURVEY A-5) reaches its level's max_tokens goes; the survivors are bucketed again
        under shift 1 and dropped by the same rule.  None when everybody stays.  Two host syncs (the survivor counts):
        only configurations that can drop at all come here."""
        ny, nx = sp.spatial_shape
        wb0 = ops.window_bucket(sp.indices, sp.grid, None, sp.batch_size, ny, nx, self.window_shape, False, self.drop_info,
                                keep_only=True)
        idx0 = wb0['keep'].bool().nonzero().squeeze(1)
        ind1 = sp.indices[idx0].contiguous()
        grid1 = ops.index_grid(ind1, sp.batch_size, ny, nx)
        wb1 = ops.window_bucket(ind1, grid1, None, sp.batch_size, ny, nx, self.window_shape, True, self.drop_info, keep_only=True)
        kept = idx0[wb1['keep'].bool().nonzero().squeeze(1)]
        return None if kept.shape[0] == sp.indices.shape[0] else kept

    def encoder_forward(self, sp: SparseConvTensor, residual=False):
        """SSTBlockV1.encoder_forward (spt_backbone.py:314-340) on the ragged layout.  residual: returns
        x + encoder(x) (the sum of forward(), spt_backbone.py:349-351), taken inside the last layer's norm."""
        if self.can_drop:
            kept = self.kept_rows(sp)
            if kept is not None:
                # dropped voxels skip the encoder and come back as zero rows (spt_backbone.py:347-349): the layers run
                # on the compacted survivors, their output is added into the survivors' rows of x
                sub = SparseConvTensor(sp.features[kept], sp.indices[kept], sp.spatial_shape, sp.batch_size)
                enc = self._encoder_forward(sub, residual=False)
                full = torch.zeros_like(sp.features).index_add(0, kept, enc.to(sp.features.dtype))
                return sp.features + full if residual else full
        return self._encoder_forward(sp, residual)

    def _encoder_forward(self, sp: SparseConvTensor, residual=False):
        plan = WindowPlan(sp)
        out = sp.features
        alias = None
        nb = len(self.encoder_blocks)
        if nb == 0:                                # NUM_BLOCKS: 0 -- the reference's x + encoder(x) with encoder = identity
            return out + out if residual else out
        for j, block in enumerate(self.encoder_blocks):
            tap, last = residual and j == 0, residual and j == nb - 1
            if tap and last:                       # a single block: its first layer taps, its last layer adds
                assert len(block.encoder_list) >= 2, 'residual folding needs >= 2 encoder layers'

                out, alias = block.encoder_list[0](out, plan, self.pos_table, self.window_shape, False, passthrough=True)
                for i, layer in enumerate(block.encoder_list[1:], 1):
                    out = layer(out, plan, self.pos_table, self.window_shape, i == 1,
                                post=alias if i == len(block.encoder_list) - 1 else None)
            elif tap:
                out, alias = block(out, plan, self.pos_table, self.window_shape, tap=True)
            else:
                out = block(out, plan, self.pos_table, self.window_shape, post=alias if last else None)
        return out

    def forward(self, sp: SparseConvTensor):
        if self.conv_down is not None:
            sp = self.conv_down(sp)
        x = sp.features
        out = self.encoder_forward(sp, residual=True)          # = x + encoder(x)
        sp = sp.replace_feature(out.to(x.dtype))
        return self.conv_out(sp)


class WCABlock(nn.Module):
    """Temporal window cross-attention stage (SiamWCA.py:272-447): one BasicShiftBlock_WCA (NUM_BLOCKS is
    forced to 1 there, :294-296), residual, subm conv_out."""

    def __init__(self, model_cfg, input_channels, indice_key, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        enc = model_cfg.ENCODER
        d_model = enc.D_MODEL
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        _check_preprocess(model_cfg.PREPROCESS)
        self.window_shape = list(model_cfg.PREPROCESS.WINDOW_SHAPE)
        self.drop_info, self.can_drop = _drop_info(model_cfg.PREPROCESS)
        self.encoder_blocks = nn.ModuleList([
            BasicShiftBlock_WCA(d_model, enc.NHEAD, enc.DIM_FEEDFORWARD, enc.DROPOUT, enc.ACTIVATION, enc.LAYER_CFG)])
        self.conv_out = post_act_block(d_model, d_model, 3, norm_fn=norm_fn, indice_key=f'{indice_key}_subm', dim=2)
        self.register_buffer('pos_table', pos_embed_table(d_model, self.window_shape, model_cfg.PREPROCESS.POS_TEMPERATURE,
                                                          bool(model_cfg.PREPROCESS.get('NORMALIZE_POS', False))), persistent=False)

    def encoder_forward(self, sp: SparseConvTensor, sp_prev: SparseConvTensor, residual=False):
        """WCABlock.encoder_forward (SiamWCA.py:342-396): joint bucketing of the two frames, two cross layers."""
        plan_prv = WindowPlan(sp_prev)
        plan = WindowPlan(sp, other=plan_prv)
        assert sp.spatial_shape == sp_prev.spatial_shape
        x = sp.features
        kept = []
        for shift in (False, True):
            wb = ops.window_bucket(plan.indices, plan.grid, plan_prv.grid, plan.batch, plan.ny, plan.nx,
                                   self.window_shape, shift, self.drop_info, keep_only=True)
            kept.append(wb['keep'].view(-1, 1).to(x.dtype))
            if self.can_drop:
                # per-shift keep sets of BOTH frames (drop_single_shift_ref_to_prv, SiamWCA.py:65-140): dropped queries
                # get no update (bmask above), dropped previous-frame tokens are no keys -- both vanish from the grids
                # this shift's attention reads
                wbp = ops.window_bucket(plan_prv.indices, plan_prv.grid, plan.grid, plan.batch, plan.ny, plan.nx,
                                        self.window_shape, shift, self.drop_info, keep_only=True)
                plan.set_shift_grids(shift, _mask_grid(plan, wb['keep']), _mask_grid(plan_prv, wbp['keep']))
        return self.encoder_blocks[0](x, plan, sp_prev.features, plan_prv, self.pos_table, self.window_shape, kept,
                                      residual=residual)

    def forward(self, sp: SparseConvTensor, sp_prev: SparseConvTensor, dtime=0):
        x = sp.features
        res = self.encoder_forward(sp, sp_prev, residual=True)      # = x + layers(x)
        sp = sp.replace_feature(res.to(x.dtype))
        return self.conv_out(sp)
