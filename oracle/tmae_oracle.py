"""CPU oracle for the T-MAE pre-training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU / numpy restatement of the reference algorithm
(codename1995/T-MAE @ 2024-10-22).  It is the *checker* for the HIP product path
under ``t-mae_amd/`` and the ``cpu_baseline`` leg of ``bench.py``.  Nothing in the
product package may import it (only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg do).

Pinning: the pure-Python reference logic (voxelisation, masking, window
partition / bucketing, cosine attention, encoder layers, decoder, target
assignment) is pinned against the reference itself, imported in the build
container by ``oracle/ref_import.py``; ``oracle/gen_golden.py`` writes the
resulting vectors to ``tests/golden``.  The arithmetic the reference delegates to
third-party packages that are absent from /root/reference (spconv, torch_scatter,
pytorch3d v0.7.1) is restated here from their published definitions -> for those
ops "parity unpinned" (self-checked against dense conv / brute force instead).

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# configuration (mirrors tools/cfgs/once_models/t_mae_ssl.yaml MODEL section)
# --------------------------------------------------------------------------- #

DEFAULT_DROP_INFO = OrderedDict([
    (0, dict(max_tokens=16, drop_range=(0, 16))),
    (1, dict(max_tokens=32, drop_range=(16, 32))),
    (2, dict(max_tokens=64, drop_range=(32, 100000))),
])


def default_model_cfg(num_stages=3):
    """Model hyper-parameters of t_mae_ssl.yaml:46-176 as a plain dict."""
    stages = [
        dict(name='sst_block_x1', stride=1, d_model=128, nhead=8, dff=256, num_blocks=2),
        dict(name='sst_block_x2', stride=2, d_model=256, nhead=8, dff=512, num_blocks=2),
        dict(name='sst_block_x4', stride=2, d_model=256, nhead=8, dff=512, num_blocks=2),
    ][:num_stages]
    fuse = [dict(stride=1, cin=128, cout=128), dict(stride=2, cin=256, cout=128),
            dict(stride=4, cin=256, cout=128)][:num_stages]
    return dict(
        point_cloud_range=[-74.88, -74.88, -5.0, 74.88, 74.88, 3.0],
        voxel_size=[0.32, 0.32, 8.0],
        grid_size=[468, 468, 1],
        vfe_mlps=[64, 128],
        stages=stages, fuse=fuse,
        window_shape=(8, 8, 1), drop_info=DEFAULT_DROP_INFO,
        pos_temperature=1000, tau_min=0.01,
        mask_ratio=0.75, num_prd_points=16, num_gt_points=64,
    )


# --------------------------------------------------------------------------- #
# A1  dynamic voxelisation
# --------------------------------------------------------------------------- #

def in_range_coords(points, pc_range, voxel_size, grid_size):
    """pcdet/utils/common_utils.py:66-76 get_in_range_mask.

    coords = ((xyz - range_min) / voxel_size).to(int64): IEEE fp32 subtract and
    divide, truncation toward zero; keep iff 0 <= c < grid on all three axes.
    points: [N,5] float32 (b,x,y,z,i).  Returns keep[N] bool, coords[N,3] int64 (x,y,z).
    """
    p = np.asarray(points, dtype=np.float32)
    rmin = np.asarray(pc_range[:3], dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    q = (p[:, 1:4] - rmin[None, :]) / vs[None, :]          # float32
    coords = np.trunc(q).astype(np.int64)                   # .to(int64) truncates
    g = np.asarray(grid_size, dtype=np.int64)
    keep = np.all((coords >= 0) & (coords < g[None, :]), axis=1)
    return keep, coords


def voxelize(points, pc_range, voxel_size, grid_size):
    """pcdet/models/backbones_3d/vfe/temporal_dyn_vfe.py:67-72.

    Returns dict(points[N',5], point_coords[N',4] (b,z,y,x), voxel_coords[M,4]
    lexicographically sorted (= torch.unique(dim=0)), inverse[N']).
    """
    p = np.asarray(points, dtype=np.float32)
    keep, c = in_range_coords(p, pc_range, voxel_size, grid_size)
    p, c = p[keep], c[keep]
    b = p[:, 0].astype(np.int64)
    coords4 = np.stack([b, c[:, 2], c[:, 1], c[:, 0]], axis=1)     # [b, z, y, x]
    gx, gy, gz = [int(v) for v in grid_size]
    key = ((coords4[:, 0] * gz + coords4[:, 1]) * gy + coords4[:, 2]) * gx + coords4[:, 3]
    ukey, inverse = np.unique(key, return_inverse=True)              # ascending == lexicographic
    vx = ukey % gx
    vy = (ukey // gx) % gy
    vz = (ukey // (gx * gy)) % gz
    vb = ukey // (gx * gy * gz)
    voxel_coords = np.stack([vb, vz, vy, vx], axis=1).astype(np.int64)
    return dict(keep=keep, points=p, point_coords=coords4, voxel_coords=voxel_coords,
                inverse=inverse.astype(np.int64))


# --------------------------------------------------------------------------- #
# torch_scatter restatements (third-party, parity unpinned)
# --------------------------------------------------------------------------- #

def segment_mean(src, index, num_segments):
    """torch_scatter.scatter(src, index, dim=0, reduce='mean'): sum / max(count, 1).
    Call site temporal_dyn_vfe.py:85."""
    out = torch.zeros((num_segments,) + tuple(src.shape[1:]), dtype=src.dtype)
    out = out.index_add(0, index, src)
    cnt = torch.zeros(num_segments, dtype=src.dtype).index_add(0, index, torch.ones_like(index, dtype=src.dtype))
    return out / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))


def segment_max(src, index, num_segments):
    """torch_scatter.scatter_max(src, index, dim=0)[0].  Call site temporal_dyn_vfe.py:113."""
    idx = index.view(-1, 1).expand_as(src)
    out = torch.zeros((num_segments, src.shape[1]), dtype=src.dtype)
    return out.scatter_reduce(0, idx, src, reduce='amax', include_self=False)


# --------------------------------------------------------------------------- #
# A2  VFE
# --------------------------------------------------------------------------- #

def batch_norm_train(x, weight, bias, eps, stats=None):
    """nn.BatchNorm in training mode (biased batch variance).  x: [N,C] or [B,C,H,W].
    If `stats` is a dict, the (mean, unbiased var, count) used for the running-stat
    update are recorded."""
    return F.batch_norm(x, None, None, weight, bias, True, 0.0, eps)


def vfe_forward(params, prefix, points, cfg):
    """TemporalDynVFE._forward, temporal_dyn_vfe.py:55-125 (TYPE mean, USE_ABSLOTE_XYZ,
    USE_CLUSTER_XYZ, no distance).  params: state_dict tensors under `prefix`
    ('vfe.'): dvfe_mlps.0.{0,3}.weight, dvfe_mlps.0.{1,4}.{weight,bias}."""
    vox = voxelize(points, cfg['point_cloud_range'], cfg['voxel_size'], cfg['grid_size'])
    pts = torch.from_numpy(vox['points'])
    coords = torch.from_numpy(vox['point_coords'])
    inv = torch.from_numpy(vox['inverse'])
    M = vox['voxel_coords'].shape[0]
    vs = torch.tensor(cfg['voxel_size'], dtype=torch.float32)
    rmin = torch.tensor(cfg['point_cloud_range'][:3], dtype=torch.float32)

    mean = segment_mean(pts[:, 1:], inv, M)                          # :85
    f_cluster = pts[:, 1:4] - mean[inv][:, :3]                       # :88-89
    f_center = torch.zeros_like(f_cluster)                           # :91-96
    f_center[:, 0] = pts[:, 1] - ((coords[:, 3] + 0.5) * vs[0] + rmin[0])
    f_center[:, 1] = pts[:, 2] - ((coords[:, 2] + 0.5) * vs[1] + rmin[1])
    f_center[:, 2] = pts[:, 3] - ((coords[:, 1] + 0.5) * vs[2] + rmin[2])
    x = torch.cat([f_center, pts[:, 1:], f_cluster], dim=-1)         # :98-109
    pfeat = x
    # make_fc_layers_GN: Linear(no bias) + BatchNorm1d(eps 1e-5) + ReLU, twice (network_utils.py:25-40)
    m = prefix + 'dvfe_mlps.0.'
    x = x.to(params[m + '0.weight'].dtype)     # float64 parameters = the high-precision run of the same function
    x = F.linear(x, params[m + '0.weight'])
    x = F.relu(batch_norm_train(x, params[m + '1.weight'], params[m + '1.bias'], 1e-5))
    x = F.linear(x, params[m + '3.weight'])
    x = F.relu(batch_norm_train(x, params[m + '4.weight'], params[m + '4.bias'], 1e-5))
    vfeat = segment_max(x, inv, M)                                   # :113
    vox.update(voxel_features=vfeat, point_features=pfeat, voxel_mean=mean)
    return vox


# --------------------------------------------------------------------------- #
# A3  random masking
# --------------------------------------------------------------------------- #

def random_masking_from_noise(noise, mask_ratio):
    """pcdet/utils/common_utils.py:49-63 with the noise injected: keep the
    int(L*(1-ratio)) smallest-noise voxels.  Returns mask[L] float32, 1 = removed."""
    L = noise.shape[0]
    len_keep = int(L * (1 - mask_ratio))
    ids = np.argsort(noise, kind='stable')
    mask = np.ones(L, dtype=np.float32)
    mask[ids[:len_keep]] = 0
    return mask


def mask_voxels(voxel_coords, noise, mask_ratio, batch_size):
    """SiamWCA_MAE.mask_voxels, SiamWCA_MAE.py:166-182.  `noise` [M] is consumed
    per sample in voxel order (voxel rows are grouped by batch index)."""
    masks = []
    start = 0
    for b in range(batch_size):
        L = int((voxel_coords[:, 0] == b).sum())
        masks.append(random_masking_from_noise(noise[start:start + L], mask_ratio))
        start += L
    return np.concatenate(masks) if masks else np.zeros(0, np.float32)


# --------------------------------------------------------------------------- #
# A4/A5  window partition + region batching
# --------------------------------------------------------------------------- #

def stable_ingroup_rank(group):
    """Canonical form of sst_ops ingroup_inds (sst_ops_gpu.cu:14-20): running index of
    each element inside its group, in ascending element order (the reference's atomic
    order is schedule-dependent; SURVEY A-5)."""
    g = np.asarray(group, dtype=np.int64)
    n = g.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    order = np.argsort(g, kind='stable')
    sg = g[order]
    is_start = np.ones(n, dtype=bool)
    is_start[1:] = sg[1:] != sg[:-1]
    start_idx = np.maximum.accumulate(np.where(is_start, np.arange(n), 0))
    out = np.empty(n, np.int64)
    out[order] = np.arange(n) - start_idx
    return out


def window_grid(sparse_shape, window_shape):
    """sst_utils.py:23-27: windows per axis = ceil(grid / win) + 1."""
    return tuple(int(math.ceil(s / w) + 1) for s, w in zip(sparse_shape, window_shape))


def get_window_coors(coors, sparse_shape, window_shape, do_shift):
    """pcdet/models/model_utils/sst_utils.py:6-58.  coors [N,4] (b,z,y,x) int64;
    sparse_shape (X,Y,Z); returns batch_win_inds[N], coors_in_win[N,3] (z,y,x)."""
    wx, wy, wz = window_shape
    sx, sy, sz = sparse_shape
    mx, my, mz = window_grid(sparse_shape, window_shape)
    per_sample = mx * my * mz
    if do_shift:
        shx, shy, shz = wx // 2, wy // 2, wz // 2
    else:
        shx, shy, shz = wx, wy, wz
    if sz == wz:
        shz = 0
    c = np.asarray(coors, dtype=np.int64)
    scx, scy, scz = c[:, 3] + shx, c[:, 2] + shy, c[:, 1] + shz
    wcx, wcy, wcz = scx // wx, scy // wy, scz // wz
    bwi = c[:, 0] * per_sample + wcx * my * mz + wcy * mz + wcz
    ciw = np.stack([scz % wz, scy % wy, scx % wx], axis=-1)
    return bwi, ciw


def drop_levels_from_counts(counts, drop_info):
    """spt_backbone.py:55-60: per-element (level, max_tokens) from its window's count."""
    lvl = -np.ones_like(counts)
    tgt = np.zeros_like(counts)
    for dl, info in drop_info.items():
        lo, hi = info['drop_range']
        m = (counts >= lo) & (counts < hi)
        lvl[m] = dl
        tgt[m] = info['max_tokens']
    return lvl, tgt


def drop_single_shift(bwi, drop_info):
    """SSTInputLayer.drop_single_shift, spt_backbone.py:47-71."""
    inner = stable_ingroup_rank(bwi)
    bincount = np.bincount(bwi)
    lvl, tgt = drop_levels_from_counts(bincount[bwi], drop_info)
    assert (tgt > 0).all() and (lvl >= 0).all()
    return inner < tgt, lvl


def flat2win_inds(bwi, lvl, drop_info):
    """sst_utils.get_flat2win_inds, sst_utils.py:79-107 (+ make_continuous_inds :61-76).
    Returns {level: (flat2win[m_l], flat_pos[m_l])}."""
    out = OrderedDict()
    for dl, info in drop_info.items():
        m = lvl == dl
        if not m.any():
            continue
        w = bwi[m]
        uniq = np.unique(w)
        conti = np.searchsorted(uniq, w)
        inner = stable_ingroup_rank(conti)
        out[dl] = (conti * info['max_tokens'] + inner, np.nonzero(m)[0])
    return out


def sst_input_layer(coords, grid_xyz, cfg):
    """SSTInputLayer.forward, spt_backbone.py:137-184 (SHUFFLE_VOXELS False).
    coords [M,4] (b,z,y,x).  Returns per-shift dicts with batch_win_inds, coors_in_win,
    drop level, flat2win inds; and voxel_keep_inds (all voxels for 8x8 windows)."""
    info = {}
    ws, di = cfg['window_shape'], cfg['drop_info']
    for i in range(2):
        bwi, ciw = get_window_coors(coords, grid_xyz, ws, i == 1)
        info[f'batch_win_inds_shift{i}'] = bwi
        info[f'coors_in_win_shift{i}'] = ciw
    M = coords.shape[0]
    keep_inds = np.arange(M)
    k0, l0 = drop_single_shift(info['batch_win_inds_shift0'], di)      # drop_voxel :73-135
    keep_inds = keep_inds[k0]
    l0 = l0[k0]
    b0 = info['batch_win_inds_shift0'][k0]
    b1 = info['batch_win_inds_shift1'][k0]
    k1, l1 = drop_single_shift(b1, di)
    keep_inds, l0, b0, l1, b1 = keep_inds[k1], l0[k1], b0[k1], l1[k1], b1[k1]
    info['voxel_keep_inds'] = keep_inds
    info['batch_win_inds_shift0'], info['batch_win_inds_shift1'] = b0, b1
    info['voxel_drop_level_shift0'], info['voxel_drop_level_shift1'] = l0, l1
    for i in range(2):
        info[f'coors_in_win_shift{i}'] = info[f'coors_in_win_shift{i}'][keep_inds]
        info[f'flat2win_inds_shift{i}'] = flat2win_inds(info[f'batch_win_inds_shift{i}'],
                                                        info[f'voxel_drop_level_shift{i}'], di)
    return info


def sst_input_layer_temporal(coords, coords_prv, grid_xyz, cfg):
    """SSTInputLayer_Temporal.forward/drop_voxel/drop_single_shift_ref_to_prv,
    SiamWCA.py:65-269: a window is dropped when empty in either frame; the level comes
    from max(count_cur, count_prev); per shift independent keep sets."""
    ws, di = cfg['window_shape'], cfg['drop_info']
    cur, prv = {}, {}
    for i in range(2):
        bwi, ciw = get_window_coors(coords, grid_xyz, ws, i == 1)
        bwi_p, ciw_p = get_window_coors(coords_prv, grid_xyz, ws, i == 1)
        n = max(int(bwi.max()) + 1 if bwi.size else 0, int(bwi_p.max()) + 1 if bwi_p.size else 0)
        bc = np.bincount(bwi, minlength=n)
        bcp = np.bincount(bwi_p, minlength=n)
        no_vox = (bc == 0) | (bcp == 0)
        bmax = np.maximum(bc, bcp)
        lvl, tgt = drop_levels_from_counts(bmax[bwi], di)
        lvl_p, tgt_p = drop_levels_from_counts(bmax[bwi_p], di)
        keep = (stable_ingroup_rank(bwi) < tgt) & ~no_vox[bwi]
        keep_p = (stable_ingroup_rank(bwi_p) < tgt_p) & ~no_vox[bwi_p]
        for d, k, b, l, c in ((cur, keep, bwi, lvl, ciw), (prv, keep_p, bwi_p, lvl_p, ciw_p)):
            d[f'voxel_keep_inds_shift{i}'] = np.nonzero(k)[0]
            d[f'voxel_drop_level_shift{i}'] = l[k]
            d[f'batch_win_inds_shift{i}'] = b[k]
            d[f'coors_in_win_shift{i}'] = c[k]
            d[f'flat2win_inds_shift{i}'] = flat2win_inds(b[k], l[k], di)
    return cur, prv


def pos_embed(coors_in_win, feat_dim, window_shape, pos_temperature, normalize_pos=False):
    """SSTInputLayer.get_pos_embed, spt_backbone.py:186-224 (normalize_pos: NORMALIZE_POS, :202-204).
    coors_in_win [N,3] (z,y,x) -> [N, feat_dim] float32."""
    wx, wy = window_shape[:2]
    c = torch.as_tensor(coors_in_win)
    y = c[:, 1] - wy / 2
    x = c[:, 2] - wx / 2
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
    return torch.cat([ex, ey], dim=-1).to(torch.float32)


def flat2window(feat, f2w, drop_info):
    """sst_utils.flat2window, sst_utils.py:118-152: scatter rows into zero-padded
    [num_windows, max_tokens, C] per level."""
    out = OrderedDict()
    for dl, (inds, pos) in f2w.items():
        T = drop_info[dl]['max_tokens']
        nwin = int(inds.max()) // T + 1
        buf = torch.zeros((nwin * T, feat.shape[-1]), dtype=feat.dtype)
        buf = buf.index_copy(0, torch.from_numpy(inds), feat[torch.from_numpy(pos)])
        out[dl] = buf.reshape(nwin, T, feat.shape[-1])
    return out


def window2flat(feat3d, f2w, num_all):
    """sst_utils.window2flat, sst_utils.py:162-187."""
    C = next(iter(feat3d.values())).shape[-1]
    dtype = next(iter(feat3d.values())).dtype
    out = torch.zeros((num_all, C), dtype=dtype)
    for dl, f in feat3d.items():
        inds, pos = f2w[dl]
        out = out.index_copy(0, torch.from_numpy(pos), f.reshape(-1, C)[torch.from_numpy(inds)])
    return out


def key_padding_mask(f2w, drop_info, num_all):
    """SSTInputLayer.get_key_padding_mask, spt_backbone.py:233-243: True = padded slot."""
    ones = torch.ones((num_all, 1))
    win = flat2window(ones, f2w, drop_info)
    return OrderedDict((k, (v == 0).squeeze(2)) for k, v in win.items())


# --------------------------------------------------------------------------- #
# A7  cosine multi-head attention (padded, as the reference computes it)
# --------------------------------------------------------------------------- #

def cosine_mha(q_in, k_in, v_in, kpm, p, prefix, nhead, tau_min):
    """cosine_msa.py:178-438 (+ _scaled_cosine_attention :114-176).  q_in/k_in/v_in:
    [nW, T, E] (batch-first here; the reference permutes to (T, nW, E)); kpm [nW, T]
    True = padded key.  Three chunks of the packed in-proj weight (:57-62); per-head
    L2-normalised q,k (F.normalize eps 1e-12); logits / clamp(tau, tau_min) (one tau, or one per head); -inf on
    padded keys; softmax; PV; out-proj."""
    nW, Tq, E = q_in.shape
    Tk = k_in.shape[1]
    w, b = p[prefix + 'in_proj_weight'], p[prefix + 'in_proj_bias']
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(k_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(v_in, w[2 * E:], b[2 * E:])
    Dh = E // nhead
    q = q.view(nW, Tq, nhead, Dh).transpose(1, 2)          # [nW,H,T,Dh]
    k = k.view(nW, Tk, nhead, Dh).transpose(1, 2)
    v = v.view(nW, Tk, nhead, Dh).transpose(1, 2)
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    attn = q @ k.transpose(-2, -1)
    tau = p[prefix + 'tau']                      # (1,1,1) shared, or (1, nhead, 1, 1) with non_shared_tau (cosine_msa.py:155-161, :453-456)
    attn = attn / (tau.view(1, nhead, 1, 1) if tau.dim() == 4 else tau.view(1, 1, 1, 1)).clamp(min=tau_min)
    attn = attn.masked_fill(kpm.view(nW, 1, 1, Tk), float('-inf'))
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(nW, Tq, E)
    return F.linear(o, p[prefix + 'out_proj.weight'], p[prefix + 'out_proj.bias'])


def window_self_attention(x, pos, f2w, kpm, p, prefix, nhead, cfg):
    """WindowAttention.forward, sst_basic_block.py:22-54: q = k = x + pos, v = x."""
    di = cfg['drop_info']
    x3 = flat2window(x, f2w, di)
    p3 = flat2window(pos, f2w, di)
    out = OrderedDict()
    for dl in x3:
        qk = x3[dl] + p3[dl]
        out[dl] = cosine_mha(qk, qk, x3[dl], kpm[dl], p, prefix + 'self_attn.', nhead, cfg['tau_min'])
    return window2flat(out, f2w, x.shape[0])


def layer_tail(src, p, prefix):
    """Post-norm tail shared by both EncoderLayer flavours
    (sst_basic_block.py:79-84, wca_block.py:98-102): LN, FFN(GELU), LN."""
    d = src.shape[-1]
    src = F.layer_norm(src, (d,), p[prefix + 'norm1.weight'], p[prefix + 'norm1.bias'], 1e-5)
    h = F.gelu(F.linear(src, p[prefix + 'linear1.weight'], p[prefix + 'linear1.bias']))
    src = src + F.linear(h, p[prefix + 'linear2.weight'], p[prefix + 'linear2.bias'])
    return F.layer_norm(src, (d,), p[prefix + 'norm2.weight'], p[prefix + 'norm2.bias'], 1e-5)


def encoder_layer(src, pos, f2w, kpm, p, prefix, nhead, cfg):
    """EncoderLayer.forward, sst_basic_block.py:77-84."""
    src = src + window_self_attention(src, pos, f2w, kpm, p, prefix + 'win_attn.', nhead, cfg)
    return layer_tail(src, p, prefix)


def sst_encoder(x, coords, grid_xyz, p, prefix, stage, cfg, capture=None):
    """SSTBlockV1.encoder_forward, spt_backbone.py:314-340: input layer, then
    NUM_BLOCKS x BasicShiftBlockV2 (layer 0 on shift 0, layer 1 on shift 1,
    sst_basic_block.py:100-114)."""
    info = sst_input_layer(coords, grid_xyz, cfg)
    # Voxels whose in-window rank reaches their level's max_tokens are dropped from the encoder (spt_backbone.py:47-135)
    # and come back as zero rows (:347-349: voxel_features_unshuffle[voxel_shuffle_inds] = ...), i.e. they keep only the
    # block's residual.  With the shipped DROP_INFO (8 x 8 windows, top level 64 tokens) nobody is dropped (SURVEY A-6).
    keep = torch.from_numpy(info['voxel_keep_inds'])
    xk = x if keep.numel() == x.shape[0] else x[keep]
    d = x.shape[1]
    per_shift = []
    for i in range(2):
        f2w = info[f'flat2win_inds_shift{i}']
        pos = pos_embed(info[f'coors_in_win_shift{i}'], d, cfg['window_shape'], cfg['pos_temperature'], cfg.get('normalize_pos', False))
        per_shift.append((pos, f2w, key_padding_mask(f2w, cfg['drop_info'], xk.shape[0])))
    out = xk
    for blk in range(stage['num_blocks']):
        for i in range(2):
            pos, f2w, kpm = per_shift[i]
            out = encoder_layer(out, pos, f2w, kpm, p,
                                f'{prefix}encoder_blocks.{blk}.encoder_list.{i}.', stage['nhead'], cfg)
    if capture is not None:
        capture['input_layer'] = info
    if keep.numel() != x.shape[0]:
        out = torch.zeros_like(x).index_copy(0, keep, out)
    return out


# --------------------------------------------------------------------------- #
# A9  2-D sparse convolution (spconv restatement; parity unpinned)
# --------------------------------------------------------------------------- #

def _keys(ind, shape):
    Y, X = shape
    return (ind[:, 0] * Y + ind[:, 1]) * X + ind[:, 2]


def sparse_rulebook(indices, spatial_shape, kind):
    """Rulebook of a 3x3 2-D sparse conv.

    kind 'subm': outputs = inputs; tap (ky,kx) reads input (oy-1+ky, ox-1+kx).
    kind 'down': SparseConv2d(k3,s2,p1): out shape floor((in+2-3)/2)+1; tap (ky,kx)
    connects input (iy,ix) to output ((iy+1-ky)/2, (ix+1-kx)/2) when exact & in range.
    indices [M,3] (b,y,x) int64.  Returns out_indices [Mo,3] (lexicographic),
    out_shape, pairs: list over the 9 taps of (in_idx, out_idx) int64 arrays.
    """
    ind = np.asarray(indices, dtype=np.int64)
    Y, X = int(spatial_shape[0]), int(spatial_shape[1])
    if kind == 'subm':
        out_shape = (Y, X)
        out_ind = ind
        okey = _keys(out_ind, out_shape)
        order = np.argsort(okey, kind='stable')
        skey = okey[order]
        pairs = []
        for ky in range(3):
            for kx in range(3):
                ny, nx = ind[:, 1] - 1 + ky, ind[:, 2] - 1 + kx       # input read by output o
                valid = (ny >= 0) & (ny < Y) & (nx >= 0) & (nx < X)
                nkey = (ind[:, 0] * Y + ny) * X + nx
                pos = np.searchsorted(skey, nkey)
                pos_c = np.minimum(pos, len(skey) - 1)
                hit = valid & (skey[pos_c] == nkey)
                o = np.nonzero(hit)[0]
                pairs.append((order[pos_c[hit]], o))
        return out_ind, out_shape, pairs
    assert kind == 'down'
    oY, oX = (Y + 2 - 3) // 2 + 1, (X + 2 - 3) // 2 + 1
    out_shape = (oY, oX)
    cand = []
    for ky in range(3):
        for kx in range(3):
            ty, tx = ind[:, 1] + 1 - ky, ind[:, 2] + 1 - kx
            ok = (ty % 2 == 0) & (tx % 2 == 0)
            oy, ox = ty // 2, tx // 2
            ok &= (oy >= 0) & (oy < oY) & (ox >= 0) & (ox < oX)
            cand.append((ok, oy, ox))
    allkeys = np.concatenate([(ind[ok, 0] * oY + oy[ok]) * oX + ox[ok] for ok, oy, ox in cand])
    ukey = np.unique(allkeys)
    out_ind = np.stack([ukey // (oY * oX), (ukey // oX) % oY, ukey % oX], axis=1)
    pairs = []
    for ok, oy, ox in cand:
        i = np.nonzero(ok)[0]
        k = (ind[i, 0] * oY + oy[i]) * oX + ox[i]
        pairs.append((i, np.searchsorted(ukey, k)))
    return out_ind, out_shape, pairs


def sparse_conv(feat, weight, pairs, num_out):
    """out[o] = sum_taps W[:,ky,kx,:] @ in[i]; weight layout spconv-2 [Cout,kH,kW,Cin]
    (detector3d_template.py:373-383)."""
    out = torch.zeros((num_out, weight.shape[0]), dtype=feat.dtype)
    t = 0
    for ky in range(3):
        for kx in range(3):
            i, o = pairs[t]
            t += 1
            if len(i) == 0:
                continue
            out = out.index_add(0, torch.from_numpy(o), feat[torch.from_numpy(i)] @ weight[:, ky, kx, :].t())
    return out


def post_act_block(feat, indices, spatial_shape, p, prefix, kind):
    """spconv_utils.post_act_block, spconv_utils.py:37-56: conv + BatchNorm1d(eps 1e-3)
    over all active rows + ReLU."""
    out_ind, out_shape, pairs = sparse_rulebook(indices, spatial_shape, kind)
    y = sparse_conv(feat, p[prefix + '0.weight'], pairs, out_ind.shape[0])
    y = F.relu(batch_norm_train(y, p[prefix + '1.weight'], p[prefix + '1.bias'], 1e-3))
    return y, out_ind, out_shape


def sst_block(feat, indices, spatial_shape, p, prefix, stage, cfg, capture=None):
    """SSTBlockV1.forward, spt_backbone.py:342-353."""
    if stage['stride'] > 1:
        feat, indices, spatial_shape = post_act_block(feat, indices, spatial_shape, p, prefix + 'conv_down.', 'down')
    coords = np.concatenate([indices[:, :1], np.zeros_like(indices[:, :1]), indices[:, 1:]], axis=1)
    grid_xyz = (spatial_shape[1], spatial_shape[0], 1)
    enc = sst_encoder(feat, coords, grid_xyz, p, prefix, stage, cfg, capture)
    feat = feat + enc
    feat, _, _ = post_act_block(feat, indices, spatial_shape, p, prefix + 'conv_out.', 'subm')
    return feat, indices, spatial_shape


# --------------------------------------------------------------------------- #
# A10  window cross-attention block
# --------------------------------------------------------------------------- #

def wca_encoder_layer(src, src_prv, cur, prv, shift, p, prefix, nhead, cfg):
    """wca_block.EncoderLayer.forward, wca_block.py:90-103 + WindowCrossAttention :26-67:
    q = cur[keep]+pos, k = prev[keep]+pos_prev, v = prev[keep]; key mask from prev;
    only kept rows receive the attention residual; LN/FFN/LN on all rows."""
    di = cfg['drop_info']
    d = src.shape[1]
    keep = torch.from_numpy(cur[f'voxel_keep_inds_shift{shift}'])
    keep_p = torch.from_numpy(prv[f'voxel_keep_inds_shift{shift}'])
    f2w, f2w_p = cur[f'flat2win_inds_shift{shift}'], prv[f'flat2win_inds_shift{shift}']
    if keep.numel() > 0:
        sel, sel_p = src[keep], src_prv[keep_p]
        pos = pos_embed(cur[f'coors_in_win_shift{shift}'], d, cfg['window_shape'], cfg['pos_temperature'], cfg.get('normalize_pos', False))
        pos_p = pos_embed(prv[f'coors_in_win_shift{shift}'], d, cfg['window_shape'], cfg['pos_temperature'], cfg.get('normalize_pos', False))
        q3, qp3 = flat2window(sel, f2w, di), flat2window(pos, f2w, di)
        k3, kp3 = flat2window(sel_p, f2w_p, di), flat2window(pos_p, f2w_p, di)
        kpm = key_padding_mask(f2w_p, di, sel_p.shape[0])
        out = OrderedDict()
        for dl in q3:
            out[dl] = cosine_mha(q3[dl] + qp3[dl], k3[dl] + kp3[dl], k3[dl], kpm[dl], p,
                                 prefix + 'win_attn.cross_attn.', nhead, cfg['tau_min'])
        attn = window2flat(out, f2w, sel.shape[0])
        src = src.index_add(0, keep, attn)
    return layer_tail(src, p, prefix)


def wca_block(feat, indices, feat_prv, indices_prv, spatial_shape, p, prefix, stage, cfg, capture=None):
    """WCABlock.forward/encoder_forward, SiamWCA.py:342-396,431-447 (NUM_BLOCKS forced to
    1, :294-296): joint bucketing, 2 cross layers, residual, subm conv_out."""
    mk = lambda ind: np.concatenate([ind[:, :1], np.zeros_like(ind[:, :1]), ind[:, 1:]], axis=1)
    grid_xyz = (spatial_shape[1], spatial_shape[0], 1)
    cur, prv = sst_input_layer_temporal(mk(indices), mk(indices_prv), grid_xyz, cfg)
    res = feat
    for shift in range(2):
        res = wca_encoder_layer(res, feat_prv, cur, prv, shift, p,
                                f'{prefix}encoder_blocks.0.encoder_list.{shift}.', stage['nhead'], cfg)
    out = feat + res
    out, _, _ = post_act_block(out, indices, spatial_shape, p, prefix + 'conv_out.', 'subm')
    if capture is not None:
        capture['temporal_cur'], capture['temporal_prv'] = cur, prv
    return out


# --------------------------------------------------------------------------- #
# A11-A13  dense decoder, targets, Chamfer
# --------------------------------------------------------------------------- #

def to_dense(feat, indices, spatial_shape, batch_size):
    """SparseConvTensor.dense(): [B, C, Y, X], zeros at inactive sites."""
    Y, X = spatial_shape
    out = torch.zeros((batch_size * Y * X, feat.shape[1]), dtype=feat.dtype)
    lin = torch.from_numpy(_keys(np.asarray(indices), (Y, X)))
    out = out.index_copy(0, lin, feat)
    return out.view(batch_size, Y, X, -1).permute(0, 3, 1, 2)


def dense_decoder(ms_feats, batch_size, p, prefix, cfg, deblocks='decoder_deblocks', conv_out='decoder_conv_out'):
    """SiamWCA_MAE.dense_conv, SiamWCA_MAE.py:231-253 with modules :79-115:
    ConvTranspose2d(k=s) + BN2d(eps 1e-3) + ReLU per scale, cat, Conv3x3 + BN + ReLU."""
    ups = []
    for i, (feat, ind, shape) in enumerate(ms_feats):
        d = to_dense(feat, ind, shape, batch_size)
        s = cfg['fuse'][i]['stride']
        y = F.conv_transpose2d(d, p[f'{prefix}{deblocks}.{i}.0.weight'], stride=s)
        y = F.relu(batch_norm_train(y, p[f'{prefix}{deblocks}.{i}.1.weight'],
                                    p[f'{prefix}{deblocks}.{i}.1.bias'], 1e-3))
        ups.append(y)
    y = F.conv2d(torch.cat(ups, dim=1), p[prefix + conv_out + '.0.weight'], padding=1)
    return F.relu(batch_norm_train(y, p[prefix + conv_out + '.1.weight'],
                                   p[prefix + conv_out + '.1.bias'], 1e-3))


def group_inner_inds(inverse, num_groups, K):
    """Canonical form of sst_ops group_inner_inds (sst_ops_gpu.cu:22-39): first K point
    indices of every voxel in point order, rows with cnt < K filled cyclically
    g[m,i] = g[m, i % cnt]."""
    inv = np.asarray(inverse, dtype=np.int64)
    rank = stable_ingroup_rank(inv)
    cnt = np.bincount(inv, minlength=num_groups)
    table = -np.ones((num_groups, K), dtype=np.int64)
    m = rank < K
    table[inv[m], rank[m]] = np.nonzero(m)[0]
    cols = np.arange(K)[None, :]
    c = np.maximum(cnt, 1)[:, None]
    src = np.where(cols < cnt[:, None], cols, cols % c)
    table = np.take_along_axis(table, src, axis=1)
    return table


def chamfer_distance(pred, gt, weights):
    """pytorch3d v0.7.1 loss.chamfer_distance defaults (batch_reduction mean,
    point_reduction mean, norm 2) with per-cloud weights: squared-L2 NN distance each
    way, mean over the cloud's own points, times weight, summed, / weights.sum().
    pred [M,P,3], gt [M,G,3], weights [M]."""
    d = ((pred[:, :, None, :] - gt[:, None, :, :]) ** 2).sum(-1)     # [M,P,G]
    cx = d.min(dim=2).values.mean(dim=1)
    cy = d.min(dim=1).values.mean(dim=1)
    wsum = weights.sum()
    if float(wsum) == 0.0:
        return (cx.sum() + cy.sum()) * 0.0
    return ((cx * weights).sum() + (cy * weights).sum()) / wsum


def voxel_centers(voxel_coords, voxel_size, pc_range):
    """common_utils.get_voxel_centers, common_utils.py:130-145 (downsample 1, dim 3).
    voxel_coords [M,3] (z,y,x)."""
    c = torch.from_numpy(np.asarray(voxel_coords)[:, ::-1].copy()).float()   # (z,y,x) -> (x,y,z)
    vs = torch.tensor(voxel_size[:3]).float()
    r = torch.tensor(pc_range[:3]).float()
    return (c + 0.5) * vs + r


# --------------------------------------------------------------------------- #
# full step
# --------------------------------------------------------------------------- #

def forward_loss(params, points, points_prev, noise, batch_size, cfg, capture=None):
    """TMAE.forward + get_training_loss (t_mae.py:11-34): TemporalDynVFE on both frames,
    SiamWCA_MAE.forward (SiamWCA_MAE.py:255-322), Chamfer loss (:154-164).
    params: state_dict (reference key names, fp32 tensors; may require grad).
    noise: [M_cur] masking noise in voxel order.  Returns scalar loss tensor."""
    cap = capture if capture is not None else {}
    cur = vfe_forward(params, 'vfe.', points, cfg)
    prv = vfe_forward(params, 'vfe.', points_prev, cfg)
    cap['vfe_cur'], cap['vfe_prv'] = cur, prv
    bp = 'backbone_3d.'
    gx, gy, _ = cfg['grid_size']
    shape0 = (gy, gx)

    def encode(feat, vcoords, tag):
        ind = np.asarray(vcoords)[:, [0, 2, 3]]
        shape = shape0
        outs = []
        for si, stage in enumerate(cfg['stages']):
            c = {}
            feat, ind, shape = sst_block(feat, ind, shape, params, f'{bp}sst_blocks.{si}.', stage, cfg, c)
            cap[f'{tag}_stage{si}'] = dict(indices=ind, shape=shape, features=feat, **c)
            outs.append((feat, ind, shape))
        return outs

    ms_prev = encode(prv['voxel_features'], prv['voxel_coords'], 'prev')
    mask = mask_voxels(cur['voxel_coords'], noise, cfg['mask_ratio'], batch_size)
    vis = np.nonzero(mask == 0)[0]
    cap['mask'] = mask
    ms_cur = encode(cur['voxel_features'][torch.from_numpy(vis)], cur['voxel_coords'][vis], 'cur')
    ms = []
    for si, stage in enumerate(cfg['stages']):
        f, ind, shape = ms_cur[si]
        fp, indp, _ = ms_prev[si]
        c = {}
        f = wca_block(f, ind, fp, indp, shape, params, f'{bp}wca_blocks.{si}.', stage, cfg, c)
        cap[f'wca_stage{si}'] = dict(features=f, **c)
        ms.append((f, ind, shape))
    spatial = dense_decoder(ms, batch_size, params, bp, cfg)
    cap['spatial_features'] = spatial
    vc = cur['voxel_coords']
    vfeat = spatial.permute(0, 2, 3, 1)[torch.from_numpy(vc[:, 0]), torch.from_numpy(vc[:, 2]),
                                        torch.from_numpy(vc[:, 3])]
    M = vc.shape[0]
    table = group_inner_inds(cur['inverse'], M, cfg['num_gt_points'])
    gt = torch.from_numpy(cur['points'][:, 1:4])[torch.from_numpy(table)]
    gt = gt - voxel_centers(vc[:, 1:], cfg['voxel_size'], cfg['point_cloud_range']).unsqueeze(1)
    pred = F.linear(vfeat, params[bp + 'decoder_pred.weight'], params[bp + 'decoder_pred.bias']).view(M, -1, 3)
    w = torch.from_numpy(mask)
    cap.update(pred_points=pred, gt_points=gt, group_inds=table)
    # SiamWCA_MAE.py:162 casts the prediction to fp32; float64 parameters (the high-precision run) stay float64
    return chamfer_distance(pred if pred.dtype == torch.float64 else pred.float(), gt, w)


# --------------------------------------------------------------------------- #
# parameter initialisation with the reference's module defaults (for tests/bench)
# --------------------------------------------------------------------------- #

def init_params(cfg, seed=0, num_point_features=4, tau=None, pred_scale=1.0):
    """Random-init state_dict with the reference's names and shapes.  `tau` overrides every
    attention temperature; `pred_scale` scales decoder_pred (a small head keeps the Chamfer loss of an untrained
    model well-conditioned: with the default scale the loss moves by 6e-3 between 1 and 8 CPU threads)."""
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()

    def lin(name, cout, cin, bias=True, std=None):
        bound = 1.0 / math.sqrt(cin)
        P[name + '.weight'] = (torch.rand(cout, cin, generator=g) * 2 - 1) * bound
        if bias:
            P[name + '.bias'] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def norm(name, c):
        P[name + '.weight'] = 1.0 + 0.1 * torch.randn(c, generator=g)
        P[name + '.bias'] = 0.1 * torch.randn(c, generator=g)

    cin = num_point_features + 6
    lin('vfe.dvfe_mlps.0.0', cfg['vfe_mlps'][0], cin, bias=False)
    norm('vfe.dvfe_mlps.0.1', cfg['vfe_mlps'][0])
    lin('vfe.dvfe_mlps.0.3', cfg['vfe_mlps'][1], cfg['vfe_mlps'][0], bias=False)
    norm('vfe.dvfe_mlps.0.4', cfg['vfe_mlps'][1])

    def enc_layer(pre, d, dff, attn_name):
        a = f'{pre}win_attn.{attn_name}.'
        P[a + 'in_proj_weight'] = (torch.rand(3 * d, d, generator=g) * 2 - 1) * math.sqrt(6.0 / (4 * d))
        P[a + 'in_proj_bias'] = 0.02 * torch.randn(3 * d, generator=g)
        # (1,1,1) shared; cfg['non_shared_tau']: one per head (cosine_msa.py:453-456), here drawn away from 1 so that heads differ
        P[a + 'tau'] = (0.3 + 1.7 * torch.rand(1, cfg['stages'][0]['nhead'], 1, 1, generator=g)) if cfg.get('non_shared_tau', False) else torch.ones(1, 1, 1)
        lin(a + 'out_proj', d, d)
        lin(pre + 'linear1', dff, d)
        lin(pre + 'linear2', d, dff)
        norm(pre + 'norm1', d)
        norm(pre + 'norm2', d)

    def spconv_w(name, cout, cin):
        P[name] = torch.randn(cout, 3, 3, cin, generator=g) * math.sqrt(2.0 / (9 * cin))

    c_prev = cfg['vfe_mlps'][-1]
    for si, st in enumerate(cfg['stages']):
        pre = f'backbone_3d.sst_blocks.{si}.'
        d = st['d_model']
        if st['stride'] > 1:
            spconv_w(pre + 'conv_down.0.weight', d, c_prev)
            norm(pre + 'conv_down.1', d)
        for blk in range(st['num_blocks']):
            for i in range(2):
                enc_layer(f'{pre}encoder_blocks.{blk}.encoder_list.{i}.', d, st['dff'], 'self_attn')
        spconv_w(pre + 'conv_out.0.weight', d, d)
        norm(pre + 'conv_out.1', d)
        c_prev = d
    for si, st in enumerate(cfg['stages']):
        pre = f'backbone_3d.wca_blocks.{si}.'
        d = st['d_model']
        for i in range(2):
            enc_layer(f'{pre}encoder_blocks.0.encoder_list.{i}.', d, st['dff'], 'cross_attn')
        spconv_w(pre + 'conv_out.0.weight', d, d)
        norm(pre + 'conv_out.1', d)
    ctot = 0
    for i, fz in enumerate(cfg['fuse']):
        s = fz['stride']
        P[f'backbone_3d.decoder_deblocks.{i}.0.weight'] = torch.randn(fz['cin'], fz['cout'], s, s, generator=g) * math.sqrt(1.0 / fz['cin'])
        norm(f'backbone_3d.decoder_deblocks.{i}.1', fz['cout'])
        ctot += fz['cout']
    cmid = ctot // len(cfg['fuse'])
    P['backbone_3d.decoder_conv_out.0.weight'] = torch.randn(cmid, ctot, 3, 3, generator=g) * math.sqrt(2.0 / (9 * ctot))
    norm('backbone_3d.decoder_conv_out.1', cmid)
    lin('backbone_3d.decoder_pred', cfg['num_prd_points'] * 3, cmid)
    P['backbone_3d.decoder_pred.weight'] *= pred_scale
    P['backbone_3d.decoder_pred.bias'] *= pred_scale
    if tau is not None and not cfg.get('non_shared_tau', False):
        for n_, t_ in P.items():
            if n_.endswith('tau'):
                t_.fill_(tau)
    return P


# --------------------------------------------------------------------------- #
# synthetic ONCE-shape scans
# --------------------------------------------------------------------------- #

def synth_frame_pair(n_points, batch_size, seed, shift=(0.5, 0.1)):
    """Deterministic synthetic scans: r = exp(U(ln2, ln105)), theta = U(0,2pi),
    z = N(-1.7, 0.3) + 3 U^4, intensity U(0,1); crop |x|,|y| <= 74.88; previous frame =
    current translated by `shift`, re-cropped.  Rows [b,x,y,z,i] float32."""
    cur, prv = [], []
    for b in range(batch_size):
        rng = np.random.default_rng(seed * 64 + b)
        r = np.exp(rng.uniform(np.log(2.0), np.log(105.0), n_points))
        th = rng.uniform(0, 2 * np.pi, n_points)
        x, y = r * np.cos(th), r * np.sin(th)
        z = rng.normal(-1.7, 0.3, n_points) + 3 * rng.uniform(0, 1, n_points) ** 4
        it = rng.uniform(0, 1, n_points)
        pts = np.stack([np.full(n_points, b), x, y, z, it], axis=1).astype(np.float32)

        def crop(p):
            m = (np.abs(p[:, 1]) <= 74.88) & (np.abs(p[:, 2]) <= 74.88)
            return p[m]
        cur.append(crop(pts))
        q = pts.copy()
        q[:, 1] += np.float32(shift[0])
        q[:, 2] += np.float32(shift[1])
        prv.append(crop(q))
    return np.concatenate(cur), np.concatenate(prv)
