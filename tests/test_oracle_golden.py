"""CPU: the oracle (oracle/tmae_oracle.py) against the golden vectors captured from the reference
(tests/golden, written by oracle/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import torch

from conftest import golden, fl


def test_f1_f2_voxelize_and_vfe(oracle):
    g = golden('F1_F2_voxelize_vfe')
    cfg = oracle.default_model_cfg(3)
    params = {k.replace('__', '.'): torch.from_numpy(g[k]) for k in g.files if k.startswith('vfe__')}
    o = oracle.vfe_forward(params, 'vfe.', g['points'], cfg)
    assert np.array_equal(o['keep'], g['keep'])
    assert np.array_equal(o['points'], g['points_kept'])
    assert np.array_equal(o['point_coords'], g['point_coords'])
    assert np.array_equal(o['voxel_coords'], g['voxel_coords'])
    assert np.array_equal(o['inverse'], g['inverse'])
    np.testing.assert_allclose(o['voxel_features'].numpy(), g['voxel_features'], atol=1e-5)


def test_voxelize_edge_cases(oracle):
    cfg = oracle.default_model_cfg(3)
    pts = np.array([[0, 74.88, 0, -1, .5], [0, -74.88, 3, -1, .5], [0, -74.98, 3, -1, .5], [0, -75.21, 3, -1, .5],
                    [0, 1, 2, -12.9, .5], [0, 1, 2, -13.1, .5], [0, 1, 2, 3.0, .5]], dtype=np.float32)
    keep, _ = oracle.in_range_coords(pts, cfg['point_cloud_range'], cfg['voxel_size'], cfg['grid_size'])
    assert keep.tolist() == [False, True, True, False, True, False, False]          # SURVEY A-1
    empty = oracle.voxelize(np.zeros((0, 5), np.float32), cfg['point_cloud_range'], cfg['voxel_size'], cfg['grid_size'])
    assert empty['voxel_coords'].shape == (0, 4) and empty['inverse'].shape == (0,)


def test_f3_mask(oracle):
    g = golden('F3_mask')
    m = oracle.mask_voxels(g['voxel_coords'], g['noise'], float(g['mask_ratio']), 2)
    assert np.array_equal(m, g['mask'])


def test_f4_window_partition(oracle):
    g = golden('F4_window_partition')
    for grid in (468, 234, 117):
        for s in (0, 1):
            bwi, ciw = oracle.get_window_coors(g[f'coords_{grid}'], (grid, grid, 1), (8, 8, 1), s == 1)
            assert np.array_equal(bwi, g[f'bwi_{grid}_s{s}'])
            assert np.array_equal(ciw, g[f'ciw_{grid}_s{s}'])
    assert oracle.window_grid((468, 468, 1), (8, 8, 1)) == (60, 60, 2)
    assert oracle.window_grid((234, 234, 1), (8, 8, 1)) == (31, 31, 2)
    assert oracle.window_grid((117, 117, 1), (8, 8, 1)) == (16, 16, 2)


def test_f5_bucketing(oracle):
    g = golden('F5_bucketing')
    cfg = oracle.default_model_cfg(3)
    info = oracle.sst_input_layer(g['A_coords'], (468, 468, 1), cfg)
    for s in (0, 1):
        assert np.array_equal(info[f'batch_win_inds_shift{s}'], g[f'A_bwi_s{s}'])
        assert np.array_equal(info[f'voxel_drop_level_shift{s}'], g[f'A_lvl_s{s}'])
        kpm = oracle.key_padding_mask(info[f'flat2win_inds_shift{s}'], cfg['drop_info'], len(g['A_coords']))
        for dl in (0, 1, 2):
            assert np.array_equal(info[f'flat2win_inds_shift{s}'][dl][0], g[f'A_f2w_s{s}_l{dl}'])
            assert np.array_equal(info[f'flat2win_inds_shift{s}'][dl][1], g[f'A_pos_s{s}_l{dl}'])
            assert np.array_equal(kpm[dl].numpy(), g[f'A_kpm_s{s}_l{dl}'])
    cur, prv = oracle.sst_input_layer_temporal(g['T_coords_cur'], g['T_coords_prv'], (468, 468, 1), cfg)
    for tag, d in (('cur', cur), ('prv', prv)):
        for s in (0, 1):
            for key in ('voxel_keep_inds', 'voxel_drop_level', 'batch_win_inds', 'coors_in_win'):
                assert np.array_equal(d[f'{key}_shift{s}'], g[f'T_{tag}_{key}_s{s}'])
            for dl, (inds, pos) in d[f'flat2win_inds_shift{s}'].items():
                assert np.array_equal(inds, g[f'T_{tag}_f2w_s{s}_l{dl}'])
                assert np.array_equal(pos, g[f'T_{tag}_pos_s{s}_l{dl}'])


def test_f5_hand_example(oracle):
    """Worked example of SiamWCA.py:692-706: drop levels for 1 / 2 / 4 tokens, level from max(cur, prev)."""
    g = golden('F5_bucketing')
    di = {0: dict(max_tokens=1, drop_range=(0, 2)), 1: dict(max_tokens=2, drop_range=(2, 4)),
          2: dict(max_tokens=4, drop_range=(4, 100000))}
    cur, prv = g['hand_cur'], g['hand_prv']
    n = max(cur.max(), prv.max()) + 1
    bmax = np.maximum(np.bincount(cur, minlength=n), np.bincount(prv, minlength=n))
    lvl, _ = oracle.drop_levels_from_counts(bmax[cur], di)
    assert lvl.tolist() == g['hand_target'].tolist()


def test_f6_pos_embed(oracle):
    g = golden('F6_pos_embed')
    for d in (128, 256):
        pe = oracle.pos_embed(g['coors_in_win'], d, (8, 8, 1), 1000).numpy()
        np.testing.assert_allclose(pe, g[f'pos_{d}'], atol=1e-6)


def test_f14_pos_embed_normalised(oracle):
    """NORMALIZE_POS: True (spt_backbone.py:202-204) -- the reference's values (oracle/gen_golden_options.py) vs the oracle."""
    g = golden('F14_options')
    for d in (128, 256):
        pe = oracle.pos_embed(g['coors_in_win'], d, (8, 8, 1), float(g['pos_temperature']), normalize_pos=True).numpy()
        np.testing.assert_allclose(pe, g[f'pos_norm_{d}'], atol=1e-6)


def test_f7_attention(oracle):
    g = golden('F7_attention')
    for case in range(4):
        pre = f'c{case}_'
        E, H, T, nW, cross = [int(v) for v in g[pre + 'meta']]
        p = {'a.' + k[len(pre) + 2:].replace('__', '.'): torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + 'w_')}
        q = torch.from_numpy(g[pre + 'q']).transpose(0, 1).clone().requires_grad_(True)
        v = torch.from_numpy(g[pre + 'v']).transpose(0, 1).clone().requires_grad_(True)
        k = torch.from_numpy(g[pre + 'k']).transpose(0, 1).clone().requires_grad_(True) if cross else q
        kpm = torch.from_numpy(g[pre + 'kpm'])
        qvalid = (torch.arange(T)[None, :] < torch.from_numpy(g[pre + 'qlens'])[:, None]).unsqueeze(-1).float()
        p = {n: t.clone().requires_grad_(True) for n, t in p.items()}
        out = oracle.cosine_mha(q, k, v, kpm, p, 'a.', H, 0.01)
        (out * torch.from_numpy(g[pre + 'gout']).transpose(0, 1) * qvalid).sum().backward()
        np.testing.assert_allclose((out * qvalid).detach().transpose(0, 1).numpy(), g[pre + 'out'], atol=1e-5)
        np.testing.assert_allclose(q.grad.transpose(0, 1).numpy(), g[pre + 'dq'], atol=2e-4)
        np.testing.assert_allclose(v.grad.transpose(0, 1).numpy(), g[pre + 'dv'], atol=2e-4)
        np.testing.assert_allclose(p['a.tau'].grad.numpy(), g[pre + 'dtau'], atol=1e-3 * max(1.0, np.abs(g[pre + 'dtau']).max()))


def test_f14_attention_one_temperature_per_head(oracle):
    """CosineMultiheadAttention(non_shared_tau=True), cosine_msa.py:453-454 / :155-158: the reference's outputs and gradients (three
    cases, temperatures on both sides of the clamp at 0.01) vs the oracle with tau [1, H, 1, 1]."""
    g = golden('F14_options')
    for case in range(3):
        pre = f'h{case}_'
        E, H, T, nW, cross = [int(v) for v in g[pre + 'meta']]
        p = {'a.' + k[len(pre) + 2:].replace('__', '.'): torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + 'w_')}
        assert tuple(p['a.tau'].shape) == (1, H, 1, 1)
        q = torch.from_numpy(g[pre + 'q']).transpose(0, 1).clone().requires_grad_(True)
        v = torch.from_numpy(g[pre + 'v']).transpose(0, 1).clone().requires_grad_(True)
        k = torch.from_numpy(g[pre + 'k']).transpose(0, 1).clone().requires_grad_(True) if cross else q
        kpm = torch.from_numpy(g[pre + 'kpm'])
        qvalid = (torch.arange(T)[None, :] < torch.from_numpy(g[pre + 'qlens'])[:, None]).unsqueeze(-1).float()
        p = {n: t.clone().requires_grad_(True) for n, t in p.items()}
        out = oracle.cosine_mha(q, k, v, kpm, p, 'a.', H, 0.01)
        (out * torch.from_numpy(g[pre + 'gout']).transpose(0, 1) * qvalid).sum().backward()
        np.testing.assert_allclose((out * qvalid).detach().transpose(0, 1).numpy(), g[pre + 'out'], atol=1e-4)
        np.testing.assert_allclose(q.grad.transpose(0, 1).numpy(), g[pre + 'dq'], atol=2e-4)
        np.testing.assert_allclose(v.grad.transpose(0, 1).numpy(), g[pre + 'dv'], atol=2e-4)
        np.testing.assert_allclose(p['a.tau'].grad.numpy(), g[pre + 'dtau'], atol=1e-3 * max(1.0, np.abs(g[pre + 'dtau']).max()))
        assert float(np.abs(g[pre + 'dtau']).reshape(-1).min()) == 0.0          # the head below the clamp takes no gradient


def test_f8_encoder_blocks(oracle):
    g = golden('F8_encoder_blocks')
    cfg = oracle.default_model_cfg(3)
    P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']),
                           pred_scale=float(g['pred_scale']) if 'pred_scale' in g.files else 1.0)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    y = oracle.sst_encoder(x, g['coords'], (468, 468, 1), P, 'backbone_3d.sst_blocks.0.', cfg['stages'][0], cfg)
    (y * torch.from_numpy(g['gout'])).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), g['y'], atol=1e-4)
    np.testing.assert_allclose(x.grad.numpy(), g['dx'], atol=1e-3)
    xc = torch.from_numpy(g['w_xc']).requires_grad_(True)
    xp = torch.from_numpy(g['w_xp']).requires_grad_(True)
    cur, prv = oracle.sst_input_layer_temporal(g['w_coords_cur'], g['coords'], (468, 468, 1), cfg)
    res = xc
    for s in (0, 1):
        res = oracle.wca_encoder_layer(res, xp, cur, prv, s, P, f'backbone_3d.wca_blocks.0.encoder_blocks.0.encoder_list.{s}.', 8, cfg)
    (res * torch.from_numpy(g['w_gout'])).sum().backward()
    np.testing.assert_allclose(res.detach().numpy(), g['w_y'], atol=1e-4)
    np.testing.assert_allclose(xc.grad.numpy(), g['w_dxc'], atol=1e-3)
    np.testing.assert_allclose(xp.grad.numpy(), g['w_dxp'], atol=1e-3)


def test_f9_sparse_conv_vs_dense(oracle):
    g = golden('F9_sparse_conv')
    ind, feat, w = g['indices'], torch.from_numpy(g['feat']), torch.from_numpy(g['weight'])
    oi, osz, pairs = oracle.sparse_rulebook(ind, (468, 468), 'subm')
    ys = oracle.sparse_conv(feat, w, pairs, len(oi))
    np.testing.assert_allclose(ys.numpy(), g['subm_out'], atol=1e-6)
    dense = oracle.to_dense(feat, ind, (468, 468), 3)
    yd = torch.nn.functional.conv2d(dense, w.permute(0, 3, 1, 2).contiguous(), padding=1).permute(0, 2, 3, 1)
    np.testing.assert_allclose(ys.numpy(), yd[ind[:, 0], ind[:, 1], ind[:, 2]].numpy(), atol=1e-5)
    oi2, osz2, pairs2 = oracle.sparse_rulebook(ind, (468, 468), 'down')
    assert osz2 == (234, 234) and np.array_equal(oi2, g['down_indices'])
    assert [len(p[0]) for p in pairs2] == g['down_pairs'].tolist()
    y2 = oracle.sparse_conv(feat, w, pairs2, len(oi2))
    yd2 = torch.nn.functional.conv2d(dense, w.permute(0, 3, 1, 2).contiguous(), stride=2, padding=1).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y2.numpy(), yd2[oi2[:, 0], oi2[:, 1], oi2[:, 2]].numpy(), atol=1e-5)


def _e2e(oracle, name, nst):
    g = golden(name)
    cfg = oracle.default_model_cfg(nst)
    P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']),
                           pred_scale=float(g['pred_scale']) if 'pred_scale' in g.files else 1.0)
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    cap = {}
    loss = oracle.forward_loss(P, g['points'], g['points_prev'], g['noise'], int(g['batch_size']), cfg, cap)
    loss.backward()
    assert abs(fl(loss) - float(g['loss'])) < 1e-4                       # north-star bar
    assert np.array_equal(cap['mask'], g['mask'])
    np.testing.assert_allclose(cap['pred_points'].detach().numpy(), g['pred_points'], atol=1e-4)
    for n, gn in zip(g['grad_names'], g['grad_norms']):
        assert abs(float(P[str(n)].grad.norm()) - gn) <= 2e-3 * max(1.0, gn), n


def test_f11_e2e_one_stage(oracle):
    _e2e(oracle, 'F11_e2e_1stage', 1)


def test_f10_e2e_three_stage(oracle):
    _e2e(oracle, 'F10_e2e_3stage', 3)


def test_e2e_ragged_batch_matches_reference(oracle):
    """F12: batch of 3 where sample 1 has no current-frame points and sample 2 no previous-frame points."""
    _e2e(oracle, 'F12_e2e_ragged', 3)


def test_chamfer_zero_weights_and_groups(oracle):
    inv = np.array([2, 0, 2, 2, 1, 0])
    t = oracle.group_inner_inds(inv, 4, 4)
    assert t.tolist() == [[1, 5, 1, 5], [4, 4, 4, 4], [0, 2, 3, 0], [-1, -1, -1, -1]]
    pred, gt = torch.randn(3, 16, 3), torch.randn(3, 64, 3)
    assert float(oracle.chamfer_distance(pred, gt, torch.zeros(3))) == 0.0
    r = oracle.stable_ingroup_rank(np.array([5, 1, 5, 5, 1]))
    assert r.tolist() == [0, 0, 1, 2, 1]


def test_chamfer_against_kdtree_nearest_neighbours(oracle):
    """Independent cross-check of the Chamfer restatement (pytorch3d is absent: parity unpinned): the
    nearest-neighbour distances of every cloud pair from scipy.spatial.cKDTree (float64, a different algorithm: tree
    search, not the [P, G] distance matrix) -- per-cloud means each way, weights, / weights.sum().  Includes clouds
    with cyclically repeated gt points (< 64 points in the voxel) and zero-weight clouds."""
    from scipy.spatial import cKDTree
    O = oracle
    rng = np.random.default_rng(5)
    M, P, G = 300, 16, 64
    pred = rng.normal(0, 0.4, (M, P, 3)).astype(np.float32)
    gt = rng.normal(0, 0.4, (M, G, 3)).astype(np.float32)
    for m in range(0, M, 7):                                  # few distinct gt points, repeated cyclically
        c = int(rng.integers(1, 9))
        gt[m] = gt[m, np.arange(G) % c]
    w = (rng.random(M) < 0.7).astype(np.float32)
    num = 0.0
    for m in range(M):
        dx = cKDTree(gt[m].astype(np.float64)).query(pred[m].astype(np.float64))[0]     # pred -> nearest gt
        dy = cKDTree(pred[m].astype(np.float64)).query(gt[m].astype(np.float64))[0]     # gt -> nearest pred
        num += float(w[m]) * ((dx ** 2).mean() + (dy ** 2).mean())
    want = num / float(w.sum())
    got = float(O.chamfer_distance(torch.from_numpy(pred), torch.from_numpy(gt), torch.from_numpy(w)))
    assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (got, want)


def test_f13_e2e_token_dropping(oracle):
    """The oracle with a DROP_INFO that really drops tokens against what the reference computed (F13)."""
    g = golden('F13_e2e_dropping')
    nst, bs = int(g['num_stages']), int(g['batch_size'])
    cfg = oracle.default_model_cfg(nst)
    cfg['drop_info'] = {i: dict(max_tokens=int(t), drop_range=(int(lo), int(hi)))
                        for i, (t, lo, hi) in enumerate(zip(g['drop_max_tokens'], g['drop_lower'], g['drop_upper']))}
    P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']), pred_scale=float(g['pred_scale']))
    cap = {}
    with torch.no_grad():
        lo = oracle.forward_loss(P, g['points'], g['points_prev'], g['noise'], bs, cfg, cap)
    assert abs(fl(lo) - float(g['loss'])) < 1e-5
    assert np.array_equal(cap['mask'], g['mask'])
    assert int(g['dropped_stage1_unmasked']) > 1000
