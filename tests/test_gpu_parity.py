"""GPU (-m gpu): the HIP path, called through the C ABI (tmae_amd.ops -> libtmae_hip.so), against
 (1) the golden vectors captured from the reference (tests/golden),
 (2) the CPU oracle on seeded inputs,
 (3) size-independent properties at BASELINE configuration sizes.
Bars: bit-exact for every integer / index / mask output; stated fp tolerances otherwise."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import build_product_model, golden, fl

pytestmark = pytest.mark.gpu

PCR = [-74.88, -74.88, -5.0, 74.88, 74.88, 3.0]
VS = [0.32, 0.32, 8.0]
GRID = [468, 468, 1]
DROP = {0: dict(max_tokens=16, drop_range=(0, 16)), 1: dict(max_tokens=32, drop_range=(16, 32)),
        2: dict(max_tokens=64, drop_range=(32, 100000))}


def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda:0')


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


def test_native_library_is_loaded():
    from tmae_amd import _lib
    assert _lib.lib.tmae_abi_version() == _lib.ABI_VERSION
    maps = open('/proc/self/maps').read()
    assert 'libtmae_hip.so' in maps


# ------------------------------------------------------------------------------------------ A1 / A2 / A5 / A12

def test_voxelize_golden_bit_exact():
    from tmae_amd import ops
    g = golden('F1_F2_voxelize_vfe')
    v = ops.voxelize(cu(g['points']), 2, PCR, VS, GRID)
    assert np.array_equal(v['points'].cpu().numpy(), g['points_kept'])
    assert np.array_equal(v['point_coords'].cpu().numpy(), g['point_coords'])
    assert np.array_equal(v['voxel_coords'].cpu().numpy(), g['voxel_coords'])
    assert np.array_equal(v['inverse'].cpu().numpy(), g['inverse'])
    vc = g['voxel_coords']
    assert v['voxels_per_sample'] == [int((vc[:, 0] == b).sum()) for b in range(2)]


def test_voxelize_edge_and_empty(oracle):
    from tmae_amd import ops
    v = ops.voxelize(torch.zeros((0, 5), device=dev()), 2, PCR, VS, GRID)
    assert v['points'].shape[0] == 0 and v['voxel_coords'].shape[0] == 0 and v['voxels_per_sample'] == [0, 0]
    far = np.array([[0, 500., 0, 0, 1], [1, 0, -300., 0, 1], [0, 0, 0, 50., 1]], np.float32)
    v = ops.voxelize(cu(far), 2, PCR, VS, GRID)
    assert v['points'].shape[0] == 0 and v['voxel_coords'].shape[0] == 0
    rng = np.random.default_rng(0)
    # points exactly on voxel boundaries and range edges: the fp32 divide must round like the reference
    k = rng.integers(0, 469, (4000, 2))
    pts = np.zeros((4000, 5), np.float32)
    pts[:, 0] = rng.integers(0, 2, 4000)
    pts[:, 1] = np.float32(-74.88) + k[:, 0].astype(np.float32) * np.float32(0.32)
    pts[:, 2] = np.float32(-74.88) + k[:, 1].astype(np.float32) * np.float32(0.32)
    pts[:, 3] = rng.uniform(-14, 4, 4000)
    o = oracle.voxelize(pts, PCR, VS, GRID)
    v = ops.voxelize(cu(pts), 2, PCR, VS, GRID)
    assert np.array_equal(v['point_coords'].cpu().numpy(), o['point_coords'])
    assert np.array_equal(v['voxel_coords'].cpu().numpy(), o['voxel_coords'])
    assert np.array_equal(v['inverse'].cpu().numpy(), o['inverse'])


def test_ingroup_rank_and_csr(oracle):
    from tmae_amd import ops
    rng = np.random.default_rng(1)
    for n, ng in ((1, 1), (1000, 7), (50000, 3000), (200000, 60000)):
        g = rng.integers(0, ng, n)
        r = ops.get_inner_win_inds(cu(g))
        assert np.array_equal(r.cpu().numpy(), oracle.stable_ingroup_rank(g))
        m = int(g.max()) + 1
        perm, off = ops.segment_csr(cu(g), m)
        perm, off = perm.cpu().numpy(), off.cpu().numpy()
        assert off[0] == 0 and off[-1] == n and np.array_equal(np.diff(off), np.bincount(g, minlength=m))
        assert np.array_equal(perm, np.argsort(g, kind='stable'))
    assert ops.get_inner_win_inds(torch.zeros(0, dtype=torch.long, device=dev())).shape[0] == 0


def test_sst_ops_utils_api(oracle):
    """The reference's operator module surface (pcdet/ops/sst_ops/sst_ops_utils.py:5-27)."""
    from pcdet.ops.sst_ops import sst_ops_utils
    rng = np.random.default_rng(2)
    inv = rng.integers(0, 300, 2000)
    inv[:300] = np.arange(300)
    pts = rng.normal(size=(2000, 3)).astype(np.float32)
    out = sst_ops_utils.group_inner_inds(cu(pts), cu(inv), 8)
    table = oracle.group_inner_inds(inv, 300, 8)
    assert np.array_equal(out.cpu().numpy(), pts[table])
    r = sst_ops_utils.get_inner_win_inds(cu(inv))
    assert np.array_equal(r.cpu().numpy(), oracle.stable_ingroup_rank(inv))


def test_vfe_features_golden():
    g = golden('F1_F2_voxelize_vfe')
    params = {k.replace('__', '.'): torch.from_numpy(g[k]) for k in g.files if k.startswith('vfe__')}
    model, _, _ = build_product_model(3, params=params, device=dev(), partial=True)
    model.train()
    bd = {'points': cu(g['points']), 'points_prev': cu(g['points']), 'batch_size': 2}
    bd = model.vfe(bd)
    for suffix in ('', '_prev'):
        np.testing.assert_allclose(bd['voxel_features' + suffix].detach().cpu().numpy(), g['voxel_features'], atol=1e-5)
        assert np.array_equal(bd['voxel_coords' + suffix].cpu().numpy(), g['voxel_coords'])


def test_segment_max_backward(oracle):
    from tmae_amd import ops
    rng = np.random.default_rng(3)
    inv = rng.integers(0, 500, 4000)
    inv[:500] = np.arange(500)
    x = rng.normal(size=(4000, 128)).astype(np.float32)
    for dt in (torch.float32, torch.bfloat16):
        xt = cu(x, dt).requires_grad_(True)
        perm, off = ops.segment_csr(cu(inv), 500)
        out, arg = ops.scatter_max(xt, cu(inv), perm, off, 500)
        xo = xt.detach().float().cpu().requires_grad_(True)
        ref = oracle.segment_max(xo, torch.from_numpy(inv), 500)
        assert torch.equal(out.detach().float().cpu(), ref.detach())
        # argmax = FIRST point (ascending id) attaining the max
        a = arg.cpu().numpy()
        xv = xt.detach().float().cpu().numpy()
        for vv in (0, 17, 499):
            rows = np.nonzero(inv == vv)[0]
            assert np.array_equal(a[vv], rows[np.argmax(xv[rows], axis=0)])
        gout = torch.randn(500, 128)
        out.backward(cu(gout.numpy(), dt))
        ref.backward(gout.to(dt).float())
        if dt == torch.float32:                                              # no ties in fp32 random data
            assert torch.equal(xt.grad.cpu(), xo.grad)
        gsum = torch.zeros(500, 128).index_add_(0, torch.from_numpy(inv), xt.grad.float().cpu())
        assert torch.equal(gsum, gout.to(dt).float())                        # each voxel's gradient lands exactly once


def test_group_points_and_chamfer(oracle):
    from tmae_amd import ops
    rng = np.random.default_rng(4)
    pts, _ = oracle.synth_frame_pair(6000, 2, seed=9)
    v = ops.voxelize(cu(pts), 2, PCR, VS, GRID)
    m = v['voxel_coords'].shape[0]
    perm, off = ops.segment_csr(v['inverse'], m)
    ginds, gt = ops.group_points(v['points'], v['voxel_coords'], perm, off, 64, PCR, VS)
    table = oracle.group_inner_inds(v['inverse'].cpu().numpy(), m, 64)
    assert np.array_equal(ginds.cpu().numpy(), table)
    gt_ref = torch.from_numpy(v['points'].cpu().numpy()[:, 1:4])[torch.from_numpy(table)] - \
        oracle.voxel_centers(v['voxel_coords'].cpu().numpy()[:, 1:], VS, PCR).unsqueeze(1)
    np.testing.assert_allclose(gt.cpu().numpy(), gt_ref.numpy(), atol=1e-6)
    pred = torch.from_numpy(rng.normal(0, 0.3, (m, 16, 3)).astype(np.float32))
    w = torch.from_numpy((rng.random(m) < 0.75).astype(np.float32))
    pg = pred.clone().to(dev()).requires_grad_(True)
    loss, _ = ops.chamfer_distance(pg, gt, weights=w.to(dev()))
    loss.backward()
    po = pred.clone().requires_grad_(True)
    lo = oracle.chamfer_distance(po, gt_ref, w)
    lo.backward()
    assert abs(fl(loss) - fl(lo)) < 1e-5                               # north-star: Chamfer within 1e-4
    np.testing.assert_allclose(pg.grad.cpu().numpy(), po.grad.numpy(), atol=1e-7)
    z, _ = ops.chamfer_distance(pg.detach(), gt, weights=torch.zeros(m, device=dev()))
    assert float(z) == 0.0


# ------------------------------------------------------------------------------------------ A3

def test_mask_golden_and_ties(oracle):
    from tmae_amd import ops
    g = golden('F3_mask')
    vc = g['voxel_coords']
    per = [int((vc[:, 0] == b).sum()) for b in range(2)]
    offs = torch.tensor([0, per[0], per[0] + per[1]], dtype=torch.int32, device=dev())
    mask, vis, nvis = ops.random_mask(cu(g['noise']), offs, 2, 1 - float(g['mask_ratio']))
    assert np.array_equal(mask.cpu().numpy(), g['mask'])
    n = int(nvis.item())
    assert n == sum(int(L * 0.25) for L in per)
    assert np.array_equal(vis[:n].cpu().numpy(), np.nonzero(g['mask'] == 0)[0])
    # heavy ties: quantised noise; ties broken by index like a stable argsort
    rng = np.random.default_rng(5)
    noise = (rng.integers(0, 7, 5000) / 8).astype(np.float32)
    offs = torch.tensor([0, 1800, 5000], dtype=torch.int32, device=dev())
    mask, _, _ = ops.random_mask(cu(noise), offs, 2, 0.25)
    ref = np.concatenate([oracle.random_masking_from_noise(noise[:1800], 0.75), oracle.random_masking_from_noise(noise[1800:], 0.75)])
    assert np.array_equal(mask.cpu().numpy(), ref)


# ------------------------------------------------------------------------------------------ A4 / A10

def _bucket(coords4, grid_n, other4=None, shift=False, batch=3):
    from tmae_amd import ops
    ind = cu(coords4[:, [0, 2, 3]], torch.int32).contiguous()
    grid = ops.index_grid(ind, batch, grid_n, grid_n)
    gother = None
    if other4 is not None:
        gother = ops.index_grid(cu(other4[:, [0, 2, 3]], torch.int32).contiguous(), batch, grid_n, grid_n)
    wb = ops.window_bucket(ind, grid, gother, batch, grid_n, grid_n, [8, 8, 1], shift, DROP)
    # the keep-only form (no flat2win, no per-level counts: the rank scans are skipped) returns the same keep mask and ranks
    ko = ops.window_bucket(ind, grid, gother, batch, grid_n, grid_n, [8, 8, 1], shift, DROP, keep_only=True)
    assert set(ko) == {'keep', 'inner'} and torch.equal(ko['keep'], wb['keep']) and torch.equal(ko['inner'], wb['inner'])
    return wb


def test_window_partition_golden():
    g = golden('F4_window_partition')
    for grid_n in (468, 234, 117):
        for s in (0, 1):
            wb = _bucket(g[f'coords_{grid_n}'], grid_n, shift=s == 1)
            assert np.array_equal(wb['batch_win_inds'].cpu().numpy(), g[f'bwi_{grid_n}_s{s}'])
            assert np.array_equal(wb['coors_in_win'].cpu().numpy(), g[f'ciw_{grid_n}_s{s}'])


def test_bucketing_golden_single_and_temporal(oracle):
    g = golden('F5_bucketing')
    c = g['A_coords']
    for s in (0, 1):
        wb = _bucket(c, 468, shift=s == 1)
        lvl = wb['level'].cpu().numpy()
        assert np.array_equal(lvl, g[f'A_lvl_s{s}'])
        assert wb['keep'].cpu().numpy().all()                                # SURVEY A-6: nothing is dropped
        assert np.array_equal(wb['inner'].cpu().numpy(), oracle.stable_ingroup_rank(g[f'A_bwi_s{s}']))
        f2w = wb['flat2win'].cpu().numpy()
        for dl in (0, 1, 2):
            assert np.array_equal(f2w[lvl == dl], g[f'A_f2w_s{s}_l{dl}'])
            nwin = int(g[f'A_kpm_s{s}_l{dl}'].shape[0])
            assert int(wb['win_per_level'][dl]) == nwin
    cur, prv = g['T_coords_cur'], g['T_coords_prv']
    for s in (0, 1):
        for tag, a, b in (('cur', cur, prv), ('prv', prv, cur)):
            wb = _bucket(a, 468, other4=b, shift=s == 1)
            keep = wb['keep'].cpu().numpy().astype(bool)
            assert np.array_equal(np.nonzero(keep)[0], g[f'T_{tag}_voxel_keep_inds_s{s}'])
            assert np.array_equal(wb['level'].cpu().numpy()[keep], g[f'T_{tag}_voxel_drop_level_s{s}'])
            assert np.array_equal(wb['batch_win_inds'].cpu().numpy()[keep], g[f'T_{tag}_batch_win_inds_s{s}'])
            assert np.array_equal(wb['coors_in_win'].cpu().numpy()[keep], g[f'T_{tag}_coors_in_win_s{s}'])
            f2w, lvl = wb['flat2win'].cpu().numpy(), wb['level'].cpu().numpy()
            for dl in (0, 1, 2):
                key = f'T_{tag}_f2w_s{s}_l{dl}'
                if key in g.files:
                    assert np.array_equal(f2w[keep & (lvl == dl)], g[key])
                else:
                    assert not (keep & (lvl == dl)).any()


# ------------------------------------------------------------------------------------------ A6 / A7

def _place(lens, base_b=0):
    """Tokens of window w (t < lens[w]) at cell (t//8, t%8) of window (1 + w//50, 1 + w%50); returns
    coords [(b,y,x)] sorted lexicographically and row id of every (w,t)."""
    items = []
    for w, L in enumerate(lens):
        wy, wx = w // 50, w % 50
        for t in range(int(L)):
            items.append((base_b, wy * 8 + t // 8, wx * 8 + t % 8, w, t))
    items.sort()
    coords = np.array([(b, y, x) for b, y, x, _, _ in items], dtype=np.int32).reshape(-1, 3)
    row = {(w, t): i for i, (_, _, _, w, t) in enumerate(items)}
    return coords, row


def _rows(padded, lens, row):
    """padded [T, nW, E] -> ragged [M, E]."""
    out = np.zeros((len(row), padded.shape[2]), np.float32)
    for (w, t), i in row.items():
        out[i] = padded[t, w]
    return out


@pytest.mark.parametrize('case', [0, 1, 2, 3])
def test_attention_golden_fwd_bwd(case):
    """Real reference CosineMultiheadAttention outputs/grads (F7) vs the ragged HIP kernel, fp32."""
    from tmae_amd import ops
    g = golden('F7_attention')
    pre = f'c{case}_'
    E, H, T, nW, cross = [int(v) for v in g[pre + 'meta']]
    W = cu(g[pre + 'w_in_proj_weight']).requires_grad_(True)
    bias = cu(g[pre + 'w_in_proj_bias'])
    Wo, bo = cu(g[pre + 'w_out_proj__weight']), cu(g[pre + 'w_out_proj__bias'])
    tau = cu(g[pre + 'w_tau']).requires_grad_(True)
    klens = (~g[pre + 'kpm']).sum(1)
    qlens = g[pre + 'qlens']
    kc, krow = _place(klens)
    qc, qrow = _place(qlens) if cross else (kc, krow)
    gq = ops.index_grid(cu(qc), 1, 468, 468)
    gk = ops.index_grid(cu(kc), 1, 468, 468)
    Xq = cu(_rows(g[pre + 'q'], qlens, qrow)).requires_grad_(True)
    Xv = cu(_rows(g[pre + 'v'], klens, krow)).requires_grad_(True)
    if cross:
        Xk = cu(_rows(g[pre + 'k'], klens, krow)).requires_grad_(True)
        q = F.linear(Xq, W[:E], bias[:E])
        k = F.linear(Xk, W[E:2 * E], bias[E:2 * E])
        v = F.linear(Xv, W[2 * E:], bias[2 * E:])
        o = ops.win_attn(q, k, v, tau, gq, gk, H, 1, 468, 468, False, 0.01)
    else:
        qk = F.linear(Xq, W[:2 * E], bias[:2 * E])
        v = F.linear(Xv, W[2 * E:], bias[2 * E:])
        o = ops.win_attn(qk, v, None, tau, gq, gk, H, 1, 468, 468, False, 0.01)
    out = F.linear(o, Wo, bo)
    (out * cu(_rows(g[pre + 'gout'], qlens, qrow))).sum().backward()
    tol = 2e-5 if float(g[pre + 'tau']) >= 0.05 else 2e-4                  # logits reach +-100 at the tau clamp
    np.testing.assert_allclose(out.detach().cpu().numpy(), _rows(g[pre + 'out'], qlens, qrow), atol=tol)
    np.testing.assert_allclose(Xq.grad.cpu().numpy(), _rows(g[pre + 'dq'], qlens, qrow), atol=3e-4)
    np.testing.assert_allclose(Xv.grad.cpu().numpy(), _rows(g[pre + 'dv'], klens, krow), atol=3e-4)
    if cross:
        np.testing.assert_allclose(Xk.grad.cpu().numpy(), _rows(g[pre + 'dk'], klens, krow), atol=3e-4)
    dtau = g[pre + 'dtau']
    np.testing.assert_allclose(tau.grad.cpu().numpy(), dtau, atol=1e-3 * max(1.0, float(np.abs(dtau).max())))
    np.testing.assert_allclose(W.grad.cpu().numpy(), g[pre + 'd_in_proj_weight'], atol=1e-3)


@pytest.mark.parametrize('case', [0, 1, 2, 3])
def test_attention_mfma_bf16_vs_reference_golden(case):
    """The BENCHED attention path -- bf16 MFMA kernels driven by class-binned work lists -- straight against the real
    reference CosineMultiheadAttention (F7), not via the fp32 kernel: relative L2 2e-2 (SURVEY 8c, F7 "bf16 2e-2 rel")
    for the output and the q / k / v input gradients.  Case 2 sits at the temperature clamp (tau 0.005 -> 0.01): logits
    reach +-100, and rounding q, k to bf16 alone (2^-9 relative per element, before any kernel arithmetic) moves them by
    ~0.2 -- its bar is 0.1, stated here, not hidden in a shared constant."""
    from tmae_amd import ops
    g = golden('F7_attention')
    pre = f'c{case}_'
    E, H, T, nW, cross = [int(v) for v in g[pre + 'meta']]
    W, bias = cu(g[pre + 'w_in_proj_weight']), cu(g[pre + 'w_in_proj_bias'])
    Wo, bo = cu(g[pre + 'w_out_proj__weight']), cu(g[pre + 'w_out_proj__bias'])
    tau = cu(g[pre + 'w_tau']).requires_grad_(True)
    klens = (~g[pre + 'kpm']).sum(1)
    qlens = g[pre + 'qlens']
    kc, krow = _place(klens)
    qc, qrow = _place(qlens) if cross else (kc, krow)
    gq = ops.index_grid(cu(qc), 1, 468, 468)
    gk = ops.index_grid(cu(kc), 1, 468, 468)
    wl = ops.window_worklist(gq, gk, 1, 468, 468, False)
    Xq = cu(_rows(g[pre + 'q'], qlens, qrow)).requires_grad_(True)
    Xv = cu(_rows(g[pre + 'v'], klens, krow)).requires_grad_(True)
    bf = torch.bfloat16
    if cross:
        Xk = cu(_rows(g[pre + 'k'], klens, krow)).requires_grad_(True)
        q = F.linear(Xq, W[:E], bias[:E]).to(bf)
        k = F.linear(Xk, W[E:2 * E], bias[E:2 * E]).to(bf)
        v = F.linear(Xv, W[2 * E:], bias[2 * E:]).to(bf)
        o = ops.win_attn(q, k, v, tau, gq, gk, H, 1, 468, 468, False, 0.01, worklist=wl)
    else:
        qk = F.linear(Xq, W[:2 * E], bias[:2 * E]).to(bf)
        v = F.linear(Xv, W[2 * E:], bias[2 * E:]).to(bf)
        o = ops.win_attn(qk, v, None, tau, gq, gk, H, 1, 468, 468, False, 0.01, worklist=wl)
    assert o.dtype == bf
    out = F.linear(o.float(), Wo, bo)
    (out * cu(_rows(g[pre + 'gout'], qlens, qrow))).sum().backward()

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

    lim = 2e-2 if float(g[pre + 'tau']) >= 0.05 else 0.1
    errs = {'out': rel(out.detach().cpu().numpy(), _rows(g[pre + 'out'], qlens, qrow)),
            'dq': rel(Xq.grad.cpu().numpy(), _rows(g[pre + 'dq'], qlens, qrow)),
            'dv': rel(Xv.grad.cpu().numpy(), _rows(g[pre + 'dv'], klens, krow))}
    if cross:
        errs['dk'] = rel(Xk.grad.cpu().numpy(), _rows(g[pre + 'dk'], klens, krow))
    for name, e in errs.items():
        assert e < lim, (case, name, e, errs)
    dtau = float(np.asarray(g[pre + 'dtau']).reshape(-1)[0])
    assert abs(float(tau.grad) - dtau) <= 3 * lim * max(1.0, abs(dtau)), (float(tau.grad), dtau)


@pytest.mark.parametrize('case', [0, 1, 2])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_attention_one_temperature_per_head_golden(case, dtype):
    """non_shared_tau (cosine_msa.py:453-454, :155-158; off in the shipped YAMLs): the real CosineMultiheadAttention with a
    temperature per head (F14: 0.005 ... 2.0 shuffled over the heads, i.e. below the clamp, below and above the MFMA kernels' split
    threshold inside one workgroup) vs the fp32 kernels (F7's tolerances) and the benched bf16 MFMA kernels with work lists (F7's
    bf16 bars: 2e-2 relative, 0.1 for what passes through the clamped head); d tau per head, exactly 0 for the clamped one."""
    from tmae_amd import ops
    g = golden('F14_options')
    pre = f'h{case}_'
    E, H, T, nW, cross = [int(v) for v in g[pre + 'meta']]
    W = cu(g[pre + 'w_in_proj_weight']).requires_grad_(True)
    bias = cu(g[pre + 'w_in_proj_bias'])
    Wo, bo = cu(g[pre + 'w_out_proj__weight']), cu(g[pre + 'w_out_proj__bias'])
    tau = cu(g[pre + 'w_tau']).requires_grad_(True)
    assert tuple(tau.shape) == (1, H, 1, 1)
    klens = (~g[pre + 'kpm']).sum(1)
    qlens = g[pre + 'qlens']
    kc, krow = _place(klens)
    qc, qrow = _place(qlens) if cross else (kc, krow)
    gq = ops.index_grid(cu(qc), 1, 468, 468)
    gk = ops.index_grid(cu(kc), 1, 468, 468)
    bf = dtype == 'bf16'
    wl = ops.window_worklist(gq, gk, 1, 468, 468, False) if bf else None
    cast = (lambda t: t.to(torch.bfloat16)) if bf else (lambda t: t)
    Xq = cu(_rows(g[pre + 'q'], qlens, qrow)).requires_grad_(True)
    Xv = cu(_rows(g[pre + 'v'], klens, krow)).requires_grad_(True)
    if cross:
        Xk = cu(_rows(g[pre + 'k'], klens, krow)).requires_grad_(True)
        q = cast(F.linear(Xq, W[:E], bias[:E]))
        k = cast(F.linear(Xk, W[E:2 * E], bias[E:2 * E]))
        v = cast(F.linear(Xv, W[2 * E:], bias[2 * E:]))
        o = ops.win_attn(q, k, v, tau, gq, gk, H, 1, 468, 468, False, 0.01, worklist=wl)
    else:
        qk = cast(F.linear(Xq, W[:2 * E], bias[:2 * E]))
        v = cast(F.linear(Xv, W[2 * E:], bias[2 * E:]))
        o = ops.win_attn(qk, v, None, tau, gq, gk, H, 1, 468, 468, False, 0.01, worklist=wl)
    out = F.linear(o.float(), Wo, bo)
    (out * cu(_rows(g[pre + 'gout'], qlens, qrow))).sum().backward()
    want = {'out': _rows(g[pre + 'out'], qlens, qrow), 'dq': _rows(g[pre + 'dq'], qlens, qrow), 'dv': _rows(g[pre + 'dv'], klens, krow)}
    got = {'out': out.detach().cpu().numpy(), 'dq': Xq.grad.cpu().numpy(), 'dv': Xv.grad.cpu().numpy()}
    if cross:
        want['dk'], got['dk'] = _rows(g[pre + 'dk'], klens, krow), Xk.grad.cpu().numpy()
    dtau = np.asarray(g[pre + 'dtau']).reshape(-1)
    mine = tau.grad.cpu().numpy().reshape(-1)
    assert tau.grad.shape == tau.shape
    clamped = np.asarray(g[pre + 'w_tau']).reshape(-1) < 0.01
    assert clamped.sum() == 1 and float(np.abs(mine[clamped]).max()) == 0.0 and float(np.abs(dtau[clamped]).max()) == 0.0
    if not bf:
        for name in want:                                                   # one head sits at the clamp: F7's clamped-case bars
            np.testing.assert_allclose(got[name], want[name], atol=2e-4 if name == 'out' else 3e-4, err_msg=name)
        np.testing.assert_allclose(mine, dtau, atol=1e-3 * max(1.0, float(np.abs(dtau).max())))
        np.testing.assert_allclose(W.grad.cpu().numpy(), g[pre + 'd_in_proj_weight'], atol=1e-3)
    else:
        def rel(a, b):
            a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
            return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
        for name in want:
            assert rel(got[name], want[name]) < 0.1, (case, name, rel(got[name], want[name]))
        # per head: the un-clamped heads with tau >= 0.05 at the 2e-2 bar F7 uses for such temperatures
        dh = E // H
        taus = np.asarray(g[pre + 'w_tau']).reshape(-1)
        for h in range(H):
            if taus[h] >= 0.05:
                assert abs(mine[h] - dtau[h]) <= 6e-2 * max(1.0, abs(dtau[h])), (h, mine[h], dtau[h])
            else:
                assert abs(mine[h] - dtau[h]) <= 0.3 * max(1.0, abs(dtau[h])), (h, mine[h], dtau[h])


def test_gather_rows_scatters_its_gradient():
    """ops.gather_rows (the visible voxels of the masked frame, SiamWCA_MAE.py:166-182): x[idx] for distinct indices; the backward puts
    every gradient row in its place of a zeroed tensor -- equal to autograd's own (sorting, accumulating) backward of x[idx]; empty
    index lists and a single row included."""
    from tmae_amd import ops
    g = torch.Generator(device=dev()).manual_seed(4)
    for (m, c, n, dt) in ((1000, 128, 250, torch.bfloat16), (37, 16, 37, torch.float32), (50, 8, 0, torch.float32), (1, 128, 1, torch.bfloat16)):
        x = torch.randn(m, c, device=dev(), generator=g).to(dt)
        idx = torch.randperm(m, device=dev(), generator=g)[:n]
        go = torch.randn(n, c, device=dev(), generator=g).to(dt)
        a = x.clone().requires_grad_(True)
        y = ops.gather_rows(a, idx)
        y.backward(go)
        b = x.clone().requires_grad_(True)
        yr = b[idx]
        yr.backward(go)
        assert torch.equal(y, yr) and torch.equal(a.grad, b.grad) and a.grad.shape == x.shape


def test_attention_bf16_and_softmax_property():
    """bf16 I/O (fp32 softmax/normalise) stays within bf16 tolerance of the fp32 kernel; with V = 1 every
    attended row must come back as exactly-normalised ones (rows of P sum to 1) -- at stage-1 size."""
    from tmae_amd import ops
    rng = np.random.default_rng(7)
    n = 40000
    c = np.unique(np.stack([rng.integers(0, 2, n), rng.integers(0, 468, n), rng.integers(0, 468, n)], 1), axis=0)
    ind = cu(c, torch.int32)
    grid = ops.index_grid(ind, 2, 468, 468)
    m = len(c)
    tau = torch.full((1, 1, 1), 0.07, device=dev())
    for d, H in ((128, 8), (256, 8)):
        qk = torch.randn(m, 2 * d, device=dev())
        v = torch.randn(m, d, device=dev())
        for shift in (False, True):
            o32 = ops.win_attn(qk, v, None, tau, grid, grid, H, 2, 468, 468, shift, 0.01)
            o16 = ops.win_attn(qk.bfloat16(), v.bfloat16(), None, tau, grid, grid, H, 2, 468, 468, shift, 0.01)
            err = (o16.float() - o32).abs().max().item()
            assert err < 0.25 and (o16.float() - o32).abs().mean().item() < 2e-2, err
            ones = ops.win_attn(qk, torch.ones_like(v), None, tau, grid, grid, H, 2, 468, 468, shift, 0.01)
            assert (ones - 1).abs().max().item() < 1e-5
            ones16 = ops.win_attn(qk.bfloat16(), torch.ones_like(v).bfloat16(), None, tau, grid, grid, H, 2, 468, 468,
                                  shift, 0.01)
            assert torch.isfinite(ones16.float()).all() and (ones16.float() - 1).abs().max().item() < 1.5e-2


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_attention_mfma_bf16_fwd_bwd_vs_fp32_kernel(seed):
    """The bf16 MFMA kernels (forward + backward) against the fp32 kernels (which the golden tests pin to the
    reference) on the same inputs: self- and cross-attention, dh 16 and 32, both shifts, tau at and above the clamp.
    (Seeded: the tau gradient is a sum over ~1e5 query rows whose rounding noise a lucky draw can hide.)"""
    from tmae_amd import ops
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)

    def cloud(n, b):
        c = np.unique(np.stack([rng.integers(0, b, n), rng.integers(0, 234, n), rng.integers(0, 234, n)], 1), axis=0)
        dense = np.stack(np.meshgrid(np.arange(40, 72), np.arange(40, 72), indexing='ij'), -1).reshape(-1, 2)
        dense = dense[rng.random(len(dense)) < 0.7]
        c = np.unique(np.concatenate([c, np.concatenate([np.zeros((len(dense), 1), np.int64), dense], 1)]), axis=0)
        return cu(c, torch.int32)

    def rel(a, b):
        a, b = a.detach(), b.detach()
        return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))

    for d, H, tauv, cross in ((128, 8, 0.3, False), (256, 8, 1.0, False), (256, 8, 0.05, True), (128, 8, 0.005, True)):
        indq = cloud(15000, 2)
        indk = cloud(30000, 2) if cross else indq
        gq = ops.index_grid(indq, 2, 234, 234)
        gk = ops.index_grid(indk, 2, 234, 234) if cross else gq
        mq, mk = indq.shape[0], indk.shape[0]
        for shift in (False, True):
            res = {}
            base = [torch.randn(mq, 2 * d if not cross else d, device=dev()), torch.randn(mk, d, device=dev()),
                    torch.randn(mk, d, device=dev()) if cross else None]
            go = torch.randn(mq, d, device=dev())
            for dt in (torch.float32, torch.bfloat16):
                # identical (bf16-representable) inputs for both kernels: the comparison isolates kernel precision
                a, b_, c_ = [None if t is None else t.bfloat16().to(dt).clone().requires_grad_(True) for t in base]
                tau = torch.full((1, 1, 1), tauv, device=dev(), requires_grad=True)
                o = ops.win_attn(a, b_, c_, tau, gq, gk, H, 2, 234, 234, shift, 0.01)
                o.backward(go.bfloat16().to(dt))
                res[dt] = (o.detach(), a.grad, b_.grad, None if c_ is None else c_.grad, tau.grad)
                if dt == torch.bfloat16:
                    # class-binned work lists (what the model uses) must give the dense-window kernel's numbers
                    wl = ops.window_worklist(gq, gk, 2, 234, 234, shift)
                    cnt = wl[:4].cpu().numpy()                                    # <= 8 (paired), <= 16, <= 32, <= 64 tokens
                    assert cnt[0] > 1 and cnt[1] > 0 and cnt[3] > 0             # the fixture populates the classes
                    a2, b2, c2 = [None if t is None else t.detach().clone().requires_grad_(True) for t in (a, b_, c_)]
                    tau2 = torch.full((1, 1, 1), tauv, device=dev(), requires_grad=True)
                    o2 = ops.win_attn(a2, b2, c2, tau2, gq, gk, H, 2, 234, 234, shift, 0.01, worklist=wl)
                    o2.backward(go.bfloat16().to(dt))
                    def same(x, y, what):
                        assert torch.isfinite(x.float()).all() and torch.isfinite(y.float()).all(), what
                        # same algorithm, different template instantiation: FMA contraction may differ by a bf16 ulp
                        r = rel(x, y)
                        assert r < 1e-3, (what, d, cross, shift, r, int((x != y).sum()), x.numel())
                    same(o2, o, 'out'), same(a2.grad, a.grad, 'da'), same(b2.grad, b_.grad, 'db')
                    if c2 is not None:
                        same(c2.grad, c_.grad, 'dc')
                    # the tau gradient sums dS*S over every window: instantiations that mask differently (C-operand
                    # bias for <= 32-token windows, per-element select for 64) differ in rounding, not in meaning
                    assert abs(float(tau2.grad) - float(tau.grad)) <= 5e-3 * max(1.0, abs(float(tau.grad)))
                    if cross:
                        # covered=True: only the rows of the orphan windows are zeroed (tmae_win_attn_zero_orphans) instead of the
                        # whole outputs -- the two clouds differ, so windows with queries and no keys (and the reverse) exist; the
                        # result must be the pre-zeroed one bit for bit, and the buffers are poisoned first to prove it
                        assert int((ops.index_grid(indq, 2, 234, 234) >= 0).sum()) == mq
                        junk = [torch.full((mk, d), float('nan'), device=dev(), dtype=torch.bfloat16) for _ in range(4)]
                        del junk                                                    # NaN-filled blocks for the allocator to hand back
                        a3, b3, c3 = [None if t is None else t.detach().clone().requires_grad_(True) for t in (a, b_, c_)]
                        tau3 = torch.full((1, 1, 1), tauv, device=dev(), requires_grad=True)
                        o3 = ops.win_attn(a3, b3, c3, tau3, gq, gk, H, 2, 234, 234, shift, 0.01, worklist=wl, covered=True)
                        o3.backward(go.bfloat16().to(dt))
                        assert torch.equal(o3, o2) and torch.equal(a3.grad, a2.grad) and torch.equal(b3.grad, b2.grad)
                        assert c3 is None or torch.equal(c3.grad, c2.grad)
                        assert torch.equal(tau3.grad, tau2.grad)
            f, h = res[torch.float32], res[torch.bfloat16]
            assert all(torch.isfinite(t.float()).all() for t in h if t is not None)
            lim = 0.03 if tauv >= 0.05 else 0.12          # logits reach +-100 at the clamp: bf16 logit error ~0.4
            assert rel(h[0], f[0]) < lim, ('out', d, cross, shift, rel(h[0], f[0]))
            assert rel(h[1], f[1]) < 2 * lim, ('da', d, cross, shift, rel(h[1], f[1]))
            assert rel(h[2], f[2]) < 2 * lim, ('db', d, cross, shift, rel(h[2], f[2]))
            if cross:
                assert rel(h[3], f[3]) < 2 * lim, ('dc', d, cross, shift, rel(h[3], f[3]))
            assert abs(float(h[4]) - float(f[4])) <= 3 * lim * max(1.0, abs(float(f[4]))), ('dtau', d, cross, shift, float(h[4]), float(f[4]))


def test_encoder_blocks_golden(oracle):
    """F8: SSTBlockV1 encoder (4 layers) and the WCA block (2 cross layers), outputs + input grads, fp32."""
    from tmae_amd.modules.sparse import SparseConvTensor
    g = golden('F8_encoder_blocks')
    cfg = oracle.default_model_cfg(3)
    P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']),
                           pred_scale=float(g['pred_scale']) if 'pred_scale' in g.files else 1.0)
    model, _, _ = build_product_model(3, params=P, device=dev())
    model.train()
    blk = model.backbone_3d.sst_blocks[0]
    c = g['coords']
    x = cu(g['x']).requires_grad_(True)
    sp = SparseConvTensor(x, cu(c[:, [0, 2, 3]], torch.int32), [468, 468], 3)
    y = blk.encoder_forward(sp)
    (y * cu(g['gout'])).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), g['y'], atol=2e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g['dx'], atol=1e-3)
    gW = blk.encoder_blocks[1].encoder_list[1].win_attn.self_attn.in_proj_weight.grad
    np.testing.assert_allclose(gW.cpu().numpy(), g['dW_last_in_proj'], atol=1e-3)
    gt = blk.encoder_blocks[0].encoder_list[0].win_attn.self_attn.tau.grad
    np.testing.assert_allclose(gt.cpu().numpy(), g['dtau_first'], atol=1e-3 * max(1.0, float(np.abs(g['dtau_first']).max())))
    wb = model.backbone_3d.wca_blocks[0]
    cc = g['w_coords_cur']
    xc = cu(g['w_xc']).requires_grad_(True)
    xp = cu(g['w_xp']).requires_grad_(True)
    spc = SparseConvTensor(xc, cu(cc[:, [0, 2, 3]], torch.int32), [468, 468], 3)
    spp = SparseConvTensor(xp, cu(c[:, [0, 2, 3]], torch.int32), [468, 468], 3)
    yc = wb.encoder_forward(spc, spp)
    (yc * cu(g['w_gout'])).sum().backward()
    np.testing.assert_allclose(yc.detach().cpu().numpy(), g['w_y'], atol=2e-4)
    np.testing.assert_allclose(xc.grad.cpu().numpy(), g['w_dxc'], atol=1e-3)
    np.testing.assert_allclose(xp.grad.cpu().numpy(), g['w_dxp'], atol=1e-3)


def test_linear_wgrad_kernel_vs_torch():
    """Split-token MFMA weight/bias gradient (csrc/wgrad.hip) vs torch in fp32 on the same bf16 inputs."""
    from tmae_amd import ops
    torch.manual_seed(0)
    for (m, n, k) in ((4096, 128, 128), (50001, 256, 128), (37777, 48, 128), (20000, 256, 1152), (9000, 512, 256),
                      (131072, 64, 64), (30011, 256, 256), (12345, 256, 512), (8200, 2048, 256), (40000, 264, 328)):
        dy = (torch.randn(m, n, device=dev()) * 0.5).bfloat16()
        x = torch.randn(m, k, device=dev()).bfloat16()
        dw, db = ops.linear_wgrad(dy, x)
        ref_w = dy.float().t() @ x.float()
        ref_b = dy.float().sum(0)
        scale = float(ref_w.abs().max())
        assert (dw - ref_w).abs().max().item() <= 2e-3 * scale, (m, n, k)
        assert (db - ref_b).abs().max().item() <= 2e-3 * float(ref_b.abs().max()) + 1e-3, (m, n, k)
    # token counts around the 32-row slice and the two-slice prefetch depth (clamped loads past the end, 1-3 slices per
    # workgroup), both tile sizes, partial output tiles
    for (m, n, k) in ((1, 128, 128), (31, 256, 256), (33, 512, 256), (64, 128, 256), (95, 512, 512), (257, 264, 328),
                      (300, 768, 256)):
        dy = torch.randn(m, n, device=dev()).bfloat16()
        x = torch.randn(m, k, device=dev()).bfloat16()
        dw, db = ops.linear_wgrad(dy, x)
        ref_w = dy.float().t() @ x.float()
        assert (dw - ref_w).abs().max().item() <= 2e-3 * max(1.0, float(ref_w.abs().max())), (m, n, k)
        assert (db - dy.float().sum(0)).abs().max().item() <= 2e-3 * max(1.0, float(dy.float().sum(0).abs().max())), (m, n, k)
    # strided views (columns of a packed buffer), as the attention projections produce them
    big = torch.randn(30000, 384, device=dev()).bfloat16()
    dy, x = big[:, 128:384], big[:, :128]
    dw, db = ops.linear_wgrad(dy, x)
    ref = dy.float().t() @ x.float()
    assert (dw - ref).abs().max().item() <= 2e-3 * float(ref.abs().max())
    # autograd wrapper: same numbers as F.linear's autograd in bf16
    xg = torch.randn(20000, 128, device=dev()).bfloat16().requires_grad_(True)
    w = torch.randn(256, 128, device=dev(), requires_grad=True)
    b = torch.randn(256, device=dev(), requires_grad=True)
    go = torch.randn(20000, 256, device=dev()).bfloat16()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = ops.linear(xg, w, b)
    y.backward(go)
    g1 = (xg.grad.clone(), w.grad.clone(), b.grad.clone())
    xg.grad = w.grad = b.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y2 = F.linear(xg, w, b)
    y2.backward(go)
    assert torch.equal(y, y2)
    assert (g1[0].float() - xg.grad.float()).abs().max().item() < 1e-1
    assert (g1[1] - w.grad).abs().max().item() <= 1e-2 * float(w.grad.abs().max())
    assert (g1[2] - b.grad).abs().max().item() <= 1e-2 * float(b.grad.abs().max())


def test_linear_wgrad_cell_sums_both_tile_sizes():
    """tmae_linear_wgrad_cells: dW, db and the per-cell column sums dcell[c, n] = sum of dy[m, n] over the tokens of cell slot c
    (0..7 x cell, 8..15 y cell) for n < pos_n, zeros past pos_n -- in ONE pass for the 128-tile kernel and (round 5) for the
    256-tile kernel, whose four waves of a row share out the one-hot products; ragged token counts, 1-3 row blocks, pos_n below n."""
    from tmae_amd import ops
    torch.manual_seed(5)
    for (m, n, k, pos_n) in ((40001, 512, 256, 512), (131072, 512, 256, 256), (9000, 768, 256, 512), (333, 512, 512, 512),
                             (50000, 256, 128, 256), (20011, 256, 256, 256), (65, 768, 256, 256)):
        ind = torch.stack([torch.zeros(m, dtype=torch.int64), torch.randint(0, 468, (m,)), torch.randint(0, 468, (m,))], 1).int().to(dev())
        cells, onehot = ops.window_cells(ind, [8, 8, 1], True, want_onehot=True)
        dy = (torch.randn(m, n, device=dev()) * 0.5).bfloat16()
        x = torch.randn(m, k, device=dev()).bfloat16()
        dw, db, dcell = ops.linear_wgrad(dy, x, True, cells=cells, pos_n=pos_n)
        ref_w = dy.float().t() @ x.float()
        ref_b = dy.float().sum(0)
        ref_c = onehot.float().t() @ dy.float()
        ref_c[:, pos_n:] = 0
        assert (dw - ref_w).abs().max().item() <= 2e-3 * max(1.0, float(ref_w.abs().max())), (m, n, k)
        assert (db - ref_b).abs().max().item() <= 2e-3 * float(ref_b.abs().max()) + 1e-3, (m, n, k)
        assert dcell.shape == (16, n)
        assert (dcell - ref_c).abs().max().item() <= 2e-3 * max(1.0, float(ref_c.abs().max())), (m, n, k, pos_n)
        assert float(dcell[:, pos_n:].abs().max()) == 0.0 if pos_n < n else True
        # the x-cell sums and the y-cell sums both add up to the bias gradient
        assert (dcell[:8, :pos_n].sum(0) - ref_b[:pos_n]).abs().max().item() <= 4e-3 * float(ref_b.abs().max()) + 1e-3
        # pos_e: the position part dcell^T E of the weight gradient added inside the slab reduction (no dcell tensor, no GEMM)
        E = torch.randn(16, k, device=dev())
        dw2, db2, none = ops.linear_wgrad(dy, x, True, cells=cells, pos_n=pos_n, pos_e=E)
        ref_w2 = ref_w + ref_c.t() @ E
        assert none is None and torch.equal(db2, db)
        assert (dw2 - ref_w2).abs().max().item() <= 2e-3 * max(1.0, float(ref_w2.abs().max())), (m, n, k, pos_n)


def test_token_gemm_kernel_vs_torch():
    """x-stationary token GEMM (csrc/token_gemm.hip) vs fp32 matmul on the same bf16 inputs: forward with bias,
    strided input (column slice of a packed buffer), ragged token counts, and the dX use on W^T."""
    from tmae_amd import ops
    torch.manual_seed(3)
    # (>= 32768 tokens: the W-in-registers kernel of csrc/token_gemm_wreg.hip; below: the chunk-streaming kernel)
    for (m, k, n) in ((8192, 128, 128), (50001, 128, 256), (33333, 256, 512), (20000, 256, 768), (9999, 256, 64),
                      (30001, 512, 256), (12000, 512, 64), (9000, 256, 2304), (70001, 256, 512), (150003, 256, 256),
                      (65536, 256, 512)):
        x = torch.randn(m, k, device=dev()).bfloat16()
        w = (torch.randn(n, k, device=dev()) * 0.1).bfloat16()
        b = torch.randn(n, device=dev()).bfloat16()
        y = ops.token_gemm(x, w, b, force=True)
        ref = x.float() @ w.float().t() + b.float()
        assert y.shape == (m, n) and y.dtype == torch.bfloat16
        assert (y.float() - ref).abs().max().item() <= 2e-2 * max(1.0, float(ref.abs().max())), (m, k, n)
        y0 = ops.token_gemm(x, w, None, force=True)
        assert (y0.float() - x.float() @ w.float().t()).abs().max().item() <= 2e-2 * max(1.0, float(ref.abs().max()))
        lib_y = F.linear(x, w, b)
        assert (y.float() - lib_y.float()).abs().max().item() <= 4e-2 * max(1.0, float(ref.abs().max()))
    # contraction 32 (the VFE's first Linear on the hi | lo split point features): one k-step, 64 output columns
    for m in (8192, 905005, 33):
        x = torch.randn(m, 32, device=dev()).bfloat16()
        w = (torch.randn(64, 32, device=dev()) * 0.3).bfloat16()
        y = torch.empty((m, 64), dtype=torch.bfloat16, device=dev())
        from tmae_amd._lib import lib, check
        check(lib.tmae_token_gemm(x.data_ptr(), 32, m, 32, w.data_ptr(), 64, torch.zeros(64, device=dev()).bfloat16().data_ptr(),
                                  y.data_ptr(), 64, torch.cuda.current_stream().cuda_stream), 'tmae_token_gemm')
        ref = x.float() @ w.float().t()
        assert (y.float() - ref).abs().max().item() <= 2e-2 * max(1.0, float(ref.abs().max())), m
    big = torch.randn(30000, 384, device=dev()).bfloat16()
    xs = big[:, 128:384]                                   # pitch 384, 256 columns
    w = (torch.randn(128, 256, device=dev()) * 0.1).bfloat16()
    y = ops.token_gemm(xs, w, None, force=True)
    ref = xs.float() @ w.float().t()
    assert (y.float() - ref).abs().max().item() <= 2e-2 * float(ref.abs().max())
    big = torch.randn(80001, 384, device=dev()).bfloat16()   # the same through the W-in-registers kernel
    xs = big[:, 128:384]
    w = (torch.randn(256, 256, device=dev()) * 0.1).bfloat16()
    y = ops.token_gemm(xs, w, None, force=True)
    ref = xs.float() @ w.float().t()
    assert (y.float() - ref).abs().max().item() <= 2e-2 * float(ref.abs().max())
    dy = torch.randn(40000, 256, device=dev()).bfloat16()
    w = (torch.randn(256, 512, device=dev()) * 0.1).bfloat16()     # [n, k]: dx = dy @ w -> [m, 512]
    dx = ops.token_gemm_dx(dy, w, force=True)
    ref = dy.float() @ w.float()
    assert dx.shape == (40000, 512) and (dx.float() - ref).abs().max().item() <= 2e-2 * float(ref.abs().max())


def test_token_gemm_wreg_kernel_every_instantiation_vs_torch():
    """The W-in-registers token GEMM (csrc/token_gemm_wreg.hip, >= 32 k tokens): every (K, N, position, accumulate)
    instantiation against fp32 matmul on the same bf16 inputs, on token counts that leave ragged last steps (a step =
    128 / 64 / 32 tokens for K = 128 / 256 / 512), workgroups without work (32 769 tokens: fewer steps than CUs for
    K = 128) and a strided x; the position columns against the one-hot formula; the in-place form against addmm."""
    from tmae_amd import ops
    from tmae_amd._lib import lib, check
    torch.manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    for (m, k, n, pos) in ((32769, 128, 128, False), (32769 + 77, 128, 256, False), (40003, 128, 384, True), (33001, 128, 128, True),
                           (65537, 128, 256, True), (32768 + 63, 256, 128, False), (50001, 256, 256, False),
                           (33333, 256, 512, False), (66001, 256, 768, True), (40001, 256, 512, True), (36001, 256, 256, True),
                           (70001, 256, 1024, False), (65536 + 31, 512, 256, False), (100001, 512, 256, False)):
        ka = k + (32 if pos else 0)
        x = torch.randn(m, k, device=dev()).bfloat16()
        w = (torch.randn(n, ka, device=dev()) * 0.1).bfloat16()
        b = torch.randn(n, device=dev()).bfloat16()
        y = torch.full((m, n), float('nan'), device=dev(), dtype=torch.bfloat16)
        xf = x.float()
        if pos:
            cells = torch.randint(0, 64, (m,), device=dev(), dtype=torch.uint8)
            check(lib.tmae_token_gemm_pos(x.data_ptr(), k, m, k, w.data_ptr(), n, b.data_ptr(), cells.data_ptr(),
                                          y.data_ptr(), n, st), 'tmae_token_gemm_pos')
            oh = torch.zeros(m, 32, device=dev())
            xc, yc = (cells & 7).long(), (cells >> 3).long()
            for gk in range(4):                       # hi / lo halves of the x-cell and y-cell tables
                oh[torch.arange(m, device=dev()), gk * 8 + (yc if gk & 1 else xc)] = 1.0
            xf = torch.cat([xf, oh], 1)
        else:
            check(lib.tmae_token_gemm(x.data_ptr(), k, m, k, w.data_ptr(), n, b.data_ptr(), y.data_ptr(), n, st),
                  'tmae_token_gemm')
        ref = xf @ w.float().t() + b.float()
        err = (y.float() - ref).abs().max().item()          # (a NaN left in y fails the comparison)
        assert err <= 2e-2 * max(1.0, float(ref.abs().max())), (m, k, n, pos, err)
    big = torch.randn(80001, 384, device=dev()).bfloat16()   # x = a column slice of a packed buffer (pitch 384)
    xs = big[:, 128:384]
    w = (torch.randn(256, 256, device=dev()) * 0.1).bfloat16()
    y = ops.token_gemm(xs, w, None, force=True)
    ref = xs.float() @ w.float().t()
    assert (y.float() - ref).abs().max().item() <= 2e-2 * float(ref.abs().max())
    # y written into a column block of a wider buffer (ldy > n): the neighbours stay untouched
    out = torch.full((40000, 768), 7.0, device=dev(), dtype=torch.bfloat16)
    x = torch.randn(40000, 256, device=dev()).bfloat16()
    w = (torch.randn(256, 256, device=dev()) * 0.1).bfloat16()
    zb = torch.zeros(256, device=dev(), dtype=torch.bfloat16)
    check(lib.tmae_token_gemm(x.data_ptr(), 256, 40000, 256, w.data_ptr(), 256, zb.data_ptr(), out[:, 256:].data_ptr(), 768, st),
          'tmae_token_gemm')
    assert float((out[:, :256].float() - 7).abs().max()) == 0 and float((out[:, 512:].float() - 7).abs().max()) == 0
    ref = x.float() @ w.float().t()
    assert (out[:, 256:512].float() - ref).abs().max().item() <= 2e-2 * float(ref.abs().max())
    # in-place accumulate (FFN-1 input gradient): dx += dy @ w
    # (contraction 768 / 384: the attention in-projections' input gradients; their weight is a column slice of the
    # position-augmented weight, pitch k + 32)
    for (m, n, k) in ((65536 + 33, 512, 256), (120001, 512, 256), (32768 + 5, 256, 128), (470001, 256, 128),
                      (65536 + 17, 768, 256), (200003, 768, 256), (32768 + 9, 384, 128), (150001, 384, 128),
                      (32768 + 41, 256, 256), (147403, 256, 256), (32768 + 3, 128, 128), (94022, 128, 128)):     # square: two column groups / half steps
        dy = torch.randn(m, n, device=dev()).bfloat16()
        w = (torch.randn(n, k + 32, device=dev()) * 0.1).bfloat16()[:, :k] if n in (768, 384) else (torch.randn(n, k, device=dev()) * 0.1).bfloat16()
        dx0 = torch.randn(m, k, device=dev()).bfloat16()
        dx = dx0.clone()
        r = ops.addmm_inplace(dx, dy, w)
        assert r.data_ptr() == dx.data_ptr()
        ref = dx0.float() + dy.float() @ w.float()
        assert (dx.float() - ref).abs().max().item() <= 2e-2 * max(1.0, float(ref.abs().max())), (m, n, k)
    # shapes outside the kernel's table fall back to torch without touching anything else
    dx = torch.randn(1000, 128, device=dev()).bfloat16()
    dy, w = torch.randn(1000, 256, device=dev()).bfloat16(), (torch.randn(256, 128, device=dev()) * 0.1).bfloat16()
    ref = dx.float() + dy.float() @ w.float()
    ops.addmm_inplace(dx, dy, w)
    assert (dx.float() - ref).abs().max().item() <= 4e-2 * float(ref.abs().max())


def _pos_case(m, d, seed):
    """Random tokens on a 468 x 468 grid + the module's position table."""
    from tmae_amd.modules.sst import pos_embed_table
    g = torch.Generator().manual_seed(seed)
    ind = torch.stack([torch.zeros(m, dtype=torch.int64), torch.randint(0, 468, (m,), generator=g),
                       torch.randint(0, 468, (m,), generator=g)], 1).int().to(dev())
    table = pos_embed_table(d, [8, 8, 1], 10000, normalize_pos=bool(seed % 2)).to(dev())     # both forms of the table (NORMALIZE_POS)
    x = torch.randn(m, d, generator=g).to(dev()).bfloat16()
    return ind, table, x


@pytest.mark.parametrize('shift', [False, True])
def test_pos_folded_in_projection_vs_materialised(shift):
    """tmae_token_gemm_pos (position embedding as a one-hot k-step of the in-projection GEMM) vs the fp32 formula
    (x + pos[cell]) W^T + b of WindowAttention.forward (sst_basic_block.py:45-52) on the same bf16 x / W: every
    width the layers use (d = 128: 128 / 256 / 384 rows; d = 256: 256 / 512 / 768 rows, chunk-streaming kernel below
    32 768 tokens and the W-in-registers one above), v rows without position; then the backward of ops.pos_proj (dx
    accumulated into the alias gradient, dW incl. the position part, db) vs autograd of the formula."""
    from tmae_amd import ops
    sh = 4 if shift else 8
    for (m, d, lo, hi, p0, p1) in ((20001, 128, 0, 384, 0, 256), (9000, 128, 0, 128, 0, 128), (30011, 128, 128, 384, 128, 256),
                                   (12345, 256, 0, 768, 0, 512), (70003, 256, 0, 768, 0, 512), (66000, 256, 0, 256, 0, 256),
                                   (131072, 256, 256, 768, 256, 512)):
        ind, table, x = _pos_case(m, d, 7 + m % 5)
        w = (torch.randn(3 * d, d, device=dev()) * 0.1)
        b = torch.randn(3 * d, device=dev())
        cells, onehot = ops.window_cells(ind, [8, 8, 1], shift, want_onehot=True)
        yc, xc = (ind[:, 1].long() + sh) % 8, (ind[:, 2].long() + sh) % 8
        assert torch.equal(cells.long(), xc + 8 * yc)
        assert torch.equal(onehot.float().argmax(1), xc) and torch.equal(onehot[:, 8:].float().argmax(1), yc)
        assert float(onehot.float().sum()) == 2 * m
        E = ops.pos_axes(table, [8, 8, 1])
        pos = table[yc * 8 + xc]                                        # [m, d] f32
        assert torch.allclose(onehot.float() @ E, pos, atol=0, rtol=0)
        wb = w.bfloat16().float()
        xr = x.float().clone().requires_grad_(True)
        wr = wb.clone().requires_grad_(True)
        br = b.bfloat16().float().clone().requires_grad_(True)
        rows = torch.arange(lo, hi, device=dev())
        usep = ((rows >= p0) & (rows < p1)).float()[None, :]
        ref = xr @ wr[lo:hi].t() + (pos @ wr[lo:hi].t()) * usep + br[lo:hi]
        wp = torch.nn.Parameter(w.clone())
        bp = torch.nn.Parameter(b.clone())
        xin = x.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, alias = ops.pos_proj(xin, wp, bp, lo, hi, p0, p1, cells, E, fork=True)
        assert out.shape == (m, hi - lo) and out.dtype == torch.bfloat16
        scale = float(ref.detach().abs().max())
        assert (out.float() - ref).abs().max().item() <= 1e-2 * scale, (m, d, lo, hi)
        gout = torch.randn(m, hi - lo, device=dev()).bfloat16()
        galias = torch.randn(m, d, device=dev()).bfloat16()
        torch.autograd.backward([out, alias], [gout, galias])
        (ref * gout.float()).sum().backward()
        dx_ref = xr.grad + galias.float()
        assert (xin.grad.float() - dx_ref).abs().max().item() <= 2e-2 * float(dx_ref.abs().max())
        for got, want, nm in ((wp.grad, wr.grad, 'dW'), (bp.grad, br.grad, 'db')):
            assert got.shape == want.shape
            err = (got - want).abs().max().item()
            assert err <= 3e-3 * float(want.abs().max()), (nm, m, d, err, float(want.abs().max()))
        if lo > 0:
            assert float(wp.grad[:lo].abs().max()) == 0.0
        # a second call finds the folded weight in the cache, an optimizer-style in-place update drops it
        c0 = wp._tmae_derived
        with torch.autocast('cuda', dtype=torch.bfloat16), torch.no_grad():
            out2 = ops.pos_proj(xin, wp, bp, lo, hi, p0, p1, cells, E)
            assert torch.equal(out2, out) and wp._tmae_derived is c0
            wp.mul_(0.5)
            out3 = ops.pos_proj(xin, wp, bp, lo, hi, p0, p1, cells, E)
        ref3 = (ref - br[lo:hi]) * 0.5 + br[lo:hi]
        assert (out3.float() - ref3).abs().max().item() <= 1e-2 * scale


def test_cross_in_projection_chains_the_previous_frame_gradient():
    """ops.pos_proj_cross(kv_alias=True): the two cross layers of a block (wca_block.py:106-145) read the same previous-frame rows;
    the second takes them through the first one's alias, so its input gradient arrives at the first node and that node's k | v
    input-gradient GEMM accumulates into it (tmae_token_gemm_acc above 32 k / 64 k rows, the library's in-place addmm below) instead
    of autograd adding two [m_prev, d] tensors.  Against the same two nodes reading x_prev independently: identical outputs, the same
    gradient for x_prev up to one bf16 rounding (the chained sum is rounded once, the added one twice), weight gradients equal."""
    from tmae_amd import ops
    for (mq, mk, d) in ((5000, 7001, 128), (20011, 40003, 128), (30000, 70001, 256)):
        indq, table, xq = _pos_case(mq, d, 3)
        indk, _, xk = _pos_case(mk, d, 4)
        cq, ck = ops.window_cells(indq, [8, 8, 1], False), ops.window_cells(indk, [8, 8, 1], False)
        E = ops.pos_axes(table, [8, 8, 1])
        gens = torch.Generator(device=dev()).manual_seed(mq)
        W = [torch.randn(3 * d, d, device=dev(), generator=gens) * 0.1 for _ in range(2)]
        Bv = [torch.randn(3 * d, device=dev(), generator=gens) for _ in range(2)]
        gq = [torch.randn(mq, d, device=dev(), generator=gens).bfloat16() for _ in range(2)]
        gkv = [torch.randn(mk, 2 * d, device=dev(), generator=gens).bfloat16() for _ in range(2)]

        def run(chain):
            wp = [torch.nn.Parameter(w.clone()) for w in W]
            bp = [torch.nn.Parameter(b.clone()) for b in Bv]
            a, k = xq.clone().requires_grad_(True), xk.clone().requires_grad_(True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                r0 = ops.pos_proj_cross(a, k, wp[0], bp[0], cq, ck, E, inplace_dx=True, kv_alias=chain)
                r1 = ops.pos_proj_cross(a, r0[3] if chain else k, wp[1], bp[1], cq, ck, E, inplace_dx=True)
            torch.autograd.backward([r0[0], r0[1], r1[0], r1[1]], [gq[0], gkv[0], gq[1], gkv[1]])
            return (r0[0], r0[1], r1[0], r1[1]), k.grad, [p.grad for p in wp + bp]
        o1, dk1, g1 = run(True)
        o0, dk0, g0 = run(False)
        for u, v in zip(o1, o0):
            assert torch.equal(u, v)
        for u, v in zip(g1, g0):
            assert torch.equal(u, v)
        scale = float(dk0.float().abs().max())
        assert (dk1.float() - dk0.float()).abs().max().item() <= 2.0 ** -7 * scale, (mq, mk, d)
        # and against fp32: d x_prev = sum over the two layers of dkv W[d:3d] (the position rows add nothing to dx)
        ref = sum(gkv[j].float() @ W[j][d:].bfloat16().float() for j in range(2))
        assert (dk1.float() - ref).abs().max().item() <= 2e-2 * float(ref.abs().max())


def test_refresh_param_copies_casts_and_transposes_in_one_launch():
    """ops.refresh_param_copies: after a weight's transposed bf16 copy was used once (the dX GEMM of ops.linear), the
    refresh writes the bf16 copy AND its transpose with tmae_multi_cast_transpose; three optimizer-style updates of
    matrices with ragged sizes, results against plain casts; the backward keeps using the refreshed transposes."""
    from tmae_amd import ops
    torch.manual_seed(1)
    ws = [torch.nn.Parameter(torch.randn(n, k, device=dev()) * 0.1) for n, k in ((256, 128), (128, 256), (512, 256), (200, 72))]
    x = [torch.randn(9000, w.shape[1], device=dev()).bfloat16().requires_grad_(True) for w in ws]
    for step in range(4):
        for w, xi in zip(ws, x):
            w.grad = None
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = ops.linear(xi, w, None)
            y.sum().backward()
            ref = (torch.ones_like(y).float() @ w.detach().bfloat16().float()).bfloat16()
            assert torch.allclose(xi.grad.float(), ref.float(), rtol=2e-2, atol=2e-2), step
            xi.grad = None
        with torch.no_grad():
            for w in ws:
                w.add_(torch.randn_like(w) * 0.01)                   # the optimizer step
        ops.refresh_param_copies(ws)
        for w in ws:
            c = w._tmae_copy
            assert c[0] == w._version and torch.equal(c[2], w.detach().bfloat16())
            T = getattr(c[2], '_tmae_T', None)
            if step >= 1 and w.shape[0] in (128, 256) and w.shape[1] % 64 == 0:   # shapes whose dX runs on the token GEMM (on W^T)
                assert T is not None and T[0] == c[2]._tmae_stamp
                assert torch.equal(T[1], w.detach().bfloat16().t().contiguous())


def test_gelu_linear_fused_backward_vs_torch():
    """ops.gelu_linear: forward = F.linear(F.gelu(x)); backward with the GELU derivative fused into the dX GEMM
    (tmae_token_gemm_dgelu) vs torch autograd in bf16, for both FFN shapes (d = 128 / dff = 256, d = 256 / dff = 512);
    token counts below and above 32768 (above: the W-in-registers kernel with h_pre in its LDS ring; ragged last step)."""
    from tmae_amd import ops
    torch.manual_seed(4)
    for (m, dff, d) in ((30001, 256, 128), (20000, 512, 256), (40003, 256, 128), (33001, 512, 256)):
        hp = (torch.randn(m, dff, device=dev()) * 1.5).bfloat16().requires_grad_(True)
        w = (torch.randn(d, dff, device=dev()) * 0.05).requires_grad_(True)
        b = torch.randn(d, device=dev(), requires_grad=True)
        go = torch.randn(m, d, device=dev()).bfloat16()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = ops.gelu_linear(hp, w, b)
        y.backward(go)
        g1 = (hp.grad.clone(), w.grad.clone(), b.grad.clone())
        hp.grad = w.grad = b.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y2 = F.linear(F.gelu(hp), w, b)
        y2.backward(go)
        assert (y.float() - y2.float()).abs().max().item() <= 4e-2 * max(1.0, float(y2.float().abs().max()))
        ref = hp.grad.float()
        assert (g1[0].float() - ref).abs().max().item() <= 2e-2 * max(1.0, float(ref.abs().max())), (m, dff, d)
        assert ((g1[0].float() - ref).norm() / ref.norm()).item() < 6e-3
        assert (g1[1] - w.grad).abs().max().item() <= 1e-2 * float(w.grad.abs().max())
        assert (g1[2] - b.grad).abs().max().item() <= 1e-2 * float(b.grad.abs().max())


def test_gelu_linear_with_residual_vs_torch():
    """ops.gelu_linear(..., residual=r) = r + F.linear(F.gelu(x)) -- `src + linear2(act)` of the encoder layer
    (sst_basic_block.py:81-83) out of linear2's GEMM (tmae_token_gemm_res: the residual tile rides in the kernel's LDS ring) --
    against fp32 torch on the same bf16 values: both FFN shapes above the kernel's size limit (ragged last step), one below it
    (the fallback: GEMM + add); the residual's gradient is the output gradient itself; refusals of the C entry point."""
    from tmae_amd import ops
    from tmae_amd._lib import lib
    torch.manual_seed(14)
    for (m, dff, d) in ((40003, 256, 128), (70001, 512, 256), (20000, 512, 256)):
        hp = (torch.randn(m, dff, device=dev()) * 1.5).bfloat16().requires_grad_(True)
        r = torch.randn(m, d, device=dev()).bfloat16().requires_grad_(True)
        w = (torch.randn(d, dff, device=dev()) * 0.05).bfloat16().float().requires_grad_(True)
        b = torch.randn(d, device=dev()).bfloat16().float().requires_grad_(True)
        go = torch.randn(m, d, device=dev()).bfloat16()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = ops.gelu_linear(hp, w, b, residual=r)
        y.backward(go)
        assert torch.equal(r.grad, go)
        g1 = (hp.grad.float().clone(), w.grad.clone(), b.grad.clone())
        hp.grad = w.grad = b.grad = r.grad = None
        h = F.gelu(hp.detach()).float()                               # the kernel's left operand: gelu in bf16, as the producer stores it
        want = r.detach().float() + h @ w.detach().t() + b.detach()
        err = (y.float() - want).abs()
        fused = m >= (65536 if dff == 512 else 32768)                  # ONE rounding of the sum; the fallback (GEMM, then add) has two
        assert bool((err <= 2.0 ** (-8 if fused else -7) * want.abs() + (2e-3 if fused else 2e-2)).all()), (m, dff, d, float(err.max()))
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y2 = F.linear(F.gelu(hp), w, b) + r
        y2.backward(go)
        ref = hp.grad.float()
        assert ((g1[0] - ref).norm() / ref.norm()).item() < 6e-3
        assert (g1[1] - w.grad).abs().max().item() <= 1e-2 * float(w.grad.abs().max())
        assert (g1[2] - b.grad).abs().max().item() <= 1e-2 * float(b.grad.abs().max())
    x = torch.randn(40000, 512, device=dev()).bfloat16()
    w = torch.randn(256, 512, device=dev()).bfloat16()
    bz = torch.zeros(256, device=dev()).bfloat16()
    r = torch.randn(40000, 256, device=dev()).bfloat16()
    y = torch.empty_like(r)
    st = torch.cuda.current_stream().cuda_stream
    args = lambda **kw: [kw.get('x', x).data_ptr(), 512, kw.get('m', 40000), kw.get('k', 512), w.data_ptr(), kw.get('n', 256), bz.data_ptr(),
                         kw.get('r', r.data_ptr()), y.data_ptr(), 256, st]
    assert lib.tmae_token_gemm_res(*args()) == 0
    assert lib.tmae_token_gemm_res(*args(m=1000)) < 0 and lib.tmae_token_gemm_res(*args(k=256)) < 0
    assert lib.tmae_token_gemm_res(*args(r=None)) < 0 and lib.tmae_token_gemm_res(*args(r=r.data_ptr() + 2)) < 0
    torch.cuda.synchronize()


def test_add_layernorm_kernel_vs_torch():
    from tmae_amd import ops
    torch.manual_seed(1)
    for d in (128, 256):
        for m in (1, 777, 40000):
            a = torch.randn(m, d, device=dev())
            b = torch.randn(m, d, device=dev()) * 0.5
            w = (1 + 0.1 * torch.randn(d, device=dev())).requires_grad_(True)
            bb = (0.1 * torch.randn(d, device=dev())).requires_grad_(True)
            go = torch.randn(m, d, device=dev())
            for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 6e-2)):
                a1, b1 = a.to(dt).clone().requires_grad_(True), b.to(dt).clone().requires_grad_(True)
                y = ops.add_layer_norm(a1, b1, w, bb, 1e-5)
                y.backward(go.to(dt))
                g_mine = (a1.grad.float().clone(), b1.grad.float().clone(), w.grad.clone(), bb.grad.clone())
                w.grad = bb.grad = None
                a2, b2 = a.to(dt).float().clone().requires_grad_(True), b.to(dt).float().clone().requires_grad_(True)
                xs = (a2 + b2) if dt == torch.float32 else (a2 + b2).to(dt).float()
                y2 = F.layer_norm(xs, (d,), w, bb, 1e-5)
                y2.backward(go.to(dt).float())
                assert (y.float() - y2).abs().max().item() <= tol, (d, m, dt)
                assert (g_mine[0] - a2.grad).abs().max().item() <= tol * 4
                assert torch.equal(g_mine[0], g_mine[1])
                sc = max(1.0, float(w.grad.abs().max()))
                assert (g_mine[2] - w.grad).abs().max().item() <= (1e-3 if dt == torch.float32 else 3e-2) * sc * max(1, m ** 0.5 / 10)
                assert (g_mine[3] - bb.grad).abs().max().item() <= (1e-3 if dt == torch.float32 else 3e-2) * sc * max(1, m ** 0.5 / 10)
                w.grad = bb.grad = None
    y = ops.add_layer_norm(a, None, w, bb, 1e-5)
    assert (y - F.layer_norm(a, (256,), w, bb, 1e-5)).abs().max().item() < 2e-5


def test_add_layernorm_residual_options_vs_torch():
    """The round-3 options of the fused add + LayerNorm: a 0/1 row mask on the second summand (cross layers), an
    alias of the first summand handed out (passthrough) and added to the output of a LATER norm (post) -- the block
    residual x + encoder(x) -- against the same graph written with torch ops: outputs and every gradient."""
    from tmae_amd import ops
    torch.manual_seed(3)
    for d in (128, 256):
        for m in (5, 4099):
            for dt, tol in ((torch.float32, 3e-5), (torch.bfloat16, 8e-2)):
                a = torch.randn(m, d, device=dev()).to(dt)
                b = (torch.randn(m, d, device=dev()) * 0.5).to(dt)
                c = (torch.randn(m, d, device=dev()) * 0.5).to(dt)
                keep = (torch.rand(m, 1, device=dev()) < 0.6).to(dt)
                w1 = (1 + 0.1 * torch.randn(d, device=dev())).requires_grad_(True)
                b1 = (0.1 * torch.randn(d, device=dev())).requires_grad_(True)
                w2 = (1 + 0.1 * torch.randn(d, device=dev())).requires_grad_(True)
                b2 = (0.1 * torch.randn(d, device=dev())).requires_grad_(True)
                go = torch.randn(m, d, device=dev()).to(dt)

                def run(mine):
                    aa, bb_, cc = (t.clone().requires_grad_(True) for t in (a, b, c))
                    if mine:
                        h, alias = ops.add_layer_norm(aa, bb_, w1, b1, 1e-5, bmask=keep, passthrough=True)
                        y = ops.add_layer_norm(h, cc, w2, b2, 1e-5, post=alias)
                    else:
                        af, bf, cf = aa.float(), bb_.float(), cc.float()
                        x1 = af + bf * keep.float()
                        x1 = x1 if dt == torch.float32 else x1.to(dt).float()
                        h = F.layer_norm(x1, (d,), w1, b1, 1e-5)
                        h = h if dt == torch.float32 else h.to(dt).float()
                        x2 = h + cf
                        x2 = x2 if dt == torch.float32 else x2.to(dt).float()
                        y = F.layer_norm(x2, (d,), w2, b2, 1e-5) + af
                    y.backward(go.to(y.dtype))
                    out = (y.detach().float(), aa.grad.float(), bb_.grad.float(), cc.grad.float(), w1.grad.clone(),
                           b1.grad.clone(), w2.grad.clone(), b2.grad.clone())
                    w1.grad = b1.grad = w2.grad = b2.grad = None
                    return out
                got, ref = run(True), run(False)
                for k, (x_, r_) in enumerate(zip(got, ref)):
                    sc = max(1.0, float(r_.abs().max()))
                    lim = tol * (4 if k in (1, 2, 3) else 1) * sc if k < 4 else (2e-3 if dt == torch.float32 else 5e-2) * sc * max(1, m ** 0.5 / 10)
                    assert (x_ - r_).abs().max().item() <= lim, (d, m, dt, k, (x_ - r_).abs().max().item(), lim)
                assert float((got[2] * (1 - keep.float())).abs().max()) == 0.0        # masked rows of b get no gradient


def test_dense_conv_epilogue_sums_and_their_two_consumers():
    """tmae_dense_conv3x3_sums: the decoder conv's two heavy launches with the column sums of their OUTPUT from the epilogue.
    (1) operator level: outputs bit-equal to tmae_dense_conv3x3(_add), sums equal to fp64 sums of the stored bf16 values to fp32
    accuracy, on a grid with ragged 16 x 16 tiles; (2) the input gradient carries its sums behind its data (ops.colsum_tail) and
    only there; (3) model level: one decoder pass (deblocks -> conv -> norm + gather) with and without the epilogue sums gives the
    same outputs and gradients (the statistics of the last norm come from the conv's moments, the column sum of the concat
    gradient from its tail)."""
    from tmae_amd import ops
    from tmae_amd._lib import lib, check
    torch.manual_seed(8)
    B, Y, X = 2, 41, 50
    st = torch.cuda.current_stream().cuda_stream
    for cin, cout, mom in ((384, 128, 2), (128, 384, 1)):
        x = torch.randn(B, Y, X, cin, device=dev()).bfloat16()
        w = (torch.randn(cout, 9 * cin, device=dev()) * 0.03).bfloat16()
        post = torch.randn(B, Y, X, cout, device=dev()).bfloat16() if mom == 1 else None
        ref = ops.dense_conv3x3_halo(x, w, 1, post=post)
        y = torch.empty_like(ref)
        sums = torch.empty(mom, cout, device=dev())
        wsb = lib.tmae_dense_conv3x3_sums_workspace(cout)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
        check(lib.tmae_dense_conv3x3_sums(x.data_ptr(), B, Y, X, cin, w.data_ptr(), cout, None if post is None else post.data_ptr(), mom,
                                          y.data_ptr(), sums.data_ptr(), ws.data_ptr(), wsb, st), 'sums')
        assert torch.equal(y, ref)
        rows = ref.view(-1, cout).double()
        want = [rows.sum(0)] + ([(rows * rows).sum(0)] if mom == 2 else [])
        for got, w_ in zip(sums, want):
            assert (got.double() - w_).abs().max().item() <= 2e-5 * max(1.0, float(w_.abs().max())) + 1e-3, (cin, cout)
        # refusals: other shapes, a missing sums pointer
        assert lib.tmae_dense_conv3x3_sums(x.data_ptr(), B, Y, X, cin, w.data_ptr(), cout, None, 3 - mom, y.data_ptr(), sums.data_ptr(),
                                           ws.data_ptr(), wsb, st) != 0
    x = torch.randn(B, Y, X, 128, device=dev()).bfloat16()
    w = (torch.randn(384, 9 * 128, device=dev()) * 0.03).bfloat16()
    y = ops.dense_conv3x3_halo(x, w, 1, tail_sums=True)
    tail = ops.colsum_tail(y, 384)
    assert tail is not None and torch.allclose(tail.double(), y.view(-1, 384).double().sum(0), rtol=2e-5, atol=1e-3)
    assert ops.colsum_tail(y.clone(), 384) is None and ops.colsum_tail(ops.dense_conv3x3_halo(x, w, 1), 384) is None
    assert ops.colsum_tail(y.permute(0, 3, 1, 2).permute(0, 2, 3, 1), 384) is not None        # the views autograd hands on


def test_decoder_with_and_without_epilogue_sums(oracle):
    """The whole model step (F10 golden case, bf16 autocast) with TMAE_DENSE_SUMS on and off: same loss to 1e-4, same decoder
    gradients to bf16 accuracy -- the conv-epilogue moments / tail sums replace passes, not results."""
    from tmae_amd import ops
    g = golden('F10_e2e_3stage')
    cfg = oracle.default_model_cfg(3)
    P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']), pred_scale=float(g['pred_scale']))
    res = []
    for on in (True, False):
        saved = ops._DENSE_SUMS
        ops._DENSE_SUMS = on
        try:
            model, _, _ = build_product_model(3, params=P, device=dev())
            model.train()
            loss, _ = _run_product(model, g['points'], g['points_prev'], g['noise'], int(g['batch_size']), amp=True)
            res.append((fl(loss), {n: p.grad.float().clone() for n, p in model.named_parameters() if 'decoder' in n},
                        {n: b.clone() for n, b in model.named_buffers() if 'decoder_conv_out' in n and 'running' in n}))
        finally:
            ops._DENSE_SUMS = saved
    (la, ga, ba), (lb, gb, bb) = res
    assert abs(la - lb) < 1e-4, (la, lb)
    for n in gb:
        assert (ga[n] - gb[n]).abs().max().item() <= 2e-2 * max(1e-3, float(gb[n].abs().max())), n
    for n in bb:
        assert torch.allclose(ba[n], bb[n], rtol=1e-4, atol=1e-5), n


def test_bn_relu_scatter_max_equals_the_two_ops():
    """ops.bn_relu_scatter_max (the VFE's last BatchNorm + ReLU applied by the voxel max as it reads the rows,
    tmae_segment_max_bn_fwd) against ops.batch_norm_relu followed by ops.scatter_max: values, argmax, running statistics and
    all gradients bit for bit (the fused kernel rounds the normalised value exactly as the separate pass stores it)."""
    from tmae_amd import ops
    torch.manual_seed(9)
    n, m = 60000, 21000
    inv = torch.randint(0, m, (n,), device=dev()).sort().values          # rows in voxel order (perm = None), some voxels empty
    perm_, offs = ops.segment_csr(inv, m)
    for c, dt in ((128, torch.bfloat16), (64, torch.float32), (256, torch.bfloat16)):
        x0 = (torch.randn(n, c, device=dev()) * 1.7 + 0.4).to(dt)
        go = torch.randn(m, c, device=dev()).to(dt)
        res = []
        for fused in (True, False):
            bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev())
            with torch.no_grad():
                bn.weight.normal_(1.0, 0.3), bn.bias.normal_(0.0, 0.3)
                bn.weight[0] = -0.7                                              # a negative scale: the max is taken AFTER the affine map
            if len(res):
                bn.load_state_dict(keep)
            keep = {k: v.clone() for k, v in bn.state_dict().items()}
            x = x0.clone().requires_grad_(True)
            if fused:
                out, arg = ops.bn_relu_scatter_max(x, bn, inv, None, offs, m)
            else:
                out, arg = ops.scatter_max(ops.batch_norm_relu(x, bn, relu=True), inv, None, offs, m)
            out.backward(go)
            res.append((out.detach().clone(), arg.clone(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                        bn.running_mean.clone(), bn.running_var.clone()))
        for k, (a, b) in enumerate(zip(*res)):
            assert torch.equal(a, b), (c, dt, k)


def test_batchnorm_gather_backward_without_the_dense_gradient():
    """ops.batch_norm_relu_gather (the decoder's last norm + the gather at the current frame's voxels as one node,
    tmae_bn_relu_bwd_gathered) against ops.batch_norm_relu followed by ops.dense_gather: same outputs bit for bit; dx / dgamma /
    dbeta equal whether the gradient arrives at the gathered rows only (the pre-training step), at the dense map too, or at the
    dense map alone."""
    from tmae_amd import ops
    torch.manual_seed(6)
    B, ny, nx = 2, 37, 53
    for c, dt, tol in ((128, torch.float32, 2e-5), (128, torch.bfloat16, 2e-2), (64, torch.bfloat16, 2e-2)):
        cells = B * ny * nx
        x0 = (torch.randn(cells, c, device=dev()) * 1.4 - 0.3).to(dt)
        pick = torch.randperm(cells, device=dev())[:cells // 5].sort().values
        ind = torch.stack([pick // (ny * nx), (pick // nx) % ny, pick % nx], 1).int().contiguous()
        rowmap = ops.index_grid(ind, B, ny, nx)
        g_rows = torch.randn(ind.shape[0], c, device=dev()).to(dt)
        g_map = torch.randn(cells, c, device=dev()).to(dt)
        for use_rows, use_map in ((True, False), (True, True), (False, True)):
            res = []
            for fused in (True, False):
                bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev())
                with torch.no_grad():
                    bn.weight.fill_(0.9), bn.bias.fill_(0.05)
                x = x0.clone().requires_grad_(True)
                if fused:
                    y, rows = ops.batch_norm_relu_gather(x, bn, True, rowmap, ind, B, ny, nx)
                else:
                    y = ops.batch_norm_relu(x, bn, relu=True)
                    rows = ops.dense_gather(y.view(B, ny, nx, c), rowmap, ind)
                outs, gs = [], []
                if use_rows:
                    outs.append(rows), gs.append(g_rows)
                if use_map:
                    outs.append(y), gs.append(g_map)
                torch.autograd.backward(outs, gs)
                res.append((y.detach().float(), rows.detach().float(), x.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                            bn.running_mean.clone(), bn.running_var.clone()))
            a, b = res
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            assert torch.equal(a[5], b[5]) and torch.equal(a[6], b[6])
            sc = max(1.0, float(b[2].abs().max()))
            assert (a[2] - b[2]).abs().max().item() <= tol * sc, (c, dt, use_rows, use_map, (a[2] - b[2]).abs().max().item())
            for k in (3, 4):
                assert (a[k] - b[k]).abs().max().item() <= max(tol, 2e-3) * max(1.0, float(b[k].abs().max())), (c, dt, k)


def test_ffn_residual_out_of_the_gemm_vs_added_in_the_norm():
    """_EncoderTail with `src + linear2(act)` written by linear2's GEMM (tmae_token_gemm_res, the default) against the same tail
    with the sum taken inside norm2 from two tensors (TMAE_FFN_RESIDUAL=add): the fused form rounds the residual
    stream to bf16 once more per layer (2^-9 relative on the norm's INPUT).  Stated tolerance, bf16 autocast, 70 k tokens, d = 256:
    norm2 output 2^-6 absolute on unit-variance rows (two bf16 roundings of a normalised value), input gradient and the FFN weight
    gradients 2 % of their largest element."""
    from tmae_amd.modules import sst
    torch.manual_seed(11)
    d, m = 256, 70000
    layer = sst._EncoderTail(torch.nn.Identity(), d, 2 * d, 'gelu').to(dev())
    src0 = torch.randn(m, d, device=dev())
    attn0 = torch.randn(m, d, device=dev()) * 0.5
    go = torch.randn(m, d, device=dev())
    res = []
    for fused in (True, False):
        saved = sst._FFN_RESIDUAL_FUSED
        sst._FFN_RESIDUAL_FUSED = fused
        try:
            layer.zero_grad()
            src = src0.clone().requires_grad_(True)
            attn = attn0.clone().requires_grad_(True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out = layer.tail(src, attn)
            out.backward(go.to(out.dtype))
            res.append((out.detach().float(), src.grad.float(), attn.grad.float(), layer.linear1.weight.grad.float().clone(),
                        layer.linear2.weight.grad.float().clone(), layer.norm2.weight.grad.float().clone()))
        finally:
            sst._FFN_RESIDUAL_FUSED = saved
    a, b = res
    assert (a[0] - b[0]).abs().max().item() <= 2.0 ** -6 * max(1.0, float(b[0].abs().max())), (a[0] - b[0]).abs().max().item()
    assert (a[0] - b[0]).abs().mean().item() <= 2e-3
    for k in (1, 2, 3, 4, 5):
        lim = 2e-2 * max(1.0, float(b[k].abs().max()))
        assert (a[k] - b[k]).abs().max().item() <= lim, (k, (a[k] - b[k]).abs().max().item(), lim)


def test_batchnorm_fork_joins_the_frame_gradients_inside_its_backward():
    """ops.batch_norm_relu(..., groups=[m0, m1], fork=True): y and its two row ranges as outputs of ONE autograd node; whichever
    of the three receive a gradient, dx / dgamma / dbeta equal those of the unforked norm followed by ops.split_rows (autograd's
    cat + add), up to the one bf16 rounding that path spends on the sum (tmae_bn_relu_bwd2 adds in fp32)."""
    from tmae_amd import ops
    torch.manual_seed(4)
    for c, dt, tol in ((128, torch.float32, 1e-5), (256, torch.bfloat16, 2e-2), (64, torch.bfloat16, 2e-2)):
        m0, m1 = 7001, 3999
        x = (torch.randn(m0 + m1, c, device=dev()) * 1.3 + 0.2).to(dt)
        ga, g0, g1 = (torch.randn(n, c, device=dev()).to(dt) for n in (m0 + m1, m0, m1))
        for use in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 0, 0), (0, 1, 0)):
            res = []
            for fork in (True, False):
                bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev())
                with torch.no_grad():
                    bn.weight.fill_(1.1), bn.bias.fill_(-0.1)
                xi = x.clone().requires_grad_(True)
                if fork:
                    y, (h0, h1) = ops.batch_norm_relu(xi, bn, relu=True, groups=[m0, m1], fork=True)
                else:
                    y = ops.batch_norm_relu(xi, bn, relu=True, groups=[m0, m1])
                    h0, h1 = ops.split_rows(y, m0)
                assert h0.shape == (m0, c) and h1.shape == (m1, c) and torch.equal(h0, y[:m0]) and torch.equal(h1, y[m0:])
                outs, gs = [], []
                for flag, o, g_ in zip(use, (y, h0, h1), (ga, g0, g1)):
                    if flag:
                        outs.append(o), gs.append(g_)
                torch.autograd.backward(outs, gs)
                res.append((y.detach().float(), xi.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone()))
            (ya, dxa, dga, dba), (yb, dxb, dgb, dbb) = res
            assert torch.equal(ya, yb)
            sc = max(1.0, float(dxb.abs().max()))
            assert (dxa - dxb).abs().max().item() <= tol * sc, (c, dt, use, (dxa - dxb).abs().max().item())
            for a, b in ((dga, dgb), (dba, dbb)):
                assert (a - b).abs().max().item() <= max(tol, 2e-3) * max(1.0, float(b.abs().max())), (c, dt, use)


def test_batchnorm_relu_kernel_vs_torch():
    from tmae_amd import ops
    torch.manual_seed(2)
    for c in (64, 128, 256):
        for m in (2, 999, 60000):
            x = torch.randn(m, c, device=dev()) * 1.5 + 0.3
            go = torch.randn(m, c, device=dev())
            for dt, tol in ((torch.float32, 1e-4), (torch.bfloat16, 4e-2)):
                bn1 = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev())
                bn2 = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev())
                with torch.no_grad():
                    bn1.weight.normal_(1, 0.2), bn1.bias.normal_(0, 0.2)
                    bn2.load_state_dict(bn1.state_dict())
                x1 = x.to(dt).clone().requires_grad_(True)
                y1 = ops.batch_norm_relu(x1, bn1, relu=True)
                y1.backward(go.to(dt))
                x2 = x.to(dt).float().clone().requires_grad_(True)
                y2 = torch.relu(bn2(x2))
                y2.backward(go.to(dt).float())
                assert (y1.float() - y2).abs().max().item() <= tol, (c, m, dt)
                assert (x1.grad.float() - x2.grad).abs().max().item() <= tol * 3 * max(1.0, float(x2.grad.abs().max())), (c, m, dt)
                sc = max(1.0, float(bn2.weight.grad.abs().max()))
                lim = (2e-3 if dt == torch.float32 else 5e-2) * sc * max(1.0, m ** 0.5 / 20)
                assert (bn1.weight.grad - bn2.weight.grad).abs().max().item() <= lim
                assert (bn1.bias.grad - bn2.bias.grad).abs().max().item() <= lim
                assert (bn1.running_mean - bn2.running_mean).abs().max().item() < 1e-4
                assert (bn1.running_var - bn2.running_var).abs().max().item() < 1e-3
                assert int(bn1.num_batches_tracked) == 1
    # running statistics queued inside defer_bn_updates() (what the detectors' forward does) = the immediate updates:
    # a module used three times (two row groups, then once more), another one once
    bn_a = [torch.nn.BatchNorm1d(128, eps=1e-3, momentum=0.01).to(dev()) for _ in range(2)]
    bn_b = [torch.nn.BatchNorm1d(64, eps=1e-3, momentum=0.01).to(dev()) for _ in range(2)]
    xa, xb = torch.randn(5000, 128, device=dev()) * 2 + 1, torch.randn(700, 64, device=dev())
    def run(i):
        ops.batch_norm_relu(xa, bn_a[i], relu=True, groups=[3000, 2000])
        ops.batch_norm_relu(xb, bn_b[i], relu=False)
        ops.batch_norm_relu(xa * 0.5, bn_a[i], relu=True)
    run(0)
    with ops.defer_bn_updates():
        run(1)
        assert int(bn_a[1].num_batches_tracked) == 0          # still queued
    for u, v in ((bn_a[0], bn_a[1]), (bn_b[0], bn_b[1])):
        assert torch.allclose(u.running_mean, v.running_mean, rtol=1e-6, atol=1e-7)
        assert torch.allclose(u.running_var, v.running_var, rtol=1e-6, atol=1e-7)
        assert int(u.num_batches_tracked) == int(v.num_batches_tracked) > 0


def test_conv_bias_before_batchnorm_is_folded_exactly():
    """USE_BIAS_BEFORE_NORM (center_head.py:19-27): conv(bias) -> BatchNorm2d(train) -> ReLU through
    modules.bev_backbone.conv_bn_relu_nhwc (bias folded away: ops.batch_norm_relu pre_bias) against the plain torch
    modules: output, input / weight gradients, a zero bias gradient where torch's is rounding noise, running statistics."""
    import copy
    from tmae_amd.modules.bev_backbone import conv_bn_relu_nhwc
    torch.manual_seed(2)
    seq = torch.nn.Sequential(torch.nn.Conv2d(32, 64, 3, padding=1, bias=True), torch.nn.BatchNorm2d(64, eps=1e-3, momentum=0.01),
                              torch.nn.ReLU()).to(dev()).train()
    seq[0].bias.data.normal_(0, 2.0)
    ref = copy.deepcopy(seq)
    x = torch.randn(2, 32, 40, 36, device=dev()).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(2, 64, 40, 36, device=dev()).contiguous(memory_format=torch.channels_last)
    ya = conv_bn_relu_nhwc(seq, xa)
    yb = ref(xb)
    ya.backward(g)
    yb.backward(g)
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-3, atol=1e-4)
    assert torch.allclose(seq[0].weight.grad, ref[0].weight.grad, rtol=1e-3, atol=1e-3)
    assert torch.allclose(seq[1].weight.grad, ref[1].weight.grad, rtol=1e-3, atol=1e-3)
    assert float(seq[0].bias.grad.abs().max()) == 0.0 and float(ref[0].bias.grad.abs().max()) < 1e-2
    assert torch.allclose(seq[1].running_mean, ref[1].running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(seq[1].running_var, ref[1].running_var, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------ A9 / A11

def test_fused_decoder_head_vs_torch_dense():
    """A11: dense() -> ConvTranspose2d -> BatchNorm2d -> ReLU -> cat (SiamWCA_MAE.py:231-250) against the dense torch
    modules, forward, all gradients and running statistics."""
    from tmae_amd import ops
    from tmae_amd.modules.sparse import SparseConvTensor
    torch.manual_seed(5)
    B, Y, X = 2, 24, 16
    specs = [(1, 32, 128), (2, 48, 128), (4, 64, 64)]              # (stride, cin, cout)
    for dt, tol in ((torch.float32, 2e-4), (torch.bfloat16, 6e-2)):
        sources, dense_in, mods = [], [], []
        for s, cin, cout in specs:
            ys, xs = Y // s, X // s
            occ = torch.rand(B, ys, xs, device=dev()) < 0.3
            occ[0, 0, 0] = True
            ind = occ.nonzero().int().contiguous()
            feat = (torch.randn(ind.shape[0], cin, device=dev())).to(dt).requires_grad_(True)
            sp = SparseConvTensor(feat, ind, [ys, xs], B)
            deconv = torch.nn.ConvTranspose2d(cin, cout, s, stride=s, bias=False).to(dev())
            bn = torch.nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(dev())
            with torch.no_grad():
                bn.weight.normal_(1, 0.2), bn.bias.normal_(0, 0.3)
                deconv.weight.mul_(6.0)                    # v of unit scale: keeps rstd (the gradient gain) O(1)
            deconv2 = torch.nn.ConvTranspose2d(cin, cout, s, stride=s, bias=False).to(dev())
            bn2 = torch.nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(dev())
            deconv2.load_state_dict(deconv.state_dict()), bn2.load_state_dict(bn.state_dict())
            sources.append((sp.features, sp.grid, sp.indices, sp.spatial_shape, deconv, bn))
            f2 = feat.detach().float().clone().requires_grad_(True)
            d = torch.zeros(B, ys, xs, cin, device=dev())
            d = d.index_put((ind[:, 0].long(), ind[:, 1].long(), ind[:, 2].long()), f2)
            dense_in.append(f2)
            mods.append((deconv2, bn2, d.permute(0, 3, 1, 2)))
        assert ops.deblocks_fusable(sources, True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
            cat = ops.deblocks_to_dense(sources, B, Y, X)
        ref = torch.cat([torch.relu(bn2(deconv2(d))) for deconv2, bn2, d in mods], dim=1).permute(0, 2, 3, 1)
        assert cat.shape == ref.shape and cat.dtype == dt
        assert (cat.float() - ref).abs().max().item() <= tol
        go = torch.randn_like(ref)
        cat.backward(go.to(dt))
        ref.backward(go.to(dt).float())
        for (feat, _, _, _, deconv, bn), f2, (deconv2, bn2, _) in zip(sources, dense_in, mods):
            for a, b, name in ((feat.grad, f2.grad, 'feat'), (deconv.weight.grad, deconv2.weight.grad, 'w'),
                               (bn.weight.grad, bn2.weight.grad, 'gamma'), (bn.bias.grad, bn2.bias.grad, 'beta')):
                if dt == torch.float32:
                    lim = tol * 3 * max(1.0, float(b.abs().max()))
                    assert (a - b).abs().max().item() <= lim, (name, (a - b).abs().max().item(), lim)
                else:       # bf16: a ReLU-mask flip of a near-zero pre-activation moves single entries; bound the norm
                    rel = ((a.float() - b).norm() / b.norm()).item()
                    assert rel <= 6e-2, (name, rel)
            assert (bn.running_mean - bn2.running_mean).abs().max().item() < 1e-4
            assert (bn.running_var - bn2.running_var).abs().max().item() < 1e-3


@pytest.mark.parametrize('dil,cin,cout', [(1, 128, 128), (2, 128, 128), (1, 384, 128), (2, 256, 256)])
def test_dense_conv3x3_own_wgrad_vs_torch(dil, cin, cout):
    """ops.dense_conv3x3 (halo-tiled implicit GEMM forward / dX, token-split weight gradient through the dense rulebook)
    against torch's Conv2d in fp32 on the same bf16-representable data: y, dX, dW; dilation 1 (decoder conv,
    SiamWCA_MAE.py:100-115) and 2 (SSTBEVBackbone, sst_bev_backbone.py:20-30), grid sizes that are no multiple of 16."""
    from tmae_amd import ops
    torch.manual_seed(3)
    B, Y, X = 2, 52, 44
    x = torch.randn(B, Y, X, cin, device=dev()).bfloat16()
    w = (torch.randn(cout, cin, 3, 3, device=dev()) * 0.05).bfloat16().float().requires_grad_(True)
    gy = torch.randn(B, Y, X, cout, device=dev()).bfloat16()
    xa = x.clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = ops.dense_conv3x3(xa, w, dil)
    y.backward(gy)
    xr = x.float().requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, padding=dil, dilation=dil).permute(0, 2, 3, 1)
    yr.backward(gy.float())

    def rel(a, b):
        return float((a.float() - b.float()).norm() / b.float().norm())
    assert rel(y, yr) < 1e-2 and rel(xa.grad, xr.grad) < 1e-2
    assert rel(w.grad, wr.grad) < 5e-3, rel(w.grad, wr.grad)
    # borders included: the rulebook's -1 entries are the zero padding
    assert (w.grad - wr.grad).abs().max().item() < 2e-2 * wr.grad.abs().max().item()


@pytest.mark.parametrize('dil,cin,shape', [(1, 64, (1, 7, 5)), (1, 384, (2, 33, 47)), (2, 128, (3, 16, 16)), (2, 256, (1, 40, 21))])
def test_dense_conv3x3_wgrad_c_abi_vs_torch(dil, cin, shape):
    """tmae_dense_conv3x3_wgrad (halo-tiled weight gradient, csrc/dense_wgrad.hip) through the C ABI against torch's conv2d
    weight gradient in fp32 on the same bf16-representable data: grids smaller than a tile, partial tiles, both dilations,
    channel counts 64 .. 384; deterministic (two calls bit-identical); argument errors."""
    from tmae_amd._lib import lib, check
    torch.manual_seed(11)
    B, Y, X = shape
    cout = 128
    x = torch.randn(B, Y, X, cin, device=dev()).bfloat16()
    dy = torch.randn(B, Y, X, cout, device=dev()).bfloat16()
    wsb = lib.tmae_dense_conv3x3_wgrad_workspace(cin, cout)
    assert wsb > 0 and lib.tmae_dense_conv3x3_wgrad_workspace(cin, 256) == 0 and lib.tmae_dense_conv3x3_wgrad_workspace(96, cout) == 0
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(2):
        dw = torch.full((cout, 9 * cin), float('nan'), device=dev())
        check(lib.tmae_dense_conv3x3_wgrad(dy.data_ptr(), x.data_ptr(), B, Y, X, cin, cout, dil, dw.data_ptr(), ws.data_ptr(), wsb, st),
              'tmae_dense_conv3x3_wgrad')
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    w = torch.zeros(cout, cin, 3, 3, device=dev(), requires_grad=True)
    yr = F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=dil, dilation=dil)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(cout, 9 * cin)                  # [cout, ky, kx, cin]
    err = (outs[0] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, float(ref.abs().max())), (err, float(ref.abs().max()))
    # refusals: unsupported width / dilation (TMAE_EARG), workspace too small (TMAE_EWS)
    assert lib.tmae_dense_conv3x3_wgrad(dy.data_ptr(), x.data_ptr(), B, Y, X, cin, 256, dil, outs[0].data_ptr(), ws.data_ptr(), wsb, st) == -1
    assert lib.tmae_dense_conv3x3_wgrad(dy.data_ptr(), x.data_ptr(), B, Y, X, cin, cout, 3, outs[0].data_ptr(), ws.data_ptr(), wsb, st) == -1
    assert lib.tmae_dense_conv3x3_wgrad(dy.data_ptr(), x.data_ptr(), B, Y, X, cin, cout, dil, outs[0].data_ptr(), ws.data_ptr(), 1024, st) == -2


def test_sparse_conv_golden_and_dense(oracle):
    from tmae_amd.modules.sparse import SparseConvTensor, SubMConv2d, SparseConv2d
    g = golden('F9_sparse_conv')
    ind = g['indices']
    feat = cu(g['feat']).requires_grad_(True)
    sp = SparseConvTensor(feat, cu(ind, torch.int32), [468, 468], 3)
    subm = SubMConv2d(16, 24, 3).to(dev())
    down = SparseConv2d(16, 24, 3, stride=2, padding=1).to(dev())
    with torch.no_grad():
        subm.weight.copy_(cu(g['weight'])), down.weight.copy_(cu(g['weight']))
    ys = subm(sp)
    np.testing.assert_allclose(ys.features.detach().cpu().numpy(), g['subm_out'], atol=1e-5)
    yd = down(sp)
    assert yd.spatial_shape == [234, 234]
    assert np.array_equal(yd.indices.cpu().numpy(), g['down_indices'])        # lexicographic output order
    np.testing.assert_allclose(yd.features.detach().cpu().numpy(), g['down_out'], atol=1e-5)
    # gradients against autograd through a dense conv on the densified input
    go_s, go_d = torch.randn_like(ys.features), torch.randn_like(yd.features)
    (ys.features * go_s).sum().backward(retain_graph=True)
    gs_feat, gs_w = feat.grad.clone(), subm.weight.grad.clone()
    feat.grad = None
    (yd.features * go_d).sum().backward()
    gd_feat, gd_w = feat.grad.clone(), down.weight.grad.clone()
    fo = torch.from_numpy(g['feat']).requires_grad_(True)
    wo = torch.from_numpy(g['weight']).requires_grad_(True)
    dense = oracle.to_dense(fo, ind, (468, 468), 3)
    wd = wo.permute(0, 3, 1, 2)
    r1 = F.conv2d(dense, wd, padding=1).permute(0, 2, 3, 1)[ind[:, 0], ind[:, 1], ind[:, 2]]
    (r1 * go_s.cpu()).sum().backward()
    np.testing.assert_allclose(gs_feat.cpu().numpy(), fo.grad.numpy(), atol=1e-4)
    np.testing.assert_allclose(gs_w.cpu().numpy(), wo.grad.numpy(), atol=1e-3)
    fo.grad, wo.grad = None, None
    oi = g['down_indices']
    dense = oracle.to_dense(fo, ind, (468, 468), 3)
    r2 = F.conv2d(dense, wo.permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)[oi[:, 0], oi[:, 1], oi[:, 2]]
    (r2 * go_d.cpu()).sum().backward()
    np.testing.assert_allclose(gd_feat.cpu().numpy(), fo.grad.numpy(), atol=1e-4)
    np.testing.assert_allclose(gd_w.cpu().numpy(), wo.grad.numpy(), atol=1e-3)


@pytest.mark.parametrize('cin,cout,kind', [(128, 128, 'subm'), (128, 256, 'down'), (256, 256, 'subm'), (256, 256, 'down'), (384, 256, 'subm'),
                                           (384, 128, 'subm')])
def test_spconv_native_implicit_gemm_vs_oracle(oracle, cin, cout, kind):
    """tmae_spconv_fwd / tmae_spconv_bwd_data (implicit GEMM over the rulebook, bf16) through the C ABI against the
    oracle's sparse conv and its autograd input gradient in fp32 on the same bf16-representable data; ragged sizes (the
    last row tile is partial) and absent neighbours included."""
    from tmae_amd import ops
    rng = np.random.default_rng(17)
    n = 7000
    c = np.unique(np.stack([rng.integers(0, 2, n), rng.integers(0, 150, n), rng.integers(0, 150, n)], 1), axis=0)
    ind = c[np.lexsort((c[:, 2], c[:, 1], c[:, 0]))]
    m = len(ind)
    assert m % 128 != 0
    oi, oshape, pairs = oracle.sparse_rulebook(ind, (150, 150), kind)
    x = torch.randn(m, cin).bfloat16().float()
    w = (torch.randn(cout, 3, 3, cin) * 0.05).bfloat16().float()
    xr, wr = x.clone().requires_grad_(True), w.clone()
    yr = oracle.sparse_conv(xr, wr, pairs, len(oi))
    gy = torch.randn(len(oi), cout).bfloat16().float()
    yr.backward(gy)
    indg = cu(ind, torch.int32)
    grid = ops.index_grid(indg, 2, 150, 150)
    if kind == 'subm':
        nbr = ops.spconv_neighbors(indg, grid, 2, 150, 150, 1)
        nbr_t = nbr.flip(1).contiguous()
    else:
        out_grid, out_ind, n_out, (oy, ox) = ops.spconv_down_outputs(grid, 2, 150, 150)
        mo = int(n_out)
        assert mo == len(oi) and np.array_equal(out_ind[:mo].cpu().numpy(), oi)
        nbr = ops.spconv_neighbors(out_ind[:mo], grid, 2, 150, 150, 2)
        nbr_t = ops.spconv_neighbors_t(indg, out_grid, 2, oy, ox, 2)
    w2d = cu(w).bfloat16().reshape(cout, 9 * cin).contiguous()
    y = ops.spconv_fwd(cu(x).bfloat16(), nbr, w2d)
    dx = ops.spconv_bwd_data(cu(gy).bfloat16(), nbr_t, w2d, cin)

    def rel(a, b):
        return float((a.float().cpu() - b).norm() / b.norm())
    assert rel(y, yr.detach()) < 6e-3, rel(y, yr.detach())                 # bf16 output rounding: 2^-9 per element
    assert rel(dx, xr.grad) < 6e-3, rel(dx, xr.grad)
    # the autograd op takes the same path and matches the gather + GEMM formulation it replaces
    xa = cu(x).bfloat16().requires_grad_(True)
    wa = cu(w).requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ya = ops.sparse_conv(xa, wa, nbr, nbr_t)
    ya.backward(cu(gy).bfloat16())
    if cin != 384:                     # (the model has no 384-channel sparse conv: the autograd op may route it differently)
        assert torch.equal(ya, y)
        assert torch.equal(xa.grad, dx)
    else:
        assert rel(ya, yr.detach()) < 6e-3 and rel(xa.grad, xr.grad) < 6e-3


def test_sparse_conv_bf16_wgrad_through_rulebook():
    """bf16 path of the sparse conv backward: the token-split MFMA weight gradient reads feature rows through the
    neighbour table (no [m, 9*cin] matrix); checked against an explicit fp32 gather + matmul."""
    from tmae_amd import ops
    from tmae_amd.modules.sparse import SparseConvTensor, SubMConv2d, SparseConv2d
    rng = np.random.default_rng(21)
    c = np.unique(np.stack([rng.integers(0, 2, 30000), rng.integers(0, 234, 30000), rng.integers(0, 234, 30000)], 1), axis=0)
    ind = cu(c, torch.int32)
    for Conv, kw, cin, cout in ((SubMConv2d, {}, 128, 128), (SparseConv2d, dict(stride=2, padding=1), 128, 256),
                                (SubMConv2d, {}, 256, 256)):
        feat = torch.randn(len(c), cin, device=dev()).bfloat16().requires_grad_(True)
        sp = SparseConvTensor(feat, ind, [234, 234], 2)
        conv = Conv(cin, cout, 3, **kw).to(dev())
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = conv(sp)
        go = torch.randn_like(y.features)
        y.features.backward(go)
        rb = sp._cache['subm' if Conv is SubMConv2d else 'down']
        nbr = rb[0].long()
        f32 = feat.detach().float()
        cols = torch.where((nbr >= 0).unsqueeze(-1), f32[nbr.clamp(min=0)], torch.zeros((), device=dev())).reshape(nbr.shape[0], -1)
        wmat = conv.weight.detach().bfloat16().float().reshape(cout, -1)
        y_ref = cols @ wmat.t()
        assert (y.features.float() - y_ref).abs().max().item() <= 2e-2 * float(y_ref.abs().max())
        dw_ref = (go.float().t() @ cols).reshape(conv.weight.shape)
        assert (conv.weight.grad - dw_ref).abs().max().item() <= 4e-3 * float(dw_ref.abs().max()), (cin, cout)
        dcols = go.float() @ wmat
        din_ref = torch.zeros_like(f32)
        for t in range(9):
            valid = nbr[:, t] >= 0
            din_ref.index_add_(0, nbr[valid, t], dcols[valid, t * cin:(t + 1) * cin])
        assert (feat.grad.float() - din_ref).abs().max().item() <= 3e-2 * float(din_ref.abs().max())


def test_dense_roundtrip_full_grid():
    from tmae_amd import ops
    rng = np.random.default_rng(8)
    c = np.unique(np.stack([rng.integers(0, 4, 150000), rng.integers(0, 468, 150000), rng.integers(0, 468, 150000)], 1), axis=0)
    ind = cu(c, torch.int32)
    grid = ops.index_grid(ind, 4, 468, 468)
    for dt in (torch.float32, torch.bfloat16):
        f = torch.randn(len(c), 128, device=dev()).to(dt).requires_grad_(True)
        d = ops.sparse_to_dense(f, grid, ind, 4, 468, 468)
        assert d.shape == (4, 468, 468, 128)
        assert torch.equal(ops.dense_gather(d, grid, ind), f)                # dense -> rows is the exact inverse
        assert float(d.float().abs().sum()) == pytest.approx(float(f.detach().float().abs().sum()), rel=1e-3)
        w = torch.randn_like(d)
        (d * w).sum().backward()
        assert torch.equal(f.grad, ops.dense_gather(w, grid, ind).to(dt))


# ------------------------------------------------------------------------------------------ end to end

def _run_product(model, pts, prv, noise, bs, amp=None):
    bd = {'points': cu(pts), 'points_prev': cu(prv), 'batch_size': bs, 'mae_noise': cu(noise)}
    model.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp is not None):
        ret, tb, _ = model(bd)
    ret['loss'].backward()
    return ret['loss'], bd


@pytest.mark.parametrize('name,nst', [('F11_e2e_1stage', 1), ('F10_e2e_3stage', 3), ('F12_e2e_ragged', 3)])
def test_e2e_golden_and_oracle(oracle, name, nst):
    """Whole step (VFE -> Siamese encoder -> masking -> WCA -> decoder -> Chamfer) in fp32 vs the reference's
    captured loss / mask / predictions / grad norms, and per-parameter gradients vs the CPU oracle."""
    g = golden(name)
    cfg = oracle.default_model_cfg(nst)
    P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']),
                           pred_scale=float(g['pred_scale']) if 'pred_scale' in g.files else 1.0)
    bs = int(g['batch_size'])
    model, _, _ = build_product_model(nst, params=P, device=dev())
    model.train()
    loss, bd = _run_product(model, g['points'], g['points_prev'], g['noise'], bs)
    assert abs(fl(loss) - float(g['loss'])) < 1e-4, (fl(loss), float(g['loss']))   # north-star bar
    assert np.array_equal(bd['voxel_mae_mask'].cpu().numpy(), g['mask'])
    assert np.array_equal(bd['voxel_coords'].cpu().numpy(), g['voxel_coords'])
    pred = model.backbone_3d.forward_ret_dict['pred_points']
    # decoder features reach |x| ~ 1e2 (BatchNorm over a mostly empty BEV grid), so fp32 summation-order noise of
    # 5e-6 relative there is ~5e-4 absolute here; the loss bar above is the north-star one
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g['pred_points'], atol=2e-3)
    sf = bd['spatial_features'].detach().double()
    assert float(sf.sum()) == pytest.approx(float(g['spatial_checksum']), rel=1e-4, abs=1.0)
    assert float(sf.abs().sum()) == pytest.approx(float(g['spatial_abs_checksum']), rel=1e-4)
    names = [str(n) for n in g['grad_names']]
    grads = dict(model.named_parameters())
    for n, gn in zip(names, g['grad_norms']):
        assert abs(float(grads[n].grad.norm()) - gn) <= 5e-3 * max(1.0, gn), n
    counts = dict(zip([str(s) for s in g['stage_count_names']], g['stage_counts']))
    feats = bd['multi_scale_3d_features']
    for si in range(nst):
        assert feats[f'x_conv{si + 1}'].features.shape[0] == counts[f'cur_M{si}']
    # per-parameter gradients against the oracle (CPU, same inputs)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lo = oracle.forward_loss(Pg, g['points'], g['points_prev'], g['noise'], bs, cfg)
    lo.backward()
    assert abs(fl(lo) - fl(loss)) < 1e-4
    # What this check covers: every gradient of the fp32 path against the oracle's.  In fp32 the decoder conv (forward, input and
    # weight gradient: `backbone_3d.decoder_conv_out.0.weight`) is the LIBRARY's (MIOpen; ops.dense_conv3x3_ok admits bf16 only), so
    # for that one tensor the comparison pins the glue around the conv, not csrc/dense_wgrad.hip / spconv_igemm.hip's halo kernels --
    # those are pinned by their operator tests against torch (test_dense_conv3x3_*), and at model level by
    # test_e2e_bf16_autocast_close_to_fp32, which compares the bf16 step's gradient of that tensor with this fp32 one.
    for n in names:
        a, b = grads[n].grad.cpu(), Pg[n].grad
        assert (a - b).abs().max().item() <= 5e-3 * max(1.0, b.abs().max().item()), n


@pytest.mark.parametrize('amp', [False, True])
def test_e2e_per_head_temperature_and_normalised_positions_vs_oracle(oracle, amp):
    """The two options round 6 added (LAYER_CFG.non_shared_tau, PREPROCESS.NORMALIZE_POS; off in the shipped YAMLs) through the WHOLE
    step of the 3-stage model built from the YAML with the options switched on: loss and every gradient (one per head for the
    temperatures) against the CPU oracle with the same options, fp32 at the golden tests' bars; under bf16 autocast the loss within
    3e-3 and the temperatures' gradients -- what the option adds -- direction-equal to the fp32 ones."""
    g = golden('F10_e2e_3stage')                       # its inputs (points of both frames, masking noise); the parameters are drawn here
    nst, bs = 3, int(g['batch_size'])
    cfg = oracle.default_model_cfg(nst)
    cfg['non_shared_tau'], cfg['normalize_pos'] = True, True
    P = oracle.init_params(cfg, seed=5, pred_scale=float(g['pred_scale']))
    taus = [k for k in P if k.endswith('.tau')]
    assert taus and all(tuple(P[k].shape) == (1, 8, 1, 1) for k in taus)

    def edit(c):
        for blk in c.MODEL.BACKBONE_3D.SST_BLOCK_LIST:
            blk.PREPROCESS.NORMALIZE_POS = True
            blk.ENCODER.LAYER_CFG['non_shared_tau'] = True
    model, _, _ = build_product_model(nst, params=P, device=dev(), cfg_edit=edit)
    model.train()
    loss, bd = _run_product(model, g['points'], g['points_prev'], g['noise'], bs, amp=True if amp else None)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lo = oracle.forward_loss(Pg, g['points'], g['points_prev'], g['noise'], bs, cfg)
    lo.backward()
    grads = dict(model.named_parameters())
    assert all(tuple(grads[k].grad.shape) == (1, 8, 1, 1) for k in taus)
    if not amp:
        assert abs(fl(lo) - fl(loss)) < 1e-4, (fl(lo), fl(loss))
        for n, p in grads.items():
            if n in Pg and Pg[n].grad is not None and p.grad is not None:
                a, b = p.grad.cpu(), Pg[n].grad
                assert (a - b).abs().max().item() <= 5e-3 * max(1.0, b.abs().max().item()), n
    else:
        assert abs(fl(lo) - fl(loss)) < 3e-3, (fl(lo), fl(loss))
        a = torch.cat([grads[k].grad.float().reshape(-1).cpu() for k in taus])
        b = torch.cat([Pg[k].grad.reshape(-1) for k in taus])
        assert float(torch.nn.functional.cosine_similarity(a, b, dim=0)) > 0.95


def test_pair_encode_equals_two_encoder_calls(oracle):
    """Running both frames through the Siamese encoder as one token list (SiamWCA_MAE.sparse_encode_pair, per-frame
    BatchNorm groups) gives the results of the reference's two calls: loss, gradients, BatchNorm running statistics."""
    g = golden('F10_e2e_3stage')
    cfg = oracle.default_model_cfg(3)
    P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']), pred_scale=float(g['pred_scale']))
    bs = int(g['batch_size'])
    res = []
    for pair in (True, False):
        model, _, _ = build_product_model(3, params=P, device=dev())
        model.train()
        model.backbone_3d.pair_encode = pair
        loss, bd = _run_product(model, g['points'], g['points_prev'], g['noise'], bs)
        res.append((fl(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None},
                    {n: b.clone() for n, b in model.named_buffers() if 'running' in n or 'num_batches' in n}))
    assert abs(res[0][0] - res[1][0]) < 2e-6
    assert res[0][1].keys() == res[1][1].keys()
    for n, a in res[0][1].items():
        b = res[1][1][n]
        assert (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item()), n
    for n, a in res[0][2].items():
        assert torch.allclose(a.float(), res[1][2][n].float(), rtol=1e-5, atol=1e-6), n


def test_e2e_waymo_shape_config(oracle):
    """BASELINE configs[3] shape: 5 point features (VFE input width 11), z range [-2,4), 6 m pillars; same model.
    (Range +-74.88 -> grid 468: with the dataset yaml's +-75.2 the reference's own decoder cannot concatenate
    118*4 = 472 with 470.)  Whole step in fp32 vs the CPU oracle."""
    cfg = oracle.default_model_cfg(3)
    cfg['point_cloud_range'] = [-74.88, -74.88, -2.0, 74.88, 74.88, 4.0]
    cfg['voxel_size'] = [0.32, 0.32, 6.0]
    P = oracle.init_params(cfg, seed=5, num_point_features=5, tau=0.3, pred_scale=0.1)
    rng = np.random.default_rng(3)
    pts, prv = oracle.synth_frame_pair(4000, 2, seed=77)
    pts[:, 3] += 1.5
    prv[:, 3] += 1.5
    pts = np.concatenate([pts, rng.uniform(0, 1, (len(pts), 1)).astype(np.float32)], 1)
    prv = np.concatenate([prv, rng.uniform(0, 1, (len(prv), 1)).astype(np.float32)], 1)
    vox = oracle.voxelize(pts, cfg['point_cloud_range'], cfg['voxel_size'], cfg['grid_size'])
    noise = rng.random(vox['voxel_coords'].shape[0]).astype(np.float32)
    model, _, ds = build_product_model(3, params=P, device=dev(), waymo_shape=True)
    assert list(ds.grid_size) == [468, 468, 1] and model.vfe.dvfe_mlps[0][0].weight.shape == (64, 11)
    model.train()
    loss, bd = _run_product(model, pts, prv, noise, 2)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    cap = {}
    lo = oracle.forward_loss(Pg, pts, prv, noise, 2, cfg, cap)
    lo.backward()
    assert abs(fl(lo) - fl(loss)) < 1e-4, (fl(lo), fl(loss))
    assert np.array_equal(bd['voxel_mae_mask'].cpu().numpy(), cap['mask'])
    assert np.array_equal(bd['voxel_coords'].cpu().numpy(), cap['vfe_cur']['voxel_coords'])
    grads = dict(model.named_parameters())
    for n in ('vfe.dvfe_mlps.0.0.weight', 'backbone_3d.sst_blocks.2.conv_down.0.weight', 'backbone_3d.decoder_pred.weight'):
        a, b = grads[n].grad.cpu(), Pg[n].grad
        assert (a - b).abs().max().item() <= 5e-3 * max(1.0, b.abs().max().item()), n


def test_e2e_bf16_autocast_close_to_fp32(oracle):
    """The benched dtype: the whole step under bf16 autocast against the same step in fp32 (which the golden tests pin
    to the reference) and against the reference's own loss.  Conditioned head (as F10): |loss difference| <= 2e-3
    (measured 3-6e-4: bf16 keeps 8 mantissa bits through ~40 layers); default head: 1 % of the loss."""
    g = golden('F10_e2e_3stage')
    cfg = oracle.default_model_cfg(3)
    for scale, bar in ((float(g['pred_scale']), 2e-3), (1.0, None)):
        P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']), pred_scale=scale)
        model, _, _ = build_product_model(3, params=P, device=dev())
        model.train()
        loss, _ = _run_product(model, g['points'], g['points_prev'], g['noise'], int(g['batch_size']), amp=True)
        g16 = {n: p.grad.clone() for n, p in model.named_parameters()}
        l32, _ = _run_product(model, g['points'], g['points_prev'], g['noise'], int(g['batch_size']))
        assert torch.isfinite(loss)
        if bar is not None:
            assert abs(fl(loss) - fl(l32)) < bar, (fl(loss), fl(l32))
            assert abs(fl(loss) - float(g['loss'])) < bar, (fl(loss), float(g['loss']))
        else:
            assert abs(fl(loss) - fl(l32)) < 0.01 * max(1.0, abs(fl(l32))), (fl(loss), fl(l32))
        assert all(torch.isfinite(v).all() for v in g16.values())
        # Gradient fidelity of the bf16 step, stated with metrics that can hold a bar (round 6; until then: the cosine of the WORST
        # big tensor, a figure that is chaotic at the third digit -- it moved 0.9532 -> 0.9496 when round 5 rewrote the attention
        # masks with outputs equal within one bf16 ulp -- and whose bar had followed it to 0.94):
        #  (1) cosine of the CONCATENATED gradient (all parameters) against the fp32 gradient: the direction the optimizer follows;
        #  (2) per big tensor, the relative L2 error against the fp32 gradient, measured against that tensor's own NOISE FLOOR: the
        #      relative L2 change of the fp32 gradient when nothing but the parameters are rounded to bf16 (the smallest perturbation
        #      any bf16 step makes, 2^-9 relative per weight, computed in fp32 throughout).  A tensor whose fp32 gradient moves by 30 %
        #      under that rounding (the stage-2 in-projections behind ~30 layers, tau = the golden case's) cannot be asked to match to
        #      5 %; one that moves by 1 % can.  The bar is a multiple of the floor plus the bf16 rounding of the gradient itself.
        g32 = {n: p.grad.clone() for n, p in model.named_parameters()}
        saved = {n: p.detach().clone() for n, p in model.named_parameters()}
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.copy_(p.to(torch.bfloat16).float())
        _run_product(model, g['points'], g['points_prev'], g['noise'], int(g['batch_size']))
        g32r = {n: p.grad.clone() for n, p in model.named_parameters()}
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.copy_(saved[n])
        names = [n for n in g32 if g32[n].norm() > 0]
        cat = lambda d: torch.cat([d[n].flatten().double() for n in names])
        cos_all = float(torch.nn.functional.cosine_similarity(cat(g16), cat(g32), dim=0))
        cos_floor = float(torch.nn.functional.cosine_similarity(cat(g32r), cat(g32), dim=0))
        rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        rows = []
        for n in names:
            if g32[n].numel() >= 16384:
                rows.append((rel(g16[n], g32[n]), rel(g32r[n], g32[n]), n))
        worst = max(rows, key=lambda r: r[0] / max(r[1], 1e-3))
        print(f'bf16 vs fp32: cosine of the concatenated gradient {cos_all:.4f} (weights-rounded fp32: {cos_floor:.4f}); worst big tensor '
              f'rel-L2 {worst[0]:.3f} at floor {worst[1]:.3f}: {worst[2]}; largest rel-L2 {max(rows)[0]:.3f} ({max(rows)[2]})')
        assert cos_all > 0.95, (cos_all, cos_floor)
        for e16, floor, n in rows:
            assert e16 <= 3.0 * floor + 0.02, (n, e16, floor)
        # the decoder conv's weight gradient: in the fp32 run it is MIOpen's (ops.dense_conv3x3_ok admits bf16 only), in the bf16 run
        # it is csrc/dense_wgrad.hip's -- the two agree to bf16 accuracy at model level (the operator tests compare that kernel
        # with torch on its own)
        n = 'backbone_3d.decoder_conv_out.0.weight'
        assert rel(g16[n], g32[n]) <= 3.0 * rel(g32r[n], g32[n]) + 0.02, (rel(g16[n], g32[n]), rel(g32r[n], g32[n]))


def test_vfe_bf16_keeps_far_range_coordinates(oracle):
    """Under bf16 autocast the VFE's first Linear must not see coordinates rounded to bf16 (0.5 m steps beyond 64 m):
    points between 64 and 74 m, voxel features of the bf16 path vs the fp32 path and vs the oracle."""
    from tmae_amd import ops
    cfg = oracle.default_model_cfg(1)
    P = oracle.init_params(cfg, seed=2)
    rng = np.random.default_rng(5)
    n = 20000
    r = rng.uniform(64.5, 74.0, n)
    th = rng.uniform(0, 2 * np.pi, n)
    pts = np.stack([np.zeros(n), np.clip(r * np.cos(th), -74.8, 74.8), np.clip(r * np.sin(th), -74.8, 74.8),
                    rng.normal(-1.5, 0.4, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
    model, _, _ = build_product_model(1, params=P, device=dev())
    model.train()
    out = {}
    for amp in (False, True):
        bd = {'points': cu(pts), 'points_prev': cu(pts), 'batch_size': 1}
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            bd = model.vfe(bd)
        out[amp] = bd['voxel_features'].float().cpu().numpy()
    ref = oracle.vfe_forward({k: v for k, v in P.items()}, 'vfe.', pts, cfg)['voxel_features'].detach().numpy()
    np.testing.assert_allclose(out[False], ref, rtol=1e-4, atol=1e-4)
    scale = np.abs(ref).max()
    err16 = np.abs(out[True] - ref).max() / scale
    # bf16 activations (8 bits) after two BatchNorm'd layers: a few 1e-2 of the range; with bf16-rounded coordinates
    # the f_center / f_cluster offsets (|.| < 0.32 m) drown in the 0.5 m rounding and this error is > 0.3
    assert err16 < 0.05, err16
    # direct check of the split representation: hi + lo reproduces the fp32 features to 2^-16 relative
    v = ops.voxelize(cu(pts), 1, PCR, VS, GRID)
    perm, offs = ops.segment_csr(v['inverse'], v['voxel_coords'].shape[0])
    args = (v['points'], v['point_coords'], v['inverse'], perm, offs, v['voxel_coords'].shape[0], PCR, VS)
    _, f32 = ops.vfe_point_features(*args)
    _, hl = ops.vfe_point_features_bf16x2(*args)
    rec = hl[:, :16].float() + hl[:, 16:].float()
    assert (rec[:, :10] - f32).abs().max().item() <= 2.0 ** -16 * 75.0
    assert (rec[:, 10:] == 0).all()
    # CSR order (what the model uses): the same rows, sorted by voxel; inverse_csr names the voxel of every row; the segment max
    # over them with perm = None equals the one over the unsorted rows with the permutation, values AND gradient
    _, hl_csr, inv_csr = ops.vfe_point_features_bf16x2(*args, csr_order=True)
    assert torch.equal(hl_csr, hl[perm.long()]) and torch.equal(inv_csr, v['inverse'][perm.long()])
    assert bool((inv_csr[1:] >= inv_csr[:-1]).all())
    mvox = v['voxel_coords'].shape[0]
    xa = torch.randn(hl.shape[0], 128, device=dev()).bfloat16()
    xb = xa[perm.long()].contiguous()
    xa.requires_grad_(True), xb.requires_grad_(True)
    ya, _ = ops.scatter_max(xa, v['inverse'], perm, offs, mvox)
    yb, _ = ops.scatter_max(xb, inv_csr, None, offs, mvox)
    assert torch.equal(ya, yb)
    go = torch.randn_like(ya)
    ya.backward(go), yb.backward(go)
    assert torch.equal(xb.grad, xa.grad[perm.long()])


def test_proj_fork_out_of_place_is_hook_safe():
    """ops.proj_fork accumulates into the alias gradient in place only on the caller's word; the default path leaves
    the tensor autograd handed over untouched (a hook / retain_grad on the alias sees the right values) and both
    paths give the same gradients."""
    from tmae_amd import ops
    torch.manual_seed(0)
    m, d = 9000, 128
    x0 = torch.randn(m, d, device=dev()).bfloat16()
    w = (torch.randn(2 * d, d, device=dev()) * 0.05).requires_grad_(True)
    b = torch.zeros(2 * d, device=dev(), requires_grad=True)
    res = {}
    for inplace in (False, True):
        x = x0.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            h, alias = ops.proj_fork(x, w, b, ((0, 2 * d, False),), fork=True, inplace_dx=inplace)
        seen = []
        if not inplace:
            alias.register_hook(lambda g_: seen.append(g_.clone()))
            alias.retain_grad()
        (h.float().sum() * 0.5 + (alias.float() * 3).sum()).backward()
        res[inplace] = x.grad.float().clone()
        if not inplace:
            assert torch.equal(seen[0].float(), torch.full_like(seen[0].float(), 3.0))
            assert torch.equal(alias.grad.float(), torch.full_like(alias.grad.float(), 3.0))    # not overwritten
        w.grad = None
    assert torch.allclose(res[False], res[True], rtol=2e-2, atol=2e-2)


def test_e2e_full_size_120k_vs_oracle(oracle):
    """BASELINE configs[1] size -- ONE 120 k-point frame pair, full 3-stage model, fp32 -- against the CPU oracle on the
    same inputs, weights and masking noise (what bench.py reports as `parity`):
    conditioned head (decoder_pred x 0.1): |loss difference| <= 1e-4 (the north-star bar), masks bit-exact, gradients of a
    few parameters <= 5e-3 relative;
    default head: the bar is 3x the oracle's OWN |loss(1 thread) - loss(all threads)|, measured here, floor 1e-4."""
    pts, prv = oracle.synth_frame_pair(120000, 1, seed=0)
    cfg = oracle.default_model_cfg(3)
    vox = oracle.voxelize(pts, cfg['point_cloud_range'], cfg['voxel_size'], cfg['grid_size'])
    noise = np.random.default_rng(0).random(vox['voxel_coords'].shape[0]).astype(np.float32)
    nthr = torch.get_num_threads()
    # --- conditioned head, forward + backward
    P = oracle.init_params(cfg, seed=0, pred_scale=0.1)
    model, _, _ = build_product_model(3, params=P, device=dev(), batch_size=1)
    model.train()
    loss, bd = _run_product(model, pts, prv, noise, 1)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    cap = {}
    lo = oracle.forward_loss(Pg, pts, prv, noise, 1, cfg, cap)
    lo.backward()
    assert abs(fl(lo) - fl(loss)) < 1e-4, (fl(lo), fl(loss))
    assert np.array_equal(bd['voxel_mae_mask'].cpu().numpy(), cap['mask'])
    assert np.array_equal(bd['voxel_coords'].cpu().numpy(), cap['vfe_cur']['voxel_coords'])
    feats = bd['multi_scale_3d_features']
    for si in range(3):
        assert feats[f'x_conv{si + 1}'].features.shape[0] == len(cap[f'cur_stage{si}']['indices'])
    grads = dict(model.named_parameters())
    for n in ('vfe.dvfe_mlps.0.0.weight', 'backbone_3d.sst_blocks.0.encoder_blocks.0.encoder_list.0.win_attn.self_attn.in_proj_weight',
              'backbone_3d.sst_blocks.1.conv_down.0.weight', 'backbone_3d.sst_blocks.2.encoder_blocks.1.encoder_list.1.linear2.weight',
              'backbone_3d.wca_blocks.1.encoder_blocks.0.encoder_list.0.win_attn.cross_attn.in_proj_weight',
              'backbone_3d.decoder_conv_out.0.weight', 'backbone_3d.decoder_pred.weight'):
        a, b = grads[n].grad.cpu(), Pg[n].grad
        assert (a - b).abs().max().item() <= 5e-3 * max(1.0, b.abs().max().item()), n
    # --- bf16 autocast on the same pair (the benched dtype)
    l16, _ = _run_product(model, pts, prv, noise, 1, amp=True)
    assert abs(float(l16) - fl(lo)) < 3e-3, (float(l16), fl(lo))
    # ... and its gradients (the position-folded in-projections run here: > 8192 tokens per stage): in-projection weights
    # incl. their position part, an FFN weight, a sparse conv and the decoder conv against the oracle's fp32 gradients.
    # The bar is what bf16 activations allow on ONE frame pair -- these gradients are sums of ~1e5 cancelling rows:
    # cosine 0.96-0.99 / relative error 0.11-0.27, the same with the position embedding folded or materialised
    # (profiles/scripts/pf_check.py) -- not a kernel tolerance (those are pinned by the operator tests above)
    for n in ('backbone_3d.sst_blocks.0.encoder_blocks.0.encoder_list.0.win_attn.self_attn.in_proj_weight',
              'backbone_3d.sst_blocks.0.encoder_blocks.1.encoder_list.1.win_attn.self_attn.in_proj_bias',
              'backbone_3d.sst_blocks.1.encoder_blocks.0.encoder_list.1.win_attn.self_attn.in_proj_weight',
              'backbone_3d.sst_blocks.2.encoder_blocks.1.encoder_list.0.win_attn.self_attn.in_proj_weight',
              'backbone_3d.wca_blocks.1.encoder_blocks.0.encoder_list.0.win_attn.cross_attn.in_proj_weight',
              'backbone_3d.wca_blocks.2.encoder_blocks.0.encoder_list.1.win_attn.cross_attn.in_proj_weight',
              'backbone_3d.sst_blocks.2.encoder_blocks.1.encoder_list.1.linear2.weight',
              'backbone_3d.sst_blocks.1.conv_down.0.weight', 'backbone_3d.decoder_conv_out.0.weight'):
        a, b = grads[n].grad.float().cpu().reshape(-1), Pg[n].grad.reshape(-1)
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        assert cos > 0.93 and rel < 0.4, (n, cos, rel)
    # --- default head, forward only
    P1 = oracle.init_params(cfg, seed=0)
    model1, _, _ = build_product_model(3, params=P1, device=dev(), batch_size=1)
    model1.train()
    with torch.no_grad():
        ret, _, _ = model1({'points': cu(pts), 'points_prev': cu(prv), 'batch_size': 1, 'mae_noise': cu(noise)})
        l_all = float(oracle.forward_loss(P1, pts, prv, noise, 1, cfg))
        torch.set_num_threads(1)
        try:
            l_one = float(oracle.forward_loss(P1, pts, prv, noise, 1, cfg))
        finally:
            torch.set_num_threads(nthr)
    spread = abs(l_one - l_all)
    bar = max(3 * spread, 1e-4)
    assert min(abs(fl(ret['loss']) - l_all), abs(fl(ret['loss']) - l_one)) <= bar, \
        (fl(ret['loss']), l_all, l_one, spread)


def test_full_size_properties(oracle):
    """BASELINE config-2 sizes (120k-point frames, batch 2): size-independent properties of the index path."""
    from tmae_amd import ops
    pts, prv = oracle.synth_frame_pair(120000, 2, seed=1)
    v = ops.voxelize(cu(pts), 2, PCR, VS, GRID)
    vc, inv, pc = v['voxel_coords'], v['inverse'], v['point_coords']
    assert torch.equal(vc[inv], pc)                                          # every point maps to its own voxel
    key = (vc[:, 0] * 468 + vc[:, 2]) * 468 + vc[:, 3]
    assert (key[1:] > key[:-1]).all()                                        # sorted, unique
    assert sum(v['voxels_per_sample']) == vc.shape[0]
    assert torch.unique(inv).numel() == vc.shape[0]
    o = oracle.voxelize(pts, PCR, VS, GRID)
    assert np.array_equal(vc.cpu().numpy(), o['voxel_coords']) and np.array_equal(inv.cpu().numpy(), o['inverse'])
    m = vc.shape[0]
    per = v['voxels_per_sample']
    offs = torch.tensor(np.concatenate([[0], np.cumsum(per)]), dtype=torch.int32, device=dev())
    noise = torch.rand(m, device=dev())
    mask, vis, nvis = ops.random_mask(noise, offs, 2, 0.25)
    assert int(nvis) == sum(int(L * 0.25) for L in per) == int((mask == 0).sum())
    ref = oracle.mask_voxels(vc.cpu().numpy(), noise.cpu().numpy(), 0.75, 2)
    assert np.array_equal(mask.cpu().numpy(), ref)
    ind = vc[:, [0, 2, 3]].int().contiguous()
    grid = ops.index_grid(ind, 2, 468, 468)
    for s in (False, True):
        wb = ops.window_bucket(ind, grid, None, 2, 468, 468, [8, 8, 1], s, DROP)
        bwi, _ = oracle.get_window_coors(vc.cpu().numpy(), (468, 468, 1), (8, 8, 1), s)
        assert np.array_equal(wb['batch_win_inds'].cpu().numpy(), bwi)
        assert np.array_equal(wb['inner'].cpu().numpy(), oracle.stable_ingroup_rank(bwi))
        assert int(wb['win_per_level'].sum()) == len(np.unique(bwi))


def test_bf16_trains_like_fp32():
    """The benched dtype must TRAIN like fp32, not only agree on one forward pass: 30 optimizer steps of the full model
    (B = 2 x 20 k-point pairs, the recipe's Adam one-cycle) from the same initial weights on the same batches with the
    same masking noise, once in fp32 and once under bf16 autocast (bench.training_curves, also printed in the bench
    line's `parity`).  Both curves must fall (3.77 -> 0.88 here) and stay within a stated band of each other: every step
    within 8 %, the means over the last ten steps within 3 %.  What was measured: the bf16 curve repeats bit for bit from
    run to run; the fp32 curve does not (library reductions on the fp32 path), and from step ~20 on its own runs differ by
    up to 6 % at single steps (0.888 / 0.939 at step 27 in two runs) -- the two dtypes were within 0.5 % of each other in
    one run and 5.7 % in another, i.e. inside fp32's own run-to-run spread.
    WHICH library call (round 5, test_step_gradients_are_bit_reproducible / profiles/round5_reproducibility.txt): of the 278
    gradients of one fp32 step exactly one differs between runs from identical state -- `decoder_conv_out.0.weight`, the weight
    gradient of the decoder's 3x3 conv, which in fp32 is the library's (MIOpen) kernel (1e-7 relative per step; Adam's
    normalisation amplifies it over 20+ steps).  The library GEMMs, reductions and every kernel of this repository repeat bit
    for bit."""
    import bench
    c = bench.training_curves(dev())
    a, b = np.array(c['fp32']), np.array(c['bf16'])
    print('fp32', c['fp32'])
    print('bf16', c['bf16'])
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert a[-5:].mean() < 0.9 * a[:5].mean() and b[-5:].mean() < 0.9 * b[:5].mean(), (c['fp32_first5_last5'], c['bf16_first5_last5'])
    assert abs(a[0] - b[0]) <= 2e-3 * abs(a[0])                          # the first step: same weights, bf16 forward error only
    assert c['max_rel_gap'] <= 0.08, c['max_rel_gap']
    assert abs(a[-10:].mean() - b[-10:].mean()) <= 0.03 * a[-10:].mean()


def test_vfe_prefetch_one_step_ahead_changes_nothing():
    """train_one_step(next_batch=...): the next batch's voxelisation enqueued between this step's forward and backward
    (TemporalDynVFE.prefetch) -- three optimizer steps with and without the lookahead give bit-identical losses and weights."""
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import model_fn_decorator
    from tmae_amd.train import SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler, train_one_step
    import bench
    cfg = cfg_from_yaml_file(os.path.join(bench.ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml'), EasyDict())
    ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=12000, batch_size=2)
    raw = [ds.batch(i) for i in range(3)]
    res = []
    for ahead in (True, False):
        torch.manual_seed(7)
        model = build_model_from_cfg(cfg, ds).to(dev()).train()
        opt = build_optimizer(model, cfg.OPTIMIZATION)
        sched, _ = build_scheduler(opt, 100, cfg.OPTIMIZATION.NUM_EPOCHS, -1, cfg.OPTIMIZATION)
        mk = lambda b: {'points': torch.from_numpy(b['points']).to(dev()), 'points_prev': torch.from_numpy(b['points_prev']).to(dev()),
                        'batch_size': b['batch_size']}
        dicts = [mk(b) for b in raw]
        losses = []
        for i in range(3):
            torch.manual_seed(100 + i)                                           # the masking noise
            nxt = dicts[i + 1] if (ahead and i + 1 < 3) else None
            loss, _, _ = train_one_step(model, opt, sched, dicts[i], i, model_fn_decorator(), amp_dtype=torch.bfloat16, settle=False,
                                        next_batch=nxt)
            if ahead and i + 1 < 3:
                assert '_vfe_prefetch' in dicts[i + 1]
            losses.append(loss.detach().clone())
        res.append((losses, torch.cat([p.detach().flatten() for p in model.parameters()]).clone()))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b), (a, b)
    assert torch.equal(res[0][1], res[1][1])


def test_step_gradients_are_bit_reproducible():
    """One forward + backward of the full model (B = 2 x 20 k-point pairs) three times from identical state and masking noise:
    under bf16 autocast (the benched path) EVERY gradient repeats bit for bit -- all reductions of this repository run in a fixed
    order, including the tau gradient's per-window partials (until round 5 the paired <= 8-token attention class wrote one
    partial per PAIR of windows, and which windows share a unit depends on the order of the atomically built work list: the 16
    tau gradients differed in their last bits) -- and in fp32 everything but the decoder conv's weight gradient, which is the
    library's (MIOpen) fp32 kernel."""
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import model_fn_decorator
    from tmae_amd.train import SyntheticTemporalDataset, build_model_from_cfg
    import bench
    cfg = cfg_from_yaml_file(os.path.join(bench.ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml'), EasyDict())
    ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=20000, batch_size=2)
    b = ds.batch(0)
    batch = {'points': torch.from_numpy(b['points']).to(dev()), 'points_prev': torch.from_numpy(b['points_prev']).to(dev()),
             'batch_size': b['batch_size']}
    torch.manual_seed(1234)
    model = build_model_from_cfg(cfg, ds).to(dev()).train()
    mf = model_fn_decorator()
    for amp in (torch.bfloat16, None):
        runs = []
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            torch.manual_seed(99)
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp is not None):
                ret = mf(model, dict(batch))[0]
            loss = (ret.loss if hasattr(ret, 'loss') else ret).mean()
            loss.backward()
            runs.append((loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][0], runs[2][0])
        differ = sorted({n for n in runs[0][1] for k in (1, 2) if not torch.equal(runs[0][1][n], runs[k][1][n])})
        if amp is not None:
            assert differ == [], differ
        else:
            # measured: exactly {'backbone_3d.decoder_conv_out.0.weight'} (MIOpen's fp32 weight-gradient kernel).  Which library
            # kernel a box picks is not ours to pin, so the assertion is on what IS ours: nothing that only passes through this
            # repository's kernels may differ (the tau gradients, the norms), and the list stays a handful
            # Explicit allow-list: the parameters whose fp32 gradient comes from a library kernel.  In fp32 the decoder
            # conv runs on MIOpen (ops.dense_conv3x3_ok admits bf16 only); its weight gradient is the only gradient that is a
            # library kernel's OUTPUT (the conv's input gradient feeds the encoder, but MIOpen's data-gradient kernel repeats bit
            # for bit on every box measured).  A new name here must be traced to a library kernel before it is added.
            print('fp32 gradients that differ between runs:', differ)
            allow = {'backbone_3d.decoder_conv_out.0.weight'}
            assert set(differ) <= allow, differ


def test_token_gemm_gelu_dual_store_vs_torch():
    """tmae_token_gemm_gelu (the FFN's first Linear + its exact GELU in one launch, csrc/token_gemm_wreg.hip GELU2): both
    shapes, ragged token counts, a strided x.  y against fp32 matmul on the same bf16 inputs; y_gelu against torch's erf
    GELU of the KERNEL's own y (the activation is taken from the bf16-rounded pre-activation, as F.gelu(y) would): one
    bf16 rounding (relative 2^-8) plus the Abramowitz-Stegun erf's absolute error (< 1.5e-7 in erf, i.e. < 1e-7 |y| in
    the result -- visible only in the far negative tail, where gelu itself is ~1e-6; the backward's gelu' has used the
    same formula since round 1).  Then the encoder tail that uses it
    (ops.proj_fork(gelu=True) -> ops.gelu_linear(h=...)) against the two-pass form: identical outputs and gradients."""
    from tmae_amd import ops
    from tmae_amd._lib import lib, check
    torch.manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream
    for (m, k, n) in ((32768 + 17, 256, 512), (70001, 256, 512), (32769, 128, 256), (100003, 128, 256)):
        big = torch.randn(m, k + 64, device=dev()).bfloat16()
        x = big[:, 32:32 + k]                                                   # pitch k + 64
        w = (torch.randn(n, k, device=dev()) * 0.1).bfloat16()
        b = torch.randn(n, device=dev()).bfloat16()
        y = torch.full((m, n), float('nan'), device=dev(), dtype=torch.bfloat16)
        yg = torch.full((m, n), float('nan'), device=dev(), dtype=torch.bfloat16)
        check(lib.tmae_token_gemm_gelu(x.data_ptr(), x.stride(0), m, k, w.data_ptr(), n, b.data_ptr(), y.data_ptr(), yg.data_ptr(), n,
                                       st), 'tmae_token_gemm_gelu')
        ref = x.float() @ w.float().t() + b.float()
        assert (y.float() - ref).abs().max().item() <= 2e-2 * max(1.0, float(ref.abs().max())), (m, k, n)
        y_plain = ops.token_gemm(x, w, b, force=True)
        assert torch.equal(y, y_plain), 'the pre-activation must not depend on the second store'
        want = F.gelu(y.float())
        bound = want.abs() * 2.0 ** -8 + 2e-7 * y.float().abs() + 1e-30
        assert bool(((yg.float() - want).abs() <= bound).all()), (m, k, n, float(((yg.float() - want).abs() / bound).max()))
        assert float((yg.float() != want.bfloat16().float()).float().mean()) < 2e-3        # almost always the same rounding
    # refused shapes: the caller falls back (ops.token_gemm_gelu does)
    x = torch.randn(1000, 256, device=dev()).bfloat16()
    w = torch.randn(512, 256, device=dev()).bfloat16()
    b = torch.zeros(512, device=dev()).bfloat16()
    y = torch.empty(1000, 512, device=dev(), dtype=torch.bfloat16)
    assert lib.tmae_token_gemm_gelu(x.data_ptr(), 256, 1000, 256, w.data_ptr(), 512, b.data_ptr(), y.data_ptr(), y.data_ptr(), 512, st) == -1
    y2, g2 = ops.token_gemm_gelu(x, w, b)
    assert torch.equal(g2, F.gelu(y2))
    # the encoder tail: fused vs two-pass, forward and every gradient
    m, d = 40000, 256
    xin = torch.randn(m, d, device=dev()).bfloat16()
    lin1, lin2 = torch.nn.Linear(d, 2 * d).to(dev()), torch.nn.Linear(2 * d, d).to(dev())
    outs = []
    for fused in (True, False):
        xi = xin.clone().requires_grad_(True)
        for p in list(lin1.parameters()) + list(lin2.parameters()):
            p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            if fused:
                h_pre, h_act, alias = ops.proj_fork(xi, lin1.weight, lin1.bias, ((0, 2 * d, False),), fork=True, gelu=True)
                out = ops.gelu_linear(h_pre, lin2.weight, lin2.bias, h=h_act) + alias
            else:
                h_pre, alias = ops.proj_fork(xi, lin1.weight, lin1.bias, ((0, 2 * d, False),), fork=True)
                out = ops.gelu_linear(h_pre, lin2.weight, lin2.bias) + alias
        out.float().square().mean().backward()
        outs.append((out.detach().float(), xi.grad.float(), lin1.weight.grad.clone(), lin2.weight.grad.clone(), lin1.bias.grad.clone()))
    for a, b_ in zip(*outs):
        assert (a - b_).abs().max().item() <= 4e-3 * max(1e-6, float(b_.abs().max())), float((a - b_).abs().max())


def test_e2e_token_dropping_vs_reference(oracle):
    """DROP_INFO that really drops tokens (max_tokens 4 / 8 / 12: a pure YAML edit the reference accepts; the shipped
    levels never drop, SURVEY A-6) on a dense cloud: half of the stage-1 voxels skip the encoder of their SST block
    (spt_backbone.py:47-135,347-349) and the window cross-attention sees per-shift keep sets of both frames
    (SiamWCA.py:65-215).  Whole step in fp32 against the reference's captured loss / mask / predictions / gradient
    norms (fixture F13, oracle/gen_golden_dropping.py) and, per parameter, against the oracle's gradients."""
    g = golden('F13_e2e_dropping')
    nst, bs = int(g['num_stages']), int(g['batch_size'])
    drop = {i: dict(max_tokens=int(t), drop_range=(int(lo), int(hi)))
            for i, (t, lo, hi) in enumerate(zip(g['drop_max_tokens'], g['drop_lower'], g['drop_upper']))}
    cfg = oracle.default_model_cfg(nst)
    cfg['drop_info'] = drop
    P = oracle.init_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']), pred_scale=float(g['pred_scale']))
    from pcdet.models import build_network
    from tmae_amd.train import SyntheticTemporalDataset
    from conftest import load_cfg
    ycfg = load_cfg(nst)
    for blk in ycfg.MODEL.BACKBONE_3D.SST_BLOCK_LIST:
        for mode in ('train', 'test'):
            blk.PREPROCESS.DROP_INFO[mode] = {str(k): {'max_tokens': v['max_tokens'], 'drop_range': list(v['drop_range'])}
                                              for k, v in drop.items()}
    ds = SyntheticTemporalDataset(ycfg.DATA_CONFIG, ycfg.CLASS_NAMES, n_points=1000, batch_size=bs)
    model = build_network(ycfg.MODEL, len(ycfg.CLASS_NAMES), ds)
    res = model.load_state_dict(P, strict=False)
    assert not res.unexpected_keys
    model = model.to(dev()).train()
    assert all(b.can_drop for b in model.backbone_3d.sst_blocks) and all(b.can_drop for b in model.backbone_3d.wca_blocks)
    loss, bd = _run_product(model, g['points'], g['points_prev'], g['noise'], bs)
    assert abs(fl(loss) - float(g['loss'])) < 1e-4, (fl(loss), float(g['loss']))
    assert np.array_equal(bd['voxel_mae_mask'].cpu().numpy(), g['mask'])
    pred = model.backbone_3d.forward_ret_dict['pred_points']
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g['pred_points'], atol=2e-3)
    sf = bd['spatial_features'].detach().double()
    assert float(sf.abs().sum()) == pytest.approx(float(g['spatial_abs_checksum']), rel=1e-4)
    grads = dict(model.named_parameters())
    bad = [(n, float(grads[n].grad.norm()), float(gn)) for n, gn in zip([str(n) for n in g['grad_names']], g['grad_norms'])
           if not abs(float(grads[n].grad.norm()) - gn) <= 5e-3 * max(1.0, gn)]
    assert not bad, bad[-6:]
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lo = oracle.forward_loss(Pg, g['points'], g['points_prev'], g['noise'], bs, cfg)
    lo.backward()
    for n in [str(n) for n in g['grad_names']]:
        a, b = grads[n].grad.cpu(), Pg[n].grad
        assert (a - b).abs().max().item() <= 5e-3 * max(1.0, b.abs().max().item()), n
    # bf16 autocast: the same keep sets, loss within the bf16 bar of the other e2e tests
    l16, _ = _run_product(model, g['points'], g['points_prev'], g['noise'], bs, amp=True)
    assert abs(float(l16.detach()) - float(g['loss'])) < 3e-3
    # the shipped levels never enter the dropping code
    m3, _, _ = build_product_model(1, device=dev())
    assert not any(b.can_drop for b in m3.backbone_3d.sst_blocks) and not any(b.can_drop for b in m3.backbone_3d.wca_blocks)
