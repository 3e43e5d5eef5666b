"""GPU (-m gpu): the fine-tune path (BASELINE configs[4]) through the C ABI against the golden
vectors captured from the reference (G1-G3) and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import build_finetune_model, golden

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda:0')


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


def _dense_heat(g):
    hm = np.zeros(tuple(g['heat_shape']), np.float32)
    ix = g['heat_nz_index']
    hm[ix[:, 0], ix[:, 1], ix[:, 2], ix[:, 3]] = g['heat_nz_value']
    return hm


def test_centerhead_targets_golden():
    """tmae_centerhead_targets vs CenterHead.assign_targets of the reference: heat map, slots and masks bit-exact
    (Gaussian radius / centre truncation follow the reference's fp32 operation order); log / sin / cos to 1e-6."""
    from tmae_amd import ops
    g = golden('G1_centerhead_targets')
    cmap = torch.tensor([-1, 0, 1, 2, 3, 4], dtype=torch.int32, device=dev())
    heat, tb, inds, mask = ops.centerhead_targets(cu(g['gt_boxes']), cmap, 5, (468, 468), [-74.88, -74.88, -5.0],
                                                  [0.32, 0.32, 8.0], 1, 500, 0.1, 2)
    assert np.array_equal(inds.cpu().numpy(), g['inds']) and np.array_equal(mask.cpu().numpy(), g['masks'])
    ref = _dense_heat(g)
    got = heat.cpu().numpy()
    assert np.array_equal(got != 0, ref != 0)
    assert np.abs(got - ref).max() <= 6e-8                     # double exp on both sides, rounded to f32: <= 1 ulp
    assert (got != ref).mean() < 1e-4
    np.testing.assert_allclose(tb.cpu().numpy(), g['target_boxes'], atol=1e-6)
    # two heads: classes split, slots are the ranks inside each head
    cm_a = torch.tensor([-1, 0, -1, -1, 1, -1], dtype=torch.int32, device=dev())        # head A = Car, Pedestrian
    ha, tba, ia, ma = ops.centerhead_targets(cu(g['gt_boxes']), cm_a, 2, (468, 468), [-74.88, -74.88, -5.0],
                                             [0.32, 0.32, 8.0], 1, 500, 0.1, 2)
    cls = g['gt_boxes'][:, :, 7]
    for b in range(cls.shape[0]):
        n_a = int(((cls[b] == 1) | (cls[b] == 4)).sum())
        assert int((ia[b, n_a:] != 0).sum()) == 0 and int(ma[b].sum()) <= n_a
    assert torch.equal(ha[:, 0], heat[:, 0]) and torch.equal(ha[:, 1], heat[:, 3])
    # empty label set
    h0, _, _, m0 = ops.centerhead_targets(torch.zeros((2, 0, 8), device=dev()), cmap, 5, (64, 64), [-74.88, -74.88, -5.0],
                                          [0.32, 0.32, 8.0], 1, 500, 0.1, 2)
    assert float(h0.abs().sum()) == 0 and int(m0.sum()) == 0


def test_focal_loss_golden():
    from tmae_amd import ops
    g1, g2 = golden('G1_centerhead_targets'), golden('G2_centerhead_losses')
    gen = torch.Generator().manual_seed(int(g2['pred_seed']))
    p = torch.rand(3, 5, 468, 468, generator=gen) * 0.98 + 0.01
    logits = torch.log(p / (1 - p)).to(dev()).requires_grad_(True)            # sigmoid(logits) = p
    heat = cu(_dense_heat(g1))
    loss = ops.focal_loss_centernet(logits, heat)
    loss.backward()
    assert abs(float(loss) - float(g2['focal'])) <= 2e-5 * abs(float(g2['focal']))
    # d loss / d p = d loss / d logits / (p (1 - p))
    gp = (logits.grad.cpu() / (p * (1 - p))).flatten()[torch.from_numpy(g2['grad_probe_index'])].numpy()
    np.testing.assert_allclose(gp, g2['grad_probe_hm'], rtol=2e-4, atol=2e-8)
    # no positives: loss = - sum of the negative terms; clamped logits carry no gradient
    x = torch.tensor([[-20.0, 0.3, 25.0]], device=dev(), requires_grad=True)
    t = torch.tensor([[0.0, 0.5, 0.2]], device=dev())
    l0 = ops.focal_loss_centernet(x, t)
    l0.backward()
    pc = torch.clamp(torch.sigmoid(x.detach()), 1e-4, 1 - 1e-4)
    ref = -(torch.log(1 - pc) * pc ** 2 * (1 - t) ** 4).sum()
    assert abs(float(l0) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert float(x.grad[0, 0]) == 0.0 and float(x.grad[0, 2]) == 0.0 and float(x.grad[0, 1]) != 0.0


def test_finetune_e2e_golden_and_oracle(oracle, ft_oracle):
    """Whole fine-tune training step (TemporalDynVFE -> SiamWCA -> SSTBEVBackbone -> CenterHead loss) in fp32 vs the
    loss captured from the reference and per-parameter gradients vs the CPU oracle."""
    g = golden('G3_finetune_e2e_3stage')
    cfg = ft_oracle.default_finetune_cfg(3)
    P = ft_oracle.init_finetune_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']))
    bs = int(g['batch_size'])
    model, _, _ = build_finetune_model(params=P, device=dev())
    model.train()
    bd = {'points': cu(g['points']), 'points_prev': cu(g['points_prev']), 'batch_size': bs, 'gt_boxes': cu(g['gt_boxes'])}
    ret, tb, _ = model(bd)
    ret['loss'].backward()
    ref = float(g['loss'])
    assert abs(float(ret['loss']) - ref) <= 1e-4 * max(1.0, abs(ref)), (float(ret['loss']), ref)
    assert abs(float(tb['hm_loss_head_0']) - float(g['hm_loss'])) <= 1e-4 * max(1.0, float(g['hm_loss']))
    assert abs(float(tb['loc_loss_head_0']) - float(g['loc_loss'])) <= 1e-4 * max(1.0, float(g['loc_loss']))
    sf = bd['spatial_features_2d'].detach().double()
    assert float(sf.sum()) == pytest.approx(float(g['x2d_checksum']), rel=1e-4, abs=1.0)
    assert float(sf.abs().sum()) == pytest.approx(float(g['x2d_abs_checksum']), rel=1e-4)
    grads = dict(model.named_parameters())
    # Gradient bars.  This step is ill-conditioned in fp32: eight dense BatchNorm layers over a BEV map whose inactive
    # region is constant amplify summation-order noise -- the CPU oracle itself moves by up to 3 % in gradient norms
    # and 11 % of the largest entry between 1 and 8 threads (the loss by 5e-5).  The bars below are that noise floor;
    # the kernels' own unit tests (and the pre-training e2e tests, 5e-3) carry the tight gradient checks.
    for n, gn in zip(g['grad_names'], g['grad_norms']):
        assert abs(float(grads[str(n)].grad.norm()) - gn) <= 6e-2 * max(1.0, gn), n
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lo = ft_oracle.finetune_loss(Pg, g['points'], g['points_prev'], g['gt_boxes'], bs, cfg)
    lo.backward()
    for n in g['grad_names']:
        a, b = grads[str(n)].grad.cpu(), Pg[str(n)].grad
        assert (a - b).abs().max().item() <= 0.25 * max(1.0, b.abs().max().item()), n
        cos = torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0)
        assert float(cos) > 0.995 or float(b.norm()) < 1e-3, (n, float(cos))


def test_finetune_bf16_step_runs_and_is_close(ft_oracle):
    g = golden('G3_finetune_e2e_3stage')
    cfg = ft_oracle.default_finetune_cfg(3)
    P = ft_oracle.init_finetune_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']))
    model, _, _ = build_finetune_model(params=P, device=dev())
    model.train()
    bd = {'points': cu(g['points']), 'points_prev': cu(g['points_prev']), 'batch_size': int(g['batch_size']),
          'gt_boxes': cu(g['gt_boxes'])}
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ret, _, _ = model(bd)
    ret['loss'].backward()
    assert torch.isfinite(ret['loss'])
    assert abs(float(ret['loss']) - float(g['loss'])) <= 0.05 * abs(float(g['loss']))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
