"""CPU: the fine-tune oracle (oracle/finetune_oracle.py) against the golden vectors captured from the reference's
CenterHead / SSTBEVBackbone / SiamWCA / loss_utils / centernet_utils (tests/golden/G*.npz, written by
oracle/gen_golden_finetune.py)."""
import numpy as np
import torch

from conftest import golden, fl


def _dense_heat(g):
    hm = np.zeros(tuple(g['heat_shape']), np.float32)
    ix = g['heat_nz_index']
    hm[ix[:, 0], ix[:, 1], ix[:, 2], ix[:, 3]] = g['heat_nz_value']
    return hm


def test_g1_centerhead_targets(ft_oracle):
    g = golden('G1_centerhead_targets')
    cfg = ft_oracle.default_finetune_cfg(3)
    t = ft_oracle.assign_targets(torch.from_numpy(g['gt_boxes']), (468, 468), cfg)
    assert np.array_equal(t['heatmaps'][0].numpy(), _dense_heat(g))
    assert np.array_equal(t['inds'][0].numpy(), g['inds']) and np.array_equal(t['masks'][0].numpy(), g['masks'])
    assert np.array_equal(t['target_boxes'][0].numpy(), g['target_boxes'])
    # the degenerate / clamped boxes of the fixture: dx = 0 is skipped, a centre outside the range lands in the last cell
    assert t['masks'][0][1].sum() < (g['gt_boxes'][1, :, -1] > 0).sum()
    assert int(t['inds'][0][0, 1]) % 468 == 467


def test_g2_losses(ft_oracle):
    g1, g2 = golden('G1_centerhead_targets'), golden('G2_centerhead_losses')
    gen = torch.Generator().manual_seed(int(g2['pred_seed']))
    pred_hm = (torch.rand(3, 5, 468, 468, generator=gen) * 0.98 + 0.01).requires_grad_(True)
    pred_box = torch.randn(3, 8, 468, 468, generator=gen).requires_grad_(True)
    heat = torch.from_numpy(_dense_heat(g1))
    fl = ft_oracle.focal_loss_centernet(pred_hm, heat)
    rl = ft_oracle.reg_loss_centernet(pred_box, torch.from_numpy(g1['masks']), torch.from_numpy(g1['inds']),
                                      torch.from_numpy(g1['target_boxes']))
    (fl + rl.sum()).backward()
    assert abs(float(fl.detach()) - float(g2['focal'])) <= 1e-5 * abs(float(g2['focal']))
    np.testing.assert_allclose(rl.detach().numpy(), g2['reg'], atol=1e-6)
    np.testing.assert_allclose(pred_hm.grad.flatten()[torch.from_numpy(g2['grad_probe_index'])].numpy(), g2['grad_probe_hm'],
                               atol=1e-7)
    gb = pred_box.grad.flatten()
    nz = gb.nonzero()[:, 0]
    assert np.array_equal(nz.numpy(), g2['grad_box_nz_index'])
    np.testing.assert_allclose(gb[nz].numpy(), g2['grad_box_nz'], atol=1e-8)


def test_g3_finetune_e2e(ft_oracle):
    g = golden('G3_finetune_e2e_3stage')
    cfg = ft_oracle.default_finetune_cfg(3)
    P = {k: v.requires_grad_(True) for k, v in ft_oracle.init_finetune_params(cfg, seed=int(g['param_seed']), tau=float(g['tau'])).items()}
    cap = {}
    loss = ft_oracle.finetune_loss(P, g['points'], g['points_prev'], g['gt_boxes'], int(g['batch_size']), cfg, cap)
    loss.backward()
    assert abs(fl(loss) - float(g['loss'])) <= 2e-5 * max(1.0, abs(float(g['loss'])))
    assert abs(fl(cap['parts']['hm_loss_head_0']) - float(g['hm_loss'])) <= 2e-5 * max(1.0, float(g['hm_loss']))
    assert abs(fl(cap['parts']['loc_loss_head_0']) - float(g['loc_loss'])) <= 2e-5 * max(1.0, float(g['loc_loss']))
    for n, gn in zip(g['grad_names'], g['grad_norms']):
        assert abs(float(P[str(n)].grad.norm()) - gn) <= 3e-3 * max(1.0, gn), n


def test_g4_decode(ft_oracle):
    g = golden('G4_decode')
    B, C, Hh, Ww, K = (int(v) for v in g['shape'])
    gen = torch.Generator().manual_seed(int(g['seed']))
    hm = torch.rand(B, C, Hh, Ww, generator=gen) ** 6
    center, cz = torch.rand(B, 2, Hh, Ww, generator=gen), torch.randn(B, 1, Hh, Ww, generator=gen) - 1
    dim = torch.rand(B, 3, Hh, Ww, generator=gen) * 3 + 0.5
    rc, rs = torch.randn(B, 1, Hh, Ww, generator=gen), torch.randn(B, 1, Hh, Ww, generator=gen)
    out = ft_oracle.decode_bbox_from_heatmap(hm, rc, rs, center, cz, dim, [-74.88, -74.88, -5.0], [0.32, 0.32, 8.0], 1, K,
                                             float(g['score_thresh']), g['limit'])
    for k in range(B):
        assert np.array_equal(out[k]['pred_boxes'].numpy(), g[f'pred_boxes_{k}'])
        assert np.array_equal(out[k]['pred_scores'].numpy(), g[f'pred_scores_{k}'])
        assert np.array_equal(out[k]['pred_labels'].numpy(), g[f'pred_labels_{k}'])


def test_rotated_iou_known_answers(ft_oracle):
    """The rotated-IoU restatement is unpinned against the reference (CUDA-only there): closed-form cases."""
    a = [0, 0, 0, 4, 2, 1, 0.0]
    assert ft_oracle.iou_bev(a, a) == 1.0 and ft_oracle.iou_bev(a, [10, 0, 0, 4, 2, 1, 0.3]) == 0.0
    assert abs(ft_oracle.overlap_bev([0, 0, 0, 2, 2, 1, np.pi / 4], [0, 0, 0, 2, 2, 1, 0]) - 8 * (np.sqrt(2) - 1)) < 1e-12
    assert abs(ft_oracle.overlap_bev(a, [1, 0, 0, 4, 2, 1, 0]) - 6.0) < 1e-12
    assert abs(ft_oracle.overlap_bev(a, [0, 0, 0, 4, 2, 1, np.pi / 2]) - 4.0) < 1e-12
    assert abs(ft_oracle.iou3d([a], [[0, 0, 0.5, 4, 2, 1, 0]])[0, 0] - 1 / 3) < 1e-12
    keep = ft_oracle.nms_bev(np.array([a, [0.2, 0, 0, 4, 2, 1, 0.0], [10, 0, 0, 4, 2, 1, 0.0]]), [0.5, 0.9, 0.1], 0.5)
    assert keep.tolist() == [1, 2]
