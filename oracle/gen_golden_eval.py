"""Pins oracle/eval_oracle.py against the UNMODIFIED reference evaluation and writes
tests/golden/G5_once_eval.npz.  TEST INFRASTRUCTURE, build container only.

Reference code run here: pcdet/datasets/once_temporal/once_eval/evaluation.py (get_evaluation_results and everything it
calls) and once_eval/eval_utils.py, loaded by path.  Stand-ins for what the image lacks: `numba` (its decorators become
identities: the functions run as plain Python), the removed alias `np.bool` (eval_utils.py:13 predates numpy 1.24),
and -- the one numerical piece -- the numba.cuda kernel `rotate_iou_gpu_eval`, replaced by eval_oracle.bev_intersection
(PARITY UNPINNED, see that file's header).
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import eval_oracle as EO                      # noqa: E402
from gen_golden import save                   # noqa: E402

REF = '/root/reference/pcdet/datasets/once_temporal/once_eval'


def load_reference_eval():
    if not hasattr(np, 'bool'):
        np.bool = np.bool_
    nb = types.ModuleType('numba')
    ident = lambda *a, **k: (a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f))   # noqa: E731
    nb.jit = ident
    nb.float32 = np.float32
    cuda = types.ModuleType('numba.cuda')
    cuda.jit = ident
    nb.cuda = cuda
    sys.modules['numba'], sys.modules['numba.cuda'] = nb, cuda
    pkg = types.ModuleType('once_eval')
    pkg.__path__ = [REF]
    sys.modules['once_eval'] = pkg
    out = {}
    for name in ('iou_utils', 'eval_utils', 'evaluation'):
        spec = importlib.util.spec_from_file_location('once_eval.' + name, os.path.join(REF, name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules['once_eval.' + name] = mod
        spec.loader.exec_module(mod)
        out[name] = mod
    out['evaluation'].rotate_iou_gpu_eval = lambda b, q, criterion=-1, device_id=0: EO.bev_intersection(b, q)
    return out


def main():
    ref = load_reference_eval()
    ev = ref['evaluation']
    classes = ['Car', 'Bus', 'Truck', 'Pedestrian', 'Cyclist']
    gts, preds = EO.synth_annos(24, seed=4)
    import copy
    ret_str, ret = ev.get_evaluation_results(copy.deepcopy(gts), copy.deepcopy(preds), list(classes))
    mine, AP, ious = EO.get_evaluation_results(gts, preds, classes)
    assert set(mine) == set(ret), (sorted(mine), sorted(ret))
    for k in ret:
        a, b = float(ret[k]), float(mine[k])
        assert (np.isnan(a) and np.isnan(b)) or a == b, (k, a, b)
    # the helper functions one by one, on every (class, difficulty) of sample 0..n
    for s in range(len(gts)):
        for cur in ('Vehicle', 'Pedestrian', 'Cyclist'):
            for d in range(4):
                g1, p1 = ev.filter_data(gts[s], preds[s], 'Overall&Distance', d, cur, True)
                g2, p2 = EO.filter_data(gts[s], preds[s], d, cur)
                assert np.array_equal(g1, g2) and np.array_equal(p1, p2)
                thr = EO.SUPERCLASS_IOU_THRESHOLDS[cur]
                sc = preds[s]['score']
                a1 = ev.accumulate_scores(ious[s], sc, g1, p1, iou_threshold=thr)
                a2 = EO.accumulate_scores(ious[s], sc, g2, p2, thr)
                assert np.array_equal(a1, a2)
                for t in (0.0, 0.3, 0.6):
                    assert tuple(ev.compute_statistics(ious[s], sc, g1, p1, score_threshold=t, iou_threshold=thr)) == \
                        EO.compute_statistics(ious[s], sc, g2, p2, t, thr)
    print(ret_str)
    out = dict(classes=np.array(classes), num_samples=len(gts), keys=np.array(sorted(ret)),
               values=np.array([float(ret[k]) for k in sorted(ret)]), AP=AP)
    for s, (g, p) in enumerate(zip(gts, preds)):
        out[f'gt_name_{s}'], out[f'gt_box_{s}'] = g['name'], g['boxes_3d']
        out[f'pd_name_{s}'], out[f'pd_box_{s}'], out[f'pd_score_{s}'] = p['name'], p['boxes_3d'], p['score']
        out[f'iou_{s}'] = ious[s]
    save('G5_once_eval', **out)
    print('oracle == reference on', len(ret), 'AP entries')


if __name__ == '__main__':
    main()
