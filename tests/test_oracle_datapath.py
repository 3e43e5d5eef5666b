"""CPU: the data-path oracle (oracle/datapath_oracle.py) against the batch captured from the reference's dataset
functions (tests/golden/D1_datapath.npz, written by oracle/gen_golden_datapath.py)."""
import numpy as np

from conftest import golden

PCR = [-74.88, -74.88, -5.0, 74.88, 74.88, 3.0]


def test_d1_datapath(dp_oracle):
    g = golden('D1_datapath')
    samples = []
    for i in range(int(g['n_samples'])):
        params = dict(flips=(['x'] if int(g[f'flip_x_{i}']) else []) + (['y'] if int(g[f'flip_y_{i}']) else []),
                      rot=float(g[f'rot_{i}']), scale=float(g[f'scale_{i}']))
        prv, cur = dp_oracle.prepare_pair(g[f'pts_{i}'], g[f'prv_{i}'], g[f'pose_cur_{i}'], g[f'pose_prv_{i}'], params,
                                          g[f'perm_{i}'], PCR)
        samples.append({'points_prev': prv, 'points': cur})
    c = dp_oracle.collate(samples)
    assert np.array_equal(c['points'], g['points']) and np.array_equal(c['points_prev'], g['points_prev'])
    assert c['batch_size'] == 3


def test_pose_quirks_and_draw_order(dp_oracle):
    # an all-zero pose means "static": that step is skipped (once_utils.py:9-10,17-18)
    p = np.array([[1.0, 2.0, 3.0, 0.5]], np.float32)
    same = dp_oracle.convert_prv_frame_to_cur(p, np.zeros(7), np.zeros(7))
    assert np.array_equal(same, p)
    moved = dp_oracle.convert_prv_frame_to_cur(p, np.array([0, 0, 0, 1.0, 1.0, 0, 0]), np.zeros(7))
    assert np.allclose(moved[0, :3], [2.0, 2.0, 3.0]) and moved[0, 3] == 0.5
    back = dp_oracle.convert_prv_frame_to_cur(p, np.zeros(7), np.array([0, 0, 0, 1.0, 1.0, 0, 0]))
    assert np.allclose(back[0, :3], [0.0, 2.0, 3.0])
    cfg = dict(flip_axes=['x', 'y'], flip_prob=0.5, rot_prob=1.0, rot_range=[-0.78539816, 0.78539816], scale_prob=1.0,
               scale_range=[0.95, 1.05])
    np.random.seed(3)
    a = dp_oracle.draw_params(cfg)
    np.random.seed(3)
    b = dp_oracle.draw_params(cfg)
    assert a == b and -0.78539816 <= a['rot'] <= 0.78539816 and 0.95 <= a['scale'] <= 1.05
