"""Generate tests/golden/F14_options.npz from the reference itself (build container only): configuration options of the
hot path that the shipped YAMLs leave off and this repository supports anyway.

TEST INFRASTRUCTURE (see gen_golden.py).  Runs the unmodified reference module (via oracle/ref_import.py), checks this repo's CPU
oracle against every captured value and stores inputs + expected outputs as data.
  F14a  SSTInputLayer.get_pos_embed with NORMALIZE_POS: True (spt_backbone.py:186-224, :202-204), d = 128 / 256.
Usage:  python oracle/gen_golden_options.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R      # noqa: E402
import tmae_oracle as O     # noqa: E402
from gen_golden import check, save      # noqa: E402


if __name__ == '__main__':
    V, B, cfg = R.build_reference_model(num_stages=3, seed=0)
    il = B.sst_blocks[0].sst_input_layer
    ciw = np.stack([np.zeros(64, np.int64), np.repeat(np.arange(8), 8), np.tile(np.arange(8), 8)], 1)
    out = dict(coors_in_win=ciw, pos_temperature=np.float64(il.pos_temperature))
    assert il.normalize_pos is False
    il.normalize_pos = True                       # the attribute NORMALIZE_POS sets (spt_backbone.py:35)
    for d in (128, 256):
        f2w = {0: (torch.arange(64), (torch.arange(64),)), 'voxel_drop_level': torch.zeros(64, dtype=torch.long),
               'batching_info': {0: {'max_tokens': 64, 'drop_range': (0, 100000)}}}
        pe = il.get_pos_embed(f2w, torch.from_numpy(ciw), d)[0][0]
        check('pos (normalised)', O.pos_embed(ciw, d, (8, 8, 1), il.pos_temperature, normalize_pos=True), pe, 1e-6)
        out[f'pos_norm_{d}'] = pe
    il.normalize_pos = False
    save('F14_options', **out)
