"""G3d: the fine-tune step's gradients in float64.  TEST INFRASTRUCTURE, build container only.

The fine-tune step (eight dense BatchNorm layers over BEV maps whose inactive region is one constant per channel) is
ill-conditioned in fp32: its weight gradients are sums of ~1e6 terms that cancel, and the CPU oracle ITSELF moves by up
to 3 % (8 threads) / 11 % (1 thread) of the largest entry between summation orders.  Comparing two fp32 runs therefore
says little; this fixture anchors the comparison to the SAME oracle function (pinned against the reference in fp32 by
gen_golden_finetune.py, G3) run in float64 on the G3 inputs: per parameter tensor its norm and its projections on four
seeded +-1 vectors, plus the error of the fp32 oracle run against them -- the yardstick for the GPU's bars."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import finetune_oracle as FO      # noqa: E402
from gen_golden import save       # noqa: E402

NPROJ = 4


def projections(name_index, g):
    """[norm, <g, r_1>, ..., <g, r_4>] in float64; r_j in {-1, +1}^numel from a generator seeded by the tensor's index."""
    g = g.detach().double().flatten().cpu()
    gen = torch.Generator().manual_seed(100003 * (name_index + 1))
    out = [float(g.norm())]
    for _ in range(NPROJ):
        r = torch.randint(0, 2, (g.numel(),), generator=gen, dtype=torch.int8).double() * 2 - 1
        out.append(float((g * r).sum()))
    return out


def main():
    g3 = np.load(os.path.join(HERE, '..', 'tests', 'golden', 'G3_finetune_e2e_3stage.npz'))
    cfg = FO.default_finetune_cfg(3)
    P = FO.init_finetune_params(cfg, seed=int(g3['param_seed']), tau=float(g3['tau']))
    bs = int(g3['batch_size'])
    names = [str(n) for n in g3['grad_names']]
    res = {}
    for dt in (torch.float64, torch.float32):
        Pg = {k: v.to(dt).clone().requires_grad_(True) for k, v in P.items()}
        lo = FO.finetune_loss(Pg, g3['points'], g3['points_prev'], g3['gt_boxes'], bs, cfg)
        lo.backward()
        res[dt] = (float(lo), np.array([projections(i, Pg[n].grad) for i, n in enumerate(names)]))
        print(dt, float(lo))
    p64, p32 = res[torch.float64][1], res[torch.float32][1]
    rel = np.abs(p32 - p64) / np.maximum(p64[:, :1], 1e-12)
    print('fp32 oracle vs float64: worst relative (of the norm)', rel.max(), 'median', np.median(rel.max(1)))
    save('G3d_finetune_grad64', names=np.array(names), loss64=res[torch.float64][0], loss32=res[torch.float32][0],
         proj64=p64, proj32=p32, nproj=NPROJ, threads=torch.get_num_threads())


if __name__ == '__main__':
    main()
