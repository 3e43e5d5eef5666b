"""Timing of the weight gradient WITH per-cell sums (tmae_linear_wgrad_cells) against the plain one (tmae_linear_wgrad) on the
in-projection shapes of one step.  usage (GPU box): python profiles/scripts/wgrad_cells_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
from tmae_amd import ops

dev = 'cuda:0'
for m, n, k, pos_n in ((466000, 512, 256, 512), (150000, 512, 256, 512), (466000, 512, 256, 256), (752000, 256, 128, 256),
                       (466000, 256, 256, 256)):
    ind = torch.stack([torch.zeros(m, dtype=torch.int64), torch.randint(0, 468, (m,)), torch.randint(0, 468, (m,))], 1).int().to(dev)
    cells = ops.window_cells(ind, [8, 8, 1], False)
    dy = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    res = []
    for f in (lambda: ops.linear_wgrad(dy, x, True), lambda: ops.linear_wgrad(dy, x, True, cells=cells, pos_n=pos_n)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    byt = m * (n + k) * 2
    print(f'm={m:7d} n={n:4d} k={k:4d} pos_n={pos_n:4d}  plain {res[0]*1e3:7.1f} us ({byt/res[0]/1e6:6.0f} GB/s)   '
          f'with cell sums {res[1]*1e3:7.1f} us ({byt/res[1]/1e6:6.0f} GB/s)', flush=True)
