"""Timing of the weight-gradient launches whose block mapping round 6 changed (csrc/wgrad.hip): the gathered sparse-conv gradient
(n = 256, k = 9 x 256 / 9 x 128: 9 / 5 output blocks per token chunk) and the self-attention in-projection (n = 768: 3 blocks), plus
two shapes the change must not move.  Library under test: TMAE_LIB_PATH (profiles/scripts/ab_lib.sh style).
usage (GPU box): python3 profiles/scripts/wgrad_ab_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
from tmae_amd._lib import lib, check
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream


def timeit(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.manual_seed(0)
for m, cin, cout in ((466000, 256, 256), (318000, 256, 256), (148000, 256, 256), (86000, 256, 256), (466000, 128, 256)):
    feat = torch.randn(m, cin, device=dev).bfloat16()
    dy = torch.randn(m, cout, device=dev).bfloat16()
    # a submanifold-like rulebook: neighbours of row r are rows near r (x +- 1) and near r +- 500 (y +- 1), 70 % present
    r = torch.arange(m, device=dev)
    cols = []
    for t in range(9):
        off = (t // 3 - 1) * 500 + (t % 3 - 1)
        j = r + off
        ok = (j >= 0) & (j < m) & ((torch.rand(m, device=dev) < 0.7) | (t == 4))
        cols.append(torch.where(ok, j, torch.full_like(j, -1)))
    nbr = torch.stack(cols, 1).int().contiguous()
    dw = torch.empty(cout, 9 * cin, device=dev)
    wsb = lib.tmae_linear_wgrad_workspace(m, cout, 9 * cin)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    us = timeit(lambda: check(lib.tmae_spconv_wgrad(dy.data_ptr(), cout, feat.data_ptr(), cin, nbr.data_ptr(), m, cout, cin, dw.data_ptr(),
                                                    ws.data_ptr(), wsb, st), 'spconv_wgrad'))
    print(f'spconv wgrad m={m:7d} {cin}->{cout}: {us:7.1f} us  {2.0 * m * 9 * cin * cout * 0.7 / us / 1e6:6.0f} TFLOP/s (70 % pairs)', flush=True)
for m, n, k in ((466000, 768, 256), (318000, 768, 256), (466000, 512, 256), (466000, 256, 512), (466000, 256, 256), (470000, 384, 128)):
    dy = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    dw = torch.empty(n, k, device=dev)
    db = torch.empty(n, device=dev)
    wsb = lib.tmae_linear_wgrad_workspace(m, n, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    us = timeit(lambda: check(lib.tmae_linear_wgrad(dy.data_ptr(), n, x.data_ptr(), k, m, n, k, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, st),
                              'linear_wgrad'))
    print(f'linear wgrad m={m:7d} n={n:4d} k={k:4d}: {us:7.1f} us  {m * (n + k) * 2 / us / 1e6:6.0f} GB/s', flush=True)
