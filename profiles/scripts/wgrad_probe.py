"""Per-shape timing of the weight-gradient kernel (csrc/wgrad.hip) against torch's GEMM on the shapes of one
T-MAE step.  usage (GPU box): python profiles/scripts/wgrad_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
from tmae_amd import ops

shapes = [(466000, 512, 256), (466000, 256, 256), (466000, 256, 512), (466000, 768, 256), (752000, 256, 128), (752000, 128, 128), (376000, 256, 128), (376000, 128, 128), (376000, 128, 256), (113000, 256, 128), (113000, 128, 128),
          (150000, 512, 256), (150000, 256, 256), (150000, 256, 512), (150000, 512, 512), (60000, 512, 256), (60000, 256, 256),
          (50000, 512, 256), (50000, 256, 256), (376000, 128, 64), (376000, 64, 16), (376000, 2048, 256)]
dev = 'cuda:0'
for m, n, k in shapes:
    dy = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    def ours():
        return ops.linear_wgrad(dy, x, want_bias=True)
    def lib():
        return dy.t() @ x
    res = []
    for f in (ours, lib):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    byt = m * (n + k) * 2
    print(f'm={m:7d} n={n:4d} k={k:4d}  ours {res[0]*1e3:7.1f} us ({byt/res[0]/1e6:6.0f} GB/s)   '
          f'hipBLASLt {res[1]*1e3:7.1f} us ({byt/res[1]/1e6:6.0f} GB/s)', flush=True)
