"""Timing of the 256-tile weight-gradient kernel on its heavy shapes for one build / variant (A/B: run under
TMAE_LIB_PATH=<debug build> TMAE_WGRAD_VAR=<n>).  usage (GPU box): python profiles/scripts/wgrad_var_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
from tmae_amd import ops
dev = 'cuda:0'
shapes = [(466000, 512, 256), (466000, 256, 512), (466000, 768, 256), (195000, 512, 256), (195000, 256, 512), (195000, 768, 256)]
ref = {}
out = []
for m, n, k in shapes:
    torch.manual_seed(m + n)
    dy = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    for _ in range(3):
        dw, db = ops.linear_wgrad(dy, x, want_bias=True)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear_wgrad(dy, x, want_bias=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    chk = float(dw.double().abs().sum())
    out.append(f'm={m} n={n} k={k}: {min(ts)*1e3:7.1f} us (min of 3 x 20)  checksum {chk:.6e}')
print(os.environ.get('TMAE_WGRAD_VAR', '0'), *out, sep='\n  ')
