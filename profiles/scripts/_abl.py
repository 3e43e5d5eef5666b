import os, sys, torch
sys.path.insert(0, 't-mae_amd')
from tmae_amd import ops
m, n, k = 466000, 512, 256
dy = torch.randn(m, n, device='cuda').bfloat16(); x = torch.randn(m, k, device='cuda').bfloat16()
for _ in range(3): ops.linear_wgrad(dy, x, want_bias=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.linear_wgrad(dy, x, want_bias=True)
e1.record(); torch.cuda.synchronize()
print(os.environ.get('TMAE_WGRAD_CYCLIC'), f'{e0.elapsed_time(e1)/20*1e3:.1f} us')
