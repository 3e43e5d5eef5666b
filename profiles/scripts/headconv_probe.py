"""Probe: the narrow head convs (csrc/headconv.hip) against the library on CenterHead's map [8, 468, 468, 64] -> k channels:
forward, input gradient, weight gradient, us per launch (HIP events, 20 launches).  python3 profiles/scripts/headconv_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
import torch
from tmae_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


B, Y, X = 8, 468, 468
x = torch.randn(B, Y, X, 64, device=dev).bfloat16().permute(0, 3, 1, 2)
mb = x.numel() * 2 / 1e6
print(f'map {B}x{Y}x{X}x64 bf16 = {mb:.0f} MB')
for k in (1, 2, 3, 5):
    conv = torch.nn.Conv2d(64, k, 3, padding=1, bias=True).to(dev)
    go = torch.randn(B, Y, X, k, device=dev).bfloat16().permute(0, 3, 1, 2)
    row = [f'k={k}']
    for mode in ('native', 'library'):
        os.environ['TMAE_HEAD_CONV'] = mode
        xa = x.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            tf = timeit(lambda: ops.conv3x3_channel_bias(xa, conv))
            y = ops.conv3x3_channel_bias(xa, conv)
        tb = timeit(lambda: torch.autograd.grad(y, (xa, conv.weight, conv.bias), go, retain_graph=True))
        row.append(f'{mode}: fwd {tf:7.1f} us ({mb / tf * 1e3:6.0f} GB/s)  bwd (dx + dw + db) {tb:7.1f} us')
    print('   '.join(row))

# the 64 -> 64 stems (conv + training-mode BatchNorm + ReLU through the module path; the conv alone through the op)
import torch.nn as nn
conv = nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev)
xn = x.permute(0, 2, 3, 1)
go = torch.randn(B, Y, X, 64, device=dev).bfloat16()
row = ['64->64']
for mode in ('native', 'library'):
    xa = xn.clone().requires_grad_(True)
    if mode == 'native':
        f = lambda: ops.conv3x3_c64(xa, conv.weight)
    else:
        f = lambda: torch.nn.functional.conv2d(xa.permute(0, 3, 1, 2), conv.weight.bfloat16().contiguous(memory_format=torch.channels_last),
                                               padding=1).permute(0, 2, 3, 1)
    tf = timeit(f)
    y = f()
    tb = timeit(lambda: torch.autograd.grad(y, (xa, conv.weight), go, retain_graph=True))
    row.append(f'{mode}: fwd {tf:7.1f} us  bwd (dx + dw) {tb:7.1f} us')
print('   '.join(row))
