"""Times the halo-tiled dense 3x3 conv (tmae_dense_conv3x3 / _dilated) on the step's shapes: HIP events, 20 launches each.
A/B of two builds: TMAE_LIB_PATH=<other libtmae .so> python3 profiles/scripts/halo_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
from tmae_amd import ops
dev = torch.device('cuda')
B, Y, X = 8, 468, 468


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


out = []
for cin, cout, dil in ((384, 128, 1), (128, 384, 1), (128, 128, 1), (128, 128, 2)):
    x = torch.randn(B, Y, X, cin, device=dev).bfloat16()
    w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).bfloat16()
    t = timeit(lambda: ops.dense_conv3x3_halo(x, w, dil))
    out.append(f'{cin}->{cout} d{dil}: {t:.3f} ms ({2.0 * B * Y * X * 9 * cin * cout / t / 1e9:.0f} TF/s)')
    del x, w
print(os.path.basename(os.environ.get('TMAE_LIB_PATH', 'libtmae_hip.so')), ' | '.join(out))
