"""Times csrc/spconv_igemm.hip on the step's heavy shapes (HIP events, 20 launches each):
  dense decoder conv forward   1.75 M cells x (9 x 384) -> 128      (full-grid rulebook)
  dense decoder conv dX        1.75 M cells x (9 x 128) -> 384
  stage-2 submanifold conv     466 k tokens x (9 x 256) -> 256      (random occupancy ~ the step's)
and, next to them, the library path they replace (MIOpen conv / gather + hipBLASLt GEMM)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
for _k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
    os.environ.setdefault(_k, '0')
from tmae_amd import ops
dev = torch.device('cuda')


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


B, Y, X = 8, 468, 468
n = B * Y * X
nbr = ops._dense_rulebook(B, Y, X, dev)
nbr_t = nbr.flip(1).contiguous()
x = torch.randn(n, 384, device=dev).bfloat16()
w = (torch.randn(128, 9 * 384, device=dev) * 0.02).bfloat16()
dy = torch.randn(n, 128, device=dev).bfloat16()
fl = 2.0 * n * 9 * 384 * 128
t = timeit(lambda: ops.spconv_fwd(x, nbr, w))
print(f'dense fwd  native {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s')
t = timeit(lambda: ops.spconv_bwd_data(dy, nbr_t, w, 384))
print(f'dense dX   native {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s')
x4 = x.view(B, Y, X, 384)
t = timeit(lambda: ops.dense_conv3x3_halo(x4, w))
print(f'dense fwd  halo   {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s')
wt = w.view(128, 3, 3, 384).flip(1, 2).permute(3, 1, 2, 0).reshape(384, 9 * 128).contiguous()
dy4 = dy.view(B, Y, X, 128)
t = timeit(lambda: ops.dense_conv3x3_halo(dy4, wt))
print(f'dense dX   halo   {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s')
ref = ops.spconv_fwd(x, nbr, w).view(B, Y, X, 128)
got = ops.dense_conv3x3_halo(x4, w)
print('halo vs ring fwd: max abs diff', float((ref.float() - got.float()).abs().max()), 'equal', bool(torch.equal(ref, got)))
ref = ops.spconv_bwd_data(dy, nbr_t, w, 384).view(B, Y, X, 384)
got = ops.dense_conv3x3_halo(dy4, wt)
print('halo vs ring dX : max abs diff', float((ref.float() - got.float()).abs().max()), 'equal', bool(torch.equal(ref, got)))
del x4, dy4, ref, got
xc = x.view(B, Y, X, 384).permute(0, 3, 1, 2)
wc = w.view(128, 3, 3, 384).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
t = timeit(lambda: torch.nn.functional.conv2d(xc, wc, padding=1), 5)
print(f'dense fwd  MIOpen {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s')
del x, dy, xc
rng = np.random.default_rng(0)
occ = rng.random((B, 234, 234)) < 0.53
ind = torch.from_numpy(np.ascontiguousarray(np.argwhere(occ).astype(np.int32))).to(dev).contiguous()
m = ind.shape[0]
grid = ops.index_grid(ind, B, 234, 234)
nb2 = ops.spconv_neighbors(ind, grid, B, 234, 234, 1)
f = torch.randn(m, 256, device=dev).bfloat16()
w2 = (torch.randn(256, 9 * 256, device=dev) * 0.02).bfloat16()
fl = 2.0 * m * 9 * 256 * 256
t = timeit(lambda: ops.spconv_fwd(f, nb2, w2))
print(f'subm 256   native {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s (dense-equivalent FLOPs, {m} tokens, '
      f'{float((nb2 >= 0).float().mean()) * 9:.1f} of 9 taps present)')
t = timeit(lambda: ops._gather9(f, nb2) @ w2.t())
print(f'subm 256   gather + hipBLASLt {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s')
