"""Probe: the W-in-registers token GEMM (csrc/token_gemm_wreg.hip) against the W-resident kernel it replaces and against
torch, on the shapes of the step.  Usage: python profiles/scripts/tgw_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
import torch
import torch.nn.functional as F
from tmae_amd import ops
from tmae_amd._lib import lib, check
dev = torch.device('cuda:0')
torch.manual_seed(0)


def run(x, w, b, y, cells=None):
    m, k = x.shape
    n = w.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    if cells is None:
        check(lib.tmae_token_gemm(x.data_ptr(), x.stride(0), m, k, w.data_ptr(), n, b.data_ptr(), y.data_ptr(), y.stride(0), st), 'tg')
    else:
        check(lib.tmae_token_gemm_pos(x.data_ptr(), x.stride(0), m, k, w.data_ptr(), n, b.data_ptr(), cells.data_ptr(),
                                      y.data_ptr(), y.stride(0), st), 'tgpos')


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
    return e0.elapsed_time(e1) / it


for (m, k, n, pos) in [(1000, 256, 512, False), (32768 + 37, 256, 512, False), (32768 + 37, 256, 256, False),
                       (70001, 256, 512, True), (70001, 256, 256, True), (70001, 256, 768, True), (40001, 128, 384, True),
                       (40001, 128, 128, False), (40001, 128, 256, False),
                       (466268, 256, 512, False), (466268, 256, 256, False), (195000, 256, 512, False),
                       (195000, 256, 256, False), (466268, 256, 768, True), (195000, 256, 768, True), (148000, 256, 256, True),
                       (148000, 256, 512, True), (470000, 128, 128, False), (470000, 128, 256, False), (470000, 128, 384, True),
                       (94000, 128, 128, True), (94000, 128, 256, True)]:
    x = torch.randn(m, k, device=dev).bfloat16()
    ka = k + (32 if pos else 0)
    w = (torch.randn(n, ka, device=dev) * 0.05).bfloat16()
    b = torch.randn(n, device=dev).bfloat16()
    cells = torch.randint(0, 64, (m,), device=dev, dtype=torch.uint8) if pos else None
    res = {}
    for mode in ('0', '1'):
        os.environ['TMAE_TG_WREG'] = mode
        y = torch.full((m, n), float('nan'), device=dev, dtype=torch.bfloat16)
        run(x, w, b, y, cells)
        torch.cuda.synchronize()
        ms = timeit(lambda: run(x, w, b, y, cells)) if m > 60000 else 0.0
        res[mode] = (y, ms)
    xf = x.float()
    if pos:
        oh = torch.zeros(m, 32, device=dev)
        xc, yc = (cells & 7).long(), (cells >> 3).long()
        for gk in range(4):
            sel = yc if gk & 1 else xc
            oh[torch.arange(m, device=dev), gk * 8 + sel] = 1.0
        xf = torch.cat([xf, oh], 1)
    ref = xf @ w.float().t() + b.float()
    e0 = (res['0'][0].float() - ref).abs().max().item()
    e1 = (res['1'][0].float() - ref).abs().max().item()
    same = torch.equal(res['0'][0], res['1'][0])
    byts = m * (k + n) * 2 + (n * ka + n) * 2
    line = f'm={m} k={k} n={n} pos={pos} | max|err| old {e0:.4f} new {e1:.4f} bit-identical {same}'
    if m > 60000:
        line += f" | old {res['0'][1]*1e3:7.1f} us {byts/res['0'][1]/1e6:6.0f} GB/s | new {res['1'][1]*1e3:7.1f} us {byts/res['1'][1]/1e6:6.0f} GB/s"
    print(line, flush=True)
    assert e1 <= max(2 * e0, 0.08), 'new kernel deviates'


# ---- contraction 512 (FFN-2 forward) and the in-place accumulate form (FFN-1 input gradients) vs torch
import torch.nn.functional as F
for (m, k, n) in [(40001, 512, 256), (466268, 512, 256), (195000, 512, 256)]:
    x = torch.randn(m, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) * 0.05).bfloat16(); b = torch.randn(n, device=dev).bfloat16()
    ref = x.float() @ w.float().t() + b.float()
    os.environ['TMAE_TG_WREG'] = '1'
    y = ops.token_gemm(x, w, b)
    err = (y.float() - ref).abs().max().item()
    t_new = timeit(lambda: ops.token_gemm(x, w, b)); t_lib = timeit(lambda: F.linear(x, w, b))
    byts = m * (k + n) * 2
    print(f'm={m} k={k} n={n} | max|err| {err:.4f} | hipBLASLt {t_lib*1e3:7.1f} us | wreg {t_new*1e3:7.1f} us {byts/t_new/1e6:6.0f} GB/s', flush=True)
    assert err <= 0.08
for (m, n, k) in [(40001, 512, 256), (466268, 512, 256), (195000, 512, 256), (470000, 256, 128), (50001, 256, 128)]:
    dy = torch.randn(m, n, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    dx0 = torch.randn(m, k, device=dev).bfloat16()
    ref = dx0.float() + dy.float() @ w.float()
    dx = dx0.clone(); ops.addmm_inplace(dx, dy, w)
    err = (dx.float() - ref).abs().max().item()
    dx1 = dx0.clone(); dx2 = dx0.clone()
    t_new = timeit(lambda: ops.addmm_inplace(dx1, dy, w)); t_lib = timeit(lambda: dx2.addmm_(dy, w))
    byts = m * (n + 2 * k) * 2
    print(f'acc m={m} {n}->{k} | max|err| {err:.4f} | addmm_ {t_lib*1e3:7.1f} us | wreg {t_new*1e3:7.1f} us {byts/t_new/1e6:6.0f} GB/s', flush=True)
    assert err <= 0.1
