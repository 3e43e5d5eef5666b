"""Probe: how fast do the library GEMMs of the step run on this box (bf16, shapes of the T-MAE layers)?
Usage: python profiles/scripts/gemm_probe.py [hipblaslt|rocblas|tunable]"""
import os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else 'hipblaslt'
if mode == 'tunable':
    os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '1'
    os.environ['PYTORCH_TUNABLEOP_TUNING'] = '1'
    os.environ['PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS'] = '30'
    os.environ['PYTORCH_TUNABLEOP_FILENAME'] = '/tmp/tunableop.csv'
import torch
import torch.nn.functional as F
if mode == 'rocblas':
    torch.backends.cuda.preferred_blas_library('cublas')
dev = torch.device('cuda:0')
shapes = [(470000, 128, 256), (470000, 128, 128), (470000, 256, 128), (466000, 256, 512), (466000, 256, 256),
          (466000, 512, 256), (195000, 256, 512), (195000, 512, 256), (470000, 1152, 128), (466000, 2304, 256), (195000, 2304, 256)]
print(mode)
for (M, K, N) in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for name, fn in (('fwd x@W^T+b', lambda: F.linear(x, w, b)), ('dX dy@W', lambda: dy @ w)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        flops = 2 * M * K * N
        byts = 2 * (M * K + M * N + K * N)
        res.append(f'{name}: {ms*1e3:7.1f} us {flops/ms/1e9:6.1f} TF/s {byts/ms/1e6:6.0f} GB/s')
    print(f'M={M} K={K} N={N} | ' + ' | '.join(res), flush=True)

# ---- our token GEMM on the eligible shapes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
from tmae_amd import ops
print('token_gemm (csrc/token_gemm.hip)')
for (M, K, N) in [(470000, 128, 256), (470000, 128, 128), (470000, 256, 128), (466000, 256, 512), (466000, 256, 256),
                  (195000, 256, 512), (195000, 256, 256), (94000, 128, 256), (466000, 512, 256), (195000, 512, 256),
                  (466000, 256, 2304)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.token_gemm(x, w, b, force=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.token_gemm(x, w, b, force=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byts = 2 * (M * K + M * N + K * N)
    print(f'M={M} K={K} N={N} | ours: {ms*1e3:7.1f} us {2*M*K*N/ms/1e9:6.1f} TF/s {byts/ms/1e6:6.0f} GB/s', flush=True)
