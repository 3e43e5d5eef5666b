"""Timing of the token GEMM on the heavy shapes: TMAE_TG_WREG=0 (W in LDS) vs 1 (W in registers); TMAE_LIB_PATH picks a build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
import torch
from tmae_amd._lib import lib, check
dev = torch.device('cuda:0')
torch.manual_seed(0)
out = []
for (m, k, n, pos) in [(466268, 256, 512, False), (466268, 256, 256, False), (195000, 256, 512, False), (195000, 256, 256, False),
                       (466268, 256, 512, True), (148000, 256, 256, True)]:
    x = torch.randn(m, k, device=dev).bfloat16()
    ka = k + (32 if pos else 0)
    w = (torch.randn(n, ka, device=dev) * 0.05).bfloat16(); b = torch.randn(n, device=dev).bfloat16()
    cells = torch.randint(0, 64, (m,), device=dev, dtype=torch.uint8) if pos else None
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        if pos:
            check(lib.tmae_token_gemm_pos(x.data_ptr(), k, m, k, w.data_ptr(), n, b.data_ptr(), cells.data_ptr(), y.data_ptr(), n, st), 'tg')
        else:
            check(lib.tmae_token_gemm(x.data_ptr(), k, m, k, w.data_ptr(), n, b.data_ptr(), y.data_ptr(), n, st), 'tg')
    t = {}
    for rep in range(2):
        for mode in ('0', '1'):
            os.environ['TMAE_TG_WREG'] = mode
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            t[mode] = min(t.get(mode, 1e9), e0.elapsed_time(e1) / 20)
    byts = m * (k + n) * 2 + (n * ka + n) * 2
    out.append(f"{m}x{k}->{n}{'p' if pos else ''}: {t['0']*1e3:6.1f} -> {t['1']*1e3:6.1f} us ({t['1']/t['0']:.3f}, {byts/t['1']/1e6:5.0f} GB/s)")
print(os.environ.get('TMAE_LIB_PATH', 'default')[-14:], ' | '.join(out), flush=True)
