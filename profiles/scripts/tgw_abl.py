import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
import torch
from tmae_amd._lib import lib, check
dev = torch.device('cuda:0')
m, k = 466268, 256
x = torch.randn(m, k, device=dev).bfloat16()
for n in (512, 256):
    w = (torch.randn(n, k, device=dev) * 0.05).bfloat16(); b = torch.randn(n, device=dev).bfloat16()
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        check(lib.tmae_token_gemm(x.data_ptr(), k, m, k, w.data_ptr(), n, b.data_ptr(), y.data_ptr(), n, st), 'tg')
    for wreg, flags in (('0', 0), ('1', 0), ('1', 1), ('1', 2), ('1', 4), ('1', 3), ('1', 5), ('1', 6), ('1', 7)):
        os.environ['TMAE_TG_WREG'] = wreg; os.environ['TMAE_TGW_FLAGS'] = str(flags)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print(f'n={n} wreg={wreg} flags={flags} (1=no stores 2=no mfma 4=no loads): {e0.elapsed_time(e1)/20*1e3:7.1f} us', flush=True)
