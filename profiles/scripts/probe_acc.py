import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 't-mae_amd'))
from tmae_amd._lib import lib, check
sink = torch.zeros(4, device='cuda'); flops = C.c_int64(0); st = torch.cuda.current_stream().cuda_stream
def run(): check(lib.tmae_probe_mfma(4096, sink.data_ptr(), C.addressof(flops), st), 'p')
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print('TMAE_PROBE_ACC', os.environ.get('TMAE_PROBE_ACC', '16'), f'{flops.value / ms / 1e9:.0f} TFLOP/s')
