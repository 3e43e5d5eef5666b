"""Times tmae_dense_conv3x3_wgrad on the decoder conv's shape (8 x 468 x 468 cells, 384 -> 128): HIP events, 10 launches."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
from tmae_amd._lib import lib, check
dev = torch.device('cuda')
B, Y, X, cin, cout = 8, 468, 468, int(os.environ.get('CIN', 384)), 128
x = torch.randn(B, Y, X, cin, device=dev).bfloat16()
dy = torch.randn(B, Y, X, cout, device=dev).bfloat16()
dw = torch.empty(cout, 9 * cin, device=dev)
wsb = lib.tmae_dense_conv3x3_wgrad_workspace(cin, cout)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    check(lib.tmae_dense_conv3x3_wgrad(dy.data_ptr(), x.data_ptr(), B, Y, X, cin, cout, 1, dw.data_ptr(), ws.data_ptr(), wsb, st), 'wgrad')
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 2.0 * B * Y * X * 9 * cin * cout
print(f'dense conv weight gradient, cin={cin}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s')
