"""Attention forward + backward on a fixed random cloud, outputs saved to an .npz (for bit-comparing two builds of the library:
TMAE_LIB_PATH=<lib> python3 profiles/scripts/attn_dump.py out.npz)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd'))
from tmae_amd import ops
dev = 'cuda:0'
rng = np.random.default_rng(3)
torch.manual_seed(3)
res = {}
for tag, d, H, tauv, cross in (('a', 256, 8, 1.0, False), ('b', 128, 8, 0.3, False), ('c', 256, 8, 0.05, True)):
    def cloud(n):
        c = np.unique(np.stack([rng.integers(0, 2, n), rng.integers(0, 234, n), rng.integers(0, 234, n)], 1), axis=0)
        dense = np.stack(np.meshgrid(np.arange(40, 104), np.arange(40, 104), indexing='ij'), -1).reshape(-1, 2)
        dense = dense[rng.random(len(dense)) < 0.7]
        c = np.unique(np.concatenate([c, np.concatenate([np.zeros((len(dense), 1), np.int64), dense], 1)]), axis=0)
        return torch.from_numpy(c).int().to(dev)
    indq = cloud(15000)
    indk = cloud(30000) if cross else indq
    gq = ops.index_grid(indq, 2, 234, 234)
    gk = ops.index_grid(indk, 2, 234, 234) if cross else gq
    mq, mk = indq.shape[0], indk.shape[0]
    for shift in (False, True):
        a = torch.randn(mq, d if cross else 2 * d, device=dev).bfloat16().requires_grad_(True)
        b = torch.randn(mk, d, device=dev).bfloat16().requires_grad_(True)
        c = torch.randn(mk, d, device=dev).bfloat16().requires_grad_(True) if cross else None
        go = torch.randn(mq, d, device=dev).bfloat16()
        tau = torch.full((1, 1, 1), tauv, device=dev, requires_grad=True)
        wl = ops.window_worklist(gq, gk, 2, 234, 234, shift)
        o = ops.win_attn(a, b, c, tau, gq, gk, H, 2, 234, 234, shift, 0.01, worklist=wl)
        o.backward(go)
        k = f'{tag}{int(shift)}'
        res[k + '_o'] = o.detach().float().cpu().numpy()
        res[k + '_da'] = a.grad.float().cpu().numpy()
        res[k + '_db'] = b.grad.float().cpu().numpy()
        if c is not None:
            res[k + '_dc'] = c.grad.float().cpu().numpy()
        res[k + '_dtau'] = tau.grad.float().cpu().numpy()
np.savez(sys.argv[1], **res)
print('saved', sys.argv[1])
