"""Generate tests/golden/F14_options.npz from the reference itself (build container only): configuration options of the
hot path that the shipped YAMLs leave off and this repository supports anyway.

TEST INFRASTRUCTURE (see gen_golden.py).  Runs the unmodified reference module (via oracle/ref_import.py), checks this repo's CPU
oracle against every captured value and stores inputs + expected outputs as data.
  F14a  SSTInputLayer.get_pos_embed with NORMALIZE_POS: True (spt_backbone.py:186-224, :202-204), d = 128 / 256.
  F14b  CosineMultiheadAttention(non_shared_tau=True) (cosine_msa.py:453-454, :155-158): one temperature per head, self and cross
        attention with padding, temperatures on both sides of the clamp and of the kernels' split threshold; outputs and all gradients.
Usage:  python oracle/gen_golden_options.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R      # noqa: E402
import tmae_oracle as O     # noqa: E402
from gen_golden import check, save      # noqa: E402


if __name__ == '__main__':
    V, B, cfg = R.build_reference_model(num_stages=3, seed=0)
    il = B.sst_blocks[0].sst_input_layer
    ciw = np.stack([np.zeros(64, np.int64), np.repeat(np.arange(8), 8), np.tile(np.arange(8), 8)], 1)
    out = dict(coors_in_win=ciw, pos_temperature=np.float64(il.pos_temperature))
    assert il.normalize_pos is False
    il.normalize_pos = True                       # the attribute NORMALIZE_POS sets (spt_backbone.py:35)
    for d in (128, 256):
        f2w = {0: (torch.arange(64), (torch.arange(64),)), 'voxel_drop_level': torch.zeros(64, dtype=torch.long),
               'batching_info': {0: {'max_tokens': 64, 'drop_range': (0, 100000)}}}
        pe = il.get_pos_embed(f2w, torch.from_numpy(ciw), d)[0][0]
        check('pos (normalised)', O.pos_embed(ciw, d, (8, 8, 1), il.pos_temperature, normalize_pos=True), pe, 1e-6)
        out[f'pos_norm_{d}'] = pe
    il.normalize_pos = False

    ref = R.load_reference()
    CM = ref['cosine_msa'].CosineMultiheadAttention
    case = 0
    for (E, H, T, nW, cross) in [(128, 8, 16, 5, False), (256, 8, 64, 3, True), (128, 8, 32, 4, False)]:
        torch.manual_seed(300 + case)
        mha = CM(E, H, dropout=0.0, tau_min=0.01, cosine=True, non_shared_tau=True)
        assert tuple(mha.tau.shape) == (1, H, 1, 1)
        taus = torch.tensor([0.005, 0.05, 0.2, 0.3, 0.5, 1.0, 2.0, 0.1])[torch.randperm(H)]
        with torch.no_grad():
            mha.tau.copy_(taus.view(1, H, 1, 1))
            mha.in_proj_bias.normal_(0, 0.02)
            mha.out_proj.bias.normal_(0, 0.02)
        lens = torch.randint(1, T + 1, (nW,))
        lens[0] = T
        kpm = torch.arange(T)[None, :] >= lens[:, None]
        q = torch.randn(T, nW, E, requires_grad=True)
        k = torch.randn(T, nW, E, requires_grad=True) if cross else None
        v = torch.randn(T, nW, E, requires_grad=True)
        qlens = torch.randint(1, T + 1, (nW,)) if cross else lens
        qvalid = (torch.arange(T)[None, :] < qlens[:, None]).t().unsqueeze(-1).float()   # [T,nW,1]
        o, _ = mha(q, k if cross else q, value=v, key_padding_mask=kpm)
        gout = torch.randn_like(o)
        (o * gout * qvalid).sum().backward()
        p = {'a.' + n: t.detach() for n, t in mha.state_dict().items()}
        qo = q.detach().transpose(0, 1).clone().requires_grad_(True)
        ko = k.detach().transpose(0, 1).clone().requires_grad_(True) if cross else None
        vo = v.detach().transpose(0, 1).clone().requires_grad_(True)
        po = {n: t.clone().requires_grad_(True) for n, t in p.items()}
        oo = O.cosine_mha(qo, ko if cross else qo, vo, kpm, po, 'a.', H, 0.01)
        (oo * (gout * qvalid).transpose(0, 1)).sum().backward()
        check('attn out', oo.transpose(0, 1) * qvalid, o * qvalid, 1e-4)
        check('attn dq', qo.grad.transpose(0, 1), q.grad, 2e-4)
        check('attn dv', vo.grad.transpose(0, 1), v.grad, 2e-4)
        check('attn dtau', po['a.tau'].grad, mha.tau.grad, 1e-3 * max(1.0, float(mha.tau.grad.abs().max())))
        check('attn dW', po['a.in_proj_weight'].grad, mha.in_proj_weight.grad, 1e-3)
        pre = f'h{case}_'
        out.update({pre + 'meta': np.array([E, H, T, nW, int(cross)]), pre + 'q': q.detach(), pre + 'v': v.detach(), pre + 'kpm': kpm,
                    pre + 'qlens': qlens, pre + 'gout': gout, pre + 'out': o.detach() * qvalid, pre + 'dq': q.grad, pre + 'dv': v.grad,
                    pre + 'dtau': mha.tau.grad, pre + 'd_in_proj_weight': mha.in_proj_weight.grad,
                    pre + 'd_in_proj_bias': mha.in_proj_bias.grad, pre + 'd_out_proj_weight': mha.out_proj.weight.grad})
        if cross:
            out.update({pre + 'k': k.detach(), pre + 'dk': k.grad})
            check('attn dk', ko.grad.transpose(0, 1), k.grad, 2e-4)
        for n, t in mha.state_dict().items():
            out[pre + 'w_' + n.replace('.', '__')] = t
        case += 1
    save('F14_options', **out)
