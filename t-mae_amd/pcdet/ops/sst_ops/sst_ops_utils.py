"""pcdet.ops.sst_ops.sst_ops_utils with the reference's two entry points (sst_ops_utils.py:5-27), on HIP."""
import torch

from tmae_amd import ops


def get_inner_win_inds(group_inds):
    """(N,) int64 group ids -> (N,) running index inside each group (deterministic: stable rank)."""
    return ops.get_inner_win_inds(group_inds)


def group_inner_inds(points, inverse_inds, K):
    """points (N,C), inverse_inds (N,) -> (M,K,C) first K points of every group, cyclic repeat when fewer."""
    m = int(inverse_inds.max().item()) + 1
    perm, offsets = ops.segment_csr(inverse_inds.contiguous().long(), m)
    cnt = (offsets[1:] - offsets[:-1]).long()
    cols = torch.arange(K, device=points.device)[None, :]
    src = torch.where(cols < cnt[:, None], cols, cols % cnt.clamp(min=1)[:, None])
    idx = perm.long()[(offsets[:-1].long()[:, None] + src).clamp(max=max(perm.shape[0] - 1, 0))]
    return points[idx]
