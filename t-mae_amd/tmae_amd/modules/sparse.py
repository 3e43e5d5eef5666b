"""Sparse tensor + 2-D sparse convolution modules with the call shapes of spconv.pytorch that the
reference uses (SparseConvTensor, SubMConv2d, SparseConv2d, SparseSequential; pcdet/utils/spconv_utils.py:28-56,
SiamWCA_MAE.py:187-193) -- implemented on the HIP rulebook / gather kernels, not on spconv.
"""
import torch
import torch.nn as nn

from .. import ops


class SparseConvTensor:
    """features [m,c]; indices [m,3] int32 (b,y,x), lexicographically ordered; spatial_shape (ny,nx)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, cache=None, groups=None):
        # groups: None, or ((rows, samples), ...) -- consecutive sample ranges that the reference runs as separate
        # calls (the two frames of the Siamese encoder); BatchNorm keeps separate statistics per group
        self.groups = groups
        self.features = features
        self.indices = indices.int().contiguous() if indices.dtype != torch.int32 else indices.contiguous()
        self.spatial_shape = [int(spatial_shape[0]), int(spatial_shape[1])]
        self.batch_size = int(batch_size)
        self._grid = grid
        self._cache = cache if cache is not None else {}      # rulebooks keyed by conv kind, shared by replace_feature
        self.frame_halves = None   # (features[:rows of group 0], features[rows of group 0:]) when the producing norm forked them

    @property
    def grid(self):
        """Dense row-index grid [batch*ny*nx] int32 (built once per index set)."""
        if self._grid is None:
            self._grid = ops.index_grid(self.indices, self.batch_size, *self.spatial_shape)
        return self._grid

    def replace_feature(self, new_features):
        return SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self._grid, self._cache,
                                self.groups)

    def dense_nhwc(self):
        ny, nx = self.spatial_shape
        return ops.sparse_to_dense(self.features, self.grid, self.indices, self.batch_size, ny, nx)

    def dense(self):
        """[batch, c, ny, nx] (channels-last memory), zeros at inactive sites (SiamWCA_MAE.py:235)."""
        return self.dense_nhwc().permute(0, 3, 1, 2)


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    """3x3 conv, weight in the spconv-2 layout [cout, kh, kw, cin] (detector3d_template.py:373-383)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None,
                 subm=False):
        super().__init__()
        if kernel_size != 3 or bias:
            raise NotImplementedError('the T-MAE path only uses 3x3 bias-free sparse convs')
        if not subm and not (stride == 2 and padding == 1):
            raise NotImplementedError('strided sparse conv: only k3 s2 p1 (spt_backbone.py:280-284)')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.subm, self.stride, self.indice_key = subm, stride, indice_key
        self.lookahead = False       # set by the backbone when another k3 s2 p1 conv follows on this conv's output sites
        self.weight = nn.Parameter(torch.empty(out_channels, 3, 3, in_channels))
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=5 ** 0.5)

    def forward(self, x: SparseConvTensor):
        ny, nx = x.spatial_shape
        if self.subm:
            rb = x._cache.get('subm')
            if rb is None:
                nbr = ops.spconv_neighbors(x.indices, x.grid, x.batch_size, ny, nx, 1)
                rb = (nbr, nbr.flip(1).contiguous())              # transposed rulebook of a subm conv = flipped taps
                x._cache['subm'] = rb
            y = ops.sparse_conv(x.features, self.weight, rb[0], rb[1])
            return x.replace_feature(y)
        rb = x._cache.get('down')
        nxt = None
        if rb is None:
            rb, nxt = _down_rulebooks(x, 2 if self.lookahead else 1)
            x._cache['down'] = rb
        nbr, nbr_t, out_ind, out_grid, oshape, groups = rb
        y = ops.sparse_conv(x.features, self.weight, nbr, nbr_t)
        # the next strided conv's rulebook (built with this one, one host sync for both) travels in the output's cache
        return SparseConvTensor(y, out_ind, oshape, x.batch_size, out_grid, groups=groups,
                                cache=None if nxt is None else {'down': nxt})


def _down_levels(x, depth):
    """enqueue the output-site kernels of `depth` chained SparseConv2d(k3, s2, p1) levels starting at x's sites; returns
    (levels, counts): what _down_rulebooks needs next and the device tensors whose values it has to read"""
    levels, counts = [], []
    grid, (ny, nx) = x.grid, x.spatial_shape
    for _ in range(depth):
        out_grid, out_ind, n_out, (oy, ox) = ops.spconv_down_outputs(grid, x.batch_size, ny, nx)
        counts.append(n_out.view(1).long())
        if x.groups is not None:
            cells = 0
            for _, nb in x.groups[:-1]:
                cells += nb * oy * ox
                counts.append((out_grid[:cells] >= 0).sum().view(1))
        levels.append((grid, ny, nx, out_grid, out_ind, oy, ox))
        grid, ny, nx = out_grid, oy, ox
    return levels, counts


def prefetch_down_rulebooks(x, depth):
    """Enqueue the output-site kernels and the copy of their counts NOW (the sites depend on x's sites only) and park the handle in
    x's rulebook cache: the strided conv that needs the rulebooks one SST block later then finds the counts on the host without
    stalling -- read at the point of use, the host waited ~10 ms for the GPU to reach kernels enqueued behind a whole stage, the GPU
    then idled until the host had woken up, and the host's lead over the GPU was gone for the rest of the forward pass."""
    levels, counts = _down_levels(x, depth)
    x._cache['down_prefetch'] = (depth, levels, ops.HostCopy(torch.cat(counts), 'down_counts'), x.grid)


def _down_rulebooks(x, depth):
    """Rulebooks of `depth` chained SparseConv2d(k3, s2, p1) levels starting at x's sites.  The output sites of a level
    depend on the previous level's sites only (never on features), so all levels are enqueued first and their counts
    -- output rows, and the rows of every sample group but the last -- come back in ONE host sync (or were prefetched:
    prefetch_down_rulebooks)."""
    pre = x._cache.pop('down_prefetch', None)
    if pre is not None and pre[0] == depth and pre[3] is x.grid:      # same levels, same sites (the grid object travels with them)
        levels, host = pre[1], pre[2].get().tolist()
    else:
        levels, counts = _down_levels(x, depth)
        host = ops.to_host(torch.cat(counts)).tolist()            # the one sync
    per = len(host) // depth
    rbs, in_ind = [], x.indices
    for d, (grid, ny, nx, out_grid, out_ind, oy, ox) in enumerate(levels):
        h = host[d * per:(d + 1) * per]
        m_out = int(h[0])
        groups = None
        if x.groups is not None:
            ends = [int(v) for v in h[1:]] + [m_out]
            groups = tuple((e - b, nb) for b, e, (_, nb) in zip([0] + ends[:-1], ends, x.groups))
        out_ind = out_ind[:m_out]
        nbr = ops.spconv_neighbors(out_ind, grid, x.batch_size, ny, nx, 2)
        nbr_t = ops.spconv_neighbors_t(in_ind, out_grid, x.batch_size, oy, ox, 2)
        rbs.append((nbr, nbr_t, out_ind, out_grid, (oy, ox), groups))
        in_ind = out_ind
    return rbs[0], (rbs[1] if depth > 1 else None)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, bias=False, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key, subm=True)


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias,
                         indice_key=indice_key)


class SparseSequential(SparseModule):
    """Children named '0','1',... like spconv.SparseSequential; dense modules act on .features."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(m, (nn.BatchNorm1d, nn.SyncBatchNorm)) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                rows = None if x.groups is None else [g[0] for g in x.groups]
                if rows is not None and len(rows) == 2 and i + 2 == len(mods):
                    # two frames in one token list and this norm ends the block: its output also leaves split by frame
                    # (SiamWCA_MAE.sparse_encode_pair) -- the halves come out of the norm's own autograd node (ops.batch_norm_relu)
                    y, halves = ops.batch_norm_relu(x.features, m, relu=True, groups=rows, fork=True)
                    x = x.replace_feature(y)
                    x.frame_halves = halves
                else:
                    x = x.replace_feature(ops.batch_norm_relu(x.features, m, relu=True, groups=rows))   # fused BN + ReLU
                i += 1
            elif isinstance(m, (nn.BatchNorm1d, nn.SyncBatchNorm)) and x.groups is not None:
                x = x.replace_feature(ops.batch_norm_relu(x.features, m, relu=False, groups=[g[0] for g in x.groups]))
            else:
                x = x.replace_feature(m(x.features))
            i += 1
        return x


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm',
                   norm_fn=None, dim=2):
    """conv + norm + ReLU (pcdet/utils/spconv_utils.py:37-56)."""
    assert dim == 2
    if conv_type == 'subm':
        conv = SubMConv2d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = SparseConv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                            indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


def replace_feature(out, new_features):
    return out.replace_feature(new_features)


def prefetch_down(blocks, x):
    """Before the first encoder block runs: if the first block keeps x's sites (no conv in front of its encoder, or a submanifold
    one) and the second one starts with the strided conv, enqueue that conv's (and, with `lookahead`, the next level's) output-site
    kernels and the copy of their counts now -- see prefetch_down_rulebooks.  Anything else: nothing (and _down_rulebooks checks
    that the sites it is asked about are the prefetched ones)."""
    if len(blocks) < 2 or not x.features.is_cuda:
        return
    first = getattr(blocks[0], 'conv_down', None)
    c0 = None if first is None else getattr(first, '0', None)
    if c0 is not None and not getattr(c0, 'subm', False):
        return
    nxt = getattr(blocks[1], 'conv_down', None)
    c1 = None if nxt is None else getattr(nxt, '0', None)
    if isinstance(c1, SparseConvolution) and not c1.subm:
        prefetch_down_rulebooks(x, 2 if c1.lookahead else 1)
