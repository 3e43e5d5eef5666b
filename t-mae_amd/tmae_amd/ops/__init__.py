"""Python operator layer over the C ABI (include/tmae_hip.h): thin wrappers + autograd Functions.

Mirrors the operator surface the reference reaches through sst_ops_utils / torch_scatter / spconv /
pytorch3d -- same argument meaning, but every call lands in a hand-written gfx950 kernel.
No tensor math happens here besides allocation and dense GEMMs handed to hipBLASLt via torch.
"""
import torch

from .. import _lib
from .._lib import lib, check, PinnedStager


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _s():
    """HIP stream of torch's current stream on the current device (the raw-handle query is ~20x cheaper than building
    a torch.cuda.Stream object; ~250 calls per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _dt(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise TypeError(f'tmae_amd ops support float32 / bfloat16 features, got {t.dtype}')


def _ws(nbytes, device):
    return torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError('tmae_amd ops run on the GPU only (no CPU fallback)')


def compute_dtype(t):
    """dtype a feature op should run in: the autocast dtype when autocast is on, else the input's."""
    if torch.is_autocast_enabled('cuda'):
        return torch.get_autocast_dtype('cuda')
    return t.dtype if t.dtype in (torch.float32, torch.bfloat16) else torch.float32


# ----------------------------------------------------------------------------- token-list GEMM

_TOKEN_GEMM_MIN_ROWS = 8192


def _tg_ok(x, k, n):
    # contraction 128 / 256: ahead of the library on every shape (profiles/scripts/gemm_probe.py).  Contraction 512:
    # the W-in-registers kernel (csrc/token_gemm_wreg.hip: n = 256, >= 64 k tokens) is ahead of hipBLASLt; the chunk-streaming
    # kernel (16 tokens per wave: LDS-bound) is 10-15 % behind it on smaller token lists, which stay with the library.
    return (x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[0] >= _TOKEN_GEMM_MIN_ROWS
            and (k in (64, 128, 256) or (k == 512 and n == 256 and x.shape[0] >= 65536))
            and n % 64 == 0 and x.shape[0] * n * 2 < 2 ** 31
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)


_ZERO_BIAS = {}


def _zero_bias(n, device):
    """A read-only bf16 zero vector for the bias-free GEMM calls (the kernels are branch-free and always add a bias):
    cached per (n, device) instead of one fill launch per call (~45 per step)."""
    key = (int(n), device)
    z = _ZERO_BIAS.get(key)
    if z is None:
        z = _ZERO_BIAS[key] = torch.zeros((int(n),), dtype=torch.bfloat16, device=device)
    return z


def token_gemm(x, w, bias=None, force=False):
    """y [m,n] = x [m,k] @ w[n,k]^T (+ bias) in bf16, fp32 accumulation: the x-stationary streaming kernel of
    csrc/token_gemm.hip on the shapes where it is ahead of the library (force=True: whenever the kernel supports the
    shape: k in {64,128,256,512}, n % 64 == 0), else the library."""
    n, k = w.shape
    ok = _tg_ok(x, k, n) or (force and x.dtype == torch.bfloat16 and k in (64, 128, 256, 512) and n % 64 == 0
                             and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)
    if ok and w.dtype == torch.bfloat16 and w.is_contiguous() and (bias is None or bias.dtype == torch.bfloat16):
        m = x.shape[0]
        y = torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
        b = _zero_bias(n, x.device) if bias is None else bias.contiguous()
        check(lib.tmae_token_gemm(_p(x), x.stride(0), m, k, _p(w), n, _p(b), _p(y), n, _s()), 'tmae_token_gemm')
        return y
    return torch.nn.functional.linear(x, w, bias)


def token_gemm_gelu(x, w, bias):
    """(y, gelu(y)) with y = x @ w^T + bias: ONE launch on the shapes of the encoder FFN's first Linear
    (tmae_token_gemm_gelu: d -> 2d for d in {128, 256}, >= 32 k tokens, bf16), otherwise the GEMM and an elementwise pass."""
    n, k = w.shape
    m = x.shape[0]
    if (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and (k, n) in ((256, 512), (128, 256)) and m >= 32768
            and bias is not None and bias.dtype == torch.bfloat16 and w.is_contiguous() and x.stride(1) == 1
            and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and m * n * 2 < 2 ** 31):
        y = torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
        yg = torch.empty_like(y)
        check(lib.tmae_token_gemm_gelu(_p(x), x.stride(0), m, k, _p(w), n, _p(bias.contiguous()), _p(y), _p(yg), n, _s()),
              'tmae_token_gemm_gelu')
        return y, yg
    y = token_gemm(x, w, bias)
    return y, torch.nn.functional.gelu(y)


def addmm_inplace(dx, dy, w, wt=None):
    """dx += dy @ w in place (w [n,k]: dx [m,k], dy [m,n]): the W-in-registers kernel's accumulate form on the shapes it
    covers (tmae_token_gemm_acc: contraction 512 -> 256, 256 -> 128, the attention in-projections' 768 -> 256 and 384 -> 128,
    the square 256 -> 256 and 128 -> 128; >= 32 k tokens), torch's addmm_ otherwise.  wt: w^T contiguous if the caller keeps one (else made here)."""
    n, k = w.shape
    m = dx.shape[0]
    if (m >= (65536 if n in (512, 768) else 32768) and (n, k) in ((512, 256), (256, 128), (768, 256), (384, 128), (256, 256), (128, 128))
            and dx.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
            and w.dtype == torch.bfloat16 and dx.is_contiguous() and dy.stride(1) == 1 and dy.stride(0) % 8 == 0
            and dy.data_ptr() % 16 == 0 and dx.data_ptr() % 16 == 0 and m * max(n, k) * 2 < 2 ** 31):
        if wt is None:
            wt = _transposed(w)
        check(lib.tmae_token_gemm_acc(_p(dy), dy.stride(0), m, n, _p(wt), k, _p(_zero_bias(k, dx.device)), _p(dx), k, _s()),
              'tmae_token_gemm_acc')
        return dx
    return dx.addmm_(dy, w)


def token_gemm_dx(dy, w, force=False):
    """dx [m,k] = dy [m,n] @ w[n,k]: the same kernel on w^T (contraction n in {128,256}, k % 64 == 0)."""
    n, k = w.shape
    if (force or _tg_ok(dy, n, k)) and w.dtype == torch.bfloat16:
        return token_gemm(dy, _transposed(w), None, force)
    return dy @ w


def _transposed(w):
    """w^T contiguous.  For a bf16 copy managed by refresh_param_copies the transposed copy is made there, together with
    the cast, in ONE launch for all weights that asked for it (the first request of a weight registers it; until the next
    refresh it is transposed on the spot); other whole weights are cached until their version counter moves."""
    st = getattr(w, '_tmae_stamp', None)
    if st is not None and w.dim() == 2:
        T = getattr(w, '_tmae_T', None)
        if T is not None and T[0] == st:
            return T[1]
        with torch.no_grad():
            t = w.t().contiguous()
        w._tmae_T = (st, t)                               # refresh_param_copies rewrites this buffer from now on
        return t
    if w._base is None and not w.requires_grad:
        return _derived(w, 'T', lambda t: t.t().contiguous())
    return w.t().contiguous()


# ----------------------------------------------------------------------------- BatchNorm running statistics

_BN_PENDING = None          # None: update immediately; a list: inside defer_bn_updates()


def _bn_running_update(bn, mean, var, count):
    """running = (1 - mom) * running + mom * batch statistic (unbiased variance), num_batches_tracked += 1 -- torch's
    BatchNorm update.  Inside `defer_bn_updates()` the update is queued and applied at the exit in multi-tensor
    launches (the step has ~26 such updates x 5 tiny launches each)."""
    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
    if _BN_PENDING is not None and bn.momentum is not None:
        _BN_PENDING.append((bn, mean, var, float(mom), float(count)))
        return
    with torch.no_grad():
        bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
        bn.running_var.mul_(1 - mom).add_(var, alpha=mom * count / (count - 1))
        bn.num_batches_tracked += 1


_BN_TABLES = {}


def _bn_flush(pending):
    """Apply the queued running-statistics updates: ONE launch of tmae_bn_running_update for fp32 CUDA buffers (a
    workgroup per buffer applies that buffer's updates in order), torch ops otherwise."""
    import struct
    with torch.no_grad():
        dev = pending[0][0].running_mean.device
        def _raw_ok(bn, mean, var):          # everything the kernel reads / writes through raw pointers
            rm, rv, nbt = bn.running_mean, bn.running_var, bn.num_batches_tracked
            n = rm.numel()
            return (rm.is_cuda and rm.device == dev and rm.dtype == torch.float32 and rm.is_contiguous()
                    and rv.device == dev and rv.dtype == torch.float32 and rv.is_contiguous() and rv.numel() == n
                    and nbt.device == dev and nbt.dtype == torch.int64            # 64-bit atomic add in the kernel
                    and mean.device == dev and var.device == dev and mean.dtype == torch.float32
                    and var.dtype == torch.float32 and mean.is_contiguous() and var.is_contiguous()
                    and mean.numel() == n and var.numel() == n)
        ok = all(_raw_ok(bn, mean, var) for bn, mean, var, _, _ in pending)
        if not ok:
            for bn, mean, var, mom, count in pending:
                bn.running_mean.mul_(1 - mom).add_(mean.reshape(-1), alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var.reshape(-1), alpha=mom * count / (count - 1))
                bn.num_batches_tracked += 1
            return
        per_buf, order, counters = {}, [], []
        for bn, mean, var, mom, count in pending:
            for buf, stat, scale in ((bn.running_mean, mean, mom), (bn.running_var, var, mom * count / (count - 1.0))):
                key = buf.data_ptr()
                if key not in per_buf:
                    per_buf[key] = (buf, [])
                    order.append(key)
                bits = struct.unpack('<q', struct.pack('<ff', 1.0 - mom, scale))[0]
                per_buf[key][1].append((stat.data_ptr(), bits))
            counters.append(bn.num_batches_tracked.data_ptr())
        bufs, upds = [], []
        for key in order:
            buf, us = per_buf[key]
            bufs.append([buf.data_ptr(), buf.numel(), len(upds), len(us)])
            upds += [list(u) for u in us]
        flat = [v for row in bufs for v in row] + [v for row in upds for v in row] + counters
        nb, nu = len(bufs), len(upds)
        # sent from pinned memory with an asynchronous copy (a pageable torch.tensor(..., device=) copy is synchronous: the
        # host would wait here for everything queued before); two staging buffers in turn
        ent = _BN_TABLES.get((dev, len(flat)))
        if ent is None:
            ent = _BN_TABLES[(dev, len(flat))] = (PinnedStager((len(flat),)), torch.empty(len(flat), dtype=torch.int64, device=dev))
        table = ent[0].upload(flat, ent[1])      # event-guarded: never rewrites a buffer whose copy is still queued
        base = table.data_ptr()
        check(lib.tmae_bn_running_update(base, nb, base + nb * 32, base + nb * 32 + nu * 16, len(counters), _s()),
              'tmae_bn_running_update')
        # keep the statistics tensors alive until the launch has read them (stream-ordered allocator reuse is safe: the
        # next kernel that could reuse their memory is enqueued after this one on the same stream)


class defer_bn_updates:
    """Context manager: BatchNorm running-statistics updates made inside are applied at the exit, in order, in one
    launch (tmae_bn_running_update; a module used twice gets its two updates applied sequentially)."""

    def __enter__(self):
        global _BN_PENDING
        self.prev = _BN_PENDING
        if _BN_PENDING is None:
            _BN_PENDING = []
        return self

    def __exit__(self, *exc):
        global _BN_PENDING
        if self.prev is not None:            # nested: the outermost context flushes
            return False
        pending, _BN_PENDING = _BN_PENDING, None
        if pending:
            _bn_flush(pending)
        return False


# ----------------------------------------------------------------------------- low-precision parameter copies

def cast_param(p, dtype):
    """p.to(dtype), served from the copy made by refresh_param_copies when that copy is still current (same storage,
    same version counter): ~340 per-parameter cast launches per step become a handful of multi-tensor launches."""
    if p.dtype == dtype:
        return p
    c = getattr(p, '_tmae_copy', None)
    if c is not None and c[0] == p._version and c[1] == p.data_ptr() and c[2].dtype == dtype:
        return c[2]
    return p.detach().to(dtype)


@torch.no_grad()
def refresh_param_copies(params, dtype=torch.bfloat16):
    """Re-cast every floating-point parameter to `dtype` with multi-tensor copies (call after optimizer.step())."""
    ps = []
    for p in params:
        if not (p.is_cuda and p.is_floating_point() and p.dtype != dtype):
            continue
        c = getattr(p, '_tmae_copy', None)      # still current (a parameter the optimizer does not own): keep it
        if c is None or c[0] != p._version or c[1] != p.data_ptr() or c[2].dtype != dtype:
            ps.append(p)
    if not ps:
        return
    dst, plain, mats = [], [], []
    for p in ps:
        c = getattr(p, '_tmae_copy', None)
        if c is None or c[2].dtype != dtype or c[2].shape != p.shape:
            c = [0, 0, torch.empty_like(p, dtype=dtype)]
            p._tmae_copy = c
        # weights whose transposed copy is in use (ops._transposed): cast + transpose in the one-launch kernel
        if (dtype == torch.bfloat16 and p.dtype == torch.float32 and p.dim() == 2 and p.is_contiguous()
                and getattr(c[2], '_tmae_T', None) is not None):
            mats.append(p)
        else:
            plain.append(p)
            dst.append(c[2])
    if plain:
        torch._foreach_copy_(dst, plain)
    if mats:
        _multi_cast_transpose(mats)
    for p in ps:
        c = p._tmae_copy
        c[0], c[1] = p._version, p.data_ptr()
        c[2]._tmae_stamp = getattr(c[2], '_tmae_stamp', 0) + 1
        T = getattr(c[2], '_tmae_T', None)
        if T is not None:                                 # rewritten above (mats) or stale (plain: dropped)
            c[2]._tmae_T = (c[2]._tmae_stamp, T[1]) if id(p) in _MCT_DONE else None
    _MCT_DONE.clear()


def plain_copy_targets(params, dtype=torch.bfloat16):
    """For the optimizer's one-launch step: per parameter the `dtype` copy it should write next to the update (created
    here if missing), or None -- wrong dtype / device, or a matrix whose TRANSPOSED copy is in use (those are re-cast
    and re-transposed together by refresh_param_copies' one-launch kernel)."""
    out = []
    for p in params:
        if not (p.is_cuda and p.dtype == torch.float32 and dtype == torch.bfloat16 and p.is_contiguous()):
            out.append(None)
            continue
        c = getattr(p, '_tmae_copy', None)
        if c is None or c[2].dtype != dtype or c[2].shape != p.shape:
            c = [0, 0, torch.empty_like(p, dtype=dtype)]
            p._tmae_copy = c
        out.append(None if (p.dim() == 2 and getattr(c[2], '_tmae_T', None) is not None) else c[2])
    return out


def mark_copies_written(params):
    """The copies of `params` (plain_copy_targets) hold the parameters' current values: the bookkeeping of
    refresh_param_copies for them."""
    for p in params:
        c = p._tmae_copy
        c[0], c[1] = p._version, p.data_ptr()
        c[2]._tmae_stamp = getattr(c[2], '_tmae_stamp', 0) + 1
        if getattr(c[2], '_tmae_T', None) is not None:      # (not reached: such parameters are not handed out)
            c[2]._tmae_T = None


_MCT_TABLES = {}
_MCT_DONE = set()


def _multi_cast_transpose(mats):
    """fp32 matrices -> their bf16 copies and transposed bf16 copies (tmae_multi_cast_transpose); the descriptor table is
    built once per set of buffers."""
    # shapes are part of the key: the table bakes n | k << 32 and the tile offsets, and the caching allocator hands the
    # same blocks to a rebuilt model whose weights have the same byte sizes but other shapes
    key = tuple((p.data_ptr(), p._tmae_copy[2].data_ptr(), p._tmae_copy[2]._tmae_T[1].data_ptr(), tuple(p.shape))
                for p in mats)
    ent = _MCT_TABLES.get(key)
    if ent is None:
        rows, tile0 = [], 0
        for p in mats:
            n, k = p.shape
            rows.append([p.data_ptr(), p._tmae_copy[2].data_ptr(), p._tmae_copy[2]._tmae_T[1].data_ptr(), n | (k << 32), tile0])
            tile0 += ((n + 31) // 32) * ((k + 31) // 32)
        ent = _MCT_TABLES[key] = (torch.tensor(rows, dtype=torch.int64, device=mats[0].device), tile0)
        if len(_MCT_TABLES) > 8:
            _MCT_TABLES.pop(next(iter(_MCT_TABLES)))
    check(lib.tmae_multi_cast_transpose(_p(ent[0]), len(mats), ent[1], _s()), 'tmae_multi_cast_transpose')
    _MCT_DONE.update(id(p) for p in mats)


# ----------------------------------------------------------------------------- token-list Linear

def linear_wgrad(dy, x, want_bias=True, out_w=None, out_b=None, cells=None, pos_n=0, pos_e=None):
    """dW [n,k] f32 = dy^T @ x, db [n] f32 = column sums of dy; dy [m,n], x [m,k] bf16 (row-major, last dim
    contiguous).  One streaming pass, token axis split over the chip (csrc/wgrad.hip).  out_w / out_b: contiguous f32
    destinations (e.g. a row slice of a packed gradient) written in place of fresh tensors.
    cells (ops.window_cells): also returns dcell [16, n] f32, the per-cell column sums of dy[:, :pos_n]
    (tmae_linear_wgrad_cells) -- the return value is then (dw, db, dcell).  pos_e (E [16, k] f32, ops.pos_axes): the position
    part dcell^T E of the in-projection's weight gradient is added to dw inside the slab reduction; dcell is then None."""
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.stride(1) == 1 and x.stride(1) == 1
    m, n = dy.shape
    k = x.shape[1]
    if out_w is not None:
        assert out_w.shape == (n, k) and out_w.dtype == torch.float32 and out_w.is_contiguous()
        dw = out_w
    else:
        dw = torch.empty((n, k), dtype=torch.float32, device=dy.device)
    if want_bias and out_b is not None:
        assert out_b.shape == (n,) and out_b.dtype == torch.float32 and out_b.is_contiguous()
        db = out_b
    else:
        db = torch.empty((n,), dtype=torch.float32, device=dy.device) if want_bias else None
    wsb = lib.tmae_linear_wgrad_workspace(m, n, k)
    ws = _ws(wsb, dy.device)
    if cells is not None:
        # (the kernels zero the cell sums per OUTPUT TILE, the reduction adds dcell^T E to every row: the position rows must end on a
        #  tile boundary -- 256 rows for the 256-tile kernel, 128 otherwise -- or cover all rows; else the unfolded form below)
        tile = 256 if (n >= 256 and k >= 256 and ((n + 255) // 256) * ((k + 255) // 256) >= 2) else 128
        fold = (pos_e is not None and pos_e.dtype == torch.float32 and pos_e.is_contiguous() and pos_e.shape == (16, k)
                and k >= 64 and (256 % k == 0 or k % 256 == 0) and (n * k) % 256 == 0 and (pos_n % tile == 0 or pos_n == n))
        dcell = None if fold else torch.empty((16, n), dtype=torch.float32, device=dy.device)
        check(lib.tmae_linear_wgrad_cells(_p(dy), dy.stride(0), _p(x), x.stride(0), m, n, k, _p(cells), int(pos_n),
                                          _p(pos_e) if fold else None, _p(dw), _p(db), _p(dcell), _p(ws), wsb, _s()),
              'tmae_linear_wgrad_cells')
        if pos_e is not None and not fold and pos_n > 0:
            dw[:pos_n].addmm_(dcell[:, :pos_n].t(), pos_e)
            dcell = None
        return dw, db, dcell
    check(lib.tmae_linear_wgrad(_p(dy), dy.stride(0), _p(x), x.stride(0), m, n, k, _p(dw), _p(db), _p(ws), wsb, _s()),
          'tmae_linear_wgrad')
    return dw, db


def _wgrad_ok(dy, x):
    return (dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0
            and dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
            and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and dy.shape[0] >= 4096)


class _Linear(torch.autograd.Function):
    """y = x W^T + b over a long token list.  Forward and dX are plain library GEMMs (they stream at HBM rate);
    the weight / bias gradient -- a reduction over 1e5+ tokens into a few tiles -- is our split-token MFMA kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        cdt = compute_dtype(x)
        x_c = x.to(cdt)
        w_c = cast_param(weight, cdt)
        y = token_gemm(x_c, w_c, None if bias is None else cast_param(bias, cdt))
        ctx.save_for_backward(x_c, w_c)
        ctx.has_bias = bias is not None
        ctx.dtypes = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_c, w_c = ctx.saved_tensors
        xdt, wdt, bdt = ctx.dtypes
        dy = dy.to(x_c.dtype)
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        dx = token_gemm_dx(dy, w_c).to(xdt) if ctx.needs_input_grad[0] else None
        if _wgrad_ok(dy, x_c):
            dw, db = linear_wgrad(dy, x_c, ctx.has_bias)
        else:
            dw = dy.t() @ x_c
            db = dy.sum(0) if ctx.has_bias else None
        return dx, dw.to(wdt), (db.to(bdt) if ctx.has_bias else None)


class _ProjFork(torch.autograd.Function):
    """Several Linear projections of ONE token list in one autograd node:
        out_s = (x [+ pos]) W[r0:r1]^T + b[r0:r1]      for every segment s = (r0, r1, use_pos)
    plus, with `fork`, x itself as a last output for the residual branch of the layer.  The backward folds every
    contribution to dx -- also the gradient that arrives through the forked alias -- into accumulating GEMMs
    (addmm, beta = 1), so autograd never runs a separate add over [m, d]; dW / db come back as one full-size tensor
    (no slice-backward zero fills).  pos = (indices, table, wy, wx, do_shift) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, segs, pos, fork, inplace_dx, gelu=False):
        cdt = compute_dtype(x)
        x_c = x.to(cdt).contiguous()
        w_c = cast_param(weight, cdt)
        b_c = None if bias is None else cast_param(bias, cdt)
        xp = None
        if any(sg[2] for sg in segs):
            indices, table, wy, wx, do_shift = pos
            xp = torch.empty_like(x_c)
            check(lib.tmae_add_pos_embed(_p(x_c), _dt(x_c), x_c.shape[0], x_c.shape[1], _p(indices), wy, wx,
                                         1 if do_shift else 0, _p(table), _p(xp), _s()), 'tmae_add_pos_embed')
        outs = []
        act = None
        if gelu:             # one segment, the whole weight: the projection and its GELU (not differentiable here: the
            r0, r1, use_pos = segs[0]        # consumer, ops.gelu_linear, differentiates through the pre-activation)
            assert len(segs) == 1 and not use_pos and r0 == 0 and r1 == w_c.shape[0]
            y, act = token_gemm_gelu(x_c, w_c, b_c)
            outs.append(y)
        else:
            for r0, r1, use_pos in segs:
                outs.append(token_gemm(xp if use_pos else x_c, w_c[r0:r1], None if b_c is None else b_c[r0:r1]))
        ctx.save_for_backward(x_c, xp, w_c)
        ctx.segs, ctx.fork, ctx.has_bias = segs, fork, bias is not None
        ctx.inplace_dx = bool(inplace_dx and fork)
        ctx.dtypes = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        ctx.set_materialize_grads(False)
        if act is not None:
            ctx.mark_non_differentiable(act)
            outs.append(act)
        if fork:
            outs.append(x.view_as(x))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        x_c, xp, w_c = ctx.saved_tensors
        xdt, wdt, bdt = ctx.dtypes
        segs = ctx.segs
        dx = grads[-1] if ctx.fork else None
        if dx is not None:
            dx = dx.to(x_c.dtype)
        rows = w_c.shape[0]
        covered = sum(r1 - r0 for (r0, r1, _), g in zip(segs, grads) if g is not None) == rows
        mk = torch.empty if covered else torch.zeros
        dW = mk((rows, w_c.shape[1]), dtype=torch.float32, device=x_c.device) if ctx.needs_input_grad[1] else None
        dB = mk((rows,), dtype=torch.float32, device=x_c.device) if ctx.has_bias and ctx.needs_input_grad[2] else None
        for (r0, r1, use_pos), dy in zip(segs, grads):
            if dy is None:
                continue
            dy = dy.to(x_c.dtype)
            if dy.stride(-1) != 1:
                dy = dy.contiguous()
            wseg = w_c if (r0 == 0 and r1 == rows) else w_c[r0:r1]     # the whole weight keeps its cached transpose
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx = token_gemm_dx(dy, wseg)
                elif ctx.inplace_dx:
                    # accumulate IN PLACE into the gradient that arrived through the alias (torch.addmm would first
                    # memcpy it into a new buffer).  Only on the caller's word (`inplace_dx`): the encoder layers
                    # guarantee that this buffer -- the dx of their add+LayerNorm backward -- has no reader left
                    # (its only other reader, the Linear on the norm's second input, is a younger node and has run).
                    addmm_inplace(dx, dy, wseg)
                else:
                    dx = torch.addmm(dx, dy, wseg)
            if dW is not None:
                inp = xp if use_pos else x_c
                if _wgrad_ok(dy, inp):          # straight into the rows of the packed gradient: no slice copies
                    linear_wgrad(dy, inp, dB is not None, out_w=dW[r0:r1], out_b=None if dB is None else dB[r0:r1])
                else:
                    dW[r0:r1] = dy.float().t() @ inp.float()
                    if dB is not None:
                        dB[r0:r1] = dy.float().sum(0)
        return (None if dx is None else dx.to(xdt), None if dW is None else dW.to(wdt),
                None if dB is None else dB.to(bdt), None, None, None, None, None)


def proj_fork(x, weight, bias, segs, pos=None, fork=False, inplace_dx=False, gelu=False):
    """See _ProjFork.  Returns the projections (and x's alias last when fork) as a tuple.
    gelu (one segment covering the whole weight): the tuple is (y, gelu(y), [alias]) -- the activation comes out of the
    same launch (token_gemm_gelu) and carries no gradient; hand it to ops.gelu_linear(y, ..., h=gelu(y)).
    inplace_dx: the backward may add into the gradient tensor it receives for the alias instead of copying it.
    Pass True only if the alias feeds exactly one consumer whose backward hands over a buffer nobody else reads
    afterwards (no hooks / retain_grad on the alias, no second use, no retain_graph replay)."""
    if not x.is_cuda or x.dim() != 2:
        raise RuntimeError('proj_fork: 2-D GPU token lists only')
    segs = tuple((int(a), int(b), bool(c)) for a, b, c in segs)
    if pos is not None:
        indices, table, window_shape, do_shift = pos
        pos = (indices, table, int(window_shape[1]), int(window_shape[0]), bool(do_shift))
    return _ProjFork.apply(x, weight, bias, segs, pos, bool(fork), bool(inplace_dx), bool(gelu))


def _derived(t, name, fn):
    """fn(t), cached on the tensor object for as long as t keeps its storage and version counter (weights that the
    optimizer does not own -- every attention in-projection under the reference's parameter grouping -- never change)."""
    c = getattr(t, '_tmae_derived', None)
    if c is None:
        c = t._tmae_derived = {}
    e = c.get(name)
    if e is not None and e[0] == t._version and e[1] == t.data_ptr():
        return e[2]
    with torch.no_grad():
        v = fn(t)
    c[name] = (t._version, t.data_ptr(), v)
    return v


def window_cells(indices, window_shape, do_shift, want_onehot=False):
    """cells [m] u8 = xc | yc << 3 of the tokens inside their (shifted) windows -- a view of a zero-padded buffer of
    round_up(m, 32) + 64 bytes, which tmae_linear_wgrad_cells may read past m -- and, on request, the explicit one-hot
    matrix [m,16] bf16 (columns 0..7 xc, 8..15 yc)."""
    m = indices.shape[0]
    buf = torch.empty(((m + 31) // 32 * 32 + 64,), dtype=torch.uint8, device=indices.device)    # the kernel zeroes the tail
    cells = buf[:m]
    onehot = torch.empty((m, 16), dtype=torch.bfloat16, device=indices.device) if want_onehot else None
    check(lib.tmae_window_cells(_p(indices), m, buf.shape[0], int(window_shape[1]), int(window_shape[0]),
                                1 if do_shift else 0, _p(buf), _p(onehot), _s()), 'tmae_window_cells')
    return (cells, onehot) if want_onehot else cells


def pos_axes(pos_table, window_shape):
    """E [16, d] f32, block diagonal: rows 0..7 = ex[xc] in columns :d/2, rows 8..15 = ey[yc] in columns d/2: -- the
    separable halves of the in-window position embedding (pos[y * wx + x] = [ex[x] | ey[y]], spt_backbone.py:186-224)."""
    def make(t):
        wx, wy = int(window_shape[0]), int(window_shape[1])
        d = t.shape[1]
        h = d // 2
        tab = t.float().view(wy, wx, d)
        if not (torch.equal(tab[:, :, :h], tab[:1, :, :h].expand(wy, wx, h))
                and torch.equal(tab[:, :, h:], tab[:, :1, h:].expand(wy, wx, d - h))):
            raise RuntimeError('position table is not separable into x and y halves')
        e = torch.zeros((16, d), dtype=torch.float32, device=t.device)
        e[:wx, :h] = tab[0, :, :h]
        e[8:8 + wy, h:] = tab[:, 0, h:]
        return e
    return _derived(pos_table, ('axes', int(window_shape[0]), int(window_shape[1])), make)


def pos_fold_weight(weight, lo, hi, pos_rows, E):
    """w_aug [hi-lo, d+32] bf16 = [W | Tx_hi Ty_hi Tx_lo Ty_lo] for rows lo:hi of `weight` (tmae_token_gemm_pos); T = W E^T
    for the rows inside pos_rows = (p0, p1) (absolute row range that takes the position), zero elsewhere."""
    def make(w):
        ws = w[lo:hi].float()
        t = torch.zeros((hi - lo, 16), dtype=torch.float32, device=w.device)
        p0, p1 = pos_rows
        if p1 > p0:
            t[p0 - lo:p1 - lo] = ws[p0 - lo:p1 - lo] @ E.t()
        t_hi = t.to(torch.bfloat16)
        t_lo = (t - t_hi.float()).to(torch.bfloat16)
        return torch.cat([ws.to(torch.bfloat16), t_hi, t_lo], dim=1).contiguous()
    return _derived(weight, ('posfold', lo, hi, pos_rows, E.data_ptr()), make)


def _pos_proj_ok(x, d, rows):
    return (x.is_cuda and x.dim() == 2 and compute_dtype(x) == torch.bfloat16 and d in (128, 256) and rows % 64 == 0
            and x.shape[0] >= _TOKEN_GEMM_MIN_ROWS and x.shape[0] * rows * 2 < 2 ** 31)


class _PosProj(torch.autograd.Function):
    """out [m, hi-lo] = (x [+ pos]) W[lo:hi]^T + b[lo:hi] -- the attention in-projections of ONE token list as one GEMM:
    rows p0:p1 of W (p0 == lo) see x + pos (q, k), the others x (v).  The position embedding never touches [m,d]: forward
    through the one-hot k-step of tmae_token_gemm_pos, backward as dW[p0:p1] += dcell^T E with the per-cell column sums of
    dOut[:, p0:p1] that the weight-gradient pass returns (tmae_linear_wgrad_cells).
    With `fork`, x itself comes back as a last output (residual branch) and its gradient is accumulated in place."""

    @staticmethod
    def forward(ctx, x, weight, bias, lo, hi, p0, p1, cells, E, fork, inplace_dx):
        assert p0 == lo
        x_c = x.to(torch.bfloat16).contiguous()
        m, d = x_c.shape
        w_aug = pos_fold_weight(weight, lo, hi, (p0, p1), E)
        b_c = _zero_bias(hi - lo, x.device) if bias is None else cast_param(bias, torch.bfloat16)[lo:hi]
        out = torch.empty((m, hi - lo), dtype=torch.bfloat16, device=x.device)
        check(lib.tmae_token_gemm_pos(_p(x_c), x_c.stride(0), m, d, _p(w_aug), hi - lo, _p(b_c), _p(cells), _p(out),
                                      hi - lo, _s()), 'tmae_token_gemm_pos')
        ctx.save_for_backward(x_c, w_aug, cells, E)
        ctx.rng = (lo, hi, p0, p1, weight.shape[0])
        # W[lo:hi]^T for the in-place input gradient, cached with the folded weight (it was transposed again in every backward:
        # one launch per attention layer and step)
        ctx.wt = (_derived(weight, ('plainT', lo, hi), lambda w: w[lo:hi].to(torch.bfloat16).t().contiguous())
                  if (inplace_dx and fork) else None)
        ctx.fork, ctx.has_bias = fork, bias is not None
        ctx.inplace_dx = bool(inplace_dx and fork)
        ctx.dtypes = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        ctx.set_materialize_grads(False)
        return (out, x.view_as(x)) if fork else out

    @staticmethod
    def backward(ctx, dout, dalias=None):
        x_c, w_aug, cells, E = ctx.saved_tensors
        lo, hi, p0, p1, rows = ctx.rng
        xdt, wdt, bdt = ctx.dtypes
        m, d = x_c.shape
        dx = dalias.to(torch.bfloat16) if dalias is not None else None
        dW = dB = None
        if dout is not None:
            dout = dout.to(torch.bfloat16)
            if dout.stride(-1) != 1:
                dout = dout.contiguous()
            w = w_aug[:, :d]                                   # the plain weight rows (pitch d + 32)
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx = dout @ w
                elif ctx.inplace_dx:
                    addmm_inplace(dx, dout, w, ctx.wt)         # see _ProjFork.backward
                else:
                    dx = torch.addmm(dx, dout, w)
            if ctx.needs_input_grad[1]:
                full = lo == 0 and hi == rows
                mk = torch.empty if full else torch.zeros
                dW = mk((rows, d), dtype=torch.float32, device=x_c.device)
                want_b = ctx.has_bias and ctx.needs_input_grad[2]
                dB = mk((rows,), dtype=torch.float32, device=x_c.device) if want_b else None
                # (p0 == lo: the position rows lead the slice; dcell^T E is added inside the slab reduction)
                linear_wgrad(dout, x_c, want_b, out_w=dW[lo:hi], out_b=None if dB is None else dB[lo:hi], cells=cells,
                             pos_n=p1 - p0, pos_e=E)
        return (None if dx is None else dx.to(xdt), None if dW is None else dW.to(wdt),
                None if dB is None else dB.to(bdt), None, None, None, None, None, None, None, None)


class _PosProjCross(torch.autograd.Function):
    """The in-projections of a cross-attention layer (wca_block.py:50-60) as ONE autograd node:
        q  [mq, d]  = (x_q + pos) W[:d]^T + b[:d]            (rows 0:d take the position)
        kv [mk, 2d] = [(x_kv + pos) W[d:2d]^T | x_kv W[2d:]^T] + b[d:]
    plus x_q's alias for the residual branch.  Two position-folded GEMMs forward; backward writes the two row ranges of ONE
    [3d, d] weight gradient (as two separate nodes each returned a zero-padded full-size gradient that autograd then
    added: two fills and two adds per layer and parameter).
    kv_alias: a fourth output, an alias of x_kv, for the NEXT cross layer to take as its x_kv (both layers of a block read the
    same previous-frame rows, wca_block.py:106-145): that layer's input gradient then arrives here through the alias and this
    layer's k | v input gradient is accumulated into it by the GEMM itself -- autograd no longer adds two [m_prev, d] gradients
    (one elementwise pass per stage) in front of the previous frame's last norm."""

    @staticmethod
    def forward(ctx, x_q, x_kv, weight, bias, cells_q, cells_k, E, inplace_dx, kv_alias=False):
        xq, xk = x_q.to(torch.bfloat16).contiguous(), x_kv.to(torch.bfloat16).contiguous()
        d = xq.shape[1]
        wq = pos_fold_weight(weight, 0, d, (0, d), E)
        wkv = pos_fold_weight(weight, d, 3 * d, (d, 2 * d), E)
        b_c = _zero_bias(3 * d, x_q.device) if bias is None else cast_param(bias, torch.bfloat16)
        q = torch.empty((xq.shape[0], d), dtype=torch.bfloat16, device=x_q.device)
        kv = torch.empty((xk.shape[0], 2 * d), dtype=torch.bfloat16, device=x_q.device)
        check(lib.tmae_token_gemm_pos(_p(xq), xq.stride(0), xq.shape[0], d, _p(wq), d, _p(b_c), _p(cells_q), _p(q), d, _s()),
              'tmae_token_gemm_pos')
        check(lib.tmae_token_gemm_pos(_p(xk), xk.stride(0), xk.shape[0], d, _p(wkv), 2 * d, _p(b_c[d:]), _p(cells_k), _p(kv),
                                      2 * d, _s()), 'tmae_token_gemm_pos')
        ctx.save_for_backward(xq, xk, wq, wkv, cells_q, cells_k, E)
        # (W[d:3d])^T for the k | v input gradient on the token GEMM (contraction 2d -> d), cached with the weight
        ctx.wkvT = _derived(weight, ('kvT', d), lambda w: w[d:3 * d].to(torch.bfloat16).t().contiguous())
        ctx.has_bias, ctx.inplace_dx = bias is not None, bool(inplace_dx)
        ctx.dtypes = (x_q.dtype, x_kv.dtype, weight.dtype, None if bias is None else bias.dtype)
        ctx.set_materialize_grads(False)
        if kv_alias:
            return q, kv, x_q.view_as(x_q), x_kv.view_as(x_kv)
        return q, kv, x_q.view_as(x_q)

    @staticmethod
    def backward(ctx, dq, dkv, dalias, dkalias=None):
        xq, xk, wq, wkv, cells_q, cells_k, E = ctx.saved_tensors
        qdt, kdt, wdt, bdt = ctx.dtypes
        d = xq.shape[1]
        dxq = dalias.to(torch.bfloat16) if dalias is not None else None
        dxk = None

        def cont(t):
            t = t.to(torch.bfloat16)
            return t if t.stride(-1) == 1 else t.contiguous()
        dq = None if dq is None else cont(dq)
        dkv = None if dkv is None else cont(dkv)
        if ctx.needs_input_grad[0] and dq is not None:
            if dxq is None:
                dxq = dq @ wq[:, :d]
            elif ctx.inplace_dx:
                addmm_inplace(dxq, dq, wq[:, :d])              # see _ProjFork.backward
            else:
                dxq = torch.addmm(dxq, dq, wq[:, :d])
        if ctx.needs_input_grad[1] and dkalias is not None:
            dxk = dkalias.to(torch.bfloat16)
            if not dxk.is_contiguous():
                dxk = dxk.contiguous()
        if ctx.needs_input_grad[1] and dkv is not None:
            if dxk is not None and ctx.inplace_dx:
                addmm_inplace(dxk, dkv, wkv[:, :d], ctx.wkvT)      # into the later layer's gradient (see kv_alias)
            elif dxk is not None:
                dxk = torch.addmm(dxk, dkv, wkv[:, :d])
            else:
                dxk = token_gemm(dkv, ctx.wkvT) if _tg_ok(dkv, 2 * d, d) else dkv @ wkv[:, :d]
        dW = dB = None
        if ctx.needs_input_grad[2]:
            mk = torch.empty if (dq is not None and dkv is not None) else torch.zeros
            dW = mk((3 * d, d), dtype=torch.float32, device=xq.device)
            want_b = ctx.has_bias and ctx.needs_input_grad[3]
            dB = mk((3 * d,), dtype=torch.float32, device=xq.device) if want_b else None
            if dq is not None:
                linear_wgrad(dq, xq, want_b, out_w=dW[:d], out_b=None if dB is None else dB[:d], cells=cells_q, pos_n=d, pos_e=E)
            if dkv is not None:
                linear_wgrad(dkv, xk, want_b, out_w=dW[d:], out_b=None if dB is None else dB[d:], cells=cells_k, pos_n=d, pos_e=E)
        return (None if dxq is None else dxq.to(qdt), None if dxk is None else dxk.to(kdt),
                None if dW is None else dW.to(wdt), None if dB is None else dB.to(bdt), None, None, None, None, None)


def pos_proj_cross(x_q, x_kv, weight, bias, cells_q, cells_k, E, inplace_dx=False, kv_alias=False):
    """See _PosProjCross: (q [mq,d], kv [mk,2d], alias of x_q[, alias of x_kv])."""
    return _PosProjCross.apply(x_q, x_kv, weight, bias, cells_q, cells_k, E, bool(inplace_dx), bool(kv_alias))


def pos_proj(x, weight, bias, lo, hi, p0, p1, cells, E, fork=False, inplace_dx=False):
    """See _PosProj (bf16 GPU path; callers check _pos_proj_ok first)."""
    return _PosProj.apply(x, weight, bias, int(lo), int(hi), int(p0), int(p1), cells, E, bool(fork), bool(inplace_dx))


class _GeluLinear(torch.autograd.Function):
    """y = gelu(h_pre) W^T + b, the second half of the encoder FFN (exact erf GELU, sst_basic_block.py:81).
    Backward: dW / db from the token-split kernel on h = gelu(h_pre); d h_pre = (dy W) * gelu'(h_pre) in ONE pass
    (tmae_token_gemm_dgelu: the GEMM's epilogue reads h_pre) instead of a GEMM plus an elementwise GeluBackward."""

    @staticmethod
    def forward(ctx, h_pre, weight, bias, h=None, residual=None):
        """residual [m, d] (optional): y = residual + gelu(h_pre) W^T + b -- `src + linear2(act)` of the encoder layer
        (sst_basic_block.py:81-83): the add in front of norm2 rides on this GEMM (tmae_token_gemm_res: the residual tile travels
        through the kernel's LDS ring), so the norm reads ONE tensor and does not write the sum again."""
        cdt = compute_dtype(h_pre)
        hp = h_pre.to(cdt).contiguous()
        if h is None or h.dtype != cdt or h.shape != hp.shape:        # h = gelu(h_pre) already made by the producing GEMM
            h = torch.nn.functional.gelu(hp)
        w_c = cast_param(weight, cdt)
        b_c = None if bias is None else cast_param(bias, cdt)
        n, k = w_c.shape                                              # [d, dff]
        m = h.shape[0]
        y = None
        if residual is not None:
            r = residual.to(cdt)
            if (cdt == torch.bfloat16 and (k, n) in ((512, 256), (256, 128)) and m >= (65536 if k == 512 else 32768)
                    and r.is_contiguous() and r.shape == (m, n) and h.is_contiguous() and w_c.is_contiguous()
                    and m * k * 2 < 2 ** 31 and r.data_ptr() % 16 == 0 and h.data_ptr() % 16 == 0):
                y = torch.empty((m, n), dtype=cdt, device=h.device)
                bb = b_c.contiguous() if b_c is not None else _zero_bias(n, h.device)
                check(lib.tmae_token_gemm_res(_p(h), k, m, k, _p(w_c), n, _p(bb), _p(r), _p(y), n, _s()), 'tmae_token_gemm_res')
            else:
                y = token_gemm(h, w_c, b_c) + r
        if y is None:
            y = token_gemm(h, w_c, b_c)
        ctx.save_for_backward(hp, h, w_c)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.dtypes = (h_pre.dtype, weight.dtype, None if bias is None else bias.dtype, None if residual is None else residual.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        hp, h, w_c = ctx.saved_tensors
        xdt, wdt, bdt, rdt = ctx.dtypes
        dy = dy.to(hp.dtype)
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        dhp = None
        if ctx.needs_input_grad[0]:
            n, k = w_c.shape                                     # [d, dff]
            if _tg_ok(dy, n, k) and w_c.dtype == torch.bfloat16 and hp.dtype == torch.bfloat16:
                m = dy.shape[0]
                dhp = torch.empty((m, k), dtype=torch.bfloat16, device=dy.device)
                zb = _zero_bias(k, dy.device)
                wt = _transposed(w_c)
                check(lib.tmae_token_gemm_dgelu(_p(dy), dy.stride(0), m, n, _p(wt), k, _p(zb), _p(hp), _p(dhp), k, _s()),
                      'tmae_token_gemm_dgelu')
            else:
                dhp = torch.ops.aten.gelu_backward(dy @ w_c, hp)
            dhp = dhp.to(xdt)
        if _wgrad_ok(dy, h):
            dw, db = linear_wgrad(dy, h, ctx.has_bias)
        else:
            dw = dy.float().t() @ h.float()
            db = dy.float().sum(0) if ctx.has_bias else None
        # the residual's gradient IS dy (no copy: a later in-place accumulation into it, ops.proj_fork(inplace_dx=True), runs
        # after this node has read it -- one stream)
        return dhp, dw.to(wdt), (db.to(bdt) if ctx.has_bias else None), None, (dy.to(rdt) if ctx.has_res else None)


def gelu_linear(h_pre, weight, bias=None, h=None, residual=None):
    """linear(gelu(h_pre), weight, bias) with the GELU backward fused into the input-gradient GEMM (GPU, 2-D).
    h: gelu(h_pre) if the caller already has it (ops.proj_fork(..., gelu=True)); never differentiated through."""
    if h_pre.is_cuda and h_pre.dim() == 2:
        return _GeluLinear.apply(h_pre, weight, bias, h, residual)
    y = torch.nn.functional.linear(torch.nn.functional.gelu(h_pre), weight, bias)
    return y if residual is None else y + residual


def linear(x, weight, bias=None):
    """torch.nn.functional.linear with the token-split weight-gradient kernel (2-D inputs on the GPU)."""
    if x.is_cuda and x.dim() == 2:
        return _Linear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


class _AddLayerNorm(torch.autograd.Function):
    """y = LayerNorm(a + b * bmask) + post: one fused pass forward, one backward (+ fixed-order gamma/beta sums).
    bmask [m] (0/1 rows of b that count), post [m,d] (added after the norm) are optional.  passthrough: also returns an
    alias of `a`; the gradient that arrives for the alias is summed into a's gradient INSIDE the backward kernel -- the
    encoder blocks hand the block input to their last norm this way (post), so the block residual `x + encoder(x)`
    costs no elementwise pass forward and no AccumulateGrad add backward."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, eps, bmask, post, passthrough):
        cdt = compute_dtype(a)
        a_c = a.to(cdt).contiguous()
        b_c = None if b is None else b.to(cdt).contiguous()
        m, d = a_c.shape
        y = torch.empty_like(a_c)
        xs = torch.empty_like(a_c) if b_c is not None else None
        mean = torch.empty((m,), dtype=torch.float32, device=a.device)
        rstd = torch.empty((m,), dtype=torch.float32, device=a.device)
        g32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        bm = None if (bmask is None or b_c is None) else bmask.to(cdt).reshape(-1).contiguous()
        po = None if post is None else post.to(cdt).contiguous()
        assert bm is None or bm.shape[0] == m
        assert po is None or po.shape == a_c.shape
        check(lib.tmae_add_layernorm_fwd(_p(a_c), _p(b_c), _dt(a_c), m, d, _p(g32), _p(b32), float(eps), _p(xs), _p(y),
                                         _p(mean), _p(rstd), _p(bm), _p(po), _s()), 'tmae_add_layernorm_fwd')
        ctx.save_for_backward(xs if xs is not None else a_c, mean, rstd, g32, bm)
        ctx.meta = (a.dtype, None if b is None else b.dtype, weight.dtype, bias.dtype, None if post is None else post.dtype)
        ctx.passthrough = bool(passthrough)
        ctx.set_materialize_grads(False)
        if passthrough:
            return y, a.view_as(a)
        return y

    @staticmethod
    def backward(ctx, dy, dalias=None):
        x, mean, rstd, g32, bm = ctx.saved_tensors
        adt, bdt, wdt, btdt, pdt = ctx.meta
        m, d = x.shape
        if dy is None:                        # only the alias was used downstream
            z = None if dalias is None else dalias.to(adt)
            return z, None, None, None, None, None, None, None
        dy = dy.to(x.dtype).contiguous()
        skip = None if dalias is None else dalias.to(x.dtype).contiguous()
        need_b = bdt is not None
        dx_skip = torch.empty_like(x) if skip is not None else None
        dx_b = torch.empty_like(x) if (bm is not None and need_b) else None
        need_dx = skip is None or (need_b and dx_b is None)
        dx = torch.empty_like(x) if need_dx else None
        dg = torch.empty((d,), dtype=torch.float32, device=x.device)
        db = torch.empty((d,), dtype=torch.float32, device=x.device)
        wsb = lib.tmae_layernorm_bwd_workspace(m, d)
        ws = _ws(wsb, x.device)
        check(lib.tmae_layernorm_bwd(_p(dy), _p(x), _dt(x), m, d, _p(mean), _p(rstd), _p(g32), _p(dx), _p(dg), _p(db),
                                     _p(skip), _p(dx_skip), _p(bm if dx_b is not None else None), _p(dx_b),
                                     _p(ws), wsb, _s()), 'tmae_layernorm_bwd')
        da = dx_skip if skip is not None else dx
        dbb = None if not need_b else (dx_b if dx_b is not None else dx)
        return (da.to(adt), None if dbb is None else dbb.to(bdt), dg.to(wdt), db.to(btdt), None, None,
                None if pdt is None else dy.to(pdt), None)


def add_layer_norm(a, b, weight, bias, eps=1e-5, bmask=None, post=None, passthrough=False):
    """LayerNorm(a + b) (b may be None) with nn.LayerNorm semantics; fused HIP kernel for d in {128, 256}.
    bmask [m] or [m,1]: b's rows are scaled by it; post [m,d]: added to the result; passthrough: returns (y, alias of a)
    -- see _AddLayerNorm."""
    if a.is_cuda and a.dim() == 2 and a.shape[1] in (128, 256):
        return _AddLayerNorm.apply(a, b, weight, bias, eps, bmask, post, passthrough)
    bb = b if (b is None or bmask is None) else b * bmask.reshape(-1, 1)
    x = a if bb is None else a + bb
    y = torch.nn.functional.layer_norm(x, (x.shape[-1],), weight, bias, eps)
    if post is not None:
        y = y + post
    return (y, a) if passthrough else y


def _sync_group(bn):
    """The process group over which `bn` shares its batch statistics, or None: a torch.nn.SyncBatchNorm module (what
    tools/train.py --sync_bn makes of every BatchNorm, as the reference does with convert_sync_batchnorm,
    tools/train.py:244-245) in training mode under an initialised process group of more than one rank."""
    import torch.distributed as dist
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not bn.training:
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    pg = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return pg if dist.get_world_size(pg) > 1 else None


def _merge_stats(mean, var, count, pg):
    """Batch statistics of the union of every rank's rows from the per-rank (mean, biased var, row count): one
    all_gather of [2c + 1] floats per rank, merged in float64 (the pairwise update of Chan et al.: no E[x^2] - E[x]^2
    cancellation).  Returns (mean, biased var) f32 [c] and the total count (float)."""
    import torch.distributed as dist
    c = mean.numel()
    mine = torch.cat([mean.reshape(-1).double(), var.reshape(-1).double(), mean.new_full((1,), float(count)).double()])
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size(pg))]
    dist.all_gather(parts, mine, group=pg)
    allp = torch.stack(parts)                                # [W, 2c + 1]
    n = allp[:, 2 * c:]                                      # [W, 1]
    tot = n.sum()
    gmean = (allp[:, :c] * n).sum(0) / tot
    gvar = ((allp[:, c:2 * c] + (allp[:, :c] - gmean) ** 2) * n).sum(0) / tot
    return gmean.float(), gvar.float(), float(tot)


class _BatchNormReLU(torch.autograd.Function):
    """`bounds` = row offsets [0, r1, ..., m]: every row range is normalised with its own batch statistics (the
    Siamese encoder runs both frames as one token list; the reference normalises each frame's call separately).
    pg (a process group, SyncBatchNorm): statistics and the two backward sums are taken over the rows of ALL ranks
    (torch.nn.SyncBatchNorm semantics: the gradients of gamma / beta stay the rank's own sums, DDP averages them)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu, bounds, pg=None, post=None, fork=False):
        """post [m, c] (single row range, no process group): y = relu?(norm(x)) + post; its gradient is dy itself.
        fork (two row ranges, no process group): also returns the two row ranges of y as outputs of THIS node; what arrives for
        them in the backward is added to dy inside the BatchNorm backward kernels (tmae_bn_relu_bwd2) -- the Siamese encoder's
        stage output goes on to the next stage whole and to the cross-attention block split by frame (SiamWCA_MAE.py:262-291),
        and autograd's way of joining the two (cat of the halves, then an add over [m, c]) was two elementwise passes per stage."""
        x = x.contiguous()
        m, c = x.shape
        ng = len(bounds) - 1
        ctx.fork = bool(fork)
        if fork:
            assert ng == 2 and pg is None and post is None
        if post is not None:
            assert ng == 1 and pg is None and post.shape == x.shape
            post = post.to(x.dtype).contiguous()
        ctx.has_post = post is not None
        y = torch.empty_like(x)
        mean = torch.empty((ng, c), dtype=torch.float32, device=x.device)
        var, rstd = torch.empty_like(mean), torch.empty_like(mean)
        g32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        counts = []
        for g in range(ng):
            r0, r1 = bounds[g], bounds[g + 1]
            wsb = lib.tmae_bn_workspace(r1 - r0, c)
            ws = _ws(wsb, x.device)
            if pg is None:
                if post is not None:
                    check(lib.tmae_bn_relu_add_fwd(_p(x), _dt(x), m, c, _p(g32), _p(b32), float(eps), 1 if relu else 0, _p(post),
                                                   _p(y), _p(mean[g]), _p(var[g]), _p(rstd[g]), _p(ws), wsb, _s()),
                          'tmae_bn_relu_add_fwd')
                else:
                    check(lib.tmae_bn_relu_fwd(_p(x[r0:r1]), _dt(x), r1 - r0, c, _p(g32), _p(b32), float(eps), 1 if relu else 0,
                                               _p(y[r0:r1]), _p(mean[g]), _p(var[g]), _p(rstd[g]), _p(ws), wsb, _s()),
                          'tmae_bn_relu_fwd')
                counts.append(float(r1 - r0))
                continue
            check(lib.tmae_bn_stats(_p(x[r0:r1]), _dt(x), r1 - r0, c, float(r1 - r0), float(eps), _p(mean[g]), _p(var[g]),
                                    _p(rstd[g]), _p(ws), wsb, _s()), 'tmae_bn_stats')
            gm, gv, tot = _merge_stats(mean[g], var[g], r1 - r0, pg)
            mean[g], var[g], rstd[g] = gm, gv, torch.rsqrt(gv + eps)
            counts.append(tot)
            check(lib.tmae_bn_apply(_p(x[r0:r1]), _dt(x), r1 - r0, c, _p(mean[g]), _p(rstd[g]), _p(g32), _p(b32),
                                    1 if relu else 0, _p(y[r0:r1]), _s()), 'tmae_bn_apply')
        ctx.save_for_backward(x, mean, rstd, g32, b32)
        ctx.relu, ctx.bounds, ctx.pg, ctx.counts = relu, bounds, pg, counts
        _BatchNormReLU.last_counts = counts            # rows behind every group's statistics (all ranks under SyncBatchNorm)
        ctx.dtypes = (weight.dtype, bias.dtype)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)       # no zero-filled "gradients" for the statistics outputs (2 fills per layer)
        if fork:
            return y, mean, var, y[:bounds[1]], y[bounds[1]:]
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _m, _v, d0=None, d1=None):
        if dy is None and d0 is None and d1 is None:
            return (None,) * 9
        x, mean, rstd, g32, b32 = ctx.saved_tensors
        m, c = x.shape
        bounds = ctx.bounds
        ng = len(bounds) - 1
        if dy is not None:
            dy = dy.to(x.dtype).contiguous()
        # per row range: the gradient (first) and, with `fork`, what arrived for that range's own output (second); a range
        # nobody sent a gradient for gets zeros
        first = [None if dy is None else dy[bounds[g]:bounds[g + 1]] for g in range(ng)]
        second = [None] * ng
        if ctx.fork:
            for g, d in enumerate((d0, d1)):
                if d is None:
                    continue
                d = d.to(x.dtype).contiguous()
                if first[g] is None:
                    first[g] = d
                else:
                    second[g] = d
        for g in range(ng):
            if first[g] is None:
                first[g] = torch.zeros((bounds[g + 1] - bounds[g], c), dtype=x.dtype, device=x.device)
        dx = torch.empty_like(x)
        dgb = torch.empty((ng, 2, c), dtype=torch.float32, device=x.device)     # [group][gamma | beta]: one sum over the groups below
        dg, db = dgb[:, 0], dgb[:, 1]
        for g in range(ng):
            r0, r1 = bounds[g], bounds[g + 1]
            wsb = lib.tmae_bn_workspace(r1 - r0, c)
            ws = _ws(wsb, x.device)
            if ctx.pg is None:
                if second[g] is not None:
                    check(lib.tmae_bn_relu_bwd2(_p(first[g]), _p(second[g]), _p(x[r0:r1]), _dt(x), r1 - r0, c, _p(mean[g]),
                                                _p(rstd[g]), _p(g32), _p(b32), 1 if ctx.relu else 0, _p(dx[r0:r1]), _p(dg[g]),
                                                _p(db[g]), _p(ws), wsb, _s()), 'tmae_bn_relu_bwd2')
                    continue
                check(lib.tmae_bn_relu_bwd(_p(first[g]), _p(x[r0:r1]), _dt(x), r1 - r0, c, _p(mean[g]), _p(rstd[g]), _p(g32),
                                           _p(b32), 1 if ctx.relu else 0, _p(dx[r0:r1]), _p(dg[g]), _p(db[g]), _p(ws), wsb,
                                           _s()), 'tmae_bn_relu_bwd')
                continue
            import torch.distributed as dist
            check(lib.tmae_bn_bwd_sums(_p(first[g]), _p(x[r0:r1]), _dt(x), r1 - r0, c, _p(mean[g]), _p(rstd[g]), _p(g32),
                                       _p(b32), 1 if ctx.relu else 0, _p(db[g]), _p(dg[g]), _p(ws), wsb, _s()),
                  'tmae_bn_bwd_sums')
            tot = torch.stack([db[g], dg[g]])                 # the sums of every rank enter dx; dgamma / dbeta stay local
            dist.all_reduce(tot, group=ctx.pg)
            tb, tg = tot[0].contiguous(), tot[1].contiguous()
            check(lib.tmae_bn_bwd_apply(_p(first[g]), _p(x[r0:r1]), _dt(x), r1 - r0, c, _p(mean[g]), _p(rstd[g]), _p(g32),
                                        _p(b32), 1 if ctx.relu else 0, _p(tb), _p(tg), float(ctx.counts[g]), _p(dx[r0:r1]),
                                        _s()), 'tmae_bn_bwd_apply')
        if ng > 1:
            dg, db = dgb.sum(0)
        else:
            dg, db = dg[0], db[0]
        return dx, dg.to(ctx.dtypes[0]), db.to(ctx.dtypes[1]), None, None, None, None, (dy if ctx.has_post else None), None


class _DeadBias(torch.autograd.Function):
    """y = x, with `bias` attached to the graph and an all-zero gradient: a per-channel bias in front of a training-mode
    BatchNorm drops out of its output, and its gradient -- the sum of a zero-mean BatchNorm input gradient -- is zero."""

    @staticmethod
    def forward(ctx, x, bias):
        ctx.meta = (bias.shape, bias.dtype, bias.device)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        shape, dtype, device = ctx.meta
        return dy, torch.zeros(shape, dtype=dtype, device=device)


def batch_norm_relu(x, bn, relu=True, groups=None, pre_bias=None, post=None, fork=False):
    """nn.BatchNorm1d `bn` (+ ReLU) over the rows of x [m,c]; fused HIP kernels in training mode for c in
    {64,128,256}; updates bn's running statistics like torch does.  `groups` (row counts summing to m): each row
    range is a separate BatchNorm call (own batch statistics, running statistics updated in order).
    pre_bias [c]: the result for x + pre_bias WITHOUT adding it (a conv bias before the norm, USE_BIAS_BEFORE_NORM,
    center_head.py:19-27): the normalised output does not depend on it, its gradient is exactly zero, only the running
    mean sees it -- saves an elementwise pass over the activation forward and a column reduction backward.
    fork (two groups): returns (y, (y[:g0], y[g0:])) -- the halves as outputs of the norm's own autograd node, see
    _BatchNormReLU.forward; where the fused kernels do not apply the halves come from ops.split_rows."""
    if pre_bias is not None and not (x.is_cuda and x.dim() == 2 and bn.training and x.shape[1] in (64, 128, 256)
                                     and groups is None and x.shape[0] > 1 and x.dtype in (torch.float32, torch.bfloat16)):
        x, pre_bias = x + pre_bias.to(x.dtype), None
    sizes = [int(x.shape[0])] if groups is None else [int(g) for g in groups]
    assert sum(sizes) == x.shape[0]
    if (x.is_cuda and x.dim() == 2 and bn.training and x.shape[1] in (64, 128, 256) and min(sizes) > 1
            and x.dtype in (torch.float32, torch.bfloat16)):
        bounds = [0]
        for g in sizes:
            bounds.append(bounds[-1] + g)
        pg = _sync_group(bn)
        halves = None
        if post is not None and (groups is not None or pg is not None):
            y, mean, var = _BatchNormReLU.apply(x, bn.weight, bn.bias, bn.eps, relu, tuple(bounds), pg)
            y = y + post
        elif fork and len(sizes) == 2 and pg is None and post is None and pre_bias is None:
            y, mean, var, h0, h1 = _BatchNormReLU.apply(x, bn.weight, bn.bias, bn.eps, relu, tuple(bounds), None, None, True)
            halves = (h0, h1)
        else:
            y, mean, var = _BatchNormReLU.apply(x, bn.weight, bn.bias, bn.eps, relu, tuple(bounds), pg, post)
        if pre_bias is not None:
            y = _DeadBias.apply(y, pre_bias)
            mean = mean + pre_bias.detach().float()
        if bn.track_running_stats:
            for g, m in enumerate(_BatchNormReLU.last_counts):      # SyncBatchNorm: the rows of all ranks (unbiased variance)
                _bn_running_update(bn, mean[g], var[g], m)
        if fork:
            return y, (halves if halves is not None else split_rows(y, sizes[0]))
        return y
    if _sync_group(bn) is not None:
        raise NotImplementedError('SyncBatchNorm: this layer shape has no fused kernel (channels in {64, 128, 256}, bf16 / fp32 '
                                  'rows on the GPU); the library path would need an RCCL-capable torch SyncBatchNorm call here')
    if len(sizes) > 1:
        ys = [bn(part) for part in torch.split(x, sizes)]
        y = torch.cat(ys, 0)
    else:
        y = bn(x)
    y = torch.relu(y) if relu else y
    y = y if post is None else y + post
    return (y, split_rows(y, sizes[0])) if fork else y


class _BNReLUGather(torch.autograd.Function):
    """y = relu?(BatchNorm(x)) over the rows of a dense map x [cells, c] AND rows = y at m sites, one autograd node
    (SiamWCA_MAE.py:100-115 then :303-312: the decoder's last norm, read back at the current frame's voxels).  When only `rows`
    carries a gradient -- pre-training: nothing else reads the dense map -- the backward never builds the dense, 79 % zero dy:
    its sums run over the m gathered rows and the apply pass picks dz through the cell -> row map (tmae_bn_relu_bwd_gathered)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu, rowmap, indices, batch, ny, nx, moments=None):
        """moments [2, c] f32 (optional): the column sums of x and x^2, already taken by x's producer (the decoder conv's epilogue):
        mean / variance come from them (double precision, a handful of [c]-sized launches) and the statistics pass over x is skipped."""
        x = x.contiguous()
        cells, c = x.shape
        assert cells == batch * ny * nx
        y = torch.empty_like(x)
        g32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        if moments is not None:
            mo = moments.double() / float(cells)
            mean64 = mo[0]
            var64 = (mo[1] - mean64 * mean64).clamp_(min=0.0)
            mean, var = mean64.float(), var64.float()
            rstd = torch.rsqrt(var64 + float(eps)).float()
            check(lib.tmae_bn_apply(_p(x), _dt(x), cells, c, _p(mean), _p(rstd), _p(g32), _p(b32), 1 if relu else 0, _p(y), _s()),
                  'tmae_bn_apply')
        else:
            mean = torch.empty((c,), dtype=torch.float32, device=x.device)
            var, rstd = torch.empty_like(mean), torch.empty_like(mean)
            wsb = lib.tmae_bn_workspace(cells, c)
            ws = _ws(wsb, x.device)
            check(lib.tmae_bn_relu_fwd(_p(x), _dt(x), cells, c, _p(g32), _p(b32), float(eps), 1 if relu else 0, _p(y), _p(mean), _p(var),
                                       _p(rstd), _p(ws), wsb, _s()), 'tmae_bn_relu_fwd')
        m = indices.shape[0]
        rows = torch.empty((m, c), dtype=x.dtype, device=x.device)
        check(lib.tmae_dense_gather(_p(y), _dt(y), batch, ny, nx, c, _p(indices), m, _p(rows), _s()), 'tmae_dense_gather')
        ctx.save_for_backward(x, mean, rstd, g32, b32, rowmap, indices)
        ctx.meta = (bool(relu), int(batch), int(ny), int(nx), weight.dtype, bias.dtype)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        return y, rows, mean, var

    @staticmethod
    def backward(ctx, dy, drows, _m, _v):
        if dy is None and drows is None:
            return (None,) * 11
        x, mean, rstd, g32, b32, rowmap, indices = ctx.saved_tensors
        relu, batch, ny, nx, wdt, bdt = ctx.meta
        cells, c = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty((c,), dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
        if dy is None:
            drows = drows.to(x.dtype).contiguous()
            wsb = lib.tmae_bn_workspace(drows.shape[0], c)
            ws = _ws(wsb, x.device)
            check(lib.tmae_bn_relu_bwd_gathered(_p(drows), _p(indices), _p(rowmap), drows.shape[0], _p(x), _dt(x), batch, ny, nx, c,
                                                _p(mean), _p(rstd), _p(g32), _p(b32), 1 if relu else 0, _p(dx), _p(dg), _p(db),
                                                _p(ws), wsb, _s()), 'tmae_bn_relu_bwd_gathered')
        else:                                        # the dense map has a reader of its own: the plain backward on the sum
            dy = dy.to(x.dtype).contiguous()
            if drows is not None:
                sc = torch.empty_like(x)
                check(lib.tmae_sparse_to_dense(_p(drows.to(x.dtype).contiguous()), _dt(x), drows.shape[0], c, _p(rowmap), batch, ny, nx,
                                               _p(sc), _s()), 'tmae_sparse_to_dense')
                dy = dy + sc
            wsb = lib.tmae_bn_workspace(cells, c)
            ws = _ws(wsb, x.device)
            check(lib.tmae_bn_relu_bwd(_p(dy), _p(x), _dt(x), cells, c, _p(mean), _p(rstd), _p(g32), _p(b32), 1 if relu else 0,
                                       _p(dx), _p(dg), _p(db), _p(ws), wsb, _s()), 'tmae_bn_relu_bwd')
        return dx, dg.to(wdt), db.to(bdt), None, None, None, None, None, None, None, None


def batch_norm_relu_gather(x_rows, bn, relu, rowmap, indices, batch, ny, nx, moments=None):
    """(y, rows): ops.batch_norm_relu over the rows of a dense map [batch * ny * nx, c] and ops.dense_gather of the result at
    `indices` ([m, 3] int32 unique sites; rowmap = ops.index_grid(indices, ...)), as one autograd node (_BNReLUGather).  Falls
    back to the two ops where the fused kernels do not apply."""
    cells, c = x_rows.shape
    ok = (x_rows.is_cuda and bn.training and c in (64, 128, 256) and cells > 1 and _sync_group(bn) is None
          and x_rows.dtype in (torch.float32, torch.bfloat16) and indices.shape[0] > 0)
    if not ok:
        y = batch_norm_relu(x_rows, bn, relu=relu)
        return y, dense_gather(y.view(batch, ny, nx, c), rowmap, indices)
    if moments is not None and (moments.numel() != 2 * c or not _DENSE_SUMS):
        moments = None
    y, rows, mean, var = _BNReLUGather.apply(x_rows, bn.weight, bn.bias, bn.eps, relu, rowmap, indices, batch, ny, nx, moments)
    if bn.track_running_stats:
        _bn_running_update(bn, mean, var, float(cells))
    return y, rows


class _SplitRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m0):
        ctx.m0 = m0
        return x[:m0], x[m0:]

    @staticmethod
    def backward(ctx, d0, d1):
        return torch.cat([d0, d1], 0), None           # one pass (narrow's own backward pads + adds twice)


def split_rows(x, m0):
    """x[:m0], x[m0:] as views; the backward is a single concatenation."""
    return _SplitRows.apply(x, int(m0))


# ----------------------------------------------------------------------------- voxelisation (A1)

def voxelize_launch(points, batch_size, pc_range, voxel_size, grid_size):
    """Enqueue the voxelisation of one frame; returns worst-case buffers + device counts (no sync).
    Replaces get_in_range_mask + unique(dim=0) (temporal_dyn_vfe.py:67-72)."""
    _need_cuda(points)
    pts = points.contiguous().float()
    n, row = pts.shape
    gx, gy, gz = (int(v) for v in grid_size)
    dev = pts.device
    out = dict(
        points=torch.empty((n, row), dtype=torch.float32, device=dev),
        point_coords=torch.empty((n, 4), dtype=torch.int64, device=dev),
        inverse=torch.empty((n,), dtype=torch.int64, device=dev),
        voxel_coords=torch.empty((min(n, batch_size * gx * gy * gz), 4), dtype=torch.int64, device=dev),
        counts=torch.zeros((2 + batch_size,), dtype=torch.int32, device=dev),
    )
    wsb = lib.tmae_voxelize_workspace(n, batch_size, gx, gy, gz)
    ws = _ws(wsb, dev)
    r, vs = [float(v) for v in pc_range[:3]], [float(v) for v in voxel_size]
    check(lib.tmae_voxelize(_p(pts), row, n, batch_size, r[0], r[1], r[2], vs[0], vs[1], vs[2], gx, gy, gz,
                            _p(out['points']), _p(out['point_coords']), _p(out['inverse']),
                            _p(out['voxel_coords']), _p(out['counts']), _p(ws), wsb, _s()), 'tmae_voxelize')
    return out


_HOST_BUFS = {}


def to_host(t):
    """A small device tensor on the host: asynchronous copy into a pinned buffer, then a wait on an event recorded right behind
    it -- not Tensor.cpu(), whose pageable destination makes the runtime stage the copy and block the calling thread in a
    stream-wide wait.  The step's two data-dependent sizes (voxel counts, strided-conv output counts) come through here."""
    t = t.detach()
    n = t.numel()
    key = (t.device, t.dtype)
    ent = _HOST_BUFS.get(key)
    if ent is None or ent[0].numel() < n:
        ent = _HOST_BUFS[key] = (torch.empty((max(n, 256),), dtype=t.dtype).pin_memory(), torch.cuda.Event())
    buf = ent[0][:n].view(t.shape)
    buf.copy_(t, non_blocking=True)
    ent[1].record(torch.cuda.current_stream(t.device))
    ent[1].synchronize()
    return buf.clone()


class HostCopy:
    """to_host() in two halves: `HostCopy(t)` enqueues the asynchronous copy of a small device tensor into a pinned buffer of its own
    and records an event behind it; `.get()` waits for that event and returns the values.  Between the two the caller enqueues
    work that does not need the values: by the time they are read the copy has long happened, the host does not stall and keeps
    its lead over the GPU (the strided convs' output counts are enqueued before the first stage and read after it)."""
    _free = {}         # (device, dtype) -> [(pinned buffer, event), ...] not in flight

    def __init__(self, t, tag):
        """Every copy in flight owns its pinned buffer and event: they come from a free list and go back to it in get().  (Until
        round 6 one buffer per `tag` was shared: a second HostCopy of the same tag made before the first one's get() -- two models
        or forwards interleaved, an abandoned prefetch -- overwrote the first one's values in place.)  `tag` only names the copy."""
        t = t.detach()
        self.tag = tag
        self.key = (t.device, t.dtype)
        free = HostCopy._free.setdefault(self.key, [])
        ent = None
        for j, e in enumerate(free):
            if e[0].numel() >= t.numel():
                ent = free.pop(j)
                break
        if ent is None:
            ent = (torch.empty((max(t.numel(), 64),), dtype=t.dtype).pin_memory(), torch.cuda.Event())
        self.ent = ent
        self.buf = ent[0][:t.numel()].view(t.shape)
        self.buf.copy_(t, non_blocking=True)
        ent[1].record(torch.cuda.current_stream(t.device))

    def get(self):
        if self.ent is None:
            raise RuntimeError(f'HostCopy({self.tag!r}).get() called twice')
        self.ent[1].synchronize()
        out = self.buf.clone()
        HostCopy._free[self.key].append(self.ent)       # an abandoned copy simply never returns its buffer (64 pinned elements)
        self.ent = self.buf = None
        return out


def voxelize_finish(out, counts_host):
    n_kept, m = int(counts_host[0]), int(counts_host[1])
    return dict(points=out['points'][:n_kept], point_coords=out['point_coords'][:n_kept],
                inverse=out['inverse'][:n_kept], voxel_coords=out['voxel_coords'][:m],
                voxels_per_sample=[int(v) for v in counts_host[2:]])


def voxelize(points, batch_size, pc_range, voxel_size, grid_size):
    out = voxelize_launch(points, batch_size, pc_range, voxel_size, grid_size)
    return voxelize_finish(out, to_host(out['counts']))


def segment_csr(inverse, m):
    """Stable point->voxel CSR: perm [n] int32, offsets [m+1] int32."""
    _need_cuda(inverse)
    n = inverse.shape[0]
    perm = torch.empty((n,), dtype=torch.int32, device=inverse.device)
    offsets = torch.empty((m + 1,), dtype=torch.int32, device=inverse.device)
    wsb = lib.tmae_segment_csr_workspace(n, m)
    ws = _ws(wsb, inverse.device)
    check(lib.tmae_segment_csr(_p(inverse), n, m, _p(perm), _p(offsets), _p(ws), wsb, _s()), 'tmae_segment_csr')
    return perm, offsets


def get_inner_win_inds(group_inds):
    """sst_ops_utils.get_inner_win_inds (pcdet/ops/sst_ops/sst_ops_utils.py:5-12): running index of each
    element in its group; deterministic (stable rank), unlike the reference's atomic order."""
    _need_cuda(group_inds)
    g = group_inds.contiguous().long()
    n = g.shape[0]
    out = torch.empty_like(g)
    if n == 0:
        return out
    ng = int(g.max().item()) + 1          # the reference syncs here as well (sst_ops.cpp:26)
    wsb = lib.tmae_ingroup_rank_workspace(n, ng)
    ws = _ws(wsb, g.device)
    check(lib.tmae_ingroup_rank(_p(g), n, ng, _p(out), _p(ws), wsb, _s()), 'tmae_ingroup_rank')
    return out


def vfe_point_features(points, point_coords, inverse, perm, offsets, m, pc_range, voxel_size):
    n, row = points.shape
    mean = torch.empty((m, row - 1), dtype=torch.float32, device=points.device)
    feats = torch.empty((n, row + 5), dtype=torch.float32, device=points.device)
    r, vs = [float(v) for v in pc_range[:3]], [float(v) for v in voxel_size]
    check(lib.tmae_vfe_point_features(_p(points), row, _p(point_coords), _p(inverse), _p(perm), _p(offsets), n, m,
                                      r[0], r[1], r[2], vs[0], vs[1], vs[2], _p(mean), _p(feats), _s()),
          'tmae_vfe_point_features')
    return mean, feats


def vfe_point_features_bf16x2(points, point_coords, inverse, perm, offsets, m, pc_range, voxel_size, csr_order=False):
    """The point features as [n,32] bf16 = [hi(16) | lo(16)], hi + lo = the fp32 feature (see the header).
    csr_order: the rows come out sorted by voxel (row j = point perm[j]) and the third result is inverse_csr [n] (the voxel
    of every row): the segment max behind the row-wise point MLP then streams (scatter_max(x, inverse_csr, None, ...))."""
    n, row = points.shape
    mean = torch.empty((m, row - 1), dtype=torch.float32, device=points.device)
    feats = torch.empty((n, 32), dtype=torch.bfloat16, device=points.device)
    inv_csr = torch.empty((n,), dtype=torch.int64, device=points.device) if csr_order else None
    r, vs = [float(v) for v in pc_range[:3]], [float(v) for v in voxel_size]
    check(lib.tmae_vfe_point_features_bf16x2(_p(points), row, _p(point_coords), _p(inverse), _p(perm), _p(offsets), n, m,
                                             r[0], r[1], r[2], vs[0], vs[1], vs[2], _p(mean), _p(feats), _p(inv_csr), _s()),
          'tmae_vfe_point_features_bf16x2')
    return (mean, feats, inv_csr) if csr_order else (mean, feats)


class _LinearSplitInput(torch.autograd.Function):
    """y = f W^T for point features given as x2 = [hi | lo] (vfe_point_features_bf16x2): one bf16 GEMM over the 32
    columns against [W | W], i.e. the inputs enter with ~16 mantissa bits although the arithmetic is bf16 MFMA.
    dW = dW2[:, :16] + dW2[:, 16:] from the token-split kernel; the points take no gradient."""

    @staticmethod
    def forward(ctx, x2, weight):
        k = weight.shape[1]

        def dup(wt):                                    # [W | 0 | W | 0]: the same weight against the hi and the lo half
            w = torch.zeros((wt.shape[0], 32), dtype=torch.bfloat16, device=wt.device)
            wb = wt.detach().to(torch.bfloat16)
            w[:, :k] = wb
            w[:, 16:16 + k] = wb
            return w
        w = _derived(weight, ('hilo', k), dup)           # once per weight version (both frames of a step share it)
        ctx.save_for_backward(x2)
        ctx.k, ctx.wdtype = k, weight.dtype
        n = w.shape[0]
        if (x2.is_cuda and x2.dtype == torch.bfloat16 and x2.is_contiguous() and n % 64 == 0 and x2.shape[0] >= _TOKEN_GEMM_MIN_ROWS
                and x2.shape[0] * n * 2 < 2 ** 31):
            # contraction 32 on the token GEMM (csrc/token_gemm.hip): the library ran this 174 MB pass at 1.4 TB/s
            y = torch.empty((x2.shape[0], n), dtype=torch.bfloat16, device=x2.device)
            check(lib.tmae_token_gemm(_p(x2), 32, x2.shape[0], 32, _p(w), n, _p(_zero_bias(n, x2.device)), _p(y), n, _s()),
                  'tmae_token_gemm')
            return y
        return torch.nn.functional.linear(x2, w)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        dy = dy.to(torch.bfloat16)
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        if _wgrad_ok(dy, x2):
            dw2, _ = linear_wgrad(dy, x2, want_bias=False)
        else:
            dw2 = dy.float().t() @ x2.float()
        k = ctx.k
        return None, (dw2[:, :k] + dw2[:, 16:16 + k]).to(ctx.wdtype)


def linear_split_input(x2, weight):
    return _LinearSplitInput.apply(x2, weight)


class _SegmentMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inverse, perm, offsets, m):
        x = x.contiguous()
        n, c = x.shape
        out = torch.empty((m, c), dtype=x.dtype, device=x.device)
        argmax = torch.empty((m, c), dtype=torch.int32, device=x.device)
        check(lib.tmae_segment_max_fwd(_p(x), _dt(x), n, m, c, _p(perm), _p(offsets), _p(out), _p(argmax), _s()),
              'tmae_segment_max_fwd')
        ctx.save_for_backward(inverse, argmax)
        ctx.n = n
        ctx.mark_non_differentiable(argmax)
        ctx.set_materialize_grads(False)
        return out, argmax

    @staticmethod
    def backward(ctx, dout, _):
        if dout is None:
            return (None,) * 5
        inverse, argmax = ctx.saved_tensors
        dout = dout.contiguous()
        m, c = dout.shape
        dx = torch.empty((ctx.n, c), dtype=dout.dtype, device=dout.device)
        check(lib.tmae_segment_max_bwd(_p(dout), _dt(dout), ctx.n, m, c, _p(inverse), _p(argmax), _p(dx), _s()),
              'tmae_segment_max_bwd')
        return dx, None, None, None, None


class _BNReLUSegMax(torch.autograd.Function):
    """scatter_max(relu(BatchNorm1d(x))) -- the tail of the VFE's MLP (temporal_dyn_vfe.py:110-113) -- without the normalised
    [points, c] tensor: one statistics pass over x, then the segment max normalises the rows as it reads them
    (tmae_segment_max_bn_fwd: values and argmax bit-equal to the two-op form).  Backward: the max's gradient scattered to its
    argmax rows, then the norm's backward with the ReLU mask recomputed from x, as before."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, inverse, perm, offsets, m):
        x = x.contiguous()
        n, c = x.shape
        mean = torch.empty((c,), dtype=torch.float32, device=x.device)
        var, rstd = torch.empty_like(mean), torch.empty_like(mean)
        g32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        wsb = lib.tmae_bn_workspace(n, c)
        ws = _ws(wsb, x.device)
        check(lib.tmae_bn_stats(_p(x), _dt(x), n, c, float(n), float(eps), _p(mean), _p(var), _p(rstd), _p(ws), wsb, _s()), 'tmae_bn_stats')
        out = torch.empty((m, c), dtype=x.dtype, device=x.device)
        argmax = torch.empty((m, c), dtype=torch.int32, device=x.device)
        check(lib.tmae_segment_max_bn_fwd(_p(x), _dt(x), n, m, c, _p(perm), _p(offsets), _p(mean), _p(rstd), _p(g32), _p(b32), 1, _p(out),
                                          _p(argmax), _s()), 'tmae_segment_max_bn_fwd')
        ctx.save_for_backward(x, mean, rstd, g32, b32, inverse, argmax)
        ctx.dtypes = (weight.dtype, bias.dtype)
        ctx.mark_non_differentiable(argmax, mean, var)
        ctx.set_materialize_grads(False)
        return out, argmax, mean, var

    @staticmethod
    def backward(ctx, dout, _a, _m, _v):
        if dout is None:
            return (None,) * 8
        x, mean, rstd, g32, b32, inverse, argmax = ctx.saved_tensors
        n, c = x.shape
        dout = dout.to(x.dtype).contiguous()
        dy = torch.empty((n, c), dtype=x.dtype, device=x.device)
        check(lib.tmae_segment_max_bwd(_p(dout), _dt(dout), n, dout.shape[0], c, _p(inverse), _p(argmax), _p(dy), _s()), 'tmae_segment_max_bwd')
        dx = torch.empty_like(x)
        dg = torch.empty((c,), dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
        wsb = lib.tmae_bn_workspace(n, c)
        ws = _ws(wsb, x.device)
        check(lib.tmae_bn_relu_bwd(_p(dy), _p(x), _dt(x), n, c, _p(mean), _p(rstd), _p(g32), _p(b32), 1, _p(dx), _p(dg), _p(db), _p(ws), wsb,
                                   _s()), 'tmae_bn_relu_bwd')
        return dx, dg.to(ctx.dtypes[0]), db.to(ctx.dtypes[1]), None, None, None, None, None


def bn_relu_scatter_max(x, bn, inverse, perm, offsets, m):
    """scatter_max(relu(bn(x)), inverse) -> (out [m, c], argmax) with bn's running statistics updated as torch does; the fused
    form (_BNReLUSegMax) for training-mode norms over 64 / 128 / 256 channels on the GPU, else the two ops."""
    n, c = x.shape
    if (x.is_cuda and bn.training and c in (64, 128, 256) and n > 1 and _sync_group(bn) is None and x.dtype in (torch.float32, torch.bfloat16)
            and x.is_contiguous() and x.data_ptr() % 16 == 0):
        out, argmax, mean, var = _BNReLUSegMax.apply(x, bn.weight, bn.bias, bn.eps, inverse, perm, offsets, m)
        if bn.track_running_stats:
            _bn_running_update(bn, mean, var, float(n))
        return out, argmax
    return scatter_max(batch_norm_relu(x, bn, relu=True), inverse, perm, offsets, m)


def scatter_max(src, inverse, perm, offsets, m):
    """torch_scatter.scatter_max(src, index, dim=0) -> (out, argmax) (temporal_dyn_vfe.py:113)."""
    return _SegmentMax.apply(src, inverse, perm, offsets, m)


def group_points(points, voxel_coords, perm, offsets, k, pc_range, voxel_size, want_inds=True):
    """sst_ops_utils.group_inner_inds + centre normalisation (SiamWCA_MAE.py:134-141)."""
    m = voxel_coords.shape[0]
    dev = points.device
    ginds = torch.empty((m, k), dtype=torch.int64, device=dev) if want_inds else None
    gt = torch.empty((m, k, 3), dtype=torch.float32, device=dev)
    r, vs = [float(v) for v in pc_range[:3]], [float(v) for v in voxel_size]
    check(lib.tmae_group_points(_p(points), points.shape[1], _p(voxel_coords), _p(perm), _p(offsets), m, k, r[0], r[1], r[2],
                                vs[0], vs[1], vs[2], _p(ginds), _p(gt), _s()), 'tmae_group_points')
    return ginds, gt


# ----------------------------------------------------------------------------- masking (A3)

def random_mask(noise, sample_offsets, batch_size, keep_frac):
    """random_masking with injected noise (common_utils.py:49-63).  Returns mask [m] f32 (1 = removed),
    vis_index [m] int32 (first n_vis valid), n_vis (device int32 tensor)."""
    _need_cuda(noise)
    m = noise.shape[0]
    dev = noise.device
    mask = torch.empty((m,), dtype=torch.float32, device=dev)
    vis = torch.empty((m,), dtype=torch.int32, device=dev)
    n_vis = torch.zeros((1,), dtype=torch.int32, device=dev)
    wsb = lib.tmae_random_mask_workspace(m, batch_size)
    ws = _ws(wsb, dev)
    check(lib.tmae_random_mask(_p(noise.contiguous().float()), _p(sample_offsets), m, batch_size, float(keep_frac),
                               _p(mask), _p(vis), _p(n_vis), _p(ws), wsb, _s()), 'tmae_random_mask')
    return mask, vis, n_vis


# ----------------------------------------------------------------------------- windows (A4/A5/A10)

class _GatherRows(torch.autograd.Function):
    """x[idx] for DISTINCT row indices (the visible voxels of the masked frame, SiamWCA_MAE.py:166-182).  The backward writes each
    gradient row to its place in a zeroed tensor; autograd's own backward of x[idx] has to allow repeated indices and sorts them
    first (an accumulating index_put: 100 us for 94 k of 376 k rows, profiles/round6 census)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.rows = x.shape[0]
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        dx = dy.new_zeros((ctx.rows,) + tuple(dy.shape[1:]))
        dx.index_copy_(0, idx, dy)
        return dx, None


def gather_rows(x, idx):
    """x[idx], idx int64 [n] without repeats -- see _GatherRows."""
    return _GatherRows.apply(x, idx)


def index_grid(indices, batch_size, ny, nx):
    """Dense row-index grid [batch*ny*nx] int32 (-1 = inactive) of a sparse tensor's indices [m,3] (b,y,x)."""
    _need_cuda(indices)
    assert indices.dtype == torch.int32 and indices.is_contiguous()
    grid = torch.empty((batch_size * ny * nx,), dtype=torch.int32, device=indices.device)
    check(lib.tmae_index_grid(_p(indices), indices.shape[0], batch_size, ny, nx, _p(grid), _s()), 'tmae_index_grid')
    return grid


def window_bucket(indices, grid, grid_other, batch_size, ny, nx, window_shape, do_shift, drop_info, keep_only=False):
    """Window partition + region batching of one shift (spt_backbone.py:47-71,137-184; sst_utils.py:6-107;
    joint two-frame form SiamWCA.py:65-140 when grid_other is given).
    keep_only: only `keep` (and `inner`) -- what the cross-attention blocks and the token-dropping paths read; flat2win and the
    per-level window counts (one device scan per drop level) are then not computed."""
    import ctypes as C
    m = indices.shape[0]
    dev = indices.device
    wx, wy = int(window_shape[0]), int(window_shape[1])
    lv = sorted(drop_info.items())
    arr = (C.c_int32 * (3 * len(lv)))()
    for i, (_, info) in enumerate(lv):
        arr[3 * i], arr[3 * i + 1], arr[3 * i + 2] = info['max_tokens'], info['drop_range'][0], info['drop_range'][1]
    out = dict(inner=torch.empty((m,), dtype=torch.int32, device=dev), keep=torch.empty((m,), dtype=torch.uint8, device=dev))
    if not keep_only:
        out.update(
            batch_win_inds=torch.empty((m,), dtype=torch.int64, device=dev),
            coors_in_win=torch.empty((m, 3), dtype=torch.int64, device=dev),
            level=torch.empty((m,), dtype=torch.int32, device=dev),
            flat2win=torch.empty((m,), dtype=torch.int64, device=dev),
            win_per_level=torch.zeros((len(lv),), dtype=torch.int32, device=dev),
        )
    wsb = lib.tmae_window_bucket_workspace(batch_size, ny, nx, wy, wx, len(lv))
    ws = _ws(wsb, dev)
    check(lib.tmae_window_bucket(_p(indices), m, _p(grid), _p(grid_other), batch_size, ny, nx, wy, wx,
                                 1 if do_shift else 0, C.cast(arr, C.c_void_p), len(lv),
                                 _p(out.get('batch_win_inds')), _p(out.get('coors_in_win')), _p(out['inner']),
                                 _p(out.get('level')), _p(out['keep']), _p(out.get('flat2win')), _p(out.get('win_per_level')),
                                 _p(ws), wsb, _s()), 'tmae_window_bucket')
    return out


class _AddPos(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, indices, table, wy, wx, do_shift):
        x = x.contiguous()
        out = torch.empty_like(x)
        check(lib.tmae_add_pos_embed(_p(x), _dt(x), x.shape[0], x.shape[1], _p(indices), wy, wx,
                                     1 if do_shift else 0, _p(table), _p(out), _s()), 'tmae_add_pos_embed')
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None, None, None


def add_pos_embed(x, indices, table, window_shape, do_shift):
    """x + pos_embed(coors_in_win) (spt_backbone.py:186-224 + sst_basic_block.py:41-44)."""
    return _AddPos.apply(x, indices, table, int(window_shape[1]), int(window_shape[0]), bool(do_shift))


# ----------------------------------------------------------------------------- attention (A6/A7)

def window_worklist(grid_q, grid_k, batch, ny, nx, do_shift):
    """Index lists of the windows holding both queries and keys, binned by 16-token tile count (the reference's
    drop levels 16/32/64 as lists, not padded tensors).  Built once per (index set, shift), reused by every layer."""
    n = lib.tmae_window_worklist_size(batch, ny, nx)
    wl = torch.empty((n,), dtype=torch.int32, device=grid_q.device)
    check(lib.tmae_window_worklist(_p(grid_q), _p(grid_k), batch, ny, nx, 1 if do_shift else 0, _p(wl), _s()),
          'tmae_window_worklist')
    return wl


class _WinAttn(torch.autograd.Function):
    """Ragged window cosine attention over projected q/k/v.
    Self mode (c is None): a = packed [m,2d] (q|k) from one GEMM over x+pos, b = v [m,d]; b None too: a = [m,3d] (q|k|v).
    Cross mode: a = q [mq,d], b = k [mk,d], c = v [mk,d]; c == 'kv': b = packed [mk,2d] (k|v)."""

    @staticmethod
    def _ptrs(a, b, c, d):
        es = a.element_size()
        if b is None:
            return (a.data_ptr(), 3 * d, a.data_ptr() + d * es, 3 * d, a.data_ptr() + 2 * d * es, 3 * d)
        if c is None:
            return (a.data_ptr(), 2 * d, a.data_ptr() + d * es, 2 * d, b.data_ptr(), d)
        if isinstance(c, str):
            return (a.data_ptr(), d, b.data_ptr(), 2 * d, b.data_ptr() + d * es, 2 * d)
        return (a.data_ptr(), d, b.data_ptr(), d, c.data_ptr(), d)

    @staticmethod
    def forward(ctx, a, b, c, tau, grid_q, grid_k, worklist, nhead, batch, ny, nx, do_shift, tau_min, covered=False):
        a = a.contiguous()
        ctx.layout = 'qkv' if b is None else 'qk_v' if c is None else 'q_kv' if isinstance(c, str) else 'q_k_v'
        if b is None:
            d = a.shape[1] // 3
            mq = mk = a.shape[0]
        elif c is None:
            b = b.contiguous()
            d = b.shape[1]
            mq = mk = a.shape[0]
        else:
            b = b.contiguous()
            if not isinstance(c, str):
                c = c.contiguous()
                assert c.dtype == a.dtype
            d = a.shape[1]
            mq, mk = a.shape[0], b.shape[0]
        assert b is None or a.dtype == b.dtype
        dh = d // nhead
        q, ldq, k, ldk, v, ldv = _WinAttn._ptrs(a, b, c, d)
        cross = ctx.layout in ('q_kv', 'q_k_v')
        tau32 = tau.detach().reshape(-1).float().contiguous()
        per_head = tau32.numel() > 1                     # non_shared_tau: tau [1, nhead, 1, 1] (cosine_msa.py:453-454)
        assert tau32.numel() == (nhead if per_head else 1), (tuple(tau.shape), nhead)
        if a.dtype != torch.bfloat16:
            worklist = None                                  # the fp32 kernels walk the dense windows
        # cross mode starts from zeros: with a work list, tokens of windows in no list are not written; and under token
        # dropping (modules/sst.py: per-shift grids with the dropped tokens masked out) a token may be in no window at all
        # covered (the caller's promise: no token dropping, every token lies in a window of the two grids) + a work list: only
        # the rows of the windows that are in no list are zeroed (tmae_win_attn_zero_orphans), not the whole tensors
        orphans = cross and bool(covered) and worklist is not None and d % 8 == 0
        alloc = torch.zeros if (cross and not orphans) else torch.empty
        out = alloc((mq, d), dtype=a.dtype, device=a.device)
        lse = alloc((mq, nhead), dtype=torch.float32, device=a.device)
        if orphans:
            check(lib.tmae_win_attn_zero_orphans(_p(grid_q), _p(grid_k), batch, ny, nx, 1 if do_shift else 0, _p(out), d, d, _p(lse),
                                                 nhead, None, 0, None, 0, 0, _s()), 'tmae_win_attn_zero_orphans')
        check(lib.tmae_win_attn_fwd(q, ldq, k, ldk, v, ldv, _dt(a), mq, mk, nhead, dh, _p(grid_q), _p(grid_k),
                                    batch, ny, nx, 1 if do_shift else 0, _p(tau32), float(tau_min), _p(out), d,
                                    _p(lse), _p(worklist), 1 if per_head else 0, _s()), 'tmae_win_attn_fwd')
        ctx.cross = cross
        ctx.orphans = orphans
        ctx.has_wl = worklist is not None
        ctx.save_for_backward(a, b if b is not None else a, c if torch.is_tensor(c) else a, tau32, grid_q, grid_k, out,
                              lse, worklist if worklist is not None else grid_q)
        ctx.meta = (d, nhead, dh, batch, ny, nx, do_shift, tau_min, mq, mk, tau.shape, tau.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, c, tau32, grid_q, grid_k, out, lse, worklist = ctx.saved_tensors
        d, nhead, dh, batch, ny, nx, do_shift, tau_min, mq, mk, tshape, tdtype = ctx.meta
        lay = ctx.layout
        if lay == 'qkv':
            b = None
        if lay in ('qkv', 'qk_v'):
            c = None
        elif lay == 'q_kv':
            c = 'kv'
        if not ctx.has_wl:
            worklist = None
        dout = dout.contiguous()
        alloc = torch.zeros_like if (ctx.cross and not ctx.orphans) else torch.empty_like
        da = alloc(a)
        db = alloc(b) if b is not None else None
        dc = alloc(c) if torch.is_tensor(c) else c
        q, ldq, k, ldk, v, ldv = _WinAttn._ptrs(a, b, c, d)
        dq, lddq, dk, lddk, dv, lddv = _WinAttn._ptrs(da, db, dc, d)
        if ctx.orphans:
            check(lib.tmae_win_attn_zero_orphans(_p(grid_q), _p(grid_k), batch, ny, nx, 1 if do_shift else 0, dq, lddq, d, None, 0,
                                                 dk, lddk, dv, lddv, d, _s()), 'tmae_win_attn_zero_orphans')
        nblk = lib.tmae_win_attn_num_blocks(batch, ny, nx, nhead, dh)
        part = torch.empty((nblk,), dtype=torch.float32, device=a.device)       # unlisted windows: skipped by tmae_win_attn_dtau
        check(lib.tmae_win_attn_bwd(q, ldq, k, ldk, v, ldv, _p(out), d, _p(dout), d, _p(lse), _dt(a), mq, mk,
                                    nhead, dh, _p(grid_q), _p(grid_k), batch, ny, nx, 1 if do_shift else 0,
                                    _p(tau32), float(tau_min), dq, lddq, dk, lddk, dv, lddv, _p(part), _p(worklist),
                                    1 if tau32.numel() > 1 else 0, _s()), 'tmae_win_attn_bwd')
        # d/d tau of logits = cos / max(tau, tau_min): -(1/tau_c) * sum dS*s, zero in the clamped branch
        dtau = torch.empty((tau32.numel(),), dtype=torch.float32, device=a.device)
        # fixed-order multi-block sum of the partials + the clamp rule, one launch (one per head with per-head temperatures)
        check(lib.tmae_win_attn_dtau(_p(part), nblk, _p(tau32), float(tau_min), _p(dtau), _p(worklist), nhead,
                                     1 if tau32.numel() > 1 else 0, _s()), 'tmae_win_attn_dtau')
        dtau = dtau.reshape(tshape).to(tdtype)
        return da, db, (dc if torch.is_tensor(dc) else None), dtau, None, None, None, None, None, None, None, None, None, None


def win_attn(a, b, c, tau, grid_q, grid_k, nhead, batch, ny, nx, do_shift, tau_min, worklist=None, covered=False):
    """covered: see _WinAttn.forward (cross attention with a work list: zero only the orphan rows)."""
    return _WinAttn.apply(a, b, c, tau, grid_q, grid_k, worklist, nhead, batch, ny, nx, do_shift, tau_min, bool(covered))


# ----------------------------------------------------------------------------- sparse conv (A9)

def spconv_down_outputs(grid_in, batch, ny, nx):
    """Active output sites of SparseConv2d(k3,s2,p1): (out_grid, out_indices[worst case], n_out dev, (oy,ox))."""
    oy, ox = (ny + 2 - 3) // 2 + 1, (nx + 2 - 3) // 2 + 1
    dev = grid_in.device
    out_grid = torch.empty((batch * oy * ox,), dtype=torch.int32, device=dev)
    out_ind = torch.empty((batch * oy * ox, 3), dtype=torch.int32, device=dev)
    n_out = torch.zeros((1,), dtype=torch.int32, device=dev)
    wsb = lib.tmae_spconv_down_outputs_workspace(batch, oy, ox)
    ws = _ws(wsb, dev)
    check(lib.tmae_spconv_down_outputs(_p(grid_in), batch, ny, nx, oy, ox, _p(out_grid), _p(out_ind), _p(n_out),
                                       _p(ws), wsb, _s()), 'tmae_spconv_down_outputs')
    return out_grid, out_ind, n_out, (oy, ox)


def spconv_neighbors(out_indices, grid_in, batch, ny, nx, stride):
    m = out_indices.shape[0]
    nbr = torch.empty((m, 9), dtype=torch.int32, device=out_indices.device)
    check(lib.tmae_spconv_neighbors(_p(out_indices), m, _p(grid_in), batch, ny, nx, stride, _p(nbr), _s()),
          'tmae_spconv_neighbors')
    return nbr


def spconv_neighbors_t(in_indices, grid_out, batch, oy, ox, stride):
    m = in_indices.shape[0]
    nbr_t = torch.empty((m, 9), dtype=torch.int32, device=in_indices.device)
    check(lib.tmae_spconv_neighbors_t(_p(in_indices), m, _p(grid_out), batch, oy, ox, stride, _p(nbr_t), _s()),
          'tmae_spconv_neighbors_t')
    return nbr_t


def _gather9(feat, nbr):
    m_out = nbr.shape[0]
    c = feat.shape[1]
    cols = torch.empty((m_out, 9 * c), dtype=feat.dtype, device=feat.device)
    check(lib.tmae_spconv_gather(_p(feat), _dt(feat), feat.shape[0], c, _p(nbr), m_out, _p(cols), _s()),
          'tmae_spconv_gather')
    return cols


def _spconv_native_ok(f, cin, cout):
    return (_SPCONV_NATIVE and f.dtype == torch.bfloat16 and cin in (128, 256) and cout in (128, 256)
            and f.stride(1) == 1 and f.stride(0) % 8 == 0 and f.data_ptr() % 16 == 0)


import os as _os
_SPCONV_NATIVE = _os.environ.get('TMAE_SPCONV', 'native') != 'gather'


def spconv_fwd(feat, nbr, w2d):
    """out [m_out, cout] = implicit-GEMM sparse conv (csrc/spconv_igemm.hip); w2d [cout, 9*cin] bf16."""
    m_out, cout, cin = nbr.shape[0], w2d.shape[0], feat.shape[1]
    out = torch.empty((m_out, cout), dtype=torch.bfloat16, device=feat.device)
    check(lib.tmae_spconv_fwd(_p(feat), feat.stride(0), feat.shape[0], cin, _p(nbr), m_out, _p(w2d), cout, _p(out), cout,
                              _s()), 'tmae_spconv_fwd')
    return out


def spconv_bwd_data(dout, nbr_t, w2d, cin):
    """din [m_in, cin] through the transposed rulebook; w2d [cout, 9*cin] (transposed copy made here, ~1 MB)."""
    cout = w2d.shape[0]
    m_in = nbr_t.shape[0]
    wt = w2d.view(cout, 9, cin).permute(2, 1, 0).contiguous().view(cin, 9 * cout)
    din = torch.empty((m_in, cin), dtype=torch.bfloat16, device=dout.device)
    check(lib.tmae_spconv_bwd_data(_p(dout), dout.stride(0), dout.shape[0], cout, _p(nbr_t), m_in, _p(wt), cin, _p(din),
                                   cin, _s()), 'tmae_spconv_bwd_data')
    return din


class _SparseConv(torch.autograd.Function):
    """out[o] = sum_t W[:,t,:] in[nbr[o,t]]; weight [cout,3,3,cin] (spconv-2 layout).  bf16 with 128 / 256 channels: the
    native implicit GEMM (forward, input gradient) and the rulebook-reading token-split kernel (weight gradient) -- the
    [m, 9 cin] im2col matrix never exists.  Other shapes / fp32: gather + one GEMM."""

    @staticmethod
    def forward(ctx, feat, weight, nbr, nbr_t):
        cdt = compute_dtype(feat)
        f = feat.to(cdt).contiguous()
        w = cast_param(weight, cdt).reshape(weight.shape[0], -1)
        ctx.native = _spconv_native_ok(f, f.shape[1], w.shape[0]) and f.shape[0] > 0
        if ctx.native:
            out = spconv_fwd(f, nbr, w.contiguous())
        else:
            out = _gather9(f, nbr) @ w.t()
        ctx.save_for_backward(f, w, nbr, nbr_t)
        ctx.wshape, ctx.wdtype, ctx.fdtype = weight.shape, weight.dtype, feat.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        f, w, nbr, nbr_t = ctx.saved_tensors
        dout = dout.to(f.dtype).contiguous()
        cin, cout = f.shape[1], dout.shape[1]
        if ctx.native and dout.data_ptr() % 16 == 0:
            din = spconv_bwd_data(dout, nbr_t, w.contiguous(), cin)
        else:
            dcols = token_gemm_dx(dout, w)                                # [m_out, 9*cin]: streaming kernel when it fits
            din = torch.empty_like(f)
            check(lib.tmae_spconv_gather_t(_p(dcols), _dt(dcols), dout.shape[0], f.shape[1], _p(nbr_t), f.shape[0],
                                           _p(din), _s()), 'tmae_spconv_gather_t')
            del dcols
        if dout.dtype == torch.bfloat16 and cin % 128 == 0 and cout % 8 == 0 and dout.shape[0] >= 4096:
            # token-split kernel reading the feature rows through the rulebook: no [m_out, 9*cin] matrix
            dw = torch.empty((cout, 9 * cin), dtype=torch.float32, device=f.device)
            wsb = lib.tmae_linear_wgrad_workspace(dout.shape[0], cout, 9 * cin)
            ws = _ws(wsb, f.device)
            check(lib.tmae_spconv_wgrad(_p(dout), dout.stride(0), _p(f), f.stride(0), _p(nbr), dout.shape[0], cout, cin,
                                        _p(dw), _p(ws), wsb, _s()), 'tmae_spconv_wgrad')
        else:
            dw = dout.t() @ _gather9(f, nbr)
        dw = dw.reshape(ctx.wshape).to(ctx.wdtype)
        return din.to(ctx.fdtype), dw, None, None


def sparse_conv(feat, weight, nbr, nbr_t):
    return _SparseConv.apply(feat, weight, nbr, nbr_t)


# ----------------------------------------------------------------------------- dense <-> sparse (A11/A12)

class _ToDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, grid, indices, batch, ny, nx):
        feat = feat.contiguous()
        c = feat.shape[1]
        out = torch.empty((batch, ny, nx, c), dtype=feat.dtype, device=feat.device)
        check(lib.tmae_sparse_to_dense(_p(feat), _dt(feat), feat.shape[0], c, _p(grid), batch, ny, nx, _p(out),
                                       _s()), 'tmae_sparse_to_dense')
        ctx.save_for_backward(indices)
        ctx.meta = (batch, ny, nx, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        (indices,) = ctx.saved_tensors
        batch, ny, nx, c = ctx.meta
        dout = dout.contiguous()
        rows = torch.empty((indices.shape[0], c), dtype=dout.dtype, device=dout.device)
        check(lib.tmae_dense_gather(_p(dout), _dt(dout), batch, ny, nx, c, _p(indices), indices.shape[0], _p(rows),
                                    _s()), 'tmae_dense_gather')
        return rows, None, None, None, None, None


def sparse_to_dense(feat, grid, indices, batch, ny, nx):
    """SparseConvTensor.dense() in channels-last: [batch, ny, nx, c] (SiamWCA_MAE.py:235)."""
    return _ToDense.apply(feat, grid, indices, batch, ny, nx)



class _DeblocksToDense(torch.autograd.Function):
    """dense() -> ConvTranspose2d(k=s, stride s) -> BatchNorm2d (batch stats) -> ReLU -> cat(dim=1) for several
    sparse sources at once, written straight into the channels-last concat buffer (csrc/deblock.hip).
    args = (feat_i, weight_i [cin, cout, s, s], gamma_i, beta_i) per source."""

    @staticmethod
    def forward(ctx, metas, batch, ny, nx, eps, pg, *args):
        n_src = len(metas)
        feats = args[0::4]
        dev = feats[0].device
        cdt = compute_dtype(feats[0])
        couts = [args[4 * i + 1].shape[1] for i in range(n_src)]
        ctot = sum(couts)
        cat = torch.empty((batch, ny, nx, ctot), dtype=cdt, device=dev)
        count = float(batch * ny * nx)
        saved, stats = [], []
        coff = 0
        for i, (grid, indices, ys, xs, s) in enumerate(metas):
            feat, w, gamma, beta = args[4 * i:4 * i + 4]
            cin, cout = w.shape[0], w.shape[1]
            assert ys * s == ny and xs * s == nx and w.shape[2] == s and w.shape[3] == s
            x_c = feat.to(cdt).contiguous()
            wmat = w.detach().to(cdt).permute(2, 3, 1, 0).reshape(s * s * cout, cin).contiguous()
            v = token_gemm(x_c, wmat)                                       # [m, s*s*cout], columns (dy, dx, cout)
            m = v.shape[0]
            mean = torch.empty((cout,), dtype=torch.float32, device=dev)
            var, rstd = torch.empty_like(mean), torch.empty_like(mean)
            g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
            rows = m * s * s
            wsb = lib.tmae_bn_workspace(rows, cout)
            ws = _ws(wsb, dev)
            check(lib.tmae_bn_stats(_p(v), _dt(v), rows, cout, count, float(eps[i]), _p(mean), _p(var), _p(rstd),
                                    _p(ws), wsb, _s()), 'tmae_bn_stats')
            if pg is not None:                    # SyncBatchNorm: every rank counts batch * ny * nx cells
                gm, gv, _ = _merge_stats(mean, var, count, pg)
                mean.copy_(gm), var.copy_(gv), rstd.copy_(torch.rsqrt(gv + float(eps[i])))
            saved += [x_c, wmat, v, mean, rstd, g32, b32, grid, indices]
            stats += [mean, var]
            coff += cout
        if n_src <= 4 and len({(ys * s, xs * s) for (_, _, ys, xs, s) in metas}) == 1 and all(s & (s - 1) == 0 for (_, _, _, _, s) in metas):
            # all channel slices of the concat buffer in one launch (whole rows per store burst): HOST tables of the sources
            import ctypes as _ct
            vp = _ct.c_void_p
            def tab(vals, ty):
                return (ty * n_src)(*vals)
            ptrs = lambda k: _ct.cast(tab([saved[9 * i + k].data_ptr() for i in range(n_src)], vp), vp)
            ints = lambda k: _ct.cast(tab([int(metas[i][k]) for i in range(n_src)], _ct.c_int), vp)
            t_v, t_mean, t_rstd, t_g, t_b, t_grid = ptrs(2), ptrs(3), ptrs(4), ptrs(5), ptrs(6), ptrs(7)
            t_ys, t_xs, t_s = ints(2), ints(3), ints(4)
            t_c = _ct.cast(tab(couts, _ct.c_int), vp)
            check(lib.tmae_deblock_scatter_multi(n_src, t_v, _dt(saved[2]), t_grid, batch, t_ys, t_xs, t_s, t_c, t_mean, t_rstd, t_g, t_b,
                                                 _p(cat), ctot, _s()), 'tmae_deblock_scatter_multi')
        else:
            coff = 0
            for i, (grid, indices, ys, xs, s) in enumerate(metas):
                x_c, wmat, v, mean, rstd, g32, b32 = saved[9 * i:9 * i + 7]
                check(lib.tmae_deblock_scatter(_p(v), _dt(v), _p(grid), batch, ys, xs, s, couts[i], _p(mean), _p(rstd), _p(g32),
                                               _p(b32), _p(cat), ctot, coff, _s()), 'tmae_deblock_scatter')
                coff += couts[i]
        ctx.save_for_backward(*saved)
        ctx.meta = (n_src, [(ys, xs, s) for (_, _, ys, xs, s) in metas], batch, ny, nx, couts, count,
                    [(args[4 * i].dtype, args[4 * i + 1].dtype, args[4 * i + 2].dtype, args[4 * i + 3].dtype)
                     for i in range(n_src)])
        ctx.pg = pg
        ctx.mark_non_differentiable(*stats)
        ctx.set_materialize_grads(False)
        return (cat, *stats)

    @staticmethod
    def backward(ctx, dcat, *_unused):
        n_src, shapes, batch, ny, nx, couts, count, dts = ctx.meta
        if dcat is None:
            return (None,) * (6 + 4 * n_src)
        saved = ctx.saved_tensors
        ctot = sum(couts)
        dev = dcat.device
        cdt = saved[2].dtype
        dcat = dcat.to(cdt)
        if not dcat.is_contiguous():
            dcat = dcat.contiguous()
        nrows = batch * ny * nx
        s_all = colsum_tail(dcat, ctot)                      # the producing conv kernel summed its own output (dense_conv3x3_halo)
        if s_all is None:
            s_all = torch.empty((ctot,), dtype=torch.float32, device=dev)
            wsb = lib.tmae_column_sums_workspace(nrows, ctot)
            ws = _ws(wsb, dev)
            check(lib.tmae_column_sums(_p(dcat), _dt(dcat), nrows, ctot, _p(s_all), _p(ws), wsb, _s()), 'tmae_column_sums')
        grads = []
        coff = 0
        for i in range(n_src):
            x_c, wmat, v, mean, rstd, g32, b32, grid, indices = saved[9 * i:9 * i + 9]
            ys, xs, s = shapes[i]
            cout = couts[i]
            m = v.shape[0]
            rows = m * s * s
            if ctx.pg is None and s in (1, 2, 4):
                # the norm's whole backward with the gradient rows read in place from dcat (no gathered copy: 353 MB for the
                # stride-4 source), three sums in one pass, the inactive cells' share, one pass for dv
                dv = torch.empty_like(v)
                dbeta = torch.empty((cout,), dtype=torch.float32, device=dev)
                dgamma = torch.empty_like(dbeta)
                wsb = lib.tmae_deblock_bn_bwd_workspace(m, s, cout)
                ws = _ws(wsb, dev)
                check(lib.tmae_deblock_bn_bwd(_p(dcat), _dt(dcat), ctot, coff, _p(indices), m, ys, xs, s, cout, _p(v), _p(mean), _p(rstd),
                                              _p(g32), _p(b32), s_all.data_ptr() + 4 * coff, float(count), _p(dv), _p(dgamma), _p(dbeta),
                                              _p(ws), wsb, _s()), 'tmae_deblock_bn_bwd')
                dfeat = token_gemm_dx(dv, wmat) if ctx.needs_input_grad[6 + 4 * i] else None
                dwmat = None
                if ctx.needs_input_grad[6 + 4 * i + 1]:
                    if _wgrad_ok(dv, x_c):
                        dwmat, _ = linear_wgrad(dv, x_c, want_bias=False)
                    else:
                        dwmat = dv.float().t() @ x_c.float()
                    cin = x_c.shape[1]
                    dwmat = dwmat.view(s, s, cout, cin).permute(3, 2, 0, 1).to(dts[i][1])
                grads += [None if dfeat is None else dfeat.to(dts[i][0]), dwmat, dgamma.to(dts[i][2]), dbeta.to(dts[i][3])]
                coff += cout
                continue
            g = torch.empty_like(v)
            check(lib.tmae_deblock_gather(_p(dcat), _dt(dcat), ctot, coff, _p(indices), m, ys, xs, s, cout, _p(g), _s()),
                  'tmae_deblock_gather')
            # the unmasked column sums of g (for the inactive cells' share below) and the two masked sums: one pass over g and v
            s_act = torch.empty((cout,), dtype=torch.float32, device=dev)
            sum_dz = torch.empty((cout,), dtype=torch.float32, device=dev)
            sum_dzx = torch.empty_like(sum_dz)
            wsb = 2 * lib.tmae_bn_workspace(rows, cout)
            ws = _ws(wsb, dev)
            check(lib.tmae_bn_bwd_sums3(_p(g), _p(v), _dt(v), rows, cout, _p(mean), _p(rstd), _p(g32), _p(b32), 1,
                                        _p(s_act), _p(sum_dz), _p(sum_dzx), _p(ws), wsb, _s()), 'tmae_bn_bwd_sums3')
            # inactive cells: z = beta - mean*rstd*gamma (constant per channel), xhat = -mean*rstd: their share of the sums
            dbeta, dgamma = torch.empty_like(sum_dz), torch.empty_like(sum_dz)
            check(lib.tmae_deblock_bn_tail(_p(mean), _p(rstd), _p(g32), _p(b32), s_all.data_ptr() + 4 * coff, _p(s_act),
                                           _p(sum_dz), _p(sum_dzx), cout, _p(dbeta), _p(dgamma), _s()), 'tmae_deblock_bn_tail')
            tb, tg, tcount = dbeta, dgamma, count
            if ctx.pg is not None:                # the sums of every rank enter dx; dgamma / dbeta stay this rank's
                import torch.distributed as dist
                tot = torch.stack([dbeta, dgamma])
                dist.all_reduce(tot, group=ctx.pg)
                tb, tg, tcount = tot[0].contiguous(), tot[1].contiguous(), count * dist.get_world_size(ctx.pg)
            dv = torch.empty_like(v)
            check(lib.tmae_bn_bwd_apply(_p(g), _p(v), _dt(v), rows, cout, _p(mean), _p(rstd), _p(g32), _p(b32), 1,
                                        _p(tb), _p(tg), tcount, _p(dv), _s()), 'tmae_bn_bwd_apply')
            dfeat = token_gemm_dx(dv, wmat) if ctx.needs_input_grad[6 + 4 * i] else None
            dwmat = None
            if ctx.needs_input_grad[6 + 4 * i + 1]:
                if _wgrad_ok(dv, x_c):
                    dwmat, _ = linear_wgrad(dv, x_c, want_bias=False)
                else:
                    dwmat = dv.float().t() @ x_c.float()
                cin = x_c.shape[1]
                dwmat = dwmat.view(s, s, cout, cin).permute(3, 2, 0, 1).to(dts[i][1])
            grads += [None if dfeat is None else dfeat.to(dts[i][0]), dwmat, dgamma.to(dts[i][2]), dbeta.to(dts[i][3])]
            coff += cout
        return (None, None, None, None, None, None, *grads)


def deblocks_to_dense(sources, batch, ny, nx):
    """sources: list of (SparseConvTensor-like features [m,cin], grid, indices, (ys, xs), ConvTranspose2d, BatchNorm2d)
    with kernel == stride and no bias.  Returns the channels-last concat [batch, ny, nx, sum(cout)] of
    relu(bn(deconv(dense(source)))) -- SiamWCA_MAE.dense_conv (SiamWCA_MAE.py:231-250) -- and updates the running
    statistics of the norms like torch does."""
    metas, args, eps = [], [], []
    for feat, grid, indices, (ys, xs), deconv, bn in sources:
        s = int(deconv.stride[0])
        metas.append((grid, indices, int(ys), int(xs), s))
        args += [feat, deconv.weight, bn.weight, bn.bias]
        eps.append(bn.eps)
    pgs = {id(pg): pg for pg in (_sync_group(src[5]) for src in sources)}
    if len(pgs) > 1:
        raise NotImplementedError('decoder norms on different SyncBatchNorm process groups')
    pg = next(iter(pgs.values()))
    out = _DeblocksToDense.apply(metas, batch, ny, nx, eps, pg, *args)
    cat, stats = out[0], out[1:]
    n = float(batch * ny * nx)
    if pg is not None:
        import torch.distributed as dist
        n *= dist.get_world_size(pg)
    for i, src in enumerate(sources):
        bn = src[5]
        if bn.track_running_stats:
            _bn_running_update(bn, stats[2 * i].detach(), stats[2 * i + 1].detach(), n)
    return cat


def deblocks_fusable(sources, training):
    for feat, grid, indices, shape, deconv, bn in sources:
        if not (isinstance(deconv, torch.nn.ConvTranspose2d) and isinstance(bn, (torch.nn.BatchNorm2d, torch.nn.SyncBatchNorm))):
            return False
        s = deconv.stride[0]
        if (deconv.kernel_size != (s, s) or deconv.stride != (s, s) or deconv.bias is not None
                or deconv.padding != (0, 0) or deconv.output_padding != (0, 0) or deconv.groups != 1):
            return False
        if not (training and bn.training and bn.affine and deconv.out_channels in (64, 128, 256) and feat.is_cuda
                and feat.shape[0] > 0):
            return False
    return True


_DENSE_NBR = {}


def _dense_rulebook(batch, ny, nx, device, dil=1):
    """nbr [batch*ny*nx, 9] of a FULL grid (row = cell, -1 past the border): the sparse-conv rulebook of a dense 3x3
    convolution with padding = dilation, built once per shape."""
    key = (batch, ny, nx, device) if dil == 1 else (batch, ny, nx, device, dil)
    nbr = _DENSE_NBR.get(key)
    if nbr is None:
        n = batch * ny * nx
        cells = torch.arange(n, device=device, dtype=torch.int32)
        if dil == 1:
            ind = torch.stack([cells // (ny * nx), (cells // nx) % ny, cells % nx], 1).contiguous()
            nbr = spconv_neighbors(ind, cells, batch, ny, nx, 1)
        else:
            y, x = (cells // nx) % ny, cells % nx
            cols = []
            for ky in range(3):
                for kx in range(3):
                    yy, xx = y + (ky - 1) * dil, x + (kx - 1) * dil
                    ok = (yy >= 0) & (yy < ny) & (xx >= 0) & (xx < nx)
                    cols.append(torch.where(ok, cells + ((ky - 1) * nx + (kx - 1)) * dil, torch.full_like(cells, -1)))
            nbr = torch.stack(cols, 1).contiguous()
        _DENSE_NBR[key] = nbr
    return nbr


_DENSE_WGRAD = _os.environ.get('TMAE_DENSE_WGRAD', 'halo')   # halo: csrc/dense_wgrad.hip; rulebook: tmae_spconv_wgrad over the full grid
_DENSE_CONV = _os.environ.get('TMAE_DENSE_CONV', 'halo')     # native: all three passes on our kernels; wgrad: only dW; miopen
_DENSE_SUMS = _os.environ.get('TMAE_DENSE_SUMS', '1') != '0'    # column sums / moments out of the decoder conv's epilogues (round 6)


class _DenseConv3x3(torch.autograd.Function):
    """Conv2d(cin, cout, 3, padding=1, bias=False) on a channels-last activation [B, Y, X, cin] (SiamWCA_MAE.py:100-115)
    as a sparse conv over the rulebook of a FULL grid: forward and input gradient on the implicit-GEMM kernel of
    csrc/spconv_igemm.hip, weight gradient on the token-split kernel of csrc/wgrad.hip (dW[cout, 9 cin] = dY^T .
    im2col(X), never materialised).  The library's implicit GEMMs ran at 0.52-0.68 PFLOP/s on this shape."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, dil=1, fork=False, moments=False):
        """fork: also returns an alias of the input for a residual shortcut (SSTBEVBackbone, sst_bev_backbone.py:35-41); the
        gradient that arrives for the alias is added to the input gradient INSIDE the conv's input-gradient kernel
        (tmae_dense_conv3x3_add) instead of by autograd's accumulation pass.
        moments: the last output is [2, cout] f32, the column sums of y and y^2 from the conv's epilogue (None-valued zeros tensor
        of shape [0] where the shape has no such kernel): what the BatchNorm behind the conv needs (ops.batch_norm_relu_gather)."""
        cdt = compute_dtype(x_nhwc)
        x = x_nhwc.to(cdt).contiguous()
        w = cast_param(weight, cdt)
        B, Y, X, cin = x.shape
        cout = w.shape[0]
        n = B * Y * X
        ctx.native = _DENSE_CONV in ('native', 'halo') and cin in (128, 256, 384) and cout % 128 == 0
        ctx.dil = dil
        if ctx.native:
            w2d = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
            mom = None
            if _DENSE_CONV == 'halo' or dil != 1:
                if moments and dil == 1:
                    y, mom = dense_conv3x3_halo(x, w2d, dil, moments=2)
                else:
                    y = dense_conv3x3_halo(x, w2d, dil)
            else:
                y = spconv_fwd(x.view(n, cin), _dense_rulebook(B, Y, X, x.device), w2d).view(B, Y, X, cout)
            ctx.save_for_backward(x, w2d)
        else:
            y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.contiguous(memory_format=torch.channels_last),
                                           padding=dil, dilation=dil).permute(0, 2, 3, 1)
            ctx.save_for_backward(x, w)
        ctx.meta = (x_nhwc.dtype, weight.dtype)
        ctx.fork = bool(fork)
        if moments:
            assert not fork
            if not ctx.native or mom is None:
                mom = torch.empty((0,), dtype=torch.float32, device=x.device)
            ctx.mark_non_differentiable(mom)
            ctx.set_materialize_grads(False)
            return y, mom
        if fork:
            ctx.set_materialize_grads(False)
            return y, x_nhwc.view_as(x_nhwc)
        return y

    @staticmethod
    def backward(ctx, dy_nhwc, dalias=None):
        x, w = ctx.saved_tensors
        if not ctx.fork:
            dalias = None                                    # (the second output is the moments tensor: no gradient)
        if dy_nhwc is None:                                  # only the alias was used downstream
            return (None if dalias is None else dalias.to(ctx.meta[0])), None, None, None, None
        skip = None if dalias is None else dalias.to(x.dtype).contiguous()
        B, Y, X, cin = x.shape
        cout = w.shape[0]
        n = B * Y * X
        dy = dy_nhwc.to(x.dtype).contiguous()
        dil = ctx.dil
        nbr = _dense_rulebook(B, Y, X, x.device, dil)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.native and (_DENSE_CONV == 'halo' or dil != 1) and cout in (128, 256, 384) and cin % 128 == 0:
                # weight_t[c, 2-ky, 2-kx, n] = w[n, ky, kx, c]: the input gradient is a conv of dY with the flipped taps
                wt = w.view(cout, 3, 3, cin).flip(1, 2).permute(3, 1, 2, 0).reshape(cin, 9 * cout).contiguous()
                # (128 -> 384: the decoder conv's input gradient leaves with its column sums behind it -- the BatchNorm backward of
                #  the deconvolutions in front of the conv wants them, _DeblocksToDense.backward)
                dx = dense_conv3x3_halo(dy, wt, dil, post=skip, tail_sums=(dil == 1 and (cout, cin) == (128, 384))).to(ctx.meta[0])
                skip = None                                  # added inside the kernel
            elif ctx.native:
                nbr_t = _DENSE_NBR.get(('t', B, Y, X, dil, x.device))
                if nbr_t is None:                      # transposed rulebook of a stride-1 conv = flipped taps
                    nbr_t = _DENSE_NBR[('t', B, Y, X, dil, x.device)] = nbr.flip(1).contiguous()
                dx = spconv_bwd_data(dy.view(n, cout), nbr_t, w, cin).view(B, Y, X, cin).to(ctx.meta[0])
            else:
                dx = torch.ops.aten.convolution_backward(
                    dy.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w.contiguous(memory_format=torch.channels_last), None,
                    [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1, [True, False, False])[0].permute(0, 2, 3, 1).to(ctx.meta[0])
        if skip is not None:                                 # a path without the fused add
            dx = skip.to(ctx.meta[0]) if dx is None else dx + skip.to(dx.dtype)
        dy2, x2 = dy.view(n, cout), x.view(n, cin)
        dw = torch.empty((cout, 9 * cin), dtype=torch.float32, device=x.device)
        if ctx.native and _DENSE_WGRAD == 'halo' and x.dtype == torch.bfloat16:
            wsb = lib.tmae_dense_conv3x3_wgrad_workspace(cin, cout)
            if wsb:                                    # 0: a shape the halo weight-gradient kernel does not cover
                ws = _ws(wsb, x.device)
                check(lib.tmae_dense_conv3x3_wgrad(_p(dy), _p(x), B, Y, X, cin, cout, int(dil), _p(dw), _p(ws), wsb, _s()),
                      'tmae_dense_conv3x3_wgrad')
                return dx, dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2).to(ctx.meta[1]), None, None, None
        wsb = lib.tmae_linear_wgrad_workspace(n, cout, 9 * cin)
        ws = _ws(wsb, x.device)
        check(lib.tmae_spconv_wgrad(_p(dy2), dy2.stride(0), _p(x2), x2.stride(0), _p(nbr), n, cout, cin, _p(dw), _p(ws),
                                    wsb, _s()), 'tmae_spconv_wgrad')
        dw = dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2).to(ctx.meta[1])
        return dx, dw, None, None, None


def _channel_sums(dy):
    """sum over (0, 2, 3) of a [B, k, Y, X] tensor in channels-last memory, fp32: the column sums of the contiguous
    [B*Y*X, k] view.  For small k the view is re-cut into rows of L = lcm(64, k) elements (column c holds channel c % k), which
    is a shape tmae_column_sums takes; anything else: torch."""
    import math
    k = dy.shape[1]
    rows = dy.permute(0, 2, 3, 1)
    rows = rows.reshape(-1, k) if rows.is_contiguous() else rows.contiguous().view(-1, k)
    n = rows.shape[0]
    L = 64 * k // math.gcd(64, k)
    if L <= 512 and (n * k) % L == 0 and rows.dtype in (torch.bfloat16, torch.float32):
        view = rows.view(n * k // L, L)
        sums = torch.empty((L,), dtype=torch.float32, device=dy.device)
        wsb = lib.tmae_column_sums_workspace(view.shape[0], L)
        ws = _ws(wsb, dy.device)
        check(lib.tmae_column_sums(_p(view), _dt(view), view.shape[0], L, _p(sums), _p(ws), wsb, _s()), 'tmae_column_sums')
        return sums.view(L // k, k).sum(0)
    return rows.float().sum(0)


class _ConvOwnBiasGrad(torch.autograd.Function):
    """A biased Conv2d on a channels-last CUDA tensor: the library's forward (bias inside, one rounding), the library's input
    and weight gradients, but the bias gradient from _channel_sums.  torch's convolution backward sums it over (0, 2, 3) of the
    NCHW view of the channels-last gradient -- a strided reduction that took 0.7 ms per head on the [8, k <= 5, 468, 468] maps of
    CenterHead's last convs (center_head.py:11-45); the same sum over the rows of the contiguous view is one short pass."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation):
        cdt = compute_dtype(x)
        x_c = x.to(cdt)
        w_c, b_c = weight.to(cdt), bias.to(cdt)
        ctx.save_for_backward(x_c, w_c)
        ctx.conf = (stride, padding, dilation, x.dtype, weight.dtype, bias.dtype)
        return torch.nn.functional.conv2d(x_c, w_c, b_c, stride, padding, dilation)

    @staticmethod
    def backward(ctx, dy):
        x, w_c = ctx.saved_tensors
        stride, padding, dilation, xdt, wdt, bdt = ctx.conf
        dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, w_c, None, list(stride), list(padding), list(dilation), False,
                                                        [0, 0], 1, [ctx.needs_input_grad[0], True, False])
        return (None if dx is None else dx.to(xdt)), dw.to(wdt), _channel_sums(dy).to(bdt), None, None, None


class _NarrowConv3x3(torch.autograd.Function):
    """Conv2d(64, k <= 8, 3, padding=1, bias=True) on a channels-last bf16 map -- the last conv of a CenterHead branch
    (center_head.py:11-45) -- forward, input gradient and weight gradient as one pass over the 64-channel map each
    (csrc/headconv.hip); the bias gradient from _channel_sums.  The library pads these shapes to 32 output columns and takes
    0.37-0.53 / 0.13 / 0.34 ms per branch on the [8, 468, 468] map."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        xr = x.permute(0, 2, 3, 1)                                       # [B, Y, X, 64], contiguous (checked by the caller)
        B, Y, X, cin = xr.shape
        k = weight.shape[0]
        w2 = _derived(weight, ('narrow3x3',),
                      lambda t: t.detach().permute(0, 2, 3, 1).reshape(k, 9 * cin).to(torch.bfloat16).contiguous())
        b32 = _derived(bias, ('f32',), lambda t: t.detach().float().contiguous())
        out = torch.empty((B, Y, X, k), dtype=torch.bfloat16, device=x.device)
        check(lib.tmae_conv3x3_c64_narrow_fwd(_p(xr), cin, B, Y, X, _p(w2), _p(b32), k, _p(out), _s()),
              'tmae_conv3x3_c64_narrow_fwd')
        ctx.save_for_backward(xr, w2)
        ctx.conf = (x.dtype, weight.dtype, bias.dtype)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xr, w2 = ctx.saved_tensors
        xdt, wdt, bdt = ctx.conf
        B, Y, X, cin = xr.shape
        k = w2.shape[0]
        dyr = dy.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()     # [B, Y, X, k]
        dx = None
        if ctx.needs_input_grad[0]:
            dxr = torch.empty((B, Y, X, cin), dtype=torch.bfloat16, device=dy.device)
            wsb = lib.tmae_conv3x3_c64_narrow_bwd_data_workspace()
            ws = _ws(wsb, dy.device)
            check(lib.tmae_conv3x3_c64_narrow_bwd_data(_p(dyr), B, Y, X, k, _p(w2), _p(dxr), cin, _p(ws), wsb, _s()),
                  'tmae_conv3x3_c64_narrow_bwd_data')
            dx = dxr.permute(0, 3, 1, 2).to(xdt)
        dw = torch.empty((k, 9 * cin), dtype=torch.float32, device=dy.device)
        wsb = lib.tmae_conv3x3_c64_narrow_wgrad_workspace(k)
        ws = _ws(wsb, dy.device)
        check(lib.tmae_conv3x3_c64_narrow_wgrad(_p(dyr), _p(xr), cin, B, Y, X, k, _p(dw), _p(ws), wsb, _s()),
              'tmae_conv3x3_c64_narrow_wgrad')
        dw = dw.view(k, 3, 3, cin).permute(0, 3, 1, 2).to(wdt)
        return dx, dw, _channel_sums(dyr.permute(0, 3, 1, 2)).to(bdt)


class _Conv3x3C64(torch.autograd.Function):
    """Conv2d(64, 64, 3, padding=1, bias=False) on a channels-last bf16 activation [B, Y, X, 64] -- the stem convs of CenterHead's
    branches (center_head.py:28-31) -- forward, input gradient (the same kernel on dY with flipped, transposed weights) and
    weight gradient on csrc/headconv.hip (tmae_conv3x3_c64, tmae_conv3x3_c64_wgrad).
    chain: a second output, an alias of the input, for the NEXT branch to read instead of the shared tensor: the five branches'
    input gradients then arrive one through the other and every input-gradient launch accumulates into what came in (the kernel's
    accumulate form) -- autograd otherwise joins them with four adds over [B, Y, X, 64]."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, chain=False):
        x = x_nhwc.to(torch.bfloat16).contiguous()
        B, Y, X, cin = x.shape
        w2 = _derived(weight, ('c64_3x3',), lambda t: t.detach().permute(0, 2, 3, 1).reshape(64, 9 * 64).to(torch.bfloat16).contiguous())
        y = torch.empty((B, Y, X, 64), dtype=torch.bfloat16, device=x.device)
        wsb = lib.tmae_conv3x3_c64_workspace()
        ws = _ws(wsb, x.device)
        check(lib.tmae_conv3x3_c64(_p(x), 64, B, Y, X, _p(w2), 0, 0, _p(y), 64, _p(ws), wsb, _s()), 'tmae_conv3x3_c64')
        ctx.save_for_backward(x, w2)
        ctx.meta = (x_nhwc.dtype, weight.dtype)
        ctx.set_materialize_grads(False)
        if chain:
            return y, x_nhwc.view_as(x_nhwc)
        return y

    @staticmethod
    def backward(ctx, dy_nhwc, dalias=None):
        x, w2 = ctx.saved_tensors
        B, Y, X, _ = x.shape
        if dy_nhwc is None:                               # only the alias was used downstream
            return (None if dalias is None else dalias.to(ctx.meta[0])), None, None
        dy = dy_nhwc.to(torch.bfloat16).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            acc = dalias is not None
            dx = dalias.to(torch.bfloat16).contiguous() if acc else torch.empty_like(x)
            wsb = lib.tmae_conv3x3_c64_workspace()
            ws = _ws(wsb, x.device)
            check(lib.tmae_conv3x3_c64(_p(dy), 64, B, Y, X, _p(w2), 1, 1 if acc else 0, _p(dx), 64, _p(ws), wsb, _s()), 'tmae_conv3x3_c64')
            dx = dx.to(ctx.meta[0])
        dw = torch.empty((64, 9 * 64), dtype=torch.float32, device=x.device)
        wsb = lib.tmae_conv3x3_c64_wgrad_workspace()
        ws = _ws(wsb, x.device)
        check(lib.tmae_conv3x3_c64_wgrad(_p(dy), 64, _p(x), 64, B, Y, X, _p(dw), _p(ws), wsb, _s()), 'tmae_conv3x3_c64_wgrad')
        return dx, dw.view(64, 3, 3, 64).permute(0, 3, 1, 2).to(ctx.meta[1]), None


class _Conv3x3C128to64(torch.autograd.Function):
    """Conv2d(128, 64, 3, padding=1, bias=False) on a channels-last bf16 activation -- CenterHead's shared conv
    (center_head.py:85-89) -- as two 64-channel halves of the contraction on the 64 -> 64 kernels of csrc/headconv.hip: forward =
    two calls on the halves of the input (the second accumulates), input gradient = one call per half of dx (channel pitch 128),
    weight gradient = one call per half of the weight's input channels."""

    @staticmethod
    def forward(ctx, x_nhwc, weight):
        x = x_nhwc.to(torch.bfloat16).contiguous()
        B, Y, X, _ = x.shape

        def halves(t):
            w = t.detach().permute(0, 2, 3, 1).to(torch.bfloat16)                  # [64, 3, 3, 128]
            return tuple(w[..., 64 * h:64 * h + 64].reshape(64, 9 * 64).contiguous() for h in range(2))
        wa, wb = _derived(weight, ('c128to64_3x3',), halves)
        y = torch.empty((B, Y, X, 64), dtype=torch.bfloat16, device=x.device)
        wsb = lib.tmae_conv3x3_c64_workspace()
        for h, wh in enumerate((wa, wb)):
            ws = _ws(wsb, x.device)
            check(lib.tmae_conv3x3_c64(x[..., 64 * h:].data_ptr(), 128, B, Y, X, _p(wh), 0, h, _p(y), 64, _p(ws), wsb, _s()),
                  'tmae_conv3x3_c64')
        ctx.save_for_backward(x, wa, wb)
        ctx.meta = (x_nhwc.dtype, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, dy_nhwc):
        x, wa, wb = ctx.saved_tensors
        B, Y, X, _ = x.shape
        dy = dy_nhwc.to(torch.bfloat16).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wsb = lib.tmae_conv3x3_c64_workspace()
            for h, wh in enumerate((wa, wb)):
                ws = _ws(wsb, x.device)
                check(lib.tmae_conv3x3_c64(_p(dy), 64, B, Y, X, _p(wh), 1, 0, dx[..., 64 * h:].data_ptr(), 128, _p(ws), wsb, _s()),
                      'tmae_conv3x3_c64')
            dx = dx.to(ctx.meta[0])
        dw = torch.empty((2, 64, 9 * 64), dtype=torch.float32, device=x.device)
        wsb = lib.tmae_conv3x3_c64_wgrad_workspace()
        for h in range(2):
            ws = _ws(wsb, x.device)
            check(lib.tmae_conv3x3_c64_wgrad(_p(dy), 64, x[..., 64 * h:].data_ptr(), 128, B, Y, X, _p(dw[h]), _p(ws), wsb, _s()),
                  'tmae_conv3x3_c64_wgrad')
        dw = torch.cat((dw[0].view(64, 3, 3, 64), dw[1].view(64, 3, 3, 64)), dim=3).permute(0, 3, 1, 2).to(ctx.meta[1])
        return dx, dw


def conv3x3_c128to64_ok(x_nhwc, conv):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels == 128 and conv.out_channels == 64
            and x_nhwc.is_cuda and x_nhwc.dim() == 4 and compute_dtype(x_nhwc) == torch.bfloat16 and x_nhwc.is_contiguous()
            and x_nhwc.numel() * 2 < (1 << 31) and _os.environ.get('TMAE_HEAD_CONV', 'native') == 'native')


def conv3x3_c128to64(x_nhwc, weight):
    return _Conv3x3C128to64.apply(x_nhwc, weight)


def conv3x3_c64_ok(x_nhwc, conv):
    """a Conv2d(64, 64, 3, padding=1) on a contiguous channels-last bf16 map that csrc/headconv.hip takes (a bias in front of a
    training-mode norm is folded away by the caller)"""
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels == 64 and conv.out_channels == 64
            and x_nhwc.is_cuda and x_nhwc.dim() == 4 and compute_dtype(x_nhwc) == torch.bfloat16 and x_nhwc.is_contiguous()
            and x_nhwc.numel() * 2 < (1 << 31) and _os.environ.get('TMAE_HEAD_CONV', 'native') == 'native')


def conv3x3_c64(x_nhwc, weight, chain=False):
    """See _Conv3x3C64; chain=True: (y, alias of x_nhwc for the next reader of the same tensor)."""
    return _Conv3x3C64.apply(x_nhwc, weight, bool(chain))


def narrow_conv3x3_ok(x, conv):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None and conv.in_channels == 64
            and 1 <= conv.out_channels <= 8 and x.is_cuda and x.dtype == torch.bfloat16 and compute_dtype(x) == torch.bfloat16
            and x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous() and x.numel() * 2 < (1 << 31)
            and _os.environ.get('TMAE_HEAD_CONV', 'native') == 'native')


def conv3x3_channel_bias(x, conv):
    """conv(x) for a biased nn.Conv2d on a channels-last CUDA tensor (training): the 64 -> k <= 8 shapes of CenterHead's last
    convs on kernels of our own (_NarrowConv3x3), anything else through the library (_ConvOwnBiasGrad)."""
    if (conv.bias is None or not x.is_cuda or not torch.is_grad_enabled() or conv.groups != 1
            or not x.is_contiguous(memory_format=torch.channels_last)):
        return conv(x)
    if narrow_conv3x3_ok(x, conv):
        return _NarrowConv3x3.apply(x, conv.weight, conv.bias)
    return _ConvOwnBiasGrad.apply(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)


_COLSUM_TAIL = 2048       # bf16 elements behind a gradient map that carries its own column sums (dense_conv3x3_halo(tail_sums=True))


def colsum_tail(t, ncol):
    """The fp32 column sums [ncol] that ride behind the data of `t` ([..., ncol] bf16, contiguous, the whole of its storage) when
    `t` came out of dense_conv3x3_halo(..., tail_sums=True); None for any other tensor.  The storage size is the signature: the
    data followed by exactly _COLSUM_TAIL spare elements is an allocation only that call makes, and a tensor autograd has summed
    with another gradient, cast or copied is a new allocation without it."""
    if not (_DENSE_SUMS and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and t.storage_offset() == 0
            and t.shape[-1] == ncol and ncol * 4 <= _COLSUM_TAIL * 2):
        return None
    st = t.untyped_storage()
    if st.nbytes() != (t.numel() + _COLSUM_TAIL) * 2:
        return None
    return torch.empty(0, dtype=torch.float32, device=t.device).set_(st, t.numel() * 2 // 4, (ncol,))


def dense_conv3x3_halo(x_nhwc, w2d, dil=1, post=None, moments=0, tail_sums=False):
    """[B, Y, X, cin] bf16 (contiguous) x w2d [cout, 9*cin] bf16 -> [B, Y, X, cout] (csrc/spconv_igemm.hip, halo kernel);
    padding = dilation in {1, 2}; post [B, Y, X, cout] bf16: added to the result inside the kernel.
    moments = 2 (cin 384 -> cout 128, dil 1): returns (y, sums [2, cout] f32 = column sums of y and of y^2, from the conv's epilogue).
    tail_sums (cin 128 -> cout 384, dil 1): y's allocation carries its fp32 column sums behind the data (read them with
    colsum_tail(y, cout)); the return value is y alone."""
    B, Y, X, cin = x_nhwc.shape
    cout = w2d.shape[0]
    if (moments == 2 or tail_sums) and dil == 1 and _DENSE_SUMS and ((moments == 2 and (cin, cout) == (384, 128)) or (tail_sums and (cin, cout) == (128, 384))):
        if post is not None:
            assert post.shape == (B, Y, X, cout) and post.dtype == torch.bfloat16 and post.is_contiguous()
        n = B * Y * X * cout
        wsb = lib.tmae_dense_conv3x3_sums_workspace(cout)
        ws = _ws(wsb, x_nhwc.device)
        if tail_sums:
            buf = torch.empty((n + _COLSUM_TAIL,), dtype=torch.bfloat16, device=x_nhwc.device)
            y = buf[:n].view(B, Y, X, cout)
            sums = torch.empty(0, dtype=torch.float32, device=x_nhwc.device).set_(buf.untyped_storage(), n * 2 // 4, (cout,))
            check(lib.tmae_dense_conv3x3_sums(_p(x_nhwc), B, Y, X, cin, _p(w2d), cout, _p(post), 1, _p(y), _p(sums), _p(ws), wsb, _s()),
                  'tmae_dense_conv3x3_sums')
            return y
        y = torch.empty((B, Y, X, cout), dtype=torch.bfloat16, device=x_nhwc.device)
        sums = torch.empty((2, cout), dtype=torch.float32, device=x_nhwc.device)
        check(lib.tmae_dense_conv3x3_sums(_p(x_nhwc), B, Y, X, cin, _p(w2d), cout, _p(post), 2, _p(y), _p(sums), _p(ws), wsb, _s()),
              'tmae_dense_conv3x3_sums')
        return y, sums
    if moments == 2:
        return dense_conv3x3_halo(x_nhwc, w2d, dil, post), None
    y = torch.empty((B, Y, X, cout), dtype=torch.bfloat16, device=x_nhwc.device)
    if post is not None:
        assert post.shape == y.shape and post.dtype == torch.bfloat16 and post.is_contiguous()
        check(lib.tmae_dense_conv3x3_add(_p(x_nhwc), B, Y, X, cin, _p(w2d), cout, int(dil), _p(post), _p(y), _s()),
              'tmae_dense_conv3x3_add')
        return y
    if dil == 1:
        check(lib.tmae_dense_conv3x3(_p(x_nhwc), B, Y, X, cin, _p(w2d), cout, _p(y), _s()), 'tmae_dense_conv3x3')
    else:
        check(lib.tmae_dense_conv3x3_dilated(_p(x_nhwc), B, Y, X, cin, _p(w2d), cout, int(dil), _p(y), _s()),
              'tmae_dense_conv3x3_dilated')
    return y


def dense_conv3x3_ok(x_nhwc, conv):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.dilation in ((1, 1), (2, 2))
            and conv.padding == conv.dilation and conv.stride == (1, 1) and conv.groups == 1 and conv.bias is None
            and x_nhwc.is_cuda and compute_dtype(x_nhwc) == torch.bfloat16 and conv.in_channels % 128 == 0
            and conv.out_channels % 8 == 0 and x_nhwc.shape[0] * x_nhwc.shape[1] * x_nhwc.shape[2] >= 4096)


def dense_conv3x3(x_nhwc, weight, dilation=1, fork=False, moments=False):
    """fork: returns (y, alias of x_nhwc); moments: returns (y, [2, cout] column sums of y and y^2, or an empty tensor) -- see
    _DenseConv3x3.forward."""
    return _DenseConv3x3.apply(x_nhwc, weight, int(dilation), bool(fork), bool(moments))


class _DenseGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, grid, indices):
        dense = dense.contiguous()                       # [batch, ny, nx, c]
        batch, ny, nx, c = dense.shape
        m = indices.shape[0]
        rows = torch.empty((m, c), dtype=dense.dtype, device=dense.device)
        check(lib.tmae_dense_gather(_p(dense), _dt(dense), batch, ny, nx, c, _p(indices), m, _p(rows), _s()),
              'tmae_dense_gather')
        ctx.save_for_backward(grid)
        ctx.meta = (batch, ny, nx, c)
        return rows

    @staticmethod
    def backward(ctx, drows):
        (grid,) = ctx.saved_tensors
        batch, ny, nx, c = ctx.meta
        drows = drows.contiguous()
        out = torch.empty((batch, ny, nx, c), dtype=drows.dtype, device=drows.device)
        check(lib.tmae_sparse_to_dense(_p(drows), _dt(drows), drows.shape[0], c, _p(grid), batch, ny, nx, _p(out),
                                       _s()), 'tmae_sparse_to_dense')
        return out, None, None


def dense_gather(dense_nhwc, grid, indices):
    """rows[r] = dense[b,y,x,:] for the (unique) sites in indices (SiamWCA_MAE.py:308-312)."""
    return _DenseGather.apply(dense_nhwc, grid, indices)


# ----------------------------------------------------------------------------- Chamfer (A13)

class _Chamfer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, weights):
        pred = pred.contiguous().float()
        gt = gt.contiguous().float()
        w = weights.contiguous().float()
        m, np_, _ = pred.shape
        ng = gt.shape[1]
        dev = pred.device
        per = torch.empty((m,), dtype=torch.float32, device=dev)
        ix = torch.empty((m, np_), dtype=torch.int8, device=dev)
        iy = torch.empty((m, ng), dtype=torch.int8, device=dev)
        check(lib.tmae_chamfer_fwd(_p(pred), _p(gt), _p(w), m, np_, ng, _p(per), _p(ix), _p(iy), _s()),
              'tmae_chamfer_fwd')
        wsum = w.sum()
        inv = torch.where(wsum > 0, 1.0 / wsum.clamp(min=1e-30), torch.zeros_like(wsum))
        ctx.save_for_backward(pred, gt, w, ix, iy, inv)
        return per.sum() * inv

    @staticmethod
    def backward(ctx, g):
        pred, gt, w, ix, iy, inv = ctx.saved_tensors
        m, np_, _ = pred.shape
        scale = (g.float() * inv).reshape(1).contiguous()
        dpred = torch.empty_like(pred)
        check(lib.tmae_chamfer_bwd(_p(pred), _p(gt), _p(w), _p(ix), _p(iy), _p(scale), m, np_, gt.shape[1],
                                   _p(dpred), _s()), 'tmae_chamfer_bwd')
        return dpred, None, None


def chamfer_distance(pred, gt, weights=None):
    """pytorch3d.loss.chamfer_distance(x, y, weights=) -> (loss, None) (SiamWCA_MAE.py:163)."""
    if weights is None:
        weights = torch.ones(pred.shape[0], device=pred.device)
    return _Chamfer.apply(pred, gt, weights), None


# ----------------------------------------------------------------------------- fine-tune path: CenterHead

def centerhead_targets(gt_boxes, cls_map, num_class_head, fm_hw, pc_range, voxel_size, stride, nmax, overlap, min_radius):
    """CenterHead.assign_targets for one head (center_head.py:107-231) on the device: gt_boxes [B, M, 8] f32
    (class 1..n in the last column, zero rows = padding), cls_map [n+1] i32 (global class id -> index in the head or
    -1).  Returns heatmap [B, C, H, W], target_boxes [B, nmax, 8], inds [B, nmax] i64, mask [B, nmax] i64."""
    _need_cuda(gt_boxes)
    g = gt_boxes.contiguous().float()
    B, M, ncode = g.shape
    H, W = int(fm_hw[0]), int(fm_hw[1])
    dev = g.device
    heat = torch.zeros((B, num_class_head, H, W), dtype=torch.float32, device=dev)
    tb = torch.zeros((B, nmax, ncode), dtype=torch.float32, device=dev)
    inds = torch.zeros((B, nmax), dtype=torch.int64, device=dev)
    mask = torch.zeros((B, nmax), dtype=torch.int64, device=dev)
    check(lib.tmae_centerhead_targets(_p(g), B, M, ncode, _p(cls_map), cls_map.numel() - 1, num_class_head, H, W,
                                      float(pc_range[0]), float(pc_range[1]), float(voxel_size[0]), float(voxel_size[1]),
                                      float(stride), int(nmax), float(overlap), int(min_radius), _p(heat), _p(tb), _p(inds),
                                      _p(mask), _s()), 'tmae_centerhead_targets')
    return heat, tb, inds, mask


class _FocalLossCenterNet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        x = logits.contiguous()
        t = target.contiguous().float()
        n = x.numel()
        out = torch.empty((4,), dtype=torch.float32, device=x.device)
        wsb = lib.tmae_focal_loss_workspace(n)
        ws = _ws(wsb, x.device)
        check(lib.tmae_focal_loss_fwd(_p(x), _dt(x), _p(t), n, _p(out), _p(ws), wsb, _s()), 'tmae_focal_loss_fwd')
        ctx.save_for_backward(x, t, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, t, out = ctx.saved_tensors
        dx = torch.empty_like(x)
        g32 = g.reshape(1).float().contiguous()
        check(lib.tmae_focal_loss_bwd(_p(x), _dt(x), _p(t), x.numel(), _p(out), _p(g32), _p(dx), _s()),
              'tmae_focal_loss_bwd')
        return dx, None


def focal_loss_centernet(logits, target):
    """FocalLossCenterNet()(clamp(sigmoid(logits), 1e-4, 1 - 1e-4), target) (center_head.py:233-244,
    loss_utils.py:273-309) as one fused forward and one fused backward kernel."""
    _need_cuda(logits)
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    return _FocalLossCenterNet.apply(logits, target)


# ----------------------------------------------------------------------------- rotated boxes: IoU / NMS (iou3d_nms)

def _boxes_pairwise(a, b, mode):
    _need_cuda(a)
    a = a[:, :7].contiguous().float()
    b = b[:, :7].contiguous().float()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(lib.tmae_boxes_pairwise(_p(a), a.shape[0], _p(b), b.shape[0], mode, _p(out), _s()), 'tmae_boxes_pairwise')
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.boxes_overlap_bev_gpu: BEV intersection areas [na, nb]."""
    return _boxes_pairwise(boxes_a, boxes_b, 0)


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.boxes_iou_bev (iou3d_nms_utils.py:31-45)."""
    return _boxes_pairwise(boxes_a, boxes_b, 1)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.boxes_iou3d_gpu (iou3d_nms_utils.py:48-81)."""
    return _boxes_pairwise(boxes_a, boxes_b, 2)


def boxes_iou3d_paired(boxes_a, boxes_b):
    """diag(boxes_iou3d_gpu(a, b)) without the matrix: out [n] (IoULossCenterNet's targets, loss_utils.py:411-420)."""
    _need_cuda(boxes_a)
    a = boxes_a[:, :7].contiguous().float()
    b = boxes_b[:, :7].contiguous().float()
    assert a.shape == b.shape
    out = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device)
    check(lib.tmae_boxes_pairwise(_p(a), a.shape[0], _p(b), b.shape[0], 6, _p(out), _s()), 'tmae_boxes_pairwise')
    return out


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.nms_gpu (iou3d_nms_utils.py:84-99): indices of the kept boxes in descending score order.
    The suppression pass runs on the device; the only host sync is reading the number of kept boxes."""
    _need_cuda(boxes)
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].contiguous().float()
    n = b.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    num = torch.zeros((1,), dtype=torch.int32, device=b.device)
    wsb = lib.tmae_nms_bev_workspace(n)
    ws = _ws(wsb, b.device)
    check(lib.tmae_nms_bev(_p(b), n, float(thresh), _p(keep), _p(num), _p(ws), wsb, _s()), 'tmae_nms_bev')
    return order[keep[:int(num.item())]].contiguous(), None


# ----------------------------------------------------------------------------- data path (frame preparation)

def frame_prepare(points, r1t1, m2, ego_radius, flip_x, flip_y, cosa, sina, scale, pc_range, batch_idx, remove_boxes=None):
    """One frame of one sample through tmae_frame_prepare: returns (rows [n, row+1] worst case, count [1] i32 on the
    device); the caller slices after its one sync.  r1t1 / m2: float64 host arrays of 12 or None.
    remove_boxes: device [nb, 8] float64 (tmae_frame_prepare_boxes: gt_sampling's point removal) or None."""
    import ctypes as C
    _need_cuda(points)
    pts = points.contiguous().float()
    n, row = pts.shape
    out = torch.empty((n, row + 1), dtype=torch.float32, device=pts.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=pts.device)
    wsb = lib.tmae_frame_prepare_workspace(n)
    ws = _ws(wsb, pts.device)

    def dptr(a):
        if a is None:
            return None, None
        arr = (C.c_double * 12)(*[float(v) for v in a])
        return arr, C.cast(arr, C.c_void_p)
    k1, p1 = dptr(r1t1)
    k2, p2 = dptr(m2)
    nb = 0 if remove_boxes is None else int(remove_boxes.shape[0])
    check(lib.tmae_frame_prepare_boxes(_p(pts), row, n, p1, p2, float(ego_radius), int(bool(flip_x)), int(bool(flip_y)), float(cosa),
                                       float(sina), float(scale), float(pc_range[0]), float(pc_range[1]), float(pc_range[3]),
                                       float(pc_range[4]), int(batch_idx), _p(remove_boxes) if nb else None, nb, _p(out),
                                       _p(cnt), _p(ws), wsb, _s()), 'tmae_frame_prepare_boxes')
    return out, cnt
