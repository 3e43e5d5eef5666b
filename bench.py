"""Benchmark of the T-MAE pre-training hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W          # starts N rank processes itself (before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # or under an external launcher (WORLD_SIZE set)

A "step" = one full pre-training iteration (VFE -> Siamese SST encoder -> masking -> WCA -> dense decoder ->
Chamfer -> backward -> Adam one-cycle) over one batch of synthetic 120k-point ONCE-shape frame pairs, inputs
resident in HBM.  value = frame-pairs/s of the whole job (all ranks).  Rank 0 prints ONE JSON line carrying the
`roofline` of the dominant hand-written kernel (timed live with HIP events on the launch stream) and the
`cpu_baseline` (the oracle timed on this box's host cores on one frame pair).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 't-mae_amd'))

# MIOpen's default find benchmarks every applicable solver for the 4 dense decoder convolutions, including the
# reference 'naive' solvers that take ~90 s on these shapes; they are never the winner, so leave them out of the search
for _k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD',
           'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
    os.environ.setdefault(_k, '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16
VALU_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-per-gpu', type=int, default=8)        # OPTIMIZATION.BATCH_SIZE_PER_GPU
    ap.add_argument('--points', type=int, default=120000)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--task', default='pretrain', choices=['pretrain', 'finetune'],
                    help='pretrain = the metric (t_mae_ssl.yaml); finetune = BASELINE configs[4]: CenterPoint head on the '
                         'two-frame encoder (t_mae.yaml, synthetic labels; informational, batch 6 = its recipe)')
    ap.add_argument('--shape', default='once', choices=['once', 'waymo'],
                    help='once = BASELINE configs[1] (the metric); waymo = configs[3] shape: 5 point features, z in [-2,4), '
                         '6 m pillars (use with --points 180000)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the short fine-tune / Waymo-shape runs that fill the `secondary` field of the N = 1 line')
    ap.add_argument('--no-vfe-prefetch', action='store_true',
                    help='voxelise every batch inside its own step (the default hands a step the batch that follows it, see step()); '
                         'for kernel traces: under rocprofv3 the slowed-down host turns the lookahead into idle gaps')
    ap.add_argument('--skip-unread-gradients', action='store_true',
                    help='variant, not the reference step: no gradients for the parameters its optimizer never owns')
    ap.add_argument('--cpu-points', type=int, default=120000)
    ap.add_argument('--probe-only', action='store_true', help='only time the roofline kernels')
    ap.add_argument('--probe-hbm-only', action='store_true',
                    help='only the four HBM-priced probes (token GEMMs, weight gradient, attention): the command of the counter '
                         'passes, whose per-kernel averages must not mix with the other probes\' launches of the same kernels')
    return ap.parse_args()


def _pmc_files():
    """Committed counter passes of `bench.py --probe-hbm-only` (profiles/roundN_pmc.json), latest round first."""
    import glob
    import re
    fs = glob.glob(os.path.join(ROOT, 'profiles', 'round*_pmc.json'))
    return sorted(fs, key=lambda f: -int(re.search(r'round(\d+)_pmc', f).group(1)))


_PMC_FILES = {'token_gemm': 'token_gemm_wreg.hip', 'token_gemm_gelu': 'token_gemm_wreg.hip', 'wgrad': 'wgrad.hip',
              'attention': 'attention_mfma.hip'}


def _pmc_entry(key):
    """(file, json, bytes per op) of the latest pass that holds a VALID figure for `key`: a missing, empty or zero entry is
    'absent' (round 4 committed a 0 for the priced kernel -- a summariser whose name pattern had gone stale -- and `0 is not
    None` put it into the driver's line)."""
    for f in _pmc_files():
        d = json.load(open(f))
        v = d.get(key, {}).get('traffic_bytes_per_op')
        if v:
            return f, d, v
    return None, None, None


def _pmc_source(key):
    f, _, _ = _pmc_entry(key)
    if f is None:
        return None
    return (f'profiles/{os.path.basename(f)} (FETCH_SIZE x2 + WRITE_SIZE, bytes per op; a committed counter pass of '
            f'`bench.py --probe-hbm-only`, not measured in this run)')


def _pmc_current(key):
    """Is the committed counter pass still about THIS kernel source?  (sha256 of the .hip file at measurement time, written by
    profiles/scripts/pmc_summary.py, against the file in this tree; None for passes of earlier rounds that carry no hash.)"""
    import hashlib
    _, d, _ = _pmc_entry(key)
    if d is None:
        return None
    want = d.get('source_sha16', {}).get(_PMC_FILES[key])
    if want is None:
        return None
    src = os.path.join(ROOT, 't-mae_amd', 'csrc', _PMC_FILES[key])
    return hashlib.sha256(open(src, 'rb').read()).hexdigest()[:16] == want


def _pmc(key):
    return _pmc_entry(key)[2]


def box_peaks(dev, reps=5):
    """What THIS box sustains: a 16-bytes-per-lane streaming copy of 1 GiB (HBM
    bytes = read + write) and a register-resident bf16 MFMA loop on random operands, both library probes
    (csrc/probe.hip), HIP events on the launch stream.  Every `roofline*` entry carries these as `peak_measured` next to
    the spec `peak`; `frac_of_measured` is the fraction that is comparable across boxes."""
    import ctypes as C
    from tmae_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    nbytes = 1 << 30
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    flops = C.c_int64(0)

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_copy = min(timed(lambda nt=nt: check(lib.tmae_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, nt, st), 'tmae_probe_copy'))
                  for nt in (0, 1))                          # default and non-temporal cache policy: the faster one
    ms_mfma = timed(lambda: check(lib.tmae_probe_mfma(8192, sink.data_ptr(), C.addressof(flops), st), 'tmae_probe_mfma'))
    del src, dst
    return {'hbm_copy_gbs': round(2 * nbytes / (ms_copy * 1e-3) / 1e9, 1), 'hbm_copy_ms': round(ms_copy, 4),
            'mfma_bf16_tflops': round(flops.value / (ms_mfma * 1e-3) / 1e12, 1), 'mfma_ms': round(ms_mfma, 4),
            'how': '1 GiB float4 streaming copy (read + write bytes) and a 16x16x32 bf16 MFMA loop on random register '
                   'operands, 512 threads per CU (csrc/probe.hip); mean of %d launches' % reps}


def _step_bytes():
    """Whole-step HBM traffic of the committed counter pass (profiles/scripts/pmc_step.sh: FETCH_SIZE x2 + WRITE_SIZE of every
    kernel of three timed steps of this configuration), newest round first."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*_step_bytes.json')), reverse=True):
        try:
            d = json.load(open(f))
            return {'bytes': d['step_hbm_bytes'], 'read': d['step_fetch_bytes'], 'written': d['step_write_bytes'],
                    'families_gb': {k: round(v['bytes'] / 1e9, 2) for k, v in sorted(d['families'].items(), key=lambda kv: -kv[1]['bytes'])},
                    'source': 'profiles/' + os.path.basename(f) + ' (a committed rocprofv3 --pmc pass of `bench.py --steps 3 --warmup 2`, not measured in this run)'}
        except Exception:
            continue
    return None


def _with_measured(entry, peaks):
    """peak_measured / frac_of_measured beside the spec-peak fraction of a roofline entry."""
    if entry is None or peaks is None:
        return entry
    pm = peaks['hbm_copy_gbs'] if entry.get('bound') == 'hbm' else peaks['mfma_bf16_tflops']
    entry['peak_measured'] = pm
    entry['frac_of_measured'] = round(entry['achieved'] / pm, 5) if pm else None
    return entry


_STAGE2_TOKENS = []


def _stage2_tokens(model, batch):
    """Stage-2 token count (previous + current frame) of `batch`: one no-grad forward pass, once per process."""
    if not _STAGE2_TOKENS:
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            model(dict(batch))
        _STAGE2_TOKENS.append(int(model.backbone_3d.last_pair_tokens[1]))
    return _STAGE2_TOKENS[0]


def token_gemm_roofline(model, batch, amp_dtype, iters=20):
    """`roofline`: the dominant hand-written kernel family of the step by GPU time (profiles/round1_g_kernel_stats.md):
    the token GEMM, here its W-in-registers kernel token_gemm_wreg_kernel<256,4,8> (csrc/token_gemm_wreg.hip, round 3;
    rounds 1-2 priced the W-resident-in-LDS kernel on the same shape) on its heaviest frequent shape -- the stage-2 FFN /
    q,k in-projection  Y[m,512] = X[m,256] W^T + b  over the token list of both frames (bench batch).  One launch per
    op, timed with HIP events on the launch stream.  Algorithmic bytes: X read once, Y written once, W and b read once
    (DESIGN.md section 4).  The probe's launches are the last 23 launches of that kernel in the process (3 warm-up + 20
    timed): that is how the profile summaries tell them from the forward pass that measures m."""
    from tmae_amd._lib import lib, check
    m = _stage2_tokens(model, batch)
    n, k = 512, 256
    dev = next(model.parameters()).device
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    b = torch.randn(n, device=dev).bfloat16()
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        check(lib.tmae_token_gemm(x.data_ptr(), k, m, k, w.data_ptr(), n, b.data_ptr(), y.data_ptr(), n, st),
              'tmae_token_gemm')
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_alg = m * (n + k) * 2 + (n * k + n) * 2
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    return {'kernel': 'token_gemm_wreg_kernel<256,4,8> (W-in-registers persistent token GEMM Y[m,512] = X[m,256] W^T + b of the '
                      'stage-2 token list; one launch per op)', 'bound': 'hbm', 'achieved': round(achieved, 2),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5),
            'traffic': _pmc('token_gemm'), 'traffic_is_of_this_source': _pmc_current('token_gemm'),
            'traffic_source': _pmc_source('token_gemm'),
            'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(bytes_alg), 'tokens': m, 'n': n, 'k': k}


def gelu_gemm_roofline(model, batch, amp_dtype, iters=20):
    """`roofline` (round 5; `roofline_gemm_gelu` in round 4): the FFN's first Linear with its exact GELU as a second store of the epilogue
    (token_gemm_wreg_kernel<256,4,8,...,GELU2>, csrc/token_gemm_wreg.hip) on the token-GEMM probe's shape: x read once, the
    pre-activation AND the activation written once each -- algorithmic bytes M (K + 2N) 2 + W."""
    from tmae_amd._lib import lib, check
    m = _stage2_tokens(model, batch)
    n, k = 512, 256
    dev = next(model.parameters()).device
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    b = torch.randn(n, device=dev).bfloat16()
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    yg = torch.empty_like(y)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        check(lib.tmae_token_gemm_gelu(x.data_ptr(), k, m, k, w.data_ptr(), n, b.data_ptr(), y.data_ptr(), yg.data_ptr(), n, st),
              'tmae_token_gemm_gelu')
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_alg = m * (2 * n + k) * 2 + (n * k + n) * 2
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    return {'kernel': 'token_gemm_wreg_kernel<256,4,8,..,GELU2> (Y = X W^T + b and gelu(Y) in one launch, stage-2 token list)',
            'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': _pmc('token_gemm_gelu'),
            'traffic_is_of_this_source': _pmc_current('token_gemm_gelu'), 'traffic_source': _pmc_source('token_gemm_gelu'),
            'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(bytes_alg), 'tokens': m, 'n': n, 'k': k}


def wgrad_roofline(model, batch, amp_dtype, iters=20):
    """`roofline_wgrad` (the step's dominant kernel until its loads were made branch-free): the token-split weight
    gradient of the d = 256 stages, wgrad256_kernel (csrc/wgrad.hip), on its most frequent heavy shape -- the
    stage-2 FFN / in-projection  dW[512,256] = dY^T X  over the token list of both frames (bench batch).  The op = that
    kernel + its slab-reduction launch, timed with HIP events on the launch stream.  Algorithmic bytes: dY and
    X read once, dW and db written once; the fp32 slabs are overhead and show up in `traffic`."""
    from tmae_amd import ops
    m = _stage2_tokens(model, batch)
    n, k = 512, 256
    dev = next(model.parameters()).device
    dy = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    for _ in range(3):
        ops.linear_wgrad(dy, x, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.linear_wgrad(dy, x, True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_alg = m * (n + k) * 2 + (n * k + n) * 4
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    return {'kernel': 'wgrad256_kernel (token-split weight gradient dW[512,256] = dY^T X of the stage-2 token list; the '
                      'op = the kernel + its slab-reduction launch)', 'bound': 'hbm',
            'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': _pmc('wgrad'), 'traffic_is_of_this_source': _pmc_current('wgrad'),
            'traffic_source': _pmc_source('wgrad'),
            'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(bytes_alg), 'tokens': m, 'n': n, 'k': k}


def attention_roofline(model, batch, amp_dtype, iters=20):
    """Times the dominant hand-written kernel of the step -- the ragged window attention backward
    (`win_attn_bwd_mfma_kernel<16,NT,PAIR>`: one launch per window size class) on the stage-1 previous-frame tensors --
    through the C ABI with HIP events on torch's current stream (the stream the ABI launches on), nothing else in
    between.  Priced by its ALGORITHMIC HBM bytes: q,k,v,out,dout read once, dq,dk,dv written
    once, + lse + the dense index grid."""
    from tmae_amd import ops
    from tmae_amd._lib import lib, check
    with torch.no_grad():
        bd = model.vfe(dict(batch))
        ind = bd['voxel_coords_prev'][:, [0, 2, 3]].int().contiguous()
        bs = int(bd['batch_size'])
        grid = ops.index_grid(ind, bs, 468, 468)
        wl = ops.window_worklist(grid, grid, bs, 468, 468, False)
    m, d, H, dh = ind.shape[0], 128, 8, 16
    dt = amp_dtype or torch.float32
    es, code = (2, 1) if dt == torch.bfloat16 else (4, 0)
    dev = ind.device
    qk = torch.randn(m, 2 * d, device=dev).to(dt)
    v = torch.randn(m, d, device=dev).to(dt)
    dout = torch.randn(m, d, device=dev).to(dt)
    out, dqk, dv = torch.empty_like(v), torch.empty_like(qk), torch.empty_like(v)
    lse = torch.empty(m, H, device=dev)
    tau = torch.ones(1, device=dev)
    nblk = lib.tmae_win_attn_num_blocks(bs, 468, 468, H, dh)
    part = torch.zeros(nblk, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    wlp = wl.data_ptr() if code == 1 else None
    q_, k_ = qk.data_ptr(), qk.data_ptr() + d * es
    check(lib.tmae_win_attn_fwd(q_, 2 * d, k_, 2 * d, v.data_ptr(), d, code, m, m, H, dh, grid.data_ptr(),
                                grid.data_ptr(), bs, 468, 468, 0, tau.data_ptr(), 0.01, out.data_ptr(), d,
                                lse.data_ptr(), wlp, 0, st), 'fwd')

    def bwd():
        check(lib.tmae_win_attn_bwd(q_, 2 * d, k_, 2 * d, v.data_ptr(), d, out.data_ptr(), d, dout.data_ptr(), d,
                                    lse.data_ptr(), code, m, m, H, dh, grid.data_ptr(), grid.data_ptr(), bs, 468, 468,
                                    0, tau.data_ptr(), 0.01, dqk.data_ptr(), 2 * d, dqk.data_ptr() + d * es, 2 * d,
                                    dv.data_ptr(), d, part.data_ptr(), wlp, 0, st), 'bwd')
    for _ in range(3):
        bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    # algorithmic bytes of the op: read q,k,v,dout (+ out on the fp32 path; the bf16 kernel rebuilds D from P.dP),
    # write dq,dk,dv; + the lse rows and one pass over the row-index grid
    units = 7 if code == 1 else 8
    bytes_alg = m * d * es * units + m * H * 4 + bs * 468 * 468 * 4
    achieved = bytes_alg / (ms * 1e-3) / 1e9
    traffic = _pmc('attention') if code == 1 else None
    return {'kernel': 'win_attn_bwd_mfma_kernel<16,NT> (stage-1 self-attention backward, previous frame; the op = its 3 '
                      'tile-class launches NT=1,2,4)', 'bound': 'hbm', 'achieved': round(achieved, 2),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
            'traffic_is_of_this_source': _pmc_current('attention') if code == 1 else None,
            'traffic_source': _pmc_source('attention'),
            'ms_per_launch': round(ms, 4), 'algorithmic_bytes': int(bytes_alg), 'tokens': int(m)}


def _cpu_now():
    """The CPU this thread runs on (diagnostic: is the launching thread being moved between cores / NUMA nodes?)."""
    try:
        import ctypes
        return int(ctypes.CDLL(None).sched_getcpu())
    except Exception:
        return -1


def _host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))          # the 1-GPU box's CPU share is 16 cores


def _oracle_case(O, n_points, stages, iters, pred_scale=1.0):
    """Warm-up + `iters` timed forward+backward passes of the oracle on ONE synthetic frame pair (seed 0)."""
    cfg = O.default_model_cfg(stages)
    P = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=0, pred_scale=pred_scale).items()}
    pts, prv = O.synth_frame_pair(n_points, 1, seed=0)
    vox = O.voxelize(pts, cfg['point_cloud_range'], cfg['voxel_size'], cfg['grid_size'])
    noise = np.random.default_rng(0).random(vox['voxel_coords'].shape[0]).astype(np.float32)
    times, loss = [], None
    for i in range(iters + 1):
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        loss = O.forward_loss(P, pts, prv, noise, 1, cfg)
        loss.backward()
        if i > 0:                           # iteration 0 = warm-up (allocator, thread pool, first-touch)
            times.append(time.perf_counter() - t0)
    return times, float(loss.detach()), (cfg, pts, prv, noise)


def igemm_roofline(batch_per_gpu, iters=10):
    """`roofline_igemm`: the heaviest MFMA-bound launch of the step -- the dense decoder conv forward (B x 468 x 468 cells,
    384 -> 128 channels, 3 x 3) on the halo-tiled implicit-GEMM kernel of csrc/spconv_igemm.hip.  Priced by FLOPs
    (2 * cells * 9 * 384 * 128) against the dense bf16 MFMA peak, timed with HIP events on the launch stream."""
    from tmae_amd import ops
    dev = torch.device('cuda', torch.cuda.current_device())
    B, Y, X = batch_per_gpu, 468, 468
    x = torch.randn(B, Y, X, 384, device=dev).bfloat16()
    w = (torch.randn(128, 9 * 384, device=dev) * 0.02).bfloat16()
    for _ in range(2):
        ops.dense_conv3x3_halo(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.dense_conv3x3_halo(x, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * Y * X * 9 * 384 * 128
    tf = flops / (ms * 1e-3) / 1e12
    return {'kernel': 'dense_conv3x3_halo_kernel<384> (dense decoder conv forward, halo-tiled implicit GEMM; one launch per '
                      'op)', 'bound': 'mfma', 'achieved': round(tf, 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(tf / MFMA_BF16_PEAK_TFLOPS, 5), 'traffic': None, 'ms_per_launch': round(ms, 4),
            'algorithmic_flops': flops, 'cells': B * Y * X}


def dense_wgrad_roofline(batch_per_gpu, iters=10):
    """`roofline_dense_wgrad`: the decoder conv's weight gradient (B x 468 x 468 cells, dW[128, 9 x 384]) on the halo kernel of
    csrc/dense_wgrad.hip + its slab reduction, through the C ABI.  MFMA-bound: 2 * cells * 9 * 384 * 128 FLOPs per launch
    against the dense bf16 MFMA peak; HIP events on the launch stream."""
    from tmae_amd._lib import lib, check
    dev = torch.device('cuda', torch.cuda.current_device())
    B, Y, X, cin, cout = batch_per_gpu, 468, 468, 384, 128
    x = torch.randn(B, Y, X, cin, device=dev).bfloat16()
    dy = torch.randn(B, Y, X, cout, device=dev).bfloat16()
    dw = torch.empty(cout, 9 * cin, device=dev)
    wsb = lib.tmae_dense_conv3x3_wgrad_workspace(cin, cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        check(lib.tmae_dense_conv3x3_wgrad(dy.data_ptr(), x.data_ptr(), B, Y, X, cin, cout, 1, dw.data_ptr(), ws.data_ptr(), wsb, st),
              'tmae_dense_conv3x3_wgrad')
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * Y * X * 9 * cin * cout
    tf = flops / (ms * 1e-3) / 1e12
    return {'kernel': 'dense_wgrad_halo_kernel<1> + dense_wgrad_reduce_kernel (dense decoder conv weight gradient dW[128, 9 x 384]; '
                      'the op = both launches)', 'bound': 'mfma', 'achieved': round(tf, 1), 'peak': MFMA_BF16_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(tf / MFMA_BF16_PEAK_TFLOPS, 5), 'traffic': None, 'ms_per_launch': round(ms, 4),
            'algorithmic_flops': flops, 'cells': B * Y * X}


def spconv_rooflines(model, iters=10):
    """`roofline_spconv` / `roofline_spconv_wgrad`: the d = 256 submanifold conv of the encoder's second stage (conv_out of
    sst_block 2: 256 -> 256, spt_backbone.py:279-304 through spconv_utils.post_act_block) over the token list of BOTH frames with
    the batch's real rulebook -- forward on spconv_igemm_ring256_kernel<256>, weight gradient on
    wgrad256_kernel<true> + the slab reduction.  MFMA-bound; FLOPs = 2 x ACTIVE (input, output) pairs x 256 x 256 (SURVEY 8d:
    the kernels also multiply the zeros of absent neighbours, which is not counted).  Needs the index sets of a forward pass
    (step_flops ran one)."""
    from tmae_amd import ops
    from tmae_amd._lib import lib, check
    bb = model.backbone_3d
    ind_p, ind_c, (ny, nx) = bb.last_stage_indices[1]
    dev = ind_p.device
    B = int(max(int(ind_p[:, 0].max()), int(ind_c[:, 0].max()))) + 1
    shift = torch.tensor([B, 0, 0], dtype=torch.int32, device=dev)
    ind = torch.cat([ind_p, ind_c + shift], 0).contiguous()
    grid = ops.index_grid(ind, 2 * B, ny, nx)
    nbr = ops.spconv_neighbors(ind, grid, 2 * B, ny, nx, 1)
    m, cin, cout = int(ind.shape[0]), 256, 256
    pairs = float((nbr >= 0).sum())
    feat = torch.randn(m, cin, device=dev).bfloat16()
    dy = torch.randn(m, cout, device=dev).bfloat16()
    w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).bfloat16()
    out = torch.empty(m, cout, device=dev, dtype=torch.bfloat16)
    dw = torch.empty(cout, 9 * cin, device=dev)
    wsb = lib.tmae_linear_wgrad_workspace(m, cout, 9 * cin)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        check(lib.tmae_spconv_fwd(feat.data_ptr(), cin, m, cin, nbr.data_ptr(), m, w.data_ptr(), cout, out.data_ptr(), cout, st), 'tmae_spconv_fwd')

    def wg():
        check(lib.tmae_spconv_wgrad(dy.data_ptr(), cout, feat.data_ptr(), cin, nbr.data_ptr(), m, cout, cin, dw.data_ptr(), ws.data_ptr(), wsb, st),
              'tmae_spconv_wgrad')
    res = {}
    flops = 2.0 * pairs * cin * cout
    for key, fn, name in (('roofline_spconv', fwd, 'spconv_igemm_ring256_kernel<256>: stage-2 '
                                                   'submanifold conv 256 -> 256 forward, both frames'),
                          ('roofline_spconv_wgrad', wg, 'wgrad256_kernel<true> + wgrad_reduce_kernel: the same conv\'s weight gradient '
                                                        'dW[256, 9 x 256] through the rulebook')):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = flops / (ms * 1e-3) / 1e12
        res[key] = {'kernel': name, 'bound': 'mfma', 'achieved': round(tf, 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(tf / MFMA_BF16_PEAK_TFLOPS, 5), 'traffic': None, 'ms_per_launch': round(ms, 4),
                    'algorithmic_flops': flops, 'rows': m, 'active_pairs': pairs, 'executed_flops': 2.0 * m * 9 * cin * cout}
    return res


def cpu_baseline(n_points, iters=3):
    """The oracle (oracle/tmae_oracle.py, a CPU restatement pinned against the reference) timed on this box's
    host cores: forward + backward of ONE frame pair, fp32, all cores -- 1 warm-up + `iters` timed iterations at
    C2 (120 k points, full model; the metric's configuration) and at C1 (8 k points, one SST stage)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import tmae_oracle as O
    cores = _host_cores()
    torch.set_num_threads(cores)
    t2, loss2, case2 = _oracle_case(O, n_points, 3, iters)
    t1, _, _ = _oracle_case(O, 8000, 1, iters)
    med2, med1 = float(np.median(t2)), float(np.median(t1))
    try:
        model = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][0]
    except Exception:
        model = 'unknown'
    out = {'value': round(1.0 / med2, 5), 'unit': 'frame-pairs/s', 'cores': cores, 'kind': 'port',
           'sample': f'C2: 1 frame pair of {n_points} pts/frame, full 3-stage model, forward+backward (no optimizer), '
                     f'fp32; 1 warm-up + {iters} timed iterations, median {med2:.2f} s '
                     f'(all: {", ".join(f"{t:.2f}" for t in t2)})',
           'c1': {'value': round(1.0 / med1, 5), 'unit': 'frame-pairs/s',
                  'sample': f'C1: 1 frame pair of 8000 pts/frame, one SST stage; 1 warm-up + {iters} timed iterations, '
                            f'median {med1:.2f} s'},
           'cpu_model': model}
    return out, (O, loss2, case2)


def chamfer_parity(O, loss_default_cpu, case, dev):
    """BASELINE.md section 3's "Chamfer |delta| vs CPU path": the GPU product in fp32 and in bf16 autocast on the SAME
    120 k-point frame pair, weights and masking noise that the cpu_baseline leg fed the oracle (default head), and on
    the conditioned head (decoder_pred x 0.1, the fixtures' choice: with the default head an untrained model's loss
    moves by ~2e-3 between thread counts of the CPU path ITSELF, measured here as `oracle_thread_spread`)."""
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import build_network
    from tmae_amd.train import SyntheticTemporalDataset
    cfg_o, pts, prv, noise = case
    ycfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml'), EasyDict())
    ds = SyntheticTemporalDataset(ycfg.DATA_CONFIG, ycfg.CLASS_NAMES, n_points=1000, batch_size=1)
    bd0 = {'points': torch.from_numpy(pts).to(dev), 'points_prev': torch.from_numpy(prv).to(dev), 'batch_size': 1,
           'mae_noise': torch.from_numpy(noise).to(dev)}

    def gpu_losses(P):
        model = build_network(ycfg.MODEL, len(ycfg.CLASS_NAMES), ds)
        res = model.load_state_dict(P, strict=False)
        assert not res.unexpected_keys
        model.to(dev).train()
        out = []
        for amp in (False, True):
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
                ret, _, _ = model(dict(bd0))
            out.append(float(ret['loss']))
        return out

    def cpu_loss(P, threads):
        torch.set_num_threads(threads)
        with torch.no_grad():
            return float(O.forward_loss(P, pts, prv, noise, 1, cfg_o))

    cores = _host_cores()
    P1 = O.init_params(cfg_o, seed=0)
    g32, g16 = gpu_losses(P1)
    spread = abs(cpu_loss(P1, 1) - loss_default_cpu)
    Pc = O.init_params(cfg_o, seed=0, pred_scale=0.1)
    lc = cpu_loss(Pc, cores)
    c32, c16 = gpu_losses(Pc)
    torch.set_num_threads(cores)
    return {'config': f'1 frame pair of {pts.shape[0]} + {prv.shape[0]} in-range points (seed 0), 3 stages, same weights '
                      f'and masking noise on both sides; forward loss',
            'default_head': {'cpu_loss': round(loss_default_cpu, 6), 'gpu_fp32': round(g32, 6), 'gpu_bf16': round(g16, 6),
                             'chamfer_abs_err_fp32': round(abs(g32 - loss_default_cpu), 7),
                             'chamfer_abs_err_bf16': round(abs(g16 - loss_default_cpu), 7),
                             'oracle_thread_spread': round(spread, 7),
                             'note': f'oracle_thread_spread = |CPU loss at 1 thread - at {cores} threads|'},
            'conditioned_head': {'cpu_loss': round(lc, 6), 'gpu_fp32': round(c32, 6), 'gpu_bf16': round(c16, 6),
                                 'chamfer_abs_err_fp32': round(abs(c32 - lc), 7),
                                 'chamfer_abs_err_bf16': round(abs(c16 - lc), 7), 'pred_scale': 0.1}}


def training_curves(dev, steps=30, points=20000, batch=2, lr_steps=60):
    """Does the bf16 step TRAIN like the fp32 step?  The same `steps` optimizer steps (full 3-stage model, B = `batch`
    pairs of `points`-point scans, the recipe's Adam one-cycle, same initial weights, same batches, same masking noise:
    torch.manual_seed before each run and the masking draws are the only device RNG calls) once in fp32 and once under bf16
    autocast; returns both loss curves.  tests/test_gpu_parity.py::test_bf16_trains_like_fp32 asserts the band."""
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import model_fn_decorator
    from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler,
                                train_one_step)
    cfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml'), EasyDict())
    ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=points, batch_size=batch)
    host = [ds.batch(i) for i in range(4)]
    batches = [{'points': torch.from_numpy(b['points']).to(dev), 'points_prev': torch.from_numpy(b['points_prev']).to(dev),
                'batch_size': b['batch_size']} for b in host]
    curves = {}
    for name, amp in (('fp32', None), ('bf16', torch.bfloat16)):
        torch.manual_seed(1234)
        model = build_model_from_cfg(cfg, ds).to(dev)
        model.train()
        opt = build_optimizer(model, cfg.OPTIMIZATION)
        sched, _ = build_scheduler(opt, lr_steps, 1, -1, cfg.OPTIMIZATION)
        mf = model_fn_decorator()
        torch.manual_seed(99)                                           # the masking noise of every step
        losses = [train_one_step(model, opt, sched, dict(batches[i % 4]), i, mf, amp_dtype=amp)[0].detach() for i in range(steps)]
        curves[name] = [round(float(v), 5) for v in torch.stack(losses).cpu()]
    a, b = np.array(curves['fp32']), np.array(curves['bf16'])
    return {'config': f'{steps} optimizer steps, B = {batch} x {points}-pt pairs (4 batches cycled), full model, Adam one-cycle '
                      f'over {lr_steps} steps, same init / batches / masking noise', 'fp32': curves['fp32'], 'bf16': curves['bf16'],
            'max_rel_gap': round(float(np.max(np.abs(a - b) / np.abs(a))), 5),
            'fp32_first5_last5': [round(float(a[:5].mean()), 5), round(float(a[-5:].mean()), 5)],
            'bf16_first5_last5': [round(float(b[:5].mean()), 5), round(float(b[-5:].mean()), 5)]}


def step_flops(model, batch, amp):
    """Algorithmic FLOPs of ONE training step on `batch` from SURVEY.md 8(d)'s formulas, evaluated on the batch's REAL
    token / window / kernel-pair counts (one extra no-grad forward pass collects the index sets): forward, and 3x that
    for forward + backward."""
    from tmae_amd import ops
    bb = model.backbone_3d
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp is not None):
        bd = dict(batch)
        model(bd)
    B = int(bd['batch_size'])
    drop = {0: dict(max_tokens=16, drop_range=(0, 16)), 1: dict(max_tokens=32, drop_range=(16, 32)),
            2: dict(max_tokens=64, drop_range=(32, 100000))}

    def win_counts(ind, grid, other, ny, nx, shift, nb):
        wb = ops.window_bucket(ind, grid, other, nb, ny, nx, [8, 8, 1], shift, drop)
        return wb['batch_win_inds']

    enc = wca = 0.0
    cfgs = [c.ENCODER for c in bb.model_cfg.SST_BLOCK_LIST]
    d_prev = bb.sst_blocks[0].encoder_blocks[0].encoder_list[0].linear1.in_features
    detail = []
    for si, (e, (ind_p, ind_c, shape)) in enumerate(zip(cfgs, bb.last_stage_indices)):
        d, dff, nblk = e.D_MODEL, e.DIM_FEEDFORWARD, e.NUM_BLOCKS
        ny, nx = shape
        nwin = B * 64 * 64 * 4 + 64
        stage = {}
        for tag, ind in (('prev', ind_p), ('cur', ind_c)):
            M = ind.shape[0]
            grid = ops.index_grid(ind, B, ny, nx)
            t2 = []
            for shift in (False, True):
                cnt = torch.bincount(win_counts(ind, grid, None, ny, nx, shift, B), minlength=1).double()
                t2.append(float((cnt * cnt).sum()))
            layers = 2 * nblk                                           # alternating shift 0 / shift 1
            f = layers * M * 2 * (4 * d * d + 2 * d * dff) + nblk * sum(4 * t * d for t in t2)
            p_subm = float((ops.spconv_neighbors(ind, grid, B, ny, nx, 1) >= 0).sum())
            f += 2 * p_subm * d * d
            if e.STRIDE > 1:
                ind_in = bb.last_stage_indices[si - 1][0 if tag == 'prev' else 1]
                iy, ix = bb.last_stage_indices[si - 1][2]
                gin = ops.index_grid(ind_in, B, iy, ix)
                p_down = float((ops.spconv_neighbors(ind, gin, B, iy, ix, 2) >= 0).sum())
                f += 2 * p_down * d_prev * d
            enc += f
            stage[tag + '_tokens'] = int(M)
        # window cross-attention: queries = current tokens, keys = previous-frame tokens of the same window
        gp, gc = ops.index_grid(ind_p, B, ny, nx), ops.index_grid(ind_c, B, ny, nx)
        m_c = ind_c.shape[0]
        w = 2 * (m_c * 2 * (2 * d * d + 2 * d * dff))
        for shift in (False, True):
            wq = win_counts(ind_c, gc, None, ny, nx, shift, B)
            wk = win_counts(ind_p, gp, None, ny, nx, shift, B)
            n = int(max(int(wq.max()), int(wk.max()))) + 1
            cq, ck = torch.bincount(wq, minlength=n).double(), torch.bincount(wk, minlength=n).double()
            kept_prev = float(ck[cq > 0].sum())
            w += kept_prev * 2 * (2 * d * d) + float((cq * ck).sum()) * 4 * d
        w += 2 * float((ops.spconv_neighbors(ind_c, gc, B, ny, nx, 1) >= 0).sum()) * d * d
        wca += w
        d_prev = d
        detail.append(stage)
    dense = B * 229.8e9
    n_pts = bd['points'].shape[0] + bd['points_prev'].shape[0] if 'points' in bd else 0
    vfe = 2.0 * n_pts * 8832
    M_all = bd['voxel_coords'].shape[0]
    head = 2.0 * M_all * 128 * 48 + 8.0 * M_all * 16 * 64
    fwd = enc + wca + dense + vfe + head
    return {'forward': fwd, 'step': 3.0 * fwd, 'encoder': enc, 'wca': wca, 'dense_decoder': dense, 'vfe': vfe,
            'head_chamfer': head, 'stages': detail}


def secondary_runs(args, log):
    """Short runs of the two other measured configurations of BASELINE.json -- configs[4] (fine-tune step, CenterPoint head)
    and configs[3] (Waymo-shaped pre-training input, 180 k points, 5 point features) -- as CHILD processes of this one after
    its own measurement is complete (3 warm-up + 8 timed steps each, same barrier / synchronize bracket): reported next to
    the headline line, never part of `value`."""
    import subprocess
    out = {}
    for name, extra in (('finetune_configs4', ['--task', 'finetune']),
                        ('waymo_shape_configs3', ['--shape', 'waymo', '--points', '180000'])):
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', '8', '--warmup', '3', '--no-cpu-baseline',
               '--no-secondary', '--batch-per-gpu', str(args.batch_per_gpu), '--dtype', args.dtype] + extra
        log(f'secondary run: {" ".join(extra)} ...')
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out[name] = {'ms_per_step': d['ms_per_step'], 'value': d['value'], 'unit': d['unit'], 'steps': d['steps'],
                         'workload': d['config']['workload'], 'final_loss': d['config']['final_loss']}
        except Exception as e:                      # a failed side run must not cost the headline line
            out[name] = {'error': f'{type(e).__name__}: {e}'[:300]}
    return out


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes through torch.distributed.run BEFORE this
    process touches the GPU (nothing here has made a HIP call yet), relay their output and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.run(cmd, env=env).returncode


def _affinity_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_tmae_affinity', os.path.join(ROOT, 't-mae_amd', 'tmae_amd', 'train', 'affinity.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(_self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # TEST-ONLY (tests/test_gpu_ddp.py): TMAE_BENCH_SHARED_GPU=1 puts every rank on GPU 0 and swaps RCCL for gloo (two
    # ranks cannot share one GPU under RCCL), so that the launcher, the rank bookkeeping and the JSON relay of an N > 1
    # run can be rehearsed on a one-GPU box.  Such a line says so in `collective_backend` and is not a measurement.
    shared_gpu = world > 1 and os.environ.get('TMAE_BENCH_SHARED_GPU') == '1'
    # Host cores of this rank: a disjoint slice of the node's allowed cores, next to the rank's GPU where sysfs shows that, set
    # BEFORE the first GPU call so that the HIP runtime's threads inherit it (tmae_amd/train/affinity.py, loaded by path: importing
    # the package loads the HIP library).  One rank on the node: nothing is pinned.
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    pin = _affinity_module().pin_rank(local_rank, local_world, device_indices=[0] * local_world if shared_gpu else None)
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        dist.init_process_group('gloo' if shared_gpu else 'nccl')          # nccl = RCCL over xGMI
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import model_fn_decorator
    from tmae_amd.train import (SyntheticTemporalDataset, build_model_from_cfg, build_optimizer, build_scheduler,
                                train_one_step, wrap_ddp)
    yaml_name = 't_mae_ssl.yaml' if args.task == 'pretrain' else 't_mae.yaml'
    cfg = cfg_from_yaml_file(os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', yaml_name), EasyDict())
    npf = 5
    if args.shape == 'waymo':
        cfg.DATA_CONFIG.POINT_CLOUD_RANGE = [-74.88, -74.88, -2.0, 74.88, 74.88, 4.0]
        for p_ in cfg.DATA_CONFIG.DATA_PROCESSOR:
            if p_.NAME == 'calculate_grid_size':
                p_.VOXEL_SIZE = [0.32, 0.32, 6.0]
        npf = 6
    ds = SyntheticTemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, n_points=args.points,
                                  batch_size=args.batch_per_gpu, rank=rank, num_point_features=npf,
                                  n_boxes=40 if args.task == 'finetune' else 0)
    torch.manual_seed(0)
    model = build_model_from_cfg(cfg, ds).to(dev)
    model.train()
    if args.skip_unread_gradients:
        cfg.OPTIMIZATION.SKIP_UNREAD_GRADIENTS = True
    opt = build_optimizer(model, cfg.OPTIMIZATION)      # before the DDP wrap: it fixes the set of reduced parameters
    ddp = wrap_ddp(model, dev_index)
    sched, _ = build_scheduler(opt, 1000, cfg.OPTIMIZATION.NUM_EPOCHS, -1, cfg.OPTIMIZATION)
    model_func = model_fn_decorator()
    amp = torch.bfloat16 if args.dtype == 'bf16' else None

    # synthetic batches, uploaded before the timed region (distinct scans, cycled)
    nb = min(args.steps + args.warmup, 4)
    batches = []
    for i in range(nb):
        b = ds.batch(i)
        batches.append({'points': torch.from_numpy(b['points']).to(dev), 'points_prev': torch.from_numpy(b['points_prev']).to(dev),
                        'batch_size': b['batch_size']})
        if 'gt_boxes' in b:
            batches[-1]['gt_boxes'] = torch.from_numpy(b['gt_boxes']).to(dev)

    # the batch a step will consume is handed to the step before it as `next_batch`: its voxelisation (index work, no gradients)
    # is enqueued between that step's forward and backward, and its counts are on the host when its own step begins
    # (TemporalDynVFE.prefetch).  Every step still voxelises exactly one batch -- the next one.
    upcoming = {}

    def step(i):
        if args.no_vfe_prefetch:
            return train_one_step(ddp, opt, sched, dict(batches[i % nb]), i, model_func, amp_dtype=amp)[0]
        cur = upcoming.pop(i, None) or dict(batches[i % nb])
        nxt = upcoming[i + 1] = dict(batches[(i + 1) % nb])
        return train_one_step(ddp, opt, sched, cur, i, model_func, amp_dtype=amp, next_batch=nxt)[0]

    if args.probe_hbm_only:
        print(json.dumps({'roofline': gelu_gemm_roofline(model, dict(batches[0]), amp),
                          'roofline_token_gemm_plain': token_gemm_roofline(model, dict(batches[0]), amp),
                          'roofline_wgrad': wgrad_roofline(model, dict(batches[0]), amp),
                          'roofline_attention': attention_roofline(model, dict(batches[0]), amp)}), flush=True)
        return
    if args.probe_only:
        print(json.dumps({'box_peaks': box_peaks(dev),
                          'roofline': gelu_gemm_roofline(model, dict(batches[0]), amp),
                          'roofline_token_gemm_plain': token_gemm_roofline(model, dict(batches[0]), amp),
                          'roofline_wgrad': wgrad_roofline(model, dict(batches[0]), amp),
                          'roofline_attention': attention_roofline(model, dict(batches[0]), amp),
                          'roofline_igemm': igemm_roofline(args.batch_per_gpu),
                          'roofline_dense_wgrad': dense_wgrad_roofline(args.batch_per_gpu),
                          **(spconv_rooflines(model) if (step_flops(model, dict(batches[0]), amp) and amp is not None) else {})}), flush=True)
        return

    def log(msg):
        if rank == 0:
            print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)

    log(f'model on {dev}, {nb} synthetic batches resident ({int(batches[0]["points"].shape[0])} + '
        f'{int(batches[0]["points_prev"].shape[0])} points each); warm-up ...')
    for i in range(args.warmup):
        tw = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        log(f'warm-up step {i}: {time.perf_counter() - tw:.2f} s')
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    host_s, cpus = 0.0, set()
    for i in range(args.steps):
        th = time.perf_counter()
        loss = step(args.warmup + i)
        host_s += time.perf_counter() - th            # time inside step(): enqueue + the step's own host syncs (not the final wait)
        cpus.add(_cpu_now())
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats(dev)
    alloc_delta = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'num_sync_all_streams')}
    alloc_delta['reserved_gb'] = round(ms1.get('reserved_bytes.all.current', 0) / 2 ** 30, 2)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    # per-rank time of the rank's own K steps (before the closing barrier): the skew between ranks
    own = torch.tensor([t_own], device=dev, dtype=torch.float64)
    per_rank = [own.clone() for _ in range(world)]
    ranks_seen = 1
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, own)
        ranks_seen = dist.get_world_size()
    per_rank = [float(v.item()) for v in per_rank]
    rank_cores = [pin['cores']]
    if world > 1:
        rank_cores = [None] * world
        dist.all_gather_object(rank_cores, pin['cores'])
    elapsed = float(t.item())
    loss_val = float(loss.detach())
    assert np.isfinite(loss_val), 'non-finite loss in the timed region'

    if rank == 0:
        pairs = args.batch_per_gpu * world * args.steps
        line = {
            'metric': ('frame-pairs/sec T-MAE pretrain, 120k-pt ONCE scans' if args.task == 'pretrain' else
                       'frame-pairs/sec T-MAE fine-tune (CenterPoint head), 120k-pt ONCE scans'),
            'value': round(pairs / elapsed, 4),
            'unit': 'frame-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            # host side of the timed region: wall time spent inside the step calls (launch enqueue + the step's host syncs) and the
            # CPUs the launching thread ran on; ms_per_step close to host_ms_per_step = the host, not the GPU, set the pace
            'host_ms_per_step': round(1e3 * host_s / args.steps, 3), 'host_cpus_seen': sorted(cpus),
            'host_affinity': len(os.sched_getaffinity(0)),
            # cores each rank's process (launcher + the threads it starts) is pinned to; disjoint across the ranks of a node
            'rank_cores': {'source': pin['source'], 'allowed': pin['allowed'],
                           'per_rank': [_affinity_module().format_cpulist(c) for c in rank_cores],
                           'disjoint': all(not (set(a) & set(b)) for i_, a in enumerate(rank_cores) for b in rank_cores[i_ + 1:])},
            # the caching allocator inside the timed region: a device malloc / free there is a host stall with the queue draining
            'allocator_in_timed_region': alloc_delta,
            'config': {'workload': (f'configs[{1 if args.shape == "once" else 3}]: {"ONCE" if args.shape == "once" else "Waymo"}-shape synthetic {args.points}-pt frame-pairs, full 3-stage SST '
                                    f'encoder + temporal cross-attn + decoder + Chamfer, fwd+bwd+Adam one-cycle') if args.task == 'pretrain' else
                                   (f'configs[4]: ONCE-shape synthetic {args.points}-pt frame-pairs with 40 boxes, SiamWCA encoder (both frames '
                                    f'unmasked) + SSTBEVBackbone + CenterHead losses, fwd+bwd+Adam one-cycle'),
                       'batch_per_gpu': args.batch_per_gpu, 'global_batch': args.batch_per_gpu * world,
                       'parallelism': f'dp{world}', 'grid': '468x468x1', 'final_loss': round(loss_val, 5),
                       'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                       **({'variant': 'skip_unread_gradients (not the reference step: see DESIGN.md)'}
                          if args.skip_unread_gradients else {})},
            'ranks': ranks_seen,
            'collective_backend': ('gloo, all ranks on GPU 0: REHEARSAL, not a measurement' if shared_gpu else 'nccl (RCCL)') if world > 1 else None,
            'rank_ms_per_step': {'min': round(1e3 * min(per_rank) / args.steps, 3),
                                 'max': round(1e3 * max(per_rank) / args.steps, 3)},
        }
        if args.task == 'pretrain':
            fl = step_flops(model, dict(batches[0]), amp)
            tfs = fl['step'] * world / (elapsed / args.steps) / 1e12
            line['roofline_step'] = {
                'bound': 'mfma', 'achieved': round(tfs / world, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(tfs / world / MFMA_BF16_PEAK_TFLOPS, 5), 'per': 'GPU',
                'flops_per_step_per_gpu': fl['step'], 'forward_flops': fl['forward'],
                'breakdown_forward': {k: fl[k] for k in ('encoder', 'wca', 'dense_decoder', 'vfe', 'head_chamfer')},
                'stage_tokens': fl['stages'],
                'note': 'algorithmic FLOPs of SURVEY.md 8(d) on the real token / window / kernel-pair counts of batch 0 '
                        '(forward x 3 for forward + backward) / measured step time / dense bf16 MFMA peak'}
        log(f'timed region done: {1e3 * elapsed / args.steps:.1f} ms/step; timing the dominant kernel ...')
        peaks = box_peaks(dev)
        line['box_peaks'] = peaks
        if 'roofline_step' in line:
            _with_measured(line['roofline_step'], peaks)
            sb = _step_bytes() if (args.shape == 'once' and args.points == 120000 and args.batch_per_gpu == 8) else None
            if sb is not None:
                # the step as a byte stream: what the present op decomposition moves, and how long that takes at the rate
                # this box's float4 copy sustains -- the floor of the decomposition, next to the measured step
                line['roofline_step'].update({
                    'hbm_bytes': sb['bytes'], 'hbm_read_bytes': sb['read'], 'hbm_written_bytes': sb['written'],
                    'hbm_gbs_achieved': round(sb['bytes'] / (elapsed / args.steps) / 1e9, 1),
                    'hbm_floor_ms_at_measured_copy_rate': round(sb['bytes'] / (peaks['hbm_copy_gbs'] * 1e9) * 1e3, 2),
                    'hbm_families_gb': sb['families_gb'], 'hbm_source': sb['source']})
        # `roofline` = the instance of the token-GEMM family that carries the most step time today: the dual-store FFN-1 kernel
        # (12 launches x 147 us per step, profiles/round4_z_kernel_stats.md); the plain instance that rounds 3-4 priced runs 5
        # launches / 0.2 ms per step and stays as a secondary entry
        plain = _with_measured(token_gemm_roofline(model, dict(batches[0]), amp), peaks)
        if amp is not None:
            line['roofline'] = _with_measured(gelu_gemm_roofline(model, dict(batches[0]), amp), peaks)
            line['roofline_token_gemm_plain'] = plain
        else:
            line['roofline'] = plain
        # round-1 history: the two kernels that led the profile before this one, still priced the same way
        line['roofline_wgrad'] = _with_measured(wgrad_roofline(model, dict(batches[0]), amp), peaks)
        line['roofline_attention'] = _with_measured(attention_roofline(model, dict(batches[0]), amp), peaks)
        if args.task == 'pretrain' and amp is not None:
            line['roofline_igemm'] = _with_measured(igemm_roofline(args.batch_per_gpu), peaks)
            # the three MFMA-bound stragglers of the per-kernel tables: priced in the line itself
            line['roofline_dense_wgrad'] = _with_measured(dense_wgrad_roofline(args.batch_per_gpu), peaks)
            for key, ent in spconv_rooflines(model).items():
                line[key] = _with_measured(ent, peaks)
        if world == 1 and not args.no_cpu_baseline:
            log('timing the CPU oracle (cpu_baseline: warm-up + 3 iterations at C2 and C1) ...')
            line['cpu_baseline'], (O, loss_cpu, case) = cpu_baseline(args.cpu_points)
            if args.task == 'pretrain' and args.shape == 'once':
                log('Chamfer |delta| of the GPU path (fp32, bf16) vs the CPU path on the same 120k-point pair ...')
                line['parity'] = chamfer_parity(O, loss_cpu, case, dev)
                line['chamfer_abs_err_fp32'] = line['parity']['default_head']['chamfer_abs_err_fp32']
                line['chamfer_abs_err_bf16'] = line['parity']['default_head']['chamfer_abs_err_bf16']
                log('30 training steps in fp32 and in bf16 (same init, batches, masking noise) ...')
                line['parity']['training_curves'] = training_curves(dev)
        if (world == 1 and not args.no_secondary and not args.no_cpu_baseline and args.task == 'pretrain'
                and args.shape == 'once' and not args.skip_unread_gradients):
            line['secondary'] = secondary_runs(args, log)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
