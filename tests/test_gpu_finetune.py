"""GPU (-m gpu): the fine-tune path (BASELINE configs[4]) through the C ABI against the golden
vectors captured from the reference (G1-G3) and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import build_finetune_model, golden, fl

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda:0')


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


def _dense_heat(g):
    hm = np.zeros(tuple(g['heat_shape']), np.float32)
    ix = g['heat_nz_index']
    hm[ix[:, 0], ix[:, 1], ix[:, 2], ix[:, 3]] = g['heat_nz_value']
    return hm


def test_centerhead_targets_golden():
    """tmae_centerhead_targets vs CenterHead.assign_targets of the reference: heat map, slots and masks bit-exact
    (Gaussian radius / centre truncation follow the reference's fp32 operation order); log / sin / cos to 1e-6."""
    from tmae_amd import ops
    g = golden('G1_centerhead_targets')
    cmap = torch.tensor([-1, 0, 1, 2, 3, 4], dtype=torch.int32, device=dev())
    heat, tb, inds, mask = ops.centerhead_targets(cu(g['gt_boxes']), cmap, 5, (468, 468), [-74.88, -74.88, -5.0],
                                                  [0.32, 0.32, 8.0], 1, 500, 0.1, 2)
    assert np.array_equal(inds.cpu().numpy(), g['inds']) and np.array_equal(mask.cpu().numpy(), g['masks'])
    ref = _dense_heat(g)
    got = heat.cpu().numpy()
    assert np.array_equal(got != 0, ref != 0)
    assert np.abs(got - ref).max() <= 6e-8                     # double exp on both sides, rounded to f32: <= 1 ulp
    assert (got != ref).mean() < 1e-4
    np.testing.assert_allclose(tb.cpu().numpy(), g['target_boxes'], atol=1e-6)
    # two heads: classes split, slots are the ranks inside each head
    cm_a = torch.tensor([-1, 0, -1, -1, 1, -1], dtype=torch.int32, device=dev())        # head A = Car, Pedestrian
    ha, tba, ia, ma = ops.centerhead_targets(cu(g['gt_boxes']), cm_a, 2, (468, 468), [-74.88, -74.88, -5.0],
                                             [0.32, 0.32, 8.0], 1, 500, 0.1, 2)
    cls = g['gt_boxes'][:, :, 7]
    for b in range(cls.shape[0]):
        n_a = int(((cls[b] == 1) | (cls[b] == 4)).sum())
        assert int((ia[b, n_a:] != 0).sum()) == 0 and int(ma[b].sum()) <= n_a
    assert torch.equal(ha[:, 0], heat[:, 0]) and torch.equal(ha[:, 1], heat[:, 3])
    # empty label set
    h0, _, _, m0 = ops.centerhead_targets(torch.zeros((2, 0, 8), device=dev()), cmap, 5, (64, 64), [-74.88, -74.88, -5.0],
                                          [0.32, 0.32, 8.0], 1, 500, 0.1, 2)
    assert float(h0.abs().sum()) == 0 and int(m0.sum()) == 0


def test_focal_loss_golden():
    from tmae_amd import ops
    g1, g2 = golden('G1_centerhead_targets'), golden('G2_centerhead_losses')
    gen = torch.Generator().manual_seed(int(g2['pred_seed']))
    p = torch.rand(3, 5, 468, 468, generator=gen) * 0.98 + 0.01
    logits = torch.log(p / (1 - p)).to(dev()).requires_grad_(True)            # sigmoid(logits) = p
    heat = cu(_dense_heat(g1))
    loss = ops.focal_loss_centernet(logits, heat)
    loss.backward()
    assert abs(fl(loss) - float(g2['focal'])) <= 2e-5 * abs(float(g2['focal']))
    # d loss / d p = d loss / d logits / (p (1 - p))
    gp = (logits.grad.cpu() / (p * (1 - p))).flatten()[torch.from_numpy(g2['grad_probe_index'])].numpy()
    np.testing.assert_allclose(gp, g2['grad_probe_hm'], rtol=2e-4, atol=2e-8)
    # no positives: loss = - sum of the negative terms; clamped logits carry no gradient
    x = torch.tensor([[-20.0, 0.3, 25.0]], device=dev(), requires_grad=True)
    t = torch.tensor([[0.0, 0.5, 0.2]], device=dev())
    l0 = ops.focal_loss_centernet(x, t)
    l0.backward()
    pc = torch.clamp(torch.sigmoid(x.detach()), 1e-4, 1 - 1e-4)
    ref = -(torch.log(1 - pc) * pc ** 2 * (1 - t) ** 4).sum()
    assert abs(fl(l0) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert float(x.grad[0, 0]) == 0.0 and float(x.grad[0, 2]) == 0.0 and float(x.grad[0, 1]) != 0.0


def test_finetune_e2e_golden_and_oracle(oracle, ft_oracle):
    """Whole fine-tune training step (TemporalDynVFE -> SiamWCA -> SSTBEVBackbone -> CenterHead loss) in fp32 vs the
    loss captured from the reference and per-parameter gradients vs the CPU oracle."""
    g = golden('G3_finetune_e2e_3stage')
    cfg = ft_oracle.default_finetune_cfg(3)
    P = ft_oracle.init_finetune_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']))
    bs = int(g['batch_size'])
    model, _, _ = build_finetune_model(params=P, device=dev())
    model.train()
    bd = {'points': cu(g['points']), 'points_prev': cu(g['points_prev']), 'batch_size': bs, 'gt_boxes': cu(g['gt_boxes'])}
    ret, tb, _ = model(bd)
    ret['loss'].backward()
    ref = float(g['loss'])
    assert abs(fl(ret['loss']) - ref) <= 1e-4 * max(1.0, abs(ref)), (fl(ret['loss']), ref)
    assert abs(float(tb['hm_loss_head_0']) - float(g['hm_loss'])) <= 1e-4 * max(1.0, float(g['hm_loss']))
    assert abs(float(tb['loc_loss_head_0']) - float(g['loc_loss'])) <= 1e-4 * max(1.0, float(g['loc_loss']))
    sf = bd['spatial_features_2d'].detach().double()
    assert float(sf.sum()) == pytest.approx(float(g['x2d_checksum']), rel=1e-4, abs=1.0)
    assert float(sf.abs().sum()) == pytest.approx(float(g['x2d_abs_checksum']), rel=1e-4)
    grads = dict(model.named_parameters())
    # Gradient bars.  This step is ill-conditioned in fp32: eight dense BatchNorm layers over a BEV map whose inactive
    # region is constant amplify summation-order noise -- the CPU oracle itself moves by up to 3 % in gradient norms
    # and 11 % of the largest entry between 1 and 8 threads (the loss by 5e-5).  The bars below are that noise floor;
    # the kernels' own unit tests (and the pre-training e2e tests, 5e-3) carry the tight gradient checks.
    for n, gn in zip(g['grad_names'], g['grad_norms']):
        assert abs(float(grads[str(n)].grad.norm()) - gn) <= 6e-2 * max(1.0, gn), n
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lo = ft_oracle.finetune_loss(Pg, g['points'], g['points_prev'], g['gt_boxes'], bs, cfg)
    lo.backward()
    for n in g['grad_names']:
        a, b = grads[str(n)].grad.cpu(), Pg[str(n)].grad
        assert (a - b).abs().max().item() <= 0.25 * max(1.0, b.abs().max().item()), n
        cos = torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0)
        assert float(cos) > 0.995 or float(b.norm()) < 1e-3, (n, float(cos))


def test_finetune_gradients_vs_float64_oracle(ft_oracle):
    """The fine-tune step's gradients against the oracle run in FLOAT64 (fixture G3d: per tensor the norm and four seeded
    +-1 projections).  fp32 cannot resolve this step's cancelling sums: the fp32 CPU oracle itself is ~2 % (median) off
    the float64 values, so the yardstick per tensor is the fp32 oracle's own error; the GPU (fp32 kernels, shifted
    BatchNorm sums, fp32 MFMA accumulation in the token-split weight gradients) is held to 4e-2 of each tensor's norm and
    4e-3 in the median -- an order of magnitude inside the fp32 CPU path's own deviation."""
    g, d = golden('G3_finetune_e2e_3stage'), golden('G3d_finetune_grad64')
    cfg = ft_oracle.default_finetune_cfg(3)
    P = ft_oracle.init_finetune_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']))
    model, _, _ = build_finetune_model(params=P, device=dev())
    model.train()
    bd = {'points': cu(g['points']), 'points_prev': cu(g['points_prev']), 'batch_size': int(g['batch_size']),
          'gt_boxes': cu(g['gt_boxes'])}
    ret, _, _ = model(bd)
    ret['loss'].backward()
    assert abs(fl(ret['loss']) - float(d['loss64'])) <= 3 * abs(float(d['loss32']) - float(d['loss64'])) + 1e-4
    grads = dict(model.named_parameters())
    p64, p32 = d['proj64'], d['proj32']
    scale = float(np.median(p64[:, 0]))
    worst, rel, relc = (0.0, None), [], []
    for i, n in enumerate(str(x) for x in d['names']):
        gg = grads[n].grad.detach().double().flatten().cpu()
        gen = torch.Generator().manual_seed(100003 * (i + 1))
        mine = [float(gg.norm())]
        for _ in range(int(d['nproj'])):
            r = torch.randint(0, 2, (gg.numel(),), generator=gen, dtype=torch.int8).double() * 2 - 1
            mine.append(float((gg * r).sum()))
        err = np.abs(np.array(mine) - p64[i]).max()
        cpu = np.abs(p32[i] - p64[i]).max()
        bar = 4e-2 * p64[i, 0] + 1e-4 * scale        # measured: median 1.1e-3, worst 2.8e-2 (a tau gradient) of the norm
        worst = max(worst, (err / bar, n))
        rel.append(err / max(p64[i, 0], 1e-4 * scale))
        relc.append(cpu / max(p64[i, 0], 1e-4 * scale))
        assert err <= bar, (n, err, bar, mine[0], p64[i, 0], p32[i, 0])
    assert np.median(rel) < 4e-3, np.median(rel)           # the fp32 CPU oracle: median 2.2e-2, worst 5.6 (!) of the norm
    print('worst err / bar', worst, '| GPU error / norm: median %.2e max %.2e | CPU fp32 oracle: median %.2e max %.2e'
          % (np.median(rel), np.max(rel), np.median(relc), np.max(relc)))


def test_finetune_bf16_step_runs_and_is_close(ft_oracle):
    g = golden('G3_finetune_e2e_3stage')
    cfg = ft_oracle.default_finetune_cfg(3)
    P = ft_oracle.init_finetune_params(cfg, seed=int(g['param_seed']), tau=float(g['tau']))
    model, _, _ = build_finetune_model(params=P, device=dev())
    model.train()
    bd = {'points': cu(g['points']), 'points_prev': cu(g['points_prev']), 'batch_size': int(g['batch_size']),
          'gt_boxes': cu(g['gt_boxes'])}
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ret, _, _ = model(bd)
    ret['loss'].backward()
    assert torch.isfinite(ret['loss'])
    rel = abs(float(ret['loss'].detach()) - float(g['loss'])) / abs(float(g['loss']))
    print(f'finetune bf16 loss vs fp32 reference: rel {rel:.2e}')
    assert rel <= 0.01                                   # measured 3.9e-3 (round 4); the bar was 5 % until round 3
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def _rand_boxes(rng, n, spread=20.0):
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.normal(-1, 0.5, n)
    b[:, 3:6] = rng.uniform(0.5, 6.0, (n, 3))
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def test_head_conv_bias_gradient_vs_torch():
    """ops.conv3x3_channel_bias (CenterHead's last convs, center_head.py:11-45: the library's forward and input / weight
    gradients, the bias gradient as a column sum over the contiguous [B*Y*X, k] view) against nn.Conv2d itself; k = 1, 2, 3, 5;
    a map whose size does not fit the periodic view (falls back to torch's sum)."""
    from tmae_amd import ops
    torch.manual_seed(5)
    for k, (B, Y, X) in ((1, (2, 32, 32)), (2, (2, 32, 32)), (3, (2, 64, 32)), (5, (4, 64, 64)), (3, (1, 7, 5))):
        conv = torch.nn.Conv2d(64, k, 3, padding=1, bias=True).cuda()
        x = torch.randn(B, 64, Y, X, device='cuda').contiguous(memory_format=torch.channels_last)
        go = torch.randn(B, k, Y, X, device='cuda').contiguous(memory_format=torch.channels_last)
        res = []
        for fn in (lambda t: ops.conv3x3_channel_bias(t, conv), conv):
            xa = x.clone().requires_grad_(True)
            conv.zero_grad()
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = fn(xa)
            y.backward(go.to(y.dtype))
            res.append((y.detach().float(), xa.grad.float(), conv.weight.grad.clone(), conv.bias.grad.clone()))
        (y1, dx1, dw1, db1), (y2, dx2, dw2, db2) = res
        assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
        assert (dw1 - dw2).abs().max().item() <= 1e-2 * max(1.0, float(dw2.abs().max()))      # a bf16 weight gradient: one ulp
        # torch sums the bias gradient in bf16 steps; ours in fp32: compare both with the fp32 sum
        want = go.to(torch.bfloat16).float().sum((0, 2, 3))
        assert (db1.float() - want).abs().max().item() <= 2e-2 * max(1.0, float(want.abs().max())), (k, db1, want)


def test_narrow_head_conv_kernels_vs_torch():
    """csrc/headconv.hip -- Conv2d(64, k <= 8, 3, padding=1, bias=True) forward, input gradient and weight gradient on a
    channels-last bf16 map (CenterHead's last convs, center_head.py:11-45) -- against torch's fp32 convolution of the same
    bf16-representable values: maps that are not multiples of the 16 x 16 block, one smaller than a block, k = 1 .. 8, a map
    large enough for several blocks per persistent workgroup; a 64-channel slice of a wider map through the C ABI; refusals."""
    from tmae_amd import ops
    from tmae_amd._lib import lib
    torch.manual_seed(6)
    for k, (B, Y, X) in ((1, (2, 32, 32)), (2, (1, 16, 16)), (3, (2, 40, 23)), (5, (3, 50, 70)), (8, (1, 7, 5)), (4, (1, 3, 90)),
                         (3, (4, 468, 468))):
        conv = torch.nn.Conv2d(64, k, 3, padding=1, bias=True).cuda()
        with torch.no_grad():
            conv.weight.copy_(conv.weight.bfloat16().float())
            conv.bias.copy_(torch.randn(k, device='cuda'))
        x = torch.randn(B, Y, X, 64, device='cuda').bfloat16().permute(0, 3, 1, 2)          # channels-last memory
        go = torch.randn(B, Y, X, k, device='cuda').bfloat16().permute(0, 3, 1, 2)
        assert ops.narrow_conv3x3_ok(x, conv)
        xa = x.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = ops.conv3x3_channel_bias(xa, conv)
        assert y.dtype == torch.bfloat16 and y.shape == (B, k, Y, X) and y.permute(0, 2, 3, 1).is_contiguous()
        y.backward(go)
        got = (y.detach().float(), xa.grad.float(), conv.weight.grad.clone(), conv.bias.grad.clone())
        conv.zero_grad()
        xf = x.float().requires_grad_(True)
        yr = conv(xf)
        yr.backward(go.float())
        want = (yr.detach(), xf.grad, conv.weight.grad.clone(), conv.bias.grad.clone())
        conv.zero_grad()
        # outputs rounded once to bf16: half an ulp of the value (+ fp32 summation order)
        for a, b, what in ((got[0], want[0], 'y'), (got[1], want[1], 'dx')):
            err = (a - b).abs()
            assert bool((err <= 2.0 ** -8 * b.abs() + 1e-5 * float(b.abs().max())).all()), (k, B, Y, X, what, float(err.max()))
        scale = float(want[2].abs().max())
        assert float((got[2] - want[2]).abs().max()) <= 2e-4 * max(1.0, scale), (k, B, Y, X, 'dw', scale)      # fp32 both sides
        assert float((got[3] - want[3]).abs().max()) <= 2e-3 * max(1.0, float(want[3].abs().max())), (k, 'db')

    # a 64-channel slice (channels 64 .. 127) of a 128-channel map: channel pitch 128
    k, B, Y, X = 3, 2, 33, 20
    wide = torch.randn(B, Y, X, 128, device='cuda').bfloat16()
    w = (torch.randn(k, 9 * 64, device='cuda') * 0.1).bfloat16()
    bias = torch.randn(k, device='cuda')
    out = torch.empty(B, Y, X, k, device='cuda', dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    sl = wide[..., 64:]
    assert lib.tmae_conv3x3_c64_narrow_fwd(sl.data_ptr(), 128, B, Y, X, w.data_ptr(), bias.data_ptr(), k, out.data_ptr(), st) == 0
    ref = torch.nn.functional.conv2d(sl.float().permute(0, 3, 1, 2), w.float().view(k, 3, 3, 64).permute(0, 3, 1, 2), bias, padding=1)
    assert float((out.float().permute(0, 3, 1, 2) - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-5
    dyr = torch.randn(B, Y, X, k, device='cuda').bfloat16()
    dwide = torch.full((B, Y, X, 128), 7.0, device='cuda', dtype=torch.bfloat16)
    wsb = lib.tmae_conv3x3_c64_narrow_bwd_data_workspace()
    ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
    assert lib.tmae_conv3x3_c64_narrow_bwd_data(dyr.data_ptr(), B, Y, X, k, w.data_ptr(), dwide[..., 64:].data_ptr(), 128,
                                                ws.data_ptr(), wsb, st) == 0
    xs = sl.float().permute(0, 3, 1, 2).requires_grad_(True)
    torch.nn.functional.conv2d(xs, w.float().view(k, 3, 3, 64).permute(0, 3, 1, 2), None, padding=1).backward(
        dyr.float().permute(0, 3, 1, 2))
    assert float((dwide[..., 64:].float().permute(0, 3, 1, 2) - xs.grad).abs().max()) <= 2.0 ** -8 * float(xs.grad.abs().max()) + 1e-5
    assert bool((dwide[..., :64] == 7.0).all())                                  # the other half of the wide map is untouched
    wsb = lib.tmae_conv3x3_c64_narrow_wgrad_workspace(k)
    ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
    dw = torch.empty(k, 9 * 64, device='cuda')
    assert lib.tmae_conv3x3_c64_narrow_wgrad(dyr.data_ptr(), sl.data_ptr(), 128, B, Y, X, k, dw.data_ptr(), ws.data_ptr(), wsb, st) == 0
    wp = w.float().view(k, 3, 3, 64).permute(0, 3, 1, 2).clone().requires_grad_(True)
    wref = torch.autograd.grad(torch.nn.functional.conv2d(sl.float().permute(0, 3, 1, 2), wp, None, padding=1), wp,
                               dyr.float().permute(0, 3, 1, 2))[0]
    assert float((dw.view(k, 3, 3, 64).permute(0, 3, 1, 2) - wref).abs().max()) <= 2e-4 * max(1.0, float(wref.abs().max()))
    # refusals: k out of range, a pitch below 64 / not a multiple of 8, a missing pointer, a short workspace
    assert lib.tmae_conv3x3_c64_narrow_fwd(sl.data_ptr(), 128, B, Y, X, w.data_ptr(), bias.data_ptr(), 9, out.data_ptr(), st) < 0
    assert lib.tmae_conv3x3_c64_narrow_fwd(sl.data_ptr(), 60, B, Y, X, w.data_ptr(), bias.data_ptr(), k, out.data_ptr(), st) < 0
    assert lib.tmae_conv3x3_c64_narrow_fwd(sl.data_ptr(), 128, B, Y, X, w.data_ptr(), None, k, out.data_ptr(), st) < 0
    assert lib.tmae_conv3x3_c64_narrow_wgrad(dyr.data_ptr(), sl.data_ptr(), 128, B, Y, X, k, dw.data_ptr(), ws.data_ptr(), 64, st) < 0
    assert lib.tmae_conv3x3_c64_narrow_wgrad_workspace(0) == 0 and lib.tmae_conv3x3_c64_narrow_wgrad_workspace(9) == 0
    torch.cuda.synchronize()


def test_branch_stems_chain_their_input_gradients():
    """SeparateHead (center_head.py:11-45): the branches' 64 -> 64 stem convs read one shared map.  Each stem hands the next branch
    an alias of its input (ops.conv3x3_c64(chain=True)), so a branch's input gradient accumulates into the later branches' inside
    the conv kernel (accumulate form) instead of autograd adding one [B, Y, X, 64] tensor per branch.  Against the same stems reading
    the map independently: identical outputs and weight gradients, the map's gradient equal up to the bf16 roundings the separate
    adds make (the chained sum is rounded once per branch as well: same count, different order)."""
    import torch.nn as nn
    from tmae_amd.modules.bev_backbone import conv_bn_relu_nhwc
    torch.manual_seed(21)
    B, Y, X, nb = 2, 52, 37, 4
    seqs = [nn.Sequential(nn.Conv2d(64, 64, 3, padding=1, bias=True), nn.BatchNorm2d(64, eps=1e-3, momentum=0.01),
                          nn.ReLU(inplace=True)).cuda().train() for _ in range(nb)]
    x0 = torch.randn(B, Y, X, 64, device='cuda').bfloat16().permute(0, 3, 1, 2)
    gos = [torch.randn(B, Y, X, 64, device='cuda').bfloat16().permute(0, 3, 1, 2) for _ in range(nb)]

    def run(chain):
        for q in seqs:
            q.zero_grad()
            q[1].running_mean.zero_(); q[1].running_var.fill_(1.0)
        xa = x0.clone().requires_grad_(True)
        src, outs = xa, []
        with torch.autocast('cuda', dtype=torch.bfloat16):
            for q in seqs:
                if chain:
                    y, src = conv_bn_relu_nhwc(q, src, chain=True)
                    assert src is not xa                        # the 64 -> 64 kernel took the conv: the alias is handed on
                else:
                    y = conv_bn_relu_nhwc(q, xa)
                outs.append(y)
        torch.autograd.backward(outs, gos)
        return [o.detach() for o in outs], xa.grad.float(), [p.grad.clone() for q in seqs for p in q.parameters()]
    o1, dx1, g1 = run(True)
    o0, dx0, g0 = run(False)
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)
    assert float((dx1 - dx0).abs().max()) <= 2.0 ** -6 * float(dx0.abs().max())
    assert float((dx1 - dx0).norm() / dx0.norm()) < 4e-3


def test_c64_stem_conv_kernels_vs_torch():
    """csrc/headconv.hip, 64 -> 64: Conv2d(64, 64, 3, padding=1, bias=False) forward, input gradient (the same kernel with flipped,
    transposed weights) and weight gradient on channels-last bf16 maps (CenterHead's stem convs, center_head.py:28-31) against
    torch's fp32 convolution of the same values; ragged maps, a map with several blocks per persistent workgroup; through
    conv_bn_relu_nhwc (the module path) against the library path; channel pitches above 64 through the C ABI; refusals."""
    import torch.nn as nn
    from tmae_amd import ops
    from tmae_amd._lib import lib
    from tmae_amd.modules.bev_backbone import conv_bn_relu_nhwc
    torch.manual_seed(8)
    for (B, Y, X) in ((1, 16, 16), (2, 40, 23), (1, 7, 5), (3, 50, 70), (2, 468, 468)):
        conv = nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda()
        with torch.no_grad():
            conv.weight.copy_(conv.weight.bfloat16().float())
        x = torch.randn(B, Y, X, 64, device='cuda').bfloat16()
        go = torch.randn(B, Y, X, 64, device='cuda').bfloat16()
        assert ops.conv3x3_c64_ok(x, conv)
        xa = x.clone().requires_grad_(True)
        y = ops.conv3x3_c64(xa, conv.weight)
        y.backward(go)
        got = (y.detach().float(), xa.grad.float(), conv.weight.grad.clone())
        conv.zero_grad()
        xf = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        yr = conv(xf)
        yr.backward(go.float().permute(0, 3, 1, 2))
        want = (yr.detach().permute(0, 2, 3, 1), xf.grad.permute(0, 2, 3, 1), conv.weight.grad.clone())
        conv.zero_grad()
        for a, b, what in ((got[0], want[0], 'y'), (got[1], want[1], 'dx')):
            err = (a - b).abs()
            assert bool((err <= 2.0 ** -8 * b.abs() + 1e-5 * float(b.abs().max())).all()), (B, Y, X, what, float(err.max()))
        assert float((got[2] - want[2]).abs().max()) <= 2e-4 * max(1.0, float(want[2].abs().max())), (B, Y, X, 'dw')

    # the module path (conv + training-mode BatchNorm + ReLU) against the library conv in front of the same fused norm: the 64 -> 64
    # stems and the 128 -> 64 shared conv (center_head.py:85-89: two 64-channel halves, the second call accumulates)
    import os
    for cin in (64, 128):
        seq = nn.Sequential(nn.Conv2d(cin, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64, eps=1e-3, momentum=0.01),
                            nn.ReLU(inplace=True)).cuda().train()
        x0 = torch.randn(2, 60, 44, cin, device='cuda').bfloat16().permute(0, 3, 1, 2)
        go = torch.randn(2, 60, 44, 64, device='cuda').bfloat16().permute(0, 3, 1, 2)
        assert (ops.conv3x3_c64_ok if cin == 64 else ops.conv3x3_c128to64_ok)(x0.permute(0, 2, 3, 1), seq[0])
        res = {}
        for mode in ('native', 'library'):
            os.environ['TMAE_HEAD_CONV'] = mode
            try:
                seq.zero_grad()
                seq[1].running_mean.zero_(); seq[1].running_var.fill_(1.0)
                xa = x0.clone().requires_grad_(True)
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    y = conv_bn_relu_nhwc(seq, xa)
                y.backward(go)
                res[mode] = (y.detach().float(), xa.grad.float(), seq[0].weight.grad.clone(), seq[1].weight.grad.clone())
            finally:
                os.environ.pop('TMAE_HEAD_CONV', None)
        for a, b, tol in zip(res['native'], res['library'], (2e-2, 3e-2, 3e-2, 3e-2)):
            assert float((a - b).norm() / (b.norm() + 1e-12)) < tol, cin
    # 128 -> 64 against torch's fp32 convolution of the same values (forward: one extra bf16 rounding of the first half's partial sum)
    conv = nn.Conv2d(128, 64, 3, padding=1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(conv.weight.bfloat16().float())
    x = torch.randn(2, 37, 50, 128, device='cuda').bfloat16()
    go = torch.randn(2, 37, 50, 64, device='cuda').bfloat16()
    xa = x.clone().requires_grad_(True)
    y = ops.conv3x3_c128to64(xa, conv.weight)
    y.backward(go)
    got = (y.detach().float(), xa.grad.float(), conv.weight.grad.clone())
    conv.zero_grad()
    xf = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = conv(xf)
    yr.backward(go.float().permute(0, 3, 1, 2))
    want = (yr.detach().permute(0, 2, 3, 1), xf.grad.permute(0, 2, 3, 1), conv.weight.grad.clone())
    conv.zero_grad()
    assert float((got[0] - want[0]).abs().max()) <= 2.0 ** -7 * float(want[0].abs().max())
    err = (got[1] - want[1]).abs()
    assert bool((err <= 2.0 ** -8 * want[1].abs() + 1e-5 * float(want[1].abs().max())).all())
    assert float((got[2] - want[2]).abs().max()) <= 2e-4 * max(1.0, float(want[2].abs().max()))

    # channel pitches: input = channels 64 .. 127 of a 128-wide map, output into channels 0 .. 63 of a 192-wide one
    B, Y, X = 2, 33, 20
    wide = torch.randn(B, Y, X, 128, device='cuda').bfloat16()
    w = (torch.randn(64, 9 * 64, device='cuda') * 0.05).bfloat16()
    owide = torch.full((B, Y, X, 192), 3.0, device='cuda', dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    wsb = lib.tmae_conv3x3_c64_workspace()
    ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
    sl = wide[..., 64:]
    w4 = w.float().view(64, 3, 3, 64).permute(0, 3, 1, 2)
    for inp_grad in (0, 1):
        assert lib.tmae_conv3x3_c64(sl.data_ptr(), 128, B, Y, X, w.data_ptr(), inp_grad, 0, owide.data_ptr(), 192, ws.data_ptr(), wsb, st) == 0
        if inp_grad == 0:
            ref = torch.nn.functional.conv2d(sl.float().permute(0, 3, 1, 2), w4, padding=1)
        else:
            ref = torch.nn.functional.conv_transpose2d(sl.float().permute(0, 3, 1, 2), w4, padding=1)
        assert float((owide[..., :64].float().permute(0, 3, 1, 2) - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-5
        assert bool((owide[..., 64:] == 3.0).all())
    dy = torch.randn(B, Y, X, 192, device='cuda').bfloat16()
    wsb2 = lib.tmae_conv3x3_c64_wgrad_workspace()
    ws2 = torch.empty(wsb2, dtype=torch.uint8, device='cuda')
    dw = torch.empty(64, 9 * 64, device='cuda')
    assert lib.tmae_conv3x3_c64_wgrad(dy[..., 128:].data_ptr(), 192, sl.data_ptr(), 128, B, Y, X, dw.data_ptr(), ws2.data_ptr(), wsb2, st) == 0
    wp = w4.clone().requires_grad_(True)
    wref = torch.autograd.grad(torch.nn.functional.conv2d(sl.float().permute(0, 3, 1, 2), wp, padding=1), wp,
                               dy[..., 128:].float().permute(0, 3, 1, 2))[0]
    assert float((dw.view(64, 3, 3, 64).permute(0, 3, 1, 2) - wref).abs().max()) <= 2e-4 * max(1.0, float(wref.abs().max()))
    assert lib.tmae_conv3x3_c64(sl.data_ptr(), 60, B, Y, X, w.data_ptr(), 0, 0, owide.data_ptr(), 192, ws.data_ptr(), wsb, st) < 0
    assert lib.tmae_conv3x3_c64(sl.data_ptr(), 128, B, Y, X, w.data_ptr(), 2, 0, owide.data_ptr(), 192, ws.data_ptr(), wsb, st) < 0
    assert lib.tmae_conv3x3_c64(sl.data_ptr(), 128, B, Y, X, w.data_ptr(), 0, 3, owide.data_ptr(), 192, ws.data_ptr(), wsb, st) < 0
    assert lib.tmae_conv3x3_c64(sl.data_ptr(), 128, B, Y, X, w.data_ptr(), 0, 0, owide.data_ptr(), 192, ws.data_ptr(), 64, st) < 0
    assert lib.tmae_conv3x3_c64_wgrad(dy.data_ptr(), 192, sl.data_ptr(), 128, B, Y, X, dw.data_ptr(), None, wsb2, st) < 0
    torch.cuda.synchronize()


@pytest.mark.parametrize('dil', [1, 2])
def test_bev_block_shortcut_fused_vs_separate_adds(dil, monkeypatch):
    """`out = conv_bn_relu(out) + out` of SSTBEVBackbone (sst_bev_backbone.py:35-41) with the shortcut added inside the norm's apply
    kernel (forward) and inside the conv's input-gradient kernel (backward: tmae_bn_relu_add_fwd, tmae_dense_conv3x3_add) against
    the same block with the two adds as separate elementwise passes, and against torch in fp32."""
    import torch.nn as nn
    from tmae_amd.modules.bev_backbone import conv_bn_relu_nhwc
    torch.manual_seed(9)
    B, C, Y, X = 2, 128, 70, 52
    seq = nn.Sequential(nn.Conv2d(C, C, 3, padding=dil, dilation=dil, bias=False), nn.BatchNorm2d(C, eps=1e-3, momentum=0.01),
                        nn.ReLU(inplace=True)).cuda().train()
    x0 = torch.randn(B, Y, X, C, device='cuda').bfloat16().permute(0, 3, 1, 2)           # channels-last memory
    go = torch.randn(B, Y, X, C, device='cuda').bfloat16().permute(0, 3, 1, 2)
    res = {}
    for mode in ('fused', 'add'):
        monkeypatch.setenv('TMAE_BEV_SHORTCUT', mode)
        seq.zero_grad()
        seq[1].running_mean.zero_(); seq[1].running_var.fill_(1.0)
        x = x0.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = conv_bn_relu_nhwc(seq, x, shortcut=True)
        y.backward(go)
        res[mode] = (y.detach().float(), x.grad.float(), seq[0].weight.grad.clone(), seq[1].weight.grad.clone(),
                     seq[1].bias.grad.clone(), seq[1].running_mean.clone(), seq[1].running_var.clone())
    f, a = res['fused'], res['add']
    assert torch.equal(f[5], a[5]) and torch.equal(f[6], a[6])                              # same statistics
    # one rounding (fused) against two (separate adds): within one bf16 ulp of the larger operand
    assert (f[0] - a[0]).abs().max().item() <= 2.0 ** -7 * float(a[0].abs().max())
    assert (f[1] - a[1]).abs().max().item() <= 2.0 ** -7 * float(a[1].abs().max())
    for i in (2, 3, 4):
        assert torch.equal(f[i], a[i])                                                      # the branch's own gradients: same inputs
    # fp32 torch reference of the block
    xr = x0.float().clone().requires_grad_(True)
    ref = nn.Sequential(nn.Conv2d(C, C, 3, padding=dil, dilation=dil, bias=False), nn.BatchNorm2d(C, eps=1e-3, momentum=0.01),
                        nn.ReLU()).cuda().train()
    ref[0].weight.data.copy_(seq[0].weight.data.bfloat16().float())
    yr = ref(xr) + xr
    yr.backward(go.float())
    assert ((f[0] - yr.detach()).norm() / yr.detach().norm()).item() < 1e-2
    assert ((f[1] - xr.grad).norm() / xr.grad.norm()).item() < 2e-2


def test_rotated_iou_and_nms_vs_oracle(ft_oracle):
    """csrc/iou3d_nms.hip vs the float64 convex-clipping restatement: BEV overlap / IoU, 3-D IoU, edge cases
    (identical, disjoint, contained, 45 degrees, touching), and the kept set + order of the rotated NMS."""
    from tmae_amd import ops
    rng = np.random.default_rng(5)
    a, b = _rand_boxes(rng, 60), _rand_boxes(rng, 45)
    special = np.array([[0, 0, 0, 4, 2, 1, 0.0], [0, 0, 0, 4, 2, 1, 0.0], [0, 0, 0, 2, 2, 1, np.pi / 4], [0, 0, 0, 2, 2, 1, 0.0],
                        [0.5, 0.2, 0, 1, 0.5, 1, 0.3], [100, 100, 0, 4, 2, 1, 1.0], [4, 0, 0, 4, 2, 1, 0.0],
                        [0, 0, 0.6, 4, 2, 1, 0.0]], np.float32)
    a = np.concatenate([special, a])
    b = np.concatenate([special, b])
    ov = ops.boxes_overlap_bev(cu(a), cu(b)).cpu().numpy()
    iou = ops.boxes_iou_bev(cu(a), cu(b)).cpu().numpy()
    i3 = ops.boxes_iou3d_gpu(cu(a), cu(b)).cpu().numpy()
    ref_ov = np.array([[ft_oracle.overlap_bev(x, y) for y in b] for x in a])
    ref_iou = np.array([[ft_oracle.iou_bev(x, y) for y in b] for x in a])
    assert np.abs(ov - ref_ov).max() <= 2e-4 * max(1.0, ref_ov.max())
    assert np.abs(iou - ref_iou).max() <= 2e-5 + 1e-4
    assert np.abs(i3 - ft_oracle.iou3d(a, b)).max() <= 2e-4
    assert abs(iou[0, 1] - 1.0) < 1e-5 and iou[0, 5] == 0.0 and abs(ov[2, 3] - 8 * (np.sqrt(2) - 1)) < 1e-4
    assert abs(ov[0, 4] - 0.5) < 1e-5                                  # contained box: its own area
    # NMS: clustered boxes so that many suppressions happen; no IoU sits within 1e-3 of the threshold
    n = 700
    centers = rng.uniform(-30, 30, (40, 2))
    bx = _rand_boxes(rng, n)
    bx[:, 0:2] = centers[rng.integers(0, 40, n)] + rng.normal(0, 0.8, (n, 2))
    sc = rng.random(n).astype(np.float32)
    full = np.array([[ft_oracle.iou_bev(x, y) for y in bx[:200]] for x in bx[:200]])
    assert (np.abs(full - 0.5) < 1e-3).sum() == 0
    keep, _ = ops.nms_gpu(cu(bx), cu(sc), 0.5)
    ref_keep = ft_oracle.nms_bev(bx, sc, 0.5)
    assert np.array_equal(keep.cpu().numpy(), ref_keep)
    keep2, _ = ops.nms_gpu(cu(bx), cu(sc), 0.5, pre_maxsize=100)
    assert np.array_equal(keep2.cpu().numpy(), ft_oracle.nms_bev(bx, sc, 0.5, pre_maxsize=100))
    k0, _ = ops.nms_gpu(torch.zeros((0, 7), device=dev()), torch.zeros((0,), device=dev()), 0.5)
    assert k0.numel() == 0


def test_decode_golden_and_eval_forward(ft_oracle):
    """Box decoding vs the reference's decode_bbox_from_heatmap (G4, bit-exact selection) and the whole evaluation
    forward of CenterPoint (decode + rotated NMS + recall record) on the fine-tune model."""
    from tmae_amd.modules.center_head import CenterHead
    g = golden('G4_decode')
    B, C, Hh, Ww, K = (int(v) for v in g['shape'])
    gen = torch.Generator().manual_seed(int(g['seed']))
    hm = torch.rand(B, C, Hh, Ww, generator=gen) ** 6
    center, cz = torch.rand(B, 2, Hh, Ww, generator=gen), torch.randn(B, 1, Hh, Ww, generator=gen) - 1
    dim = torch.rand(B, 3, Hh, Ww, generator=gen) * 3 + 0.5
    rc, rs = torch.randn(B, 1, Hh, Ww, generator=gen), torch.randn(B, 1, Hh, Ww, generator=gen)
    d = lambda t: t.to(dev())
    out = CenterHead._decode(d(hm), d(rc), d(rs), d(center), d(cz), d(dim), [-74.88, -74.88, -5.0], [0.32, 0.32, 8.0], 1, K,
                             float(g['score_thresh']), d(torch.from_numpy(g['limit'])))
    for k in range(B):
        assert np.array_equal(out[k]['pred_labels'].cpu().numpy(), g[f'pred_labels_{k}'])
        np.testing.assert_allclose(out[k]['pred_scores'].cpu().numpy(), g[f'pred_scores_{k}'], atol=1e-7)
        np.testing.assert_allclose(out[k]['pred_boxes'].cpu().numpy(), g[f'pred_boxes_{k}'], atol=2e-5)
    # evaluation forward of the detector
    g3 = golden('G3_finetune_e2e_3stage')
    cfg = ft_oracle.default_finetune_cfg(3)
    P = ft_oracle.init_finetune_params(cfg, seed=int(g3['param_seed']), tau=float(g3['tau']))
    P['dense_head.heads_list.0.hm.1.bias'] = P['dense_head.heads_list.0.hm.1.bias'] + 2.0      # some scores above 0.1
    model, _, _ = build_finetune_model(params=P, device=dev())
    model.eval()
    bd = {'points': cu(g3['points']), 'points_prev': cu(g3['points_prev']), 'batch_size': int(g3['batch_size']),
          'gt_boxes': cu(g3['gt_boxes'])}
    with torch.no_grad():
        preds, recall = model(bd)
    assert len(preds) == int(g3['batch_size']) and recall['gt_num'] > 0
    for p in preds:
        n = p['pred_boxes'].shape[0]
        assert p['pred_boxes'].shape == (n, 7) and p['pred_scores'].shape == (n,) and n <= 500
        assert n == 0 or (int(p['pred_labels'].min()) >= 1 and int(p['pred_labels'].max()) <= 5)
        if n > 1:                                            # survivors of the NMS do not overlap above the threshold
            from tmae_amd import ops
            iou = ops.boxes_iou_bev(p['pred_boxes'], p['pred_boxes'])
            iou.fill_diagonal_(0)
            assert float(iou.max()) <= 0.5 + 1e-4
            assert bool((p['pred_scores'][:-1] >= p['pred_scores'][1:]).all())


def test_datapath_golden():
    """On-device two-frame data path (ego removal, pose alignment, flip / rotation / scaling, crop, shuffle, collate)
    vs the batch the reference's dataset functions produce from the same scans and random draws (D1)."""
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from tmae_amd.data import TemporalPairPipeline
    import os
    g = golden('D1_datapath')
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 't-mae_amd', 'tools', 'cfgs', 'once_models')
    cfg = cfg_from_yaml_file(os.path.join(root, 't_mae_ssl.yaml'), EasyDict())
    pipe = TemporalPairPipeline(cfg.DATA_CONFIG, training=True)
    ns = int(g['n_samples'])
    samples = [dict(points=g[f'pts_{i}'], points_prev=g[f'prv_{i}'], pose=g[f'pose_cur_{i}'], pose_prev=g[f'pose_prv_{i}'])
               for i in range(ns)]
    params = [dict(flips=(['x'] if int(g[f'flip_x_{i}']) else []) + (['y'] if int(g[f'flip_y_{i}']) else []),
                   rot=float(g[f'rot_{i}']), scale=float(g[f'scale_{i}'])) for i in range(ns)]
    perms = [g[f'perm_{i}'] for i in range(ns)]
    out = pipe(samples, dev(), params=params, perms=perms)
    for key in ('points', 'points_prev'):
        got, ref = out[key].cpu().numpy(), g[key]
        assert got.shape == ref.shape, (key, got.shape, ref.shape)          # same points kept: crop / ego decisions agree
        assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[:, 4], ref[:, 4])
        np.testing.assert_allclose(got[:, 1:4], ref[:, 1:4], rtol=3e-7, atol=2e-5)   # fp32 rotation: fma / order, <= 2 ulp
    # the pipeline's own draws follow the reference's call order on np.random
    np.random.seed(100)
    p0 = pipe.draw()
    assert p0['flips'] == params[0]['flips'] and p0['rot'] == params[0]['rot'] and p0['scale'] == params[0]['scale']
    # a batch straight into the model
    from conftest import build_product_model
    model, _, _ = build_product_model(3, device=dev())
    model.train()
    ret, _, _ = model(dict(out))
    assert torch.isfinite(ret['loss'])


def _iou_head(g):
    """The product's CenterHead under the fixture's config (IoU branch, two heads, multi-class NMS)."""
    import json
    from pcdet.config import EasyDict
    from tmae_amd.modules.center_head import CenterHead
    cfg = EasyDict(json.loads(str(g['head_cfg'])))
    head = CenterHead(cfg, input_channels=16, num_class=5, class_names=['Car', 'Bus', 'Truck', 'Pedestrian', 'Cyclist'],
                      grid_size=np.array([96, 96, 1]), point_cloud_range=g['pc_range'], voxel_size=[0.32, 0.32, 8.0],
                      predict_boxes_when_training=False)
    sd = {str(n): torch.from_numpy(np.array(g[f'state_{i}'])) for i, n in enumerate(g['state_names'])}
    assert set(sd) == set(head.state_dict().keys())
    head.load_state_dict(sd)
    return head.to(dev())


def test_iou_head_loss_vs_reference():
    """CenterHead with an IoU branch (center_head.py:254-276, loss_utils.IoULossCenterNet) against the reference head run
    on the CPU (fixture G6): the ground-truth boxes per slot, every loss term, the total and the gradient norms."""
    g = golden('G6_iou_head')
    head = _iou_head(g)
    head.train()
    x = cu(g['x'])
    head({'spatial_features_2d': x, 'gt_boxes': cu(g['gt_boxes']), 'batch_size': x.shape[0]})
    tgt = head.forward_ret_dict['target_dicts']
    for hi in range(2):
        assert np.array_equal(tgt['masks'][hi].cpu().numpy(), g[f'masks_{hi}'])
        np.testing.assert_allclose(tgt['iou_boxes'][hi].cpu().numpy(), g[f'iou_boxes_{hi}'], atol=1e-6)
    loss, tb = head.get_loss()
    loss.backward()
    ref = dict(zip((str(k) for k in g['tb_names']), g['tb_values']))
    for k, v in ref.items():
        assert abs(float(tb[k]) - v) <= 2e-4 * max(1.0, abs(v)), (k, float(tb[k]), v)
    assert abs(fl(loss) - float(g['loss'])) <= 2e-4 * float(g['loss'])
    grads = dict(head.named_parameters())
    for n, gn in zip(g['grad_names'], g['grad_norms']):
        assert abs(float(grads[str(n)].grad.norm()) - gn) <= 2e-2 * max(1.0, gn), (n, float(grads[str(n)].grad.norm()), gn)


def test_iou_head_multi_class_nms_vs_reference():
    """Evaluation forward with IoU-rectified scores and per-class rotated NMS (model_nms_utils.py:28-46) vs the reference."""
    g = golden('G6_iou_head')
    head = _iou_head(g)
    head.eval()
    x = cu(g['x'])
    with torch.no_grad():
        dd = head({'spatial_features_2d': x, 'batch_size': x.shape[0]})
    for k, fd in enumerate(dd['final_box_dicts']):
        rs, rl, rb = g[f'det_scores_{k}'], g[f'det_labels_{k}'], g[f'det_boxes_{k}']
        s, l, b = fd['pred_scores'].cpu().numpy(), fd['pred_labels'].cpu().numpy(), fd['pred_boxes'].cpu().numpy()
        assert len(s) == len(rs) and np.array_equal(l, rl), (k, len(s), len(rs))
        np.testing.assert_allclose(s, rs, rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(b, rb, rtol=2e-4, atol=2e-4)


def test_train_cli_accepts_the_reference_launch_line(tmp_path):
    """tools/train.py with the flags of the reference's own launch line (tools/scripts/once_train.sh:17-20: --workers
    --extra_tag --max_ckpt_save_num 1 --num_epochs_to_eval 1 --amp --fixed_gap_eval 1) on the fine-tune config: two short
    epochs, one checkpoint kept, the last one evaluated."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = tmp_path / 'run'
    cmd = [sys.executable, os.path.join(ROOT, 't-mae_amd', 'tools', 'train.py'), '--cfg_file',
           os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae.yaml'), '--workers', '8', '--extra_tag', 't',
           '--max_ckpt_save_num', '1', '--num_epochs_to_eval', '1', '--amp', '--fixed_gap_eval', '1', '--synthetic', '--sync_bn',
           '--synthetic_points', '20000', '--iters_per_epoch', '2', '--epochs', '2', '--batch_size', '2', '--output_dir', str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ck = sorted(p.name for p in (out / 'ckpt').glob('*.pth'))
    assert ck == ['checkpoint_epoch_2.pth'], ck
    assert 'EVALUATION' in (r.stdout + r.stderr) or any((out / 'eval').rglob('*'))
    # and the reference's test line (tools/scripts/once_test.sh:8-10) on that checkpoint
    cmd = [sys.executable, os.path.join(ROOT, 't-mae_amd', 'tools', 'test.py'), '--cfg_file',
           os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae.yaml'), '--workers', '8', '--extra_tag', 't',
           '--ckpt', str(out / 'ckpt' / 'checkpoint_epoch_2.pth'), '--fixed_gap_eval', '1', '--synthetic',
           '--synthetic_points', '20000', '--synthetic_samples', '4', '--batch_size', '2', '--output_dir', str(out / 'test')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_once_loader_golden_and_real_data_cli(tmp_path):
    """The ONCE-format reader + on-device pipeline against what the reference's own ONCETemporalDataset returned on the
    same tiny ONCE-layout directory (fixture D2): same points kept in the same order (coordinates <= 2 ulp: the reference
    rotates with an fp32 BLAS matmul), gt_boxes bit for bit; then the loader as tools/train.py / tools/test.py use it
    (reader threads, resampling, evaluation split) and both tools WITHOUT --synthetic."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, finetune_data_cfg, write_once_directory
    from tmae_amd.data import ONCETemporalDataset, TemporalPairPipeline, build_dataloader
    g = golden('D2_once_dataset')
    root = tmp_path / 'once'
    write_once_directory(root, g)
    cfg = finetune_data_cfg()
    ds = ONCETemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True, root_path=root)
    pipe = TemporalPairPipeline(cfg.DATA_CONFIG, training=True, class_names=cfg.CLASS_NAMES, reference_rng_order=True)

    def lazy(i):                                             # the fixture seeded np.random per sample
        def read():
            np.random.seed(500 + i)
            return ds.raw_sample(i)
        return read
    n = len(ds)
    out = pipe([lazy(i) for i in range(n)], dev())           # its OWN draws, in the reference's per-sample order
    for key in ('points', 'points_prev'):
        got, ref = out[key].cpu().numpy(), g[key]
        assert got.shape == ref.shape, (key, got.shape, ref.shape)
        assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[:, 4], ref[:, 4])
        np.testing.assert_allclose(got[:, 1:4], ref[:, 1:4], rtol=3e-7, atol=2e-5)
    assert np.array_equal(out['gt_boxes'], g['gt_boxes'])
    assert [str(f) for f in out['frame_id']] == [str(g[f'info_frame_{int(p[0])}']) for p in g['picks']]
    # evaluation mode (no augmentation, no shuffle) on sample 2
    pipe_t = TemporalPairPipeline(cfg.DATA_CONFIG, training=False, class_names=cfg.CLASS_NAMES)
    ds_t = ONCETemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=False, root_path=root)
    np.random.seed(77)
    o2 = pipe_t([ds_t.raw_sample(2)], dev())
    for key, ref in (('points', g['test_points']), ('points_prev', g['test_points_prev'])):
        got = o2[key].cpu().numpy()[:, 1:]
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=3e-7, atol=2e-5)
    assert np.array_equal(o2['gt_boxes'][0], g['test_gt_boxes'].astype(np.float32))
    # the loader: reader threads, every sample once per epoch, a different order in the next epoch
    _, loader, sampler = build_dataloader(cfg.DATA_CONFIG, cfg.CLASS_NAMES, 3, dist=False, root_path=root, workers=2,
                                          training=True, device=dev())
    seen = []
    for ep in range(2):
        sampler.set_epoch(ep)
        frames = []
        for batch in loader:
            assert batch['points'].is_cuda and batch['points'].shape[1] == 5 and batch['batch_size'] in (1, 2, 3)
            assert batch['gt_boxes'].shape[0] == batch['batch_size'] and batch['gt_boxes'].shape[2] == 8
            cls = batch['gt_boxes'][..., 7]
            assert ((cls >= 0) & (cls <= 5)).all() and (np.abs(batch['gt_boxes'][..., 6]) <= np.pi + 1e-6).all()
            frames += [str(f) for f in batch['frame_id']]
        assert len(frames) == n and len(set(frames)) == n
        seen.append(frames)
    assert seen[0] != seen[1]
    # a sample that loses every box is replaced by another one (once_temporal_dataset.py:199-202)
    calls = []
    s_bad = dict(ds.raw_sample(0), gt_names=np.array(['Tricycle'] * len(ds.raw_sample(0)['gt_names'])))
    b = TemporalPairPipeline(cfg.DATA_CONFIG, training=True, class_names=cfg.CLASS_NAMES)(
        [s_bad, ds.raw_sample(1)], dev(), resample=lambda: (calls.append(1), ds.raw_sample(3))[1])
    assert calls == [1] and b['batch_size'] == 2 and b['gt_boxes'].shape[0] == 2
    # tools/train.py and tools/test.py on the directory, no --synthetic (the reference's launch line otherwise)
    out_dir = tmp_path / 'run'
    yaml_ft = os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae.yaml')
    cmd = [sys.executable, os.path.join(ROOT, 't-mae_amd', 'tools', 'train.py'), '--cfg_file', yaml_ft, '--workers', '2',
           '--extra_tag', 't', '--max_ckpt_save_num', '1', '--num_epochs_to_eval', '1', '--amp', '--epochs', '2', '--batch_size', '2',
           '--data_path', str(root), '--output_dir', str(out_dir),
           '--set', 'DATA_CONFIG.DATA_AUGMENTOR.DISABLE_AUG_LIST', 'gt_sampling']       # this directory has no label database
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert sorted(p.name for p in (out_dir / 'ckpt').glob('*.pth')) == ['checkpoint_epoch_2.pth']
    assert 'Total samples for ONCE dataset: %d' % n in (r.stdout + r.stderr)
    # stale higher-epoch checkpoints of an earlier run never cost the one just written (pruned by age, before the save)
    import shutil
    import time as _time
    for e in (7, 8):
        shutil.copy(out_dir / 'ckpt' / 'checkpoint_epoch_2.pth', out_dir / 'ckpt' / f'checkpoint_epoch_{e}.pth')
    old = _time.time() - 1000
    for e in (2, 7, 8):
        os.utime(out_dir / 'ckpt' / f'checkpoint_epoch_{e}.pth', (old + e, old + e))
    cmd2 = cmd[:cmd.index('--epochs') + 1] + ['4'] + cmd[cmd.index('--epochs') + 2:]
    cmd2[cmd2.index('--max_ckpt_save_num') + 1] = '3'
    k = cmd2.index('--set')                                               # --set takes the REMAINDER of the line: keep it last
    cmd2 = cmd2[:k] + ['--ckpt', str(out_dir / 'ckpt' / 'checkpoint_epoch_2.pth')] + cmd2[k:]
    r = subprocess.run(cmd2, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    left = sorted(p.name for p in (out_dir / 'ckpt').glob('*.pth'))
    assert 'checkpoint_epoch_3.pth' in left and 'checkpoint_epoch_4.pth' in left and len(left) == 3, left
    cmd = [sys.executable, os.path.join(ROOT, 't-mae_amd', 'tools', 'test.py'), '--cfg_file', yaml_ft, '--workers', '2',
           '--ckpt', str(out_dir / 'ckpt' / 'checkpoint_epoch_4.pth'), '--batch_size', '2', '--data_path', str(root),
           '--output_dir', str(out_dir / 'test')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'AP' in (r.stdout + r.stderr)
    # pre-training on the same directory (t_mae_ssl.yaml trains on the raw_large split: point it at the tiny train split)
    yaml_ssl = os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae_ssl.yaml')
    cmd = [sys.executable, os.path.join(ROOT, 't-mae_amd', 'tools', 'train.py'), '--cfg_file', yaml_ssl, '--workers', '2', '--amp',
           '--epochs', '1', '--batch_size', '2', '--data_path', str(root), '--output_dir', str(tmp_path / 'ssl'),
           '--set', 'DATA_CONFIG.DATA_SPLIT.train', 'train']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / 'ssl' / 'ckpt' / 'checkpoint_epoch_1.pth').exists()


def test_gt_sampling_device_path_golden(tmp_path):
    """gt_sampling end to end on the device against what the reference's DataBaseSampler + ONCETemporalDataset returned on
    the same tiny directory and label database (fixture D3): scene points inside the pasted boxes removed from BOTH frames
    (tmae_frame_prepare_boxes), the pasted points in front of both frames and through the same augmentation, the shuffle
    over [pasted (current) | pasted (previous) | previous | current] -- same points in the same order (<= 2 ulp), boxes bit
    for bit.  Then the loader and tools/train.py with the fine-tune recipe's full augmentor queue."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, finetune_data_cfg, write_once_directory
    from tmae_amd.data import ONCETemporalDataset, TemporalPairPipeline, build_dataloader
    g = golden('D3_gt_sampling')
    root = tmp_path / 'once'
    write_once_directory(root, g)
    cfg = finetune_data_cfg(gt_sampling=True)
    ds = ONCETemporalDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True, root_path=root)
    pipe = TemporalPairPipeline(cfg.DATA_CONFIG, training=True, class_names=cfg.CLASS_NAMES, reference_rng_order=True,
                                root_path=root)
    pipe.total_epochs = 1

    def lazy(i):
        def read():
            np.random.seed(int(g['seed_base']) + i)
            return ds.raw_sample(i)
        return read
    n = len(ds)
    out = pipe([lazy(i) for i in range(n)], dev())
    for key in ('points', 'points_prev'):
        got, ref = out[key].cpu().numpy(), g[key]
        assert got.shape == ref.shape, (key, got.shape, ref.shape)          # same points removed, pasted and kept
        assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[:, 4], ref[:, 4])
        np.testing.assert_allclose(got[:, 1:4], ref[:, 1:4], rtol=3e-7, atol=2e-5)
    assert np.array_equal(out['gt_boxes'], g['gt_boxes'])
    # the loader with the database, then the tool
    _, loader, sampler = build_dataloader(cfg.DATA_CONFIG, cfg.CLASS_NAMES, 3, dist=False, root_path=root, workers=2,
                                          training=True, total_epochs=2, device=dev())
    assert loader.pipeline.sampler is not None
    nb = sum(int((b['gt_boxes'][..., 7] > 0).sum()) for b in loader)
    own = sum(int(sum(str(x) in cfg.CLASS_NAMES for x in info['annos']['name'])) for info in ds.once_infos if 'annos' in info)
    assert nb > 0.8 * own
    yml = os.path.join(ROOT, 't-mae_amd', 'tools', 'cfgs', 'once_models', 't_mae.yaml')      # gt_sampling is in the shipped recipe
    cmd = [sys.executable, os.path.join(ROOT, 't-mae_amd', 'tools', 'train.py'), '--cfg_file', str(yml), '--workers', '2', '--amp',
           '--epochs', '1', '--batch_size', '2', '--data_path', str(root), '--output_dir', str(tmp_path / 'run')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'Database filter by min points' in (r.stdout + r.stderr)
    assert (tmp_path / 'run' / 'ckpt' / 'checkpoint_epoch_1.pth').exists()
