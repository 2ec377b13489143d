"""-m gpu: the bf16x3 MFMA convolution path against torch fp32 convolutions on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _stack(cin, cout_last, seed):
    torch.manual_seed(seed)
    seq = nn.Sequential(nn.Conv2d(cin, 128, 3, padding=1), nn.ReLU(inplace=True),
                        nn.Conv2d(128, 128, 1), nn.ReLU(inplace=True),
                        nn.Conv2d(128, 128, 1), nn.ReLU(inplace=True),
                        nn.Conv2d(128, cout_last, 1))
    return seq.eval()


def _run_stack(seq, x, gpu, in_map=None, fuse_tail="epilogue"):
    """x: (B, C, h, w) fp32 CPU -> interior of the stack's fp32 output, (B, cout, h, w)."""
    from magnet_amd import lib
    from magnet_amd.convnet import ConvStackMFMA
    B, C, h, w = x.shape
    st = ConvStackMFMA(seq.to(gpu), in_map=in_map)
    # "epilogue": 1x1 tail fused into the 3x3 kernel (default); "chain": separate fused-chain kernel; "separate": one launch per layer
    st.fuse_epilogue = fuse_tail == "epilogue"
    st.fuse_tail = fuse_tail == "chain"
    ctot = st.cin_pad()
    rows = B * (h + 2) * (w + 2)
    hi = torch.zeros((rows, ctot), dtype=torch.bfloat16, device=gpu); lo = torch.zeros_like(hi)
    if in_map is None:
        lib.pack_split(x.to(gpu), hi, lo, ctot, 0)
    else:
        for src, n, dst in in_map:
            lib.pack_split(x[:, src:src + n].contiguous().to(gpu), hi, lo, ctot, dst)
    out, ld = st.run(hi, lo, ctot, rows, w + 2, {})
    cout = seq[-1].out_channels
    o = out.view(B, h + 2, w + 2, ld)[:, 1:-1, 1:-1, :cout].permute(0, 3, 1, 2).contiguous().cpu()
    seq.cpu()
    return o


@pytest.mark.parametrize("fuse_tail", ["epilogue", "chain", "separate"])
@pytest.mark.parametrize("cin,cout,h,w,B", [(320, 2, 12, 16, 2), (256, 144, 9, 21, 1), (64, 2, 30, 40, 3)])
def test_conv_stack_matches_fp32(hip_lib, gpu, cin, cout, h, w, B, fuse_tail):
    seq = _stack(cin, cout, seed=cin + cout)
    x = torch.randn(B, cin, h, w, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref = seq(x)
    got = _run_stack(seq, x, gpu, fuse_tail=fuse_tail)
    err = (got - ref).abs().max().item(); scale = ref.abs().max().item()
    print(f"[conv {cin}->{cout} fused_tail={fuse_tail}] max|d|={err:.3e} max|ref|={scale:.3f} rel={err / scale:.2e}")
    assert torch.isfinite(got).all() and err <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize("cin,cout", [(320, 2), (256, 144)])
def test_conv_stack_chip_filling_shape_matches_fp32(hip_lib, gpu, cin, cout):
    """>= one 256-row tile per CU (4 frames of 120x160: 79 056 rows): the launcher picks the 8-wave ping-pong / register-window
    kernel with the column-owned fused tail (smaller shapes above run the 4-wave kernel); ragged last tile included."""
    seq = _stack(cin, cout, seed=cin + cout + 1)
    x = torch.randn(4, cin, 120, 160, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        ref = seq(x)
    got = _run_stack(seq, x, gpu, fuse_tail="epilogue")
    err = (got - ref).abs().max().item(); scale = ref.abs().max().item()
    print(f"[conv {cin}->{cout} 4x120x160] max|d|={err:.3e} max|ref|={scale:.3f} rel={err / scale:.2e}")
    assert torch.isfinite(got).all() and err <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize("B,h,w,n_pred", [(1, 9, 21, 1), (2, 12, 16, 3), (4, 120, 160, 2)])
def test_mask_head_with_fused_upsampling_equals_the_two_launch_form(hip_lib, gpu, B, h, w, n_pred):
    """MAGNET.py:172-173 in one launch (the mask head's last 1x1 layer soft-maxes its logits and writes the x4-upsampled
    predictions) against mask head -> (rows, 144) fp32 -> magnet_upsample_depth_cl_n: same arithmetic in the same order, so the
    outputs must be bit-identical.  Small shapes run the 4-wave kernel, 4 x 120 x 160 the 8-wave one (ragged last tile)."""
    from magnet_amd import lib
    from magnet_amd.convnet import ConvStackMFMA
    seq = _stack(256, 144, seed=17).to(gpu)
    st = ConvStackMFMA(seq)
    assert st.can_fuse_upsample(gpu)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, 256, h, w, generator=g)
    depths = [torch.cat([torch.rand(B, 1, h, w, generator=g) * 5 + 0.5, torch.rand(B, 1, h, w, generator=g) * 0.5 + 0.05], dim=1).to(gpu)
              for _ in range(n_pred)]
    rows = B * (h + 2) * (w + 2)
    hi = torch.zeros((rows, 256), dtype=torch.bfloat16, device=gpu); lo = torch.zeros_like(hi)
    lib.pack_split(x.to(gpu), hi, lo, 256, 0)
    mask, ld = st.run(hi, lo, 256, rows, w + 2, {})
    ref = lib.upsample_depth_cl_n(depths, mask, ld)
    d = torch.stack(depths)
    outs = torch.full((n_pred, B, 2, 4 * h, 4 * w), float("nan"), dtype=torch.float32, device=gpu)
    none, ld2 = st.run(hi, lo, 256, rows, w + 2, {}, upsample=(d, outs))
    torch.cuda.synchronize()
    assert none is None and ld2 == 144
    for i in range(n_pred):
        assert torch.equal(outs[i], ref[i]), f"prediction {i}: max|d| = {(outs[i] - ref[i]).abs().max().item():.3e}"
    # and against the reference's own formulation on the CPU (MAGNET.py:15-27)
    m = mask.view(B, h + 2, w + 2, ld)[:, 1:-1, 1:-1, :144].permute(0, 3, 1, 2).cpu()
    mk = torch.softmax(m.view(B, 1, 9, 4, 4, h, w), dim=2)
    up = torch.nn.functional.unfold(depths[0].cpu(), [3, 3], padding=1).view(B, 2, 9, 1, 1, h, w)
    want = torch.sum(mk * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 4 * h, 4 * w)
    assert (outs[0].cpu() - want).abs().max().item() <= 5e-6


@pytest.mark.parametrize("B,h,w", [(2, 12, 16), (4, 120, 160)])
def test_gnet_with_fused_gaussian_update_equals_the_two_launch_form(hip_lib, gpu, B, h, w):
    """MAGNET.py:62 + 60-69 in one launch (the head's last layer updates (mu, sigma) itself) against G-Net -> (rows, 16) fp32 ->
    magnet_gaussian_update_cl: bit-identical.  Both kernel forms (4-wave / 8-wave with a ragged last tile)."""
    from magnet_amd import lib
    from magnet_amd.convnet import ConvStackMFMA
    seq = _stack(320, 2, seed=29).to(gpu)
    st = ConvStackMFMA(seq)
    assert st.can_fuse_gauss(gpu)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, 320, h, w, generator=g)
    gmm = torch.cat([torch.rand(B, 1, h, w, generator=g) * 5 + 0.5, torch.rand(B, 1, h, w, generator=g) * 0.5 + 0.05], dim=1).to(gpu)
    rows = B * (h + 2) * (w + 2)
    hi = torch.zeros((rows, 320), dtype=torch.bfloat16, device=gpu); lo = torch.zeros_like(hi)
    lib.pack_split(x.to(gpu), hi, lo, 320, 0)
    o, ld = st.run(hi, lo, 320, rows, w + 2, {})
    ref = lib.gaussian_update_cl(o, ld, gmm, h, w)
    out = torch.full_like(gmm, float("nan"))
    none, ld2 = st.run(hi, lo, 320, rows, w + 2, {}, gauss=(gmm, out))
    torch.cuda.synchronize()
    assert none is None and ld2 == 16
    assert torch.equal(out, ref), f"max|d| = {(out - ref).abs().max().item():.3e}"


def test_conv_stack_channel_map_odd_D(hip_lib, gpu):
    """G-Net with D = 5: cost channels [0,5), x_d3 at channel offset 8 of the 288-wide buffer."""
    seq = _stack(256 + 5, 2, seed=3)
    x = torch.randn(2, 261, 12, 16, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        ref = seq(x)
    got = _run_stack(seq, x, gpu, in_map=[(0, 5, 0), (5, 256, 8)])
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item())


def test_single_layer_asymmetric(hip_lib, gpu):
    """One 3x3 layer with a one-hot weight: a transposed operand or a wrong tap offset cannot pass."""
    from magnet_amd import lib
    from magnet_amd.convnet import ConvStackMFMA
    conv = nn.Conv2d(32, 128, 3, padding=1, bias=False)
    with torch.no_grad():
        conv.weight.zero_()
        conv.weight[7, 3, 0, 2] = 1.0           # out ch 7 <- in ch 3 at (dy=-1, dx=+1)
        conv.weight[100, 31, 2, 1] = -2.0       # out ch 100 <- in ch 31 at (dy=+1, dx=0)
    seq = nn.Sequential(conv).eval()
    x = torch.randn(1, 32, 10, 14, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        ref = seq(x)
    got = _run_stack(seq, x, gpu)
    assert torch.allclose(got, ref, atol=1e-6)
    assert got[0, 7].abs().sum() > 0 and got[0, 8].abs().sum() == 0


def test_gaussian_update_and_upsample_cl(hip_lib, gpu):
    from magnet_amd import lib
    from oracle import oracle
    g = torch.Generator().manual_seed(9)
    B, h, w = 2, 7, 11
    o = torch.randn(B, 2, h, w, generator=g) * 2; gmm = torch.rand(B, 2, h, w, generator=g) + 0.2
    pad = torch.full((B, h + 2, w + 2, 16), 3.0)
    pad[:, 1:-1, 1:-1, :2] = o.permute(0, 2, 3, 1)
    got = lib.gaussian_update_cl(pad.to(gpu).view(-1, 16), 16, gmm.to(gpu), h, w).cpu().numpy()
    np.testing.assert_allclose(got, oracle.gaussian_update(o.numpy(), gmm.numpy()), rtol=1e-6, atol=1e-6)
    m = torch.randn(B, 144, h, w, generator=g) * 2; d = torch.rand(B, 2, h, w, generator=g) * 4
    mp = torch.full((B, h + 2, w + 2, 144), -5.0)
    mp[:, 1:-1, 1:-1] = m.permute(0, 2, 3, 1)
    got = lib.upsample_depth_cl(d.to(gpu), mp.to(gpu).view(-1, 144), 144).cpu().numpy()
    np.testing.assert_allclose(got, oracle.upsample_depth_via_mask(d.numpy(), m.numpy(), 4), rtol=0, atol=5e-6)


def test_depth_metrics_match_reference(hip_lib, gpu, golden):
    """Device reductions vs utils.compute_depth_errors (G7 golden values) incl. validate()'s clamping/masking."""
    from magnet_amd import metrics as M
    from oracle import oracle
    gt = torch.from_numpy(golden["G7_gt"]); pr = torch.from_numpy(golden["G7_pred"]); var = torch.from_numpy(golden["G7_var"])
    n = gt.numel()
    pred = torch.stack([pr, var.sqrt()], 0).view(1, 2, 1, n)
    m = M.compute_depth_errors(pred.to(gpu), gt.view(1, 1, 1, n).to(gpu), 1e-3, 10.0)[0]
    for k, v in zip(golden["G7_keys"], golden["G7_vals"]):
        assert abs(m[str(k)] - v) <= 2e-5 * max(1.0, abs(v)), (k, m[str(k)], v)
    # masking + clamping as in test_MaGNet.py:43,58-79
    g = torch.Generator().manual_seed(4)
    gt2 = torch.rand(2, 1, 24, 32, generator=g) * 12                       # some > max_depth -> invalid
    gt2[0, 0, :3] = 0.0
    pr2 = torch.rand(2, 2, 24, 32, generator=g) * 11 + 0.01
    pr2[0, 0, 5, 5] = float("inf"); pr2[1, 0, 6, 6] = float("nan"); pr2[1, 0, 7, 7] = -1.0
    got = M.compute_depth_errors(pr2.to(gpu), gt2.to(gpu), 1e-3, 10.0)
    for b in range(2):
        gtn = gt2[b, 0].numpy().copy(); gtn[gtn > 10.0] = 0.0
        mask = np.logical_and(gtn > 1e-3, gtn < 10.0)
        pn = pr2[b, 0].numpy().copy()
        pn[pn < 1e-3] = 1e-3; pn[pn > 10.0] = 10.0; pn[np.isinf(pn)] = 10.0; pn[np.isnan(pn)] = 1e-3
        ref = oracle.compute_depth_errors(gtn[mask], pn[mask], np.square(pr2[b, 1].numpy())[mask])
        for k in M.METRIC_ORDER:
            assert abs(got[b][k] - float(ref[k])) <= 2e-5 * max(1.0, abs(float(ref[k]))), (b, k, got[b][k], ref[k])


def test_depth_metrics_kitti_crops_and_determinism(hip_lib, gpu):
    """garg / eigen evaluation windows (test_MaGNet.py:63-71) against the same masking done in numpy + the oracle metrics, at the
    KITTI evaluation size; and the sums are bit-identical run to run (fixed summation order, no atomics)."""
    from magnet_amd import metrics as M
    from oracle import oracle
    H, W = 352, 1216
    g = torch.Generator().manual_seed(6)
    gt = torch.rand(1, 1, H, W, generator=g) * 90
    gt[torch.rand(1, 1, H, W, generator=g) < 0.7] = 0.0                     # sparse LiDAR ground truth
    pr = torch.rand(1, 2, H, W, generator=g) * 80 + 0.5
    for kind in ("garg", "eigen"):
        y0, y1, x0, x1 = M.crop_window(kind, H, W)
        got = M.compute_depth_errors(pr.to(gpu), gt.to(gpu), 1e-3, 80.0, crop=kind)[0]
        gtn = gt[0, 0].numpy().copy(); gtn[gtn > 80.0] = 0.0
        mask = np.logical_and(gtn > 1e-3, gtn < 80.0)
        ev = np.zeros_like(mask); ev[y0:y1, x0:x1] = True
        mask &= ev
        pn = np.clip(pr[0, 0].numpy(), 1e-3, 80.0)
        ref = oracle.compute_depth_errors(gtn[mask], pn[mask], np.square(pr[0, 1].numpy())[mask])
        for k in M.METRIC_ORDER:
            assert abs(got[k] - float(ref[k])) <= 2e-5 * max(1.0, abs(float(ref[k]))), (kind, k, got[k], ref[k])
    a = M.depth_metric_sums(pr.to(gpu), gt.to(gpu), 1e-3, 80.0)
    for _ in range(3):
        assert torch.equal(a, M.depth_metric_sums(pr.to(gpu), gt.to(gpu), 1e-3, 80.0))


def test_eval_driver_runs(hip_lib, gpu, tmp_path):
    """The test_MaGNet.py-shaped loop end to end on tiny synthetic windows (one NaN pose -> a dropped view)."""
    import eval_synthetic as E
    from magnet_amd.magnet import MAGNET
    from tests.stubs import StubDNet, StubFNet, make_args, seeded_magnet_weights
    args = make_args(D=5, iters=2, dpv_h=12, dpv_w=16, V=2)
    args.min_depth, args.max_depth = 1e-3, 10.0
    model = MAGNET(args, d_net=StubDNet(1), f_net=StubFNet(2, fdim=8))
    seeded_magnet_weights(model, 3)
    model = model.to(gpu).eval()
    loader = E.SyntheticWindows(2, 2, 2, 48, 64, nan_every=1)
    m = E.validate(model, args, loader, gpu)
    assert set(m.keys()) == set(E.M.METRIC_ORDER) and all(np.isfinite(v) for v in m.values())
    E.M.log_metrics(str(tmp_path / "test_acc.txt"), m, "synthetic")
    assert (tmp_path / "test_acc.txt").read_text().count("\n") == 4


# ---- round 4: the fp16 + block-scaled e4m3 operand format (ABI v302; conv_mfma.hip WIN == 4 loop, magnet_pack_mx) ----------------------
def _pack_mx_planes(x, gpu, ctot=None, c_off=0):
    from magnet_amd import lib
    B, C, h, w = x.shape
    ctot = ctot or C
    rows = B * (h + 2) * (w + 2)
    f16 = torch.zeros((rows, ctot), dtype=torch.float16, device=gpu)
    qr = torch.zeros((rows, ctot), dtype=torch.int16, device=gpu)
    sc = torch.zeros((ctot // 32, rows), dtype=torch.int32, device=gpu)
    lib.pack_mx(x.to(gpu), f16, qr, sc, ctot, c_off, rows)
    return f16, qr, sc, rows


def test_pack_mx_equals_the_torch_restatement(hip_lib, gpu):
    """magnet_pack_mx against convnet.split_mx (torch: fp16 RNE, E8M0 block exponent, e4m3 RNE): every plane bit for bit."""
    from magnet_amd.convnet import split_mx
    g = torch.Generator().manual_seed(3)
    B, C, h, w = 2, 96, 10, 14
    x = torch.randn(B, C, h, w, generator=g) * torch.exp(torch.randn(B, C, 1, 1, generator=g) * 2.0)
    x[0, :32, 2, 3] = 0.0                                              # an all-zero block
    x[1, 40, 5, 5] = 3.0e4; x[1, 41, 5, 5] = 1.0e-7                    # fp16-range extremes inside one block
    f16, qr, sc, rows = _pack_mx_planes(x, gpu)
    xi = x.permute(0, 2, 3, 1).contiguous()                            # (B, h, w, C)
    hi, q, s = split_mx(xi)
    inner = lambda t: t.view(B, h + 2, w + 2, -1)[:, 1:-1, 1:-1].cpu()
    assert torch.equal(inner(f16), hi)
    assert torch.equal(inner(qr), q)
    assert torch.equal(sc.view(C // 32, B, h + 2, w + 2)[:, :, 1:-1, 1:-1].permute(1, 2, 3, 0).cpu(), s)
    # border rows untouched (zeros)
    assert f16.view(B, h + 2, w + 2, C)[:, 0].abs().sum().item() == 0 and sc.view(C // 32, B, h + 2, w + 2)[:, :, :, 0].abs().sum().item() == 0


def test_pack_mx_non_finite_inputs(hip_lib, gpu):
    """include/magnet_hip.h (v302 domain note): a NaN is written through to the fp16 plane and to both e4m3 planes — it is not turned
    into a finite bound by the clamp — while the block exponents come from the block's finite entries; +-Inf and |x| > 65504 clamp to
    +-65504; every other (row, block) is untouched by a neighbour's NaN."""
    from magnet_amd.convnet import split_mx
    g = torch.Generator().manual_seed(5)
    B, C, h, w = 1, 64, 6, 8
    x = torch.randn(B, C, h, w, generator=g)
    x[0, 3, 2, 4] = float("nan")
    x[0, 40, 1, 1] = float("inf"); x[0, 41, 1, 1] = -1.0e9
    f16, qr, sc, rows = _pack_mx_planes(x, gpu)
    f = f16.view(B, h + 2, w + 2, C)[:, 1:-1, 1:-1].cpu()
    q8 = qr.view(B, h + 2, w + 2, C)[:, 1:-1, 1:-1].cpu().contiguous().view(torch.uint8).view(B, h, w, C // 32, 64)
    assert torch.isnan(f[0, 2, 4, 3]) and int(torch.isnan(f).sum()) == 1
    assert (int(q8[0, 2, 4, 0, 3]) & 0x7f) == 0x7f and (int(q8[0, 2, 4, 0, 32 + 3]) & 0x7f) == 0x7f        # e4m3 NaN in the hi and lo bytes
    assert f[0, 1, 1, 40].item() == 65504.0 and f[0, 1, 1, 41].item() == -65504.0
    # everything that is not in the NaN's own (pixel, block) equals the finite restatement of the clamped tensor
    xc = torch.nan_to_num(x, nan=0.0).clamp(-65504.0, 65504.0).permute(0, 2, 3, 1).contiguous()
    hi, q, s_ = split_mx(xc)
    keep = torch.ones(B, h, w, C, dtype=torch.bool); keep[0, 2, 4, :32] = False
    assert torch.equal(f[keep], hi[keep])
    assert torch.equal(qr.view(B, h + 2, w + 2, C)[:, 1:-1, 1:-1].cpu()[keep], q[keep])


@pytest.mark.parametrize("cin,cout", [(320, 2), (256, 144)])
def test_conv_stack_mx_format_matches_fp32(hip_lib, gpu, cin, cout):
    """The 2-unit operand format (fp16 main term + block-scaled e4m3 correction terms) on the chip-filling shape, against torch fp32:
    the same bar as the bf16x3 form (2e-5 of the output scale), and against the bf16x3 kernel on the same inputs."""
    from magnet_amd import lib
    from magnet_amd.convnet import ConvStackMFMA
    seq = _stack(cin, cout, seed=cin + cout + 1)
    x = torch.randn(4, cin, 120, 160, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        ref = seq(x)
    B, C, h, w = x.shape
    st = ConvStackMFMA(seq.to(gpu))
    f16, qr, sc, rows = _pack_mx_planes(x, gpu)
    out, ld = st.run(f16, qr, C, rows, w + 2, {}, mx=(sc, rows))
    got = out.view(B, h + 2, w + 2, ld)[:, 1:-1, 1:-1, :cout].permute(0, 3, 1, 2).contiguous().cpu()
    seq.cpu()
    err = (got - ref).abs().max().item(); scale = ref.abs().max().item()
    rms = ((got - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()).item()
    print(f"[conv mx {cin}->{cout} 4x120x160] max|d|={err:.3e} max|ref|={scale:.3f} rel={err / scale:.2e} rms rel={rms:.2e}")
    assert torch.isfinite(got).all() and err <= 2e-5 * max(1.0, scale)
