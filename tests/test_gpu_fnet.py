"""-m gpu: the F-Net on the matrix-core path (row N3): small kernels against torch fp32 references of the same op, the
whole FNetMFMA against the golden samples captured from the reference's PSMNet (G10) and against the torch module, and
MAGNET end to end with the F-Net feeding the matcher's layouts directly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from magnet_amd import fnet, lib
from magnet_amd.convnet import split_bf16
from tests.stubs import StubDNet, make_args, procedural_images, seeded_fnet_state, seeded_magnet_weights

pytestmark = pytest.mark.gpu


def _planes(x_cl, pad, gpu):
    """(N,h,w,C) fp32 -> zero-bordered split planes ((N*(h+2p)*(w+2p), C) hi, lo) on the GPU."""
    x = F.pad(x_cl, (0, 0, pad, pad, pad, pad))
    hi, lo = split_bf16(x.reshape(-1, x.shape[-1]).to(gpu))
    return hi, lo


def _join(hi, lo, N, hp, wp):
    return (hi.float() + lo.float()).cpu().reshape(N, hp, wp, -1)


def test_stem(hip_lib, gpu):
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 37, 50, generator=g); w = torch.randn(32, 3, 3, 3, generator=g) * 0.2; b = torch.randn(32, generator=g)
    H2, W2 = 19, 25
    hi = torch.zeros((2 * (H2 + 2) * (W2 + 2), 32), dtype=torch.bfloat16, device=gpu); lo = torch.zeros_like(hi)
    lib.fnet_stem(img.to(gpu), w.reshape(32, 27).contiguous().to(gpu), b.to(gpu), hi, lo)
    exp = F.relu(F.conv2d(img, w, b, stride=2, padding=1)).permute(0, 2, 3, 1)
    got = _join(hi, lo, 2, H2 + 2, W2 + 2)
    np.testing.assert_allclose(got[:, 1:-1, 1:-1].numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)
    assert not got[:, 0].any() and not got[:, :, 0].any() and not got[:, -1].any() and not got[:, :, -1].any()


def test_space_to_depth(hip_lib, gpu):
    x = torch.randn(2, 11, 14, 32, generator=torch.Generator().manual_seed(2))        # odd height: phase rows beyond the edge
    hi, lo = _planes(x, 1, gpu)
    H4, W4 = 6, 7
    oh = torch.zeros((2 * (H4 + 4) * (W4 + 4), 128), dtype=torch.bfloat16, device=gpu); ol = torch.zeros_like(oh)
    lib.space_to_depth(hi, lo, oh, ol, 2, 32, 11, 14, 2)
    got = _join(oh, ol, 2, H4 + 4, W4 + 4)[:, 2:-2, 2:-2]
    xs = (hi.float() + lo.float()).cpu().reshape(2, 13, 16, 32)[:, 1:-1, 1:-1]
    xs = F.pad(xs, (0, 0, 0, 0, 0, 1))                                                 # virtual zero row 11
    for py in (0, 1):
        for px in (0, 1):
            ph = py * 2 + px
            assert torch.equal(got[..., ph * 32:(ph + 1) * 32], xs[:, py::2, px::2])


@pytest.mark.parametrize("k", [8, 16, 64])
def test_avgpool(hip_lib, gpu, k):
    x = torch.randn(2, 64, 80, 128, generator=torch.Generator().manual_seed(3))
    buf = torch.zeros(2, 68, 84, 320); buf[:, 2:-2, 2:-2, 64:192] = x
    hi, lo = split_bf16(buf.reshape(-1, 320).to(gpu))
    ph, pw = 64 // k, 80 // k
    oh = torch.zeros((2 * ph * pw, 128), dtype=torch.bfloat16, device=gpu); ol = torch.zeros_like(oh)
    lib.avgpool_cl(hi[:, 64:192], lo[:, 64:192], 320, 2, 64, 80, 2, k, 128, oh, ol)
    xr = (hi.float() + lo.float()).cpu().reshape(2, 68, 84, 320)[:, 2:-2, 2:-2, 64:192]
    exp = F.avg_pool2d(xr.permute(0, 3, 1, 2), k, k).permute(0, 2, 3, 1).reshape(-1, 128)
    np.testing.assert_allclose((oh.float() + ol.float()).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("ph,pw", [(1, 2), (3, 5), (15, 20)])
def test_upsample_bilinear(hip_lib, gpu, ph, pw):
    q = torch.randn(2, ph, pw, 32, generator=torch.Generator().manual_seed(4))
    h, w = 60, 80
    oh = torch.zeros((2 * (h + 4) * (w + 4), 320), dtype=torch.bfloat16, device=gpu); ol = torch.zeros_like(oh)
    lib.upsample_bilinear_cl(q.reshape(-1, 32).to(gpu), 32, ph, pw, 32, oh[:, 224:256], ol[:, 224:256], 320, 2, h, w, 2)
    got = _join(oh, ol, 2, h + 4, w + 4)
    exp = F.interpolate(q.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    np.testing.assert_allclose(got[:, 2:-2, 2:-2, 224:256].numpy(), exp.numpy(), rtol=0, atol=2e-5)
    assert not got[..., :224].any() and not got[..., 256:].any() and not got[:, :2].any() and not got[:, :, :2].any()


@pytest.mark.parametrize("cfg", [
    dict(cin=32, cout=32, k=3, dil=1, res=True, relu=False),       # layer1 residual unit tail
    dict(cin=64, cout=64, k=3, dil=1, res=False, relu=True),
    dict(cin=128, cout=128, k=3, dil=2, res=True, relu=False),     # layer4 (dilated)
    dict(cin=64, cout=128, k=1, dil=1, res=False, relu=False),     # projection shortcut
])
def test_conv_extensions(hip_lib, gpu, cfg):
    """conv_mfma with the F-Net fields (dilation, residual input, border zeroing, 32/64-wide tiles) vs torch fp32 conv."""
    g = torch.Generator().manual_seed(5)
    N, h, w, pad = 2, 13, 21, 2
    cin, cout, k, dil = cfg["cin"], cfg["cout"], cfg["k"], cfg["dil"]
    x = torch.randn(N, h, w, cin, generator=g); wt = torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5); b = torch.randn(cout, generator=g)
    r = torch.randn(N, h, w, cout, generator=g)
    xh, xl = _planes(x, pad, gpu); rh, rl = _planes(r, pad, gpu)
    wh, wl = fnet._pack_taps(wt.to(gpu))
    rows, wp = N * (h + 2 * pad) * (w + 2 * pad), w + 2 * pad
    oh = torch.full((rows, cout), 7.0, dtype=torch.bfloat16, device=gpu); ol = torch.full_like(oh, 7.0)     # poisoned
    lib.conv_mfma(xh, xl, cin, cin, wh, wl, b.to(gpu), k * k, wp, cfg["relu"], rows, out_hi=oh, out_lo=ol,
                  add=(rh, rl, cout) if cfg["res"] else None, border=(h + 2 * pad, pad), dil=dil)
    got = _join(oh, ol, N, h + 2 * pad, wp)
    xr = (xh.float() + xl.float()).cpu().reshape(N, h + 2 * pad, wp, cin)[:, pad:-pad, pad:-pad]
    exp = F.conv2d(xr.permute(0, 3, 1, 2), wt, b, padding=dil if k == 3 else 0, dilation=dil).permute(0, 2, 3, 1)
    if cfg["res"]:
        exp = exp + (rh.float() + rl.float()).cpu().reshape(N, h + 2 * pad, wp, cout)[:, pad:-pad, pad:-pad]
    if cfg["relu"]:
        exp = F.relu(exp)
    np.testing.assert_allclose(got[:, pad:-pad, pad:-pad].numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)
    border = got.clone(); border[:, pad:-pad, pad:-pad] = 0
    assert not border.any()                                         # border outputs are zeros (next layer's padding)


def test_conv_repad_outputs(hip_lib, gpu):
    """Last layer: interior rows re-addressed into border-0 / border-1 grids, fp32 and single-plane bf16."""
    g = torch.Generator().manual_seed(6)
    N, h, w, pad, cin, cout = 3, 9, 12, 2, 128, 64
    x = torch.randn(N, h, w, cin, generator=g); wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    xh, xl = _planes(x, pad, gpu)
    wh, wl = fnet._pack_taps(wt.to(gpu))
    rows, wp = N * (h + 4) * (w + 4), w + 4
    zero_b = torch.zeros(cout, device=gpu)
    xr = (xh.float() + xl.float()).cpu().reshape(N, h + 4, wp, cin)[:, 2:-2, 2:-2]
    exp = F.conv2d(xr.permute(0, 3, 1, 2), wt).permute(0, 2, 3, 1)
    o0 = torch.empty((N, h, w, cout), dtype=torch.float32, device=gpu)
    lib.conv_mfma(xh, xl, cin, cin, wh, wl, zero_b, 1, wp, False, rows, out_f32=o0, border=(h + 4, 2), repad=1, out_ld=cout)
    np.testing.assert_allclose(o0.cpu().numpy(), exp.numpy(), rtol=1e-4, atol=1e-4)
    o1 = torch.zeros((N, h + 2, w + 2, cout), dtype=torch.bfloat16, device=gpu)
    lib.conv_mfma(xh, xl, cin, cin, wh, wl, zero_b, 1, wp, False, rows, out_bf16=o1, border=(h + 4, 2), repad=2, out_ld=cout)
    o1c = o1.float().cpu()
    np.testing.assert_allclose(o1c[:, 1:-1, 1:-1].numpy(), exp.numpy(), rtol=1e-2, atol=1e-2)              # bf16 storage
    assert torch.equal(o1c[:, 1:-1, 1:-1], o0.cpu().to(torch.bfloat16).float())                            # = RNE of the fp32 result
    assert not o1c[:, 0].any() and not o1c[:, :, 0].any()


def test_G10_fnet_mfma_vs_reference_golden(hip_lib, gpu, golden):
    m = seeded_fnet_state(fnet.PSMNet(feature_dim=64), seed=10).eval()
    img = procedural_images(2, 256, 320)
    out = fnet.FNetMFMA(m.to(gpu)).run(img.to(gpu)).cpu()
    assert tuple(out.shape) == (2, 64, 64, 80)
    scale = float(np.abs(golden["G10_feat_sparse"]).max())
    err = np.abs(out[:, :, ::4, ::4].numpy() - golden["G10_feat_sparse"]).max() / scale
    print(f"F-Net MFMA vs reference (G10): max err / max|feat| = {err:.2e}")
    assert err < 2e-4
    np.testing.assert_allclose(out.abs().mean(dim=(0, 2, 3)).numpy(), golden["G10_feat_absmean"], rtol=1e-3)


def test_fnet_matcher_layouts(hip_lib, gpu):
    """n_ref mode: ref (B,h,w,F) and zero-bordered source features == the NCHW output re-laid-out; bf16 = RNE of fp32."""
    m = seeded_fnet_state(fnet.PSMNet(feature_dim=64), seed=3).eval().to(gpu)
    run = fnet.FNetMFMA(m)
    img = procedural_images(3, 256, 256).to(gpu)
    nchw = run.run(img)
    ref_cl, src_pad = run.run(img, n_ref=1, feat_dtype="fp32")
    assert torch.equal(ref_cl, nchw[:1].permute(0, 2, 3, 1)) and torch.equal(src_pad[:, 1:-1, 1:-1], nchw[1:].permute(0, 2, 3, 1))
    assert not src_pad[:, 0].any() and not src_pad[:, :, -1].any()
    ref_b, src_b = run.run(img, n_ref=1, feat_dtype="bf16")
    assert ref_b.dtype == torch.bfloat16 and torch.equal(ref_b, ref_cl.to(torch.bfloat16)) and torch.equal(src_b, src_pad.to(torch.bfloat16))


def test_fnet_two_stream_split_is_bit_identical(hip_lib, gpu):
    """Batches of >= FNetMFMA.split_min_images images run as two half-batches on two HIP streams (the halves' launches fill each other's
    epilogue gaps).  Per image the same kernels see the same data: the split run must equal the one-stream run bit for bit — NCHW output,
    and the matcher-layout outputs with the reference / source boundary inside the first half, at the split point and inside the second."""
    m = seeded_fnet_state(fnet.PSMNet(feature_dim=64), seed=13).eval().to(gpu)
    run = fnet.FNetMFMA(m)
    assert run.split_min_images <= 9
    img = procedural_images(9, 256, 256).to(gpu)                      # odd count: halves of 5 and 4 images
    one = fnet.FNetMFMA(m); one.split_min_images = 10 ** 9
    assert torch.equal(run.run(img), one.run(img))
    for n_ref in (2, 5, 7, 9):
        for fd in ("fp32", "bf16"):
            a = [t.clone() for t in run.run(img, n_ref=n_ref, feat_dtype=fd)]
            b = one.run(img, n_ref=n_ref, feat_dtype=fd)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (n_ref, fd)
            assert a[1].shape[0] == 9 - n_ref and (a[1].numel() == 0 or (not a[1][:, 0].any() and not a[1][:, :, -1].any()))
    # a second call re-uses the buffers and streams
    assert torch.equal(run.run(img), one.run(img))


def test_magnet_with_fnet_mfma(hip_lib, gpu):
    """MAGNET.forward with a real PSMNet F-Net: matrix-core F-Net (features handed over in the matcher's layouts) vs the
    torch F-Net + pack path; depth abs_rel difference far below the 1e-4 bar."""
    from magnet_amd.magnet import MAGNET
    from magnet_amd import synth
    args = make_args(D=16, iters=2, dpv_h=64, dpv_w=80, fdim=64, V=2)
    args.FNET_architecture, args.FNET_feature_dim = "PSM-Net", 64
    f = seeded_fnet_state(fnet.FNET(args).f_net, seed=5)
    fn = fnet.FNET(args); fn.f_net = f
    model = MAGNET(args, d_net=StubDNet(0), f_net=fn, feat_dtype="fp32").to(gpu).eval()
    seeded_magnet_weights(model, seed=4)
    wl = synth.Workload("t", "scannet", 64, 80, V=2, D=16, F=64)
    inp = synth.make_inputs(wl, B=2, seed=11)
    ref_img = procedural_images(2, 256, 320).to(gpu); nb = procedural_images(4, 256, 320).flip(0).to(gpu)
    outs = {}
    for flag in (True, False):
        model.fnet_mfma = flag
        with torch.no_grad():
            outs[flag] = model(ref_img, nb, inp["nghbr_poses"].to(gpu), inp["is_valid"], inp["cam_intrins"], mode="test")
    a, b = outs[True][-1][:, 0].cpu(), outs[False][-1][:, 0].cpu()
    rel = ((a - b).abs() / b.abs().clamp_min(1e-3)).mean().item()
    print(f"MAGNET with F-Net on MFMA vs torch F-Net: mean |d mu|/mu = {rel:.2e}")
    assert rel < 2e-5


@pytest.mark.parametrize("hw", [(352, 1216), (258, 330), (480, 640)])
def test_fnet_mfma_shapes_vs_torch(hip_lib, gpu, hw):
    """KITTI / odd (H/2 = 129, W/2 = 165: the space-to-depth phases run past the edge) / ScanNet input sizes against the same
    module's torch forward on the GPU (MIOpen fp32)."""
    H, W = hw
    m = seeded_fnet_state(fnet.PSMNet(feature_dim=64), seed=7).eval().to(gpu)
    img = procedural_images(2, H, W).to(gpu)
    with torch.no_grad():
        ref = m(img)
    out = fnet.FNetMFMA(m).run(img)
    assert out.shape == ref.shape
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"F-Net {H}x{W}: max err / max|feat| = {err:.2e}")
    assert err < 2e-4


def test_fnet_batch_invariance_at_bench_size(hip_lib, gpu):
    """40 images of 480x640 in one pass (the F-Net benchmark batch) give, image by image, exactly what a one-image pass gives."""
    m = seeded_fnet_state(fnet.PSMNet(feature_dim=64), seed=11).eval().to(gpu)
    run = fnet.FNetMFMA(m)
    imgs = torch.randn(40, 3, 480, 640, generator=torch.Generator().manual_seed(12)).to(gpu)
    full = run.run(imgs).clone()
    for i in (0, 39):
        one = run.run(imgs[i:i + 1].contiguous())
        assert torch.isfinite(one).all() and torch.equal(one[0], full[i]), f"image {i}"
