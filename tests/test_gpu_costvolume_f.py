"""-m gpu: est_costvolume_F (the F-Net training volume, reference homography.py:10-75) — forward and the
hand-written backward through the C ABI, against the golden vectors captured from the reference (G9) and the
CPU oracle on larger seeded cases."""
import numpy as np
import pytest
import torch

from magnet_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


def _bins(D, dmin=1e-3, dmax=10.0):
    return torch.tensor(np.exp(np.log(dmax + 1 - dmin) * (np.arange(D) + 0.5) / D) - (1 - dmin), dtype=torch.float32).view(1, D, 1, 1)


def _run(gpu, d_center, ref, src, poses, is_valid, intM, rays, path, gout=None, softmax=False):
    from magnet_amd import homography
    rf = torch.as_tensor(ref).to(gpu).requires_grad_(gout is not None)
    sf = torch.as_tensor(src).to(gpu).requires_grad_(gout is not None)
    poses = torch.as_tensor(poses)
    cam = {"intM": torch.as_tensor(intM), "unit_ray_array_2D": torch.as_tensor(rays)}
    if softmax:
        out = homography.est_costvolume_F(torch.as_tensor(d_center), rf, sf, poses[:, :, :3, :3].to(gpu), poses[:, :, :3, 3].to(gpu),
                                          torch.as_tensor(is_valid), cam, path=path)
    else:
        bins = [float(v) for v in torch.as_tensor(d_center).reshape(-1)]
        out = homography._CostVolumeF.apply(rf, sf, bins, poses.to(gpu).contiguous(), torch.as_tensor(is_valid).int().to(gpu),
                                            cam["intM"].to(gpu), cam["unit_ray_array_2D"].to(gpu), path)
    if gout is None:
        return out.detach().cpu().numpy()
    (out * torch.as_tensor(gout).to(gpu)).sum().backward()
    return out.detach().cpu().numpy(), rf.grad.cpu().numpy(), sf.grad.cpu().numpy()


def _g9(g):
    return (g["G9_d_center"], g["G9_ref_feat"], g["G9_nghbr_feat"], g["G9_poses"], g["G9_is_valid"], g["G9_intM"], g["G9_rays"])


def test_G9_raw_generic_bitwise(hip_lib, gpu, golden):
    """Generic kernel, mode 1: bit-identical to the reference's volume before its softmax."""
    raw = _run(gpu, *_g9(golden), path=1)
    assert np.array_equal(raw, golden["G9_raw"])


def test_G9_raw_candidate_lane(hip_lib, gpu, golden):
    raw = _run(gpu, *_g9(golden), path=0)
    np.testing.assert_allclose(raw, golden["G9_raw"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("path", [0, 1])
def test_G9_softmax(hip_lib, gpu, golden, path):
    sm = _run(gpu, *_g9(golden), path=path, softmax=True)
    np.testing.assert_allclose(sm, golden["G9_softmax"], rtol=0, atol=1e-6)


def test_G9_gradients_vs_reference_autograd(hip_lib, gpu, golden):
    raw, gr, gs = _run(gpu, *_g9(golden), path=0, gout=golden["G9_gout"])
    np.testing.assert_allclose(gr, golden["G9_grad_ref"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gs, golden["G9_grad_src"], rtol=1e-4, atol=1e-5)
    B = golden["G9_ref_feat"].shape[0]
    assert not gs[2 * B + 0].any()                                  # invalid view: exactly zero


@pytest.mark.parametrize("case", [
    dict(h=24, w=40, V=2, D=64, F=64, B=2),          # one candidate block, F = 64 (CPL = 2)
    dict(h=30, w=44, V=4, D=80, F=64, B=1),          # the reference's training bins (two candidate blocks), ragged tiles
    dict(h=9, w=21, V=3, D=7, F=24, B=3),            # ragged everything, partial channel chunks
    dict(h=16, w=16, V=2, D=130, F=128, B=1),        # three candidate blocks, CPL = 4
])
def test_forward_backward_vs_oracle(hip_lib, gpu, case):
    wl = synth.Workload("f", "scannet", case["h"], case["w"], V=case["V"], D=case["D"], F=case["F"])
    inp = synth.make_inputs(wl, B=case["B"], seed=77, invalid=[(0, case["V"] - 1)] if case["B"] > 1 else ())
    dc = _bins(case["D"])
    gout = torch.randn(case["B"], case["D"], case["h"], case["w"], generator=torch.Generator().manual_seed(5))
    args = (dc.numpy(), inp["ref_feat"].numpy(), inp["nghbr_feat"].numpy(), inp["nghbr_poses"].numpy(), inp["is_valid"].numpy(),
            inp["cam_intrins"]["intM"].numpy(), inp["cam_intrins"]["unit_ray_array_2D"].numpy())
    o_raw, o_gr, o_gs = oracle.cost_volume_f_raw(*args, gout=gout.numpy())
    raw1 = _run(gpu, *args, path=1)
    assert np.array_equal(raw1, o_raw)                              # generic kernel: bitwise
    raw, gr, gs = _run(gpu, *args, path=0, gout=gout.numpy())
    scale = max(1.0, float(np.abs(o_raw).max()))
    np.testing.assert_allclose(raw, o_raw, rtol=2e-5, atol=2e-5 * scale)
    for got, exp in ((gr, o_gr), (gs, o_gs)):
        tol = 2e-5 * max(1.0, float(np.abs(exp).max()))
        np.testing.assert_allclose(got, exp, rtol=1e-4, atol=tol)


def test_softmax_chain_gradient(hip_lib, gpu):
    """Gradient through the drop-in est_costvolume_F including its softmax (torch autograd on top of the HIP
    backward) against the oracle: d/d raw of sum(W * softmax(raw)) computed in fp64, then the oracle's backward."""
    wl = synth.Workload("f", "scannet", 20, 28, V=3, D=32, F=32)
    inp = synth.make_inputs(wl, B=2, seed=3)
    dc = _bins(32)
    W = torch.randn(2, 32, 20, 28, generator=torch.Generator().manual_seed(8))
    args = (dc.numpy(), inp["ref_feat"].numpy(), inp["nghbr_feat"].numpy(), inp["nghbr_poses"].numpy(), inp["is_valid"].numpy(),
            inp["cam_intrins"]["intM"].numpy(), inp["cam_intrins"]["unit_ray_array_2D"].numpy())
    raw = oracle.cost_volume_f_raw(*args).astype(np.float64)
    e = np.exp(raw - raw.max(axis=1, keepdims=True)); sm = e / e.sum(axis=1, keepdims=True)
    Wn = W.numpy().astype(np.float64)
    g_raw = sm * (Wn - (Wn * sm).sum(axis=1, keepdims=True))
    _, o_gr, o_gs = oracle.cost_volume_f_raw(*args, gout=g_raw.astype(np.float32))
    sm_hip, gr, gs = _run(gpu, *args, path=0, gout=W.numpy(), softmax=True)
    np.testing.assert_allclose(sm_hip, sm, rtol=0, atol=2e-6)
    for got, exp in ((gr, o_gr), (gs, o_gs)):
        np.testing.assert_allclose(got, exp, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(exp).max())))


def test_backward_rejects_bad_mode(hip_lib, gpu):
    from magnet_amd import lib
    r = torch.zeros(1, 4, 4, 8, device=gpu)
    with pytest.raises(lib.MagnetError):
        lib.cost_volume_f_backward(r, torch.zeros(1, 6, 6, 8, device=gpu), torch.eye(4, device=gpu).view(1, 1, 4, 4),
                                   torch.ones(1, 1, dtype=torch.int32, device=gpu), torch.eye(3, device=gpu).view(1, 3, 3),
                                   torch.ones(1, 3, 16, device=gpu), [1.0, 2.0], torch.zeros(1, 3, 4, 4, device=gpu))


def test_backward_kernels_agree(hip_lib, gpu):
    """Gather backward (default: no atomics, deterministic) vs the tile-privatised scatter kernel (LDS hash table, path bit
    0x1000) vs the per-item atomic kernel (path bit 0x2000) vs the oracle, on a shape with long epipolar runs and an invalid
    view; the gather path must also be bit-identical run to run."""
    from magnet_amd import lib
    wl = synth.Workload("f", "scannet", 28, 36, V=3, D=80, F=64)
    inp = synth.make_inputs(wl, B=2, seed=21, invalid=[(1, 0)])
    inp["nghbr_poses"][0, 1, :3, 3] = torch.tensor([0.5, 0.1, 0.0])                 # strong lateral parallax
    dc = _bins(80)
    gout = torch.randn(2, 80, 28, 36, generator=torch.Generator().manual_seed(6))
    args = (dc.numpy(), inp["ref_feat"].numpy(), inp["nghbr_feat"].numpy(), inp["nghbr_poses"].numpy(), inp["is_valid"].numpy(),
            inp["cam_intrins"]["intM"].numpy(), inp["cam_intrins"]["unit_ray_array_2D"].numpy())
    _, o_gr, o_gs = oracle.cost_volume_f_raw(*args, gout=gout.numpy())
    ref_cl = lib.pack_features(inp["ref_feat"].to(gpu), lib.FEAT_F32, pad=0); src_pad = lib.pack_features(inp["nghbr_feat"].to(gpu), lib.FEAT_F32, pad=1)
    bins = [float(v) for v in dc.reshape(-1)]
    common = (ref_cl, src_pad, inp["nghbr_poses"].to(gpu), inp["is_valid"].int().to(gpu), inp["cam_intrins"]["intM"].to(gpu),
              inp["cam_intrins"]["unit_ray_array_2D"].to(gpu), bins, gout.to(gpu))
    a1 = lib.cost_volume_f_backward(*common, path=0); a2 = lib.cost_volume_f_backward(*common, path=0)
    assert torch.equal(a1[0], a2[0]) and torch.equal(a1[1], a2[1])      # production gather: fixed summation order, no atomics
    for path in (0, 0x4000, 0x1000, 0x2000):                            # gather (16- / 32-texel segments), hash scatter, plain atomics
        gr, gs = lib.cost_volume_f_backward(*common, path=path)
        gr = gr.permute(0, 3, 1, 2).cpu().numpy(); gs = gs[:, 1:-1, 1:-1].permute(0, 3, 1, 2).cpu().numpy()
        for got, exp in ((gr, o_gr), (gs, o_gs)):
            np.testing.assert_allclose(got, exp, rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(exp).max())))


def test_backward_full_training_shape_properties(hip_lib, gpu):
    """The reference's F-Net training shape (120x160, V = 4, D = 80 SID bins, F = 64): too large for the oracle in a test, so
    size-independent properties: the backward is linear in the upstream gradient, a zero gradient gives zeros, and
    <grad_cost, cost(ref + e*dref) - cost(ref)> / e matches <grad_ref, dref> (the backward is the adjoint of the forward)."""
    from magnet_amd import lib
    wl = synth.Workload("f", "scannet", 120, 160, V=4, D=80, F=64)
    B = 2
    inp = synth.make_inputs(wl, B=B, seed=31, smooth_feats=True)
    bnd = np.exp(np.log(10.0 + 1 - 1e-3) * np.arange(81) / 80) - (1 - 1e-3)
    bins = [float(v) for v in ((bnd[:-1] + bnd[1:]) / 2).astype(np.float32)]
    ref_cl = lib.pack_features(inp["ref_feat"].to(gpu), lib.FEAT_F32, pad=0); src_pad = lib.pack_features(inp["nghbr_feat"].to(gpu), lib.FEAT_F32, pad=1)
    geo = (inp["nghbr_poses"].to(gpu), inp["is_valid"].int().to(gpu), inp["cam_intrins"]["intM"].to(gpu), inp["cam_intrins"]["unit_ray_array_2D"].to(gpu))
    g = torch.Generator().manual_seed(32)
    g1 = torch.randn(B, 80, 120, 160, generator=g).to(gpu); g2 = torch.randn(B, 80, 120, 160, generator=g).to(gpu)
    bw = lambda gg: lib.cost_volume_f_backward(ref_cl, src_pad, *geo, bins, gg)
    r1, s1 = bw(g1); r2, s2 = bw(g2); r12, s12 = bw(g1 + 2.0 * g2)
    for a, b_ in ((r12, r1 + 2.0 * r2), (s12, s1 + 2.0 * s2)):
        assert torch.allclose(a, b_, rtol=1e-3, atol=2e-4 * float(b_.abs().max()))
    r0, s0 = bw(torch.zeros_like(g1))
    assert not r0.any() and not s0.any()
    # adjoint test on the reference features (the forward is linear in them)
    fwd = lambda rc: lib.cost_volume_cw(rc, src_pad, None, *geo, 0.0, k_list=bins, mode=1)
    dref = torch.randn(ref_cl.shape, generator=torch.Generator().manual_seed(33)).to(gpu)
    lhs = ((fwd(ref_cl + dref) - fwd(ref_cl)).double() * g1.double()).sum().item()
    rhs = (r1.double() * dref.double()).sum().item()
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
