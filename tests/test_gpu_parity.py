"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from magnet_amd import synth
from oracle import oracle
from tests.parity import assert_cost_parity, oracle_cost, to_dev
from tests.stubs import StubDNet, StubFNet, make_args, seeded_magnet_weights

pytestmark = pytest.mark.gpu


def _hip_cost(inp, k_list, device, feat_dtype="fp32", d_volume=None, path=0, kappa=5, out=None):
    from magnet_amd.homography import CostVolumeCW
    d = to_dev(inp, device)
    cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"],
                      d["cam_intrins"], kappa, feat_dtype=feat_dtype, path=path)
    if d_volume is not None:
        return cv(d_volume=d_volume.to(device), out=out)
    return cv(ref_gmm=d["ref_gmms"], k_list=k_list, out=out)


def _golden_tiny(g):
    return dict(ref_feat=torch.from_numpy(g["G2_ref_feat"]), nghbr_feat=torch.from_numpy(g["G2_nghbr_feat"]),
                ref_gmms=torch.from_numpy(g["G2_ref_gmms"]), nghbr_gmms=torch.from_numpy(g["G2_nghbr_gmms"]),
                nghbr_poses=torch.from_numpy(g["G2_nghbr_poses"]), is_valid=torch.from_numpy(g["G2_is_valid"]),
                cam_intrins={"intM": torch.from_numpy(g["G2_intM"]), "unit_ray_array_2D": torch.from_numpy(g["G2_rays"])})


def test_device_is_gfx950(hip_lib, gpu):
    assert hip_lib.magnet_device_count() >= 1
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ---- pack ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 8, 5, 7), (2, 64, 12, 16), (1, 64, 120, 160), (2, 72, 9, 33)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("pad", [0, 1])
def test_pack_features(hip_lib, gpu, shape, dtype, pad):
    from magnet_amd import lib
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
    out = torch.full((shape[0], shape[2] + 2 * pad, shape[3] + 2 * pad, shape[1]), 3.0,
                     dtype=lib.feat_torch_dtype(lib.feat_enum(dtype)), device=gpu)     # poisoned: border must be zeroed
    got = lib.pack_features(x.to(gpu), lib.feat_enum(dtype), pad=pad, out=out).cpu()
    exp = torch.nn.functional.pad(x, (pad, pad, pad, pad)).permute(0, 2, 3, 1).contiguous()
    if dtype == "bf16":
        exp = exp.to(torch.bfloat16)
    assert got.dtype == exp.dtype and torch.equal(got, exp)        # bit-exact (RNE for bf16), zero border


def test_pack_gmm(hip_lib, gpu):
    from magnet_amd import lib
    g = torch.rand(3, 2, 13, 21, generator=torch.Generator().manual_seed(2)) + 0.5
    got = lib.pack_gmm(g.to(gpu)).cpu()
    exp = torch.nn.functional.pad(g, (1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous()
    assert torch.equal(got, exp)


# ---- cost volume: golden (reference) vectors -------------------------------------------------------
@pytest.mark.parametrize("path", [2, 1, 3])
@pytest.mark.parametrize("fused", [True, False])
def test_cost_volume_tiny_golden(hip_lib, gpu, golden, path, fused):
    """The reference's own output on the edge-case vector (invalid view, behind-camera pose, OOB)."""
    inp = _golden_tiny(golden)
    k = list(golden["G1_k_D5"])
    dv = None if fused else torch.from_numpy(golden["G2_d_volume"])
    got = _hip_cost(inp, k, gpu, d_volume=dv, path=path)
    assert_cost_parity(got, golden["G2_cost"], path=path, label=f"tiny fused={fused}")


def test_cost_volume_reference_signature(hip_lib, gpu, golden):
    """homography.est_costvolume_CW(d_volume, ref_feat, nghbr_feat, ref_gmms, nghbr_gmms, R, t,
    is_valid, cam_intrins, thres) with is_valid / cam_intrins on the CPU, as test_MaGNet.py passes them."""
    from magnet_amd import homography
    inp = _golden_tiny(golden)
    d = to_dev(inp, gpu)
    R = d["nghbr_poses"][:, :, :3, :3]; t = d["nghbr_poses"][:, :, :3, 3]
    got = homography.est_costvolume_CW(torch.from_numpy(golden["G2_d_volume"]).to(gpu), d["ref_feat"],
                                       d["nghbr_feat"], d["ref_gmms"], d["nghbr_gmms"], R, t,
                                       inp["is_valid"], inp["cam_intrins"], 5)
    assert got.shape == (2, 5, 12, 16) and got.dtype == torch.float32 and got.device.type == "cuda"
    assert_cost_parity(got, golden["G2_cost"], path=0, label="reference signature")


def test_cost_volume_C1_golden_subsample(hip_lib, gpu, golden):
    wl = synth.WORKLOADS["C1"]
    inp = synth.make_inputs(wl, B=1, seed=0)
    for path in (2, 1, 3):
        got = _hip_cost(inp, list(golden["G1_k_D16"]), gpu, path=path).cpu().numpy()
        assert_cost_parity(got[:, :, ::5, ::7], golden["G2_C1_cost_sub"], path=path, label="C1 golden")


# ---- cost volume: oracle on seeded inputs ------------------------------------------------------------
CASES = [
    # name, workload, B, seed, feat_dtype, invalid
    ("C1", "C1", 1, 0, "fp32", ()),
    ("C1-b2-invalid", "C1", 2, 1, "fp32", ((0, 1),)),
    ("C2-fp32", "C2", 1, 0, "fp32", ()),
    ("C2-bf16", "C2", 2, 1, "bf16", ((1, 2),)),
    ("C5", "C5", 1, 2, "fp32", ()),
    ("shipped-D5", "shipped", 2, 0, "fp32", ()),
]


@pytest.mark.parametrize("path", [2, 1, 3])
@pytest.mark.parametrize("name,wlname,B,seed,fdt,invalid", CASES)
def test_cost_volume_vs_oracle(hip_lib, gpu, name, wlname, B, seed, fdt, invalid, path):
    wl = synth.WORKLOADS[wlname]
    inp = synth.make_inputs(wl, B=B, seed=seed, invalid=list(invalid), round_bf16=(fdt == "bf16"))
    k = oracle.depth_sampling(3, wl.D)
    orc = oracle_cost(inp, k)
    got = _hip_cost(inp, k, gpu, feat_dtype=fdt, path=path)
    assert_cost_parity(got, orc, path=path, label=name)


def test_cost_volume_kitti_wide_aspect(hip_lib, gpu):
    """C4: 88x304, V=4, D=128, KITTI forward motion (long epipolar segments, many OOB samples)."""
    wl = synth.WORKLOADS["C4"]
    inp = synth.make_inputs(wl, B=1, seed=0)
    k = oracle.depth_sampling(3, wl.D)
    orc = oracle_cost(inp, k)
    for path in (2, 1, 3):
        got = _hip_cost(inp, k, gpu, path=path)
        assert_cost_parity(got, orc, path=path, label="C4 kitti")


def test_cost_volume_ragged_grid_and_odd_D(hip_lib, gpu):
    """Grid not a multiple of the 16x4 tile, D not a multiple of 4, F=8, V=1."""
    wl = synth.Workload("ragged", "scannet", 13, 19, V=1, D=7, F=8)
    inp = synth.make_inputs(wl, B=3, seed=4)
    k = oracle.depth_sampling(3, wl.D)
    orc = oracle_cost(inp, k)
    for path in (2, 1, 3):
        got = _hip_cost(inp, k, gpu, path=path)
        assert_cost_parity(got, orc, path=path, label="ragged")


@pytest.mark.parametrize("V,D,F,fdt", [(6, 24, 32, "bf16"), (1, 256, 8, "fp32"), (3, 40, 128, "fp32"), (2, 9, 128, "bf16"),
                                        (8, 16, 72, "fp32")])
def test_cost_volume_shape_sweep(hip_lib, gpu, V, D, F, fdt):
    """Channel counts off the F = 64 fast path (partial / multi-chunk lanes), D up to the ABI limit, many views."""
    wl = synth.Workload("sweep", "7scenes", 10, 23, V=V, D=D, F=F)
    inp = synth.make_inputs(wl, B=2, seed=V * 100 + D, round_bf16=(fdt == "bf16"), invalid=[(1, 0)])
    k = oracle.depth_sampling(3, D)
    orc = oracle_cost(inp, k)
    for path in (2, 1) + ((3,) if D <= 128 else ()):
        got = _hip_cost(inp, k, gpu, feat_dtype=fdt, path=path)
        assert_cost_parity(got, orc, path=path, label=f"sweep V={V} D={D} F={F} {fdt}")


def test_cost_volume_nan_and_degenerate_inputs(hip_lib, gpu):
    """NaN / zero sigma / zero depth in the reference gmm and a singular pose: the reference's arithmetic yields 0
    for those samples (closed gate, out-of-range sample); every kernel must agree with the oracle and stay finite."""
    wl = synth.Workload("nan", "scannet", 12, 16, V=2, D=8, F=8)
    inp = synth.make_inputs(wl, B=2, seed=21)
    inp["ref_gmms"][0, 0, 3, 4] = float("nan")          # mu NaN
    inp["ref_gmms"][0, 1, 5, 6] = float("nan")          # sigma NaN
    inp["ref_gmms"][1, 1, 2, :] = 0.0                   # sigma 0: all candidates identical
    inp["ref_gmms"][1, 0, 7, :] = 0.0                   # depth 0
    inp["nghbr_poses"][1, 1, :3, :3] = 0.0              # singular rotation
    k = oracle.depth_sampling(3, 8)
    orc = oracle_cost(inp, k)
    assert np.isfinite(orc).all()
    for path in (2, 1, 3):
        got = _hip_cost(inp, k, gpu, path=path)
        assert_cost_parity(got, orc, path=path, label="nan/degenerate")


def test_cost_volume_all_views_invalid_is_zero(hip_lib, gpu):
    wl = synth.Workload("inv", "scannet", 12, 16, V=2, D=5, F=8)
    inp = synth.make_inputs(wl, B=1, seed=5, invalid=[(0, 0), (0, 1)])
    for path in (2, 1, 3):
        got = _hip_cost(inp, oracle.depth_sampling(3, 5), gpu, path=path)
        assert torch.count_nonzero(got) == 0


def test_kernel_selection_stats(hip_lib, gpu):
    """path 0 = candidate-lane kernel for every D (several candidate blocks above 64), path 3 = worklist
    kernel up to D = 128, path 1 = generic; `stats` counts tiles per kernel class."""
    from magnet_amd import lib
    from magnet_amd.homography import CostVolumeCW
    for D, path, which in ((16, 0, 0), (130, 0, 0), (130, 1, 1), (100, 3, 0), (7, 2, 0)):
        wl = synth.Workload("st", "scannet", 12, 16, V=2, D=D, F=8)
        inp = synth.make_inputs(wl, B=2, seed=7)
        d = to_dev(inp, gpu)
        k = oracle.depth_sampling(3, D)
        stats = torch.zeros(4, dtype=torch.int32, device=gpu)
        cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"],
                          d["cam_intrins"], 5, path=path)
        got = cv(ref_gmm=d["ref_gmms"], k_list=k, stats=stats)
        st = stats.cpu().tolist()
        assert st[which] == 2 * 3 * 1 and st[1 - which] == 0, (D, path, st)   # B=2 frames x (12/4) x (16/16) tiles
        assert_cost_parity(got, oracle_cost(inp, k), path=path, label=f"D={D} path={path}")
    wl = synth.Workload("st", "scannet", 12, 16, V=2, D=130, F=8)
    inp = synth.make_inputs(wl, B=1, seed=7); d = to_dev(inp, gpu)
    cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"],
                      d["cam_intrins"], 5, path=3)
    with pytest.raises(lib.MagnetError, match="worklist kernel does not take"):
        cv(ref_gmm=d["ref_gmms"], k_list=oracle.depth_sampling(3, 130))


def test_cost_volume_strided_output_into_gnet_buffer(hip_lib, gpu):
    """The kernel writes the first D channels of G-Net's (B, D+256, h, w) input directly."""
    wl = synth.Workload("s", "scannet", 12, 16, V=2, D=5, F=8)
    inp = synth.make_inputs(wl, B=2, seed=6)
    k = oracle.depth_sampling(3, 5)
    for path in (2, 1, 3):
        buf = torch.full((2, 5 + 3, 12, 16), 7.0, device=gpu)
        _hip_cost(inp, k, gpu, out=buf[:, :5], path=path)
        dense = _hip_cost(inp, k, gpu, path=path)
        assert torch.equal(buf[:, :5], dense) and torch.all(buf[:, 5:] == 7.0)


def test_cost_volume_split_channel_last_output(hip_lib, gpu):
    """cost_hi/cost_lo: the matcher writes split-bf16 planes of the conv kernel's zero-bordered channel-last buffer.
    Must equal splitting the fp32 cost volume, touch only the D cost channels of interior rows, for D = 64 and D = 5."""
    from magnet_amd.convnet import split_bf16
    from magnet_amd.homography import CostVolumeCW
    for D, F in ((64, 16), (5, 8)):
        wl = synth.Workload("sp", "scannet", 13, 19, V=2, D=D, F=F)
        inp = synth.make_inputs(wl, B=2, seed=11)
        d = to_dev(inp, gpu)
        k = oracle.depth_sampling(3, D)
        cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"], d["cam_intrins"], 5)
        dense = cv(ref_gmm=d["ref_gmms"], k_list=k)                                    # (B,D,h,w) fp32
        ld = 96
        rows = 2 * 15 * 21
        hi = torch.full((rows, ld), 7.0, dtype=torch.bfloat16, device=gpu); lo = torch.full_like(hi, 7.0)
        cv(ref_gmm=d["ref_gmms"], k_list=k, out_split=(hi, lo, ld))
        eh, el = split_bf16(dense.permute(0, 2, 3, 1).contiguous())                    # (B,h,w,D)
        hi4 = hi.view(2, 15, 21, ld); lo4 = lo.view(2, 15, 21, ld)
        assert torch.equal(hi4[:, 1:-1, 1:-1, :D], eh) and torch.equal(lo4[:, 1:-1, 1:-1, :D], el)
        poison = torch.full((1,), 7.0, dtype=torch.bfloat16, device=gpu)
        assert torch.all(hi4[:, 1:-1, 1:-1, D:] == poison) and torch.all(hi4[:, 0] == poison) and torch.all(hi4[:, :, 0] == poison)
        assert torch.all(lo4[:, -1] == poison) and torch.all(lo4[:, :, -1] == poison)


def test_linearity_in_reference_features(hip_lib, gpu):
    """Size-independent property at the full C2 shape: the score is linear in the reference
    features for fixed gates (scaling ref features by 2 scales the cost volume by exactly 2)."""
    wl = synth.WORKLOADS["C2"]
    inp = synth.make_inputs(wl, B=1, seed=3)
    k = oracle.depth_sampling(3, wl.D)
    for path in (2, 1, 3):
        a = _hip_cost(inp, k, gpu, feat_dtype="bf16", path=path)
        inp2 = dict(inp); inp2["ref_feat"] = inp["ref_feat"] * 2.0
        b = _hip_cost(inp2, k, gpu, feat_dtype="bf16", path=path)
        assert torch.equal(b, a * 2.0)


# ---- gaussian update / upsample ------------------------------------------------------------------------
def test_gaussian_update_golden(hip_lib, gpu, golden):
    from magnet_amd import lib
    got = lib.gaussian_update(torch.from_numpy(golden["G4_raw"]).to(gpu), torch.from_numpy(golden["G4_prev"]).to(gpu))
    np.testing.assert_allclose(got.cpu().numpy(), golden["G4_out"], rtol=0, atol=1e-6)
    x = torch.randn(3, 2, 37, 53, generator=torch.Generator().manual_seed(2)) * 3
    g = torch.rand(3, 2, 37, 53, generator=torch.Generator().manual_seed(3)) + 0.1
    got = lib.gaussian_update(x.to(gpu), g.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, oracle.gaussian_update(x.numpy(), g.numpy()), rtol=1e-6, atol=1e-6)


def test_gnet_forward_golden(hip_lib, gpu, golden):
    """GNET.forward (conv stack on MIOpen + HIP tail) against the reference's output, with the
    reference's seeded weights reconstructed from the same generator."""
    from magnet_amd.magnet import GNET
    torch.manual_seed(11)
    net = GNET(ch_in=261)                     # same construction order as the golden generator
    assert torch.equal(net.state_dict()["gnet.6.bias"], torch.from_numpy(golden["G4_sd_gnet.6.bias"]))
    net = net.to(gpu).eval()
    with torch.no_grad():
        y = net(torch.from_numpy(golden["G4_x"]).to(gpu), torch.from_numpy(golden["G4_prev"]).to(gpu))
    np.testing.assert_allclose(y.cpu().numpy(), golden["G4_out"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k", [4, 2])
def test_upsample(hip_lib, gpu, golden, k):
    from magnet_amd import lib
    if k == 4:
        got = lib.upsample_depth(torch.from_numpy(golden["G5_depth"]).to(gpu), torch.from_numpy(golden["G5_mask"]).to(gpu), 4)
        np.testing.assert_allclose(got.cpu().numpy(), golden["G5_out"], rtol=0, atol=3e-6)
    g = torch.Generator().manual_seed(9)
    d = torch.rand(2, 2, 11, 70, generator=g) * 4; m = torch.randn(2, 9 * k * k, 11, 70, generator=g) * 2
    got = lib.upsample_depth(d.to(gpu), m.to(gpu), k).cpu().numpy()
    np.testing.assert_allclose(got, oracle.upsample_depth_via_mask(d.numpy(), m.numpy(), k), rtol=0, atol=5e-6)


# ---- full loop ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", ["mfma", "torch"])
def test_magnet_forward_matches_reference_output(hip_lib, gpu, golden, backend):
    """G6: the reference's MAGNET.forward output (stub D-Net/F-Net, I=3, one invalid view) vs ours:
    final-depth abs_rel delta < 1e-6 (BASELINE.json's parity bar is 1e-4; the measured agreement is 1e-7-class, and the bar sits
    next to it so that a systematic matcher error of 0.1 % cannot pass: tests/test_gpu_golden_r3.py has the negative control)."""
    from magnet_amd.magnet import MAGNET
    args = make_args(D=5, iters=3, dpv_h=12, dpv_w=16)
    m = MAGNET(args, d_net=StubDNet(seed=21), f_net=StubFNet(seed=22, fdim=8), conv_backend=backend)
    seeded_magnet_weights(m, seed=23)
    m = m.to(gpu).eval()
    intr = synth.make_intrinsics("scannet", 12, 16, 2)
    with torch.no_grad():
        preds = m(torch.from_numpy(golden["G6_ref_img"]).to(gpu), torch.from_numpy(golden["G6_nghbr_imgs"]).to(gpu),
                  torch.from_numpy(golden["G6_poses"]).to(gpu), torch.from_numpy(golden["G6_is_valid"]), intr, mode="test")
    assert len(preds) == 3
    for i, p in enumerate(preds):
        ref = golden[f"G6_pred{i}"]
        assert tuple(p.shape) == ref.shape == (2, 2, 48, 64)
        got = p.cpu().numpy()
        ar = oracle.abs_rel(ref[:, 0], got[:, 0])
        print(f"[G6 iter {i}] abs_rel(ours vs reference mu) = {ar:.3e}; max|dmu|={np.abs(got[:,0]-ref[:,0]).max():.3e}")
        assert ar < 1e-6                                  # measured 7e-8 (torch convolutions) .. 2.8e-7 (matrix-core convolutions)
        np.testing.assert_allclose(got[:, 1], ref[:, 1], rtol=5e-3, atol=1e-6)


@pytest.mark.parametrize("backend", ["mfma", "torch"])
def test_full_loop_abs_rel_C3_shape(hip_lib, gpu, backend):
    """C3 shape (120x160, V=4, D=64, I=3, bf16 storage): HIP loop vs an oracle loop (oracle matcher +
    torch-CPU G-Net + oracle tail/upsample) on identical inputs: abs_rel delta < 1e-4."""
    from magnet_amd.magnet import MAGNET
    wl = synth.WORKLOADS["C3"]
    inp = synth.make_inputs(wl, B=1, seed=0, round_bf16=False)          # the HAND-OVER is fp32 (what an F-Net produces) ...
    args = make_args(D=wl.D, iters=3, dpv_h=wl.h, dpv_w=wl.w)
    m = MAGNET(args, d_net=StubDNet(1), f_net=StubFNet(2), feat_dtype="bf16", conv_backend=backend)
    # ... the kernel stores bf16 (round to nearest even), so the oracle is fed exactly those rounded values
    inp_orc = dict(inp, ref_feat=inp["ref_feat"].to(torch.bfloat16).float(), nghbr_feat=inp["nghbr_feat"].to(torch.bfloat16).float())
    seeded_magnet_weights(m, seed=5)
    x_d3 = torch.randn(1, 256, wl.h, wl.w, generator=torch.Generator().manual_seed(7)) * 0.5
    # oracle loop on the CPU
    k = oracle.depth_sampling(3, wl.D)
    gmm = inp["ref_gmms"].clone(); cpu_preds = []
    with torch.no_grad():
        mask = m.mask_head(x_d3)
        for _ in range(3):
            cost = torch.from_numpy(oracle_cost(dict(inp_orc, ref_gmms=gmm), k))
            raw = m.g_net.gnet(torch.cat([cost, x_d3], dim=1))
            gmm = torch.from_numpy(oracle.gaussian_update(raw.numpy(), gmm.numpy()))
            cpu_preds.append(oracle.upsample_depth_via_mask(gmm.numpy(), mask.numpy(), 4))
    m = m.to(gpu).eval()
    d = to_dev(inp, gpu)
    with torch.no_grad():
        preds = m.match_and_refine(d["ref_gmms"], x_d3.to(gpu), d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"],
                                   d["nghbr_poses"], inp["is_valid"], inp["cam_intrins"], mode="test")
    for i, (p, c) in enumerate(zip(preds, cpu_preds)):
        got = p.cpu().numpy()
        ar = oracle.abs_rel(np.abs(c[:, 0]) + 1e-3, np.abs(got[:, 0]) + 1e-3)
        print(f"[C3 loop iter {i}] abs_rel delta = {ar:.3e}")
        assert np.isfinite(got).all() and ar < 1e-4


def test_hoisted_invariant_conv_matches_unhoisted(hip_lib, gpu):
    """I = 3: computing the x_d3 part of G-Net's first layer once per forward (hoist_invariant) must agree with
    recomputing the full first layer every iteration, for D = 64 and the shipped D = 5."""
    from magnet_amd.magnet import MAGNET
    for D, fd in ((64, 16), (5, 8)):
        wl = synth.Workload("h", "scannet", 12, 16, V=2, D=D, F=fd)
        inp = synth.make_inputs(wl, B=2, seed=13)
        d = to_dev(inp, gpu)
        x_d3 = (torch.randn(2, 256, 12, 16, generator=torch.Generator().manual_seed(3)) * 0.5).to(gpu)
        args = make_args(D=D, iters=3, dpv_h=12, dpv_w=16)
        m = MAGNET(args, d_net=StubDNet(1), f_net=StubFNet(2, fdim=fd))
        seeded_magnet_weights(m, seed=9)
        m = m.to(gpu).eval()
        outs = []
        for hoist in (True, False):
            m.hoist_invariant = hoist
            with torch.no_grad():
                outs.append([p.clone() for p in m.match_and_refine(d["ref_gmms"], x_d3, d["ref_feat"], d["nghbr_feat"],
                                                                   d["nghbr_gmms"], d["nghbr_poses"], inp["is_valid"],
                                                                   inp["cam_intrins"], mode="test")])
        for a, b in zip(*outs):
            assert torch.isfinite(a).all()
            assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), (D, (a - b).abs().max().item())


def test_graph_replay_matches_eager(hip_lib, gpu):
    """HIP-graph capture of the refinement loop (magnet_amd/graph.py): replays with new inputs, validity and poses equal eager runs."""
    from magnet_amd.graph import GraphedRefine
    from magnet_amd.magnet import MAGNET
    args = make_args(D=16, iters=2, dpv_h=24, dpv_w=32, V=3)
    model = MAGNET(args, d_net=StubDNet(0), f_net=StubFNet(1), feat_dtype="bf16").to(gpu).eval()
    seeded_magnet_weights(model, seed=2)
    wl = synth.Workload("g", "scannet", 24, 32, V=3, D=16, F=64)

    def inputs(seed, invalid):
        inp = to_dev(synth.make_inputs(wl, B=2, seed=seed, invalid=invalid), gpu)
        x_d3 = torch.randn(2, 256, 24, 32, generator=torch.Generator().manual_seed(seed)).to(gpu) * 0.5
        return inp, (inp["ref_gmms"], x_d3, inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"])

    inp0, t0 = inputs(1, ())
    g = GraphedRefine(model, *t0, inp0["is_valid"], inp0["cam_intrins"])
    for seed, invalid in ((1, ()), (5, [(1, 2)]), (9, [(0, 0), (1, 1)])):
        inp, t = inputs(seed, invalid)
        with torch.no_grad():
            eager = [o.clone() for o in model.match_and_refine(*t, inp["is_valid"], inp["cam_intrins"], mode="test")]
        got = g(*t, is_valid=inp["is_valid"], cam_intrins=inp["cam_intrins"])
        torch.cuda.synchronize()
        assert len(got) == len(eager) and all(torch.equal(a, b) for a, b in zip(got, eager))


def test_cost_volume_bench_size_batch_invariance(hip_lib, gpu):
    """BASELINE full size (C2, 64 frames per launch, bf16 features — the bench.py launch): frames of the batched launch equal
    the same frames launched alone (bitwise: no cross-frame state, no 32-bit offset overflow at 6 GB of inputs), and three of them
    are checked against the oracle."""
    wl = synth.WORKLOADS["C2"]
    B = 64
    inp = synth.make_inputs(wl, B=B, seed=123, round_bf16=True)
    k = oracle.depth_sampling(3, wl.D)
    full = _hip_cost(inp, k, gpu, feat_dtype="bf16", path=2)       # exact kernel; the production matcher has the same test in test_gpu_fast_matcher.py
    V = wl.V

    def frame(b):
        sel = lambda t: t[b:b + 1].contiguous()
        nb = inp["nghbr_feat"].view(V, B, *inp["nghbr_feat"].shape[1:])[:, b:b + 1].reshape(V, *inp["nghbr_feat"].shape[1:]).contiguous()
        ng = inp["nghbr_gmms"].view(V, B, *inp["nghbr_gmms"].shape[1:])[:, b:b + 1].reshape(V, *inp["nghbr_gmms"].shape[1:]).contiguous()
        return dict(ref_feat=sel(inp["ref_feat"]), nghbr_feat=nb, ref_gmms=sel(inp["ref_gmms"]), nghbr_gmms=ng,
                    nghbr_poses=sel(inp["nghbr_poses"]), is_valid=sel(inp["is_valid"]),
                    cam_intrins={kk: sel(v) for kk, v in inp["cam_intrins"].items()})

    for b in (0, 31, 63):
        one = frame(b)
        alone = _hip_cost(one, k, gpu, feat_dtype="bf16", path=2)
        assert torch.equal(alone[0], full[b]), f"frame {b}: batched launch differs from the single-frame launch"
        assert_cost_parity(alone, oracle_cost(one, k), path=2, label=f"C2 bench-size frame {b}")


def test_full_step_bench_size_batch_invariance(hip_lib, gpu):
    """The whole C2 step at the bench.py batch (64 frames: 1.26 M activation rows per convolution launch) gives, frame by frame,
    exactly what a single-frame step gives: the matcher, the matrix-core stacks with their fused tails, update and upsampling."""
    from magnet_amd.magnet import MAGNET
    wl = synth.WORKLOADS["C2"]
    B = 64
    args = make_args(D=wl.D, iters=1, dpv_h=wl.h, dpv_w=wl.w, V=wl.V)
    model = MAGNET(args, d_net=StubDNet(0), f_net=StubFNet(1), feat_dtype="bf16").to(gpu).eval()
    seeded_magnet_weights(model, seed=6)
    inp = to_dev(synth.make_inputs(wl, B=B, seed=321, round_bf16=True), gpu)
    x_d3 = (torch.randn(B, 256, wl.h, wl.w, generator=torch.Generator().manual_seed(4)) * 0.5).to(gpu)
    with torch.no_grad():
        full = model.match_and_refine(inp["ref_gmms"], x_d3, inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"],
                                      inp["is_valid"], inp["cam_intrins"], mode="test")[-1].clone()
        # run-to-run: every kernel of the step sums in a fixed order and the LDS-DMA rings are ordered by counted waits + barriers,
        # so a second pass over the same inputs is bit-identical everywhere (a rare staging race would show up here)
        for _ in range(2):
            again = model.match_and_refine(inp["ref_gmms"], x_d3, inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"],
                                           inp["is_valid"], inp["cam_intrins"], mode="test")[-1]
            assert torch.equal(again, full)
    # what bench.py times since round 6: the same step captured once and REPLAYED as a HIP graph (magnet_amd/graph.py) — bit-identical
    # to the eager step at the contract's batch, on the first replay and on a later one
    from magnet_amd.graph import GraphedRefine
    g = GraphedRefine(model, inp["ref_gmms"], x_d3, inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"],
                      inp["is_valid"], inp["cam_intrins"], mode="test")
    for _ in range(3):
        rep = g(*g.static)[-1]
        torch.cuda.synchronize()
        assert torch.equal(rep, full), "graph replay differs from the eager step"
    del g
    V = wl.V
    for b in (0, 63):
        sel = lambda t: t[b:b + 1].contiguous()
        nb = inp["nghbr_feat"].view(V, B, *inp["nghbr_feat"].shape[1:])[:, b:b + 1].reshape(V, *inp["nghbr_feat"].shape[1:]).contiguous()
        ng = inp["nghbr_gmms"].view(V, B, *inp["nghbr_gmms"].shape[1:])[:, b:b + 1].reshape(V, *inp["nghbr_gmms"].shape[1:]).contiguous()
        with torch.no_grad():
            one = model.match_and_refine(sel(inp["ref_gmms"]), sel(x_d3), sel(inp["ref_feat"]), nb, ng, sel(inp["nghbr_poses"]),
                                         sel(inp["is_valid"]), {kk: sel(v) for kk, v in inp["cam_intrins"].items()}, mode="test")[-1]
        assert torch.isfinite(one).all() and torch.equal(one[0], full[b]), f"frame {b}"


def test_full_step_bench_size_I3_hoisted_invariant(hip_lib, gpu):
    """BASELINE config 3 at the bench.py batch: 64 frames, I = 3 — the refinement loop with the loop-invariant x_d3 convolution hoisted
    out of the iterations (magnet.py: _refine_mfma).  (a) Frame by frame the three outputs equal a single-frame run bit for bit (no
    cross-frame state in the hoisted partial sums, the per-iteration launches or the matcher); (b) frame 63 against the ORACLE loop on
    the CPU (oracle matcher, torch-CPU G-Net / mask head, oracle update and upsampling; models/MAGNET.py:150-173): abs_rel < 1e-4
    (north_star) for every iteration."""
    from magnet_amd.magnet import MAGNET
    wl = synth.WORKLOADS["C3"]
    B, I = 64, 3
    args = make_args(D=wl.D, iters=I, dpv_h=wl.h, dpv_w=wl.w, V=wl.V)
    model = MAGNET(args, d_net=StubDNet(0), f_net=StubFNet(1), feat_dtype="bf16").eval()
    seeded_magnet_weights(model, seed=8)
    cpu_inp = synth.make_inputs(wl, B=B, seed=654, round_bf16=True)
    x_d3_cpu = torch.randn(B, 256, wl.h, wl.w, generator=torch.Generator().manual_seed(5)) * 0.5
    V, b = wl.V, B - 1
    k = oracle.depth_sampling(3, wl.D)
    with torch.no_grad():                                              # the oracle loop for frame 63 (CPU)
        idx = [v * B + b for v in range(V)]
        case = dict(ref_feat=cpu_inp["ref_feat"][b:b + 1], nghbr_feat=cpu_inp["nghbr_feat"][idx], nghbr_gmms=cpu_inp["nghbr_gmms"][idx],
                    nghbr_poses=cpu_inp["nghbr_poses"][b:b + 1], is_valid=cpu_inp["is_valid"][b:b + 1],
                    cam_intrins={kk: vv[b:b + 1] for kk, vv in cpu_inp["cam_intrins"].items()})
        gmm, x3 = cpu_inp["ref_gmms"][b:b + 1].clone(), x_d3_cpu[b:b + 1]
        mask = model.mask_head(x3)
        want = []
        for _ in range(I):
            cost = torch.from_numpy(oracle_cost(dict(case, ref_gmms=gmm), k))
            raw = model.g_net.gnet(torch.cat([cost, x3], dim=1))
            gmm = torch.from_numpy(oracle.gaussian_update(raw.numpy(), gmm.numpy()))
            want.append(oracle.upsample_depth_via_mask(gmm.numpy(), mask.numpy(), 4))
    model = model.to(gpu)
    inp = to_dev(cpu_inp, gpu)
    x_d3 = x_d3_cpu.to(gpu)
    with torch.no_grad():
        full = [o.clone() for o in model.match_and_refine(inp["ref_gmms"], x_d3, inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"],
                                                          inp["nghbr_poses"], inp["is_valid"], inp["cam_intrins"], mode="test")]
    assert len(full) == I
    for f in (0, b):
        sel = lambda t: t[f:f + 1].contiguous()
        nb = inp["nghbr_feat"].view(V, B, *inp["nghbr_feat"].shape[1:])[:, f:f + 1].reshape(V, *inp["nghbr_feat"].shape[1:]).contiguous()
        ng = inp["nghbr_gmms"].view(V, B, *inp["nghbr_gmms"].shape[1:])[:, f:f + 1].reshape(V, *inp["nghbr_gmms"].shape[1:]).contiguous()
        with torch.no_grad():
            one = model.match_and_refine(sel(inp["ref_gmms"]), sel(x_d3), sel(inp["ref_feat"]), nb, ng, sel(inp["nghbr_poses"]),
                                         sel(inp["is_valid"]), {kk: sel(v) for kk, v in inp["cam_intrins"].items()}, mode="test")
        for i in range(I):
            assert torch.isfinite(one[i]).all() and torch.equal(one[i][0], full[i][f]), f"frame {f}, iteration {i}"
    for i in range(I):
        got = full[i][b:b + 1].cpu().numpy()
        abs_rel = oracle.abs_rel(np.abs(want[i][:, 0]) + 1e-3, np.abs(got[:, 0]) + 1e-3)
        print(f"[C3 bench size, frame {b}, iteration {i}] abs_rel(production HIP loop vs oracle loop) = {abs_rel:.3e}")
        assert abs_rel < 1e-4


def test_refine_with_inputs_in_kernel_layouts(hip_lib, gpu):
    """match_and_refine fed features already in the matcher's layouts and x_d3 already in the G-Net input buffer (what
    backbones on the matrix-core path hand over) returns exactly what the NCHW-input call returns; with and without the
    side-stream repack."""
    from magnet_amd import lib
    from magnet_amd.magnet import MAGNET
    wl = synth.Workload("pk", "scannet", 24, 32, V=2, D=64, F=64)
    inp = synth.make_inputs(wl, B=2, seed=2)
    d = to_dev(inp, gpu)
    x_d3 = torch.randn(2, 256, wl.h, wl.w, generator=torch.Generator().manual_seed(3)).to(gpu) * 0.5
    m = MAGNET(make_args(D=wl.D, iters=2, dpv_h=wl.h, dpv_w=wl.w), d_net=StubDNet(1), f_net=StubFNet(2), feat_dtype="bf16")
    seeded_magnet_weights(m, 5)
    m = m.to(gpu).eval()
    outs = []
    with torch.no_grad():
        for overlap in (True, False):
            m.overlap_pack = overlap
            outs.append(m.match_and_refine(d["ref_gmms"], x_d3, d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"],
                                           d["is_valid"], d["cam_intrins"], mode="test"))
        packed = (lib.pack_features(d["ref_feat"], lib.feat_enum("bf16"), pad=0), lib.pack_features(d["nghbr_feat"], lib.feat_enum("bf16"), pad=1))
        gh, gl, ctot, coff = m.gnet_input_buffer(2, wl.h, wl.w, gpu)
        lib.pack_split(x_d3, gh, gl, ctot, coff)
        outs.append(m.match_and_refine(d["ref_gmms"], None, None, None, d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"],
                                       d["cam_intrins"], mode="test", packed_feats=packed, x_d3_in_place=True))
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert len(o) == 2 and all(torch.equal(a, b) for a, b in zip(o, outs[0]))
