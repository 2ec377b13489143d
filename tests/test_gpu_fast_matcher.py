"""-m gpu: the PRODUCTION matcher (cost_volume_fast.hip, `path` 0/4) — tolerance parity against the oracle.

The production kernel does not reproduce the reference's fp32 rounding sequence (homography.py:131-148): parity is
(a) gate-flip fraction <= 1e-5 against the oracle's gate bits (homography.py:157-158), read back through the ABI's
`gate_bits` debug output, (b) |hip - oracle| <= 2e-5 + 2e-5 |oracle| on every entry none of whose gates flipped,
(c) abs_rel of the refinement loop's depth < 1e-4 (north_star).  The exact kernels (path 1/2/3) keep their bitwise /
zero-flip tests in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from magnet_amd import synth
from oracle import oracle
from tests.parity import assert_tolerant_parity, flip_scale, oracle_cost, pos_eps, position_sensitivity, to_dev
from tests.stubs import StubDNet, StubFNet, make_args, seeded_magnet_weights

pytestmark = pytest.mark.gpu


def _run(inp, k_list, device, feat_dtype="fp32", path=4, want_gates=True, kappa=5):
    from magnet_amd.homography import CostVolumeCW
    d = to_dev(inp, device)
    cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"],
                      d["cam_intrins"], kappa, feat_dtype=feat_dtype, path=path)
    B, F, h, w = inp["ref_feat"].shape
    V = inp["nghbr_feat"].shape[0] // B
    gates = torch.zeros(B, V, len(k_list), h, w, dtype=torch.uint8, device=device) if want_gates else None
    cost = cv(ref_gmm=d["ref_gmms"], k_list=k_list, gate_bits=gates)
    return cost, gates


def _check(inp, k, gpu, fdt="fp32", label="", path=4):
    orc, og, _ = oracle_cost(inp, k, aux=True)
    cost, gates = _run(inp, k, gpu, feat_dtype=fdt, path=path)
    h, w = inp["ref_feat"].shape[-2:]
    sens = position_sensitivity(inp, k, og, device=gpu)
    st = assert_tolerant_parity(cost, orc, gates, og, label=label, sens=sens, eps=pos_eps(h, w), flip_rate_scale=flip_scale(h, w))
    # the production launch (no debug output) is a different template instance: it must give the same volume
    plain, _ = _run(inp, k, gpu, feat_dtype=fdt, path=path, want_gates=False)
    assert torch.equal(plain, cost), f"{label}: gate-bit instance and production instance differ"
    return st


def _golden_tiny(g):
    return dict(ref_feat=torch.from_numpy(g["G2_ref_feat"]), nghbr_feat=torch.from_numpy(g["G2_nghbr_feat"]),
                ref_gmms=torch.from_numpy(g["G2_ref_gmms"]), nghbr_gmms=torch.from_numpy(g["G2_nghbr_gmms"]),
                nghbr_poses=torch.from_numpy(g["G2_nghbr_poses"]), is_valid=torch.from_numpy(g["G2_is_valid"]),
                cam_intrins={"intM": torch.from_numpy(g["G2_intM"]), "unit_ray_array_2D": torch.from_numpy(g["G2_rays"])})


def test_fast_tiny_golden(hip_lib, gpu, golden):
    """The reference's own output on the edge-case vector (invalid view, behind-camera pose, out of bounds)."""
    inp = _golden_tiny(golden)
    cost, _ = _run(inp, list(golden["G1_k_D5"]), gpu, want_gates=False)
    _, og, _ = oracle_cost(inp, list(golden["G1_k_D5"]), aux=True)
    sens = position_sensitivity(inp, list(golden["G1_k_D5"]), og, device=gpu)
    assert_tolerant_parity(cost, golden["G2_cost"], n_views=2, label="tiny golden (reference output)", sens=sens, eps=pos_eps(12, 16))
    _check(inp, list(golden["G1_k_D5"]), gpu, label="tiny golden")


def test_fast_gate_bits_match_reference_G3(hip_lib, gpu, golden):
    """G3 = the non-zero pattern of the reference's `_compute_cost_CW` output (= its gate bits wherever the feature cost is
    non-zero) for every valid (b, v) of the tiny vector: the `gate_bits` output of both candidate-lane kernels must
    reproduce it — the exact kernel bit for bit, the production matcher up to one flipped gate."""
    inp = _golden_tiny(golden)
    ref = golden["G3_gate_nonzero"].astype(bool)                                   # (B,V,D,h,w), zeros for the invalid view
    _, g_fast = _run(inp, list(golden["G1_k_D5"]), gpu, path=4)
    _, g_exact = _run(inp, list(golden["G1_k_D5"]), gpu, path=2)
    assert np.array_equal(g_exact.cpu().numpy().astype(bool), ref)
    assert int((g_fast.cpu().numpy().astype(bool) != ref).sum()) <= 1


def test_fast_C1_golden_subsample(hip_lib, gpu, golden):
    wl = synth.WORKLOADS["C1"]
    inp = synth.make_inputs(wl, B=1, seed=0)
    k = list(golden["G1_k_D16"])
    cost, _ = _run(inp, k, gpu, want_gates=False)
    _, og, _ = oracle_cost(inp, k, aux=True)
    sens = position_sensitivity(inp, k, og, device=gpu)[:, :, ::5, ::7]
    assert_tolerant_parity(cost.cpu().numpy()[:, :, ::5, ::7], golden["G2_C1_cost_sub"], n_views=2, label="C1 golden",
                           sens=sens, eps=pos_eps(wl.h, wl.w))


CASES = [
    # name, workload, B, seed, feat_dtype, invalid
    ("C1", "C1", 1, 0, "fp32", ()),
    ("C1-b2-invalid", "C1", 2, 1, "fp32", ((0, 1),)),
    ("C2-fp32", "C2", 1, 0, "fp32", ()),
    ("C2-bf16", "C2", 2, 1, "bf16", ((1, 2),)),
    ("C4", "C4", 1, 0, "fp32", ()),
    ("C5", "C5", 1, 2, "fp32", ()),
    ("shipped-D5", "shipped", 2, 0, "fp32", ()),
    # bf16 F = 64 at D > 32 (cost_volume_v3.hip's production instance): C4 (D = 128: two candidate blocks, long KITTI segments) and
    # C5 (V = 6, an invalid view: the compacted view table)
    ("C4-bf16", "C4", 1, 0, "bf16", ()),
    ("C5-bf16-invalid", "C5", 2, 2, "bf16", ((0, 3), (1, 0))),
]


@pytest.mark.parametrize("name,wlname,B,seed,fdt,invalid", CASES)
def test_fast_vs_oracle_baseline_shapes(hip_lib, gpu, name, wlname, B, seed, fdt, invalid):
    """Every BASELINE.json shape: gate-flip fraction <= 1e-5, value tolerance elsewhere."""
    wl = synth.WORKLOADS[wlname]
    inp = synth.make_inputs(wl, B=B, seed=seed, invalid=list(invalid), round_bf16=(fdt == "bf16"))
    _check(inp, oracle.depth_sampling(3, wl.D), gpu, fdt=fdt, label=name)


def test_fast_ragged_grid_and_odd_D(hip_lib, gpu):
    wl = synth.Workload("ragged", "scannet", 13, 19, V=1, D=7, F=8)
    inp = synth.make_inputs(wl, B=3, seed=4)
    _check(inp, oracle.depth_sampling(3, wl.D), gpu, label="ragged")


@pytest.mark.parametrize("V,D,F,fdt", [(6, 24, 32, "bf16"), (1, 256, 8, "fp32"), (3, 40, 128, "fp32"), (2, 9, 128, "bf16"),
                                        (8, 16, 72, "fp32"), (3, 64, 32, "bf16"), (2, 130, 128, "bf16"), (5, 70, 64, "bf16"),
                                        (2, 64, 48, "bf16")])
def test_fast_shape_sweep(hip_lib, gpu, V, D, F, fdt):
    """Channel counts off the F = 64 path, several candidate blocks, matrix-pipe correlation with 1 / 2 / 4 K steps."""
    wl = synth.Workload("sweep", "7scenes", 10, 23, V=V, D=D, F=F)
    inp = synth.make_inputs(wl, B=2, seed=V * 100 + D, round_bf16=(fdt == "bf16"), invalid=[(1, 0)])
    _check(inp, oracle.depth_sampling(3, D), gpu, fdt=fdt, label=f"sweep V={V} D={D} F={F} {fdt}")


def test_many_views_small_D_lds_budget(hip_lib, gpu):
    """D <= 32 stages reference vectors (8 pixels) in LDS next to the per-view tables: with V = 25, fp32 F = 64 that is more
    than 64 KB per workgroup.  `path = 4` (production or error) must say so, `path = 0` must fall back to an exact kernel and
    still give the oracle's volume; V = 12 fits and runs the production kernel."""
    from magnet_amd import lib
    wl = synth.Workload("views", "7scenes", 10, 23, V=25, D=5, F=64)
    inp = synth.make_inputs(wl, B=1, seed=77, invalid=[(0, 3)])
    k = oracle.depth_sampling(3, wl.D)
    with pytest.raises(lib.MagnetError):
        _run(inp, k, gpu, path=4, want_gates=False)
    cost, _ = _run(inp, k, gpu, path=0, want_gates=False)
    orc = oracle_cost(inp, k)
    np.testing.assert_allclose(cost.cpu().numpy(), orc, rtol=2e-5, atol=2e-5)
    wl12 = synth.Workload("views12", "7scenes", 10, 23, V=12, D=5, F=64)
    _check(synth.make_inputs(wl12, B=1, seed=78), k, gpu, label="V=12 D=5")


def test_wide_grid_runs_the_batched_view_kernel(hip_lib, gpu):
    """Matching grids wider than 512 (full-resolution grids: long epipolar segments) are routed to the round-2 D > 32 kernel
    (cost_volume_fast64.hip), which reads the interleaved (mu, sigma) map: same tolerance contract; both output forms."""
    wl = synth.Workload("wide", "kitti", 6, 528, V=3, D=48, F=64)
    inp = synth.make_inputs(wl, B=2, seed=91, invalid=[(1, 2)])
    k = oracle.depth_sampling(3, wl.D)
    _check(inp, k, gpu, label="wide grid (w = 528) fp32")
    _check(synth.make_inputs(wl, B=1, seed=92, round_bf16=True), k, gpu, fdt="bf16", label="wide grid (w = 528) bf16")


@pytest.mark.parametrize("name,fdt", [("C2L", "bf16"), ("C4L", "fp32")])
def test_full_resolution_grids_vs_oracle(hip_lib, gpu, name, fdt):
    """BASELINE.json's "640x480 warp kernel" / "352x1216 wide-aspect warp stress" taken literally: the matcher at the FULL-resolution
    matching grids C2L (480 x 640, D = 64, bf16) and C4L (352 x 1216, D = 128, fp32), one frame, for whichever kernel launch_cv_fast
    routes them to (w > 512: cost_volume_fast64.hip): gate bits against the oracle's (flip fraction <= 1e-5) and the value tolerance."""
    wl = synth.WORKLOADS[name]
    inp = synth.make_inputs(wl, B=1, seed=5, round_bf16=(fdt == "bf16"))
    st = _check(inp, oracle.depth_sampling(3, wl.D), gpu, fdt=fdt, label=f"{name} full grid")
    assert st["gate_flip_frac"] <= 1e-5 * flip_scale(wl.h, wl.w)        # the contract rate scales with the grid (tests/parity.py: flip_scale)


def _axis_poses(B, V, step):
    """Translations along +x, -x, +y, -y (repeating), no rotation: the candidates of a pixel travel along one image axis, in both
    directions — the four (mode, direction) cases of the texel-pair item lists (cost_volume_fast.hip / cost_volume_fast64.hip).  A 7 %
    component along the other axis slants the segments a little: with an exactly axis-parallel translation every sample would sit ON a
    texel boundary of the other axis (v = y + 0.5 for all depths), where a 1-ulp difference picks the other quad and the
    position-sensitivity model of tests/parity.py (the slope inside the oracle's quad) does not describe the difference."""
    poses = torch.eye(4).repeat(B, V, 1, 1)
    for v in range(V):
        major = v // 2 % 2
        sgn = step * (1.0 if v % 2 == 0 else -1.0) * (1.0 + 0.25 * (v // 4))
        poses[:, v, major, 3] = sgn
        poses[:, v, 1 - major, 3] = 0.07 * sgn * (1.0 if v % 3 else -1.0)
    return poses


@pytest.mark.parametrize("name,h,w,V,D,F,fdt,step", [
    ("D16 per-view kernel", 24, 40, 4, 16, 64, "fp32", 0.25),
    ("D5 per-view kernel, bf16", 20, 36, 4, 5, 32, "bf16", 0.4),
    ("D64 batched views, wide grid, bf16 pair items", 10, 528, 4, 64, 64, "bf16", 2.0),
    ("D64 batched views, wide grid, fp32 quad items", 10, 528, 4, 64, 64, "fp32", 2.0),
    ("D128 long segments, table overflow -> view by view", 6, 640, 8, 128, 64, "bf16", 6.0),
])
def test_pair_items_all_travel_directions(hip_lib, gpu, name, h, w, V, D, F, fdt, step):
    """Round 5: texel-pair items.  Source views translated along +x / -x / +y / -y make every (row / column mode, up / down direction)
    case of the shared-pair bookkeeping run in one launch; a long baseline with wide sigma stretches the segments over many quads (the
    last case makes a view group need more than the 128 table entries, i.e. the view-by-view path).  Same tolerance contract."""
    wl = synth.Workload("axes", "scannet", h, w, V=V, D=D, F=F)
    inp = synth.make_inputs(wl, B=2, seed=17, round_bf16=(fdt == "bf16"), invalid=[(1, 1)])
    inp["nghbr_poses"] = _axis_poses(2, V, step)
    inp["ref_gmms"][:, 1] *= 2.0                                      # wider candidate spread: more distinct quads per pixel
    _check(inp, oracle.depth_sampling(3, D), gpu, fdt=fdt, label=name)


def test_large_batch_key_range(hip_lib, gpu):
    """launch_cv_v3 declines batches whose quad keys over all views of a frame ((V - 1) * B * map + map) do not fit its 24-bit
    multiply: V = 4, 30 x 40 grid (map = 1344), B = 4200 -> 1.69e7 > 2^24, so the call falls to the round-2 kernel,
    whose keys are per map.  The last two frames are compared with the oracle run on those frames alone."""
    wl = synth.Workload("bigB", "scannet", 30, 40, V=4, D=64, F=16)
    B = 4200
    one = synth.make_inputs(wl, B=2, seed=31, round_bf16=True)
    k = oracle.depth_sampling(3, wl.D)
    rep = lambda t, n: torch.cat([t[:1].expand(n - 2, *t.shape[1:]), t], dim=0)
    inp = dict(one)
    inp["ref_feat"] = rep(one["ref_feat"], B); inp["ref_gmms"] = rep(one["ref_gmms"], B)
    V = wl.V
    nf = one["nghbr_feat"].view(V, 2, *one["nghbr_feat"].shape[1:]); ng = one["nghbr_gmms"].view(V, 2, *one["nghbr_gmms"].shape[1:])
    inp["nghbr_feat"] = torch.cat([nf[:, :1].expand(V, B - 2, *nf.shape[2:]), nf], dim=1).reshape(V * B, *nf.shape[2:])
    inp["nghbr_gmms"] = torch.cat([ng[:, :1].expand(V, B - 2, *ng.shape[2:]), ng], dim=1).reshape(V * B, *ng.shape[2:])
    inp["nghbr_poses"] = rep(one["nghbr_poses"], B); inp["is_valid"] = rep(one["is_valid"], B)
    inp["cam_intrins"] = {kk: rep(v, B) for kk, v in one["cam_intrins"].items()}
    cost, _ = _run(inp, k, gpu, feat_dtype="bf16", path=4, want_gates=False)
    orc, og, _ = oracle_cost(one, k, aux=True)
    sens = position_sensitivity(one, k, og, device=gpu)
    assert_tolerant_parity(cost[-2:], orc, label="B = 4200 (last two frames)", sens=sens, eps=pos_eps(wl.h, wl.w), n_views=V)
    assert torch.equal(cost[0], cost[1])                               # identical frames, identical results


def test_quad_only_call_on_a_shape_the_quad_kernel_declines_is_E_SHAPE(hip_lib, gpu):
    """C ABI contract (include/magnet_hip.h): with only the quad-form (mu, sigma) map given, a shape that ends on a kernel reading
    the interleaved map returns MAGNET_E_SHAPE — the one code a caller may answer by packing the other map and calling again."""
    from magnet_amd import lib
    wl = synth.Workload("wide", "kitti", 6, 528, V=2, D=40, F=64)
    d = to_dev(synth.make_inputs(wl, B=1, seed=93), gpu)
    k = oracle.depth_sampling(3, wl.D)
    ref_cl = lib.pack_features(d["ref_feat"], lib.feat_enum("fp32"), pad=0)
    src_pad = lib.pack_features(d["nghbr_feat"], lib.feat_enum("fp32"), pad=1)
    quad = lib.pack_gmm_quad(d["nghbr_gmms"])
    intr = d["cam_intrins"]
    with pytest.raises(lib.MagnetError) as ei:
        lib.cost_volume_cw(ref_cl, src_pad, None, d["nghbr_poses"], d["is_valid"].to(gpu).int(), intr["intM"].to(gpu).float(),
                           intr["unit_ray_array_2D"].to(gpu).float().contiguous(), 5.0, ref_gmm=d["ref_gmms"], k_list=k, path=4, src_gmm_quad=quad)
    assert ei.value.code == lib.E_SHAPE
    both = lib.cost_volume_cw(ref_cl, src_pad, lib.pack_gmm(d["nghbr_gmms"]), d["nghbr_poses"], d["is_valid"].to(gpu).int(), intr["intM"].to(gpu).float(),
                              intr["unit_ray_array_2D"].to(gpu).float().contiguous(), 5.0, ref_gmm=d["ref_gmms"], k_list=k, path=4, src_gmm_quad=quad)
    assert torch.isfinite(both).all()


def test_fast_nan_and_degenerate_inputs(hip_lib, gpu):
    """NaN / zero sigma / zero depth in the reference gmm and a singular pose (test_gpu_parity.py's case)."""
    wl = synth.Workload("nan", "scannet", 12, 16, V=2, D=8, F=8)
    inp = synth.make_inputs(wl, B=2, seed=21)
    inp["ref_gmms"][0, 0, 3, 4] = float("nan")
    inp["ref_gmms"][0, 1, 5, 6] = float("nan")
    inp["ref_gmms"][1, 1, 2, :] = 0.0
    inp["ref_gmms"][1, 0, 7, :] = 0.0
    inp["nghbr_poses"][1, 1, :3, :3] = 0.0
    _check(inp, oracle.depth_sampling(3, 8), gpu, label="nan/degenerate")


def test_fast_all_views_invalid_is_zero(hip_lib, gpu):
    wl = synth.Workload("inv", "scannet", 12, 16, V=2, D=5, F=8)
    inp = synth.make_inputs(wl, B=1, seed=5, invalid=[(0, 0), (0, 1)])
    got, _ = _run(inp, oracle.depth_sampling(3, 5), gpu, want_gates=False)
    assert torch.count_nonzero(got) == 0


def test_fast_auto_path_selects_production_matcher(hip_lib, gpu):
    """path 0 = production matcher for fused sampling; with an explicit d_volume it must fall to the exact kernel."""
    from magnet_amd.homography import CostVolumeCW
    wl = synth.Workload("sel", "scannet", 24, 32, V=2, D=16, F=16)
    inp = synth.make_inputs(wl, B=1, seed=9)
    k = oracle.depth_sampling(3, wl.D)
    a, _ = _run(inp, k, gpu, path=0, want_gates=False)
    b, _ = _run(inp, k, gpu, path=4, want_gates=False)
    assert torch.equal(a, b)
    d = to_dev(inp, gpu)
    cv0 = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"], d["cam_intrins"], 5, path=0)
    cv2 = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"], d["cam_intrins"], 5, path=2)
    dv = synth.depth_volume_from_gmm(inp["ref_gmms"], k).to(gpu)
    assert torch.equal(cv0(d_volume=dv), cv2(d_volume=dv))
    from magnet_amd import lib
    cv4 = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"], d["cam_intrins"], 5, path=4)
    with pytest.raises(lib.MagnetError, match="production matcher needs fused sampling"):
        cv4(d_volume=dv)


def test_fast_split_output_equals_dense(hip_lib, gpu):
    """cost_hi/cost_lo written by the production matcher = split of its own fp32 volume (D = 64 matrix-pipe form, D = 5)."""
    from magnet_amd.convnet import split_bf16
    from magnet_amd.homography import CostVolumeCW
    for D, F, fdt in ((64, 64, "bf16"), (5, 8, "fp32")):
        wl = synth.Workload("sp", "scannet", 13, 19, V=2, D=D, F=F)
        inp = synth.make_inputs(wl, B=2, seed=11, round_bf16=(fdt == "bf16"))
        d = to_dev(inp, gpu)
        k = oracle.depth_sampling(3, D)
        cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"], d["cam_intrins"], 5,
                          feat_dtype=fdt, path=4)
        dense = cv(ref_gmm=d["ref_gmms"], k_list=k)
        ld = 96
        hi = torch.full((2 * 15 * 21, ld), 7.0, dtype=torch.bfloat16, device=gpu); lo = torch.full_like(hi, 7.0)
        cv(ref_gmm=d["ref_gmms"], k_list=k, out_split=(hi, lo, ld))
        eh, el = split_bf16(dense.permute(0, 2, 3, 1).contiguous())
        hi4 = hi.view(2, 15, 21, ld); lo4 = lo.view(2, 15, 21, ld)
        assert torch.equal(hi4[:, 1:-1, 1:-1, :D], eh) and torch.equal(lo4[:, 1:-1, 1:-1, :D], el)
        poison = torch.full((1,), 7.0, dtype=torch.bfloat16, device=gpu)
        assert torch.all(hi4[:, 1:-1, 1:-1, D:] == poison) and torch.all(hi4[:, 0] == poison) and torch.all(lo4[:, :, -1] == poison)


@pytest.mark.parametrize("fdt,w,step", [("bf16", 200, 1.5), ("fp32", 200, 1.5), ("bf16", 43, 0.3)])
def test_two_pixel_batches_equal_the_one_pixel_loop(hip_lib, gpu, fdt, w, step):
    """Round 5: the split-output form of the D > 32 kernel (what MAGNET.forward runs) correlates the view groups of TWO neighbouring
    pixels in one batch; the (B,D,h,w) fp32 form keeps the one-pixel loop.  Same arithmetic per candidate and per item: the split
    planes must be the bf16 split of the dense volume bit for bit — on long segments too, where a two-pixel batch has more than 64
    items and falls back to view-by-view, then to the pixel alone (D = 128: two candidate blocks; axis-parallel baselines of 1.5 m with
    doubled sigma: segments of ~90 texels), and on a ragged row (w = 43: a last segment of 3 pixels, i.e. a batch with one pixel)."""
    from magnet_amd.convnet import split_bf16
    from magnet_amd.homography import CostVolumeCW
    wl = synth.Workload("px2", "scannet", 6, w, V=4, D=128, F=64)
    inp = synth.make_inputs(wl, B=2, seed=23, round_bf16=(fdt == "bf16"), invalid=[(0, 2)])
    inp["nghbr_poses"] = _axis_poses(2, 4, step)
    inp["ref_gmms"][:, 1] *= 2.0
    d = to_dev(inp, gpu)
    k = oracle.depth_sampling(3, wl.D)
    cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"], d["cam_intrins"], 5, feat_dtype=fdt, path=4)
    dense = cv(ref_gmm=d["ref_gmms"], k_list=k)
    ld = 128 + 256
    hi = torch.zeros((2 * 8 * (w + 2), ld), dtype=torch.bfloat16, device=gpu); lo = torch.zeros_like(hi)
    cv(ref_gmm=d["ref_gmms"], k_list=k, out_split=(hi, lo, ld))
    eh, el = split_bf16(dense.permute(0, 2, 3, 1).contiguous())
    hi4 = hi.view(2, 8, w + 2, ld); lo4 = lo.view(2, 8, w + 2, ld)
    assert torch.count_nonzero(dense) > 0.02 * dense.numel()
    assert torch.equal(hi4[:, 1:-1, 1:-1, :128], eh) and torch.equal(lo4[:, 1:-1, 1:-1, :128], el)


def test_fast_batch_independence_at_bench_size(hip_lib, gpu):
    """64 frames per launch (bench.py's step): every frame equals the same frame run alone (no cross-frame state, XCD remap
    bijective), and frames 0 / 63 are within tolerance of the oracle."""
    wl = synth.WORKLOADS["C2"]
    inp = synth.make_inputs(wl, B=64, seed=1234, round_bf16=True)
    k = oracle.depth_sampling(3, wl.D)
    full, _ = _run(inp, k, gpu, feat_dtype="bf16", path=0, want_gates=False)
    V = wl.V
    for b in (0, 31, 63):
        idx = [v * 64 + b for v in range(V)]
        one = dict(ref_feat=inp["ref_feat"][b:b + 1], nghbr_feat=inp["nghbr_feat"][idx], ref_gmms=inp["ref_gmms"][b:b + 1],
                   nghbr_gmms=inp["nghbr_gmms"][idx], nghbr_poses=inp["nghbr_poses"][b:b + 1], is_valid=inp["is_valid"][b:b + 1],
                   cam_intrins={kk: vv[b:b + 1] for kk, vv in inp["cam_intrins"].items()})
        alone, _ = _run(one, k, gpu, feat_dtype="bf16", path=0, want_gates=False)
        assert torch.equal(alone[0], full[b])
        if b != 31:
            _check(one, k, gpu, fdt="bf16", label=f"C2 bench-size frame {b}")


def test_fast_forward_abs_rel_vs_oracle_forward(hip_lib, gpu):
    """north_star: depth maps within abs_rel 1e-4.  MAGNET.forward (C3 shape, B = 2, I = 3, bf16 feature storage, stand-in backbones,
    MFMA convolutions, production matcher) against the ORACLE forward on the CPU: the same backbones' outputs (features rounded to
    bf16 as the kernel stores them), oracle matcher, torch-CPU G-Net / mask head, oracle Gaussian update and upsampling
    (models/MAGNET.py:130-175).  (Rounds 1-3 compared this loop with the exact HIP matcher: HIP against HIP.)"""
    from magnet_amd.magnet import MAGNET
    wl = synth.WORKLOADS["C3"]
    args = make_args(D=wl.D, iters=3, dpv_h=wl.h, dpv_w=wl.w)
    inp = synth.make_inputs(wl, B=2, seed=3, round_bf16=True)
    m = MAGNET(args, d_net=StubDNet(1), f_net=StubFNet(2, fdim=wl.F), feat_dtype="bf16")
    seeded_magnet_weights(m, 3)
    g = torch.Generator().manual_seed(0)
    ref = torch.rand(2, 3, 4 * wl.h, 4 * wl.w, generator=g)
    ngh = torch.rand(2 * wl.V, 3, 4 * wl.h, 4 * wl.w, generator=g)
    k = oracle.depth_sampling(3, wl.D)
    with torch.no_grad():                                              # the oracle forward (CPU)
        gmms, x_d3 = m.d_net(torch.cat((ref, ngh), dim=0))
        feat = m.f_net(torch.cat((ref, ngh), dim=0)).to(torch.bfloat16).float()
        gmm, x3 = gmms[:2].clone(), x_d3[:2]
        case = dict(ref_feat=feat[:2], nghbr_feat=feat[2:], nghbr_gmms=gmms[2:], nghbr_poses=inp["nghbr_poses"], is_valid=inp["is_valid"],
                    cam_intrins=inp["cam_intrins"])
        mask = m.mask_head(x3)
        want = []
        for _ in range(3):
            cost = torch.from_numpy(oracle_cost(dict(case, ref_gmms=gmm), k))
            raw = m.g_net.gnet(torch.cat([cost, x3], dim=1))
            gmm = torch.from_numpy(oracle.gaussian_update(raw.numpy(), gmm.numpy()))
            want.append(oracle.upsample_depth_via_mask(gmm.numpy(), mask.numpy(), 4))
    m = m.to(gpu).eval()
    assert m.matcher_path == 0
    with torch.no_grad():
        outs = [o.cpu().numpy() for o in m(ref.to(gpu), ngh.to(gpu), inp["nghbr_poses"].to(gpu), inp["is_valid"], inp["cam_intrins"], mode="test")]
    for i, (a_, b_) in enumerate(zip(outs, want)):
        abs_rel = oracle.abs_rel(np.abs(b_[:, 0]) + 1e-3, np.abs(a_[:, 0]) + 1e-3)
        print(f"[forward, iteration {i}] abs_rel(production HIP forward vs oracle forward) = {abs_rel:.3e}")
        assert np.isfinite(a_).all() and abs_rel < 1e-4
