"""The CPU oracle against the golden vectors captured from the reference.
The oracle is the checker for every GPU parity test, so it is pinned first — bit-for-bit where
the arithmetic is integer-free fp32 with a defined order (G2/G3), to rounding elsewhere."""
import hashlib

import numpy as np
import pytest

from magnet_amd import synth
from oracle import oracle


def _sha(*arrays):
    m = hashlib.sha256()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(bytes.fromhex(m.hexdigest()), dtype=np.uint8)


@pytest.mark.parametrize("D", [5, 16, 64, 128])
def test_G1_depth_sampling(golden, D):
    k = np.array(oracle.depth_sampling(3, D))
    assert k.shape == (D,)
    np.testing.assert_allclose(k, golden[f"G1_k_D{D}"], rtol=0, atol=2e-15)
    assert np.all(np.diff(k) > 0) and abs(k[0] + k[-1]) < 1e-12     # sorted, symmetric


def test_G1_shipped_values():
    # D=5 -> [-1.9194,-0.5457,0,0.5457,1.9194]
    k = oracle.depth_sampling(3, 5)
    np.testing.assert_allclose(k, [-1.9194, -0.5457, 0.0, 0.5457, 1.9194], atol=5e-5)


def _tiny_inputs(g):
    return dict(ref_feat=g["G2_ref_feat"], nghbr_feat=g["G2_nghbr_feat"], ref_gmms=g["G2_ref_gmms"],
                nghbr_gmms=g["G2_nghbr_gmms"], poses=g["G2_nghbr_poses"], is_valid=g["G2_is_valid"],
                intM=g["G2_intM"], rays=g["G2_rays"])


def test_A2_depth_volume_bitwise(golden):
    dv = oracle.depth_volume(golden["G2_ref_gmms"], golden["G1_k_D5"])
    assert np.array_equal(dv, golden["G2_d_volume"])


@pytest.mark.parametrize("fused", [True, False])
def test_G2_tiny_bitwise(golden, fused):
    """B=2,V=3,F=8,12x16,D=5 incl. an invalid view, a 90-degree/behind-camera pose and sigma=3 m."""
    t = _tiny_inputs(golden)
    out = oracle.cost_volume_cw(None if fused else golden["G2_d_volume"], t["ref_gmms"], golden["G1_k_D5"],
                                t["ref_feat"], t["nghbr_feat"], t["nghbr_gmms"], t["poses"], t["is_valid"],
                                t["intM"], t["rays"], 5.0)
    assert np.array_equal(out, golden["G2_cost"]), np.abs(out - golden["G2_cost"]).max()


def test_G2_reference_signature(golden):
    t = _tiny_inputs(golden)
    R = t["poses"][:, :, :3, :3]; tt = t["poses"][:, :, :3, 3]
    out = oracle.est_costvolume_CW(golden["G2_d_volume"], t["ref_feat"], t["nghbr_feat"], t["ref_gmms"],
                                   t["nghbr_gmms"], R, tt, t["is_valid"],
                                   {"intM": t["intM"], "unit_ray_array_2D": t["rays"]}, 5)
    assert np.array_equal(out, golden["G2_cost"])


def test_G3_per_view_fp64_and_gates(golden):
    t = _tiny_inputs(golden)
    out, gates, fc = oracle.cost_volume_cw(None, t["ref_gmms"], golden["G1_k_D5"], t["ref_feat"],
                                           t["nghbr_feat"], t["nghbr_gmms"], t["poses"], t["is_valid"],
                                           t["intM"], t["rays"], 5.0, return_aux=True)
    weighted = fc.astype(np.float64) * gates
    assert np.array_equal(weighted, golden["G3_weighted_cost_f64"])
    assert np.array_equal((weighted != 0).astype(np.uint8), golden["G3_gate_nonzero"])
    assert gates[1, 1].sum() == 0                      # the invalid view contributes nothing
    assert 0.1 < gates.mean() < 0.9                    # neither all-pass nor all-fail


def test_G2_invalid_view_still_divides_by_V(golden):
    """homography.py:120 divides by ALL views: dropping a view's validity scales nothing else."""
    t = _tiny_inputs(golden)
    iv = t["is_valid"].copy(); iv[:] = 0; iv[0, 0] = 1
    out = oracle.cost_volume_cw(None, t["ref_gmms"], golden["G1_k_D5"], t["ref_feat"], t["nghbr_feat"],
                                t["nghbr_gmms"], t["poses"], iv, t["intM"], t["rays"], 5.0)
    assert np.all(out[1] == 0)
    _, gates, fc = oracle.cost_volume_cw(None, t["ref_gmms"], golden["G1_k_D5"], t["ref_feat"], t["nghbr_feat"],
                                         t["nghbr_gmms"], t["poses"], iv, t["intM"], t["rays"], 5.0, return_aux=True)
    np.testing.assert_array_equal(out[0], (fc[0, 0].astype(np.float64) * gates[0, 0]).astype(np.float32) / np.float32(3))


def test_G2_C1_shape_subsample_bitwise(golden):
    """Config 1 of BASELINE.json: 128x160, V=2, D=16, F=64 — inputs regenerated from the seed."""
    wl = synth.WORKLOADS["C1"]
    inp = synth.make_inputs(wl, B=1, seed=0)
    sha = _sha(inp["ref_feat"].numpy(), inp["nghbr_feat"].numpy(), inp["ref_gmms"].numpy(),
               inp["nghbr_gmms"].numpy(), inp["nghbr_poses"].numpy())
    assert np.array_equal(sha, golden["G2_C1_input_sha"]), "synthetic generator drifted from the golden inputs"
    out = oracle.cost_volume_cw(None, inp["ref_gmms"], golden["G1_k_D16"], inp["ref_feat"], inp["nghbr_feat"],
                                inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"],
                                inp["cam_intrins"]["intM"], inp["cam_intrins"]["unit_ray_array_2D"], 5.0)
    assert np.array_equal(out[:, :, ::5, ::7], golden["G2_C1_cost_sub"])
    s = golden["G2_C1_cost_sum"]
    np.testing.assert_allclose([out.astype(np.float64).sum(), np.abs(out).astype(np.float64).sum()], s, rtol=1e-12)


def test_G4_gaussian_update(golden):
    out = oracle.gaussian_update(golden["G4_raw"], golden["G4_prev"])
    np.testing.assert_allclose(out, golden["G4_out"], rtol=0, atol=1e-6)
    assert np.all(out[:, 1] > 0)


def test_G5_upsample(golden):
    out = oracle.upsample_depth_via_mask(golden["G5_depth"], golden["G5_mask"], 4)
    assert out.shape == golden["G5_out"].shape == (2, 2, 24, 32)
    np.testing.assert_allclose(out, golden["G5_out"], rtol=0, atol=3e-6)


def test_G7_metrics(golden):
    m = oracle.compute_depth_errors(golden["G7_gt"], golden["G7_pred"], golden["G7_var"].copy())
    for k, v in zip(golden["G7_keys"], golden["G7_vals"]):
        assert abs(float(m[str(k)]) - v) <= 1e-12 * max(1.0, abs(v)), k


def test_G8_relative_poses(golden):
    exts = golden["G8_exts"]
    poses, valid = oracle.relative_poses(exts[2], [exts[i] for i in (0, 1, 3, 4)])
    assert np.array_equal(valid, golden["G8_valid"])
    np.testing.assert_allclose(poses, golden["G8_poses"], rtol=0, atol=1e-6)
    assert valid[2].sum() == 0 and valid[1, 0] == 0      # NaN reference / NaN neighbour


# ---- G9: est_costvolume_F (F-Net training volume) and its autograd gradients -------------------------
def _g9(g):
    return (g["G9_d_center"], g["G9_ref_feat"], g["G9_nghbr_feat"], g["G9_poses"], g["G9_is_valid"], g["G9_intM"], g["G9_rays"])


def test_G9_costvolume_F_raw_bitwise(golden):
    """B=2,V=3,F=8,12x16,D=10 incl. an invalid view and a strong-parallax pose: the volume before the softmax."""
    raw = oracle.cost_volume_f_raw(*_g9(golden))
    assert np.array_equal(raw, golden["G9_raw"])


def test_G9_costvolume_F_softmax(golden):
    p = golden["G9_poses"]
    cam = {"intM": golden["G9_intM"], "unit_ray_array_2D": golden["G9_rays"]}
    sm = oracle.est_costvolume_F(golden["G9_d_center"], golden["G9_ref_feat"], golden["G9_nghbr_feat"],
                                 p[:, :, :3, :3], p[:, :, :3, 3], golden["G9_is_valid"], cam)
    np.testing.assert_allclose(sm, golden["G9_softmax"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(sm.sum(axis=1), 1.0, atol=1e-6)


def test_G9_costvolume_F_gradients(golden):
    """The oracle's analytic fp64 gradients against the reference's autograd (fp32) on the same upstream gradient."""
    _, gr, gs = oracle.cost_volume_f_raw(*_g9(golden), gout=golden["G9_gout"])
    np.testing.assert_allclose(gr, golden["G9_grad_ref"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(gs, golden["G9_grad_src"], rtol=1e-5, atol=2e-6)
    # invalid view (frame 0, view 2) receives exactly zero gradient
    B = golden["G9_ref_feat"].shape[0]
    assert not gs[2 * B + 0].any() and not golden["G9_grad_src"][2 * B + 0].any()


BASELINE_GOLDEN = (("C2", 0, True), ("C4", 0, False), ("C5", 2, False))      # name, seed, features rounded to bf16


def baseline_golden_inputs(name, seed, bf16):
    wl = synth.WORKLOADS[name]
    return wl, synth.make_inputs(wl, B=1, seed=seed, round_bf16=bf16)


@pytest.mark.parametrize("name,seed,bf16", BASELINE_GOLDEN)
def test_G2_baseline_shapes_bitwise(golden_r2, name, seed, bf16):
    """The reference's est_costvolume_CW output at the full C2 (bf16-rounded features) / C4 / C5 shapes: the oracle reproduces
    the stored subsample and checksums bit for bit (the seeded inputs are identified by their sha256)."""
    wl, inp = baseline_golden_inputs(name, seed, bf16)
    assert np.array_equal(_sha(inp["ref_feat"].numpy(), inp["nghbr_feat"].numpy(), inp["ref_gmms"].numpy(),
                               inp["nghbr_gmms"].numpy(), inp["nghbr_poses"].numpy()), golden_r2[f"G2_{name}_input_sha"])
    out = oracle.cost_volume_cw(None, inp["ref_gmms"], oracle.depth_sampling(3, wl.D), inp["ref_feat"], inp["nghbr_feat"],
                                inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"], inp["cam_intrins"]["intM"],
                                inp["cam_intrins"]["unit_ray_array_2D"], 5.0)
    assert np.array_equal(out[:, ::3, ::5, ::7], golden_r2[f"G2_{name}_cost_sub"])
    s = golden_r2[f"G2_{name}_cost_sum"]
    assert out.astype(np.float64).sum() == s[0] and np.abs(out).astype(np.float64).sum() == s[1]
