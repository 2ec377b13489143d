"""-m gpu: round-3 reference fixtures (golden_r3.npz, made by tests/golden/make_golden_r3.py from the imported reference).

G13  the reference's own MAGNET.forward(mode='test') at D = 64, I = 3, V = 4 (one invalid view), 60 x 80 grid: the D > 32
     PRODUCTION matcher (cost_volume_v3.hip) inside the full loop, against reference-generated numbers (models/MAGNET.py:130-175).
G14  est_costvolume_CW on the SMOOTH synthetic variant at C2 / C4: there the production matcher meets the plain
     2e-5 + 2e-5|c| bound with NO position term — the position term of the default tests is an artefact of white-noise features.
Also here: the production loop against the ORACLE loop at the C4 and C5 shapes."""
import numpy as np
import pytest
import torch

from magnet_amd import synth
from oracle import oracle
from tests.parity import assert_tolerant_parity, oracle_cost, pos_eps, position_sensitivity, to_dev
from tests.stubs import StubDNet, StubFNet, g13_case, make_args, seeded_magnet_weights

pytestmark = pytest.mark.gpu

G13_BAR = 1e-6        # abs_rel of our depth maps against the reference's own MAGNET.forward output (measured <= 1.5e-7)


@pytest.mark.parametrize("backend", ["mfma", "torch"])
def test_magnet_forward_D64_matches_reference_output(hip_lib, gpu, golden_r3, backend):
    """G13: every iteration's output within abs_rel 1e-6 of the reference's (measured 3e-8 .. 1.4e-7; BASELINE.json's parity bar is
    1e-4, but with the reference on the CPU a 0.1 % systematic cost error moves this fixture by only 1.5e-5 .. 3.4e-5, so 1e-4 would
    not see it: test_G13_bar_detects_a_cost_scale_error below is the negative control), sigma within 5e-3 relative; fp32 feature
    storage (the reference cannot run bf16)."""
    from magnet_amd.magnet import MAGNET
    args, ref_img, nghbr_imgs, poses, valid, intr, seeds = g13_case()
    m = MAGNET(args, d_net=StubDNet(seed=seeds["d"]), f_net=StubFNet(seed=seeds["f"], fdim=64), conv_backend=backend, feat_dtype="fp32")
    seeded_magnet_weights(m, seed=seeds["w"], gain=seeds["gain"])
    m = m.to(gpu).eval()
    assert m.matcher_path == 0                                     # the production path
    with torch.no_grad():
        preds = m(ref_img.to(gpu), nghbr_imgs.to(gpu), poses.to(gpu), valid, intr, mode="test")
    assert len(preds) == 3
    for i, p in enumerate(preds):
        got = p.cpu().numpy()
        assert got.shape == (2, 2, 240, 320)
        ref = golden_r3[f"G13_pred{i}_sub"]
        sub = got[:, :, ::4, ::5]
        ar = oracle.abs_rel(ref[:, 0], sub[:, 0])
        print(f"[G13 {backend} iter {i}] abs_rel(ours vs reference mu) = {ar:.3e}; max|dmu| = {np.abs(sub[:, 0] - ref[:, 0]).max():.3e}; "
              f"max rel dsigma = {np.abs(sub[:, 1] / ref[:, 1] - 1).max():.3e}")
        assert np.isfinite(got).all() and ar < G13_BAR
        np.testing.assert_allclose(sub[:, 1], ref[:, 1], rtol=5e-3, atol=1e-6)
        s = golden_r3[f"G13_pred{i}_sum"]
        np.testing.assert_allclose(got.astype(np.float64).sum(), s[0], rtol=1e-4)
        np.testing.assert_allclose(np.abs(got).astype(np.float64).sum(), s[1], rtol=1e-4)


def test_G13_bar_detects_a_cost_scale_error(hip_lib, gpu, golden_r3, monkeypatch):
    """Negative control of G13: the same forward with every cost volume scaled by 1.001 (a 0.1 % systematic matcher error, injected
    behind the exact generic kernel, which writes the NCHW volume the step then repacks) must FAIL the G13 bar in every iteration
    (the reference moves by 1.5e-5 / 2.6e-5 / 3.4e-5 under that scale); unscaled, the same path passes it."""
    from magnet_amd.homography import CostVolumeCW
    from magnet_amd.magnet import MAGNET
    args, ref_img, nghbr_imgs, poses, valid, intr, seeds = g13_case()
    m = MAGNET(args, d_net=StubDNet(seed=seeds["d"]), f_net=StubFNet(seed=seeds["f"], fdim=64), conv_backend="mfma", feat_dtype="fp32")
    seeded_magnet_weights(m, seed=seeds["w"], gain=seeds["gain"])
    m = m.to(gpu).eval()
    m.matcher_path = 1
    orig = CostVolumeCW.__call__
    scale = [1.0]

    def scaled(self, *a, **kw):
        res = orig(self, *a, **kw)
        if kw.get("out") is not None and scale[0] != 1.0:
            kw["out"].mul_(scale[0])
        return res
    monkeypatch.setattr(CostVolumeCW, "__call__", scaled)
    for sc, must_pass in ((1.0, True), (1.001, False)):
        scale[0] = sc
        with torch.no_grad():
            preds = m(ref_img.to(gpu), nghbr_imgs.to(gpu), poses.to(gpu), valid, intr, mode="test")
        for i, p in enumerate(preds):
            ar = oracle.abs_rel(golden_r3[f"G13_pred{i}_sub"][:, 0], p.cpu().numpy()[:, :, ::4, ::5][:, 0])
            print(f"[G13 negative control, cost x {sc}, iter {i}] abs_rel = {ar:.3e} (bar {G13_BAR:.0e})")
            assert (ar < G13_BAR) == must_pass


def _run(inp, k_list, device, feat_dtype, path=4):
    from magnet_amd.homography import CostVolumeCW
    d = to_dev(inp, device)
    cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"],
                      d["cam_intrins"], 5, feat_dtype=feat_dtype, path=path)
    B, F, h, w = inp["ref_feat"].shape
    V = inp["nghbr_feat"].shape[0] // B
    gates = torch.zeros(B, V, len(k_list), h, w, dtype=torch.uint8, device=device)
    return cv(ref_gmm=d["ref_gmms"], k_list=k_list, gate_bits=gates), gates


@pytest.mark.parametrize("name,bf16", [("C2", True), ("C4", False)])
def test_production_matcher_plain_bound_on_smooth_inputs(hip_lib, gpu, golden_r3, name, bf16):
    """Smooth features / (mu, sigma) maps (synth.make_inputs smooth=6).  The full contract holds as everywhere; on top of it
      * fewer than 1e-3 of the entries are outside the PLAIN bound 2e-5 + 2e-5|c| (no position term) — measured 2e-4 (C2) and
        7e-4 (C4); on the default white-noise inputs it is 10 - 15 %;
      * none of those is an entry whose score is locally flat according to the pinned fp64 model (S * eps <= 1e-5,
        tests/test_parity_model.py): what is left sits where the bilinear field is steep whatever the features are (the zero
        border of the source maps);
      * the same against the reference's own subsample (G14).
    I.e. the position term of the tolerance is an artefact of texel-to-texel independent features, not kernel error."""
    wl = synth.WORKLOADS[name]
    inp = synth.make_inputs(wl, B=1, seed=3, round_bf16=bf16, smooth=6)
    k = oracle.depth_sampling(3, wl.D)
    orc, og, _ = oracle_cost(inp, k, aux=True)
    cost, gates = _run(inp, k, gpu, "bf16" if bf16 else "fp32")
    sens = position_sensitivity(inp, k, og, device=gpu)
    eps = pos_eps(wl.h, wl.w)
    st = assert_tolerant_parity(cost, orc, gates, og, n_views=wl.V, label=f"{name} smooth", sens=sens, eps=eps)
    assert st["gate_flip_frac"] <= 1e-5 and st["frac_over_2e5"] < 1e-3, st
    got = cost.cpu().numpy().astype(np.float64)
    flipped = (gates.cpu().numpy().astype(bool) != og.astype(bool)).any(axis=1)
    flat = sens * eps <= 1e-5
    over = np.abs(got - orc) > 2e-5 + 2e-5 * np.abs(orc)
    print(f"[{name} smooth] locally flat entries: {flat.mean():.4f}; over the plain bound: {over.mean():.2e} (all on non-flat entries: {not (over & flat & ~flipped).any()})")
    assert over.mean() < 1e-3 and not (over & flat & ~flipped).any()
    ref_sub = golden_r3[f"G14_{name}_cost_sub"]
    sl = (slice(None), slice(None, None, 3), slice(None, None, 5), slice(None, None, 7))
    bad = np.abs(got[sl] - ref_sub) > 2e-5 + 2e-5 * np.abs(ref_sub)
    assert not (bad & flat[sl] & ~flipped[sl]).any() and bad.mean() < 1e-3


@pytest.mark.parametrize("name,B,iters", [("C4", 1, 2), ("C5", 1, 2)])
def test_production_loop_vs_oracle_loop(hip_lib, gpu, name, B, iters):
    """The refinement loop with the PRODUCTION matcher and the matrix-core convolutions against an oracle loop (oracle matcher +
    torch-CPU G-Net + oracle tail / upsampling) at the C4 (KITTI, D = 128, fp32) and C5 (7-Scenes, V = 6) shapes: abs_rel
    < 1e-4 per iteration (the C3 shape is in test_gpu_parity.py)."""
    from magnet_amd.magnet import MAGNET
    wl = synth.WORKLOADS[name]
    inp = synth.make_inputs(wl, B=B, seed=11)
    args = make_args(D=wl.D, iters=iters, dpv_h=wl.h, dpv_w=wl.w, V=wl.V)
    m = MAGNET(args, d_net=StubDNet(1), f_net=StubFNet(2), feat_dtype="fp32", conv_backend="mfma")
    seeded_magnet_weights(m, seed=5, gain=0.3)
    x_d3 = torch.randn(B, 256, wl.h, wl.w, generator=torch.Generator().manual_seed(7)) * 0.5
    k = oracle.depth_sampling(3, wl.D)
    gmm = inp["ref_gmms"].clone(); cpu_preds = []
    with torch.no_grad():
        mask = m.mask_head(x_d3)
        for _ in range(iters):
            cost = torch.from_numpy(oracle_cost(dict(inp, ref_gmms=gmm), k))
            raw = m.g_net.gnet(torch.cat([cost, x_d3], dim=1))
            gmm = torch.from_numpy(oracle.gaussian_update(raw.numpy(), gmm.numpy()))
            cpu_preds.append(oracle.upsample_depth_via_mask(gmm.numpy(), mask.numpy(), 4))
    m = m.to(gpu).eval()
    d = to_dev(inp, gpu)
    with torch.no_grad():
        preds = m.match_and_refine(d["ref_gmms"], x_d3.to(gpu), d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"],
                                   d["nghbr_poses"], inp["is_valid"], inp["cam_intrins"], mode="test")
    assert len(preds) == iters
    for i, (p, c) in enumerate(zip(preds, cpu_preds)):
        got = p.cpu().numpy()
        ar = oracle.abs_rel(np.abs(c[:, 0]) + 1e-3, np.abs(got[:, 0]) + 1e-3)
        print(f"[{name} production loop vs oracle loop, iter {i}] abs_rel delta = {ar:.3e}")
        assert np.isfinite(got).all() and ar < 1e-4
