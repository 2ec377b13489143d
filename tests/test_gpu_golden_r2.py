"""-m gpu: the HIP path directly against reference-generated fixtures at the BASELINE.json shapes (golden_r2.npz, made by
tests/golden/make_golden_r2.py from the imported reference) — no oracle in between — and one training step."""
import numpy as np
import pytest
import torch

from magnet_amd import synth
from oracle import oracle
from tests.parity import assert_cost_parity, assert_tolerant_parity, oracle_cost, pos_eps, position_sensitivity, to_dev
from tests.stubs import StubDNet, StubFNet, magnet_nll_loss, seeded_magnet_weights, train_case
from tests.test_oracle_golden import BASELINE_GOLDEN, baseline_golden_inputs

pytestmark = pytest.mark.gpu


def _hip_cost(inp, k_list, device, feat_dtype, path):
    from magnet_amd.homography import CostVolumeCW
    d = to_dev(inp, device)
    cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"],
                      d["cam_intrins"], 5, feat_dtype=feat_dtype, path=path)
    return cv(ref_gmm=d["ref_gmms"], k_list=k_list)


@pytest.mark.parametrize("name,seed,bf16", BASELINE_GOLDEN)
def test_cost_volume_baseline_golden_subsample(hip_lib, gpu, golden_r2, name, seed, bf16):
    """C2 (bf16 storage) / C4 / C5: generic kernel bitwise, exact candidate-lane kernel within the re-association
    tolerance with no flipped entry, production matcher within the tolerance-parity contract — all against the REFERENCE's
    own numbers."""
    wl, inp = baseline_golden_inputs(name, seed, bf16)
    k = oracle.depth_sampling(3, wl.D)
    ref_sub = golden_r2[f"G2_{name}_cost_sub"]
    fdt = "bf16" if bf16 else "fp32"
    for path in (1, 2):
        got = _hip_cost(inp, k, gpu, fdt, path).cpu().numpy()
        assert_cost_parity(got[:, ::3, ::5, ::7], ref_sub, path=path, label=f"{name} golden")
        if path == 1:
            s = golden_r2[f"G2_{name}_cost_sum"]
            assert got.astype(np.float64).sum() == s[0] and np.abs(got).astype(np.float64).sum() == s[1]
    got = _hip_cost(inp, k, gpu, fdt, 4).cpu().numpy()
    _, og, _ = oracle_cost(inp, k, aux=True)
    sens = position_sensitivity(inp, k, og, device=gpu)[:, ::3, ::5, ::7]
    assert_tolerant_parity(got[:, ::3, ::5, ::7], ref_sub, n_views=wl.V, label=f"{name} golden", sens=sens, eps=pos_eps(wl.h, wl.w))


def test_training_step_gradients_match_reference(hip_lib, gpu, golden_r2):
    """MAGNET(mode='train') + the reference's loss + backward (train_MaGNet.py:87-98): loss and the gradients of g_net /
    mask_head against the reference's autograd (G11).  The matcher is forward-only (its inputs are detached in the
    reference too); everything downstream of it must be differentiable."""
    from magnet_amd.magnet import MAGNET
    args, ref_img, nghbr_imgs, poses, valid, intr, gt, gt_mask = train_case()
    m = MAGNET(args, d_net=StubDNet(seed=21), f_net=StubFNet(seed=22, fdim=8))
    seeded_magnet_weights(m, seed=23, gain=0.25)
    m = m.to(gpu).train()
    preds = m(ref_img.to(gpu), nghbr_imgs.to(gpu), poses.to(gpu), valid, intr, mode="train")
    assert len(preds) == 3 and all(p.requires_grad for p in preds)
    loss = magnet_nll_loss(preds, gt.to(gpu), gt_mask.to(gpu))
    loss.backward()
    np.testing.assert_allclose(loss.item(), golden_r2["G11_loss"][0], rtol=2e-4)
    for key, prm in (("gnet0", m.g_net.gnet[0].weight), ("mask0", m.mask_head[0].weight),
                     ("gnet6", m.g_net.gnet[6].weight), ("mask6", m.mask_head[6].weight)):
        assert prm.grad is not None, key
        g = prm.grad.detach().cpu().numpy().astype(np.float64).reshape(-1)
        ref_sub = golden_r2[f"G11_grad_{key}_sub"].astype(np.float64)
        scale = np.abs(ref_sub).max()
        assert np.abs(g[::97] - ref_sub).max() <= 2e-3 * scale, (key, np.abs(g[::97] - ref_sub).max(), scale)
        np.testing.assert_allclose(np.abs(g).sum(), golden_r2[f"G11_grad_{key}_sum"][1], rtol=2e-3)
    for i, p_ in enumerate(preds):
        np.testing.assert_allclose(p_.detach().abs().double().sum().item(), golden_r2[f"G11_pred{i}_sum"][1], rtol=1e-4)
