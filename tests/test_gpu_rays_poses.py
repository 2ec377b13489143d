"""-m gpu: row N4 on the device — unit rays generated from (fx, fy, cx, cy, sx, sy, left, top) instead of the loaders' table,
relative poses / is_valid computed from float64 extrinsics on the GPU — and the C5 end-to-end leg (matrix-core F-Net ->
production matcher at 480x640, V = 6, D = 64)."""
import numpy as np
import pytest
import torch

from magnet_amd import data, fnet, lib, synth
from oracle import oracle
from tests.parity import to_dev
from tests.stubs import c5_case, StubDNet, make_args, procedural_images, seeded_fnet_state, seeded_magnet_weights

pytestmark = pytest.mark.gpu


def _cases(g):
    return (("scannet", data.cam_intrinsics(g["G12_scannet_K"], 1296, 968, 120, 160), g["G12_scannet_rays"], 120, 160),
            ("7scenes", data.cam_intrinsics_7scenes(120, 160), g["G12_7scenes_rays"], 120, 160),
            ("kitti", data.cam_intrinsics_kitti(g["G12_kitti_K"], 1242, 375, 88, 304), g["G12_kitti_rays"], 88, 304))


def test_make_rays_bitwise_equals_reference_loader_tables(hip_lib, gpu, golden_r2):
    """magnet_make_rays against the tables the REFERENCE's loaders compute (G12), bit for bit."""
    for name, ci, ref_rays, h, w in _cases(golden_r2):
        prm = ci["ray_params"][None].repeat(2, 1).to(gpu)
        got = lib.make_rays(prm, h, w).cpu().numpy()
        assert got.shape == (2, 3, h * w)
        assert np.array_equal(got[0], ref_rays) and np.array_equal(got[1], ref_rays), name


@pytest.mark.parametrize("path", [4, 2, 1, 3])
def test_matcher_with_ray_params_equals_table(hip_lib, gpu, path):
    """cam_intrins carrying only 'ray_params' (no 12*h*w-byte table): every kernel returns exactly what it returns with the
    loader's table (production / candidate-lane / generic generate the rays in the kernel; the worklist kernel gets a table
    built on the device)."""
    from magnet_amd.homography import CostVolumeCW
    wl = synth.Workload("rp", "kitti", 22, 76, V=2, D=8 if path != 4 else 40, F=16)
    inp = synth.make_inputs(wl, B=2, seed=5)
    cam = synth.CAMERAS["kitti"]
    K = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1.0]])
    ci = data.cam_intrinsics_kitti(K, 1242, 375, wl.h, wl.w)
    full = {"intM": ci["intM"][None].repeat(2, 1, 1), "unit_ray_array_2D": ci["unit_ray_array_2D"][None].repeat(2, 1, 1)}
    lean = {"intM": full["intM"], "ray_params": ci["ray_params"][None].repeat(2, 1)}
    d = to_dev(inp, gpu)
    k = oracle.depth_sampling(3, wl.D)
    outs = []
    for cam_intrins in (full, lean):
        cv = CostVolumeCW(d["ref_feat"], d["nghbr_feat"], d["nghbr_gmms"], d["nghbr_poses"], d["is_valid"], cam_intrins, 5, path=path)
        outs.append(cv(ref_gmm=d["ref_gmms"], k_list=k))
    assert torch.equal(outs[0], outs[1]) and torch.count_nonzero(outs[0]) > 0


def test_relative_poses_on_device_match_reference(hip_lib, gpu, golden):
    """G8 (utils.data_preprocess incl. a NaN reference and a NaN neighbour): poses to 1e-6, validity exactly; outputs stay on
    the device."""
    from magnet_amd.preprocess import data_preprocess_device
    exts = golden["G8_exts"]
    data_array = [{"extM": torch.from_numpy(e), "tag": i} for i, e in enumerate(exts)]
    ref, nghbrs, poses, valid = data_preprocess_device(data_array, 3, gpu)
    assert ref["tag"] == 2 and poses.is_cuda and valid.is_cuda and poses.dtype == torch.float32 and valid.dtype == torch.int32
    np.testing.assert_allclose(poses.cpu().numpy(), golden["G8_poses"], rtol=0, atol=1e-6)
    assert np.array_equal(valid.cpu().numpy(), golden["G8_valid"])
    # a singular reference extrinsic -> every view of that frame invalid, zero poses
    er = torch.eye(4, dtype=torch.float64).repeat(2, 1, 1); er[1, 2] = er[1, 1]
    en = torch.eye(4, dtype=torch.float64).repeat(2, 3, 1, 1)
    p2, v2 = lib.relative_poses(er.to(gpu), en.to(gpu))
    assert v2.cpu().tolist() == [[1, 1, 1], [0, 0, 0]] and not p2[1].any() and torch.equal(p2[0].cpu(), en[0].float())


def test_C5_end_to_end_fnet_to_matcher_vs_reference(hip_lib, gpu, golden_r4):
    """BASELINE config 5 with the F-Net in the loop, against the REFERENCE's own numbers (fixture G15: models.FNET.FNET +
    MAGNET.forward on the CPU, tests/golden/make_golden_r4.py): 480x640 images, V = 6 source views, D = 64, 7-Scenes intrinsics.
    Ours: PSMNet F-Net on the matrix cores writing the matcher's layouts, rays generated in the kernel from 8 scalars, production
    matcher, matrix-core G-Net / mask head with the fused update / upsampling.  (The D-Net is a seeded stand-in on both sides: it
    needs torch.hub)  The F-Net runs split-bf16 x 3 arithmetic over 27 layers, so the bar is abs_rel 1e-4
    (north_star), not the 1e-6 of the matcher-only fixtures; the torch-F-Net + exact-matcher forward must meet 1e-5."""
    from magnet_amd.magnet import MAGNET
    args, ref_img, nb, poses, valid, cam_full, seeds = c5_case()
    V = 6
    fn = fnet.FNET(args); fn.f_net = seeded_fnet_state(fnet.FNET(args).f_net, seed=seeds["f"])
    model = MAGNET(args, d_net=StubDNet(seeds["d"]), f_net=fn, feat_dtype="fp32").to(gpu).eval()
    seeded_magnet_weights(model, seed=seeds["w"])
    ci = data.cam_intrinsics_7scenes(120, 160, with_table=False)
    lean = {"intM": ci["intM"][None], "ray_params": ci["ray_params"][None]}
    ref_sub, ref_sum = golden_r4["G15_pred_sub"], golden_r4["G15_pred_sum"]
    for name, mfma_fnet, path, cam, bar in (("production", True, 0, lean, 1e-4), ("torch F-Net + exact matcher", False, 2, cam_full, 1e-5)):
        model.fnet_mfma, model.matcher_path = mfma_fnet, path
        with torch.no_grad():
            out = model(ref_img.to(gpu), nb.to(gpu), poses.to(gpu), valid, cam, mode="test")
        got = out[-1].cpu().numpy()
        assert got.shape == (1, 2, 480, 640) and np.isfinite(got).all()
        sub = got[:, :, ::8, ::8]
        ar = oracle.abs_rel(ref_sub[:, 0], sub[:, 0])
        print(f"C5 end to end, {name} vs the reference's FNET + MAGNET.forward: abs_rel = {ar:.2e}; max rel dsigma = {np.abs(sub[:, 1] / ref_sub[:, 1] - 1).max():.2e}")
        assert ar < bar
        np.testing.assert_allclose(got.astype(np.float64).sum(), ref_sum[0], rtol=1e-3)
