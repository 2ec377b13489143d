"""-m gpu: row N4 on the device — unit rays generated from (fx, fy, cx, cy, sx, sy, left, top) instead of the loaders' table,
relative poses / is_valid computed from float64 extrinsics on the GPU — and the C5 end-to-end leg (matrix-core F-Net ->
production matcher at 480x640, V = 6, D = 64)."""
import numpy as np
import pytest
import torch

from magnet_amd import data, fnet, lib, synth
from oracle import oracle
from tests.parity import to_dev
from tests.stubs import StubDNet, make_args, procedural_images, seeded_fnet_state, seeded_magnet_weights

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


def test_C5_end_to_end_fnet_to_matcher(hip_lib, gpu):
    """BASELINE config 5 leg: 480x640 images, V = 6 source views, D = 64, 7-Scenes intrinsics (rays generated in the kernel),
    PSMNet F-Net on the matrix cores writing the matcher's layouts, production matcher, matrix-core G-Net / mask head —
    against the same forward with the torch F-Net + pack path and the exact matcher.  (The D-Net is a seeded stand-in: it
    needs torch.hub)"""
    from magnet_amd.magnet import MAGNET
    V, D = 6, 64
    args = make_args(D=D, iters=1, dpv_h=120, dpv_w=160, fdim=64, V=V)
    args.FNET_architecture, args.FNET_feature_dim = "PSM-Net", 64
    fn = fnet.FNET(args); fn.f_net = seeded_fnet_state(fnet.FNET(args).f_net, seed=5)
    model = MAGNET(args, d_net=StubDNet(0), f_net=fn, feat_dtype="fp32").to(gpu).eval()
    seeded_magnet_weights(model, seed=4)
    gen = torch.Generator().manual_seed(3)
    poses = synth.make_poses("7scenes", 1, V, gen).to(gpu)
    valid = torch.ones(1, V, dtype=torch.int32)
    ci = data.cam_intrinsics_7scenes(120, 160, with_table=False)
    lean = {"intM": ci["intM"][None], "ray_params": ci["ray_params"][None]}
    full = {kk: vv[None] for kk, vv in data.cam_intrinsics_7scenes(120, 160).items() if kk != "ray_params"}
    ref_img = procedural_images(1, 480, 640).to(gpu); nb = procedural_images(V, 480, 640).flip(0).to(gpu)
    outs = {}
    for name, mfma_fnet, path, cam in (("production", True, 0, lean), ("reference-ish", False, 2, full)):
        model.fnet_mfma, model.matcher_path = mfma_fnet, path
        with torch.no_grad():
            outs[name] = model(ref_img, nb, poses, valid, cam, mode="test")
    a, b = outs["production"][-1][:, 0].cpu(), outs["reference-ish"][-1][:, 0].cpu()
    assert tuple(a.shape) == (1, 480, 640) and torch.isfinite(a).all()
    rel = ((a - b).abs() / b.abs().clamp_min(1e-3)).mean().item()
    print(f"C5 end to end (F-Net MFMA + production matcher + in-kernel rays) vs (torch F-Net + exact matcher + table): abs_rel = {rel:.2e}")
    assert rel < 1e-4
