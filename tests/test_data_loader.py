"""CPU: the ScanNet-format folder loader and the cam_intrins producer (row N4; reference data/dataloader_scannet.py)."""
import os

import numpy as np
import pytest
import torch

from magnet_amd import data, synth
from magnet_amd.preprocess import data_preprocess


def _make_scene(root, scene, n_frames, raw_wh=(64, 48), lost=(), missing=()):
    from PIL import Image
    sdir = os.path.join(root, scene)
    for sub in ("color", "depth", "pose", "intrinsic"):
        os.makedirs(os.path.join(sdir, sub))
    K = np.eye(4); K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 58.0, 57.0, 31.5, 23.5
    np.savetxt(os.path.join(sdir, "intrinsic", "intrinsic_color.txt"), K)
    rng = np.random.RandomState(0)
    poses = {}
    for i in range(n_frames):
        if i in missing:
            continue
        Image.fromarray(rng.randint(0, 255, (raw_wh[1], raw_wh[0], 3), dtype=np.uint8)).save(os.path.join(sdir, "color", f"{i}.jpg"))
        Image.fromarray((rng.rand(raw_wh[1], raw_wh[0]) * 4000 + 500).astype(np.uint16)).save(os.path.join(sdir, "depth", f"{i}.png"))
        T = np.eye(4); T[:3, 3] = [0.01 * i, 0.02 * i, -0.005 * i]
        c, s = np.cos(0.01 * i), np.sin(0.01 * i); T[0, 0], T[0, 2], T[2, 0], T[2, 2] = c, s, -s, c
        poses[i] = T
        with open(os.path.join(sdir, "pose", f"{i}.txt"), "w") as f:
            for row in (np.full((4, 4), -np.inf) if i in lost else T):
                f.write(" ".join("-inf" if not np.isfinite(v) else repr(float(v)) for v in row) + "\n")
    return K, poses


def test_cam_intrinsics_matches_reference_formula():
    c = synth.CAMERAS["scannet"]
    K = np.eye(4); K[0, 0], K[1, 1], K[0, 2], K[1, 2] = c["fx"], c["fy"], c["cx"], c["cy"]
    got = data.cam_intrinsics(K, c["raw_w"], c["raw_h"], 120, 160)
    exp = synth.make_intrinsics("scannet", 120, 160, 1)
    assert torch.equal(got["intM"], exp["intM"][0]) and torch.equal(got["unit_ray_array_2D"], exp["unit_ray_array_2D"][0])
    assert got["unit_ray_array_2D"].shape == (3, 120 * 160) and torch.all(got["unit_ray_array_2D"][2] == 1)


def test_window_indices_and_missing_frames():
    have = set(range(0, 100))
    assert data.window_indices(50, 4, 20, have.__contains__) == [30, 40, 50, 60, 70]
    assert data.window_indices(50, 2, 10, have.__contains__) == [40, 50, 60]
    # near the end of a scan the missing neighbours are mirrored to the other side, half a step closer (:160-165)
    assert data.window_indices(95, 4, 20, have.__contains__) == [75, 85, 95, 95 - 10 - 5, 95 - 20 - 5]


def test_folder_loader_end_to_end(tmp_path):
    pytest.importorskip("PIL")
    K, poses = _make_scene(str(tmp_path), "scene0000_00", 12, lost=(4,))
    ds = data.ScanNetFolder(str(tmp_path), [("scene0000_00", 6), ("scene0000_00", 5)], n_views=4, window_radius=4,
                            input_hw=(32, 48), dpv_hw=(8, 12))
    arr, intr = ds[0]
    assert [d["img_idx"] for d in arr] == ["2", "4", "6", "8", "10"]
    assert arr[2]["gt_dmap"].shape == (1, 32, 48) and arr[0]["gt_dmap"] == 0.0
    assert 0.4 < float(arr[2]["gt_dmap"].min()) and float(arr[2]["gt_dmap"].max()) < 4.6          # millimetres -> metres
    np.testing.assert_allclose(arr[2]["extM"], np.linalg.inv(poses[6]), atol=1e-12)               # cam->world file, world->cam out
    assert np.isnan(arr[1]["extM"]).all()                                                          # lost pose
    img = arr[2]["img"]
    assert img.shape == (3, 32, 48) and img.dtype == torch.float32
    lo = (0 - np.array(data.IMAGENET_MEAN)) / np.array(data.IMAGENET_STD); hi = (1 - np.array(data.IMAGENET_MEAN)) / np.array(data.IMAGENET_STD)
    for ch in range(3):
        assert lo[ch] - 1e-6 <= float(img[ch].min()) and float(img[ch].max()) <= hi[ch] + 1e-6
    exp = data.cam_intrinsics(K, 64, 48, 8, 12)
    assert torch.equal(intr["intM"], exp["intM"])
    # batch of two windows -> the reference's data_preprocess contract
    data_array, cam = next(data.batches(ds, batch_size=2))
    assert data_array[0]["img"].shape == (2, 3, 32, 48) and data_array[2]["extM"].shape == (2, 4, 4) and cam["intM"].shape == (2, 3, 3)
    ref_dat, nghbr_dats, nghbr_poses, is_valid = data_preprocess(data_array, 2)
    assert nghbr_poses.shape == (2, 4, 4, 4) and is_valid.tolist() == [[1, 0, 1, 1], [1, 1, 1, 1]]  # window of frame 6 holds lost frame 4
    rel = poses_rel = np.linalg.inv(poses[8]) @ poses[6]                                           # ext_nghbr @ inv(ext_ref)
    np.testing.assert_allclose(nghbr_poses[0, 2].numpy(), rel, atol=1e-6)
    assert ref_dat["gt_dmap"].shape == (2, 1, 32, 48) and not nghbr_poses[0, 1].any()


def test_seven_scenes_layout(tmp_path):
    pytest.importorskip("PIL")
    from PIL import Image
    sdir = tmp_path / "chess" / "seq-03"
    sdir.mkdir(parents=True)
    rng = np.random.RandomState(1)
    for i in range(0, 40, 5):
        Image.fromarray(rng.randint(0, 255, (48, 64, 3), dtype=np.uint8)).save(sdir / ("frame-%06d.color.png" % i))
        Image.fromarray((rng.rand(48, 64) * 3000 + 400).astype(np.uint16)).save(sdir / ("frame-%06d.depth.png" % i))
        T = np.eye(4); T[:3, 3] = [0.02 * i, 0.0, 0.01 * i]
        np.savetxt(sdir / ("frame-%06d.pose.txt" % i), T)
    ds = data.SevenScenesFolder(str(tmp_path), [("chess", 3, 20)], n_views=2, window_radius=10, input_hw=(48, 64), dpv_hw=(12, 16))
    arr, intr = ds[0]
    assert [d["img_idx"] for d in arr] == ["10", "20", "30"] and arr[1]["gt_dmap"].shape == (1, 48, 64)
    np.testing.assert_allclose(intr["intM"].numpy(), [[585 * 16 / 64, 0, 320 * 16 / 64], [0, 585 * 12 / 48, 240 * 12 / 48], [0, 0, 1]], rtol=1e-6)
    np.testing.assert_allclose(arr[2]["extM"][:3, 3], [-0.6, 0.0, -0.3], atol=1e-12)                # inverse of the cam->world file
    data_array, cam = next(data.batches(ds, 1))
    _, _, poses, valid = data_preprocess(data_array, 1)
    assert valid.tolist() == [[1, 1]] and poses.shape == (1, 2, 4, 4)


def test_cam_intrinsics_match_reference_loaders(golden_r2):
    """G12: intM and the unit-ray table of the ScanNet / 7-Scenes / KITTI loaders (computed by the reference's own methods,
    tests/golden/make_golden_r2.py) — reproduced bit for bit, including KITTI's KB-crop margins."""
    import numpy as np
    from magnet_amd import data
    g = golden_r2
    ci = data.cam_intrinsics(g["G12_scannet_K"], 1296, 968, 120, 160)
    assert np.array_equal(ci["intM"].numpy(), g["G12_scannet_intM"]) and np.array_equal(ci["unit_ray_array_2D"].numpy(), g["G12_scannet_rays"])
    ci = data.cam_intrinsics_7scenes(120, 160)
    assert np.array_equal(ci["intM"].numpy(), g["G12_7scenes_intM"]) and np.array_equal(ci["unit_ray_array_2D"].numpy(), g["G12_7scenes_rays"])
    ci = data.cam_intrinsics_kitti(g["G12_kitti_K"], 1242, 375, 88, 304)
    assert np.array_equal(ci["intM"].numpy(), g["G12_kitti_intM"]) and np.array_equal(ci["unit_ray_array_2D"].numpy(), g["G12_kitti_rays"])
    assert ci["ray_params"].dtype.is_floating_point and tuple(ci["ray_params"].shape) == (8,) and float(ci["ray_params"][6]) == 13.0 and float(ci["ray_params"][7]) == 23.0
    assert "unit_ray_array_2D" not in data.cam_intrinsics_7scenes(120, 160, with_table=False)
