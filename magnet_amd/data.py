"""Dataset-free host side of the path (row N4 of SURVEY.md §8f): camera-intrinsics producer and a loader for
ScanNet-format scene folders, yielding exactly what the reference's loaders hand to `validate()` /
`MAGNET.forward` (reference: data/dataloader_scannet.py:16-217, utils/utils.py:64-98).

    <root>/<scene>/color/<i>.jpg   depth/<i>.png (uint16 millimetres)   pose/<i>.txt (4x4 cam->world)
                   intrinsic/intrinsic_color.txt (4x4)

No torchvision, no dataset split files, no JSON side tables: the raw image size is read from the first colour image
unless given.  Everything here is host code feeding the device path; nothing is timed by bench.py.
"""
from __future__ import annotations

import os

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def read_matrix_txt(path: str) -> np.ndarray:
    """4x4 float64 matrix from a whitespace-separated text file (first four rows)."""
    m = np.eye(4)
    with open(path, "r") as f:
        rows = [ln.split() for ln in f.read().strip().splitlines()[:4]]
    for i, r in enumerate(rows):
        m[i, :] = [float(x) for x in r[:4]]
    return m


def read_pose_txt(path: str) -> np.ndarray:
    """ScanNet stores cam->world; the model wants world->cam (dataloader_scannet.py:16-28).  A lost pose (-inf entries)
    inverts to NaN, which `preprocess.data_preprocess` turns into is_valid = 0, like the reference (utils.py:84-94)."""
    with np.errstate(all="ignore"):
        try:
            return np.linalg.inv(read_matrix_txt(path))
        except np.linalg.LinAlgError:
            return np.full((4, 4), np.nan)


def cam_intrinsics(K: np.ndarray, raw_w: int, raw_h: int, dpv_h: int, dpv_w: int, crop=(0, 0), with_table: bool = True) -> dict:
    """K (>=3x3, pixels of the image the network sees BEFORE any crop) -> {'intM' (3,3), 'unit_ray_array_2D' (3, dpv_h*dpv_w),
    'ray_params' (8,) float64}: intrinsics scaled to the matching grid and the ray ((x+0.5)*sx - cx + left)/fx,
    ((y+0.5)*sy - cy + top)/fy, 1 of every grid pixel, row-major (float64 math, one cast).
      ScanNet / 7-Scenes (dataloader_scannet.py:113-153, dataloader_7scenes.py:72-116): raw_w x raw_h = the raw image, no crop.
      KITTI (dataloader_kitti.py:83-127): raw_w x raw_h = the CROPPED network input (1216 x 352), crop = (left, top) margins of the
      KB crop inside the raw frame — see cam_intrinsics_kitti.
    'ray_params' = (fx, fy, cx, cy, sx, sy, left, top): what the kernel needs to generate the rays itself (the 12*h*w-byte
    table then never exists: pass with_table=False)."""
    K = np.asarray(K, dtype=np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    left, top = float(crop[0]), float(crop[1])
    intM = np.zeros((3, 3))
    intM[2, 2] = 1.0
    intM[0, 0], intM[1, 1] = fx * (dpv_w / raw_w), fy * (dpv_h / raw_h)
    intM[0, 2], intM[1, 2] = (cx - left) * (dpv_w / raw_w), (cy - top) * (dpv_h / raw_h)
    sx, sy = raw_w / dpv_w, raw_h / dpv_h
    out = {"intM": torch.from_numpy(intM.astype(np.float32)),
           "ray_params": torch.tensor([fx, fy, cx, cy, sx, sy, left, top], dtype=torch.float64)}
    if with_table:
        xs = np.arange(dpv_w, dtype=np.float64)[None, :] + 0.5
        ys = np.arange(dpv_h, dtype=np.float64)[:, None] + 0.5
        rays = np.ones((3, dpv_h, dpv_w))
        rays[0] = (xs * sx - cx + left) / fx
        rays[1] = (ys * sy - cy + top) / fy
        out["unit_ray_array_2D"] = torch.from_numpy(rays.reshape(3, -1).astype(np.float32))
    return out


def cam_intrinsics_kitti(K_cam2: np.ndarray, raw_w: int, raw_h: int, dpv_h: int, dpv_w: int, img_h: int = 352, img_w: int = 1216,
                         with_table: bool = True) -> dict:
    """KITTI (dataloader_kitti.py:94-127): the network input is the 1216 x 352 KB crop of the raw_w x raw_h frame
    (top margin raw_h - 352, left margin (raw_w - 1216) / 2, both truncated to int); K_cam2 is the raw calibration."""
    top = int(raw_h - img_h)
    left = int((raw_w - img_w) / 2)
    return cam_intrinsics(K_cam2, img_w, img_h, dpv_h, dpv_w, crop=(left, top), with_table=with_table)


def cam_intrinsics_7scenes(dpv_h: int, dpv_w: int, img_h: int = 480, img_w: int = 640, with_table: bool = True) -> dict:
    """7-Scenes (dataloader_7scenes.py:83-116): fixed fx = fy = 585, principal point (320, 240) of the 640 x 480 frames."""
    K = np.array([[585.0, 0.0, 320.0], [0.0, 585.0, 240.0], [0.0, 0.0, 1.0]])
    return cam_intrinsics(K, img_w, img_h, dpv_h, dpv_w, with_table=with_table)


def window_indices(center: int, n_views: int, window_radius: int, exists) -> list:
    """Frame indices of the local window, reference frame in the middle (dataloader_scannet.py:81-89,160-165): offsets
    k * (radius // (n_views // 2)), k = -n/2..n/2; a missing neighbour is mirrored to the other side, half a step closer."""
    step = window_radius // (n_views // 2)
    out = []
    for k in range(-(n_views // 2), n_views // 2 + 1):
        off = k * step
        if exists(center + off):
            out.append(center + off)
        else:
            out.append(center - off - int(np.sign(off)) * int(step * 0.5))
    return out


class ScanNetFolder:
    """samples: list of (scene_name, frame_index).  __getitem__ -> (data_array, cam_intrins): a list of n_views + 1 dicts
    {'img' (3,H,W) normalised fp32, 'gt_dmap' (1,H,W) metres for the reference frame else 0.0, 'extM' (4,4) float64,
    'scene_name', 'img_idx'} and the intrinsics dict — the structure of dataloader_scannet.py:155-217."""

    def __init__(self, root, samples, n_views=4, window_radius=20, input_hw=(480, 640), dpv_hw=(120, 160), raw_wh=None):
        from PIL import Image  # noqa: F401  (fail early if Pillow is missing)
        self.root, self.samples = root, list(samples)
        self.n_views, self.window_radius = n_views, window_radius
        self.img_h, self.img_w = input_hw
        self.dpv_h, self.dpv_w = dpv_hw
        self.raw_wh = dict(raw_wh or {})
        self.center = n_views // 2

    def __len__(self):
        return len(self.samples)

    def _scene(self, name):
        return os.path.join(self.root, name)

    # file naming of the format; SevenScenesFolder overrides these
    def _color(self, sdir, k):
        return os.path.join(sdir, "color", f"{k}.jpg")

    def _depth(self, sdir, k):
        return os.path.join(sdir, "depth", f"{k}.png")

    def _pose(self, sdir, k):
        return os.path.join(sdir, "pose", f"{k}.txt")

    def _intrinsics(self, scene, sdir, idx):
        raw_w, raw_h = self._raw_wh(scene, idx)
        return cam_intrinsics(read_matrix_txt(os.path.join(sdir, "intrinsic", "intrinsic_color.txt")), raw_w, raw_h, self.dpv_h, self.dpv_w)

    def _split(self, sample):
        scene, idx = sample
        return scene, self._scene(scene), int(idx)

    def _depth_metres(self, raw):
        return raw.astype(np.float32) / 1000.0                                  # uint16 millimetres

    def _raw_wh(self, scene, any_idx):
        if scene not in self.raw_wh:
            from PIL import Image
            with Image.open(self._color(self._scene(scene), any_idx)) as im:
                self.raw_wh[scene] = im.size                                   # (W, H)
        return self.raw_wh[scene]

    def __getitem__(self, i):
        from PIL import Image
        scene, sdir, idx = self._split(self.samples[i])
        frames = window_indices(idx, self.n_views, self.window_radius, lambda k: os.path.exists(self._color(sdir, k)))
        intr = self._intrinsics(scene, sdir, idx)
        mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1); std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
        data_array = []
        for j, k in enumerate(frames):
            with Image.open(self._color(sdir, k)) as im:
                rgb = im.convert("RGB").resize((self.img_w, self.img_h), resample=Image.BILINEAR)
            img = torch.from_numpy(np.asarray(rgb).astype(np.float32) / 255.0).permute(2, 0, 1)
            img = (img - mean) / std
            if j == self.center:
                with Image.open(self._depth(sdir, k)) as dm:
                    d = self._depth_metres(np.asarray(dm.resize((self.img_w, self.img_h), resample=Image.NEAREST)))
                gt = torch.from_numpy(d)[None]
            else:
                gt = 0.0
            data_array.append({"img": img, "gt_dmap": gt, "extM": read_pose_txt(self._pose(sdir, k)),
                               "scene_name": scene, "img_idx": str(k)})
        return data_array, intr


class SevenScenesFolder(ScanNetFolder):
    """7-Scenes layout (data/dataloader_7scenes.py): <root>/<scene>/seq-XX/frame-XXXXXX.{color.png,depth.png,pose.txt}; samples are
    (scene, sequence id, frame index); the published intrinsics fx = fy = 585, c = (320, 240) apply to the 640x480 frames and are
    scaled by the INPUT size, as the reference does (:78-100)."""

    def _split(self, sample):
        scene, seq, idx = sample
        return "%s_seq-%02d" % (scene, int(seq)), os.path.join(self.root, scene, "seq-%02d" % int(seq)), int(idx)

    def _depth_metres(self, raw):
        raw = raw.copy()
        raw[raw == 65535] = 0                                                    # the sensor's "no measurement" code (:150)
        return raw.astype(np.float32) / 1000.0

    def _color(self, sdir, k):
        return os.path.join(sdir, "frame-%06d.color.png" % k)

    def _depth(self, sdir, k):
        return os.path.join(sdir, "frame-%06d.depth.png" % k)

    def _pose(self, sdir, k):
        return os.path.join(sdir, "frame-%06d.pose.txt" % k)

    def _intrinsics(self, scene, sdir, idx):
        K = np.eye(3); K[0, 0] = K[1, 1] = 585.0; K[0, 2], K[1, 2] = 320.0, 240.0
        return cam_intrinsics(K, self.img_w, self.img_h, self.dpv_h, self.dpv_w)


def collate(items):
    """What torch's default_collate makes of a list of (data_array, cam_intrins): per-frame dicts with a leading batch
    dimension ('extM' becomes a (B,4,4) float64 tensor), intrinsics stacked to (B,3,3) / (B,3,hw)."""
    n_frames = len(items[0][0])
    data_array = []
    for f in range(n_frames):
        ds = [it[0][f] for it in items]
        gt = ds[0]["gt_dmap"]
        data_array.append({
            "img": torch.stack([d["img"] for d in ds]),
            "gt_dmap": torch.stack([d["gt_dmap"] for d in ds]) if torch.is_tensor(gt) else torch.zeros(len(ds), dtype=torch.float64),
            "extM": torch.from_numpy(np.stack([d["extM"] for d in ds])),
            "scene_name": [d["scene_name"] for d in ds], "img_idx": [d["img_idx"] for d in ds]})
    intr = {k: torch.stack([it[1][k] for it in items]) for k in items[0][1]}
    return data_array, intr


def batches(dataset, batch_size=1):
    """Sequential batches of (data_array, cam_intrins), the iteration protocol of the reference's test loaders."""
    for s in range(0, len(dataset), batch_size):
        yield collate([dataset[i] for i in range(s, min(s + batch_size, len(dataset)))])
