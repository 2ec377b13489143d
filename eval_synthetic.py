#!/usr/bin/env python3
"""test_MaGNet.py-shaped driver on synthetic frames (SURVEY.md §8b "Python-side counterpart"; BASELINE config 1).

Same flow as the reference's validate() (test_MaGNet.py:27-81): loader -> data_preprocess -> model(ref_img,
nghbr_imgs, nghbr_poses, is_valid, cam_intrins, mode='test') -> clamp + mask -> depth metrics -> running average ->
log_metrics line.  Differences: frames come from a seeded synthetic window generator instead of the dataset loaders,
the backbones are caller-provided (stub D-Net/F-Net here), and the metric reductions run on the device.

    python eval_synthetic.py --frames 8 --batch 2 [--D 5] [--iters 3] [--V 4] [--log out.txt]
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from magnet_amd import metrics as M  # noqa: E402
from magnet_amd import synth  # noqa: E402
from magnet_amd.magnet import MAGNET  # noqa: E402
from magnet_amd.preprocess import data_preprocess, data_preprocess_device  # noqa: E402,F401


class SyntheticWindows:
    """Yields (data_array, cam_intrins) like ScannetLoader(...).data (data/dataloader_scannet.py:155-217): a list of
    V+1 dicts with 'img' (B,3,H,W), 'gt_dmap' (B,1,H,W), 'extM' (B,4,4 float64), reference frame in the middle."""

    def __init__(self, n_batches, batch, V, H, W, camera="scannet", seed=0, nan_every=0):
        self.n, self.B, self.V, self.H, self.W, self.cam, self.seed, self.nan_every = n_batches, batch, V, H, W, camera, seed, nan_every

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for it in range(self.n):
            rel = synth.make_poses(self.cam, self.B, self.V, g).double()           # ref -> neighbour
            ref_ext = torch.eye(4, dtype=torch.float64).expand(self.B, 4, 4).clone()
            ref_ext[:, :3, 3] = torch.randn(self.B, 3, generator=g, dtype=torch.float64)
            frames = []
            for i in range(self.V + 1):
                ext = ref_ext if i == self.V // 2 else rel[:, i - (i > self.V // 2)] @ ref_ext
                ext = ext.clone()
                if self.nan_every and i == 0 and it % self.nan_every == 0:
                    ext[0] = float("nan")                                           # a lost pose: is_valid must drop the view
                frames.append({"img": torch.rand(self.B, 3, self.H, self.W, generator=g),
                               "gt_dmap": torch.rand(self.B, 1, self.H, self.W, generator=g) * 3 + 1,
                               "extM": ext})
            yield frames, synth.make_intrinsics(self.cam, self.H // 4, self.W // 4, self.B)


def validate(model, args, test_loader, device):
    """The reference's validate() (test_MaGNet.py:27-81), metrics reduced on the device."""
    with torch.no_grad():
        metrics = M.RunningAverageDict()
        for data_array, cam_intrins in test_loader:
            cur_batch_size = data_array[0]["img"].size()[0]
            # relative poses / validity on the device (float64 inverse there; they never come back to the host)
            ref_dat, nghbr_dats, nghbr_poses, is_valid = data_preprocess_device(data_array, cur_batch_size, device)
            ref_img = ref_dat["img"].to(device)
            gt_dmap = ref_dat["gt_dmap"].to(device)
            nghbr_imgs = torch.cat([d["img"].to(device) for d in nghbr_dats], dim=0)       # view-major
            pred_list = model(ref_img, nghbr_imgs, nghbr_poses, is_valid, cam_intrins, mode="test")
            crop = "garg" if getattr(args, "garg_crop", False) else ("eigen" if getattr(args, "eigen_crop", False) else None)
            for m in M.compute_depth_errors(pred_list[-1], gt_dmap, args.min_depth, args.max_depth, crop=crop):   # test_MaGNet.py:58-79
                metrics.update(m)
        return metrics.get_value()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8); ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--V", type=int, default=4); ap.add_argument("--D", type=int, default=5); ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--input_height", type=int, default=480); ap.add_argument("--input_width", type=int, default=640)
    ap.add_argument("--min_depth", type=float, default=1e-3); ap.add_argument("--max_depth", type=float, default=10.0)
    ap.add_argument("--feat_dtype", default="fp32"); ap.add_argument("--log", default="")
    ap.add_argument("--dataset_path", default="", help="root of ScanNet-format scene folders (magnet_amd/data.py) instead of synthetic frames")
    ap.add_argument("--split", default="", help="text file of '<scene> <frame index>' lines (data_split/scannet_*.txt format)")
    ap.add_argument("--window_radius", type=int, default=20)
    ap.add_argument("--dataset_format", default="scannet", choices=["scannet", "7scenes"],
                    help="folder layout; the split file has '<scene> <frame>' or '<scene> <sequence> <frame>' lines")
    ap.add_argument("--garg_crop", action="store_true", help="KITTI: evaluate inside the Garg ECCV16 window (test_MaGNet.py:67-68)")
    ap.add_argument("--eigen_crop", action="store_true", help="KITTI: evaluate inside the Eigen NIPS14 window (test_MaGNet.py:69-70)")
    ap.add_argument("--psmnet", action="store_true", help="use the PSMNet F-Net (matrix-core path) instead of the stub F-Net")
    a = ap.parse_args()
    from magnet_amd.standin import StubDNet, StubFNet, make_args, seeded_magnet_weights
    if not torch.cuda.is_available():
        raise SystemExit("eval_synthetic.py needs an MI355X (no CPU fallback)")
    device = torch.device("cuda:0")
    args = make_args(D=a.D, iters=a.iters, dpv_h=a.input_height // 4, dpv_w=a.input_width // 4, V=a.V)
    args.min_depth, args.max_depth = a.min_depth, a.max_depth
    args.garg_crop, args.eigen_crop = a.garg_crop, a.eigen_crop
    f_net = StubFNet(2)
    if a.psmnet:
        from magnet_amd.fnet import FNET
        args.FNET_architecture, args.FNET_feature_dim = "PSM-Net", 64
        f_net = FNET(args)                                   # random init unless a checkpoint is loaded by the caller
    model = MAGNET(args, d_net=StubDNet(1), f_net=f_net, feat_dtype=a.feat_dtype)
    seeded_magnet_weights(model, 3)
    model = model.to(device).eval()
    if a.dataset_path:
        from magnet_amd import data
        with open(a.split) as f:
            samples = [ln.split()[:3 if a.dataset_format == "7scenes" else 2] for ln in f if ln.strip()]
        Folder = data.SevenScenesFolder if a.dataset_format == "7scenes" else data.ScanNetFolder
        ds = Folder(a.dataset_path, samples, n_views=a.V, window_radius=a.window_radius,
                                input_hw=(a.input_height, a.input_width), dpv_hw=(a.input_height // 4, a.input_width // 4))
        loader = data.batches(ds, a.batch)
        title = "scannet-format folder %s (%d windows) V=%d D=%d iters=%d" % (a.dataset_path, len(ds), a.V, a.D, a.iters)
    else:
        loader = SyntheticWindows((a.frames + a.batch - 1) // a.batch, a.batch, a.V, a.input_height, a.input_width, nan_every=3)
        title = "synthetic frames=%d V=%d D=%d iters=%d" % (a.frames, a.V, a.D, a.iters)
    m = validate(model, args, loader, device)
    M.log_metrics(a.log, m, title)


if __name__ == "__main__":
    main()
