"""Depth metrics and logging with the reference's names and formats (utils/utils.py:106-197), with the per-pixel
reductions on the device (magnet_depth_metrics) so the (B,2,H,W) predictions are never copied to the host —
the reference's validate() does `.cpu().numpy()` on full maps (test_MaGNet.py:54-56)."""
from __future__ import annotations

import ctypes
import math

import torch

from . import lib

METRIC_ORDER = ("abs_rel", "abs_diff", "sq_rel", "rmse", "rmse_log", "irmse", "log_10", "silog", "a1", "a2", "a3", "nll")


def crop_window(kind, H: int, W: int):
    """Evaluation window (y0, y1, x0, x1) of the reference's KITTI crops (test_MaGNet.py:63-71): 'garg' (Garg ECCV16),
    'eigen' (Eigen NIPS14) or None (whole frame)."""
    if kind in (None, "", "none"):
        return None
    if kind == "garg":
        return int(0.40810811 * H), int(0.99189189 * H), int(0.03594771 * W), int(0.96405229 * W)
    if kind == "eigen":
        return int(0.3324324 * H), int(0.91351351 * H), int(0.0359477 * W), int(0.96405229 * W)
    raise ValueError(f"unknown crop {kind!r}")


def depth_metric_sums(pred: torch.Tensor, gt: torch.Tensor, min_depth: float, max_depth: float, crop=None) -> torch.Tensor:
    """pred (B,2,H,W) fp32 [mu, sigma]; gt (B,1,H,W) or (B,H,W) fp32 -> (B,16) float64 sums (device).
    crop: None, 'garg', 'eigen' or an explicit (y0, y1, x0, x1) window.  Fixed summation order: deterministic."""
    l = lib.load()
    if not getattr(l, "_metrics_proto", False):
        l.magnet_depth_metrics.restype = ctypes.c_int
        l.magnet_depth_metrics.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32] * 2 + [ctypes.c_float] * 2 + [ctypes.c_void_p]
        l.magnet_depth_metrics_crop.restype = ctypes.c_int
        l.magnet_depth_metrics_crop.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32] * 3 + [ctypes.c_float] * 2 + [ctypes.c_int32] * 4 + [ctypes.c_void_p]
        l._metrics_proto = True
    p = lib._dev(pred.detach().float().contiguous(), "pred", torch.float32)
    g = lib._dev(gt.detach().float().contiguous(), "gt", torch.float32)
    B, two, H, W = p.shape
    if two != 2 or g.numel() != B * H * W:
        raise lib.MagnetError(f"depth_metric_sums: pred {tuple(p.shape)} / gt {tuple(g.shape)} mismatch")
    sums = torch.empty((B, 16), dtype=torch.float64, device=p.device)
    win = crop_window(crop, H, W) if isinstance(crop, (str, type(None))) else tuple(int(c) for c in crop)
    with torch.cuda.device(p.device):
        if win is None:
            lib._check(l.magnet_depth_metrics(p.data_ptr(), g.data_ptr(), sums.data_ptr(), B, H * W, float(min_depth),
                                              float(max_depth), lib._stream(p)), "magnet_depth_metrics")
        else:
            lib._check(l.magnet_depth_metrics_crop(p.data_ptr(), g.data_ptr(), sums.data_ptr(), B, H, W, float(min_depth),
                                                   float(max_depth), *win, lib._stream(p)), "magnet_depth_metrics_crop")
    return sums


def metrics_from_sums(s) -> dict:
    """One frame's 16 sums -> the reference's metric dict (same keys as utils.compute_depth_errors)."""
    s = [float(x) for x in s]
    n = s[0]
    if n <= 0:
        return {k: float("nan") for k in METRIC_ORDER}
    mean_err = s[6] / n
    return dict(a1=s[9] / n, a2=s[10] / n, a3=s[11] / n, abs_diff=s[1] / n, abs_rel=s[2] / n, sq_rel=s[3] / n,
                rmse=math.sqrt(s[4] / n), log_10=s[7] / n, irmse=math.sqrt(s[8] / n), rmse_log=math.sqrt(s[5] / n),
                silog=math.sqrt(max(s[5] / n - mean_err * mean_err, 0.0)) * 100, nll=s[12] / n)


def compute_depth_errors(pred, gt, min_depth, max_depth, crop=None) -> list:
    """Per-frame metric dicts for a batch (device reductions, one small D2H of B x 16 doubles); crop: see depth_metric_sums."""
    return [metrics_from_sums(row) for row in depth_metric_sums(pred, gt, min_depth, max_depth, crop).cpu().tolist()]


class RunningAverage:
    def __init__(self):
        self.avg = 0
        self.count = 0

    def append(self, value):
        self.avg = (value + self.count * self.avg) / (self.count + 1)
        self.count += 1

    def get_value(self):
        return self.avg


class RunningAverageDict:
    """utils.RunningAverageDict (utils/utils.py:160-174)."""

    def __init__(self):
        self._dict = None

    def update(self, new_dict):
        if self._dict is None:
            self._dict = {key: RunningAverage() for key in new_dict}
        for key, value in new_dict.items():
            self._dict[key].append(value)

    def get_value(self):
        return {key: value.get_value() for key, value in self._dict.items()}


def format_metrics(metrics) -> str:
    return "%.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f" % tuple(metrics[k] for k in METRIC_ORDER)


def log_metrics(txt_path, metrics, first_line):
    """utils.log_metrics (utils/utils.py:177-197): same header, same 12-column line, appended to txt_path."""
    header = "abs_rel abs_diff sq_rel rmse rmse_log irmse log_10 silog a1 a2 a3 NLL"
    print("{}".format(first_line)); print(header); print(format_metrics(metrics))
    if txt_path:
        with open(txt_path, "a") as f:
            f.write("{}\n".format(first_line)); f.write(header + "\n"); f.write(format_metrics(metrics) + "\n\n")
