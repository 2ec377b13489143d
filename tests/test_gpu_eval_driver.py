"""-m gpu: the test_MaGNet.py-shaped driver end to end on a generated ScanNet-format folder: loader -> data_preprocess ->
MAGNET (PSMNet F-Net on the matrix cores, stub D-Net) -> device metrics -> log line."""
import os
import subprocess
import sys

import pytest

from tests.test_data_loader import _make_scene

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eval_driver_on_scannet_folder(hip_lib, gpu, tmp_path):
    pytest.importorskip("PIL")
    _make_scene(str(tmp_path), "scene0001_00", 14, raw_wh=(320, 256), lost=(3,))
    split = tmp_path / "split.txt"
    split.write_text("scene0001_00 6\nscene0001_00 7\nscene0001_00 8\n")
    log = tmp_path / "log.txt"
    out = subprocess.run([sys.executable, os.path.join(REPO, "eval_synthetic.py"), "--dataset_path", str(tmp_path), "--split", str(split),
                          "--V", "4", "--D", "8", "--iters", "2", "--batch", "2", "--window_radius", "4", "--input_height", "256",
                          "--input_width", "320", "--psmnet", "--log", str(log)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = log.read_text()
    assert "abs_rel" in text and "scannet-format folder" in text and "nan" not in text.lower()
