"""utils.data_preprocess / split_data_array (utils/utils.py:64-98): reference frame = the middle of the window,
relative pose = ext_nghbr @ inv(ext_ref) in float64, NaN anywhere -> is_valid = 0 and a zero pose; poses are float32
(B,V,4,4), is_valid int32 (B,V) on the CPU (the matcher uploads and caches it)."""
from __future__ import annotations

import torch


def split_data_array(data_array):
    n_frames = len(data_array)
    ref_idx = n_frames // 2
    return data_array[ref_idx], [data_array[idx] for idx in range(n_frames) if idx != ref_idx]


def data_preprocess_device(data_array, cur_batch_size, device):
    """data_preprocess with the float64 inverse / product / NaN test on the GPU (lib.relative_poses): the extrinsics go up as
    (V+1) x B x 128 bytes, poses and is_valid never come back to the host — they are what the matcher reads.
    Returns (ref_dat, nghbr_dats, nghbr_poses (B,V,4,4) fp32 cuda, is_valid (B,V) int32 cuda)."""
    from . import lib
    ref_dat, nghbr_dats = split_data_array(data_array)
    B = cur_batch_size
    ref = torch.as_tensor(ref_dat["extM"]).double()[:B].to(device)
    ngh = torch.stack([torch.as_tensor(d["extM"]).double()[:B] for d in nghbr_dats], 1).to(device)
    poses, valid = lib.relative_poses(ref.contiguous(), ngh.contiguous())
    return ref_dat, nghbr_dats, poses, valid


def data_preprocess(data_array, cur_batch_size):
    ref_dat, nghbr_dats = split_data_array(data_array)
    num_views = len(nghbr_dats)
    ref = torch.as_tensor(ref_dat["extM"]).double()                                   # (B,4,4)
    ngh = torch.stack([torch.as_tensor(d["extM"]).double() for d in nghbr_dats], 1)   # (B,V,4,4)
    B = cur_batch_size
    ref_bad = torch.isnan(ref[:B]).flatten(1).any(1)                                  # (B,)
    ngh_bad = torch.isnan(ngh[:B]).flatten(2).any(2)                                  # (B,V)
    safe_ref = torch.where(ref_bad[:, None, None], torch.eye(4, dtype=torch.float64, device=ref.device).expand(B, 4, 4), ref[:B])
    inv = torch.linalg.inv(safe_ref)                                                   # float64, as np.linalg.inv
    pose = torch.nan_to_num(ngh[:B], nan=0.0) @ inv[:, None]
    bad = ref_bad[:, None] | ngh_bad | torch.isnan(pose).flatten(2).any(2)
    nghbr_poses = torch.where(bad[:, :, None, None], torch.zeros_like(pose), pose).to(torch.float32).cpu()
    is_valid = (~bad).to(torch.int32).cpu()
    if nghbr_poses.shape != (B, num_views, 4, 4):
        raise ValueError("data_preprocess: inconsistent extM shapes")
    return ref_dat, nghbr_dats, nghbr_poses, is_valid
