"""Shared parity helpers for the GPU tests: how a HIP cost volume is compared with the oracle.

The consistency gate |z - mu_w| < kappa*sigma_w (reference homography.py:157-158) is a hard
threshold, so a 1-ulp difference in any upstream quantity can flip a gate and change one cost
entry by a full feature dot product.  Parity is therefore judged as:
  (a) the fraction of entries that differ by more than rounding ("flip candidates") is <= flip_frac,
  (b) everywhere else |hip - oracle| <= atol + rtol*|oracle|,
  (c) downstream, the final-depth abs_rel delta stays < 1e-4 (tests that run the full loop).
"""
import numpy as np
import torch

from oracle import oracle


def to_dev(inp, device):
    out = {}
    for k, v in inp.items():
        if k == "cam_intrins":
            out[k] = v                      # stays on the CPU like the reference's loader hands it over
        elif k == "is_valid":
            out[k] = v
        else:
            out[k] = v.to(device)
    return out


def oracle_cost(inp, k_list, kappa=5.0, d_volume=None, aux=False):
    return oracle.cost_volume_cw(d_volume, inp["ref_gmms"], k_list, inp["ref_feat"], inp["nghbr_feat"],
                                 inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"],
                                 inp["cam_intrins"]["intM"], inp["cam_intrins"]["unit_ray_array_2D"],
                                 kappa, return_aux=aux)


def cost_stats(hip, orc, atol=2e-5, rtol=2e-5):
    hip = hip.detach().cpu().numpy() if isinstance(hip, torch.Tensor) else np.asarray(hip)
    diff = np.abs(hip.astype(np.float64) - orc.astype(np.float64))
    bad = diff > (atol + rtol * np.abs(orc))
    good = ~bad
    return dict(frac_bitwise=float(np.mean(hip == orc)), frac_flip=float(bad.mean()),
                max_abs_nonflip=float(diff[good].max()) if good.any() else 0.0,
                max_abs=float(diff.max()), n=int(diff.size), finite=bool(np.isfinite(hip).all()))


# Tolerances by kernel (include/magnet_hip.h `path`):
#   path 1 (generic gather kernel): every operation mirrors the oracle -> BITWISE equality.
#   path 0/4 (PRODUCTION matcher, cost_volume_fast.hip): tolerance parity — its geometry is not the reference's rounding
#   sequence, so a small fraction of consistency gates flips: gate-flip fraction <= 1e-5, every entry outside
#   the value tolerance must be explained by a flipped gate of one of its views, everything else within 2e-5 + 2e-5*|oracle|.
#   path 2 (exact candidate-lane kernel) and 3 (worklist kernel): gates and sample positions are still exactly the oracle's; the 64-channel
#   sum is re-associated (dot products per tap, then the bilinear combine), so values agree to fp32
#   accumulation noise: |d| <= 2e-5 + 2e-5*|oracle| (costs are O(1..10), sums of 64 products of
#   N(0,1)-scale numbers), and NO entry may differ by more than that (gate flips would).
WORKLIST_ATOL = 2e-5      # measured on MI355X: max |d| 3.8e-6 over all test shapes
WORKLIST_RTOL = 2e-5


GATE_FLIP_FRAC = 1e-5     # SURVEY.md §7 / VERDICT r1: allowed fraction of flipped consistency gates (production matcher)


def assert_tolerant_parity(hip, orc, hip_gates=None, orc_gates=None, n_views=4, label=""):
    """Production matcher (path 0/4).  With gate bits from both sides (B,V,D,h,w): the gate-flip fraction is <= 1e-5 (at
    least one flip is tolerated on tiny inputs) and every value outside the tolerance sits on an entry with a flipped gate.
    Without gate bits: the fraction of out-of-tolerance entries is <= n_views * 1e-5."""
    st = cost_stats(hip, orc, WORKLIST_ATOL, WORKLIST_RTOL)
    hipn = hip.detach().cpu().numpy() if isinstance(hip, torch.Tensor) else np.asarray(hip)
    assert st["finite"] or not np.isfinite(orc).all(), f"{label}: non-finite values in the HIP cost volume"
    if hip_gates is not None:
        hg = hip_gates.detach().cpu().numpy() if isinstance(hip_gates, torch.Tensor) else np.asarray(hip_gates)
        flipped = hg.astype(bool) != orc_gates.astype(bool)                       # (B,V,D,h,w)
        n_flip, n_gate = int(flipped.sum()), int(flipped.size)
        st["gate_flips"], st["gates"], st["gate_flip_frac"] = n_flip, n_gate, n_flip / n_gate
        print(f"[parity {label} production] {st}")
        assert n_flip <= max(1, int(GATE_FLIP_FRAC * n_gate)), f"{label}: {st}"
        diff = np.abs(hipn.astype(np.float64) - orc.astype(np.float64))
        bad = diff > (WORKLIST_ATOL + WORKLIST_RTOL * np.abs(orc))
        unexplained = bad & ~flipped.any(axis=1)
        assert not unexplained.any(), f"{label}: {int(unexplained.sum())} entries differ without a flipped gate: {st}"
    else:
        print(f"[parity {label} production] {st}")
        assert st["frac_flip"] <= max(n_views * GATE_FLIP_FRAC, 1.5 / st["n"]), f"{label}: {st}"
    return st


def assert_cost_parity(hip, orc, path=0, flip_frac=0.0, label="", n_views=4):
    if path in (0, 4):
        return assert_tolerant_parity(hip, orc, n_views=n_views, label=label)
    if path == 1:
        st = cost_stats(hip, orc, 0.0, 0.0)
        print(f"[parity {label} generic] {st}")
        assert st["finite"] and st["frac_bitwise"] == 1.0, f"{label}: generic kernel not bitwise: {st}"
        return st
    st = cost_stats(hip, orc, WORKLIST_ATOL, WORKLIST_RTOL)
    print(f"[parity {label} fast(path={path})] {st}")
    assert st["finite"], f"{label}: non-finite values in the HIP cost volume"
    assert st["frac_flip"] <= flip_frac, f"{label}: {st}"
    return st
