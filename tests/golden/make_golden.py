#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (Python, CPU).

Runs only in the build container (needs /root/reference); the GPU box never sees the reference,
only the .npz files this script writes.  Everything written is DATA (inputs, expected outputs,
checksums) — no reference source text.  Usage:  python tests/golden/make_golden.py

Vector set:
  G1 depth_sampling            G2 est_costvolume_CW (tiny full tensors + C1-shape subsample)
  G3 _compute_cost_CW fp64+gates for one (b,v)       G4 GNET.forward        G5 upsample_depth_via_mask
  G6 MAGNET.forward with stub D-Net/F-Net (I=3)      G7 compute_depth_errors G8 data_preprocess
It also prints how the CPU oracle (oracle/) compares with the reference on each case.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

import models.submodules.homography as ref_h          # noqa: E402  (reference)
import models.MAGNET as ref_m                          # noqa: E402  (reference)
import utils.utils as ref_u                            # noqa: E402  (reference)

from magnet_amd import synth                           # noqa: E402
from oracle import oracle                              # noqa: E402
from tests.stubs import StubDNet, StubFNet, make_args, seeded_magnet_weights, seeded_fnet_state, procedural_images  # noqa: E402

torch.set_num_threads(8)


def sha(*arrays) -> str:
    m = hashlib.sha256()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


def tiny_case(seed=3):
    """B=2, V=3, F=8, 12x16, D=5 with one invalid view, one pose that throws pixels out of
    bounds / behind the camera, and one pixel row with huge sigma."""
    wl = synth.Workload("tiny", "scannet", 12, 16, V=3, D=5, F=8)
    inp = synth.make_inputs(wl, B=2, seed=seed, invalid=[(1, 1)])
    poses = inp["nghbr_poses"]
    # wild pose: 90deg yaw + large backwards translation -> many samples behind the camera / OOB
    poses[0, 2, :3, :3] = torch.tensor([[0., 0., 1.], [0., 1., 0.], [-1., 0., 0.]])
    poses[0, 2, :3, 3] = torch.tensor([0.3, -0.2, -2.5])
    inp["ref_gmms"][1, 1, 5, :] = 3.0      # sigma = 3 m at mu in [1,4]: candidates cross z=0
    return wl, inp


def run_ref_cw(inp, k_list, thres=5):
    dv = synth.depth_volume_from_gmm(inp["ref_gmms"], k_list)
    R = inp["nghbr_poses"][:, :, :3, :3]
    t = inp["nghbr_poses"][:, :, :3, 3]
    out = ref_h.est_costvolume_CW(dv, inp["ref_feat"], inp["nghbr_feat"], inp["ref_gmms"],
                                  inp["nghbr_gmms"], R, t, inp["is_valid"], inp["cam_intrins"], thres)
    return dv, out


def compare(name, ref_out, orc_out, gates_ref=None, gates_orc=None):
    ref_out = np.asarray(ref_out); orc_out = np.asarray(orc_out)
    diff = np.abs(ref_out - orc_out)
    msg = f"[{name}] oracle vs reference: max|d|={diff.max():.3e} mean|d|={diff.mean():.3e} " \
          f"frac(|d|>1e-4)={np.mean(diff > 1e-4):.3e} mean|ref|={np.abs(ref_out).mean():.3f}"
    if gates_ref is not None:
        msg += f" gate mismatches={int((gates_ref != gates_orc).sum())}/{gates_ref.size}"
    print(msg)


def main():
    out = {}

    # ---- G1 ------------------------------------------------------------------------------
    class _A:  # minimal stand-in carrying the two attributes depth_sampling reads
        pass
    for D in (5, 16, 64, 128):
        a = _A(); a.sampling_range = 3; a.n_samples = D
        k_ref = np.array(ref_m.MAGNET.depth_sampling(a), dtype=np.float64)
        out[f"G1_k_D{D}"] = k_ref
        k_orc = np.array(oracle.depth_sampling(3, D))
        print(f"[G1 D={D}] max|k_oracle-k_ref|={np.abs(k_orc - k_ref).max():.2e}")

    # ---- G2/G3 tiny ----------------------------------------------------------------------
    wl, inp = tiny_case()
    k5 = list(out["G1_k_D5"])
    dv, cv = run_ref_cw(inp, k5)
    for k, v in inp.items():
        if k == "cam_intrins":
            out["G2_intM"] = v["intM"].numpy(); out["G2_rays"] = v["unit_ray_array_2D"].numpy()
        else:
            out["G2_" + k] = v.numpy()
    out["G2_d_volume"] = dv.numpy(); out["G2_cost"] = cv.numpy()
    # per-(b,v) fp64 weighted cost + gate bits straight from _compute_cost_CW (G3), all valid (b,v)
    B, V, D, h, w = 2, wl.V, wl.D, wl.h, wl.w
    g3 = np.zeros((B, V, D, h, w), np.float64); g3_gate = np.zeros((B, V, D, h, w), np.uint8)
    for b in range(B):
        K = inp["cam_intrins"]["intM"][b]; Ray = inp["cam_intrins"]["unit_ray_array_2D"][b]
        for v in range(V):
            if inp["is_valid"][b, v].item() != 1:
                continue
            Rm = inp["nghbr_poses"][b, v, :3, :3]; tv = inp["nghbr_poses"][b, v, :3, 3]
            idm = torch.eye(3)
            args = dict(
                ref_feat_=inp["ref_feat"][b:b + 1].repeat(D, 1, 1, 1),
                nghbr_feat_=inp["nghbr_feat"][B * v + b][None].repeat(D, 1, 1, 1),
                nghbr_mu_=inp["nghbr_gmms"][B * v + b, 0:1][None].repeat(D, 1, 1, 1),
                nghbr_sigma_=inp["nghbr_gmms"][B * v + b, 1:2][None].repeat(D, 1, 1, 1),
                d_volume=dv[b], term1_cam=idm.matmul(tv).reshape(3, 1), term2_cam=idm.matmul(Rm).matmul(Ray),
                term1_pix=K.matmul(tv).reshape(3, 1), term2_pix=K.matmul(Rm).matmul(Ray),
                device=torch.device("cpu"), thres=5)
            wc = ref_h._compute_cost_CW(**args)
            g3[b, v] = wc.numpy()
            # the function returns feat_cost*gate only; its non-zero pattern is the gate wherever
            # feat_cost != 0 (out-of-bounds samples have feat_cost == 0 and sigma_w == 0 -> gate 0)
            g3_gate[b, v] = (wc.numpy() != 0).astype(np.uint8)
    out["G3_weighted_cost_f64"] = g3; out["G3_gate_nonzero"] = g3_gate
    o_cv, o_g, o_fc = oracle.cost_volume_cw(None, inp["ref_gmms"], k5, inp["ref_feat"], inp["nghbr_feat"],
                                            inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"],
                                            inp["cam_intrins"]["intM"], inp["cam_intrins"]["unit_ray_array_2D"],
                                            5.0, return_aux=True)
    compare("G2 tiny", cv.numpy(), o_cv)
    orc_nonzero = ((o_fc * o_g) != 0).astype(np.uint8)
    print(f"[G3 tiny] nonzero-pattern mismatches: {int((orc_nonzero != g3_gate).sum())}/{g3_gate.size}; "
          f"max|weighted diff|={np.abs(o_fc * o_g - g3).max():.3e}; gated-on frac={g3_gate.mean():.3f}")
    o_dv = oracle.depth_volume(inp["ref_gmms"], k5)
    print(f"[A2] depth_volume bitwise equal to reference: {np.array_equal(o_dv, dv.numpy())}")

    # ---- G2 at the C1 shape (inputs regenerated from the seed; checksum stored) -------------
    for name, B_, seed in (("C1", 1, 0),):
        wl1 = synth.WORKLOADS[name]
        inp1 = synth.make_inputs(wl1, B=B_, seed=seed)
        k = list(out[f"G1_k_D{wl1.D}"])
        dv1, cv1 = run_ref_cw(inp1, k)
        out[f"G2_{name}_input_sha"] = np.frombuffer(bytes.fromhex(sha(
            inp1["ref_feat"].numpy(), inp1["nghbr_feat"].numpy(), inp1["ref_gmms"].numpy(),
            inp1["nghbr_gmms"].numpy(), inp1["nghbr_poses"].numpy())), dtype=np.uint8)
        out[f"G2_{name}_cost_sub"] = cv1.numpy()[:, :, ::5, ::7].copy()
        out[f"G2_{name}_cost_sum"] = np.array([cv1.double().sum().item(), cv1.abs().double().sum().item()])
        o1 = oracle.cost_volume_cw(None, inp1["ref_gmms"], k, inp1["ref_feat"], inp1["nghbr_feat"],
                                   inp1["nghbr_gmms"], inp1["nghbr_poses"], inp1["is_valid"],
                                   inp1["cam_intrins"]["intM"], inp1["cam_intrins"]["unit_ray_array_2D"], 5.0)
        compare(f"G2 {name}", cv1.numpy(), o1)

    # ---- G4 GNET --------------------------------------------------------------------------
    torch.manual_seed(11)
    gnet = ref_m.GNET(ch_in=256 + 5, ch_out=2)
    x = torch.randn(2, 261, 12, 16); prev = inp["ref_gmms"]
    with torch.no_grad():
        y = gnet(x, prev)
        raw = gnet.gnet(x)
    out["G4_x"] = x.numpy(); out["G4_prev"] = prev.numpy(); out["G4_out"] = y.numpy(); out["G4_raw"] = raw.numpy()
    for kk, vv in gnet.state_dict().items():
        if vv.numel() <= 4096 * 8:
            out["G4_sd_" + kk] = vv.numpy()
    out["G4_sd_sha"] = np.frombuffer(bytes.fromhex(sha(*[v.numpy() for v in gnet.state_dict().values()])), np.uint8)
    print(f"[G4] oracle gaussian_update vs reference: max|d|="
          f"{np.abs(oracle.gaussian_update(raw.numpy(), prev.numpy()) - y.numpy()).max():.3e}")

    # ---- G5 upsample ----------------------------------------------------------------------
    torch.manual_seed(12)
    dep = torch.rand(2, 2, 6, 8) * 3 + 1; msk = torch.randn(2, 9 * 16, 6, 8)
    up = ref_m.upsample_depth_via_mask(dep, msk, 4)
    out["G5_depth"] = dep.numpy(); out["G5_mask"] = msk.numpy(); out["G5_out"] = up.numpy()
    print(f"[G5] oracle upsample vs reference: max|d|="
          f"{np.abs(oracle.upsample_depth_via_mask(dep.numpy(), msk.numpy(), 4) - up.numpy()).max():.3e}")

    # ---- G6 full MAGNET.forward with stub backbones ----------------------------------------
    args = make_args(D=5, iters=3, dpv_h=12, dpv_w=16)
    model = object.__new__(ref_m.MAGNET)
    torch.nn.Module.__init__(model)
    model.args = args
    model.d_net = StubDNet(seed=21); model.f_net = StubFNet(seed=22, fdim=8)
    model.sampling_range = args.MAGNET_sampling_range; model.n_samples = args.MAGNET_num_samples
    model.weighting = args.MAGNET_mvs_weighting
    model.train_iter = args.MAGNET_num_train_iter; model.test_iter = args.MAGNET_num_test_iter
    model.dpv_height = args.dpv_height; model.dpv_width = args.dpv_width
    model.k_list = model.depth_sampling(); model.downsample_ratio = args.downsample_ratio
    model.g_net = ref_m.GNET(ch_in=256 + model.n_samples, ch_out=2)
    h_dim = 128
    model.mask_head = torch.nn.Sequential(
        torch.nn.Conv2d(256, h_dim, 3, padding=1), torch.nn.ReLU(inplace=True),
        torch.nn.Conv2d(h_dim, h_dim, 1), torch.nn.ReLU(inplace=True),
        torch.nn.Conv2d(h_dim, h_dim, 1), torch.nn.ReLU(inplace=True),
        torch.nn.Conv2d(h_dim, 9 * 16, 1))
    model.upsample_depth = ref_m.upsample_depth_via_mask
    seeded_magnet_weights(model, seed=23)
    model.eval()
    gen = torch.Generator().manual_seed(31)
    B6, V6 = 2, 3
    ref_img = torch.rand(B6, 3, 48, 64, generator=gen)
    nghbr_imgs = torch.rand(V6 * B6, 3, 48, 64, generator=gen)
    wl6 = synth.Workload("g6", "scannet", 12, 16, V=V6, D=5, F=8)
    poses6 = synth.make_poses("scannet", B6, V6, gen)
    valid6 = torch.ones(B6, V6, dtype=torch.int32); valid6[0, 1] = 0
    intr6 = synth.make_intrinsics("scannet", 12, 16, B6)
    with torch.no_grad():
        preds = model(ref_img, nghbr_imgs, poses6, valid6, intr6, mode="test")
    out["G6_ref_img"] = ref_img.numpy(); out["G6_nghbr_imgs"] = nghbr_imgs.numpy()
    out["G6_poses"] = poses6.numpy(); out["G6_is_valid"] = valid6.numpy()
    for i, p in enumerate(preds):
        out[f"G6_pred{i}"] = p.numpy()
    out["G6_weights_sha"] = np.frombuffer(bytes.fromhex(sha(
        *[v.numpy() for v in model.g_net.state_dict().values()],
        *[v.numpy() for v in model.mask_head.state_dict().values()])), np.uint8)
    print(f"[G6] MAGNET.forward(stub nets): {len(preds)} outputs of shape {tuple(preds[0].shape)}, "
          f"mean mu={preds[-1][:, 0].mean():.4f} mean sigma={preds[-1][:, 1].mean():.4f}")

    # ---- G7 metrics -------------------------------------------------------------------------
    rng = np.random.RandomState(5)
    gt = rng.uniform(0.5, 8, size=(4000,)).astype(np.float32)
    pr = (gt * rng.uniform(0.8, 1.25, size=gt.shape)).astype(np.float32)
    var = rng.uniform(1e-7, 0.5, size=gt.shape).astype(np.float32)
    m = ref_u.compute_depth_errors(gt, pr, var.copy())
    out["G7_gt"] = gt; out["G7_pred"] = pr; out["G7_var"] = var
    keys = sorted(m.keys()); out["G7_keys"] = np.array(keys); out["G7_vals"] = np.array([float(m[k]) for k in keys])
    mo = oracle.compute_depth_errors(gt, pr, var.copy())
    print(f"[G7] oracle metrics max rel diff: {max(abs(float(mo[k]) - float(m[k])) / (abs(float(m[k])) + 1e-12) for k in keys):.2e}")

    # ---- G8 data_preprocess -------------------------------------------------------------------
    g = torch.Generator().manual_seed(8)
    def rand_ext(n):
        om = torch.randn(n, 3, generator=g, dtype=torch.float64) * 0.3
        E = torch.zeros(n, 4, 4, dtype=torch.float64)
        E[:, :3, :3] = synth._rodrigues(om); E[:, :3, 3] = torch.randn(n, 3, generator=g, dtype=torch.float64)
        E[:, 3, 3] = 1
        return E
    exts = [rand_ext(3) for _ in range(5)]          # 5 frames: ref = index 2
    exts[0][1] = float("nan"); exts[2][2] = float("nan")   # a NaN neighbour, a NaN reference
    data_array = [{"extM": e} for e in exts]
    # environment shim (not reference logic): with numpy 2.x np.linalg.inv(Tensor) returns a Tensor,
    # which the reference then feeds to torch.from_numpy; hand it an ndarray as numpy 1.x did.
    _inv = np.linalg.inv
    ref_u.np.linalg.inv = lambda a: _inv(np.asarray(a))
    try:
        _, _, poses8, valid8 = ref_u.data_preprocess(data_array, 3)
    finally:
        ref_u.np.linalg.inv = _inv
    out["G8_exts"] = np.stack([e.numpy() for e in exts]); out["G8_poses"] = poses8.numpy(); out["G8_valid"] = valid8.numpy()
    po, vo = oracle.relative_poses(exts[2].numpy(), [exts[i].numpy() for i in (0, 1, 3, 4)])
    print(f"[G8] oracle relative_poses: max|d|={np.abs(po - poses8.numpy()).max():.2e} valid equal={np.array_equal(vo, valid8.numpy())}")

    # ---- G9 est_costvolume_F: forward (raw + softmax) and autograd gradients ---------------------------
    wl9 = synth.Workload("g9", "scannet", 12, 16, V=3, D=10, F=8)
    inp9 = synth.make_inputs(wl9, B=2, seed=9, invalid=[(0, 2)])
    inp9["nghbr_poses"][1, 0, :3, 3] = torch.tensor([0.4, -0.3, -1.2])       # strong parallax / out-of-image samples
    d_center = torch.tensor(np.exp(np.log(10.0 + 1 - 1e-3) * (np.arange(10) + 0.5) / 10) - (1 - 1e-3), dtype=torch.float32).view(1, 10, 1, 1)
    rf = inp9["ref_feat"].clone().requires_grad_(True); sf = inp9["nghbr_feat"].clone().requires_grad_(True)
    R9 = inp9["nghbr_poses"][:, :, :3, :3]; t9 = inp9["nghbr_poses"][:, :, :3, 3]
    # instrumentation (not reference logic): expose the cost volume before the final softmax
    _sm = ref_h.F.softmax
    ref_h.F.softmax = lambda x, dim: x
    try:
        raw9 = ref_h.est_costvolume_F(d_center, rf, sf, R9, t9, inp9["is_valid"], inp9["cam_intrins"])
    finally:
        ref_h.F.softmax = _sm
    gout9 = torch.randn(raw9.shape, generator=torch.Generator().manual_seed(91))
    (raw9 * gout9).sum().backward()
    with torch.no_grad():
        soft9 = ref_h.est_costvolume_F(d_center, rf, sf, R9, t9, inp9["is_valid"], inp9["cam_intrins"])
    out["G9_d_center"] = d_center.numpy(); out["G9_ref_feat"] = inp9["ref_feat"].numpy(); out["G9_nghbr_feat"] = inp9["nghbr_feat"].numpy()
    out["G9_poses"] = inp9["nghbr_poses"].numpy(); out["G9_is_valid"] = inp9["is_valid"].numpy()
    out["G9_intM"] = inp9["cam_intrins"]["intM"].numpy(); out["G9_rays"] = inp9["cam_intrins"]["unit_ray_array_2D"].numpy()
    out["G9_raw"] = raw9.detach().numpy(); out["G9_softmax"] = soft9.numpy(); out["G9_gout"] = gout9.numpy()
    out["G9_grad_ref"] = rf.grad.numpy(); out["G9_grad_src"] = sf.grad.numpy()
    o_raw, o_gr, o_gs = oracle.cost_volume_f_raw(d_center, inp9["ref_feat"], inp9["nghbr_feat"], inp9["nghbr_poses"], inp9["is_valid"],
                                                 inp9["cam_intrins"]["intM"], inp9["cam_intrins"]["unit_ray_array_2D"], gout=gout9)
    print(f"[G9] oracle est_costvolume_F raw bitwise: {np.array_equal(o_raw, raw9.detach().numpy())} "
          f"(max|d|={np.abs(o_raw - raw9.detach().numpy()).max():.2e}); softmax max|d|="
          f"{np.abs(oracle.softmax_dim1(o_raw) - soft9.numpy()).max():.2e}; grad_ref max|d|={np.abs(o_gr - rf.grad.numpy()).max():.2e} "
          f"grad_src max|d|={np.abs(o_gs - sf.grad.numpy()).max():.2e} (|grad| max {rf.grad.abs().max():.2f}/{sf.grad.abs().max():.2f})")

    # ---- G10 F-Net (PSMNet, eval mode) on a procedural image with crc-seeded weights: sparse output samples ----------
    from models.submodules.F_psmnet import PSMNet as RefPSMNet
    ref_f = seeded_fnet_state(RefPSMNet(feature_dim=64), seed=10).eval()
    img10 = procedural_images(2, 256, 320)
    with torch.no_grad():
        f10 = ref_f(img10)                                                       # (2, 64, 64, 80)
    out["G10_feat_sparse"] = f10[:, :, ::4, ::4].contiguous().numpy()
    out["G10_feat_absmean"] = f10.abs().mean(dim=(0, 2, 3)).numpy()
    from magnet_amd.fnet import PSMNet as OurPSMNet
    ours = seeded_fnet_state(OurPSMNet(feature_dim=64), seed=10).eval()
    with torch.no_grad():
        o10 = ours(img10)
    print(f"[G10] F-Net reference output {tuple(f10.shape)} |max| {f10.abs().max():.3f}; magnet_amd.fnet.PSMNet (torch path) "
          f"max|d| = {(o10 - f10).abs().max():.2e}")

    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB, {len(out)} arrays")


if __name__ == "__main__":
    main()
