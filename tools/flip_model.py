"""Dev tool (CPU): model the production matcher geometry in numpy fp32 and count gate flips against the oracle (tools/README.md).
usage: python tools/flip_model.py [workload]"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from magnet_amd import synth
from oracle import oracle
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
wlname = sys.argv[1] if len(sys.argv) > 1 else "C2"
wl = synth.WORKLOADS[wlname]
B = 1
inp = synth.make_inputs(wl, B=B, seed=0, round_bf16=(wl.feat_dtype == "bf16"))
k = oracle.depth_sampling(3, wl.D)
orc, gates, fc = oracle.cost_volume_cw(None, inp["ref_gmms"], k, inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"], inp["cam_intrins"]["intM"], inp["cam_intrins"]["unit_ray_array_2D"], 5.0, return_aux=True)
h, w, V, D = wl.h, wl.w, wl.V, wl.D
K = inp["cam_intrins"]["intM"].numpy()[0].astype(f32)
rays = inp["cam_intrins"]["unit_ray_array_2D"].numpy()[0].astype(f32)  # (3,hw)
mu = inp["ref_gmms"].numpy()[0, 0].reshape(-1); sg = inp["ref_gmms"].numpy()[0, 1].reshape(-1)
kk = np.asarray(k, dtype=np.float64).astype(f32)
sgm = inp["nghbr_gmms"].numpy()   # (V*B,2,h,w)
for variant in ("fma_exactdiv", "fma_rcp1ulp", "nofma_exactdiv"):
    tot = 0; flips = 0
    for v in range(V):
        T = inp["nghbr_poses"].numpy()[0, v].astype(f32)
        R = T[:3, :3]; t = T[:3, 3]
        # table values as the oracle builds them (fp32 with the same roundings: approximate with float64->f32 of dot; tiny diff irrelevant for flip stats)
        KR = (K.astype(np.float64) @ R.astype(np.float64)).astype(f32)
        Kt = (K.astype(np.float64) @ t.astype(np.float64)).astype(f32)
        rp = (KR.astype(np.float64) @ rays.astype(np.float64)).astype(f32)    # (3,hw)
        rcz = (R[2].astype(np.float64) @ rays.astype(np.float64)).astype(f32)
        if variant.startswith("nofma"):
            d = (mu[None, :] + (sg[None, :] * kk[:, None]).astype(f32)).astype(f32)
            Px = (Kt[0] + (rp[0][None] * d).astype(f32)).astype(f32); Py = (Kt[1] + (rp[1][None] * d).astype(f32)).astype(f32); Pz = (Kt[2] + (rp[2][None] * d).astype(f32)).astype(f32)
            zw = (t[2] + (rcz[None] * d).astype(f32)).astype(f32)
        else:
            d = fma(np.broadcast_to(sg[None, :], (D, h*w)), np.broadcast_to(kk[:, None], (D, h*w)), np.broadcast_to(mu[None, :], (D, h*w)))
            bc = lambda a: np.broadcast_to(a, (D, h*w))
            Px = fma(bc(rp[0][None]), d, bc(np.full(1, Kt[0], f32))); Py = fma(bc(rp[1][None]), d, bc(np.full(1, Kt[1], f32))); Pz = fma(bc(rp[2][None]), d, bc(np.full(1, Kt[2], f32)))
            zw = fma(bc(rcz[None]), d, bc(np.full(1, t[2], f32)))
        r = (f32(1.0) / Pz).astype(f32)
        if variant.endswith("rcp1ulp"):
            rng = np.random.default_rng(1)
            r = (r.view(np.int32) + rng.integers(-1, 2, r.shape, dtype=np.int32)).view(f32)
        ixs = fma(Px, r, np.full_like(Px, 0.5)); iys = fma(Py, r, np.full_like(Px, 0.5))   # shifted by +1
        x0 = np.floor(ixs); y0 = np.floor(iys); bx = (ixs - x0).astype(f32); by = (iys - y0).astype(f32)
        ax = (f32(1) - bx).astype(f32); ay = (f32(1) - by).astype(f32)
        inw = (ixs >= 0) & (ixs < w + 1) & (iys >= 0) & (iys < h + 1)
        xi = np.where(inw, x0, 0).astype(np.int64); yi = np.where(inw, y0, 0).astype(np.int64)
        pm = np.pad(sgm[v * B + 0], ((0, 0), (1, 1), (1, 1)))   # (2,h+2,w+2)
        def bil(img):
            a = img[yi, xi]; b_ = img[yi, xi + 1]; c = img[yi + 1, xi]; dd = img[yi + 1, xi + 1]
            vv = (a * (ax * ay).astype(f32)).astype(f32)
            vv = fma(b_, (bx * ay).astype(f32), vv); vv = fma(c, (ax * by).astype(f32), vv); vv = fma(dd, (bx * by).astype(f32), vv)
            return vv
        mu_w = bil(pm[0]); sg_w = bil(pm[1])
        gate = inw & (np.abs((zw - mu_w).astype(f32)) < (sg_w * f32(5.0)).astype(f32))
        og = gates[0, v].reshape(D, -1).astype(bool)
        flips += int((gate != og).sum()); tot += gate.size
    print(wlname, variant, "gate flips", flips, "of", tot, "=", flips / tot, " open frac", og.mean())
