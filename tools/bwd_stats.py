import os, sys, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd import lib, synth
dev = torch.device('cuda:0')
for name, cam, h, w, V, dmax in (("scannet", "scannet", 120, 160, 4, 10.0), ("kitti", "kitti", 88, 304, 2, 80.0)):
    wl = synth.Workload(name, cam, h, w, V=V, D=80, F=64)
    B = 16
    inp = synth.make_inputs(wl, B=B, seed=1, smooth_feats=True)
    D = 80
    b = np.exp(np.log(dmax + 1 - 1e-3) * np.arange(D + 1) / D) - (1 - 1e-3)
    bins = [float(x) for x in ((b[:-1] + b[1:]) / 2).astype(np.float32)]
    ref_cl = lib.pack_features(inp["ref_feat"].to(dev), lib.FEAT_F32, pad=0); src_pad = lib.pack_features(inp["nghbr_feat"].to(dev), lib.FEAT_F32, pad=1)
    g = torch.randn(B, D, h, w, device=dev)
    common = (ref_cl, src_pad, inp["nghbr_poses"].to(dev), inp["is_valid"].int().to(dev), inp["cam_intrins"]["intM"].to(dev), inp["cam_intrins"]["unit_ray_array_2D"].to(dev), bins, g)
    for path in (0, 0x2000):
        st = torch.zeros(4, dtype=torch.int32, device=dev)
        lib.cost_volume_f_backward(*common, path=path, stats=None if os.environ.get('NO_STATS') else st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): lib.cost_volume_f_backward(*common, path=path)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(name, hex(path), f"{dt*1e3:.2f} ms", st.tolist())
