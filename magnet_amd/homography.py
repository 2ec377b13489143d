"""Host mirror of the reference's `models/submodules/homography.py` for the consistency-weighted
matcher.  Same function name, argument order and meaning as the reference so that
`import magnet_amd.homography as homography` drops in at models/MAGNET.py:160-164; all compute goes
through the C ABI (magnet_amd.lib -> libmagnet_hip.so).  No CPU path exists here.

Two entry points:

* `est_costvolume_CW(d_volume, ref_feat, nghbr_feat, ref_gmms, nghbr_gmms, R, t, is_valid,
  cam_intrins, thres)` — the reference signature (homography.py:79-80).  Packs the NCHW features
  on every call, exactly as stateless as the reference.
* `CostVolumeCW` — the per-forward state the MAGNET loop uses: features are packed ONCE per
  forward (they do not change across refinement iterations, MAGNET.py:142-144 vs :151), intrinsics /
  validity live on the device, candidates are sampled inside the kernel from (mu, sigma, k_list).
"""
from __future__ import annotations

import weakref
import zlib

import torch

from . import lib

# cam_intrins arrive as CPU tensors (the reference re-uploads them per call and per batch item,
# homography.py:89-90).  Cache device copies keyed on the CPU tensor's storage + version.
_INTRINS_CACHE: dict = {}
_VALID_CACHE: dict = {}


_SMALL_BYTES = 1 << 20


def _to_device_cached(t: torch.Tensor, device, dtype, cache: dict):
    """Device copy of a CPU tensor the caller hands over on every call (cam_intrins, homography.py:89-90 re-uploads them).
    <= 1 MB (intM; ray tables of small batches): keyed on a digest of the WHOLE content — a fresh tensor with the same content
    (what a DataLoader hands over every step) hits, an in-place edit anywhere misses.
    Larger (the ray table of a big batch, ~15 MB at B = 64: hashing it would cost milliseconds per call): keyed on the tensor
    OBJECT (weakref identity + data_ptr + _version) plus a strided content sample; a different object never hits, entries whose
    tensor has died are dropped, at most 4 large copies stay resident."""
    if t.is_cuda:
        return t.to(device=device, dtype=dtype).contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes <= _SMALL_BYTES:
        buf = memoryview(t.contiguous().numpy()).cast("B")
        digest = bytes(buf) if len(buf) <= 4096 else (zlib.crc32(buf), zlib.adler32(buf), len(buf))
        key = ("small", tuple(t.shape), str(t.dtype), str(device), dtype, digest)
        hit = cache.get(key)
        if hit is not None:
            return hit[1]
        d = t.to(device=device, dtype=dtype).contiguous()
        cache[key] = (None, d)
    else:
        flat = t.reshape(-1)
        sample = flat[:: max(1, flat.numel() // 1024)]
        key = ("large", t.data_ptr(), t._version, tuple(t.shape), str(t.dtype), str(device), dtype, bytes(sample.contiguous().numpy().tobytes()))
        hit = cache.get(key)
        if hit is not None and hit[0]() is t:
            return hit[1]
        for k_ in [k_ for k_, v_ in cache.items() if k_[0] == "large" and (v_[0]() is None or k_ == key)]:
            del cache[k_]                                       # dead owners, and a stale entry of this key
        large = [k_ for k_ in cache if k_[0] == "large"]
        while len(large) >= 4:
            del cache[large.pop(0)]
        d = t.to(device=device, dtype=dtype).contiguous()
        cache[key] = (weakref.ref(t), d)
    if len(cache) > 64:
        for k_ in list(cache)[:32]:
            del cache[k_]
    return d


def _valid_to_device(is_valid: torch.Tensor, device):
    if is_valid.is_cuda:
        return is_valid.to(device=device, dtype=torch.int32).contiguous()
    key = (bytes(is_valid.to(torch.int32).contiguous().numpy().tobytes()), tuple(is_valid.shape), str(device))
    d = _VALID_CACHE.get(key)
    if d is None:
        if len(_VALID_CACHE) > 256:
            _VALID_CACHE.clear()
        d = is_valid.to(device=device, dtype=torch.int32).contiguous()
        _VALID_CACHE[key] = d
    return d


def _poses_from_Rt(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    B, V = R.shape[:2]
    poses = torch.zeros((B, V, 4, 4), dtype=torch.float32, device=R.device)
    poses[:, :, :3, :3] = R
    poses[:, :, :3, 3] = t
    poses[:, :, 3, 3] = 1.0
    return poses


class CostVolumeCW:
    """Per-forward matcher state: packed features + device-side geometry.

    feat_dtype: 'fp32' (bit-faithful storage) or 'bf16' (features rounded once to bf16, fp32 math —
    BASELINE config C2; parity is defined against the oracle fed the same rounded features)."""

    # When set to a list, every launch appends a (start, end) pair of HIP events recorded on the launch
    # stream around the fused kernel (bench.py's live per-launch timing for the roofline figure).
    event_sink = None

    def __init__(self, ref_feat, nghbr_feat, nghbr_gmms, nghbr_poses, is_valid, cam_intrins, thres,
                 feat_dtype="fp32", path: int = 0, packed=None):
        """packed = (ref_cl (B,h,w,F), src_pad (V*B,h+2,w+2,F)): features already in the kernel's layouts (the
        matrix-core F-Net writes them directly, magnet_amd/fnet.py); ref_feat / nghbr_feat are then ignored."""
        if packed is not None:
            self.ref_cl, self.src_pad = packed
            dev = self.ref_cl.device
            self.fe = lib.feat_enum(self.ref_cl.dtype)
            self.B, self.h, self.w, self.F = self.ref_cl.shape
            self.V = self.src_pad.shape[0] // self.B
        else:
            dev = ref_feat.device
            if not ref_feat.is_cuda:
                raise lib.MagnetError("CostVolumeCW: features must be on the GPU (no CPU fallback)")
            self.fe = lib.feat_enum(feat_dtype)
            self.B, self.F, self.h, self.w = ref_feat.shape
            self.V = nghbr_feat.shape[0] // self.B
            self.ref_cl = lib.pack_features(ref_feat.detach().float().contiguous(), self.fe, pad=0)
            self.src_pad = lib.pack_features(nghbr_feat.detach().float().contiguous(), self.fe, pad=1)
        # the source (mu, sigma) maps are packed on first use, in the layout the selected kernel reads: the production matcher
        # for D > 32 takes the quad form (3 fma per bilinear sample), every other kernel the interleaved zero-bordered map
        self._gmm_nchw = nghbr_gmms.detach().float().contiguous()
        self._gmm_pad = None
        self._gmm_quad = None
        self.poses = nghbr_poses.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.is_valid = _valid_to_device(is_valid, dev)
        self.intM = _to_device_cached(cam_intrins["intM"], dev, torch.float32, _INTRINS_CACHE)
        # rays: the loader's (B,3,h*w) table, or — when the dict only carries 'ray_params' (B,8) float64 (magnet_amd.data
        # with_table=False) — generated inside the kernel from those 8 scalars (worklist kernel: table built on the device)
        self.rays, self.ray_params = None, None
        if "unit_ray_array_2D" in cam_intrins:
            self.rays = _to_device_cached(cam_intrins["unit_ray_array_2D"], dev, torch.float32, _INTRINS_CACHE)
        else:
            self.ray_params = _to_device_cached(cam_intrins["ray_params"], dev, torch.float64, _INTRINS_CACHE)
            if (path & 0xff) == 3:
                self.rays = lib.make_rays(self.ray_params, self.h, self.w)
        self.kappa = float(thres)
        self.path = path

    def __call__(self, ref_gmm=None, k_list=None, d_volume=None, out=None, stats=None, out_split=None, gate_bits=None):
        if d_volume is not None:
            d_volume = d_volume.detach().float().contiguous()
        if ref_gmm is not None:
            ref_gmm = ref_gmm.detach().float().contiguous()
        sink = CostVolumeCW.event_sink
        D = d_volume.shape[1] if d_volume is not None else len(k_list)
        # the D > 32 production kernel reads the quad-form (mu, sigma) map; matching grids wider than 512 (long epipolar segments)
        # go to the round-2 kernel, which reads the interleaved one (cost_volume_v3.hip:launch_cv_v3 decides; mirrored here only
        # to pack the right map first — a wrong guess costs one MAGNET_E_SHAPE retry below, not a wrong result)
        quad = (self.path & 0xff) in (0, 4) and D > 32 and d_volume is None and stats is None and self.w <= 512
        if quad and self._gmm_quad is None:
            self._gmm_quad = lib.pack_gmm_quad(self._gmm_nchw)
        if not quad and self._gmm_pad is None:
            self._gmm_pad = lib.pack_gmm(self._gmm_nchw)
        if sink is not None:                                  # the event pair brackets the matcher launch alone: the map is packed above
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()

        def launch():
            return lib.cost_volume_cw(self.ref_cl, self.src_pad, self._gmm_pad, self.poses, self.is_valid,
                                      self.intM, self.rays, self.kappa, ref_gmm=ref_gmm, k_list=k_list,
                                      d_volume=d_volume, out=out, path=self.path, stats=stats, out_split=out_split,
                                      gate_bits=gate_bits, ray_params=self.ray_params, src_gmm_quad=self._gmm_quad if quad else None)
        try:
            res = launch()
        except lib.MagnetError as e:
            # the one condition handled here: the kernel that reads the quad form declined the shape and the interleaved map was
            # not packed yet (the library says so with MAGNET_E_SHAPE); everything else propagates
            if not (e.code == lib.E_SHAPE and quad and self._gmm_pad is None and (self.path & 0xff) in (0, 4)):
                raise
            self._gmm_pad = lib.pack_gmm(self._gmm_nchw)
            if sink is not None:
                e0.record()                                   # re-armed: the declined launch and the pack are not kernel time
            res = launch()
        if sink is not None:
            e1.record()
            sink.append((e0, e1))
        return res


def est_costvolume_CW(d_volume, ref_feat, nghbr_feat, ref_gmms, nghbr_gmms,
                      R, t, is_valid, cam_intrins, thres, feat_dtype="fp32"):
    """Drop-in for homography.est_costvolume_CW (reference homography.py:79-121).

    d_volume (B,D,h,w); ref_feat (B,F,h,w); nghbr_feat (V*B,F,h,w) view-major; ref_gmms is accepted
    and ignored exactly like the reference (never read there); nghbr_gmms (V*B,2,h,w); R (B,V,3,3);
    t (B,V,3); is_valid (B,V) int (CPU or GPU); cam_intrins dict of (CPU or GPU) tensors; thres = kappa.
    Returns (B,D,h,w) fp32 on ref_feat.device."""
    poses = _poses_from_Rt(R.detach().float(), t.detach().float())
    cv = CostVolumeCW(ref_feat, nghbr_feat, nghbr_gmms, poses, is_valid, cam_intrins, thres, feat_dtype)
    return cv(d_volume=d_volume)


class _CostVolumeF(torch.autograd.Function):
    """Raw (pre-softmax) feature-matching volume of est_costvolume_F with its hand-written backward
    (cost_volume_f_bwd.hip).  Features are differentiable; bins, poses and intrinsics are data."""

    @staticmethod
    def forward(ctx, ref_feat, nghbr_feat, bins, poses, is_valid, intM, rays, path, bwd_path=0):
        ref_cl = lib.pack_features(ref_feat.detach().float().contiguous(), lib.FEAT_F32, pad=0)
        src_pad = lib.pack_features(nghbr_feat.detach().float().contiguous(), lib.FEAT_F32, pad=1)
        ctx.bins = bins
        ctx.bwd_path = bwd_path
        ctx.save_for_backward(ref_cl, src_pad, poses, is_valid, intM, rays)
        return lib.cost_volume_cw(ref_cl, src_pad, None, poses, is_valid, intM, rays, 0.0, k_list=bins,
                                  path=path, mode=1)

    @staticmethod
    def backward(ctx, grad_cost):
        ref_cl, src_pad, poses, is_valid, intM, rays = ctx.saved_tensors
        g_ref_cl, g_src_pad = lib.cost_volume_f_backward(ref_cl, src_pad, poses, is_valid, intM, rays, ctx.bins,
                                                         grad_cost.float().contiguous(), path=ctx.bwd_path)
        # channel-last -> NCHW (+ drop the gradient of the zero padding): layout plumbing, outside the hot loop
        g_ref = g_ref_cl.permute(0, 3, 1, 2).contiguous()
        g_src = g_src_pad[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).contiguous()
        return g_ref, g_src, None, None, None, None, None, None, None


def est_costvolume_F(d_center, ref_feat, nghbr_feat, R, t, is_valid, cam_intrins, path: int = 0, bwd_path: int = 0):
    """Drop-in for homography.est_costvolume_F (reference homography.py:10-46), the matching volume the
    F-Net is trained through (MAGNET.py:197-200): differentiable w.r.t. ref_feat and nghbr_feat.

    d_center (1,D,1,1) fixed depth bins (CPU or GPU tensor); ref_feat (B,F,h,w); nghbr_feat (V*B,F,h,w)
    view-major; R (B,V,3,3); t (B,V,3); is_valid (B,V); cam_intrins dict.  Returns softmax over D of the
    view-averaged feature correlation, (B,D,h,w) fp32.  path = 1 selects the bit-faithful generic kernel."""
    if not ref_feat.is_cuda:
        raise lib.MagnetError("est_costvolume_F: features must be on the GPU (no CPU fallback)")
    dev = ref_feat.device
    bins = [float(v) for v in d_center.detach().float().reshape(-1).cpu().tolist()]
    poses = _poses_from_Rt(R.detach().float(), t.detach().float()).to(dev).contiguous()
    iv = _valid_to_device(is_valid, dev)
    intM = _to_device_cached(cam_intrins["intM"], dev, torch.float32, _INTRINS_CACHE)
    if "unit_ray_array_2D" in cam_intrins:
        rays = _to_device_cached(cam_intrins["unit_ray_array_2D"], dev, torch.float32, _INTRINS_CACHE)
    else:       # table-free dict (magnet_amd.data with_table=False): the F forward / backward read the table, build it on the device
        rays = lib.make_rays(_to_device_cached(cam_intrins["ray_params"], dev, torch.float64, _INTRINS_CACHE),
                             ref_feat.shape[2], ref_feat.shape[3])
    raw = _CostVolumeF.apply(ref_feat, nghbr_feat, bins, poses, iv, intM, rays, path, bwd_path)
    return torch.softmax(raw, dim=1)                                   # homography.py:45
