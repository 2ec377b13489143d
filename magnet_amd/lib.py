"""ctypes binding of libmagnet_hip.so (the C ABI in include/magnet_hip.h).

torch is used for device memory and streams only: every call passes `tensor.data_ptr()` and the
current HIP stream through the C boundary.  There is NO CPU fallback — if the library is missing,
or an argument lives on the CPU, the call raises.  torch must be imported before the library is
loaded so that libmagnet_hip.so binds to the same libamdhip64 (same SONAME) torch has loaded.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmagnet_hip.so")

FEAT_F32, FEAT_BF16 = 0, 1
MAX_CANDIDATES = 256

API_SYMBOLS = ("magnet_version", "magnet_last_error", "magnet_device_count", "magnet_pack_features",
               "magnet_pack_gmm", "magnet_pack_gmm_quad",
               "magnet_cost_volume_cw", "magnet_cost_volume_f_backward", "magnet_cost_volume_f_backward_ws",
               "magnet_cost_volume_f_backward_workspace", "magnet_gaussian_update",
               "magnet_upsample_depth")


# argument-error codes of include/magnet_hip.h (positive return values; negative = -(hipError_t))
E_NULL, E_DIM, E_DTYPE, E_ALIGN, E_NODEVICE, E_SHAPE = 1, 2, 3, 4, 5, 6


class MagnetError(RuntimeError):
    """code = the C ABI's return value when the error came from the library (None for host-side checks):
    E_SHAPE = "this kernel / output form does not take the shape" (the only condition callers may fall back on)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


class MagnetCostVolumeArgs(ctypes.Structure):
    """Mirror of `struct MagnetCostVolumeArgs` (include/magnet_hip.h)."""
    _fields_ = [
        ("B", ctypes.c_int32), ("V", ctypes.c_int32), ("F", ctypes.c_int32), ("D", ctypes.c_int32),
        ("h", ctypes.c_int32), ("w", ctypes.c_int32),
        ("kappa", ctypes.c_float), ("feat_dtype", ctypes.c_int32),
        ("ref_feat_cl", ctypes.c_void_p), ("src_feat_pad", ctypes.c_void_p),
        ("src_gmm_pad", ctypes.c_void_p), ("ref_gmm", ctypes.c_void_p),
        ("k_list", ctypes.c_void_p), ("d_volume", ctypes.c_void_p),
        ("poses", ctypes.c_void_p), ("is_valid", ctypes.c_void_p),
        ("intM", ctypes.c_void_p), ("rays", ctypes.c_void_p),
        ("cost", ctypes.c_void_p),
        ("path", ctypes.c_int32), ("stats", ctypes.c_void_p),
        ("cost_batch_stride", ctypes.c_int64),
        ("cost_hi", ctypes.c_void_p), ("cost_lo", ctypes.c_void_p), ("cost_ld", ctypes.c_int64),
        ("mode", ctypes.c_int32),
        ("gate_bits", ctypes.c_void_p),
        ("ray_params", ctypes.c_void_p),
        ("src_gmm_quad", ctypes.c_void_p),
        ("dev_flags", ctypes.c_uint32),
    ]


_lib = None


def use_dev_build():
    """tools/ only: bind to libmagnet_hip_dev.so (python -m magnet_amd.build --dev), the build that honours dev_flags and the
    MAGNET_* environment switches.  Must be called before the first load(); never called by the package itself."""
    global LIB_PATH
    if _lib is not None:
        raise MagnetError("use_dev_build() must be called before the library is loaded")
    LIB_PATH = os.path.join(_HERE, "libmagnet_hip_dev.so")


def load() -> ctypes.CDLL:
    """Load the library (once).  Raises MagnetError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MagnetError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m magnet_amd.build` "
            "(or __graft_entry__.build()). magnet_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    I, P = ctypes.c_int32, ctypes.c_void_p
    lib.magnet_version.restype = ctypes.c_int
    lib.magnet_last_error.restype = ctypes.c_char_p
    lib.magnet_device_count.restype = ctypes.c_int
    lib.magnet_pack_features.restype = ctypes.c_int
    lib.magnet_pack_features.argtypes = [P, P, I, I, I, I, I, I, P]
    lib.magnet_pack_gmm.restype = ctypes.c_int
    lib.magnet_pack_gmm.argtypes = [P, P, I, I, I, P]
    lib.magnet_pack_gmm_quad.restype = ctypes.c_int
    lib.magnet_pack_gmm_quad.argtypes = [P, P, I, I, I, P]
    lib.magnet_cost_volume_cw.restype = ctypes.c_int
    lib.magnet_cost_volume_cw.argtypes = [ctypes.POINTER(MagnetCostVolumeArgs), P]
    lib.magnet_cost_volume_f_backward.restype = ctypes.c_int
    lib.magnet_cost_volume_f_backward.argtypes = [ctypes.POINTER(MagnetCostVolumeArgs), P, P, P, P]
    lib.magnet_gaussian_update.restype = ctypes.c_int
    lib.magnet_gaussian_update.argtypes = [P, P, P, I, I, P]
    lib.magnet_upsample_depth.restype = ctypes.c_int
    lib.magnet_upsample_depth.argtypes = [P, P, P, I, I, I, I, I, P]
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().magnet_last_error().decode("utf-8", "replace")
        raise MagnetError(f"{what} failed (rc={rc}): {msg}", code=rc)


def _dev(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise MagnetError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise MagnetError(f"{name} is on {t.device}; magnet_amd runs on the GPU only (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise MagnetError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise MagnetError(f"{name} must be contiguous")
    return t


def _stream(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def feat_torch_dtype(feat_dtype: int):
    return torch.bfloat16 if feat_dtype == FEAT_BF16 else torch.float32


def feat_enum(dtype) -> int:
    if dtype in (torch.bfloat16, "bf16", "bfloat16", FEAT_BF16):
        return FEAT_BF16
    if dtype in (torch.float32, "fp32", "float32", FEAT_F32):
        return FEAT_F32
    raise MagnetError(f"unsupported feature storage dtype {dtype!r} (fp32 or bf16)")


def pack_features(feat_nchw: torch.Tensor, feat_dtype: int = FEAT_F32, pad: int = 0, out: torch.Tensor | None = None):
    """(N,F,h,w) fp32 NCHW -> (N,h+2*pad,w+2*pad,F) channel-last in fp32 or bf16 storage; pad=1 adds a
    one-texel zero border (the source-view layout of the matcher)."""
    x = _dev(feat_nchw, "feat_nchw", torch.float32)
    N, F, h, w = x.shape
    shape = (N, h + 2 * pad, w + 2 * pad, F)
    if out is None:
        out = torch.empty(shape, dtype=feat_torch_dtype(feat_dtype), device=x.device)
    else:
        _dev(out, "out", feat_torch_dtype(feat_dtype))
        if tuple(out.shape) != shape:
            raise MagnetError(f"out has shape {tuple(out.shape)}, expected {shape}")
    with torch.cuda.device(x.device):
        _check(load().magnet_pack_features(x.data_ptr(), out.data_ptr(), N, F, h, w, feat_dtype, int(pad), _stream(x)),
               "magnet_pack_features")
    return out


def pack_gmm(gmm_nchw: torch.Tensor, out: torch.Tensor | None = None):
    """(N,2,h,w) fp32 [mu,sigma] planes -> (N,h+2,w+2,2) interleaved with a zero border."""
    g = _dev(gmm_nchw, "gmm_nchw", torch.float32)
    N, two, h, w = g.shape
    if two != 2:
        raise MagnetError(f"gmm_nchw must be (N,2,h,w), got {tuple(g.shape)}")
    if out is None:
        out = torch.empty((N, h + 2, w + 2, 2), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _check(load().magnet_pack_gmm(g.data_ptr(), _dev(out, "out", torch.float32).data_ptr(), N, h, w, _stream(g)),
               "magnet_pack_gmm")
    return out


def pack_gmm_quad(gmm_nchw: torch.Tensor, out: torch.Tensor | None = None):
    """(N,2,h,w) fp32 [mu,sigma] planes -> (N,h+2,w+2,8): the zero-bordered map per quad origin in quad form
    (MagnetCostVolumeArgs.src_gmm_quad; the production matcher's bilinear (mu, sigma) samples are 3 fma each)."""
    g = _dev(gmm_nchw, "gmm_nchw", torch.float32)
    N, two, h, w = g.shape
    if two != 2:
        raise MagnetError(f"gmm_nchw must be (N,2,h,w), got {tuple(g.shape)}")
    if out is None:
        out = torch.empty((N, h + 2, w + 2, 8), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _check(load().magnet_pack_gmm_quad(g.data_ptr(), _dev(out, "out", torch.float32).data_ptr(), N, h, w, _stream(g)),
               "magnet_pack_gmm_quad")
    return out


def cost_volume_cw(ref_feat_cl, src_feat_pad, src_gmm_pad, poses, is_valid, intM, rays, kappa,
                   ref_gmm=None, k_list=None, d_volume=None, out=None, path: int = 0, stats=None, out_split=None,
                   mode: int = 0, gate_bits=None, ray_params=None, src_gmm_quad=None, dev_flags: int = 0):
    """Launch the fused matching kernel.  All tensors on one GPU; see MagnetCostVolumeArgs.

    ref_feat_cl (B,h,w,F) from pack_features(pad=0); src_feat_pad (V*B,h+2,w+2,F) from
    pack_features(pad=1) (fp32 or bf16, same dtype); src_gmm_pad (V*B,h+2,w+2,2) from pack_gmm;
    poses (B,V,4,4), is_valid (B,V) int32, intM (B,3,3), rays (B,3,h*w).
    Either d_volume (B,D,h,w) or (ref_gmm (B,2,h,w), k_list: sequence of D python floats)."""
    r = _dev(ref_feat_cl, "ref_feat_cl")
    s = _dev(src_feat_pad, "src_feat_pad", r.dtype)
    fe = feat_enum(r.dtype)
    B, h, w, F = r.shape
    if s.shape[0] % B != 0 or tuple(s.shape[1:]) != (h + 2, w + 2, F):
        raise MagnetError(f"src_feat_pad shape {tuple(s.shape)} does not match ref_feat_cl {tuple(r.shape)} "
                          "(expected (V*B, h+2, w+2, F))")
    V = s.shape[0] // B
    a = MagnetCostVolumeArgs()
    a.B, a.V, a.F, a.h, a.w = B, V, F, h, w
    a.kappa = float(kappa)
    a.feat_dtype = fe
    a.ref_feat_cl, a.src_feat_pad = r.data_ptr(), s.data_ptr()
    a.mode = int(mode)
    if src_gmm_pad is None:
        if mode != 1 and src_gmm_quad is None:
            raise MagnetError("need src_gmm_pad (pack_gmm) or src_gmm_quad (pack_gmm_quad)")
        g = s                                                      # est_costvolume_F mode has no (mu,sigma) maps; or only the quad form is given
    else:
        g = _dev(src_gmm_pad, "src_gmm_pad", torch.float32)
        if tuple(g.shape) != (V * B, h + 2, w + 2, 2):
            raise MagnetError(f"src_gmm_pad shape {tuple(g.shape)}, expected {(V * B, h + 2, w + 2, 2)}")
        a.src_gmm_pad = g.data_ptr()
    keep = [r, s, g]
    kbuf = None
    if d_volume is not None:
        dv = _dev(d_volume, "d_volume", torch.float32)
        if dv.dim() != 4 or dv.shape[0] != B or tuple(dv.shape[2:]) != (h, w):
            raise MagnetError(f"d_volume shape {tuple(dv.shape)}, expected (B,D,h,w) = ({B},D,{h},{w})")
        D = dv.shape[1]
        a.d_volume = dv.data_ptr(); keep.append(dv)
    else:
        if k_list is None or (ref_gmm is None and mode != 1):
            raise MagnetError("need d_volume, or ref_gmm and k_list (mode 1: k_list = depth bins)")
        D = len(k_list)
        kbuf = (ctypes.c_double * D)(*[float(k) for k in k_list])
        a.k_list = ctypes.addressof(kbuf)
        if mode != 1:
            rg = _dev(ref_gmm, "ref_gmm", torch.float32)
            if tuple(rg.shape) != (B, 2, h, w):
                raise MagnetError(f"ref_gmm shape {tuple(rg.shape)}, expected {(B, 2, h, w)}")
            a.ref_gmm = rg.data_ptr(); keep.append(rg)
    a.D = D
    po = _dev(poses, "poses", torch.float32); iv = _dev(is_valid, "is_valid", torch.int32)
    K = _dev(intM, "intM", torch.float32)
    if rays is None and ray_params is None:
        raise MagnetError("need rays (B,3,h*w) or ray_params (B,8) float64")
    if rays is not None:
        ry = _dev(rays, "rays", torch.float32)
        if tuple(ry.shape) != (B, 3, h * w):
            raise MagnetError(f"rays shape {tuple(ry.shape)}, expected {(B, 3, h * w)}")
        a.rays = ry.data_ptr()
    else:
        ry = _dev(ray_params, "ray_params", torch.float64)                 # the kernel generates the rays (N4)
        if tuple(ry.shape) != (B, 8):
            raise MagnetError(f"ray_params shape {tuple(ry.shape)}, expected {(B, 8)}")
        a.ray_params = ry.data_ptr()
    if tuple(po.shape) != (B, V, 4, 4) or tuple(iv.shape) != (B, V) or tuple(K.shape) != (B, 3, 3):
        raise MagnetError("poses/is_valid/intM shape mismatch: "
                          f"{tuple(po.shape)} {tuple(iv.shape)} {tuple(K.shape)}")
    a.poses, a.is_valid, a.intM = po.data_ptr(), iv.data_ptr(), K.data_ptr()
    if out_split is not None:
        # (hi, lo, ld): split-bf16 planes of the conv kernel's zero-bordered channel-last buffer, written in place
        oh, ol, ld = out_split
        for t in (oh, ol):
            if not t.is_cuda or t.dtype != torch.bfloat16:
                raise MagnetError("out_split planes must be bf16 GPU tensors")
        a.cost_hi, a.cost_lo, a.cost_ld = oh.data_ptr(), ol.data_ptr(), int(ld)
    elif out is None:
        out = torch.empty((B, D, h, w), dtype=torch.float32, device=r.device)
    else:
        # `out` may be the leading-D-channel slice of a larger (B, D+C, h, w) buffer (G-Net's input)
        if not out.is_cuda or out.dtype != torch.float32 or tuple(out.shape) != (B, D, h, w):
            raise MagnetError(f"out must be a float32 GPU tensor of shape {(B, D, h, w)}")
        if out.stride()[1:] != (h * w, w, 1) or (B > 1 and out.stride(0) < D * h * w):
            raise MagnetError(f"out strides {out.stride()} unsupported (need dense (D,h,w) frames)")
        a.cost_batch_stride = out.stride(0) if B > 1 else 0
    if out_split is None:
        a.cost = out.data_ptr()
    # `path` = 0..4; for the dev tools' convenience bits 8.. of the python argument are forwarded as dev_flags (ignored by the
    # product build of the library)
    a.path = int(path) & 0xff
    a.dev_flags = (int(path) >> 8) | int(dev_flags)
    if src_gmm_quad is not None:
        gq = _dev(src_gmm_quad, "src_gmm_quad", torch.float32)
        if tuple(gq.shape) != (V * B, h + 2, w + 2, 8):
            raise MagnetError(f"src_gmm_quad shape {tuple(gq.shape)}, expected {(V * B, h + 2, w + 2, 8)}")
        a.src_gmm_quad = gq.data_ptr(); keep.append(gq)
    if stats is not None:
        a.stats = _dev(stats, "stats").data_ptr()
    if gate_bits is not None:
        # debug output: (B,V,D,h,w) uint8 consistency-gate bits (zero it first: invalid views are not written)
        if not gate_bits.is_cuda or gate_bits.dtype != torch.uint8 or tuple(gate_bits.shape) != (B, V, D, h, w) \
                or not gate_bits.is_contiguous():
            raise MagnetError(f"gate_bits must be a contiguous uint8 GPU tensor of shape {(B, V, D, h, w)}")
        a.gate_bits = gate_bits.data_ptr()
    for t in (s, g, po, iv, K, ry):
        if t.device != r.device:
            raise MagnetError("all tensors must be on the same device")
    with torch.cuda.device(r.device):
        _check(load().magnet_cost_volume_cw(ctypes.byref(a), _stream(r)), "magnet_cost_volume_cw")
    return out


def cost_volume_f_backward(ref_feat_cl, src_feat_pad, poses, is_valid, intM, rays, d_center, grad_cost, path: int = 0,
                           stats=None):
    """Gradients of the mode-1 volume (est_costvolume_F before its softmax) w.r.t. the two feature maps.

    Same tensors as the forward call (fp32 features) + grad_cost (B,D,h,w).  Returns
    (grad_ref_cl (B,h,w,F), grad_src_pad (V*B,h+2,w+2,F)), both fp32 channel-last."""
    r = _dev(ref_feat_cl, "ref_feat_cl", torch.float32)
    s = _dev(src_feat_pad, "src_feat_pad", torch.float32)
    B, h, w, F = r.shape
    if s.shape[0] % B != 0 or tuple(s.shape[1:]) != (h + 2, w + 2, F):
        raise MagnetError(f"src_feat_pad shape {tuple(s.shape)} does not match ref_feat_cl {tuple(r.shape)}")
    V = s.shape[0] // B
    D = len(d_center)
    g = _dev(grad_cost, "grad_cost", torch.float32)
    if tuple(g.shape) != (B, D, h, w):
        raise MagnetError(f"grad_cost shape {tuple(g.shape)}, expected {(B, D, h, w)}")
    po = _dev(poses, "poses", torch.float32); iv = _dev(is_valid, "is_valid", torch.int32)
    K = _dev(intM, "intM", torch.float32); ry = _dev(rays, "rays", torch.float32)
    if tuple(po.shape) != (B, V, 4, 4) or tuple(iv.shape) != (B, V) or tuple(K.shape) != (B, 3, 3) \
            or tuple(ry.shape) != (B, 3, h * w):
        raise MagnetError("poses/is_valid/intM/rays shape mismatch")
    a = MagnetCostVolumeArgs()
    a.B, a.V, a.F, a.D, a.h, a.w = B, V, F, D, h, w
    a.feat_dtype = FEAT_F32
    a.mode = 1
    a.path = int(path)                                             # dev: 0x2000 = the per-item atomic kernel
    if stats is not None:
        a.stats = _dev(stats, "stats", torch.int32).data_ptr()       # [_, flushed texels, units merged in LDS, units sent to global atomics]
    a.ref_feat_cl, a.src_feat_pad = r.data_ptr(), s.data_ptr()
    kbuf = (ctypes.c_double * D)(*[float(k) for k in d_center])
    a.k_list = ctypes.addressof(kbuf)
    a.poses, a.is_valid, a.intM, a.rays = po.data_ptr(), iv.data_ptr(), K.data_ptr(), ry.data_ptr()
    grad_ref = torch.empty_like(r)
    grad_src = torch.zeros_like(s)
    l = load()
    with torch.cuda.device(r.device):
        if (int(path) & 0xff) == 0:
            # gather path: deterministic, no atomics; workspace = projection terms + per-(view, bin, tile) bounding boxes
            l.magnet_cost_volume_f_backward_workspace.restype = ctypes.c_int64
            l.magnet_cost_volume_f_backward_workspace.argtypes = [ctypes.c_void_p]
            l.magnet_cost_volume_f_backward_ws.restype = ctypes.c_int
            l.magnet_cost_volume_f_backward_ws.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_void_p]
            nbytes = int(l.magnet_cost_volume_f_backward_workspace(ctypes.byref(a)))
            if nbytes < 0:
                raise MagnetError("magnet_cost_volume_f_backward_workspace: " + l.magnet_last_error().decode())
            ws = torch.empty(nbytes, dtype=torch.uint8, device=r.device)
            _check(l.magnet_cost_volume_f_backward_ws(ctypes.byref(a), g.data_ptr(), grad_ref.data_ptr(), grad_src.data_ptr(),
                                                      ws.data_ptr(), nbytes, _stream(r)), "magnet_cost_volume_f_backward_ws")
        else:
            _check(l.magnet_cost_volume_f_backward(ctypes.byref(a), g.data_ptr(), grad_ref.data_ptr(),
                                                   grad_src.data_ptr(), _stream(r)), "magnet_cost_volume_f_backward")
    return grad_ref, grad_src


def gaussian_update(gnet_out, gmm_in, out=None):
    """(B,2,h,w) G-Net output + previous [mu,sigma] -> new [mu,sigma] (MAGNET.py:60-69)."""
    o = _dev(gnet_out, "gnet_out", torch.float32); g = _dev(gmm_in, "gmm_in", torch.float32)
    if o.shape != g.shape or o.dim() != 4 or o.shape[1] != 2:
        raise MagnetError(f"gaussian_update: shapes {tuple(o.shape)} / {tuple(g.shape)}, expected (B,2,h,w)")
    if out is None:
        out = torch.empty_like(g)
    B, _, h, w = g.shape
    with torch.cuda.device(g.device):
        _check(load().magnet_gaussian_update(o.data_ptr(), g.data_ptr(), _dev(out, "out", torch.float32).data_ptr(),
                                             B, h * w, _stream(g)), "magnet_gaussian_update")
    return out


def upsample_depth(depth, up_mask, k: int, out=None):
    """Learned convex upsampling (MAGNET.py:15-27): (B,C,h,w),(B,9*k*k,h,w) -> (B,C,k*h,k*w)."""
    d = _dev(depth, "depth", torch.float32); m = _dev(up_mask, "up_mask", torch.float32)
    B, C, h, w = d.shape
    if tuple(m.shape) != (B, 9 * k * k, h, w):
        raise MagnetError(f"up_mask shape {tuple(m.shape)}, expected {(B, 9 * k * k, h, w)}")
    if out is None:
        out = torch.empty((B, C, k * h, k * w), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _check(load().magnet_upsample_depth(d.data_ptr(), m.data_ptr(), _dev(out, "out", torch.float32).data_ptr(),
                                            B, C, h, w, k, _stream(d)), "magnet_upsample_depth")
    return out


# ---------------------------------------------------------------------------------------------------
# G-Net / mask-head convolutions on the matrix cores (include/magnet_hip.h: magnet_conv_mfma & friends)
# ---------------------------------------------------------------------------------------------------
class MagnetConvArgs(ctypes.Structure):
    """Mirror of `struct MagnetConvArgs` (include/magnet_hip.h)."""
    _fields_ = [
        ("in_hi", ctypes.c_void_p), ("in_lo", ctypes.c_void_p), ("w_hi", ctypes.c_void_p), ("w_lo", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("out_hi", ctypes.c_void_p), ("out_lo", ctypes.c_void_p), ("out_f32", ctypes.c_void_p),
        ("rows", ctypes.c_int64),
        ("cin", ctypes.c_int32), ("cout_pad", ctypes.c_int32), ("taps", ctypes.c_int32), ("wp", ctypes.c_int32),
        ("relu", ctypes.c_int32), ("out_mode", ctypes.c_int32), ("in_ld", ctypes.c_int32),
        ("addend", ctypes.c_void_p), ("addend_ld", ctypes.c_int32),
        ("dil", ctypes.c_int32), ("out_ld", ctypes.c_int32),
        ("add_hi", ctypes.c_void_p), ("add_lo", ctypes.c_void_p), ("add_ld", ctypes.c_int32),
        ("border_hp", ctypes.c_int32), ("border_pad", ctypes.c_int32), ("repad", ctypes.c_int32),
        ("tail_w_hi", ctypes.c_void_p), ("tail_w_lo", ctypes.c_void_p), ("tail_bias", ctypes.c_void_p),
        ("tail_cout_pad", ctypes.c_int32),
        ("up_depth", ctypes.c_void_p), ("up_out", ctypes.c_void_p),
        ("up_npred", ctypes.c_int32), ("up_B", ctypes.c_int32), ("up_h", ctypes.c_int32), ("up_w", ctypes.c_int32),
        ("gu_in", ctypes.c_void_p), ("gu_out", ctypes.c_void_p),
        ("in_sc", ctypes.c_void_p), ("w_sc", ctypes.c_void_p), ("sc_rows", ctypes.c_int64),
    ]


API_SYMBOLS = API_SYMBOLS + ("magnet_conv_mfma", "magnet_pack_split", "magnet_pack_mx", "magnet_gaussian_update_cl",
                             "magnet_upsample_depth_cl", "magnet_upsample_depth_cl_n")


def _conv_protos(lib):
    if getattr(lib, "_conv_protos_done", False):
        return lib
    I, P = ctypes.c_int32, ctypes.c_void_p
    lib.magnet_conv_mfma.restype = ctypes.c_int
    lib.magnet_conv_mfma.argtypes = [ctypes.POINTER(MagnetConvArgs), P]
    lib.magnet_pack_split.restype = ctypes.c_int
    lib.magnet_pack_split.argtypes = [P, P, P, I, I, I, I, I, I, ctypes.c_int64, P]
    lib.magnet_pack_mx.restype = ctypes.c_int
    lib.magnet_pack_mx.argtypes = [P, P, P, P, I, I, I, I, I, I, ctypes.c_int64, ctypes.c_int64, P]
    lib.magnet_gaussian_update_cl.restype = ctypes.c_int
    lib.magnet_gaussian_update_cl.argtypes = [P, I, P, P, I, I, I, P]
    lib.magnet_upsample_depth_cl.restype = ctypes.c_int
    lib.magnet_upsample_depth_cl.argtypes = [P, P, I, P, I, I, I, P]
    lib.magnet_upsample_depth_cl_n.restype = ctypes.c_int
    lib.magnet_upsample_depth_cl_n.argtypes = [P, P, I, P, I, I, I, I, P]
    lib._conv_protos_done = True
    return lib


def _bf16_ptr(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.bfloat16:
        raise MagnetError(f"{name} must be a bf16 GPU tensor")
    return t.data_ptr()


def conv_mfma(in_hi, in_lo, in_ld, cin, w_hi, w_lo, bias, taps, wp, relu, rows, out_hi=None, out_lo=None, out_f32=None,
              addend=None, dil=0, out_ld=0, add=None, border=None, repad=0, out_bf16=None, tail=None, upsample=None, gauss=None,
              mx=None):
    """One convolution layer on the matrix cores.  in_hi/in_lo: bf16 tensors whose data_ptr is row 0 (possibly a
    channel-offset view of a wider buffer, `in_ld` = its row pitch in elements); weights (taps, cout_pad, cin) bf16.
    F-Net extras (include/magnet_hip.h): dil (3x3 dilation), out_ld (write a channel slice: out tensors may then be
    views), add = (hi, lo, ld) split-bf16 residual input, border = (hp, pad) zero the border outputs, repad (re-address
    interior rows to a grid with border repad-1), out_bf16 = single bf16 output plane.
    tail = (w_hi, w_lo, bias, cout_pad): the stack's three 1x1 successors fused into this launch (result in out_f32).
    upsample = (depths (n,B,2,h,w) fp32, outs (n,B,2,4h,4w) fp32): with tail cout_pad 144, the learned convex upsampling runs in
    the tail's last layer (models/MAGNET.py:15-27) and only `outs` is written.
    gauss = (gmm_in (B,2,h,w), gmm_out): with tail cout_pad 16 (G-Net's head) the Gaussian update of models/MAGNET.py:60-69 runs
    in the tail's last layer and only `gmm_out` is written."""
    lib = _conv_protos(load())
    a = MagnetConvArgs()
    if mx is not None:
        # mx = (in_sc, w_sc, sc_rows): the fp16 + block-scaled e4m3 operand format (include/magnet_hip.h v302): in_hi / w_hi are fp16
        # planes, in_lo / w_lo the 2-byte-per-channel containers of the e4m3 hi | lo bytes (any 2-byte dtype), scales as int32 tensors
        for t, n in ((in_hi, "in_hi"), (w_hi, "w_hi")):
            if not t.is_cuda or t.dtype != torch.float16:
                raise MagnetError(f"conv_mfma (mx): {n} must be an fp16 GPU tensor")
        for t, n in ((in_lo, "in_lo"), (w_lo, "w_lo")):
            if not t.is_cuda or t.element_size() != 2:
                raise MagnetError(f"conv_mfma (mx): {n} must be a 2-byte GPU tensor (e4m3 hi | lo bytes per 32-channel block)")
        isc, wsc, sc_rows = mx
        if isc.dtype != torch.int32 or wsc.dtype != torch.int32 or not isc.is_cuda or not wsc.is_cuda:
            raise MagnetError("conv_mfma (mx): scale planes must be int32 GPU tensors")
        a.in_sc, a.w_sc, a.sc_rows = isc.data_ptr(), wsc.data_ptr(), int(sc_rows)
    else:
        for t, n in ((in_hi, "in_hi"), (in_lo, "in_lo"), (w_hi, "w_hi"), (w_lo, "w_lo")):
            if not t.is_cuda or t.dtype != torch.bfloat16:
                raise MagnetError(f"conv_mfma: {n} must be a bf16 GPU tensor")
    a.in_hi, a.in_lo, a.w_hi, a.w_lo = in_hi.data_ptr(), in_lo.data_ptr(), w_hi.data_ptr(), w_lo.data_ptr()
    a.bias = _dev(bias, "bias", torch.float32).data_ptr()
    a.rows, a.cin, a.cout_pad, a.taps, a.wp = int(rows), int(cin), int(w_hi.shape[1]), int(taps), int(wp)
    a.relu, a.in_ld = int(bool(relu)), int(in_ld)
    if addend is not None:
        a.addend, a.addend_ld = _dev(addend, "addend", torch.float32).data_ptr(), int(addend.shape[1])
    a.dil, a.out_ld, a.repad = int(dil), int(out_ld), int(repad)
    if add is not None:
        a.add_hi, a.add_lo, a.add_ld = _bf16_ptr(add[0], "add_hi"), _bf16_ptr(add[1], "add_lo"), int(add[2])
    if border is not None:
        a.border_hp, a.border_pad = int(border[0]), int(border[1])
    if tail is not None:
        a.tail_w_hi, a.tail_w_lo = _bf16_ptr(tail[0], "tail w_hi"), _bf16_ptr(tail[1], "tail w_lo")
        a.tail_bias, a.tail_cout_pad = _dev(tail[2], "tail bias", torch.float32).data_ptr(), int(tail[3])
    if gauss is not None:                                 # (gmm_in (B,2,h,w), gmm_out): Gaussian update behind G-Net's 16-channel fused tail
        gi, go = gauss
        if tail is None or gi.dim() != 4 or gi.shape[1] != 2 or go.shape != gi.shape:
            raise MagnetError("conv_mfma: gauss = (gmm_in (B,2,h,w), gmm_out (B,2,h,w)) with a fused tail")
        a.gu_in, a.gu_out = _dev(gi, "gmm_in", torch.float32).data_ptr(), _dev(go, "gmm_out", torch.float32).data_ptr()
        a.up_B, a.up_h, a.up_w = int(gi.shape[0]), int(gi.shape[2]), int(gi.shape[3])
        a.out_mode = 1
    elif upsample is not None:
        d, o = upsample
        if tail is None or d.dim() != 5 or d.shape[2] != 2 or tuple(o.shape) != (d.shape[0], d.shape[1], 2, 4 * d.shape[3], 4 * d.shape[4]):
            raise MagnetError("conv_mfma: upsample = (depths (n,B,2,h,w), outs (n,B,2,4h,4w)) with a fused tail")
        a.up_depth, a.up_out = _dev(d, "upsample depths", torch.float32).data_ptr(), _dev(o, "upsample outs", torch.float32).data_ptr()
        a.up_npred, a.up_B, a.up_h, a.up_w = int(d.shape[0]), int(d.shape[1]), int(d.shape[3]), int(d.shape[4])
        a.out_mode = 1
    elif out_f32 is not None:
        if not out_f32.is_cuda or out_f32.dtype != torch.float32:
            raise MagnetError("conv_mfma: out_f32 must be a float32 GPU tensor")
        a.out_mode, a.out_f32 = 1, out_f32.data_ptr()
    elif out_bf16 is not None:
        a.out_mode, a.out_hi = 2, _bf16_ptr(out_bf16, "out_bf16")
    else:
        a.out_mode, a.out_hi, a.out_lo = 0, _bf16_ptr(out_hi, "out_hi"), _bf16_ptr(out_lo, "out_lo")
    with torch.cuda.device(in_hi.device):
        _check(lib.magnet_conv_mfma(ctypes.byref(a), _stream(in_hi)), "magnet_conv_mfma")


def pack_split(x_nchw, out_hi, out_lo, ctot, c_off):
    """fp32 (N,C,h,w) (dense, or a leading-channel slice of a wider NCHW tensor) -> interior of the split-bf16
    padded channel-last buffer (N,h+2,w+2,ctot), channels [c_off, c_off+C)."""
    lib = _conv_protos(load())
    if not x_nchw.is_cuda or x_nchw.dtype != torch.float32:
        raise MagnetError("pack_split: input must be a float32 GPU tensor")
    N, C, h, w = x_nchw.shape
    if x_nchw.stride()[1:] != (h * w, w, 1):
        raise MagnetError(f"pack_split: unsupported input strides {x_nchw.stride()}")
    with torch.cuda.device(x_nchw.device):
        _check(lib.magnet_pack_split(x_nchw.data_ptr(), out_hi.data_ptr(), out_lo.data_ptr(), N, C, h, w, int(ctot),
                                     int(c_off), int(x_nchw.stride(0)) if N > 1 else 0, _stream(x_nchw)), "magnet_pack_split")


def pack_mx(x_nchw, out_f16, out_qr, out_sc, ctot, c_off, sc_rows):
    """fp32 (N,C,h,w) -> channels [c_off, c_off+C) of the interior of a padded channel-last buffer (N,h+2,w+2,ctot) in the fp16 + e4m3
    operand format of conv_mfma(mx=...): out_f16 fp16 plane, out_qr 2-byte container plane (hi | lo e4m3 bytes per 32-channel block),
    out_sc int32 [ctot / 32][sc_rows] E8M0 pairs."""
    lib = _conv_protos(load())
    if not x_nchw.is_cuda or x_nchw.dtype != torch.float32:
        raise MagnetError("pack_mx: input must be a float32 GPU tensor")
    N, C, h, w = x_nchw.shape
    if x_nchw.stride()[1:] != (h * w, w, 1):
        raise MagnetError(f"pack_mx: unsupported input strides {x_nchw.stride()}")
    if out_f16.dtype != torch.float16 or out_qr.element_size() != 2 or out_sc.dtype != torch.int32:
        raise MagnetError("pack_mx: out_f16 fp16, out_qr a 2-byte dtype, out_sc int32")
    rows = N * (h + 2) * (w + 2)
    for name, t, need in (("out_f16", out_f16, rows * int(ctot)), ("out_qr", out_qr, rows * int(ctot)), ("out_sc", out_sc, (int(ctot) // 32) * int(sc_rows))):
        if not t.is_cuda or t.device != x_nchw.device or not t.is_contiguous() or t.numel() < need:
            raise MagnetError(f"pack_mx: {name} must be a contiguous tensor on {x_nchw.device} with at least {need} elements")
    if int(ctot) % 32 or int(c_off) % 32 or int(c_off) + C > int(ctot) or int(sc_rows) < rows:
        raise MagnetError("pack_mx: ctot / c_off must be multiples of 32 with c_off + C <= ctot, and sc_rows >= N (h+2) (w+2)")
    with torch.cuda.device(x_nchw.device):
        _check(lib.magnet_pack_mx(x_nchw.data_ptr(), out_f16.data_ptr(), out_qr.data_ptr(), out_sc.data_ptr(), N, C, h, w, int(ctot),
                                  int(c_off), int(sc_rows), int(x_nchw.stride(0)) if N > 1 else 0, _stream(x_nchw)), "magnet_pack_mx")


def gaussian_update_cl(gnet_out_pad, ld, gmm_in, h, w, out=None):
    lib = _conv_protos(load())
    g = _dev(gmm_in, "gmm_in", torch.float32)
    if out is None:
        out = torch.empty_like(g)
    with torch.cuda.device(g.device):
        _check(lib.magnet_gaussian_update_cl(_dev(gnet_out_pad, "gnet_out_pad", torch.float32).data_ptr(), int(ld),
                                             g.data_ptr(), out.data_ptr(), g.shape[0], h, w, _stream(g)),
               "magnet_gaussian_update_cl")
    return out


def upsample_depth_cl(depth, mask_pad, ld, out=None):
    lib = _conv_protos(load())
    d = _dev(depth, "depth", torch.float32)
    B, C, h, w = d.shape
    if C != 2:
        raise MagnetError("upsample_depth_cl: depth must be (B,2,h,w)")
    if out is None:
        out = torch.empty((B, 2, 4 * h, 4 * w), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _check(lib.magnet_upsample_depth_cl(d.data_ptr(), _dev(mask_pad, "mask_pad", torch.float32).data_ptr(), int(ld),
                                            out.data_ptr(), B, h, w, _stream(d)), "magnet_upsample_depth_cl")
    return out


def upsample_depth_cl_n(depths, mask_pad, ld):
    """Every prediction of the refinement loop upsampled with the same mask in ONE launch (models/MAGNET.py:173): `depths` is a
    list of (B,2,h,w) tensors; returns the list of (B,2,4h,4w) outputs (contiguous slices of one buffer)."""
    lib = _conv_protos(load())
    if len(depths) == 1:
        return [upsample_depth_cl(depths[0], mask_pad, ld)]
    d = torch.stack([_dev(x, "depth", torch.float32) for x in depths])
    n, B, C, h, w = d.shape
    if C != 2:
        raise MagnetError("upsample_depth_cl_n: depths must be (B,2,h,w)")
    out = torch.empty((n, B, 2, 4 * h, 4 * w), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _check(lib.magnet_upsample_depth_cl_n(d.data_ptr(), _dev(mask_pad, "mask_pad", torch.float32).data_ptr(), int(ld),
                                              out.data_ptr(), n, B, h, w, _stream(d)), "magnet_upsample_depth_cl_n")
    return [out[i] for i in range(n)]


API_SYMBOLS = API_SYMBOLS + ("magnet_conv1x1_chain",)


def conv1x1_chain(in_hi, in_lo, w_hi, w_lo, bias, out, rows, cout_pad):
    """relu(1x1 128->128), relu(1x1 128->128), 1x1 128->cout_pad in one launch (see include/magnet_hip.h)."""
    lib = _conv_protos(load())
    if not getattr(lib, "_chain_proto", False):
        P = ctypes.c_void_p
        lib.magnet_conv1x1_chain.restype = ctypes.c_int
        lib.magnet_conv1x1_chain.argtypes = [P, P, P, P, P, P, ctypes.c_int64, ctypes.c_int32, P]
        lib._chain_proto = True
    for t, n in ((in_hi, "in_hi"), (in_lo, "in_lo"), (w_hi, "w_hi"), (w_lo, "w_lo")):
        if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise MagnetError(f"conv1x1_chain: {n} must be a contiguous bf16 GPU tensor")
    with torch.cuda.device(in_hi.device):
        _check(lib.magnet_conv1x1_chain(in_hi.data_ptr(), in_lo.data_ptr(), w_hi.data_ptr(), w_lo.data_ptr(),
                                        _dev(bias, "bias", torch.float32).data_ptr(),
                                        _dev(out, "out", torch.float32).data_ptr(), int(rows), int(cout_pad),
                                        _stream(in_hi)), "magnet_conv1x1_chain")


API_SYMBOLS = API_SYMBOLS + ("magnet_depth_metrics", "magnet_depth_metrics_crop", "magnet_make_rays", "magnet_relative_poses")


def make_rays(ray_params, h: int, w: int):
    """(B,8) float64 GPU {fx, fy, cx, cy, sx, sy, left, top} -> unit_ray_array_2D (B,3,h*w) fp32 on the device, bit-identical to
    the loaders' host table (dataloader_scannet.py:139-147, dataloader_kitti.py:113-118)."""
    lib = load()
    prm = _dev(ray_params, "ray_params", torch.float64)
    if prm.dim() != 2 or prm.shape[1] != 8:
        raise MagnetError(f"ray_params shape {tuple(prm.shape)}, expected (B, 8)")
    B = prm.shape[0]
    out = torch.empty((B, 3, h * w), dtype=torch.float32, device=prm.device)
    lib.magnet_make_rays.restype = ctypes.c_int
    lib.magnet_make_rays.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    with torch.cuda.device(prm.device):
        _check(lib.magnet_make_rays(prm.data_ptr(), out.data_ptr(), B, int(h), int(w), _stream(prm)), "magnet_make_rays")
    return out


def relative_poses(ext_ref, ext_nghbr):
    """utils.data_preprocess on the device (utils/utils.py:72-98): float64 GPU extrinsics ext_ref (B,4,4), ext_nghbr (B,V,4,4) ->
    (poses (B,V,4,4) fp32, is_valid (B,V) int32), both on the device, ready for the matcher."""
    lib = load()
    er = _dev(ext_ref, "ext_ref", torch.float64); en = _dev(ext_nghbr, "ext_nghbr", torch.float64)
    if er.dim() != 3 or tuple(er.shape[1:]) != (4, 4) or en.dim() != 4 or en.shape[0] != er.shape[0] or tuple(en.shape[2:]) != (4, 4):
        raise MagnetError(f"relative_poses: shapes {tuple(er.shape)} {tuple(en.shape)}, expected (B,4,4) and (B,V,4,4)")
    B, V = en.shape[:2]
    poses = torch.empty((B, V, 4, 4), dtype=torch.float32, device=er.device)
    valid = torch.empty((B, V), dtype=torch.int32, device=er.device)
    lib.magnet_relative_poses.restype = ctypes.c_int
    lib.magnet_relative_poses.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    with torch.cuda.device(er.device):
        _check(lib.magnet_relative_poses(er.data_ptr(), en.data_ptr(), poses.data_ptr(), valid.data_ptr(), B, V, _stream(er)),
               "magnet_relative_poses")
    return poses, valid


# ---- F-Net non-GEMM layers (row N3) -------------------------------------------------------------------------------
API_SYMBOLS = API_SYMBOLS + ("magnet_fnet_stem", "magnet_space_to_depth", "magnet_avgpool_cl", "magnet_upsample_bilinear_cl")


def _fnet_protos(lib):
    if getattr(lib, "_fnet_protos_done", False):
        return lib
    I, P = ctypes.c_int32, ctypes.c_void_p
    lib.magnet_fnet_stem.restype = ctypes.c_int
    lib.magnet_fnet_stem.argtypes = [P, P, P, P, P, I, I, I, P]
    lib.magnet_space_to_depth.restype = ctypes.c_int
    lib.magnet_space_to_depth.argtypes = [P, P, P, P, I, I, I, I, I, P]
    lib.magnet_avgpool_cl.restype = ctypes.c_int
    lib.magnet_avgpool_cl.argtypes = [P, P, I, I, I, I, I, I, I, P, P, P]
    lib.magnet_upsample_bilinear_cl.restype = ctypes.c_int
    lib.magnet_upsample_bilinear_cl.argtypes = [P, I, I, I, I, P, P, I, I, I, I, I, P]
    lib._fnet_protos_done = True
    return lib


def fnet_stem(img, wgt, bias, out_hi, out_lo):
    """(N,3,H,W) fp32 image -> 32-channel split planes (N,H2+2,W2+2,32): 3x3/s2 conv + folded BN + ReLU (F_psmnet.py:40)."""
    lib = _fnet_protos(load())
    x = _dev(img, "img", torch.float32)
    N, C, H, W = x.shape
    if C != 3:
        raise MagnetError(f"fnet_stem: expected 3 input channels, got {C}")
    with torch.cuda.device(x.device):
        _check(lib.magnet_fnet_stem(x.data_ptr(), _dev(wgt, "wgt", torch.float32).data_ptr(),
                                    _dev(bias, "bias", torch.float32).data_ptr(), _bf16_ptr(out_hi, "out_hi"),
                                    _bf16_ptr(out_lo, "out_lo"), N, H, W, _stream(x)), "magnet_fnet_stem")


def space_to_depth(in_hi, in_lo, out_hi, out_lo, N, C, H2, W2, opad):
    lib = _fnet_protos(load())
    with torch.cuda.device(in_hi.device):
        _check(lib.magnet_space_to_depth(_bf16_ptr(in_hi, "in_hi"), _bf16_ptr(in_lo, "in_lo"), _bf16_ptr(out_hi, "out_hi"),
                                         _bf16_ptr(out_lo, "out_lo"), N, C, H2, W2, opad, _stream(in_hi)), "magnet_space_to_depth")


def avgpool_cl(in_hi, in_lo, ld, N, h, w, pad, k, C, out_hi, out_lo):
    lib = _fnet_protos(load())
    with torch.cuda.device(in_hi.device):
        _check(lib.magnet_avgpool_cl(_bf16_ptr(in_hi, "in_hi"), _bf16_ptr(in_lo, "in_lo"), ld, N, h, w, pad, k, C,
                                     _bf16_ptr(out_hi, "out_hi"), _bf16_ptr(out_lo, "out_lo"), _stream(in_hi)), "magnet_avgpool_cl")


def upsample_bilinear_cl(x, in_ld, ph, pw, C, out_hi, out_lo, out_ld, N, h, w, pad):
    lib = _fnet_protos(load())
    with torch.cuda.device(x.device):
        _check(lib.magnet_upsample_bilinear_cl(_dev(x, "x", torch.float32).data_ptr(), in_ld, ph, pw, C, _bf16_ptr(out_hi, "out_hi"),
                                               _bf16_ptr(out_lo, "out_lo"), out_ld, N, h, w, pad, _stream(x)),
               "magnet_upsample_bilinear_cl")
