"""F-Net (row N3 of SURVEY.md §8f): the PSMNet feature extractor of the reference
(models/FNET.py:7-20, models/submodules/F_psmnet.py:37-124) — 5/6 of the matcher's input bytes come from it.

Two things live here:

* `PSMNet` / `FNET`: plain nn.Modules with the reference's architecture and **state_dict keys** (`firstconv.0.0.weight`,
  `layer2.7.conv1.0.1.running_var`, `branch3.1.0.weight`, `lastconv.2.weight`, ...), so the reference's F-Net
  checkpoints load unchanged.  This is the torch path (training / autograd, e.g. under `MAGNET_F`).
* `FNetMFMA`: the inference path on MI355X.  BatchNorm (eval) is folded into the convolutions; every convolution
  runs on the bf16x3 matrix-core kernel (`magnet_conv_mfma`, fp32-grade) over zero-bordered channel-last activations
  kept as split-bf16 planes; the stride-2 layers become stride-1 GEMMs through a space-to-depth rearrangement; residual
  adds, ReLU, border zeroing and the 320-channel concatenation happen in the convolution epilogues (channel-slice
  writes); the last 1x1 layer writes the features **directly in the matcher's layouts** (reference features
  (B,h,w,F), source features (V*B,h+2,w+2,F) zero-bordered, fp32 or bf16) — the NCHW fp32 feature tensor of the
  reference and the pack pass over it never exist.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib
from .convnet import split_bf16

# (name, planes, blocks, stride, dilation) — F_psmnet.py:44-47
_TRUNK = (("layer1", 32, 3, 1, 1), ("layer2", 64, 16, 2, 1), ("layer3", 128, 3, 1, 1), ("layer4", 128, 3, 1, 2))
# (name, pooling window) in the order the reference concatenates them LAST-to-first (F_psmnet.py:50-64,122)
_SPP = (("branch1", 64), ("branch2", 32), ("branch3", 16), ("branch4", 8))


def _conv_bn(cin, cout, k, stride=1, dilation=1):
    """Conv2d (no bias) + BatchNorm2d as a 2-element Sequential (keys `.0.weight`, `.1.*`), F_psmnet.py:10-16."""
    pad = dilation if k == 3 else 0
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dilation, bias=False),
                         nn.BatchNorm2d(cout))


class ResidualUnit(nn.Module):
    """conv3x3-BN-ReLU, conv3x3-BN, plus the (optionally projected) input; no ReLU after the sum (F_psmnet.py:19-34)."""

    def __init__(self, cin, cout, stride, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(_conv_bn(cin, cout, 3, stride, dilation), nn.ReLU(inplace=True))
        self.conv2 = _conv_bn(cout, cout, 3, 1, dilation)
        self.downsample = _conv_bn(cin, cout, 1, stride) if (stride != 1 or cin != cout) else None

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + (x if self.downsample is None else self.downsample(x))


class PSMNet(nn.Module):
    def __init__(self, feature_dim=32):
        super().__init__()
        self.firstconv = nn.Sequential(_conv_bn(3, 32, 3, 2), nn.ReLU(inplace=True),
                                       _conv_bn(32, 32, 3), nn.ReLU(inplace=True),
                                       _conv_bn(32, 32, 3), nn.ReLU(inplace=True))
        width = 32
        for name, planes, blocks, stride, dilation in _TRUNK:
            units = [ResidualUnit(width if i == 0 else planes, planes, stride if i == 0 else 1, dilation) for i in range(blocks)]
            setattr(self, name, nn.Sequential(*units))
            width = planes
        for name, k in _SPP:
            setattr(self, name, nn.Sequential(nn.AvgPool2d((k, k), stride=(k, k)), _conv_bn(128, 32, 1), nn.ReLU(inplace=True)))
        self.lastconv = nn.Sequential(_conv_bn(320, 128, 3), nn.ReLU(inplace=True),
                                      nn.Conv2d(128, feature_dim, kernel_size=1, bias=False))
        for m in self.modules():                                    # F_psmnet.py:71-76: N(0, sqrt(2 / (k*k*cout)))
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))

    def forward(self, x):
        x = self.layer1(self.firstconv(x))
        raw = self.layer2(x)                                        # H/4, 64 ch
        skip = self.layer4(self.layer3(raw))                        # H/4, 128 ch
        size = skip.shape[2:]
        pyramid = [F.interpolate(getattr(self, name)(skip), size=size, mode="bilinear", align_corners=True)
                   for name, _ in _SPP]                             # branch1..branch4
        feat = torch.cat([raw, skip] + pyramid[::-1], dim=1)        # raw, skip, branch4, branch3, branch2, branch1 (:122)
        return self.lastconv(feat)


class FNET(nn.Module):
    """models/FNET.py:7-20: `args.FNET_architecture == 'PSM-Net'`, `args.FNET_feature_dim` output channels."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        if getattr(args, "FNET_architecture", "PSM-Net") != "PSM-Net":
            raise lib.MagnetError(f"unknown FNET_architecture {args.FNET_architecture!r}")
        self.f_net = PSMNet(feature_dim=args.FNET_feature_dim)

    def forward(self, img):
        return self.f_net(img)


# ======================================================================================================
# inference on the matrix cores
# ======================================================================================================
def _fold(seq: nn.Sequential):
    """(Conv2d, BatchNorm2d) in eval mode -> fp32 weight (cout,cin,kh,kw), bias (cout), folded in fp64."""
    conv, bn = seq[0], seq[1]
    w = conv.weight.detach().double()
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    return (w * scale.view(-1, 1, 1, 1)).float(), (bn.bias.detach().double() - bn.running_mean.detach().double() * scale).float()


def _pack_taps(w: torch.Tensor):
    """(cout, cin, kh, kw) fp32 -> split bf16 planes (kh*kw, cout, cin), cin contiguous."""
    cout, cin, kh, kw = w.shape
    return split_bf16(w.permute(2, 3, 0, 1).reshape(kh * kw, cout, cin).contiguous())


def _pack_s2d(w: torch.Tensor):
    """3x3 stride-2 pad-1 weights (cout, C, 3, 3) -> the equivalent 2x2-window weights over a space-to-depth input:
    planes (4, cout, 4*C); tap (ty,tx) in {-1,0}^2 -> index (ty+1)*2+(tx+1); channel (py*2+px)*C + c."""
    cout, C = w.shape[:2]
    out = torch.zeros((4, cout, 4 * C), dtype=torch.float32, device=w.device)
    k_of = {(-1, 1): 0, (0, 0): 1, (0, 1): 2}                       # (tap offset, phase) -> kernel index; (-1, 0) has none
    for (ty, py), ky in k_of.items():
        for (tx, px), kx in k_of.items():
            ph = py * 2 + px
            out[(ty + 1) * 2 + (tx + 1), :, ph * C:(ph + 1) * C] = w[:, :, ky, kx]
    return split_bf16(out)


class FNetMFMA:
    """Inference runner for a `PSMNet` (or the reference's own PSMNet instance: same attribute structure)."""

    # every convolution launch appends (start_event, end_event, flops) when set to a list (tools/bench_fnet.py)
    event_sink = None

    # Batches of at least this many images run as TWO half-batches on two HIP streams (round 5).  Every launch of this chain ends in
    # a burst of output stores that does not depend on K (20 - 35 % of the 32- / 64-channel layers' launches) while the matrix pipes
    # idle; a second, independent chain out of phase fills those gaps (profiles/r5/fnet_two_streams.log).  Per image the two
    # halves run the same kernels on the same data, so the result is bit-identical to the one-stream run (tests/test_gpu_fnet.py).
    split_min_images = 8
    split_parts = 2

    def __init__(self, psm: nn.Module):
        self.psm = psm
        self._packed = None
        self._key = None
        self._bufs = {}
        self._bufs_sig = None
        self._streams = {}

    # ---- weights ---------------------------------------------------------------------------------------------
    def _params_key(self, device):
        return tuple((t.data_ptr(), t._version) for t in list(self.psm.parameters()) + list(self.psm.buffers())) + (str(device),)

    @torch.no_grad()
    def packed(self, device):
        key = self._params_key(device)
        if self._packed is not None and self._key == key:
            return self._packed
        P = {}

        def put(name, w, b, s2d=False):
            w = w.to(device); b = b.to(device)
            hi, lo = (_pack_s2d if s2d else _pack_taps)(w)
            P[name] = (hi, lo, b.contiguous(), w.shape[0])

        psm = self.psm
        w0, b0 = _fold(psm.firstconv[0])
        P["stem"] = (w0.to(device).reshape(32, 27).contiguous(), b0.to(device).contiguous())
        put("firstconv.2", *_fold(psm.firstconv[2]))
        put("firstconv.4", *_fold(psm.firstconv[4]))
        for name, planes, blocks, stride, dilation in _TRUNK:
            layer = getattr(psm, name)
            for i in range(blocks):
                u = layer[i]
                put(f"{name}.{i}.conv1", *_fold(u.conv1[0]), s2d=(i == 0 and stride == 2))
                put(f"{name}.{i}.conv2", *_fold(u.conv2))
                if u.downsample is not None:
                    put(f"{name}.{i}.downsample", *_fold(u.downsample))
        for name, _ in _SPP:
            put(name, *_fold(getattr(psm, name)[1]))
        put("lastconv.0", *_fold(psm.lastconv[0]))
        wl = psm.lastconv[2].weight.detach().float()
        put("lastconv.2", wl, torch.zeros(wl.shape[0]))
        self.feature_dim = wl.shape[0]
        if self.feature_dim not in (16, 32, 64, 128):
            raise lib.MagnetError(f"FNetMFMA: feature_dim {self.feature_dim} unsupported (16, 32, 64, 128)")
        self._packed, self._key = P, key
        return P

    # ---- activations -----------------------------------------------------------------------------------------
    def _buffers(self, dev, N, H, W, slot=0, sig=None):
        sig = sig if sig is not None else (str(dev), N, H, W)
        if self._bufs_sig != sig:                                   # one batch shape at a time: the buffers are large
            self._bufs.clear()
            self._bufs_sig = sig
        key = (str(dev), N, H, W, slot)
        b = self._bufs.get(key)
        if b is not None:
            return b
        H2, W2 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        H4, W4 = (H2 - 1) // 2 + 1, (W2 - 1) // 2 + 1
        if H4 < 64 or W4 < 64:
            raise lib.MagnetError(f"FNetMFMA: input {H}x{W} too small for the 64x64 pooling branch (needs H/4, W/4 >= 64)")
        rows_a, rows_b = N * (H2 + 2) * (W2 + 2), N * (H4 + 4) * (W4 + 4)

        def planes(rows, c):
            return (torch.zeros((rows, c), dtype=torch.bfloat16, device=dev), torch.zeros((rows, c), dtype=torch.bfloat16, device=dev))

        b = {"dims": (H2, W2, H4, W4, rows_a, rows_b),
             "A": [planes(rows_a, 32) for _ in range(3)],
             "S": planes(rows_b, 128),
             "B": [planes(rows_b, 64) for _ in range(3)],
             "C": [planes(rows_b, 128) for _ in range(3)],
             "cat": planes(rows_b, 320),
             "pool": {k: (planes(N * (H4 // k) * (W4 // k), 128),
                          torch.empty((N * (H4 // k) * (W4 // k), 32), dtype=torch.float32, device=dev)) for _, k in _SPP}}
        self._bufs[key] = b
        return b

    def _conv(self, name, src, in_ld, cin, taps, wp, rows, relu, dst=None, out_ld=0, add=None, border=None, dil=0, **kw):
        hi, lo, bias, cout = self._packed[name]
        sink = FNetMFMA.event_sink
        if sink is not None:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        lib.conv_mfma(src[0], src[1], in_ld, cin, hi, lo, bias, taps, wp, relu, rows,
                      out_hi=None if dst is None else dst[0], out_lo=None if dst is None else dst[1],
                      out_ld=out_ld, add=add, border=border, dil=dil, **kw)
        if sink is not None:
            e1.record()
            sink.append((e0, e1, 2.0 * rows * cin * taps * cout))

    @torch.no_grad()
    def run(self, img: torch.Tensor, n_ref: int | None = None, feat_dtype="fp32"):
        """img (N,3,H,W) fp32 on the GPU.  n_ref None: returns the (N,F,H/4,W/4) fp32 NCHW feature tensor (drop-in for
        the module's forward).  n_ref = B: returns (ref_cl (B,h,w,F), src_pad (N-B,h+2,w+2,F)) in `feat_dtype`
        storage — the matcher's input layouts, written by the last layer's epilogue."""
        if not img.is_cuda:
            raise lib.MagnetError("FNetMFMA: image must be on the GPU (no CPU fallback)")
        if self.psm.training:
            raise lib.MagnetError("FNetMFMA folds BatchNorm running statistics: call .eval() on the F-Net first")
        img = img.detach().float().contiguous()
        N, _, H, W = img.shape
        if n_ref is not None and not (0 <= int(n_ref) <= N):
            raise lib.MagnetError(f"FNetMFMA: n_ref = {n_ref} outside [0, {N}] (the reference images lead the batch)")
        dev = img.device
        self.packed(dev)
        H4, W4 = ((H - 1) // 2 + 1 - 1) // 2 + 1, ((W - 1) // 2 + 1 - 1) // 2 + 1
        Fd = self.feature_dim
        sig = (str(dev), N, H, W)
        if self._bufs_sig != sig:
            self._bufs.clear()
            self._bufs_sig = sig
        # the outputs of the whole batch (the halves of a split run write their slices)
        if n_ref is None:
            outs = (torch.empty((N, H4, W4, Fd), dtype=torch.float32, device=dev), None)
            fe = None
        else:
            fe = lib.feat_enum(feat_dtype)
            dt = lib.feat_torch_dtype(fe)
            okey = ("out", n_ref, fe)
            outs = self._bufs.get(okey)
            if outs is None:
                outs = self._bufs[okey] = (torch.empty((n_ref, H4, W4, Fd), dtype=dt, device=dev),
                                           torch.zeros((N - n_ref, H4 + 2, W4 + 2, Fd), dtype=dt, device=dev))   # zero border, never rewritten
        # Two-stream split: k contiguous, balanced, NON-EMPTY parts; split_parts <= 1 (or a batch too small for two parts of >= 4
        # images) takes the one-stream path
        k = min(int(self.split_parts), N // 4) if N >= self.split_min_images else 1
        if k >= 2:
            cuts = [(N * i + k - 1) // k for i in range(k + 1)]       # the first parts take the remainder; N >= 4k: none is empty
            parts = [(cuts[i], cuts[i + 1]) for i in range(k)]
            main = torch.cuda.current_stream(dev)
            streams = self._streams.setdefault(str(dev), [])
            while len(streams) < k:
                streams.append(torch.cuda.Stream(device=dev))
            ready = torch.cuda.Event(); ready.record(main)
            # `img` and `outs` were allocated on `main`; the side streams use them: tell the caching allocator, so that neither block
            # can be recycled before the side streams are done even if an exception skips the joins below
            for st in streams[:k]:
                img.record_stream(st)
                for t in outs:
                    if t is not None:
                        t.record_stream(st)
            started = []
            try:
                for slot, ((lo, hi), st) in enumerate(zip(parts, streams)):
                    st.wait_event(ready)                             # the images (and the previous consumer of the outputs) are on `main`
                    started.append(st)
                    with torch.cuda.stream(st):
                        self._run_part(img[lo:hi], slot, sig, n_ref, fe, outs, lo)
            finally:
                for st in started:                                   # always join: `main` never runs ahead of a side stream's work
                    done = torch.cuda.Event(); done.record(st)
                    main.wait_event(done)
        else:
            self._run_part(img, 0, sig, n_ref, fe, outs, 0)
        if n_ref is None:
            return outs[0].permute(0, 3, 1, 2).contiguous()
        return outs

    def _run_part(self, img, slot, sig, n_ref, fe, outs, first):
        """Images [first, first + len(img)) of the batch through the whole chain on the current stream, with buffer set `slot`."""
        N, _, H, W = img.shape
        dev = img.device
        P = self._packed
        buf = self._buffers(dev, N, H, W, slot, sig)
        H2, W2, H4, W4, rows_a, rows_b = buf["dims"]
        A, S, Bb, C, cat = buf["A"], buf["S"], buf["B"], buf["C"], buf["cat"]
        wpa, wpb = W2 + 2, W4 + 4
        ba, bb = (H2 + 2, 1), (H4 + 4, 2)

        # ---- H/2 stage: firstconv + layer1 (32 channels, border 1) ----
        lib.fnet_stem(img, P["stem"][0], P["stem"][1], A[0][0], A[0][1])                          # F_psmnet.py:40
        self._conv("firstconv.2", A[0], 32, 32, 9, wpa, rows_a, True, dst=A[1], border=ba)
        self._conv("firstconv.4", A[1], 32, 32, 9, wpa, rows_a, True, dst=A[0], border=ba)
        x = 0
        for i in range(3):                                                                       # layer1
            t, o = (x + 1) % 3, (x + 2) % 3
            self._conv(f"layer1.{i}.conv1", A[x], 32, 32, 9, wpa, rows_a, True, dst=A[t], border=ba)
            self._conv(f"layer1.{i}.conv2", A[t], 32, 32, 9, wpa, rows_a, False, dst=A[o], add=(A[x][0], A[x][1], 32), border=ba)
            x = o
        # ---- H/4 stage (border 2: layer4 is dilated) ----
        lib.space_to_depth(A[x][0], A[x][1], S[0], S[1], N, 32, H2, W2, 2)
        self._conv("layer2.0.conv1", S, 128, 128, 4, wpb, rows_b, True, dst=Bb[0], border=bb)    # 3x3 stride 2
        self._conv("layer2.0.downsample", S, 128, 32, 1, wpb, rows_b, False, dst=Bb[1], border=bb)   # 1x1 stride 2 = phase 0
        self._conv("layer2.0.conv2", Bb[0], 64, 64, 9, wpb, rows_b, False, dst=Bb[2], add=(Bb[1][0], Bb[1][1], 64), border=bb)
        cur, ld = Bb[2], 64
        x = 2
        raw = (cat[0][:, 0:64], cat[1][:, 0:64])
        for i in range(1, 16):
            t, o = (x + 1) % 3, (x + 2) % 3
            self._conv(f"layer2.{i}.conv1", cur, ld, 64, 9, wpb, rows_b, True, dst=Bb[t], border=bb)
            last = i == 15                                                                       # output_raw -> concat[:, 0:64]
            self._conv(f"layer2.{i}.conv2", Bb[t], 64, 64, 9, wpb, rows_b, False, dst=raw if last else Bb[o],
                       out_ld=320 if last else 0, add=(cur[0], cur[1], ld), border=bb)
            cur, ld, x = (raw, 320, o) if last else (Bb[o], 64, o)
        skip = (cat[0][:, 64:192], cat[1][:, 64:192])
        x = 0
        units = [("layer3", i, 0) for i in range(3)] + [("layer4", i, 2) for i in range(3)]
        for n_unit, (name, i, dil) in enumerate(units):
            t, o = (x + 1) % 3, (x + 2) % 3
            cin = 64 if (name == "layer3" and i == 0) else 128
            self._conv(f"{name}.{i}.conv1", cur, ld, cin, 9, wpb, rows_b, True, dst=C[t], border=bb, dil=dil)
            if f"{name}.{i}.downsample" in P:
                self._conv(f"{name}.{i}.downsample", cur, ld, cin, 1, wpb, rows_b, False, dst=C[x], border=bb)
                res = (C[x][0], C[x][1], 128)
            else:
                res = (cur[0], cur[1], ld)
            last = n_unit == len(units) - 1                                                      # output_skip -> concat[:, 64:192]
            self._conv(f"{name}.{i}.conv2", C[t], 128, 128, 9, wpb, rows_b, False, dst=skip if last else C[o],
                       out_ld=320 if last else 0, add=res, border=bb, dil=dil)
            cur, ld, x = (skip, 320, o) if last else (C[o], 128, o)
        # ---- SPP branches: pool -> 1x1 conv + BN + ReLU -> bilinear back to H/4, into their concat slices ----
        for slot, (name, k) in enumerate(_SPP):                     # branch1 -> channels 288:320 ... branch4 -> 192:224
            (p_hi, p_lo), q = buf["pool"][k]
            ph, pw = H4 // k, W4 // k
            lib.avgpool_cl(skip[0], skip[1], 320, N, H4, W4, 2, k, 128, p_hi, p_lo)
            hi, lo, bias, _ = P[name]
            lib.conv_mfma(p_hi, p_lo, 128, 128, hi, lo, bias, 1, 1, True, N * ph * pw, out_f32=q)
            off = 288 - 32 * slot
            lib.upsample_bilinear_cl(q, 32, ph, pw, 32, cat[0][:, off:off + 32], cat[1][:, off:off + 32], 320, N, H4, W4, 2)
        # ---- lastconv ----
        self._conv("lastconv.0", cat, 320, 320, 9, wpb, rows_b, True, dst=C[0], border=bb)
        Fd = self.feature_dim
        img_rows = (H4 + 4) * wpb
        if n_ref is None:
            self._conv("lastconv.2", C[0], 128, 128, 1, wpb, rows_b, False, border=bb, repad=1, out_f32=outs[0][first:first + N], out_ld=Fd)
            return
        ref_cl, src_pad = outs
        okw = (lambda t: {"out_bf16": t}) if fe == lib.FEAT_BF16 else (lambda t: {"out_f32": t})
        nr = max(0, min(n_ref - first, N))                          # reference images of this part (they lead the batch)
        if nr > 0:
            self._conv("lastconv.2", C[0], 128, 128, 1, wpb, nr * img_rows, False, border=bb, repad=1, out_ld=Fd, **okw(ref_cl[first:first + nr]))
        if N > nr:
            tail = (C[0][0][nr * img_rows:], C[0][1][nr * img_rows:])
            s0 = first + nr - n_ref                                 # first source image of this part, counted among the source images
            self._conv("lastconv.2", tail, 128, 128, 1, wpb, (N - nr) * img_rows, False, border=bb, repad=2, out_ld=Fd,
                       **okw(src_pad[s0:s0 + (N - nr)]))
