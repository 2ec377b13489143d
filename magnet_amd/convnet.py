"""Runs the reference's small conv stacks (GNET.gnet, mask_head — models/MAGNET.py:51-56,111-116:
Conv3x3(pad 1)-ReLU-Conv1x1-ReLU-Conv1x1-ReLU-Conv1x1) on the bf16x3 MFMA kernel (csrc/conv_mfma.hip).

Activations are zero-bordered channel-last buffers (B, h+2, w+2, C) kept as two bf16 planes (hi, lo); the
weights of the nn.Conv2d modules are re-packed (and re-split) lazily whenever a parameter changes, so the
module keeps its reference state_dict and stays trainable through the torch path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import lib


def _round_up(x, m):
    return (x + m - 1) // m * m


def split_bf16(x: torch.Tensor):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def mx_quant(v: torch.Tensor):
    """Block-scaled OCP e4m3 (MX-fp8) of v (..., nblk, 32) fp32: one E8M0 exponent per block with max |v| / 2^e in [128, 256) (<= 448), e = -127
    for an all-zero block.  Returns (bytes uint8 (..., nblk, 32), e + 127 int32 (..., nblk)).  The rule of magnet_pack_mx (conv_mfma.hip)."""
    m = v.abs().amax(-1)
    _, ex = torch.frexp(m)
    e = (ex.to(torch.int32) - 8).clamp(min=-127)
    e = torch.where(m > 0, e, torch.full_like(e, -127))
    q = (v * torch.exp2(-e.float()).unsqueeze(-1)).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), e + 127


def split_mx(x: torch.Tensor):
    """x (..., K) fp32, K % 32 == 0 -> the fp16 + e4m3 operand format of magnet_conv_mfma v302: (hi (..., K) fp16, qr (..., K) int16 container of
    the [32 hi bytes | 32 lo bytes] per 32-element block, sc (..., K / 32) int32 = E8M0(hi) | E8M0(lo) << 8)."""
    hi = x.to(torch.float16)
    lo = x - hi.float()
    shp = x.shape[:-1] + (x.shape[-1] // 32, 32)
    q, eq = mx_quant(hi.float().reshape(shp))
    r, er = mx_quant(lo.reshape(shp))
    qr = torch.cat([q, r], dim=-1).contiguous().view(torch.int16).reshape(x.shape)
    return hi.contiguous(), qr, (eq | (er << 8)).to(torch.int32).contiguous()


def _cout_pad(c):
    if c <= 16:
        return 16
    if c == 144:
        return 144
    return _round_up(c, 128)


class ConvStackMFMA:
    """`seq`: nn.Sequential of Conv2d / ReLU as in the reference.  `in_map`: how the first layer's input channels are
    laid out in the (wider, 32-aligned) input buffer: list of (src_start, length, dst_start)."""

    # When set to a list, every layer launch appends (start_event, end_event, flops, taps) recorded on the launch
    # stream (bench.py's live timing of the convolution kernel).
    event_sink = None

    def __init__(self, seq: nn.Sequential, in_map=None):
        self.layers = []
        mods = list(seq)
        i = 0
        while i < len(mods):
            conv = mods[i]
            if not isinstance(conv, nn.Conv2d):
                raise lib.MagnetError(f"ConvStackMFMA: unexpected module {type(conv).__name__}")
            k = conv.kernel_size
            if k not in ((1, 1), (3, 3)) or conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1 or \
                    conv.padding != ((k[0] - 1) // 2, (k[1] - 1) // 2):
                raise lib.MagnetError("ConvStackMFMA supports 1x1 and 3x3 (pad 1) stride-1 convolutions")
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            self.layers.append((conv, relu))
            i += 2 if relu else 1
        self.in_map = in_map
        # Fused 1x1-chain kernel (magnet_conv1x1_chain): bit-identical results, but MEASURED SLOWER on MI355X
        # (1.41 + 1.04 ms vs 0.94 + 0.79 ms per 64-frame step for the two stacks): its 101 KB LDS tile allows one
        # workgroup per CU, so its load phase is not overlapped, whereas the separate launches are HBM-bound at
        # high occupancy.  Off by default; kept (and tested) as the starting point for a 64-row-tile variant.
        self.fuse_tail = False
        # The same three 1x1 layers fused into the EPILOGUE of the stack's first (3x3) layer: the 128 x 128 tile is already
        # on the CU, weights come straight from L2, LDS need stays at the K ring's 64 KB (2 workgroups per CU).
        self.fuse_epilogue = True
        self._chain = None
        self._packed = None
        self._key = None

    def can_fuse_gauss(self, device):
        """True when run(..., gauss=...) is available: the reference's G-Net (3x3 + three 1x1, 2 outputs) with the fused epilogue on."""
        self.packed(device)
        return self._chain is not None and self.fuse_epilogue and self._chain["cout_pad"] == 16

    def can_fuse_upsample(self, device):
        """True when run(..., upsample=...) is available: the reference's mask head (3x3 + three 1x1, 9 * 16 outputs) with the fused
        epilogue on."""
        self.packed(device)
        return self._chain is not None and self.fuse_epilogue and self._chain["cout_pad"] == 144

    def cin_pad(self):
        c0 = self.layers[0][0].in_channels
        if self.in_map is None:
            return _round_up(c0, 32)
        return _round_up(max(d + n for _, n, d in self.in_map), 32)

    def _params_key(self, device):
        return tuple((p.data_ptr(), p._version) for conv, _ in self.layers for p in conv.parameters()) + (str(device),)

    @torch.no_grad()
    def packed(self, device):
        key = self._params_key(device)
        if self._packed is not None and self._key == key:
            return self._packed
        out = []
        cin_p = self.cin_pad()
        for li, (conv, relu) in enumerate(self.layers):
            W = conv.weight.detach().to(device=device, dtype=torch.float32)          # (Cout, Cin, kh, kw)
            cout, cin, kh, kw = W.shape
            cp = _cout_pad(cout)
            Wp = torch.zeros((kh * kw, cp, cin_p), dtype=torch.float32, device=device)
            Wt = W.permute(2, 3, 0, 1).reshape(kh * kw, cout, cin)                   # (tap, Cout, Cin)
            if li == 0 and self.in_map is not None:
                for src, n, dst in self.in_map:
                    Wp[:, :cout, dst:dst + n] = Wt[:, :, src:src + n]
            else:
                Wp[:, :cout, :cin] = Wt
            bias = torch.zeros(cp, dtype=torch.float32, device=device)
            if conv.bias is not None:
                bias[:cout] = conv.bias.detach().to(device=device, dtype=torch.float32)
            hi, lo = split_bf16(Wp)
            out.append(dict(w_hi=hi, w_lo=lo, bias=bias, taps=kh * kw, cin=cin_p, cout=cout, cout_pad=cp, relu=relu))
            cin_p = cp                                                                 # next layer reads all padded channels
            if li + 1 < len(self.layers) and cp % 32 != 0:
                raise lib.MagnetError("hidden layer width must be a multiple of 32")
        # the reference's stacks end in 1x1(128->128)+ReLU, 1x1(128->128)+ReLU, 1x1(128->cout): fused into one launch
        self._chain = None
        if len(out) == 4 and all(o["taps"] == 1 for o in out[1:]) and out[1]["relu"] and out[2]["relu"] and \
                not out[3]["relu"] and out[0]["cout_pad"] == 128 and out[1]["cout_pad"] == 128 and \
                out[2]["cout_pad"] == 128 and out[3]["cout_pad"] in (16, 128, 144) and out[0]["relu"]:
            self._chain = dict(
                w_hi=torch.cat([o["w_hi"].reshape(-1) for o in out[1:]]).contiguous(),
                w_lo=torch.cat([o["w_lo"].reshape(-1) for o in out[1:]]).contiguous(),
                bias=torch.cat([o["bias"] for o in out[1:]]).contiguous(), cout_pad=out[3]["cout_pad"])
        self._packed, self._key = out, key
        return out

    @torch.no_grad()
    def packed_mx(self, device):
        """First layer's weights in the fp16 + e4m3 operand format (conv_mfma.hip, WIN == 4 loop): (w_f16 (taps, cout_pad, cin), w_qr
        (taps, cout_pad, cin) int16 container, w_sc (taps, cin / 32, cout_pad) int32)."""
        pk = self.packed(device)[0]
        key = ("mx", self._key)
        if getattr(self, "_mx_key", None) != key:
            w = (pk["w_hi"].float() + pk["w_lo"].float())                   # the 16-bit-mantissa weights the bf16x3 path multiplies by
            hi, qr, sc = split_mx(w)
            self._mx, self._mx_key = dict(w_hi=hi, w_lo=qr, w_sc=sc.permute(0, 2, 1).contiguous()), key
        return self._mx

    @torch.no_grad()
    def packed_first_split(self, device, n_var, inv_off):
        """First layer split by input channels of the (padded) input buffer: channels [0, n_var) vary per refinement
        iteration (the cost volume), channels [inv_off, cin_pad) are loop-invariant (x_d3), everything between is padding.
        Returns (variable part, invariant part): variable = weights over buffer channels [0, round_up(n_var,32)) (+ bias,
        ReLU); invariant = weights over buffer channels [inv_off, cin_pad) only — its launch reads the buffer at a channel
        offset, so no MFMA work is spent on the cost channels' zero weights (no bias, no ReLU)."""
        pk = self.packed(device)[0]
        key = ("split", n_var, inv_off, self._key)
        if getattr(self, "_split_key", None) == key:
            return self._split
        if inv_off % 32 or inv_off < n_var:
            raise lib.MagnetError("packed_first_split: the invariant channels must start at a multiple of 32 behind the variable ones")
        w = (pk["w_hi"].float() + pk["w_lo"].float())                       # exact: hi + lo is how the kernel sees them
        cv = _round_up(n_var, 32)
        wv = w[:, :, :cv].clone(); wv[:, :, n_var:] = 0
        wi = w[:, :, inv_off:].clone()
        # re-splitting hi + lo is exact (16 mantissa bits): hi, lo are recovered bit for bit
        vh, vl = split_bf16(wv.contiguous()); ih, il = split_bf16(wi.contiguous())
        var = dict(w_hi=vh, w_lo=vl, bias=pk["bias"], taps=pk["taps"], cin=cv, cout_pad=pk["cout_pad"], relu=pk["relu"])
        inv = dict(w_hi=ih, w_lo=il, bias=torch.zeros_like(pk["bias"]), taps=pk["taps"], cin=pk["cin"] - inv_off,
                   cout_pad=pk["cout_pad"], relu=False)
        self._split_key, self._split = key, (var, inv)
        return self._split

    def run_invariant(self, in_hi, in_lo, in_ld, rows, wp, work, n_var, inv_off):
        """Loop-invariant partial sums of the first layer: fp32 (rows, cout_pad), computed once per forward.  in_hi / in_lo:
        channel 0 of the buffer; the launch starts at channel inv_off."""
        _, inv = self.packed_first_split(in_hi.device, n_var, inv_off)
        key = ("partial", rows, inv["cout_pad"])
        if key not in work:
            work[key] = torch.empty((rows, inv["cout_pad"]), dtype=torch.float32, device=in_hi.device)
        lib.conv_mfma(in_hi[:, inv_off:], in_lo[:, inv_off:], in_ld, inv["cin"], inv["w_hi"], inv["w_lo"], inv["bias"], inv["taps"], wp,
                      False, rows, out_f32=work[key])
        return work[key]

    def run(self, in_hi, in_lo, in_ld, rows, wp, work, first_addend=None, n_var=None, inv_off=None, upsample=None, gauss=None, mx=None):
        """in_hi/in_lo: bf16 views whose data_ptr is row 0, channel 0 of this stack's input; `work`: dict for cached
        hidden buffers.  Returns (fp32 tensor (rows, cout_pad_last), cout_pad_last).
        upsample = (depths (n,B,2,h,w), outs (n,B,2,4h,4w)): the mask head's stack only (144 outputs, fused tail) — the learned
        convex upsampling runs in the tail's last layer and only `outs` is written (returns (None, 144)); see can_fuse_upsample().
        gauss = (gmm_in, gmm_out): G-Net's stack only — the Gaussian update runs behind the head (returns (None, 16)); can_fuse_gauss().
        mx = (in_sc, sc_rows): the input planes are in the fp16 + e4m3 operand format (in_hi fp16, in_lo the e4m3 container, in_sc the
        E8M0 plane of lib.pack_mx); fused-epilogue stacks with at least 65 536 rows only."""
        packs = self.packed(in_hi.device)
        if first_addend is not None:
            # first layer over the per-iteration channels only; the invariant part arrives as `first_addend`
            packs = [self.packed_first_split(in_hi.device, n_var, inv_off)[0]] + list(packs[1:])
        cur_hi, cur_lo, cur_ld = in_hi, in_lo, in_ld
        if self._chain is not None and self.fuse_epilogue:
            pk, ch = packs[0], self._chain
            key = ("out", rows, ch["cout_pad"])
            if key not in work and upsample is None and gauss is None:
                work[key] = torch.empty((rows, ch["cout_pad"]), dtype=torch.float32, device=in_hi.device)
            sink = ConvStackMFMA.event_sink
            if sink is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                sink.append((e0, e1, 2.0 * rows * (pk["cout_pad"] * pk["cin"] * pk["taps"] + 128 * (256 + ch["cout_pad"])), pk["taps"]))
            if mx is not None:
                if first_addend is not None:
                    raise lib.MagnetError("ConvStackMFMA.run: the fp16 + e4m3 format has no addend form")
                wm = self.packed_mx(in_hi.device)
                lib.conv_mfma(cur_hi, cur_lo, cur_ld, pk["cin"], wm["w_hi"], wm["w_lo"], pk["bias"], pk["taps"], wp, pk["relu"], rows,
                              out_f32=None if (upsample is not None or gauss is not None) else work[key],
                              tail=(ch["w_hi"], ch["w_lo"], ch["bias"], ch["cout_pad"]), upsample=upsample, gauss=gauss,
                              mx=(mx[0], wm["w_sc"], mx[1]))
            else:
                lib.conv_mfma(cur_hi, cur_lo, cur_ld, pk["cin"], pk["w_hi"], pk["w_lo"], pk["bias"], pk["taps"], wp, pk["relu"], rows,
                              out_f32=None if (upsample is not None or gauss is not None) else work[key], addend=first_addend,
                              tail=(ch["w_hi"], ch["w_lo"], ch["bias"], ch["cout_pad"]), upsample=upsample, gauss=gauss)
            if sink is not None:
                e1.record()
            return (None if (upsample is not None or gauss is not None) else work[key]), ch["cout_pad"]
        if self._chain is not None and self.fuse_tail and first_addend is None:
            pk = packs[0]
            key = ("hid", 0, rows, 128)
            if key not in work:
                work[key] = (torch.empty((rows, 128), dtype=torch.bfloat16, device=in_hi.device),
                             torch.empty((rows, 128), dtype=torch.bfloat16, device=in_hi.device))
            oh, ol = work[key]
            sink = ConvStackMFMA.event_sink
            if sink is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                sink.append((e0, e1, 2.0 * rows * pk["cout_pad"] * pk["cin"] * pk["taps"], pk["taps"]))
            lib.conv_mfma(cur_hi, cur_lo, cur_ld, pk["cin"], pk["w_hi"], pk["w_lo"], pk["bias"], pk["taps"], wp,
                          pk["relu"], rows, out_hi=oh, out_lo=ol)
            if sink is not None:
                e1.record()
            ch = self._chain
            key = ("out", rows, ch["cout_pad"])
            if key not in work:
                work[key] = torch.empty((rows, ch["cout_pad"]), dtype=torch.float32, device=in_hi.device)
            if sink is not None:
                c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
                c0.record()
                sink.append((c0, c1, 2.0 * rows * 128 * (256 + ch["cout_pad"]), 1))
            lib.conv1x1_chain(oh, ol, ch["w_hi"], ch["w_lo"], ch["bias"], work[key], rows, ch["cout_pad"])
            if sink is not None:
                c1.record()
            return work[key], ch["cout_pad"]
        for li, pk in enumerate(packs):
            last = li == len(packs) - 1
            sink = ConvStackMFMA.event_sink
            if sink is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                sink.append((e0, e1, 2.0 * rows * pk["cout_pad"] * pk["cin"] * pk["taps"], pk["taps"]))
            if last:
                key = ("out", rows, pk["cout_pad"])
                if key not in work:
                    work[key] = torch.empty((rows, pk["cout_pad"]), dtype=torch.float32, device=in_hi.device)
                lib.conv_mfma(cur_hi, cur_lo, cur_ld, pk["cin"], pk["w_hi"], pk["w_lo"], pk["bias"], pk["taps"], wp,
                              pk["relu"], rows, out_f32=work[key])
                if sink is not None:
                    e1.record()
                return work[key], pk["cout_pad"]
            key = ("hid", li & 1, rows, pk["cout_pad"])
            if key not in work:
                work[key] = (torch.empty((rows, pk["cout_pad"]), dtype=torch.bfloat16, device=in_hi.device),
                             torch.empty((rows, pk["cout_pad"]), dtype=torch.bfloat16, device=in_hi.device))
            oh, ol = work[key]
            lib.conv_mfma(cur_hi, cur_lo, cur_ld, pk["cin"], pk["w_hi"], pk["w_lo"], pk["bias"], pk["taps"], wp,
                          pk["relu"], rows, out_hi=oh, out_lo=ol, addend=first_addend if li == 0 else None)
            if sink is not None:
                e1.record()
            cur_hi, cur_lo, cur_ld = oh, ol, pk["cout_pad"]
