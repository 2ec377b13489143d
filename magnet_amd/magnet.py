"""Host mirror of the reference's `models/MAGNET.py`: same class names, constructor argument,
`forward(ref_img, nghbr_imgs, nghbr_poses, is_valid, cam_intrins, mode)` signature, return value
(list of (B,2,H,W) tensors) and state_dict keys (`g_net.gnet.*`, `mask_head.*`, `d_net.*`, `f_net.*`).

What runs where (inference, conv_backend='mfma'):
  * candidate sampling + warping + consistency-weighted score  -> HIP kernel (lib.cost_volume_cw, production matcher)
  * G-Net / mask-head convolutions                              -> HIP matrix-core kernel (lib.conv_mfma, bf16x3 split)
  * Gaussian update tail, convex upsampling                     -> HIP kernels (lib.gaussian_update*/upsample_depth*)
  * a PSMNet-structured F-Net                                   -> HIP matrix-core path (magnet_amd/fnet.py)
  * D-Net                                                       -> caller-provided module (out of scope)
Under autograd (mode='train' with trainable g_net / mask_head) the convolutions, the Gaussian update and the convex
upsampling are torch ops, so gradients reach g_net and mask_head exactly as in the reference (train_MaGNet.py:87-98);
the matcher is forward-only there too (the reference detaches its inputs, MAGNET.py:154,167-168).
"""
from __future__ import annotations

import math
from statistics import NormalDist

import torch
import torch.nn as nn

from . import lib
from .convnet import ConvStackMFMA
from .homography import CostVolumeCW, est_costvolume_F


def depth_sampling(sampling_range, n_samples) -> list:
    """k_j = mid-points of the N(0,1) quantile bins over +-sampling_range
    (reference MAGNET.depth_sampling, models/MAGNET.py:120-128; float64 on the host)."""
    P_total = math.erf(sampling_range / math.sqrt(2.0))
    nd = NormalDist()
    k = [nd.inv_cdf((1 - P_total) / 2 + (i / n_samples) * P_total) for i in range(n_samples + 1)]
    return [(k[i + 1] + k[i]) / 2 for i in range(n_samples)]


def _upsample_depth_torch(depth, up_mask, k):
    """Differentiable form of the learned convex upsampling (reference models/MAGNET.py:15-27): per low-resolution
    pixel, softmax over the 9 neighbours of each of the k*k sub-pixels, convex combination of the zero-padded 3x3 patch."""
    N, C, H, W = depth.shape
    wgt = torch.softmax(up_mask.reshape(N, 1, 9, k, k, H, W), dim=2)
    patch = nn.functional.unfold(depth, kernel_size=3, padding=1).reshape(N, C, 9, 1, 1, H, W)
    up = (wgt * patch).sum(dim=2)                                          # (N, C, k, k, H, W)
    return up.permute(0, 1, 4, 2, 5, 3).reshape(N, C, k * H, k * W)


def upsample_depth_via_mask(depth, up_mask, k):
    """Learned convex upsampling (reference models/MAGNET.py:15-27).  HIP kernel when no gradient is needed; under
    autograd (MagnetLoss is computed on these tensors, train_MaGNet.py:87-98) the torch form, so that g_net and
    mask_head receive their gradients."""
    if torch.is_grad_enabled() and (depth.requires_grad or up_mask.requires_grad):
        return _upsample_depth_torch(depth, up_mask, int(k))
    return lib.upsample_depth(depth.detach().float().contiguous(), up_mask.detach().float().contiguous(), int(k))


def load_checkpoint(fpath, model):
    """Same behaviour as the reference's loader (models/MAGNET.py:31-43): accepts {'model': sd} or a
    bare state_dict, strips a leading 'module.'."""
    ckpt = torch.load(fpath, map_location="cpu")
    if "model" in ckpt:
        ckpt = ckpt["model"]
    load_dict = {}
    for k, v in ckpt.items():
        load_dict[k.replace("module.", "") if k.startswith("module.") else k] = v
    model.load_state_dict(load_dict)
    return model


class GNET(nn.Module):
    """conv3x3(ch_in->128)-ReLU-1x1-ReLU-1x1-ReLU-1x1(->2), then the Gaussian update
    (reference models/MAGNET.py:47-70).  Inference (no_grad) uses the fused HIP tail; under autograd
    the tail is evaluated with torch ops so g_net stays trainable like the reference's."""

    def __init__(self, ch_in, ch_out=2):
        super().__init__()
        h_dim = 128
        self.gnet = nn.Sequential(
            nn.Conv2d(ch_in, h_dim, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(h_dim, h_dim, 1), nn.ReLU(inplace=True),
            nn.Conv2d(h_dim, h_dim, 1), nn.ReLU(inplace=True),
            nn.Conv2d(h_dim, ch_out, 1))

    def forward(self, cost_volume, ref_gmm):
        d_output = self.gnet(cost_volume)
        if torch.is_grad_enabled() and d_output.requires_grad:
            mu_0, sigma_0 = torch.split(ref_gmm, 1, dim=1)
            mu_1, sigma_1 = torch.split(d_output, 1, dim=1)
            mu_new = mu_0 + (mu_1 * sigma_0)
            sigma_new = (nn.functional.elu(sigma_1) + 1.0 + 1e-10) * sigma_0
            return torch.cat([mu_new, sigma_new], dim=1)
        return lib.gaussian_update(d_output.float().contiguous(), ref_gmm.detach().float().contiguous())


class MAGNET(nn.Module):
    """Drop-in for the reference's MAGNET (models/MAGNET.py:73-175).

    `args` carries the same fields (MAGNET_sampling_range, MAGNET_num_samples, MAGNET_mvs_weighting,
    MAGNET_num_train_iter, MAGNET_num_test_iter, dpv_height, dpv_width, downsample_ratio).  The frozen
    backbones are out of this build's scope: pass them as `d_net` (img -> ((N,2,h,w), (N,256,h,w))) and
    `f_net` (img -> (N,F,h,w)); both are required (an omitted backbone is an error naming the argument).
    `feat_dtype`: 'fp32' or 'bf16' storage of F-Net features inside the matcher.
    `conv_backend`: 'mfma' runs g_net / mask_head on the bf16x3 matrix-core kernel at inference (csrc/conv_mfma.hip,
    fp32-grade); 'torch' keeps them on nn.Conv2d (MIOpen).  Autograd always takes the torch path."""

    def __init__(self, args, d_net: nn.Module | None = None, f_net: nn.Module | None = None,
                 feat_dtype: str = "fp32", conv_backend: str = "mfma"):
        super().__init__()
        self.args = args
        if d_net is None or f_net is None:
            # The reference builds its frozen backbones itself (MAGNET.py:97-108: DNET needs torch.hub + a checkpoint).  This module
            # accelerates the matching path only: the caller constructs them (the reference's own classes, or magnet_amd.fnet.FNET for
            # the F-Net) and hands them over — there is no import fallback to a MaGNet checkout.
            raise lib.MagnetError("MAGNET(args, d_net=..., f_net=...): both backbone modules are required "
                                  "(d_net: img -> ((N,2,h,w), (N,256,h,w)); f_net: img -> (N,F,h,w)); see INTEGRATION.md")
        # Reference behaviour (MAGNET.py:80-92): the frozen backbones are loaded from args.DNET_ckpt / args.FNET_ckpt.  The modules
        # come from the caller here, but the loading stays this constructor's job: a checkpoint path that is set is loaded into the
        # passed module (same loader, so a bad path or a key mismatch raises exactly as in the reference); a path that is unset
        # (None / '') means the caller has initialised the module itself — said once, loudly, because a train_MaGNet.py-style run
        # would otherwise proceed against randomly initialised frozen backbones.
        for name, net, key in (("d_net", d_net, "DNET_ckpt"), ("f_net", f_net, "FNET_ckpt")):
            path = getattr(args, key, None)
            if path:
                load_checkpoint(path, net)
            else:
                import warnings
                warnings.warn(f"MAGNET: args.{key} is not set — `{name}` is used as passed (the reference loads it from "
                              f"args.{key}, models/MAGNET.py:80-92)", stacklevel=2)
        self.d_net = d_net
        self.f_net = f_net
        for net in (self.d_net, self.f_net):
            for prm in net.parameters():
                prm.requires_grad = False
            net.eval()

        self.sampling_range = args.MAGNET_sampling_range
        self.n_samples = args.MAGNET_num_samples
        self.weighting = args.MAGNET_mvs_weighting
        self.train_iter = args.MAGNET_num_train_iter
        self.test_iter = args.MAGNET_num_test_iter
        self.dpv_height = args.dpv_height
        self.dpv_width = args.dpv_width
        self.k_list = self.depth_sampling()
        self.downsample_ratio = args.downsample_ratio
        self.feat_dtype = feat_dtype
        self.matcher_path = 0          # kernel selection of the matcher (include/magnet_hip.h `path`)
        if conv_backend not in ("mfma", "torch"):
            raise lib.MagnetError(f"conv_backend must be 'mfma' or 'torch', got {conv_backend!r}")
        self.conv_backend = conv_backend
        self._work = {}                # cached device workspaces of the MFMA conv path, keyed by shape
        self.fuse_upsample = True      # the stacks' tails finish the job: G-Net's head applies the Gaussian update, the mask head
                                       # writes the upsampled predictions itself (no (B,144,h,w) mask in HBM); False: separate launches
        self.hoist_invariant = True    # I >= 2: compute the x_d3 part of G-Net's first layer once per forward
        self._stacks = None
        # mask head on a side stream next to matcher + G-Net.  Measured on MI355X: no gain (10.15 vs 10.05 ms per C2 step) —
        # the workgroups of the two queues do not co-reside usefully; kept as an option, off by default.
        self.overlap_mask_head = False
        # x_d3's NCHW fp32 -> split channel-last repack (HBM-bound, 0.6 ms per 64 C2 frames) on a side stream next to the first
        # matcher launch (latency / VALU-bound): the two overlap almost completely
        self.overlap_pack = False      # measured: 8.01 vs 8.08 ms per C2 step, and it stretches the matcher launch 1.06 -> 1.57 ms
        self._side = {}
        self.fuse_conv_tail = True     # 1x1 layers of g_net / mask_head fused into their 3x3 layer's epilogue
        self.fnet_mfma = True          # run a PSMNet-structured f_net on the matrix-core path (magnet_amd/fnet.py)
        self._fnet = None

        dnet_fdim = 256
        self.g_net = GNET(ch_in=dnet_fdim + self.n_samples, ch_out=2)
        h_dim = 128
        self.mask_head = nn.Sequential(
            nn.Conv2d(dnet_fdim, h_dim, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(h_dim, h_dim, 1), nn.ReLU(inplace=True),
            nn.Conv2d(h_dim, h_dim, 1), nn.ReLU(inplace=True),
            nn.Conv2d(h_dim, 9 * self.downsample_ratio * self.downsample_ratio, 1))
        self.upsample_depth = upsample_depth_via_mask

    def depth_sampling(self):
        return depth_sampling(self.sampling_range, self.n_samples)

    # -- the hot path proper: everything after the backbones --------------------------------------
    def gnet_input_buffer(self, B, h, w, device):
        """The split-bf16 zero-bordered channel-last buffer (B*(h+2)*(w+2), ctot) that holds [cost (D, padded to 64) | x_d3 (256)]
        for the matrix-core G-Net / mask head: (hi, lo, ctot, c_off of x_d3).  A D-Net running on the matrix-core path writes
        its x_d3 output into channels [c_off, c_off+256) of the interior rows and calls match_and_refine(x_d3_in_place=True):
        no NCHW tensor, no repack pass."""
        D = self.n_samples
        # x_d3 starts at a multiple of 64 channels = 128 bytes: its 128-byte pack stores and the convolutions' K slices are line-
        # aligned (at D = 5 the 8-channel offset of rounds 1-2 made every store straddle two lines: the x_d3 pack ran at 2.0
        # instead of 5.4 TB/s), and the loop-invariant convolution can read the x_d3 channels alone
        Dp = (D + 63) // 64 * 64
        if self._stacks is None:
            self._stacks = (ConvStackMFMA(self.g_net.gnet, in_map=[(0, D, 0), (D, 256, Dp)]),
                            ConvStackMFMA(self.mask_head))
        ctot = self._stacks[0].cin_pad()
        rows = B * (h + 2) * (w + 2)
        wkey = (str(device), B, h, w, ctot)
        work = self._work.get(wkey)
        if work is None:
            self._work.clear()                       # one shape at a time: the buffers are large
            work = self._work[wkey] = {
                "gin": (torch.zeros((rows, ctot), dtype=torch.bfloat16, device=device),   # zero border / zero pad
                        torch.zeros((rows, ctot), dtype=torch.bfloat16, device=device)),  # channels stay zero
                "cost": torch.empty((B, D, h, w), dtype=torch.float32, device=device)}
        return work["gin"][0], work["gin"][1], ctot, Dp

    def match_and_refine(self, ref_gmms, x_d3, ref_feat_4, nghbr_feat_4, nghbr_gmms, nghbr_poses,
                         is_valid, cam_intrins, mode="test", packed_feats=None, x_d3_in_place=False):
        """MAGNET.py:146-175 from backbone outputs.  Returns the list of upsampled (B,2,H,W).
        packed_feats: (ref_cl, src_pad) straight from the matrix-core F-Net instead of NCHW feature tensors.
        x_d3_in_place: x_d3 already sits in gnet_input_buffer() (x_d3 may then be None)."""
        thres = int(self.weighting.split("CW")[1])
        matcher = CostVolumeCW(ref_feat_4, nghbr_feat_4, nghbr_gmms, nghbr_poses, is_valid, cam_intrins,
                               thres, feat_dtype=self.feat_dtype, path=self.matcher_path, packed=packed_feats)
        B, _, h, w = ref_gmms.shape
        n_iter = self.train_iter if mode == "train" else self.test_iter
        training = torch.is_grad_enabled() and any(p.requires_grad for p in
                                                   list(self.g_net.parameters()) + list(self.mask_head.parameters()))
        if self.conv_backend == "mfma" and not training and self.downsample_ratio == 4 and (x_d3_in_place or x_d3.shape[1] == 256):
            return self._refine_mfma(matcher, ref_gmms, None if x_d3_in_place else x_d3, n_iter)
        if x_d3_in_place:
            raise lib.MagnetError("x_d3_in_place needs the matrix-core convolution path (conv_backend='mfma', inference)")

        # ---- torch (MIOpen) convolutions: training, or conv_backend='torch' ----
        # G-Net input buffer: cost volume first, then x_d3 (MAGNET.py:167); x_d3 is copied once
        gnet_in = torch.empty((B, self.n_samples + x_d3.shape[1], h, w), dtype=torch.float32, device=x_d3.device)
        gnet_in[:, self.n_samples:] = x_d3
        cost_view = gnet_in[:, :self.n_samples]          # the kernel writes here directly (strided frames)
        pred_list = [ref_gmms]
        for _ in range(n_iter):
            matcher(ref_gmm=pred_list[-1].detach(), k_list=self.k_list, out=cost_view)  # MAGNET.py:153-164
            # autograd saves G-Net's input, so training gets a fresh tensor per iteration (as the
            # reference's torch.cat does, MAGNET.py:167); inference reuses the buffer in place.
            g_in = gnet_in.clone() if training else gnet_in
            new_pred = self.g_net(g_in, pred_list[-1].detach())                         # MAGNET.py:167-168
            pred_list.append(new_pred)
        mask = self.mask_head(x_d3)                                                       # MAGNET.py:172
        return [self.upsample_depth(pred, mask, self.downsample_ratio) for pred in pred_list[1:]]

    def _refine_mfma(self, matcher, ref_gmms, x_d3, n_iter):
        """Inference loop with g_net / mask_head on the matrix cores.  One zero-bordered channel-last buffer
        (B, h+2, w+2, Ctot) in split-bf16 holds [cost (D, padded to 64) | x_d3 (256)]: x_d3 is packed once and is
        read in place by the mask head (channel offset) and by every G-Net iteration; only the D cost channels
        are re-packed per iteration (replaces torch.cat, MAGNET.py:167)."""
        B, _, h, w = ref_gmms.shape
        D = self.n_samples
        dev = ref_gmms.device
        gin_hi, gin_lo, ctot, Dp = self.gnet_input_buffer(B, h, w, dev)
        g_stack, m_stack = self._stacks
        g_stack.fuse_epilogue = m_stack.fuse_epilogue = self.fuse_conv_tail
        rows, wp = B * (h + 2) * (w + 2), w + 2
        work = self._work[(str(dev), B, h, w, ctot)]
        # Streams.  x_d3's repack (HBM-bound) runs on a side stream beside the first matcher launch (latency / VALU-bound);
        # the mask head depends on x_d3 only and may run there too (overlap_mask_head: measured no gain — two matrix-core
        # kernels do not co-reside usefully).  Events mark where the chains meet.
        main = torch.cuda.current_stream(dev)
        pack_stream = self._side_stream(dev) if (self.overlap_pack and x_d3 is not None) or self.overlap_mask_head else main
        ev_pack = torch.cuda.Event()
        if pack_stream is not main:
            pack_stream.wait_stream(main)            # the packed features exist; the previous forward no longer reads `gin`
        if x_d3 is not None:
            x_d3 = x_d3.detach().float().contiguous()
            if pack_stream is not main:
                x_d3.record_stream(pack_stream)
            with torch.cuda.stream(pack_stream):
                lib.pack_split(x_d3, gin_hi, gin_lo, ctot, Dp)
        ev_pack.record(pack_stream)
        mask_out = None
        if self.overlap_mask_head:
            with torch.cuda.stream(pack_stream):
                mask_out = m_stack.run(gin_hi[:, Dp:], gin_lo[:, Dp:], ctot, rows, wp, work.setdefault("mask", {}))  # MAGNET.py:172
                ev_mask = torch.cuda.Event(); ev_mask.record(pack_stream)
        pred_list = [ref_gmms.detach().float().contiguous()]
        # With more than one refinement iteration the x_d3 part of G-Net's first layer (256 of the 256+D input
        # channels) is loop-invariant: compute W_x * x_d3 once, add it in the epilogue of the per-iteration
        # convolution over the D cost channels only (K = 9*(256+D) -> 9*round_up(D,32) per iteration).
        partial = None
        if n_iter >= 2 and self.hoist_invariant:
            # the invariant convolution reads the x_d3 channels [Dp, Dp+256) only: nothing the matcher writes (a frame with a non-
            # finite cost cannot poison it), K = 9 * 256
            main.wait_event(ev_pack)
            partial = g_stack.run_invariant(gin_hi, gin_lo, ctot, rows, wp, work, D, Dp)
        split_out = self.matcher_path in (0, 2, 4)
        for _ in range(n_iter):
            if split_out:
                # the candidate-lane kernels write the D cost channels of the G-Net input buffer directly
                try:
                    matcher(ref_gmm=pred_list[-1], k_list=self.k_list, out_split=(gin_hi, gin_lo, ctot))  # MAGNET.py:153-164
                except lib.MagnetError as e:
                    if e.code != lib.E_SHAPE:        # bad arguments / HIP failures are real errors: never retried on another path
                        raise
                    split_out = False                # a shape only the generic kernel takes (V >= 26, very wide F): NCHW + repack
            if not split_out:
                matcher(ref_gmm=pred_list[-1], k_list=self.k_list, out=work["cost"])
                lib.pack_split(work["cost"], gin_hi, gin_lo, ctot, 0)
            main.wait_event(ev_pack)                                                                 # x_d3 channels are in place
            if self.fuse_upsample and g_stack.can_fuse_gauss(dev):
                new_pred = torch.empty_like(pred_list[-1])                                           # MAGNET.py:62 + 60-69 in one launch
                g_stack.run(gin_hi, gin_lo, ctot, rows, wp, work, first_addend=partial, n_var=D, inv_off=Dp, gauss=(pred_list[-1], new_pred))
                pred_list.append(new_pred)
                continue
            g_out, g_ld = g_stack.run(gin_hi, gin_lo, ctot, rows, wp, work, first_addend=partial, n_var=D, inv_off=Dp)  # MAGNET.py:62
            pred_list.append(lib.gaussian_update_cl(g_out, g_ld, pred_list[-1], h, w))               # MAGNET.py:60-69
        if len(pred_list) == 1:                               # no refinement iteration: the reference returns [] (MAGNET.py:172-173)
            return []
        if mask_out is None:
            main.wait_event(ev_pack)
            if self.fuse_upsample and self.downsample_ratio == 4 and m_stack.can_fuse_upsample(dev):
                # MAGNET.py:172-173 in one launch: behind the mask head's last 1x1 layer the workgroup soft-maxes its own logits (through
                # LDS) and writes every iteration's x4-upsampled prediction; the (B, 144, h, w) mask never reaches HBM
                d = torch.stack(pred_list[1:])
                outs = torch.empty((d.shape[0], B, 2, 4 * h, 4 * w), dtype=torch.float32, device=dev)
                m_stack.run(gin_hi[:, Dp:], gin_lo[:, Dp:], ctot, rows, wp, work.setdefault("mask", {}), upsample=(d, outs))
                return [outs[i] for i in range(outs.shape[0])]
            mask_out = m_stack.run(gin_hi[:, Dp:], gin_lo[:, Dp:], ctot, rows, wp, work.setdefault("mask", {}))      # MAGNET.py:172
        else:
            main.wait_event(ev_mask)
        mask_pad, mask_ld = mask_out
        return lib.upsample_depth_cl_n(pred_list[1:], mask_pad, mask_ld)                              # MAGNET.py:173 (one launch)

    def forward(self, ref_img, nghbr_imgs, nghbr_poses, is_valid, cam_intrins, mode="train"):
        B = ref_img.shape[0]
        with torch.no_grad():
            mono_gmms, x_d3 = self.d_net(torch.cat((ref_img, nghbr_imgs), dim=0))        # MAGNET.py:135
            mono_gmms = mono_gmms.detach()
            ref_gmms = mono_gmms[:B, ...]
            x_d3 = x_d3[:B, ...]
            nghbr_gmms = mono_gmms[B:, ...]
            runner = self._fnet_runner()
            if runner is not None and ref_img.is_cuda:
                # F-Net on the matrix cores; its last layer writes the matcher's layouts (no NCHW features, no pack)
                packed = runner.run(torch.cat((ref_img, nghbr_imgs), dim=0), n_ref=B, feat_dtype=self.feat_dtype)
                return self.match_and_refine(ref_gmms, x_d3, None, None, nghbr_gmms, nghbr_poses, is_valid, cam_intrins,
                                             mode, packed_feats=packed)
            feat_4 = self.f_net(torch.cat((ref_img, nghbr_imgs), dim=0))                 # MAGNET.py:142
            ref_feat_4 = feat_4[:B, ...]
            nghbr_feat_4 = feat_4[B:, ...]
        return self.match_and_refine(ref_gmms, x_d3, ref_feat_4, nghbr_feat_4, nghbr_gmms, nghbr_poses,
                                     is_valid, cam_intrins, mode)

    def _side_stream(self, dev):
        st = self._side.get(str(dev))
        if st is None:
            st = self._side[str(dev)] = torch.cuda.Stream(device=dev)
        return st

    def _fnet_runner(self):
        """FNetMFMA for a PSMNet-structured F-Net (ours or the reference's own class) when conv_backend == 'mfma'."""
        if self.conv_backend != "mfma" or not self.fnet_mfma or self.f_net.training:
            return None            # BatchNorm in training mode cannot be folded: the module's own forward runs (reference semantics)
        if self._fnet is None:
            from .fnet import FNetMFMA
            psm = getattr(self.f_net, "f_net", self.f_net)
            ok = all(hasattr(psm, a) for a in ("firstconv", "layer1", "layer2", "layer3", "layer4", "branch1", "lastconv"))
            self._fnet = FNetMFMA(psm) if ok else False
        return self._fnet or None


class MAGNET_F(nn.Module):
    """Mirror of the reference's F-Net training wrapper (models/MAGNET.py:179-202): F-Net features of the reference
    and source images -> est_costvolume_F (softmax over D fixed depth bins of the view-averaged feature correlation).
    The volume and its gradient w.r.t. the features run in hand-written HIP (magnet_amd.homography.est_costvolume_F);
    `f_net` is the caller's module (the reference's FNET; backbones are out of scope here) and trains through it."""

    def __init__(self, args, f_net: nn.Module):
        super().__init__()
        self.f_net = f_net

    def forward(self, ref_img, nghbr_imgs, nghbr_poses, is_valid, cam_intrins, d_center):
        B = ref_img.shape[0]
        feat_4 = self.f_net(torch.cat((ref_img, nghbr_imgs), dim=0))            # MAGNET.py:188
        ref_feat_4, nghbr_feat_4 = feat_4[:B], feat_4[B:]
        Rs_src = nghbr_poses[:, :, :3, :3]                                       # MAGNET.py:193-194
        ts_src = nghbr_poses[:, :, :3, 3]
        return est_costvolume_F(d_center, ref_feat_4, nghbr_feat_4, Rs_src, ts_src, is_valid, cam_intrins)
