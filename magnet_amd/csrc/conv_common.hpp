// conv_common.hpp — launch parameters of the matrix-core convolution kernels (conv_mfma.hip), shared with api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace magnet {

struct ConvParams {
    const uint16_t* in_hi;  const uint16_t* in_lo;    // activations, row 0 of the flattened padded grid
    const uint16_t* w_hi;   const uint16_t* w_lo;     // [taps][cout_pad][cin]
    const float*    bias;                             // [cout_pad]
    uint16_t* out_hi; uint16_t* out_lo;               // OUT mode 0: bf16 planes [rows][cout_pad]
    float*    out_f32;                                // OUT mode 1: fp32 [rows][cout_pad]
    long long rows;                                   // B*(h+2)*(w+2)
    int cin, cout_pad, taps, wp, relu, out_mode;
    int in_ld;                                        // elements between consecutive input rows (>= cin)
    const float* addend;                              // optional fp32 (rows, addend_ld) added before bias/ReLU
    int addend_ld;
    int tap_off[9];                                   // input row offset of each tap (dilation / 2x2 windows: host-computed)
    int min_off, max_off;                             // min / max of tap_off[0..taps)
    int tap_n, tap_o0, tap_sy, tap_sx;                // taps = tap_n^2; offset of tap (ty,tx) = (ty+o0)*sy + (tx+o0)*sx — the kernel
                                                      // steps (ty,tx) with scalar counters instead of loading tap_off[] per K step
    int out_ld;                                       // elements between consecutive output rows (>= cout_pad)
    const uint16_t* add_hi; const uint16_t* add_lo;   // optional split-bf16 addend (residual connections), rows x add_ld
    int add_ld;
    int img_rows, hp, pad;                            // img_rows = hp*wp > 0: rows are decoded to (image, y, x) and outputs of
                                                      // the `pad`-wide border are written as zeros (next layer's zero padding)
    int repad;                                        // > 0: interior rows only, re-addressed to a grid with border repad-1
    // fused 1x1 tail (cout_pad == 128 only): relu(1x1 128->128), relu(1x1 128->128), 1x1 128->tail_cout applied to the
    // workgroup's tile before it leaves the CU; weights [128][128],[128][128],[tail_cout][128] concatenated, bias likewise
    const uint16_t* tail_w_hi; const uint16_t* tail_w_lo; const float* tail_bias;
    int tail_cout;                                    // 16, 128 or 144; result fp32 (rows, tail_cout) at out_f32
    // fused convex upsampling (tail_cout == 144 only; models/MAGNET.py:15-27): the tail's last layer keeps its (rows, 144) mask
    // logits in registers, soft-maxes the 9 neighbour weights of each of the 16 sub-pixels and writes the x4-upsampled (mu, sigma)
    // of up_npred stacked predictions: depth (up_npred, B, 2, up_h, up_w) -> up_out (up_npred, B, 2, 4 up_h, 4 up_w); nothing is
    // written to out_f32.  Rows are positions of (B, up_h + 2, up_w + 2) grids (wp = up_w + 2).
    const float* up_depth; float* up_out;
    int up_npred, up_h, up_w, up_B;
    // fused Gaussian update (tail_cout == 16 only; models/MAGNET.py:60-69): the G-Net head's two outputs (o0, o1) of an interior
    // position update (mu, sigma) in place of the (rows, 16) fp32 write: gu_in (up_B, 2, up_h, up_w) -> gu_out, same layout
    const float* gu_in; float* gu_out;
    int variant;                                      // dev: bit 1 = 8-wave ping-pong K loop (conv_mfma.hip, PP) instead of the default
    // round 4, "2-unit" operand format of the 128-wide 3x3 layers (in_sc != nullptr): in_hi / w_hi = fp16 planes, in_lo / w_lo = per
    // 64-byte (row, 32-channel chunk) slice the 32 e4m3 bytes of hi then the 32 e4m3 bytes of lo = x - fp16(x), each block scaled by
    // its E8M0 exponent: in_sc [cin / 32][sc_rows] u32 {E8M0 of the hi block, E8M0 of the lo block, 0, 0}, w_sc [taps][cin / 32][cout_pad] u32
    const uint32_t* in_sc; const uint32_t* w_sc;
    long long sc_rows;                                // rows of one chunk plane of in_sc (>= rows)
};

struct ChainParams {
    const uint16_t* in_hi;  const uint16_t* in_lo;    // (rows, 128)
    const uint16_t* w_hi;   const uint16_t* w_lo;     // [128][128], [128][128], [cout_pad][128] concatenated
    const float*    bias;                             // 128 + 128 + cout_pad
    float*          out;                              // (rows, cout_pad) fp32
    long long rows;
    int cout_pad;
};

hipError_t launch_conv_mfma(const ConvParams&, hipStream_t);
hipError_t launch_conv1x1_chain(const ChainParams&, hipStream_t);
hipError_t launch_pack_mx(const float* in, uint16_t* out_f16, uint8_t* out_qr, uint32_t* out_sc, int N, int C, int h, int w, int ctot, int c_off,
                          long long sc_rows, long long in_img_stride, hipStream_t s);

}  // namespace magnet
