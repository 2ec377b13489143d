// Dev microbenchmark: cycles per wave for the exact per-candidate geometry (project + taps + window test),
// pure VALU (no memory), at 8 waves/SIMD, with 1 or 2 independent candidates interleaved per iteration.
#include "../../magnet_amd/csrc/cv_common.hpp"
#include <stdio.h>
using namespace magnet;
#define N_IT 2048
template <int ILP, int VARIANT>
__global__ __launch_bounds__(256) void k(float* out, PixelView pv0, GridConst gc, float fw, float fh, int Wp) {
    PixelView pv = pv0; pv.rpx += threadIdx.x * 1e-3f;
    float d[ILP]; float acc = 0.f; unsigned qa = 0;
    for (int i = 0; i < ILP; ++i) d[i] = 1.0f + 0.01f * (threadIdx.x & 63) + i;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            float ix, iy, zw;
            project(pv, gc, d[i], ix, iy, zw);
            if (VARIANT >= 1) {
                const float x0f = __builtin_floorf(ix), y0f = __builtin_floorf(iy);
                const float x1 = x0f + 1.0f, y1 = y0f + 1.0f;
                const float ax = x1 - ix, bx = ix - x0f, ay = y1 - iy, by = iy - y0f;
                const float nw = ax * ay, ne = bx * ay, sw = ax * by, se = bx * by;
                const bool inwin = (ix >= -1.0f) && (ix < fw) && (iy >= -1.0f) && (iy < fh);
                const unsigned qi = inwin ? (unsigned)(__mul24((int)y0f + 1, Wp) + ((int)x0f + 1)) : 0u;
                acc += nw + ne * 2.f + sw * 3.f + se * 4.f; qa += qi;
            }
            acc += zw + ix + iy;
            d[i] += 1e-4f;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + qa;
}
template <int ILP, int VARIANT> void run(const char* name, float* d) {
    PixelView pv{150.f, 2.f, 1.f, 1.f, 10.f, 5.f, 0.01f, 0.02f};
    GridConst gc{80.f, 60.f, 1.f / 80.f, 1.f / 60.f, 80.f, 60.f};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL((k<ILP, VARIANT>), dim3(blocks), dim3(256), 0, 0, d, pv, gc, 160.f, 120.f, 162);
    hipEventRecord(e0); hipLaunchKernelGGL((k<ILP, VARIANT>), dim3(blocks), dim3(256), 0, 0, d, pv, gc, 160.f, 120.f, 162); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double calls_per_simd = (double)blocks * 4 * N_IT * ILP / 1024.0;
    printf("%-34s %8.3f ms -> %.1f ns per wave-candidate per SIMD (= %.0f cycles @2.3GHz)\n", name, ms, ms * 1e6 / calls_per_simd, ms * 1e6 / calls_per_simd * 2.3);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<1, 0>("project only, ILP1", d); run<2, 0>("project only, ILP2", d);
    run<1, 1>("project+taps+qi, ILP1", d); run<2, 1>("project+taps+qi, ILP2", d); run<4, 1>("project+taps+qi, ILP4", d);
    return 0;
}
