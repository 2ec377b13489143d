// Dev microbenchmark: issue rate of common VALU ops on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define N_IT 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float s) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    double dacc = threadIdx.x; int ia = threadIdx.x;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[8]; for (int i = 0; i < 8; ++i) v[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    const f2 s2 = {s, s * 1.0001f}, h2 = {0.5f, 0.25f};
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], s, 0.5f);            // v_fma_f32 (8 independent chains)
            if (OP == 1) a[i] = a[i] * s;                                  // v_mul_f32
            if (OP == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);               // v_rcp_f32
            if (OP == 3) { dacc += (double)a[i]; }                         // cvt f64 + add f64
            if (OP == 4) { ia = ia * 3 + i; }                              // v_mul_lo_u32 (+add)
            if (OP == 5) { ia = __mul24(ia, 3) + i; }                      // v_mad_u32_u24
            if (OP == 6) a[i] = __builtin_amdgcn_fmed3f(a[i], -10.f, s);   // v_med3_f32
            if (OP == 7) a[i] = __builtin_floorf(a[i] * s);                // v_floor + mul
            if (OP == 8) v[i] = __builtin_elementwise_fma(v[i], s2, h2);   // v_pk_fma_f32
            if (OP == 9) v[i] = v[i] * s2;                                 // v_pk_mul_f32
            if (OP == 10) v[i] = v[i] + s2;                                // v_pk_add_f32
        }
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i] + v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r + (float)dacc + ia;
}
template <int OP> void run(const char* name, int ops_per_inner, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;        // 8 WGs/CU -> 8 waves/SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instrs = (double)blocks * 4 * N_IT * 8 * ops_per_inner;   // per launch
    double per_simd = wave_instrs / 1024.0;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", 1, d); run<1>("v_mul_f32", 1, d); run<2>("v_rcp_f32", 1, d);
    run<3>("cvt_f64_f32+add_f64", 2, d); run<4>("v_mul_lo_u32+add", 2, d); run<5>("v_mad_u32_u24", 1, d);
    run<6>("v_med3_f32", 1, d); run<7>("v_mul+v_floor", 2, d);
    run<8>("v_pk_fma_f32", 1, d); run<9>("v_pk_mul_f32", 1, d); run<10>("v_pk_add_f32", 1, d);
    return 0;
}
