// Dev microbenchmark: issue rate of VALU ops on gfx950 (cycles per wave64 instruction per SIMD), 8 waves/SIMD, 8 independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define N_IT 2048
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float s, int si) {
    float a[8]; unsigned u[8]; f2 v[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 77u + i; v[i] = f2{a[i], a[i] * 0.5f}; }
    double dacc = threadIdx.x;
    const f2 s2 = {s, s * 1.0001f}, h2c = {0.5f, 0.25f};
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], s, 0.5f);
            if (OP == 1) a[i] = a[i] * s;
            if (OP == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 3) { dacc += (double)a[i]; }
            if (OP == 4) u[i] = u[i] * 3u + i;
            if (OP == 5) u[i] = __umul24(u[i], 3u) + i;
            if (OP == 6) a[i] = __builtin_amdgcn_fmed3f(a[i], -10.f, s);
            if (OP == 7) a[i] = __builtin_floorf(a[i]);
            if (OP == 8) v[i] = __builtin_elementwise_fma(v[i], s2, h2c);
            if (OP == 9) v[i] = v[i] * s2;
            if (OP == 10) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, u[i]), __builtin_bit_cast(bf2, u[(i + 1) & 7]), a[i], false);
            if (OP == 11) a[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, u[i]), __builtin_bit_cast(h2, u[(i + 1) & 7]), a[i], false);
            if (OP == 12) a[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0xB1, 0xf, 0xf, true));
            if (OP == 13) u[i] = __builtin_amdgcn_mbcnt_lo(u[i], u[(i + 1) & 7]);
            if (OP == 14) a[i] = (u[i] & 1) ? a[i] : s;                       // v_cmp + v_cndmask (2 ops)
            if (OP == 15) u[i] = (unsigned)__builtin_amdgcn_ds_bpermute((int)(u[i] & 252), (int)u[(i + 1) & 7]);
            if (OP == 16) u[i] = (unsigned)(int)a[i] + u[i];                 // cvt_i32_f32 + add
            if (OP == 17) u[i] = (unsigned)__builtin_amdgcn_readlane((int)u[i], si) + u[(i + 1) & 7];   // v_readlane + add
            if (OP == 18) a[i] = __builtin_fmaf(__uint_as_float(u[i] << 16), __uint_as_float(u[(i + 1) & 7] << 16), a[i]);  // shift x2 + fma
            if (OP == 19) a[i] = __builtin_fmaf(__uint_as_float(u[i] & 0xffff0000u), __uint_as_float(u[(i + 1) & 7] & 0xffff0000u), a[i]);
        }
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i] + v[i].x + v[i].y + u[i];
    out[blockIdx.x * 256 + threadIdx.x] = r + (float)dacc;
}
template <int OP> void run(const char* name, int ops_per_inner, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 3);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 3); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)blocks * 4 * N_IT * 8 * ops_per_inner / 1024.0;
    printf("%-34s %8.3f ms -> %.2f cycles per wave-instr per SIMD @2.4GHz (%d instr per step)\n", name, ms, ms * 1e6 / per_simd * 2.4, ops_per_inner);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", 1, d); run<1>("v_mul_f32", 1, d); run<2>("v_rcp_f32", 1, d);
    run<3>("cvt_f64_f32+add_f64", 2, d); run<4>("v_mul_lo_u32+add", 2, d); run<5>("v_mad_u32_u24", 1, d);
    run<6>("v_med3_f32", 1, d); run<7>("v_floor_f32", 1, d);
    run<8>("v_pk_fma_f32", 1, d); run<9>("v_pk_mul_f32", 1, d);
    run<10>("v_dot2c_f32_bf16", 1, d); run<11>("v_dot2_f32_f16", 1, d); run<12>("v_add_f32 dpp quad_perm", 1, d);
    run<13>("v_mbcnt_lo", 1, d); run<14>("v_and+v_cmp+v_cndmask", 3, d); run<15>("ds_bpermute (+and)", 2, d);
    run<16>("v_cvt_i32_f32+add", 2, d); run<17>("v_readlane+add", 2, d); run<18>("2x lshl + fma", 3, d); run<19>("2x and + fma", 3, d);
    return 0;
}
