// pp_barrier.hip — what does one barrier interval of the convolution's ping-pong loop cost beyond its MFMAs?
// 8 waves per workgroup (one workgroup per CU, 2 waves per SIMD) in two groups half a step apart, as conv_mfma.hip's PP loops: in every
// barrier interval one group issues NM independent-accumulator bf16 MFMAs (16x16x32, 16 cycles each) and the other only waits at the
// barrier (its LOAD phase, here empty).  cycles per interval - 16 * NM = the fixed cost of an interval (barrier round trip, pipe refill).
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -o pp_barrier pp_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NM, bool LOADS>
__global__ __launch_bounds__(512, 2) void pp_kernel(float* out, const uint4* src, int iters) {
    __shared__ uint4 lds[4096];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, grp = wv >> 2;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
    f32x4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a = __builtin_bit_cast(bf16x8_t, lds[lane]), b = __builtin_bit_cast(bf16x8_t, lds[64 + lane]);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        // LOAD phase: nothing (or 16 ds_read_b128 whose results feed the next MFMAs), then the barrier
        if (LOADS) {
            uint4 t = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) { const uint4 v = lds[((it + r) * 64 + lane) & 4095]; t.x ^= v.x; t.y ^= v.y; }
            a = __builtin_bit_cast(bf16x8_t, make_uint4(t.x, t.y, __builtin_bit_cast(uint4, a).z, __builtin_bit_cast(uint4, a).w));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // COMPUTE phase
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 15], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NM, bool LOADS>
static void run(float* out, uint4* src, int ncu, double ghz) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((pp_kernel<NM, LOADS>), dim3(ncu), dim3(512), 0, 0, out, src, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pp_kernel<NM, LOADS>), dim3(ncu), dim3(512), 0, 0, out, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // every loop trip = 2 barrier intervals (each group computes once)
    const double cyc = ms * 1e-3 * ghz * 1e9 / (2.0 * iters);
    printf("NM = %3d MFMAs per interval%s: %7.1f cycles per interval at %.2f GHz = %5.1f %% matrix pipe, fixed %6.1f cycles\n", NM,
           LOADS ? " + 16 ds_read_b128 in the LOAD phase" : "", cyc, ghz, 100.0 * 16 * NM / cyc, cyc - 16.0 * NM);
}

int main() {
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    int khz = 2400000; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    float* out; uint4* src; hipMalloc(&out, (size_t)ncu * 512 * 4); hipMalloc(&src, 4096 * 16); hipMemset(src, 0x3c, 4096 * 16);
    printf("(cycles assume the nominal %.2f GHz; the chip may clock lower under this load — compare the rows, not the absolute numbers)\n", ghz);
    run<24, false>(out, src, ncu, ghz); run<48, false>(out, src, ncu, ghz); run<96, false>(out, src, ncu, ghz); run<144, false>(out, src, ncu, ghz);
    run<48, true>(out, src, ncu, ghz); run<96, true>(out, src, ncu, ghz); run<144, true>(out, src, ncu, ghz);
    return 0;
}
