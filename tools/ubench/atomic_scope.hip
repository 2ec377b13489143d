// Dev microbenchmark: fp32 global atomic-add rate on gfx950 at agent scope versus workgroup scope into a
// per-XCD private copy (selected with the XCC_ID hardware register), plus a correctness check of the latter.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define N_IT 64
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }
template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, size_t n_floats, unsigned* xcc_hist) {
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(xcc_hist + xcc * 2 + ((blockIdx.x % 8) == xcc ? 0 : 1), 1u);
    float* base = (MODE == 0) ? buf : buf + (size_t)xcc * n_floats;
    unsigned s = blockIdx.x * 2654435761u + (threadIdx.x >> 3) * 40503u;
    for (int it = 0; it < N_IT; ++it) {
        s = s * 1664525u + 1013904223u;
        const size_t texel = (s >> 8) % (n_floats / 64);                 // a 64-float "texel"
        float* p = base + texel * 64 + (threadIdx.x & 7) * 4;
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) __hip_atomic_fetch_add(p + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else           __hip_atomic_fetch_add(p + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
__global__ void reduce8(const float* in, float* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) { float s = 0; for (int c = 0; c < 8; ++c) s += in[c * n + i]; out[i] = s; }
}
int main() {
    const size_t n = (size_t)5 << 20;               // 20 MB region (x8 copies for mode 1)
    float *a, *b, *r; unsigned* hist;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4 * 8); hipMalloc(&r, n * 4); hipMalloc(&hist, 64 * 4);
    hipMemset(hist, 0, 64 * 4);
    const int blocks = 256 * 32;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipMemset(a, 0, n * 4); else hipMemset(b, 0, n * 4 * 8);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, a, n, hist);
            else           hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, b, n, hist);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double atoms = (double)blocks * 256 * N_IT * 4;
        printf("mode %d (%s): %.3f ms, %.1f G atomics/s\n", mode, mode ? "workgroup scope, per-XCC copy" : "agent scope", ms, atoms / ms * 1e-6);
    }
    hipLaunchKernelGGL(reduce8, dim3((n + 255) / 256), dim3(256), 0, 0, b, r, n);
    std::vector<float> ha(n), hr(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), r, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; double tot = 0;
    for (size_t i = 0; i < n; ++i) { bad += ha[i] != hr[i]; tot += ha[i]; }
    printf("sum check: total %.0f (expected %.0f), mismatching elements %zu\n", tot, (double)blocks * 256 * N_IT * 4, bad);
    unsigned hh[64]; hipMemcpy(hh, hist, 64 * 4, hipMemcpyDeviceToHost);
    for (int x = 0; x < 16; ++x) if (hh[2 * x] || hh[2 * x + 1]) printf("xcc %d: blocks with id%%8==xcc %u, other %u\n", x, hh[2 * x], hh[2 * x + 1]);
    return 0;
}
