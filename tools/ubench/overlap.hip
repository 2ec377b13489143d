// Dev microbenchmark (round 3): do vector-memory instructions overlap with vector-ALU work of OTHER waves on the same SIMD?
// Blocks of 512 threads = 8 waves = 2 per SIMD; wave role by (wave >> 2) & 1 (so each SIMD gets one wave of each role):
//   role 0: NV v_fma_f32 per trip;   role 1: NL L1-resident global_load_dwordx4 per trip (pattern: 16 lines x 4 lanes x 16 B).
// mode 0: only role-0 waves work, mode 1: only role-1 waves work, mode 2: both.  Time per trip from the wall clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
#define R8(x) x x x x x x x x
template <int SADDR>
__global__ __launch_bounds__(512) void k(const unsigned char* buf, int nit, int mode, float* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int role = (wv >> 2) & 1;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, c0 = 1.0001f, c1 = 0.5f;
    u4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    const unsigned char* base = buf + (size_t)(blockIdx.x & 255) * 65536;                  // scalar
    const unsigned off = wv * 8192 + (lane / 4) * 384 + (lane & 3) * 16;
    const unsigned char* p64 = base + off;
    if (role == 0 && mode != 1) {
        for (int it = 0; it < nit; ++it)
            asm volatile(R8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0), "v"(c1));
    }
    if (role == 1 && mode != 0) {
        for (int it = 0; it < nit; ++it) {
            if (SADDR)
                asm volatile("global_load_dwordx4 %0, %4, %5\n global_load_dwordx4 %1, %4, %5 offset:64\n global_load_dwordx4 %2, %4, %5 offset:1024\n global_load_dwordx4 %3, %4, %5 offset:1088\n s_waitcnt vmcnt(0)\n"
                             : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(off), "s"(base) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:64\n global_load_dwordx4 %2, %4, off offset:1024\n global_load_dwordx4 %3, %4, off offset:1088\n s_waitcnt vmcnt(0)\n"
                             : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(p64) : "memory");
        }
    }
    if (a0 + a1 + a2 + a3 + (float)(d0.x + d1.y + d2.z + d3.w) == 1234.5f) out[0] = 1;
}
int main() {
    unsigned char* buf; float* out;
    hipMalloc(&buf, 256 * 65536 + 4096); hipMemset(buf, 0, 256 * 65536 + 4096); hipMalloc(&out, 64);
    for (int saddr = 0; saddr < 2; ++saddr)
        for (int W = 1; W <= 4; W *= 2) {          // W blocks per CU: W waves of each role per SIMD
            float ms[3];
            for (int mode = 0; mode < 3; ++mode) {
                const int nit = 2000;
                if (saddr) hipLaunchKernelGGL(k<1>, dim3(256 * W), dim3(512), 0, 0, buf, nit, mode, out);
                else hipLaunchKernelGGL(k<0>, dim3(256 * W), dim3(512), 0, 0, buf, nit, mode, out);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
                if (saddr) hipLaunchKernelGGL(k<1>, dim3(256 * W), dim3(512), 0, 0, buf, nit, mode, out);
                else hipLaunchKernelGGL(k<0>, dim3(256 * W), dim3(512), 0, 0, buf, nit, mode, out);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[mode], e0, e1);
            }
            printf("%s addr, %d wave(s) of each role per SIMD: 32 v_fma per trip alone %.3f ms, 4 loads per trip alone %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n",
                   saddr ? "sgpr+voff" : "64-bit vgpr", W, ms[0], ms[1], ms[2], ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1]);
        }
    return 0;
}
