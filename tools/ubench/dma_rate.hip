// Dev microbenchmark (round 4): what an LDS-DMA wave-load (buffer_load_* ... lds) costs next to the same load into VGPRs, by width
// and lane pattern — the matcher's quad / texel fetch patterns.  L1/L2-resident (16 KB per wave), W waves per SIMD.
// cycles per wave-instruction per CU = wall time x 2.4 GHz / (nit x 8 x W)  [4 SIMDs stream; x4 = per SIMD]
// Build: hipcc --offload-arch=gfx950 -O2 -o dma_rate dma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef void __attribute__((address_space(3)))* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

// PAT 0: DMA dword, 8 lanes x 4 B per 32-byte slot, slots scattered (stride 96 B)      1: the same into VGPRs (buffer_load_dword)
//     2: DMA dwordx4, 8 lanes x 16 B per 128-byte texel, texels scattered (stride 384 B)  3: the same into VGPRs
//     4: DMA dword, lane*4 contiguous                                                   5: DMA dwordx4, lane*16 contiguous
//     6: PAT 0 with 40 of the 64 lanes out of range (bounds check -> zero)            7: PAT 2 with 32 of 64 lanes out of range
template <int PAT>
__global__ __launch_bounds__(256) void k(u64* out, int nit, const unsigned char* gbuf) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = gbuf + (size_t)(blockIdx.x & 255) * 65536;
    u64 a = (u64)base; unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((u64)hi << 32) | lo), 0, 65536, 0x00020000);
    unsigned off;
    switch (PAT) {
        case 0: case 1: case 6: off = (lane >> 3) * 96 + (lane & 7) * 4; break;
        case 2: case 3: case 7: off = (lane >> 3) * 384 + (lane & 7) * 16; break;
        case 4: off = lane * 4; break;
        default: off = lane * 16; break;
    }
    off += wv * 16384;
    if (PAT == 6 && lane >= 24) off = 0x7fffff00u;
    if (PAT == 7 && lane >= 32) off = 0x7fffff00u;
    unsigned char* l0 = smem + wv * 8192;
    u4 d[8]; for (int i = 0; i < 8; ++i) d[i] = u4{0, 0, 0, 0};
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nit; ++it) {
        if (PAT == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i].x += __builtin_amdgcn_raw_buffer_load_b32(rs, off + i * 1024, 0, 0);
        } else if (PAT == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { u4 t = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + i * 1024, 0, 0)); d[i].x += t.x; d[i].w += t.w; }
        } else if (PAT == 0 || PAT == 4 || PAT == 6) {
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(l0 + i * 256), 4, off + i * 1024, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(l0 + (i & 7) * 1024), 16, off + i * 1024, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const u64 t1 = __builtin_readcyclecounter();
    unsigned r = 0; for (int i = 0; i < 8; ++i) r += d[i].x + d[i].w;
    r += ((unsigned*)smem)[lane];
    if (r == 0x12345679u) out[0] = r;
    if (lane == 0) out[1 + blockIdx.x * 4 + wv] = t1 - t0;
}

static u64* d_out; static unsigned char* d_buf;
template <int PAT> static void run(const char* name) {
    printf("%-70s", name);
    const int nit = 1000;
    for (int W : {1, 2, 4, 5}) {
        const int blocks = 256 * W;
        hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 32768, 0, d_out, nit, d_buf);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 32768, 0, d_out, nit, d_buf); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  W=%d %6.2f/CU", W, ms * 1e-3 * 2.4e9 / ((double)nit * 8 * W * 4));
    }
    printf("\n");
}
int main() {
    hipMalloc(&d_out, (1 + 256 * 8 * 4) * sizeof(u64));
    hipMalloc(&d_buf, (size_t)256 * 65536 + 65536); hipMemset(d_buf, 0, (size_t)256 * 65536 + 65536);
    printf("# cycles per wave-load per CU at 2.4 GHz (wall), W waves per SIMD, 8 loads in flight per wave\n");
    run<0>("LDS-DMA dword   8 lanes x 4 B per 32-byte slot (scattered slots)");
    run<1>("VGPR    dword   8 lanes x 4 B per 32-byte slot (scattered slots)");
    run<2>("LDS-DMA dwordx4 8 lanes x 16 B per 128-byte texel (scattered)");
    run<3>("VGPR    dwordx4 8 lanes x 16 B per 128-byte texel (scattered)");
    run<4>("LDS-DMA dword   lane*4");
    run<5>("LDS-DMA dwordx4 lane*16");
    run<6>("LDS-DMA dword   slots, 40 of 64 lanes out of range");
    run<7>("LDS-DMA dwordx4 texels, 32 of 64 lanes out of range");
    return 0;
}
