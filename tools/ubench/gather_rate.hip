// Dev microbenchmark (round 3): what a CU can GATHER from the L2 / Infinity Cache / HBM in 128-byte texels, issued the way
// the matcher's correlation issues them: a wave-load = 16 texel units x (4 lanes x 16 B), two loads per unit pair (32 B per lane),
// NL wave-loads in flight, then s_waitcnt vmcnt(0).  Texel indices are pseudo-random inside a region of `region` bytes per XCD-ish
// slice (block b uses region slice b % nslice), so the region size selects the level that serves the misses.
// Reports bytes per clock per CU (2.4 GHz nominal) and per-wave cycles per batch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int NL, int ADJ>
__global__ __launch_bounds__(256) void k_gather(const unsigned char* buf, u64 region_texels, int nslice, int nit, u64* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned char* base = buf + (u64)(blockIdx.x % nslice) * region_texels * 128;
    unsigned int h = (blockIdx.x * 4 + wv) * 2654435761u + 12345u;
    u4 acc = u4{0, 0, 0, 0};
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nit; ++it) {
        u4 d[NL];
#pragma unroll
        for (int a = 0; a < NL / 2; ++a) {
            h = h * 1664525u + 1013904223u;
            // 16 units per wave-load pair; ADJ: units come as 2x2 quads (taps t, t+1, t+W, t+W+1 with W = 162), else independent
            unsigned int u = lane >> 2;
            unsigned int r = (h ^ ((ADJ ? (u >> 2) : u) * 0x9E3779B9u));
            r ^= r >> 15; r *= 0x85EBCA6Bu; r ^= r >> 13;
            u64 tex = (u64)r % (region_texels - 200);
            if (ADJ) tex += (u & 1) + ((u >> 1) & 1) * 162;
            const unsigned char* p = base + tex * 128 + (lane & 3) * 32;
            d[2 * a] = *reinterpret_cast<const u4*>(p);
            d[2 * a + 1] = *reinterpret_cast<const u4*>(p + 16);
        }
#pragma unroll
        for (int a = 0; a < NL; ++a) acc += d[a];
    }
    const u64 t1 = __builtin_readcyclecounter();
    if (acc.x == 0x12345678u) out[0] = acc.y;
    if (lane == 0) out[1 + blockIdx.x * 4 + wv] = t1 - t0;
}

template <int NL, int ADJ> void run(const unsigned char* d_buf, u64* d_out, u64 region_bytes, int W, int nit) {
    const int blocks = 256 * W;
    const u64 region_texels = region_bytes / 128;
    const int nslice = 1;                       // every block gathers from the SAME region: region <= 4 MB stays in each XCD's L2, <= 256 MB in the Infinity Cache
    hipLaunchKernelGGL((k_gather<NL, ADJ>), dim3(blocks), dim3(256), 0, 0, d_buf, region_texels, nslice, nit, d_out);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather<NL, ADJ>), dim3(blocks), dim3(256), 0, 0, d_buf, region_texels, nslice, nit, d_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * nit * NL * 1024.0;
    printf("  NL=%d %s region %6.1f MB  W=%d: %7.1f GB/s = %5.1f B/clk/CU (texel bytes through the TA; unique <= that)\n", NL, ADJ ? "quads" : "indep",
           region_bytes / 1048576.0, W, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
    unsigned char* d_buf; u64* d_out;
    hipMalloc(&d_buf, (1ull << 31) + (1 << 20)); hipMemset(d_buf, 1, (1ull << 31) + (1 << 20));
    hipMalloc(&d_out, (1 + 256 * 8 * 4) * sizeof(u64));
    const u64 regions[] = {64ull << 10, 1ull << 20, 3ull << 20, 24ull << 20, 128ull << 20, 1ull << 30};
    for (u64 rg : regions)
        for (int W : {2, 5, 8}) {
            run<8, 0>(d_buf, d_out, rg, W, 400);
            run<8, 1>(d_buf, d_out, rg, W, 400);
            run<2, 1>(d_buf, d_out, rg, W, 1600);
        }
    return 0;
}
