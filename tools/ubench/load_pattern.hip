// Load-pattern microbenchmark (GPU box: hipcc --offload-arch=gfx950 -O3 load_pattern.hip -o /tmp/lp && /tmp/lp): 1 GiB of fp16
// [BH = 1024 heads][T = 4096][128] read once, a wave owning 64-token tiles (16 KB) of one head like the fused K kernels:
//   A: lane = channel pair, 64 x 4-byte loads per lane and tile (each wave instruction covers one 256-byte token row)
//   B: lane = 8 channels of a token, 16 x 16-byte loads per lane and tile (each instruction covers 4 token rows = 1 KB)
// tiles per wave = 8 (one slab of a head), 2 workgroups of 4 waves per head -- k_main_kernel's grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const uint16_t* __restrict__ x, uint32_t* __restrict__ out, int T, int tiles_per_slab) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bh = blockIdx.y;
    const int tile_lo = blockIdx.x * tiles_per_slab, tile_hi = tile_lo + tiles_per_slab;
    uint32_t acc = 0;
    for (int tile = tile_lo + wave; tile < tile_hi; tile += 4) {
        if (MODE == 0) {
            const uint32_t* xw = (const uint32_t*)(x + (bh * T + (int64_t)tile * 64) * 128) + lane;
            uint32_t r[64];
#pragma unroll
            for (int i = 0; i < 64; i++) r[i] = xw[i * 64];
#pragma unroll
            for (int i = 0; i < 64; i++) acc ^= r[i];
        } else {
            const uint4* xw = (const uint4*)(x + (bh * T + (int64_t)tile * 64) * 128) + lane;
            uint4 r[16];
#pragma unroll
            for (int i = 0; i < 16; i++) r[i] = xw[i * 64];
#pragma unroll
            for (int i = 0; i < 16; i++) acc ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
        }
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

int main() {
    const int BH = 1024, T = 4096;
    uint16_t* x; uint32_t* out;
    hipMalloc(&x, (size_t)BH * T * 128 * 2);
    hipMalloc(&out, 4096);
    hipMemset(x, 1, (size_t)BH * T * 128 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++)
        for (int nslab : {2, 4, 8}) {
            const int tps = T / 64 / nslab;
            auto launch = [&]() {
                dim3 g(nslab, BH);
                if (mode == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, x, out, T, tps);
                else hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, x, out, T, tps);
            };
            for (int i = 0; i < 3; i++) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("mode %c slabs %d: %.3f ms  %.0f GB/s read\n", "AB"[mode], nslab, ms, (double)BH * T * 256 / ms / 1e6);
        }
    return 0;
}
