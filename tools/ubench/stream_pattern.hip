// Streaming-read microbenchmark for the per-head reduction kernels (Gram matrix, K selection): 1 GiB fp16 [BH = 1024][T = 4096][128]
// read once.  grid (nslab, BH); a workgroup of 4 waves sweeps its slab of one head, wave w taking the 16-token steps w, w + 4, ...
// (4 KB contiguous per step, one 16-byte load per lane and 4 token rows per instruction), NSTG steps of loads in flight per wave.
// The number of resident workgroups per CU is forced by a dynamic LDS allocation.  Compile on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 stream_pattern.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NSTG, int WORK>   // WORK: dependent VALU instructions per step standing in for the consumer (0, 64, 256)
__global__ __launch_bounds__(256) void k(const uint16_t* __restrict__ x, uint32_t* __restrict__ out, int T, int tok_per_slab) {
    extern __shared__ uint32_t dummy[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bh = blockIdx.y;
    const int s_lo = blockIdx.x * tok_per_slab, nstep = tok_per_slab / 16;
    const char* base = (const char*)(x + (bh * T + s_lo) * 128);
    uint4 st[NSTG][4];
    auto issue = [&](int step, uint4 (&r)[4]) {
        const int s = step < nstep ? step : 0;
#pragma unroll
        for (int p = 0; p < 4; p++) r[p] = *(const uint4*)(base + (size_t)((uint32_t)(16 * s + 4 * p) * 256u + (uint32_t)lane * 16u));
    };
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NSTG; i++) issue(wave + 4 * i, st[i]);
    for (int s0 = wave; s0 < nstep; s0 += 4 * NSTG) {
#pragma unroll
        for (int i = 0; i < NSTG; i++) {
            uint32_t v = 0;
#pragma unroll
            for (int p = 0; p < 4; p++) v ^= st[i][p].x ^ st[i][p].y ^ st[i][p].z ^ st[i][p].w;
#pragma unroll
            for (int q = 0; q < WORK; q++) v = v * 1664525u + 1013904223u;
            acc ^= v;
            issue(s0 + 4 * (i + NSTG), st[i]);
        }
    }
    if (acc == 0x12345678u) out[tid] = acc + dummy[0];
}

int main() {
    const int BH = 1024, T = 4096;
    uint16_t* x; uint32_t* out;
    hipMalloc(&x, (size_t)BH * T * 128 * 2);
    hipMalloc(&out, 4096);
    hipMemset(x, 1, (size_t)BH * T * 128 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kfn, const char* name, int nslab, int wg_per_cu) {
        const size_t lds = 160 * 1024 / wg_per_cu - 1024;
        hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int tps = T / nslab;
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kfn, dim3(nslab, BH), dim3(256), lds, 0, x, out, T, tps);
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kfn, dim3(nslab, BH), dim3(256), lds, 0, x, out, T, tps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-18s nslab %d  wg/CU %d : %.3f ms  %.0f GB/s\n", name, nslab, wg_per_cu, ms, (double)BH * T * 256 / ms / 1e6);
    };
    for (int wgcu : {2, 3, 4, 8})
        for (int nslab : {1, 4}) {
            run(k<2, 0>, "stg2 work0", nslab, wgcu);
            run(k<4, 0>, "stg4 work0", nslab, wgcu);
            run(k<8, 0>, "stg8 work0", nslab, wgcu);
            run(k<3, 256>, "stg3 work256", nslab, wgcu);
        }
    return 0;
}
