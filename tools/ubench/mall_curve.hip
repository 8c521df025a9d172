// Read bandwidth against working set: is a tensor that was just streamed still in the 256 MB Infinity Cache (MALL) when the next
// kernel of a chain reads it again, and how fast does it come back?  (the measurement that decides
// between a layer-group-ordered chain and an on-chip-resident kernel.)
//
// One launch reads `ws` bytes once with the chains' access pattern (16 bytes per lane, 4 KB contiguous per wave-step, 3 steps in
// flight per wave, 2048 workgroups of 4 waves striding the buffer).  For every working set 16 MiB .. 1 GiB the launch is repeated
// back to back (same buffer) until 16 GiB have been read; GB/s = bytes / event time.  Second table: a PAIR of kernels per
// working set -- "producer" (reads ws, writes ws / 8: the shape of select / main) then "consumer" (reads ws again) -- the order of
// a chain run one layer group at a time; third: a kernel that reads ws AND writes ws (error matrix out) followed by the reader.
// Compile on the GPU box:  hipcc --offload-arch=gfx950 -O3 mall_curve.hip -o /tmp/mall && /tmp/mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int NSTG = 3;

// WR: 0 read only; 1 write 1 / 8 of what is read (a uint2 per 4 uint4 read... one 16-byte store per 8 loads); 2 write as much as read
template <int WR>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ x, uint4* __restrict__ y, uint32_t* __restrict__ out,
                                          uint64_t nvec) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a wave-step = 4 instructions x 64 lanes x 16 bytes = 4 KB contiguous; steps are dealt round-robin to (block, wave)
    const uint64_t nstep = nvec / 256;                                 // 256 vectors per wave-step
    const uint64_t first = (uint64_t)blockIdx.x * 4 + wave, stride = (uint64_t)gridDim.x * 4;
    uint4 st[NSTG][4];
    auto issue = [&](uint64_t step, uint4 (&r)[4]) {
        const uint64_t s = step < nstep ? step : first;
#pragma unroll
        for (int p = 0; p < 4; p++) r[p] = x[s * 256 + p * 64 + lane];
    };
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NSTG; i++) issue(first + stride * i, st[i]);
    for (uint64_t s0 = first; s0 < nstep; s0 += stride * NSTG) {
#pragma unroll
        for (int i = 0; i < NSTG; i++) {
            const uint64_t s = s0 + stride * i;
            uint4 v = st[i][0];
#pragma unroll
            for (int p = 1; p < 4; p++) { v.x ^= st[i][p].x; v.y ^= st[i][p].y; v.z ^= st[i][p].z; v.w ^= st[i][p].w; }
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            if (s < nstep) {
                if (WR == 1) { if ((lane & 1) == 0) y[(s * 256) / 8 + (lane >> 1)] = v; }
                if (WR == 2) {
#pragma unroll
                    for (int p = 0; p < 4; p++) y[s * 256 + p * 64 + lane] = st[i][p];
                }
            }
            issue(s + stride * NSTG, st[i]);
        }
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

int main() {
    const size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;
    uint4 *x, *y; uint32_t* out;
    hipMalloc(&x, GiB); hipMalloc(&y, GiB); hipMalloc(&out, 4096);
    hipMemset(x, 1, GiB); hipMemset(y, 2, GiB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const std::vector<size_t> sets = {16, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 768, 1024};
    printf("| working set MiB | re-read GB/s (us per pass) | producer(r ws, w ws/8) GB/s of bytes moved | its reader GB/s | producer(r ws, w ws) GB/s | its reader (reads what was WRITTEN) GB/s |\n|---:|---:|---:|---:|---:|---:|\n");
    for (size_t ws_mib : sets) {
        const size_t ws = ws_mib * MiB;
        const uint64_t nvec = ws / 16;
        const int reps = (int)((16 * GiB) / ws);
        const int grid = 2048;
        float ms;
        // (a) the same buffer again and again
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, x, y, out, nvec);
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, x, y, out, nvec);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double a_gbs = (double)ws * reps / ms / 1e6, a_us = ms * 1e3 / reps;
        // (b) producer (read x, write 1/8) then reader of x: per-kernel times through separate event pairs are too coarse for
        // small sets, so time the alternation and the producer alone, and subtract
        auto timed = [&](auto fn, int n) {
            for (int i = 0; i < 2; i++) fn();
            hipEventRecord(e0);
            for (int i = 0; i < n; i++) fn();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            return (double)t / n;
        };
        const int n2 = reps / 2 > 4 ? reps / 2 : 4;
        // rotate the producer over DIFFERENT buffers is not what a chain does: the chain's producer reads the group that the
        // previous consumer just finished with (evicted or not), so "alone" here = producer after producer on the same set
        const double p1 = timed([&] { hipLaunchKernelGGL(rd<1>, dim3(grid), dim3(256), 0, 0, x, y, out, nvec); }, n2);
        const double p1r = timed([&] { hipLaunchKernelGGL(rd<1>, dim3(grid), dim3(256), 0, 0, x, y, out, nvec);
                                       hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, x, y, out, nvec); }, n2);
        const double p2 = timed([&] { hipLaunchKernelGGL(rd<2>, dim3(grid), dim3(256), 0, 0, x, y, out, nvec); }, n2);
        const double p2r = timed([&] { hipLaunchKernelGGL(rd<2>, dim3(grid), dim3(256), 0, 0, x, y, out, nvec);
                                       hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, y, x, out, nvec); }, n2);
        printf("| %zu | %.0f (%.1f) | %.0f | %.0f | %.0f | %.0f |\n", ws_mib, a_gbs, a_us, (double)ws * 1.125 / p1 / 1e6,
               (double)ws / (p1r - p1) / 1e6, (double)ws * 2 / p2 / 1e6, (double)ws / (p2r - p2) / 1e6);
        fflush(stdout);
    }
    // (c) cold reference: 1 GiB buffers alternating (nothing can stay)
    {
        float ms;
        hipEventRecord(e0);
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(rd<0>, dim3(2048), dim3(256), 0, 0, (i & 1) ? y : x, y, out, (uint64_t)(GiB / 16));
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("\ncold stream (two 1 GiB buffers alternating): %.0f GB/s\n", (double)GiB * 8 / ms / 1e6);
    }
    return 0;
}
