// Does the blockIdx -> address mapping of a streaming kernel matter?  1 GiB read once (16 bytes per lane, 4 KB per wave-step, 3 steps
// in flight per wave, 4 waves per workgroup) with the wave-steps dealt
//   rr      round-robin over ALL (workgroup, wave) pairs (tools/ubench/mall_curve.hip: 6.1 - 6.3 TB/s),
//   blkN    in contiguous runs of N KB per workgroup, the runs themselves in blockIdx order (N = 64 .. 1024; the chain kernels give a
//           workgroup 256 KB - 1 MB of one head: tools/ubench/stream_pattern.hip reads that shape at 5.1 TB/s),
// for grids of 1024 .. 8192 workgroups.  Compile on the GPU box: hipcc --offload-arch=gfx950 -O3 stream_map.hip -o /tmp/sm && /tmp/sm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int NSTG = 3;

// run_steps: wave-steps (4 KB) per contiguous run of a workgroup; 0 = round-robin over everything
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ x, uint32_t* __restrict__ out, uint64_t nstep, uint32_t run_steps) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint4 st[NSTG][4];
    uint32_t acc = 0;
    auto body = [&](uint64_t first, uint64_t stride, uint64_t end) {
        auto issue = [&](uint64_t step, uint4 (&r)[4]) {
            const uint64_t s = step < end ? step : first;
#pragma unroll
            for (int p = 0; p < 4; p++) r[p] = x[s * 256 + p * 64 + lane];
        };
#pragma unroll
        for (int i = 0; i < NSTG; i++) issue(first + stride * i, st[i]);
        for (uint64_t s0 = first; s0 < end; s0 += stride * NSTG) {
#pragma unroll
            for (int i = 0; i < NSTG; i++) {
                uint4 v = st[i][0];
#pragma unroll
                for (int p = 1; p < 4; p++) { v.x ^= st[i][p].x; v.y ^= st[i][p].y; v.z ^= st[i][p].z; v.w ^= st[i][p].w; }
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
                issue(s0 + stride * (i + NSTG), st[i]);
            }
        }
    };
    if (run_steps == 0) body((uint64_t)blockIdx.x * 4 + wave, (uint64_t)gridDim.x * 4, nstep);
    else {
        // runs r = blockIdx.x, blockIdx.x + gridDim.x, ...: each a contiguous run of run_steps wave-steps, the 4 waves interleaved inside
        for (uint64_t r = blockIdx.x; r * run_steps < nstep; r += gridDim.x) {
            const uint64_t lo = r * run_steps, hi = lo + run_steps < nstep ? lo + run_steps : nstep;
            body(lo + wave, 4, hi);
        }
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

int main() {
    const size_t GiB = (size_t)1 << 30;
    uint4* x; uint32_t* out;
    hipMalloc(&x, GiB); hipMalloc(&out, 4096);
    hipMemset(x, 1, GiB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint64_t nstep = GiB / 4096;
    printf("| workgroups | rr | blk64K | blk256K | blk1M |\n|---:|---:|---:|---:|---:|\n");
    for (int grid : {1024, 2048, 4096, 8192}) {
        printf("| %d |", grid);
        for (uint32_t run : {0u, 16u, 64u, 256u}) {
            for (int i = 0; i < 2; i++) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, x, out, nstep, run);
            hipEventRecord(e0);
            for (int i = 0; i < 8; i++) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, x, out, nstep, run);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf(" %.0f |", (double)GiB * 8 / ms / 1e6);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
