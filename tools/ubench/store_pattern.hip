// Store-pattern microbenchmark (GPU box: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o /tmp/sp && /tmp/sp):
// 1 GiB of fp16 written by 256-thread workgroups that own 16 rows of 4096 elements each (the decompress kernels' shape),
//   A: a lane stores its 32 contiguous bytes as two 16-byte stores (each instruction covers 2 KB at 50 % density)
//   B: the same bytes, but instruction 0 writes the first KB of the wave's 2 KB and instruction 1 the second (full lines)
//   C: like A with a dependent 4-byte load per row in front (the code word)
//   D: like A, but a block's rows are spread over the tensor (row = block + j * blocks) instead of consecutive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* __restrict__ out, const uint32_t* __restrict__ code, int rows_per_block) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
    for (int r = 0; r < rows_per_block; r++) {
        uint32_t c = 0x1234u + r;
        if (MODE == 2) c = code[(row0 + r) * 256 + tid];
        const uint4 v0 = make_uint4(c, c + 1, c + 2, c + 3), v1 = make_uint4(c + 4, c + 5, c + 6, c + 7);
        uint4* rowp = out + (MODE == 3 ? (int64_t)blockIdx.x + (int64_t)r * gridDim.x : row0 + r) * 512;   // 4096 halves = 512 uint4
        if (MODE == 1) {
            rowp[wave * 128 + lane] = v0;
            rowp[wave * 128 + 64 + lane] = v1;
        } else {
            rowp[tid * 2] = v0;
            rowp[tid * 2 + 1] = v1;
        }
    }
}

int main() {
    const int64_t n_rows = 131072;
    uint4* out; uint32_t* code;
    hipMalloc(&out, n_rows * 8192);
    hipMalloc(&code, n_rows * 1024);
    hipMemset(code, 1, n_rows * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; mode++)
        for (int rpb : {1, 4, 16, 64}) {
            auto launch = [&]() {
                dim3 g((unsigned)(n_rows / rpb));
                if (mode == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, out, code, rpb);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, out, code, rpb);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, out, code, rpb);
                else hipLaunchKernelGGL(k<3>, g, dim3(256), 0, 0, out, code, rpb);
            };
            for (int i = 0; i < 3; i++) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("mode %c rows/block %2d: %.3f ms  %.0f GB/s written\n", "ABCD"[mode], rpb, ms, n_rows * 8192.0 / ms / 1e6);
        }
    return 0;
}
