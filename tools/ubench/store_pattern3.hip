// Store-pattern microbenchmark 3 (GPU box: hipcc --offload-arch=gfx950 -O3 store_pattern3.hip -o /tmp/sp3 && /tmp/sp3):
// why does a 4-byte load per lane and row in front of the decompressor's 32-byte store cost a third of the write rate
// (store_pattern2: 6.0 TB/s stores only, 3.5-4.5 with the load)?  Variants of "one row ahead":
//   0 stores only                       1 one row ahead (code word from a 128 MB buffer)
//   2 one row ahead, code from a 16 KB buffer (always an L2 hit)
//   3 one row ahead, full-line stores (instruction 0 writes the first KB of the wave's 2 KB, instruction 1 the second)
//   4 the block's 16 KB of code loaded up front with 16-byte loads (4 instructions per wave) into LDS, rows read it from there
//   5 like 4 plus full-line stores
//   6 a fifth wave loads the code of the NEXT block-row group into LDS while four waves only read LDS and store
//   7 code word loaded through the scalar cache (one s_load_dwordx4 per wave and row, wave-uniform data)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int V>
__global__ __launch_bounds__(V == 6 ? 320 : 256) void k(uint4* __restrict__ out, const uint32_t* __restrict__ code, int mask, const void* __restrict__ aux) {
    __shared__ uint32_t lds[2][16 * 256];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    auto store_row = [&](int r, uint32_t cc) {
        const uint4 v0 = make_uint4(cc, cc + 1, cc + 2, cc + 3), v1 = make_uint4(cc + 4, cc + 5, cc + 6, cc + 7);
        uint4* rowp = out + (row0 + r) * 512;
        if (V == 8) {
            // what v_permlane32_swap gives the decompressor for free: instruction 0 carries the lower 32 lanes' two halves
            // (lanes < 32 write the even 16-byte slots of the wave's first KB, lanes >= 32 the odd ones), instruction 1 the upper
            const int wave = tid >> 6, lane = tid & 63;
            rowp[wave * 128 + (lane & 31) * 2 + (lane >> 5)] = v0;
            rowp[wave * 128 + 64 + (lane & 31) * 2 + (lane >> 5)] = v1;
        } else if (V == 3 || V == 5) {
            const int wave = tid >> 6, lane = tid & 63;
            rowp[wave * 128 + lane] = v0;
            rowp[wave * 128 + 64 + lane] = v1;
        } else {
            rowp[tid * 2] = v0;
            rowp[tid * 2 + 1] = v1;
        }
    };
    if (V == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) store_row(r, 0x1234u + r);
    } else if (V == 1 || V == 2 || V == 3 || V == 8) {
        auto ld = [&](int r) { return code[(((row0 + r) * 256) & (int64_t)mask) + tid]; };
        uint32_t n0 = ld(0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t cc = n0;
            if (r + 1 < 16) n0 = ld(r + 1);
            store_row(r, cc);
        }
    } else if (V == 10 || V == 11) {
        // the bare decompressor: code word + the group's scale and zero point (fp32, four lanes share a group) one row ahead
        const float* sc = (const float*)aux;
        const float* mn = sc + (int64_t)131072 * 64;
        const int NR = V == 11 ? 8 : 16;
        const int64_t rb = V == 11 ? (int64_t)blockIdx.x * 8 : row0;
        auto ldc = [&](int r) { return code[(rb + r) * 256 + tid]; };
        auto lds_ = [&](int r) { return sc[(rb + r) * 64 + (tid >> 2)]; };
        auto ldm = [&](int r) { return mn[(rb + r) * 64 + (tid >> 2)]; };
        uint32_t n0 = ldc(0); float s0 = lds_(0), m0 = ldm(0);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const uint32_t cc = n0; const float s1 = s0, m1 = m0;
            if (r + 1 < NR) { n0 = ldc(r + 1); s0 = lds_(r + 1); m0 = ldm(r + 1); }
            const uint32_t x = cc + __float_as_uint(s1 * 3.0f + m1);
            const uint4 v0 = make_uint4(x, x + 1, x + 2, x + 3), v1 = make_uint4(x + 4, x + 5, x + 6, x + 7);
            uint4* rowp = out + (rb + r) * 512;
            const int wave = tid >> 6, lane = tid & 63;
            rowp[wave * 128 + (lane & 31) * 2 + (lane >> 5)] = v0;
            rowp[wave * 128 + 64 + (lane & 31) * 2 + (lane >> 5)] = v1;
        }
    } else if (V == 4 || V == 5) {
        const uint4* src = (const uint4*)(code + row0 * 256);
#pragma unroll
        for (int i = 0; i < 4; i++) ((uint4*)lds[0])[tid + 256 * i] = src[tid + 256 * i];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) store_row(r, lds[0][r * 256 + tid]);
    } else if (V == 6) {
        // blocks of 16 rows, each handled as 4 groups of 4 rows; wave 4 stages group g + 1 while waves 0-3 store group g
        const int wave = tid >> 6, lane = tid & 63;
        const uint4* src = (const uint4*)(code + row0 * 256);
        if (wave == 4) {
#pragma unroll
            for (int i = 0; i < 4; i++) ((uint4*)lds[0])[lane + 64 * i] = src[lane + 64 * i];
        }
        __syncthreads();
        for (int g = 0; g < 4; g++) {
            if (wave == 4) {
                if (g + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 4; i++) ((uint4*)lds[(g + 1) & 1])[lane + 64 * i] = src[(g + 1) * 256 + lane + 64 * i];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) store_row(4 * g + r, lds[g & 1][r * 256 + tid]);
            }
            __syncthreads();
        }
    } else if (V == 7) {
        const uint4* src = (const uint4*)(code + (row0 * 256 + (tid >> 6) * 64));
        const uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)src);
        const uint32_t phi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)src >> 32));
        const uint4* us = (const uint4*)(((uintptr_t)phi << 32) | plo);
        uint4 n0 = us[0];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint4 cc = n0;
            if (r + 1 < 16) n0 = us[(r + 1) * 64];
            store_row(r, cc.x + cc.y + cc.z + cc.w + tid);
        }
    }
}

template <int V>
void run(uint4* out, const uint32_t* code, const char* what, const void* aux = nullptr) {
    const int64_t n_rows = 131072;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int mask = V == 2 ? 4095 : 0x7FFFFFFF;
    auto launch = [&]() { hipLaunchKernelGGL(k<V>, dim3((unsigned)(n_rows / (V == 11 ? 8 : 16))), dim3(V == 6 ? 320 : 256), 0, 0, out, code, mask, aux); };
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("variant %d (%s): %.3f ms  %.0f GB/s written\n", V, what, ms, n_rows * 8192.0 / ms / 1e6);
    fflush(stdout);
}

int main() {
    const int64_t n_rows = 131072;
    uint4* out; uint32_t* code;
    hipMalloc(&out, n_rows * 8192);
    hipMalloc(&code, n_rows * 1024);
    hipMemset(code, 1, n_rows * 1024);
    void* aux; hipMalloc(&aux, n_rows * 64 * 4 * 2); hipMemset(aux, 0, n_rows * 64 * 4 * 2);
    for (int rep = 0; rep < 2; rep++) {
        run<0>(out, code, "stores only");
        run<1>(out, code, "one row ahead");
        run<2>(out, code, "one row ahead, code always in L2");
        run<3>(out, code, "one row ahead, full-line stores");
        run<4>(out, code, "code of the block up front through LDS");
        run<5>(out, code, "code up front through LDS, full-line stores");
        run<6>(out, code, "loader wave + four storing waves");
        run<7>(out, code, "code through the scalar cache");
        run<8>(out, code, "one row ahead, full lines per instruction from lanes l and l + 32");
        run<10>(out, code, "8 + scale and zero point loads", aux);
        run<11>(out, code, "10 with 8 rows per block", aux);
    }
    return 0;
}
