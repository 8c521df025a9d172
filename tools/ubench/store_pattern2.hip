// Store-pattern microbenchmark 2 (GPU box: hipcc --offload-arch=gfx950 -O3 store_pattern2.hip -o /tmp/sp2 && /tmp/sp2):
// the decompress kernel's shape (256-thread workgroups, 16 rows of 4096 fp16 each, a lane stores 32 contiguous bytes per row, a
// 4-byte code word per lane and row in front of the store) with the knobs that could explain why the real kernel writes at
// 2.7 TB/s: row order inside the block (ROT: start row = block index mod 16, in steps of one row), how far ahead the code word is
// loaded (DEPTH 0: no load, 1: dependent, 2: one row ahead, 3: two rows ahead, 4: all 16 up front), nontemporal stores (NT),
// resident blocks per CU (launch bound / dynamic LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int DEPTH, int ROT, int NT>
__global__ __launch_bounds__(256) void k(uint4* __restrict__ out, const uint32_t* __restrict__ code) {
    extern __shared__ uint32_t pad[];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int rot = ROT == 1 ? (int)(blockIdx.x & 15) : ROT == 2 ? 4 * (int)((blockIdx.x ^ (blockIdx.x >> 2)) & 3) : 0;
    auto row_of = [&](int r) { return row0 + ((r + rot) & 15); };
    auto ld = [&](int r) { return code[row_of(r) * 256 + tid]; };
    uint32_t c[16];
    if (DEPTH == 4) {
#pragma unroll
        for (int r = 0; r < 16; r++) c[r] = ld(r);
    }
    uint32_t n0 = 0, n1 = 0;
    if (DEPTH == 2 || DEPTH == 3) n0 = ld(0);
    if (DEPTH == 3) n1 = ld(1);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        uint32_t cc = 0x1234u + r;
        if (DEPTH == 1) cc = ld(r);
        if (DEPTH == 2) { cc = n0; if (r + 1 < 16) n0 = ld(r + 1); }
        if (DEPTH == 3) { cc = n0; n0 = n1; if (r + 2 < 16) n1 = ld(r + 2); }
        if (DEPTH == 4) cc = c[r];
        const uint4 v0 = make_uint4(cc, cc + 1, cc + 2, cc + 3), v1 = make_uint4(cc + 4, cc + 5, cc + 6, cc + 7);
        uint4* rowp = out + row_of(r) * 512;
        if (NT) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store((u4){v0.x, v0.y, v0.z, v0.w}, (u4*)&rowp[tid * 2]);
            __builtin_nontemporal_store((u4){v1.x, v1.y, v1.z, v1.w}, (u4*)&rowp[tid * 2 + 1]);
        } else {
            rowp[tid * 2] = v0;
            rowp[tid * 2 + 1] = v1;
        }
    }
    if (tid == 1000) pad[0] = 1;
}

template <int DEPTH, int ROT, int NT>
void run(uint4* out, const uint32_t* code, int lds_kb, const char* what) {
    const int64_t n_rows = 131072;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kf = k<DEPTH, ROT, NT>;
    hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    auto launch = [&]() { hipLaunchKernelGGL(kf, dim3((unsigned)(n_rows / 16)), dim3(256), (size_t)lds_kb * 1024, 0, out, code); };
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("depth %d rot %d nt %d lds %2d KB (%s): %.3f ms  %.0f GB/s written\n", DEPTH, ROT, NT, lds_kb, what, ms, n_rows * 8192.0 / ms / 1e6);
}

int main() {
    const int64_t n_rows = 131072;
    uint4* out; uint32_t* code;
    hipMalloc(&out, n_rows * 8192);
    hipMalloc(&code, n_rows * 1024);
    hipMemset(code, 1, n_rows * 1024);
    for (int lds : {0, 35, 52}) {      // 35 KB: 4 blocks per CU, 52 KB: 3
        run<0, 0, 0>(out, code, lds, "stores only");
        run<0, 1, 0>(out, code, lds, "stores only, rotated by one row");
        run<0, 2, 0>(out, code, lds, "stores only, rotated by four rows");
        run<0, 0, 1>(out, code, lds, "stores only, nontemporal");
        run<0, 1, 1>(out, code, lds, "stores only, rotated, nontemporal");
        run<1, 0, 0>(out, code, lds, "dependent load");
        run<1, 1, 0>(out, code, lds, "dependent load, rotated");
        run<2, 0, 0>(out, code, lds, "one row ahead");
        run<2, 1, 0>(out, code, lds, "one row ahead, rotated");
        run<2, 2, 0>(out, code, lds, "one row ahead, rotated by four");
        run<2, 1, 1>(out, code, lds, "one row ahead, rotated, nontemporal");
        run<3, 1, 0>(out, code, lds, "two rows ahead, rotated");
        run<4, 0, 0>(out, code, lds, "all loads up front");
        run<4, 1, 0>(out, code, lds, "all loads up front, rotated");
        run<4, 1, 1>(out, code, lds, "all loads up front, rotated, nontemporal");
    }
    return 0;
}
