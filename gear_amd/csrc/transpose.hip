// transpose.hip -- batched fp16 [bh, R, C] -> [bh, C, R] through a 64x64 LDS tile (gfx950).
// K arrives token-major [T, D] from the model but is quantized per channel along T; the reference's caller does
// key_states.transpose(2, 3).contiguous() with a generic strided copy (modeling_llamagear.py:268, :403).  Both
// sides here move 16 bytes per lane; the tile pitch (66 halfs = 33 dwords) keeps the column reads conflict-free.
#include "common.h"

namespace {
constexpr int TT = 64, TP = 66;

__global__ __launch_bounds__(256) void transpose_f16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                            int R, int C) {
    __shared__ uint16_t tile[TT][TP];
    const int64_t bh = blockIdx.z;
    const int r0 = blockIdx.y * TT, c0 = blockIdx.x * TT;
    const uint16_t* xb = x + bh * (int64_t)R * C;
    uint16_t* yb = y + bh * (int64_t)R * C;
    const int l8 = threadIdx.x & 7, rr = threadIdx.x >> 3;  // 8 lanes x 8 halfs per tile row, 32 rows per pass
#pragma unroll
    for (int p = 0; p < 2; p++) {
        int r = r0 + rr + 32 * p, c = c0 + l8 * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < R && c < C) v = *(const uint4*)(xb + (int64_t)r * C + c);
        uint32_t* d = (uint32_t*)&tile[rr + 32 * p][l8 * 8];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; p++) {
        int c = rr + 32 * p;          // output row = input column
        int rb = l8 * 8;              // 8 consecutive input rows -> 8 consecutive output columns
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            w[j] = (uint32_t)tile[rb + 2 * j][c] | ((uint32_t)tile[rb + 2 * j + 1][c] << 16);
        if (c0 + c < C && r0 + rb < R) *(uint4*)(yb + (int64_t)(c0 + c) * R + r0 + rb) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
}  // namespace

extern "C" int gear_transpose_f16(const void* x, int64_t bh, int R, int C, void* y, void* stream) {
    GEAR_CHECK_ARG(x && y, "gear_transpose_f16: null pointer");
    GEAR_CHECK_ARG(bh > 0 && bh <= 65535 && R > 0 && C > 0, "gear_transpose_f16: bad shape");
    GEAR_CHECK_ARG(R % 8 == 0 && C % 8 == 0, "gear_transpose_f16: both dims must be multiples of 8 (R=%d C=%d)", R, C);
    dim3 grid((C + TT - 1) / TT, (R + TT - 1) / TT, (unsigned)bh);
    hipLaunchKernelGGL(transpose_f16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, R, C);
    GEAR_CHECK_LAUNCH("gear_transpose_f16");
    return 0;
}
