// gemv.hip -- decompress-into-attention GEMV over the packed cache (gfx950, wave64).
//
//   out[ba, n] = sum_k a[ba, k] * (scale[bw, k, n/g] * code[bw, k, n] + zero[bw, k, n/g])
//
// The packed dimension is the OUTPUT dimension ("outer dim", gemv_cuda.cu:255-263), so one int32 word holds
// 32/bits different outputs of the same reduction index k.  A lane owns VEC consecutive words of a row
// (VEC=4 -> one 16-byte load) and keeps VEC*32/bits fp32 accumulators; lanes of a block tile
// (chunk-lanes x row-lanes); rows are strided over row-lanes and blocks (split-K).  No cross-lane traffic
// in the main loop; one xor-shuffle butterfly over the row-lanes at the end.
//
// Per element the arithmetic is  acc += (a*scale) * code  and the  a*zero  terms are summed once per
// (row, group) -- same sums as gemv_cuda.cu:331-335, re-associated; fp32 throughout, one fp16 rounding.
#include "common.h"

template <int BITS, typename ST, int VEC>
__global__ __launch_bounds__(64) void gemv_outer_kernel(const uint16_t* __restrict__ a, const uint32_t* __restrict__ qB,
                                                         const ST* __restrict__ scale, const ST* __restrict__ zero,
                                                         int n_rep, int K, int NW, int NG, int group, int64_t ldq,
                                                         int64_t lds, int cw_log2, int rows_per_split,
                                                         uint16_t* __restrict__ out16, float* __restrict__ part) {
    constexpr int CPW = 32 / BITS;
    constexpr int NACC = VEC * CPW;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const int CW = 1 << cw_log2;         // chunk-lanes per block (power of two, <= blockDim.x)
    const int RL = GEAR_WAVE >> cw_log2;  // row-lanes per block (block = one wave)
    const int cl = threadIdx.x & (CW - 1);
    const int rl = threadIdx.x >> cw_log2;
    const int NC = (NW + VEC - 1) / VEC;  // chunks per row
    const int chunk = blockIdx.x * CW + cl;
    const int64_t ba = blockIdx.z;
    const int64_t bw = ba / n_rep;
    const int k_begin = blockIdx.y * rows_per_split;
    const int k_end = min(K, k_begin + rows_per_split);
    const bool chunk_ok = chunk < NC;
    const int w0 = chunk * VEC;  // first word of this lane

    float acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = 0.0f;
    float zacc[VEC];  // sum_k a_k * zero_k(group of word v)
#pragma unroll
    for (int v = 0; v < VEC; v++) zacc[v] = 0.0f;

    const uint16_t* arow = a + ba * K;
    if (chunk_ok) {
        for (int k = k_begin + rl; k < k_end; k += RL) {
            const uint32_t* wp = qB + (bw * K + k) * ldq + w0;
            uint32_t words[VEC];
            if (VEC == 4) {
                if (w0 + 3 < NW) {
                    uint4 t = *(const uint4*)wp;
                    words[0] = t.x; words[1] = t.y; words[2] = t.z; words[3] = t.w;
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; v++) words[v] = (w0 + v < NW) ? wp[v] : 0u;
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; v++) words[v] = (w0 + v < NW) ? wp[v] : 0u;
            }
            const float av = h2f_bits(arow[k]);
            const ST* sp = scale + (bw * K + k) * lds;
            const ST* zp = zero + (bw * K + k) * lds;
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                int g = min(((w0 + v) * CPW) / group, NG - 1);
                float sa = ld_st<ST>(sp + g) * av;
                zacc[v] = fmaf(ld_st<ST>(zp + g), av, zacc[v]);
                uint32_t wv = words[v];
#pragma unroll
                for (int j = 0; j < CPW; j++) {
                    float c = (float)((wv >> (BITS * j)) & MASK);
                    acc[v * CPW + j] = fmaf(sa, c, acc[v * CPW + j]);
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; v++)
#pragma unroll
        for (int j = 0; j < CPW; j++) acc[v * CPW + j] += zacc[v];

    // reduce over the row-lanes (blocks are exactly one wave; row-lane bits sit above the chunk-lane bits)
    for (int m = CW; m < GEAR_WAVE; m <<= 1) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] += __shfl_xor(acc[j], m, GEAR_WAVE);
    }
    if (!chunk_ok || rl != 0) return;
    const int N = NW * CPW;
    const int n0 = w0 * CPW;
    if (part) {
        float* pp = part + ((int64_t)blockIdx.y * gridDim.z + ba) * N + n0;
#pragma unroll
        for (int j = 0; j < NACC; j++)
            if (n0 + j < N) pp[j] = acc[j];
    } else {
        uint16_t* op = out16 + ba * N + n0;
#pragma unroll
        for (int j = 0; j < NACC; j++)
            if (n0 + j < N) op[j] = f2h_bits(acc[j]);
    }
}

__global__ __launch_bounds__(256) void gemv_reduce_kernel(const float* __restrict__ part, int splits, int64_t total,
                                                          uint16_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = 0.0f;
    for (int p = 0; p < splits; p++) s += part[(int64_t)p * total + i];
    out[i] = f2h_bits(s);
}

namespace {
struct GemvPlan {
    int vec, cw_log2, threads, grid_x, splits, rows_per_split;
};

GemvPlan plan_gemv(int64_t BA, int K, int N, int bits) {
    GemvPlan p;
    int cpw = 32 / bits;
    int NW = N / cpw;
    p.vec = (NW % 4 == 0) ? 4 : 1;
    int NC = (NW + p.vec - 1) / p.vec;
    // chunk-lanes: power of two, at most 64 (so that the row-lane butterfly stays inside a wave)
    int cw = 1, lg = 0;
    while (cw < NC && cw < 64) { cw <<= 1; lg++; }
    p.cw_log2 = lg;
    p.threads = 64;  // one wave per block: plenty of blocks, no LDS pass
    int RL = p.threads / cw;
    p.grid_x = (NC + cw - 1) / cw;
    // split K so that the grid has >= ~1024 blocks but every row-lane still loops >= 4 times
    int64_t blocks = (int64_t)p.grid_x * BA;
    int max_splits = K / (RL * 4);
    if (max_splits < 1) max_splits = 1;
    int want = (int)((1024 + blocks - 1) / blocks);
    if (want < 1) want = 1;
    p.splits = want < max_splits ? want : max_splits;
    if (p.splits > 64) p.splits = 64;
    p.rows_per_split = (K + p.splits - 1) / p.splits;
    // keep splits exact
    p.splits = (K + p.rows_per_split - 1) / p.rows_per_split;
    return p;
}
}  // namespace

extern "C" size_t gear_gemv_outer_workspace(int64_t BA, int K, int N, int bits) {
    if (bits != 2 && bits != 4) return 0;
    GemvPlan p = plan_gemv(BA, K, N, bits);
    return p.splits > 1 ? sizeof(float) * (size_t)p.splits * (size_t)BA * (size_t)N : 0;
}

extern "C" int gear_gemv_outer(const void* a, const void* qB, const void* scale, const void* zero, int64_t BA, int n_rep,
                               int K, int N, int group, int bits, int mode, int64_t ldq, int64_t lds, void* out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4, "gear_gemv_outer: bits must be 2 or 4 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_gemv_outer: bad mode %d", mode);
    GEAR_CHECK_ARG(BA > 0 && K > 0 && N > 0, "gear_gemv_outer: empty problem");
    GEAR_CHECK_ARG(n_rep >= 1 && BA % n_rep == 0, "gear_gemv_outer: BA=%lld not divisible by n_rep=%d", (long long)BA, n_rep);
    int cpw = 32 / bits;
    GEAR_CHECK_ARG(N % cpw == 0, "gear_gemv_outer: N=%d must be a multiple of %d", N, cpw);
    GEAR_CHECK_ARG(group > 0 && group % cpw == 0, "gear_gemv_outer: group %d must be a multiple of %d", group, cpw);
    GEAR_CHECK_ARG(BA <= 65535, "gear_gemv_outer: batch*heads=%lld exceeds 65535", (long long)BA);
    GEAR_CHECK_ARG(a && qB && scale && zero && out, "gear_gemv_outer: null pointer");
    int NW = N / cpw;
    int NG = (N + group - 1) / group;
    if (ldq == 0) ldq = NW;
    if (lds == 0) lds = NG;
    GemvPlan p = plan_gemv(BA, K, N, bits);
    GEAR_CHECK_ARG(p.vec == 1 || ldq % 4 == 0, "gear_gemv_outer: ldq=%lld must be a multiple of 4 words", (long long)ldq);
    float* part = nullptr;
    if (p.splits > 1) {
        size_t need = sizeof(float) * (size_t)p.splits * (size_t)BA * (size_t)N;
        GEAR_CHECK_ARG(workspace && workspace_bytes >= need, "gear_gemv_outer: workspace too small (%zu < %zu)", workspace_bytes, need);
        part = (float*)workspace;
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 block(p.threads), grid(p.grid_x, p.splits, (unsigned)BA);
#define GO(B, STT, V)                                                                                               \
    hipLaunchKernelGGL((gemv_outer_kernel<B, STT, V>), grid, block, 0, st, (const uint16_t*)a, (const uint32_t*)qB,  \
                       (const STT*)scale, (const STT*)zero, n_rep, K, NW, NG, group, ldq, lds, p.cw_log2,            \
                       p.rows_per_split, (uint16_t*)out, part)
    if (mode == 0) {
        if (bits == 2) { if (p.vec == 4) GO(2, uint16_t, 4); else GO(2, uint16_t, 1); }
        else           { if (p.vec == 4) GO(4, uint16_t, 4); else GO(4, uint16_t, 1); }
    } else {
        if (bits == 2) { if (p.vec == 4) GO(2, float, 4); else GO(2, float, 1); }
        else           { if (p.vec == 4) GO(4, float, 4); else GO(4, float, 1); }
    }
#undef GO
    GEAR_CHECK_LAUNCH("gear_gemv_outer");
    if (p.splits > 1) {
        int64_t total = BA * (int64_t)N;
        hipLaunchKernelGGL(gemv_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, p.splits, total,
                           (uint16_t*)out);
        GEAR_CHECK_LAUNCH("gear_gemv_outer(reduce)");
    }
    return 0;
}
