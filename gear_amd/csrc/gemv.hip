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

// ---- the reference extension's own argument layout (gemv_cuda.h:13-21, kernels gemv_cuda.cu:264-434): K innermost --------------
// kernel [BW][N / fpi][K] int32, scaling factors / zeros [BW][N / group][K]: what cuda_bmm_fA_qB_outer hands to
// kivi_gemv.gemv_forward_cuda_outer_dim AFTER its per-call transpose(1, 2).contiguous() (matmul.py:205, :215-216).  One wave per
// packed word row: the lanes stride over K (contiguous: coalesced 4-byte and 2-byte loads), fpi accumulators per lane, one DPP
// sum per output at the end.  Same arithmetic as gemv_outer_kernel (scale * a and zero * a in fp32, fp32 accumulation, one fp16
// rounding).  For callers that keep the reference's binding line by line; the native layout (gear_gemv_outer) needs no re-layout.
template <int BITS, typename ST>
__global__ __launch_bounds__(256) void gemv_outer_dim_kernel(const uint16_t* __restrict__ a, const uint32_t* __restrict__ qBt,
                                                             const ST* __restrict__ scale_t, const ST* __restrict__ zero_t, int n_rep,
                                                             int K, int NW, int NG, int group, uint16_t* __restrict__ out16) {
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const int lane = threadIdx.x & 63;
    const int w = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (w >= NW) return;
    const int64_t ba = blockIdx.y, bw = ba / n_rep;
    const int g = min((w * CPW) / group, NG - 1);
    const uint32_t* q = qBt + (bw * NW + w) * (int64_t)K;
    const ST* sp = scale_t + (bw * NG + g) * (int64_t)K;
    const ST* zp = zero_t + (bw * NG + g) * (int64_t)K;
    const uint16_t* arow = a + ba * (int64_t)K;
    float acc[CPW];
#pragma unroll
    for (int j = 0; j < CPW; j++) acc[j] = 0.0f;
    float zacc = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const uint32_t wv = q[k];
        const float av = h2f_bits(arow[k]);
        const float sa = ld_st<ST>(sp + k) * av;
        zacc = fmaf(ld_st<ST>(zp + k), av, zacc);
#pragma unroll
        for (int j = 0; j < CPW; j++) acc[j] = fmaf(sa, (float)((wv >> (BITS * j)) & MASK), acc[j]);
    }
#pragma unroll
    for (int j = 0; j < CPW; j++) acc[j] = wave_sum_dpp(acc[j] + zacc);
    if (lane == 0) {
        uint16_t* op = out16 + ba * (int64_t)(NW * CPW) + (int64_t)w * CPW;
#pragma unroll
        for (int j = 0; j < CPW; j++) op[j] = f2h_bits(acc[j]);
    }
}

extern "C" int gear_gemv_outer_dim(const void* in_feats, const void* kernel, const void* scaling_factors, const void* zeros, int64_t BA,
                                   int n_rep, int K, int N, int group, int bits, int mode, void* out, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4, "gear_gemv_outer_dim: bits must be 2 or 4 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_gemv_outer_dim: bad mode %d", mode);
    GEAR_CHECK_ARG(BA > 0 && K > 0 && N > 0, "gear_gemv_outer_dim: empty problem");
    GEAR_CHECK_ARG(n_rep >= 1 && BA % n_rep == 0, "gear_gemv_outer_dim: BA=%lld not divisible by n_rep=%d", (long long)BA, n_rep);
    const int cpw = 32 / bits;
    GEAR_CHECK_ARG(N % cpw == 0, "gear_gemv_outer_dim: N=%d must be a multiple of %d", N, cpw);
    GEAR_CHECK_ARG(group > 0 && group % cpw == 0, "gear_gemv_outer_dim: group %d must be a multiple of %d", group, cpw);
    GEAR_CHECK_ARG(BA <= 65535, "gear_gemv_outer_dim: batch*heads=%lld exceeds 65535", (long long)BA);
    GEAR_CHECK_ARG(in_feats && kernel && scaling_factors && zeros && out, "gear_gemv_outer_dim: null pointer");
    const int NW = N / cpw, NG = (N + group - 1) / group;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((NW + 3) / 4), (unsigned)BA);
#define GOD(B, STT)                                                                                                         \
    hipLaunchKernelGGL((gemv_outer_dim_kernel<B, STT>), grid, dim3(256), 0, st, (const uint16_t*)in_feats, (const uint32_t*)kernel, \
                       (const STT*)scaling_factors, (const STT*)zeros, n_rep, K, NW, NG, group, (uint16_t*)out)
    if (mode == 0) { if (bits == 2) GOD(2, uint16_t); else GOD(4, uint16_t); }
    else { if (bits == 2) GOD(2, float); else GOD(4, float); }
#undef GOD
    GEAR_CHECK_LAUNCH("gear_gemv_outer_dim");
    return 0;
}

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

// ================================================================================================ GEMV + low-rank epilogue
// matmul_withlrap (cuda_supported_gear/modeling_llamagear.py:54-111) = the GEMV above + the low-rank correction of the error
// (prefill factors and one factor pair per 64-token decode block, stacked on a leading dim).  The reference adds it with ~8 eager
// matmul / permute / slice-assign launches per call; here the split-K reduction of the GEMV becomes the epilogue that also adds
//   key   (K = head_dim, N = tokens):  out[n] += sum_c (a . Q_seg(n)[:, c]) P_seg(n)[n - start(seg), c]
//   value (K = tokens, N = head_dim):  out[d] += sum_seg sum_c (sum_{t in seg} a[t] Q_seg[t - start, c]) P_seg[d, c]
// in fp32 with one fp16 rounding at the end (the reference rounds a @ Q, (a @ Q) @ P^T and the sum to fp16 one by one).
namespace {

struct LrapArgs {
    const float* part;        // [splits][BA][N]
    int splits;
    int64_t BA;
    int n_rep, K, N, r;
    const uint16_t* a;        // [BA][K]
    const uint16_t* P0;       // key: [BW][Tp][r]   value: [BW][N][r]
    const uint16_t* Q0;       // key: [BW][K][r]    value: [BW][Tp][r]
    const uint16_t* P1;       // key: [nbuf][BW][blk][r]   value: [nbuf][BW][N][r]
    const uint16_t* Q1;       // key: [nbuf][BW][K][r]     value: [nbuf][BW][blk][r]
    int Tp, nbuf, blk;
    uint16_t* out;            // [BA][N]
};

constexpr int LR_MAXR = 16;
constexpr int LR_TOK = 256;      // tokens per workgroup of the key epilogue

// key: grid (ceil(N / 256), BA), 256 threads = 256 tokens.  The tokens of a workgroup touch the prefill segment and / or a few
// 64-token blocks: their a . Q vectors are computed first (LDS), then every thread adds its token's term.
__global__ __launch_bounds__(256) void lrap_key_kernel(LrapArgs g) {
    __shared__ float aq[(LR_TOK / 16 + 2) * LR_MAXR];      // segment slots of this workgroup: slot 0 = prefill, 1 + i = block b0 + i
    __shared__ float av[256];
    const int tid = threadIdx.x;
    const int64_t ba = blockIdx.y, bw = ba / g.n_rep, BW = g.BA / g.n_rep;
    const int n0 = blockIdx.x * LR_TOK, n1 = min(g.N, n0 + LR_TOK);
    const int r = g.r, K = g.K;
    for (int k = tid; k < K; k += 256) av[k] = h2f_bits(g.a[ba * K + k]);
    const bool has_pre = n0 < g.Tp;
    const int b0 = (max(n0, g.Tp) - g.Tp) / g.blk;                                   // first block this workgroup touches
    const int b1 = (n1 > g.Tp && g.nbuf > 0) ? min(g.nbuf, (n1 - 1 - g.Tp) / g.blk + 1) : b0;
    const int nslot = 1 + max(0, b1 - b0);
    __syncthreads();
    for (int o = tid; o < nslot * r; o += 256) {
        const int slot = o / r, c = o % r;
        float s = 0.0f;
        if (slot == 0) {
            if (has_pre) {
                const uint16_t* q = g.Q0 + bw * (int64_t)K * r + c;
                for (int k = 0; k < K; k++) s = fmaf(av[k], h2f_bits(q[(int64_t)k * r]), s);
            }
        } else {
            const uint16_t* q = g.Q1 + ((int64_t)(b0 + slot - 1) * BW + bw) * (int64_t)K * r + c;
            for (int k = 0; k < K; k++) s = fmaf(av[k], h2f_bits(q[(int64_t)k * r]), s);
        }
        aq[slot * LR_MAXR + c] = s;
    }
    __syncthreads();
    const int n = n0 + tid;
    if (n >= n1) return;
    float s = 0.0f;
    for (int p = 0; p < g.splits; p++) s += g.part[((int64_t)p * g.BA + ba) * g.N + n];
    const uint16_t* prow = nullptr;
    const float* w = aq;
    if (n < g.Tp) prow = g.P0 + (bw * (int64_t)g.Tp + n) * r;
    else if (g.nbuf > 0) {
        const int b = (n - g.Tp) / g.blk;
        if (b < g.nbuf) {
            prow = g.P1 + (((int64_t)b * BW + bw) * g.blk + (n - g.Tp - b * g.blk)) * (int64_t)r;
            w = aq + (1 + b - b0) * LR_MAXR;
        }
    }
    if (prow)
        for (int c = 0; c < r; c++) s = fmaf(w[c], h2f_bits(prow[c]), s);
    g.out[ba * g.N + n] = f2h_bits(s);
}

// value: grid (BA), 256 threads.  A wave takes 64-token pieces in turn: lane = token, w[c] = wave sum of a[t] Q[t][c]; the pieces
// of the prefill segment share P0 (their w are summed first), a decode block is applied at once: acc[d] += sum_c w[c] P[d][c]
// with lanes d = lane, lane + 64.  N = head_dim <= 128.
__global__ __launch_bounds__(256) void lrap_value_kernel(LrapArgs g) {
    __shared__ float accs[4][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ba = blockIdx.x, bw = ba / g.n_rep, BW = g.BA / g.n_rep;
    const int r = g.r, N = g.N;
    const uint16_t* arow = g.a + ba * g.K;
    float acc0 = 0.0f, acc1 = 0.0f;                    // channels lane, lane + 64
    auto apply = [&](const float (&w)[LR_MAXR], const uint16_t* P) {      // P [N][r]
        if (lane < N) for (int c = 0; c < r; c++) acc0 = fmaf(w[c], h2f_bits(P[(int64_t)lane * r + c]), acc0);
        if (lane + 64 < N) for (int c = 0; c < r; c++) acc1 = fmaf(w[c], h2f_bits(P[(int64_t)(lane + 64) * r + c]), acc1);
    };
    // prefill segment: pieces of 64 tokens round-robin over the waves
    {
        float w[LR_MAXR];
#pragma unroll
        for (int c = 0; c < LR_MAXR; c++) w[c] = 0.0f;
        for (int t0 = wave * 64; t0 < g.Tp; t0 += 256) {
            const int t = t0 + lane;
            const float x = t < g.Tp ? h2f_bits(arow[t]) : 0.0f;
            const uint16_t* q = g.Q0 + (bw * (int64_t)g.Tp + min(t, g.Tp - 1)) * r;
#pragma unroll
            for (int c = 0; c < LR_MAXR; c++)
                if (c < r) w[c] += x * h2f_bits(q[c]);
        }
#pragma unroll
        for (int c = 0; c < LR_MAXR; c++)
            if (c < r) w[c] = wave_sum_dpp(w[c]);
        if (g.Tp > 0) apply(w, g.P0 + bw * (int64_t)N * r);
    }
    for (int b = wave; b < g.nbuf; b += 4) {
        float w[LR_MAXR];
        const int tl = lane < g.blk ? lane : g.blk - 1;
        const float x = lane < g.blk ? h2f_bits(arow[g.Tp + b * g.blk + lane]) : 0.0f;
        const uint16_t* q = g.Q1 + (((int64_t)b * BW + bw) * g.blk + tl) * (int64_t)r;
#pragma unroll
        for (int c = 0; c < LR_MAXR; c++) w[c] = c < r ? wave_sum_dpp(x * h2f_bits(q[c])) : 0.0f;
        apply(w, g.P1 + ((int64_t)b * BW + bw) * (int64_t)N * r);
    }
    accs[wave][lane] = acc0;
    accs[wave][lane + 64] = acc1;
    __syncthreads();
    if (tid < N) {
        float s = accs[0][tid] + accs[1][tid] + accs[2][tid] + accs[3][tid];
        for (int p = 0; p < g.splits; p++) s += g.part[((int64_t)p * g.BA + ba) * N + tid];
        g.out[ba * N + tid] = f2h_bits(s);
    }
}

}  // namespace

extern "C" size_t gear_gemv_outer_lrap_workspace(int64_t BA, int K, int N, int bits) {
    if (bits != 2 && bits != 4) return 0;
    GemvPlan p = plan_gemv(BA, K, N, bits);
    return sizeof(float) * (size_t)(p.splits > 1 ? p.splits : 1) * (size_t)BA * (size_t)N;
}

extern "C" int gear_gemv_outer_lrap(const void* a, const void* qB, const void* scale, const void* zero, int64_t BA, int n_rep,
                                    int K, int N, int group, int bits, int mode, int kind, const void* P0, const void* Q0, int Tp,
                                    const void* P1, const void* Q1, int nbuf, int blk, int r, void* out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4, "gear_gemv_outer_lrap: bits must be 2 or 4 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_gemv_outer_lrap: bad mode %d", mode);
    GEAR_CHECK_ARG(BA > 0 && K > 0 && N > 0 && BA <= 65535, "gear_gemv_outer_lrap: bad problem size");
    GEAR_CHECK_ARG(n_rep >= 1 && BA % n_rep == 0, "gear_gemv_outer_lrap: BA=%lld not divisible by n_rep=%d", (long long)BA, n_rep);
    const int cpw = 32 / bits;
    GEAR_CHECK_ARG(N % cpw == 0 && group > 0 && group % cpw == 0, "gear_gemv_outer_lrap: N / group must be multiples of %d", cpw);
    GEAR_CHECK_ARG(a && qB && scale && zero && out && workspace, "gear_gemv_outer_lrap: null pointer");
    GEAR_CHECK_ARG(kind == 0 || kind == 1, "gear_gemv_outer_lrap: kind must be 0 (key) or 1 (value)");
    GEAR_CHECK_ARG(r >= 1 && r <= LR_MAXR, "gear_gemv_outer_lrap: rank must be in [1, %d]", LR_MAXR);
    GEAR_CHECK_ARG(Tp >= 0 && nbuf >= 0 && (nbuf == 0 || (blk >= 16 && blk <= 64 && P1 && Q1)) && (Tp == 0 || (P0 && Q0)),
                   "gear_gemv_outer_lrap: bad factor geometry");
    const int span = Tp + nbuf * blk;
    GEAR_CHECK_ARG(kind == 0 ? span <= N : (span <= K && N <= 128), "gear_gemv_outer_lrap: factors cover %d tokens, the payload %d", span,
                   kind == 0 ? N : K);
    const int NW = N / cpw, NG = (N + group - 1) / group;
    const GemvPlan p = plan_gemv(BA, K, N, bits);
    GEAR_CHECK_ARG(p.vec == 1 || NW % 4 == 0, "gear_gemv_outer_lrap: row pitch must be a multiple of 4 words");
    GEAR_CHECK_ARG(workspace_bytes >= gear_gemv_outer_lrap_workspace(BA, K, N, bits), "gear_gemv_outer_lrap: workspace too small");
    float* part = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int64_t ldq = NW, lds = NG;
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
    GEAR_CHECK_LAUNCH("gear_gemv_outer_lrap(gemv)");
    LrapArgs g;
    g.part = part; g.splits = p.splits; g.BA = BA; g.n_rep = n_rep; g.K = K; g.N = N; g.r = r;
    g.a = (const uint16_t*)a; g.P0 = (const uint16_t*)P0; g.Q0 = (const uint16_t*)Q0; g.P1 = (const uint16_t*)P1; g.Q1 = (const uint16_t*)Q1;
    g.Tp = Tp; g.nbuf = nbuf; g.blk = blk; g.out = (uint16_t*)out;
    if (kind == 0) {
        GEAR_CHECK_ARG(K <= 256, "gear_gemv_outer_lrap: key side needs head_dim <= 256 (got %d)", K);
        hipLaunchKernelGGL(lrap_key_kernel, dim3((unsigned)((N + LR_TOK - 1) / LR_TOK), (unsigned)BA), dim3(256), 0, st, g);
    } else {
        hipLaunchKernelGGL(lrap_value_kernel, dim3((unsigned)BA), dim3(256), 0, st, g);
    }
    GEAR_CHECK_LAUNCH("gear_gemv_outer_lrap(epilogue)");
    return 0;
}
