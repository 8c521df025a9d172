// gemv_f16.hip -- the projections of the decode token step (batch <= 4): y[b, n] = sum_k x[b, k] W[n, k], fp16 in / out,
// fp32 accumulate, with the glue ops of a decoder layer folded into the weight stream.
//
// HBM-bound weight streaming.  Measured on MI355X (tools/exp_gemv.py): ONE output row per wave with four 16-byte loads
// per lane in flight beats 2/4/8 rows per wave by 10-25 % (5.6 / 4.6 / 6.2 / 4.8 / 6.3 TB/s on the Llama-2-7B qkv / o /
// gate-up / down / lm_head shapes), so a workgroup is 4 waves = 4 rows (or 4/SK rows with the K dimension split over SK
// waves for the short-and-wide shapes); lane-level partial sums meet in LDS, where the epilogues also find their partner
// rows:
//   prologue (NORM): v = fp16(x + delta) is rebuilt by every lane for the chunk it is about to use and sum(v^2)
//   accumulates beside the dot product; the projection is linear, so the row scale rsqrt(mean(v^2) + eps) is applied once
//   to the finished dot product:  y[n] = inv * sum_k W[n,k] (w_norm[k] v[k])  (w_norm may be pre-folded into W: pass NULL).
//   torch rounds fp16(v * inv) before the weight multiply; the two differ by fp16 rounding only.
//   epilogue ADD   : y = res_in + W x (the residual stream update after o_proj / down_proj)
//   epilogue SwiGLU: W rows are interleaved (gate_i, up_i); the block's 4 rows are two finished pairs
//   epilogue RoPE  : W = [q heads | k heads | v heads] x 128 rows; a block takes rows {2u, 2u+1, 2u+64, 2u+65} of one
//                    head = two rotation pairs, rotates them (fp16 op by op like HF) and stores q / the window slot.
// Replaces the library GEMV that F.linear dispatches for M = 1 (3.6 TB/s average over a Llama-2-7B layer) and the
// add+RMSNorm / RoPE+append / SwiGLU launches (decode_ops.hip) inside FastGearDecoder.  The model's projections are not
// part of the reference's hot path; this keeps the decode-tokens/s harness from being bounded by them.
#include <stdlib.h>

#include "common.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot8(const uint4& a, const uint4& b, float acc) {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.x), __builtin_bit_cast(half2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.y), __builtin_bit_cast(half2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.z), __builtin_bit_cast(half2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.w), __builtin_bit_cast(half2_t, b.w), acc, false);
    return acc;
}
__device__ __forceinline__ uint4 hadd8(const uint4& a, const uint4& b) {
    uint4 r;
    r.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, a.x) + __builtin_bit_cast(half2_t, b.x));
    r.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, a.y) + __builtin_bit_cast(half2_t, b.y));
    r.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, a.z) + __builtin_bit_cast(half2_t, b.z));
    r.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, a.w) + __builtin_bit_cast(half2_t, b.w));
    return r;
}
__device__ __forceinline__ uint4 hmul8(const uint4& a, const uint4& b) {
    uint4 r;
    r.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, a.x) * __builtin_bit_cast(half2_t, b.x));
    r.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, a.y) * __builtin_bit_cast(half2_t, b.y));
    r.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, a.z) * __builtin_bit_cast(half2_t, b.z));
    r.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, a.w) * __builtin_bit_cast(half2_t, b.w));
    return r;
}

enum { EPI_PLAIN = 0, EPI_SWIGLU = 1, EPI_ROPE = 2 };

struct GemvArgs {
    const uint16_t* x;        // [B, K] activations (the residual stream when NORM)
    const uint16_t* delta;    // NORM: optional addend [B, K]
    const uint16_t* nw;       // NORM: optional norm weight [K] (NULL: folded into W)
    uint16_t* res_out;        // NORM && delta: x + delta
    const uint16_t* res_in;   // EPI_PLAIN: optional residual added to the result [B, N]
    float eps;
    const uint16_t* W;
    uint16_t* y;
    int K, N;
    int Hq, Hkv, pos, slot, wcap;
    float log2_theta;
    const int* dyn;
    uint16_t *q_out, *kwin, *vwin;
    int half_n;               // EPI_SWIGLU with BLOCKED weights [gate rows | up rows]: N / 2 (0: the rows are interleaved gate, up, gate, ...)
};

template <int NB, int SK, int NORM, int EPI>
__global__ __launch_bounds__(EPI == EPI_ROPE ? 256 * SK : 256) void gemv_tok_kernel(GemvArgs a) {
    constexpr int WPB = EPI == EPI_ROPE ? 4 * SK : 4;   // waves per block (RoPE blocks always hold 4 rows)
    constexpr int RPB = WPB / SK;                       // rows per block
    static_assert(EPI != EPI_SWIGLU || SK <= 2, "SwiGLU blocks hold at least one pair");
    __shared__ float part[WPB][NB], ssp[WPB][NB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsub = wave / SK, kp = wave % SK;
    int row, head = 0, pu = 0;
    if (EPI == EPI_ROPE) {
        head = blockIdx.x >> 5;
        pu = blockIdx.x & 31;
        row = head * 128 + 2 * pu + (rsub & 1) + 64 * (rsub >> 1);
    } else {
        row = min((int)blockIdx.x * RPB + rsub, a.N - 1);
        // (gate_proj / up_proj as the modules hold them -- two row blocks of one buffer: pair (2 j, 2 j + 1) = rows j and N / 2 + j)
        if (EPI == EPI_SWIGLU && a.half_n) row = (row & 1) * a.half_n + (row >> 1);
    }
    const int nchunk = a.K / 8;
    float acc[NB], ss[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = ss[b] = 0.0f;
    const uint4* Wv = (const uint4*)a.W + (int64_t)row * nchunk;
    const uint4* xv = (const uint4*)a.x;
    const uint4* dv = (const uint4*)a.delta;
    const uint4* nv = (const uint4*)a.nw;
    uint4* rv = (uint4*)a.res_out;
    // NORM 1: the residual stream as it is (norm weight folded into W) -- a loop without run-time branches, so that it
    // unrolls with four weight loads in flight like the plain GEMV; NORM 2: optional addend / norm weight / residual write;
    // NORM 3 (round 6): the norm weight applied on the fly and nothing else -- the attention hook's usual step (the modules' own
    // weights, no addend): NORM 2's run-time branches kept its loop from unrolling (gate / up 34.8 us against 28.9 folded)
    const bool writer = NORM == 2 && a.res_out && blockIdx.x == 0 && rsub == 0;   // this row's K parts cover every chunk once
#pragma unroll 4
    for (int c = lane + 64 * kp; c < nchunk; c += 64 * SK) {
        const uint4 w = Wv[c];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            uint4 xa = xv[(int64_t)b * nchunk + c];
            if (NORM == 2) {
                if (a.delta) xa = hadd8(xa, dv[(int64_t)b * nchunk + c]);
                if (writer) rv[(int64_t)b * nchunk + c] = xa;
            }
            if (NORM) ss[b] = dot8(xa, xa, ss[b]);
            if (NORM == 2) {
                if (a.nw) xa = hmul8(xa, nv[c]);
            }
            if (NORM == 3) xa = hmul8(xa, nv[c]);      // norm weight on the fly, nothing else: no run-time branch in the loop
            acc[b] = dot8(w, xa, acc[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        float t = acc[b], u = ss[b];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            t += __shfl_xor(t, m, 64);
            if (NORM) u += __shfl_xor(u, m, 64);
        }
        if (lane == 0) { part[wave][b] = t; ssp[wave][b] = u; }
    }
    __syncthreads();
    // finished value of (block row r, batch row b), rounded to fp16 like the library GEMV's output
    auto val = [&](int r, int b) {
        float t = 0.0f, u = 0.0f;
#pragma unroll
        for (int k = 0; k < SK; k++) { t += part[r * SK + k][b]; u += ssp[r * SK + k][b]; }
        if (NORM) t *= rsqrtf(u / (float)a.K + a.eps);
        return hround(t);
    };
    const int t = threadIdx.x;
    if (EPI == EPI_PLAIN) {
        if (t < RPB * NB) {
            const int r = t / NB, b = t % NB, n = blockIdx.x * RPB + r;
            if (n < a.N) {
                float v = val(r, b);
                if (a.res_in) v = hround(v + h2f_bits(a.res_in[(int64_t)b * a.N + n]));
                a.y[(int64_t)b * a.N + n] = f2h_bits(v);
            }
        }
    } else if (EPI == EPI_SWIGLU) {   // (gate, up) pairs -> fp16(silu(gate)) * up, like silu_mul_kernel
        if (t < (RPB / 2) * NB) {
            const int pr = t / NB, b = t % NB, n = blockIdx.x * RPB + 2 * pr;
            if (n + 1 < a.N) {
                const float g = val(2 * pr, b), u = val(2 * pr + 1, b);
                const float sg = hround(g / (1.0f + expf(-g)));
                a.y[(int64_t)b * (a.N / 2) + n / 2] = f2h_bits(sg * u);
            }
        }
    } else {
        if (t < 2 * NB) {   // one rotation pair (or two V elements) per thread
            const int e = t / NB, b = t % NB;
            const int Hq = a.Hq, Hkv = a.Hkv;
            const int p = 2 * pu + e;
            const float x1 = val(e, b), x2 = val(2 + e, b);
            int pos = a.pos, slot = a.slot;
            if (a.dyn) { pos = a.dyn[0]; slot = a.dyn[1]; }
            if (head >= Hq + Hkv) {
                uint16_t* dst = a.vwin + (((int64_t)b * Hkv + (head - Hq - Hkv)) * a.wcap + slot) * 128;
                dst[p] = f2h_bits(x1);
                dst[p + 64] = f2h_bits(x2);
            } else {   // same arithmetic as rope_append_kernel
                uint16_t* dst = (head < Hq) ? a.q_out + ((int64_t)b * Hq + head) * 128
                                            : a.kwin + (((int64_t)b * Hkv + (head - Hq)) * a.wcap + slot) * 128;
                const float inv_freq = exp2f(-(float)(2 * p) / 128.0f * a.log2_theta);
                const float ang = (float)pos * inv_freq;
                const float c = hround(cosf(ang)), sn = hround(sinf(ang));
                dst[p] = f2h_bits(hround(hround(x1 * c) + hround(-x2 * sn)));
                dst[p + 64] = f2h_bits(hround(hround(x2 * c) + hround(x1 * sn)));
            }
        }
    }
}

template <int SK, int NORM, int EPI>
void launch_nb(int B, unsigned blocks, hipStream_t st, const GemvArgs& a) {
    dim3 grid(blocks), block(EPI == EPI_ROPE ? 256 * SK : 256);
    if (B == 1) hipLaunchKernelGGL((gemv_tok_kernel<1, SK, NORM, EPI>), grid, block, 0, st, a);
    else if (B == 2) hipLaunchKernelGGL((gemv_tok_kernel<2, SK, NORM, EPI>), grid, block, 0, st, a);
    else if (B == 3) hipLaunchKernelGGL((gemv_tok_kernel<3, SK, NORM, EPI>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemv_tok_kernel<4, SK, NORM, EPI>), grid, block, 0, st, a);
}

// K split over the waves of a block.  Measured (tools/exp_gemv.py, GB/s of weight bytes, qkv / o / gate-up / down / lm_head):
//   SK=1  5696 4473 6156 4784 6338     SK=2  5907 4608 6376 4788 6581     SK=4  5497 4188 6224 5015 6488
int pick_sk(int K, int N, int max_sk) {
    int sk = 2;
    if ((int64_t)N <= 8192 && K >= 8192) sk = 4;
    if (sk != 1 && sk != 2 && sk != 4) sk = 1;
    return sk > max_sk ? max_sk : sk;
}

template <int NORM, int EPI>
void launch_sk(int sk, int B, int N, hipStream_t st, const GemvArgs& a) {
    if constexpr (EPI != EPI_SWIGLU) {
        if (sk == 4) return launch_nb<4, NORM, EPI>(B, (unsigned)N, st, a);
    }
    if (sk >= 2) return launch_nb<2, NORM, EPI>(B, (unsigned)((N + 1) / 2), st, a);
    launch_nb<1, NORM, EPI>(B, (unsigned)((N + 3) / 4), st, a);
}

}  // namespace

extern "C" int gear_gemv_f16_add(const void* x, const void* W, int B, int K, int N, const void* res_in, void* y,
                                 void* stream) {
    GEAR_CHECK_ARG(x && W && y, "gear_gemv_f16: null pointer");
    GEAR_CHECK_ARG(B >= 1 && B <= 4, "gear_gemv_f16: batch must be in [1,4] (got %d)", B);
    GEAR_CHECK_ARG(K > 0 && K % 8 == 0 && N > 0, "gear_gemv_f16: K=%d must be a positive multiple of 8", K);
    GemvArgs a = {};
    a.x = (const uint16_t*)x; a.W = (const uint16_t*)W; a.y = (uint16_t*)y; a.K = K; a.N = N;
    a.res_in = (const uint16_t*)res_in;
    launch_sk<0, EPI_PLAIN>(pick_sk(K, N, 4), B, N, (hipStream_t)stream, a);
    GEAR_CHECK_LAUNCH("gear_gemv_f16");
    return 0;
}

extern "C" int gear_gemv_f16(const void* x, const void* W, int B, int K, int N, void* y, void* stream) {
    return gear_gemv_f16_add(x, W, B, K, N, nullptr, y, stream);
}

extern "C" int gear_gemv_f16_norm(const void* x, const void* delta, const void* norm_w, float eps, const void* W, int B,
                                  int K, int N, int swiglu, void* res_out, void* y, void* stream) {
    GEAR_CHECK_ARG(x && W && y, "gear_gemv_f16_norm: null pointer");
    GEAR_CHECK_ARG(B >= 1 && B <= 4, "gear_gemv_f16_norm: batch must be in [1,4] (got %d)", B);
    GEAR_CHECK_ARG(K > 0 && K % 8 == 0 && N > 0, "gear_gemv_f16_norm: K=%d must be a positive multiple of 8", K);
    GEAR_CHECK_ARG(!swiglu || N % 4 == 0, "gear_gemv_f16_norm: SwiGLU needs interleaved (gate, up) rows, N %% 4 == 0");
    GEAR_CHECK_ARG(!delta || (res_out && res_out != x && res_out != delta),
                   "gear_gemv_f16_norm: the new residual needs its own buffer (other workgroups still read the inputs)");
    GemvArgs a = {};
    a.x = (const uint16_t*)x; a.delta = (const uint16_t*)delta; a.nw = (const uint16_t*)norm_w;
    a.res_out = delta ? (uint16_t*)res_out : nullptr;
    a.eps = eps; a.W = (const uint16_t*)W; a.y = (uint16_t*)y; a.K = K; a.N = N;
    const bool folded = !delta && !norm_w;
    a.half_n = swiglu == 2 ? N / 2 : 0;
    const bool nw_only = !delta && norm_w;
    if (swiglu) {
        if (folded) launch_sk<1, EPI_SWIGLU>(pick_sk(K, N, 2), B, N, (hipStream_t)stream, a);
        else if (nw_only) launch_sk<3, EPI_SWIGLU>(pick_sk(K, N, 2), B, N, (hipStream_t)stream, a);
        else launch_sk<2, EPI_SWIGLU>(pick_sk(K, N, 2), B, N, (hipStream_t)stream, a);
    } else {
        if (folded) launch_sk<1, EPI_PLAIN>(pick_sk(K, N, 4), B, N, (hipStream_t)stream, a);
        else if (nw_only) launch_sk<3, EPI_PLAIN>(pick_sk(K, N, 4), B, N, (hipStream_t)stream, a);
        else launch_sk<2, EPI_PLAIN>(pick_sk(K, N, 4), B, N, (hipStream_t)stream, a);
    }
    GEAR_CHECK_LAUNCH("gear_gemv_f16_norm");
    return 0;
}

extern "C" int gear_gemv_qkv_rope(const void* x, const void* delta, const void* norm_w, float eps, const void* Wqkv, int B,
                                  int K, int Hq, int Hkv, int D, int pos, int slot, int wcap, float theta,
                                  const void* dyn_state, void* res_out, void* q_out, void* kwin, void* vwin, void* stream) {
    GEAR_CHECK_ARG(D == 128, "gear_gemv_qkv_rope: head_dim must be 128 (got %d)", D);
    GEAR_CHECK_ARG(x && Wqkv && q_out && kwin && vwin, "gear_gemv_qkv_rope: null pointer");
    GEAR_CHECK_ARG(B >= 1 && B <= 4, "gear_gemv_qkv_rope: batch must be in [1,4] (got %d)", B);
    GEAR_CHECK_ARG(K > 0 && K % 8 == 0 && Hq > 0 && Hkv > 0, "gear_gemv_qkv_rope: bad shape");
    GEAR_CHECK_ARG(dyn_state || (slot >= 0 && slot < wcap && pos >= 0), "gear_gemv_qkv_rope: bad pos / slot");
    GEAR_CHECK_ARG(!delta || (res_out && res_out != x && res_out != delta),
                   "gear_gemv_qkv_rope: the new residual needs its own buffer");
    GemvArgs a = {};
    a.x = (const uint16_t*)x; a.delta = (const uint16_t*)delta; a.nw = (const uint16_t*)norm_w;
    a.res_out = delta ? (uint16_t*)res_out : nullptr;
    a.eps = eps; a.W = (const uint16_t*)Wqkv; a.K = K; a.N = (Hq + 2 * Hkv) * 128;
    a.Hq = Hq; a.Hkv = Hkv; a.pos = pos; a.slot = slot; a.wcap = wcap; a.log2_theta = log2f(theta);
    a.dyn = (const int*)dyn_state; a.q_out = (uint16_t*)q_out; a.kwin = (uint16_t*)kwin; a.vwin = (uint16_t*)vwin;
    const unsigned blocks = (unsigned)((Hq + 2 * Hkv) * 32);
    const bool folded = !delta && !norm_w, nw_only = !delta && norm_w;
    if (pick_sk(K, a.N, 2) == 2) {
        if (folded) launch_nb<2, 1, EPI_ROPE>(B, blocks, (hipStream_t)stream, a);
        else if (nw_only) launch_nb<2, 3, EPI_ROPE>(B, blocks, (hipStream_t)stream, a);
        else launch_nb<2, 2, EPI_ROPE>(B, blocks, (hipStream_t)stream, a);
    } else {
        if (folded) launch_nb<1, 1, EPI_ROPE>(B, blocks, (hipStream_t)stream, a);
        else if (nw_only) launch_nb<1, 3, EPI_ROPE>(B, blocks, (hipStream_t)stream, a);
        else launch_nb<1, 2, EPI_ROPE>(B, blocks, (hipStream_t)stream, a);
    }
    GEAR_CHECK_LAUNCH("gear_gemv_qkv_rope");
    return 0;
}
