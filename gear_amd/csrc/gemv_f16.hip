// gemv_f16.hip -- y[b, n] = sum_k x[b, k] W[n, k]  for the decode token step (b <= 4), fp16 in / fp16 out, fp32 accumulate.
// HBM-bound weight streaming: one wave owns 4 output rows at a time, lanes stride the K dimension with 16-byte loads
// (4 row vectors + 1 activation vector per trip, 16 v_dot2_f32_f16 per batch row), one butterfly per row at the end.
// Replaces the library GEMV that F.linear dispatches for M = 1 (3.6 TB/s average over a Llama-2-7B layer, 1.65 TB/s on
// the 4096x4096 o_proj) inside FastGearDecoder; the model's projections are not part of the reference's hot path, this
// only keeps the decode-tokens/s harness from being bounded by them.
#include "common.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot8(const uint4& a, const uint4& b, float acc) {
#if __has_builtin(__builtin_amdgcn_fdot2)
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.x), __builtin_bit_cast(half2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.y), __builtin_bit_cast(half2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.z), __builtin_bit_cast(half2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a.w), __builtin_bit_cast(half2_t, b.w), acc, false);
#else
    float fa[8], fb[8];
    unpack8(a, fa);
    unpack8(b, fb);
#pragma unroll
    for (int i = 0; i < 8; i++) acc = fmaf(fa[i], fb[i], acc);
#endif
    return acc;
}

template <int NB>
__global__ __launch_bounds__(256) void gemv_f16_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ W,
                                                       uint16_t* __restrict__ y, int K, int N) {
    constexpr int RPW = 4;  // rows per wave per trip
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * RPW;
    if (n0 >= N) return;
    const int nchunk = K / 8;
    float acc[NB][RPW];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < RPW; r++) acc[b][r] = 0.0f;
    const uint4* Wv = (const uint4*)W;
    const uint4* xv = (const uint4*)x;
    int rows[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) rows[r] = min(n0 + r, N - 1);
#pragma unroll 2
    for (int c = lane; c < nchunk; c += 64) {
        uint4 w[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) w[r] = Wv[(int64_t)rows[r] * nchunk + c];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const uint4 xa = xv[(int64_t)b * nchunk + c];
#pragma unroll
            for (int r = 0; r < RPW; r++) acc[b][r] = dot8(w[r], xa, acc[b][r]);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            float v = acc[b][r];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0 && n0 + r < N) y[(int64_t)b * N + n0 + r] = f2h_bits(v);
        }
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused token-step variants: the glue launches around the projections (residual add + RMSNorm before, SwiGLU gate or
// RoPE + window append after) ride along with the weight stream instead of costing a launch + a latency chain each.
//
//   prologue (NORM): v = fp16(x + delta) is rebuilt by every lane for the chunk it is about to use, sum(v^2) accumulates
//   beside the dot products, and because the projection is linear the row scale rsqrt(mean(v^2) + eps) is applied once
//   to the finished dot product:  y[n] = inv * sum_k W[n,k] (w_norm[k] v[k]).  (torch rounds fp16(v * inv) before the
//   weight multiply; the two differ by fp16 rounding only.)  Wave 0 of block 0 writes the new residual stream.
//   epilogue 1 (SwiGLU): W rows are interleaved (gate_i, up_i), a wave's 4 rows are two finished pairs.
//   epilogue 2 (RoPE + append): W = [q heads | k heads | v heads] x 128 rows; a wave takes rows {2u, 2u+1, 2u+64, 2u+65}
//   of one head = two rotation pairs, rotates them (fp16 op by op like HF) and stores q / the window slot directly.
struct GemvEx {
    const uint16_t* x;
    const uint16_t* delta;
    const uint16_t* nw;
    uint16_t* res_out;
    float eps;
    const uint16_t* W;
    uint16_t* y;
    int K, N;
    int Hq, Hkv, pos, slot, wcap;
    float log2_theta;
    const int* dyn;
    uint16_t *q_out, *kwin, *vwin;
};

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

template <int NB, bool NORM, int EPI>
__global__ __launch_bounds__(256) void gemv_f16_ex_kernel(GemvEx a) {
    constexpr int RPW = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.x * 4 + wave;
    int rows[RPW];
    int n0 = unit * RPW, head = 0, pu = 0;
    if (EPI == 2) {
        head = unit >> 5;
        pu = unit & 31;
        if (head >= a.Hq + 2 * a.Hkv) return;
        rows[0] = head * 128 + 2 * pu; rows[1] = rows[0] + 1; rows[2] = rows[0] + 64; rows[3] = rows[0] + 65;
    } else {
        if (n0 >= a.N) return;
#pragma unroll
        for (int r = 0; r < RPW; r++) rows[r] = min(n0 + r, a.N - 1);
    }
    const int nchunk = a.K / 8;
    float acc[NB][RPW], ss[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        ss[b] = 0.0f;
#pragma unroll
        for (int r = 0; r < RPW; r++) acc[b][r] = 0.0f;
    }
    const uint4* Wv = (const uint4*)a.W;
    const uint4* xv = (const uint4*)a.x;
    const uint4* dv = (const uint4*)a.delta;
    const uint4* nv = (const uint4*)a.nw;
    uint4* rv = (uint4*)a.res_out;
    const bool writer = NORM && a.res_out && blockIdx.x == 0 && wave == 0;
#pragma unroll 2
    for (int c = lane; c < nchunk; c += 64) {
        uint4 w[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) w[r] = Wv[(int64_t)rows[r] * nchunk + c];
        uint4 g;
        if (NORM) g = nv[c];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            uint4 xa = xv[(int64_t)b * nchunk + c];
            if (NORM) {
                if (a.delta) xa = hadd8(xa, dv[(int64_t)b * nchunk + c]);
                if (writer) rv[(int64_t)b * nchunk + c] = xa;
                ss[b] = dot8(xa, xa, ss[b]);
                xa = hmul8(xa, g);
            }
#pragma unroll
            for (int r = 0; r < RPW; r++) acc[b][r] = dot8(w[r], xa, acc[b][r]);
        }
    }
    int pos = a.pos, slot = a.slot;
    if (EPI == 2 && a.dyn) { pos = a.dyn[0]; slot = a.dyn[1]; }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        float inv = 1.0f;
        if (NORM) {
            float t = ss[b];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
            inv = rsqrtf(t / (float)a.K + a.eps);
        }
        float v[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            float t = acc[b][r];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
            v[r] = hround(t * inv);
        }
        if (lane != 0) continue;
        if (EPI == 0) {
#pragma unroll
            for (int r = 0; r < RPW; r++)
                if (n0 + r < a.N) a.y[(int64_t)b * a.N + n0 + r] = f2h_bits(v[r]);
        } else if (EPI == 1) {   // (gate, up) pairs -> fp16(silu(gate)) * up, like silu_mul_kernel
#pragma unroll
            for (int r = 0; r < RPW; r += 2) {
                if (n0 + r + 1 < a.N) {
                    const float sg = hround(v[r] / (1.0f + expf(-v[r])));
                    a.y[(int64_t)b * (a.N / 2) + (n0 + r) / 2] = f2h_bits(sg * v[r + 1]);
                }
            }
        } else {
            const int Hq = a.Hq, Hkv = a.Hkv;
            if (head >= Hq + Hkv) {
                uint16_t* dst = a.vwin + (((int64_t)b * Hkv + (head - Hq - Hkv)) * a.wcap + slot) * 128;
                dst[2 * pu] = f2h_bits(v[0]); dst[2 * pu + 1] = f2h_bits(v[1]);
                dst[2 * pu + 64] = f2h_bits(v[2]); dst[2 * pu + 65] = f2h_bits(v[3]);
            } else {
                uint16_t* dst = (head < Hq) ? a.q_out + ((int64_t)b * Hq + head) * 128
                                            : a.kwin + (((int64_t)b * Hkv + (head - Hq)) * a.wcap + slot) * 128;
#pragma unroll
                for (int e = 0; e < 2; e++) {   // same arithmetic as rope_append_kernel
                    const int p = 2 * pu + e;
                    const float inv_freq = exp2f(-(float)(2 * p) / 128.0f * a.log2_theta);
                    const float ang = (float)pos * inv_freq;
                    const float c = hround(cosf(ang)), sn = hround(sinf(ang));
                    const float x1 = v[e], x2 = v[2 + e];
                    dst[p] = f2h_bits(hround(hround(x1 * c) + hround(-x2 * sn)));
                    dst[p + 64] = f2h_bits(hround(hround(x2 * c) + hround(x1 * sn)));
                }
            }
        }
    }
}

template <bool NORM, int EPI>
void launch_ex(int B, dim3 grid, hipStream_t st, const GemvEx& a) {
    if (B == 1) hipLaunchKernelGGL((gemv_f16_ex_kernel<1, NORM, EPI>), grid, dim3(256), 0, st, a);
    else if (B == 2) hipLaunchKernelGGL((gemv_f16_ex_kernel<2, NORM, EPI>), grid, dim3(256), 0, st, a);
    else if (B == 3) hipLaunchKernelGGL((gemv_f16_ex_kernel<3, NORM, EPI>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemv_f16_ex_kernel<4, NORM, EPI>), grid, dim3(256), 0, st, a);
}

}  // namespace

extern "C" int gear_gemv_f16(const void* x, const void* W, int B, int K, int N, void* y, void* stream) {
    GEAR_CHECK_ARG(x && W && y, "gear_gemv_f16: null pointer");
    GEAR_CHECK_ARG(B >= 1 && B <= 4, "gear_gemv_f16: batch must be in [1,4] (got %d)", B);
    GEAR_CHECK_ARG(K > 0 && K % 8 == 0 && N > 0, "gear_gemv_f16: K=%d must be a positive multiple of 8", K);
    dim3 grid((N + 15) / 16), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (B == 1) hipLaunchKernelGGL(gemv_f16_kernel<1>, grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)W, (uint16_t*)y, K, N);
    else if (B == 2) hipLaunchKernelGGL(gemv_f16_kernel<2>, grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)W, (uint16_t*)y, K, N);
    else if (B == 3) hipLaunchKernelGGL(gemv_f16_kernel<3>, grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)W, (uint16_t*)y, K, N);
    else hipLaunchKernelGGL(gemv_f16_kernel<4>, grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)W, (uint16_t*)y, K, N);
    GEAR_CHECK_LAUNCH("gear_gemv_f16");
    return 0;
}


extern "C" int gear_gemv_f16_norm(const void* x, const void* delta, const void* norm_w, float eps, const void* W, int B,
                                  int K, int N, int swiglu, void* res_out, void* y, void* stream) {
    GEAR_CHECK_ARG(x && norm_w && W && y, "gear_gemv_f16_norm: null pointer");
    GEAR_CHECK_ARG(B >= 1 && B <= 4, "gear_gemv_f16_norm: batch must be in [1,4] (got %d)", B);
    GEAR_CHECK_ARG(K > 0 && K % 8 == 0 && N > 0, "gear_gemv_f16_norm: K=%d must be a positive multiple of 8", K);
    GEAR_CHECK_ARG(!swiglu || N % 4 == 0, "gear_gemv_f16_norm: SwiGLU needs interleaved (gate, up) rows, N %% 4 == 0");
    GEAR_CHECK_ARG(!delta || (res_out && res_out != x && res_out != delta),
                   "gear_gemv_f16_norm: the new residual needs its own buffer (other workgroups still read the inputs)");
    GemvEx a = {};
    a.x = (const uint16_t*)x; a.delta = (const uint16_t*)delta; a.nw = (const uint16_t*)norm_w;
    a.res_out = delta ? (uint16_t*)res_out : nullptr;
    a.eps = eps; a.W = (const uint16_t*)W; a.y = (uint16_t*)y; a.K = K; a.N = N;
    dim3 grid((N + 15) / 16);
    if (swiglu) launch_ex<true, 1>(B, grid, (hipStream_t)stream, a);
    else launch_ex<true, 0>(B, grid, (hipStream_t)stream, a);
    GEAR_CHECK_LAUNCH("gear_gemv_f16_norm");
    return 0;
}

extern "C" int gear_gemv_qkv_rope(const void* x, const void* delta, const void* norm_w, float eps, const void* Wqkv, int B,
                                  int K, int Hq, int Hkv, int D, int pos, int slot, int wcap, float theta,
                                  const void* dyn_state, void* res_out, void* q_out, void* kwin, void* vwin, void* stream) {
    GEAR_CHECK_ARG(D == 128, "gear_gemv_qkv_rope: head_dim must be 128 (got %d)", D);
    GEAR_CHECK_ARG(x && norm_w && Wqkv && q_out && kwin && vwin, "gear_gemv_qkv_rope: null pointer");
    GEAR_CHECK_ARG(B >= 1 && B <= 4, "gear_gemv_qkv_rope: batch must be in [1,4] (got %d)", B);
    GEAR_CHECK_ARG(K > 0 && K % 8 == 0 && Hq > 0 && Hkv > 0, "gear_gemv_qkv_rope: bad shape");
    GEAR_CHECK_ARG(dyn_state || (slot >= 0 && slot < wcap && pos >= 0), "gear_gemv_qkv_rope: bad pos / slot");
    GEAR_CHECK_ARG(!delta || (res_out && res_out != x && res_out != delta),
                   "gear_gemv_qkv_rope: the new residual needs its own buffer");
    GemvEx a = {};
    a.x = (const uint16_t*)x; a.delta = (const uint16_t*)delta; a.nw = (const uint16_t*)norm_w;
    a.res_out = delta ? (uint16_t*)res_out : nullptr;
    a.eps = eps; a.W = (const uint16_t*)Wqkv; a.K = K; a.N = (Hq + 2 * Hkv) * 128;
    a.Hq = Hq; a.Hkv = Hkv; a.pos = pos; a.slot = slot; a.wcap = wcap; a.log2_theta = log2f(theta);
    a.dyn = (const int*)dyn_state; a.q_out = (uint16_t*)q_out; a.kwin = (uint16_t*)kwin; a.vwin = (uint16_t*)vwin;
    dim3 grid(((Hq + 2 * Hkv) * 32 + 3) / 4);
    launch_ex<true, 2>(B, grid, (hipStream_t)stream, a);
    GEAR_CHECK_LAUNCH("gear_gemv_qkv_rope");
    return 0;
}
