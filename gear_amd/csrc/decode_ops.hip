// decode_ops.hip -- the small per-token glue ops of a decode step, one launch each (gfx950):
//   gear_rope_append : RoPE on the new token's q / k, q out, k and v appended to the fp16 residual window
//   gear_add_rmsnorm : residual add + RMSNorm (the Llama pre-norm), both results written
//   gear_silu_mul    : SwiGLU gate
// They replace the ~25 eager torch launches per layer that the attention hook's caller issues around the cache
// (cuda_supported_gear/modeling_llamagear.py:193-205 q/k/v views + rotary, :502-560 decoder layer); arithmetic follows
// torch eager on fp16 tensors (each elementwise op rounded to fp16) so that results stay within fp16 rounding of it.
#include <math.h>

#include "common.h"

namespace {

// qkv: [B, (Hq + 2 Hkv) * 128] = [q heads | k heads | v heads]
__global__ __launch_bounds__(256) void rope_append_kernel(const uint16_t* __restrict__ qkv, int B, int Hq, int Hkv, int pos,
                                                          float log2_theta, uint16_t* __restrict__ q_out,
                                                          uint16_t* __restrict__ kwin, uint16_t* __restrict__ vwin,
                                                          int slot, int W, const int* __restrict__ dyn) {
    if (dyn) { pos = dyn[0]; slot = dyn[1]; }   // device-side state {pos, slot, T, W} (hipGraph replay)
    const int HT = Hq + 2 * Hkv;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (b, head, pair index 0..63)
    if (i >= (int64_t)B * HT * 64) return;
    const int p = (int)(i % 64);
    const int h = (int)((i / 64) % HT);
    const int b = (int)(i / (64 * HT));
    const uint16_t* src = qkv + ((int64_t)b * HT + h) * 128;
    if (h >= Hq + Hkv) {  // V head: plain copy of two elements
        const int hv = h - Hq - Hkv;
        uint16_t* dst = vwin + (((int64_t)b * Hkv + hv) * W + slot) * 128;
        dst[p] = src[p];
        dst[p + 64] = src[p + 64];
        return;
    }
    // inv_freq = theta^(-2p/128); HF computes cos/sin in fp32 and casts them to fp16
    const float inv_freq = exp2f(-(float)(2 * p) / 128.0f * log2_theta);
    const float ang = (float)pos * inv_freq;
    const float c = hround(cosf(ang)), sn = hround(sinf(ang));
    const float x1 = h2f_bits(src[p]), x2 = h2f_bits(src[p + 64]);
    const float lo = hround(hround(x1 * c) + hround(-x2 * sn));   // q*cos + rotate_half(q)*sin, fp16 op by op
    const float hi = hround(hround(x2 * c) + hround(x1 * sn));
    uint16_t* dst = (h < Hq) ? q_out + ((int64_t)b * Hq + h) * 128
                             : kwin + (((int64_t)b * Hkv + (h - Hq)) * W + slot) * 128;
    dst[p] = f2h_bits(lo);
    dst[p + 64] = f2h_bits(hi);
}

// res_out = res_in + delta (fp16 add; delta may be null) ; y = weight * fp16(res_out * rsqrt(mean(res_out^2) + eps))
// One block per row; every thread keeps its 16-byte vectors (up to 4 -> H <= 8192) in registers between the two passes.
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(const uint16_t* __restrict__ res_in, const uint16_t* __restrict__ delta,
                                                          const uint16_t* __restrict__ weight, int H, float eps,
                                                          uint16_t* __restrict__ res_out, uint16_t* __restrict__ y) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int nvec = H / 8;
    float v[4][8];
    float ss = 0.0f;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int iv = threadIdx.x + 256 * p;
        if (iv < nvec) {
            unpack8(*(const uint4*)(res_in + row * H + iv * 8), v[p]);
            if (delta) {
                float d[8];
                unpack8(*(const uint4*)(delta + row * H + iv * 8), d);
#pragma unroll
                for (int j = 0; j < 8; j++) v[p][j] = hround(v[p][j] + d[j]);
            }
            if (res_out) *(uint4*)(res_out + row * H + iv * 8) = pack8(v[p]);
#pragma unroll
            for (int j = 0; j < 8; j++) ss = fmaf(v[p][j], v[p][j], ss);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int iv = threadIdx.x + 256 * p;
        if (iv < nvec) {
            float w[8], o[8];
            unpack8(*(const uint4*)(weight + iv * 8), w);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = w[j] * hround(v[p][j] * inv);
            *(uint4*)(y + row * H + iv * 8) = pack8(o);
        }
    }
}

// out[b, i] = fp16(silu(gate)) * up,  gate_up = [B, 2 I] = [gate | up]
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ gate_up, int64_t B, int I,
                                                       uint16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * I) return;
    const int64_t b = i / I, c = i % I;
    const float g = h2f_bits(gate_up[b * 2 * I + c]), u = h2f_bits(gate_up[b * 2 * I + I + c]);
    const float s = hround(g / (1.0f + expf(-g)));
    out[i] = f2h_bits(s * u);
}

}  // namespace

extern "C" int gear_rope_append(const void* qkv, int B, int Hq, int Hkv, int D, int pos, float theta, void* q_out,
                                void* kwin, void* vwin, int slot, int W, void* stream) {
    GEAR_CHECK_ARG(D == 128, "gear_rope_append: head_dim must be 128 (got %d)", D);
    GEAR_CHECK_ARG(qkv && q_out && kwin && vwin, "gear_rope_append: null pointer");
    GEAR_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && slot >= 0 && slot < W && pos >= 0, "gear_rope_append: bad arguments");
    const int64_t n = (int64_t)B * (Hq + 2 * Hkv) * 64;
    hipLaunchKernelGGL(rope_append_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)qkv, B, Hq, Hkv, pos, log2f(theta), (uint16_t*)q_out, (uint16_t*)kwin, (uint16_t*)vwin,
                       slot, W, (const int*)nullptr);
    GEAR_CHECK_LAUNCH("gear_rope_append");
    return 0;
}

extern "C" int gear_rope_append_dyn(const void* qkv, int B, int Hq, int Hkv, int D, const void* dyn_state, float theta,
                                    void* q_out, void* kwin, void* vwin, int W, void* stream) {
    GEAR_CHECK_ARG(D == 128, "gear_rope_append_dyn: head_dim must be 128 (got %d)", D);
    GEAR_CHECK_ARG(qkv && q_out && kwin && vwin && dyn_state, "gear_rope_append_dyn: null pointer");
    const int64_t n = (int64_t)B * (Hq + 2 * Hkv) * 64;
    hipLaunchKernelGGL(rope_append_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)qkv, B, Hq, Hkv, 0, log2f(theta), (uint16_t*)q_out, (uint16_t*)kwin, (uint16_t*)vwin,
                       0, W, (const int*)dyn_state);
    GEAR_CHECK_LAUNCH("gear_rope_append_dyn");
    return 0;
}

__global__ void decode_state_advance_kernel(int* st) {  // one token appended: pos, slot and W move on
    st[0] += 1;
    st[1] += 1;
    st[3] += 1;
}

extern "C" int gear_decode_state_advance(void* state, void* stream) {
    GEAR_CHECK_ARG(state, "gear_decode_state_advance: null pointer");
    hipLaunchKernelGGL(decode_state_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)state);
    GEAR_CHECK_LAUNCH("gear_decode_state_advance");
    return 0;
}

extern "C" int gear_add_rmsnorm(const void* res_in, const void* delta, const void* weight, int64_t rows, int H, float eps,
                                void* res_out, void* y, void* stream) {
    GEAR_CHECK_ARG(res_in && weight && y && rows > 0 && H > 0, "gear_add_rmsnorm: bad arguments");
    GEAR_CHECK_ARG(H % 8 == 0 && H <= 8192, "gear_add_rmsnorm: hidden size %d must be a multiple of 8 and <= 8192", H);
    hipLaunchKernelGGL(add_rmsnorm_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)res_in,
                       (const uint16_t*)delta, (const uint16_t*)weight, H, eps, (uint16_t*)res_out, (uint16_t*)y);
    GEAR_CHECK_LAUNCH("gear_add_rmsnorm");
    return 0;
}

extern "C" int gear_silu_mul(const void* gate_up, int64_t B, int I, void* out, void* stream) {
    GEAR_CHECK_ARG(gate_up && out && B > 0 && I > 0, "gear_silu_mul: bad arguments");
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)((B * I + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)gate_up, B, I, (uint16_t*)out);
    GEAR_CHECK_LAUNCH("gear_silu_mul");
    return 0;
}
