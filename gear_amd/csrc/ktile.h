// ktile.h -- pieces of the fused K path that the single-launch block compressor (block_fused.hip) shares with kfused.hip:
// the element-by-element tile arithmetic of a 64-token x 128-channel tile held as lane = channel pair (both arithmetic modes),
// the LDS error-tile geometry and the matrix-core operand loads from it.
#pragma once
#include "common.h"

namespace {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef short short2v __attribute__((ext_vector_type(2)));

constexpr int KD = 128;          // head_dim
constexpr int ET_PITCH = 144;    // halfs per LDS row of an error tile [64 tokens][128 channels (+16 pad)]: 72 words, so the
                                 // 4 token rows x 4 x 8 bytes a 16-lane group gathers for ds_read_b64_tr_b16 hit 32 different banks

__device__ __forceinline__ uint32_t sort_key16(uint32_t hbits) {  // fp16 bits -> ascending-order key (16 bit); -0 == +0
    if (hbits == 0x8000u) hbits = 0u;
    return (hbits & 0x8000u) ? (~hbits & 0xFFFFu) : (hbits | 0x8000u);
}
// "larger = selected first": side 0 = the k largest values, side 1 = the k smallest
__device__ __forceinline__ uint32_t order_key(uint32_t hbits, int side) {
    const uint32_t kx = sort_key16(hbits);
    return side == 0 ? kx : 0xFFFFu - kx;
}

__device__ __forceinline__ uint32_t vbfi(uint32_t m, uint32_t a, uint32_t bb) {   // (a & m) | (bb & ~m)
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(bb));
    return r;
}
__device__ __forceinline__ uint32_t pkmin16(uint32_t a, uint32_t bb) {
    uint32_t r;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bb));
    return r;
}
__device__ __forceinline__ uint32_t pkmax16(uint32_t a, uint32_t bb) {
    uint32_t r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bb));
    return r;
}
__device__ __forceinline__ float fmin_raw(float a, float bb) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bb));
    return r;
}
__device__ __forceinline__ float fmax_raw(float a, float bb) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bb));
    return r;
}
__device__ __forceinline__ uint32_t pkminu16(uint32_t a, uint32_t bb) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bb));
    return r;
}
// One 64-token tile of one wave.  xr[i] = (channel 2*lane, channel 2*lane+1) of token i.
// Generic arithmetic (both modes): element by element, as compress_rows_kernel / quant_pack.hip do it.
template <int BITS, int MODE, int G, typename ST>
__device__ __forceinline__ void tile_generic(const uint32_t (&xr)[64], uint32_t mA0, uint32_t mA1, uint32_t mB0, uint32_t mB1,
                                             float meanA, float meanB, uint32_t (&ew)[64], uint32_t (&cwA)[64 * BITS / 32],
                                             uint32_t (&cwB)[64 * BITS / 32], float (&scA)[64 / G], float (&mnA)[64 / G],
                                             float (&scB)[64 / G], float (&mnB)[64 / G]) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int CPW = 32 / BITS;
    constexpr int NW = 64 / CPW;
#pragma unroll
    for (int w = 0; w < NW; w++) { cwA[w] = 0u; cwB[w] = 0u; }
#pragma unroll
    for (int i = 0; i < 64; i++) ew[i] = 0u;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t m0 = h ? mB0 : mA0, m1 = h ? mB1 : mA1;
        const float mean = h ? meanB : meanA;
        const float fill = (MODE == 0) ? hround(mean) : mean;
#pragma unroll
        for (int gi = 0; gi < 64 / G; gi++) {
            float v[G];
            float lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int i = 0; i < G; i++) {
                const int tk = gi * G + i;
                const bool o = ((tk < 32 ? m0 >> tk : m1 >> (tk - 32)) & 1u) != 0u;
                const float xv = h2f_bits((uint16_t)((xr[tk] >> (16 * h)) & 0xFFFFu));
                v[i] = o ? fill : xv;
                lo = fminf(lo, v[i]);
                hi = fmaxf(hi, v[i]);
            }
            QuantParams<MODE> qp = make_qparams<MODE>(lo, hi, LEVELS);
            const float inv = (qp.scale != 0.0f) ? div_rn(1.0f, qp.scale) : 0.0f;
            (h ? scB : scA)[gi] = qp.scale;
            (h ? mnB : mnA)[gi] = qp.mn;
#pragma unroll
            for (int i = 0; i < G; i++) {
                const int tk = gi * G + i;
                const bool o = ((tk < 32 ? m0 >> tk : m1 >> (tk - 32)) & 1u) != 0u;
                int qv;
                if (qp.scale == 0.0f) qv = 0;
                else if (MODE == 0) {
                    const float t1 = hround(v[i] - qp.mn);
                    float c = hround(div_rn(t1, qp.scale));
                    c = fminf(fmaxf(c, 0.0f), (float)LEVELS);
                    qv = (int)rintf(c);
                } else {
                    const float t = v[i] - qp.mn;
                    float c = t * inv;
                    float r = rintf(c);
                    if (fabsf(fabsf(c - r) - 0.5f) < 1e-5f) r = rintf(div_rn(t, qp.scale));
                    qv = (int)fminf(fmaxf(r, 0.0f), (float)LEVELS);
                }
                (h ? cwB : cwA)[tk / CPW] |= (uint32_t)qv << (BITS * (tk % CPW));
                const float d = (MODE == 0) ? dequant_one<0>(qv, qp.scale, qp.mn) : hround(dequant_one<1>(qv, qp.scale, qp.mn));
                const float e = o ? 0.0f : (v[i] - d);
                ew[tk] |= (uint32_t)f2h_bits(e) << (16 * h);
            }
        }
    }
}

// MFMA operand of lane (x31 = lane & 31, kg = lane >> 5): channel 32 I + x31, tokens t0 + 8 kg .. + 7 of an error tile
// [64 tokens][ET_PITCH] in LDS.  TR: two ds_read_b64_tr_b16 -- inside a 16-lane group lane i supplies the address of 4
// consecutive channels of token row i / 4 and receives column i of the [4 tokens][16 channels] block (the hardware's 4x4
// transpose); otherwise eight 16-bit reads.
template <bool TR>
__device__ __forceinline__ half8_t load_operand(const uint16_t* etile, int t0, int I, int lane) {
    const int x31 = lane & 31, kg = lane >> 5;
    if (TR) {
        const int i = lane & 15, c0 = 32 * I + 16 * ((lane >> 4) & 1);
        const uint16_t* p = etile + (t0 + 8 * kg + (i >> 2)) * ET_PITCH + c0 + 4 * (i & 3);
        const uint32_t addr = (uint32_t)(uintptr_t)p;      // LDS byte address (the low 32 bits of a shared pointer)
        typedef short short4v __attribute__((ext_vector_type(4)));
        short4v lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * ET_PITCH * 2) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        union { half8_t h; short4v s[2]; } cv;
        cv.s[0] = lo;
        cv.s[1] = hi;
        return cv.h;
    } else {
        union { half8_t h; uint16_t u[8]; } cv;
#pragma unroll
        for (int j = 0; j < 8; j++) cv.u[j] = etile[(t0 + 8 * kg + j) * ET_PITCH + 32 * I + x31];
        return cv.h;
    }
}

}  // namespace
