// dense16.h -- the dense half of the fp32-arithmetic row compressors as one function: a lane's 16 consecutive elements of a
// row after the outlier selection -> group min / max (with the mean fill), quantize, pack, error, stores.  Shared by
// compress_rows.hip (workgroup and wave-per-row kernels) and rows_multi.hip (short rows, several per wave): same instructions,
// same bits.  The single-instruction helpers it stands on come first.
#pragma once
#include "common.h"

namespace {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bfi32(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

// flag bits (one per element) -> BITS bits per element, for the CPW elements of one packed word
template <int BITS>
__device__ __forceinline__ uint32_t spread_flags(uint32_t f) {
    if (BITS == 2) {
        uint32_t x = f & 0xFFFFu;
        x = (x | (x << 8)) & 0x00FF00FFu;
        x = (x | (x << 4)) & 0x0F0F0F0Fu;
        x = (x | (x << 2)) & 0x33333333u;
        x = (x | (x << 1)) & 0x55555555u;
        return x * 3u;
    } else if (BITS == 4) {
        uint32_t x = f & 0xFFu;
        x = (x | (x << 12)) & 0x000F000Fu;
        x = (x | (x << 6)) & 0x03030303u;
        x = (x | (x << 3)) & 0x11111111u;
        return x * 15u;
    } else {
        uint32_t x = f & 0xFu;
        x = (x | (x << 14)) & 0x00030003u;
        x = (x | (x << 7)) & 0x01010101u;
        return x * 255u;
    }
}

// single-instruction forms the compiler does not pick by itself here: v_bfi_b32 for (a & m) | (b & ~m) (it shares b & ~m
// between the +inf and -inf variants: 4 instructions for two results), packed / scalar min and max WITHOUT the
// canonicalising max(x, x) that IEEE minNum semantics put in front of every operand (inputs are never signalling NaNs here:
// they are loaded fp16 data or +-inf constants), and the fp16 -> fp32 convert folded into the subtract (v_fma_mix_f32).
__device__ __forceinline__ uint32_t vbfi(uint32_t m, uint32_t a, uint32_t bb) {
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
// float(half HI ? high : low of w) - mn, one rounding (the convert is exact)
template <int HI>
__device__ __forceinline__ float sub_mix(uint32_t w, float one, float negmn) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(one), "v"(negmn));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(one), "v"(negmn));
    return r;
}

// ------------------------------------------------------------------------------------------------------------------
// One lane's 16 consecutive elements of a row, after the selection: group min / max (with the mean fill), quantize, pack,
// error -- the dense half of compress_rows_fp32_kernel as a function (same instructions, same bits), used by the
// wave-per-row kernel below once per 1024-element chunk of the row.
template <int BITS>
__device__ __forceinline__ void dense16(const uint32_t (&rw)[8], const uint32_t (&m)[8], uint32_t outl, float mean, int group,
                                        int group_shift, int lane, uint32_t* __restrict__ code_row, float* __restrict__ scale_row,
                                        float* __restrict__ mn_row, uint32_t ooff, uint16_t* __restrict__ err_row, uint32_t loff) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int WPL = BITS / 2;
    constexpr int CPW = 32 / BITS;
    constexpr int HC = 16 / BITS;
    // ---------------- group min / max over the elements that are not outliers (packed fp16 min/max are exact), plus the
    // fill value (the fp32 row mean, compress_function.py:279-283 / :315-319) when the lane holds an outlier
    uint32_t lo2, hi2;
    {
        const uint32_t PINF = 0x7C007C00u, NINF = 0xFC00FC00u;
        lo2 = vbfi(m[0], PINF, rw[0]);
        hi2 = vbfi(m[0], NINF, rw[0]);
#pragma unroll
        for (int w = 1; w < 8; w++) {
            lo2 = pkmin16(lo2, vbfi(m[w], PINF, rw[w]));
            hi2 = pkmax16(hi2, vbfi(m[w], NINF, rw[w]));
        }
    }
    float lo = fmin_raw(h2f_bits((uint16_t)(lo2 & 0xFFFFu)), h2f_bits((uint16_t)(lo2 >> 16)));
    float hi = fmax_raw(h2f_bits((uint16_t)(hi2 & 0xFFFFu)), h2f_bits((uint16_t)(hi2 >> 16)));
    lo = fmin_raw(lo, outl ? mean : INFINITY);
    hi = fmax_raw(hi, outl ? mean : -INFINITY);
    const int lanes_per_group = group / 16;
    if (lanes_per_group == 4) {   // the usual group of 64: the four lanes of a DPP quad
        lo = fmin_raw(lo, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, lo), 0xB1, 0xF, 0xF, true)));
        hi = fmax_raw(hi, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hi), 0xB1, 0xF, 0xF, true)));
        lo = fmin_raw(lo, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, lo), 0x4E, 0xF, 0xF, true)));
        hi = fmax_raw(hi, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hi), 0x4E, 0xF, 0xF, true)));
    } else {
        for (int mm = 1; mm < lanes_per_group; mm <<= 1) {
            lo = fminf(lo, __shfl_xor(lo, mm, 64));
            hi = fmaxf(hi, __shfl_xor(hi, mm, 64));
        }
    }
    const float qscale = div_rn(hi - lo, (float)LEVELS), qmn = lo;        // make_qparams<1>
    // (v_rcp_f32 is within 1 ulp: far inside the 1e-5 tie guard below.)  Zero-range group: every code 0 (defect B6)
    const float inv = (qscale != 0.0f) ? __builtin_amdgcn_rcpf(qscale) : 0.0f;
    // ---------------- quantize: reciprocal multiply; a lane that sees a quotient within 1e-5 of a rounding tie (the only
    // place where t * (1/s) and t / s can round differently; 1e-3 for 8-bit codes) redoes its elements by division
    constexpr float TIE = (BITS == 8) ? 0.499f : 0.49999f;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v rq[8];                                       // codes as floats, (even, odd) element pairs: one register pair per word
    bool tie = false;
    const float one = 1.0f, negmn = -qmn;
    float dmax = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        float2v t = {sub_mix<0>(rw[w], one, negmn), sub_mix<1>(rw[w], one, negmn)};
        const float2v c = t * inv;                       // v_pk_mul_f32
        float2v rr = {rintf(c.x), rintf(c.y)};
        const float2v d = c - rr;                        // v_pk_add_f32
        rq[w] = rr;
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(dmax) : "v"(dmax), "v"(d.x), "v"(d.y));
    }
    tie = dmax > TIE;
    if (tie) {
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const float xa = h2f_bits((uint16_t)(rw[w] & 0xFFFFu)), xb = h2f_bits((uint16_t)(rw[w] >> 16));
            rq[w].x = (qscale != 0.0f) ? rintf(div_rn(xa - qmn, qscale)) : 0.0f;
            rq[w].y = (qscale != 0.0f) ? rintf(div_rn(xb - qmn, qscale)) : 0.0f;
        }
    }
#pragma unroll
    for (int w = 0; w < 8; w++) {
        rq[w].x = __builtin_amdgcn_fmed3f(rq[w].x, 0.0f, (float)LEVELS);
        rq[w].y = __builtin_amdgcn_fmed3f(rq[w].y, 0.0f, (float)LEVELS);
    }
    // ---------------- pack: Horner chains in fp32 over the HC codes of each 16-bit half (exact: < 2^16)
    // (even, odd) element pairs ride one v_pk_fma_f32: A = sum 4^BITS^i code[2i], B = the same over the odd elements,
    // half word = A + 2^BITS B
    uint32_t words[WPL];
#pragma unroll
    for (int w = 0; w < WPL; w++) {
        uint32_t hw[2];
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const int p0 = (w * CPW + hf * HC) / 2;          // first element pair of the half word
            float2v ab = rq[p0 + HC / 2 - 1];
            const float2v base2 = {(float)(1 << (2 * BITS)), (float)(1 << (2 * BITS))};
#pragma unroll
            for (int i = HC / 2 - 2; i >= 0; i--) ab = __builtin_elementwise_fma(ab, base2, rq[p0 + i]);
            hw[hf] = (uint32_t)fmaf(ab.y, (float)(1 << BITS), ab.x);
        }
        words[w] = hw[0] | (hw[1] << 16);
    }
    if (outl) {   // filled positions: every outlier of the group carries quant(mean)
        const float cq = (mean - qmn) * inv;
        float cm = rintf(cq);
        if (fabsf(cq - cm) > TIE) cm = (qscale != 0.0f) ? rintf(div_rn(mean - qmn, qscale)) : 0.0f;
        cm = __builtin_amdgcn_fmed3f(cm, 0.0f, (float)LEVELS);
        const uint32_t qrep = (uint32_t)cm * (0xFFFFFFFFu / (uint32_t)LEVELS);
#pragma unroll
        for (int w = 0; w < WPL; w++) words[w] = bfi32(spread_flags<BITS>(outl >> (w * CPW)), qrep, words[w]);
    }
    uint32_t* cp = code_row + ooff / CPW;
#pragma unroll
    for (int w = 0; w < WPL; w++) cp[w] = words[w];
    if ((lane & (lanes_per_group - 1)) == 0) {
        scale_row[ooff >> group_shift] = qscale;
        mn_row[ooff >> group_shift] = qmn;
    }
    if (err_row) {
        // error = x - fp16(code * scale + mn) (mul then add, unfused, like the reference), 0 at the outlier positions
        uint32_t ew[8];
        const float2v qs2 = {qscale, qscale}, mn2 = {qmn, qmn};
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const float2v dq = rq[w] * qs2 + mn2;            // -ffp-contract=off: v_pk_mul_f32 then v_pk_add_f32 (two roundings)
            const uint32_t dw = f2h2_bits(dq.x, dq.y);
            uint32_t e2;   // x - d in packed fp16 (the optimiser otherwise negates d in fp32 first), outlier halves cleared
            asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e2) : "v"(rw[w]), "v"(dw));
            ew[w] = vbfi(m[w], 0u, e2);
        }
        uint4* ep = (uint4*)(err_row + loff);
        ep[0] = make_uint4(ew[0], ew[1], ew[2], ew[3]);
        ep[1] = make_uint4(ew[4], ew[5], ew[6], ew[7]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dense16s: dense16 for a row whose outliers have been REPLACED by fp16(row mean) in the staged copy (one 2-byte LDS store per
// outlier, where dense16's callers store a 0xFFFF mark): no mask anywhere in the common path -- plain packed min / max
// (three-operand forms), no clamp (every element, the substitutes included, lies inside [mn, mx]), the quotient rounded inside
// a fused multiply-add (1.5 * 2^23 addend: rint of the exact product) with a second one for the distance to the tie, the
// outlier slots' error cleared by 2-byte stores behind the row's 16-byte ones.  `outl` = the lane's 16 outlier flags, `sv` = the
// substitute's bits.  Same bits as dense16: a group in which the substitute is itself the minimum or maximum (then the
// non-outlier range is not what the substituted data shows) sends the whole WAVE through dense16 with masks rebuilt from the flags.
template <int BITS>
__device__ __forceinline__ void dense16s(const uint32_t (&rs)[8], uint32_t outl, float mean, uint32_t sv, int group, int group_shift,
                                         int lane, uint32_t* __restrict__ code_row, float* __restrict__ scale_row,
                                         float* __restrict__ mn_row, uint32_t ooff, uint16_t* __restrict__ err_row, uint32_t loff) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int WPL = BITS / 2;
    constexpr int CPW = 32 / BITS;
    constexpr int HC = 16 / BITS;
    typedef float float2v __attribute__((ext_vector_type(2)));
    uint32_t lo2 = rs[0], hi2 = rs[0];
#pragma unroll
    for (int w = 1; w < 7; w += 2) {
        asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(lo2) : "v"(lo2), "v"(rs[w]), "v"(rs[w + 1]));
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(hi2) : "v"(hi2), "v"(rs[w]), "v"(rs[w + 1]));
    }
    lo2 = pkmin16(lo2, rs[7]);
    hi2 = pkmax16(hi2, rs[7]);
    float lo = fmin_raw(h2f_bits((uint16_t)(lo2 & 0xFFFFu)), h2f_bits((uint16_t)(lo2 >> 16)));
    float hi = fmax_raw(h2f_bits((uint16_t)(hi2 & 0xFFFFu)), h2f_bits((uint16_t)(hi2 >> 16)));
    float gf = outl ? 1.0f : 0.0f;                        // does the group hold an outlier?
    const int lanes_per_group = group / 16;
    if (lanes_per_group == 4) {   // the usual group of 64: the four lanes of a DPP quad
#define GEAR_D16_DPP(v, ctl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctl, 0xF, 0xF, true))
        lo = fmin_raw(lo, GEAR_D16_DPP(lo, 0xB1)); hi = fmax_raw(hi, GEAR_D16_DPP(hi, 0xB1)); gf = fmax_raw(gf, GEAR_D16_DPP(gf, 0xB1));
        lo = fmin_raw(lo, GEAR_D16_DPP(lo, 0x4E)); hi = fmax_raw(hi, GEAR_D16_DPP(hi, 0x4E)); gf = fmax_raw(gf, GEAR_D16_DPP(gf, 0x4E));
#undef GEAR_D16_DPP
    } else {
        for (int mm = 1; mm < lanes_per_group; mm <<= 1) {
            lo = fminf(lo, __shfl_xor(lo, mm, 64));
            hi = fmaxf(hi, __shfl_xor(hi, mm, 64));
            gf = fmaxf(gf, __shfl_xor(gf, mm, 64));
        }
    }
    const bool g = gf != 0.0f;
    {
        const float fs = h2f_bits((uint16_t)sv);
        if (__any(g && (lo == fs || hi == fs))) {
            uint32_t m[8];
#pragma unroll
            for (int w = 0; w < 8; w++) m[w] = (((outl >> (2 * w)) & 1u) ? 0x0000FFFFu : 0u) | (((outl >> (2 * w + 1)) & 1u) ? 0xFFFF0000u : 0u);
            dense16<BITS>(rs, m, outl, mean, group, group_shift, lane, code_row, scale_row, mn_row, ooff, err_row, loff);
            return;
        }
    }
    lo = fmin_raw(lo, g ? mean : INFINITY);
    hi = fmax_raw(hi, g ? mean : -INFINITY);
    const float qscale = div_rn(hi - lo, (float)LEVELS), qmn = lo;        // make_qparams<1>
    const float inv = (qscale != 0.0f) ? __builtin_amdgcn_rcpf(qscale) : 0.0f;
    constexpr float TIE = (BITS == 8) ? 0.499f : 0.49999f;
    float2v rq[8];
    const float one = 1.0f, negmn = -qmn;
    const float2v inv2 = {inv, inv}, magic2 = {12582912.0f, 12582912.0f};
    float dmax = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const float2v t = {sub_mix<0>(rs[w], one, negmn), sub_mix<1>(rs[w], one, negmn)};
        const float2v sb = __builtin_elementwise_fma(t, inv2, magic2);
        const float2v rr = sb - magic2;
        const float2v d = __builtin_elementwise_fma(t, inv2, -rr);
        rq[w] = rr;
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(dmax) : "v"(dmax), "v"(d.x), "v"(d.y));
    }
    if (dmax > TIE) {
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const float xa = h2f_bits((uint16_t)(rs[w] & 0xFFFFu)), xb = h2f_bits((uint16_t)(rs[w] >> 16));
            rq[w].x = (qscale != 0.0f) ? rintf(div_rn(xa - qmn, qscale)) : 0.0f;
            rq[w].y = (qscale != 0.0f) ? rintf(div_rn(xb - qmn, qscale)) : 0.0f;
        }
    }
    uint32_t words[WPL];
#pragma unroll
    for (int w = 0; w < WPL; w++) {
        uint32_t hw[2];
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const int p0 = (w * CPW + hf * HC) / 2;          // first element pair of the half word
            float2v ab = rq[p0 + HC / 2 - 1];
            const float2v base2 = {(float)(1 << (2 * BITS)), (float)(1 << (2 * BITS))};
#pragma unroll
            for (int i = HC / 2 - 2; i >= 0; i--) ab = __builtin_elementwise_fma(ab, base2, rq[p0 + i]);
            hw[hf] = (uint32_t)fmaf(ab.y, (float)(1 << BITS), ab.x);
        }
        words[w] = hw[0] | (hw[1] << 16);
    }
    if (outl) {   // filled positions: every outlier of the group carries quant(mean)
        const float cq = (mean - qmn) * inv;
        float cm = rintf(cq);
        if (fabsf(cq - cm) > TIE) cm = (qscale != 0.0f) ? rintf(div_rn(mean - qmn, qscale)) : 0.0f;
        cm = __builtin_amdgcn_fmed3f(cm, 0.0f, (float)LEVELS);
        const uint32_t qrep = (uint32_t)cm * (0xFFFFFFFFu / (uint32_t)LEVELS);
#pragma unroll
        for (int w = 0; w < WPL; w++) words[w] = bfi32(spread_flags<BITS>(outl >> (w * CPW)), qrep, words[w]);
    }
    uint32_t* cp = code_row + ooff / CPW;
#pragma unroll
    for (int w = 0; w < WPL; w++) cp[w] = words[w];
    if ((lane & (lanes_per_group - 1)) == 0) {
        scale_row[ooff >> group_shift] = qscale;
        mn_row[ooff >> group_shift] = qmn;
    }
    if (err_row) {
        uint32_t ew[8];
        const float2v qs2 = {qscale, qscale}, mn2 = {qmn, qmn};
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const float2v dq = rq[w] * qs2 + mn2;            // -ffp-contract=off: v_pk_mul_f32 then v_pk_add_f32 (two roundings)
            const uint32_t dw = f2h2_bits(dq.x, dq.y);
            asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(ew[w]) : "v"(rs[w]), "v"(dw));
        }
        uint4* ep = (uint4*)(err_row + loff);
        ep[0] = make_uint4(ew[0], ew[1], ew[2], ew[3]);
        ep[1] = make_uint4(ew[4], ew[5], ew[6], ew[7]);
        // (same thread, same addresses, program order: the zeros land on top of the row's stores)
        for (uint32_t mm = outl; mm; mm &= mm - 1u) err_row[loff + (uint32_t)__builtin_ctz(mm)] = (uint16_t)0u;
    }
}

}  // namespace
