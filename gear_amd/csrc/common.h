// common.h -- shared device/host helpers for libgear_hip.so (gfx950 only; wave64).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include "../../include/gear_hip.h"

#define GEAR_WAVE 64

// ------------------------------------------------------------------ host side error plumbing
void gear_set_error(const char* fmt, ...);

#define GEAR_CHECK_ARG(cond, ...)         \
    do {                                  \
        if (!(cond)) {                    \
            gear_set_error(__VA_ARGS__);  \
            return -1;                    \
        }                                 \
    } while (0)

#define GEAR_CHECK_LAUNCH(name)                                                     \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            gear_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return -2;                                                              \
        }                                                                           \
    } while (0)

// run-time options (api.hip): read from the environment once at first use, settable through gear_set_option()
struct GearOptions {
    int attn_generic;      // decode attention: the generic variable-chunk kernel instead of the 128-token-chunk one
    int lowrank_generic;   // power iteration: the generic multi-pass kernels instead of the Gram formulation
    int rows_hist_only;    // row compressor: always the radix-select (exact fallback) selection
    int rows_wg_only;      // row compressor: never the wave-per-row kernel (the workgroup kernel for every row)
    int rows_v1;           // row compressor: first-generation kernel also for fp32 arithmetic (cross-check)
    int rows_masked;       // wave-per-row compressor, dense part: outlier masks (dense16) or the substituted row (dense16s, round 6: 6 % faster
                           // without an error matrix, 5 % slower with one -- its 2-byte zero stores): 0 = substituted when no error
                           // matrix is written, 1 = always masks, -1 = always substituted; same bits
    int kfused_generic;    // fused K path: the element-by-element tile body instead of the packed one
    int kselect_slow;      // fused K path: always the exact slow selection (no candidate lists)
    int kfused_no_tr;      // fused K path: 16-bit LDS reads for the MFMA operands instead of ds_read_b64_tr_b16
    int gram_fused;        // low-rank step: 1 = Gram matrix and solve in ONE kernel per head (round 1-3); 2 = slab kernels with
                           // workgroup barriers + solve; 0 = wave-private slab kernel + solve
    int gram_nstg;         // wave-private Gram kernel: steps of loads in flight per wave (2, 3 = default, 4)
    int decomp_general;    // row decompressor: the general row loop also for full blocks (never the straight-line 16-row path)
    int attn_gqa_group;    // decode attention, Hq > Hkv: 1 = ONE workgroup serves the 2 / 4 / 8 query heads of a KV head (the chunk's payload
                           // loaded once); 0 = one workgroup per query head (default: measured 2 - 3 x faster up to batch 4, profiles/
                           // r5_attn_experiments.md -- the repeated reads hit L2 and the chip wants the parallelism); -1 = by launch size
    int kfused_nslab;      // fused K path: slabs per head of k_main_kernel (0 = by head count)
    int attn_fold;         // decode attention (vector short-chunk kernel, fp16 baseline): the merge of a head's partial results folded into
                           // the partial launch (last-arriving workgroup merges): 1 = on; off by default (measured no faster than the
                           // reduce kernel as a second launch: csrc/attention.hip, fold_wanted)
    int decomp_rpb;        // row decompressor, rank 4 / 8 on whole waves: rows per workgroup behind one load of the column factor block:
                           // 0 = 128 / 64 / 32 while >= 1024 workgroups remain, else 16; 32 / 64 / 128 = that many whatever the size; -1 = 16
    int kfused_eout;       // fused K path, k_dense_kernel: 1 = the error matrix written to the workspace (+ BH T 256 bytes) and the Q pass reading it
                           // (lr_qpass_tm_mfma_kernel) instead of rebuilding it from K and the stored codes (k_qpass_kernel)
    int kfused_main;       // fused K path, fp32 arithmetic: 1 = k_main_kernel (register-resident tiles with outlier masks, rounds 2 - 5) instead of
                           // k_dense_kernel (LDS-resident slabs, substituted outliers, round 6)
    int kfused_one;        // fused K path, fp32 arithmetic: the single-read kernel (kone.hip: selection + dense part + Gram in one launch):
                           // 1 = wherever its plan fits; 0 / -1 = never (select + main as two kernels: faster, profiles/r6_kone.md)
    int attn_mfma;         // decode attention, matrix-core variant of the short-chunk kernel: 0 = for grouped-query shapes with a
                           // workgroup per CU (measured faster there), 1 = whenever it applies (tests, A/B runs), -1 = never
    int attn_keep_chunk_index;  // decode attention over a cache with sparse tiles: also load the outlier chunk indices (round 4)
    int attn_win_chunk;    // decode attention: the fp16 window as one more chunk of the split: 0 = when the vector short-chunk kernel runs
                           // (faster since every chunk of a head runs on one XCD: 21.0 -> 20.0 us per layer at batch 1), 1 = whenever
                           // the short-chunk kernel runs, -1 = never (the window inside the reduce kernel: rounds 1 - 4)
};
GearOptions& gear_options();

static inline bool gear_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ------------------------------------------------------------------ device helpers
// round a float to the nearest fp16 value (RNE) and come back: one torch-eager fp16 op boundary
__device__ __forceinline__ float hround(float f) { return __half2float(__float2half_rn(f)); }

__device__ __forceinline__ float h2f_bits(uint16_t b) {
    __half_raw r;
    r.x = b;
    return __half2float(__half(r));
}
__device__ __forceinline__ uint16_t f2h_bits(float f) {
    __half h = __float2half_rn(f);
    return __half_as_ushort(h);
}

// two floats -> packed fp16 pair (lo in bits 0..15), round to nearest even: ONE instruction on gfx950 (v_cvt_pk_f16_f32)
// where two converts + a pack / SDWA write take two or three
__device__ __forceinline__ uint32_t f2h2_bits(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// IEEE-correct fp32 division (never the v_rcp approximation): torch semantics
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }

// unpack 8 fp16 from a 16-byte vector
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = h2f_bits((uint16_t)(v.x & 0xFFFFu));
    f[1] = h2f_bits((uint16_t)(v.x >> 16));
    f[2] = h2f_bits((uint16_t)(v.y & 0xFFFFu));
    f[3] = h2f_bits((uint16_t)(v.y >> 16));
    f[4] = h2f_bits((uint16_t)(v.z & 0xFFFFu));
    f[5] = h2f_bits((uint16_t)(v.z >> 16));
    f[6] = h2f_bits((uint16_t)(v.w & 0xFFFFu));
    f[7] = h2f_bits((uint16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = (uint32_t)f2h_bits(f[0]) | ((uint32_t)f2h_bits(f[1]) << 16);
    v.y = (uint32_t)f2h_bits(f[2]) | ((uint32_t)f2h_bits(f[3]) << 16);
    v.z = (uint32_t)f2h_bits(f[4]) | ((uint32_t)f2h_bits(f[5]) << 16);
    v.w = (uint32_t)f2h_bits(f[6]) | ((uint32_t)f2h_bits(f[7]) << 16);
    return v;
}

// scale / zero-point storage type: fp16 (mode 0) or float (mode 1)
template <typename ST>
__device__ __forceinline__ float ld_st(const ST* p);
template <>
__device__ __forceinline__ float ld_st<uint16_t>(const uint16_t* p) { return h2f_bits(*p); }
template <>
__device__ __forceinline__ float ld_st<float>(const float* p) { return *p; }

template <typename ST>
__device__ __forceinline__ void st_st(ST* p, float v);
template <>
__device__ __forceinline__ void st_st<uint16_t>(uint16_t* p, float v) { *p = f2h_bits(v); }
template <>
__device__ __forceinline__ void st_st<float>(float* p, float v) { *p = v; }

// ------------------------------------------------------------------ DPP reductions (no LDS crossbar)
// sum over the 16 lanes of a DPP row (lanes 16i .. 16i+15); every lane of the row gets the total.
// quad_perm xor 1 (0xB1), xor 2 (0x4E), row_half_mirror (0x141), row_mirror (0x140).
#define GEAR_DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true)))
__device__ __forceinline__ float row16_sum(float v) {
    v = GEAR_DPP_ADD(v, 0xB1);
    v = GEAR_DPP_ADD(v, 0x4E);
    v = GEAR_DPP_ADD(v, 0x141);
    v = GEAR_DPP_ADD(v, 0x140);
    return v;
}

// sum over all 64 lanes, returned to every lane through an SGPR: 4 DPP steps inside the rows of 16, row_bcast15 / row_bcast31
// across the rows (lane 63 holds the total), one v_readlane.  6 VALU instead of 6 x (4 VALU + ds_bpermute) for a shuffle
// butterfly (__shfl_xor computes a bounds-checked lane index for every step).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = row16_sum(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// inclusive prefix sum over the 64 lanes of a packed counter (fields must not overflow into each other)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
#define GEAR_DPP_SHR(x, ctrl, rm) ((x) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rm), 0xF, true))
    v = GEAR_DPP_SHR(v, 0x111, 0xF);   // row_shr:1
    v = GEAR_DPP_SHR(v, 0x112, 0xF);   // row_shr:2
    v = GEAR_DPP_SHR(v, 0x114, 0xF);   // row_shr:4
    v = GEAR_DPP_SHR(v, 0x118, 0xF);   // row_shr:8
#undef GEAR_DPP_SHR
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast31 -> rows 2, 3
    return v;
}

// ------------------------------------------------------------------ the group quantizer arithmetic
// MODE 0: fp16-stepwise (cuda_supported_gear/quant/new_pack.py:237-240, :277-278)
// MODE 1: fp32         (GenerationBench/.../Simulated/compress_function.py:24-28)
template <int MODE>
struct QuantParams {
    float mn, scale;  // as stored (exact fp16 values in MODE 0)
    int levels;
};

template <int MODE>
__device__ __forceinline__ QuantParams<MODE> make_qparams(float mn, float mx, int levels) {
    QuantParams<MODE> p;
    p.mn = mn;
    p.levels = levels;
    if (MODE == 0) {
        float range = hround(mx - mn);
        p.scale = hround(div_rn(range, (float)levels));
    } else {
        p.scale = div_rn(mx - mn, (float)levels);
    }
    return p;
}

template <int MODE>
__device__ __forceinline__ int quant_one(float v, const QuantParams<MODE>& p) {
    if (p.scale == 0.0f) return 0;  // zero-range group: defined as code 0 (reference: NaN, defect B6)
    float c;
    if (MODE == 0) {
        float t1 = hround(v - p.mn);
        c = hround(div_rn(t1, p.scale));
    } else {
        c = div_rn(v - p.mn, p.scale);
    }
    c = fminf(fmaxf(c, 0.0f), (float)p.levels);
    return (int)rintf(c);  // half-to-even
}

// dequantized value as the reference computes it (MODE 0: two fp16 roundings; MODE 1: fp32 mul then add, unfused)
template <int MODE>
__device__ __forceinline__ float dequant_one(int q, float scale, float mn) {
    if (MODE == 0) {
        return hround(hround((float)q * scale) + mn);
    } else {
        return __fadd_rn(__fmul_rn((float)q, scale), mn);
    }
}
