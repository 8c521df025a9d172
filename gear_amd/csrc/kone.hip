// kone.hip -- the K side of GEAR compress with ONE read of K: selection + fill + quantize + pack + error + Gram in a single launch.
//
// Reference semantics: gears_channelQ (GenerationBench/GenerationTest/GEARLM/Simulated/compress_function.py:261-296: per channel
// the k smallest / k largest of its T values come out, the row mean goes in, `bit**2 - 1`-level group quantization along T) + the
// Gram matrix of the error for fake_poweriteration_group (:69-98, :213-220); the fused reference path: new_pack.py:253-311.
//
// kfused.hip's chain reads K three times: k_select_kernel (per-channel outliers over all T), k_main_kernel (dense part + Gram),
// k_qpass_kernel (Q = E W with E rebuilt).  Here the first two are one launch -- VERDICT r5 item 1:
//
//   * a head's T tokens are cut into slabs of 256 tokens; one workgroup (512 threads) per slab keeps its [256][128] fp16 slab in
//     LDS (64 KB, rows rotated by 64 bytes per token so that column reads, 16-byte row writes and the matrix cores' transposing
//     reads are all bank-conflict free without padding) from its only HBM read to its last matrix-core read;
//   * the per-channel selection over ALL T tokens is an exchange among the S slab workgroups of a head, which the block map puts on
//     ONE XCD (block b runs on XCD b % 8), through that XCD's L2:
//       E1  every slab publishes the candidates of its tokens (elements beyond a threshold guess made from 256 sampled tokens of
//           the head -- every slab computes the same guess), bucketed by OWNER (slab o owns channels [128 o / S, 128 (o + 1) / S))
//           and its channel sums;
//           the owner finds, per (channel, side), the k-th largest composite key (value, then lower token first) among the
//           candidates of all slabs by bisection with the counts on the scalar unit, checks the guess (count in [k, cap]), sums
//           the row means and counts the selected entries per slab;
//       E2  the owner publishes thresholds, means and per-slab list offsets; every slab marks its own outliers (LDS bitmaps),
//           writes ITS part of the sorted sparse lists (a slab's tokens are a contiguous range, so list position = offset of
//           the slab + rank inside it, by popcount) and SUBSTITUTES the outliers in its LDS slab by fp16(mean);
//   * the dense part then runs without a single outlier mask: lane = (channel pair, half of a 64-token group), 32 tokens in 32
//     registers, packed min / max, quantize, Horner pack, error back into the slab in place; the codes and error entries under
//     the (few) outlier slots are patched sparsely.  (A substitute can only disturb a group's min / max when fp16(mean) is itself
//     the extreme of the group -- every kept element on one side of the row mean; such groups are detected and redone with exact
//     masks.)  tile_fast of kfused.hip spends a quarter of its instructions on those masks.
//   * G = E^T E of the slab on the matrix cores straight from the LDS slab, added into ONE 64 KB matrix per head with fp32
//     atomics (it stays in the XCD's L2; kfused.hip's slab partials went through HBM).
//
// Hand-off protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"; the same as block_fused.hip): payload by agent-scope
// relaxed stores (write-through), `s_waitcnt vmcnt(0)`, workgroup barrier, flag by an agent-scope relaxed store; the consumer polls
// the flags with agent-scope relaxed loads + s_sleep (one wave), barrier, agent-scope loads of the payload; no fences.  Every poll
// is bounded: a time-out raises a status word (gear_kone_timeouts()) instead of hanging the GPU.
// Forward progress: a head's S workgroups are at most 8 S <= 256 consecutive block ids apart and the device holds >= 256 of these
// workgroups (checked at launch); the hardware dispatches in block order.
//
// A head whose threshold guess fails (a candidate count outside [k, cap], a bucket overflow) raises headfail[bh]; its slabs stop
// after E2 and the caller runs kfused.hip's exact chain for the flagged heads (launches that return at once for every other head).
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "ktile.h"

namespace {

constexpr int KO_NT = 4;                 // 64-token tiles per slab
constexpr int KO_ROWS = 64 * KO_NT;      // tokens per slab
constexpr int KO_THREADS = 512;
constexpr int KO_MAXS = 64;              // slabs per head (T <= 16384)
constexpr int KO_KPL = 8;                // candidate keys per lane at the owner (32 lanes per list): list capacity <= 256

typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ void st_agent(void* p, uint32_t v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_agent(const void* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ void st_agent2(void* p, uint32_t lo, uint32_t hi) {
    __hip_atomic_store((gu64*)p, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 ld_agent2(const void* p) {
    const unsigned long long v = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

__device__ uint32_t g_kone_timeouts = 0u;
__device__ uint32_t g_kone_fallbacks = 0u;      // heads handed to the exact chain (cumulative)

// Phase clocks (measurement builds only: make -C gear_amd/csrc EXTRA=-DGEAR_KO_CLK): thread 0 of every workgroup stores s_memtime at
// the marked places, gear_debug_ko_clk copies the table out (tools/exp_kone_clk.py).
#ifdef GEAR_KO_CLK
__device__ unsigned long long ko_clk_buf[16 * 16384];
#define KO_CLK(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) ko_clk_buf[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#define KD_CLK_DECL unsigned long long kd_t_ = __builtin_readcyclecounter(), kd_acc_[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long kd_t0_ = kd_t_
#define KD_CLK(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); kd_acc_[k] += n_ - kd_t_; kd_t_ = n_; } while (0)
#define KD_CLK_OUT do { const unsigned id_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && id_ < 16384) { ko_clk_buf[id_ * 16] = kd_t0_; for (int q_ = 0; q_ < 6; q_++) ko_clk_buf[id_ * 16 + 1 + q_] = kd_acc_[q_]; ko_clk_buf[id_ * 16 + 7] = __builtin_readcyclecounter(); } } while (0)
#else
#define KO_CLK(k) do { } while (0)
#define KD_CLK_DECL do { } while (0)
#define KD_CLK(k) do { } while (0)
#define KD_CLK_OUT do { } while (0)
#endif

struct KoArgs {
    const uint16_t* x;       // [BH][T][128]
    int64_t BH;
    int T, S, k;
    float zthr, rlen;
    int nlmax;               // lists per owner (2 * ceil(128 / S))
    int bcap;                // entries per (slab -> owner) bucket (multiple of 2)
    int lcap;                // candidate keys per list at the owner (multiple of 32, <= 32 * KO_KPL)
    int xbytes;              // LDS bytes of the time-shared area (buckets -> owner lists -> side bitmaps)
    // exchange area (global), 8-byte granules {data, epoch}
    uint32_t epoch;          // tag of this call (never 0)
    unsigned long long* xc_sum;   // [BH][S][2][128]: channel sums, sums of squares of the slab (float bits)
    unsigned long long* xc_cnt;   // [BH][S owners][S slabs]: entries of the slab for the owner
    unsigned long long* xc_ent;   // [BH][S owners][S slabs][bcap]
    unsigned long long* kthr;     // [BH][256]: composite threshold of list 2 * channel + side (0: the guess failed)
    unsigned long long* base;     // [BH][S][128]: (offset of the slab in list 2 c) | (offset in list 2 c + 1) << 16
    uint32_t* headfail;      // [BH] (zeroed by the caller): heads the exact chain must redo
    // outputs
    uint32_t* obits;         // [BH][T/64][128][2]
    uint16_t* oidx;          // [BH][128][2][kcap]
    uint16_t* oval;
    int kcap, o_off, tok_base;
    uint32_t* code;          // [BH][128][ldc]
    void* scale;             // [BH][128][lds] float
    void* mn;
    int64_t ldc, lds;
    int t_off;
    float* G;                // [BH][128][128] or null (no low-rank step)
    int dbg;                 // measurement builds: 1 = no atomics, 2 = no Gram at all, 4 = no clearing of G, 8 = stop after E2
};

// ---------------------------------------------------------------------------------------------------- LDS slab geometry
// byte offset of (row, channel c) inside the slab: 16-byte chunk (c >> 3) rotated by 4 chunks per row & 3
__device__ __forceinline__ uint32_t ko_off(int row, int c) {
    return (uint32_t)row * 256u + (uint32_t)((((c >> 3) + 4 * (row & 3)) & 15) << 4) + (uint32_t)((c & 7) << 1);
}

// MFMA operand (as ktile.h's load_operand<true>) from the rotated slab: lane (x31, kg) gets channel 32 I + x31, tokens t0 + 8 kg .. + 7
__device__ __forceinline__ half8_t ko_operand(const unsigned char* slab, int t0, int I, int lane) {
    const int kg = lane >> 5, i = lane & 15, c0 = 32 * I + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
    const int row = t0 + 8 * kg + (i >> 2);
    const uint32_t addr = (uint32_t)(uintptr_t)(slab + ko_off(row, c0));        // (row + 4 has the same rotation: + 1024 bytes)
    typedef short short4v __attribute__((ext_vector_type(4)));
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(addr) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    union { half8_t h; short4v s[2]; } cv;
    cv.s[0] = lo;
    cv.s[1] = hi;
    return cv.h;
}

__device__ __forceinline__ int ko_owner(int ch, int S) { return (ch * S) >> 7; }
__device__ __forceinline__ int ko_c0(int o, int S) { return (128 * o + S - 1) / S; }     // first channel of owner o

// Exchange granule: one naturally aligned 8-byte {data, tag} written by ONE agent-scope store; the tag is the call's epoch, so a
// granule of this call is told from whatever the memory held before (no zeroing, no separate flag, no drain before a flag).
// Wait until the n <= 64 granules p[lane * stride] carry the epoch.  Called by ONE wave; bounded.  Returns the data word.
__device__ __forceinline__ uint32_t ko_poll(const unsigned long long* p, int64_t stride, int n, int lane, uint32_t epoch) {
    unsigned spins = 0;
    uint2 v = make_uint2(0u, epoch);
    while (true) {
        if (lane < n) v = ld_agent2(p + lane * stride);
        if (__all(v.y == epoch)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) {
            if (lane == 0) atomicAdd(&g_kone_timeouts, 1u);
            break;
        }
    }
    return v.x;
}
// one granule that a poll has (almost certainly) seen complete: load, check the tag, retry while it is not there yet
__device__ __forceinline__ uint32_t ko_get(const unsigned long long* p, uint32_t epoch) {
    uint2 v = ld_agent2(p);
    unsigned spins = 0;
    while (v.y != epoch) {
        __builtin_amdgcn_s_sleep(1);
        v = ld_agent2(p);
        if (++spins > (1u << 22)) { atomicAdd(&g_kone_timeouts, 1u); break; }
    }
    return v.x;
}

template <int HI>
__device__ __forceinline__ float ko_mix(uint32_t w, float addend) {      // float(half HI ? high : low of w) + addend
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(addend));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(addend));
    return r;
}

// ---------------------------------------------------------------------------------------------------- the dense part
// Lane (cp = channel pair, hf = half of the 64-token tile): tokens row0 .. row0 + 31 of channels 2 cp, 2 cp + 1, the outliers in the
// slab already replaced by fp16(mean).  fp32 "simulated" arithmetic (compress_function.py:24-33 through :116-125), bit-exact with
// kfused.hip's tile_fast: exact min / max, scale = (mx - mn) / levels by IEEE division, quotient by reciprocal multiply with a
// 1e-5 tie guard and exact re-division, dequant = code * scale + mn unfused, error = x - fp16(dequant).
template <int BITS, int G>
__device__ __forceinline__ void ko_dense(unsigned char* slab, int row0, int cp, int hf, uint32_t mA, uint32_t mB, bool gA, bool gB,
                                         float meanA, float meanB, uint32_t sA, uint32_t sB, uint32_t (&cwA)[32 * BITS / 32],
                                         uint32_t (&cwB)[32 * BITS / 32], float& qsA_o, float& loA_o, float& qsB_o, float& loB_o) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int HC = 16 / BITS;
    constexpr float TIE = 0.49999f;
    const uint32_t PINF = 0x7C007C00u, NINF = 0xFC00FC00u;
    // the lane's word (row0 + i, cp): four bases, one per row & 3 (row0 is a multiple of 32)
    uint32_t base[4];
#pragma unroll
    for (int r = 0; r < 4; r++) base[r] = (uint32_t)row0 * 256u + (uint32_t)((((cp >> 2) + 4 * r) & 15) << 4) + (uint32_t)((cp & 3) << 2);
    // (two passes over the lane's 32 words in LDS -- min / max, then quantize half word by half word -- instead of 32 registers held
    // across both: k_dense_kernel keeps 32 Gram accumulators alive through this function at four waves per SIMD)
    uint32_t lo2 = PINF, hi2 = NINF;
#pragma unroll
    for (int i = 0; i < 32; i += 2) {       // (three-operand packed min / max of gfx950: two new words per instruction)
        const uint32_t w0 = *(const uint32_t*)(slab + base[i & 3] + 256 * i);
        const uint32_t w1 = *(const uint32_t*)(slab + base[(i + 1) & 3] + 256 * (i + 1));
        asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(lo2) : "v"(lo2), "v"(w0), "v"(w1));
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(hi2) : "v"(hi2), "v"(w0), "v"(w1));
    }
    if (G == 64) {
        lo2 = pkmin16(lo2, (uint32_t)__shfl_xor((int)lo2, 32, 64));
        hi2 = pkmax16(hi2, (uint32_t)__shfl_xor((int)hi2, 32, 64));
    }
    float loA = h2f_bits((uint16_t)(lo2 & 0xFFFFu)), loB = h2f_bits((uint16_t)(lo2 >> 16));
    float hiA = h2f_bits((uint16_t)(hi2 & 0xFFFFu)), hiB = h2f_bits((uint16_t)(hi2 >> 16));
    // a substitute that is itself the group's extreme: redo min / max over the kept elements only (exact masks)
    {
        const float fsA = h2f_bits((uint16_t)sA), fsB = h2f_bits((uint16_t)sB);
        const bool rare = (gA && (loA == fsA || hiA == fsA)) || (gB && (loB == fsB || hiB == fsB));
        if (__any(rare)) {
            float la = INFINITY, ha = -INFINITY, lb = INFINITY, hb = -INFINITY;
#pragma unroll 1
            for (int i = 0; i < 32; i++) {
                // (w[] is indexed by the loop counter only through the LDS: re-read the word)
                const uint32_t wi = *(const uint32_t*)(slab + (uint32_t)(row0 + i) * 256u + (uint32_t)((((cp >> 2) + 4 * (i & 3)) & 15) << 4) + (uint32_t)((cp & 3) << 2));
                const float xa = h2f_bits((uint16_t)(wi & 0xFFFFu)), xb = h2f_bits((uint16_t)(wi >> 16));
                if (!((mA >> i) & 1u)) { la = fmin_raw(la, xa); ha = fmax_raw(ha, xa); }
                if (!((mB >> i) & 1u)) { lb = fmin_raw(lb, xb); hb = fmax_raw(hb, xb); }
            }
            if (G == 64) {
                la = fmin_raw(la, __shfl_xor(la, 32, 64)); ha = fmax_raw(ha, __shfl_xor(ha, 32, 64));
                lb = fmin_raw(lb, __shfl_xor(lb, 32, 64)); hb = fmax_raw(hb, __shfl_xor(hb, 32, 64));
            }
            loA = la; hiA = ha; loB = lb; hiB = hb;
        }
    }
    // the fill value (fp32 row mean, compress_function.py:279-283) takes part in min / max when the group holds an outlier
    loA = fmin_raw(loA, gA ? meanA : INFINITY); hiA = fmax_raw(hiA, gA ? meanA : -INFINITY);
    loB = fmin_raw(loB, gB ? meanB : INFINITY); hiB = fmax_raw(hiB, gB ? meanB : -INFINITY);
    const float qsA = div_rn(hiA - loA, (float)LEVELS), qsB = div_rn(hiB - loB, (float)LEVELS);
    const float invA = (qsA != 0.0f) ? __builtin_amdgcn_rcpf(qsA) : 0.0f, invB = (qsB != 0.0f) ? __builtin_amdgcn_rcpf(qsB) : 0.0f;
    qsA_o = qsA; loA_o = loA; qsB_o = qsB; loB_o = loB;
    const float2v inv2 = {invA, invB}, qs2 = {qsA, qsB}, mn2 = {loA, loB};
    const float nloA = -loA, nloB = -loB;
    const float2v magic2 = {12582912.0f, 12582912.0f}, radix2 = {(float)(1 << BITS), (float)(1 << BITS)};
#pragma unroll
    for (int hb = 0; hb < 32 / HC; hb++) {
        const int tb = hb * HC;
        float2v rq[HC];
        float dmax = 0.0f;
        uint32_t w[HC];
#pragma unroll
        for (int j = 0; j < HC; j++) w[j] = *(const uint32_t*)(slab + base[(tb + j) & 3] + 256 * (tb + j));
        // quotient by reciprocal multiply, rounded to an integer by the 1.5 * 2^23 addend inside ONE fused multiply-add (rint of the
        // exact product; 0 <= quotient <= levels), distance to that integer by a second one: three packed fp32 instructions where
        // multiply + 2 x v_rndne_f32 + subtract were four of the same issue class (profiles/r6_valu_rate.md)
#pragma unroll
        for (int j = 0; j < HC; j++) {
            const float2v t = {ko_mix<0>(w[j], nloA), ko_mix<1>(w[j], nloB)};
            const float2v sb = __builtin_elementwise_fma(t, inv2, magic2);
            const float2v rr = sb - magic2;
            const float2v d = __builtin_elementwise_fma(t, inv2, -rr);
            rq[j] = rr;
            asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(dmax) : "v"(dmax), "v"(d.x), "v"(d.y));
        }
        if (dmax > TIE) {                                // within 1e-5 of a rounding tie: redo by exact division
#pragma unroll
            for (int j = 0; j < HC; j++) {
                const float xa = h2f_bits((uint16_t)(w[j] & 0xFFFFu)), xb = h2f_bits((uint16_t)(w[j] >> 16));
                rq[j].x = (qsA != 0.0f) ? rintf(div_rn(xa - loA, qsA)) : 0.0f;
                rq[j].y = (qsB != 0.0f) ? rintf(div_rn(xb - loB, qsB)) : 0.0f;
            }
        }
        // every element lies in [mn, mx] (the substitutes too): 0 <= quotient <= levels, no clamp
        float2v hn = {0.0f, 0.0f};
#pragma unroll
        for (int j = HC - 1; j >= 0; j--) hn = __builtin_elementwise_fma(hn, radix2, rq[j]);      // exact: < 2^16
        const uint32_t hwA = (uint32_t)hn.x, hwB = (uint32_t)hn.y;
        if ((hb & 1) == 0) { cwA[hb >> 1] = hwA; cwB[hb >> 1] = hwB; }
        else { cwA[hb >> 1] |= hwA << 16; cwB[hb >> 1] |= hwB << 16; }
#pragma unroll
        for (int j = 0; j < HC; j++) {
            const float2v dq = rq[j] * qs2 + mn2;        // -ffp-contract=off: v_pk_mul_f32, v_pk_add_f32
            const uint32_t dw = f2h2_bits(dq.x, dq.y);
            uint32_t e2;
            asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e2) : "v"(w[j]), "v"(dw));
            *(uint32_t*)(slab + base[(tb + j) & 3] + 256 * (tb + j)) = e2;
        }
        // (no instruction moves across: without it the scheduler hoists the later blocks' LDS reads and conversions over this one and
        // the function's live range grows by 60 registers -- which a wave alone cannot use anyway: profiles/r6_valu_rate.md)
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the outlier slots: code = quant(mean) (fill, then quantize: compress_function.py:276-286), error = 0
    asm volatile("" ::: "memory");      // (the 2-byte stores below hit words stored above through another type: keep the order)
    if (mA | mB) {
        constexpr int CPW = 32 / BITS;
        float cmA = 0.f, cmB = 0.f;
        {
            const float ca = (meanA - loA) * invA, cb = (meanB - loB) * invB;
            cmA = rintf(ca); cmB = rintf(cb);
            if (fabsf(ca - cmA) > TIE) cmA = (qsA != 0.0f) ? rintf(div_rn(meanA - loA, qsA)) : 0.0f;
            if (fabsf(cb - cmB) > TIE) cmB = (qsB != 0.0f) ? rintf(div_rn(meanB - loB, qsB)) : 0.0f;
            cmA = __builtin_amdgcn_fmed3f(cmA, 0.0f, (float)LEVELS);
            cmB = __builtin_amdgcn_fmed3f(cmB, 0.0f, (float)LEVELS);
        }
        const uint32_t rA = (uint32_t)cmA, rB = (uint32_t)cmB;
        uint32_t m = mA;
        while (m) {
            const int j = __ffs((int)m) - 1;
            m &= m - 1u;
            const int wi = j / CPW, sh = BITS * (j % CPW);
#pragma unroll
            for (int q = 0; q < 32 * BITS / 32; q++)
                if (wi == q) cwA[q] = (cwA[q] & ~((uint32_t)LEVELS << sh)) | (rA << sh);
            *(uint16_t*)(slab + (uint32_t)(row0 + j) * 256u + (uint32_t)((((cp >> 2) + 4 * (j & 3)) & 15) << 4) + (uint32_t)((cp & 3) << 2)) = 0;
        }
        m = mB;
        while (m) {
            const int j = __ffs((int)m) - 1;
            m &= m - 1u;
            const int wi = j / CPW, sh = BITS * (j % CPW);
#pragma unroll
            for (int q = 0; q < 32 * BITS / 32; q++)
                if (wi == q) cwB[q] = (cwB[q] & ~((uint32_t)LEVELS << sh)) | (rB << sh);
            *(uint16_t*)(slab + (uint32_t)(row0 + j) * 256u + (uint32_t)((((cp >> 2) + 4 * (j & 3)) & 15) << 4) + (uint32_t)((cp & 3) << 2) + 2u) = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- the kernel
// block L of the 1-D grid -> (slab, head): with BH % 8 == 0 and 8 S <= 256 a head's slabs are blocks of ONE residue mod 8 (one XCD)
__device__ __forceinline__ void ko_block_map(int L, int64_t BH, int S, int& slab, int64_t& bh) {
    if ((BH & 7) == 0 && 8 * S <= 256) {
        const int xcd = L & 7, i = L >> 3;
        slab = i % S;
        bh = xcd + 8 * (int64_t)(i / S);
    } else {
        slab = L % S;
        bh = L / S;
    }
}

template <int BITS, int G>
__global__ __launch_bounds__(KO_THREADS, 4) void k_one_kernel(KoArgs a) {
    constexpr int CPW = 32 / BITS;
    constexpr int NWL = 32 * BITS / 32;               // code words per lane and channel
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* slabp = sm;                                        // [256][256 B], rotated rows
    uint32_t* X = (uint32_t*)(sm + KO_ROWS * 256);                    // time-shared: buckets | owner keys | side bitmaps
    uint32_t* fx = (uint32_t*)(sm + KO_ROWS * 256 + a.xbytes);        // fixed part (KO_FIXED words + lists of the owner)
    uint32_t* bcnt = fx;                 // [64]   entries per owner bucket
    uint32_t* cntS = fx + 64;            // [64]   entries of each slab for this owner
    uint32_t* misc = fx + 128;           // [16]
    uint32_t* thr16 = fx + 144;          // [128]  thi | tlo << 16 per channel (P1 - P2) ...
    uint32_t* base_l = fx + 144;         //        ... list offsets of this slab, two 16-bit halves per channel (P5 -)
    float* mean_l = (float*)(fx + 272);  // [128]
    uint32_t* kthr_l = fx + 400;         // [256]  thresholds (P4 -)
    uint32_t* chinfo = fx + 656;         // [128]  owner | first list of the channel at its owner << 8
    uint32_t* c0tab = fx + 784;          // [64]   first channel of an owner
    uint32_t* lcnt = fx + 848;           // [nlmax]  candidates per list at the owner
    uint32_t* selcnt = lcnt + a.nlmax;   // [nlmax * S] selected entries per (list, slab) at the owner

    int slab;
    int64_t bh;
    ko_block_map((int)blockIdx.x, a.BH, a.S, slab, bh);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, S = a.S, k = a.k;
    const int ntiles = T >> 6;
    const int ntl = min(KO_NT, ntiles - slab * KO_NT);                 // tiles of this slab
    const int nrows = ntl * 64;
    const uint16_t* xh = a.x + bh * (int64_t)T * KD;
    const int r16 = tid >> 4, q16 = tid & 15;                          // row group / 16-byte chunk of the loads

    // ---------------------------------------------------------------- P0: the slab -> registers -> LDS (its only trip from HBM)
    KO_CLK(0);
    uint4 xv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = min(r16 + 32 * i, nrows - 1);
        xv[i] = *(const uint4*)(xh + ((int64_t)slab * KO_ROWS + row) * KD + 8 * q16);
    }
    if (tid < 64) { bcnt[tid] = 0u; c0tab[tid] = (uint32_t)ko_c0(tid, S); }
    if (tid < 16) misc[tid] = 0u;
    if (tid < KD) { const int o = ko_owner(tid, S); chinfo[tid] = (uint32_t)o | ((uint32_t)((tid - ko_c0(o, S)) * 2) << 8); }
    for (int e = tid; e < a.nlmax * (S + 1); e += KO_THREADS) lcnt[e] = 0u;          // lcnt and selcnt
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = r16 + 32 * i;
        *(uint4*)(slabp + (uint32_t)row * 256u + (uint32_t)(((q16 + 4 * (row & 3)) & 15) << 4)) = xv[i];
    }
    __syncthreads();

    KO_CLK(1);
    uint32_t kept[6];                    // this thread's share of the slab's candidate entries
    uint32_t keptm = 0u;                 // which of them exist
    uint32_t kepto[2] = {0u, 0u};        // their owners, 8 bits each
    if (k > 0) {
        // ------------------------------------------------------------ P1: channel sums and sums of squares on the matrix cores, E0
        {
            // D1 = ones[32 x 16] * X[16 tokens x 32 channels]: every row holds the column sums; D2 = X^T X block (I, I): its diagonal
            // holds the sums of squares.  wave -> (channel block I, half of the 16-token steps)
            const int I = wave & 3, kh = wave >> 2;
            float16_t acc1, acc2;
#pragma unroll
            for (int q = 0; q < 16; q++) { acc1[q] = 0.0f; acc2[q] = 0.0f; }
            half8_t ones;
#pragma unroll
            for (int q = 0; q < 8; q++) ones[q] = (_Float16)1.0f;
            const int ks_n = ntl * 4;                                   // 16-token steps of the slab
            for (int ks = kh; ks < ks_n; ks += 2) {
                const half8_t b = ko_operand(slabp, 16 * ks, I, lane);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, acc2, 0, 0, 0);
            }
            float* st = (float*)X;                                      // [2 halves][2: sum, sum of squares][128]
            const int col = lane & 31, kg = lane >> 5;
            if (lane < 32) st[(kh * 2) * KD + 32 * I + lane] = acc1[0];          // row 0 of D1, column lane
            // C layout: lane l, reg q -> row (q & 3) + 8 (q >> 2) + 4 (l >> 5), col l & 31: the diagonal element of column col sits in
            // the lane whose kg == (col >> 2) & 1, register (col & 3) + 4 (col >> 3)
            if (((col >> 2) & 1) == kg) {
                const int qd = (col & 3) + 4 * (col >> 3);
                float dv = 0.0f;
#pragma unroll
                for (int q = 0; q < 16; q++) dv = (q == qd) ? acc2[q] : dv;
                st[(kh * 2 + 1) * KD + 32 * I + col] = dv;
            }
        }
        __syncthreads();
        {
            const float* st = (const float*)X;
            if (tid < 2 * KD) {
                const float v = st[tid] + st[2 * KD + tid];
                st_agent2(a.xc_sum + (bh * S + slab) * (int64_t)(2 * KD) + tid, __builtin_bit_cast(uint32_t, v), a.epoch);
            }
        }
        KO_CLK(2);
        if (wave == 0) (void)ko_poll(a.xc_sum + bh * S * (int64_t)(2 * KD), 2 * KD, S, lane, a.epoch);
        __syncthreads();
        KO_CLK(3);
        {   // the head's statistics: every slab adds the S slab sums in the same order -> the same mean and thresholds everywhere
            float* tot = (float*)X;
            if (tid < 2 * KD) {
                float t = 0.0f;
                for (int s0 = 0; s0 < S; s0 += 16) {
                    uint2 v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) v[u] = ld_agent2(a.xc_sum + (bh * S + min(s0 + u, S - 1)) * (int64_t)(2 * KD) + tid);
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        if (s0 + u < S) {
                            const uint32_t w = v[u].y == a.epoch ? v[u].x : ko_get(a.xc_sum + (bh * S + s0 + u) * (int64_t)(2 * KD) + tid, a.epoch);
                            t += __builtin_bit_cast(float, w);
                        }
                    }
                }
                tot[tid] = t;
            }
            __syncthreads();
            if (tid < KD) {
                const float mean = ((T & (T - 1)) == 0) ? tot[tid] * a.rlen : tot[tid] / (float)T;
                const float sd = sqrtf(fmaxf(tot[KD + tid] * a.rlen - mean * mean, 0.0f));
                uint32_t th = f2h_bits(mean + a.zthr * sd), tl = f2h_bits(mean - a.zthr * sd);
                // never +-0: then clamp(x) != x happens only for x strictly outside [tl, th]
                if ((th & 0x7FFFu) == 0u) th = 0x0001u;
                if ((tl & 0x7FFFu) == 0u) tl = 0x8001u;
                thr16[tid] = th | (tl << 16);
                mean_l[tid] = mean;
            }
            __syncthreads();
        }
        KO_CLK(4);
        // ------------------------------------------------------------ P2: candidates (from the registers the slab came through) -> owner buckets
        {
            uint32_t thi[4], tlo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t t0 = thr16[8 * q16 + 2 * j], t1 = thr16[8 * q16 + 2 * j + 1];
                thi[j] = (t0 & 0xFFFFu) | (t1 << 16);
                tlo[j] = (t0 >> 16) | (t1 & 0xFFFF0000u);
            }
            const int bcap = a.bcap;
            // a bit per element beyond its channel's thresholds: clamp(x) != x, four packed instructions per word, then the lanes
            // walk their set bits (a wave's loop runs as long as its busiest lane: ~7 of 64 elements)
            unsigned long long m64 = 0ull;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t wv[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
                uint32_t m = 0u;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    m = (m << 1) | pkminu16(pkmin16(pkmax16(wv[j], tlo[j]), thi[j]) ^ wv[j], 0x00010001u);     // bit 3 - j: low half, bit 19 - j: high half
                if (r16 + 32 * i >= nrows) m = 0u;
                m64 |= (unsigned long long)((m & 0xFu) | ((m >> 12) & 0xF0u)) << (8 * i);
            }
            while (m64) {
                const int b = __ffsll((long long)m64) - 1;
                m64 &= m64 - 1ull;
                const int i = b >> 3, r = b & 7, h = r >> 2, j = 3 - (r & 3);
                const int row = r16 + 32 * i, ch = 8 * q16 + 2 * j + h;
                const uint32_t bits = *(const uint16_t*)(slabp + ko_off(row, ch));
                const int side = h2f_bits((uint16_t)bits) < h2f_bits((uint16_t)(thr16[ch] >> 16)) ? 1 : 0;
                const uint32_t ci = chinfo[ch];
                const uint32_t o = ci & 0xFFu, list = (ci >> 8) + (uint32_t)side;
                const uint32_t slot = atomicAdd(&bcnt[o], 1u);
                if (slot < (uint32_t)bcap) X[o * bcap + slot] = (bits << 16) | ((uint32_t)row << 8) | list;
            }
        }
        __syncthreads();
        KO_CLK(5);
        // ------------------------------------------------------------ P3: publish E1 (entries and counts as tagged granules: no drain, no flag)
        {
            const int tot = S * a.bcap;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int e = tid + KO_THREADS * i;
                kept[i] = 0u;
                if (e < tot) {
                    const int o = e / a.bcap, j = e - o * a.bcap;
                    if (j < (int)min(bcnt[o], (uint32_t)a.bcap)) {
                        kept[i] = X[e];
                        keptm |= 1u << i;
                        kepto[i >> 2] |= (uint32_t)o << (8 * (i & 3));
                        st_agent2(a.xc_ent + ((bh * S + o) * S + slab) * (int64_t)a.bcap + j, kept[i], a.epoch);
                    }
                }
            }
            if (tid < S) st_agent2(a.xc_cnt + (bh * S + tid) * (int64_t)S + slab, bcnt[tid], a.epoch);
        }
        // ------------------------------------------------------------ P4: the owner's lists
        KO_CLK(6);
        const int me = slab;
        const int c0 = ko_c0(me, S), nch = ko_c0(me + 1, S) - c0, nl = 2 * nch;
        const int lcap = a.lcap;
        if (wave == 0) {
            const uint32_t cn = ko_poll(a.xc_cnt + (bh * S + me) * (int64_t)S, 1, S, lane, a.epoch);
            if (lane < S) cntS[lane] = cn;
        }
        __syncthreads();                                           // (also: every thread has read its bucket entries out of X)
        {
            KO_CLK(7);
            // thread j takes entry j of every slab's segment for this owner
            bool bad = false;
            const int j = min(tid, a.bcap - 1);
            for (int s0 = 0; s0 < S; s0 += 16) {
                uint2 v[16];
#pragma unroll
                for (int u = 0; u < 16; u++)
                    v[u] = ld_agent2(a.xc_ent + ((bh * S + me) * S + min(s0 + u, S - 1)) * (int64_t)a.bcap + j);
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int s = s0 + u;
                    if (s < S) {
                        const uint32_t cn = cntS[s];
                        if (cn > (uint32_t)a.bcap) bad = true;                  // a bucket overflowed: the head takes the exact chain
                        if (tid < (int)min(cn, (uint32_t)a.bcap)) {
                            const uint32_t e = v[u].y == a.epoch ? v[u].x : ko_get(a.xc_ent + ((bh * S + me) * S + s) * (int64_t)a.bcap + j, a.epoch);
                            const int list = (int)(e & 0xFFu), side = list & 1;
                            const uint32_t tok = (uint32_t)(s * KO_ROWS) + ((e >> 8) & 0xFFu);
                            const uint32_t key = ((order_key(e >> 16, side) << 14) | (0x3FFFu - tok)) + 1u;
                            const uint32_t slot = atomicAdd(&lcnt[list], 1u);
                            if (slot < (uint32_t)lcap) X[list * lcap + slot] = key;
                        }
                    }
                }
            }
            if (bad) misc[1] = 1u;
        }
        __syncthreads();
        KO_CLK(8);
        {   // the k-th largest key of every list: lanes 0-31 of a wave take list lA, lanes 32-63 list lA + 1
            const int half = lane >> 5, l32 = lane & 31;
            for (int lA = 2 * wave; lA < nl; lA += 16) {
                const int l = lA + half;
                const int n = (int)lcnt[l];
                const bool valid = n >= k && n <= lcap && misc[1] == 0u;
                uint32_t key[KO_KPL];
#pragma unroll
                for (int j = 0; j < KO_KPL; j++) {
                    const int g = l32 + 32 * j;
                    key[j] = (valid && g < n) ? X[l * lcap + g] : 0u;
                }
                // The k-th largest composite key.  First the 16-bit value part V* (largest V with count(value >= V) >= k; the search
                // starts at the threshold guess every candidate lies beyond and ends at the list's maximum), then -- only when more
                // entries share V* than are still wanted -- the token part among them.  Counts through ballots on the scalar unit.
                const int kpl = (lcap + 31) >> 5;
                uint32_t hv[KO_KPL];                                    // value part + 1 (0: no entry)
                uint32_t mx = 0u;
#pragma unroll
                for (int j = 0; j < KO_KPL; j++) { hv[j] = key[j] ? ((key[j] - 1u) >> 14) + 1u : 0u; mx = max(mx, hv[j]); }
#pragma unroll
                for (int d = 16; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
                const uint32_t mx0 = (uint32_t)__builtin_amdgcn_readlane((int)mx, 0), mx1 = (uint32_t)__builtin_amdgcn_readlane((int)mx, 32);
                uint32_t lo0 = 1u, hi0 = max(mx0, 1u), lo1 = 1u, hi1 = max(mx1, 1u);
                while (lo0 < hi0 || lo1 < hi1) {
                    const uint32_t mid0 = lo0 + ((hi0 - lo0 + 1u) >> 1), mid1 = lo1 + ((hi1 - lo1 + 1u) >> 1);
                    const uint32_t mid = half ? mid1 : mid0;
                    int cA = 0, cB = 0;
#pragma unroll
                    for (int j = 0; j < KO_KPL; j++) {
                        if (j < kpl) {
                            const unsigned long long b = __ballot(hv[j] >= mid);
                            cA += __popc((uint32_t)b);
                            cB += __popc((uint32_t)(b >> 32));
                        }
                    }
                    if (lo0 < hi0) { if (cA >= k) lo0 = mid0; else hi0 = mid0 - 1u; }
                    if (lo1 < hi1) { if (cB >= k) lo1 = mid1; else hi1 = mid1 - 1u; }
                }
                // entries above V* and at V*
                int gA = 0, gB = 0, eA = 0, eB = 0;
                {
                    const uint32_t vs = half ? lo1 : lo0;
#pragma unroll
                    for (int j = 0; j < KO_KPL; j++) {
                        if (j < kpl) {
                            const unsigned long long bg = __ballot(hv[j] > vs), be = __ballot(hv[j] == vs);
                            gA += __popc((uint32_t)bg); gB += __popc((uint32_t)(bg >> 32));
                            eA += __popc((uint32_t)be); eB += __popc((uint32_t)(be >> 32));
                        }
                    }
                }
                const int needA = k - gA, needB = k - gB;               // ties at V* still to take (lower token first)
                uint32_t tl0 = 0u, th0 = (eA > needA) ? 0x3FFFu : 0u, tl1 = 0u, th1 = (eB > needB) ? 0x3FFFu : 0u;
                {
                    const uint32_t vs = half ? lo1 : lo0;
                    while (tl0 < th0 || tl1 < th1) {                    // largest D with count(value == V* and token part >= D) >= need
                        const uint32_t mid0 = tl0 + ((th0 - tl0 + 1u) >> 1), mid1 = tl1 + ((th1 - tl1 + 1u) >> 1);
                        const uint32_t mid = half ? mid1 : mid0;
                        int cA = 0, cB = 0;
#pragma unroll
                        for (int j = 0; j < KO_KPL; j++) {
                            if (j < kpl) {
                                const unsigned long long b = __ballot(hv[j] == vs && ((key[j] - 1u) & 0x3FFFu) >= mid);
                                cA += __popc((uint32_t)b);
                                cB += __popc((uint32_t)(b >> 32));
                            }
                        }
                        if (tl0 < th0) { if (cA >= needA) tl0 = mid0; else th0 = mid0 - 1u; }
                        if (tl1 < th1) { if (cB >= needB) tl1 = mid1; else th1 = mid1 - 1u; }
                    }
                }
                // composite threshold: ((V* - 1) << 14 | D) + 1
                lo0 = (((lo0 - 1u) << 14) | tl0) + 1u;
                lo1 = (((lo1 - 1u) << 14) | tl1) + 1u;
                const uint32_t kt = valid ? (half ? lo1 : lo0) : 0u;
                if (l32 == 0) kthr_l[l] = kt;                             // 0: the guess failed for this list -> the head takes the exact chain
                // selected entries per slab (their tokens say which): the slabs' offsets in the sorted list
#pragma unroll
                for (int j = 0; j < KO_KPL; j++) {
                    if (kt != 0u && key[j] >= kt) {
                        const uint32_t tok = 0x3FFFu - ((key[j] - 1u) & 0x3FFFu);
                        atomicAdd(&selcnt[l * S + (int)(tok >> 8)], 1u);
                    }
                }
            }
        }
        __syncthreads();
        KO_CLK(9);
        {   // publish E2: thresholds and per-slab list offsets of this owner's channels
            if (tid < nl) st_agent2(a.kthr + bh * 256 + 2 * c0 + tid, kthr_l[tid], a.epoch);
            for (int e = tid; e < nch * S; e += KO_THREADS) {
                const int c = e / S, s = e - c * S;
                uint32_t b0 = 0u, b1 = 0u;
                for (int s2 = 0; s2 < s; s2++) { b0 += selcnt[(2 * c) * S + s2]; b1 += selcnt[(2 * c + 1) * S + s2]; }
                st_agent2(a.base + (bh * S + s) * (int64_t)KD + c0 + c, b0 | (b1 << 16), a.epoch);
            }
        }
        // ------------------------------------------------------------ P5: everybody's thresholds
        KO_CLK(10);
        if (wave == 0) {                                           // the first threshold of every owner ...
            const int oc = 2 * ko_c0(min(lane, S - 1), S);
            unsigned spins = 0;
            while (true) {
                const uint2 v = ld_agent2(a.kthr + bh * 256 + oc);
                if (__all(v.y == a.epoch)) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(&g_kone_timeouts, 1u); break; }
            }
        }
        __syncthreads();                                           // ... (and the owner is done with its keys in X)
        KO_CLK(11);
        if (tid < 256) {
            const uint32_t kt = ko_get(a.kthr + bh * 256 + tid, a.epoch);
            kthr_l[tid] = kt;
            if (kt == 0u) misc[0] = 1u;
        } else if (tid >= 384) base_l[tid - 384] = ko_get(a.base + (bh * S + slab) * (int64_t)KD + (tid - 384), a.epoch);
        // the side bitmaps of the slab: X[side][tile][channel][2 words]
        for (int e = tid; e < 2 * KO_NT * KD * 2; e += KO_THREADS) X[e] = 0u;
        __syncthreads();
        if (misc[0] != 0u && slab == 0 && tid == 0) { a.headfail[bh] = 1u; atomicAdd(&g_kone_fallbacks, 1u); }
        if (misc[0] != 0u || (a.dbg & 8)) return;                        // the exact chain redoes this head
        // ------------------------------------------------------------ P6: mark, substitute, this slab's part of the sparse lists
        bool sel[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            sel[i] = false;
            if ((keptm >> i) & 1u) {
                const int o = (int)((kepto[i >> 2] >> (8 * (i & 3))) & 0xFFu);
                const int list = (int)(kept[i] & 0xFFu), side = list & 1, ch = (int)c0tab[o] + (list >> 1);
                const int row = (int)((kept[i] >> 8) & 0xFFu);
                const uint32_t tok = (uint32_t)(slab * KO_ROWS + row);
                const uint32_t key = ((order_key(kept[i] >> 16, side) << 14) | (0x3FFFu - tok)) + 1u;
                if (key >= kthr_l[2 * ch + side]) {
                    sel[i] = true;
                    atomicOr(&X[((side * KO_NT + (row >> 6)) * KD + ch) * 2 + ((row >> 5) & 1)], 1u << (row & 31));
                    *(uint16_t*)(slabp + ko_off(row, ch)) = f2h_bits(mean_l[ch]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (sel[i]) {
                const int o = (int)((kepto[i >> 2] >> (8 * (i & 3))) & 0xFFu);
                const int list = (int)(kept[i] & 0xFFu), side = list & 1, ch = (int)c0tab[o] + (list >> 1);
                const int row = (int)((kept[i] >> 8) & 0xFFu);
                const uint32_t* bm = &X[((side * KO_NT) * KD + ch) * 2];           // + tile * 256 words
                uint32_t r = (base_l[ch] >> (16 * side)) & 0xFFFFu;
                const int wq = row >> 5;                                            // 32-token word of the slab
                for (int q = 0; q < wq; q++) r += (uint32_t)__popc(bm[(q >> 1) * (KD * 2) + (q & 1)]);
                r += (uint32_t)__popc(bm[(wq >> 1) * (KD * 2) + (wq & 1)] & ((1u << (row & 31)) - 1u));
                const int64_t lbase = ((bh * KD + ch) * 2 + (side == 0 ? 1 : 0)) * (int64_t)a.kcap + a.o_off;
                a.oidx[lbase + r] = (uint16_t)(slab * KO_ROWS + row + a.tok_base);
                a.oval[lbase + r] = (uint16_t)(kept[i] >> 16);
            }
        }
        // the merged bitmap, tile-major, for the Q pass: [bh][tile][channel][2 words]
        for (int e = tid; e < ntl * KD * 2; e += KO_THREADS)
            a.obits[((bh * ntiles + (int64_t)slab * KO_NT) * KD) * 2 + e] = X[e] | X[KO_NT * KD * 2 + e];
    } else {
        // no outliers: nothing to select
        for (int e = tid; e < 2 * KO_NT * KD * 2; e += KO_THREADS) X[e] = 0u;
        if (tid < KD) mean_l[tid] = 0.0f;
        __syncthreads();
    }

    KO_CLK(12);
    // ---------------------------------------------------------------- P7: the dense part, wave -> (tile, channel half)
    {
        const int tl = wave >> 1;
        if (tl < ntl) {
            const int hf = lane >> 5, cp = 32 * (wave & 1) + (lane & 31);
            const uint4 mA4 = *(const uint4*)&X[(tl * KD + 2 * cp) * 2];                       // side 0: A.w0, A.w1, B.w0, B.w1
            const uint4 mB4 = *(const uint4*)&X[((KO_NT + tl) * KD + 2 * cp) * 2];            // side 1
            const uint32_t a0 = mA4.x | mB4.x, a1 = mA4.y | mB4.y, b0 = mA4.z | mB4.z, b1 = mA4.w | mB4.w;
            const uint32_t mA = hf ? a1 : a0, mB = hf ? b1 : b0;
            const bool gA = G == 64 ? ((a0 | a1) != 0u) : (mA != 0u), gB = G == 64 ? ((b0 | b1) != 0u) : (mB != 0u);
            const float meanA = mean_l[2 * cp], meanB = mean_l[2 * cp + 1];
            uint32_t cwA[NWL], cwB[NWL];
            float qsA, loA, qsB, loB;
            ko_dense<BITS, G>(slabp, tl * 64 + hf * 32, cp, hf, mA, mB, gA, gB, meanA, meanB, f2h_bits(meanA), f2h_bits(meanB), cwA, cwB,
                              qsA, loA, qsB, loB);
            const int tile_g = slab * KO_NT + tl;
            const int tok = a.t_off + tile_g * 64 + hf * 32;
            uint32_t* cA = a.code + (bh * KD + 2 * cp) * a.ldc + tok / CPW;
            uint32_t* cB = cA + a.ldc;
            if constexpr (NWL == 2) {
                *(uint2*)cA = make_uint2(cwA[0], cwA[1]);
                *(uint2*)cB = make_uint2(cwB[0], cwB[1]);
            } else {
                *(uint4*)cA = make_uint4(cwA[0], cwA[1], cwA[2], cwA[3]);
                *(uint4*)cB = make_uint4(cwB[0], cwB[1], cwB[2], cwB[3]);
            }
            if (G == 32 || hf == 0) {
                float* sA = (float*)a.scale + (bh * KD + 2 * cp) * a.lds + tok / G;
                float* nA = (float*)a.mn + (bh * KD + 2 * cp) * a.lds + tok / G;
                sA[0] = qsA; nA[0] = loA; sA[a.lds] = qsB; nA[a.lds] = loB;
            }
        }
    }
    KO_CLK(13);
    if (!a.G || (a.dbg & 2)) return;
    __syncthreads();
    // ---------------------------------------------------------------- P8: G += E^T E of the slab (matrix cores, fp32 atomics)
    // The ten 32x32 blocks on and above the block diagonal ONLY (k_solve_kernel mirrors on load): waves 0 / 1 take two blocks that
    // share an operand, waves 2 .. 7 one block each -- at most two operand sets per 16-token step and wave.
    {
        // wave:        0            1            2      3      4      5      6      7
        // blocks:  (0,0) (0,1)  (2,2) (2,3)    (0,2)  (0,3)  (1,1)  (1,2)  (1,3)  (3,3)
        const int tI[8] = {0, 2, 0, 0, 1, 1, 1, 3}, tJ[8] = {1, 3, 2, 3, 1, 2, 3, 3};
        const int I = tI[wave], J = tJ[wave];
        const bool two = wave < 2;                          // also the diagonal block (I, I)
        const bool diag = I == J;
        float16_t acc0, acc1;
#pragma unroll
        for (int q = 0; q < 16; q++) { acc0[q] = 0.0f; acc1[q] = 0.0f; }
        const int ks_n = ntl * 4;
        for (int ks = 0; ks < ks_n; ks++) {
            const half8_t fi = ko_operand(slabp, 16 * ks, I, lane);
            if (diag) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fi, fi, acc0, 0, 0, 0);
            } else {
                const half8_t fj = ko_operand(slabp, 16 * ks, J, lane);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fi, fj, acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fi, fi, acc1, 0, 0, 0);
            }
        }
        KO_CLK(14);
        if (!(a.dbg & 1)) {
            float* g = a.G + bh * (int64_t)(KD * KD);
            const int x31 = lane & 31, kg = lane >> 5;
            // C layout of the 32x32 MFMA: lane l, reg q -> row (q & 3) + 8 (q >> 2) + 4 (l >> 5), col l & 31
#pragma unroll
            for (int q = 0; q < 16; q++)
                unsafeAtomicAdd(&g[(32 * I + (q & 3) + 8 * (q >> 2) + 4 * kg) * KD + 32 * J + x31], acc0[q]);
            if (two) {
#pragma unroll
                for (int q = 0; q < 16; q++)
                    unsafeAtomicAdd(&g[(32 * I + (q & 3) + 8 * (q >> 2) + 4 * kg) * KD + 32 * I + x31], acc1[q]);
            }
        }
        KO_CLK(15);
    }
}

// ---------------------------------------------------------------------------------------------------- the chain's dense kernel
// k_dense_kernel: what k_main_kernel (kfused.hip) does -- fill + quantize + pack + error + per-head Gram, behind k_select_kernel's
// bitmap and row means -- with this file's slab machinery instead of register-resident tiles.  A workgroup of 256 threads walks
// `spw` consecutive 128-token slabs of one head through TWO LDS buffers: the next slab arrives by LDS-DMA (global_load_lds_dwordx4,
// no registers, nothing to wait for until the buffer is needed: the rotation of the slab layout is applied on the SOURCE side, lane
// l of a 4-row piece fetching the 16-byte chunk that belongs at its linear place) while the current one is worked on:
//   barrier -> substitute (the outliers of k_select's bitmap replaced in LDS by fp16(mean): one 2-byte store each) -> barrier ->
//   dense part (ko_dense: mask-free, half of tile_fast's vector instructions; error left in place) -> barrier ->
//   Gram: the ten upper 32x32 blocks of G = E^T E split 3 / 3 / 2 / 2 over the four waves, accumulated IN REGISTERS across the slabs.
// No exchange, no atomics: the partial Gram matrix of the workgroup is stored once (upper blocks; k_solve_kernel mirrors on load
// and adds the workgroups of a head).  Bit-identical payload to k_main_kernel (same arithmetic); the Gram sums differ in their
// order of additions only.  (First version, 512 threads / 256-token slabs / register prefetch at four waves per SIMD: 1.0 ms against
// k_main_kernel's 0.55 -- 128 registers do not hold the accumulators, the prefetch and the dense part, and a scratch reload
// between a prefetch and its use waits for the prefetch.)
struct KdArgs {
    const uint16_t* x;       // [BH][T][128]
    const uint32_t* obits;   // [BH][T/64][128][2] or null
    const float* omean;      // [BH][128] (with obits)
    int T, spw, nwg;         // slabs per workgroup, workgroups per head
    uint32_t* code; void* scale; void* mn;
    int64_t ldc, lds;
    int t_off;
    float* gpart;            // [BH][nwg][128][128] (upper 32x32 blocks written), or null
    uint16_t* eout;          // [BH][T][128] fp16: the error matrix written out for the Q pass (lr_qpass_tm_mfma_kernel), or null
    const uint32_t* only_if;
};

constexpr int KD_NT = 2;                  // 64-token tiles per slab
constexpr int KD_ROWS = 64 * KD_NT;
constexpr int KD_THREADS = 256;
constexpr int KD_BUF = KD_ROWS * 256;     // bytes of one slab buffer
constexpr int KD_XB = KD_NT * KD * 2 * 4; // bytes of one slab's bitmap
constexpr int KD_LDS = 2 * KD_BUF + 2 * KD_XB + KD * 4;

typedef short kd_s4 __attribute__((ext_vector_type(4)));

// one LDS-DMA piece: 64 lanes x 16 bytes from the lanes' own global addresses to lds_addr + 16 lane (M0 saved and restored: the
// compiler owns it).  Invisible to the compiler's wait counting: the caller waits (s_waitcnt vmcnt) before the data is read.
__device__ __forceinline__ void kd_dma16(const void* g, uint32_t lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
// workgroup barrier that waits for this wave's LDS traffic only (__syncthreads() also drains the vector-memory queue: the DMA in flight)
__device__ __forceinline__ void kd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MFMA operand of k-step ks from the rotated slab through the compiler's own transposing read (its wait counts, its scheduling):
// `off` = the lane's byte offset for k-step 0 (ko_operand's address), + 4096 per k-step (16 rows of 256 bytes, same rotation)
__device__ __forceinline__ half8_t kd_operand(unsigned char* slab, uint32_t off) {
    typedef __attribute__((address_space(3))) kd_s4* lp_t;
    union { half8_t h; kd_s4 s[2]; } cv;
    cv.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(slab + off));
    cv.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(slab + off + 1024u));
    return cv.h;
}
__device__ __forceinline__ uint32_t kd_operand_off(int S, int lane) {
    const int kg = lane >> 5, i = lane & 15, c0 = 32 * S + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
    return ko_off(8 * kg + (i >> 2), c0);
}

// Which upper 32x32 blocks a wave accumulates (kfused.hip's split: 10 operand-set reads per k-step over the workgroup)
// w0: (0,0) (0,1) (1,1)   w1: (2,2) (2,3) (3,3)   w2: (0,2) (0,3)   w3: (1,2) (1,3)
template <int W> struct KdBlocks;
template <> struct KdBlocks<0> { static constexpr int n = 3; static constexpr int I[3] = {0, 0, 1}, J[3] = {0, 1, 1}; static constexpr int need = 0x3; };
template <> struct KdBlocks<1> { static constexpr int n = 3; static constexpr int I[3] = {2, 2, 3}, J[3] = {2, 3, 3}; static constexpr int need = 0xC; };
template <> struct KdBlocks<2> { static constexpr int n = 2; static constexpr int I[3] = {0, 0, 0}, J[3] = {2, 3, 3}; static constexpr int need = 0xD; };
template <> struct KdBlocks<3> { static constexpr int n = 2; static constexpr int I[3] = {1, 1, 1}, J[3] = {2, 3, 3}; static constexpr int need = 0xE; };

template <int W>
__device__ __forceinline__ void kd_gram(unsigned char* buf, int ks_n, int lane, float16_t (&acc)[3]) {
    typedef KdBlocks<W> WB;
    uint32_t off[4];
#pragma unroll
    for (int S = 0; S < 4; S++) off[S] = kd_operand_off(S, lane);
#pragma unroll 2
    for (int ks = 0; ks < ks_n; ks++) {
        half8_t f[4];
#pragma unroll
        for (int S = 0; S < 4; S++)
            if (WB::need & (1 << S)) f[S] = kd_operand(buf, off[S] + 4096u * ks);
#pragma unroll
        for (int b = 0; b < WB::n; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[WB::I[b]], f[WB::J[b]], acc[b], 0, 0, 0);
    }
}
template <int W>
__device__ __forceinline__ void kd_gram_store(float* g, int lane, const float16_t (&acc)[3]) {
    typedef KdBlocks<W> WB;
    const int x31 = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int b = 0; b < WB::n; b++)
#pragma unroll
        for (int q = 0; q < 16; q++) g[(32 * WB::I[b] + (q & 3) + 8 * (q >> 2) + 4 * kg) * KD + 32 * WB::J[b] + x31] = acc[b][q];
}

template <int BITS, int G>
__global__ __launch_bounds__(KD_THREADS, 2) void k_dense_kernel(KdArgs a) {
    constexpr int CPW = 32 / BITS;
    constexpr int NWL = 32 * BITS / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float* mean_l = (float*)(sm + 2 * KD_BUF + 2 * KD_XB);            // [128]
    const int wg = blockIdx.x;
    const int64_t bh = blockIdx.y;
    if (a.only_if && a.only_if[bh] == 0u) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, ntiles = T >> 6;
    const int nslab = (ntiles + KD_NT - 1) / KD_NT;
    const int s_lo = wg * a.spw, s_hi = min(nslab, s_lo + a.spw);
    if (s_lo >= s_hi) return;                                         // (the launcher leaves no workgroup without a slab when gpart is read)
    if (tid < KD) mean_l[tid] = a.obits ? a.omean[bh * KD + tid] : 0.0f;
    if (!a.obits) {
        for (int i = tid; i < 2 * KD_XB / 4; i += KD_THREADS) ((uint32_t*)(sm + 2 * KD_BUF))[i] = 0u;
    }
    float16_t acc[3];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[b][q] = 0.0f;
    // DMA source of this lane inside a 4-row piece: row lane >> 4, the chunk whose rotated place is lane & 15
    const unsigned char* xl = (const unsigned char*)(a.x + bh * (int64_t)T * KD) + (lane >> 4) * 256 + ((((lane & 15) - 4 * (lane >> 4)) & 15) << 4);
    const unsigned char* bl = a.obits ? (const unsigned char*)(a.obits + bh * (int64_t)ntiles * (KD * 2)) + lane * 16 : nullptr;
    const uint32_t sm0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)sm);
    // wave w brings rows 32 w .. 32 w + 31 of the slab (8 pieces), waves 0 / 1 one half of the slab's bitmap each
    auto dma_slab = [&](int slab, int b) {
        const int row0 = slab * KD_ROWS + 32 * wave;
        const unsigned char* g = xl + (int64_t)row0 * 256;
        const uint32_t l = sm0 + (uint32_t)(b * KD_BUF + 32 * wave * 256);
        if (row0 < T) {                                               // (T is a multiple of 64: a wave's 32 rows are all there or none is)
#pragma unroll
            for (int j = 0; j < 8; j++) kd_dma16(g + j * 1024, l + j * 1024);
        }
        if (bl && wave < KD_NT && slab * KD_NT + wave < ntiles) {
            // (tile = 128 channels x 8 bytes = 1 KiB: the wave's piece)
            kd_dma16(bl + (int64_t)(slab * KD_NT + wave) * 1024, sm0 + (uint32_t)(2 * KD_BUF + b * KD_XB + wave * 1024));
        }
    };
    dma_slab(s_lo, 0);
    const int btile = tid >> 7, bch = tid & 127;
    KD_CLK_DECL;
#pragma unroll 1
    for (int slab = s_lo; slab < s_hi; slab++) {
        const int b = (slab - s_lo) & 1;
        unsigned char* buf = sm + b * KD_BUF;
        const uint32_t* X = (const uint32_t*)(sm + 2 * KD_BUF + b * KD_XB);
        const int ntl = min(KD_NT, ntiles - slab * KD_NT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of the slab have landed
        kd_barrier();                                                 // ... everyone's; and everyone is done with the other buffer
        KD_CLK(0);
        if (slab + 1 < s_hi) dma_slab(slab + 1, b ^ 1);
        if (a.obits && btile < ntl) {   // substitute: one 2-byte store per outlier of (tile, channel)
            const uint2 bm = *(const uint2*)&X[(btile * KD + bch) * 2];
            unsigned long long m = (unsigned long long)bm.x | ((unsigned long long)bm.y << 32);
            const uint16_t sv = f2h_bits(mean_l[bch]);
            while (m) {
                const int t = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                *(uint16_t*)(buf + ko_off(btile * 64 + t, bch)) = sv;
            }
        }
        kd_barrier();
        KD_CLK(1);
        {
            const int tl = wave >> 1;
            if (tl < ntl) {
                const int hf = lane >> 5, cp = 32 * (wave & 1) + (lane & 31);
                const uint4 m4 = *(const uint4*)&X[(tl * KD + 2 * cp) * 2];                    // A.w0, A.w1, B.w0, B.w1
                const uint32_t mA = hf ? m4.y : m4.x, mB = hf ? m4.w : m4.z;
                const bool gA = G == 64 ? ((m4.x | m4.y) != 0u) : (mA != 0u), gB = G == 64 ? ((m4.z | m4.w) != 0u) : (mB != 0u);
                const float meanA = mean_l[2 * cp], meanB = mean_l[2 * cp + 1];
                uint32_t cwA[NWL], cwB[NWL];
                float qsA, loA, qsB, loB;
                ko_dense<BITS, G>(buf, tl * 64 + hf * 32, cp, hf, mA, mB, gA, gB, meanA, meanB, f2h_bits(meanA), f2h_bits(meanB), cwA, cwB,
                                  qsA, loA, qsB, loB);
                const int tok = a.t_off + (slab * KD_NT + tl) * 64 + hf * 32;
                uint32_t* cA = a.code + (bh * KD + 2 * cp) * a.ldc + tok / CPW;
                uint32_t* cB = cA + a.ldc;
                if constexpr (NWL == 2) {
                    *(uint2*)cA = make_uint2(cwA[0], cwA[1]);
                    *(uint2*)cB = make_uint2(cwB[0], cwB[1]);
                } else {
                    *(uint4*)cA = make_uint4(cwA[0], cwA[1], cwA[2], cwA[3]);
                    *(uint4*)cB = make_uint4(cwB[0], cwB[1], cwB[2], cwB[3]);
                }
                if (G == 32 || hf == 0) {
                    float* sA = (float*)a.scale + (bh * KD + 2 * cp) * a.lds + tok / G;
                    float* nA = (float*)a.mn + (bh * KD + 2 * cp) * a.lds + tok / G;
                    sA[0] = qsA; nA[0] = loA; sA[a.lds] = qsB; nA[a.lds] = loB;
                }
            }
        }
        KD_CLK(2);
        if (a.gpart) {
            kd_barrier();
            KD_CLK(3);
            if (a.eout && 32 * wave < 64 * ntl) {
                // the error rows 32 wave .. + 31 of the slab leave for the Q pass: linear 16-byte LDS reads, the rotation undone
                // on the global side -- every store instruction covers four whole 256-byte rows.  The kernel is bound by its vector
                // instructions, the writes ride under them.
                uint16_t* eg = a.eout + (bh * T + (int64_t)slab * KD_ROWS + 32 * wave + (lane >> 4)) * KD + 8 * (((lane & 15) - 4 * (lane >> 4)) & 15);
                const unsigned char* el = buf + (32 * wave) * 256 + lane * 16;
#pragma unroll
                for (int j = 0; j < 8; j++) *(uint4*)(eg + j * 4 * KD) = *(const uint4*)(el + j * 1024);
            }
            const int ks_n = ntl * 4;
            switch (wave) {
            case 0: kd_gram<0>(buf, ks_n, lane, acc); break;
            case 1: kd_gram<1>(buf, ks_n, lane, acc); break;
            case 2: kd_gram<2>(buf, ks_n, lane, acc); break;
            default: kd_gram<3>(buf, ks_n, lane, acc); break;
            }
        }
        KD_CLK(4);
    }
    KD_CLK_OUT;
    if (!a.gpart) return;
    float* g = a.gpart + (bh * a.nwg + wg) * (int64_t)(KD * KD);
    switch (wave) {
    case 0: kd_gram_store<0>(g, lane, acc); break;
    case 1: kd_gram_store<1>(g, lane, acc); break;
    case 2: kd_gram_store<2>(g, lane, acc); break;
    default: kd_gram_store<3>(g, lane, acc); break;
    }
}

inline size_t ko_align(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int KO_FIXED = 848;            // words of the fixed LDS part in front of the owner's list counters

struct KoPlan {
    int S, nlmax, bcap, lcap, xbytes;
    size_t shmem;
    size_t o_cnt, o_ent, o_sum, o_fail, o_kthr, o_base, total, zero_bytes;     // workspace offsets
};

// target = expected candidates per (channel, side) over all T tokens (as kfused.hip: k + 5 sqrt(k) + 8)
bool ko_plan(int64_t BH, int T, int k, KoPlan& p) {
    if (T % 64 || T < 64 || T > 16384) return false;
    const int ntiles = T / 64;
    p.S = (ntiles + KO_NT - 1) / KO_NT;
    if (p.S > KO_MAXS) return false;
    const int nch = (128 + p.S - 1) / p.S, nl = 2 * nch;
    p.nlmax = nl;
    if (k > 0) {
        if (4 * k > T) return false;
        const double target = k + 5.0 * sqrt((double)k) + 8.0;
        int lcap = (int)(1.6 * target + 16.0);
        lcap = ((lcap + 31) / 32) * 32;
        if (lcap > 32 * KO_KPL) return false;
        if (target > 0.25 * T) return false;                        // the guess would let a quarter of the column through
        p.lcap = lcap;
        const double per_bucket = nl * target / p.S;
        p.bcap = ((int)(1.6 * per_bucket + 24.0) + 1) & ~1;
        if (p.bcap > KO_THREADS) return false;                      // the owner reads a segment with 256 threads, two entries each
        if ((int64_t)p.S * p.bcap > 6 * KO_THREADS) return false;   // six kept entries per thread
        const size_t xb = (size_t)std::max(std::max(p.S * p.bcap, nl * lcap), 2 * KO_NT * KD * 2) * 4;
        p.xbytes = (int)((xb + 15) & ~(size_t)15);
    } else {
        p.lcap = 32; p.bcap = 2;
        p.xbytes = 2 * KO_NT * KD * 2 * 4;
    }
    p.shmem = (size_t)KO_ROWS * 256 + p.xbytes + (size_t)(KO_FIXED + p.nlmax * (p.S + 1)) * 4;
    if (p.shmem > 160 * 1024) return false;
    size_t off = 0;
    p.o_fail = off;  off += ko_align((size_t)BH * 4);
    p.zero_bytes = off;
    p.o_cnt = off;   off += ko_align((size_t)BH * p.S * p.S * 8);
    p.o_ent = off;   off += ko_align(k > 0 ? (size_t)BH * p.S * p.S * p.bcap * 8 : 0);
    p.o_sum = off;   off += ko_align((size_t)BH * p.S * 2 * KD * 8);
    p.o_kthr = off;  off += ko_align((size_t)BH * 256 * 8);
    p.o_base = off;  off += ko_align((size_t)BH * p.S * KD * 8);
    p.total = off;
    return true;
}

double ko_inv_norm_cdf(double p) {  // Acklam's rational approximation (relative error 1.2e-9), 0 < p < 1
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                               1.383577518672690e+02,  -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                               6.680131188771972e+01,  -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00,  2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    const double plow = 0.02425;
    if (p < plow) {
        double q = sqrt(-2 * log(p));
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    if (p > 1 - plow) {
        double q = sqrt(-2 * log(1 - p));
        return -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    double q = p - 0.5, rr = q * q;
    return (((((a[0] * rr + a[1]) * rr + a[2]) * rr + a[3]) * rr + a[4]) * rr + a[5]) * q /
           (((((b[0] * rr + b[1]) * rr + b[2]) * rr + b[3]) * rr + b[4]) * rr + 1);
}

}  // namespace

// Does the single-read kernel take this shape?  (fp32 "simulated" arithmetic only: the streaming cache's fp16-stepwise mode keeps
// kfused.hip's chain.)
bool gear_kone_supported(int64_t BH, int T, int group, int bits, int mode, int k) {
    KoPlan p;
    if (mode != GEAR_MODE_FP32 || (group != 64 && group != 32) || (bits != 2 && bits != 4)) return false;
    return ko_plan(BH, T, k, p);
}

size_t gear_kone_workspace(int64_t BH, int T, int k) {
    KoPlan p;
    return ko_plan(BH, T, k, p) ? p.total + 256 : 0;
}

// The chain's dense kernel (k_dense_kernel above) in place of k_main_kernel: fp32 arithmetic, T a multiple of 64.  gpart receives
// nwg upper-block partial Gram matrices per head (returned through *nwg_out).  Returns 1 when the shape is not taken.
int gear_kdense_launch(const void* x, const void* obits, const void* omean, int64_t BH, int T, int group, int bits, void* code, void* scale,
                       void* mn, int64_t ldc, int64_t lds, int t_off, float* gpart, void* eout, int nwg, const uint32_t* only_if, hipStream_t st) {
    if (T % 64 || (group != 64 && group != 32) || (bits != 2 && bits != 4) || BH > 65535 || nwg < 1) return 1;
    const int nslab = (T / 64 + KD_NT - 1) / KD_NT;
    if (nwg > nslab) return 1;                                      // (every partial Gram matrix the solve adds must be written)
    KdArgs a;
    a.x = (const uint16_t*)x; a.obits = (const uint32_t*)obits; a.omean = (const float*)omean; a.T = T;
    a.nwg = nwg; a.spw = (nslab + nwg - 1) / nwg;
    if ((int64_t)a.spw * (nwg - 1) >= nslab) return 1;              // (the last workgroup of a head would hold no slab)
    a.code = (uint32_t*)code; a.scale = scale; a.mn = mn; a.ldc = ldc; a.lds = lds; a.t_off = t_off;
    a.gpart = gpart; a.eout = gpart ? (uint16_t*)eout : nullptr; a.only_if = only_if;
    const size_t shmem = (size_t)KD_LDS;
    const dim3 grid((unsigned)nwg, (unsigned)BH);
#define KD_GO(B, GG)                                                                                                    \
    do {                                                                                                                \
        auto kfn = k_dense_kernel<B, GG>;                                                                               \
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);            \
        hipLaunchKernelGGL(kfn, grid, dim3(KD_THREADS), shmem, st, a);                                                  \
    } while (0)
    if (bits == 2) { if (group == 64) KD_GO(2, 64); else KD_GO(2, 32); }
    else { if (group == 64) KD_GO(4, 64); else KD_GO(4, 32); }
#undef KD_GO
    GEAR_CHECK_LAUNCH("gear_kdense_launch");
    return 0;
}

// headfail (device, [BH] words): non-zero for the heads the caller must redo with the exact chain
const uint32_t* gear_kone_headfail(void* ws, int64_t BH, int T, int k) {
    KoPlan p;
    if (!ko_plan(BH, T, k, p)) return nullptr;
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    return (const uint32_t*)(base + p.o_fail);
}

int gear_kone_launch(const void* x, int64_t BH, int T, int group, int bits, int k, void* code, void* scale, void* mn, int64_t ldc,
                     int64_t lds, int t_off, void* obits, void* oidx, void* oval, int kcap, int o_off, float* G, void* ws,
                     hipStream_t st) {
    KoPlan p;
    if (!ko_plan(BH, T, k, p)) { gear_set_error("gear_kone_launch: unsupported shape"); return -1; }
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    if (hipMemsetAsync(base, 0, p.zero_bytes, st) != hipSuccess) { gear_set_error("gear_kone_launch: memset failed"); return -2; }
    // the Gram matrices start at zero (the slabs add into them)
    if (G && hipMemsetAsync(G, 0, (size_t)BH * KD * KD * 4, st) != hipSuccess) { gear_set_error("gear_kone_launch: memset failed"); return -2; }
    KoArgs a;
    a.x = (const uint16_t*)x; a.BH = BH; a.T = T; a.S = p.S; a.k = k;
    a.rlen = 1.0f / (float)T;
    a.zthr = 0.0f;
    if (k > 0) {
        const double target = k + 5.0 * sqrt((double)k) + 8.0;
        a.zthr = (float)(-ko_inv_norm_cdf(target / (double)T));
    }
    a.nlmax = p.nlmax; a.bcap = p.bcap; a.lcap = p.lcap; a.xbytes = p.xbytes;
    a.headfail = (uint32_t*)(base + p.o_fail);
    a.xc_cnt = (unsigned long long*)(base + p.o_cnt); a.xc_ent = (unsigned long long*)(base + p.o_ent);
    a.xc_sum = (unsigned long long*)(base + p.o_sum);
    a.kthr = (unsigned long long*)(base + p.o_kthr); a.base = (unsigned long long*)(base + p.o_base);
    {   // a fresh, non-zero tag per call: the granules of every earlier call never match it
        static uint32_t g_epoch = 0x5EED0000u;
        uint32_t ep = __atomic_add_fetch(&g_epoch, 1u, __ATOMIC_RELAXED);
        if (ep == 0u) ep = __atomic_add_fetch(&g_epoch, 1u, __ATOMIC_RELAXED);
        a.epoch = ep;
    }
    a.obits = (uint32_t*)obits; a.oidx = (uint16_t*)oidx; a.oval = (uint16_t*)oval; a.kcap = kcap; a.o_off = o_off; a.tok_base = t_off;
    a.code = (uint32_t*)code; a.scale = scale; a.mn = mn; a.ldc = ldc; a.lds = lds; a.t_off = t_off;
    a.G = G;
    { const char* e = getenv("GEAR_KONE_DBG"); a.dbg = e ? atoi(e) : 0; }
    const dim3 grid((unsigned)(BH * p.S));
    static int n_cu = 0;
    if (!n_cu) {
        int dev_ = 0;
        (void)hipGetDevice(&dev_);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess || n_cu <= 0) n_cu = 1;
    }
#define KO_GO(B, GG)                                                                                                    \
    do {                                                                                                                \
        auto kfn = k_one_kernel<B, GG>;                                                                                 \
        if (p.S > 1 && k > 0) {                                                                                         \
            /* forward progress of the exchange: a head's S workgroups (8 S consecutive block ids at most) must co-reside; */ \
            /* the occupancy query costs milliseconds on the host: once per instantiation and LDS size */              \
            static size_t occ_shmem = (size_t)-1;                                                                       \
            static int occ = 0;                                                                                         \
            if (occ_shmem != p.shmem) {                                                                                 \
                (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.shmem);  \
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kfn, KO_THREADS, p.shmem) != hipSuccess) occ = 0; \
                occ_shmem = p.shmem;                                                                                    \
            }                                                                                                           \
            GEAR_CHECK_ARG((int64_t)occ * n_cu >= 8 * p.S, "gear_kone_launch: only %d workgroups fit on the device at once, the "  \
                           "exchange of a head needs %d", occ * n_cu, 8 * p.S);                                         \
        } else (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.shmem);   \
        hipLaunchKernelGGL(kfn, grid, dim3(KO_THREADS), p.shmem, st, a);                                                \
    } while (0)
    if (bits == 2) { if (group == 64) KO_GO(2, 64); else KO_GO(2, 32); }
    else { if (group == 64) KO_GO(4, 64); else KO_GO(4, 32); }
#undef KO_GO
    GEAR_CHECK_LAUNCH("gear_kone_launch");
    return 0;
}

extern "C" int gear_kone_fallback_heads(void) {
    uint32_t v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_kone_fallbacks), sizeof(v)) != hipSuccess) return -1;
    return (int)v;
}
#ifdef GEAR_KO_CLK
extern "C" int gear_debug_ko_clk(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ko_clk_buf), sizeof(unsigned long long) * 16 * 16384);
}
#endif

// number of exchange polls that ran into their bound since the library was loaded (0 on a healthy device); synchronises
extern "C" int gear_kone_timeouts(void) {
    uint32_t v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_kone_timeouts), sizeof(v)) != hipSuccess) return -1;
    return (int)v;
}
