// attention.hip -- single-token decode attention straight over the compressed GEAR cache (gfx950, head_dim 128).
//
//   scores[t] = q . Khat[t,:] / sqrt(D)          Khat = dequant(K codes) + Qk Pk^T, outlier entries = stored value + Qk Pk^T
//   out[d]    = softmax(scores) . Vhat[:,d]      Vhat likewise; an fp16 window of recent tokens (the reference's residual
//                                                buffer, modeling_llamagear.py:256-261, :329-333) joins the same softmax.
//
// Replaces, in ONE pass over the packed bytes: cuda_bmm_fA_qB_outer for K and V (cuda_supported_gear/quant/matmul.py:178),
// the low-rank bmm chains of matmul_withlrap (modeling_llamagear.py:64-108), the fp32 softmax (:313) and the
// [B,H,1,T] score round trip between them -- plus the sparse outlier term the reference's fused path never stores.
//
// Flash-decoding split: grid (splits, B*Hq).  A workgroup owns a chunk of TC compressed tokens of one query head:
//   1. K side, lanes along tokens (one packed word = 32/bits tokens of one channel), 4 row-subsets over channels,
//      partial sums merged in LDS;  + Qk[t,:] . (Pk^T q);  + sum over K outliers in the chunk q[d] (val - dequant)
//   2. chunk-local softmax statistics (m, l), p = exp(s - m)
//   3. V side, lanes along (token-subset, packed word = 32/bits channels);  w = Qv^T p for the low-rank term;
//      + V outliers of the chunk's tokens that fall into this head
// A second kernel merges the splits and the fp16 window and applies Pv w.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "ktile.h"

namespace {

constexpr int AD = 128;       // head_dim
constexpr int TC_MAX = 2048;  // chunk of compressed tokens per workgroup

struct AttnArgs {
    const uint16_t* q;       // [B*Hq, 128]
    // K payload (channel-major)
    const uint32_t* kcode;   // [B*Hkv, 128, ldk] words
    const void* kscale;      // [B*Hkv, 128, lsk]
    const void* kmn;
    const uint16_t* kP;      // [B*Hkv, 128, rk]  channel-side factor
    const uint16_t* kQ;      // [B*Hkv, Tf, rk]   token-side factor (Tf = token capacity of the factor tensor)
    const uint16_t* koidx;   // [B*Hkv, 128, 2*kk] token indices (each half sorted ascending)
    const uint16_t* koval;
    // V payload (token-major)
    const uint32_t* vcode;   // [B*Hkv, Tcap, 128/CPW]
    const void* vscale;      // [B*Hkv, Tcap, 128/g]
    const void* vmn;
    const uint16_t* vP;      // [B*Hkv, 128, rv]
    const uint16_t* vQ;      // [B*Hkv, Tf, rv]
    const uint16_t* voidx;   // [B, Tcap, 2*kv] column index hkv*128 + d (each half sorted ascending)
    const uint16_t* voval;
    int B, Hq, Hkv, T;       // T = number of compressed tokens
    int ldk, lsk;            // K code / scale row pitch
    int tcap_v, tf_k, tf_v;  // token capacity (row count) of V tensors / factor tensors
    int group, rk, rv, kk, kv;
    // K outlier lists of a streaming cache: every (channel, side) list has room for kk_stride entries, the first
    // kk0 + kkb * (blocks appended so far) are valid -- kk0 from the prompt segment, kkb per `seglen` tokens after seg0.
    // Plain payloads: kk_stride == kk, kkb == 0.
    int kk_stride, kk0, kkb;
    // chunk-major sparse tiles of a streaming cache (gear_cache_tiles_build): per (KV head, 128-token chunk) up to ktile_cap
    // entries (channel | token-in-chunk << 7 | fp16(value - dequant) << 16), kcnt = their number (-1: too many, use the lists);
    // per (KV head, 64-token block) up to vtile_cap entries (token-in-block | channel << 6 | fp16 delta << 16).  The attention
    // chunk then adds its outlier corrections from entries whose position is known at launch: no search, no dependent loads.
    const uint32_t* ktile; const int* kcnt; int ktile_cap, nck;
    const uint32_t* vtile; const int* vcnt; int vtile_cap, nblk;
    int seg0, seglen;        // low-rank factor segments: tokens [0, seg0) use channel factors #0, then one set per `seglen`
                             // tokens (seglen == 0: a single segment).  kP / vP are [nseg, B*Hkv, 128, r].
    int64_t kP_seg_stride, vP_seg_stride;
    const int* dyn;          // optional device state {pos, slot, T, W}: T and W are read from it (hipGraph replay)
    int tc, splits;
    int pslots;              // partial-result slots per query head: splits, + 1 when the fp16 window rides along as one more chunk
    const uint16_t* kwin;    // fp16 window [B*Hkv, wcap, 128] (only read by the window chunk of attn_decode_partial_small)
    const uint16_t* vwin;
    int W, wcap;
    float qscale;
    float* part_o;           // [B*Hq, splits, 128]
    float* part_w;           // [B*Hq, splits, 16]
    float* part_ml;          // [B*Hq, splits, 2]
    // optional chunk index of the sorted outlier lists (gear_outlier_chunk_index): entry [list][b] = first position in
    // the list whose index is >= b * 128.  K lists (bhk, d, side) x (T / 128 + 1) token bounds; V lists (b, t, side) x
    // (Hkv + 1) column bounds.  Lets a 128-token chunk find its outliers without a binary search per list.
    const uint8_t* kochunk;
    const uint8_t* vochunk;
    int nbk, nbv;
    // log2 of group / seglen / (Hq / Hkv) when they are powers of two, else -1: the short-chunk kernel's index arithmetic then has
    // no integer division (each is ~20 vector instructions; the kernel is bound by instruction issue at batch 1)
    int gshift, seglen_shift, nrep_shift;
    // merge folded into the partial kernel (round 6): 1 = every workgroup publishes its partial result with agent-scope stores,
    // counts itself in on arrive[bhq] and the LAST one of a query head merges all of the head's partials into out_fold (what
    // attn_decode_reduce_kernel does in a launch of its own)
    int fold;
    uint32_t* arrive;        // [B*Hq] arrival counters (zero between launches: the merging workgroup resets its head's)
    uint16_t* out_fold;      // [B*Hq, 128]
};
__device__ __forceinline__ uint32_t div_sh(uint32_t x, int d, int sh) { return sh >= 0 ? x >> sh : x / (uint32_t)d; }

__device__ __forceinline__ float block_reduce_max(float v, float* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// the same without the leading barrier: for a `red` array nobody can still be reading (its first use in a pass, or a second array)
__device__ __forceinline__ float block_reduce_max_nb(float v, float* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_reduce_sum_nb(float v, float* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// first index i in [0, n) with a[i] >= key (a ascending)
__device__ __forceinline__ int lower_bound_u16(const uint16_t* a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((int)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// valid entries of a K outlier list when the cache holds Tc compressed tokens
__device__ __forceinline__ int k_list_len(const AttnArgs& a, int Tc) {
    if (a.kkb == 0) return a.kk;
    const int nblk = (a.seglen > 0 && Tc > a.seg0) ? (Tc - a.seg0) / a.seglen : 0;
    return min(a.kk_stride, a.kk0 + nblk * a.kkb);
}

// Positions of a K outlier list that can hold tokens of the chunk [t0, t0 + tn).  Plain payload: one range (binary search to the
// first entry >= t0, the caller stops at the first entry past the chunk).  Streaming cache: the prompt segment's kk0 entries are
// searched the same way; the entries of the 64-token blocks need no search -- block j of the cache wrote positions
// kk0 + j * kkb .. of every list, so the chunk's blocks map to a fixed position range.
__device__ __forceinline__ void k_list_ranges(const AttnArgs& a, const uint16_t* oi, int Tc, int t0, int tn, int (&r0)[2], int (&r1)[2],
                                              bool search = true) {
    r0[0] = r1[0] = r0[1] = r1[1] = 0;
    if (a.kkb == 0) {
        r1[0] = a.kk;
        r0[0] = lower_bound_u16(oi, a.kk, t0);
        return;
    }
    if (t0 < a.seg0 && a.kk0 > 0 && search) {
        r1[0] = a.kk0;
        r0[0] = lower_bound_u16(oi, a.kk0, t0);
    }
    const int te = min(t0 + tn, Tc);
    if (te > a.seg0 && a.seglen > 0) {
        const int b0 = max(t0 - a.seg0, 0) / a.seglen, b1 = (te - a.seg0 + a.seglen - 1) / a.seglen;
        r0[1] = a.kk0 + b0 * a.kkb;
        r1[1] = min(a.kk0 + b1 * a.kkb, a.kk_stride);
    }
}


template <int BITS, typename ST>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(AttnArgs a) {
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    constexpr int NWV = AD / CPW;  // packed words per V token row
    constexpr int MAXSEG = TC_MAX / 64 + 2;
    __shared__ float qs[AD];
    __shared__ float u[MAXSEG][16];
    __shared__ float s[TC_MAX];
    __shared__ float oacc[AD];
    __shared__ float wacc[4][MAXSEG][16];   // one copy per wave, merged in a fixed order
    __shared__ float red[4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int split = blockIdx.x;
    const int64_t bhq = blockIdx.y;
    const int b = (int)(bhq / a.Hq), hq = (int)(bhq % a.Hq);
    const int n_rep = a.Hq / a.Hkv;
    const int hkv = hq / n_rep;
    const int64_t bhk = (int64_t)b * a.Hkv + hkv;
    const int t0 = split * a.tc;
    const int Tc = a.dyn ? a.dyn[2] : a.T;
    const int tn = min(a.tc, Tc - t0);  // tokens in this chunk
    if (tn <= 0) {                      // grid planned for the cache capacity: chunks beyond the current length
        const int64_t pe = bhq * a.pslots + split;
        if (tid < AD) a.part_o[pe * AD + tid] = 0.0f;
        if (tid == 0) { a.part_ml[pe * 2] = -INFINITY; a.part_ml[pe * 2 + 1] = 0.0f; }
        return;
    }
    const ST* kscale = (const ST*)a.kscale;
    const ST* kmn = (const ST*)a.kmn;
    const ST* vscale = (const ST*)a.vscale;
    const ST* vmn = (const ST*)a.vmn;

    if (tid < AD) {
        qs[tid] = h2f_bits(a.q[bhq * AD + tid]) * a.qscale;
        oacc[tid] = 0.0f;
    }
    // factor segments touched by this chunk
    auto seg_of = [&](int t) { return (a.seglen == 0 || t < a.seg0) ? 0 : 1 + (t - a.seg0) / a.seglen; };
    const int seg_first = seg_of(t0), nseg_c = seg_of(t0 + tn - 1) - seg_first + 1;
    for (int i = tid; i < 4 * MAXSEG * 16; i += 256) (&wacc[0][0][0])[i] = 0.0f;
    for (int i = tid; i < a.tc; i += 256) s[i] = 0.0f;
    __syncthreads();
    // u[seg] = Pk[seg]^T q: one factor row (rk contiguous fp16) per thread, reduced over the 128 channels with
    // wave shuffles + one LDS add per wave (the serial 128-deep load loop it replaces dominated small chunks)
    for (int i = tid; i < MAXSEG * 16; i += 256) (&u[0][0])[i] = 0.0f;
    __syncthreads();
    if (a.rk > 0) {
        for (int i = tid; i < nseg_c * AD; i += 256) {   // (i / 128) is wave-uniform: 64 | 128
            const int sl = i / AD, d = i % AD;
            const uint16_t* pk = a.kP + (int64_t)(seg_first + sl) * a.kP_seg_stride + (bhk * AD + d) * a.rk;
            const float qd = qs[d];
            float pr[16];
#pragma unroll
            for (int c = 0; c < 16; c++) pr[c] = 0.0f;
            if (a.rk == 8) { float t[8]; unpack8(*(const uint4*)pk, t);
#pragma unroll
                for (int c = 0; c < 8; c++) pr[c] = qd * t[c]; }
            else {
#pragma unroll
                for (int c = 0; c < 16; c++) if (c < a.rk) pr[c] = qd * h2f_bits(pk[c]); }
#pragma unroll
            for (int c = 0; c < 16; c++) {
                if (c < a.rk) {
                    float v = pr[c];
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                    if (lane == 0) atomicAdd(&u[sl][c], v);
                }
            }
        }
    }
    // ------------------------------------------------------------------ 1. K side
    {
        // lanes = (packed word of the chunk) x (channel subset).  The word-lane count is the smallest power of two
        // covering a full chunk (a.tc / CPW words, at most 64), so that a short chunk (128 tokens = 8 words at 2 bits)
        // still keeps all 256 threads busy: 8 word lanes x 32 channel subsets, 4 channels each, merged with wave
        // shuffles + one LDS add per wave.
        const int nw = (tn + CPW - 1) / CPW;        // words in this chunk
        const int w0 = t0 / CPW;                    // t0 is a multiple of CPW
        const int nwc = a.tc / CPW;
        int wl = 64;
        while (wl > 1 && (wl >> 1) >= nwc) wl >>= 1;
        const int lw = tid & (wl - 1), dsub = tid / wl, nds = 256 / wl;
        for (int wb = 0; wb < nw; wb += wl) {
            const int w = wb + lw;
            float acc[CPW];
#pragma unroll
            for (int j = 0; j < CPW; j++) acc[j] = 0.0f;
            float zacc = 0.0f;
            if (w < nw) {
                const int g = ((w0 + w) * CPW) / a.group;
#pragma unroll 4
                for (int d = dsub; d < AD; d += nds) {
                    const uint32_t word = a.kcode[(bhk * AD + d) * (int64_t)a.ldk + w0 + w];
                    const float sc = ld_st<ST>(kscale + (bhk * AD + d) * (int64_t)a.lsk + g);
                    const float mnv = ld_st<ST>(kmn + (bhk * AD + d) * (int64_t)a.lsk + g);
                    const float qd = qs[d];
                    const float sa = sc * qd;
                    zacc = fmaf(mnv, qd, zacc);
#pragma unroll
                    for (int j = 0; j < CPW; j++) acc[j] = fmaf(sa, (float)((word >> (BITS * j)) & MASK), acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < CPW; j++) {
                float v = acc[j] + zacc;
                for (int msk = wl; msk < 64; msk <<= 1) v += __shfl_xor(v, msk, 64);
                acc[j] = v;
            }
            // the four waves add their channel subsets in a fixed order (float atomics would make the association,
            // and with it the generated tokens, run-to-run dependent)
            for (int turn = 0; turn < 4; turn++) {
                if ((tid >> 6) == turn && lane < wl) {
#pragma unroll
                    for (int j = 0; j < CPW; j++) {
                        const int t = w * CPW + j;
                        if (t < tn) s[t] += acc[j];
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    // low-rank: s[t] += Qk[t,:] . u
    if (a.rk > 0) {
        for (int t = tid; t < tn; t += 256) {
            const uint16_t* qp = a.kQ + (bhk * a.tf_k + t0 + t) * (int64_t)a.rk;
            const float* us = u[seg_of(t0 + t) - seg_first];
            float acc = 0.0f;
            if (a.rk == 8) {
                float tq[8];
                unpack8(*(const uint4*)qp, tq);
#pragma unroll
                for (int c = 0; c < 8; c++) acc = fmaf(tq[c], us[c], acc);
            } else {
                for (int c = 0; c < a.rk; c++) acc = fmaf(h2f_bits(qp[c]), us[c], acc);
            }
            s[t] += acc;
        }
    }
    __syncthreads();
    // K outliers inside the chunk: s[t] += q[d] (val - dequant(t, d))
    if (a.kk > 0 && tid < AD) {
        const int d = tid;
        const float qd = qs[d];
        for (int side = 0; side < 2; side++) {
            const uint16_t* oi = a.koidx + ((bhk * AD + d) * 2 + side) * (int64_t)a.kk_stride;
            const uint16_t* ov = a.koval + ((bhk * AD + d) * 2 + side) * (int64_t)a.kk_stride;
            int r0[2], r1[2];
            k_list_ranges(a, oi, Tc, t0, tn, r0, r1);
            for (int rg = 0; rg < 2; rg++) {
                for (int i = r0[rg]; i < r1[rg]; i++) {
                    const int t = oi[i];
                    if (t >= t0 + tn) break;
                    if (t < t0) continue;
                    const uint32_t word = a.kcode[(bhk * AD + d) * (int64_t)a.ldk + t / CPW];
                    const int g = t / a.group;
                    const float sc = ld_st<ST>(kscale + (bhk * AD + d) * (int64_t)a.lsk + g);
                    const float mnv = ld_st<ST>(kmn + (bhk * AD + d) * (int64_t)a.lsk + g);
                    const float deq = fmaf(sc, (float)((word >> (BITS * (t % CPW))) & MASK), mnv);
                    atomicAdd(&s[t - t0], qd * (h2f_bits(ov[i]) - deq));
                }
            }
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ 2. chunk softmax statistics
    float lm = -INFINITY;
    for (int t = tid; t < tn; t += 256) lm = fmaxf(lm, s[t]);
    const float m = block_reduce_max(lm, red);
    float ls = 0.0f;
    for (int t = tid; t < tn; t += 256) {
        float p = __expf(s[t] - m);
        s[t] = p;
        ls += p;
    }
    const float l = block_reduce_sum(ls, red);  // (contains the barrier that publishes s[])
    // ------------------------------------------------------------------ 3. V side
    {
        const int wv = tid % NWV, rsub = tid / NWV;
        constexpr int NRS = 256 / NWV;  // row-subsets
        float acc[CPW];
#pragma unroll
        for (int j = 0; j < CPW; j++) acc[j] = 0.0f;
        float zacc = 0.0f;
        const int gv = (wv * CPW) / a.group;
        const int ngv = AD / a.group;
#pragma unroll 4
        for (int t = rsub; t < tn; t += NRS) {
            const int64_t row = bhk * a.tcap_v + t0 + t;
            const uint32_t word = a.vcode[row * NWV + wv];
            const float sc = ld_st<ST>(vscale + row * ngv + gv);
            const float mnv = ld_st<ST>(vmn + row * ngv + gv);
            const float p = s[t];
            const float sa = sc * p;
            zacc = fmaf(mnv, p, zacc);
#pragma unroll
            for (int j = 0; j < CPW; j++) acc[j] = fmaf(sa, (float)((word >> (BITS * j)) & MASK), acc[j]);
        }
        // lanes with equal wv inside a wave differ in the bits above log2(NWV)
#pragma unroll
        for (int j = 0; j < CPW; j++) {
            float v = acc[j] + zacc;
            for (int msk = NWV; msk < 64; msk <<= 1) v += __shfl_xor(v, msk, 64);
            acc[j] = v;
        }
        for (int turn = 0; turn < 4; turn++) {   // fixed merge order, see the K side
            if ((tid >> 6) == turn && lane < NWV) {
#pragma unroll
                for (int j = 0; j < CPW; j++) oacc[wv * CPW + j] += acc[j];
            }
            __syncthreads();
        }
    }
    // w[seg] = Qv^T p per factor segment: 64-token slabs (aligned, never straddle a segment) go round-robin to the
    // waves; the term Pv[seg] w[seg] is added to the partial output below (the unnormalised partial is linear in p)
    if (a.rv > 0) {
        const int wave = tid >> 6;
        for (int sb = wave * 64; sb < tn; sb += 256) {
            const int t = sb + lane;
            float wl[16];
#pragma unroll
            for (int c = 0; c < 16; c++) wl[c] = 0.0f;
            if (t < tn) {
                const uint16_t* qp = a.vQ + (bhk * a.tf_v + t0 + t) * (int64_t)a.rv;
                const float p = s[t];
                if (a.rv == 8) {
                    float tq[8];
                    unpack8(*(const uint4*)qp, tq);
#pragma unroll
                    for (int c = 0; c < 8; c++) wl[c] = p * tq[c];
                } else {
#pragma unroll
                    for (int c = 0; c < 16; c++)
                        if (c < a.rv) wl[c] = p * h2f_bits(qp[c]);
                }
            }
            const int sl = seg_of(t0 + sb) - seg_first;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                if (c < a.rv) {
                    float v = wl[c];
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                    if (lane == 0) wacc[wave][sl][c] += v;
                }
            }
        }
    }
    // V outliers of the chunk's tokens that fall into this head's 128 columns
    if (a.kv > 0) {
        const int c_lo = hkv * AD, c_hi = c_lo + AD;
        const int ngv = AD / a.group;
        for (int t = tid; t < tn; t += 256) {
            const float p = s[t];
            const int64_t orow = (int64_t)b * a.tcap_v + t0 + t;
            const int64_t row = bhk * a.tcap_v + t0 + t;
            for (int side = 0; side < 2; side++) {
                const uint16_t* oi = a.voidx + (orow * 2 + side) * a.kv;
                const uint16_t* ov = a.voval + (orow * 2 + side) * a.kv;
                for (int i = lower_bound_u16(oi, a.kv, c_lo); i < a.kv; i++) {
                    const int col = oi[i];
                    if (col >= c_hi) break;
                    const int d = col - c_lo;
                    const uint32_t word = a.vcode[row * NWV + d / CPW];
                    const float sc = ld_st<ST>(vscale + row * ngv + d / a.group);
                    const float mnv = ld_st<ST>(vmn + row * ngv + d / a.group);
                    const float deq = fmaf(sc, (float)((word >> (BITS * (d % CPW))) & MASK), mnv);
                    atomicAdd(&oacc[d], p * (h2f_bits(ov[i]) - deq));
                }
            }
        }
    }
    __syncthreads();
    const int64_t po = bhq * a.pslots + split;
#define WSUM(sl, c) ((wacc[0][sl][c] + wacc[1][sl][c]) + (wacc[2][sl][c] + wacc[3][sl][c]))
    if (tid < AD) {
        float o = oacc[tid];
        for (int sl = 0; sl < nseg_c && a.rv > 0; sl++) {
            const uint16_t* pv = a.vP + (int64_t)(seg_first + sl) * a.vP_seg_stride + (bhk * AD + tid) * a.rv;
            if (a.rv == 8) {
                float t[8];
                unpack8(*(const uint4*)pv, t);
#pragma unroll
                for (int c = 0; c < 8; c++) o = fmaf(t[c], WSUM(sl, c), o);
            } else {
                for (int c = 0; c < a.rv; c++) o = fmaf(h2f_bits(pv[c]), WSUM(sl, c), o);
            }
        }
        a.part_o[po * AD + tid] = o;
    }
#undef WSUM
    if (tid == 0) {
        a.part_ml[po * 2] = m;
        a.part_ml[po * 2 + 1] = l;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Short-chunk variant (chunk = 128 tokens = two 64-token slabs; the plan for contexts up to 8k, ranks 0 or 8).
//
// With ~14 KB of payload per workgroup the kernel above is bound by two fixed costs, both measured (tools/exp_attn.py: its
// time does not change between 1k and 4k tokens): a chain of dependent global-load phases (factors -> K -> Qk -> V -> Qv
// -> Pv), and instruction fetch -- a dispatch starts with a cold instruction cache and ~27 KB of unrolled straight-line
// code is ~420 cache lines fetched from L2 one after the other.  Here (1) none of the chunk's addresses depends on computed
// data, so EVERY load is issued into registers before the first barrier, and (2) the code is kept small: one factor row per
// thread (a slab lies inside one factor segment because segments are multiples of 64 tokens), cross-lane sums by a halving
// butterfly (N values over 2^k lanes in N-ish shuffles instead of N*k), per-wave LDS slots merged in a fixed order
// (deterministic; float atomics only in the outlier terms).
constexpr int SC = 128;

// fp16 attention over tn <= 128 consecutive tokens of ONE KV head (rows of 128 fp16 at kb / vb) for the NREP query heads that share
// it: partial output, chunk maximum and exponent sum into slot `slot` of every head's partial arrays.  All 256 threads; every load of
// the chunk (64 KB at 128 tokens: sixteen 16-byte loads per thread) is issued before the first barrier.  Used for the fp16 window of a
// compressed cache -- one more chunk of the flash-decoding split, so that the reduce kernel only merges -- and, chunk after chunk,
// for the UNCOMPRESSED fp16 cache that the reference's harness times beside the compressed models (cuda_supported_gear/test.py:41-62).
// K: 8 threads per token (16 channels each), 32 tokens per pass; V: 16 threads per token row (8 channels each), 16 token subsets.
// ---------------------------------------------------------------------------------------------------------------------
// Publishing a chunk's partial result -- and, with a.fold, the merge of a query head's partials by the LAST workgroup to arrive.
// Protocol (MI355X_MICROARCH.md, inter-workgroup visibility; the one block_fused.hip / kone.hip use): partials by agent-scope
// (write-through) 8-byte stores, `s_waitcnt vmcnt(0)`, barrier, ONE returning agent-scope add on the head's counter; whoever reads
// pslots - 1 is last, loads every partial with agent-scope loads and merges exactly as attn_decode_reduce_kernel does (same slot
// partition, same order of additions: bit-identical output), then puts the counter back to 0.  Saves the reduce launch and the
// kernel boundary in front of it (~6 us per layer at batch 1 against ~4 us of drain + counter + merge here).
constexpr int RS_MAX_F = 66;     // (= RS_MAX below: 65 chunks of 128 tokens + the window)
// arrival counters: eight regions of 65536 heads, a launch takes the next region (two launches in flight on different streams never
// share a counter); zero at load time, zero again after every launch
__device__ uint32_t g_attn_arrive[8 * 65536];

typedef __attribute__((address_space(1))) unsigned long long attn_gu64;
typedef __attribute__((address_space(1))) uint32_t attn_gu32;
__device__ __forceinline__ void st_agent2f(float* p, float x, float y) {
    __hip_atomic_store((attn_gu64*)p, (unsigned long long)__builtin_bit_cast(uint32_t, x) | ((unsigned long long)__builtin_bit_cast(uint32_t, y) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_agent2f(const float* p) {
    const unsigned long long v = __hip_atomic_load((attn_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__builtin_bit_cast(float, (uint32_t)v), __builtin_bit_cast(float, (uint32_t)(v >> 32)));
}

// o: this thread's channel (tid < 128) of the partial output; m, l: the chunk's maximum and exponent sum (block-uniform).
// scr: >= 224 floats of LDS, og: >= 512 floats of LDS, both free at this point.  256 threads.
template <bool FOLD_OK>
__device__ __forceinline__ void part_publish(const AttnArgs& a, int64_t bhq, int slot, float o, float m, float l, float* __restrict__ scr,
                                             float* __restrict__ og) {
    const int tid = threadIdx.x;
    const int64_t po = bhq * a.pslots + slot;
    if (!FOLD_OK || !a.fold) {
        if (tid < AD) a.part_o[po * AD + tid] = o;
        if (tid == 0) {
            a.part_ml[po * 2] = m;
            a.part_ml[po * 2 + 1] = l;
        }
        return;
    }
    const float o2 = __shfl_down(o, 1, 64);
    if (tid < AD && !(tid & 1)) st_agent2f(a.part_o + po * AD + tid, o, o2);
    if (tid == 0) st_agent2f(a.part_ml + po * 2, m, l);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t* flag = (uint32_t*)scr;
    if (tid == 0) flag[0] = __hip_atomic_fetch_add((attn_gu32*)(a.arrive + bhq), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ns = a.pslots;
    if (flag[0] + 1u != (uint32_t)ns) return;
    // ---- the last workgroup of this query head: merge
    float* sml = scr + 8;                 // [2][RS_MAX_F]
    float* coef = sml + 2 * RS_MAX_F;     // [RS_MAX_F]
    float* stat = coef + RS_MAX_F + 3;    // [2]
    const int d2 = tid & 63, grp = tid >> 6;
    float2 mlv = make_float2(-INFINITY, 0.0f);
    if (tid < ns) mlv = ld_agent2f(a.part_ml + (bhq * ns + tid) * 2);
    constexpr int NPO = (RS_MAX_F + 3) / 4;
    float2 pv[NPO];
#pragma unroll
    for (int i = 0; i < NPO; i++) {
        const int sl = min(grp + 4 * i, ns - 1);                       // (clamped: every load unconditional)
        pv[i] = ld_agent2f(a.part_o + (bhq * ns + sl) * AD + 2 * d2);
    }
    if (tid < RS_MAX_F) { sml[tid] = mlv.x; sml[RS_MAX_F + tid] = mlv.y; }
    __syncthreads();
    if (tid < 64) {   // softmax statistics over <= 65 slots: two slots per lane (attn_decode_reduce_kernel's arithmetic, no window)
        const float m1 = tid < ns ? sml[tid] : -INFINITY, l1 = tid < ns ? sml[RS_MAX_F + tid] : 0.0f;
        const float m2 = tid + 64 < ns ? sml[tid + 64] : -INFINITY, l2 = tid + 64 < ns ? sml[RS_MAX_F + tid + 64] : 0.0f;
        float M = fmaxf(fmaxf(m1, m2), fmaxf(-INFINITY, -INFINITY));
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) M = fmaxf(M, __shfl_xor(M, x, 64));
        const float c1 = tid < ns ? __expf(m1 - M) : 0.0f;
        const float c2 = tid + 64 < ns ? __expf(m2 - M) : 0.0f;
        coef[tid] = c1;
        if (tid + 64 < RS_MAX_F) coef[tid + 64] = c2;
        float L = fmaf(c1, l1, fmaf(c2, l2, 0.0f)) + 0.0f;
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) L += __shfl_xor(L, x, 64);
        if (tid == 0) { stat[0] = M; stat[1] = L; }
    }
    __syncthreads();
    float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
    for (int i = 0; i < NPO; i++) {
        const int sl = grp + 4 * i;
        if (sl < ns) { o0 = fmaf(coef[sl], pv[i].x, o0); o1 = fmaf(coef[sl], pv[i].y, o1); }
    }
    og[grp * AD + 2 * d2] = o0;
    og[grp * AD + 2 * d2 + 1] = o1;
    __syncthreads();
    if (tid < AD) a.out_fold[bhq * AD + tid] = f2h_bits(((og[tid] + og[AD + tid]) + (og[2 * AD + tid] + og[3 * AD + tid])) / stat[1]);
    if (tid == 0) __hip_atomic_store((attn_gu32*)(a.arrive + bhq), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NREP>
__device__ __forceinline__ void f16_chunk(const AttnArgs& a, const uint16_t* __restrict__ kb, const uint16_t* __restrict__ vb, int tn,
                                          int64_t bhq0, int slot, float* __restrict__ s, float (*__restrict__ op)[AD],
                                          float* __restrict__ red, float* __restrict__ fscr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tn <= 0) {
#pragma unroll 1
        for (int r = 0; r < NREP; r++) part_publish<NREP == 1>(a, bhq0 + r, slot, 0.0f, -INFINITY, 0.0f, fscr, &op[0][0]);
        return;
    }
    const int part = tid & 7, j0 = tid >> 3;           // K: channels 16 part .., tokens j0 + 32 i
    const int c8 = tid & 15, ts = tid >> 4;            // V: channels 8 c8 .., tokens ts + 16 i
    uint4 kr[4][2], vr[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {                      // (tokens past the chunk: the last row again, dropped below)
        const uint4* pk = (const uint4*)(kb + (uint32_t)min(j0 + 32 * i, tn - 1) * AD + part * 16);
        kr[i][0] = pk[0];
        kr[i][1] = pk[1];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) vr[i] = *(const uint4*)(vb + (uint32_t)min(ts + 16 * i, tn - 1) * AD + c8 * 8);
    uint4 qh[NREP][2];
#pragma unroll
    for (int r = 0; r < NREP; r++) {
        const uint4* pq = (const uint4*)(a.q + (bhq0 + r) * AD + part * 16);
        qh[r][0] = pq[0];
        qh[r][1] = pq[1];
    }
#pragma unroll 1
    for (int r = 0; r < NREP; r++) {
        uint4 q0 = qh[0][0], q1 = qh[0][1];
#pragma unroll
        for (int rr = 1; rr < NREP; rr++) { if (r == rr) { q0 = qh[rr][0]; q1 = qh[rr][1]; } }
        float qf[16];
        unpack8(q0, qf);
        unpack8(q1, qf + 8);
#pragma unroll
        for (int c = 0; c < 16; c++) qf[c] *= a.qscale;
        if (r > 0) __syncthreads();                    // (the previous head's reads of s / op are done)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float t[8], acc = 0.0f;
            unpack8(kr[i][0], t);
#pragma unroll
            for (int c = 0; c < 8; c++) acc = fmaf(qf[c], t[c], acc);
            unpack8(kr[i][1], t);
#pragma unroll
            for (int c = 0; c < 8; c++) acc = fmaf(qf[8 + c], t[c], acc);
            acc += __shfl_xor(acc, 4, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 1, 64);
            if (part == 0) s[j0 + 32 * i] = acc;
        }
        __syncthreads();
        const float sv = tid < tn ? s[min(tid, SC - 1)] : -INFINITY;
        const float m = block_reduce_max_nb(sv, red);           // (red: 8 floats -- maximum and sum have their own four, no leading barriers)
        const float p = tid < tn ? __expf(sv - m) : 0.0f;
        if (tid < SC) s[tid] = p;
        const float l = block_reduce_sum_nb(p, red + 4);        // (contains the barrier that publishes s[])
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; c++) acc[c] = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float pt = s[ts + 16 * i];           // (0 beyond the chunk)
            float t[8];
            unpack8(vr[i], t);
#pragma unroll
            for (int c = 0; c < 8; c++) acc[c] = fmaf(pt, t[c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {                  // the wave's four token subsets (lane bits 4, 5)
            acc[c] += __shfl_xor(acc[c], 16, 64);
            acc[c] += __shfl_xor(acc[c], 32, 64);
        }
        if (lane < 16) {
            *(float4*)&op[wave][c8 * 8] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *(float4*)&op[wave][c8 * 8 + 4] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        __syncthreads();
        const float ov = tid < AD ? (op[0][tid] + op[1][tid]) + (op[2][tid] + op[3][tid]) : 0.0f;
        if (NREP == 1 && a.fold) __syncthreads();        // (op is about to be reused by the merge)
        part_publish<NREP == 1>(a, bhq0 + r, slot, ov, m, l, fscr, &op[0][0]);
    }
}

// Halving butterfly over the lane bits TOP, TOP/2, ... (STEPS of them): on entry every lane holds N partial values, on exit
// v[0 .. (N >> STEPS) - 1] hold the sums over the lane group of the values with index base + i, where
// base = sum over steps k of (lane & (TOP >> k)) ? N >> (k + 1) : 0.  Returns base.
template <int N, int TOP, int STEPS>
__device__ __forceinline__ int halving_reduce(float (&v)[N], int lane) {
    int base = 0;
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
        const int mask = TOP >> k, half = N >> (k + 1);
        const bool upper = (lane & mask) != 0;
        if (upper) base += half;
#pragma unroll
        for (int i = 0; i < half; i++) {
            float lo = v[i], hi = v[i + half];
            asm volatile("" : "+v"(lo), "+v"(hi));   // keep two plain selects (the optimiser otherwise turns them into a
                                                     // lane-dependent index into the whole array: N compare/select pairs)
            const float send = upper ? lo : hi;
            const float keep = upper ? hi : lo;
            v[i] = keep + __shfl_xor(send, mask, 64);
        }
    }
    return base;
}

// 8 values summed over the 64 lanes of a wave; lanes 0, 8, ..., 56 end up with value index lane / 8 in the return value
__device__ __forceinline__ float reduce8_over_wave(float (&v)[8], int lane) {
    halving_reduce<8, 32, 3>(v, lane);
    float r = v[0];
    r += __shfl_xor(r, 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

//
// NREP > 1 (grouped-query attention): ONE workgroup serves all NREP query heads of a KV head -- grid (splits, B * Hkv) -- so the
// chunk's codes, scale / zero point, factor rows and sparse-tile entries are loaded ONCE into registers and used NREP times
// (the reference's `mqa` mapping reads the shared KV head once per query head: gemv_cuda.cu:276-279; before round 5 so did this
// kernel -- 70B's 8 query heads per KV head read the compressed cache 8 times per token).  NREP == 1: one query head per workgroup,
// grid (splits, B * Hq), any Hq / Hkv.
// RS: factor row length in the payload (a.rk / a.rv are 0 or RS) -- 4, 8 or 16.  Rank 4 (BASELINE configs[1]) computes at width
// 8 with the upper half zero: its rows are 8-byte loads.
// acc[j] += sa * code_j for the CPW codes of word w, two instructions per element instead of three (extract, convert, multiply-add):
// a code field masked out of the word IS an fp16 number -- the subnormal code * 2^-24 -- and v_fma_mix_f32 takes an fp16 half (low or
// high, by op_sel) as a factor of an fp32 multiply-add.  One shift + one mask give the pair (j, j + CPW / 2); the caller's scale
// carries the 2^24.  Same bits as fmaf(sa, (float)code, acc): the product is the same real number, rounded once.
template <int BITS>
__device__ __forceinline__ void dq_fma_word(uint32_t w, float sa24, float (&acc)[32 / BITS]) {
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t FM = ((1u << BITS) - 1u) * 0x00010001u;
#pragma unroll
    for (int k = 0; k < CPW / 2; k++) {
        const uint32_t y = (w >> (BITS * k)) & FM;
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(acc[k]) : "v"(sa24), "v"(y));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc[k + CPW / 2]) : "v"(sa24), "v"(y));
    }
}
constexpr float DQ_2P24 = 16777216.0f;

template <int BITS, typename ST, int RS, int NREP>
__global__ __launch_bounds__(256, NREP == 1 ? 5 : 1) void attn_decode_partial_small(AttnArgs a) {
    constexpr bool R16 = RS == 16;
    constexpr int RW = R16 ? 16 : 8;   // factor row width the arithmetic runs at
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    constexpr int NWC = SC / CPW;    // K: packed words per channel in the chunk (8 | 16) = lanes along tokens
    constexpr int NDS = 256 / NWC;   // K: channel subsets (32 | 16)
    constexpr int KIT = AD / NDS;    // K: channels per thread (4 | 8)
    constexpr int NWV = AD / CPW;    // V: packed words per token row (8 | 16) = lanes along channels
    constexpr int NRS = 256 / NWV;   // V: token subsets (32 | 16)
    constexpr int VIT = SC / NRS;    // V: tokens per thread (4 | 8)
    constexpr int XS = BITS == 2 ? 3 : 2;   // butterfly steps over the in-wave subset lanes (64 / NWC = 8 | 4 of them)
    __shared__ float qs[AD];
    __shared__ float up[4][RW];       // Pk[seg(slab)]^T q: waves 0,1 -> slab 0 (channels 0-63 / 64-127), waves 2,3 -> slab 1
    __shared__ float sp[4][SC];      // K side: per-wave partial scores
    __shared__ float s[SC];
    __shared__ float op[4][AD];      // V side: per-wave partial outputs
    __shared__ float ot[2][AD];      // Pv[seg(slab)] (Qv^T p)_slab
    __shared__ float oacc[AD];
    __shared__ float wsl[2][RW];      // Qv^T p of the two slabs
    __shared__ float red[8];         // [0..3]: chunk maximum per wave, [4..7]: exponent sum per wave
    __shared__ float ksp[SC];        // K outliers through the sparse tile: sum of q[d] (value - dequant) per token
    __shared__ float vsp[AD];        // V outliers through the sparse tiles: sum of p[t] (value - dequant) per channel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid (heads of a batch entry -- query heads, or KV heads when a workgroup serves a group --, chunks, batch): no division, and
    // the HEAD is the fastest dimension: workgroup i runs on XCD i % 8, so with a multiple of 8 heads every chunk of a head runs on
    // ONE XCD and the sectors its chunks share meet in one L2 -- a K code row's 32 bytes per chunk are half a 64-byte sector, a K
    // scale / zero-point sector serves 8 chunks, the channel factors every chunk.  PMC at batch 16 (7B layer, 276 MB of payload):
    // FETCH_SIZE 674 MB with the chunk fastest (2.4 x, at 4.8 TB/s: the kernel was bound by its own re-fetches) -> see DESIGN section 6
    const int split = blockIdx.y;
    const int b = blockIdx.z;
    int hkv;
    int64_t bhq0;                      // first (NREP == 1: the only) query head of this workgroup
    if (NREP == 1) {
        hkv = (int)div_sh(blockIdx.x, a.Hq / max(a.Hkv, 1), a.nrep_shift);
        bhq0 = (int64_t)b * a.Hq + blockIdx.x;
    } else {
        hkv = (int)blockIdx.x;
        bhq0 = (int64_t)b * a.Hq + (int64_t)hkv * NREP;
    }
    const int64_t bhk = (int64_t)b * a.Hkv + hkv;
    if (split == a.splits) {           // the extra workgroup of the row: the fp16 window as one more chunk (a.pslots == a.splits + 1)
        const int Wn = a.dyn ? a.dyn[3] : a.W;
        f16_chunk<NREP>(a, a.kwin + bhk * a.wcap * (int64_t)AD, a.vwin + bhk * a.wcap * (int64_t)AD, Wn, bhq0, a.splits, s, op, red,
                        &sp[0][0]);
        return;
    }
    const int t0 = split * SC;
    const int Tc = a.dyn ? a.dyn[2] : a.T;
    const int tn = min(SC, Tc - t0);
    if (tn <= 0) {
#pragma unroll 1
        for (int r = 0; r < NREP; r++) part_publish<NREP == 1>(a, bhq0 + r, split, 0.0f, -INFINITY, 0.0f, &sp[0][0], &op[0][0]);
        return;
    }
    const ST* kscale = (const ST*)a.kscale;
    const ST* kmn = (const ST*)a.kmn;
    const ST* vscale = (const ST*)a.vscale;
    const ST* vmn = (const ST*)a.vmn;

    // ------------------------------------------------------------------ every load of the chunk
    const int dq = tid & (AD - 1), slab = tid >> 7;   // factor rows: one (slab, channel) per thread
    // factor segment of each 64-token slab: block-uniform (scalar) arithmetic, then a select by slab
    auto seg_at = [&](int t) { return (a.seglen == 0 || t < a.seg0) ? 0 : 1 + (int)div_sh((uint32_t)(t - a.seg0), a.seglen, a.seglen_shift); };
    const int seg_s0 = seg_at(t0), seg_s1 = seg_at(min(t0 + 64, t0 + tn - 1));
    float qvr[NREP];                   // this thread's channel of every query head of the group
#pragma unroll
    for (int r = 0; r < NREP; r++) qvr[r] = h2f_bits(a.q[(bhq0 + r) * AD + dq]) * a.qscale;
    const int lw = tid & (NWC - 1), dsub = tid / NWC;
    const bool kval = lw * CPW < tn;
    // Addresses = uniform 64-bit base (head / chunk, scalar registers) + a 32-bit lane offset; lanes past the end of a partial
    // last chunk load a valid clamped address and get scale = mn = 0 instead of branching around each load.
    uint32_t kw[KIT];
    float ksc[KIT], kmv[KIT];
    {
        const uint32_t* kc_b = a.kcode + bhk * AD * (int64_t)a.ldk + t0 / CPW;
        const ST* ks_b = kscale + bhk * AD * (int64_t)a.lsk;
        const ST* km_b = kmn + bhk * AD * (int64_t)a.lsk;
        const uint32_t lwc = kval ? (uint32_t)lw : 0u;
        const uint32_t gk = div_sh((uint32_t)(t0 + (int)lwc * CPW), a.group, a.gshift);
        // (one multiply per pitch; the channels of a thread are NDS rows apart: a block-uniform step)
        const uint32_t ko0 = (uint32_t)dsub * (uint32_t)a.ldk + lwc, kstep = (uint32_t)(NDS * a.ldk);
        const uint32_t so0 = (uint32_t)dsub * (uint32_t)a.lsk + gk, sstep = (uint32_t)(NDS * a.lsk);
#pragma unroll
        for (int i = 0; i < KIT; i++) {
            kw[i] = kc_b[ko0 + (uint32_t)i * kstep];
            const float sc = ld_st<ST>(ks_b + so0 + (uint32_t)i * sstep), mv = ld_st<ST>(km_b + so0 + (uint32_t)i * sstep);
            ksc[i] = kval ? sc : 0.0f;
            kmv[i] = kval ? mv : 0.0f;
        }
    }
    const int wv = tid & (NWV - 1), rsub = tid / NWV;
    uint32_t vw[VIT];
    float vsc[VIT], vmv[VIT];
    {
        const uint32_t gv = div_sh((uint32_t)(wv * CPW), a.group, a.gshift), ngv = div_sh((uint32_t)AD, a.group, a.gshift);
        const uint32_t* vc_b = a.vcode + (bhk * a.tcap_v + t0) * (int64_t)NWV;
        const ST* vs_b = vscale + (bhk * a.tcap_v + t0) * (int64_t)ngv;
        const ST* vm_b = vmn + (bhk * a.tcap_v + t0) * (int64_t)ngv;
#pragma unroll
        for (int i = 0; i < VIT; i++) {
            const int t = rsub + NRS * i;
            const bool ok = t < tn;
            const uint32_t tc = ok ? (uint32_t)t : 0u;
            vw[i] = vc_b[tc * NWV + wv];
            const uint32_t so = (a.gshift >= 0 ? tc << (7 - a.gshift) : tc * ngv) + gv;       // (AD = 2^7)
            const float sc = ld_st<ST>(vs_b + so), mv = ld_st<ST>(vm_b + so);
            vsc[i] = ok ? sc : 0.0f;
            vmv[i] = ok ? mv : 0.0f;
        }
    }
    // Every load below is UNCONDITIONAL (a dummy address -- the query row -- where the payload has no such part): a load inside an
    // `if` whose result meets a default value at the join makes the compiler wait for it at the end of the block, and the four
    // such blocks that stood here were four serialized memory round trips before the first score was computed (per-phase
    // timestamps: 9.4 of the workgroup's 16.8 us with outliers in the cache; 7.5 of 14.0 now).  The values are only looked at
    // under the same conditions further down.
    const uint4* dummy16 = (const uint4*)(a.q + bhq0 * AD);
    const uint8_t* dummy1 = (const uint8_t*)dummy16;
    const uint32_t trow = (uint32_t)min(tid, tn - 1);                       // (threads past the chunk read its last row)
    // (segment offsets: two block-uniform products on the scalar unit and a select, not a 64-bit multiply per lane)
    const int64_t kfo0 = (int64_t)seg_s0 * a.kP_seg_stride, kfo1 = (int64_t)seg_s1 * a.kP_seg_stride;
    const int64_t vfo0 = (int64_t)seg_s0 * a.vP_seg_stride, vfo1 = (int64_t)seg_s1 * a.vP_seg_stride;
    const uint4* kpp = a.rk ? (const uint4*)(a.kP + (slab ? kfo1 : kfo0) + bhk * AD * RS + (uint32_t)dq * RS) : dummy16;
    const uint4* kqp = a.rk ? (const uint4*)(a.kQ + (bhk * a.tf_k + t0) * (int64_t)RS + trow * RS) : dummy16;
    const uint4* vpp = a.rv ? (const uint4*)(a.vP + (slab ? vfo1 : vfo0) + bhk * AD * RS + (uint32_t)dq * RS) : dummy16;
    const uint4* vqp = a.rv ? (const uint4*)(a.vQ + (bhk * a.tf_v + t0) * (int64_t)RS + trow * RS) : dummy16;
    auto ldrow = [](const uint4* p) {                  // one factor row: 16 bytes, or 8 (rank 4) with zeros above
        if (RS == 4) { const uint2 t = *(const uint2*)p; return make_uint4(t.x, t.y, 0u, 0u); }
        return p[0];
    };
    const uint4 kp8 = ldrow(kpp), kq8 = ldrow(kqp), vp8 = ldrow(vpp), vq8 = ldrow(vqp);
    const uint4 kp8b = R16 ? kpp[1] : kp8, kq8b = R16 ? kqp[1] : kq8, vp8b = R16 ? vpp[1] : vp8, vq8b = R16 ? vqp[1] : vq8;   // columns 8..15 (rank 16)

    // outlier list ranges of this chunk (chunk index present): K list (channel dq, side tid >> 7), V list (token tid & 127,
    // side tid >> 7) -- two byte loads each, issued with everything else
    const bool has_kidx = a.kochunk && (a.kkb == 0 || t0 < a.seg0);
    const uint8_t* kip = has_kidx ? a.kochunk + ((bhk * AD + dq) * 2 + (tid >> 7)) * a.nbk + split : dummy1;
    const uint8_t* vip = a.vochunk ? a.vochunk + (((int64_t)b * a.tcap_v + t0 + min(tid & (SC - 1), tn - 1)) * 2 + (tid >> 7)) * a.nbv + hkv : dummy1;
    const int ki0 = kip[0], ki1 = kip[1], vi0 = vip[0], vi1 = vip[1];

    // sparse tiles of this chunk: counts + the first two K entries / one V entry per 64-token block and thread, position known
    const bool has_ktile = a.ktile && a.kk > 0, has_vtile = a.vtile && a.kv > 0;
    const int64_t vb = bhk * a.nblk + 2 * split;
    const int* kcp = has_ktile ? a.kcnt + bhk * a.nck + split : (const int*)dummy16;
    const uint32_t* ktp = has_ktile ? a.ktile + (bhk * a.nck + split) * (int64_t)a.ktile_cap + tid : (const uint32_t*)dummy16;
    const int* vcp = has_vtile ? a.vcnt + vb : (const int*)dummy16;
    const uint32_t* vtp = has_vtile ? a.vtile + vb * a.vtile_cap + tid : (const uint32_t*)dummy16;
    const int kc_raw = kcp[0], vc_raw0 = vcp[0], vc_raw1 = vcp[1];
    // (the second entries at offsets the compiler cannot compare with the first's: `ktp[has_ktile ? 256 : 0]` became "copy ke0 unless
    // has_ktile" -- a copy of a value just requested, i.e. s_waitcnt vmcnt(0) in the middle of the requests; found in the ISA)
    uint32_t koff1 = has_ktile ? 256u : 0u, voff1 = has_vtile ? (uint32_t)a.vtile_cap : 0u;
    asm volatile("" : "+v"(koff1), "+v"(voff1));
    const uint32_t ke0 = ktp[0], ke1 = ktp[koff1], ve0 = vtp[0], ve1 = vtp[voff1];

    // everything below runs once per query head of the group on the registers loaded above (a rolled loop: the code stays at
    // its one-head size, which is what a cold instruction cache charges for)
#pragma unroll 1
    for (int r = 0; r < NREP; r++) {
    float qv = qvr[0];
#pragma unroll
    for (int rr = 1; rr < NREP; rr++) qv = (r == rr) ? qvr[rr] : qv;
    if (NREP > 1 && r > 0) __syncthreads();          // (the previous head's reads of the LDS arrays are done)
    // Outliers through the sparse tiles (the streaming cache's normal case) ride on the barriers the dense path has anyway: the
    // K entries are added into ksp[] while the dense scores are being computed, the V entries into vsp[] while the dense V products
    // are (round 4 parked the scores in LDS, added, and read them back: four barriers and two exposed rounds of LDS atomics;
    // 21.1 -> see DESIGN.md section 6).  Tiles that overflowed (count -1) and payloads without tiles take the list paths below.
    const bool ktile_ok = a.kk > 0 && has_ktile && kc_raw >= 0;
    const bool vtile_ok = a.kv > 0 && has_vtile && vc_raw0 >= 0 && (tn <= 64 || vc_raw1 >= 0);
    if (tid < SC) ksp[tid] = 0.0f;
    if (tid < AD) vsp[tid] = 0.0f;
    // ------------------------------------------------------------------ 1. scores
    if (tid < AD) qs[tid] = qv;
    if (a.rk) {   // up[wave][:] = sum over this wave's 64 channels of q[d] Pk[seg(slab)][d][:]
        float pr[8];
        unpack8(kp8, pr);
#pragma unroll
        for (int c = 0; c < 8; c++) pr[c] *= qv;
        const float r = reduce8_over_wave(pr, lane);
        if ((lane & 7) == 0) up[wave][lane >> 3] = r;
        if (R16) {
            unpack8(kp8b, pr);
#pragma unroll
            for (int c = 0; c < 8; c++) pr[c] *= qv;
            const float r2 = reduce8_over_wave(pr, lane);
            if ((lane & 7) == 0) up[wave][8 + (lane >> 3)] = r2;
        }
    }
    __syncthreads();
    if (ktile_ok) {   // chunk-major tile: ksp[t] += q[d] * (value - dequant), entries prefetched with everything else
        const int kc_n = kc_raw;
        if (tid < kc_n) atomicAdd(&ksp[(ke0 >> 7) & 127u], qs[ke0 & 127u] * h2f_bits((uint16_t)(ke0 >> 16)));
        if (tid + 256 < kc_n) atomicAdd(&ksp[(ke1 >> 7) & 127u], qs[ke1 & 127u] * h2f_bits((uint16_t)(ke1 >> 16)));
        const uint32_t* kt = a.ktile + (bhk * a.nck + split) * (int64_t)a.ktile_cap;
        for (int e = tid + 512; e < kc_n; e += 256) {
            const uint32_t ke = kt[e];
            atomicAdd(&ksp[(ke >> 7) & 127u], qs[ke & 127u] * h2f_bits((uint16_t)(ke >> 16)));
        }
    }
    {
        float acc[CPW];
#pragma unroll
        for (int j = 0; j < CPW; j++) acc[j] = 0.0f;
        float zacc = 0.0f;
#pragma unroll
        for (int i = 0; i < KIT; i++) {
            const float qd = qs[dsub + NDS * i];
            zacc = fmaf(kmv[i], qd, zacc);
            dq_fma_word<BITS>(kw[i], (ksc[i] * qd) * DQ_2P24, acc);
        }
#pragma unroll
        for (int j = 0; j < CPW; j++) acc[j] += zacc;
        const int base = halving_reduce<CPW, 32, XS>(acc, lane);   // over the wave's channel subsets; 2 tokens left per lane
        *(float2*)&sp[wave][lw * CPW + base] = make_float2(acc[0], acc[1]);
    }
    __syncthreads();
    float sv = -INFINITY;
    if (tid < tn) {
        float v = (sp[0][tid] + sp[1][tid]) + (sp[2][tid] + sp[3][tid]);
        if (a.rk) {   // tid < 128: slab = tid / 64 = wave
            float tq[8], acc = 0.0f;
            unpack8(kq8, tq);
#pragma unroll
            for (int c = 0; c < 8; c++) acc = fmaf(tq[c], up[2 * wave][c] + up[2 * wave + 1][c], acc);
            if (R16) {
                unpack8(kq8b, tq);
#pragma unroll
                for (int c = 0; c < 8; c++) acc = fmaf(tq[c], up[2 * wave][8 + c] + up[2 * wave + 1][8 + c], acc);
            }
            v += acc;
        }
        if (ktile_ok) v += ksp[tid];
        sv = v;
    }
    if (a.kk > 0 && !ktile_ok) {   // K outliers inside the chunk through the lists: s[t] += q[d] (val - dequant(t, d))
        if (tid < SC) s[tid] = sv;
        __syncthreads();
        {   // one sorted list per (channel, side) = per thread; its entries inside the chunk are [i0, i1)
            const int side = tid >> 7;
            const int64_t list = (bhk * AD + dq) * 2 + side;
            const uint16_t* oi = a.koidx + list * (int64_t)a.kk_stride;
            const uint16_t* ov = a.koval + list * (int64_t)a.kk_stride;
            int r0[2], r1[2];
            if (a.kochunk && a.kkb == 0) { r0[0] = ki0; r1[0] = ki1; r0[1] = r1[1] = 0; }
            else {
                k_list_ranges(a, oi, Tc, t0, tn, r0, r1, a.kochunk == nullptr);
                if (a.kochunk && t0 < a.seg0) { r0[0] = ki0; r1[0] = ki1; }        // prompt segment through its chunk index
            }
            const int64_t ch = bhk * AD + dq;
            for (int rg = 0; rg < 2; rg++) {
            const int i0 = r0[rg], i1 = r1[rg];
            for (int base = i0; base < i1; base += 4) {   // 4 entries per trip: their loads fly together
                int tt[4];
                float val[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool ok = base + j < i1;
                    tt[j] = ok ? (int)oi[base + j] : 0x7FFFFFFF;
                    val[j] = ok ? h2f_bits(ov[base + j]) : 0.0f;
                }
                uint32_t word[4];
                float sc[4], mnv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool ok = tt[j] < t0 + tn && tt[j] >= t0;
                    const int t = ok ? tt[j] : t0;
                    word[j] = a.kcode[ch * (int64_t)a.ldk + t / CPW];
                    sc[j] = ld_st<ST>(kscale + ch * (int64_t)a.lsk + t / a.group);
                    mnv[j] = ld_st<ST>(kmn + ch * (int64_t)a.lsk + t / a.group);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (tt[j] < t0 + tn && tt[j] >= t0) {
                        const int t = tt[j];
                        const float deq = fmaf(sc[j], (float)((word[j] >> (BITS * (t % CPW))) & MASK), mnv[j]);
                        atomicAdd(&s[t - t0], qv * (val[j] - deq));
                    }
                }
                if (tt[3] >= t0 + tn) break;   // (without the chunk index: stop at the first entry past the chunk)
            }
            }
        }
        __syncthreads();
        if (tid < SC) sv = s[tid];
    }
    // ------------------------------------------------------------------ 2. chunk softmax statistics
    // (two arrays and no leading barriers: red[0..3] was last read before the previous barrier of this pass, red[4..7] before the
    // loop-top barrier of the previous head -- two workgroup barriers fewer on the chunk's critical path)
    const float m = block_reduce_max_nb(sv, red);
    const float p = tid < tn ? __expf(sv - m) : 0.0f;
    if (tid < SC) s[tid] = p;
    const float l = block_reduce_sum_nb(p, red + 4);   // (contains the barrier that publishes s[])
    // ------------------------------------------------------------------ 3. V side
    if (vtile_ok) {   // block tiles: vsp[d] += p[t] * (value - dequant); read after the barriers of the dense part below
        const int vc_n0 = vc_raw0, vc_n1 = (tn > 64) ? vc_raw1 : 0;
        if (tid < vc_n0) atomicAdd(&vsp[(ve0 >> 6) & 127u], s[ve0 & 63u] * h2f_bits((uint16_t)(ve0 >> 16)));
        if (tid < vc_n1) atomicAdd(&vsp[(ve1 >> 6) & 127u], s[64 + (ve1 & 63u)] * h2f_bits((uint16_t)(ve1 >> 16)));
        for (int hb = 0; hb < 2; hb++) {   // tiles longer than one entry per thread (rare)
            const int n = hb ? vc_n1 : vc_n0;
            const uint32_t* vt = a.vtile + (bhk * a.nblk + 2 * split + hb) * (int64_t)a.vtile_cap;
            for (int e = tid + 256; e < n; e += 256) {
                const uint32_t ve = vt[e];
                atomicAdd(&vsp[(ve >> 6) & 127u], s[64 * hb + (ve & 63u)] * h2f_bits((uint16_t)(ve >> 16)));
            }
        }
    }
    {
        float acc[CPW];
#pragma unroll
        for (int j = 0; j < CPW; j++) acc[j] = 0.0f;
        float zacc = 0.0f;
#pragma unroll
        for (int i = 0; i < VIT; i++) {
            const float pt = s[rsub + NRS * i];
            zacc = fmaf(vmv[i], pt, zacc);
            dq_fma_word<BITS>(vw[i], (vsc[i] * pt) * DQ_2P24, acc);
        }
#pragma unroll
        for (int j = 0; j < CPW; j++) acc[j] += zacc;
        const int base = halving_reduce<CPW, 32, XS>(acc, lane);   // over the wave's token subsets
        *(float2*)&op[wave][wv * CPW + base] = make_float2(acc[0], acc[1]);
    }
    if (a.rv && tid < SC) {   // wsl[slab][:] = sum over the slab's 64 tokens of p[t] Qv[t][:]   (slab = wave)
        float wl[8];
        unpack8(vq8, wl);
#pragma unroll
        for (int c = 0; c < 8; c++) wl[c] *= p;
        const float r = reduce8_over_wave(wl, lane);
        if ((lane & 7) == 0) wsl[wave][lane >> 3] = r;
        if (R16) {
            unpack8(vq8b, wl);
#pragma unroll
            for (int c = 0; c < 8; c++) wl[c] *= p;
            const float r2 = reduce8_over_wave(wl, lane);
            if ((lane & 7) == 0) wsl[wave][8 + (lane >> 3)] = r2;
        }
    }
    __syncthreads();
    if (a.rv) {   // the unnormalised partial is linear in p: ot[slab][d] = Pv[seg(slab)][d][:] . wsl[slab]
        float t[8], acc = 0.0f;
        unpack8(vp8, t);
#pragma unroll
        for (int c = 0; c < 8; c++) acc = fmaf(t[c], wsl[slab][c], acc);
        if (R16) {
            unpack8(vp8b, t);
#pragma unroll
            for (int c = 0; c < 8; c++) acc = fmaf(t[c], wsl[slab][8 + c], acc);
        }
        ot[slab][dq] = (slab * 64 < tn) ? acc : 0.0f;
        __syncthreads();
    }
    float o = 0.0f;
    if (tid < AD) {
        o = (op[0][tid] + op[1][tid]) + (op[2][tid] + op[3][tid]);
        if (a.rv) o += ot[0][tid] + ot[1][tid];
        if (vtile_ok) o += vsp[tid];
    }
    if (a.kv > 0 && !vtile_ok) {   // V outliers of the chunk's tokens that fall into this head's 128 columns, through the lists
        if (tid < AD) oacc[tid] = o;
        __syncthreads();
        const int tok = tid & (SC - 1), side = tid >> 7;   // one sorted list per (token, side) = per thread
        if (tok < tn) {
            const int c_lo = hkv * AD, c_hi = c_lo + AD;
            const int ngv = AD / a.group;
            const int64_t orow = (int64_t)b * a.tcap_v + t0 + tok;
            const int64_t row = bhk * a.tcap_v + t0 + tok;
            const int64_t list = orow * 2 + side;
            const uint16_t* oi = a.voidx + list * a.kv;
            const uint16_t* ov = a.voval + list * a.kv;
            const float pt = s[tok];
            int i0, i1 = a.kv;
            if (a.vochunk) { i0 = vi0; i1 = vi1; }
            else i0 = lower_bound_u16(oi, a.kv, c_lo);
            for (int base = i0; base < i1; base += 4) {
                int cc[4];
                float val[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool ok = base + j < i1;
                    cc[j] = ok ? (int)oi[base + j] : 0x7FFFFFFF;
                    val[j] = ok ? h2f_bits(ov[base + j]) : 0.0f;
                }
                uint32_t word[4];
                float sc[4], mnv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int d = cc[j] < c_hi ? cc[j] - c_lo : 0;
                    word[j] = a.vcode[row * NWV + d / CPW];
                    sc[j] = ld_st<ST>(vscale + row * ngv + d / a.group);
                    mnv[j] = ld_st<ST>(vmn + row * ngv + d / a.group);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (cc[j] < c_hi) {
                        const int d = cc[j] - c_lo;
                        const float deq = fmaf(sc[j], (float)((word[j] >> (BITS * (d % CPW))) & MASK), mnv[j]);
                        atomicAdd(&oacc[d], pt * (val[j] - deq));
                    }
                }
                if (cc[3] >= c_hi) break;
            }
        }
        __syncthreads();
        if (tid < AD) o = oacc[tid];
    }
    if (NREP == 1 && a.fold) __syncthreads();          // (sp / op are about to be reused by the merge)
    part_publish<NREP == 1>(a, bhq0 + r, split, o, m, l, &sp[0][0], &op[0][0]);
    }   // query heads of the group
}

// ---------------------------------------------------------------------------------------------------------------------
// Matrix-core variant of the short-chunk kernel (round 5): dequantize ONCE per KV head, contract on the matrix cores.
//
// The kernel above is bound by vector instructions per QUERY head: extract + convert + multiply-add per cached element, then
// ~250 instructions of cross-lane butterflies to sum over channels / tokens -- and for grouped-query attention all of it again for
// every query head of the group.  Here the chunk's CODES go into an LDS tile once, as exact fp16 numbers (no scale, no zero
// point: one instruction per element -- `(w >> s) & 0x00030003` is the pair of fp16 subnormals code * 2^-24 as it stands), and
//     S[h, t] = sum_c (q[h, c] sc[c, g(t)]) code[c, t]  +  sum_c q[h, c] mn[c, g(t)]
//     O[h, d] = sum_t (p[h, t] sc[t, g(d)]) code[t, d]  +  sum_t p[h, t] mn[t, g(d)]
// are v_mfma_f32_32x32x16_f16 with the scaled q / p rows as the A operand (fp16 head + remainder: exact to 2^-22; up to 8 query
// heads are rows of the SAME instruction) and the integer tile as the B operand through ds_read_b64_tr_b16 (ktile.h).  The
// matrix cores do the multiply-adds AND the sums over the 128 channels / tokens; the constant terms are two short reductions.
// Outliers: the stored corrections (value - dequant, fp16) are scattered into the zeroed tile -- one LDS store per entry, no
// atomics, whatever the number of query heads -- and one more pass of MFMAs with the unscaled q / p adds them.
// Token / channel ORDER inside the tile: a code word's pairs come out as (j, j + CPW/2); they are stored as they come, so tile
// column 2k + b of a CPW-group is element k + (CPW/2) b; scores / outputs are written back through the same map.
// Requirements (else the kernel above): group 64, sparse tiles in the view or no outliers, Hq / Hkv in {1, 2, 4, 8}.
__device__ __forceinline__ int mt_logical(int p, int cpw) {      // tile column -> element index
    const int h = cpw >> 1, r = p & (cpw - 1);
    return (p & ~(cpw - 1)) | (r >> 1) | ((r & 1) ? h : 0);
}
__device__ __forceinline__ int mt_physical(int e, int cpw) {     // element index -> tile column
    const int h = cpw >> 1, r = e & (cpw - 1);
    return (e & ~(cpw - 1)) | ((r & (h - 1)) << 1) | (r >= h ? 1 : 0);
}
// the CPW codes of a word as fp16 numbers, pairs (j, j + CPW/2), into CPW consecutive halfs of a tile row (16-byte aligned).
// The fields are stored as they stand in the word: an fp16 whose only set bits are a code's is the SUBNORMAL code * 2^-24, exact,
// and the matrix cores take subnormal fp16 operands at full precision (checked by the parity tests: a flush would zero every
// score) -- one shift + one mask per pair; the accumulators are multiplied by MT_CODE_SCALE afterwards.  (The first version built
// 1024 + code by an OR and subtracted 1024: twice the instructions of a kernel that is bound by instruction issue.)
constexpr float MT_CODE_SCALE = 16777216.0f;
template <int BITS>
__device__ __forceinline__ void mt_store_codes(uint16_t* dst, uint32_t w) {
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t FM = ((1u << BITS) - 1u) * 0x00010001u;
    uint32_t z[CPW / 2];
#pragma unroll
    for (int k = 0; k < CPW / 2; k++) z[k] = (w >> (BITS * k)) & FM;
#pragma unroll
    for (int v4 = 0; v4 < CPW / 8; v4++) ((uint4*)dst)[v4] = make_uint4(z[4 * v4], z[4 * v4 + 1], z[4 * v4 + 2], z[4 * v4 + 3]);
}

// MFMA B operands of lane (x31, kg) from a tile of row pitch PITCH halfs, ALL eight contraction steps of column block I at once:
// bo[2 ks], bo[2 ks + 1] = rows 16 ks + 8 kg .. + 7 of column 32 I + x31, through the transposing LDS read (ktile.h's load_operand
// with the pitch as a parameter).  Sixteen reads are issued, then ONE wait -- per step (read, wait, MFMA) a pass of 8 steps was eight
// LDS round trips long, and a workgroup makes five to seven passes.  The wait statement names every result register as an in / out
// operand: inline asm results count as available to the compiler the moment the statement ends, and nothing else would keep an
// MFMA from being scheduled in front of the wait.
typedef short mt_short4 __attribute__((ext_vector_type(4)));
template <int PITCH>
__device__ __forceinline__ void mt_operands(const uint16_t* tile, int I, int lane, mt_short4 (&bo)[16]) {
    const int kg = lane >> 5, i = lane & 15, c0 = 32 * I + 16 * ((lane >> 4) & 1);
    const uint16_t* p = tile + (8 * kg + (i >> 2)) * PITCH + c0 + 4 * (i & 3);
    const uint32_t addr = (uint32_t)(uintptr_t)p;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        if (ks < 4) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bo[2 * ks]) : "v"(addr), "n"((16 * (ks & 3)) * PITCH * 2) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bo[2 * ks + 1]) : "v"(addr), "n"((16 * (ks & 3) + 4) * PITCH * 2) : "memory");
        } else {   // (the 16-bit offset field ends at 65535 bytes: the second half of the rows from a second base)
            const uint32_t addr2 = addr + 64 * PITCH * 2;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bo[2 * ks]) : "v"(addr2), "n"((16 * (ks & 3)) * PITCH * 2) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bo[2 * ks + 1]) : "v"(addr2), "n"((16 * (ks & 3) + 4) * PITCH * 2) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(bo[0]), "+v"(bo[1]), "+v"(bo[2]), "+v"(bo[3]), "+v"(bo[4]), "+v"(bo[5]), "+v"(bo[6]), "+v"(bo[7]), "+v"(bo[8]),
                   "+v"(bo[9]), "+v"(bo[10]), "+v"(bo[11]), "+v"(bo[12]), "+v"(bo[13]), "+v"(bo[14]), "+v"(bo[15])
                 :
                 : "memory");
}

// Phase clocks of the kernel below (profiles/r5_attn_experiments.md; tools/exp_attn_clk.py): compiled in only with -DGEAR_ATTN_CLK
// (`make -C gear_amd/csrc CXXFLAGS+=-DGEAR_ATTN_CLK`); thread 0 of the first 8192 workgroups stores s_memtime at the phase boundaries.
#ifdef GEAR_ATTN_CLK
__device__ unsigned long long attn_clk_buf[8 * 8192];
#define ATTN_CLK(k) do { if (tid == 0) { const unsigned bid_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (bid_ < 8192) attn_clk_buf[bid_ * 8 + (k)] = __builtin_readcyclecounter(); } } while (0)
#else
#define ATTN_CLK(k) do { } while (0)
#endif
// Layout of a workgroup's life (3 workgroup barriers; +3 per side when the chunk has outliers):
//   loads (all up front) | K tile: codes [channel][token] + pad columns = Pk of the chunk's two factor segments; A rows q sc (head +
//   remainder) per token group; constant term  ||  every wave, for ITS 32 token columns: scores = MFMA(q sc, codes) + const, u =
//   MFMA(q, pad) -> Qk[t] . u added in registers, outlier pass  ||  32 lanes per query head: softmax statistics, p and the A rows
//   p sc for the V side, its constant term -- while all threads put the V codes [token][channel] + pad = Qv into the tile  ||  every
//   wave, for ITS 32 channel columns: out = MFMA(p sc, codes) + const, w = MFMA(p, pad) per 64-token slab -> Pv[c] . w added in
//   registers, outlier pass, partial output stored by the lane that owns it.
template <int BITS, typename ST, int RS, int NREP>
__global__ __launch_bounds__(256) void attn_decode_partial_mfma(AttnArgs a) {
    constexpr bool R16 = RS == 16;
    constexpr int RW = R16 ? 16 : 8;
    constexpr int CPW = 32 / BITS;
    constexpr int WPR = SC / CPW;            // words per 128-element row (8 | 16)
    constexpr int RSUB = 256 / WPR;          // rows covered per pass (32 | 16)
    constexpr int WPT = AD / RSUB;           // words per thread and side (4 | 8)
    constexpr int MP = AD + 2 * RW;          // tile row pitch in halfs (144 | 160): 128 code columns + 2 RW factor columns
    __shared__ __attribute__((aligned(16))) uint16_t tile[SC * MP + 32];     // [row = contraction index][column]
    __shared__ __attribute__((aligned(16))) uint16_t aop[2][2][NREP][AD];   // A operands: [group][head / remainder][query head][k]
    __shared__ __attribute__((aligned(16))) uint16_t araw[2][NREP][AD];     // q exact; then p head / remainder
    __shared__ __attribute__((aligned(16))) float s[NREP][SC];              // scores
    __shared__ float vsm[2][2][SC];                                         // V [scale, zero point][channel group][token]
    // Pk^T q of the two segments, then Qv^T p of the two slabs.  Every wave computes ALL of it (the pad column block is one more
    // MFMA pass, cheaper than a workgroup barrier) and stores the same bits to the same place: one copy serves the four waves.
    __shared__ __attribute__((aligned(16))) float ubuf[NREP][2 * RW];
    __shared__ float cst[NREP][4];
    __shared__ float mlh[NREP][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid (heads of a batch entry, chunks, batch), the head fastest: all chunks of a head on one XCD (see the vector kernel)
    const int split = blockIdx.y;
    const int b = blockIdx.z;
    int hkv;
    int64_t bhq0;
    if (NREP == 1) {
        hkv = (int)div_sh(blockIdx.x, a.Hq / max(a.Hkv, 1), a.nrep_shift);
        bhq0 = (int64_t)b * a.Hq + blockIdx.x;
    } else {
        hkv = (int)blockIdx.x;
        bhq0 = (int64_t)b * a.Hq + (int64_t)hkv * NREP;
    }
    const int64_t bhk = (int64_t)b * a.Hkv + hkv;
    const int t0 = split * SC;
    const int Tc = a.dyn ? a.dyn[2] : a.T;
    const int tn = min(SC, Tc - t0);
    if (tn <= 0) {
#pragma unroll 1
        for (int r = 0; r < NREP; r++) {
            const int64_t pe = (bhq0 + r) * a.pslots + split;
            if (tid < AD) a.part_o[pe * AD + tid] = 0.0f;
            if (tid == 0) { a.part_ml[pe * 2] = -INFINITY; a.part_ml[pe * 2 + 1] = 0.0f; }
        }
        return;
    }
    const ST* kscale = (const ST*)a.kscale;
    const ST* kmn = (const ST*)a.kmn;
    const ST* vscale = (const ST*)a.vscale;
    const ST* vmn = (const ST*)a.vmn;
    ATTN_CLK(0);

    // ------------------------------------------------------------------ every load of the chunk (all unconditional)
    const int dq = tid & (AD - 1), half_ = tid >> 7;   // (channel | token, group | slab | side)
    const int x31 = lane & 31, kg = lane >> 5;
    const int g_w = wave >> 1;                          // the 64-element group the wave's 32 tile columns lie in
    const int el_l = mt_logical(32 * wave + x31, CPW);  // the token (K side) / channel (V side) of this lane's tile column
    auto seg_at = [&](int t) { return (a.seglen == 0 || t < a.seg0) ? 0 : 1 + (t - a.seg0) / a.seglen; };
    const int seg_s0 = seg_at(t0), seg_s1 = seg_at(min(t0 + 64, t0 + tn - 1));
    uint16_t qb[NREP];
#pragma unroll
    for (int r = 0; r < NREP; r++) qb[r] = a.q[(bhq0 + r) * AD + dq];
    const int wl = tid & (WPR - 1), rsub = tid / WPR;
    uint32_t kw[WPT], vw[WPT];
    {
        const uint32_t* kc_b = a.kcode + bhk * AD * (int64_t)a.ldk + t0 / CPW;
        const uint32_t wlc = (wl * CPW < tn) ? (uint32_t)wl : 0u;
#pragma unroll
        for (int i = 0; i < WPT; i++) kw[i] = kc_b[(uint32_t)(rsub + RSUB * i) * (uint32_t)a.ldk + wlc];
        const uint32_t* vc_b = a.vcode + (bhk * a.tcap_v + t0) * (int64_t)WPR;
#pragma unroll
        for (int i = 0; i < WPT; i++) vw[i] = vc_b[(uint32_t)min(rsub + RSUB * i, tn - 1) * WPR + wl];
    }
    // scale / zero point: K of (channel dq, 64-token group half_), V of (token dq, 64-channel group half_)
    const uint32_t gk = (uint32_t)(min(t0 + 64 * half_, t0 + tn - 1)) / 64u;
    const float ksc1 = ld_st<ST>(kscale + (bhk * AD + dq) * (int64_t)a.lsk + gk), kmn1 = ld_st<ST>(kmn + (bhk * AD + dq) * (int64_t)a.lsk + gk);
    const bool tok_ok = dq < tn;
    const int64_t vrow = (bhk * a.tcap_v + t0 + min(dq, tn - 1)) * 2 + half_;
    float vsc1 = ld_st<ST>(vscale + vrow), vmn1 = ld_st<ST>(vmn + vrow);
    const uint4* dummy16 = (const uint4*)(a.q + bhq0 * AD);
    const int64_t fseg = half_ ? (int64_t)seg_s1 : (int64_t)seg_s0;
    auto ldrow = [](const uint4* p) {
        if (RS == 4) { const uint2 t = *(const uint2*)p; return make_uint4(t.x, t.y, 0u, 0u); }
        return p[0];
    };
    // factor rows for the tile's pad columns: Pk[segment of slab half_][channel dq] and Qv[token dq]; and the rows this LANE
    // multiplies in registers: Qk[its token], Pv[both segments][its channel]
    const uint4* kpp = a.rk ? (const uint4*)(a.kP + fseg * a.kP_seg_stride + bhk * AD * RS + (uint32_t)dq * RS) : dummy16;
    const uint4* vqp = a.rv ? (const uint4*)(a.vQ + (bhk * a.tf_v + t0) * (int64_t)RS + (uint32_t)min(dq, tn - 1) * RS) : dummy16;
    const uint4* kql = a.rk ? (const uint4*)(a.kQ + (bhk * a.tf_k + t0) * (int64_t)RS + (uint32_t)min(el_l, tn - 1) * RS) : dummy16;
    const uint4* vpl0 = a.rv ? (const uint4*)(a.vP + (int64_t)seg_s0 * a.vP_seg_stride + bhk * AD * RS + (uint32_t)el_l * RS) : dummy16;
    const uint4* vpl1 = a.rv ? (const uint4*)(a.vP + (int64_t)seg_s1 * a.vP_seg_stride + bhk * AD * RS + (uint32_t)el_l * RS) : dummy16;
    const uint4 kp8 = ldrow(kpp), vq8 = ldrow(vqp), kq8 = ldrow(kql), vp08 = ldrow(vpl0), vp18 = ldrow(vpl1);
    const uint4 kp8b = R16 ? kpp[1] : kp8, vq8b = R16 ? vqp[1] : vq8, kq8b = R16 ? kql[1] : kq8, vp08b = R16 ? vpl0[1] : vp08,
                vp18b = R16 ? vpl1[1] : vp18;
    const bool has_ktile = a.ktile && a.kk > 0, has_vtile = a.vtile && a.kv > 0;
    const int64_t vb = bhk * a.nblk + 2 * split;
    const int* kcp = has_ktile ? a.kcnt + bhk * a.nck + split : (const int*)dummy16;
    const uint32_t* ktp = has_ktile ? a.ktile + (bhk * a.nck + split) * (int64_t)a.ktile_cap + tid : (const uint32_t*)dummy16;
    const int* vcp = has_vtile ? a.vcnt + vb : (const int*)dummy16;
    const uint32_t* vtp = has_vtile ? a.vtile + vb * a.vtile_cap + tid : (const uint32_t*)dummy16;
    const int kc_n = has_ktile ? kcp[0] : 0, vc_n0 = has_vtile ? vcp[0] : 0, vc_n1 = (has_vtile && tn > 64) ? vcp[1] : 0;
    uint32_t koff1 = has_ktile ? 256u : 0u, voff1 = has_vtile ? (uint32_t)a.vtile_cap : 0u;      // (as in the vector kernel above)
    asm volatile("" : "+v"(koff1), "+v"(voff1));
    const uint32_t ke0 = ktp[0], ke1 = ktp[koff1], ve0 = vtp[0], ve1 = vtp[voff1];
    if (!tok_ok) { vsc1 = 0.0f; vmn1 = 0.0f; }

    typedef union { uint4 u; half8_t h; } U8;
    auto zero_tile = [&]() {
        for (int i = tid; i < (SC * MP * 2) / 16; i += 256) ((uint4*)tile)[i] = make_uint4(0u, 0u, 0u, 0u);
    };
    // one MFMA pass over the contraction steps [ks0, ks1) of tile column block I with the rows A0 (+ A1) [NREP][AD]
    mt_short4 bo[16];                                   // the B operands of the current column block (mt_operands)
    auto mfma_pass = [&](const uint16_t* A0, const uint16_t* A1, int ks0, int ks1, float16_t acc) {
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            if (ks < ks0 || ks >= ks1) continue;
            U8 ah, al;
            ah.u = al.u = make_uint4(0u, 0u, 0u, 0u);
            if (x31 < NREP) {
                ah.u = *(const uint4*)(A0 + x31 * AD + 16 * ks + 8 * kg);
                if (A1) al.u = *(const uint4*)(A1 + x31 * AD + 16 * ks + 8 * kg);
            }
            union { half8_t h; mt_short4 s[2]; } cv;
            cv.s[0] = bo[2 * ks];
            cv.s[1] = bo[2 * ks + 1];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, cv.h, acc, 0, 0, 0);
            if (A1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, cv.h, acc, 0, 0, 0);
        }
        return acc;
    };
    auto split16 = [](float x, uint16_t& hi, uint16_t& lo) {
        hi = f2h_bits(x);
        lo = f2h_bits(x - h2f_bits(hi));
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    float16_t zero16;
#pragma unroll
    for (int q = 0; q < 16; q++) zero16[q] = 0.0f;

    // ------------------------------------------------------------------ 1. K tile, A rows, constant term
    ATTN_CLK(1);
#pragma unroll
    for (int i = 0; i < WPT; i++) mt_store_codes<BITS>(tile + (rsub + RSUB * i) * MP + CPW * wl, kw[i]);
    ATTN_CLK(2);
    if (a.rk) {
        *(uint4*)(tile + dq * MP + AD + RW * half_) = kp8;
        if (R16) *(uint4*)(tile + dq * MP + AD + RW * half_ + 8) = kp8b;
    }
    {
        float cp[8];
#pragma unroll
        for (int r = 0; r < 8; r++) cp[r] = 0.0f;
#pragma unroll
        for (int r = 0; r < NREP; r++) {
            const float qf = h2f_bits(qb[r]) * a.qscale;
            uint16_t hi, lo;
            split16(qf * ksc1, hi, lo);
            aop[half_][0][r][dq] = hi;
            aop[half_][1][r][dq] = lo;
            if (half_ == 0) araw[0][r][dq] = qb[r];
            cp[r] = qf * kmn1;
        }
        const float cr = reduce8_over_wave(cp, lane);       // over the wave's 64 channels; lanes 0, 8, .. hold head lane / 8
        if ((lane & 7) == 0 && (lane >> 3) < NREP) cst[lane >> 3][wave] = cr;
    }
    vsm[0][half_][dq] = vsc1;
    vsm[1][half_][dq] = vmn1;
    __syncthreads();
    ATTN_CLK(3);
    // ------------------------------------------------------------------ 2. scores of this wave's 32 token columns
    mt_operands<MP>(tile, wave, lane, bo);
    float16_t acc = mfma_pass(&aop[g_w][0][0][0], &aop[g_w][1][0][0], 0, 8, zero16);
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] *= MT_CODE_SCALE;
    if (a.rk) {   // u[m][r] of both segments: q (exact) against the pad columns; through the wave's LDS slot to the lanes that need it
        mt_operands<MP>(tile, 4, lane, bo);
        const float16_t au = mfma_pass(&araw[0][0][0], nullptr, 0, 8, zero16);
        if (x31 < 2 * RW) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q + 4 * kg < NREP) ubuf[q + 4 * kg][x31] = au[q] * a.qscale;
        }
        wave_sync();
    }
    if (kc_n != 0) {                                       // K outliers: stored corrections through the zeroed tile
        __syncthreads();
        zero_tile();
        __syncthreads();
        if (kc_n > 0) {
            const uint32_t* kt = a.ktile + (bhk * a.nck + split) * (int64_t)a.ktile_cap;
            for (int e = tid; e < kc_n; e += 256) {
                const uint32_t ke = e == tid ? ke0 : (e == tid + 256 ? ke1 : kt[e]);
                tile[(ke & 127u) * MP + mt_physical((int)((ke >> 7) & 127u), CPW)] = (uint16_t)(ke >> 16);
            }
        } else {   // the chunk's tile overflowed (count -1): the sorted lists, one per (channel dq, side half_)
            const int64_t chn = bhk * AD + dq;
            const uint16_t* oi = a.koidx + (chn * 2 + half_) * (int64_t)a.kk_stride;
            const uint16_t* ov = a.koval + (chn * 2 + half_) * (int64_t)a.kk_stride;
            int r0[2], r1[2];
            k_list_ranges(a, oi, Tc, t0, tn, r0, r1);
            for (int rg = 0; rg < 2; rg++)
                for (int i = r0[rg]; i < r1[rg]; i++) {
                    const int t = oi[i];
                    if (t >= t0 + tn) break;
                    if (t < t0) continue;
                    const uint32_t word = a.kcode[chn * (int64_t)a.ldk + t / CPW];
                    const float sc = ld_st<ST>(kscale + chn * (int64_t)a.lsk + t / a.group), mv = ld_st<ST>(kmn + chn * (int64_t)a.lsk + t / a.group);
                    const float deq = fmaf(sc, (float)((word >> (BITS * (t % CPW))) & ((1u << BITS) - 1u)), mv);
                    tile[dq * MP + mt_physical(t - t0, CPW)] = f2h_bits(h2f_bits(ov[i]) - deq);
                }
        }
        __syncthreads();
        mt_operands<MP>(tile, wave, lane, bo);
        const float16_t ad = mfma_pass(&araw[0][0][0], nullptr, 0, 8, zero16);
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = fmaf(a.qscale, ad[q], acc[q]);
    }
    {   // finish the scores of (head q + 4 kg, token el_l) in registers: + constant term + Qk[token] . u[head][segment of the token]
        float tq[RW];
        if (a.rk) {
            unpack8(kq8, tq);
            if (R16) unpack8(kq8b, tq + 8);
        }
        const int uo = (el_l >> 6) * RW;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int m = q + 4 * kg;
            if (m < NREP) {
                float v = acc[q] + (cst[m][2 * g_w] + cst[m][2 * g_w + 1]);
                if (a.rk) {
                    float accl = 0.0f;
#pragma unroll
                    for (int c = 0; c < RW; c++) accl = fmaf(tq[c], ubuf[m][uo + c], accl);
                    v += accl;
                }
                s[m][el_l] = v;
            }
        }
    }
    __syncthreads();
    ATTN_CLK(4);
    // ------------------------------------------------------------------ 3. softmax statistics (32 lanes per query head) and V tile
    {
        const int h = tid >> 5, j = tid & 31;
        if (h < NREP) {
            float v[4], mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = j + 32 * i;
                v[i] = t < tn ? s[h][t] : -INFINITY;
                mx = fmaxf(mx, v[i]);
            }
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
            float sum = 0.0f, c0 = 0.0f, c1 = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = j + 32 * i;
                const float pv = t < tn ? __expf(v[i] - mx) : 0.0f;
                sum += pv;
                uint16_t hi, lo;
                split16(pv, hi, lo);
                araw[0][h][t] = hi;
                araw[1][h][t] = lo;
                split16(pv * vsm[0][0][t], hi, lo);
                aop[0][0][h][t] = hi;
                aop[0][1][h][t] = lo;
                split16(pv * vsm[0][1][t], hi, lo);
                aop[1][0][h][t] = hi;
                aop[1][1][h][t] = lo;
                c0 = fmaf(pv, vsm[1][0][t], c0);
                c1 = fmaf(pv, vsm[1][1][t], c1);
            }
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) {
                sum += __shfl_xor(sum, d, 64);
                c0 += __shfl_xor(c0, d, 64);
                c1 += __shfl_xor(c1, d, 64);
            }
            if (j == 0) { mlh[h][0] = mx; mlh[h][1] = sum; cst[h][0] = c0; cst[h][1] = c1; }
        }
    }
#pragma unroll
    for (int i = 0; i < WPT; i++) mt_store_codes<BITS>(tile + (rsub + RSUB * i) * MP + CPW * wl, vw[i]);
    if (a.rv && half_ == 0) {
        *(uint4*)(tile + dq * MP + AD) = vq8;
        if (R16) *(uint4*)(tile + dq * MP + AD + 8) = vq8b;
    }
    __syncthreads();
    ATTN_CLK(5);
    // ------------------------------------------------------------------ 4. outputs of this wave's 32 channel columns
    mt_operands<MP>(tile, wave, lane, bo);
    acc = mfma_pass(&aop[g_w][0][0][0], &aop[g_w][1][0][0], 0, 8, zero16);
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] *= MT_CODE_SCALE;
    if (a.rv) {   // w[m][r] = sum over a slab's tokens of p[m][t] Qv[t][r], slab by slab (contraction steps 0-3 / 4-7)
        mt_operands<MP>(tile, 4, lane, bo);
        const float16_t w0 = mfma_pass(&araw[0][0][0], &araw[1][0][0], 0, 4, zero16);
        const float16_t w1 = mfma_pass(&araw[0][0][0], &araw[1][0][0], 4, 8, zero16);
        if (x31 < RW) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q + 4 * kg < NREP) { ubuf[q + 4 * kg][x31] = w0[q]; ubuf[q + 4 * kg][RW + x31] = w1[q]; }
        }
        wave_sync();
    }
    if (vc_n0 != 0 || vc_n1 != 0) {                        // V outliers of the chunk's two 64-token blocks
        __syncthreads();
        zero_tile();
        __syncthreads();
        for (int hb = 0; hb < 2; hb++) {
            const int n = hb ? vc_n1 : vc_n0;
            const uint32_t* vt = a.vtile + (vb + hb) * (int64_t)a.vtile_cap;
            for (int e = tid; e < n; e += 256) {
                const uint32_t ve = e == tid ? (hb ? ve1 : ve0) : vt[e];
                tile[(64 * hb + (int)(ve & 63u)) * MP + mt_physical((int)((ve >> 6) & 127u), CPW)] = (uint16_t)(ve >> 16);
            }
        }
        if (((dq < 64) ? vc_n0 : vc_n1) < 0 && tok_ok) {   // this token's block overflowed its tile: the row's sorted list, side half_
            const int c_lo = hkv * AD, c_hi = c_lo + AD;
            const int64_t orow = (int64_t)b * a.tcap_v + t0 + dq, row = bhk * a.tcap_v + t0 + dq;
            const uint16_t* oi = a.voidx + (orow * 2 + half_) * a.kv;
            const uint16_t* ov = a.voval + (orow * 2 + half_) * a.kv;
            for (int i = lower_bound_u16(oi, a.kv, c_lo); i < a.kv; i++) {
                const int col = oi[i];
                if (col >= c_hi) break;
                const int d = col - c_lo;
                const uint32_t word = a.vcode[row * WPR + d / CPW];
                const float sc = ld_st<ST>(vscale + row * 2 + d / 64), mv = ld_st<ST>(vmn + row * 2 + d / 64);
                const float deq = fmaf(sc, (float)((word >> (BITS * (d % CPW))) & ((1u << BITS) - 1u)), mv);
                tile[dq * MP + mt_physical(d, CPW)] = f2h_bits(h2f_bits(ov[i]) - deq);
            }
        }
        __syncthreads();
        mt_operands<MP>(tile, wave, lane, bo);
        acc = mfma_pass(&araw[0][0][0], &araw[1][0][0], 0, 8, acc);
    }
    {   // partial output of (head q + 4 kg, channel el_l): + constant term + Pv[segment][channel] . w[head][slab], stored by its owner
        float t0v[RW], t1v[RW];
        if (a.rv) {
            unpack8(vp08, t0v);
            unpack8(vp18, t1v);
            if (R16) { unpack8(vp08b, t0v + 8); unpack8(vp18b, t1v + 8); }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int m = q + 4 * kg;
            if (m < NREP) {
                float o = acc[q] + cst[m][g_w];
                if (a.rv) {
                    float accl = 0.0f;
#pragma unroll
                    for (int c = 0; c < RW; c++) accl = fmaf(t0v[c], ubuf[m][c], fmaf(t1v[c], ubuf[m][RW + c], accl));
                    o += accl;
                }
                a.part_o[((bhq0 + m) * a.pslots + split) * AD + el_l] = o;
            }
        }
    }
    if (tid < NREP) {
        const int64_t po = (bhq0 + tid) * a.pslots + split;
        a.part_ml[po * 2] = mlh[tid][0];
        a.part_ml[po * 2 + 1] = mlh[tid][1];
    }
    ATTN_CLK(6);
}

// merge the splits (+ the fp16 window, unless it came as one more chunk: a.pslots == a.splits + 1, W_arg == 0), normalise.
// grid (B*Hq), block 512 = 4 groups x 128 channels: the groups share the splits / window rows between them so that every
// thread's loads are one batch (the whole kernel is a latency chain).  Up to 65 partial slots (64 chunks + the window chunk).
constexpr int RS_MAX = 66;
__global__ __launch_bounds__(512) void attn_decode_reduce_kernel(AttnArgs a, const uint16_t* __restrict__ kwin,
                                                                 const uint16_t* __restrict__ vwin, int W_arg, int wcap,
                                                                 uint16_t* __restrict__ out, float* __restrict__ lse) {
    __shared__ float qs[AD];
    __shared__ float sw[128];
    __shared__ float sml[2][RS_MAX];          // per slot: chunk maximum, exponent sum
    __shared__ float coef[RS_MAX + 128];      // per slot, then per window token
    __shared__ float og[4][AD];
    __shared__ float stat[2];
    const int tid = threadIdx.x, d = tid & (AD - 1), grp = tid >> 7;
    const bool win_here = a.pslots == a.splits;        // the window is this kernel's job (rounds 1-4; the generic partial kernel)
    int W = win_here ? W_arg : 0;
    const int64_t bhq = blockIdx.x;
    const int b = (int)(bhq / a.Hq), hq = (int)(bhq % a.Hq);
    const int hkv = hq / (a.Hq / a.Hkv);
    const int64_t bhk = (int64_t)b * a.Hkv + hkv;
    if (a.dyn && win_here) W = a.dyn[3];
    const int ns = (a.dyn || a.T > 0) ? a.pslots : 0;
    // loads that depend on nothing: the split statistics, the first partial outputs of this thread (the ones the final sum
    // starts with), q and -- window here -- this thread's 16 channels of window tokens tid / 8 and 64 + tid / 8
    const int j = tid >> 3, part = tid & 7;
    uint4 k0 = {0, 0, 0, 0}, k1 = {0, 0, 0, 0}, k2 = {0, 0, 0, 0}, k3 = {0, 0, 0, 0};
    if (j < W) {
        const uint4* kr = (const uint4*)(kwin + (bhk * wcap + j) * (int64_t)AD + part * 16);
        k0 = kr[0];
        k1 = kr[1];
    }
    if (j + 64 < W) {
        const uint4* kr = (const uint4*)(kwin + (bhk * wcap + j + 64) * (int64_t)AD + part * 16);
        k2 = kr[0];
        k3 = kr[1];
    }
    float mi = -INFINITY, li = 0.0f;
    if (tid < ns) {
        mi = a.part_ml[(bhq * a.pslots + tid) * 2];
        li = a.part_ml[(bhq * a.pslots + tid) * 2 + 1];
    }
    // partial outputs of slots grp, grp + 4, ... (17 per thread at most): issued now, used after the statistics
    constexpr int NPO = (RS_MAX + 3) / 4;
    float po_r[NPO];
#pragma unroll
    for (int i = 0; i < NPO; i++) {
        const int sl = grp + 4 * i;
        po_r[i] = a.part_o[(bhq * a.pslots + min(sl, max(ns - 1, 0))) * AD + d];       // (clamped: every load unconditional)
    }
    if (tid < RS_MAX) { sml[0][tid] = mi; sml[1][tid] = li; }
    if (W > 0) {                                            // (block-uniform)
        if (tid < AD) qs[tid] = h2f_bits(a.q[bhq * AD + tid]) * a.qscale;
        __syncthreads();
        // window scores: 8 threads per token, two tokens per thread group
        float t[8], acc = 0.0f, acc2 = 0.0f;
        unpack8(k0, t);
#pragma unroll
        for (int c = 0; c < 8; c++) acc = fmaf(qs[part * 16 + c], t[c], acc);
        unpack8(k1, t);
#pragma unroll
        for (int c = 0; c < 8; c++) acc = fmaf(qs[part * 16 + 8 + c], t[c], acc);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 1, 64);
        if (j < W && part == 0) sw[j] = acc;
        if (W > 64) {                                     // (block-uniform)
            unpack8(k2, t);
#pragma unroll
            for (int c = 0; c < 8; c++) acc2 = fmaf(qs[part * 16 + c], t[c], acc2);
            unpack8(k3, t);
#pragma unroll
            for (int c = 0; c < 8; c++) acc2 = fmaf(qs[part * 16 + 8 + c], t[c], acc2);
            acc2 += __shfl_xor(acc2, 4, 64);
            acc2 += __shfl_xor(acc2, 2, 64);
            acc2 += __shfl_xor(acc2, 1, 64);
            if (j + 64 < W && part == 0) sw[j + 64] = acc2;
        }
    }
    __syncthreads();
    if (tid < 64) {   // softmax statistics over <= 65 slots and <= 128 window tokens: two slots and two tokens per lane
        const float m1 = tid < ns ? sml[0][tid] : -INFINITY, l1 = tid < ns ? sml[1][tid] : 0.0f;
        const float m2 = tid + 64 < ns ? sml[0][tid + 64] : -INFINITY, l2 = tid + 64 < ns ? sml[1][tid + 64] : 0.0f;
        const float sj = tid < W ? sw[tid] : -INFINITY;
        const float sj2 = tid + 64 < W ? sw[tid + 64] : -INFINITY;
        float M = fmaxf(fmaxf(m1, m2), fmaxf(sj, sj2));
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) M = fmaxf(M, __shfl_xor(M, x, 64));
        const float c1 = tid < ns ? __expf(m1 - M) : 0.0f;
        const float c2 = tid + 64 < ns ? __expf(m2 - M) : 0.0f;
        const float cw = tid < W ? __expf(sj - M) : 0.0f;
        const float cw2 = tid + 64 < W ? __expf(sj2 - M) : 0.0f;
        coef[tid] = c1;
        if (tid + 64 < RS_MAX) coef[tid + 64] = c2;
        coef[RS_MAX + tid] = cw;
        coef[RS_MAX + 64 + tid] = cw2;
        float L = fmaf(c1, l1, fmaf(c2, l2, cw)) + cw2;
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) L += __shfl_xor(L, x, 64);
        if (tid == 0) { stat[0] = M; stat[1] = L; }
    }
    __syncthreads();
    float o = 0.0f;
#pragma unroll
    for (int i = 0; i < NPO; i++) {
        const int sl = grp + 4 * i;
        if (sl < ns) o = fmaf(coef[sl], po_r[i], o);
    }
#pragma unroll 8
    for (int r = grp; r < W; r += 4) o = fmaf(coef[RS_MAX + r], h2f_bits(vwin[(bhk * wcap + r) * (int64_t)AD + d]), o);
    og[grp][d] = o;
    __syncthreads();
    if (tid < AD) {
        out[bhq * AD + tid] = f2h_bits(((og[0][tid] + og[1][tid]) + (og[2][tid] + og[3][tid])) / stat[1]);
        if (lse && tid == 0) lse[bhq] = stat[0] + logf(stat[1]);
    }
}

// arrival counters of a folded launch: the next of the eight regions of g_attn_arrive
uint32_t* fold_counters() {
    static uint32_t* base = nullptr;
    static unsigned next = 0;
    if (!base && hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_attn_arrive)) != hipSuccess) return nullptr;
    const unsigned r = __atomic_fetch_add(&next, 1u, __ATOMIC_RELAXED) & 7u;
    return base + (size_t)r * 65536;
}
// fold the merge into the partial kernel?  Only with option attn_fold = 1.  Measured (round 6, 7B layer at 4k, caches rotated /
// warm): batch 1 compressed 20.0 / 17.2 -> 19.7 / 18.3 us, fp16 baseline 18.4 -> 20.0 us, batch 4 43.7 -> 44.9 us, decode 335.8 ->
// 331.6 tokens/s: the drain of the write-through partials + the returning counter add + the last workgroup's round trip for the
// partials cost what the reduce launch and its boundary cost.  VERDICT r5 item 5's fold, built and not adopted.
bool fold_wanted(int64_t n_workgroups) {
    (void)n_workgroups;
    return gear_options().attn_fold > 0;
}

int plan_splits(int T, int bits, int64_t bhq, bool fast_ranks, int* tc_out, bool* small) {
    *small = false;
    if (T <= 0) { *tc_out = 64; return 1; }
    // contexts up to 8k (ranks 0 / 8): 128-token chunks (<= 64 splits for the reduce kernel) -> attn_decode_partial_small
    if (T <= 64 * SC && fast_ranks && !gear_options().attn_generic) { *tc_out = SC; *small = true; return (T + SC - 1) / SC; }
    // enough workgroups to cover the chip a few times over, chunks a multiple of 64 tokens
    int splits = 1;
    while (splits < 64 && (int64_t)splits * bhq < 1024 && T / (splits * 2) >= 128) splits *= 2;
    int tc = ((T + splits - 1) / splits + 63) / 64 * 64;
    if (tc > TC_MAX) tc = TC_MAX;
    splits = (T + tc - 1) / tc;
    *tc_out = tc;
    (void)bits;
    return splits;
}

}  // namespace
#ifdef GEAR_ATTN_CLK
extern "C" int gear_debug_attn_clk(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(attn_clk_buf), sizeof(unsigned long long) * 8 * 8192);
}
#endif

extern "C" size_t gear_attn_decode_workspace(int B, int Hq, int T, int bits) {
    // sized for the largest plan (64 splits) so that a buffer obtained for the cache capacity serves every shorter length
    (void)T; (void)bits;
    return (size_t)B * Hq * RS_MAX * (AD + 16 + 2) * sizeof(float) + 256;     // (64 chunks + the window chunk + 1)
}

namespace {

// sorted uint16 lists [n_lists][k] -> out[list][b] = first position whose value is >= b * step, b = 0 .. n_bounds - 1
__global__ __launch_bounds__(256) void outlier_chunk_index_kernel(const uint16_t* __restrict__ oidx, int64_t n_lists, int k,
                                                                  int step, int n_bounds, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lists * n_bounds) return;
    const int64_t list = i / n_bounds;
    const int bnd = (int)(i % n_bounds);
    out[i] = (uint8_t)lower_bound_u16(oidx + list * k, k, bnd * step);
}

}  // namespace

namespace {
// lists (o, i), o < n_outer, i < inner: list id = o * outer_pitch + first + i; source oidx + id * list_stride (k valid entries),
// destination out + id * out_pitch
__global__ __launch_bounds__(256) void outlier_chunk_index_ex_kernel(const uint16_t* __restrict__ oidx, int64_t inner,
                                                                     int64_t outer_pitch, int64_t first, int64_t total, int k,
                                                                     int list_stride, int step, int n_bounds,
                                                                     uint8_t* __restrict__ out, int out_pitch) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total * n_bounds) return;
    const int64_t l = i / n_bounds;
    const int bnd = (int)(i % n_bounds);
    const int64_t id = (l / inner) * outer_pitch + first + (l % inner);
    out[id * out_pitch + bnd] = (uint8_t)lower_bound_u16(oidx + id * list_stride, k, bnd * step);
}
}  // namespace

extern "C" int gear_outlier_chunk_index_ex(const void* oidx, int64_t n_outer, int64_t inner, int64_t outer_pitch, int64_t first,
                                           int k, int list_stride, int step, int n_bounds, void* out, int out_pitch,
                                           void* stream) {
    GEAR_CHECK_ARG(oidx && out && n_outer > 0 && inner > 0 && n_bounds > 0 && step > 0 && out_pitch >= n_bounds && list_stride >= k,
                   "gear_outlier_chunk_index_ex: bad arguments");
    GEAR_CHECK_ARG(k >= 0 && k <= 255, "gear_outlier_chunk_index_ex: list length %d must be in [0,255]", k);
    const int64_t total = n_outer * inner, n = total * n_bounds;
    hipLaunchKernelGGL(outlier_chunk_index_ex_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)oidx, inner, outer_pitch, first, total, k, list_stride, step, n_bounds, (uint8_t*)out,
                       out_pitch);
    GEAR_CHECK_LAUNCH("gear_outlier_chunk_index_ex");
    return 0;
}

extern "C" int gear_outlier_chunk_index(const void* oidx, int64_t n_lists, int k, int step, int n_bounds, void* out,
                                        void* stream) {
    GEAR_CHECK_ARG(oidx && out && n_lists > 0 && n_bounds > 0 && step > 0, "gear_outlier_chunk_index: bad arguments");
    GEAR_CHECK_ARG(k > 0 && k <= 255, "gear_outlier_chunk_index: list length %d must be in [1,255] (positions are stored as bytes)", k);
    GEAR_CHECK_ARG((int64_t)(n_bounds - 1) * step <= 65536, "gear_outlier_chunk_index: bounds exceed the uint16 index range");
    const int64_t n = n_lists * n_bounds;
    hipLaunchKernelGGL(outlier_chunk_index_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)oidx, n_lists, k, step, n_bounds, (uint8_t*)out);
    GEAR_CHECK_LAUNCH("gear_outlier_chunk_index");
    return 0;
}

namespace {

struct StreamExtras {
    int kk_stride, kk0, kkb, nbk_pitch;
    const void *ktile, *kcnt, *vtile, *vcnt;
    int ktile_cap, nck, vtile_cap, nblk;
};

int attn_decode_impl(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP,
                                const void* kQ, const void* koidx, const void* koval, const void* vcode,
                                const void* vscale, const void* vmn, const void* vP, const void* vQ, const void* voidx,
                                const void* voval, const void* kwin, const void* vwin, int B, int Hq, int Hkv, int D, int T,
                                int W, int ldk, int lsk, int tcap_v, int tf_k, int tf_v, int group, int bits, int mode,
                                int rk, int rv, int kk, int kv, int seg0, int seglen, int wcap, const void* dyn_state,
                                float qscale, void* out, void* lse, void* workspace, size_t workspace_bytes, void* stream,
                     const void* kochunk, const void* vochunk, const StreamExtras* ex = nullptr) {
    const int kk_stride = ex ? ex->kk_stride : 0, kk0 = ex ? ex->kk0 : 0, kkb = ex ? ex->kkb : 0, nbk_pitch = ex ? ex->nbk_pitch : 0;
    GEAR_CHECK_ARG(wcap >= W, "gear_attn_decode: window pitch %d smaller than the window %d", wcap, W);
    GEAR_CHECK_ARG(seglen == 0 || (seglen % 64 == 0 && seg0 % 64 == 0 && seg0 >= 0),
                   "gear_attn_decode: factor segments must be multiples of 64 tokens (seg0=%d seglen=%d)", seg0, seglen);
    GEAR_CHECK_ARG(D == AD, "gear_attn_decode: head_dim must be 128 (got %d)", D);
    GEAR_CHECK_ARG(bits == 2 || bits == 4, "gear_attn_decode: bits must be 2 or 4 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_attn_decode: bad mode %d", mode);
    GEAR_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "gear_attn_decode: bad head counts %d / %d", Hq, Hkv);
    GEAR_CHECK_ARG(T >= 0 && W >= 0 && W <= 128 && T + W > 0, "gear_attn_decode: need 0 <= W <= 128 and T + W > 0");
    GEAR_CHECK_ARG(T <= 64 * TC_MAX, "gear_attn_decode: at most %d compressed tokens", 64 * TC_MAX);
    GEAR_CHECK_ARG((int64_t)B * Hq <= 65535, "gear_attn_decode: too many (batch, head) pairs");
    const int cpw = 32 / bits;
    GEAR_CHECK_ARG(group % cpw == 0 && AD % group == 0 && (T == 0 || T % cpw == 0),
                   "gear_attn_decode: group %d / T %d incompatible with %d-bit packing", group, T, bits);
    GEAR_CHECK_ARG(rk >= 0 && rk <= 16 && rv >= 0 && rv <= 16, "gear_attn_decode: ranks must be <= 16");
    GEAR_CHECK_ARG(q && out && workspace, "gear_attn_decode: null pointer");
    GEAR_CHECK_ARG(T == 0 || (kcode && kscale && kmn && vcode && vscale && vmn), "gear_attn_decode: payload missing");
    GEAR_CHECK_ARG(!dyn_state || (kwin && vwin), "gear_attn_decode: a device-side state needs the window buffers");
    GEAR_CHECK_ARG(W == 0 || (kwin && vwin), "gear_attn_decode: window missing");
    GEAR_CHECK_ARG(workspace_bytes >= gear_attn_decode_workspace(B, Hq, T, bits), "gear_attn_decode: workspace too small");
    AttnArgs a;
    a.q = (const uint16_t*)q;
    a.kcode = (const uint32_t*)kcode; a.kscale = kscale; a.kmn = kmn;
    a.kP = (const uint16_t*)kP; a.kQ = (const uint16_t*)kQ; a.koidx = (const uint16_t*)koidx; a.koval = (const uint16_t*)koval;
    a.vcode = (const uint32_t*)vcode; a.vscale = vscale; a.vmn = vmn;
    a.vP = (const uint16_t*)vP; a.vQ = (const uint16_t*)vQ; a.voidx = (const uint16_t*)voidx; a.voval = (const uint16_t*)voval;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.T = T;
    a.ldk = ldk; a.lsk = lsk; a.tcap_v = tcap_v; a.tf_k = tf_k; a.tf_v = tf_v;
    a.group = group; a.rk = (kP && kQ) ? rk : 0; a.rv = (vP && vQ) ? rv : 0;
    a.kk = (koidx && koval) ? kk : 0; a.kv = (voidx && voval) ? kv : 0;
    a.kk_stride = kk_stride > 0 ? kk_stride : a.kk; a.kk0 = kk0; a.kkb = a.kk ? kkb : 0;
    GEAR_CHECK_ARG(a.kk_stride >= a.kk && (a.kkb == 0 || (seglen > 0 && kk0 >= 0)), "gear_attn_decode: bad outlier list geometry");
    a.qscale = qscale;
    a.seg0 = seg0; a.seglen = seglen;
    a.gshift = gear_is_pow2(group) ? __builtin_ctz((unsigned)group) : -1;
    a.seglen_shift = gear_is_pow2(seglen) ? __builtin_ctz((unsigned)seglen) : -1;
    a.nrep_shift = (Hkv > 0 && Hq % Hkv == 0 && gear_is_pow2(Hq / Hkv)) ? __builtin_ctz((unsigned)(Hq / Hkv)) : -1;
    a.dyn = (const int*)dyn_state;
    // chunk index of the outlier lists: only with 128-token bounds (K) / one bound per KV head (V), the small kernel's chunks
    a.kochunk = (a.kk && (a.kkb > 0 ? nbk_pitch > 0 : a.kk_stride == a.kk)) ? (const uint8_t*)kochunk : nullptr;
    a.vochunk = a.kv ? (const uint8_t*)vochunk : nullptr;
    a.nbk = nbk_pitch > 0 ? nbk_pitch : (T + SC - 1) / SC + 1;
    a.ktile = nullptr; a.kcnt = nullptr; a.vtile = nullptr; a.vcnt = nullptr; a.ktile_cap = a.nck = a.vtile_cap = a.nblk = 0;
    if (ex && ex->ktile && ex->kcnt && a.kk && ex->ktile_cap >= 512) {
        a.ktile = (const uint32_t*)ex->ktile; a.kcnt = (const int*)ex->kcnt; a.ktile_cap = ex->ktile_cap; a.nck = ex->nck;
    }
    if (ex && ex->vtile && ex->vcnt && a.kv && ex->vtile_cap >= 256) {
        a.vtile = (const uint32_t*)ex->vtile; a.vcnt = (const int*)ex->vcnt; a.vtile_cap = ex->vtile_cap; a.nblk = ex->nblk;
    }
    // With sparse tiles in the view the chunk indices are dead weight for the short-chunk kernel: it still issued their loads with
    // everything else (two bytes per thread, each from a different cache line of a [lists][bounds] table: ~16 KB of sectors per
    // workgroup for a 14 KB chunk) although only a tile that overflowed (count -1: rare) would look at them -- that case searches
    // the lists instead.  Option attn_keep_chunk_index restores the round-4 behaviour (A/B runs).
    if (!gear_options().attn_keep_chunk_index) {
        if (a.ktile) a.kochunk = nullptr;
        if (a.vtile) a.vochunk = nullptr;
    }
    a.nbv = Hkv + 1;
    a.kP_seg_stride = (int64_t)B * Hkv * AD * a.rk;
    a.vP_seg_stride = (int64_t)B * Hkv * AD * a.rv;
    bool small;
    const bool r8 = (a.rk == 0 || a.rk == 8) && (a.rv == 0 || a.rv == 8);
    const bool r16 = !r8 && (a.rk == 0 || a.rk == 16) && (a.rv == 0 || a.rv == 16);
    const bool r4 = !r8 && !r16 && (a.rk == 0 || a.rk == 4) && (a.rv == 0 || a.rv == 4);     // (BASELINE configs[1]: rank 4)
    a.splits = plan_splits(T, bits, (int64_t)B * Hq, r8 || r16 || r4, &a.tc, &small);
    // matrix-core variant of the short-chunk kernel (see attn_decode_partial_mfma): group 64, no outliers or outliers through sparse
    // tiles.  Measured (profiles/r5_attn_experiments.md): 1.3 - 2.5 x faster than one workgroup per query head for grouped-query
    // shapes once the launch has a workgroup per CU; slower for multi-head attention (one query head per KV head: the vector kernel's
    // work is not repeated there) and for a single KV head at batch 1.  attn_mfma: 0 = by that rule, 1 = whenever it applies,
    // -1 = never.
    const int n_rep = Hq / Hkv;
    const int nrep_m = (n_rep == 2 || n_rep == 4 || n_rep == 8) ? n_rep : 1;
    const int mfo = gear_options().attn_mfma;
    const bool mf_shape = small && group == 64 && mfo >= 0 && (a.kk == 0 || a.ktile) && (a.kv == 0 || a.vtile) &&
                          (mfo > 0 || (nrep_m > 1 && (int64_t)B * Hkv * a.splits >= 256));
    // the fp16 window: one more chunk of the split on the vector short-chunk kernel (its own workgroup per query head in the same
    // launch; the reduce kernel then only merges) -- with every chunk of a head on one XCD this is the faster arrangement (one layer
    // at batch 1, caches rotated: 21.0 -> 20.0 us; 7B decode 326 -> 335 tok/s graph-replayed; before the XCD mapping it measured 1 us
    // SLOWER) --, the reduce kernel's job under the matrix-core kernel and the long-context kernel.  attn_win_chunk: 0 = that rule,
    // 1 = whenever the short-chunk kernel runs (the matrix-core kernel then does not), -1 = never.
    const int wco = gear_options().attn_win_chunk;
    const bool win_chunk = small && (T > 0 || dyn_state) && kwin && vwin && (W > 0 || dyn_state) && (wco > 0 || (wco == 0 && !mf_shape));
    a.pslots = a.splits + (win_chunk ? 1 : 0);
    a.kwin = (const uint16_t*)kwin; a.vwin = (const uint16_t*)vwin; a.W = W; a.wcap = wcap;
    float* ws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    a.part_o = ws;
    a.part_w = a.part_o + (size_t)B * Hq * a.pslots * AD;
    a.part_ml = a.part_w + (size_t)B * Hq * a.pslots * 16;
    hipStream_t st = (hipStream_t)stream;
    a.fold = 0; a.arrive = nullptr; a.out_fold = (uint16_t*)out;
    bool fold_possible = false;
    if (T > 0 || a.dyn) {
        dim3 grid(a.pslots, (unsigned)(B * Hq));
        // grouped-query attention on the short-chunk kernel: option attn_gqa_group = 1: one workgroup per (chunk, KV head) serves the
        // group's 2 / 4 / 8 query heads; default: one workgroup per query head (see common.h for the measurement)
        const int gq = gear_options().attn_gqa_group;
        const bool group_on = gq > 0 || (gq < 0 && (int64_t)B * Hq * a.splits >= 32768);
        const int nrep_t = (small && group_on && (n_rep == 2 || n_rep == 4 || n_rep == 8)) ? n_rep : 1;
        const dim3 gridg((unsigned)Hkv, a.pslots, (unsigned)B), grids((unsigned)Hq, a.pslots, (unsigned)B);
#define GOS(BI, STT, RSV)                                                                                                  \
    do {                                                                                                                   \
        if (nrep_t == 8) hipLaunchKernelGGL((attn_decode_partial_small<BI, STT, RSV, 8>), gridg, dim3(256), 0, st, a);      \
        else if (nrep_t == 4) hipLaunchKernelGGL((attn_decode_partial_small<BI, STT, RSV, 4>), gridg, dim3(256), 0, st, a); \
        else if (nrep_t == 2) hipLaunchKernelGGL((attn_decode_partial_small<BI, STT, RSV, 2>), gridg, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((attn_decode_partial_small<BI, STT, RSV, 1>), grids, dim3(256), 0, st, a);                  \
    } while (0)
        const bool mf = mf_shape && a.pslots == a.splits;
        // the merge inside the partial launch: vector short-chunk kernel with one query head per workgroup, nothing left for the
        // reduce kernel but the merge (the window is a chunk of the split, or there is none), no log-sum-exp output wanted
        fold_possible = small && !mf && nrep_t == 1 && !lse && (win_chunk || (W == 0 && !dyn_state)) && a.pslots <= RS_MAX_F &&
                        fold_wanted((int64_t)B * Hq * a.pslots);
        if (fold_possible && (a.arrive = fold_counters()) != nullptr) a.fold = 1;
        const dim3 gridm((unsigned)(nrep_m > 1 ? Hkv : Hq), a.pslots, (unsigned)B);
#define GOM(BI, STT, RSV)                                                                                                  \
    do {                                                                                                                   \
        if (nrep_m == 8) hipLaunchKernelGGL((attn_decode_partial_mfma<BI, STT, RSV, 8>), gridm, dim3(256), 0, st, a);       \
        else if (nrep_m == 4) hipLaunchKernelGGL((attn_decode_partial_mfma<BI, STT, RSV, 4>), gridm, dim3(256), 0, st, a);  \
        else if (nrep_m == 2) hipLaunchKernelGGL((attn_decode_partial_mfma<BI, STT, RSV, 2>), gridm, dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((attn_decode_partial_mfma<BI, STT, RSV, 1>), gridm, dim3(256), 0, st, a);                   \
    } while (0)
#define GO(BI, STT)                                                                                             \
    do {                                                                                                        \
        if (mf && r16) GOM(BI, STT, 16);                                                                        \
        else if (mf && r4) GOM(BI, STT, 4);                                                                     \
        else if (mf) GOM(BI, STT, 8);                                                                           \
        else if (small && r16) GOS(BI, STT, 16);                                                                \
        else if (small && r4) GOS(BI, STT, 4);                                                                  \
        else if (small) GOS(BI, STT, 8);                                                                        \
        else hipLaunchKernelGGL((attn_decode_partial_kernel<BI, STT>), grid, dim3(256), 0, st, a);              \
    } while (0)
        if (mode == 0) { if (bits == 2) GO(2, uint16_t); else GO(4, uint16_t); }
        else           { if (bits == 2) GO(2, float); else GO(4, float); }
#undef GO
#undef GOS
#undef GOM
        GEAR_CHECK_LAUNCH("gear_attn_decode(partial)");
    }
    if (a.fold) return 0;                                  // (the last workgroup of every query head has merged)
    hipLaunchKernelGGL(attn_decode_reduce_kernel, dim3((unsigned)(B * Hq)), dim3(512), 0, st, a, (const uint16_t*)kwin,
                       (const uint16_t*)vwin, W, wcap, (uint16_t*)out, (float*)lse);
    GEAR_CHECK_LAUNCH("gear_attn_decode(reduce)");
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// The UNCOMPRESSED baseline: single-token attention over an fp16 K / V cache [B, Hkv, tcap, 128] (T valid tokens), the same
// flash-decoding split (128-token chunks, f16_chunk above) and the same reduce kernel.  What the reference's harness runs as
// model "None" beside gearl / KIVI (cuda_supported_gear/test.py:41-62); bench.py times it at the shapes of the compressed cache.
namespace {
template <int NREP>
__global__ __launch_bounds__(256) void attn_f16_partial_kernel(AttnArgs a, const uint16_t* __restrict__ k,
                                                               const uint16_t* __restrict__ v, int tcap) {
    __shared__ float s[SC];
    __shared__ float op[4][AD];
    __shared__ float red[8];
    __shared__ float fscr[224];
    const int split = blockIdx.x;
    int b, hkv;
    int64_t bhq0;
    if (NREP == 1) {
        bhq0 = blockIdx.y;
        b = (int)(bhq0 / a.Hq);
        hkv = (int)(bhq0 % a.Hq) / (a.Hq / a.Hkv);
    } else {
        b = (int)blockIdx.y / a.Hkv;
        hkv = (int)blockIdx.y % a.Hkv;
        bhq0 = (int64_t)b * a.Hq + (int64_t)hkv * NREP;
    }
    const int64_t bhk = (int64_t)b * a.Hkv + hkv;
    const int t0 = split * SC;
    const int64_t row0 = (bhk * tcap + t0) * (int64_t)AD;
    f16_chunk<NREP>(a, k + row0, v + row0, min(SC, a.T - t0), bhq0, split, s, op, red, fscr);
}
}  // namespace

extern "C" int gear_attn_decode_f16(const void* q, const void* k, const void* v, int B, int Hq, int Hkv, int D, int T, int tcap,
                                    float qscale, void* out, void* lse, void* workspace, size_t workspace_bytes, void* stream) {
    GEAR_CHECK_ARG(q && k && v && out && workspace, "gear_attn_decode_f16: null pointer");
    GEAR_CHECK_ARG(D == AD, "gear_attn_decode_f16: head_dim must be 128 (got %d)", D);
    GEAR_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "gear_attn_decode_f16: bad head counts %d / %d", Hq, Hkv);
    GEAR_CHECK_ARG(T > 0 && T <= tcap && T <= (RS_MAX - 1) * SC, "gear_attn_decode_f16: need 0 < T <= min(tcap, %d) (got %d)",
                   (RS_MAX - 1) * SC, T);
    GEAR_CHECK_ARG((int64_t)B * Hq <= 65535, "gear_attn_decode_f16: too many (batch, head) pairs");
    GEAR_CHECK_ARG(workspace_bytes >= gear_attn_decode_workspace(B, Hq, T, 2), "gear_attn_decode_f16: workspace too small");
    AttnArgs a{};
    a.q = (const uint16_t*)q;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.T = T;
    a.qscale = qscale;
    a.splits = a.pslots = (T + SC - 1) / SC;
    float* ws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    a.part_o = ws;
    a.part_w = a.part_o + (size_t)B * Hq * a.pslots * AD;
    a.part_ml = a.part_w + (size_t)B * Hq * a.pslots * 16;
    hipStream_t st = (hipStream_t)stream;
    const int n_rep = Hq / Hkv;
    const int gq = gear_options().attn_gqa_group;
    const bool group_on = gq > 0 || (gq < 0 && (int64_t)B * Hq * a.splits >= 32768);
    const int nrep_t = (group_on && (n_rep == 2 || n_rep == 4 || n_rep == 8)) ? n_rep : 1;
    const dim3 grid(a.splits, (unsigned)(nrep_t > 1 ? B * Hkv : B * Hq));
    a.out_fold = (uint16_t*)out;
    if (nrep_t == 1 && !lse && a.pslots <= RS_MAX_F && fold_wanted((int64_t)B * Hq * a.pslots) && (a.arrive = fold_counters()) != nullptr) a.fold = 1;
    const uint16_t *kp = (const uint16_t*)k, *vp = (const uint16_t*)v;
    if (nrep_t == 8) hipLaunchKernelGGL(attn_f16_partial_kernel<8>, grid, dim3(256), 0, st, a, kp, vp, tcap);
    else if (nrep_t == 4) hipLaunchKernelGGL(attn_f16_partial_kernel<4>, grid, dim3(256), 0, st, a, kp, vp, tcap);
    else if (nrep_t == 2) hipLaunchKernelGGL(attn_f16_partial_kernel<2>, grid, dim3(256), 0, st, a, kp, vp, tcap);
    else hipLaunchKernelGGL(attn_f16_partial_kernel<1>, grid, dim3(256), 0, st, a, kp, vp, tcap);
    GEAR_CHECK_LAUNCH("gear_attn_decode_f16(partial)");
    if (a.fold) return 0;
    hipLaunchKernelGGL(attn_decode_reduce_kernel, dim3((unsigned)(B * Hq)), dim3(512), 0, st, a, (const uint16_t*)nullptr,
                       (const uint16_t*)nullptr, 0, 0, (uint16_t*)out, (float*)lse);
    GEAR_CHECK_LAUNCH("gear_attn_decode_f16(reduce)");
    return 0;
}

extern "C" int gear_attn_decode_dyn(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP,
                                const void* kQ, const void* koidx, const void* koval, const void* vcode,
                                const void* vscale, const void* vmn, const void* vP, const void* vQ, const void* voidx,
                                const void* voval, const void* kwin, const void* vwin, int B, int Hq, int Hkv, int D, int T,
                                int W, int ldk, int lsk, int tcap_v, int tf_k, int tf_v, int group, int bits, int mode,
                                int rk, int rv, int kk, int kv, int seg0, int seglen, int wcap, const void* dyn_state,
                                float qscale, void* out, void* lse, void* workspace, size_t workspace_bytes, void* stream) {
    return attn_decode_impl(q, kcode, kscale, kmn, kP, kQ, koidx, koval, vcode, vscale, vmn, vP, vQ, voidx, voval, kwin, vwin,
                            B, Hq, Hkv, D, T, W, ldk, lsk, tcap_v, tf_k, tf_v, group, bits, mode, rk, rv, kk, kv, seg0, seglen,
                            wcap, dyn_state, qscale, out, lse, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int gear_attn_decode_idx(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP,
                                    const void* kQ, const void* koidx, const void* koval, const void* kochunk,
                                    const void* vcode, const void* vscale, const void* vmn, const void* vP, const void* vQ,
                                    const void* voidx, const void* voval, const void* vochunk, const void* kwin,
                                    const void* vwin, int B, int Hq, int Hkv, int D, int T, int W, int ldk, int lsk,
                                    int tcap_v, int tf_k, int tf_v, int group, int bits, int mode, int rk, int rv, int kk,
                                    int kv, float qscale, void* out, void* lse, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    return attn_decode_impl(q, kcode, kscale, kmn, kP, kQ, koidx, koval, vcode, vscale, vmn, vP, vQ, voidx, voval, kwin, vwin,
                            B, Hq, Hkv, D, T, W, ldk, lsk, tcap_v, tf_k, tf_v, group, bits, mode, rk, rv, kk, kv, 0, 0, W,
                            nullptr, qscale, out, lse, workspace, workspace_bytes, stream, kochunk, vochunk);
}

extern "C" int gear_attn_decode(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP,
                                const void* kQ, const void* koidx, const void* koval, const void* vcode,
                                const void* vscale, const void* vmn, const void* vP, const void* vQ, const void* voidx,
                                const void* voval, const void* kwin, const void* vwin, int B, int Hq, int Hkv, int D, int T,
                                int W, int ldk, int lsk, int tcap_v, int tf_k, int tf_v, int group, int bits, int mode,
                                int rk, int rv, int kk, int kv, float qscale, void* out, void* lse, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return gear_attn_decode_dyn(q, kcode, kscale, kmn, kP, kQ, koidx, koval, vcode, vscale, vmn, vP, vQ, voidx, voval, kwin,
                                vwin, B, Hq, Hkv, D, T, W, ldk, lsk, tcap_v, tf_k, tf_v, group, bits, mode, rk, rv, kk, kv,
                                0, 0, W, nullptr, qscale, out, lse, workspace, workspace_bytes, stream);
}

extern "C" int gear_attn_decode_seg(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP,
                                    const void* kQ, const void* koidx, const void* koval, const void* vcode,
                                    const void* vscale, const void* vmn, const void* vP, const void* vQ, const void* voidx,
                                    const void* voval, const void* kwin, const void* vwin, int B, int Hq, int Hkv, int D,
                                    int T, int W, int ldk, int lsk, int tcap_v, int tf_k, int tf_v, int group, int bits,
                                    int mode, int rk, int rv, int kk, int kv, int seg0, int seglen, int wcap, float qscale,
                                    void* out, void* lse, void* workspace, size_t workspace_bytes, void* stream) {
    return gear_attn_decode_dyn(q, kcode, kscale, kmn, kP, kQ, koidx, koval, vcode, vscale, vmn, vP, vQ, voidx, voval, kwin,
                                vwin, B, Hq, Hkv, D, T, W, ldk, lsk, tcap_v, tf_k, tf_v, group, bits, mode, rk, rv, kk, kv,
                                seg0, seglen, wcap, nullptr, qscale, out, lse, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Sparse tiles of a streaming cache: re-derive, for K chunks [c0, c1) and V blocks [b0, b1), every outlier entry from the
// sorted lists, with the correction value - dequant(code) precomputed (the codes of compressed tokens never change).
namespace {

__device__ __forceinline__ int block_excl_scan_i32(int v, int* lds, int nthreads, int* total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < (nthreads + 63) / 64; w++) {
        const int t = lds[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

template <int BITS, typename ST>
__global__ __launch_bounds__(256) void ktile_build_kernel(AttnArgs a, int T, int c0, uint32_t* __restrict__ ktile,
                                                          int* __restrict__ kcnt) {
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    __shared__ int wl[4];
    const int c = c0 + blockIdx.x, tid = threadIdx.x;
    const int64_t bhk = blockIdx.y;
    const int d = tid & 127, side = tid >> 7;
    const int t0 = c * 128, tn = min(128, T - t0);
    const int64_t list = (bhk * AD + d) * 2 + side;
    const uint16_t* oi = a.koidx + list * (int64_t)a.kk_stride;
    const uint16_t* ov = a.koval + list * (int64_t)a.kk_stride;
    int r0[2] = {0, 0}, r1[2] = {0, 0};
    if (tn > 0) k_list_ranges(a, oi, T, t0, tn, r0, r1);
    int n = 0;
    for (int rg = 0; rg < 2; rg++)
        for (int i = r0[rg]; i < r1[rg]; i++) {
            const int t = oi[i];
            if (t >= t0 + tn) break;
            if (t >= t0) n++;
        }
    int total;
    int pos = block_excl_scan_i32(n, wl, 256, &total);
    if (tid == 0) kcnt[bhk * a.nck + c] = (total <= a.ktile_cap) ? total : -1;
    if (total > a.ktile_cap) return;
    const ST* kscale = (const ST*)a.kscale;
    const ST* kmn = (const ST*)a.kmn;
    uint32_t* kt = ktile + (bhk * a.nck + c) * (int64_t)a.ktile_cap;
    const int64_t ch = bhk * AD + d;
    for (int rg = 0; rg < 2; rg++)
        for (int i = r0[rg]; i < r1[rg]; i++) {
            const int t = oi[i];
            if (t >= t0 + tn) break;
            if (t < t0) continue;
            const uint32_t word = a.kcode[ch * (int64_t)a.ldk + t / CPW];
            const float sc = ld_st<ST>(kscale + ch * (int64_t)a.lsk + t / a.group), mv = ld_st<ST>(kmn + ch * (int64_t)a.lsk + t / a.group);
            const float deq = fmaf(sc, (float)((word >> (BITS * (t % CPW))) & MASK), mv);
            kt[pos++] = (uint32_t)d | ((uint32_t)(t - t0) << 7) | ((uint32_t)f2h_bits(h2f_bits(ov[i]) - deq) << 16);
        }
}

template <int BITS, typename ST>
__global__ __launch_bounds__(128) void vtile_build_kernel(AttnArgs a, int b0, uint32_t* __restrict__ vtile, int* __restrict__ vcnt) {
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    constexpr int NWV = AD / CPW;
    __shared__ int wl[4];
    const int blk = b0 + blockIdx.x, hkv = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int tok = tid & 63, side = tid >> 6;
    const int64_t bhk = (int64_t)b * a.Hkv + hkv;
    const int t = blk * 64 + tok;
    const int c_lo = hkv * AD, c_hi = c_lo + AD;
    const uint16_t* oi = a.voidx + (((int64_t)b * a.tcap_v + t) * 2 + side) * a.kv;
    const uint16_t* ov = a.voval + (((int64_t)b * a.tcap_v + t) * 2 + side) * a.kv;
    const int i0 = lower_bound_u16(oi, a.kv, c_lo);
    int n = 0;
    for (int i = i0; i < a.kv && (int)oi[i] < c_hi; i++) n++;
    int total;
    int pos = block_excl_scan_i32(n, wl, 128, &total);
    if (tid == 0) vcnt[bhk * a.nblk + blk] = (total <= a.vtile_cap) ? total : -1;
    if (total > a.vtile_cap) return;
    const ST* vscale = (const ST*)a.vscale;
    const ST* vmn = (const ST*)a.vmn;
    const int ngv = AD / a.group;
    const int64_t row = bhk * a.tcap_v + t;
    uint32_t* vt = vtile + (bhk * a.nblk + blk) * (int64_t)a.vtile_cap;
    for (int i = i0; i < i0 + n; i++) {
        const int d = (int)oi[i] - c_lo;
        const uint32_t word = a.vcode[row * NWV + d / CPW];
        const float sc = ld_st<ST>(vscale + row * ngv + d / a.group), mv = ld_st<ST>(vmn + row * ngv + d / a.group);
        const float deq = fmaf(sc, (float)((word >> (BITS * (d % CPW))) & MASK), mv);
        vt[pos++] = (uint32_t)tok | ((uint32_t)d << 6) | ((uint32_t)f2h_bits(h2f_bits(ov[i]) - deq) << 16);
    }
}

void view_to_args(const gear_cache_view* c, AttnArgs& a, StreamExtras& ex) {
    a = AttnArgs{};
    a.kcode = (const uint32_t*)c->kcode; a.kscale = c->kscale; a.kmn = c->kmn;
    a.koidx = (const uint16_t*)c->koidx; a.koval = (const uint16_t*)c->koval;
    a.vcode = (const uint32_t*)c->vcode; a.vscale = c->vscale; a.vmn = c->vmn;
    a.voidx = (const uint16_t*)c->voidx; a.voval = (const uint16_t*)c->voval;
    a.B = c->B; a.Hkv = c->Hkv; a.ldk = c->ldk; a.lsk = c->lsk; a.tcap_v = c->tcap; a.group = c->group;
    a.kk = (c->koidx && c->koval) ? (c->kkb ? c->kk_cap : c->kk0) : 0; a.kv = (c->voidx && c->voval) ? c->kv : 0;
    a.kk_stride = c->kk_cap; a.kk0 = c->kk0; a.kkb = c->kkb; a.seg0 = c->seg0; a.seglen = c->seglen;
    a.ktile_cap = c->ktile_cap; a.nck = c->nck; a.vtile_cap = c->vtile_cap; a.nblk = c->nblk;
    ex.kk_stride = c->kk_cap; ex.kk0 = c->kk0; ex.kkb = c->kkb; ex.nbk_pitch = c->kochunk ? c->nbk_pitch : 0;
    ex.ktile = c->ktile; ex.kcnt = c->kcnt; ex.vtile = c->vtile; ex.vcnt = c->vcnt;
    ex.ktile_cap = c->ktile_cap; ex.nck = c->nck; ex.vtile_cap = c->vtile_cap; ex.nblk = c->nblk;
}

}  // namespace

extern "C" int gear_cache_tiles_build(const gear_cache_view* c, int T, int k_chunk0, int k_chunk1, int v_blk0, int v_blk1,
                                      void* stream) {
    GEAR_CHECK_ARG(c && c->D == AD && (c->bits == 2 || c->bits == 4), "gear_cache_tiles_build: bad cache view");
    GEAR_CHECK_ARG(T >= 0 && T <= c->tcap && T % 64 == 0, "gear_cache_tiles_build: T must be a multiple of 64 within the capacity");
    AttnArgs a;
    StreamExtras ex;
    view_to_args(c, a, ex);
    hipStream_t st = (hipStream_t)stream;
    const int64_t BH = (int64_t)c->B * c->Hkv;
    if (c->ktile && c->kcnt && a.kk > 0 && k_chunk1 > k_chunk0) {
        GEAR_CHECK_ARG(k_chunk0 >= 0 && k_chunk1 <= c->nck && c->ktile_cap >= 512, "gear_cache_tiles_build: bad K chunk range");
        const dim3 grid((unsigned)(k_chunk1 - k_chunk0), (unsigned)BH);
#define KT(B, STT) hipLaunchKernelGGL((ktile_build_kernel<B, STT>), grid, dim3(256), 0, st, a, T, k_chunk0, (uint32_t*)c->ktile, (int*)c->kcnt)
        if (c->mode == 0) { if (c->bits == 2) KT(2, uint16_t); else KT(4, uint16_t); }
        else { if (c->bits == 2) KT(2, float); else KT(4, float); }
#undef KT
    }
    if (c->vtile && c->vcnt && a.kv > 0 && v_blk1 > v_blk0) {
        GEAR_CHECK_ARG(v_blk0 >= 0 && v_blk1 <= c->nblk && v_blk1 * 64 <= T && c->vtile_cap >= 256, "gear_cache_tiles_build: bad V block range");
        const dim3 grid((unsigned)(v_blk1 - v_blk0), (unsigned)c->Hkv, (unsigned)c->B);
#define VT(B, STT) hipLaunchKernelGGL((vtile_build_kernel<B, STT>), grid, dim3(128), 0, st, a, v_blk0, (uint32_t*)c->vtile, (int*)c->vcnt)
        if (c->mode == 0) { if (c->bits == 2) VT(2, uint16_t); else VT(4, uint16_t); }
        else { if (c->bits == 2) VT(2, float); else VT(4, float); }
#undef VT
    }
    GEAR_CHECK_LAUNCH("gear_cache_tiles_build");
    return 0;
}

// The streaming cache's attention entry point: as gear_attn_decode_dyn over a cache view (see include/gear_hip.h).
extern "C" int gear_attn_decode_cache(const gear_cache_view* c, const void* q, int Hq, int T, int W, const void* dyn_state,
                                      float qscale, void* out, void* lse, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    GEAR_CHECK_ARG(c, "gear_attn_decode_cache: null cache view");
    AttnArgs tmp;
    StreamExtras ex;
    view_to_args(c, tmp, ex);
    // kk (the argument the kernels treat as "there are K outliers"): with growing lists the capacity, the length follows T
    int kk = 0;
    if (c->koidx && c->koval) {
        kk = c->kk_cap;
        if (c->kkb == 0) kk = c->kk0;
    }
    return attn_decode_impl(q, c->kcode, c->kscale, c->kmn, c->kP, c->kQ, c->koidx, c->koval, c->vcode, c->vscale, c->vmn, c->vP,
                            c->vQ, c->voidx, c->voval, c->kwin, c->vwin, c->B, Hq, c->Hkv, c->D, T, W, c->ldk, c->lsk, c->tcap,
                            c->tcap, c->tcap, c->group, c->bits, c->mode, c->rk, c->rv, kk, c->kv, c->seg0, c->seglen, c->wcap,
                            dyn_state, qscale, out, lse, workspace, workspace_bytes, stream, c->kochunk, c->vochunk, &ex);
}
