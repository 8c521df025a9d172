// vsel.hip -- exact outlier selection of V token rows whose heads are spread over several GPUs (head-sharded GEAR cache).
//
// The simulated path picks, per token, the k smallest and the k largest values of the row ACROSS ALL heads
// (gears_tokenQ: GenerationBench/GenerationTest/GEARLM/Simulated/compress_function.py:297-333, two torch.topk over H * D columns)
// and fills them with the row mean before quantizing.  A rank holds H / world heads of every row.  Three launches + one all-gather
// give every rank exactly the unsharded selection (rounds 1-3 selected k / world per shard; round 4 did this with ~25 torch
// launches incl. two topk, gear_amd/parallel.py exact_v_selection -- still the cross-check of the tests):
//
//   vsel_cand_kernel   per local row: the k best elements per side as GLOBAL 32-bit composites (16-bit order key << 16 | 0xFFFF -
//                      global column: ties -> lower column, the build's rule) + the exact fp64 sum of the row's local part
//                                                                                                    -> cand uint32 [rows][2k + 2]
//   (all-gather of cand over the ranks: 4 (2k + 2) bytes per row and rank -- RCCL / torch.distributed, gear_amd/parallel.py)
//   vsel_thr_kernel    per row: the k-th largest composite per side over world * k candidates, the fill value from the summed
//                      row sums                                                                        -> thr uint32 [rows][2], fill [rows]
//   compress_rows_kernel<.., EXT> (compress_rows.hip): the row compressor with the selection GIVEN -- marks the local elements at or
//                      beyond the thresholds, fills, quantizes, packs, writes the error and the (0xFFFF-padded) sorted lists.
//
// Composites are unique per row (they contain the column), so "the k-th largest" is one element and count(c >= thr) == k over all
// ranks: the shards' outlier sets partition the unsharded set, and with the same fill the concatenated shard payloads are the
// unsharded payload (tests/test_gpu_parallel.py: bit for bit in the cache's fp16-stepwise arithmetic).
#include <math.h>

#include "common.h"
#include "rowgeom.h"

namespace {

__device__ __forceinline__ int wave_count(bool pred) { return __popcll(__ballot(pred)); }     // v_cmp + s_bcnt1: counts on the scalar unit
__device__ __forceinline__ int lanes_below(unsigned long long m, int lane) { return __popcll(m & ((1ull << lane) - 1ull)); }

// k-th largest of the n composites c[0 .. n) in LDS (all different, 28 significant bits), n >= k: bisection, counts by ballot
__device__ __forceinline__ uint32_t kth_largest28(const uint32_t* c, int n, int k, int lane) {
    uint32_t lo_b = 0u, hi_b = 0x0FFFFFFFu;
    for (int it = 0; it < 28; it++) {
        const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
        int cnt = 0;
        for (int i = lane; i < n + lane; i += 64) cnt += wave_count(i < n && c[min(i, n - 1)] >= mid);     // (uniform trip count)
        if (cnt >= k) lo_b = mid; else hi_b = mid - 1u;
    }
    return lo_b;
}

// One wave per row, 4 rows per workgroup.  Per side: a threshold GUESS from the local row's mean and standard deviation lets about
// k + 5 sqrt(k) + 8 elements pass; if between k and VS_CAND of them do, the k-th largest is found among those few (two per lane, in
// registers) -- otherwise (heavy tails, constant rows, k close to the row length) among all elements.  Either way exact.
// EC > 0: the lane's EC = len / 64 elements live in registers and every loop over them is unrolled (the bisection is a chain of
// compare -> ballot -> scalar count -> next pivot: with the operands in LDS every link waited for a load; 485 us -> see DESIGN.md);
// EC == 0: any multiple of 64 (e.g. 13B's 1280-element shard rows), elements in LDS.
constexpr int VS_CAND = 128;
template <int EC>
__global__ __launch_bounds__(256) void vsel_cand_kernel(const uint16_t* __restrict__ x, RowGeom gm, int64_t n_rows, int len, int k,
                                                        int col0, float zthr, uint32_t* __restrict__ cand) {
    extern __shared__ uint32_t vs_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rows) return;
    uint32_t* comp = vs_lds + (size_t)wave * ((EC ? 0 : len) + VS_CAND);
    uint32_t* cl = comp + (EC ? 0 : len);
    const int64_t row_base = row_base_of(gm, r);
    const int E = EC ? EC : (len >> 6);                     // elements per lane (len is a multiple of 64)
    constexpr int ER = EC ? EC : 1;
    uint32_t creg[ER];
    double sum = 0.0;
    float s1 = 0.0f, s2 = 0.0f;
    {
        uint16_t hbv[ER];
        if (EC) {                                           // (all loads first)
#pragma unroll
            for (int i = 0; i < ER; i++) {
                const int j = i * 64 + lane;
                int seg, pos;
                seg_pos(gm, j, seg, pos);
                hbv[i] = x[row_base + (int64_t)seg * gm.seg_stride + pos];
            }
        }
#pragma unroll
        for (int i = 0; i < (EC ? ER : 1); i++) {
            for (int ii = i; ii < (EC ? i + 1 : E); ii++) {
                const int j = ii * 64 + lane;
                uint16_t hb;
                if (EC) hb = hbv[i];
                else {
                    int seg, pos;
                    seg_pos(gm, j, seg, pos);
                    hb = x[row_base + (int64_t)seg * gm.seg_stride + pos];
                }
                const float f = h2f_bits(hb);
                sum += (double)f;
                s1 += f;
                s2 = fmaf(f, f, s2);
                const uint32_t c = (sort_key(hb) << 12) | (uint32_t)(4095 - j);
                if (EC) creg[i] = c; else comp[j] = c;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d, 64);
        s1 += __shfl_xor(s1, d, 64);
        s2 += __shfl_xor(s2, d, 64);
    }
    const float mu = s1 / (float)len, sd = sqrtf(fmaxf(s2 / (float)len - mu * mu, 0.0f));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t* out = cand + r * (int64_t)(2 * k + 2);
    for (int side = 0; side < 2; side++) {                  // 0: the k largest values, 1: the k smallest
        // composite of the side: larger = selected first
        auto cof = [&](uint32_t c) { return side == 0 ? c : (((0xFFFFu - (c >> 12)) << 12) | (c & 0xFFFu)); };
        auto el = [&](int i) { return cof(EC ? creg[EC ? i : 0] : comp[i * 64 + lane]); };
        uint32_t T = 0u;
        bool found = false;
        if (zthr > 0.0f) {
            const uint32_t kg = sort_key(f2h_bits(side == 0 ? mu + zthr * sd : mu - zthr * sd));
            const uint32_t tg = (side == 0 ? kg : 0xFFFFu - kg) << 12;        // candidates: order key of the side >= the guess
            int n = 0;
#pragma unroll
            for (int i = 0; i < (EC ? ER : 1); i++) {
                for (int ii = i; ii < (EC ? i + 1 : E); ii++) {
                    const uint32_t c = el(ii);
                    const unsigned long long m = __ballot(c >= tg);
                    if (c >= tg) { const int sl = n + lanes_below(m, lane); if (sl < VS_CAND) cl[sl] = c; }
                    n += __popcll(m);
                }
            }
            if (n >= k && n <= VS_CAND) {                    // (wave-uniform)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint32_t c0 = lane < n ? cl[lane] : 0u, c1 = lane + 64 < n ? cl[lane + 64] : 0u;     // (0: below every pivot >= 1)
                uint32_t lo_b = 1u, hi_b = 0x0FFFFFFFu;      // the k-th largest of the n candidates (all different, all >= 1)
                for (int it = 0; it < 28; it++) {
                    const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
                    const int cnt = wave_count(c0 >= mid) + wave_count(c1 >= mid);
                    if (cnt >= k) lo_b = mid; else hi_b = mid - 1u;
                }
                T = lo_b;
                found = true;
            }
            __builtin_amdgcn_wave_barrier();                 // (cl is rewritten for the other side)
        }
        if (!found) {
            uint32_t lo_b = 0u, hi_b = 0x0FFFFFFFu;          // largest T with count(c >= T) >= k, over the whole row
            for (int it = 0; it < 28; it++) {
                const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < (EC ? ER : 1); i++)
                    for (int ii = i; ii < (EC ? i + 1 : E); ii++) cnt += wave_count(el(ii) >= mid);
                if (cnt >= k) lo_b = mid; else hi_b = mid - 1u;
            }
            T = lo_b;
        }
        int n = 0;                                           // emit the k selected (composites are unique: exactly k pass)
#pragma unroll
        for (int i = 0; i < (EC ? ER : 1); i++) {
            for (int ii = i; ii < (EC ? i + 1 : E); ii++) {
                const int j = ii * 64 + lane;
                const uint32_t c = el(ii);
                const unsigned long long m = __ballot(c >= T);
                if (c >= T) {
                    const int sl = n + lanes_below(m, lane);
                    // global composite: 16-bit order key of the side, then "lower GLOBAL column first" in 16 bits
                    if (sl < k) out[side * k + sl] = ((c >> 12) << 16) | (uint32_t)(0xFFFF - (col0 + j));
                }
                n += __popcll(m);
            }
        }
    }
    if (lane == 0) {
        const unsigned long long sb = (unsigned long long)__double_as_longlong(sum);
        out[2 * k] = (uint32_t)sb;
        out[2 * k + 1] = (uint32_t)(sb >> 32);
    }
}

// one wave per row: cand_all uint32 [world][n_rows][2k + 2] -> thr uint32 [n_rows][2] (large side, small side), fill [n_rows].
// The k-th largest of world * k composites: bisection on the 16-bit order key first; the column part only decides among equal
// keys (one element, unless the boundary value is tied).  PER > 0: the lane's PER = ceil(world k / 64) composites in registers
// (see vsel_cand_kernel); PER == 0: any count, in LDS.
template <int PER>
__global__ __launch_bounds__(256) void vsel_thr_kernel(const uint32_t* __restrict__ cand_all, int world, int64_t n_rows, int k,
                                                       double len_total, int mode, uint32_t* __restrict__ thr,
                                                       float* __restrict__ fill) {
    extern __shared__ uint32_t vt_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rows) return;
    const int n = world * k;
    uint32_t* c = vt_lds + (size_t)wave * n;
    const int64_t pitch = 2 * k + 2;
    const int per = PER ? PER : ((n + 63) >> 6);
    constexpr int PR = PER ? PER : 1;
    // element i = q * 64 + lane of the row's candidate list (rank i / k, slot i % k): position known per lane once
    int64_t off[PR];
    if (PER) {
#pragma unroll
        for (int q = 0; q < PR; q++) {
            const int i = min(q * 64 + lane, n - 1), w = i / k;
            off[q] = ((int64_t)w * n_rows + r) * pitch + (i - w * k);
        }
    }
    for (int side = 0; side < 2; side++) {
        uint32_t cr[PR];
        if (PER) {
#pragma unroll
            for (int q = 0; q < PR; q++) cr[q] = (q * 64 + lane < n) ? cand_all[off[q] + side * k] : 0u;    // (0: key 0, never >= a pivot >= 1)
        } else {
            __builtin_amdgcn_wave_barrier();
            for (int w = 0; w < world; w++) {
                const uint32_t* src = cand_all + ((int64_t)w * n_rows + r) * pitch + side * k;
                for (int j = lane; j < k; j += 64) c[w * k + j] = src[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        auto el = [&](int q) { const int i = q * 64 + lane; return PER ? cr[PER ? q : 0] : (i < n ? c[i] : 0u); };
        uint32_t lo_b = 0u, hi_b = 0xFFFFu;                  // K = the largest key with count(key >= K) >= k
        for (int it = 0; it < 16; it++) {
            const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);            // (>= 1: padding zeros never count)
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < (PER ? PR : 1); q++)
                for (int qq = q; qq < (PER ? q + 1 : per); qq++) cnt += wave_count((el(qq) >> 16) >= mid);
            if (cnt >= k) lo_b = mid; else hi_b = mid - 1u;
        }
        const uint32_t K = lo_b;
        int gt = 0, eq = 0;
        uint32_t mn_ = 0xFFFFu;
#pragma unroll
        for (int q = 0; q < (PER ? PR : 1); q++)
            for (int qq = q; qq < (PER ? q + 1 : per); qq++) {
                const bool ok = qq * 64 + lane < n;
                const uint32_t cv = el(qq), kk = cv >> 16;
                gt += wave_count(ok && kk > K);
                eq += wave_count(ok && kk == K);
                if (ok && kk == K) mn_ = min(mn_, cv & 0xFFFFu);
            }
        const int need = k - gt;                             // 1 <= need <= eq: how many of the elements with key K are selected
        // the need-th largest column part (16 bits: "lower global column first" = larger value) among the elements with key K
        uint32_t lo_c = 0u, hi_c = 0xFFFFu;
        if (need == eq) {                                    // (the usual case: no tie across the boundary) -> the smallest of them
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mn_ = min(mn_, (uint32_t)__shfl_xor((int)mn_, d, 64));
            lo_c = mn_;
        } else {
            for (int it = 0; it < 16; it++) {
                const uint32_t mid = lo_c + ((hi_c - lo_c + 1u) >> 1);
                int cnt = 0;
#pragma unroll
                for (int q = 0; q < (PER ? PR : 1); q++)
                    for (int qq = q; qq < (PER ? q + 1 : per); qq++) {
                        const uint32_t cv = el(qq);
                        cnt += wave_count(qq * 64 + lane < n && (cv >> 16) == K && (cv & 0xFFFFu) >= mid);
                    }
                if (cnt >= need) lo_c = mid; else hi_c = mid - 1u;
            }
        }
        if (lane == 0) thr[r * 2 + side] = (K << 16) | lo_c;
    }
    if (lane == 0) {
        // (a rank's sum and the total are exact in fp64 -- hence independent of the order -- as long as the row's |x| stay below
        // 2^12: fp16 values are multiples of 2^-24, 65535 of them below 2^12 sum to less than 2^28 = 52 bits of 2^-24.  Beyond that
        // the fill of a shard can differ from the unsharded kernel's in the last place: gear_hip.h states the range)
        double tot = 0.0;
        for (int w = 0; w < world; w++) {
            const uint32_t* sp = cand_all + ((int64_t)w * n_rows + r) * pitch + 2 * k;
            tot += __longlong_as_double((long long)((unsigned long long)sp[0] | ((unsigned long long)sp[1] << 32)));
        }
        const float m = (float)(tot / len_total);             // (the row kernels' and the oracle's rounding: fp64 quotient -> float)
        fill[r] = (mode == 0) ? hround(m) : m;
    }
}

double vs_inv_norm_cdf(double p) {  // Acklam's rational approximation, 0 < p < 0.5 is all that is needed here
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                               1.383577518672690e+02,  -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                               6.680131188771972e+01,  -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00,  2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    if (p < 0.02425) {
        const double q = sqrt(-2 * log(p));
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    const double q = p - 0.5, rr = q * q;
    return (((((a[0] * rr + a[1]) * rr + a[2]) * rr + a[3]) * rr + a[4]) * rr + a[5]) * q /
           (((((b[0] * rr + b[1]) * rr + b[2]) * rr + b[3]) * rr + b[4]) * rr + 1);
}

}  // namespace

// cuda_supported_gear has no counterpart (the reference is single-GPU); semantics: compress_function.py:297-333 on the full row.
extern "C" int gear_vsel_candidates(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride, int nseg,
                                    int seglen, int64_t seg_stride, int k, int col0, void* cand, void* stream) {
    GEAR_CHECK_ARG(x && cand && n_rows > 0 && n_rows < 0x7FFFFFFFLL && nseg > 0 && seglen > 0, "gear_vsel_candidates: bad arguments");
    const int64_t len = (int64_t)nseg * seglen;
    GEAR_CHECK_ARG(len % 64 == 0 && len <= 4096, "gear_vsel_candidates: local row length %lld must be a multiple of 64, <= 4096", (long long)len);
    GEAR_CHECK_ARG(k > 0 && k <= len, "gear_vsel_candidates: k = %d per side exceeds the local row length %lld (sparsity > 1 / world: "
                                       "not a configuration the exact selection supports)", k, (long long)len);
    GEAR_CHECK_ARG(col0 >= 0 && col0 + len <= 0xFFFF, "gear_vsel_candidates: global column out of range (a full row has at most 65535 elements)");
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) l++; return l; };
    RowGeom gm{rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride, gear_is_pow2(seglen) ? ilog2(seglen) : -1, 0,
               outer_stride, inner_stride, seg_stride, rows_inner};
    const int Ec = (int)(len / 64);
    const bool regs = Ec == 2 || Ec == 4 || Ec == 8 || Ec == 16 || Ec == 32;
    const size_t lds = (size_t)4 * ((regs ? 0 : len) + VS_CAND) * 4;
    // threshold guess: let about k + 5 sqrt(k) + 8 elements of a normal row pass per side (validated by the count: between k and
    // VS_CAND); no guess (z = 0: the whole-row bisection) when that is a large part of the row
    const double target = k + 5.0 * sqrt((double)k) + 8.0, pfrac = target / (double)len;
    const float zthr = (target <= 0.75 * VS_CAND && pfrac < 0.45) ? (float)(-vs_inv_norm_cdf(pfrac)) : 0.0f;
#define VC_GO(ECV)                                                                                                              \
    do {                                                                                                                       \
        auto kfn = vsel_cand_kernel<ECV>;                                                                                      \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)x, gm, \
                           n_rows, (int)len, k, col0, zthr, (uint32_t*)cand);                                                  \
    } while (0)
    if (Ec == 2) VC_GO(2); else if (Ec == 4) VC_GO(4); else if (Ec == 8) VC_GO(8); else if (Ec == 16) VC_GO(16);
    else if (Ec == 32) VC_GO(32); else VC_GO(0);
#undef VC_GO
    GEAR_CHECK_LAUNCH("gear_vsel_candidates");
    return 0;
}

extern "C" int gear_vsel_thresholds(const void* cand_all, int world, int64_t n_rows, int k, int64_t row_len_total, int mode, void* thr,
                                    void* fill, void* stream) {
    GEAR_CHECK_ARG(cand_all && thr && fill && world >= 1 && n_rows > 0 && k > 0 && row_len_total > 0, "gear_vsel_thresholds: bad arguments");
    GEAR_CHECK_ARG(row_len_total <= 0xFFFF, "gear_vsel_thresholds: a full row has at most 65535 elements");
    GEAR_CHECK_ARG((int64_t)world * k <= 4096, "gear_vsel_thresholds: world * k = %lld candidates per side exceed 4096", (long long)world * k);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_vsel_thresholds: bad mode %d", mode);
    const int per = (world * k + 63) / 64;
    const int perT = per <= 1 ? 1 : (per <= 2 ? 2 : (per <= 4 ? 4 : (per <= 8 ? 8 : 0)));
    const size_t lds = perT ? 0 : (size_t)4 * world * k * 4;
#define VT_GO(PV)                                                                                                               \
    do {                                                                                                                       \
        auto kfn = vsel_thr_kernel<PV>;                                                                                        \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), lds, (hipStream_t)stream, (const uint32_t*)cand_all, \
                           world, n_rows, k, (double)row_len_total, mode, (uint32_t*)thr, (float*)fill);                       \
    } while (0)
    if (perT == 1) VT_GO(1); else if (perT == 2) VT_GO(2); else if (perT == 4) VT_GO(4); else if (perT == 8) VT_GO(8); else VT_GO(0);
#undef VT_GO
    GEAR_CHECK_LAUNCH("gear_vsel_thresholds");
    return 0;
}
