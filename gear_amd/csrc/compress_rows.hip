// compress_rows.hip -- one pass per row: outlier top-k select -> mean fill -> group quantize -> bit-pack -> error.
//
// A "row" is the unit over which the reference selects outliers
// (GenerationBench/.../Simulated/compress_function.py:261-333):
//   * V / token quantization : row = one token across ALL heads  (len = H*D, H segments of D contiguous fp16)
//   * K / channel quantization: row = one channel across all tokens (len = T, contiguous in the K^T layout
//                               [B,H,D,T] that the attention hook passes, modeling_llamagear.py:268)
// Quantization groups are `group` consecutive elements inside a segment.  One workgroup owns one row, each lane 16
// consecutive elements in registers; ties are broken by LOWER INDEX FIRST (the oracle's rule); the survivors are quantized
// exactly like quant_pack.hip.  Two kernels live here:
//   compress_rows_fp32_kernel  fp32 arithmetic (mode 1, the simulated path): Gaussian-threshold candidates compacted in
//                              index order + 17-step key bisection (two-level radix select as the exact fallback), dense
//                              part on packed fp16 / packed fp32 -- the kernel the bench and the roofline are about
//   compress_rows_kernel       first generation: fp16-stepwise arithmetic (mode 0, the fused path's block compress) and,
//                              with the option rows_v1, the fp32 mode as a cross-check
//
// Outputs per row: packed codes, scale, mn, optional error (0 at outlier positions), and the sparse part
// (column index within the row as uint16 + original fp16 value, sorted by index; first k = smallest side).
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "rowgeom.h"
#include "dense16.h"

namespace {

// inclusive scan over the block of a 64-bit packed counter (fields never overflow into each other)
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long* wave_tot,
                                                              unsigned long long* total_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    __syncthreads();  // wave_tot reuse
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int w = 0; w < nw; w++) {
        unsigned long long t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    if (total_out) *total_out = tot;
    return base + inc - v;
}

// wave 0 finds, in a 256-bin histogram, the bin where the running count from the top (FROM_TOP) or from the
// bottom crosses `need`; returns (bin, how many to take from that bin) through LDS.
template <bool FROM_TOP>
__device__ __forceinline__ void find_crossing(const uint32_t* hist, int need, int* out_bin, int* out_take) {
    // called by one full wave (64 lanes); lane l owns bins 4l .. 4l+3
    const int lane = threadIdx.x & 63;
    uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    uint32_t s = h0 + h1 + h2 + h3;
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    uint32_t total = __shfl(inc, 63, 64);
    uint32_t below = inc - s;  // elements in bins < 4*lane
    uint32_t hh[4] = {h0, h1, h2, h3};
    if (FROM_TOP) {
        // count of elements in bins > b
        uint32_t above = total - below - s;  // bins > 4*lane+3
        for (int j = 3; j >= 0; j--) {
            if (above < (uint32_t)need && above + hh[j] >= (uint32_t)need) {
                *out_bin = 4 * lane + j;
                *out_take = need - (int)above;
            }
            above += hh[j];
        }
    } else {
        uint32_t bl = below;
        for (int j = 0; j < 4; j++) {
            if (bl < (uint32_t)need && bl + hh[j] >= (uint32_t)need) {
                *out_bin = 4 * lane + j;
                *out_take = need - (int)bl;
            }
            bl += hh[j];
        }
    }
}


// ---- candidate-based exact top-k (fast path) ------------------------------------------------------------------
// Every one of the k largest elements of a row is >= tau_hi whenever at least k elements are >= tau_hi.  Take, in each
// wave, the ceil(k/nw)-th largest of the 64 per-lane maxima (bitonic sort in registers); tau_hi = the minimum of those
// over the waves: at least k distinct elements (lane maxima) are >= tau_hi.  Typically only ~1-2 % of the row passes
// the threshold; the survivors are ranked exactly (value, then lower index first) in LDS.  Same for the small side.
constexpr int CAND_CAP = 640;

template <bool DESC>
__device__ __forceinline__ uint32_t wave_bitonic_sort(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            uint32_t o = __shfl_xor(v, j, 64);
            bool up = ((lane & k2) == 0);
            bool lower = ((lane & j) == 0);
            uint32_t mx = max(v, o), mn = min(v, o);
            bool take_first = (lower == up);            // "first" = max for a descending sort, min for ascending
            v = DESC ? (take_first ? mx : mn) : (take_first ? mn : mx);
        }
    }
    return v;  // lane i holds the i-th element of the sorted order
}

// EXT: the selection is GIVEN (head-sharded V rows, vsel.hip): per row two 32-bit composite thresholds (large side, small side) and
// the fill value; an element is an outlier of a side when its GLOBAL composite (order key of the side << 16 | 0xFFFF - (col0 + j))
// is at or beyond the threshold.  A rank then holds between 0 and k outliers of a row per side: unused list slots carry the index
// 0xFFFF (beyond every head bound) and the value 0.
struct ExtSel {
    const uint32_t* thr;             // [n_rows][2]
    const float* fill;               // [n_rows]
    int col0;                        // global column of this rank's first element
};

template <int BITS, int MODE, typename ST, bool EXT = false>
__global__ void compress_rows_kernel(const uint16_t* __restrict__ x, RowGeom gm, int len, int group, int k, float zthr,
                                     uint32_t* __restrict__ code, ST* __restrict__ scale, ST* __restrict__ mn,
                                     uint16_t* __restrict__ err, uint16_t* __restrict__ oidx,
                                     uint16_t* __restrict__ oval, float* __restrict__ omean, ExtSel ext = ExtSel{nullptr, nullptr, 0}) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int WPL = BITS / 2;
    constexpr int CPW = 32 / BITS;
    __shared__ uint32_t hist[3][256];
    __shared__ unsigned long long wave_tot[16];
    __shared__ double wave_sum[16];
    __shared__ int sh[8];  // 0 bin_hi, 1 need_hi, 2 bin_lo, 3 need_lo, 4 thr_hi, 5 take_hi, 6 thr_lo, 7 take_lo
    __shared__ uint32_t cand[2][CAND_CAP];   // composite (key, index) of the hi / lo candidates
    __shared__ uint32_t omask[2][512];       // per-element outlier bitmaps (hi / lo), row length <= 16384
    __shared__ uint32_t wave_thr[2][16];
    __shared__ uint32_t ncand[2];
    extern __shared__ uint32_t rawlds[];     // [blockDim.x][8]: the lane's 16 raw fp16 values (tier-0 path only)

    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    const int j0 = tid * 16;  // first element of this lane inside the row
    const bool active = j0 < len;
    int seg = 0, pos = 0;
    if (active) seg_pos(gm, j0, seg, pos);
    const int64_t row_base = row_base_of(gm, r);
    const int64_t off = row_base + (int64_t)seg * gm.seg_stride + pos;  // element offset of this lane's 16 values
    const int64_t ooff = row_base_out(gm, r) + (int64_t)seg * gm.o_seg_stride + pos;   // ... in the payload tensors

    uint4 ra = make_uint4(0, 0, 0, 0), rb = ra;
    if (active) {
        const uint4* p = (const uint4*)(x + off);
        ra = p[0];
        rb = p[1];
    }
    uint32_t hb[16];  // raw fp16 bits
    hb[0] = ra.x & 0xFFFFu; hb[1] = ra.x >> 16; hb[2] = ra.y & 0xFFFFu; hb[3] = ra.y >> 16;
    hb[4] = ra.z & 0xFFFFu; hb[5] = ra.z >> 16; hb[6] = ra.w & 0xFFFFu; hb[7] = ra.w >> 16;
    hb[8] = rb.x & 0xFFFFu; hb[9] = rb.x >> 16; hb[10] = rb.y & 0xFFFFu; hb[11] = rb.y >> 16;
    hb[12] = rb.z & 0xFFFFu; hb[13] = rb.z >> 16; hb[14] = rb.w & 0xFFFFu; hb[15] = rb.w >> 16;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = h2f_bits((uint16_t)hb[j]);

    uint32_t flag_lo = 0, flag_hi = 0;  // bit j: element j of this lane is an outlier (small / large side)
    if (EXT && k > 0) {
        const uint32_t thr_l = ext.thr[r * 2], thr_s = ext.thr[r * 2 + 1];
        uint16_t* oi = oidx + lrow_of(gm, r) * (int64_t)(2 * k);
        uint16_t* ov = oval + lrow_of(gm, r) * (int64_t)(2 * k);
        for (int i = tid; i < 2 * k; i += blockDim.x) { oi[i] = 0xFFFFu; ov[i] = 0u; }      // (slots this rank does not fill)
        if (active) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t kx = sort_key(hb[j]), inv = (uint32_t)(0xFFFF - (ext.col0 + j0 + j));
                if (((kx << 16) | inv) >= thr_l) flag_hi |= 1u << j;
                if ((((0xFFFFu - kx) << 16) | inv) >= thr_s) flag_lo |= 1u << j;
            }
        }
        // sorted lists: exclusive scan of the per-lane counts (the barriers inside also order the padding stores above against
        // the entries below)
        unsigned long long cnt = (unsigned long long)__popc(flag_hi) | ((unsigned long long)__popc(flag_lo) << 32);
        unsigned long long slot = block_excl_scan(cnt, wave_tot, nullptr);
        int slot_hi = (int)(slot & 0xFFFFFFFFull), slot_lo = (int)(slot >> 32);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (flag_lo & (1u << j)) {
                if (slot_lo < k) { oi[slot_lo] = (uint16_t)(j0 + j); ov[slot_lo] = (uint16_t)hb[j]; }
                slot_lo++;
            }
            if (flag_hi & (1u << j)) {
                if (slot_hi < k) { oi[k + slot_hi] = (uint16_t)(j0 + j); ov[k + slot_hi] = (uint16_t)hb[j]; }
                slot_hi++;
            }
        }
        const float mean = ext.fill[r];
        const float fill = (MODE == 0) ? hround(mean) : mean;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((flag_lo | flag_hi) & (1u << j)) v[j] = fill;
    }
    if (!EXT && k > 0) {
        // ---------------- row mean (of the ORIGINAL row, compress_function.py:276 / :312)
        // summed in fp64: fp16 values add exactly there, so the mean is the correctly rounded one whatever the order -- the oracle's,
        // the block compressor's and the short-row kernel's, and the one a head-sharded job reconstructs from per-rank sums
        // (cache.py: exact cross-shard V selection; an fp32 tree sum moved the fp16 fill value about once in 500 rows)
        double s = 0.0;
        if (active) {
#pragma unroll
            for (int j = 0; j < 16; j++) s += (double)v[j];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) wave_sum[wave] = s;
        uint32_t key[16];
#pragma unroll
        for (int j = 0; j < 16; j++) key[j] = sort_key(hb[j]);
        // ---------------- fast path (tier 0): Gaussian-guess thresholds, validated by the survivor counts
        // tau = mean +- z*sigma of THIS row (z from the host: about 2.2 k / len of a normal row passes).  Any threshold
        // is valid as long as at least k elements pass it on each side; if not (or if more than 128 pass) the row falls
        // back to the histogram radix select below.  Survivors: exact k-th composite by ballot bisection on ONE wave per
        // side (popcounts run on the scalar unit), then one bitonic sort of the <= 64 selected by index.
        bool use_hist = (k > 64) || (zthr <= 0.0f);
        bool payload_done = false;
        if (!use_hist) {
            float s2 = 0.0f;
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) s2 = fmaf(v[j], v[j], s2);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s2 += __shfl_xor(s2, d, 64);
            if (lane == 0) wave_thr[0][wave] = __float_as_uint(s2);
            if (tid < 2) ncand[tid] = 0u;
            for (int i = tid; i < 2 * 512; i += blockDim.x) (&omask[0][0])[i] = 0u;
            __syncthreads();
            float tot1 = 0.0f, tot2 = 0.0f;
            for (int w = 0; w < nw; w++) { tot1 += wave_sum[w]; tot2 += __uint_as_float(wave_thr[0][w]); }
            const float mu = tot1 / (float)len;
            const float sd = sqrtf(fmaxf(tot2 / (float)len - mu * mu, 0.0f));
            const float thi = mu + zthr * sd, tlo = mu - zthr * sd;
            // survivors: packed fp16 subtract against the threshold, sign bits gathered into one mask per side
            // (3 VALU per two elements), then each lane emits its own few hits ((raw bits << 16) | index) -- about
            // 2 % of the elements survive, so the emission loop runs ~3 times per wave
            typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
            const uint32_t rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
            const uint32_t th = f2h_bits(thi), tl = f2h_bits(tlo);
            const half2_t thi2 = __builtin_bit_cast(half2_t, th | (th << 16)), tlo2 = __builtin_bit_cast(half2_t, tl | (tl << 16));
            uint32_t sg_hi = 0u, sg_lo = 0u;   // sign(x - thi): set where x < thi ; sign(tlo - x): set where x > tlo
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const half2_t xv = __builtin_bit_cast(half2_t, rw[w]);
                const uint32_t dh = __builtin_bit_cast(uint32_t, (half2_t)(xv - thi2));
                const uint32_t dl = __builtin_bit_cast(uint32_t, (half2_t)(tlo2 - xv));
                sg_hi |= (dh & 0x80008000u) >> w;
                sg_lo |= (dl & 0x80008000u) >> w;
                rawlds[tid * 8 + w] = rw[w];
            }
            uint32_t mh = active ? (~sg_hi & 0xFF00FF00u) : 0u, ml = active ? (~sg_lo & 0xFF00FF00u) : 0u;
            // bit 15-w <-> element 2w, bit 31-w <-> element 2w+1
            while (mh) {
                const int b = 31 - __clz(mh);
                mh &= ~(1u << b);
                const int j = b >= 24 ? 2 * (31 - b) + 1 : 2 * (15 - b);
                const uint32_t bits = (rawlds[tid * 8 + (j >> 1)] >> (16 * (j & 1))) & 0xFFFFu;
                uint32_t sl = atomicAdd(&ncand[0], 1u);
                if (sl < 128u) cand[0][sl] = (bits << 16) | (uint32_t)(j0 + j);
            }
            while (ml) {
                const int b = 31 - __clz(ml);
                ml &= ~(1u << b);
                const int j = b >= 24 ? 2 * (31 - b) + 1 : 2 * (15 - b);
                const uint32_t bits = (rawlds[tid * 8 + (j >> 1)] >> (16 * (j & 1))) & 0xFFFFu;
                uint32_t sl = atomicAdd(&ncand[1], 1u);
                if (sl < 128u) cand[1][sl] = (bits << 16) | (uint32_t)(j0 + j);
            }
            __syncthreads();
            const uint32_t nh = ncand[0], nl = ncand[1];
            use_hist = (nh < (uint32_t)k) || (nl < (uint32_t)k) || (nh > 128u) || (nl > 128u);   // block-uniform
            if (!use_hist) {
                // wave 0 finishes the large side, wave 1 (or wave 0 again) the small side
                for (int side = 0; side < 2; side++) {
                    if (wave != ((nw > 1) ? side : 0)) continue;
                    const uint32_t n = side == 0 ? nh : nl;
                    // large side: select the k LARGEST composites; small side: the k SMALLEST -> flip to "largest of ~c"
                    const bool v0 = (uint32_t)lane < n, v1 = (uint32_t)(lane + 64) < n;
                    const uint32_t c0 = v0 ? cand[side][lane] : 0u, c1 = v1 ? cand[side][lane + 64] : 0u;
                    const uint32_t k0 = sort_key(c0 >> 16), k1 = sort_key(c1 >> 16), i0 = c0 & 0xFFFFu, i1 = c1 & 0xFFFFu;
                    // order value, larger = selected first: large side (key desc, index asc); small side (key asc, index asc)
                    uint32_t o0 = side == 0 ? ((k0 << 16) | (0xFFFFu - i0)) : ~((k0 << 16) | i0);
                    uint32_t o1 = side == 0 ? ((k1 << 16) | (0xFFFFu - i1)) : ~((k1 << 16) | i1);
                    if (!v0) o0 = 0u;
                    if (!v1) o1 = 0u;
                    uint32_t lo_b = 0u, hi_b = 0xFFFFFFFFu;   // largest T with count(o >= T) >= k
                    for (int it = 0; it < 32; it++) {
                        const uint32_t mid = lo_b + ((hi_b - lo_b) >> 1) + ((hi_b - lo_b) & 1u);
                        const int cnt = __popcll(__ballot(o0 >= mid)) + __popcll(__ballot(o1 >= mid));
                        if (cnt >= k) lo_b = mid; else hi_b = mid - 1u;
                    }
                    const bool s0 = v0 && o0 >= lo_b, s1 = v1 && o1 >= lo_b;
                    const unsigned long long b0 = __ballot(s0), b1 = __ballot(s1);
                    const int p0 = __popcll(b0 & ((1ull << lane) - 1ull));
                    const int p1 = __popcll(b0) + __popcll(b1 & ((1ull << lane) - 1ull));
                    uint32_t* selb = &cand[side][128];     // scratch behind the candidates (CAND_CAP >= 192)
                    // sort value: (index << 16) | raw fp16 bits  -> ascending by index
                    if (s0) selb[p0] = (i0 << 16) | (c0 >> 16);
                    if (s1) selb[p1] = (i1 << 16) | (c1 >> 16);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    uint32_t sv = lane < k ? selb[lane] : 0xFFFFFFFFu;
                    sv = wave_bitonic_sort<false>(sv);
                    if (lane < k) {
                        const uint32_t idx = sv >> 16;
                        uint16_t* oi = oidx + lrow_of(gm, r) * (int64_t)(2 * k) + (side == 0 ? k : 0);
                        uint16_t* ov = oval + lrow_of(gm, r) * (int64_t)(2 * k) + (side == 0 ? k : 0);
                        oi[lane] = (uint16_t)idx;
                        ov[lane] = (uint16_t)(sv & 0xFFFFu);
                        atomicOr(&omask[side][idx >> 5], 1u << (idx & 31));
                    }
                }
                __syncthreads();
                if (active) {
                    flag_hi = (omask[0][j0 >> 5] >> (j0 & 31)) & 0xFFFFu;
                    flag_lo = (omask[1][j0 >> 5] >> (j0 & 31)) & 0xFFFFu;
                }
                payload_done = true;
            }
        }
        if (use_hist) {
            // ---------------- level-1 histogram on the high key byte
            for (int i = tid; i < 3 * 256; i += blockDim.x) (&hist[0][0])[i] = 0u;
            __syncthreads();
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) atomicAdd(&hist[0][key[j] >> 8], 1u);
            }
            __syncthreads();
            if (wave == 0) {
                find_crossing<true>(hist[0], k, &sh[0], &sh[1]);
                find_crossing<false>(hist[0], k, &sh[2], &sh[3]);
            }
            __syncthreads();
            const uint32_t bin_hi = (uint32_t)sh[0], bin_lo = (uint32_t)sh[2];
            const int need_hi = sh[1], need_lo = sh[3];
            // ---------------- level-2 histograms on the low key byte inside the two boundary bins
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if ((key[j] >> 8) == bin_hi) atomicAdd(&hist[1][key[j] & 255u], 1u);
                    if ((key[j] >> 8) == bin_lo) atomicAdd(&hist[2][key[j] & 255u], 1u);
                }
            }
            __syncthreads();
            if (wave == 0) {
                find_crossing<true>(hist[1], need_hi, &sh[4], &sh[5]);
                find_crossing<false>(hist[2], need_lo, &sh[6], &sh[7]);
            }
            __syncthreads();
            const uint32_t thr_hi = (bin_hi << 8) | (uint32_t)sh[4];
            const uint32_t thr_lo = (bin_lo << 8) | (uint32_t)sh[6];
            const int take_hi = sh[5], take_lo = sh[7];
            // ---------------- tie ranks (lower index first): exclusive scan of per-lane equal counts
            unsigned long long eq = 0;
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    eq += (key[j] == thr_hi) ? 1ull : 0ull;
                    eq += (key[j] == thr_lo) ? (1ull << 32) : 0ull;
                }
            }
            unsigned long long ex = block_excl_scan(eq, wave_tot, nullptr);
            int rank_hi = (int)(ex & 0xFFFFFFFFull), rank_lo = (int)(ex >> 32);
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (key[j] < thr_lo) flag_lo |= 1u << j;
                    else if (key[j] == thr_lo) { if (rank_lo < take_lo) flag_lo |= 1u << j; rank_lo++; }
                    if (key[j] > thr_hi) flag_hi |= 1u << j;
                    else if (key[j] == thr_hi) { if (rank_hi < take_hi) flag_hi |= 1u << j; rank_hi++; }
                }
            }
        }
        // every path has crossed a barrier since wave_sum was written
        double tot = 0.0;
        for (int w = 0; w < nw; w++) tot += wave_sum[w];
        const float mean = (float)(tot / (double)len);
        // ---------------- output slots (sorted by index) and the sparse payload
        if (!payload_done) {
            unsigned long long cnt = (unsigned long long)__popc(flag_hi) | ((unsigned long long)__popc(flag_lo) << 32);
            unsigned long long slot = block_excl_scan(cnt, wave_tot, nullptr);
            int slot_hi = (int)(slot & 0xFFFFFFFFull), slot_lo = (int)(slot >> 32);
            uint16_t* oi = oidx + lrow_of(gm, r) * (int64_t)(2 * k);
            uint16_t* ov = oval + lrow_of(gm, r) * (int64_t)(2 * k);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (flag_lo & (1u << j)) {
                    if (slot_lo < k) { oi[slot_lo] = (uint16_t)(j0 + j); ov[slot_lo] = (uint16_t)hb[j]; }
                    slot_lo++;
                }
                if (flag_hi & (1u << j)) {
                    if (slot_hi < k) { oi[k + slot_hi] = (uint16_t)(j0 + j); ov[k + slot_hi] = (uint16_t)hb[j]; }
                    slot_hi++;
                }
            }
        }
        if (tid == 0 && omean) omean[r] = mean;
        // ---------------- fill (compress_function.py:279-283 / :315-319)
        const float fill = (MODE == 0) ? hround(mean) : mean;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((flag_lo | flag_hi) & (1u << j)) v[j] = fill;
    }

    // ---------------- group quantization (identical arithmetic to quant_pack.hip)
    const int lanes_per_group = group / 16;
    float lo = v[0], hi = v[0];
#pragma unroll
    for (int j = 1; j < 16; j++) {
        lo = fminf(lo, v[j]);
        hi = fmaxf(hi, v[j]);
    }
    for (int m = 1; m < lanes_per_group; m <<= 1) {
        lo = fminf(lo, __shfl_xor(lo, m, 64));
        hi = fmaxf(hi, __shfl_xor(hi, m, 64));
    }
    if (!active) return;
    QuantParams<MODE> qp = make_qparams<MODE>(lo, hi, LEVELS);
    const float inv = (qp.scale != 0.0f) ? div_rn(1.0f, qp.scale) : 0.0f;
    uint32_t words[WPL];
#pragma unroll
    for (int w = 0; w < WPL; w++) words[w] = 0u;
    float e[16];
    const uint32_t outl = flag_lo | flag_hi;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        int q = quant_fast<BITS, MODE>(v[j], qp.mn, qp.scale, inv, LEVELS);
        words[j / CPW] |= (uint32_t)q << (BITS * (j % CPW));
        float d = (MODE == 0) ? dequant_one<0>(q, qp.scale, qp.mn) : hround(dequant_one<1>(q, qp.scale, qp.mn));
        e[j] = (outl & (1u << j)) ? 0.0f : (v[j] - d);
    }
    uint32_t* cp = code + ooff / CPW;
#pragma unroll
    for (int w = 0; w < WPL; w++) cp[w] = words[w];
    if ((tid & (lanes_per_group - 1)) == 0) {
        st_st<ST>(scale + (ooff >> gm.group_shift), qp.scale);
        st_st<ST>(mn + (ooff >> gm.group_shift), qp.mn);
    }
    if (err) {
        uint4* ep = (uint4*)(err + off);
        ep[0] = pack8(e);
        ep[1] = pack8(e + 8);
    }
}


// =====================================================================================================
// fp32-arithmetic rows (MODE 1), second generation of the workgroup kernel above.  Selection: tier-0 Gaussian thresholds
// from the row's sum / sum of squares -> candidates compacted IN INDEX ORDER into one LDS region per wave (DPP prefix sum of
// the lane counts) -> one wave per side finds the threshold value by 17 rounds of ballot bisection on the 16-bit order key,
// ties at the threshold resolve "lower index first" by position, outputs come out sorted by index without a sort; the
// two-level histogram radix select stays as the exact fallback.  The dense part was rebuilt around what the PMC counters
// showed (VALU-issue-bound, ~1000 VALU instructions per wave and row of which ~100 were per-element compare/select chains
// and ~32 branches):
//   * the lane's 16 elements stay PACKED (8 words of 2 x fp16) wherever the arithmetic is exact in fp16: row sum and
//     sum of squares by v_dot2_f32_f16, group min/max by v_pk_min/max_f16 with the outlier halves masked to +-inf,
//     error = x - dequant by v_pk_add_f16 (a single rounding of an exact difference, identical to rounding the fp32
//     difference), outlier halves of the error cleared by one v_bfi per word;
//   * the per-element outlier handling is gone: the selecting wave marks each outlier's half-word in LDS (the raw-copy
//     region, free after the candidate emission), a lane reads its 8 mask words back with two ds_read_b128; the code of
//     a filled position (the same for every outlier of a group: quant(mean)) is patched into the packed word with a
//     2-bit-per-element mask spread from the lane's 16 flag bits;
//   * quantization is branch-free: reciprocal multiply, one OR-ed "within 1e-5 of a rounding tie" flag per lane, and
//     only a lane that raises it redoes its 16 divisions exactly; codes are packed by a Horner chain in fp32 (exact
//     below 2^16) instead of 16 convert + shift-or pairs; the clamp is one v_med3.
// Bit-exact against the same oracle as the first kernel (tests/test_gpu_compress.py runs both).
// =====================================================================================================

template <int BITS, typename ST>
__global__ void compress_rows_fp32_kernel(const uint16_t* __restrict__ x, RowGeom gm, int len, int group, int k, float zthr,
                                          float rlen,
                                          uint32_t* __restrict__ code, ST* __restrict__ scale, ST* __restrict__ mn,
                                          uint16_t* __restrict__ err, uint16_t* __restrict__ oidx,
                                          uint16_t* __restrict__ oval, float* __restrict__ omean) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int WPL = BITS / 2;
    constexpr int CPW = 32 / BITS;
    constexpr int HC = 16 / BITS;            // codes per 16-bit half of a packed word
    __shared__ unsigned long long wave_tot[16];
    __shared__ float wave_sum[16];
    __shared__ int sh[8];
    __shared__ __attribute__((aligned(16))) uint32_t wave_thr[2][16];
    // Dynamic LDS, sized by the host for the row length (a 512-element row of an 8-way head shard needs 3.2 KB, not the
    // 20 KB of the 16384-element worst case: LDS is what bounds the number of resident one-wave workgroups):
    //   [ rawlds: blockDim x 8 words (raw copy during the emission, then the half-word outlier marks)
    //   | cand: 2 sides x nw regions x wcap composites ]      -- the radix-select histograms (3 x 256) alias this part
    //   [ omask: 2 x ceil(len / 32) words ]
    extern __shared__ uint32_t dynlds[];
    const int nw_ = (blockDim.x + 63) >> 6, wcap_ = min(64, CAND_CAP / nw_);
    uint32_t* rawlds = dynlds;
    uint32_t* cand0 = dynlds + blockDim.x * 8;
    uint32_t* cand[2] = {cand0, cand0 + nw_ * wcap_};
    uint32_t (*hist)[256] = (uint32_t (*)[256])dynlds;
    const int uw = max(768, (int)blockDim.x * 8 + 2 * nw_ * wcap_ + 256), mw = (len + 31) >> 5;   // + 2 x 128 packed candidates
    uint32_t* omask[2] = {dynlds + uw, dynlds + uw + mw};

    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    const int j0 = tid * 16;
    const bool active = j0 < len;
    int seg = 0, pos = 0;
    if (active) seg_pos(gm, j0, seg, pos);
    // uniform row base (scalar registers) + a 32-bit lane offset: the loads and stores address as SGPR base + VGPR offset
    // instead of carrying 64-bit vector arithmetic per pointer (the host checks that a row spans < 2^31 elements)
    const int64_t row_base = row_base_of(gm, r);
    const uint32_t loff = (uint32_t)seg * (uint32_t)gm.seg_stride + (uint32_t)pos;
    const uint16_t* xrow = x + row_base;
    const int64_t orow_base = row_base_out(gm, r);
    const uint32_t ooff = (uint32_t)seg * (uint32_t)gm.o_seg_stride + (uint32_t)pos;
    uint32_t* code_row = code + orow_base / CPW;                // row bases are multiples of the group size
    ST* scale_row = scale + (orow_base >> gm.group_shift);
    ST* mn_row = mn + (orow_base >> gm.group_shift);
    uint16_t* err_row = err ? err + row_base : nullptr;

    uint4 ra = make_uint4(0, 0, 0, 0), rb = ra;
    if (active) {
        const uint4* p = (const uint4*)(xrow + loff);
        ra = p[0];
        rb = p[1];
    }
    const uint32_t rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};

    uint32_t outl = 0u;                      // bit j: element j of this lane is an outlier (either side)
    uint32_t m[8];                           // 0xFFFF in the half-word of every outlier element
#pragma unroll
    for (int w = 0; w < 8; w++) m[w] = 0u;
    float mean = 0.0f;
    if (k > 0) {
        // ---------------- row sum / sum of squares on the packed halves (exact products, fp32 accumulate)
        const half2v ones = {(_Float16)1.0f, (_Float16)1.0f};
        float s = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const half2v xv = __builtin_bit_cast(half2v, rw[w]);
            s = __builtin_amdgcn_fdot2(xv, ones, s, false);
            s2 = __builtin_amdgcn_fdot2(xv, xv, s2, false);
        }
        s = wave_sum_dpp(s);
        s2 = wave_sum_dpp(s2);
        if (lane == 0) { wave_sum[wave] = s; wave_thr[0][wave] = __float_as_uint(s2); }
        if (tid < 2 * mw) omask[0][tid] = 0u;                                  // (2 mw <= blockDim + 2)
        if (tid + (int)blockDim.x < 2 * mw) omask[0][tid + blockDim.x] = 0u;
        __syncthreads();
        // row totals: lane w < nw picks up wave w's partial sums, one DPP reduction each, results in SGPRs
        const float tot1 = wave_sum_dpp(lane < nw ? wave_sum[lane] : 0.0f);
        const float tot2 = wave_sum_dpp(lane < nw ? __uint_as_float(wave_thr[0][lane]) : 0.0f);
        // (a power-of-two length divides exactly by a multiply with rlen = 1 / len from the host; the oracle's mean is
        // sum / len in fp32)
        mean = ((len & (len - 1)) == 0) ? tot1 * rlen : tot1 / (float)len;
        bool use_hist = (k > 64) || (zthr <= 0.0f);
        bool payload_done = false;
        uint32_t flag_lo = 0u, flag_hi = 0u;
        if (!use_hist) {
            // ---------------- tier 0: Gaussian-guess thresholds, validated by the survivor counts (see the kernel above)
            // (the thresholds are a guess that the survivor counts validate: approximate reciprocal / square root do)
            const float sd = __builtin_amdgcn_sqrtf(fmaxf(tot2 * rlen - mean * mean, 0.0f));
            const float thi = mean + zthr * sd, tlo = mean - zthr * sd;
            const uint32_t th = f2h_bits(thi), tl = f2h_bits(tlo);
            const half2v thi2 = __builtin_bit_cast(half2v, th | (th << 16)), tlo2 = __builtin_bit_cast(half2v, tl | (tl << 16));
            // sign bits of x - thi / tlo - x gathered in element order: bit 2w <- element 2w, bit 16 + 2w <- element 2w + 1
            uint32_t sg_hi = 0u, sg_lo = 0u;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const half2v xv = __builtin_bit_cast(half2v, rw[w]);
                const uint32_t dh = __builtin_bit_cast(uint32_t, (half2v)(xv - thi2));
                const uint32_t dl = __builtin_bit_cast(uint32_t, (half2v)(tlo2 - xv));
                sg_hi |= (dh >> (15 - 2 * w)) & (0x00010001u << (2 * w));
                sg_lo |= (dl >> (15 - 2 * w)) & (0x00010001u << (2 * w));
            }
            *(uint4*)&rawlds[tid * 8] = ra;
            *(uint4*)&rawlds[tid * 8 + 4] = rb;
            // candidates (sign clear: x >= thi / x <= tlo), one bit per element: large side in bits 0..15, small side in 16..31
            const uint32_t mh = active ? (~(sg_hi | (sg_hi >> 15)) & 0xFFFFu) : 0u;
            const uint32_t ml = active ? (~(sg_lo | (sg_lo >> 15)) & 0xFFFFu) : 0u;
            // Candidates are compacted IN INDEX ORDER into one region per wave (wave-level prefix sum of the lane counts on
            // DPP, no returning LDS atomics): the selecting wave then sees them sorted by index, so ties at the threshold
            // value resolve "lower index first" by position and the outputs come out sorted without a sort.
            const int wcap = wcap_;
            const uint32_t cntp = (uint32_t)__popc(mh) | ((uint32_t)__popc(ml) << 16);
            const uint32_t incl = wave_incl_scan_u32(cntp);
            const uint32_t excl = incl - cntp;
            if (lane == 63) wave_thr[1][wave] = incl;
            {   // one loop over the lane's candidate bits, lowest first (a wave runs as many trips as its busiest lane has bits)
                uint32_t cm = mh | (ml << 16);
                uint32_t slot_h = excl & 0xFFFFu, slot_l = excl >> 16;
                const uint16_t* rawh = (const uint16_t*)rawlds + tid * 16;
                uint32_t* cbase = cand[0] + wave * wcap;
                const uint32_t side_off = (uint32_t)(nw_ * wcap_);
                while (cm) {
                    const uint32_t b = (uint32_t)__builtin_ctz(cm);
                    cm &= cm - 1u;
                    const uint32_t side = b >> 4, j = b & 15u;
                    const uint32_t bits = rawh[j];
                    const uint32_t slot = side ? slot_l : slot_h;
                    if (slot < (uint32_t)wcap) cbase[side * side_off + slot] = (bits << 16) | (uint32_t)(j0 + (int)j);
                    slot_l += side;
                    slot_h += 1u - side;
                }
            }
            // the raw copy is dead now: the lane clears its slot, which becomes its half-word outlier marks
            *(uint4*)&rawlds[tid * 8] = make_uint4(0, 0, 0, 0);
            *(uint4*)&rawlds[tid * 8 + 4] = make_uint4(0, 0, 0, 0);
            __syncthreads();
            uint32_t nh = 0, nl = 0;
            bool over = false;
            if (nw <= 4) {   // one 16-byte LDS read, the rest on the scalar unit (the counts are wave-uniform)
                const uint4 c4 = *(const uint4*)&wave_thr[1][0];
                const uint32_t cs[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)c4.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)c4.y),
                                        (uint32_t)__builtin_amdgcn_readfirstlane((int)c4.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)c4.w)};
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t c = (w < nw) ? cs[w] : 0u;
                    nh += c & 0xFFFFu;
                    nl += c >> 16;
                    over |= ((c & 0xFFFFu) > (uint32_t)wcap) || ((c >> 16) > (uint32_t)wcap);
                }
            } else
            for (int w = 0; w < nw; w++) {
                const uint32_t c = wave_thr[1][w];
                nh += c & 0xFFFFu;
                nl += c >> 16;
                over |= ((c & 0xFFFFu) > (uint32_t)wcap) || ((c >> 16) > (uint32_t)wcap);
            }
            use_hist = over || (nh < (uint32_t)k) || (nl < (uint32_t)k) || (nh > 128u) || (nl > 128u);   // block-uniform
            if (!use_hist) {
                for (int side = 0; side < 2; side++) {
                    if (wave != ((nw > 1) ? side : 0)) continue;
                    const uint32_t n = side == 0 ? nh : nl;
                    // The regions hold the candidates in index order, wave after wave.  The selecting wave first packs them
                    // into one contiguous list (slot `lane` of every region -> list position prefix + lane: wave-uniform
                    // prefixes, no per-lane search for "which region holds position g"), then takes positions lane, lane + 64.
                    uint32_t* packed = cand0 + 2 * nw * wcap + side * 128;     // (behind the regions of both sides)
                    {
                        uint32_t pre = 0u;
                        for (int w = 0; w < nw; w++) {
                            const uint32_t c = wave_thr[1][w];
                            const uint32_t cw = side == 0 ? (c & 0xFFFFu) : (c >> 16);
                            if ((uint32_t)lane < cw) packed[pre + lane] = cand[side][w * wcap + lane];
                            pre += cw;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    const bool v0 = (uint32_t)lane < n, v1 = (uint32_t)(lane + 64) < n;
                    const uint32_t c0 = v0 ? packed[lane] : 0u, c1 = v1 ? packed[lane + 64] : 0u;
                    // order key, larger = selected first (large side: the value key; small side: its complement); +1 so that
                    // 0 means "no candidate"
                    const uint32_t ka = sort_key(c0 >> 16), kb = sort_key(c1 >> 16);
                    const uint32_t x0 = v0 ? (side == 0 ? ka : 0xFFFFu - ka) + 1u : 0u;
                    const uint32_t x1 = v1 ? (side == 0 ? kb : 0xFFFFu - kb) + 1u : 0u;
                    uint32_t lo_b = 1u, hi_b = 0x10000u;   // largest Kt with count(x >= Kt) >= k   (count(x >= 1) = n >= k)
                    for (int it = 0; it < 17; it++) {
                        const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
                        const int cnt = __popcll(__ballot(x0 >= mid)) + __popcll(__ballot(x1 >= mid));
                        if (cnt >= k) lo_b = mid; else hi_b = mid - 1u;
                    }
                    const int above = __popcll(__ballot(x0 > lo_b)) + __popcll(__ballot(x1 > lo_b));
                    const int need = k - above;                      // how many of the ties at the threshold value are taken
                    const bool t0 = x0 == lo_b, t1 = x1 == lo_b;
                    const unsigned long long bt0 = __ballot(t0), bt1 = __ballot(t1);
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    const int r0 = __popcll(bt0 & lt), r1 = __popcll(bt0) + __popcll(bt1 & lt);
                    const bool s0 = x0 > lo_b || (t0 && r0 < need), s1 = x1 > lo_b || (t1 && r1 < need);
                    const unsigned long long b0 = __ballot(s0), b1 = __ballot(s1);
                    const int p0 = __popcll(b0 & lt), p1 = __popcll(b0) + __popcll(b1 & lt);
                    uint16_t* oi = oidx + lrow_of(gm, r) * (int64_t)(2 * k) + (side == 0 ? k : 0);
                    uint16_t* ov = oval + lrow_of(gm, r) * (int64_t)(2 * k) + (side == 0 ? k : 0);
                    if (s0) {
                        const uint32_t idx = c0 & 0xFFFFu;
                        oi[p0] = (uint16_t)idx;
                        ov[p0] = (uint16_t)(c0 >> 16);
                        atomicOr(&omask[side][idx >> 5], 1u << (idx & 31));
                        ((uint16_t*)rawlds)[idx] = (uint16_t)0xFFFFu;
                    }
                    if (s1) {
                        const uint32_t idx = c1 & 0xFFFFu;
                        oi[p1] = (uint16_t)idx;
                        ov[p1] = (uint16_t)(c1 >> 16);
                        atomicOr(&omask[side][idx >> 5], 1u << (idx & 31));
                        ((uint16_t*)rawlds)[idx] = (uint16_t)0xFFFFu;
                    }
                }
                __syncthreads();
                if (active) {
                    flag_hi = (omask[0][j0 >> 5] >> (j0 & 31)) & 0xFFFFu;
                    flag_lo = (omask[1][j0 >> 5] >> (j0 & 31)) & 0xFFFFu;
                    const uint4 ma = *(const uint4*)&rawlds[tid * 8], mb = *(const uint4*)&rawlds[tid * 8 + 4];
                    m[0] = ma.x; m[1] = ma.y; m[2] = ma.z; m[3] = ma.w;
                    m[4] = mb.x; m[5] = mb.y; m[6] = mb.z; m[7] = mb.w;
                }
                payload_done = true;
            }
        }
        if (use_hist) {
            // ---------------- exact fallback: two-level radix select on the 16-bit order key (as in the kernel above)
            uint32_t hb[16], key[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                hb[j] = (rw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                key[j] = sort_key(hb[j]);
            }
            for (int i = tid; i < 3 * 256; i += blockDim.x) (&hist[0][0])[i] = 0u;
            __syncthreads();
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) atomicAdd(&hist[0][key[j] >> 8], 1u);
            }
            __syncthreads();
            if (wave == 0) {
                find_crossing<true>(hist[0], k, &sh[0], &sh[1]);
                find_crossing<false>(hist[0], k, &sh[2], &sh[3]);
            }
            __syncthreads();
            const uint32_t bin_hi = (uint32_t)sh[0], bin_lo = (uint32_t)sh[2];
            const int need_hi = sh[1], need_lo = sh[3];
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if ((key[j] >> 8) == bin_hi) atomicAdd(&hist[1][key[j] & 255u], 1u);
                    if ((key[j] >> 8) == bin_lo) atomicAdd(&hist[2][key[j] & 255u], 1u);
                }
            }
            __syncthreads();
            if (wave == 0) {
                find_crossing<true>(hist[1], need_hi, &sh[4], &sh[5]);
                find_crossing<false>(hist[2], need_lo, &sh[6], &sh[7]);
            }
            __syncthreads();
            const uint32_t thr_hi = (bin_hi << 8) | (uint32_t)sh[4];
            const uint32_t thr_lo = (bin_lo << 8) | (uint32_t)sh[6];
            const int take_hi = sh[5], take_lo = sh[7];
            unsigned long long eq = 0;
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    eq += (key[j] == thr_hi) ? 1ull : 0ull;
                    eq += (key[j] == thr_lo) ? (1ull << 32) : 0ull;
                }
            }
            unsigned long long ex = block_excl_scan(eq, wave_tot, nullptr);
            int rank_hi = (int)(ex & 0xFFFFFFFFull), rank_lo = (int)(ex >> 32);
            if (active) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (key[j] < thr_lo) flag_lo |= 1u << j;
                    else if (key[j] == thr_lo) { if (rank_lo < take_lo) flag_lo |= 1u << j; rank_lo++; }
                    if (key[j] > thr_hi) flag_hi |= 1u << j;
                    else if (key[j] == thr_hi) { if (rank_hi < take_hi) flag_hi |= 1u << j; rank_hi++; }
                }
            }
            unsigned long long cnt = (unsigned long long)__popc(flag_hi) | ((unsigned long long)__popc(flag_lo) << 32);
            unsigned long long slot = block_excl_scan(cnt, wave_tot, nullptr);
            int slot_hi = (int)(slot & 0xFFFFFFFFull), slot_lo = (int)(slot >> 32);
            uint16_t* oi = oidx + lrow_of(gm, r) * (int64_t)(2 * k);
            uint16_t* ov = oval + lrow_of(gm, r) * (int64_t)(2 * k);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (flag_lo & (1u << j)) {
                    if (slot_lo < k) { oi[slot_lo] = (uint16_t)(j0 + j); ov[slot_lo] = (uint16_t)hb[j]; }
                    slot_lo++;
                }
                if (flag_hi & (1u << j)) {
                    if (slot_hi < k) { oi[k + slot_hi] = (uint16_t)(j0 + j); ov[k + slot_hi] = (uint16_t)hb[j]; }
                    slot_hi++;
                }
            }
            const uint32_t fl = flag_lo | flag_hi;
#pragma unroll
            for (int w = 0; w < 8; w++) m[w] = (((fl >> (2 * w)) & 1u) ? 0xFFFFu : 0u) | (((fl >> (2 * w + 1)) & 1u) ? 0xFFFF0000u : 0u);
            (void)payload_done;
        }
        outl = flag_lo | flag_hi;
        if (tid == 0 && omean) omean[r] = mean;
    }

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
    if (!active) return;
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
    if ((tid & (lanes_per_group - 1)) == 0) {
        st_st<ST>(scale_row + (ooff >> gm.group_shift), qscale);
        st_st<ST>(mn_row + (ooff >> gm.group_shift), qmn);
    }
    if (err) {
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

// Wave-per-row variant for rows of exactly 1024 C elements (C = 1..5 or 8 chunks of 16 elements per lane): the same algorithm as
// compress_rows_fp32_kernel's fast path -- Gaussian-guess thresholds validated by the survivor counts, candidates compacted
// in index order, 17-round key bisection, dense part -- with ONE wave per row, so that the per-row machinery (row statistics,
// prefix scans, the bisection and its outputs) runs once per row instead of once per 1024 elements and on all four waves of a
// workgroup instead of two, and no workgroup barrier or cross-wave LDS traffic is left (four independent rows per workgroup).
// A row whose guess fails (counts outside [k, 128] on a side) is marked by the impossible index 0xFFFF in its first list slot and
// left to the SLOW instantiation of this kernel (exact in-wave selection over all of the row's elements), which is launched
// behind the fast one over all rows and returns at once for every row that is not marked -- a separate instantiation, so that
// the rare path's registers do not count against the common one.
template <int BITS, int C, bool SLOW>
__global__ __launch_bounds__(256, (C <= 4 ? 4 : (C <= 5 ? 3 : 2))) void compress_rows_wave_kernel(const uint16_t* __restrict__ x, RowGeom gm, int64_t n_rows, int group,
                                                                    int k, float zthr, float rlen, uint32_t* __restrict__ code,
                                                                    float* __restrict__ scale, float* __restrict__ mn,
                                                                    uint16_t* __restrict__ err, uint16_t* __restrict__ oidx,
                                                                    uint16_t* __restrict__ oval, float* __restrict__ omean, int masked) {
    constexpr int LEN = 1024 * C;
    constexpr int CPW = 32 / BITS;
    constexpr int WW = LEN / 2 + 256 + 2 * (LEN / 32);     // words of LDS per wave: row image | candidates [2][128] | outlier bits [2][LEN/32]
    extern __shared__ __attribute__((aligned(16))) uint32_t wdyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto process_row = [&](const int64_t r) {
    uint32_t* rowimg = wdyn + wave * WW;      // the raw row while candidates are emitted, then half-word outlier marks
    uint32_t* cand = rowimg + LEN / 2;
    uint32_t* omask = cand + 256;
    uint32_t loff[C], ooff[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        int seg = 0, pos = 0;
        seg_pos(gm, c * 1024 + lane * 16, seg, pos);
        loff[c] = (uint32_t)seg * (uint32_t)gm.seg_stride + (uint32_t)pos;
        ooff[c] = (uint32_t)seg * (uint32_t)gm.o_seg_stride + (uint32_t)pos;
    }
    const int64_t row_base = row_base_of(gm, r);
    const uint16_t* xrow = x + row_base;
    const int64_t orow_base = row_base_out(gm, r);
    uint32_t* code_row = code + orow_base / CPW;
    float* scale_row = scale + (orow_base >> gm.group_shift);
    float* mn_row = mn + (orow_base >> gm.group_shift);
    uint16_t* err_row = err ? err + row_base : nullptr;

    uint4 ra[C], rb[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const uint4* p = (const uint4*)(xrow + loff[c]);
        ra[c] = p[0];
        rb[c] = p[1];
    }
    uint32_t flag[C];
#pragma unroll
    for (int c = 0; c < C; c++) flag[c] = 0u;
    float mean = 0.0f;
    if (k > 0) {
        const int64_t lrow = lrow_of(gm, r);
        // ---------------- row sum / sum of squares (exact products, fp32 accumulate), one DPP reduction each
        const half2v ones = {(_Float16)1.0f, (_Float16)1.0f};
        // (the workgroup kernel reduces each wave's 1024 elements first and adds the wave sums as a DPP tree, ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7));
        // chunk c here holds what wave c holds there, so the chunks are reduced one by one and added in the same order: the
        // mean is the fill value and must come out with the same bits)
        float wsum[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, s2 = 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t rw[8] = {ra[c].x, ra[c].y, ra[c].z, ra[c].w, rb[c].x, rb[c].y, rb[c].z, rb[c].w};
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const half2v xv = __builtin_bit_cast(half2v, rw[w]);
                s = __builtin_amdgcn_fdot2(xv, ones, s, false);
                s2 = __builtin_amdgcn_fdot2(xv, xv, s2, false);
            }
            wsum[c] = wave_sum_dpp(s);
        }
        const float tot1 = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7])), tot2 = wave_sum_dpp(s2);
        mean = ((LEN & (LEN - 1)) == 0) ? tot1 * rlen : tot1 / (float)LEN;
        bool ok = !SLOW && zthr > 0.0f;
        uint32_t mh[C], ml[C], excl[C];
        uint32_t nh = 0u, nl = 0u, bh[C], bl[C];
        if (!SLOW && ok) {
            const float sd = __builtin_amdgcn_sqrtf(fmaxf(tot2 * rlen - mean * mean, 0.0f));
            const float thi = mean + zthr * sd, tlo = mean - zthr * sd;
            const uint32_t th = f2h_bits(thi), tl = f2h_bits(tlo);
            const half2v thi2 = __builtin_bit_cast(half2v, th | (th << 16)), tlo2 = __builtin_bit_cast(half2v, tl | (tl << 16));
#pragma unroll
            for (int c = 0; c < C; c++) {
                const uint32_t rw[8] = {ra[c].x, ra[c].y, ra[c].z, ra[c].w, rb[c].x, rb[c].y, rb[c].z, rb[c].w};
                uint32_t sg_hi = 0u, sg_lo = 0u;
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    const half2v xv = __builtin_bit_cast(half2v, rw[w]);
                    const uint32_t dh = __builtin_bit_cast(uint32_t, (half2v)(xv - thi2));
                    const uint32_t dl = __builtin_bit_cast(uint32_t, (half2v)(tlo2 - xv));
                    sg_hi |= (dh >> (15 - 2 * w)) & (0x00010001u << (2 * w));
                    sg_lo |= (dl >> (15 - 2 * w)) & (0x00010001u << (2 * w));
                }
                *(uint4*)&rowimg[c * 512 + lane * 8] = ra[c];
                *(uint4*)&rowimg[c * 512 + lane * 8 + 4] = rb[c];
                mh[c] = ~(sg_hi | (sg_hi >> 15)) & 0xFFFFu;
                ml[c] = ~(sg_lo | (sg_lo >> 15)) & 0xFFFFu;
                const uint32_t cntp = (uint32_t)__popc(mh[c]) | ((uint32_t)__popc(ml[c]) << 16);
                const uint32_t incl = wave_incl_scan_u32(cntp);
                excl[c] = incl - cntp;
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                bh[c] = nh; bl[c] = nl;
                nh += tot & 0xFFFFu;
                nl += tot >> 16;
            }
            ok = nh >= (uint32_t)k && nl >= (uint32_t)k && nh <= 128u && nl <= 128u;
        }
        if (!SLOW && !ok) {
            if (lane == 0) oidx[lrow * (int64_t)(2 * k)] = (uint16_t)0xFFFFu;      // left to the SLOW instantiation
            return;
        }
        if constexpr (SLOW) {
            // ---------------- the guess failed (heavy tails, few distinct values, constant rows ...): exact selection on the whole
            // row inside the wave.  Per side: 17-round bisection on the order key over all 16 C elements of every lane, then, in
            // index order (chunk, lane, element): everything above the threshold value plus the first `need` ties.  Lane-local
            // except for the counts (DPP sums / prefix scans); rare, so it is written for size, not speed.
            uint32_t selm[2][C];
#pragma unroll
            for (int c = 0; c < C; c++) {
                *(uint4*)&rowimg[c * 512 + lane * 8] = ra[c];
                *(uint4*)&rowimg[c * 512 + lane * 8 + 4] = rb[c];
            }
            const uint16_t* rawrow = (const uint16_t*)rowimg;
            for (int side = 0; side < 2; side++) {
                auto key_of = [&](int c, int j) {
                    const uint32_t ky = sort_key(rawrow[c * 1024 + lane * 16 + j]);
                    return (side == 0 ? ky : 0xFFFFu - ky) + 1u;
                };
                auto count = [&](uint32_t thr, bool strict) {     // wave-wide #{key >= thr} or #{key > thr}
                    int cnt = 0;
                    for (int c = 0; c < C; c++)
                        for (int j = 0; j < 16; j++) {
                            const uint32_t ky = key_of(c, j);
                            cnt += (strict ? ky > thr : ky >= thr) ? 1 : 0;
                        }
                    return (int)wave_sum_dpp((float)cnt);         // (<= 4096: exact in fp32)
                };
                uint32_t lo_b = 1u, hi_b = 0x10000u;
                for (int it = 0; it < 17; it++) {
                    const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
                    if (count(mid, false) >= k) lo_b = mid; else hi_b = mid - 1u;
                }
                const uint32_t vstar = lo_b;
                int need = k - count(vstar, true);                // ties still to take, in index order
                for (int c = 0; c < C; c++) {
                    uint32_t gt = 0u, eq = 0u;
                    for (int j = 0; j < 16; j++) {
                        const uint32_t ky = key_of(c, j);
                        gt |= (ky > vstar ? 1u : 0u) << j;
                        eq |= (ky == vstar ? 1u : 0u) << j;
                    }
                    const uint32_t ne = (uint32_t)__popc(eq);
                    const uint32_t incl = wave_incl_scan_u32(ne);
                    const int before = (int)(incl - ne);
                    int take = need - before;
                    take = take < 0 ? 0 : (take > (int)ne ? (int)ne : take);
                    uint32_t sel = gt, rem = eq;
                    for (int t = 0; t < take; t++) { sel |= rem & (0u - rem); rem &= rem - 1u; }
                    selm[side][c] = sel;
                    need -= __builtin_amdgcn_readlane((int)incl, 63);     // (may go negative: no more ties taken)
                }
            }
            // outputs in index order + marks
            uint32_t pos_h = 0u, pos_l = 0u;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const uint32_t cntp = (uint32_t)__popc(selm[0][c]) | ((uint32_t)__popc(selm[1][c]) << 16);
                const uint32_t incl = wave_incl_scan_u32(cntp);
                const uint32_t ex = incl - cntp;
                uint32_t ph = pos_h + (ex & 0xFFFFu), pl = pos_l + (ex >> 16);
                uint16_t* oi = oidx + lrow * (int64_t)(2 * k);
                uint16_t* ov = oval + lrow * (int64_t)(2 * k);
                for (uint32_t mm = selm[0][c]; mm; mm &= mm - 1u) {
                    const int j = __builtin_ctz(mm), idx = c * 1024 + lane * 16 + j;
                    oi[k + ph] = (uint16_t)idx;
                    ov[k + ph] = rawrow[idx];
                    ph++;
                }
                for (uint32_t mm = selm[1][c]; mm; mm &= mm - 1u) {
                    const int j = __builtin_ctz(mm), idx = c * 1024 + lane * 16 + j;
                    oi[pl] = (uint16_t)idx;
                    ov[pl] = rawrow[idx];
                    pl++;
                }
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                pos_h += tot & 0xFFFFu;
                pos_l += tot >> 16;
            }
#pragma unroll
            for (int c = 0; c < C; c++) {
                *(uint4*)&rowimg[c * 512 + lane * 8] = make_uint4(0, 0, 0, 0);
                *(uint4*)&rowimg[c * 512 + lane * 8 + 4] = make_uint4(0, 0, 0, 0);
                flag[c] = selm[0][c] | selm[1][c];
                for (uint32_t mm = flag[c]; mm; mm &= mm - 1u)
                    ((uint16_t*)rowimg)[c * 1024 + lane * 16 + __builtin_ctz(mm)] = (uint16_t)0xFFFFu;
            }
        } else {
        // ---------------- candidates in index order: chunk after chunk, lane after lane, element after element
#pragma unroll
        for (int c = 0; c < C; c++) {
            uint32_t cm = mh[c] | (ml[c] << 16);
            uint32_t slot_h = bh[c] + (excl[c] & 0xFFFFu), slot_l = bl[c] + (excl[c] >> 16);
            const uint16_t* rawh = (const uint16_t*)rowimg + c * 1024 + lane * 16;
            while (cm) {
                const uint32_t b = (uint32_t)__builtin_ctz(cm);
                cm &= cm - 1u;
                const uint32_t side = b >> 4, j = b & 15u;
                const uint32_t bits = rawh[j];
                const uint32_t slot = side ? slot_l : slot_h;
                cand[side * 128u + slot] = (bits << 16) | (uint32_t)(c * 1024 + lane * 16 + (int)j);
                slot_l += side;
                slot_h += 1u - side;
            }
            // (the raw copy stays: the selected elements are replaced in it by fp16(mean), and the dense part reads it back;
            // option rows_masked: the slot becomes the lane's half-word outlier marks)
            if (masked) {
                *(uint4*)&rowimg[c * 512 + lane * 8] = make_uint4(0, 0, 0, 0);
                *(uint4*)&rowimg[c * 512 + lane * 8 + 4] = make_uint4(0, 0, 0, 0);
            }
        }
        for (int i = lane; i < 2 * (LEN / 32); i += 64) omask[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---------------- per side: the k-th largest order key by bisection, ties "lower index first" by position
        for (int side = 0; side < 2; side++) {
            const uint32_t n = side == 0 ? nh : nl;
            const bool v0 = (uint32_t)lane < n, v1 = (uint32_t)(lane + 64) < n;
            const uint32_t c0 = v0 ? cand[side * 128 + lane] : 0u, c1 = v1 ? cand[side * 128 + lane + 64] : 0u;
            const uint32_t ka = sort_key(c0 >> 16), kb = sort_key(c1 >> 16);
            const uint32_t x0 = v0 ? (side == 0 ? ka : 0xFFFFu - ka) + 1u : 0u;
            const uint32_t x1 = v1 ? (side == 0 ? kb : 0xFFFFu - kb) + 1u : 0u;
            uint32_t lo_b = 1u, hi_b = 0x10000u;
            for (int it = 0; it < 17; it++) {
                const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
                const int cnt = __popcll(__ballot(x0 >= mid)) + __popcll(__ballot(x1 >= mid));
                if (cnt >= k) lo_b = mid; else hi_b = mid - 1u;
            }
            const int above = __popcll(__ballot(x0 > lo_b)) + __popcll(__ballot(x1 > lo_b));
            const int need = k - above;
            const bool t0 = x0 == lo_b, t1 = x1 == lo_b;
            const unsigned long long bt0 = __ballot(t0), bt1 = __ballot(t1);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int r0 = __popcll(bt0 & lt), r1 = __popcll(bt0) + __popcll(bt1 & lt);
            const bool s0 = x0 > lo_b || (t0 && r0 < need), s1 = x1 > lo_b || (t1 && r1 < need);
            const unsigned long long b0 = __ballot(s0), b1 = __ballot(s1);
            const int p0 = __popcll(b0 & lt), p1 = __popcll(b0) + __popcll(b1 & lt);
            uint16_t* oi = oidx + lrow * (int64_t)(2 * k) + (side == 0 ? k : 0);
            uint16_t* ov = oval + lrow * (int64_t)(2 * k) + (side == 0 ? k : 0);
            // the selected element's stand-in in the staged row: fp16(mean) for dense16s, the 0xFFFF mark for dense16 (option rows_masked)
            const uint16_t subst = masked ? (uint16_t)0xFFFFu : f2h_bits(mean);
            if (s0) {
                const uint32_t idx = c0 & 0xFFFFu;
                oi[p0] = (uint16_t)idx;
                ov[p0] = (uint16_t)(c0 >> 16);
                atomicOr(&omask[side * (LEN / 32) + (idx >> 5)], 1u << (idx & 31));
                ((uint16_t*)rowimg)[idx] = subst;
            }
            if (s1) {
                const uint32_t idx = c1 & 0xFFFFu;
                oi[p1] = (uint16_t)idx;
                ov[p1] = (uint16_t)(c1 >> 16);
                atomicOr(&omask[side * (LEN / 32) + (idx >> 5)], 1u << (idx & 31));
                ((uint16_t*)rowimg)[idx] = subst;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int j0 = c * 1024 + lane * 16;
            flag[c] = ((omask[j0 >> 5] | omask[LEN / 32 + (j0 >> 5)]) >> (j0 & 31)) & 0xFFFFu;
        }
        }
        if (lane == 0 && omean) omean[r] = mean;
    }
    // ---------------- dense part, chunk by chunk
    if constexpr (!SLOW) {
        if (k > 0 && !masked) {
            // the common path: the staged row with its outliers replaced by fp16(mean) -> the mask-free dense16s
            const uint32_t sv = f2h_bits(mean);
#pragma unroll
            for (int c = 0; c < C; c++) {
                const uint4 sa = *(const uint4*)&rowimg[c * 512 + lane * 8], sb = *(const uint4*)&rowimg[c * 512 + lane * 8 + 4];
                const uint32_t rs[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
                dense16s<BITS>(rs, flag[c], mean, sv, group, gm.group_shift, lane, code_row, scale_row, mn_row, ooff[c], err_row, loff[c]);
            }
            return;
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++) {
        const uint32_t rw[8] = {ra[c].x, ra[c].y, ra[c].z, ra[c].w, rb[c].x, rb[c].y, rb[c].z, rb[c].w};
        uint32_t m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        if (k > 0) {
            const uint4 ma = *(const uint4*)&rowimg[c * 512 + lane * 8], mb = *(const uint4*)&rowimg[c * 512 + lane * 8 + 4];
            m[0] = ma.x; m[1] = ma.y; m[2] = ma.z; m[3] = ma.w;
            m[4] = mb.x; m[5] = mb.y; m[6] = mb.z; m[7] = mb.w;
        }
        dense16<BITS>(rw, m, flag[c], mean, group, gm.group_shift, lane, code_row, scale_row, mn_row, ooff[c], err_row, loff[c]);
    }
    };
    if (!SLOW) {
        const int64_t r = (int64_t)blockIdx.x * 4 + wave;
        if (r < n_rows) process_row(r);
    } else {
        // fallback pass (launched only with k > 0): a wave looks at the marks of 8 rows and takes the marked ones
        const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 8;
        const bool marked = lane < 8 && r0 + lane < n_rows && oidx[lrow_of(gm, r0 + lane) * (int64_t)(2 * k)] == (uint16_t)0xFFFFu;
        for (unsigned long long todo = __ballot(marked); todo; todo &= todo - 1ull) process_row(r0 + __builtin_ctzll(todo));
    }
}

}  // namespace

bool gear_rows_multi_supported(int64_t len, int group, int k);
int gear_rows_multi_launch(const void* x, const void* gm, int64_t n_rows, int64_t len, int group, int bits, int mode, int k,
                           void* code, void* scale, void* mn, void* err, void* oidx, void* oval, void* omean, hipStream_t st);

static double inv_norm_cdf(double p) {
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                               1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                               6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    if (p < 0.02425) {
        double q = sqrt(-2 * log(p));
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    if (p > 1 - 0.02425) return -inv_norm_cdf(1 - p);
    double q = p - 0.5, rr = q * q;
    return (((((a[0] * rr + a[1]) * rr + a[2]) * rr + a[3]) * rr + a[4]) * rr + a[5]) * q /
           (((((b[0] * rr + b[1]) * rr + b[2]) * rr + b[3]) * rr + b[4]) * rr + 1);
}

int gear_compress_rows_geom(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride,
                            int nseg, int seglen, int64_t seg_stride, int64_t o_outer_stride, int64_t o_inner_stride,
                            int64_t o_seg_stride, int o_list_outer, int group, int bits, int mode, int k, void* code,
                            void* scale, void* mn, void* err, void* oidx, void* oval, void* omean, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_compress_rows: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_compress_rows: bad mode %d", mode);
    GEAR_CHECK_ARG(n_rows > 0 && n_rows < 0x7FFFFFFFLL && nseg > 0 && seglen > 0, "gear_compress_rows: empty input");
    const int64_t len = (int64_t)nseg * seglen;
    GEAR_CHECK_ARG(len <= 16384, "gear_compress_rows: row length %lld exceeds 16384", (long long)len);
    GEAR_CHECK_ARG(group >= 16 && gear_is_pow2(group / 16) && group % 16 == 0 && group <= 1024 && seglen % group == 0,
                   "gear_compress_rows: group %d must be a power of two in [16,1024] dividing the segment length %d", group, seglen);
    GEAR_CHECK_ARG(k >= 0 && 2 * (int64_t)k <= len, "gear_compress_rows: k=%d out of range for row length %lld", k, (long long)len);
    GEAR_CHECK_ARG(outer_stride % group == 0 && inner_stride % group == 0 && (nseg == 1 || seg_stride % group == 0),
                   "gear_compress_rows: strides must be multiples of the group size");
    GEAR_CHECK_ARG(o_outer_stride % group == 0 && o_inner_stride % group == 0 && (nseg == 1 || o_seg_stride % group == 0) &&
                   (nseg - 1) * o_seg_stride + seglen < 0x7FFFFFFFLL && o_seg_stride >= 0,
                   "gear_compress_rows: bad output strides");
    GEAR_CHECK_ARG(o_list_outer >= rows_inner, "gear_compress_rows: bad sparse-list row pitch");
    GEAR_CHECK_ARG(x && code && scale && mn, "gear_compress_rows: null pointer");
    GEAR_CHECK_ARG((nseg - 1) * seg_stride + seglen < 0x7FFFFFFFLL && seg_stride >= 0,
                   "gear_compress_rows: a row must span fewer than 2^31 elements");
    GEAR_CHECK_ARG(k == 0 || (oidx && oval), "gear_compress_rows: outlier buffers required when k > 0");
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) l++; return l; };
    RowGeom gm{rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride,
               gear_is_pow2(seglen) ? ilog2(seglen) : -1, ilog2(group), o_outer_stride, o_inner_stride, o_seg_stride, o_list_outer};
    int threads = (int)((len / 16 + 63) / 64 * 64);
    // tier-0 threshold: let about 2.2 k of a normal row's elements pass on each side (at most 128 may)
    float zthr = 0.0f;
    if (k > 0 && k <= 58 && !gear_options().rows_hist_only) {
        double frac = 2.2 * (double)k / (double)len;
        if (frac < 0.45) zthr = (float)(-inv_norm_cdf(frac));
    }
    hipStream_t st = (hipStream_t)stream;
    // short rows (head shards: 128 .. 512 elements, a handful of outliers): eight rows per wave (rows_multi.hip)
    if (!gear_options().rows_wg_only && !gear_options().rows_v1 && !gear_options().rows_hist_only && gear_rows_multi_supported(len, group, k)) {
        gear_rows_multi_launch(x, &gm, n_rows, len, group, bits, mode, k, code, scale, mn, err, oidx, oval, omean, st);
        GEAR_CHECK_LAUNCH("gear_compress_rows");
        return 0;
    }
    dim3 block(threads), grid((unsigned)n_rows);
#define GO(B, M, STT)                                                                                                  \
    hipLaunchKernelGGL((compress_rows_kernel<B, M, STT>), grid, block, (size_t)threads * 32, st, (const uint16_t*)x, gm, (int)len, group, k, zthr, \
                       (uint32_t*)code, (STT*)scale, (STT*)mn, (uint16_t*)err, (uint16_t*)oidx, (uint16_t*)oval,       \
                       (float*)omean)
    const int nwh = threads / 64, wcaph = (64 < CAND_CAP / nwh) ? 64 : CAND_CAP / nwh;
    const int uwh = (768 > threads * 8 + 2 * nwh * wcaph + 256) ? 768 : threads * 8 + 2 * nwh * wcaph + 256;
    const size_t lds2 = ((size_t)uwh + 2 * (size_t)((len + 31) / 32)) * 4;
#define GO2(B)                                                                                                         \
    hipLaunchKernelGGL((compress_rows_fp32_kernel<B, float>), grid, block, lds2, st, (const uint16_t*)x, gm, (int)len, group, k, zthr, 1.0f / (float)len, \
                       (uint32_t*)code, (float*)scale, (float*)mn, (uint16_t*)err, (uint16_t*)oidx, (uint16_t*)oval,   \
                       (float*)omean)
    if (mode == 0) {
        if (bits == 2) GO(2, 0, uint16_t);
        else if (bits == 4) GO(4, 0, uint16_t);
        else GO(8, 0, uint16_t);
    } else if (gear_options().rows_v1) {   // first-generation workgroup kernel (kept for A/B runs and as a cross-check)
        if (bits == 2) GO(2, 1, float);
        else if (bits == 4) GO(4, 1, float);
        else GO(8, 1, float);
    } else {
        // rows of 1024 .. 5120 or 8192 elements: one wave per row (compress_rows_wave_kernel)
        const bool wave_rows = !gear_options().rows_wg_only && len % 1024 == 0 && (len <= 5120 || len == 8192) && (k == 0 || zthr > 0.0f);
        if (wave_rows) {
            const int Cc = (int)(len / 1024);
            const size_t wlds = (size_t)4 * (len / 2 + 256 + 2 * (len / 32)) * 4;
            const dim3 wgrid((unsigned)((n_rows + 3) / 4));
#define GOW(B, CC, SL)                                                                                                  \
    do {                                                                                                                \
        auto kfn = compress_rows_wave_kernel<B, CC, SL>;                                                                \
        const dim3 wg = SL ? dim3((unsigned)((n_rows + 31) / 32)) : wgrid;                                              \
        if (wlds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds); \
        hipLaunchKernelGGL(kfn, wg, dim3(256), wlds, st, (const uint16_t*)x, gm, n_rows, group, k, zthr,                \
                           1.0f / (float)len, (uint32_t*)code, (float*)scale, (float*)mn, (uint16_t*)err, (uint16_t*)oidx, \
                           (uint16_t*)oval, (float*)omean,                                   \
                           (gear_options().rows_masked > 0 || (gear_options().rows_masked == 0 && err != nullptr)) ? 1 : 0);                                 \
    } while (0)
#define GOWC(B)                                                                                                         \
    do {                                                                                                                \
        if (Cc == 1) { GOW(B, 1, false); if (k > 0) GOW(B, 1, true); }                                                  \
        else if (Cc == 2) { GOW(B, 2, false); if (k > 0) GOW(B, 2, true); }                                             \
        else if (Cc == 3) { GOW(B, 3, false); if (k > 0) GOW(B, 3, true); }                                             \
        else if (Cc == 4) { GOW(B, 4, false); if (k > 0) GOW(B, 4, true); }                                             \
        else if (Cc == 5) { GOW(B, 5, false); if (k > 0) GOW(B, 5, true); }                                             \
        else { GOW(B, 8, false); if (k > 0) GOW(B, 8, true); }                                                          \
    } while (0)
            if (bits == 2) GOWC(2); else if (bits == 4) GOWC(4); else GOWC(8);
#undef GOWC
#undef GOW
        } else {
            if (bits == 2) GO2(2);
            else if (bits == 4) GO2(4);
            else GO2(8);
        }
    }
#undef GO2
#undef GO
    GEAR_CHECK_LAUNCH("gear_compress_rows");
    return 0;
}

// The row compressor with the outlier selection given from outside (head-sharded V rows: vsel.hip finds the thresholds over all
// ranks).  Same geometry arguments as gear_compress_rows_geom; one workgroup per row (the rows of a shard are short).
int gear_compress_rows_ext(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride, int nseg,
                           int seglen, int64_t seg_stride, int64_t o_outer_stride, int64_t o_inner_stride, int64_t o_seg_stride,
                           int o_list_outer, int group, int bits, int mode, int k, int col0, const void* thr, const void* fill,
                           void* code, void* scale, void* mn, void* err, void* oidx, void* oval, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_compress_rows_ext: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_compress_rows_ext: bad mode %d", mode);
    GEAR_CHECK_ARG(n_rows > 0 && n_rows < 0x7FFFFFFFLL && nseg > 0 && seglen > 0, "gear_compress_rows_ext: empty input");
    const int64_t len = (int64_t)nseg * seglen;
    GEAR_CHECK_ARG(len <= 16384 && len % 16 == 0, "gear_compress_rows_ext: bad row length %lld", (long long)len);
    GEAR_CHECK_ARG(group >= 16 && gear_is_pow2(group / 16) && group % 16 == 0 && group <= 1024 && seglen % group == 0,
                   "gear_compress_rows_ext: group %d must be a power of two in [16,1024] dividing the segment length %d", group, seglen);
    GEAR_CHECK_ARG(k > 0 && k <= 32767 && thr && fill && oidx && oval, "gear_compress_rows_ext: selection inputs / list outputs missing");
    GEAR_CHECK_ARG(x && code && scale && mn, "gear_compress_rows_ext: null pointer");
    GEAR_CHECK_ARG(outer_stride % group == 0 && inner_stride % group == 0 && (nseg == 1 || seg_stride % group == 0) &&
                   o_outer_stride % group == 0 && o_inner_stride % group == 0 && (nseg == 1 || o_seg_stride % group == 0),
                   "gear_compress_rows_ext: strides must be multiples of the group size");
    GEAR_CHECK_ARG(o_list_outer >= rows_inner && col0 >= 0 && col0 + len <= 0xFFFF, "gear_compress_rows_ext: bad list pitch / column base");
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) l++; return l; };
    RowGeom gm{rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride,
               gear_is_pow2(seglen) ? ilog2(seglen) : -1, ilog2(group), o_outer_stride, o_inner_stride, o_seg_stride, o_list_outer};
    const int threads = (int)((len / 16 + 63) / 64 * 64);
    const ExtSel ext{(const uint32_t*)thr, (const float*)fill, col0};
    dim3 block(threads), grid((unsigned)n_rows);
    hipStream_t st = (hipStream_t)stream;
#define GOX(B, M, STT)                                                                                                 \
    hipLaunchKernelGGL((compress_rows_kernel<B, M, STT, true>), grid, block, (size_t)threads * 32, st, (const uint16_t*)x, gm, (int)len, group, k, 0.0f, \
                       (uint32_t*)code, (STT*)scale, (STT*)mn, (uint16_t*)err, (uint16_t*)oidx, (uint16_t*)oval, (float*)nullptr, ext)
    if (mode == 0) { if (bits == 2) GOX(2, 0, uint16_t); else if (bits == 4) GOX(4, 0, uint16_t); else GOX(8, 0, uint16_t); }
    else { if (bits == 2) GOX(2, 1, float); else if (bits == 4) GOX(4, 1, float); else GOX(8, 1, float); }
#undef GOX
    GEAR_CHECK_LAUNCH("gear_compress_rows_ext");
    return 0;
}

extern "C" int gear_compress_rows(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride,
                                  int64_t inner_stride, int nseg, int seglen, int64_t seg_stride, int group, int bits,
                                  int mode, int k, void* code, void* scale, void* mn, void* err, void* oidx, void* oval,
                                  void* omean, void* stream) {
    return gear_compress_rows_geom(x, n_rows, rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride, outer_stride,
                                   inner_stride, seg_stride, rows_inner, group, bits, mode, k, code, scale, mn, err, oidx, oval, omean,
                                   stream);
}
