// vsel.hip -- exact outlier selection of V token rows whose heads are spread over several GPUs (head-sharded GEAR cache).
//
// The simulated path picks, per token, the k smallest and the k largest values of the row ACROSS ALL heads
// (gears_tokenQ: GenerationBench/GenerationTest/GEARLM/Simulated/compress_function.py:297-333, two torch.topk over H * D columns)
// and fills them with the row mean before quantizing.  A rank holds H / world heads of every row.  Three launches + one all-gather
// give every rank exactly the unsharded selection (rounds 1-3 selected k / world per shard; round 4 did this with ~25 torch
// launches incl. two topk, gear_amd/parallel.py exact_v_selection -- still the cross-check of the tests):
//
//   vsel_cand_kernel   per local row: the k best elements per side as GLOBAL composites (16-bit order key, then lower global column
//                      first -- the build's tie rule) + the exact fp64 sum of the row's local part          -> cand [rows][2k + 1]
//   (all-gather of cand over the ranks: 8 (2k + 1) bytes per row and rank -- RCCL / torch.distributed, gear_amd/parallel.py)
//   vsel_thr_kernel    per row: the k-th largest composite per side over world * k candidates, the fill value from the summed
//                      row sums                                                                               -> thr [rows][2], fill [rows]
//   compress_rows_kernel<.., EXT> (compress_rows.hip): the row compressor with the selection GIVEN -- marks the local elements at or
//                      beyond the thresholds, fills, quantizes, packs, writes the error and the (0xFFFF-padded) sorted lists.
//
// Composites are unique per row (they contain the column), so "the k-th largest" is one element and count(c >= thr) == k over all
// ranks: the shards' outlier sets partition the unsharded set, and with the same fill the concatenated shard payloads are the
// unsharded payload (tests/test_gpu_parallel.py: bit for bit in the cache's fp16-stepwise arithmetic).
#include "common.h"
#include "rowgeom.h"

namespace {

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// one wave per row, 4 rows per workgroup.  LDS per wave: len composites (32 bit: key << 12 | 4095 - local column).
__global__ __launch_bounds__(256) void vsel_cand_kernel(const uint16_t* __restrict__ x, RowGeom gm, int64_t n_rows, int len, int k,
                                                        int col0, unsigned long long* __restrict__ cand) {
    extern __shared__ uint32_t vs_lds[];
    __shared__ int slot_ctr[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rows) return;
    uint32_t* comp = vs_lds + (size_t)wave * len;
    const int64_t row_base = row_base_of(gm, r);
    const int E = len >> 6;                                 // elements per lane (len is a multiple of 64)
    double sum = 0.0;
    for (int i = 0; i < E; i++) {
        const int j = i * 64 + lane;
        int seg, pos;
        seg_pos(gm, j, seg, pos);
        const uint16_t hb = x[row_base + (int64_t)seg * gm.seg_stride + pos];
        sum += (double)h2f_bits(hb);
        comp[j] = (sort_key(hb) << 12) | (uint32_t)(4095 - j);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane < 2) slot_ctr[wave][lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unsigned long long* out = cand + r * (int64_t)(2 * k + 1);
    for (int side = 0; side < 2; side++) {                  // 0: the k largest values, 1: the k smallest
        auto cof = [&](uint32_t c) { return side == 0 ? c : (((0xFFFFu - (c >> 12)) << 12) | (c & 0xFFFu)); };
        uint32_t lo_b = 0u, hi_b = 0x0FFFFFFFu;             // largest T with count(c >= T) >= k
        for (int it = 0; it < 28; it++) {
            const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
            int cnt = 0;
            for (int i = 0; i < E; i++) cnt += (cof(comp[i * 64 + lane]) >= mid) ? 1 : 0;
            cnt = wave_sum_i32(cnt);
            if (cnt >= k) lo_b = mid; else hi_b = mid - 1u;
        }
        for (int i = 0; i < E; i++) {
            const int j = i * 64 + lane;
            const uint32_t c = cof(comp[j]);
            if (c >= lo_b) {
                const int s = atomicAdd(&slot_ctr[wave][side], 1);
                // global composite: 16-bit order key of the side, then "lower GLOBAL column first" in 20 bits
                if (s < k) out[side * k + s] = ((unsigned long long)(c >> 12) << 20) | (unsigned long long)(0xFFFFF - (col0 + j));
            }
        }
    }
    if (lane == 0) out[2 * k] = (unsigned long long)__double_as_longlong(sum);
}

// one wave per row: cand_all [world][n_rows][2k + 1] -> thr [n_rows][2] (large side, small side), fill [n_rows]
__global__ __launch_bounds__(256) void vsel_thr_kernel(const unsigned long long* __restrict__ cand_all, int world, int64_t n_rows, int k,
                                                       double len_total, int mode, unsigned long long* __restrict__ thr,
                                                       float* __restrict__ fill) {
    extern __shared__ unsigned long long vt_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rows) return;
    const int n = world * k;
    unsigned long long* c = vt_lds + (size_t)wave * n;
    const int64_t pitch = 2 * k + 1;
    for (int side = 0; side < 2; side++) {
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n; i += 64) c[i] = cand_all[((int64_t)(i / k) * n_rows + r) * pitch + side * k + i % k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        unsigned long long lo_b = 0ull, hi_b = (1ull << 36) - 1ull;
        for (int it = 0; it < 36; it++) {
            const unsigned long long mid = lo_b + ((hi_b - lo_b + 1ull) >> 1);
            int cnt = 0;
            for (int i = lane; i < n; i += 64) cnt += (c[i] >= mid) ? 1 : 0;
            cnt = wave_sum_i32(cnt);
            if (cnt >= k) lo_b = mid; else hi_b = mid - 1ull;
        }
        if (lane == 0) thr[r * 2 + side] = lo_b;
    }
    if (lane == 0) {
        double tot = 0.0;                                    // (every rank's sum is exact in fp64, so is the total: any order)
        for (int w = 0; w < world; w++) tot += __longlong_as_double((long long)cand_all[((int64_t)w * n_rows + r) * pitch + 2 * k]);
        const float m = (float)(tot / len_total);             // (the row kernels' and the oracle's rounding: fp64 quotient -> float)
        fill[r] = (mode == 0) ? hround(m) : m;
    }
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
    GEAR_CHECK_ARG(col0 >= 0 && col0 + len <= 0xFFFFF, "gear_vsel_candidates: global column out of range");
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) l++; return l; };
    RowGeom gm{rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride, gear_is_pow2(seglen) ? ilog2(seglen) : -1, 0,
               outer_stride, inner_stride, seg_stride, rows_inner};
    const size_t lds = (size_t)4 * len * 4;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)vsel_cand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(vsel_cand_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)x, gm,
                       n_rows, (int)len, k, col0, (unsigned long long*)cand);
    GEAR_CHECK_LAUNCH("gear_vsel_candidates");
    return 0;
}

extern "C" int gear_vsel_thresholds(const void* cand_all, int world, int64_t n_rows, int k, int64_t row_len_total, int mode, void* thr,
                                    void* fill, void* stream) {
    GEAR_CHECK_ARG(cand_all && thr && fill && world >= 1 && n_rows > 0 && k > 0 && row_len_total > 0, "gear_vsel_thresholds: bad arguments");
    GEAR_CHECK_ARG((int64_t)world * k <= 4096, "gear_vsel_thresholds: world * k = %lld candidates per side exceed 4096", (long long)world * k);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_vsel_thresholds: bad mode %d", mode);
    const size_t lds = (size_t)4 * world * k * 8;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)vsel_thr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(vsel_thr_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), lds, (hipStream_t)stream,
                       (const unsigned long long*)cand_all, world, n_rows, k, (double)row_len_total, mode,
                       (unsigned long long*)thr, (float*)fill);
    GEAR_CHECK_LAUNCH("gear_vsel_thresholds");
    return 0;
}
