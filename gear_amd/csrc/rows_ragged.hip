// rows_ragged.hip -- gears_channelQ / gears_tokenQ for rows whose length is NOT a multiple of the quantization group: the
// reference's fake_groupwise_channel_asymmetric_quantization_cluster quantizes the first floor(len / g) * g elements of a row in
// groups of g and leaves the tail as it is (GenerationBench/GenerationTest/GEARLM/Simulated/compress_function.py:107-122), while
// the outlier selection and the fill value of gears_channelQ (:261-296) work on the WHOLE row: the k smallest / k largest of all
// len elements, replaced by the mean of all len elements for the quantization and restored afterwards.  That is every prompt of
// the simulated path whose length is not a multiple of the group (K rows = a channel over the tokens), and every decode step after
// it (the dispatcher re-compresses the whole cache, :421-584).
//
// Like rows_whole.hip this writes the quantize -> dequantize result (fp16) and optionally the error x - y for the low-rank step;
// there is no packed payload for a ragged row.  One workgroup per row, the row and its outlier bitmap in LDS, any length up to
// 16384 and any group >= 1 that keeps the scale / mn tables in LDS (at most 2048 groups).  Selection: 16 rounds of bisection on the
// 16-bit order key per side (block-wide counts), ties at the threshold "lower index first" by a scan over contiguous chunks --
// the same rule the row compressor and oracle/gear_oracle.c use.
#include "common.h"
#include "ktile.h"

namespace {

struct RGeom {
    int rows_inner;
    int64_t outer_stride, inner_stride;
    int nseg, seglen;
    int64_t seg_stride;
};

__device__ __forceinline__ int64_t r_elem_off(const RGeom& g, int64_t base, int j) {
    const int seg = j / g.seglen, pos = j - seg * g.seglen;
    return base + (int64_t)seg * g.seg_stride + pos;
}

constexpr int RR_MAX_GROUPS = 2048;

__device__ __forceinline__ int block_sum_i32(int v, int* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <int MODE>
__global__ __launch_bounds__(256) void quant_rows_ragged_kernel(const uint16_t* __restrict__ x, RGeom g, int len, int group, int q_len,
                                                                int levels, int k, uint16_t* __restrict__ y,
                                                                uint16_t* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t* xs = (uint16_t*)smem;                                     // [len] the row
    uint32_t* obit = (uint32_t*)(smem + (((size_t)len * 2 + 15) & ~(size_t)15));   // [(len + 31) / 32] outlier bitmap
    float* gsc = (float*)(obit + ((len + 31) / 32 + 3) / 4 * 4);        // [q_len / group] scale
    float* gmn = gsc + RR_MAX_GROUPS;                                    // [q_len / group] mn
    __shared__ int red[4];
    __shared__ double dred[4];
    __shared__ int chunk_cnt[256];

    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (r / g.rows_inner) * g.outer_stride + (r % g.rows_inner) * g.inner_stride;

    // ---- the row -> LDS; its exact sum (fp16 values add exactly in fp64)
    double ds = 0.0;
    for (int j = tid; j < len; j += 256) {
        const uint16_t b = x[r_elem_off(g, base, j)];
        xs[j] = b;
        ds += (double)h2f_bits(b);
    }
    for (int i = tid; i < (len + 31) / 32; i += 256) obit[i] = 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ds += __shfl_xor(ds, d, 64);
    if (lane == 0) dred[wave] = ds;
    __syncthreads();
    const float mean = (float)(((dred[0] + dred[1]) + (dred[2] + dred[3])) / (double)len);

    // ---- selection over the whole row: side 0 = the k largest, side 1 = the k smallest
    if (k > 0) {
        const int cs = (len + 255) / 256;                              // contiguous chunk of a thread for the tie scan
        for (int side = 0; side < 2; side++) {
            uint32_t lo_b = 0u, hi_b = 0xFFFFu;                        // largest K with count(key >= K) >= k
            for (int it = 0; it < 16; it++) {
                const uint32_t mid = (lo_b + hi_b + 1u) >> 1;
                int c = 0;
                for (int j = tid; j < len; j += 256) c += order_key(xs[j], side) >= mid ? 1 : 0;
                c = block_sum_i32(c, red);
                if (c >= k) lo_b = mid; else hi_b = mid - 1u;
            }
            const uint32_t kth = lo_b;
            int above = 0, ties = 0;
            for (int j = tid; j < len; j += 256) {
                const uint32_t kx = order_key(xs[j], side);
                if (kx > kth) { above++; atomicOr(&obit[j >> 5], 1u << (j & 31)); }
            }
            above = block_sum_i32(above, red);
            const int need = k - above;                                // ties to take, lowest index first (>= 1)
            const int j0 = tid * cs, j1 = min(len, j0 + cs);
            for (int j = j0; j < j1; j++) ties += order_key(xs[j], side) == kth ? 1 : 0;
            chunk_cnt[tid] = ties;
            __syncthreads();
            int before = 0;
            for (int t = 0; t < tid; t++) before += chunk_cnt[t];
            for (int j = j0; j < j1 && before < need; j++) {
                if (order_key(xs[j], side) == kth) {
                    atomicOr(&obit[j >> 5], 1u << (j & 31));
                    before++;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();

    // ---- group parameters of the quantized prefix: outliers enter min / max as the fill value (:276-283 then :111-112)
    const float fill = (MODE == 0) ? hround(mean) : mean;
    const int ngroups = q_len / group;
    for (int m = tid; m < ngroups; m += 256) {
        float lo = INFINITY, hi = -INFINITY;
        for (int j = m * group; j < (m + 1) * group; j++) {
            const float v = ((obit[j >> 5] >> (j & 31)) & 1u) ? fill : h2f_bits(xs[j]);
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
        const QuantParams<MODE> qp = make_qparams<MODE>(lo, hi, levels);
        gsc[m] = qp.scale;
        gmn[m] = qp.mn;
    }
    __syncthreads();

    // ---- quantize -> dequantize the prefix; outliers and the tail keep their original value (error 0 there)
    for (int j = tid; j < len; j += 256) {
        const int64_t o = r_elem_off(g, base, j);
        const uint16_t xb = xs[j];
        uint16_t yb = xb;
        float e = 0.0f;
        if (j < q_len && !((obit[j >> 5] >> (j & 31)) & 1u)) {
            QuantParams<MODE> qp;
            qp.scale = gsc[j / group];
            qp.mn = gmn[j / group];
            qp.levels = levels;
            const float v = h2f_bits(xb);
            const int q = quant_one<MODE>(v, qp);
            const float d = (MODE == 0) ? dequant_one<0>(q, qp.scale, qp.mn) : hround(dequant_one<1>(q, qp.scale, qp.mn));
            yb = f2h_bits(d);
            e = v - d;
        }
        y[o] = yb;
        if (err) err[o] = f2h_bits(e);
    }
}

}  // namespace

extern "C" int gear_quant_rows_ragged(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride,
                                      int nseg, int seglen, int64_t seg_stride, int group, int bits, int mode, int k, void* y,
                                      void* err, void* stream) {
    GEAR_CHECK_ARG(x && y, "gear_quant_rows_ragged: null pointer");
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_quant_rows_ragged: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_quant_rows_ragged: bad mode %d", mode);
    GEAR_CHECK_ARG(n_rows > 0 && n_rows < 0x7FFFFFFFLL && rows_inner > 0 && nseg > 0 && seglen > 0, "gear_quant_rows_ragged: empty input");
    const int64_t len = (int64_t)nseg * seglen;
    GEAR_CHECK_ARG(len <= 16384, "gear_quant_rows_ragged: row length %lld exceeds 16384", (long long)len);
    GEAR_CHECK_ARG(group >= 1, "gear_quant_rows_ragged: group %d", group);
    const int q_len = (int)(len / group) * group;
    GEAR_CHECK_ARG(q_len / group <= RR_MAX_GROUPS, "gear_quant_rows_ragged: %d groups per row (at most %d)", q_len / group, RR_MAX_GROUPS);
    GEAR_CHECK_ARG(k >= 0 && 2 * (int64_t)k <= len, "gear_quant_rows_ragged: k=%d out of range for row length %lld", k, (long long)len);
    RGeom g{rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride};
    hipStream_t st = (hipStream_t)stream;
    const size_t shmem = (((size_t)len * 2 + 15) & ~(size_t)15) + (size_t)(((len + 31) / 32 + 3) / 4 * 4) * 4 + 2u * RR_MAX_GROUPS * 4;
    const dim3 grid((unsigned)n_rows), block(256);
    const int levels = (1 << bits) - 1;
    if (mode == 0)
        hipLaunchKernelGGL(quant_rows_ragged_kernel<0>, grid, block, shmem, st, (const uint16_t*)x, g, (int)len, group, q_len, levels, k,
                           (uint16_t*)y, (uint16_t*)err);
    else
        hipLaunchKernelGGL(quant_rows_ragged_kernel<1>, grid, block, shmem, st, (const uint16_t*)x, g, (int)len, group, q_len, levels, k,
                           (uint16_t*)y, (uint16_t*)err);
    GEAR_CHECK_LAUNCH("gear_quant_rows_ragged");
    return 0;
}
