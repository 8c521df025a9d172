// rows_whole.hip -- group quantization with ONE group per row (the KCVT variants of the simulated dispatcher): the group
// spans the whole sequence of a channel (K: fake_groupwise_channel_asymmetric_quantization_new(key, bits, seq_len),
// GenerationBench/GenerationTest/GEARLM/Simulated/compress_function.py:441-446, :496-510, :555-568) or a whole token row
// across all heads (V: fake_groupwise_token_asymmetric_quantization(value, bits, num_head * sep_dim), :447-452, :511-525,
// :569-582) -- optionally around the sparse outliers of gears_channelQ / gears_tokenQ (:261-333): the k smallest / k largest
// of the row are replaced by the row mean for the quantization and restored afterwards.
//
// The simulated path returns the quantize -> dequantize result, so this kernel writes that (fp16) plus, optionally, the
// error x - y for the low-rank step; there is no packed payload for these variants.  One workgroup per row, two passes over
// the row (min / max, then quantize): rows are at most 16384 elements and come back from L2.
#include "common.h"

namespace {

struct WGeom {
    int rows_inner;
    int64_t outer_stride, inner_stride;
    int nseg, seglen;
    int64_t seg_stride;
};

__device__ __forceinline__ int64_t elem_off(const WGeom& g, int64_t base, int j) {
    const int seg = j / g.seglen, pos = j - seg * g.seglen;
    return base + (int64_t)seg * g.seg_stride + pos;
}

template <int BITS, int MODE>
__global__ __launch_bounds__(256) void quant_rows_whole_kernel(const uint16_t* __restrict__ x, WGeom g, int len,
                                                               const uint16_t* __restrict__ oidx, int k,
                                                               uint16_t* __restrict__ y, uint16_t* __restrict__ err) {
    constexpr int LEVELS = (1 << BITS) - 1;
    __shared__ uint32_t obit[512];                 // outlier bitmap of the row (len <= 16384)
    __shared__ float red[3][4];
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (r / g.rows_inner) * g.outer_stride + (r % g.rows_inner) * g.inner_stride;
    for (int i = tid; i < (len + 31) / 32; i += 256) obit[i] = 0u;
    __syncthreads();
    for (int i = tid; i < 2 * k; i += 256) {          // any list length (the reference has no limit on it)
        const uint32_t j = oidx[r * (int64_t)(2 * k) + i];
        atomicOr(&obit[j >> 5], 1u << (j & 31u));
    }
    __syncthreads();
    // ---- pass 1: sum of the ORIGINAL row (the fill value is its mean, :276, :312), min / max of the non-outliers
    float s = 0.0f, lo = INFINITY, hi = -INFINITY;
    for (int j = tid; j < len; j += 256) {
        const float v = h2f_bits(x[elem_off(g, base, j)]);
        s += v;
        if (!((obit[j >> 5] >> (j & 31)) & 1u)) { lo = fminf(lo, v); hi = fmaxf(hi, v); }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d, 64);
        lo = fminf(lo, __shfl_xor(lo, d, 64));
        hi = fmaxf(hi, __shfl_xor(hi, d, 64));
    }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = lo; red[2][wave] = hi; }
    __syncthreads();
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    lo = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
    hi = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    if (k > 0) {
        const float mean = s / (float)len;
        const float fill = (MODE == 0) ? hround(mean) : mean;
        lo = fminf(lo, fill);
        hi = fmaxf(hi, fill);
    }
    const QuantParams<MODE> qp = make_qparams<MODE>(lo, hi, LEVELS);
    // ---- pass 2: quantize -> dequantize; outliers keep their original value (restored, :290-291 / :326-327), error 0 there
    for (int j = tid; j < len; j += 256) {
        const int64_t o = elem_off(g, base, j);
        const uint16_t xb = x[o];
        uint16_t yb;
        float e = 0.0f;
        if ((obit[j >> 5] >> (j & 31)) & 1u) {
            yb = xb;
        } else {
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

extern "C" int gear_quant_rows_whole(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride,
                                     int64_t inner_stride, int nseg, int seglen, int64_t seg_stride, int bits, int mode,
                                     const void* oidx, int k, void* y, void* err, void* stream) {
    GEAR_CHECK_ARG(x && y, "gear_quant_rows_whole: null pointer");
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_quant_rows_whole: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_quant_rows_whole: bad mode %d", mode);
    GEAR_CHECK_ARG(n_rows > 0 && n_rows < 0x7FFFFFFFLL && rows_inner > 0 && nseg > 0 && seglen > 0, "gear_quant_rows_whole: empty input");
    const int64_t len = (int64_t)nseg * seglen;
    GEAR_CHECK_ARG(len <= 16384, "gear_quant_rows_whole: row length %lld exceeds 16384", (long long)len);
    GEAR_CHECK_ARG(k >= 0 && 2 * (int64_t)k <= len && (k == 0 || oidx), "gear_quant_rows_whole: bad outlier count %d", k);
    WGeom g{rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n_rows), block(256);
#define GO(B, M) hipLaunchKernelGGL((quant_rows_whole_kernel<B, M>), grid, block, 0, st, (const uint16_t*)x, g, (int)len, \
                                    (const uint16_t*)oidx, k, (uint16_t*)y, (uint16_t*)err)
    if (mode == 0) { if (bits == 2) GO(2, 0); else if (bits == 4) GO(4, 0); else GO(8, 0); }
    else { if (bits == 2) GO(2, 1); else if (bits == 4) GO(4, 1); else GO(8, 1); }
#undef GO
    GEAR_CHECK_LAUNCH("gear_quant_rows_whole");
    return 0;
}
