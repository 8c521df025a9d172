// rowgeom.h -- row geometry of the row compressors (compress_rows.hip, rows_multi.hip): a row is nseg segments of seglen
// contiguous fp16 elements, rows are addressed (outer, inner) with strides; payload tensors may have their own strides.
#pragma once
#include "common.h"

namespace {

struct RowGeom {
    int rows_inner;            // row r -> (r / rows_inner, r % rows_inner)
    int64_t outer_stride;      // elements
    int64_t inner_stride;      // elements
    int nseg, seglen;          // row = nseg segments of seglen contiguous elements
    int64_t seg_stride;        // elements
    int seglen_shift;          // log2(seglen) if it is a power of two, else -1
    int group_shift;           // log2(group) (group is a power of two)
    // geometry of code / scale / mn (elements of the fp16 tensor they describe): equal to the input's unless the payload is
    // written in place into a larger pre-allocated tensor (the streaming cache); the error output follows the input.
    int64_t o_outer_stride, o_inner_stride, o_seg_stride;
    int o_list_outer;          // row r's sparse list is list row (r / rows_inner) * o_list_outer + r % rows_inner
};

// integer divisions by run-time values cost ~30-100 VALU instructions per lane on this VALU-bound kernel: every
// geometry quotient goes through shifts (power-of-two group / segment length) or 32-bit scalar math (row index)
__device__ __forceinline__ void seg_pos(const RowGeom& gm, int j, int& seg, int& pos) {
    if (gm.nseg == 1) { seg = 0; pos = j; }
    else if (gm.seglen_shift >= 0) { seg = j >> gm.seglen_shift; pos = j & (gm.seglen - 1); }
    else { seg = j / gm.seglen; pos = j % gm.seglen; }
}
__device__ __forceinline__ int64_t row_base_of(const RowGeom& gm, int64_t r) {
    const uint32_t ru = (uint32_t)r, ri = (uint32_t)gm.rows_inner;   // n_rows < 2^31 (checked on the host)
    const uint32_t qo = ru / ri;
    return (int64_t)qo * gm.outer_stride + (int64_t)(ru - qo * ri) * gm.inner_stride;
}

__device__ __forceinline__ int64_t lrow_of(const RowGeom& gm, int64_t r) {
    const uint32_t ru = (uint32_t)r, ri = (uint32_t)gm.rows_inner;
    const uint32_t qo = ru / ri;
    return (int64_t)qo * gm.o_list_outer + (int64_t)(ru - qo * ri);
}
__device__ __forceinline__ int64_t row_base_out(const RowGeom& gm, int64_t r) {
    const uint32_t ru = (uint32_t)r, ri = (uint32_t)gm.rows_inner;
    const uint32_t qo = ru / ri;
    return (int64_t)qo * gm.o_outer_stride + (int64_t)(ru - qo * ri) * gm.o_inner_stride;
}

__device__ __forceinline__ uint32_t sort_key(uint32_t hbits) {  // fp16 bits -> ascending-order key (16 bit)
    if (hbits == 0x8000u) hbits = 0u;  // -0 == +0 (the oracle / torch.topk compare values)
    return (hbits & 0x8000u) ? (~hbits & 0xFFFFu) : (hbits | 0x8000u);
}
__device__ __forceinline__ uint32_t key_to_bits(uint32_t key) {  // inverse of sort_key
    return (key & 0x8000u) ? (key & 0x7FFFu) : (~key & 0xFFFFu);
}

template <int BITS, int MODE>
__device__ __forceinline__ int quant_fast(float v, float mn, float scale, float inv, int levels) {
    if (scale == 0.0f) return 0;
    if (MODE == 0) {
        float t1 = hround(v - mn);
        float c = hround(div_rn(t1, scale));
        c = fminf(fmaxf(c, 0.0f), (float)levels);
        return (int)rintf(c);
    } else {
        // (v - mn) / scale with an IEEE-exact result: multiply by the reciprocal, and redo the division only when the
        // approximate quotient is within 1e-5 of a rounding tie (x.5) -- the only place the two can round differently.
        float t = v - mn;
        float c = t * inv;
        float r = rintf(c);
        if (fabsf(fabsf(c - r) - 0.5f) < (BITS == 8 ? 1e-3f : 1e-5f)) {   // (8-bit quotients reach 255: 1 ulp is 3e-5)
            c = div_rn(t, scale);
            r = rintf(c);
        }
        r = fminf(fmaxf(r, 0.0f), (float)levels);
        return (int)r;
    }
}

}  // namespace
