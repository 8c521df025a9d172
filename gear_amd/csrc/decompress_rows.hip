// decompress_rows.hip -- gear_decompress_rows: the C entry point + the fp16-arithmetic instantiations of the row decompressor
// (decompress_rows_impl.h holds the kernel; the fp32-arithmetic instantiations are compiled in decompress_rows_m1.hip).
#define DEC_PART 0
#define DEC_ENTRY gear_decompress_rows_m0
#include "decompress_rows_impl.h"

int gear_decompress_rows_m1(const void* code, const void* scale, const void* mn, int64_t n_rows, int rows_inner,
                                    int64_t outer_stride, int64_t inner_stride, int nseg, int seglen, int64_t seg_stride,
                                    int group, int bits, int mode, int kind, const void* P, const void* Q, int r, int T,
                                    int D, const void* oidx, const void* oval, int k, void* out, void* stream);

extern "C" int gear_decompress_rows(const void* code, const void* scale, const void* mn, int64_t n_rows, int rows_inner,
                                    int64_t outer_stride, int64_t inner_stride, int nseg, int seglen, int64_t seg_stride,
                                    int group, int bits, int mode, int kind, const void* P, const void* Q, int r, int T,
                                    int D, const void* oidx, const void* oval, int k, void* out, void* stream) {
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_decompress_rows: bad mode %d", mode);
    return mode == 0 ? gear_decompress_rows_m0(code, scale, mn, n_rows, rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride, group, bits,
                                               mode, kind, P, Q, r, T, D, oidx, oval, k, out, stream)
                     : gear_decompress_rows_m1(code, scale, mn, n_rows, rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride, group, bits,
                                               mode, kind, P, Q, r, T, D, oidx, oval, k, out, stream);
}
