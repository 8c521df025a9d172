// decompress_rows.hip -- packed codes + low-rank factors + sparse outliers -> fp16 rows (gfx950).
//
//   out = fp16( fp16(dequant(code)) + sum_c Q[t,c] P[d,c] ),   outlier positions: out = fp16( value + sum_c ... )
//
// which is how the simulated path assembles its result (GenerationBench/.../Simulated/compress_function.py:204-220:
// `output` already holds the restored outliers and is fp16; `output + error_lr` in fp32; the dispatcher's .half()).
// Rows and segments are described exactly as in compress_rows.hip.
//   kind 0 (V):   row = (b, t), element j -> head j / seglen, channel j % seglen
//   kind 1 (K^T): row = (bh, d), element j -> token j
#include "common.h"

namespace {

template <int N>
__device__ __forceinline__ void load_halfs(const uint16_t* p, float* f) {
    if (N == 8) {
        uint4 v = *(const uint4*)p;
        unpack8(v, f);
    } else if (N == 16) {
        uint4 a = ((const uint4*)p)[0], b = ((const uint4*)p)[1];
        unpack8(a, f);
        unpack8(b, f + 8);
    } else if (N == 4) {
        uint2 v = *(const uint2*)p;
        f[0] = h2f_bits((uint16_t)(v.x & 0xFFFFu)); f[1] = h2f_bits((uint16_t)(v.x >> 16));
        f[2] = h2f_bits((uint16_t)(v.y & 0xFFFFu)); f[3] = h2f_bits((uint16_t)(v.y >> 16));
    }
}

template <int BITS, int MODE, typename ST, int KIND, int RV>
__global__ void decompress_rows_kernel(const uint32_t* __restrict__ code, const ST* __restrict__ scale,
                                       const ST* __restrict__ mn, int rows_inner, int64_t outer_stride,
                                       int64_t inner_stride, int nseg, int seglen, int64_t seg_stride, int len,
                                       int group, const uint16_t* __restrict__ P, const uint16_t* __restrict__ Q, int r,
                                       int T, int D, const uint16_t* __restrict__ oidx,
                                       const uint16_t* __restrict__ oval, int k, uint16_t* __restrict__ out) {
    constexpr int WPL = BITS / 2;
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    extern __shared__ uint32_t lds[];  // [len/32] mask words, then len uint16 values
    uint32_t* lmask = lds;
    uint16_t* lval = (uint16_t*)(lds + (len + 31) / 32);
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const int j0 = tid * 16;
    const bool active = j0 < len;
    if (k > 0) {
        for (int i = tid; i < (len + 31) / 32; i += blockDim.x) lmask[i] = 0u;
        __syncthreads();
        for (int i = tid; i < 2 * k; i += blockDim.x) {
            uint32_t idx = oidx[row * (int64_t)(2 * k) + i];
            atomicOr(&lmask[idx >> 5], 1u << (idx & 31));
            lval[idx] = oval[row * (int64_t)(2 * k) + i];
        }
        __syncthreads();
    }
    if (!active) return;
    const int seg = j0 / seglen, pos = j0 % seglen;
    const int ro = (int)(row / rows_inner), ri = (int)(row % rows_inner);
    const int64_t off = (int64_t)ro * outer_stride + (int64_t)ri * inner_stride + (int64_t)seg * seg_stride + pos;
    const int64_t g = off / group;
    const float s = ld_st<ST>(scale + g), m = ld_st<ST>(mn + g);
    float f[16];
#pragma unroll
    for (int w = 0; w < WPL; w++) {
        uint32_t word = code[off / CPW + w];
#pragma unroll
        for (int j = 0; j < CPW; j++) {
            float d = dequant_one<MODE>((int)((word >> (BITS * j)) & MASK), s, m);
            f[w * CPW + j] = (MODE == 0) ? d : hround(d);
        }
    }
    if (k > 0) {
        uint32_t mbits = (lmask[j0 >> 5] >> (j0 & 31)) & 0xFFFFu;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (mbits & (1u << j)) f[j] = h2f_bits(lval[j0 + j]);
    }
    if (r > 0) {
        // fixed vector fv[r] and a contiguous 16 x r block gb
        const uint16_t *fvp, *gbp;
        if (KIND == 0) {  // row = (b, t): bh = b * nseg + seg; fixed = Q[bh, t, :], block = P[bh, pos.., :]
            const int64_t bh = (int64_t)ro * nseg + seg;
            fvp = Q + (bh * T + ri) * r;
            gbp = P + (bh * D + pos) * r;
        } else {          // row = (bh, d): fixed = P[bh, d, :], block = Q[bh, j0.., :]
            fvp = P + ((int64_t)ro * D + ri) * r;
            gbp = Q + ((int64_t)ro * T + j0) * r;
        }
        if (RV > 0) {   // r == RV: factor rows are RV contiguous fp16 -> vector loads, fully unrolled
            float fv[RV > 0 ? RV : 1];
            load_halfs<RV>(fvp, fv);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                float gv[RV > 0 ? RV : 1];
                load_halfs<RV>(gbp + j * RV, gv);
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < RV; c++) acc = fmaf(fv[c], gv[c], acc);
                f[j] += acc;
            }
        } else {
            float fv[16];
            for (int c = 0; c < r; c++) fv[c] = h2f_bits(fvp[c]);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                float acc = 0.0f;
                for (int c = 0; c < r; c++) acc = fmaf(fv[c], h2f_bits(gbp[j * r + c]), acc);
                f[j] += acc;
            }
        }
    }
    uint4* op = (uint4*)(out + off);
    op[0] = pack8(f);
    op[1] = pack8(f + 8);
}

}  // namespace

extern "C" int gear_decompress_rows(const void* code, const void* scale, const void* mn, int64_t n_rows, int rows_inner,
                                    int64_t outer_stride, int64_t inner_stride, int nseg, int seglen, int64_t seg_stride,
                                    int group, int bits, int mode, int kind, const void* P, const void* Q, int r, int T,
                                    int D, const void* oidx, const void* oval, int k, void* out, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_decompress_rows: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_decompress_rows: bad mode %d", mode);
    GEAR_CHECK_ARG(kind == 0 || kind == 1, "gear_decompress_rows: bad kind %d", kind);
    GEAR_CHECK_ARG(n_rows > 0 && n_rows < 0x7FFFFFFFLL && nseg > 0 && seglen > 0, "gear_decompress_rows: empty input");
    const int64_t len = (int64_t)nseg * seglen;
    GEAR_CHECK_ARG(len <= 16384 && len % 16 == 0, "gear_decompress_rows: bad row length %lld", (long long)len);
    GEAR_CHECK_ARG(group >= 16 && group % 16 == 0 && seglen % group == 0, "gear_decompress_rows: bad group %d", group);
    GEAR_CHECK_ARG(r >= 0 && r <= 16, "gear_decompress_rows: rank must be in [0,16]");
    GEAR_CHECK_ARG(r == 0 || (P && Q), "gear_decompress_rows: low-rank factors missing");
    GEAR_CHECK_ARG(k >= 0 && (k == 0 || (oidx && oval)), "gear_decompress_rows: outlier buffers missing");
    GEAR_CHECK_ARG(code && scale && mn && out, "gear_decompress_rows: null pointer");
    if (kind == 0) GEAR_CHECK_ARG(rows_inner == T && seglen == D, "gear_decompress_rows: kind 0 needs rows_inner == T and seglen == D");
    if (kind == 1) GEAR_CHECK_ARG(rows_inner == D && nseg == 1 && seglen == T, "gear_decompress_rows: kind 1 needs rows_inner == D, one segment of T");
    int threads = (int)((len / 16 + 63) / 64 * 64);
    size_t shmem = k > 0 ? (size_t)((len + 31) / 32) * 4 + (size_t)len * 2 : 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 block(threads), grid((unsigned)n_rows);
#define GO(B, M, STT, KD, RVV)                                                                                         \
    hipLaunchKernelGGL((decompress_rows_kernel<B, M, STT, KD, RVV>), grid, block, shmem, st, (const uint32_t*)code,          \
                       (const STT*)scale, (const STT*)mn, rows_inner, outer_stride, inner_stride, nseg, seglen,         \
                       seg_stride, (int)len, group, (const uint16_t*)P, (const uint16_t*)Q, r, T, D,                    \
                       (const uint16_t*)oidx, (const uint16_t*)oval, k, (uint16_t*)out)
#define GOR(B, M, STT, KD) do { if (r == 8) GO(B, M, STT, KD, 8); else if (r == 4) GO(B, M, STT, KD, 4); \
                                else if (r == 16) GO(B, M, STT, KD, 16); else GO(B, M, STT, KD, 0); } while (0)
#define GOK(B, M, STT) do { if (kind == 0) GOR(B, M, STT, 0); else GOR(B, M, STT, 1); } while (0)
    if (mode == 0) {
        if (bits == 2) GOK(2, 0, uint16_t);
        else if (bits == 4) GOK(4, 0, uint16_t);
        else GOK(8, 0, uint16_t);
    } else {
        if (bits == 2) GOK(2, 1, float);
        else if (bits == 4) GOK(4, 1, float);
        else GOK(8, 1, float);
    }
#undef GOK
#undef GOR
#undef GO
    GEAR_CHECK_LAUNCH("gear_decompress_rows");
    return 0;
}
