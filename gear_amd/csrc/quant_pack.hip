// quant_pack.hip -- group-wise asymmetric quantize + bit-pack, and unpack + dequantize (gfx950).
//
// HBM-bound byte work: every lane moves 16-byte vectors, min/max are reduced with wave shuffles,
// codes never round-trip memory as int32 (the reference materialises an int32 code tensor and runs ~8 eager
// elementwise passes: cuda_supported_gear/quant/new_pack.py:237-246).
#include "common.h"

// =====================================================================================================
// K1: along-last-dim quantizer.  Flat view: element e = row*L + j; lane i owns elements [16 i, 16 i + 16).
// Lanes of one group are consecutive and group/16 is a power of two, so the group reduction is an
// xor-shuffle butterfly of width group/16.
// =====================================================================================================
template <int BITS, int MODE, typename ST>
__global__ __launch_bounds__(256) void quant_pack_lastdim_kernel(const uint4* __restrict__ x, int64_t n_lanes,
                                                                 int lanes_per_group, uint32_t* __restrict__ code,
                                                                 ST* __restrict__ scale, ST* __restrict__ mn,
                                                                 uint4* __restrict__ err) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int WPL = BITS / 2;  // int32 words produced per lane (16 codes)
    constexpr int CPW = 32 / BITS; // codes per word
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n_lanes;
    int64_t ii = active ? i : (n_lanes - 1);  // keep every lane in the shuffles
    float v[16];
    uint4 a = x[2 * ii], b = x[2 * ii + 1];
    unpack8(a, v);
    unpack8(b, v + 8);
    float lo = v[0], hi = v[0];
#pragma unroll
    for (int j = 1; j < 16; j++) {
        lo = fminf(lo, v[j]);
        hi = fmaxf(hi, v[j]);
    }
    for (int m = 1; m < lanes_per_group; m <<= 1) {
        lo = fminf(lo, __shfl_xor(lo, m, GEAR_WAVE));
        hi = fmaxf(hi, __shfl_xor(hi, m, GEAR_WAVE));
    }
    QuantParams<MODE> qp = make_qparams<MODE>(lo, hi, LEVELS);
    uint32_t words[WPL];
#pragma unroll
    for (int w = 0; w < WPL; w++) words[w] = 0u;
    float e[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        int q = quant_one<MODE>(v[j], qp);
        words[j / CPW] |= (uint32_t)q << (BITS * (j % CPW));
        if (MODE == 0) {
            e[j] = v[j] - dequant_one<0>(q, qp.scale, qp.mn);  // rounded to fp16 by pack8
        } else {
            e[j] = v[j] - hround(dequant_one<1>(q, qp.scale, qp.mn));
        }
    }
    if (!active) return;
#pragma unroll
    for (int w = 0; w < WPL; w++) code[i * WPL + w] = words[w];
    if ((i & (lanes_per_group - 1)) == 0) {
        int64_t g = i / lanes_per_group;
        st_st<ST>(scale + g, qp.scale);
        st_st<ST>(mn + g, qp.mn);
    }
    if (err) {
        err[2 * i] = pack8(e);
        err[2 * i + 1] = pack8(e + 8);
    }
}

template <int BITS, int MODE>
static int launch_quant_pack_lastdim(const void* x, int64_t n_lanes, int lpg, void* code, void* scale, void* mn,
                                     void* err, hipStream_t st) {
    using ST = typename std::conditional<MODE == 0, uint16_t, float>::type;
    dim3 block(256), grid((unsigned)((n_lanes + 255) / 256));
    hipLaunchKernelGGL((quant_pack_lastdim_kernel<BITS, MODE, ST>), grid, block, 0, st, (const uint4*)x, n_lanes, lpg,
                       (uint32_t*)code, (ST*)scale, (ST*)mn, (uint4*)err);
    GEAR_CHECK_LAUNCH("gear_quant_pack_lastdim");
    return 0;
}

extern "C" int gear_quant_pack_lastdim(const void* x, int64_t rows, int L, int group, int bits, int mode, void* code,
                                       void* scale, void* mn, void* err, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_quant_pack_lastdim: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_quant_pack_lastdim: bad mode %d", mode);
    GEAR_CHECK_ARG(rows > 0 && L > 0, "gear_quant_pack_lastdim: empty input");
    GEAR_CHECK_ARG(group > 0 && L % group == 0, "gear_quant_pack_lastdim: last dim %d not divisible by group %d", L, group);
    GEAR_CHECK_ARG(group % 16 == 0 && gear_is_pow2(group / 16) && group <= 1024,
                   "gear_quant_pack_lastdim: group must be a power of two in [16,1024] (got %d)", group);
    GEAR_CHECK_ARG(x && code && scale && mn, "gear_quant_pack_lastdim: null pointer");
    int64_t n_lanes = rows * (int64_t)L / 16;
    GEAR_CHECK_ARG((n_lanes + 255) / 256 < 0x7FFFFFFFLL, "gear_quant_pack_lastdim: tensor too large");
    int lpg = group / 16;
    hipStream_t st = (hipStream_t)stream;
#define GO(B, M) return launch_quant_pack_lastdim<B, M>(x, n_lanes, lpg, code, scale, mn, err, st)
    if (mode == 0) {
        if (bits == 2) GO(2, 0);
        if (bits == 4) GO(4, 0);
        GO(8, 0);
    } else {
        if (bits == 2) GO(2, 1);
        if (bits == 4) GO(4, 1);
        GO(8, 1);
    }
#undef GO
}

// =====================================================================================================
// K2: token-major K tile [bh, T, D]; groups of `group` consecutive tokens per channel; packed along T.
// One block = one (bh, token-group).  Thread (cl, w): channel chunk cl (8 channels, one 16-byte vector per
// token row) x word w of the group (CPW consecutive tokens).  Loads are 256-byte row segments (D = 128);
// per-channel min/max over the group goes through LDS; each thread emits 8 finished words (32 bytes).
// =====================================================================================================
template <int BITS, int MODE, typename ST>
__global__ void quant_pack_k_kernel(const uint4* __restrict__ x, int T, int D, int group, uint32_t* __restrict__ code,
                                    ST* __restrict__ scale, ST* __restrict__ mn, uint4* __restrict__ err) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int CPW = 32 / BITS;
    extern __shared__ float smem[];  // [2][wpg][D]
    const int D8 = D / 8;
    const int wpg = group / CPW;  // words (threads along T) per group
    const int cl = threadIdx.x % D8;
    const int w = threadIdx.x / D8;
    const int ng = T / group;
    const int64_t bh = blockIdx.x / ng;
    const int G = blockIdx.x % ng;
    const int t0 = G * group + w * CPW;
    const uint4* xb = x + (bh * T + t0) * (int64_t)D8 + cl;

    uint4 raw[CPW];
#pragma unroll
    for (int j = 0; j < CPW; j++) raw[j] = xb[(int64_t)j * D8];
    float lo[8], hi[8];
    {
        float f[8];
        unpack8(raw[0], f);
#pragma unroll
        for (int c = 0; c < 8; c++) lo[c] = hi[c] = f[c];
#pragma unroll
        for (int j = 1; j < CPW; j++) {
            unpack8(raw[j], f);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                lo[c] = fminf(lo[c], f[c]);
                hi[c] = fmaxf(hi[c], f[c]);
            }
        }
    }
    float* slo = smem;
    float* shi = smem + wpg * D;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        slo[w * D + cl * 8 + c] = lo[c];
        shi[w * D + cl * 8 + c] = hi[c];
    }
    __syncthreads();
    for (int ww = 0; ww < wpg; ww++) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            lo[c] = fminf(lo[c], slo[ww * D + cl * 8 + c]);
            hi[c] = fmaxf(hi[c], shi[ww * D + cl * 8 + c]);
        }
    }
    QuantParams<MODE> qp[8];
#pragma unroll
    for (int c = 0; c < 8; c++) qp[c] = make_qparams<MODE>(lo[c], hi[c], LEVELS);
    uint32_t words[8];
#pragma unroll
    for (int c = 0; c < 8; c++) words[c] = 0u;
#pragma unroll
    for (int j = 0; j < CPW; j++) {
        float f[8], e[8];
        unpack8(raw[j], f);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            int q = quant_one<MODE>(f[c], qp[c]);
            words[c] |= (uint32_t)q << (BITS * j);
            if (MODE == 0)
                e[c] = f[c] - dequant_one<0>(q, qp[c].scale, qp[c].mn);
            else
                e[c] = f[c] - hround(dequant_one<1>(q, qp[c].scale, qp[c].mn));
        }
        if (err) err[(bh * T + t0 + j) * (int64_t)D8 + cl] = pack8(e);
    }
    // code [bh, T/CPW, D]: word row = G*wpg + w
    uint4* cw = (uint4*)(code + ((bh * (T / CPW) + (int64_t)G * wpg + w) * D + cl * 8));
    cw[0] = make_uint4(words[0], words[1], words[2], words[3]);
    cw[1] = make_uint4(words[4], words[5], words[6], words[7]);
    if (w == 0) {
        int64_t so = (bh * ng + G) * D + cl * 8;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            st_st<ST>(scale + so + c, qp[c].scale);
            st_st<ST>(mn + so + c, qp[c].mn);
        }
    }
}

template <int BITS, int MODE>
static int launch_quant_pack_k(const void* x, int64_t bh, int T, int D, int group, void* code, void* scale, void* mn,
                               void* err, hipStream_t st) {
    using ST = typename std::conditional<MODE == 0, uint16_t, float>::type;
    constexpr int CPW = 32 / BITS;
    int wpg = group / CPW;
    int threads = (D / 8) * wpg;
    size_t shmem = sizeof(float) * 2 * (size_t)wpg * D;
    dim3 block(threads), grid((unsigned)(bh * (T / group)));
    hipLaunchKernelGGL((quant_pack_k_kernel<BITS, MODE, ST>), grid, block, shmem, st, (const uint4*)x, T, D, group,
                       (uint32_t*)code, (ST*)scale, (ST*)mn, (uint4*)err);
    GEAR_CHECK_LAUNCH("gear_quant_pack_k");
    return 0;
}

extern "C" int gear_quant_pack_k(const void* x, int64_t bh, int T, int D, int group, int bits, int mode, void* code,
                                 void* scale, void* mn, void* err, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_quant_pack_k: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_quant_pack_k: bad mode %d", mode);
    GEAR_CHECK_ARG(bh > 0 && T > 0 && D > 0, "gear_quant_pack_k: empty input");
    GEAR_CHECK_ARG(D % 8 == 0, "gear_quant_pack_k: head_dim %d must be a multiple of 8", D);
    int cpw = 32 / bits;
    GEAR_CHECK_ARG(group > 0 && T % group == 0, "gear_quant_pack_k: T=%d not divisible by group %d", T, group);
    GEAR_CHECK_ARG(group % cpw == 0, "gear_quant_pack_k: group %d must be a multiple of %d", group, cpw);
    int threads = (D / 8) * (group / cpw);
    GEAR_CHECK_ARG(threads <= 1024, "gear_quant_pack_k: (D/8)*(group/fpi)=%d exceeds 1024 threads", threads);
    GEAR_CHECK_ARG(sizeof(float) * 2 * (size_t)(group / cpw) * D <= 64 * 1024, "gear_quant_pack_k: tile too large for LDS");
    GEAR_CHECK_ARG(bh * (int64_t)(T / group) < 0x7FFFFFFFLL, "gear_quant_pack_k: tensor too large");
    GEAR_CHECK_ARG(x && code && scale && mn, "gear_quant_pack_k: null pointer");
    hipStream_t st = (hipStream_t)stream;
#define GO(B, M) return launch_quant_pack_k<B, M>(x, bh, T, D, group, code, scale, mn, err, st)
    if (mode == 0) {
        if (bits == 2) GO(2, 0);
        if (bits == 4) GO(4, 0);
        GO(8, 0);
    } else {
        if (bits == 2) GO(2, 1);
        if (bits == 4) GO(4, 1);
        GO(8, 1);
    }
#undef GO
}

// =====================================================================================================
// K3: unpack + dequantize
// =====================================================================================================
template <int BITS, int MODE, typename ST>
__global__ __launch_bounds__(256) void unpack_dequant_lastdim_kernel(const uint32_t* __restrict__ code,
                                                                     const ST* __restrict__ scale,
                                                                     const ST* __restrict__ mn, int64_t n_lanes,
                                                                     int lanes_per_group, uint4* __restrict__ out) {
    constexpr int WPL = BITS / 2;
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lanes) return;
    int64_t g = i / lanes_per_group;
    float s = ld_st<ST>(scale + g), m = ld_st<ST>(mn + g);
    float f[16];
#pragma unroll
    for (int w = 0; w < WPL; w++) {
        uint32_t word = code[i * WPL + w];
#pragma unroll
        for (int j = 0; j < CPW; j++) f[w * CPW + j] = dequant_one<MODE>((int)((word >> (BITS * j)) & MASK), s, m);
    }
    out[2 * i] = pack8(f);
    out[2 * i + 1] = pack8(f + 8);
}

extern "C" int gear_unpack_dequant_lastdim(const void* code, const void* scale, const void* mn, int64_t rows, int L,
                                           int group, int bits, int mode, void* out, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_unpack_dequant_lastdim: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_unpack_dequant_lastdim: bad mode %d", mode);
    GEAR_CHECK_ARG(rows > 0 && L > 0 && group > 0 && L % group == 0 && group % 16 == 0,
                   "gear_unpack_dequant_lastdim: need L %% group == 0 and group %% 16 == 0 (L=%d group=%d)", L, group);
    GEAR_CHECK_ARG(code && scale && mn && out, "gear_unpack_dequant_lastdim: null pointer");
    int64_t n_lanes = rows * (int64_t)L / 16;
    int lpg = group / 16;
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256), grid((unsigned)((n_lanes + 255) / 256));
#define GO(B, M, STT)                                                                                             \
    hipLaunchKernelGGL((unpack_dequant_lastdim_kernel<B, M, STT>), grid, block, 0, st, (const uint32_t*)code,      \
                       (const STT*)scale, (const STT*)mn, n_lanes, lpg, (uint4*)out)
    if (mode == 0) {
        if (bits == 2) GO(2, 0, uint16_t);
        else if (bits == 4) GO(4, 0, uint16_t);
        else GO(8, 0, uint16_t);
    } else {
        if (bits == 2) GO(2, 1, float);
        else if (bits == 4) GO(4, 1, float);
        else GO(8, 1, float);
    }
#undef GO
    GEAR_CHECK_LAUNCH("gear_unpack_dequant_lastdim");
    return 0;
}

// token-major K: thread = (word row, 8-channel chunk): reads 8 words (32 B), writes CPW rows x 16 B.
template <int BITS, int MODE, typename ST>
__global__ __launch_bounds__(256) void unpack_dequant_k_kernel(const uint32_t* __restrict__ code,
                                                               const ST* __restrict__ scale, const ST* __restrict__ mn,
                                                               int64_t n_items, int T, int D, int group,
                                                               uint4* __restrict__ out) {
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const int D8 = D / 8;
    const int NW = T / CPW;
    int cl = (int)(i % D8);
    int64_t rw = i / D8;  // bh * NW + w
    int w = (int)(rw % NW);
    int64_t bh = rw / NW;
    const uint4* cp = (const uint4*)(code + rw * D + cl * 8);
    uint4 c0 = cp[0], c1 = cp[1];
    uint32_t words[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    int G = (w * CPW) / group;
    int64_t so = (bh * (T / group) + G) * D + cl * 8;
    float s[8], m[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        s[c] = ld_st<ST>(scale + so + c);
        m[c] = ld_st<ST>(mn + so + c);
    }
#pragma unroll
    for (int j = 0; j < CPW; j++) {
        float f[8];
#pragma unroll
        for (int c = 0; c < 8; c++) f[c] = dequant_one<MODE>((int)((words[c] >> (BITS * j)) & MASK), s[c], m[c]);
        out[(bh * T + (int64_t)w * CPW + j) * D8 + cl] = pack8(f);
    }
}

extern "C" int gear_unpack_dequant_k(const void* code, const void* scale, const void* mn, int64_t bh, int T, int D,
                                     int group, int bits, int mode, void* out, void* stream) {
    GEAR_CHECK_ARG(bits == 2 || bits == 4 || bits == 8, "gear_unpack_dequant_k: bits must be 2, 4 or 8 (got %d)", bits);
    GEAR_CHECK_ARG(mode == 0 || mode == 1, "gear_unpack_dequant_k: bad mode %d", mode);
    int cpw = 32 / bits;
    GEAR_CHECK_ARG(bh > 0 && T > 0 && D > 0 && D % 8 == 0 && group > 0 && T % group == 0 && group % cpw == 0,
                   "gear_unpack_dequant_k: bad shape T=%d D=%d group=%d", T, D, group);
    GEAR_CHECK_ARG(code && scale && mn && out, "gear_unpack_dequant_k: null pointer");
    int64_t n_items = bh * (int64_t)(T / cpw) * (D / 8);
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256), grid((unsigned)((n_items + 255) / 256));
#define GO(B, M, STT)                                                                                          \
    hipLaunchKernelGGL((unpack_dequant_k_kernel<B, M, STT>), grid, block, 0, st, (const uint32_t*)code,         \
                       (const STT*)scale, (const STT*)mn, n_items, T, D, group, (uint4*)out)
    if (mode == 0) {
        if (bits == 2) GO(2, 0, uint16_t);
        else if (bits == 4) GO(4, 0, uint16_t);
        else GO(8, 0, uint16_t);
    } else {
        if (bits == 2) GO(2, 1, float);
        else if (bits == 4) GO(4, 1, float);
        else GO(8, 1, float);
    }
#undef GO
    GEAR_CHECK_LAUNCH("gear_unpack_dequant_k");
    return 0;
}
