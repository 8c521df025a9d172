// rows_multi.hip -- the row compressor for SHORT rows: eight (or four) rows per wave, eight (or sixteen) lanes per row.
//
// Same function as compress_rows.hip (gears_tokenQ / gears_channelQ of GenerationBench/.../Simulated/compress_function.py:261-333
// + group quantization + bit-pack + error, identical outputs), for the rows a head shard produces: with the KV heads split over
// N GPUs a V token row shrinks to (H / N) * 128 elements while the number of rows stays -- Llama-2-7B over 8 GPUs: 512 elements,
// 70B (8 KV heads) over 8 GPUs: 128 elements and one outlier per side.  The workgroup kernel spends a whole wave (and its
// selection machinery: statistics, candidate compaction, 17-round bisection) on such a row with 8 .. 32 of its 64 lanes holding
// data.  Here a wave owns 8 rows, lane l of a row group holds chunks l, l + 8, ... (16 consecutive elements each, NCH <= 4 per
// lane, rows of 128 .. 512 elements), so loads and stores stay coalesced per chunk index and a quantization group (16 .. 128
// elements) is 1 .. 8 adjacent lanes.  Selection: k rounds of "largest remaining composite (order key, lower index first)" per
// side -- a lane-local max over its elements and a 3-step DPP max over the 8 lanes per round; for the small k of a shard
// (k = round(full-row k / N): 1 .. 5) that is a few hundred instructions per wave, shared by 8 rows.  Lists come out sorted by
// index through a chunk-major prefix count over the row group.  Dispatch: gear_compress_rows_geom (compress_rows.hip).
// Round 4: LPR = 16 lanes per row (four rows per wave) for rows of 256 / 512 / 768 elements -- the 512-element rows of Llama-2-7B on
// 8 GPUs with TWO chunks per lane instead of four (187 registers -> two waves per SIMD was this kernel's limit): 176 -> ~150 us.
#include "common.h"
#include "rowgeom.h"
#include "dense16.h"

namespace {

typedef short short2v __attribute__((ext_vector_type(2)));

template <int LPR>
__device__ __forceinline__ uint32_t grp_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm: xor 1
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm: xor 2
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror: the other quad
    if (LPR == 16) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror: the other 8 lanes
    return v;
}

template <int BITS, int MODE, typename ST, int NCH, int LPR>
__global__ __launch_bounds__(256) void compress_rows_multi_kernel(const uint16_t* __restrict__ x, RowGeom gm, int64_t n_rows, int len,
                                                                  int group, int k, uint32_t* __restrict__ code,
                                                                  ST* __restrict__ scale, ST* __restrict__ mn,
                                                                  uint16_t* __restrict__ err, uint16_t* __restrict__ oidx,
                                                                  uint16_t* __restrict__ oval, float* __restrict__ omean) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int WPL = BITS / 2;      // code words per 16 elements
    constexpr int CPW = 32 / BITS;
    __shared__ uint32_t rawlds[256 * 8 * NCH];   // a lane's raw words, for the (rare) list emission with a run-time element index

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int RPW = 64 / LPR;       // rows per wave
    const int l = lane & (LPR - 1), rg = lane / LPR;
    int64_t r = ((int64_t)blockIdx.x * 4 + wave) * RPW + rg;
    const bool valid = r < n_rows;     // a row group past the end works on the last row and stores nothing
    if (!valid) r = n_rows - 1;
    const int64_t row_base = row_base_of(gm, r), orow_base = row_base_out(gm, r);

    int j0[NCH];
    int64_t off[NCH], ooff[NCH];
    uint32_t raw[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        j0[c] = (c * LPR + l) * 16;
        int seg, pos;
        seg_pos(gm, j0[c], seg, pos);
        off[c] = row_base + (int64_t)seg * gm.seg_stride + pos;
        ooff[c] = orow_base + (int64_t)seg * gm.o_seg_stride + pos;
        const uint4* p = (const uint4*)(x + off[c]);
        const uint4 a = p[0], b = p[1];
        raw[c][0] = a.x; raw[c][1] = a.y; raw[c][2] = a.z; raw[c][3] = a.w;
        raw[c][4] = b.x; raw[c][5] = b.y; raw[c][6] = b.z; raw[c][7] = b.w;
    }

    uint32_t fl_lo[NCH], fl_hi[NCH];   // bit j: element j of chunk c is an outlier (small / large side)
#pragma unroll
    for (int c = 0; c < NCH; c++) fl_lo[c] = fl_hi[c] = 0u;
    float fill = 0.0f;
    if (k > 0) {
        // ---------------- row mean of the ORIGINAL row (compress_function.py:276 / :312)
        //   summed in fp64: fp16 values add exactly there, so the mean is the correctly rounded one whatever the order
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
#pragma unroll
            for (int w = 0; w < 8; w++) {
                s += (double)h2f_bits((uint16_t)(raw[c][w] & 0xFFFFu));
                s += (double)h2f_bits((uint16_t)(raw[c][w] >> 16));
                rawlds[(tid * NCH + c) * 8 + w] = raw[c][w];
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (LPR == 16) s += __shfl_xor(s, 8, 64);
        const float mean = (float)(s / (double)len);
        fill = (MODE == 0) ? hround(mean) : mean;
        if (valid && l == 0 && omean) omean[r] = mean;
        // ---------------- selection: k rounds per side of "largest remaining composite"
        //   composite = order key << 16 | (0xFFFF - index): unique inside a row, larger = selected first, ties lower index first.
        //   Winners come out in strictly decreasing order, so round i is "the largest composite below the previous winner": with
        //   t = composite - previous (mod 2^32) every composite below the previous winner wraps above every one that is not, and
        //   the maximum of t finds it -- no element is ever modified, and the small side's composites are the large side's with
        //   the key half inverted.
        uint32_t comp[NCH][16];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t hb = (raw[c][j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                comp[c][j] = (sort_key(hb) << 16) | (0xFFFFu - (uint32_t)(j0[c] + j));
            }
        }
#pragma unroll 1
        for (int side = 0; side < 2; side++) {
            uint32_t prev = 0u;                 // (round 0: nothing wraps, the maximum of t is the largest composite)
#pragma unroll 1
            for (int round = 0; round < k; round++) {
                uint32_t m = 0u;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
#pragma unroll
                    for (int j = 0; j < 16; j++) m = max(m, comp[c][j] - prev);
                }
                prev += grp_max_u32<LPR>(m);
            }
            // selected on this side: composite >= the last winner
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                uint32_t f = 0u;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    f |= (comp[c][j] >= prev) ? (1u << j) : 0u;
                    comp[c][j] ^= 0xFFFF0000u;          // side 1: key -> 0xFFFF - key
                }
                if (side == 0) fl_hi[c] = f; else fl_lo[c] = f;
            }
        }
        // ---------------- sparse lists, sorted by index: chunk-major prefix count over the row group
        {
            uint16_t* oi = oidx + lrow_of(gm, r) * (int64_t)(2 * k);
            uint16_t* ov = oval + lrow_of(gm, r) * (int64_t)(2 * k);
            uint32_t running = 0u;                                    // small-side count | large-side count << 16
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint32_t cnt = (uint32_t)__popc(fl_lo[c]) | ((uint32_t)__popc(fl_hi[c]) << 16);
                uint32_t inc = cnt;
#pragma unroll
                for (int d = 1; d < LPR; d <<= 1) {
                    const uint32_t t = __shfl_up(inc, d, LPR);
                    if (l >= d) inc += t;
                }
                const uint32_t tot = __shfl(inc, LPR - 1, LPR);
                const uint32_t base = running + inc - cnt;
                running += tot;
                if (valid) {
                    uint32_t slot = base & 0xFFFFu;
                    for (uint32_t f = fl_lo[c]; f; f &= f - 1u) {
                        const int j = __builtin_ctz(f);
                        oi[slot] = (uint16_t)(j0[c] + j);
                        ov[slot] = (uint16_t)(rawlds[(tid * NCH + c) * 8 + (j >> 1)] >> (16 * (j & 1)));
                        slot++;
                    }
                    slot = (uint32_t)k + (base >> 16);
                    for (uint32_t f = fl_hi[c]; f; f &= f - 1u) {
                        const int j = __builtin_ctz(f);
                        oi[slot] = (uint16_t)(j0[c] + j);
                        ov[slot] = (uint16_t)(rawlds[(tid * NCH + c) * 8 + (j >> 1)] >> (16 * (j & 1)));
                        slot++;
                    }
                }
            }
        }
    }

    // ---------------- group quantization, one chunk at a time
    // fp32 arithmetic (the simulated path): the packed dense half of the long-row kernels (dense16.h: packed fp16 min / max,
    // reciprocal multiply with a tie guard, Horner packing, packed error) -- the element-by-element loop below it came to 56
    // vector instructions per element against 20, and this kernel is vector-bound (147 us for config 3's 8-GPU V shard)
    if constexpr (MODE == 1) {
        // (row bases are multiples of the group size -- the host checks the strides -- so the row-relative offsets split off)
        uint32_t* code_row = code + orow_base / CPW;
        float* scale_row = (float*)scale + (orow_base >> gm.group_shift);
        float* mn_row = (float*)mn + (orow_base >> gm.group_shift);
        uint16_t* err_row = err ? err + row_base : nullptr;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const uint32_t outl = fl_lo[c] | fl_hi[c];
            uint32_t m[8];
            const short2v o2 = __builtin_bit_cast(short2v, outl | (outl << 16));
#pragma unroll
            for (int w = 0; w < 8; w++) {          // half-word masks of elements 2w (low half) and 2w + 1 (high half)
                short2v t = o2 << (short2v){(short)(15 - 2 * w), (short)(14 - 2 * w)};
                t = t >> (short2v){15, 15};
                m[w] = __builtin_bit_cast(uint32_t, t);
            }
            if (valid)
                dense16<BITS>(raw[c], m, outl, fill, group, gm.group_shift, lane, code_row, scale_row, mn_row,
                              (uint32_t)(ooff[c] - orow_base), err_row, (uint32_t)(off[c] - row_base));
        }
        return;
    }
    const int lanes_per_group = group >> 4;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t outl = fl_lo[c] | fl_hi[c];
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float f = h2f_bits((uint16_t)((raw[c][j >> 1] >> (16 * (j & 1))) & 0xFFFFu));
            v[j] = (outl & (1u << j)) ? fill : f;
        }
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
        const QuantParams<MODE> qp = make_qparams<MODE>(lo, hi, LEVELS);
        const float inv = (qp.scale != 0.0f) ? div_rn(1.0f, qp.scale) : 0.0f;
        uint32_t words[WPL];
#pragma unroll
        for (int w = 0; w < WPL; w++) words[w] = 0u;
        float e[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int q = quant_fast<BITS, MODE>(v[j], qp.mn, qp.scale, inv, LEVELS);
            words[j / CPW] |= (uint32_t)q << (BITS * (j % CPW));
            const float d = (MODE == 0) ? dequant_one<0>(q, qp.scale, qp.mn) : hround(dequant_one<1>(q, qp.scale, qp.mn));
            e[j] = (outl & (1u << j)) ? 0.0f : (v[j] - d);
        }
        if (valid) {
            uint32_t* cp = code + ooff[c] / CPW;
#pragma unroll
            for (int w = 0; w < WPL; w++) cp[w] = words[w];
            if ((l & (lanes_per_group - 1)) == 0) {
                st_st<ST>(scale + (ooff[c] >> gm.group_shift), qp.scale);
                st_st<ST>(mn + (ooff[c] >> gm.group_shift), qp.mn);
            }
            if (err) {
                uint4* ep = (uint4*)(err + off[c]);
                ep[0] = pack8(e);
                ep[1] = pack8(e + 8);
            }
        }
    }
}

}  // namespace

// rows this kernel takes: 128 .. 512 elements in whole 128-element steps (8 lanes x NCH chunks of 16) or 256 / 512 / 768 (16 lanes x
// NCH <= 3 chunks), groups of 16 .. 128, few outliers
// (1280-element rows -- 13B on 4 GPUs, five chunks per lane -- measured SLOWER here than the one-row-per-workgroup kernel: 0.72 against
// 0.61 ms for config 4's V shard; they stay there)
static bool rows_multi_lpr16(int64_t len) { return len % 256 == 0 && len >= 256 && len <= 768; }
bool gear_rows_multi_supported(int64_t len, int group, int k) {
    const bool shape = (len >= 128 && len <= 512 && len % 128 == 0) || rows_multi_lpr16(len);
    return shape && group >= 16 && group <= 128 && k >= 0 && k <= 16;
}

int gear_rows_multi_launch(const void* x, const void* gmv, int64_t n_rows, int64_t len, int group, int bits, int mode, int k,
                           void* code, void* scale, void* mn, void* err, void* oidx, void* oval, void* omean, hipStream_t st) {
    const RowGeom gm = *(const RowGeom*)gmv;
    const bool l16 = rows_multi_lpr16(len);
    const int nch = (int)(len / (l16 ? 256 : 128));
    const int rows_per_block = l16 ? 16 : 32;
    const dim3 grid((unsigned)((n_rows + rows_per_block - 1) / rows_per_block)), block(256);
#define GOM(B, M, STT, N, LP)                                                                                                   \
    hipLaunchKernelGGL((compress_rows_multi_kernel<B, M, STT, N, LP>), grid, block, 0, st, (const uint16_t*)x, gm, n_rows, (int)len, group, k, \
                       (uint32_t*)code, (STT*)scale, (STT*)mn, (uint16_t*)err, (uint16_t*)oidx, (uint16_t*)oval, (float*)omean)
#define GOMN(B, M, STT)                                      \
    do {                                                     \
        if (l16) {                                           \
            if (nch == 1) GOM(B, M, STT, 1, 16);             \
            else if (nch == 2) GOM(B, M, STT, 2, 16);        \
            else GOM(B, M, STT, 3, 16);                      \
        } else if (nch == 1) GOM(B, M, STT, 1, 8);           \
        else GOM(B, M, STT, 3, 8);                           \
    } while (0)
    if (mode == 0) {
        if (bits == 2) GOMN(2, 0, uint16_t); else if (bits == 4) GOMN(4, 0, uint16_t); else GOMN(8, 0, uint16_t);
    } else {
        if (bits == 2) GOMN(2, 1, float); else if (bits == 4) GOMN(4, 1, float); else GOMN(8, 1, float);
    }
#undef GOMN
#undef GOM
    return 0;
}
