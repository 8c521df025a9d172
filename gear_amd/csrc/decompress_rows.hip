// decompress_rows.hip -- packed codes + low-rank factors + sparse outliers -> fp16 rows (gfx950).
//
//   out = fp16( fp16(dequant(code)) + sum_c Q[t,c] P[d,c] ),   outlier positions: out = fp16( value + sum_c ... )
//
// which is how the simulated path assembles its result (GenerationBench/.../Simulated/compress_function.py:204-220:
// `output` already holds the restored outliers and is fp16; `output + error_lr` in fp32; the dispatcher's .half()).
// Rows and segments are described exactly as in compress_rows.hip.
//   kind 0 (V):   row = (b, t), element j -> head j / seglen, channel j % seglen
//   kind 1 (K^T): row = (bh, d), element j -> token j
//
// A workgroup owns RPB consecutive rows.  A lane keeps the same 16 columns for every row, so the 16 x r block of
// the factor that varies along the row (P rows for V, Q rows for K^T) is loaded ONCE into registers (packed fp16 pairs,
// consumed by v_dot2_f32_f16) and reused; only the r-vector of the other factor changes per row.  One 32-byte store per
// lane per row.  Outliers: the block keeps an fp16 table of 4 rows in LDS (0xFFFF = no outlier), refilled every 4 rows
// from entries prefetched one fill ahead; a lane reads its 32 bytes of the table row and selects half-words branch-free.
#include <stdlib.h>

#include "common.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

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

struct DGeom {
    int rows_inner;
    int64_t outer_stride, inner_stride;
    int nseg, seglen;
    int64_t seg_stride;
    int len, group, T, D, r, k, rpb;
    int patch;   // outliers: 1 = overwrite in global memory after the dense pass (no LDS table), 0 = LDS table
    int trows;   // rows covered by one fill of the LDS outlier table (divides rpb; the block refills it rpb / trows times)
    int rpar;    // short rows (len / 16 < 64 lanes): rpar rows side by side in the block's one wave, lane = (row slot, 16 columns)
    int64_t n_rows;
};

// RV: compile-time rank (4 / 8 / 16) or 0 for the generic runtime-rank path.
// TB: launch bound bucket (256 / 512 / 1024 threads) -- the register budget for the 16 x r factor block.
template <int BITS, int MODE, typename ST, int KIND, int RV, int TB>
__global__ __launch_bounds__(TB) void decompress_rows_kernel(const uint32_t* __restrict__ code, const ST* __restrict__ scale,
                                       const ST* __restrict__ mn, DGeom g, const uint16_t* __restrict__ P,
                                       const uint16_t* __restrict__ Q, const uint16_t* __restrict__ oidx,
                                       const uint16_t* __restrict__ oval, uint16_t* __restrict__ out) {
    constexpr int WPL = BITS / 2;
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    constexpr int RVS = RV > 0 ? RV : 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t dsm[];   // [trows][len] fp16 outlier values (0xFFFF = none)
    const int tid = threadIdx.x;
    // short rows: the wave holds rpar rows at a time, lane = (row slot sub, 16-column chunk lc); otherwise one row, lane = chunk
    const int lpr = g.len >> 4;
    const int lc = g.rpar > 1 ? tid % lpr : tid, sub = g.rpar > 1 ? tid / lpr : 0;
    const int j0 = lc * 16;
    const bool active = j0 < g.len;
    const int64_t row0 = (int64_t)blockIdx.x * g.rpb;
    const int r = g.r;
    // LDS outlier table: trows x len fp16, 0xFFFF (a NaN no payload value has) = "no outlier here"
    uint16_t* lval = (uint16_t*)dsm;
    // the sparse part of `trows` rows of the block goes to LDS at a time: the dense pass then patches its own elements and
    // every global store stays a full 32-byte vector (scattered 2-byte stores cost a line read-modify-write each).  The
    // table is refilled every trows rows so that its size (and with it the number of resident blocks) does not grow with
    // the number of rows a block keeps its register-resident factor block for.
    // The entries of the next fill are loaded into registers one fill ahead (2 per thread cover 4 rows x 2k <= 512 entries
    // for 256 threads; more than that falls back to loading inside the fill), so a fill is LDS work only.
    constexpr int PF = 2;
    uint16_t pf_idx[PF], pf_val[PF];
    const int per_row_t = 2 * g.k, fill_n = g.trows * per_row_t;
    const bool pf_ok = fill_n <= PF * (int)blockDim.x;
    // entry e = tid + q * blockDim of a fill is entry pf_c[q] of table row pf_r[q]: the same for every fill (one division here,
    // none per fill)
    int pf_r[PF], pf_c[PF];
#pragma unroll
    for (int q = 0; q < PF; q++) {
        const int e = tid + q * (int)blockDim.x;
        pf_r[q] = (g.k > 0 && e < fill_n) ? e / per_row_t : -1;
        pf_c[q] = (g.k > 0 && e < fill_n) ? e - pf_r[q] * per_row_t : 0;
    }
    // Blocks walk their 16 rows in a rotated order (a multiple of the table period, by block index) so that blocks running in
    // near lockstep do not all write at the same offset of their 128 KB regions (HBM channel camping, tools/ubench/
    // store_pattern.hip; worth 2 % here)
    const int rot = (g.rpb == 16 && g.rpar == 1 && row0 + 16 <= g.n_rows) ? 4 * (int)((blockIdx.x ^ (blockIdx.x >> 2)) & 3) : 0;
    auto phys = [&](int ri) { return rot ? ((ri + rot) & 15) : ri; };
    auto prefetch_entries = [&](int rbase) {
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const int64_t row = row0 + phys(rbase) + pf_r[q];
            pf_idx[q] = 0; pf_val[q] = 0;
            if (pf_r[q] >= 0 && row < g.n_rows && rbase < g.rpb) {
                pf_idx[q] = oidx[row * per_row_t + pf_c[q]];
                pf_val[q] = oval[row * per_row_t + pf_c[q]];
            }
        }
    };
    auto fill_table = [&](int rbase) {
        for (int i = tid; i < g.trows * (g.len / 8); i += blockDim.x)
            ((uint4*)lval)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        __syncthreads();
        if (pf_ok) {
#pragma unroll
            for (int q = 0; q < PF; q++)
                if (pf_r[q] >= 0 && row0 + phys(rbase) + pf_r[q] < g.n_rows) lval[(size_t)pf_r[q] * g.len + pf_idx[q]] = pf_val[q];
            prefetch_entries(rbase + g.trows);
        } else {
            for (int e = tid; e < fill_n; e += blockDim.x) {
                const int ri = e / per_row_t;
                const int64_t row = row0 + phys(rbase) + ri;
                if (row < g.n_rows) lval[(size_t)ri * g.len + oidx[row * per_row_t + e % per_row_t]] = oval[row * per_row_t + e % per_row_t];
            }
        }
        __syncthreads();
    };
    if (g.k > 0 && !g.patch && pf_ok) prefetch_entries(0);
    // all rows of the block share the outer index (rpb divides rows_inner)
    const int ro = (int)(row0 / g.rows_inner);
    const int seg = active ? j0 / g.seglen : 0, pos = active ? j0 % g.seglen : 0;

    // ---- the 16 x r factor block of this lane's columns (row-independent)
    // kept as packed fp16 pairs: the row term is RV/2 v_dot2_f32_f16 per element (exact products, fp32 accumulate) and
    // the block costs RV/2 registers per column instead of RV
    constexpr int RV2 = RV > 0 ? RV / 2 : 1;
    uint32_t gb[16][RV2];
    const uint16_t* gbp = nullptr;
    if (active && r > 0) {
        if (KIND == 0) gbp = P + (((int64_t)ro * g.nseg + seg) * g.D + pos) * r;   // P[bh, pos.., :]
        else gbp = Q + ((int64_t)ro * g.T + j0) * r;                                // Q[bh, j0.., :]
        if (RV > 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (RV == 4) { uint2 t = *(const uint2*)(gbp + j * 4); gb[j][0] = t.x; gb[j][1 % RV2] = t.y; }
                else {
#pragma unroll
                    for (int h = 0; h < RV2 / 4; h++) {
                        const uint4 t = ((const uint4*)(gbp + j * RVS))[h];
                        gb[j][(4 * h) % RV2] = t.x; gb[j][(4 * h + 1) % RV2] = t.y;
                        gb[j][(4 * h + 2) % RV2] = t.z; gb[j][(4 * h + 3) % RV2] = t.w;
                    }
                }
            }
        }
    }

    // software pipeline over the block's rows: the loads of row i+1 are issued before row i is computed and stored
    // Row ri of the block: all rows share the outer index and rpb divides rows_inner, so everything is the first row's
    // offset plus ri times a stride -- no integer division inside the row loop (a run-time 64-bit quotient is > 100 VALU
    // instructions, and the loop had three of them per row).
    struct RowIn { uint32_t words[WPL]; float s, m; uint4 fv0, fv1; int64_t off; };
    const int rin0 = (int)(row0 % g.rows_inner);
    const int64_t off0 = (int64_t)ro * g.outer_stride + (int64_t)rin0 * g.inner_stride + (int64_t)seg * g.seg_stride + pos;
    const int64_t gi0 = off0 / g.group;
    const int gstep = (int)(g.inner_stride / g.group);          // (the host checks inner_stride % group == 0)
    const uint16_t* fv0p = nullptr;
    if (RV > 0 && r > 0)
        fv0p = (KIND == 0) ? Q + ((((int64_t)ro * g.nseg + seg) * g.T) + rin0) * r : P + ((int64_t)ro * g.D + rin0) * r;
    auto fetch = [&](int li, RowIn& in) {
        const int ri = phys(li);
        in.off = off0 + (int64_t)ri * g.inner_stride;
        const int64_t gi = gi0 + (int64_t)ri * gstep;
        in.s = ld_st<ST>(scale + gi);
        in.m = ld_st<ST>(mn + gi);
#pragma unroll
        for (int w = 0; w < WPL; w++) in.words[w] = code[in.off / CPW + w];
        if (RV > 0 && r > 0) {
            const uint16_t* fvp = fv0p + (int64_t)ri * r;
            if (RV == 4) { uint2 t = *(const uint2*)fvp; in.fv0 = make_uint4(t.x, t.y, 0, 0); }
            else in.fv0 = *(const uint4*)fvp;
            if (RV == 16) in.fv1 = ((const uint4*)fvp)[1];
        }
    };
    const int nrows = active ? (int)((g.n_rows - row0) < g.rpb ? (g.n_rows - row0) : g.rpb) : 0;
    const bool table = g.k > 0 && !g.patch;
    const int nrows_blk = (int)((g.n_rows - row0) < g.rpb ? (g.n_rows - row0) : g.rpb);   // (block-uniform)
    auto before_row = [&](int ri) {
        if (table && (ri & (g.trows - 1)) == 0) {     // (trows is a power of two)
            if (ri) __syncthreads();        // everyone is done reading the previous fill
            fill_table(ri);
        }
    };
    auto compute_row = [&](int ri, const RowIn& cur) {
        float f[16];
#pragma unroll
        for (int w = 0; w < WPL; w++) {
#pragma unroll
            for (int j = 0; j < CPW; j++)
                f[w * CPW + j] = dequant_one<MODE>((int)((cur.words[w] >> (BITS * j)) & MASK), cur.s, cur.m);
        }
        // the dequantized values as packed fp16 (MODE 1: this is the reference's cast of the fp32 result; MODE 0: exact)
        uint4 d0 = pack8(f), d1 = pack8(f + 8);
        if (table) {
            // outlier elements: the stored value replaces the dequantized one (the low-rank term still adds).  Branch-free:
            // the lane's 32 bytes of the table row, half-words that are not the sentinel select the table value
            // (per word: xor, min(x, 1), 0 - x, v_bfi -- the first version walked 16 branchy ds_read_u16 blocks)
            const int rt = ri & (g.trows - 1);
            const uint4 t0 = *(const uint4*)&lval[(size_t)rt * g.len + j0], t1 = *(const uint4*)&lval[(size_t)rt * g.len + j0 + 8];
            auto sel = [](uint32_t tw, uint32_t dw) {
                uint32_t x = ~tw, mk, rr;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(mk) : "v"(x), "v"(0x00010001u));
                asm("v_pk_sub_u16 %0, %1, %2" : "=v"(mk) : "v"(0u), "v"(mk));
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(rr) : "v"(mk), "v"(tw), "v"(dw));
                return rr;
            };
            d0 = make_uint4(sel(t0.x, d0.x), sel(t0.y, d0.y), sel(t0.z, d0.z), sel(t0.w, d0.w));
            d1 = make_uint4(sel(t1.x, d1.x), sel(t1.y, d1.y), sel(t1.z, d1.z), sel(t1.w, d1.w));
        }
        if (r > 0) {
            unpack8(d0, f);
            unpack8(d1, f + 8);
        }
        if (r > 0) {
            if (RV > 0) {
                const uint32_t fv[8] = {cur.fv0.x, cur.fv0.y, cur.fv0.z, cur.fv0.w, cur.fv1.x, cur.fv1.y, cur.fv1.z, cur.fv1.w};
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    float acc = f[j];   // the dot-product chain starts from the dequantized value (no zero-fill, no final add)
#pragma unroll
                    for (int c = 0; c < RV2; c++)
                        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, fv[c]), __builtin_bit_cast(half2_t, gb[j][c]), acc, false);
                    f[j] = acc;
                }
            } else {
                const int rin = rin0 + phys(ri);
                const uint16_t* fvp = (KIND == 0) ? Q + ((((int64_t)ro * g.nseg + seg) * g.T) + rin) * r
                                                  : P + ((int64_t)ro * g.D + rin) * r;
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
        uint4* op = (uint4*)(out + cur.off);
        if (r > 0) {
            op[0] = pack8(f);
            op[1] = pack8(f + 8);
        } else {
            op[0] = d0;
            op[1] = d1;
        }
    };
    // Two named row buffers in ping-pong (no register copies between them: a copy of a buffer whose loads are still in flight
    // makes the compiler wait for them on the spot, which is what a rotating "cur = next" pipeline did): the loads of row
    // i + 1 are issued before row i is computed and stored.
    // (step li of the loop: block rows li * rpar .. li * rpar + rpar - 1, this lane's is li * rpar + sub)
    RowIn bufA = {}, bufB = {};
    const int R = g.rpar;
    if (sub < nrows) fetch(sub, bufA);
    for (int rb = 0; rb < nrows_blk; rb += 2 * R) {
        before_row(rb);
        if (active) {
            if (rb + R + sub < nrows) fetch(rb + R + sub, bufB);
            if (rb + sub < nrows) compute_row(rb + sub, bufA);
        }
        if (rb + R < nrows_blk) {
            before_row(rb + R);
            if (active) {
                if (rb + 2 * R + sub < nrows) fetch(rb + 2 * R + sub, bufA);
                if (rb + R + sub < nrows) compute_row(rb + R + sub, bufB);
            }
        }
    }
    if (g.k > 0 && g.patch) {
        // Sparse pass over the rows this block has just written: the lines are still dirty in L2, so the 2-byte stores
        // merge there instead of costing an HBM read-modify-write each (what a separate kernel pays), and the dense loop
        // above stays free of per-element checks and of the LDS table that capped occupancy at 2 blocks per CU.
        __syncthreads();   // (workgroup-scope release/acquire: the dense stores are in L2 before any overwrite is issued)
        const int per_row = 2 * g.k;
        const int total = (int)min((int64_t)g.rpb, g.n_rows - row0) * per_row;
        for (int e = tid; e < total; e += blockDim.x) {
            const int ri = e / per_row;
            const int64_t row = row0 + ri;
            const int rin = (int)(row % g.rows_inner);
            const uint32_t idx = oidx[row * per_row + e % per_row];
            float v = h2f_bits(oval[row * per_row + e % per_row]);
            const int sg = (int)idx / g.seglen, ps = (int)idx % g.seglen;
            if (r > 0) {
                int64_t bh, t, d;
                if (KIND == 0) { bh = (int64_t)ro * g.nseg + sg; t = rin; d = ps; }
                else { bh = ro; t = idx; d = rin; }
                const uint16_t* qp = Q + (bh * g.T + t) * r;
                const uint16_t* pp = P + (bh * g.D + d) * r;
                float acc = 0.0f;
                if (RV > 0) {
                    float qa[RVS], pb[RVS];
                    load_halfs<RVS>(qp, qa);
                    load_halfs<RVS>(pp, pb);
#pragma unroll
                    for (int c = 0; c < RVS; c++) acc = fmaf(qa[c], pb[c], acc);
                } else {
                    for (int c = 0; c < r; c++) acc = fmaf(h2f_bits(qp[c]), h2f_bits(pp[c]), acc);
                }
                v += acc;
            }
            out[(int64_t)ro * g.outer_stride + (int64_t)rin * g.inner_stride + (int64_t)sg * g.seg_stride + ps] = f2h_bits(v);
        }
    }
}

// Sparse restore, one thread per outlier entry (launched after the dense kernel on the same stream): plenty of
// independent threads hide the dependent index -> factor -> store chain that stalled the dense workgroups.
template <int KIND, int RV>
__global__ __launch_bounds__(256) void decompress_sparse_kernel(DGeom g, const uint16_t* __restrict__ P,
                                                                const uint16_t* __restrict__ Q,
                                                                const uint16_t* __restrict__ oidx,
                                                                const uint16_t* __restrict__ oval,
                                                                uint16_t* __restrict__ out) {
    constexpr int RVS = RV > 0 ? RV : 1;
    const int per_row = 2 * g.k;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= g.n_rows * per_row) return;
    const int64_t row = e / per_row;
    const int ro = (int)(row / g.rows_inner), rin = (int)(row % g.rows_inner);
    const uint32_t idx = oidx[e];
    float v = h2f_bits(oval[e]);
    const int sg = (int)idx / g.seglen, ps = (int)idx % g.seglen;
    const int r = g.r;
    if (r > 0) {
        int64_t bh, t, d;
        if (KIND == 0) { bh = (int64_t)ro * g.nseg + sg; t = rin; d = ps; }
        else { bh = ro; t = idx; d = rin; }
        const uint16_t* qp = Q + (bh * g.T + t) * r;
        const uint16_t* pp = P + (bh * g.D + d) * r;
        float acc = 0.0f;
        if (RV > 0) {
            float a[RVS], b[RVS];
            load_halfs<RVS>(qp, a);
            load_halfs<RVS>(pp, b);
#pragma unroll
            for (int c = 0; c < RVS; c++) acc = fmaf(a[c], b[c], acc);
        } else {
            for (int c = 0; c < r; c++) acc = fmaf(h2f_bits(qp[c]), h2f_bits(pp[c]), acc);
        }
        v += acc;
    }
    const int64_t off = (int64_t)ro * g.outer_stride + (int64_t)rin * g.inner_stride + (int64_t)sg * g.seg_stride + ps;
    out[off] = f2h_bits(v);
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
    GEAR_CHECK_ARG(rows_inner > 0 && n_rows % rows_inner == 0, "gear_decompress_rows: n_rows must be a multiple of rows_inner");
    GEAR_CHECK_ARG(inner_stride % group == 0 && outer_stride % group == 0 && (nseg == 1 || seg_stride % group == 0),
                   "gear_decompress_rows: strides must be multiples of the group size");
    if (kind == 0) GEAR_CHECK_ARG(rows_inner == T && seglen == D, "gear_decompress_rows: kind 0 needs rows_inner == T and seglen == D");
    if (kind == 1) GEAR_CHECK_ARG(rows_inner == D && nseg == 1 && seglen == T, "gear_decompress_rows: kind 1 needs rows_inner == D, one segment of T");
    const int patch = 0;   // (an in-kernel global patch pass measured 0.81 ms vs 0.72 ms for the LDS table: not used)
    // rows per block, measured on the 7B / 4k tensors (V / K^T ms): factors only: 16 rows 0.48 / 0.42, 8 rows 0.51 / 0.47
    // (fewer reloads of the lane's factor block).  The LDS outlier table covers 4 rows (35 KB) and is refilled inside the
    // block: outliers + factors 0.58 / 0.54 with 16 rows per block (0.64 / 0.62 when the block itself was 4 rows, 0.70 /
    // 0.75 with an 8-row table = 70 KB = half the resident blocks)
    int rpb = patch ? 8 : (r > 0 ? 16 : 8);
    while (rpb > 1 && rows_inner % rpb != 0) rpb >>= 1;
    int trows = rpb < 4 ? rpb : 4;            // rows per fill of the LDS outlier table (35 KB at 4096 columns)
    if (trows > rpb) trows = rpb;
    while (trows > 1 && (rpb % trows != 0 || (size_t)trows * ((len / 32 + 1) * 4 + len * 2) > 72 * 1024)) trows >>= 1;
    // short rows (head shards: 128 .. 512 elements): 8 / 4 / 2 rows side by side in the one wave of the block
    int rpar = 1;
    if (len / 16 < 64 && 64 % (len / 16) == 0 && rpb % (64 / (len / 16)) == 0) {
        rpar = (int)(64 / (len / 16));
        if (trows < rpar) trows = rpar;            // one fill of the table covers at least the rows in flight
    }
    const size_t shmem = (k > 0 && !patch) ? (size_t)trows * len * 2 : 0;
    GEAR_CHECK_ARG(shmem <= 72 * 1024, "gear_decompress_rows: row too long for the LDS outlier table");
    DGeom g{rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride, (int)len, group, T, D, r, k, rpb, patch, trows, rpar, n_rows};
    int threads = (int)((len / 16 + 63) / 64 * 64);
    hipStream_t st = (hipStream_t)stream;
    dim3 block(threads), grid((unsigned)((n_rows + rpb - 1) / rpb));
#define GOT(B, M, STT, KD, RVV, TBB)                                                                                    \
    do {                                                                                                                \
        auto kfn = decompress_rows_kernel<B, M, STT, KD, RVV, TBB>;                                                     \
        if (shmem > 48 * 1024)                                                                                          \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);        \
        hipLaunchKernelGGL(kfn, grid, block, shmem, st, (const uint32_t*)code, (const STT*)scale, (const STT*)mn, g,     \
                           (const uint16_t*)P, (const uint16_t*)Q, (const uint16_t*)oidx, (const uint16_t*)oval,        \
                           (uint16_t*)out);                                                                             \
    } while (0)
#define GO(B, M, STT, KD, RVV) do { if (threads <= 256) GOT(B, M, STT, KD, RVV, 256); else if (threads <= 512) GOT(B, M, STT, KD, RVV, 512); \
                                    else GOT(B, M, STT, KD, RVV, 1024); } while (0)
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
#undef GOT
    GEAR_CHECK_LAUNCH("gear_decompress_rows");
    return 0;
}
