// decompress_rows_impl.h (included by decompress_rows.hip and decompress_rows_m1.hip) -- packed codes + low-rank factors + sparse outliers -> fp16 rows (gfx950).
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

#include <type_traits>

#include "common.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));



// a block-uniform 64-bit offset, moved to scalar registers (kept an integer: a pointer rebuilt from integers loses its address
// space and its loads become FLAT ones, which also count on lgkmcnt)
__device__ __forceinline__ int64_t uni64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

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
    int og, ig, sg;   // outer / inner / segment stride in groups
    int xcd;          // launch index -> block in XCD-contiguous order (grid a multiple of 8)
    int general;      // never the straight-line path for full blocks (option decomp_general: test coverage of the general loop)
};

// RV: compile-time rank (4 / 8 / 16) or 0 for the generic runtime-rank path.
// TB: launch bound bucket (256 / 512 / 1024 threads) -- the register budget for the 16 x r factor block.
template <int BITS, int MODE, typename ST, int KIND, int RV, int TB>
__global__ __launch_bounds__(TB, (TB == 256 ? 2 : 1)) void decompress_rows_kernel(const uint32_t* __restrict__ code, const ST* __restrict__ scale,
                                       const ST* __restrict__ mn, DGeom g, const uint16_t* __restrict__ P,
                                       const uint16_t* __restrict__ Q, const uint16_t* __restrict__ oidx,
                                       const uint16_t* __restrict__ oval, uint16_t* __restrict__ out) {
    constexpr int WPL = BITS / 2;
    constexpr int CPW = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    constexpr int RVS = RV > 0 ? RV : 1;
    // straight-line path: rows whose inputs are in flight ahead of the row being computed (measured at config 3: two rows ahead
    // 0.368 against 0.384 ms for K^T, 0.441 against 0.421 ms for V)
    constexpr int PFD = (KIND == 1 && TB < 1024) ? 2 : 1;      // (1024 threads: 128 registers per lane, no room for a third row)
    extern __shared__ __attribute__((aligned(16))) uint32_t dsm[];   // [trows][len] ~(fp16 outlier value) (0 = none)
    const int tid = threadIdx.x;
    // short rows: the wave holds rpar rows at a time, lane = (row slot sub, 16-column chunk lc); otherwise one row, lane = chunk
    const int lpr = g.len >> 4;
    const int lc = g.rpar > 1 ? tid % lpr : tid, sub = g.rpar > 1 ? tid / lpr : 0;
    const int j0 = lc * 16;
    const bool active = j0 < g.len;
    // XCD-aware order (g.xcd, K^T): workgroup i runs on XCD i % 8, so the launch index is turned around -- XCD x walks the
    // consecutive blocks [x * n / 8, (x + 1) * n / 8).  The 8 blocks that share a head's 64 KB column factor (Q[bh]: 128 rows = 8
    // blocks) then run on ONE XCD at about the same time and its L2 fetches the factor once instead of eight L2s fetching it once
    // each: PMC FETCH_SIZE of the K^T call at config 3 0.78 -> 0.25 GB (the time moves by ~2 %: those reads were Infinity Cache hits)
    const uint32_t bid = g.xcd ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int64_t row0 = (int64_t)bid * g.rpb;
    const int r = g.r;
    // LDS outlier table: trows x len half-words holding the COMPLEMENT of the outlier's fp16 bits; 0 = "no outlier here" (the
    // complement of 0xFFFF, a NaN no payload value has)
    uint16_t* lval = (uint16_t*)dsm;
    // the sparse part of `trows` rows of the block goes to LDS at a time: the dense pass then patches its own elements and
    // every global store stays a full 32-byte vector (scattered 2-byte stores cost a line read-modify-write each).  The
    // table is refilled every trows rows so that its size (and with it the number of resident blocks) does not grow with
    // the number of rows a block keeps its register-resident factor block for.
    // The entries of the next fill are loaded into registers one fill ahead (2 per thread cover 4 rows x 2k <= 512 entries
    // for 256 threads; more than that falls back to loading inside the fill), so a fill is LDS work only.
    constexpr int PF = 2;
    uint16_t pf_idx[PF], pf_val[PF];   // (kept exactly as loaded: any arithmetic on them here would wait for the loads on the spot)
    const int per_row_t = 2 * g.k, fill_n = g.trows * per_row_t;
    const bool pf_ok = fill_n <= PF * (int)blockDim.x;
    // entry e = tid + q * blockDim of a fill is entry c of table row rr, the same for every fill (one division here, none per
    // fill); kept as rr << 16 | c, rr = -1: nothing to do
    int pf_rc[PF];
#pragma unroll
    for (int q = 0; q < PF; q++) {
        const int e = tid + q * (int)blockDim.x;
        const int rr = (g.k > 0 && e < fill_n) ? e / per_row_t : -1;
        pf_rc[q] = (rr << 16) | (rr >= 0 ? e - rr * per_row_t : 0);
    }
    // Blocks walk their 16 rows in a rotated order (a multiple of the table period, by block index) so that blocks running in
    // near lockstep do not all write at the same offset of their 128 KB regions (HBM channel camping, tools/ubench/
    // store_pattern.hip; worth 2 % here)
    const int rot = (g.rpb == 16 && g.rpar == 1 && row0 + 16 <= g.n_rows) ? 4 * (int)((blockIdx.x ^ (blockIdx.x >> 2)) & 3) : 0;
    auto phys = [&](int ri) { return rot ? ((ri + rot) & 15) : ri; };
    auto prefetch_entries = [&](int rbase) {
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const int rr = pf_rc[q] >> 16, c = pf_rc[q] & 0xFFFF;
            const int64_t row = row0 + phys(rbase) + rr;
            pf_idx[q] = 0; pf_val[q] = 0;
            if (rr >= 0 && row < g.n_rows && rbase < g.rpb) {
                pf_idx[q] = oidx[row * per_row_t + c];
                pf_val[q] = oval[row * per_row_t + c];
            }
        }
    };
    // (the same for a full block, every load issued by every lane -- a lane without an entry re-reads entry 0 of the fill's
    // first row and drops it: loads under a per-lane condition make the compiler wait for ALL memory operations at the join)
    auto prefetch_entries_all = [&](int rbase) {
#pragma unroll
        for (int q = 0; q < PF; q++) {
            const int rr = pf_rc[q] >> 16, c = pf_rc[q] & 0xFFFF;
            const int64_t e = (row0 + phys(rbase) + (rr < 0 ? 0 : rr)) * per_row_t + c;
            pf_idx[q] = oidx[e];
            pf_val[q] = oval[e];
        }
    };
    auto zero_table = [&]() {
        for (int i = tid; i < g.trows * (g.len / 8); i += blockDim.x) ((uint4*)lval)[i] = make_uint4(0u, 0u, 0u, 0u);
    };
    auto fill_table = [&](int rbase) {
        zero_table();
        __syncthreads();
        if (pf_ok) {
#pragma unroll
            for (int q = 0; q < PF; q++) {
                const int rr = pf_rc[q] >> 16;
                // (an index beyond the row = an unused slot of a head shard's list, 0xFFFF: gear_compress_value_sharded)
                if (rr >= 0 && row0 + phys(rbase) + rr < g.n_rows && (int)pf_idx[q] < g.len) lval[rr * g.len + pf_idx[q]] = (uint16_t)~pf_val[q];
            }
            prefetch_entries(rbase + g.trows);
        } else {
            for (int e = tid; e < fill_n; e += blockDim.x) {
                const int ri = e / per_row_t;
                const int64_t row = row0 + phys(rbase) + ri;
                if (row < g.n_rows) {
                    const int ix = oidx[row * per_row_t + e % per_row_t];
                    if (ix < g.len) lval[(size_t)ri * g.len + ix] = (uint16_t)~oval[row * per_row_t + e % per_row_t];
                }
            }
        }
        __syncthreads();
    };
    // all rows of the block share the outer index (rpb divides rows_inner)
    const int ro = (int)(row0 / g.rows_inner);
    const int seg = active ? j0 / g.seglen : 0, pos = active ? j0 % g.seglen : 0;
    // (store exchange, see compute_row: this lane writes the 16-byte half `tid >> 5 & 1` of lane (tid & 31) of its wave in the
    // first store and of lane 32 + (tid & 31) in the second; st_a / st_b = their offsets relative to this lane's own.  With
    // short rows side by side the lower 32 lanes hold whole rows, so the exchange puts whole rows into one instruction)
    const bool wave_full = g.rpar > 1 || ((tid | 63) + 1) * 16 <= g.len;
    int st_a = 0, st_b = 0;
    if (wave_full) {
        auto off_of = [&](int ln) {          // element offset of lane ln's first column, its row slot included
            const int lcs = g.rpar > 1 ? ln % lpr : ln, subs = g.rpar > 1 ? ln / lpr : 0, js = lcs * 16;
            return (js / g.seglen) * (int)g.seg_stride + js % g.seglen + subs * (int)g.inner_stride;
        };
        const int la = (tid & ~63) + (tid & 31), hf = 8 * ((tid >> 5) & 1), own = off_of(tid);
        st_a = off_of(la) + hf - own;
        st_b = off_of(la + 32) + hf - own;
    }

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
    struct RowIn { uint32_t words[WPL]; float s, m; uint4 fv0, fv1; int off; };
    // Addresses = a block-uniform 64-bit base (scalar registers: everything that depends on the block's first row only) + a
    // 32-bit offset made of the lane's column part and ri times a stride (the host checks that a slab stays below 2^31
    // elements): one multiply and a few adds per row and stream instead of 64-bit multiply-adds per lane.
    const int rin0 = (int)(row0 % g.rows_inner);
    const int64_t base_off = (int64_t)ro * g.outer_stride + (int64_t)rin0 * g.inner_stride;   // (a multiple of the group size)
    uint16_t* outb = out + uni64(base_off);
    const uint32_t* codeb = code + uni64(base_off / CPW);
    const int64_t gbase = uni64((int64_t)ro * g.og + (int64_t)rin0 * g.ig);          // = base_off / group
    const ST* scaleb = scale + gbase;
    const ST* mnb = mn + gbase;
    const int lane_off = seg * (int)g.seg_stride + pos;
    const int lane_g = seg * g.sg + pos / g.group;
    const int istride = (int)g.inner_stride;
    const int gstep = g.ig;                                    // (the host checks inner_stride % group == 0)
    const uint16_t* fvb = nullptr;
    int fv_lane = 0;
    if (RV > 0 && r > 0) {
        fvb = (KIND == 0) ? Q + uni64(((int64_t)ro * g.nseg * g.T + rin0) * r) : P + uni64(((int64_t)ro * g.D + rin0) * r);
        fv_lane = (KIND == 0) ? seg * g.T * r : 0;
    }
    auto fetch = [&](int li, RowIn& in) {
        const int ri = phys(li);
        in.off = lane_off + ri * istride;
        const int gi = lane_g + ri * gstep;
        in.s = ld_st<ST>(scaleb + gi);
        in.m = ld_st<ST>(mnb + gi);
#pragma unroll
        for (int w = 0; w < WPL; w++) in.words[w] = codeb[in.off / CPW + w];
        if (RV > 0 && r > 0) {
            const uint16_t* fvp = fvb + fv_lane + ri * r;
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
    // the dequantized row as packed fp16, outlier values restored
    auto dense_row = [&](int ri, const uint32_t (&words)[WPL], float cs, float cm, const bool table, uint4& d0, uint4& d1) __attribute__((always_inline)) {
        struct { uint32_t words[WPL]; float s, m; } cur;
#pragma unroll
        for (int w = 0; w < WPL; w++) cur.words[w] = words[w];
        cur.s = cs; cur.m = cm;
        float f[16];
        if constexpr (BITS == 2) {
            // Two-bit codes: a group has four dequantized values.  Compute them once per row-lane (the lane's 16 columns lie
            // in one group), keep them as two packed-fp16 registers and let v_perm_b32 pick the pair of every output word:
            // 4 instructions per element pair (packed shift, mask, packed multiply-add to byte selectors, permute) where
            // extract + convert + multiply + add + pack came to 11.
            const uint32_t l01 = f2h2_bits(dequant_one<MODE>(0, cur.s, cur.m), dequant_one<MODE>(1, cur.s, cur.m));
            const uint32_t l23 = f2h2_bits(dequant_one<MODE>(2, cur.s, cur.m), dequant_one<MODE>(3, cur.s, cur.m));
            const uint32_t w = cur.words[0], w1 = w << 1, wh = w >> 31;
            uint32_t dw[8];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                // both halves = bits [8b - 1, 8b + 14] of the code word: byte b of it, doubled, with room for its top bit
                const uint32_t y = __builtin_amdgcn_perm(wh, w1, (uint32_t)(b | ((b + 1) << 8) | (b << 16) | ((b + 1) << 24)));
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const u16x2 sh = {(uint16_t)(4 * h), (uint16_t)(4 * h + 2)};
                    u16x2 t = __builtin_bit_cast(u16x2, y) >> sh;            // {2 c_even, 2 c_odd} in bits 1..2 of the halves
                    t = (t & (uint16_t)6) * (uint16_t)0x0101 + (uint16_t)0x0100;   // byte selectors {2c, 2c + 1} per half
                    dw[2 * b + h] = __builtin_amdgcn_perm(l23, l01, __builtin_bit_cast(uint32_t, t));
                }
            }
            d0 = make_uint4(dw[0], dw[1], dw[2], dw[3]);
            d1 = make_uint4(dw[4], dw[5], dw[6], dw[7]);
        } else {
#pragma unroll
            for (int w = 0; w < WPL; w++) {
#pragma unroll
                for (int j = 0; j < CPW; j++)
                    f[w * CPW + j] = dequant_one<MODE>((int)((cur.words[w] >> (BITS * j)) & MASK), cur.s, cur.m);
            }
            // the dequantized values as packed fp16 (MODE 1: this is the reference's cast of the fp32 result; MODE 0: exact)
            d0 = pack8(f);
            d1 = pack8(f + 8);
        }
        if (table) {
            // outlier elements: the stored value replaces the dequantized one (the low-rank term still adds).  Branch-free:
            // the lane's 32 bytes of the table row, half-words that are not the sentinel select the table value
            // (per word: min(x, 1), 0 - x, one three-input bit operation -- the first version walked 16 branchy ds_read_u16 blocks)
            const int rt = ri & (g.trows - 1);
            const uint4 t0 = *(const uint4*)&lval[(size_t)rt * g.len + j0], t1 = *(const uint4*)&lval[(size_t)rt * g.len + j0 + 8];
            auto sel = [](uint32_t tn, uint32_t dw) {
                uint32_t mk;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(mk) : "v"(tn), "v"(0x00010001u));
                asm("v_pk_sub_u16 %0, %1, %2" : "=v"(mk) : "v"(0u), "v"(mk));
                uint32_t rr;                             // mk ? ~tn : dw  (truth table with S0 = 0xF0, S1 = 0xCC, S2 = 0xAA)
                asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x3a" : "=v"(rr) : "v"(mk), "v"(tn), "v"(dw));
                return rr;
            };
            d0 = make_uint4(sel(t0.x, d0.x), sel(t0.y, d0.y), sel(t0.z, d0.z), sel(t0.w, d0.w));
            d1 = make_uint4(sel(t1.x, d1.x), sel(t1.y, d1.y), sel(t1.z, d1.z), sel(t1.w, d1.w));
        }
    };
    // the lane's 32 bytes of a row -> memory (swp: the store halves exchanged between lanes l and l + 32 -- every lane of the wave
    // must be in this call)
    auto store_row = [&](int off, uint4 d0, uint4 d1, const bool swp) __attribute__((always_inline)) {
        if (swp) {
            // Full lines per store instruction: a lane's 32 bytes as two 16-byte stores cover every 128-byte line of the wave's
            // 2 KB half per instruction, and with reads in flight the half-written lines cost a sixth of the write rate
            // (tools/ubench/store_pattern3.hip: 3.9 -> 4.8 TB/s).  v_permlane32_swap puts both halves of the lower 32 lanes
            // into the first instruction (lanes >= 32 carry the second halves) and those of the upper 32 lanes into the second.
            // (the builtin, not inline asm: the instruction needs wait states after a vector write of its operands, which the
            // compiler only inserts for instructions it knows -- with asm the first word of a block's first row came out wrong)
            const u32x2 s0 = __builtin_amdgcn_permlane32_swap(d0.x, d1.x, false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane32_swap(d0.y, d1.y, false, false);
            const u32x2 s2 = __builtin_amdgcn_permlane32_swap(d0.z, d1.z, false, false);
            const u32x2 s3 = __builtin_amdgcn_permlane32_swap(d0.w, d1.w, false, false);
            d0 = make_uint4(s0.x, s1.x, s2.x, s3.x);
            d1 = make_uint4(s0.y, s1.y, s2.y, s3.y);
            *(uint4*)(outb + off + st_a) = d0;
            *(uint4*)(outb + off + st_b) = d1;
        } else {
            uint4* op = (uint4*)(outb + off);
            op[0] = d0;
            op[1] = d1;
        }
    };
    // (swp: exchange the store halves between lanes l and l + 32 -- every lane of the wave must be in this call)
    auto compute_row = [&](int ri, const RowIn& cur, const bool table, const bool swp) __attribute__((always_inline)) {
        float f[16];
        uint4 d0, d1;
        dense_row(ri, cur.words, cur.s, cur.m, table, d0, d1);
        if (r > 0) {
            if (RV > 0) {
                unpack8(d0, f);
                unpack8(d1, f + 8);
                const uint32_t fv[8] = {cur.fv0.x, cur.fv0.y, cur.fv0.z, cur.fv0.w, cur.fv1.x, cur.fv1.y, cur.fv1.z, cur.fv1.w};
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    float acc = f[j];   // the dot-product chain starts from the dequantized value (no zero-fill, no final add)
#pragma unroll
                    for (int c = 0; c < RV2; c++)
                        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, fv[c]), __builtin_bit_cast(half2_t, gb[j][c]), acc, false);
                    f[j] = acc;
                }
                // (v_cvt_pk_f16_f32: two results per conversion, round-to-nearest-even like the single one)
                d0 = make_uint4(f2h2_bits(f[0], f[1]), f2h2_bits(f[2], f[3]), f2h2_bits(f[4], f[5]), f2h2_bits(f[6], f[7]));
                d1 = make_uint4(f2h2_bits(f[8], f[9]), f2h2_bits(f[10], f[11]), f2h2_bits(f[12], f[13]), f2h2_bits(f[14], f[15]));
            } else {
                unpack8(d0, f);
                unpack8(d1, f + 8);
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
                d0 = pack8(f);
                d1 = pack8(f + 8);
            }
        }
        store_row(cur.off, d0, d1, swp);
    };
    // ---- a full block of 16 rows, every lane of the wave inside the row: straight-line code.  Sixteen unrolled steps, every
    // load issued by every lane, so the compiler's s_waitcnt for row i's inputs is exact -- "at most the loads of row i + 1 and
    // the stores of row i - 1 still in flight".  The general loop below issues its loads under run-time conditions, and at every
    // join the compiler can only wait for ALL outstanding memory operations: there the loads of row i + 1 were waited for
    // right after they had been issued and every row waited for the previous row's stores to be acknowledged (the kernel ran
    // at 2.2 - 2.5 TB/s where tools/ubench/store_pattern3.hip writes the same bytes at 4.5).
    // Short rows (rpar of them side by side, TB == 256 only): the same sixteen steps with rpar rows each; the table period in
    // steps is then a run-time number, so that flavour keeps a uniform branch around the fill.
    const bool fast = BITS <= 4 && !g.general && wave_full && (RV > 0 || r == 0) &&
                      (g.rpb == 16 * g.rpar || ((RV == 4 || RV == 8) && TB < 1024 && r > 0 && g.rpar == 1 && g.rpb % 8 == 0 && g.rpb > 16)) && nrows_blk == g.rpb &&
                      (g.rpar > 1 ? TB == 256 : g.trows == 4) && (!table || pf_ok);
    if (fast) {
        auto rows16 = [&](auto tc, auto rc) __attribute__((always_inline)) {
            constexpr bool TBL = decltype(tc)::value;
            constexpr bool R1 = decltype(rc)::value;          // one row per step (the table period is four steps)
            const int R = R1 ? 1 : g.rpar;
            // (not with 1024 threads -- rows of 16384 columns: 128 registers per lane do not hold the 64 accumulators and the factor block)
            if constexpr (R1 && (RV == 4 || RV == 8) && TB < 1024) {
                if (r > 0) {
                    // ---- low-rank term on the matrix cores, four rows at a time.  v_mfma_f32_4x4x4_16B_f16 computes, in every group
                    // of four lanes, D[i][j] = C[i][j] + sum_k A_(lane i)[k] B_(lane j)[k] with D[.][j] in lane j's four registers
                    // (tools/ubench/mfma4_layout.hip): lane 4 b + i supplies the row factor of the group's row i (ONE 8- or 16-byte load
                    // per lane and FOUR rows instead of one per row), every lane its own column's factor block, and the accumulators
                    // start from the four rows' dequantized values -- 2 (rank 8) matrix-core instructions per column and four rows
                    // where the vector ALU issued 16 v_dot2_f32_f16.  Results differ from the dot-product chain's in the order of
                    // the fp32 additions only (<= 1 ulp of the fp16 result).
                    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                    typedef float float4_t __attribute__((ext_vector_type(4)));
                    struct RowIn4 { uint32_t words[4][WPL]; float s[4], m[4]; uint4 fva; };
                    auto fetch4 = [&](int gi, RowIn4& in) __attribute__((always_inline)) {
                        const int p0 = phys(4 * gi);                     // (rot is a multiple of 4: a group stays a group)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int off = lane_off + (p0 + q) * istride;
                            const int gix = lane_g + (p0 + q) * gstep;
                            in.s[q] = ld_st<ST>(scaleb + gix);
                            in.m[q] = ld_st<ST>(mnb + gix);
#pragma unroll
                            for (int w = 0; w < WPL; w++) in.words[q][w] = codeb[off / CPW + w];
                        }
                        const uint16_t* fvp = fvb + fv_lane + (p0 + (tid & 3)) * r;
                        if (RV == 4) { const uint2 t = *(const uint2*)fvp; in.fva = make_uint4(t.x, t.y, 0, 0); }
                        else in.fva = *(const uint4*)fvp;
                    };
                    // (the conversion as the compiler's own v_cvt_pk_f16_f32, not common.h's inline-asm f2h2_bits: the matrix cores'
                    // results are not interlocked, the wait states in front of their first reader are the compiler's to insert, and it
                    // cannot see into an asm statement -- with f2h2_bits the first row of every group read accumulators that the
                    // later columns' instructions had not written yet)
                    typedef float float2c __attribute__((ext_vector_type(2)));
                    auto cvt2 = [](float a, float b) __attribute__((always_inline)) {
                        const float2c v = {a, b};
                        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2_t));
                    };
                    auto compute4 = [&](int gi, const RowIn4& in) __attribute__((always_inline)) {
                        const int p0 = phys(4 * gi);
                        float4_t acc[16];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            uint4 d0, d1;
                            dense_row(4 * gi + q, in.words[q], in.s[q], in.m[q], TBL, d0, d1);
                            float f[16];
                            unpack8(d0, f);
                            unpack8(d1, f + 8);
#pragma unroll
                            for (int j = 0; j < 16; j++) acc[j][q] = f[j];
                        }
                        const half4_t a_lo = __builtin_bit_cast(half4_t, make_uint2(in.fva.x, in.fva.y));
                        const half4_t a_hi = __builtin_bit_cast(half4_t, make_uint2(in.fva.z, in.fva.w));
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            acc[j] = __builtin_amdgcn_mfma_f32_4x4x4f16(a_lo, __builtin_bit_cast(half4_t, make_uint2(gb[j][0], gb[j][1 % RV2])), acc[j], 0, 0, 0);
                            if (RV == 8)
                                acc[j] = __builtin_amdgcn_mfma_f32_4x4x4f16(a_hi, __builtin_bit_cast(half4_t, make_uint2(gb[j][2 % RV2], gb[j][3 % RV2])), acc[j], 0, 0, 0);
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint4 d0 = make_uint4(cvt2(acc[0][q], acc[1][q]), cvt2(acc[2][q], acc[3][q]), cvt2(acc[4][q], acc[5][q]), cvt2(acc[6][q], acc[7][q]));
                            const uint4 d1 = make_uint4(cvt2(acc[8][q], acc[9][q]), cvt2(acc[10][q], acc[11][q]), cvt2(acc[12][q], acc[13][q]), cvt2(acc[14][q], acc[15][q]));
                            store_row(lane_off + (p0 + q) * istride, d0, d1, true);
                        }
                    };
                    auto refill = [&](int gi, int ng) __attribute__((always_inline)) {
                        if (gi) __syncthreads();            // everyone is done reading the previous fill
                        zero_table();
                        __syncthreads();
#pragma unroll
                        for (int q = 0; q < PF; q++)
                            if (pf_rc[q] >= 0 && (int)pf_idx[q] < g.len) lval[(pf_rc[q] >> 16) * g.len + pf_idx[q]] = (uint16_t)~pf_val[q];
                        prefetch_entries_all(4 * min(gi + 1, ng - 1));
                        __syncthreads();
                    };
                    if (g.rpb == 16) {
                        // sixteen rows: straight-line code over the four groups
                        RowIn4 gbuf[2];
                        if (TBL) prefetch_entries_all(0);
                        fetch4(0, gbuf[0]);
#pragma unroll
                        for (int gi = 0; gi < 4; gi++) {
                            if (TBL) {
                                if (gi) __syncthreads();            // everyone is done reading the previous fill
                                zero_table();
                                __syncthreads();
#pragma unroll
                                for (int q = 0; q < PF; q++)
                                    if (pf_rc[q] >= 0 && (int)pf_idx[q] < g.len) lval[(pf_rc[q] >> 16) * g.len + pf_idx[q]] = (uint16_t)~pf_val[q];
                                if (gi + 1 < 4) prefetch_entries_all(4 * (gi + 1));
                                __syncthreads();
                            }
                            if (gi + 1 < 4) fetch4(gi + 1, gbuf[(gi + 1) & 1]);
                            compute4(gi, gbuf[gi & 1]);
                        }
                    } else {
                        // 32 .. 128 rows behind ONE load of the lane's column factor block (16 instructions that touch 64 lines each: paid
                        // per 16 rows it was 0.09 / 0.06 ms of K^T's / V's 0.37 ms at config 3): a rolled loop over pairs of groups, two
                        // group buffers in ping-pong, the prefetch behind the last group re-reads that group (unconditional loads: the
                        // compiler's wait counts stay exact)
                        const int ng = g.rpb >> 2;
                        RowIn4 gbufA, gbufB;
                        if (TBL) prefetch_entries_all(0);
                        fetch4(0, gbufA);
#pragma unroll 1
                        for (int gp = 0; gp < ng; gp += 2) {
                            if (TBL) refill(gp, ng);
                            fetch4(gp + 1, gbufB);
                            compute4(gp, gbufA);
                            if (TBL) refill(gp + 1, ng);
                            fetch4(min(gp + 2, ng - 1), gbufA);
                            compute4(gp + 1, gbufB);
                        }
                    }
                    return;
                }
            }
            RowIn buf[PFD + 1];
            if (TBL) prefetch_entries_all(0);
#pragma unroll
            for (int i = 0; i < PFD; i++) fetch(i * R + sub, buf[i]);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (TBL && (R1 ? (i & 3) == 0 : ((i * R) & (g.trows - 1)) == 0)) {
                    if (i) __syncthreads();            // everyone is done reading the previous fill
                    zero_table();
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < PF; q++)
                        if (pf_rc[q] >= 0 && (int)pf_idx[q] < g.len) lval[(pf_rc[q] >> 16) * g.len + pf_idx[q]] = (uint16_t)~pf_val[q];
                    if (R1) { if (i + 4 < 16) prefetch_entries_all(i + 4); }
                    else if (i * R + g.trows < g.rpb) prefetch_entries_all(i * R + g.trows);
                    __syncthreads();
                }
                if (i + PFD < 16) fetch((i + PFD) * R + sub, buf[(i + PFD) % (PFD + 1)]);
                compute_row(i * R + sub, buf[i % (PFD + 1)], TBL, true);
            }
        };
        if (TB == 256 && g.rpar > 1) {
            if (table) rows16(std::true_type{}, std::false_type{});
            else rows16(std::false_type{}, std::false_type{});
        } else {
            if (table) rows16(std::true_type{}, std::true_type{});
            else rows16(std::false_type{}, std::true_type{});
        }
    } else {
        // Two named row buffers in ping-pong (no register copies between them: a copy of a buffer whose loads are still in
        // flight makes the compiler wait for them on the spot, which is what a rotating "cur = next" pipeline did): the loads
        // of row i + 1 are issued before row i is computed and stored.
        // (step li of the loop: block rows li * rpar .. li * rpar + rpar - 1, this lane's is li * rpar + sub)
        if (table && pf_ok) prefetch_entries(0);
        const bool swp_gen = wave_full && g.rpar == 1;      // (rows side by side: the last step of a block may be partly empty)
        RowIn bufA = {}, bufB = {};
        const int R = g.rpar;
        if (sub < nrows) fetch(sub, bufA);
        for (int rb = 0; rb < nrows_blk; rb += 2 * R) {
            before_row(rb);
            if (active) {
                if (rb + R + sub < nrows) fetch(rb + R + sub, bufB);
                if (rb + sub < nrows) compute_row(rb + sub, bufA, table, swp_gen);
            }
            if (rb + R < nrows_blk) {
                before_row(rb + R);
                if (active) {
                    if (rb + 2 * R + sub < nrows) fetch(rb + 2 * R + sub, bufA);
                    if (rb + R + sub < nrows) compute_row(rb + R + sub, bufB, table, swp_gen);
                }
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
            if ((int)idx >= g.len) continue;                       // (unused slot of a head shard's list)
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
    if ((int)idx >= g.len) return;                                 // (unused slot of a head shard's list)
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

// (compiled twice: DEC_PART 0 = fp16 scale arithmetic, 1 = fp32 -- two translation units halve the build's critical path)
int DEC_ENTRY(const void* code, const void* scale, const void* mn, int64_t n_rows, int rows_inner,
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
    GEAR_CHECK_ARG(outer_stride / group < 0x7FFFFFFFLL && (int64_t)nseg * (nseg > 1 ? seg_stride : 0) + seglen + 128 * inner_stride < 0x7FFFFFFFLL,
                   "gear_decompress_rows: the rows of a block (up to 128) must span fewer than 2^31 elements");
    if (kind == 0) GEAR_CHECK_ARG(rows_inner == T && seglen == D, "gear_decompress_rows: kind 0 needs rows_inner == T and seglen == D");
    if (kind == 1) GEAR_CHECK_ARG(rows_inner == D && nseg == 1 && seglen == T, "gear_decompress_rows: kind 1 needs rows_inner == D, one segment of T");
    const int patch = 0;   // (an in-kernel global patch pass measured 0.81 ms vs 0.72 ms for the LDS table: not used)
    // rows per block, measured on the 7B / 4k tensors (V / K^T ms): factors only: 16 rows 0.48 / 0.42, 8 rows 0.51 / 0.47
    // (fewer reloads of the lane's factor block).  The LDS outlier table covers 4 rows (35 KB) and is refilled inside the
    // block: outliers + factors 0.58 / 0.54 with 16 rows per block (0.64 / 0.62 when the block itself was 4 rows, 0.70 /
    // 0.75 with an 8-row table = 70 KB = half the resident blocks)
    int rpb = patch ? 8 : 16;
    while (rpb > 1 && rows_inner % rpb != 0) rpb >>= 1;
    // short rows (head shards: 128 .. 512 elements): 8 / 4 / 2 rows side by side in the one wave of the block, and 16 such steps
    // per block -- the lane's factor block (r / 2 registers per column, 32 KB per wave at rank 16) is loaded once per block, and
    // with 16 ROWS per block a 128-element row cost eight times its own bytes in factor loads (config 5 on 8 GPUs: 0.216 ms)
    int rpar = 1;
    if (len / 16 < 64 && 64 % (len / 16) == 0) {
        const int rp = (int)(64 / (len / 16));
        int rb = 16 * rp;
        while (rb > rp && rows_inner % rb != 0) rb >>= 1;
        if (rows_inner % rb == 0) { rpar = rp; rpb = rb; }
    }
    // rank 4 / 8 on whole waves (the matrix-core group path): up to 128 rows behind one load of the column factor block, as long as
    // >= 1024 workgroups remain (4 per CU)
    if ((r == 4 || r == 8) && rpar == 1 && rpb == 16 && !patch && bits <= 4 && len % 1024 == 0 && len <= 8192 && !gear_options().decomp_general)
        for (int c = 128; c >= 32; c >>= 1) {
            const int forced = gear_options().decomp_rpb;           // (tests: 32 / 64 / 128 rows per block whatever the size; -1: always 16)
            if (rows_inner % c == 0 && n_rows % c == 0 && (forced > 0 ? c == forced : (forced == 0 && n_rows / c >= 1024))) { rpb = c; break; }
        }
    int trows = rpb < 4 ? rpb : 4;            // rows per fill of the LDS outlier table (35 KB at 4096 columns)
    if (trows > rpb) trows = rpb;
    while (trows > 1 && (rpb % trows != 0 || (size_t)trows * ((len / 32 + 1) * 4 + len * 2) > 72 * 1024)) trows >>= 1;
    if (trows < rpar) trows = rpar;           // one fill of the table covers at least the rows in flight
    const size_t shmem = (k > 0 && !patch) ? (size_t)trows * len * 2 : 0;
    GEAR_CHECK_ARG(shmem <= 72 * 1024, "gear_decompress_rows: row too long for the LDS outlier table");
    DGeom g{rows_inner, outer_stride, inner_stride, nseg, seglen, seg_stride, (int)len, group, T, D, r, k, rpb, patch, trows, rpar, n_rows,
            (int)(outer_stride / group), (int)(inner_stride / group), (int)((nseg > 1 ? seg_stride : 0) / group), 0,
            gear_options().decomp_general};
    g.xcd = kind == 1 && r > 0 && ((n_rows + rpb - 1) / rpb) % 8 == 0;
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
#if DEC_PART == 0
    if (bits == 2) GOK(2, 0, uint16_t);
    else if (bits == 4) GOK(4, 0, uint16_t);
    else GOK(8, 0, uint16_t);
#else
    if (bits == 2) GOK(2, 1, float);
    else if (bits == 4) GOK(4, 1, float);
    else GOK(8, 1, float);
#endif
#undef GOK
#undef GOR
#undef GO
#undef GOT
    GEAR_CHECK_LAUNCH("gear_decompress_rows");
    return 0;
}
