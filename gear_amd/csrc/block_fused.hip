// block_fused.hip -- the decode-time block boundary of the streaming cache in ONE launch.
//
// Every `residual` = 64 new tokens the attention hook compresses the fp16 window of every layer: K per channel over the 64 tokens,
// V per token (cuda_supported_gear/modeling_llamagear.py:265-286, :335-378: key_compression / value_compression on the
// 64-token block -> quantize + pack + error -> headwise_lrap), and the simulated path adds the sparse outliers of the block
// (GenerationBench/.../Simulated/compress_function.py:261-333).  The chain of kernels built for prefill-size tensors
// (kfused.hip, compress_rows.hip, lowrank_gram.hip, attention.hip's tile builders) takes ~10 launches for that; the tiles here
// are 64 x 128 fp16 = 16 KB per (layer, head, K | V), so the whole block fits on chip:
//
//   one wave = one tile, kept in LDS ([64 tokens][136 halves]: x first, overwritten in place by the error).  K tile: lane = channel
//   pair walking the 64 tokens -> exact row mean (fp64 accumulation of fp16 values), top / bottom-kk selection by repeated scans
//   (ties: lower token first), fill, group quantization along T, bit-packing into the channel-major K^T cache rows at the block's
//   token offset, the outlier lists and the block's entries of the 128-token sparse tile.  V tile: lane = token walking its 128
//   channels -> group quantization along D, packing, the 64-token sparse tile.  Both leave the fp16 error tile in LDS and run the
//   SAME low-rank step on it, written on the 64-dimensional token side because a block has only 64 rows:
//       Y0 = E P0 (matrix cores; P0 as fp16 head + remainder) ; G' = E E^T (64 x 64, matrix cores, accumulators stay in
//       registers) ; Y = G'^(loop-1) Y0 ; Q' = orth(Y) (CholeskyQR2, fp64 small Gram) ; P = E^T Q' (matrix cores, transposing
//       LDS reads)
//   which spans the same subspace as the reference's iteration  P <- (E^T E)^(loop-1) P0, orth, Q = E P, orth, P = E^T Q
//   (new_pack.py:298-304): span(Q') = span(E (E^T E)^(loop-1) P0) = span(G'^(loop-1) E P0), and Q' P^T = Q' Q'^T E depends
//   on that span only.  (The chain iterates on the 128 x 128 matrix E^T E, which is the cheaper side only for T > 128.)
//
//   V outliers are selected per TOKEN ROW ACROSS THE HEADS (gears_tokenQ, compress_function.py:297-333), i.e. across tiles.
//   Every workgroup does `rows_per_blk` "row duties" before its tile: one wave reads one token row of all H heads, finds the exact top / bottom-kv sets (16-round bisection on the 16-bit order key, both sides at once;
//   ties by index), writes the sorted lists, the chunk index bytes and -- write-through -- a 128-bit outlier mask per (row,
//   head) and the row mean, then raises the row's flag.  The V tiles (the LAST NB*H workgroups) poll the 64 flags of their
//   rows -- lane = token = row -- and read mask and mean with agent-scope loads.  A poll that outlasts its bound sets the status
//   word instead of hanging the GPU.  (Row duties never wait; a V tile's rows belong to workgroups dispatched before it or at most
//   63 after it, so forward progress needs 64 resident workgroups: one XCD holds 256.)
#include "common.h"
#include "lowrank_solve.h"
#include "ktile.h"

namespace {

typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int BT_PITCH = 136;        // halfs per tile row: 272 bytes = 17 x 16 (aligned 16-byte operand reads), 68 words

struct BlkArgs {
    const uint16_t* kwin;    // [NB*H][wcap][128] fp16 window of every (layer*batch, head): the block's K
    const uint16_t* vwin;
    int wcap;
    int NB, H;               // NB = layers * batch
    int t_off;               // tokens already compressed (multiple of 64): where the block goes
    // K payload (channel-major rows)
    uint32_t* kcode; uint16_t* kscale; uint16_t* kmn; int64_t ldk, lsk;
    uint16_t* koidx; uint16_t* koval; int kk, kk_cap, o_off;
    uint32_t* ktile; int* kcnt; int ktile_cap, nck;
    // V payload (token-major rows)
    uint32_t* vcode; uint16_t* vscale; uint16_t* vmn; int tcap;
    uint16_t* voidx; uint16_t* voval; int kv; uint8_t* vochunk;
    uint32_t* vtile; int* vcnt; int vtile_cap, nblk;
    // factors
    int rk, rv, loop;
    const float* P0k; const float* P0v;      // [NB*H][128][r]
    uint16_t* kP; uint16_t* vP; int64_t p_inner, kp_stride, vp_stride;
    uint16_t* kQ; uint16_t* vQ;              // [NB*H][tcap][r]
    // hand-off of the V row selection
    uint32_t* flags;         // [NB*64]
    float* vfill;            // [NB*64] row means
    uint32_t* vmask;         // [NB*64][H][4]: even channels (64 bit), odd channels (64 bit)
    uint32_t* status;
    uint32_t epoch;
    int rows_per_blk;
};

__device__ __forceinline__ uint32_t pkmaxu16(uint32_t a, uint32_t bb) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bb));
    return r;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_total_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}

// ================================================================================================ V row duty
// One wave, one token row (nb, t) across the H heads: lane = channel pair, word h of a lane = head h.  The row's order keys
// (two per word) live in LDS, kw[h * 64 + lane]: every lane reads only what it wrote, the loops over h stay rolled.
__device__ __forceinline__ void vrow_select(const BlkArgs& a, uint32_t* kw, int nb, int t, int lane) {
    const int H = a.H, kv = a.kv;
    const uint32_t* xrow = (const uint32_t*)(a.vwin + ((int64_t)nb * H * a.wcap + t) * KD) + lane;      // + h * wcap * 64 words
    const int64_t hstride = (int64_t)a.wcap * (KD / 2);
    const int H8 = (H + 7) & ~7;
    // the row's words stay in LDS behind the keys when both fit (up to 34 heads): the emission loop then needs no second trip to
    // global memory (its eight-loads-at-a-time round trips were a third of this function)
    const bool x_in_lds = 2 * H8 * 64 * 4 <= 64 * BT_PITCH * 2;
    uint32_t* xs = kw + H8 * 64;
    double s = 0.0;
#pragma unroll 1
    for (int h0 = 0; h0 < H; h0 += 16) {             // sixteen heads' loads in flight at a time
        uint32_t w16[16];
#pragma unroll
        for (int j = 0; j < 16; j++) w16[j] = (h0 + j < H) ? xrow[(h0 + j) * hstride] : 0u;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (h0 + j < H) {
                s += (double)h2f_bits((uint16_t)(w16[j] & 0xFFFFu)) + (double)h2f_bits((uint16_t)(w16[j] >> 16));
                kw[(h0 + j) * 64 + lane] = sort_key16(w16[j] & 0xFFFFu) | (sort_key16(w16[j] >> 16) << 16);
                if (x_in_lds) xs[(h0 + j) * 64 + lane] = w16[j];
            }
        }
    }
    for (int h = H; h < H8; h++) kw[h * 64 + lane] = 0u;        // padding words: key 0 (below every finite value's key)
    s = wave_sum_f64(s);
    const float mean = (float)(s / (double)(H * KD));
    const int row = nb * 64 + t;
    // ---- thresholds: large side = the largest K with #{key >= K} >= kv; small side = the smallest K with #{key <= K} >= kv
    // Wave-wide counts through the scalar unit: one compare per key (the high halves as 32-bit compares against the threshold
    // shifted up, the low halves as 16-bit compares), ballot -> s_bcnt1 -> scalar add, eight words per trip with their LDS reads
    // issued together (the ballots are convergent operations: the compiler does not unroll a loop around them by itself, and one
    // word per trip -- read, wait, four compares, four scalar counts -- made this function 60 us of a 200 us kernel).  The padding
    // keys are never >= a threshold (thresholds start at 1) and always <= one: their count is taken off the small side.
    const uint32_t padle = (uint32_t)(2 * (H8 - H) * 64);
    auto counts = [&](uint32_t midL, uint32_t midS) -> uint32_t {   // (#{key >= midL}) | (#{key <= midS}) << 16 over the row
        const uint32_t upL = midL << 16, upS = (midS << 16) | 0xFFFFu;
        const uint16_t loLv = (uint16_t)midL, loSv = (uint16_t)midS;
        uint32_t nL = 0u, nS = 0u;
#pragma unroll 1
        for (int h0 = 0; h0 < H8; h0 += 8) {
            uint32_t k8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) k8[j] = kw[(h0 + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                nL += (uint32_t)__popcll(__ballot(k8[j] >= upL)) + (uint32_t)__popcll(__ballot((uint16_t)k8[j] >= loLv));
                nS += (uint32_t)__popcll(__ballot(k8[j] <= upS)) + (uint32_t)__popcll(__ballot((uint16_t)k8[j] <= loSv));
            }
        }
        return nL | ((nS - padle) << 16);
    };
    uint32_t loL = 1u, hiL = 0xFFFFu, loS = 0u, hiS = 0xFFFFu;
#pragma unroll 1
    for (int it = 0; it < 16; it++) {
        const uint32_t midL = loL + ((hiL - loL + 1u) >> 1), midS = loS + ((hiS - loS) >> 1);
        const uint32_t c = counts(midL, midS);
        if (loL < hiL) { if ((int)(c & 0xFFFFu) >= kv) loL = midL; else hiL = midL - 1u; }
        if (loS < hiS) { if ((int)(c >> 16) >= kv) hiS = midS; else loS = midS + 1u; }
    }
    // (16 rounds settle both brackets: the large one halves a range of 65535 keys, the small one of 65536; every finite fp16 value
    // has a key >= 0x03FF, so #{key >= 1} is the whole row)
    const uint32_t vL = loL, vS = loS;
    const uint32_t cge = counts(vL, vS);
    const uint32_t cgt = counts(vL == 0xFFFFu ? 0xFFFFu : vL + 1u, vS == 0u ? 0u : vS - 1u);
    const int n_geL = (int)(cge & 0xFFFFu), n_leS = (int)(cge >> 16);
    const int n_gtL = (vL == 0xFFFFu) ? 0 : (int)(cgt & 0xFFFFu), n_ltS = (vS == 0u) ? 0 : (int)(cgt >> 16);
    const int needL = kv - n_gtL, needS = kv - n_ltS;            // ties taken at the threshold value, in index order
    const bool part = n_geL > kv || n_leS > kv;
    // ---- emission in index order (head, lane, half)
    uint16_t* oi = a.voidx + ((int64_t)nb * a.tcap + a.t_off + t) * (2 * kv);
    uint16_t* ov = a.voval + ((int64_t)nb * a.tcap + a.t_off + t) * (2 * kv);
    int baseS = 0, baseL = 0, tieS = 0, tieL = 0;
    uint32_t mk0 = 0u, mk1 = 0u, mk2 = 0u, mk3 = 0u;             // lane h keeps head h's mask
    uint32_t chS = 0u, chL = 0u;                                  // lane h keeps the list positions at the start of head h
#pragma unroll 1
    for (int h0 = 0; h0 < H; h0 += 8) {              // the values of eight heads are loaded together (an L2 round trip per head
        uint32_t xw8[8];                              // in a dependent loop was most of this function's time)
        // (clamped indices, no load under a per-element condition: `h < H ? load : 0` compiles to a load and a full wait per head)
        if (x_in_lds) {
#pragma unroll
            for (int j = 0; j < 8; j++) xw8[j] = xs[min(h0 + j, H - 1) * 64 + lane];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) xw8[j] = xrow[(int64_t)min(h0 + j, H - 1) * hstride];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int h = h0 + j;
            if (h < H) {
                const uint32_t k2 = kw[h * 64 + lane];
                const uint32_t xw = xw8[j];
                const uint32_t kA = k2 & 0xFFFFu, kB = k2 >> 16;
                bool sLA = kA > vL, sLB = kB > vL, sSA = kA < vS, sSB = kB < vS;
                const bool eLA = kA == vL, eLB = kB == vL, eSA = kA == vS, eSB = kB == vS;
                if (part) {
                    const uint32_t c = (uint32_t)((eLA ? 1 : 0) + (eLB ? 1 : 0)) | ((uint32_t)((eSA ? 1 : 0) + (eSB ? 1 : 0)) << 16);
                    const uint32_t inc = wave_incl_scan_u32(c);
                    const uint32_t exc = inc - c;
                    const int rL = tieL + (int)(exc & 0xFFFFu), rS = tieS + (int)(exc >> 16);
                    sLA |= eLA && rL < needL;
                    sLB |= eLB && (rL + (eLA ? 1 : 0)) < needL;
                    sSA |= eSA && rS < needS;
                    sSB |= eSB && (rS + (eSA ? 1 : 0)) < needS;
                    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    tieL += (int)(tot & 0xFFFFu);
                    tieS += (int)(tot >> 16);
                } else {
                    sLA |= eLA; sLB |= eLB; sSA |= eSA; sSB |= eSB;
                }
                const uint32_t c = (uint32_t)((sLA ? 1 : 0) + (sLB ? 1 : 0)) | ((uint32_t)((sSA ? 1 : 0) + (sSB ? 1 : 0)) << 16);
                const uint32_t inc = wave_incl_scan_u32(c);
                const uint32_t exc = inc - c;
                int pL = baseL + (int)(exc & 0xFFFFu), pS = baseS + (int)(exc >> 16);
                const uint16_t col = (uint16_t)(h * KD + 2 * lane);
                if (sSA) { oi[pS] = col; ov[pS] = (uint16_t)(xw & 0xFFFFu); pS++; }
                if (sSB) { oi[pS] = (uint16_t)(col + 1); ov[pS] = (uint16_t)(xw >> 16); }
                if (sLA) { oi[kv + pL] = col; ov[kv + pL] = (uint16_t)(xw & 0xFFFFu); pL++; }
                if (sLB) { oi[kv + pL] = (uint16_t)(col + 1); ov[kv + pL] = (uint16_t)(xw >> 16); }
                const uint64_t bA = __ballot(sLA || sSA), bB = __ballot(sLB || sSB);
                if (lane == h) {
                    mk0 = (uint32_t)bA; mk1 = (uint32_t)(bA >> 32); mk2 = (uint32_t)bB; mk3 = (uint32_t)(bB >> 32);
                    chS = (uint32_t)baseS; chL = (uint32_t)baseL;
                }
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                baseL += (int)(tot & 0xFFFFu);
                baseS += (int)(tot >> 16);
            }
        }
    }
    if (a.vochunk) {
        const int64_t id = ((int64_t)nb * a.tcap + a.t_off + t) * 2;
        if (lane < H) {
            a.vochunk[id * (H + 1) + lane] = (uint8_t)chS;
            a.vochunk[(id + 1) * (H + 1) + lane] = (uint8_t)chL;
        }
        // the terminal entry (end of the last head's range) from lane 0: with H == 64 there is no lane H
        if (lane == 0) {
            a.vochunk[id * (H + 1) + H] = (uint8_t)kv;
            a.vochunk[(id + 1) * (H + 1) + H] = (uint8_t)kv;
        }
    }
    // ---- hand-off to the V tiles: write-through (agent-scope) stores, drained, then the flag
    if (lane < H) {
        gu64* mp = (gu64*)(a.vmask + ((int64_t)row * H + lane) * 4);
        __hip_atomic_store(mp, (unsigned long long)mk0 | ((unsigned long long)mk1 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mp + 1, (unsigned long long)mk2 | ((unsigned long long)mk3 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) __hip_atomic_store((gu32*)(a.vfill + row), __builtin_bit_cast(uint32_t, mean), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store((gu32*)(a.flags + row), a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ================================================================================================ low-rank step on the LDS tile
// LDS plan (RP = 8: 20 480 bytes per tile with the 136-half row pitch, i.e. eight tiles per CU -- all 2048 tiles of a Llama-2-7B
// boundary resident at once; the first version's 28 KB gave five per CU and two rounds): the tile, ONE [64][RP] float buffer
// (Y, rewritten in place behind a barrier; at the end it holds Q' as matrix-core operand), R^-1 and R (fp64), and the small
// Gram matrix in the 16 padding bytes of the tile's rows (RP * RP <= 64 entries).  P0 goes from global memory straight into
// the A operand registers.
__host__ __device__ constexpr size_t blk_lr_lds_bytes(int RP) {      // behind the tile
    return (size_t)64 * RP * 4 + (size_t)2 * RP * RP * 8 + (RP * RP <= 64 ? 0 : (size_t)RP * RP * 8);
}

// MFMA B operand of lane (x31, kg): channel 32 I + x31, tokens t0 + 8 kg .. + 7, through the transposing LDS read (ktile.h's
// load_operand with this kernel's row pitch)
__device__ __forceinline__ half8_t load_operand_bt(const uint16_t* tile, int t0, int I, int lane) {
    const int kg = lane >> 5, i = lane & 15, c0 = 32 * I + 16 * ((lane >> 4) & 1);
    const uint16_t* p = tile + (t0 + 8 * kg + (i >> 2)) * BT_PITCH + c0 + 4 * (i & 3);
    const uint32_t addr = (uint32_t)(uintptr_t)p;
    typedef short short4v __attribute__((ext_vector_type(4)));
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * BT_PITCH * 2) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    union { half8_t h; short4v s[2]; } cv;
    cv.s[0] = lo;
    cv.s[1] = hi;
    return cv.h;
}

// tile: fp16 error [64 tokens][BT_PITCH] in LDS (complete, visible).  P0h: float [128][r] of this head (global).  Writes P_out fp16
// [128][r] and Q_out fp16 [64][r] (the block's 64 token rows).  One wave.
template <int RP>
__device__ __forceinline__ void lowrank_tile(uint16_t* tile, unsigned char* sm, const float* __restrict__ P0h, int r, int loop,
                                             uint16_t* __restrict__ P_out, uint16_t* __restrict__ Q_out, int lane) {
    constexpr bool MD_IN_PAD = RP * RP <= 64;
    constexpr int MDS = MD_IN_PAD ? BT_PITCH / 4 : 1;   // doubles from one Md entry to the next (row padding: one entry per row)
    float* Y = (float*)sm;                              // [64][RP]
    uint16_t* Ah = (uint16_t*)Y;                        // at the end: Q' as operand [8][RP][8] halves, head ...
    uint16_t* Al = Ah + 64 * RP;                        // ... and remainder
    double* Rinv = (double*)(Y + 64 * RP);              // [2][RP][RP]
    double* Md = MD_IN_PAD ? (double*)(tile + KD) : Rinv + 2 * RP * RP;
    const int n = lane & 31, kg = lane >> 5;
    union U { uint4 u; half8_t h; };
    // ---- A operand of Y0 = E P0 from global: lane (m = n < RP, kg) holds P0[16 ks + 8 kg + j][m], j < 8, as head + remainder
    float pw[8][8];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
        for (int j = 0; j < 8; j++) pw[ks][j] = (n < r) ? P0h[(16 * ks + 8 * kg + j) * r + n] : 0.0f;
    // ---- Y0 = E P0: C[m = rank column][n = token]
#pragma unroll
    for (int half = 0; half < 2; half++) {
        float16_t acc;
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            union { half8_t h; uint16_t u[8]; } ah, al;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                ah.u[j] = f2h_bits(pw[ks][j]);
                al.u[j] = f2h_bits(pw[ks][j] - h2f_bits(ah.u[j]));
            }
            U b;
            b.u = *(const uint4*)(tile + (32 * half + n) * BT_PITCH + 16 * ks + 8 * kg);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, b.h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, b.h, acc, 0, 0, 0);
        }
#pragma unroll
        for (int qb = 0; qb < (RP + 7) / 8; qb++) {
            const int c0 = 8 * qb + 4 * kg;
            if (c0 < RP) *(float4*)&Y[(32 * half + n) * RP + c0] = make_float4(acc[4 * qb], acc[4 * qb + 1], acc[4 * qb + 2], acc[4 * qb + 3]);
        }
    }
    // ---- G' = E E^T: g[I][J][q] = G'[32 I + (q & 3) + 8 (q >> 2) + 4 kg][32 J + n]
    float16_t g00, g01, g10, g11;
    if (loop > 1) {
#pragma unroll
        for (int q = 0; q < 16; q++) { g00[q] = 0.0f; g01[q] = 0.0f; g10[q] = 0.0f; g11[q] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            U f0, f1;
            f0.u = *(const uint4*)(tile + n * BT_PITCH + 16 * ks + 8 * kg);
            f1.u = *(const uint4*)(tile + (32 + n) * BT_PITCH + 16 * ks + 8 * kg);
            g00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.h, f0.h, g00, 0, 0, 0);
            g01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0.h, f1.h, g01, 0, 0, 0);
            g10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1.h, f0.h, g10, 0, 0, 0);
            g11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1.h, f1.h, g11, 0, 0, 0);
        }
    }
    __syncthreads();
    // ---- Y <- G' Y, loop - 1 times, in place (G' symmetric: column 32 J + n of G' = this lane's accumulators over its 32 rows)
#pragma unroll 1
    for (int it = 0; it + 1 < loop; it++) {
        float s0[RP], s1[RP];
#pragma unroll
        for (int c = 0; c < RP; c++) { s0[c] = 0.0f; s1[c] = 0.0f; }
#pragma unroll
        for (int I = 0; I < 2; I++) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = 32 * I + (q & 3) + 8 * (q >> 2) + 4 * kg;
                const float ga = I == 0 ? g00[q] : g10[q], gb = I == 0 ? g01[q] : g11[q];
#pragma unroll
                for (int c4 = 0; c4 < RP; c4 += 4) {
                    const float4 y = *(const float4*)&Y[row * RP + c4];
                    s0[c4] = fmaf(ga, y.x, s0[c4]); s0[c4 + 1] = fmaf(ga, y.y, s0[c4 + 1]);
                    s0[c4 + 2] = fmaf(ga, y.z, s0[c4 + 2]); s0[c4 + 3] = fmaf(ga, y.w, s0[c4 + 3]);
                    s1[c4] = fmaf(gb, y.x, s1[c4]); s1[c4 + 1] = fmaf(gb, y.y, s1[c4 + 1]);
                    s1[c4 + 2] = fmaf(gb, y.z, s1[c4 + 2]); s1[c4 + 3] = fmaf(gb, y.w, s1[c4 + 3]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < RP; c++) {
            s0[c] += __shfl_xor(s0[c], 32, 64);
            s1[c] += __shfl_xor(s1[c], 32, 64);
        }
        __syncthreads();                                  // every read of the old Y is done
        const int orow = kg == 0 ? n : 32 + n;
#pragma unroll
        for (int c4 = 0; c4 < RP; c4 += 4)
            *(float4*)&Y[orow * RP + c4] = kg == 0 ? make_float4(s0[c4], s0[c4 + 1], s0[c4 + 2], s0[c4 + 3])
                                                    : make_float4(s1[c4], s1[c4 + 1], s1[c4 + 2], s1[c4 + 3]);
        __syncthreads();
    }
    // ---- Q' = orth(Y): CholeskyQR twice, small Gram in fp64; lane = token row
    float yr[RP];
#pragma unroll
    for (int c4 = 0; c4 < RP; c4 += 4) {
        const float4 y = *(const float4*)&Y[lane * RP + c4];
        yr[c4] = y.x; yr[c4 + 1] = y.y; yr[c4 + 2] = y.z; yr[c4 + 3] = y.w;
    }
#pragma unroll 1
    for (int rep = 0; rep < 2; rep++) {
        for (int o = lane; o < RP * RP; o += 64) {
            const int ca = o / RP, cb = o % RP;
            double sacc = 0.0;
#pragma unroll 8
            for (int i = 0; i < 64; i++) sacc += (double)Y[i * RP + ca] * (double)Y[i * RP + cb];
            Md[o * MDS] = sacc;
        }
        __syncthreads();
        chol_inverse_wave<RP, MDS>(Md, Rinv, lane);
        __syncthreads();
        float yn[RP];
#pragma unroll
        for (int c = 0; c < RP; c++) {
            double sacc = 0.0;
#pragma unroll
            for (int ca = 0; ca < RP; ca++)
                if (ca <= c) sacc += (double)yr[ca] * Rinv[ca * RP + c];
            yn[c] = (float)sacc;
        }
#pragma unroll
        for (int c = 0; c < RP; c++) yr[c] = yn[c];
        if (rep == 0) {
#pragma unroll
            for (int c4 = 0; c4 < RP; c4 += 4) *(float4*)&Y[lane * RP + c4] = make_float4(yr[c4], yr[c4 + 1], yr[c4 + 2], yr[c4 + 3]);
        }
        __syncthreads();
    }
    // ---- Q' out (fp16) and, over the dead Y, as matrix-core operand [k / 8][m][k % 8], k = token
    {
        uint16_t qh[RP];
#pragma unroll
        for (int c = 0; c < RP; c++) {
            qh[c] = f2h_bits(yr[c]);
            const int pos = ((lane >> 3) * RP + c) * 8 + (lane & 7);
            Ah[pos] = qh[c];
            Al[pos] = f2h_bits(yr[c] - h2f_bits(qh[c]));
        }
        uint16_t* qo = Q_out + (int64_t)lane * r;
        if (r == RP) {
#pragma unroll
            for (int c4 = 0; c4 < RP; c4 += 4)
                *(uint2*)(qo + c4) = make_uint2((uint32_t)qh[c4] | ((uint32_t)qh[c4 + 1] << 16), (uint32_t)qh[c4 + 2] | ((uint32_t)qh[c4 + 3] << 16));
        } else {
#pragma unroll
            for (int c = 0; c < RP; c++)
                if (c < r) qo[c] = qh[c];
        }
    }
    __syncthreads();
    // ---- P = E^T Q': C[m = rank column][n = channel], k = token
    float16_t pacc[4];
#pragma unroll
    for (int J = 0; J < 4; J++)
#pragma unroll
        for (int q = 0; q < 16; q++) pacc[J][q] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        U ah, al;
        ah.u = al.u = make_uint4(0, 0, 0, 0);
        if (n < RP) {
            ah.u = *(const uint4*)&Ah[((2 * ks + kg) * RP + n) * 8];
            al.u = *(const uint4*)&Al[((2 * ks + kg) * RP + n) * 8];
        }
#pragma unroll
        for (int J = 0; J < 4; J++) {
            const half8_t b = load_operand_bt(tile, 16 * ks, J, lane);
            pacc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, b, pacc[J], 0, 0, 0);
            pacc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, b, pacc[J], 0, 0, 0);
        }
    }
#pragma unroll
    for (int J = 0; J < 4; J++) {
        const int ch = 32 * J + n;
#pragma unroll
        for (int qb = 0; qb < (RP + 7) / 8; qb++) {
            const int c0 = 8 * qb + 4 * kg;
            if (c0 >= RP) continue;
            if (r == RP) {
                uint2 v;
                v.x = (uint32_t)f2h_bits(pacc[J][4 * qb]) | ((uint32_t)f2h_bits(pacc[J][4 * qb + 1]) << 16);
                v.y = (uint32_t)f2h_bits(pacc[J][4 * qb + 2]) | ((uint32_t)f2h_bits(pacc[J][4 * qb + 3]) << 16);
                *(uint2*)(P_out + (int64_t)ch * r + c0) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (c0 + i < r) P_out[(int64_t)ch * r + c0 + i] = f2h_bits(pacc[J][4 * qb + i]);
            }
        }
    }
}

// ================================================================================================ the kernel
// grid 2 * NB * H workgroups of one wave: [0, NB*H) = K tiles (+ the V row duties), [NB*H, 2 NB*H) = V tiles.
// The tile [64 tokens][BT_PITCH] lives in LDS from the first load to the last matrix-core read: x first, overwritten in place
// by the error; the loops over tokens / channels stay rolled (a first, register-resident version unrolled everything: 75 000
// instructions per kernel, 117 spilled registers).
template <int BITS, int G, int RP>
__global__ __launch_bounds__(64) void block_compress_kernel(BlkArgs a) {
    constexpr int CPW = 32 / BITS;
    constexpr int LEVELS = (1 << BITS) - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* tile = (uint16_t*)smem;                                    // [64][BT_PITCH]
    uint32_t* tw = (uint32_t*)smem;                                      // the same as words: row pitch BT_PITCH / 2
    constexpr int WP = BT_PITCH / 2;
    unsigned char* lrsm = smem + (size_t)64 * BT_PITCH * 2;
    const int lane = threadIdx.x;
    const int64_t NBH = (int64_t)a.NB * a.H;
    const bool isK = (int64_t)blockIdx.x < NBH;
    const int64_t bh = isK ? (int64_t)blockIdx.x : (int64_t)blockIdx.x - NBH;
    const int tok0 = a.t_off;
    int r;
    const float* P0h = nullptr;
    uint16_t* P_out = nullptr;
    uint16_t* Q_out = nullptr;

    // ---------------------------------------------------------------- V row duties: every workgroup, before its tile
    // (a row duty never waits, so the V tiles -- which then wait for their 64 rows -- cannot starve a producer: the rows of
    // group nb belong to workgroups around 64 nb / rows_per_blk, all of which start, do their rows first and are dispatched
    // before or at most 63 workgroups after the waiting tile)
    if (a.kv > 0) {
        for (int j = 0; j < a.rows_per_blk; j++) {
            const int64_t row = (int64_t)blockIdx.x * a.rows_per_blk + j;
            if (row >= (int64_t)a.NB * 64) break;
            vrow_select(a, tw, (int)(row >> 6), (int)(row & 63), lane);
        }
    }
    if (isK) {
        // ---------------------------------------------------------------- K tile: lane = channel pair
        constexpr int NG = 64 / G;
        const uint16_t* xb = a.kwin + bh * (int64_t)a.wcap * KD;
        {
            const uint32_t* xw = (const uint32_t*)xb + lane;
#pragma unroll 32
            for (int i = 0; i < 64; i++) tw[i * WP + lane] = xw[i * 64];
        }
        const int kk = a.kk;
        uint64_t cLA = 0, cLB = 0, cSA = 0, cSB = 0;                    // chosen tokens: large / small side, channel A / B
        float fillA = 0.f, fillB = 0.f;
        if (kk > 0) {
            double sA = 0.0, sB = 0.0;                                   // exact: fp16 values accumulate without rounding in fp64
#pragma unroll 8
            for (int i = 0; i < 64; i++) {
                const uint32_t w = tw[i * WP + lane];
                sA += (double)h2f_bits((uint16_t)(w & 0xFFFFu));
                sB += (double)h2f_bits((uint16_t)(w >> 16));
            }
            fillA = hround((float)(sA / 64.0));
            fillB = hround((float)(sB / 64.0));
#pragma unroll 1
            for (int side = 0; side < 2; side++) {        // 0: the kk largest, 1: the kk smallest; ties -> lower token first
                const float sgn = side ? -1.0f : 1.0f;    // compare sgn * value: the small side is the large side of -x
                uint64_t cA = 0, cB = 0;
#pragma unroll 1
                for (int p = 0; p < kk; p++) {
                    float bA = -INFINITY, bB = -INFINITY;
                    int iA = 64, iB = 64;
#pragma unroll 4
                    for (int tk = 0; tk < 64; tk++) {
                        const uint32_t w = tw[tk * WP + lane];
                        const float va = sgn * h2f_bits((uint16_t)(w & 0xFFFFu)), vb = sgn * h2f_bits((uint16_t)(w >> 16));
                        const bool ta = ((cA >> tk) & 1ull) == 0ull && (iA == 64 || va > bA);
                        const bool tb = ((cB >> tk) & 1ull) == 0ull && (iB == 64 || vb > bB);
                        bA = ta ? va : bA; iA = ta ? tk : iA;
                        bB = tb ? vb : bB; iB = tb ? tk : iB;
                    }
                    cA |= 1ull << iA;
                    cB |= 1ull << iB;
                }
                if (side == 0) { cLA = cA; cLB = cB; } else { cSA = cA; cSB = cB; }
            }
        }
        const uint64_t oA = cLA | cSA, oB = cLB | cSB;
        float dqA[NG], dqB[NG];                   // what the attention reconstructs at a filled position of group gi
        QuantParams<0> qAs[NG], qBs[NG];
        uint32_t* codeA = a.kcode + (bh * KD + 2 * lane) * a.ldk + tok0 / CPW;
        uint32_t* codeB = codeA + a.ldk;
        uint16_t* sclA = a.kscale + (bh * KD + 2 * lane) * a.lsk + tok0 / G;
        uint16_t* mnlA = a.kmn + (bh * KD + 2 * lane) * a.lsk + tok0 / G;
        // ---- pass 1: the groups' parameters
#pragma unroll
        for (int gi = 0; gi < NG; gi++) {
            float loA = INFINITY, hiA = -INFINITY, loB = INFINITY, hiB = -INFINITY;
#pragma unroll 8
            for (int i = 0; i < G; i++) {
                const int tk = gi * G + i;
                const uint32_t w = tw[tk * WP + lane];
                const float xa = h2f_bits((uint16_t)(w & 0xFFFFu)), xbv = h2f_bits((uint16_t)(w >> 16));
                const float va = ((oA >> tk) & 1ull) ? fillA : xa, vb = ((oB >> tk) & 1ull) ? fillB : xbv;
                loA = fminf(loA, va); hiA = fmaxf(hiA, va);
                loB = fminf(loB, vb); hiB = fmaxf(hiB, vb);
            }
            qAs[gi] = make_qparams<0>(loA, hiA, LEVELS);
            qBs[gi] = make_qparams<0>(loB, hiB, LEVELS);
            sclA[gi] = f2h_bits(qAs[gi].scale); mnlA[gi] = f2h_bits(qAs[gi].mn);
            sclA[a.lsk + gi] = f2h_bits(qBs[gi].scale); mnlA[a.lsk + gi] = f2h_bits(qBs[gi].mn);
            dqA[gi] = fmaf(qAs[gi].scale, (float)quant_one<0>(fillA, qAs[gi]), qAs[gi].mn);
            dqB[gi] = fmaf(qBs[gi].scale, (float)quant_one<0>(fillB, qBs[gi]), qBs[gi].mn);
        }
        // ---- sparse part (x is still in the LDS tile): sorted lists (slot 0 = smallest, slot 1 = largest) + this block's entries
        // of the 128-token tile
        if (kk > 0) {
            const int chunk = tok0 >> 7, tin = tok0 & 127;
            int tbase = -1;                       // first tile entry of this block (-1: the tile is not maintained / overflowed)
            if (a.ktile && a.kcnt) {
                int* cp = a.kcnt + bh * a.nck + chunk;
                const int old = tin ? *cp : 0;
                const int tot = old + 2 * KD * kk;              // 128 channels x 2 sides x kk
                const bool ok = old >= 0 && tot <= a.ktile_cap;
                tbase = ok ? old : -1;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) *cp = ok ? tot : -1;
            }
            uint32_t* kt = a.ktile ? a.ktile + (bh * a.nck + chunk) * (int64_t)a.ktile_cap : nullptr;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int ch = 2 * lane + h;
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    uint64_t m = side == 0 ? (h ? cLB : cLA) : (h ? cSB : cSA);
                    const int64_t lbase = ((bh * KD + ch) * 2 + (side == 0 ? 1 : 0)) * (int64_t)a.kk_cap + a.o_off;
                    int pos = 0;
                    while (m) {
                        const int tk = __builtin_ctzll(m);
                        m &= m - 1ull;
                        const uint16_t vb = (uint16_t)(tw[tk * WP + lane] >> (16 * h));
                        a.koidx[lbase + pos] = (uint16_t)(tok0 + tk);
                        a.koval[lbase + pos] = vb;
                        if (tbase >= 0) {
                            const float dqv = (NG == 1 || tk < G) ? (h ? dqB[0] : dqA[0]) : (h ? dqB[NG - 1] : dqA[NG - 1]);
                            kt[tbase + (ch * 2 + side) * kk + pos] =
                                (uint32_t)ch | ((uint32_t)(tin + tk) << 7) | ((uint32_t)f2h_bits(h2f_bits(vb) - dqv) << 16);
                        }
                        pos++;
                    }
                }
            }
        }
        // ---- pass 2: codes, packed along T, and the error in place of x
#pragma unroll
        for (int gi = 0; gi < NG; gi++) {
            const QuantParams<0> qA = qAs[gi], qB = qBs[gi];
#pragma unroll 1
            for (int wd = 0; wd < G / CPW; wd++) {
                uint32_t cwA = 0u, cwB = 0u;
#pragma unroll 4
                for (int jj = 0; jj < CPW; jj++) {
                    const int tk = gi * G + wd * CPW + jj;
                    const uint32_t w = tw[tk * WP + lane];
                    const bool isoA = ((oA >> tk) & 1ull) != 0ull, isoB = ((oB >> tk) & 1ull) != 0ull;
                    const float va = isoA ? fillA : h2f_bits((uint16_t)(w & 0xFFFFu)), vb = isoB ? fillB : h2f_bits((uint16_t)(w >> 16));
                    const int qa = quant_one<0>(va, qA), qb = quant_one<0>(vb, qB);
                    cwA |= (uint32_t)qa << (BITS * jj);
                    cwB |= (uint32_t)qb << (BITS * jj);
                    const float ea = isoA ? 0.0f : (va - dequant_one<0>(qa, qA.scale, qA.mn));
                    const float eb = isoB ? 0.0f : (vb - dequant_one<0>(qb, qB.scale, qB.mn));
                    tw[tk * WP + lane] = (uint32_t)f2h_bits(ea) | ((uint32_t)f2h_bits(eb) << 16);      // the error, in place
                }
                codeA[gi * (G / CPW) + wd] = cwA;
                codeB[gi * (G / CPW) + wd] = cwB;
            }
        }
        r = a.rk;
        if (r > 0) {
            P0h = a.P0k + bh * KD * r;
            P_out = a.kP + (bh / a.p_inner) * a.kp_stride + (bh % a.p_inner) * (int64_t)(KD * r);
            Q_out = a.kQ + (bh * a.tcap + tok0) * (int64_t)r;
        }
    } else {
        // ---------------------------------------------------------------- V tile: lane = token, its row of the LDS tile
        constexpr int NGV = KD / G;
        const int nb = (int)(bh / a.H), hh = (int)(bh % a.H);
        const uint16_t* xrow = a.vwin + (bh * a.wcap + lane) * (int64_t)KD;
        uint4* trow = (uint4*)(tile + lane * BT_PITCH);                  // 16 chunks of 8 channels
#pragma unroll
        for (int j = 0; j < 16; j++) trow[j] = ((const uint4*)xrow)[j];
        uint64_t mE = 0, mO = 0;                  // outlier bits of the even / odd channels (bit w = channel 2w / 2w + 1)
        float fill = 0.f;
        if (a.kv > 0 && a.H == 1) {
            // ONE KV head per rank (70B head shards): the token row IS this lane's 128 channels, so the tile selects for itself --
            // no row duties, no flags, no waiting (with H < 4 the 64 / (2 H) serial row duties per workgroup cost more than the
            // launches they saved: 367 us against the chain's 222 at 80 layers x 1 head).  Exact fp64 row mean; top / bottom-kv
            // by repeated scans, ties: lower channel first; sorted lists, the chunk-index bytes, masks and fill as vrow_select.
            double s = 0.0;
#pragma unroll 1
            for (int j = 0; j < 16; j++) {
                float f[8];
                unpack8(trow[j], f);
#pragma unroll
                for (int e = 0; e < 8; e++) s += (double)f[e];
            }
            fill = hround((float)(s / (double)KD));
            const int kv = a.kv;
            uint64_t selE[2] = {0ull, 0ull}, selO[2] = {0ull, 0ull};      // [0] = large side, [1] = small side
#pragma unroll 1
            for (int side = 0; side < 2; side++) {
                const float sgn = side == 0 ? 1.0f : -1.0f;
#pragma unroll 1
                for (int rnd = 0; rnd < kv; rnd++) {
                    float best = 0.f;
                    int bi = KD;
#pragma unroll 1
                    for (int j = 0; j < 16; j++) {
                        float f[8];
                        unpack8(trow[j], f);
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const int d = 8 * j + e;
                            const bool taken = (((d & 1) ? selO[side] : selE[side]) >> (d >> 1)) & 1ull;
                            const float v = sgn * f[e];
                            const bool t = !taken && (bi == KD || v > best);      // strict: the first (lowest) channel wins a tie
                            best = t ? v : best;
                            bi = t ? d : bi;
                        }
                    }
                    if (bi & 1) selO[side] |= 1ull << (bi >> 1); else selE[side] |= 1ull << (bi >> 1);
                }
            }
            mE = selE[0] | selE[1];
            mO = selO[0] | selO[1];
            const int64_t lrow = (int64_t)nb * a.tcap + tok0 + lane;
            uint16_t* oi = a.voidx + lrow * (2 * kv);
            uint16_t* ov = a.voval + lrow * (2 * kv);
#pragma unroll 1
            for (int side = 0; side < 2; side++) {                        // list slot 0 = the small side, then the large side
                uint64_t e = selE[1 - side], o = selO[1 - side];
                int pos = side * kv;
                while (e | o) {
                    const int we = e ? __builtin_ctzll(e) : 64, wo = o ? __builtin_ctzll(o) : 64;
                    int d;
                    if (we <= wo) { d = 2 * we; e &= e - 1ull; } else { d = 2 * wo + 1; o &= o - 1ull; }
                    oi[pos] = (uint16_t)d;
                    ov[pos] = tile[lane * BT_PITCH + d];
                    pos++;
                }
            }
            if (a.vochunk) {                                              // head bounds of a one-head row: [0, kv] per side
                uint8_t* vc = a.vochunk + lrow * 4;
                vc[0] = 0; vc[1] = (uint8_t)kv; vc[2] = 0; vc[3] = (uint8_t)kv;
            }
        } else if (a.kv > 0) {
            gu32* fp = (gu32*)(a.flags + nb * 64 + lane);
            unsigned spins = 0;
            while (true) {
                const uint32_t f = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(f == a.epoch)) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 21)) {       // ~ a second: never reached unless the row duties cannot run
                    if (lane == 0) atomicOr(a.status, 1u);
                    break;
                }
            }
            gu64* mp = (gu64*)(a.vmask + (((int64_t)nb * 64 + lane) * a.H + hh) * 4);
            mE = __hip_atomic_load(mp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mO = __hip_atomic_load(mp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fill = hround(__builtin_bit_cast(float, __hip_atomic_load((gu32*)(a.vfill + nb * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        }
        const int64_t prow = bh * a.tcap + tok0 + lane;
        uint32_t* cp = a.vcode + prow * (KD / CPW);
        // the block's sparse tile: (token | channel << 6 | fp16(value - dequant) << 16), entries of a lane consecutive
        const bool tiles = a.kv > 0 && a.vtile && a.vcnt;
        int pos = 0;
        bool tile_ok = false;
        uint32_t* vt = nullptr;
        if (tiles) {
            const int blk = tok0 >> 6;
            const uint32_t cnt = (uint32_t)(__popcll(mE) + __popcll(mO));
            const uint32_t inc = wave_incl_scan_u32(cnt);
            const int total = __builtin_amdgcn_readlane((int)inc, 63);
            pos = (int)(inc - cnt);
            tile_ok = total <= a.vtile_cap;
            if (lane == 0) a.vcnt[bh * a.nblk + blk] = tile_ok ? total : -1;
            vt = a.vtile + (bh * a.nblk + blk) * (int64_t)a.vtile_cap;
        }
#pragma unroll 1
        for (int gi = 0; gi < NGV; gi++) {
            const uint32_t gE = (uint32_t)(mE >> (gi * (G / 2))) & (uint32_t)((1ull << (G / 2)) - 1ull);   // this group's words
            const uint32_t gO = (uint32_t)(mO >> (gi * (G / 2))) & (uint32_t)((1ull << (G / 2)) - 1ull);
            float lo = INFINITY, hi = -INFINITY;
#pragma unroll 2
            for (int jj = 0; jj < G / 8; jj++) {
                const uint4 v4 = trow[gi * (G / 8) + jj];
                float f[8];
                unpack8(v4, f);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const bool o = (((e & 1) ? gO : gE) >> (4 * jj + (e >> 1))) & 1u;
                    const float v = o ? fill : f[e];
                    lo = fminf(lo, v);
                    hi = fmaxf(hi, v);
                }
            }
            const QuantParams<0> qp = make_qparams<0>(lo, hi, LEVELS);
            a.vscale[prow * NGV + gi] = f2h_bits(qp.scale);
            a.vmn[prow * NGV + gi] = f2h_bits(qp.mn);
            if (tiles && tile_ok) {                  // (before pass 2 replaces x by the error)
                const float dqv = fmaf(qp.scale, (float)quant_one<0>(fill, qp), qp.mn);
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    uint32_t m = par ? gO : gE;
                    while (m) {
                        const int w = __builtin_ctz(m);
                        m &= m - 1u;
                        const int d = gi * G + 2 * w + par;
                        vt[pos++] = (uint32_t)lane | ((uint32_t)d << 6) |
                                    ((uint32_t)f2h_bits(h2f_bits(tile[lane * BT_PITCH + d]) - dqv) << 16);
                    }
                }
            }
            uint32_t cw = 0u;
#pragma unroll 2
            for (int jj = 0; jj < G / 8; jj++) {
                const uint4 v4 = trow[gi * (G / 8) + jj];
                float f[8], ef[8];
                unpack8(v4, f);
                uint32_t bits = 0u;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const bool o = (((e & 1) ? gO : gE) >> (4 * jj + (e >> 1))) & 1u;
                    const float v = o ? fill : f[e];
                    const int qv = quant_one<0>(v, qp);
                    bits |= (uint32_t)qv << (BITS * e);
                    ef[e] = o ? 0.0f : (v - dequant_one<0>(qv, qp.scale, qp.mn));
                }
                trow[gi * (G / 8) + jj] = pack8(ef);                     // the error, in place
                if (BITS == 4) cp[gi * (G / 8) + jj] = bits;
                else {
                    cw |= bits << (16 * (jj & 1));
                    if (jj & 1) { cp[(gi * (G / 8) + jj) >> 1] = cw; cw = 0u; }
                }
            }
        }
        r = a.rv;
        if (r > 0) {
            P0h = a.P0v + bh * KD * r;
            P_out = a.vP + (bh / a.p_inner) * a.vp_stride + (bh % a.p_inner) * (int64_t)(KD * r);
            Q_out = a.vQ + (bh * a.tcap + tok0) * (int64_t)r;
        }
    }
    if (r <= 0) return;
    __syncthreads();
    lowrank_tile<RP>(tile, lrsm, P0h, r, a.loop, P_out, Q_out, lane);
}

}  // namespace

static uint32_t g_block_epoch = 0;

extern "C" size_t gear_compress_block_workspace(int64_t NB, int H) {
    if (NB <= 0 || H <= 0) return 0;
    return 256 + (size_t)NB * 64 * (4 + 4 + (size_t)H * 16) + 256;
}

extern "C" int gear_compress_block(const gear_cache_view* c, int t_off, int o_off, int loop, const void* P0k, const void* P0v,
                                   void* kP_out, void* vP_out, int64_t p_inner, int64_t kp_outer_stride, int64_t vp_outer_stride,
                                   void* sync_ws, size_t sync_ws_bytes, void* stream) {
    GEAR_CHECK_ARG(c && c->kwin && c->vwin && c->kcode && c->kscale && c->kmn && c->vcode && c->vscale && c->vmn,
                   "gear_compress_block: null pointer in the cache view");
    GEAR_CHECK_ARG(c->D == KD, "gear_compress_block: head_dim must be 128");
    GEAR_CHECK_ARG(c->mode == GEAR_MODE_FP16_STEPWISE, "gear_compress_block: fp16-stepwise arithmetic only (mode 0)");
    GEAR_CHECK_ARG((c->bits == 2 || c->bits == 4) && (c->group == 32 || c->group == 64), "gear_compress_block: bits 2 / 4, group 32 / 64");
    GEAR_CHECK_ARG(c->B > 0 && c->Hkv > 0 && c->Hkv <= 64, "gear_compress_block: need 1 <= Hkv <= 64 (got %d)", c->Hkv);
    GEAR_CHECK_ARG(c->wcap >= 64, "gear_compress_block: the window must hold 64 tokens");
    GEAR_CHECK_ARG(t_off >= 0 && t_off % 64 == 0 && t_off + 64 <= c->tcap, "gear_compress_block: bad token offset %d (capacity %d)", t_off, c->tcap);
    GEAR_CHECK_ARG((int64_t)c->ldk * (32 / c->bits) >= t_off + 64 && (int64_t)c->lsk * c->group >= t_off + 64, "gear_compress_block: K row pitch too small");
    GEAR_CHECK_ARG((c->ldk * 4) % 16 == 0, "gear_compress_block: K code row pitch must be a multiple of 16 bytes");
    const int kk = (c->koidx && c->koval) ? c->kkb : 0, kv = (c->voidx && c->voval) ? c->kv : 0;
    GEAR_CHECK_ARG(kk >= 0 && kk <= 16, "gear_compress_block: at most 16 K outliers per side, channel and block (got %d)", kk);
    GEAR_CHECK_ARG(kk == 0 || (o_off >= 0 && o_off + kk <= c->kk_cap && t_off + 64 <= 65536), "gear_compress_block: bad K outlier geometry");
    GEAR_CHECK_ARG(kv >= 0 && 2 * kv <= c->Hkv * KD && kv <= 255 * 256, "gear_compress_block: bad V outlier count");
    GEAR_CHECK_ARG(!(kv > 0 && c->vochunk) || kv <= 255, "gear_compress_block: the V chunk index holds byte positions (kv <= 255)");
    GEAR_CHECK_ARG(c->rk >= 0 && c->rk <= 16 && c->rv >= 0 && c->rv <= 16, "gear_compress_block: rank must be in [0, 16]");
    if (c->rk > 0) GEAR_CHECK_ARG(P0k && kP_out && c->kQ && loop >= 1 && p_inner >= 1, "gear_compress_block: K low-rank needs P0, P_out, Q, loop >= 1");
    if (c->rv > 0) GEAR_CHECK_ARG(P0v && vP_out && c->vQ && loop >= 1 && p_inner >= 1, "gear_compress_block: V low-rank needs P0, P_out, Q, loop >= 1");
    const int64_t NB = c->B;
    const int H = c->Hkv;
    GEAR_CHECK_ARG(NB * H <= (int64_t)1 << 30, "gear_compress_block: too many tiles");
    if (kv > 0) GEAR_CHECK_ARG(sync_ws && sync_ws_bytes >= gear_compress_block_workspace(NB, H), "gear_compress_block: sync workspace too small");
    BlkArgs a;
    a.kwin = (const uint16_t*)c->kwin; a.vwin = (const uint16_t*)c->vwin; a.wcap = c->wcap;
    a.NB = (int)NB; a.H = H; a.t_off = t_off;
    a.kcode = (uint32_t*)c->kcode; a.kscale = (uint16_t*)c->kscale; a.kmn = (uint16_t*)c->kmn; a.ldk = c->ldk; a.lsk = c->lsk;
    a.koidx = (uint16_t*)c->koidx; a.koval = (uint16_t*)c->koval; a.kk = kk; a.kk_cap = c->kk_cap; a.o_off = o_off;
    a.ktile = (uint32_t*)c->ktile; a.kcnt = (int*)c->kcnt; a.ktile_cap = c->ktile_cap; a.nck = c->nck;
    a.vcode = (uint32_t*)c->vcode; a.vscale = (uint16_t*)c->vscale; a.vmn = (uint16_t*)c->vmn; a.tcap = c->tcap;
    a.voidx = (uint16_t*)c->voidx; a.voval = (uint16_t*)c->voval; a.kv = kv; a.vochunk = (uint8_t*)c->vochunk;
    a.vtile = (uint32_t*)c->vtile; a.vcnt = (int*)c->vcnt; a.vtile_cap = c->vtile_cap; a.nblk = c->nblk;
    a.rk = c->rk; a.rv = c->rv; a.loop = loop;
    a.P0k = (const float*)P0k; a.P0v = (const float*)P0v;
    a.kP = (uint16_t*)kP_out; a.vP = (uint16_t*)vP_out; a.p_inner = p_inner; a.kp_stride = kp_outer_stride; a.vp_stride = vp_outer_stride;
    a.kQ = (uint16_t*)c->kQ; a.vQ = (uint16_t*)c->vQ;
    if (a.ktile && a.kcnt) GEAR_CHECK_ARG((t_off >> 7) < c->nck && c->ktile_cap >= 0, "gear_compress_block: K tile chunk out of range");
    if (a.vtile && a.vcnt) GEAR_CHECK_ARG((t_off >> 6) < c->nblk, "gear_compress_block: V tile block out of range");
    char* base = (char*)(((uintptr_t)sync_ws + 255) & ~(uintptr_t)255);
    a.status = (uint32_t*)base;
    a.flags = (uint32_t*)(base + 64);
    a.vfill = (float*)(base + 64 + (size_t)NB * 64 * 4);
    a.vmask = (uint32_t*)(base + 64 + (size_t)NB * 64 * 8);
    if (!sync_ws) { a.status = nullptr; a.flags = nullptr; a.vfill = nullptr; a.vmask = nullptr; }
    // a fresh, non-zero tag per call: the flags of every earlier call (and the zeros the buffer starts with) never match it
    uint32_t ep = __atomic_add_fetch(&g_block_epoch, 1u, __ATOMIC_RELAXED);
    if (ep == 0u) ep = __atomic_add_fetch(&g_block_epoch, 1u, __ATOMIC_RELAXED);
    a.epoch = ep;
    a.rows_per_blk = (int)((NB * 64 + 2 * NB * H - 1) / (2 * NB * H));     // every workgroup (K and V tiles) takes its share
    if (H == 1) a.rows_per_blk = 0;                                        // (one head: a V tile selects for itself, no row duties)
    const int rmax = a.rk > a.rv ? a.rk : a.rv;
    const int RP = rmax <= 4 ? 4 : (rmax <= 8 ? 8 : 16);
    const size_t shmem = (size_t)64 * BT_PITCH * 2 + (rmax > 0 ? blk_lr_lds_bytes(RP) : 0);
    const dim3 grid((unsigned)(2 * NB * H));
    hipStream_t st = (hipStream_t)stream;
    // Forward progress of the flag hand-off (file header): a V tile's rows belong to workgroups dispatched before it or at most 63
    // after it, so the device must hold 64 of these workgroups AT ONCE (and dispatch them in order, as the hardware does).  Checked
    // at launch time, per kernel instantiation and LDS size, against the device's own occupancy figure -- a configuration that
    // cannot co-reside (a future larger tile, a partitioned GPU with a handful of CUs) is refused with an error instead of spinning
    // into the 1 s poll bound (round 4: only the bound existed).
    static int n_cu = 0;
    if (!n_cu) {
        int dev_ = 0;
        (void)hipGetDevice(&dev_);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess || n_cu <= 0) n_cu = 1;
    }
#define BLK_GO3(B, GG, RR)                                                                                         \
    do {                                                                                                           \
        auto kfn = block_compress_kernel<B, GG, RR>;                                                               \
        if (shmem > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        if (a.rows_per_blk > 0) {                                                                                  \
            static size_t occ_shmem = (size_t)-1;                                                                  \
            static int occ = 0;                                                                                    \
            if (occ_shmem != shmem) {                                                                              \
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kfn, 64, shmem) != hipSuccess) occ = 0; \
                occ_shmem = shmem;                                                                                 \
            }                                                                                                      \
            GEAR_CHECK_ARG((int64_t)occ * n_cu >= 64, "gear_compress_block: only %d workgroups of this kernel fit on the device " \
                           "at once (%d per CU x %d CUs); the row-duty hand-off needs 64 resident -- use the kernel chain", \
                           occ * n_cu, occ, n_cu);                                                                 \
        }                                                                                                          \
        hipLaunchKernelGGL(kfn, grid, dim3(64), shmem, st, a);                                                     \
    } while (0)
#define BLK_GO(B, GG)                                                                                              \
    do {                                                                                                           \
        if (RP == 4) BLK_GO3(B, GG, 4); else if (RP == 8) BLK_GO3(B, GG, 8); else BLK_GO3(B, GG, 16);              \
    } while (0)
    if (c->bits == 2) { if (c->group == 64) BLK_GO(2, 64); else BLK_GO(2, 32); }
    else { if (c->group == 64) BLK_GO(4, 64); else BLK_GO(4, 32); }
#undef BLK_GO3
#undef BLK_GO
    GEAR_CHECK_LAUNCH("gear_compress_block");
    return 0;
}

// status word of the sync workspace: non-zero when a V tile gave up waiting for its rows' selection (device memory, 4 bytes)
extern "C" const void* gear_compress_block_status_ptr(const void* sync_ws) {
    return (const void*)(((uintptr_t)sync_ws + 255) & ~(uintptr_t)255);
}
