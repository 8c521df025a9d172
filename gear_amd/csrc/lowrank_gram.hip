// lowrank_gram.hip -- the power iteration of lowrank.hip restructured around the Gram matrix, for head_dim 128.
//
//   for i < loop: [last: P = orth(P)]  Q = E P  [last: Q = orth(Q)]  P = E^T Q          (reference order)
// With G = E^T E (128 x 128, per head) every P-update is P <- G P, so
//   P_a = G^(loop-1) P0 ; P' = orth(P_a) ; Q = E P' ; Q^T Q = P'^T G P' = R^T R ; Q' = E (P' R^-1) ;  P_out = G (P' R^-1)
// i.e. the whole iteration needs ONE pass over E on the matrix cores (the Gram matrix, fp16 inputs are exact, fp32
// accumulate), a tiny per-head solve that never leaves LDS, and ONE more pass Q' = E W with W = P' R^-1.
// The reference makes 2*loop passes over E (new_pack.py:298-304); same mathematics, fp32-level differences.
//
// Kernel 1 (one workgroup per head): stream E through LDS (3 tiles of loads in flight), v_mfma_f32_32x32x16_f16 on the 10
//   upper-triangular 32x32 blocks of G (contraction over tokens), then the solve: strip matmuls on the symmetric G,
//   CholeskyQR2 in fp64 for orth(P) and one Cholesky for Q, both column-parallel on one wave.
// Kernel 2: Q' = E W -- token-major layout on the matrix cores (MFMA B operands straight from global memory, W split into
//   fp16 head + remainder); K^T layout: lanes own 8 tokens and stream the 128 channels with fp32 FMAs.
#include <stdlib.h>

#include "common.h"
#include "lowrank_solve.h"
#include "ktile.h"

int gear_ksolve_launch(const float* gpart, int nslab, int loop, const float* P0, int r, int64_t BH, float* Wout, void* P_out,
                       int out_f16, int64_t p_inner, int64_t p_outer_stride, hipStream_t st);

namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int GD = GS_GD;        // head_dim (the Gram trick is built for 128)

// four 16-byte staging registers with compile-time selection (an array passed by reference into the staging lambdas ended up in
// scratch memory: 144 bytes of private segment, every tile through scratch_store / scratch_load)
struct Stg4 {
    uint4 a, b, c, d;
    __device__ __forceinline__ void set(int p, const uint4& v) { if (p == 0) a = v; else if (p == 1) b = v; else if (p == 2) c = v; else d = v; }
    __device__ __forceinline__ uint4 get(int p) const { return p == 0 ? a : (p == 1 ? b : (p == 2 ? c : d)); }
};
constexpr int KT_PITCH = 72;     // halfs per LDS row of the K^T staging tile [128][64 (+8 pad)]
constexpr int TM_PITCH = 136;    // halfs per LDS row of the token-major staging tile [64][128 (+8 pad)]
constexpr int GP = GS_GP;        // float pitch of G in LDS: rows AND columns are bank-conflict-free

__host__ __device__ constexpr int blk_index(int I, int J) {  // upper-triangular block (I <= J) -> 0..9
    return I * 4 - (I * (I - 1)) / 2 + (J - I);
}

// LDS layout (bytes): [0, 65536) G fp32 (its head doubles as the staging tile while the Gram matrix is
// still in registers), then Pa, Pb fp32 [128][RP], then small fp64 scratch.
// SPLIT = false: one workgroup per head, Gram matrix + solve (rounds 1-3; option gram_fused).
// SPLIT = true (round 4): grid (nslab, heads) -- the workgroup streams ITS SLAB of the head's tokens, writes the complete (mirrored)
// partial Gram matrix to gpart [head][slab][128][128] and is done; the solve is kfused.hip's k_solve_kernel (G in registers, 10 KB of
// LDS: every head resident at once).  In the fused form the two workgroups of a CU start together and reach their solve phases
// together: HBM idles for ~60 us per round of 512 heads, twice per 32-layer call; and a head shard's 128 heads fill half the chip.
template <int RP, bool TOKEN_MAJOR, bool SPLIT>
__global__ __launch_bounds__(256, 2) void lr_gram_solve_kernel(const uint16_t* __restrict__ E, int S, int loop,
                                                               const float* __restrict__ P0, int r,
                                                               float* __restrict__ Wout, void* __restrict__ P_out,
                                                               int out_f16, int64_t p_inner, int64_t p_outer_stride,
                                                               float* __restrict__ gpart, int tok_per_slab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* G = (float*)smem;                                  // [128][GP]
    uint16_t* tile = (uint16_t*)smem;                         // staging (aliases G during phase 1)
    float* Pa = (float*)(smem + GD * GP * 4);                 // [128][RP]
    float* Pb = Pa + GD * RP;                                 // [128][RP]
    double* Md = (double*)(Pb + GD * RP);                     // [RP][RP]
    double* Rinv = Md + RP * RP;                              // [RP][RP]

    const int64_t bh = SPLIT ? blockIdx.y : blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 31, kg = lane >> 5;
    const uint16_t* Eb = E + bh * (int64_t)S * GD;
    // the slab's token range [s_lo, s_hi) (the whole head when fused)
    const int s_lo = SPLIT ? (int)blockIdx.x * tok_per_slab : 0;
    const int S_full = S;
    if (SPLIT) S = min(S_full, s_lo + tok_per_slab);

    float16_t acc[10];
#pragma unroll
    for (int b = 0; b < 10; b++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[b][q] = 0.0f;

    // ------------------------------------------------------------------ phase 1: G = E^T E on the matrix cores
    // staging: global -> registers (issued one tile ahead) -> LDS; the loads of tile t+1 fly during the MFMAs of tile t
    // Every load of the main loop is UNCONDITIONAL: a load under a per-lane condition makes the compiler lose count of what is
    // outstanding, and its s_waitcnt in front of the first LDS store of a trip became vmcnt(0) -- the pipeline of "three tiles in
    // flight" drained once per trip (measured in round 4 with the solve split off: 314 us for the 1 GiB tensor = 3.4 TB/s).
    // Token-major: thread (l16, rr) loads the four consecutive token rows 4 rr .. 4 rr + 3 of the tile from clamped (always
    // valid) 32-bit offsets; what lies beyond the slab / the sequence is zeroed in LDS after the store.
    const char* Ebc = (const char*)Eb;
    auto stage_load = [&](int t0, Stg4& stg) {
        if (TOKEN_MAJOR) {
            const int l16 = tid & 15, rr = tid >> 4;
            // (ONE code path: a fast / clamped pair of branches made the compiler split the 16-byte loads into dwords)
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t t = (uint32_t)min(t0 + 4 * rr + p, S_full - 1);
                stg.set(p, *(const uint4*)(Ebc + (size_t)((t * GD + (uint32_t)l16 * 8) * 2u)));
            }
        } else {             // K^T: E^T [128 channels][S tokens]; tile [128][64]: 8 lanes per channel row, 32 rows per pass
            const int l8 = tid & 7, rr = tid >> 3;
            const int t = min(t0 + l8 * 8, S_full - 8);                             // (S % 8 == 0: checked by the host)
#pragma unroll
            for (int p = 0; p < 4; p++) stg.set(p, *(const uint4*)(Eb + (int64_t)(rr + 32 * p) * S_full + t));
        }
    };
    auto stage_store = [&](int t0, const Stg4& stg) {
        // (no select on the loaded registers: `cond ? stg : 0` is scheduled right behind the load and waits for it there)
        if (TOKEN_MAJOR) {
            const int l16 = tid & 15, rr = tid >> 4;
#pragma unroll
            for (int p = 0; p < 4; p++) *(uint4*)(tile + (4 * rr + p) * TM_PITCH + l16 * 8) = stg.get(p);
        } else {
            const int l8 = tid & 7, rr = tid >> 3;
#pragma unroll
            for (int p = 0; p < 4; p++) *(uint4*)(tile + (rr + 32 * p) * KT_PITCH + l8 * 8) = stg.get(p);
        }
        if (t0 + 64 > S) {   // block-uniform, last tile of a ragged slab (or a tile beyond it) only: its tail is zeroed in LDS
            const uint4 zero = make_uint4(0, 0, 0, 0);
            if (TOKEN_MAJOR) {
                const int l16 = tid & 15, rr = tid >> 4;
                for (int p = 0; p < 4; p++)
                    if (t0 + 4 * rr + p >= S) *(uint4*)(tile + (4 * rr + p) * TM_PITCH + l16 * 8) = zero;
            } else {
                const int l8 = tid & 7, rr = tid >> 3;
                for (int p = 0; p < 4; p++)
                    if (t0 + l8 * 8 >= S) *(uint4*)(tile + (rr + 32 * p) * KT_PITCH + l8 * 8) = zero;
            }
        }
    };
    auto tile_mfma = [&]() {
        // wave w owns tokens [16w, 16w+16) of the tile; lane (x, kg) holds channel x of each 32-block, tokens 8kg..8kg+7
        half8_t f[4];
#pragma unroll
        for (int I = 0; I < 4; I++) {
            if (TOKEN_MAJOR) {
                union { half8_t h; uint16_t u[8]; } cv;
#pragma unroll
                for (int j = 0; j < 8; j++) cv.u[j] = tile[(16 * wave + 8 * kg + j) * TM_PITCH + 32 * I + x];
                f[I] = cv.h;
            } else {
                union { half8_t h; uint4 u; } cv;
                cv.u = *(const uint4*)(tile + (32 * I + x) * KT_PITCH + 16 * wave + 8 * kg);
                f[I] = cv.h;
            }
        }
#pragma unroll
        for (int I = 0; I < 4; I++)
#pragma unroll
            for (int J = I; J < 4; J++)
                acc[blk_index(I, J)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[I], f[J], acc[blk_index(I, J)], 0, 0, 0);
    };
    // two staging sets, two tiles (32 KB per workgroup, 64 KB per CU) of loads in flight, no branch around a load inside a trip (a
    // tile beyond the slab is loaded from a clamped, valid position and zeroed in LDS: at most one per slab), so the waits in
    // front of the LDS stores are exact counts.  (Three sets spilled: the accumulators alone are 160 of the 256 registers.)
    Stg4 stgA, stgB;
    const int t_last = max(s_lo, ((S_full >> 6) << 6) - 64);          // a tile start that is valid to load from
    stage_load(s_lo, stgA);
    stage_load(s_lo + 64 < S_full ? s_lo + 64 : t_last, stgB);
    for (int t0 = s_lo; t0 < S; t0 += 128) {
        __syncthreads();
        stage_store(t0, stgA);
        __syncthreads();
        stage_load(t0 + 128 < S_full ? t0 + 128 : t_last, stgA);
        tile_mfma();
        __syncthreads();
        stage_store(t0 + 64, stgB);
        __syncthreads();
        stage_load(t0 + 192 < S_full ? t0 + 192 : t_last, stgB);
        tile_mfma();
    }
    // the four waves hold partial Gram matrices over disjoint token subsets: add them into LDS one wave at a time
    // (deterministic, no fp32 LDS atomics); the mirrored lower blocks are written too -- with pitch 129 neither the
    // row-wise nor the column-wise accesses conflict.  C layout of the 32x32 MFMA: lane l, reg q ->
    // row (q&3) + 8*(q>>2) + 4*(l>>5), col l&31
    for (int turn = 0; turn < 4; turn++) {
        __syncthreads();
        if (wave == turn) {
#pragma unroll
            for (int I = 0; I < 4; I++)
#pragma unroll
                for (int J = I; J < 4; J++) {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int row = 32 * I + (q & 3) + 8 * (q >> 2) + 4 * kg, col = 32 * J + x;
                        const float v = acc[blk_index(I, J)][q];
                        if (turn == 0) G[row * GP + col] = v;   // blocks on or above the diagonal only: G is symmetric and
                        else G[row * GP + col] += v;            // matmulG reads G(d, e) below the diagonal as G(e, d)
                    }
                }
        }
    }
    __syncthreads();
    if (SPLIT) {
        // the slab's partial Gram matrix, complete (the solve reads whole rows): 32 lanes cover one 512-byte row
        float* gp = gpart + (bh * gridDim.x + blockIdx.x) * (int64_t)(GD * GD);
        for (int idx = tid; idx < GD * 32; idx += 256) {
            const int d = idx >> 5, e0 = (idx & 31) * 4, dlow = d & ~31;      // columns below the diagonal block: mirrored
            float4 v;
            if (e0 < dlow) v = make_float4(G[e0 * GP + d], G[(e0 + 1) * GP + d], G[(e0 + 2) * GP + d], G[(e0 + 3) * GP + d]);
            else v = make_float4(G[d * GP + e0], G[d * GP + e0 + 1], G[d * GP + e0 + 2], G[d * GP + e0 + 3]);
            *(float4*)&gp[d * GD + e0] = v;
        }
        return;
    }
    // ------------------------------------------------------------------ phase 2: the solve, entirely in LDS (lowrank_solve.h)
    // head bh of P_out lives at (bh / p_inner) * p_outer_stride + (bh % p_inner) * 128 * r (a per-segment factor tensor of a cache)
    const int64_t po = (bh / p_inner) * p_outer_stride + (bh % p_inner) * (int64_t)(GD * r);
    gram_solve_phase2<RP>(G, Pa, Pb, Md, Rinv, P0 + bh * GD * r, r, loop, Wout + bh * GD * RP,
                          out_f16 ? (void*)((uint16_t*)P_out + po) : (void*)((float*)P_out + po), out_f16);
}

// ---------------------------------------------------------------------------------------------- Gram matrix, wave-private streaming
// Token-major E, round 4.  grid (nslab, heads).  The 16-token steps of the slab are dealt out to the four waves round-robin and a
// wave does everything for ITS steps alone: four 16-byte loads per lane (16 token rows = 4 KB, contiguous in memory) -> its own
// 16-row LDS tile -> the transposing LDS read (ds_read_b64_tr_b16, ktile.h) -> the ten 32x32x16 matrix-core products.  No
// workgroup barrier inside the stream (the version above has two per 64-token tile, and every wave waits for the slowest load of
// the workgroup), NSTG steps of loads in flight per wave, every load unconditional from a clamped address.  The waves' partial
// sums meet in LDS at the end, the complete (mirrored) matrix goes to gpart [head][slab][128][128] for k_solve_kernel.
template <int NSTG, bool ROTATE, int VAR = 0>
__global__ __launch_bounds__(256, 2) void lr_gram_wave_kernel(const uint16_t* __restrict__ E, int S_full, float* __restrict__ gpart,
                                                              int tok_per_slab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t bh = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t* tile = (uint16_t*)smem + wave * 16 * ET_PITCH;  // this wave's [16 tokens][ET_PITCH]
    const int x = lane & 31, kg = lane >> 5;
    const char* Ebc = (const char*)(E + bh * (int64_t)S_full * GD);
    const int s_lo = (int)blockIdx.x * tok_per_slab, S = min(S_full, s_lo + tok_per_slab);
    const int nstep = (S - s_lo + 15) >> 4;                   // 16-token steps of the slab; wave w takes steps w, w + 4, ...
    const int r4 = lane >> 4, l16 = lane & 15;                // load p of a step: token row 4 p + r4, bytes 16 l16 .. + 15

    float16_t acc[10];
#pragma unroll
    for (int b = 0; b < 10; b++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[b][q] = 0.0f;

    // ROTATE (measurement builds only): workgroups of different heads start at different places of their slab, in case lockstep
    // marching through the same offsets of regions 2^k bytes apart camped on a few HBM channels (tools/ubench/store_pattern.hip saw
    // that for stores).  Measured here: 2 % -- and the summation order then depends on the head's index in the call, which a head
    // shard's bit-for-bit comparison with the unsharded run cannot have.  Off in the product.  Logical step j -> physical step:
    const int ngrp = (nstep + 3) >> 2;
    const int rotg = ROTATE ? (int)((((uint32_t)bh * 2654435761u) >> 12) % (uint32_t)ngrp) : 0;
    auto phys = [&](int j) {
        int g = (j >> 2) + rotg;
        if (g >= ngrp) g -= ngrp;
        return 4 * g + (j & 3);
    };
    Stg4 stg[NSTG];
    auto issue = [&](int j, Stg4& st) {                       // (a step beyond the slab re-reads a valid row: never consumed)
        const int step = j < 4 * ngrp ? phys(j) : 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t t = (uint32_t)min(s_lo + 16 * step + 4 * p + r4, S_full - 1);
            st.set(p, *(const uint4*)(Ebc + (size_t)((t * GD + (uint32_t)l16 * 8) * 2u)));
        }
    };
    auto consume = [&](int j, const Stg4& st) {
        const int t0 = s_lo + 16 * phys(j);
        if (VAR == 2) {                                       // (elimination build: loads only)
#pragma unroll
            for (int p = 0; p < 4; p++) { const uint4 v = st.get(p); acc[0][p] += __builtin_bit_cast(float, v.x ^ v.y ^ v.z ^ v.w); }
            return;
        }
#pragma unroll
        for (int p = 0; p < 4; p++) *(uint4*)(tile + (4 * p + r4) * ET_PITCH + l16 * 8) = st.get(p);
        if (t0 + 16 > S) {                                    // wave-uniform: the ragged last step only
            for (int p = 0; p < 4; p++)
                if (t0 + 4 * p + r4 >= S) *(uint4*)(tile + (4 * p + r4) * ET_PITCH + l16 * 8) = make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        half8_t f[4];
#pragma unroll
        for (int I = 0; I < 4; I++) f[I] = load_operand<true>(tile, 0, I, lane);
        if (VAR == 1) {                                       // (elimination build: no matrix-core work)
#pragma unroll
            for (int I = 0; I < 4; I++) acc[I][0] += (float)f[I][0];
            return;
        }
#pragma unroll
        for (int I = 0; I < 4; I++)
#pragma unroll
            for (int J = I; J < 4; J++)
                acc[blk_index(I, J)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[I], f[J], acc[blk_index(I, J)], 0, 0, 0);
        __builtin_amdgcn_wave_barrier();                      // (the tile is read before the next step overwrites it)
    };
#pragma unroll
    for (int i = 0; i < NSTG; i++) issue(wave + 4 * i, stg[i]);
    // the ring is walked with compile-time slots: NSTG steps per trip
    for (int st0 = wave; st0 < 4 * ngrp; st0 += 4 * NSTG) {
#pragma unroll
        for (int i = 0; i < NSTG; i++) {
            const int j = st0 + 4 * i;
            if (j < 4 * ngrp && phys(j) < nstep) consume(j, stg[i]);     // wave-uniform
            issue(j + 4 * NSTG, stg[i]);
        }
    }
    if (VAR == 3) return;                                     // (elimination build: no cross-wave sum, no output)
    // ---- the four waves' partial Gram matrices meet in LDS BLOCK BY BLOCK, all waves working at once (the first version added
    // whole matrices one wave at a time: four serialized turns of 160 LDS read-modify-writes, ~35 us per workgroup during which its
    // share of HBM idled -- 75 us of a 295 us kernel): every wave drops its 32 x 32 partial of block (I, J) into its own
    // [32][33] buffer, after a barrier thread (r, c) adds the four and writes element (r, c) of the block and, for an
    // off-diagonal block, element (r, c) of the mirrored block (read transposed from the same buffers) -- both stores are rows.
    __syncthreads();                                          // every wave is done with its tile: the buffers alias them
    float* buf = (float*)smem;                                // [4 waves][32][33]
    float* gp = gpart + (bh * gridDim.x + blockIdx.x) * (int64_t)(GD * GD);
#pragma unroll
    for (int I = 0; I < 4; I++)
#pragma unroll
        for (int J = I; J < 4; J++) {
            const int b = blk_index(I, J);
#pragma unroll
            for (int q = 0; q < 16; q++) buf[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg) * 33 + x] = acc[b][q];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = tid + 256 * i, r = e >> 5, c = e & 31;
                const float v = (buf[r * 33 + c] + buf[(32 + r) * 33 + c]) + (buf[(64 + r) * 33 + c] + buf[(96 + r) * 33 + c]);
                gp[(32 * I + r) * GD + 32 * J + c] = v;
                if (I != J) {
                    const float w = (buf[c * 33 + r] + buf[(32 + c) * 33 + r]) + (buf[(64 + c) * 33 + r] + buf[(96 + c) * 33 + r]);
                    gp[(32 * J + r) * GD + 32 * I + c] = w;
                }
            }
            __syncthreads();
        }
}


template <int N>
__device__ __forceinline__ void store_halfs(uint16_t* p, const float* f) {  // N in {4, 8, 16}, p aligned to 2N bytes
    if (N == 4) {
        uint2 v;
        v.x = (uint32_t)f2h_bits(f[0]) | ((uint32_t)f2h_bits(f[1]) << 16);
        v.y = (uint32_t)f2h_bits(f[2]) | ((uint32_t)f2h_bits(f[3]) << 16);
        *(uint2*)p = v;
    } else {
#pragma unroll
        for (int i = 0; i < N / 8; i++) ((uint4*)p)[i] = pack8(f + 8 * i);
    }
}

// ---------------------------------------------------------------------------------------------- Q' = E W
// K^T layout: E^T [bh][128][S].  Lane owns 8 consecutive tokens and walks the 128 channel rows.
template <int RP>
__global__ __launch_bounds__(256) void lr_qpass_kt_kernel(const uint16_t* __restrict__ Et, const float* __restrict__ W,
                                                          int S, int r, void* __restrict__ Q_out, int out_f16) {
    __shared__ float Ws[GD * RP];
    const int64_t bh = blockIdx.y;
    for (int i = threadIdx.x; i < GD * RP; i += 256) Ws[i] = W[bh * GD * RP + i];
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (t0 >= S) return;
    const uint16_t* p = Et + bh * (int64_t)GD * S + t0;
    float acc[8][RP];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int c = 0; c < RP; c++) acc[j][c] = 0.0f;
    // two register sets of 4 channel rows: the loads of the next 4 rows are in flight while the current 4 are consumed (a
    // plain unrolled loop waited for all of its loads before the first FMA)
    uint4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) ra[i] = *(const uint4*)(p + (int64_t)i * S);
    auto consume = [&](const uint4 (&rr)[4], int d0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float m[8];
            unpack8(rr[i], m);
#pragma unroll
            for (int c = 0; c < RP; c++) {
                const float w = Ws[(d0 + i) * RP + c];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j][c] = fmaf(m[j], w, acc[j][c]);
            }
        }
    };
#pragma unroll 1
    for (int d = 0; d < GD; d += 8) {
#pragma unroll
        for (int i = 0; i < 4; i++) rb[i] = *(const uint4*)(p + (int64_t)(d + 4 + i) * S);
        consume(ra, d);
        if (d + 8 < GD) {
#pragma unroll
            for (int i = 0; i < 4; i++) ra[i] = *(const uint4*)(p + (int64_t)(d + 8 + i) * S);
        }
        consume(rb, d + 4);
    }
    if (out_f16 && r == RP && t0 + 8 <= S) {   // 8 tokens x RP halfs = one contiguous run per lane
        uint16_t* qo = (uint16_t*)Q_out + (bh * S + t0) * (int64_t)RP;
#pragma unroll
        for (int j = 0; j < 8; j++) store_halfs<RP>(qo + j * RP, acc[j]);
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int c = 0; c < RP; c++) {   // static indices only: a run-time index would push acc[][] into scratch memory
            if (t0 + j < S && c < r) {
                int64_t o = (bh * S + t0 + j) * r + c;
                if (out_f16) ((uint16_t*)Q_out)[o] = f2h_bits(acc[j][c]);
                else ((float*)Q_out)[o] = acc[j][c];
            }
        }
    }
}

// token-major layout on the matrix cores: Q'^T [c][token] = W^T [c][channel] . E^T [channel][token], one
// v_mfma_f32_32x32x16_f16 tile = 32 tokens x (RP of 32 rows used).  The B operand of lane (token n, k-half) is 8
// consecutive channels of ONE token row = one 16-byte global load (no LDS staging, no transpose); no 16-lane DPP
// reductions, no fp32 FMAs on the vector ALU (a VALU version of this pass cost ~110 VALU per 16 bytes).  W (fp32) is split
// into an fp16 head and an fp16 remainder (two MFMAs per k-step), which keeps ~22 bits of it.
// C layout (lane l, register q): row (q & 3) + 8 (q >> 2) + 4 (l >> 5), column l & 31: a lane ends up with 4 consecutive
// rank columns of its token = one 8-byte store.
template <int RP>
__global__ __launch_bounds__(256) void lr_qpass_tm_mfma_kernel(const uint16_t* __restrict__ E, const float* __restrict__ W,
                                                               int S, int r, void* __restrict__ Q_out, int out_f16,
                                                               int q_tcap, int q_toff) {
    constexpr int NT = 4;   // 32-token tiles per wave
    __shared__ __attribute__((aligned(16))) uint16_t Ah[16 * RP * 8], Al[16 * RP * 8];   // [k / 8][m][k % 8]
    const int64_t bh = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < GD * RP; idx += 256) {
        const int k = idx / RP, m = idx % RP;
        const float w = W[(bh * GD + k) * RP + m];
        const uint16_t hi = f2h_bits(w);
        const uint16_t lo = f2h_bits(w - h2f_bits(hi));
        const int pos = ((k >> 3) * RP + m) * 8 + (k & 7);
        Ah[pos] = hi;
        Al[pos] = lo;
    }
    __syncthreads();
    const int n = lane & 31, kg = lane >> 5;
    const int tbase = (blockIdx.x * 4 + wave) * NT * 32;
    union U { uint4 u; half8_t h; };
    uint4 raw[2][8];
    auto load_tile = [&](int it, uint4 (&dst)[8]) {
        const int token = tbase + it * 32 + n;
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            dst[ks] = make_uint4(0, 0, 0, 0);
            if (token < S) dst[ks] = *(const uint4*)(E + (bh * S + token) * (int64_t)GD + 16 * ks + 8 * kg);
        }
    };
    load_tile(0, raw[0]);
#pragma unroll
    for (int it = 0; it < NT; it++) {
        if (tbase + it * 32 >= S) break;
        if (it + 1 < NT) load_tile(it + 1, raw[(it + 1) & 1]);
        float16_t acc;
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            U ah, al, b;
            ah.u = al.u = make_uint4(0, 0, 0, 0);
            if (n < RP) {
                ah.u = *(const uint4*)&Ah[((2 * ks + kg) * RP + n) * 8];
                al.u = *(const uint4*)&Al[((2 * ks + kg) * RP + n) * 8];
            }
            b.u = raw[it & 1][ks];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, b.h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, b.h, acc, 0, 0, 0);
        }
        const int token = tbase + it * 32 + n;
        if (token >= S) continue;
#pragma unroll
        for (int qb = 0; qb < (RP + 7) / 8; qb++) {   // register block qb holds rank columns 8 qb + 4 kg + (0..3)
            const int c0 = 8 * qb + 4 * kg;
            if (c0 >= RP) continue;
            if (out_f16 && r == RP) {
                uint2 v;
                v.x = (uint32_t)f2h_bits(acc[4 * qb]) | ((uint32_t)f2h_bits(acc[4 * qb + 1]) << 16);
                v.y = (uint32_t)f2h_bits(acc[4 * qb + 2]) | ((uint32_t)f2h_bits(acc[4 * qb + 3]) << 16);
                *(uint2*)((uint16_t*)Q_out + (bh * q_tcap + q_toff + token) * (int64_t)RP + c0) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (c0 + i < r) {
                        const int64_t o = (bh * (int64_t)q_tcap + q_toff + token) * r + c0 + i;
                        if (out_f16) ((uint16_t*)Q_out)[o] = f2h_bits(acc[4 * qb + i]);
                        else ((float*)Q_out)[o] = acc[4 * qb + i];
                    }
                }
            }
        }
    }
}

// slabs per head of the split Gram kernel: enough workgroups for the chip (two per CU) when a call has few heads (head shards),
// at least four 64-token tiles per slab; one slab when the heads alone fill it
inline int gram_nslab(int64_t bh, int S) {
    int n = 1;
    while (n < 8 && bh * n < 512 && S / (2 * n) >= 256) n *= 2;
    return n;
}

template <int RP>
int run_gram(const uint16_t* E, int transposed, int64_t bh, int S, int r, int loop, const float* P0, void* P_out,
             void* Q_out, int out_dtype, float* Wws, hipStream_t st, int64_t p_inner, int64_t p_outer_stride, int q_tcap,
             int q_toff) {
    const int of16 = out_dtype == GEAR_DTYPE_F16;
    size_t shmem = (size_t)GD * GP * 4 + 2 * (size_t)GD * RP * 4 + 3 * (size_t)RP * RP * 8 + 16;
    const bool split = gear_options().gram_fused != 1;   // 1: fused kernel (rounds 1-3); 2: split, barrier kernel
    const int nslab = gram_nslab(bh, S), tps = ((S + 63) / 64 + nslab - 1) / nslab * 64;
    float* gpart = (float*)((char*)Wws + (((size_t)bh * GD * RP * 4 + 255) & ~(size_t)255));
#define GRAM_GO(TM)                                                                                                      \
    do {                                                                                                                 \
        if (split && TM && gear_options().gram_fused != 2) {                                                             \
            const int nstg = gear_options().gram_nstg;                                                                   \
            /* default: 3 steps in flight, NO rotation of the start position (a head's summation order must not depend on where the  \
               head sits in the call: a head shard's factors are compared bit for bit with the unsharded run's); the other          \
               instantiations are the measurement builds of tools/exp_gram_elim.py */                                              \
            auto kfn = nstg == 2 ? lr_gram_wave_kernel<2, false> : (nstg == 4 ? lr_gram_wave_kernel<4, false>              \
                       : (nstg == 5 ? lr_gram_wave_kernel<3, true> : (nstg == 6 ? lr_gram_wave_kernel<3, false, 1>          \
                       : (nstg == 7 ? lr_gram_wave_kernel<3, false, 2> : (nstg == 8 ? lr_gram_wave_kernel<3, false, 3>    \
                       : lr_gram_wave_kernel<3, false>)))));                                                              \
            const size_t wsh = (size_t)4 * 16 * ET_PITCH * 2 > (size_t)4 * 32 * 33 * 4 ? (size_t)4 * 16 * ET_PITCH * 2 : (size_t)4 * 32 * 33 * 4; \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsh);           \
            hipLaunchKernelGGL(kfn, dim3((unsigned)nslab, (unsigned)bh), dim3(256), wsh, st, E, S, gpart, tps);          \
            gear_ksolve_launch(gpart, nslab, loop, P0, r, bh, Wws, P_out, of16, p_inner, p_outer_stride, st);            \
        } else if (split) {                                                                                              \
            auto kfn = lr_gram_solve_kernel<RP, TM, true>;                                                               \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);         \
            hipLaunchKernelGGL(kfn, dim3((unsigned)nslab, (unsigned)bh), dim3(256), shmem, st, E, S, loop, P0, r, Wws,   \
                               P_out, of16, p_inner, p_outer_stride, gpart, tps);                                        \
            gear_ksolve_launch(gpart, nslab, loop, P0, r, bh, Wws, P_out, of16, p_inner, p_outer_stride, st);            \
        } else {                                                                                                         \
            auto kfn = lr_gram_solve_kernel<RP, TM, false>;                                                              \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);         \
            hipLaunchKernelGGL(kfn, dim3((unsigned)bh), dim3(256), shmem, st, E, S, loop, P0, r, Wws, P_out, of16,       \
                               p_inner, p_outer_stride, (float*)nullptr, 0);                                             \
        }                                                                                                                \
    } while (0)
    if (transposed) {
        GRAM_GO(false);
        hipLaunchKernelGGL((lr_qpass_kt_kernel<RP>), dim3((S + 2047) / 2048, (unsigned)bh), dim3(256), 0, st, E, Wws, S, r,
                           Q_out, of16);
    } else {
        GRAM_GO(true);
        hipLaunchKernelGGL((lr_qpass_tm_mfma_kernel<RP>), dim3((S + 511) / 512, (unsigned)bh), dim3(256), 0, st, E, Wws, S,
                               r, Q_out, of16, q_tcap, q_toff);
    }
#undef GRAM_GO
    GEAR_CHECK_LAUNCH("gear_lowrank(gram)");
    return 0;
}

}  // namespace

// Q' = E W alone (token-major fp16 E [bh][S][128], W [bh][128][RP] from the solve): the fused K chain's Q pass when k_dense_kernel
// has written the error matrix out (kfused.hip)
int gear_lr_qpass_tm_launch(const void* E, const float* W, int64_t bh, int S, int r, void* Q_out, int out_f16, int q_tcap, int q_toff,
                            hipStream_t st) {
    const int RP = r <= 4 ? 4 : (r <= 8 ? 8 : 16);
    const dim3 grid((unsigned)((S + 511) / 512), (unsigned)bh);
    if (RP == 4) hipLaunchKernelGGL((lr_qpass_tm_mfma_kernel<4>), grid, dim3(256), 0, st, (const uint16_t*)E, W, S, r, Q_out, out_f16, q_tcap, q_toff);
    else if (RP == 8) hipLaunchKernelGGL((lr_qpass_tm_mfma_kernel<8>), grid, dim3(256), 0, st, (const uint16_t*)E, W, S, r, Q_out, out_f16, q_tcap, q_toff);
    else hipLaunchKernelGGL((lr_qpass_tm_mfma_kernel<16>), grid, dim3(256), 0, st, (const uint16_t*)E, W, S, r, Q_out, out_f16, q_tcap, q_toff);
    GEAR_CHECK_LAUNCH("gear_lr_qpass_tm_launch");
    return 0;
}

// workspace of the Gram path: W [bh][128][RP] floats, then the slabs' partial Gram matrices [bh][nslab][128][128]
size_t gear_lowrank_gram_workspace(int64_t bh, int S, int RP) {
    return (((size_t)bh * GD * RP * 4 + 255) & ~(size_t)255) + (size_t)bh * gram_nslab(bh, S) * GD * GD * 4 + 512;
}

// Called by gear_lowrank() when the fast path applies: fp16 error, Dm == 128, S % 8 == 0 (K^T layout).  The _ex form writes
// P at a per-head offset and Q at a row offset of a larger tensor (token-major E only): the streaming cache's layouts.
int gear_lowrank_gram_ex(const void* E, int transposed, int64_t bh, int S, int r, int loop, const void* P0, void* P_out,
                         int64_t p_inner, int64_t p_outer_stride, void* Q_out, int q_tcap, int q_toff, int out_dtype,
                         void* workspace, hipStream_t st) {
    const int RP = r <= 4 ? 4 : (r <= 8 ? 8 : 16);
    float* Wws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (RP == 4) return run_gram<4>((const uint16_t*)E, transposed, bh, S, r, loop, (const float*)P0, P_out, Q_out, out_dtype, Wws, st, p_inner, p_outer_stride, q_tcap, q_toff);
    if (RP == 8) return run_gram<8>((const uint16_t*)E, transposed, bh, S, r, loop, (const float*)P0, P_out, Q_out, out_dtype, Wws, st, p_inner, p_outer_stride, q_tcap, q_toff);
    return run_gram<16>((const uint16_t*)E, transposed, bh, S, r, loop, (const float*)P0, P_out, Q_out, out_dtype, Wws, st, p_inner, p_outer_stride, q_tcap, q_toff);
}

int gear_lowrank_gram(const void* E, int transposed, int64_t bh, int S, int r, int loop, const void* P0, void* P_out,
                      void* Q_out, int out_dtype, void* workspace, hipStream_t st) {
    return gear_lowrank_gram_ex(E, transposed, bh, S, r, loop, P0, P_out, bh, 0, Q_out, S, 0, out_dtype, workspace, st);
}
