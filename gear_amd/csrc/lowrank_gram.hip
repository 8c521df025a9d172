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

namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int GD = GS_GD;        // head_dim (the Gram trick is built for 128)
constexpr int KT_PITCH = 72;     // halfs per LDS row of the K^T staging tile [128][64 (+8 pad)]
constexpr int TM_PITCH = 136;    // halfs per LDS row of the token-major staging tile [64][128 (+8 pad)]
constexpr int GP = GS_GP;        // float pitch of G in LDS: rows AND columns are bank-conflict-free

__host__ __device__ constexpr int blk_index(int I, int J) {  // upper-triangular block (I <= J) -> 0..9
    return I * 4 - (I * (I - 1)) / 2 + (J - I);
}

// LDS layout (bytes): [0, 65536) G fp32 (its head doubles as the staging tile while the Gram matrix is
// still in registers), then Pa, Pb fp32 [128][RP], then small fp64 scratch.
template <int RP, bool TOKEN_MAJOR>
__global__ __launch_bounds__(256, 2) void lr_gram_solve_kernel(const uint16_t* __restrict__ E, int S, int loop,
                                                               const float* __restrict__ P0, int r,
                                                               float* __restrict__ Wout, void* __restrict__ P_out,
                                                               int out_f16, int64_t p_inner, int64_t p_outer_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* G = (float*)smem;                                  // [128][GP]
    uint16_t* tile = (uint16_t*)smem;                         // staging (aliases G during phase 1)
    float* Pa = (float*)(smem + GD * GP * 4);                 // [128][RP]
    float* Pb = Pa + GD * RP;                                 // [128][RP]
    double* Md = (double*)(Pb + GD * RP);                     // [RP][RP]
    double* Rinv = Md + RP * RP;                              // [RP][RP]

    const int64_t bh = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 31, kg = lane >> 5;
    const uint16_t* Eb = E + bh * (int64_t)S * GD;

    float16_t acc[10];
#pragma unroll
    for (int b = 0; b < 10; b++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[b][q] = 0.0f;

    // ------------------------------------------------------------------ phase 1: G = E^T E on the matrix cores
    // staging: global -> registers (issued one tile ahead) -> LDS; the loads of tile t+1 fly during the MFMAs of tile t
    uint4 stgA[4], stgB[4];
    auto stage_load = [&](int t0, uint4 (&stg)[4]) {
        if (TOKEN_MAJOR) {   // tile [64 tokens][128 channels]: 16 lanes per token row, 16 rows per pass
            const int l16 = tid & 15, rr = tid >> 4;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                int t = t0 + rr + 16 * p;
                stg[p] = make_uint4(0, 0, 0, 0);
                if (t < S) stg[p] = *(const uint4*)(Eb + (int64_t)t * GD + l16 * 8);
            }
        } else {             // K^T: E^T [128 channels][S tokens]; tile [128][64]: 8 lanes per channel row, 32 rows per pass
            const int l8 = tid & 7, rr = tid >> 3;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                int d = rr + 32 * p, t = t0 + l8 * 8;
                stg[p] = make_uint4(0, 0, 0, 0);
                if (t < S) stg[p] = *(const uint4*)(Eb + (int64_t)d * S + t);
            }
        }
    };
    auto stage_store = [&](const uint4 (&stg)[4]) {
        if (TOKEN_MAJOR) {
            const int l16 = tid & 15, rr = tid >> 4;
#pragma unroll
            for (int p = 0; p < 4; p++) *(uint4*)(tile + (rr + 16 * p) * TM_PITCH + l16 * 8) = stg[p];
        } else {
            const int l8 = tid & 7, rr = tid >> 3;
#pragma unroll
            for (int p = 0; p < 4; p++) *(uint4*)(tile + (rr + 32 * p) * KT_PITCH + l8 * 8) = stg[p];
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
    uint4 stgC[4];
    stage_load(0, stgA);
    if (64 < S) stage_load(64, stgB);
    if (128 < S) stage_load(128, stgC);
    for (int t0 = 0; t0 < S; t0 += 192) {       // three tiles per trip, three tiles (48 KB per workgroup) of loads in flight
        __syncthreads();
        stage_store(stgA);
        __syncthreads();
        if (t0 + 192 < S) stage_load(t0 + 192, stgA);
        tile_mfma();
        if (t0 + 64 < S) {
            __syncthreads();
            stage_store(stgB);
            __syncthreads();
            if (t0 + 256 < S) stage_load(t0 + 256, stgB);
            tile_mfma();
        }
        if (t0 + 128 < S) {
            __syncthreads();
            stage_store(stgC);
            __syncthreads();
            if (t0 + 320 < S) stage_load(t0 + 320, stgC);
            tile_mfma();
        }
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
    // ------------------------------------------------------------------ phase 2: the solve, entirely in LDS (lowrank_solve.h)
    __syncthreads();
    // head bh of P_out lives at (bh / p_inner) * p_outer_stride + (bh % p_inner) * 128 * r (a per-segment factor tensor of a cache)
    const int64_t po = (bh / p_inner) * p_outer_stride + (bh % p_inner) * (int64_t)(GD * r);
    gram_solve_phase2<RP>(G, Pa, Pb, Md, Rinv, P0 + bh * GD * r, r, loop, Wout + bh * GD * RP,
                          out_f16 ? (void*)((uint16_t*)P_out + po) : (void*)((float*)P_out + po), out_f16);
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

template <int RP>
int run_gram(const uint16_t* E, int transposed, int64_t bh, int S, int r, int loop, const float* P0, void* P_out,
             void* Q_out, int out_dtype, float* Wws, hipStream_t st, int64_t p_inner, int64_t p_outer_stride, int q_tcap,
             int q_toff) {
    const int of16 = out_dtype == GEAR_DTYPE_F16;
    size_t shmem = (size_t)GD * GP * 4 + 2 * (size_t)GD * RP * 4 + 3 * (size_t)RP * RP * 8 + 16;
    if (transposed) {
        auto kfn = lr_gram_solve_kernel<RP, false>;
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(kfn, dim3((unsigned)bh), dim3(256), shmem, st, E, S, loop, P0, r, Wws, P_out, of16, p_inner, p_outer_stride);
        hipLaunchKernelGGL((lr_qpass_kt_kernel<RP>), dim3((S + 2047) / 2048, (unsigned)bh), dim3(256), 0, st, E, Wws, S, r,
                           Q_out, of16);
    } else {
        auto kfn = lr_gram_solve_kernel<RP, true>;
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(kfn, dim3((unsigned)bh), dim3(256), shmem, st, E, S, loop, P0, r, Wws, P_out, of16, p_inner, p_outer_stride);
        hipLaunchKernelGGL((lr_qpass_tm_mfma_kernel<RP>), dim3((S + 511) / 512, (unsigned)bh), dim3(256), 0, st, E, Wws, S,
                               r, Q_out, of16, q_tcap, q_toff);
    }
    GEAR_CHECK_LAUNCH("gear_lowrank(gram)");
    return 0;
}

}  // namespace

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
