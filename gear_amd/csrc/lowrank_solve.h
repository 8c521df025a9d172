// lowrank_solve.h -- the per-head solve of the Gram-matrix power iteration, entirely in LDS (shared by lowrank_gram.hip,
// which accumulates G = E^T E itself, and kfused.hip, whose fused quantize + Gram kernel hands over partial Gram matrices).
//
//   P_a = G^(loop-1) P0 ; P' = orth(P_a) (CholeskyQR2, fp64 Gram) ; T1 = G P' ; Q^T Q = P'^T T1 = R^T R ;
//   W = P' R^-1 (the matrix of the Q pass, Q' = E W) ; P_out = T1 R^-1
// -- the reference's order of operations (cuda_supported_gear/quant/new_pack.py:298-304,
// GenerationBench/.../Simulated/compress_function.py:85-94) written on G; see lowrank_gram.hip for the derivation.
//
// Called by all 256 threads of a workgroup.  LDS: G fp32 [128][GS_GP] with the blocks on or above the block diagonal
// valid (matmulG reads G(d, e) below it as G(e, d)); Pa, Pb fp32 [128][RP]; Md fp64 [RP][RP]; Rinv fp64 [2][RP][RP].
#pragma once
#include "common.h"

constexpr int GS_GD = 128;   // head_dim
constexpr int GS_GP = 129;   // float pitch of G in LDS: rows AND columns are bank-conflict-free

__host__ __device__ constexpr size_t gram_solve_lds_bytes(int RP) {
    return (size_t)GS_GD * GS_GP * 4 + 2 * (size_t)GS_GD * RP * 4 + 3 * (size_t)RP * RP * 8 + 16;
}

// Md = R^T R  ->  Rinv = R^-1 (upper); dependent / zero columns -> 0.  One wave: lane m owns column m of R and of R^-1; the
// pivots' reciprocal square roots come from v_rsq_f64 + Newton.  Md fp64 [RP][RP], Rinv fp64 [2][RP][RP] (the second half stages
// R for the back substitution), both in LDS.  The caller orders the LDS accesses around the call (barrier or wave fence).
// MDS: stride (in doubles) between consecutive entries of Md (1 = dense).
template <int RP, int MDS = 1>
__device__ __forceinline__ void chol_inverse_wave(const double* __restrict__ Md, double* __restrict__ Rinv, int lane) {
    double* Rl = Rinv + RP * RP;   // R staged for the back substitution
    const int m = lane;
    double col[RP], rin[RP], rinvd[RP];
    bool dead[RP];
#pragma unroll
    for (int j = 0; j < RP; j++) {
        double sacc = (m < RP) ? Md[(j * RP + m) * MDS] : 0.0;
#pragma unroll
        for (int kk = 0; kk < RP; kk++) {
            if (kk < j) {
                const double rkj = __shfl(col[kk], j, 64);    // R[kk][j]
                sacc -= rkj * col[kk];
            }
        }
        const double dj = __shfl(sacc, j, 64), dg = Md[(j * RP + j) * MDS];
        dead[j] = !(dj > 1e-12 * dg) || !(dg > 0.0);
        double rs = 1.0;
        if (!dead[j]) {
            rs = __builtin_amdgcn_rsq(dj);
            rs = rs * (1.5 - 0.5 * dj * rs * rs);
            rs = rs * (1.5 - 0.5 * dj * rs * rs);
        }
        rinvd[j] = dead[j] ? 0.0 : rs;
        col[j] = (m == j) ? (dead[j] ? 1.0 : dj * rs) : ((m > j && !dead[j]) ? sacc * rs : 0.0);
    }
    if (m < RP) {
#pragma unroll
        for (int kk = 0; kk < RP; kk++) Rl[kk * RP + m] = col[kk];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    bool deadm = false;
#pragma unroll
    for (int j = 0; j < RP; j++) deadm = (m == j) ? dead[j] : deadm;
    // column m of R^-1 by back substitution
#pragma unroll
    for (int i = RP - 1; i >= 0; i--) {
        double sacc = 0.0;
#pragma unroll
        for (int kk = 0; kk < RP; kk++)
            if (kk > i) sacc += Rl[i * RP + kk] * ((kk <= m) ? rin[kk] : 0.0);
        rin[i] = (i == m) ? rinvd[i] : ((i < m && !deadm) ? -sacc * rinvd[i] : 0.0);
    }
    if (m < RP) {
#pragma unroll
        for (int i = 0; i < RP; i++) Rinv[i * RP + m] = rin[i];
    }
}

// Wout: fp32 [128][RP] of this head.  P_out: this head's [128][r] block (fp16 or fp32).
// GREG: the thread's 64 values of G (row d = tid / 2, columns e(i) = (i & 31) + 64 (i >> 5) + 32 (tid & 1), i = 0..63, i.e. two
// runs of 32 consecutive columns) are passed in registers and G in LDS is not used: 10 KB of LDS per workgroup instead of 76,
// so all 1024 heads of a 32-layer call are resident at once instead of two rounds of 512.
template <int RP, bool GREG = false>
__device__ __forceinline__ void gram_solve_phase2(float* __restrict__ G, float* __restrict__ Pa, float* __restrict__ Pb,
                                                  double* __restrict__ Md, double* __restrict__ Rinv,
                                                  const float* __restrict__ P0h, int r, int loop,
                                                  float* __restrict__ Wout, void* __restrict__ P_out, int out_f16,
                                                  const float* greg = nullptr) {
    constexpr int GD = GS_GD, GP = GS_GP;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < GD * RP; i += 256) {
        int d = i / RP, c = i % RP;
        Pa[i] = (c < r) ? P0h[d * r + c] : 0.0f;
    }
    __syncthreads();
    // Y = G X ([128][RP]).  Thread (d = tid / 2, h = tid & 1) accumulates the RP outputs of row d over half of the e range
    // (e = (i & 31) + 64 (i >> 5) + 32 h: with the pitch of 129 floats the 64 lanes of a wave hit 64 different banks, for
    // the direct access G[d][e] and for the mirrored one G[e][d] alike), then the two halves meet through DPP.
    auto matmulG = [&](const float* X, float* Y) {
        const int d = tid >> 1, h = tid & 1;
        const int dlow = d & ~31;        // e < dlow lies below the diagonal blocks
        float acc[RP];
#pragma unroll
        for (int c = 0; c < RP; c++) acc[c] = 0.0f;
        auto step = [&](int i, float g) {
            const int e = (i & 31) + 64 * (i >> 5) + 32 * h;
#pragma unroll
            for (int c4 = 0; c4 < RP; c4 += 4) {
                const float4 xv = *(const float4*)&X[e * RP + c4];
                acc[c4] = fmaf(g, xv.x, acc[c4]);
                acc[c4 + 1] = fmaf(g, xv.y, acc[c4 + 1]);
                acc[c4 + 2] = fmaf(g, xv.z, acc[c4 + 2]);
                acc[c4 + 3] = fmaf(g, xv.w, acc[c4 + 3]);
            }
        };
        if (GREG) {     // (fully unrolled: the register array needs compile-time indices)
#pragma unroll
            for (int i = 0; i < 64; i++) step(i, greg[i]);
        } else {
#pragma unroll 4
            for (int i = 0; i < 64; i++) {
                const int e = (i & 31) + 64 * (i >> 5) + 32 * h;
                step(i, (e < dlow) ? G[e * GP + d] : G[d * GP + e]);
            }
        }
#pragma unroll
        for (int c = 0; c < RP; c++) acc[c] = GEAR_DPP_ADD(acc[c], 0xB1);   // + the partner lane (quad_perm xor 1)
        if (h == 0) {
#pragma unroll
            for (int c4 = 0; c4 < RP; c4 += 4) *(float4*)&Y[d * RP + c4] = make_float4(acc[c4], acc[c4 + 1], acc[c4 + 2], acc[c4 + 3]);
        }
        __syncthreads();
    };
    // Md = R^T R  ->  Rinv = R^-1 (upper); dependent / zero columns -> 0.  Lane m of wave 0 owns column m of R and of
    // R^-1; the pivots' reciprocal square roots come from v_rsq_f64 + Newton.
    auto chol_inverse = [&]() {
        if (tid < 64) chol_inverse_wave<RP>(Md, Rinv, lane);
        __syncthreads();
    };
    auto gram_small = [&](const float* A, const float* B) {  // Md = A^T B  (fp64 accumulate), 4 lanes per output
        for (int o = tid >> 2; o < RP * RP; o += 64) {
            const int a = o / RP, b = o % RP, part = tid & 3;
            double sacc = 0.0;
#pragma unroll 8
            for (int i = 0; i < GD / 4; i++) {
                const int d = 4 * i + part;
                sacc += (double)A[d * RP + a] * (double)B[d * RP + b];
            }
            sacc += __shfl_xor(sacc, 1, 64);
            sacc += __shfl_xor(sacc, 2, 64);
            if (part == 0) Md[o] = sacc;
        }
        __syncthreads();
    };
    auto apply_rinv = [&](const float* X, float* Y) {  // Y = X Rinv
        for (int i = tid; i < GD * RP; i += 256) {
            int d = i / RP, c = i % RP;
            double s = 0.0;
            for (int a = 0; a <= c; a++) s += (double)X[d * RP + a] * Rinv[a * RP + c];
            Y[i] = (float)s;
        }
        __syncthreads();
    };
    float* cur = Pa;
    float* oth = Pb;
    for (int it = 0; it + 1 < loop; it++) {  // P <- G P, loop-1 times
        matmulG(cur, oth);
        float* t = cur; cur = oth; oth = t;
    }
    // P' = orth(P): CholeskyQR twice (fp64 Gram) -- stable for the column scaling power iteration produces
    for (int rep = 0; rep < 2; rep++) {
        gram_small(cur, cur);
        chol_inverse();
        apply_rinv(cur, oth);
        float* t = cur; cur = oth; oth = t;
    }
    // T1 = G P' ; Q^T Q = P'^T T1 ; W = P' R^-1 ; P_out = T1 R^-1
    matmulG(cur, oth);            // oth = T1
    gram_small(cur, oth);
    chol_inverse();
    for (int i = tid; i < GD * RP; i += 256) {
        int d = i / RP, c = i % RP;
        double sw = 0.0, sp = 0.0;
        for (int a = 0; a <= c; a++) {
            sw += (double)cur[d * RP + a] * Rinv[a * RP + c];
            sp += (double)oth[d * RP + a] * Rinv[a * RP + c];
        }
        Wout[d * RP + c] = (float)sw;
        if (c < r) {
            if (out_f16) ((uint16_t*)P_out)[d * r + c] = f2h_bits((float)sp);
            else ((float*)P_out)[d * r + c] = (float)sp;
        }
    }
}
