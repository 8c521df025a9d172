// lowrank.hip -- rank-r approximation of the quantization error by power iteration (gfx950).
//
// Reference algorithm (cuda_supported_gear/quant/new_pack.py:291-311 headwise_lrap;
// GenerationBench/.../Simulated/compress_function.py:69-98 fake_poweriteration_group), all in fp32:
//     for i in range(loop): [last: P = orth(P)]  Q = E P  [last: Q = orth(Q)]  P = E^T Q
// The approximation Q P^T depends only on span(P) before the last E P and span(Q) after it, so the
// orthonormalisation may be any stable one; here: Cholesky-QR with the r x r Gram matrix accumulated in
// fp64 (the reference calls torch.linalg.qr = Householder on LAPACK / cuSOLVER).
//
// Building blocks over a stored matrix M [R x C] (row-major, fp16 or fp32), thin factor X with RP (= r padded
// to 4/8/16) columns in fp32:
//     rowdot : Y[R x RP]  = M   X[C x RP]      (row-local dot products, 16 lanes per row, 8 columns per lane)
//     coldot : Y[C x RP] += M^T X[R x RP]      (lanes own 8 columns and stream rows; LDS + atomics at the end)
// E stored as [S x Dm] ("normal", V and the build's K error) or as its transpose [Dm x S] (the K^T layout).
#include <stdlib.h>

#include "common.h"

namespace {

template <typename ET>
__device__ __forceinline__ void load8(const ET* p, float* f);
template <>
__device__ __forceinline__ void load8<uint16_t>(const uint16_t* p, float* f) {
    uint4 v = *(const uint4*)p;
    unpack8(v, f);
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float* f) {
    float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ---------------------------------------------------------------------------------------------- rowdot
// grid (ceil(R/128), BH), block 256 = 16 row-groups x 16 lanes.  C128: C == 128 -> X lives in registers.
template <typename ET, int RP, bool C128>
__global__ __launch_bounds__(256) void lr_rowdot_kernel(const ET* __restrict__ M, const float* __restrict__ X,
                                                        float* __restrict__ Y, int R, int C) {
    const int64_t bh = blockIdx.y;
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const ET* Mb = M + bh * (int64_t)R * C;
    const float* Xb = X + bh * (int64_t)C * RP;
    float* Yb = Y + bh * (int64_t)R * RP;
    float xr[C128 ? 8 : 1][RP];
    if (C128) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int c = 0; c < RP; c++) xr[i][c] = Xb[(l16 * 8 + i) * RP + c];
    }
    for (int i = 0; i < 8; i++) {
        const int row = blockIdx.x * 128 + grp + 16 * i;
        float acc[RP];
#pragma unroll
        for (int c = 0; c < RP; c++) acc[c] = 0.0f;
        if (row < R) {
            if (C128) {
                float m[8];
                load8<ET>(Mb + (int64_t)row * C + l16 * 8, m);
#pragma unroll
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int c = 0; c < RP; c++) acc[c] = fmaf(m[j], xr[j][c], acc[c]);
            } else {
                for (int c0 = l16 * 8; c0 < C; c0 += 128) {
                    float m[8];
                    load8<ET>(Mb + (int64_t)row * C + c0, m);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float* xp = Xb + (int64_t)(c0 + j) * RP;
#pragma unroll
                        for (int c = 0; c < RP; c++) acc[c] = fmaf(m[j], xp[c], acc[c]);
                    }
                }
            }
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1)
#pragma unroll
            for (int c = 0; c < RP; c++) acc[c] += __shfl_xor(acc[c], d, 64);
        if (l16 == 0 && row < R) {
#pragma unroll
            for (int c = 0; c < RP; c++) Yb[(int64_t)row * RP + c] = acc[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------- coldot
// grid (ceil(C/128), row_splits, BH), block 256 = 16 row-groups x 16 lanes (8 columns per lane).
template <typename ET, int RP>
__global__ __launch_bounds__(256) void lr_coldot_kernel(const ET* __restrict__ M, const float* __restrict__ X,
                                                        float* __restrict__ Y, int R, int C, int rows_per_split) {
    __shared__ float red[4][16][8 * RP];
    const int64_t bh = blockIdx.z;
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 128 + l16 * 8;
    const bool col_ok = c0 < C;
    const ET* Mb = M + bh * (int64_t)R * C;
    const float* Xb = X + bh * (int64_t)R * RP;
    float* Yb = Y + bh * (int64_t)C * RP;
    const int r_begin = blockIdx.y * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);
    float acc[8][RP];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int c = 0; c < RP; c++) acc[j][c] = 0.0f;
    if (col_ok) {
        for (int row = r_begin + grp; row < r_end; row += 16) {
            float m[8], xv[RP];
            load8<ET>(Mb + (int64_t)row * C + c0, m);
#pragma unroll
            for (int c = 0; c < RP; c++) xv[c] = Xb[(int64_t)row * RP + c];
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int c = 0; c < RP; c++) acc[j][c] = fmaf(m[j], xv[c], acc[j][c]);
        }
    }
    // the 4 row-groups of a wave sit at lane bits 4,5
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int c = 0; c < RP; c++) {
            float v = acc[j][c];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[j][c] = v;
        }
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int c = 0; c < RP; c++) red[wave][l16][j * RP + c] = acc[j][c];
    }
    __syncthreads();
    // 16 lanes x 8 cols x RP values, summed over the 4 waves, then one atomic per value
    for (int idx = threadIdx.x; idx < 16 * 8 * RP; idx += 256) {
        const int ll = idx / (8 * RP), rem = idx % (8 * RP);
        const int col = blockIdx.x * 128 + ll * 8 + rem / RP;
        if (col < C) {
            float s = red[0][ll][rem] + red[1][ll][rem] + red[2][ll][rem] + red[3][ll][rem];
            atomicAdd(&Yb[(int64_t)col * RP + rem % RP], s);
        }
    }
}

// ---------------------------------------------------------------------------------------------- Cholesky-QR
// G[bh] (RP x RP, fp64) += A^T A for A [n x RP] fp32.  grid (splits, BH), block 256.
template <int RP>
__global__ __launch_bounds__(256) void lr_gram_kernel(const float* __restrict__ A, double* __restrict__ G, int n,
                                                      int rows_per_split) {
    __shared__ float tile[64][RP];
    constexpr int NP = RP * RP;
    constexpr int NSUB = 256 / NP > 0 ? 256 / NP : 1;
    const int64_t bh = blockIdx.y;
    const float* Ab = A + bh * (int64_t)n * RP;
    const int p = threadIdx.x % NP, rsub = threadIdx.x / NP;
    const int pi = p / RP, pj = p % RP;
    const int r_begin = blockIdx.x * rows_per_split, r_end = min(n, r_begin + rows_per_split);
    double acc = 0.0;
    for (int r0 = r_begin; r0 < r_end; r0 += 64) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 64 * RP; idx += 256) {
            int rr = r0 + idx / RP;
            tile[idx / RP][idx % RP] = (rr < r_end) ? Ab[(int64_t)rr * RP + idx % RP] : 0.0f;
        }
        __syncthreads();
        if (threadIdx.x < NP * NSUB) {
            for (int rr = rsub; rr < 64; rr += NSUB) acc += (double)tile[rr][pi] * (double)tile[rr][pj];
        }
    }
    if (threadIdx.x < NP * NSUB) atomicAdd(&G[bh * NP + p], acc);
}

// A <- A R^-1 with G = R^T R.  grid (ceil(n/256), BH), block 256 (one row per thread).
template <int RP>
__global__ __launch_bounds__(256) void lr_chol_apply_kernel(float* __restrict__ A, const double* __restrict__ G, int n) {
    __shared__ double Rinv[RP][RP];
    const int64_t bh = blockIdx.y;
    if (threadIdx.x == 0) {
        double Rm[RP][RP];
        bool dead[RP];
        const double* g = G + bh * RP * RP;
        for (int j = 0; j < RP; j++) {
            for (int i = 0; i < RP; i++) Rm[i][j] = 0.0;
        }
        for (int j = 0; j < RP; j++) {
            double d = g[j * RP + j];
            for (int kk = 0; kk < j; kk++) d -= Rm[kk][j] * Rm[kk][j];
            dead[j] = !(d > 1e-12 * g[j * RP + j]) || !(g[j * RP + j] > 0.0);
            if (dead[j]) {
                Rm[j][j] = 1.0;  // column contributes nothing (orth of a zero / dependent column -> 0)
                continue;
            }
            double rjj = sqrt(d);
            Rm[j][j] = rjj;
            for (int m = j + 1; m < RP; m++) {
                double s = g[j * RP + m];
                for (int kk = 0; kk < j; kk++) s -= Rm[kk][j] * Rm[kk][m];
                Rm[j][m] = s / rjj;
            }
        }
        // invert the upper-triangular R
        for (int j = 0; j < RP; j++) {
            for (int i = 0; i < RP; i++) Rinv[i][j] = 0.0;
            if (dead[j]) continue;
            Rinv[j][j] = 1.0 / Rm[j][j];
            for (int i = j - 1; i >= 0; i--) {
                double s = 0.0;
                for (int kk = i + 1; kk <= j; kk++) s += Rm[i][kk] * Rinv[kk][j];
                Rinv[i][j] = dead[i] ? 0.0 : -s / Rm[i][i];
            }
        }
    }
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    float* a = A + (bh * (int64_t)n + row) * RP;
    double av[RP];
#pragma unroll
    for (int c = 0; c < RP; c++) av[c] = (double)a[c];
#pragma unroll
    for (int j = 0; j < RP; j++) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i <= j; i++) s += av[i] * Rinv[i][j];
        a[j] = (float)s;
    }
}

// [bh, n, r] <-> padded [bh, n, RP] helpers
__global__ void lr_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t total, int r, int RP) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int c = (int)(i % RP);
    dst[i] = (c < r) ? src[(i / RP) * r + c] : 0.0f;
}
__global__ void lr_unpad_kernel(const float* __restrict__ src, void* __restrict__ dst, int64_t total, int r, int RP,
                                int out_f16) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = src[(i / r) * RP + (i % r)];
    if (out_f16) ((uint16_t*)dst)[i] = f2h_bits(v);
    else ((float*)dst)[i] = v;
}

struct LrWs {
    float* P;
    float* Q;
    double* G;
};

template <typename ET, int RP>
int run_lowrank(const ET* E, int transposed, int64_t bh, int S, int Dm, int r, int loop, const float* P0, void* P_out,
                void* Q_out, int out_dtype, LrWs ws, hipStream_t st) {
    const int R = transposed ? Dm : S, C = transposed ? S : Dm;  // stored matrix M [R x C]
    auto rowdot = [&](const float* X, float* Y) {
        dim3 grid((R + 127) / 128, (unsigned)bh);
        if (C == 128) hipLaunchKernelGGL((lr_rowdot_kernel<ET, RP, true>), grid, dim3(256), 0, st, E, X, Y, R, C);
        else hipLaunchKernelGGL((lr_rowdot_kernel<ET, RP, false>), grid, dim3(256), 0, st, E, X, Y, R, C);
    };
    auto coldot = [&](const float* X, float* Y) {
        (void)hipMemsetAsync(Y, 0, sizeof(float) * (size_t)bh * C * RP, st);
        int chunks = (C + 127) / 128;
        int splits = 1;
        while ((int64_t)chunks * splits * bh < 1024 && R / (splits * 2) >= 64) splits *= 2;
        int rps = (R + splits - 1) / splits;
        splits = (R + rps - 1) / rps;
        dim3 grid(chunks, splits, (unsigned)bh);
        hipLaunchKernelGGL((lr_coldot_kernel<ET, RP>), grid, dim3(256), 0, st, E, X, Y, R, C, rps);
    };
    auto orth = [&](float* A, int n) {
        (void)hipMemsetAsync(ws.G, 0, sizeof(double) * (size_t)bh * RP * RP, st);
        int splits = 1;
        while ((int64_t)splits * bh < 512 && n / (splits * 2) >= 256) splits *= 2;
        int rps = (n + splits - 1) / splits;
        splits = (n + rps - 1) / rps;
        hipLaunchKernelGGL((lr_gram_kernel<RP>), dim3(splits, (unsigned)bh), dim3(256), 0, st, A, ws.G, n, rps);
        hipLaunchKernelGGL((lr_chol_apply_kernel<RP>), dim3((n + 255) / 256, (unsigned)bh), dim3(256), 0, st, A, ws.G, n);
    };
    {
        int64_t total = bh * (int64_t)Dm * RP;
        hipLaunchKernelGGL(lr_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P0, ws.P, total, r, RP);
    }
    for (int it = 0; it < loop; it++) {
        const bool last = (it == loop - 1);
        if (last) orth(ws.P, Dm);
        if (!transposed) rowdot(ws.P, ws.Q); else coldot(ws.P, ws.Q);   // Q = E P
        if (last) orth(ws.Q, S);
        if (!transposed) coldot(ws.Q, ws.P); else rowdot(ws.Q, ws.P);   // P = E^T Q
    }
    {
        int64_t tp = bh * (int64_t)Dm * r, tq = bh * (int64_t)S * r;
        hipLaunchKernelGGL(lr_unpad_kernel, dim3((unsigned)((tp + 255) / 256)), dim3(256), 0, st, ws.P, P_out, tp, r, RP,
                           out_dtype == GEAR_DTYPE_F16);
        hipLaunchKernelGGL(lr_unpad_kernel, dim3((unsigned)((tq + 255) / 256)), dim3(256), 0, st, ws.Q, Q_out, tq, r, RP,
                           out_dtype == GEAR_DTYPE_F16);
    }
    GEAR_CHECK_LAUNCH("gear_lowrank");
    return 0;
}

inline int pad_rank(int r) { return r <= 4 ? 4 : (r <= 8 ? 8 : 16); }

}  // namespace

int gear_lowrank_gram(const void* E, int transposed, int64_t bh, int S, int r, int loop, const void* P0, void* P_out,
                      void* Q_out, int out_dtype, void* workspace, hipStream_t st);
size_t gear_lowrank_gram_workspace(int64_t bh, int S, int RP);

extern "C" size_t gear_lowrank_workspace(int64_t bh, int S, int Dm, int r) {
    if (r < 1 || r > 16) return 0;
    size_t RP = (size_t)pad_rank(r);
    size_t n = (size_t)bh * ((size_t)S + (size_t)Dm) * RP * sizeof(float) + (size_t)bh * RP * RP * sizeof(double);
    if (Dm == 128) {                   // the Gram formulation: W + the slabs' partial Gram matrices (lowrank_gram.hip)
        const size_t g = gear_lowrank_gram_workspace(bh, S, (int)RP);
        if (g > n) n = g;
    }
    return n + 256;
}

extern "C" int gear_lowrank(const void* E, int e_dtype, int transposed, int64_t bh, int S, int Dm, int r, int loop,
                            const void* P0, void* P_out, void* Q_out, int out_dtype, void* workspace,
                            size_t workspace_bytes, void* stream) {
    GEAR_CHECK_ARG(r >= 1 && r <= 16, "gear_lowrank: rank must be in [1,16] (got %d)", r);
    GEAR_CHECK_ARG(loop >= 1, "gear_lowrank: loop must be >= 1 (got %d)", loop);
    GEAR_CHECK_ARG(bh > 0 && bh <= 65535 && S > 0 && Dm > 0, "gear_lowrank: bad shape");
    GEAR_CHECK_ARG(e_dtype == GEAR_DTYPE_F16 || e_dtype == GEAR_DTYPE_F32, "gear_lowrank: bad dtype");
    GEAR_CHECK_ARG(out_dtype == GEAR_DTYPE_F16 || out_dtype == GEAR_DTYPE_F32, "gear_lowrank: bad out dtype");
    const int C = transposed ? S : Dm;
    GEAR_CHECK_ARG(C % 8 == 0, "gear_lowrank: contiguous dim %d must be a multiple of 8", C);
    GEAR_CHECK_ARG(E && P0 && P_out && Q_out && workspace, "gear_lowrank: null pointer");
    GEAR_CHECK_ARG(workspace_bytes >= gear_lowrank_workspace(bh, S, Dm, r), "gear_lowrank: workspace too small");
    // fast path: fp16 error, head_dim 128 -> Gram-matrix formulation on the matrix cores (lowrank_gram.hip)
    if (e_dtype == GEAR_DTYPE_F16 && Dm == 128 && (!transposed || S % 8 == 0) && !gear_options().lowrank_generic)
        return gear_lowrank_gram(E, transposed, bh, S, r, loop, P0, P_out, Q_out, out_dtype, workspace, (hipStream_t)stream);
    const int RP = pad_rank(r);
    LrWs ws;
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    ws.G = (double*)base;
    ws.P = (float*)(base + sizeof(double) * (size_t)bh * RP * RP);
    ws.Q = ws.P + (size_t)bh * Dm * RP;
    hipStream_t st = (hipStream_t)stream;
#define GO(ET, RPV) return run_lowrank<ET, RPV>((const ET*)E, transposed, bh, S, Dm, r, loop, (const float*)P0, P_out, Q_out, out_dtype, ws, st)
    if (e_dtype == GEAR_DTYPE_F16) {
        if (RP == 4) GO(uint16_t, 4);
        if (RP == 8) GO(uint16_t, 8);
        GO(uint16_t, 16);
    } else {
        if (RP == 4) GO(float, 4);
        if (RP == 8) GO(float, 8);
        GO(float, 16);
    }
#undef GO
}
