// How many cycles does a SIMD of gfx950 need per wave64 vector instruction?  DESIGN rounds 2-5 charged 4 (CDNA3's SIMD-16 figure);
// MI355X_MICROARCH.md says 2 (SIMD-32).  Every floor this build argues from hangs on that constant, so: measure it.
//
// Per instruction: a block of 128 copies (assembler .rept, fixed registers), 512 iterations, either INDEPENDENT (16 rotating
// destinations, constant sources) or one DEPENDENT chain (destination = first source), with 1 / 2 / 4 / 8 waves per SIMD
// (workgroups of 256 threads = one wave per SIMD, 1 / 2 / 4 / 8 workgroups per CU, 256 CUs).  Reported: shader cycles
// (s_memtime) from a wave's first to its last instruction, divided by the instructions ALL waves of its SIMD issued in that time
// = cycles per instruction of the SIMD; and the effective clock = s_memtime ticks per s_memrealtime tick (100 MHz).
// Compile on the GPU box: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

constexpr int NREP = 128, ITERS = 512;

// registers: v2..v9 sources (v[2:3], v[4:5], v[6:7] as 64-bit pairs), v10..v41 destinations
#define CLOB "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",     \
             "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36",   \
             "v37", "v38", "v39", "v40", "v41", "vcc", "memory"

// D32 / D64: i-th rotating destination; the dependent flavours name v10 / v[10:11] as destination and first source
#define REPT(body) ".set i, 0\n.rept 128\n" body "\n.set i, i + 1\n.endr\n"
#define D32 "v[10 + (i & 15)]"
#define D64 "v[10 + 2 * (i & 15):11 + 2 * (i & 15)]"

struct Op { const char* name; int id; };

template <int OP, bool DEP>
__device__ __forceinline__ void block() {
    if constexpr (OP == 0) { if (DEP) asm volatile(REPT("v_fma_f32 v10, v10, v2, v3") ::: CLOB); else asm volatile(REPT("v_fma_f32 " D32 ", v2, v3, v4") ::: CLOB); }
    if constexpr (OP == 1) { if (DEP) asm volatile(REPT("v_add_f32 v10, v10, v3") ::: CLOB); else asm volatile(REPT("v_add_f32 " D32 ", v2, v3") ::: CLOB); }
    if constexpr (OP == 2) { if (DEP) asm volatile(REPT("v_pk_fma_f32 v[10:11], v[10:11], v[2:3], v[4:5]") ::: CLOB); else asm volatile(REPT("v_pk_fma_f32 " D64 ", v[2:3], v[4:5], v[6:7]") ::: CLOB); }
    if constexpr (OP == 3) { if (DEP) asm volatile(REPT("v_pk_add_f32 v[10:11], v[10:11], v[4:5]") ::: CLOB); else asm volatile(REPT("v_pk_add_f32 " D64 ", v[2:3], v[4:5]") ::: CLOB); }
    if constexpr (OP == 4) { if (DEP) asm volatile(REPT("v_pk_mul_f32 v[10:11], v[10:11], v[2:3]") ::: CLOB); else asm volatile(REPT("v_pk_mul_f32 " D64 ", v[2:3], v[4:5]") ::: CLOB); }
    if constexpr (OP == 5) { if (DEP) asm volatile(REPT("v_pk_add_f16 v10, v10, v8") ::: CLOB); else asm volatile(REPT("v_pk_add_f16 " D32 ", v8, v9") ::: CLOB); }
    if constexpr (OP == 6) { if (DEP) asm volatile(REPT("v_pk_fma_f16 v10, v10, v8, v9") ::: CLOB); else asm volatile(REPT("v_pk_fma_f16 " D32 ", v8, v9, v9") ::: CLOB); }
    if constexpr (OP == 7) { if (DEP) asm volatile(REPT("v_pk_min_f16 v10, v10, v8") ::: CLOB); else asm volatile(REPT("v_pk_min_f16 " D32 ", v8, v9") ::: CLOB); }
    if constexpr (OP == 8) { if (DEP) asm volatile(REPT("v_cvt_f32_f16 v10, v10") ::: CLOB); else asm volatile(REPT("v_cvt_f32_f16 " D32 ", v8") ::: CLOB); }
    if constexpr (OP == 9) { if (DEP) asm volatile(REPT("v_cvt_f16_f32 v10, v10") ::: CLOB); else asm volatile(REPT("v_cvt_f16_f32 " D32 ", v2") ::: CLOB); }
    if constexpr (OP == 10) { if (DEP) asm volatile(REPT("v_cvt_pk_f16_f32 v10, v10, v3") ::: CLOB); else asm volatile(REPT("v_cvt_pk_f16_f32 " D32 ", v2, v3") ::: CLOB); }
    if constexpr (OP == 11) { if (DEP) asm volatile(REPT("v_fma_mix_f32 v10, v10, v2, v3 op_sel_hi:[0,0,0]") ::: CLOB); else asm volatile(REPT("v_fma_mix_f32 " D32 ", v8, v2, v3 op_sel:[1,0,0] op_sel_hi:[1,0,0]") ::: CLOB); }
    if constexpr (OP == 12) { if (DEP) asm volatile(REPT("v_dot2_f32_f16 v10, v8, v9, v10") ::: CLOB); else asm volatile(REPT("v_dot2_f32_f16 " D32 ", v8, v9, v2") ::: CLOB); }
    if constexpr (OP == 13) { if (DEP) asm volatile(REPT("v_add_f64 v[10:11], v[10:11], v[4:5]") ::: CLOB); else asm volatile(REPT("v_add_f64 " D64 ", v[2:3], v[4:5]") ::: CLOB); }
    if constexpr (OP == 14) { if (DEP) asm volatile(REPT("v_fma_f64 v[10:11], v[10:11], v[2:3], v[4:5]") ::: CLOB); else asm volatile(REPT("v_fma_f64 " D64 ", v[2:3], v[4:5], v[6:7]") ::: CLOB); }
    if constexpr (OP == 15) { if (DEP) asm volatile(REPT("v_cvt_f64_f32 v[10:11], v10") ::: CLOB); else asm volatile(REPT("v_cvt_f64_f32 " D64 ", v2") ::: CLOB); }
    if constexpr (OP == 16) { if (DEP) asm volatile(REPT("v_add_u32 v10, v10, v9") ::: CLOB); else asm volatile(REPT("v_add_u32 " D32 ", v8, v9") ::: CLOB); }
    if constexpr (OP == 17) { if (DEP) asm volatile(REPT("v_and_b32 v10, v10, v9") ::: CLOB); else asm volatile(REPT("v_and_b32 " D32 ", v8, v9") ::: CLOB); }
    if constexpr (OP == 18) { if (DEP) asm volatile(REPT("v_bfi_b32 v10, v10, v8, v9") ::: CLOB); else asm volatile(REPT("v_bfi_b32 " D32 ", v8, v9, v9") ::: CLOB); }
    if constexpr (OP == 19) { if (DEP) asm volatile(REPT("v_perm_b32 v10, v10, v8, v9") ::: CLOB); else asm volatile(REPT("v_perm_b32 " D32 ", v8, v9, v9") ::: CLOB); }
    if constexpr (OP == 20) { if (DEP) asm volatile(REPT("v_lshlrev_b32 v10, 3, v10") ::: CLOB); else asm volatile(REPT("v_lshlrev_b32 " D32 ", 3, v8") ::: CLOB); }
    if constexpr (OP == 21) { if (DEP) asm volatile(REPT("v_max3_f32 v10, v10, v2, v3") ::: CLOB); else asm volatile(REPT("v_max3_f32 " D32 ", v2, v3, v4") ::: CLOB); }
    if constexpr (OP == 22) { if (DEP) asm volatile(REPT("v_med3_f32 v10, v10, v2, v3") ::: CLOB); else asm volatile(REPT("v_med3_f32 " D32 ", v2, v3, v4") ::: CLOB); }
    if constexpr (OP == 23) { if (DEP) asm volatile(REPT("v_rndne_f32 v10, v10") ::: CLOB); else asm volatile(REPT("v_rndne_f32 " D32 ", v2") ::: CLOB); }
    if constexpr (OP == 24) { if (DEP) asm volatile(REPT("v_rcp_f32 v10, v10") ::: CLOB); else asm volatile(REPT("v_rcp_f32 " D32 ", v2") ::: CLOB); }
    if constexpr (OP == 25) { if (DEP) asm volatile(REPT("v_cmp_gt_f32 vcc, v10, v2\nv_cndmask_b32 v10, v3, v4, vcc") ::: CLOB); else asm volatile(REPT("v_cmp_gt_f32 vcc, v2, v3\nv_cndmask_b32 " D32 ", v3, v4, vcc") ::: CLOB); }
    if constexpr (OP == 26) { if (DEP) asm volatile(REPT("v_mov_b32_dpp v10, v10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") ::: CLOB); else asm volatile(REPT("v_mov_b32_dpp " D32 ", v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") ::: CLOB); }
    if constexpr (OP == 27) { if (DEP) asm volatile(REPT("v_cvt_u32_f32 v10, v10") ::: CLOB); else asm volatile(REPT("v_cvt_u32_f32 " D32 ", v2") ::: CLOB); }
    if constexpr (OP == 28) { if (DEP) asm volatile(REPT("v_mul_f32 v10, v10, v2") ::: CLOB); else asm volatile(REPT("v_mul_f32 " D32 ", v2, v3") ::: CLOB); }
    if constexpr (OP == 29) { if (DEP) asm volatile(REPT("v_pk_max_f16 v10, v10, v8") ::: CLOB); else asm volatile(REPT("v_pk_mul_f16 " D32 ", v8, v9") ::: CLOB); }
    if constexpr (OP == 30) { if (DEP) asm volatile(REPT("v_lshl_or_b32 v10, v10, 2, v9") ::: CLOB); else asm volatile(REPT("v_lshl_or_b32 " D32 ", v8, 2, v9") ::: CLOB); }
    if constexpr (OP == 31) { if (DEP) asm volatile(REPT("v_mad_u32_u24 v10, v10, v9, v9") ::: CLOB); else asm volatile(REPT("v_mul_u32_u24 " D32 ", v8, v9") ::: CLOB); }
}

static const char* OPNAME[] = {"v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_add_f16", "v_pk_fma_f16",
                               "v_pk_min_f16", "v_cvt_f32_f16", "v_cvt_f16_f32", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_dot2_f32_f16",
                               "v_add_f64", "v_fma_f64", "v_cvt_f64_f32", "v_add_u32", "v_and_b32", "v_bfi_b32", "v_perm_b32",
                               "v_lshlrev_b32", "v_max3_f32", "v_med3_f32", "v_rndne_f32", "v_rcp_f32", "v_cmp + v_cndmask (2 instr)",
                               "v_mov_b32_dpp", "v_cvt_u32_f32", "v_mul_f32", "v_pk_max_f16 (dep) / v_pk_mul_f16", "v_lshl_or_b32",
                               "v_mad_u32_u24 (dep) / v_mul_u32_u24"};
constexpr int NOPS = 32;

template <int OP, bool DEP>
__global__ __launch_bounds__(256) void rate(unsigned long long* out) {
    // sources: 1.0f / 0.0f patterns that keep every chain finite and normal (no denormal or inf slow paths)
    asm volatile("v_mov_b32 v2, 1.0\nv_mov_b32 v3, 0\nv_mov_b32 v4, 0\nv_mov_b32 v5, 0\nv_mov_b32 v6, 0\nv_mov_b32 v7, 0\n"
                 "v_mov_b32 v8, 0x3c003c00\nv_mov_b32 v9, 0\n"
                 ".set i, 0\n.rept 32\nv_mov_b32 v[10 + i], 0\n.set i, i + 1\n.endr\n" ::: CLOB);
    if (OP == 13 || OP == 14) asm volatile("v_mov_b32 v2, 0\nv_mov_b32 v3, 0x3ff00000\n" ::: CLOB);      // v[2:3] = 1.0 (fp64)
    if (OP == 2 || OP == 4) asm volatile("v_mov_b32 v3, 1.0\n" ::: CLOB);                                  // packed multiplier (1, 1)
    if (OP == 24) asm volatile("v_mov_b32 v10, 1.0\n" ::: CLOB);
    __syncthreads();
    const unsigned long long r0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) block<OP, DEP>();
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}

struct Res { double cyc_wave, cyc_wall, ns_wall, ghz; };

// cyc_wave: a wave's own s_memtime span / instructions of its SIMD (median over waves; valid only while all waves run together);
// cyc_wall: the kernel's hipEvent time x the effective clock / instructions per SIMD (includes launch + tail: ~1 %)
template <int OP, bool DEP>
Res run(unsigned long long* dout, int wps) {
    const int grid = 256 * wps;
    std::vector<unsigned long long> h((size_t)grid * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<OP, DEP>), dim3(grid), dim3(256), 0, 0, dout);       // warm (clocks, instruction cache)
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rate<OP, DEP>), dim3(grid), dim3(256), 0, 0, dout);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), dout, (size_t)grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < grid * 4; w++) { cyc.push_back((double)h[2 * w]); clk.push_back((double)h[2 * w] / ((double)h[2 * w + 1] * 10.0)); }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double n_instr = (double)NREP * ITERS * (OP == 25 ? 2 : 1);
    const double ghz = clk[clk.size() / 2];
    const double ns = (double)ms * 1e6 / (n_instr * wps);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return {cyc[cyc.size() / 2] / (n_instr * wps), ns * ghz, ns, ghz};
}

template <int OP>
void row(unsigned long long* dout) {
    printf("| `%s` |", OPNAME[OP]);
    double ghz = 0;
    for (int wps : {1, 2, 3, 4, 6, 8}) { const Res r = run<OP, false>(dout, wps); printf(" %.2f (%.2f) |", r.cyc_wall, r.cyc_wave); ghz = r.ghz; }
    for (int wps : {1, 2}) { const Res r = run<OP, true>(dout, wps); printf(" %.2f |", r.cyc_wall); }
    printf(" %.2f |\n", ghz);
    fflush(stdout);
}

template <int OP>
void rows(unsigned long long* dout) {
    if constexpr (OP < NOPS) { row<OP>(dout); rows<OP + 1>(dout); }
}

int main() {
    unsigned long long* dout;
    (void)hipMalloc(&dout, (size_t)256 * 8 * 8 * sizeof(unsigned long long));
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int wall_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs, clockRate %d kHz, wall clock %d kHz\n\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, wall_khz);
    printf("Shader cycles per wave64 instruction of ONE SIMD = kernel time (hipEvents) x effective clock / instructions issued on the SIMD;\n"
           "in brackets the same from a wave's own s_memtime span (median).  w = waves per SIMD, all running the same instruction.\n\n");
    printf("| instruction | indep 1 w | 2 w | 3 w | 4 w | 6 w | 8 w | dep chain 1 w | dep 2 w | GHz |\n");
    printf("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n");
    rows<0>(dout);
    return 0;
}
