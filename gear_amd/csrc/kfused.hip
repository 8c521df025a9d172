// kfused.hip -- the K side of GEAR compress, fused, reading token-major K [BH][T][128] exactly as the model produces it.
//
// Reference semantics: gears_channelQ + fake_poweriteration_group
// (GenerationBench/GenerationTest/GEARLM/Simulated/compress_function.py:261-296, :69-98, :204-220) and, in fp16-stepwise
// mode, key_compression (cuda_supported_gear/modeling_llamagear.py:23-38 with new_pack.py:253-311).  The reference does
// `key.transpose(2, 3).contiguous()` first (modeling_llamagear.py:268, :403) and then makes 2 + 2*loop passes over the
// data; round 1 of this build did transpose -> row compressor -> Gram -> Q pass with the fp16 error matrix making a round
// trip through HBM in the K^T layout.  Here:
//
//   k_select_kernel   one sweep: per-channel outlier selection over all T tokens (exact top-k / bottom-k, ties "lower
//                     index first").  Lanes own channel pairs and stream tokens; thresholds guessed from a token
//                     sample, validated by the candidate counts, exact slow path otherwise.  Out: the sparse payload
//                     (sorted lists), the row means (fill value) and a bitmap [BH][T/64][128] x 64 bit.
//   k_main_kernel     one sweep, everything dense: a wave owns a 64-token x 128-channel tile with lane = channel pair, so a
//                     quantization group (64 or 32 tokens of one channel) lives in ONE lane's registers: min/max, scale,
//                     quantize, bit-pack along T (straight into the channel-major K^T payload layout the decode
//                     attention streams, with a row pitch / token offset so that the streaming cache is written in
//                     place) and the fp16 error, which goes into the wave's LDS tile and from there into
//                     v_mfma_f32_32x32x16_f16 for the per-head Gram matrix G = E^T E.  The error matrix never reaches HBM
//                     (no E^T, no transpose kernel, no Gram pass over HBM).
//   k_solve_kernel    per head: sum the slabs' partial Gram matrices, power iteration + CholeskyQR2 in LDS
//                     (lowrank_solve.h) -> W, P.
//   k_qpass_kernel    Q' = E W with E rebuilt on the fly: x tile + the codes / scale / mn k_main_kernel just stored + the
//                     outlier bitmap -> the same E bits -> LDS tile -> matrix cores (W as fp16 head + remainder).
//
// HBM traffic per K tensor of n elements: read 2n (select) + read 2n, write n*b/8 + 8n/g (main) + read 2n + n*b/8 + 8n/g
// (Q pass) = 6.9n bytes against 12.6n for the round-1 chain (transpose r+w 4n, rows r 2n + w 2.3n, Gram 2n, Q 2n + ...) and
// 8.3n with the error written once and read back (measured: k_main 0.60 -> 0.47 ms, Q pass 0.22 -> 0.34 ms).
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "lowrank_solve.h"
#include "ktile.h"


int gear_lowrank_gram_ex(const void* E, int transposed, int64_t bh, int S, int r, int loop, const void* P0, void* P_out,
                         int64_t p_inner, int64_t p_outer_stride, void* Q_out, int q_tcap, int q_toff, int out_dtype,
                         void* workspace, hipStream_t st);
size_t gear_lowrank_gram_workspace(int64_t bh, int S, int RP);
int gear_compress_rows_ext(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride, int nseg,
                           int seglen, int64_t seg_stride, int64_t o_outer_stride, int64_t o_inner_stride, int64_t o_seg_stride,
                           int o_list_outer, int group, int bits, int mode, int k, int col0, const void* thr, const void* fill,
                           void* code, void* scale, void* mn, void* err, void* oidx, void* oval, void* stream);
int gear_compress_rows_geom(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride,
                            int nseg, int seglen, int64_t seg_stride, int64_t o_outer_stride, int64_t o_inner_stride,
                            int64_t o_seg_stride, int o_list_outer, int group, int bits, int mode, int k, void* code,
                            void* scale, void* mn, void* err, void* oidx, void* oval, void* omean, void* stream);

// kone.hip: selection + dense part + Gram in ONE launch with one read of K (fp32 arithmetic, shapes it plans for)
bool gear_kone_supported(int64_t BH, int T, int group, int bits, int mode, int k);
size_t gear_kone_workspace(int64_t BH, int T, int k);
const uint32_t* gear_kone_headfail(void* ws, int64_t BH, int T, int k);
int gear_kone_launch(const void* x, int64_t BH, int T, int group, int bits, int k, void* code, void* scale, void* mn, int64_t ldc,
                     int64_t lds, int t_off, void* obits, void* oidx, void* oval, int kcap, int o_off, float* G, void* ws,
                     hipStream_t st);
int gear_kdense_launch(const void* x, const void* obits, const void* omean, int64_t BH, int T, int group, int bits, void* code, void* scale,
                       void* mn, int64_t ldc, int64_t lds, int t_off, float* gpart, void* eout, int nwg, const uint32_t* only_if, hipStream_t st);
int gear_lr_qpass_tm_launch(const void* E, const float* W, int64_t bh, int S, int r, void* Q_out, int out_f16, int q_tcap, int q_toff,
                            hipStream_t st);

namespace {

// phase clocks of k_select_kernel (measurement builds only: EXTRA=-DGEAR_KS_CLK; tools/exp_kselect_clk.py)
#ifdef GEAR_KS_CLK
__device__ unsigned long long ks_clk_buf[8 * 4096];
#define KS_CLK_AT(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) ks_clk_buf[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define KS_CLK_AT(k) do { } while (0)
#endif

constexpr int KS_CAP = 160;      // candidate slots per (channel, side) list
constexpr int KS_STRIDE = 164;   // words between lists in LDS: 4 words of skew keep the quad-per-list reads conflict-free
constexpr int KS_B = 8;          // words per candidate batch

// sum over the 4 lanes of a DPP quad, result in all four (quad_perm xor 1, xor 2)
__device__ __forceinline__ int quad_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    return v;
}

// ================================================================================================ select
// block L of the 1-D grid -> (quarter q of the 128 channels, head bh).  With BH % 8 == 0 the four quarters of a head are
// blocks L, L+8, L+16, L+24: the same XCD (block b runs on XCD b % 8), dispatched together, so the two 64-byte halves of
// every 128-byte line of K meet in one L2.
__device__ __forceinline__ void select_block_map(int L, int64_t BH, int& q, int64_t& bh) {
    if ((BH & 7) == 0) {
        const int xcd = L & 7, i = L >> 3;
        q = i & 3;
        bh = xcd + 8 * (int64_t)(i >> 2);
    } else {
        q = L & 3;
        bh = L >> 2;
    }
}

struct SelArgs {
    const uint16_t* x;      // [BH][T][128]
    int64_t BH;
    int T, k;
    float zthr;             // candidate threshold = mean +- zthr * sd of the token sample
    int sstride;            // phase A samples every sstride-th token of a stream
    float rlen;             // 1 / T
    uint32_t* obits;        // [BH][T/64][128][2] words: bit t%64 of (tile t/64, channel) = outlier (either side)
    float* omean;           // [BH][128]
    uint16_t* oidx;         // [BH][128][2][kcap]: side slot 0 = the k smallest, 1 = the k largest, each ascending by token
    uint16_t* oval;
    int kcap, o_off, tok_base;
    uint32_t* todo_cnt;     // lists the candidate path could not decide (count outside [k, KS_CAP]): handled by k_select_fix_kernel
    uint32_t* todo;         // [BH * 256] entries (bh * 128 + channel) * 2 + side
    const uint32_t* only_if; // null, or [BH] words: heads whose word is 0 are skipped (the exact chain behind kone.hip)
};

__global__ __launch_bounds__(256) void k_select_kernel(SelArgs a) {
    // dynamic LDS: cand [64 lists][KS_STRIDE] | cnt [64] | kthr [64] | scratch (phase A: stat [16][16][4] floats; phase B: stash
    // [256][KS_B]; phase C: per wave 2 bitmaps + 1 prefix array of T/32 words)
    extern __shared__ __attribute__((aligned(16))) uint32_t sl[];
    uint32_t* cand = sl;
    uint32_t* cnt = sl + 64 * KS_STRIDE;
    uint32_t* kthr = cnt + 64;        // per list: composite threshold of the fast path (0: take the slow path)
    uint32_t* scr = kthr + 64;
    __shared__ uint32_t thr_lds[32];

    int q;
    int64_t bh;
    select_block_map((int)blockIdx.x, a.BH, q, bh);
    if (a.only_if && a.only_if[bh] == 0u) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c2 = lane & 15, ts = lane >> 4, s = wave * 4 + ts;      // channel pair in the quarter, stream id (16 streams)
    const int T = a.T, k = a.k;
    const uint16_t* xq = a.x + bh * (int64_t)T * KD + 32 * q;         // this quarter's first channel of token 0
    const int per_stream = (T + 15) >> 4;                             // tokens of a stream: t = s + 16 i
    // word (token t, this lane's channel pair): a block-uniform base (scalar registers) + a 32-bit byte offset, every load
    // issued by every lane (an index past the end is clamped and the word dropped) -- loads under a condition make the
    // compiler wait for ALL outstanding loads at the join, and a 64-bit multiply per load was 10 of this loop's instructions
    const char* xbase = (const char*)xq;
    const uint32_t lane_b = 4u * (uint32_t)c2 + 256u * (uint32_t)s;
    auto ldw = [&](int i) { return *(const uint32_t*)(xbase + (lane_b + 4096u * (uint32_t)i)); };     // token s + 16 i

    // ---------------------------------------------------------------- phase A: sample statistics -> threshold guess
    KS_CLK_AT(0);
    {
        float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
        constexpr int KS_A = 32;      // sampled tokens per trip: their loads fly together (8 per trip made this phase 8 dependent
                                      // round trips per workgroup: 104 us of the kernel for a quarter of its bytes)
        for (int i0 = 0; i0 < per_stream; i0 += KS_A * a.sstride) {
            uint32_t w[KS_A];
#pragma unroll
            for (int j = 0; j < KS_A; j++) w[j] = ldw(min(i0 + j * a.sstride, per_stream - 1));
#pragma unroll
            for (int j = 0; j < KS_A; j++) {
                const uint32_t wj = (i0 + j * a.sstride < per_stream) ? w[j] : 0u;
                const float f0 = h2f_bits((uint16_t)(wj & 0xFFFFu)), f1 = h2f_bits((uint16_t)(wj >> 16));
                s1a += f0; s2a = fmaf(f0, f0, s2a);
                s1b += f1; s2b = fmaf(f1, f1, s2b);
            }
        }
        float* stat = (float*)scr;                                   // [16 streams][16 pairs][4]
        *(float4*)&stat[(s * 16 + c2) * 4] = make_float4(s1a, s2a, s1b, s2b);
        if (tid < 64) cnt[tid] = 0u;
        __syncthreads();
        if (tid < 32) {
            const int pr = tid >> 1, h = tid & 1;
            float t1 = 0.f, t2 = 0.f;
            for (int st = 0; st < 16; st++) {
                t1 += stat[(st * 16 + pr) * 4 + 2 * h];
                t2 += stat[(st * 16 + pr) * 4 + 2 * h + 1];
            }
            // number of sampled tokens (all streams): streams s < T % 16 ... every stream sees ceil((T - s) / 16) tokens
            int n = 0;
            for (int st = 0; st < 16; st++) {
                const int cntst = (T > st) ? ((T - st + 15) >> 4) : 0;
                n += (cntst + a.sstride - 1) / a.sstride;
            }
            const float rn = 1.0f / (float)max(n, 1);
            const float mean = t1 * rn;
            const float sd = sqrtf(fmaxf(t2 * rn - mean * mean, 0.0f));
            uint32_t th = f2h_bits(mean + a.zthr * sd), tl = f2h_bits(mean - a.zthr * sd);
            // never +-0: then clamp(x) != x happens only for x strictly outside [tl, th] (no -0 / +0 artefacts)
            if ((th & 0x7FFFu) == 0u) th = 0x0001u;
            if ((tl & 0x7FFFu) == 0u) tl = 0x8001u;
            thr_lds[tid] = th | (tl << 16);
        }
        __syncthreads();
    }
    const uint32_t tw0 = thr_lds[2 * c2], tw1 = thr_lds[2 * c2 + 1];
    const half2v thi2 = __builtin_bit_cast(half2v, (tw0 & 0xFFFFu) | (tw1 << 16));
    const half2v tlo2 = __builtin_bit_cast(half2v, (tw0 >> 16) | (tw1 & 0xFFFF0000u));

    KS_CLK_AT(1);
    // ---------------------------------------------------------------- phase B: stream every token once
    // Row sums in fp32 (exact products on v_dot2, fp32 accumulate; the reference's own torch.mean is an fp32 reduction too).
    // Candidates: x outside [tlo, thi] <=> clamp(x) != x -- two packed min / max, one xor, one packed "min(d, 1)" per word
    // give a per-lane bit mask (bit 7-j: word j low half, bit 23-j: high half); the lanes then walk their set bits.
    float suma = 0.f, sumb = 0.f;
    {
        const half2v sel_a = {(_Float16)1.0f, (_Float16)0.0f}, sel_b = {(_Float16)0.0f, (_Float16)1.0f};
        const uint32_t thi_u = __builtin_bit_cast(uint32_t, thi2), tlo_u = __builtin_bit_cast(uint32_t, tlo2);
        const float tloA = h2f_bits((uint16_t)(tlo_u & 0xFFFFu)), tloB = h2f_bits((uint16_t)(tlo_u >> 16));
        uint32_t* stash = scr + tid;                                                  // word j of this lane: stash[j * 256]
        // LDS byte offsets (from cnt) of the lane's four lists: [half][side] -> counter, candidate region
        const uint32_t l0 = (uint32_t)(2 * c2) * 2u;
        const int nb = (per_stream + KS_B - 1) / KS_B;
        auto load_batch = [&](int b, uint32_t (&dst)[KS_B]) {      // (a batch past the end re-reads the last one; never used)
            const int i0 = min(b, nb - 1) * KS_B;
#pragma unroll
            for (int j = 0; j < KS_B; j++) dst[j] = ldw(min(i0 + j, per_stream - 1));
        };
        auto process_batch = [&](int b, const uint32_t (&cur)[KS_B]) {
            const int nv = min(KS_B, per_stream - b * KS_B);                       // valid words of this batch (T % 16 == 0)
            uint32_t m = 0u;
#pragma unroll
            for (int j = 0; j < KS_B; j++) {
                const uint32_t wj = (j < nv) ? cur[j] : 0u;
                const half2v xv = __builtin_bit_cast(half2v, wj);
                suma = __builtin_amdgcn_fdot2(xv, sel_a, suma, false);
                sumb = __builtin_amdgcn_fdot2(xv, sel_b, sumb, false);
                const uint32_t cl = pkmin16(pkmax16(wj, tlo_u), thi_u);
                m = (m << 1) | pkminu16(cl ^ wj, 0x00010001u);
                stash[j * 256] = wj;
            }
            m &= ((0xFFu << (KS_B - nv)) & 0xFFu) * 0x00010001u;
            const int tb0 = s + 16 * (b * KS_B + KS_B - 1);                         // token of word j: tb0 - 16 (p & 15)
            while (m) {
                const int p = 31 - __clz(m);
                m &= ~(1u << p);
                const int half = p >> 4, jr = p & 15;                               // word j = KS_B - 1 - jr
                const uint32_t bits = (stash[(KS_B - 1 - jr) * 256] >> (16 * half)) & 0xFFFFu;
                const bool is_lo = h2f_bits((uint16_t)bits) < (half ? tloB : tloA); // outside [tlo, thi] and not below: above
                const uint32_t list = l0 + 2u * (uint32_t)half + (is_lo ? 1u : 0u);
                const uint32_t slot = atomicAdd(&cnt[list], 1u);
                if (slot < (uint32_t)KS_CAP) cand[list * KS_STRIDE + slot] = (bits << 16) | (uint32_t)(tb0 - 16 * jr);
            }
        };
        // three named batch buffers, the loop body unrolled by three: a batch is processed two batch times after its loads were
        // issued (a "cur = next" register copy made the compiler wait for the loads issued ONE batch time before, which is less
        // than an HBM round trip: ~1.2 us of stall per batch, 40 us per workgroup)
        uint32_t bufA[KS_B], bufB[KS_B], bufC[KS_B];
        // (every load_batch is unconditional -- the conditions guard LDS work only -- so that the compiler can count: "the
        // batch I need is followed by exactly two younger ones")
        load_batch(0, bufA);
        load_batch(1, bufB);
#pragma unroll 1
        for (int b = 0; b < nb; b += 3) {
            load_batch(b + 2, bufC);
            process_batch(b, bufA);
            load_batch(b + 3, bufA);
            if (b + 1 < nb) process_batch(b + 1, bufB);
            load_batch(b + 4, bufB);
            if (b + 2 < nb) process_batch(b + 2, bufC);
        }
    }
    KS_CLK_AT(2);
    __syncthreads();                                   // the stash is dead: its space takes the 16 streams' partial sums
    KS_CLK_AT(3);
    float* sum_lds = (float*)scr;                      // [16][32]
    sum_lds[s * 32 + 2 * c2] = suma;
    sum_lds[s * 32 + 2 * c2 + 1] = sumb;
    __syncthreads();
    if (tid < 32) {
        float tot = 0.f;
        for (int st = 0; st < 16; st++) tot += sum_lds[st * 32 + tid];
        a.omean[bh * KD + 32 * q + tid] = ((T & (T - 1)) == 0) ? tot * a.rlen : tot / (float)T;
    }
    if (k <= 0) return;
    __syncthreads();                                   // (phase C reuses the scratch area)

    KS_CLK_AT(4);
    // ---------------------------------------------------------------- phase C.1: the threshold of every list at once
    // One DPP quad per list (16 lists per wave, 64 per workgroup): lane i4 of the quad holds candidates i4, i4 + 4, ... as
    // composite keys (16-bit order key of the value, then 14 bits "earlier token first"), unique per list, and the quad finds
    // the k-th largest by 31 rounds of bisection -- compare + add-with-carry per candidate, two DPP adds per round, no
    // scalar dependency chain.  A list whose count is outside [k, KS_CAP] is left to the exact slow path below.
    {
        const int L = 16 * wave + (lane >> 2), i4 = lane & 3, side = L & 1;
        const int n = (int)cnt[L];
        const bool fast = n >= k && n <= KS_CAP;
        int jm = fast ? (n + 3) >> 2 : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) jm = max(jm, __shfl_xor(jm, d, 64));
        jm = __builtin_amdgcn_readfirstlane(jm);                                   // candidates per lane, max over the wave
        uint32_t key[KS_CAP / 4];
#pragma unroll
        for (int j = 0; j < KS_CAP / 4; j++) {
            const int g = i4 + 4 * j;
            uint32_t c = 0u;
            if (fast && g < n) c = cand[L * KS_STRIDE + g];
            key[j] = (fast && g < n) ? (((order_key(c >> 16, side) << 14) | (0x3FFFu - (c & 0x3FFFu))) + 1u) : 0u;
        }
        // The k-th largest composite key = the largest K with count(key >= K) >= k.  Two stages instead of 31 bisection rounds over
        // the 30 key bits (20 % of the kernel: profiles/r6_kselect_phase_clocks.md): 16 rounds on the 16-bit value part, then the
        // ties at that value -- the `need` earliest tokens among them -- by walking down from the top of the tie range (one
        // maximum per tie taken; data with distinct values at the cut take exactly one), 14 more rounds only past four ties.
        auto count_ge = [&](uint32_t thr) __attribute__((always_inline)) {
            int c = 0;
#pragma unroll
            for (int blk = 0; blk < KS_CAP / 32; blk++) {
                if (blk * 8 < jm) {
#pragma unroll
                    for (int j = blk * 8; j < blk * 8 + 8; j++) c += (key[j] >= thr) ? 1 : 0;
                }
            }
            return quad_sum_i32(c);
        };
        uint32_t lo_v = 0u, hi_v = 0xFFFFu;                                        // value part: largest V with count(value >= V) >= k
        for (int it = 0; it < 16; it++) {
            const uint32_t mid = lo_v + ((hi_v - lo_v + 1u) >> 1);
            const bool take = count_ge((mid << 14) + 1u) >= k;
            lo_v = take ? mid : lo_v;
            hi_v = take ? hi_v : mid - 1u;
        }
        const uint32_t base = (lo_v << 14) + 1u, top = ((lo_v + 1u) << 14) + 1u;    // composites of value lo_v: [base, top)
        const int need = k - count_ge(top);                                         // ties still to take (>= 1 for a fast list)
        uint32_t lo_b = base;
        int nmax = fast ? need : 0;                                                 // the most ties any list of the wave takes
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d, 64));
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        if (nmax > 4) {
            // many equal values at the cut: bisection on the token part, inside the tie range
            uint32_t hi_b = top - 1u;
            for (int it = 0; it < 14; it++) {
                const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
                const bool take = count_ge(mid) >= k;
                lo_b = take ? mid : lo_b;
                hi_b = take ? hi_b : mid - 1u;
            }
        } else {
            uint32_t bound = top;                                                   // (exclusive)
            for (int t = 0; t < nmax; t++) {
                uint32_t m = 0u;
#pragma unroll
                for (int blk = 0; blk < KS_CAP / 32; blk++) {
                    if (blk * 8 < jm) {
#pragma unroll
                        for (int j = blk * 8; j < blk * 8 + 8; j++) m = max(m, key[j] < bound ? key[j] : 0u);
                    }
                }
                m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true));
                m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true));
                if (t < need) bound = m;                                            // (m >= base: `need` ties exist)
            }
            lo_b = bound;
        }
        if (i4 == 0) kthr[L] = fast ? lo_b : 0u;
    }
    KS_CLK_AT(5);
    __syncthreads();
    KS_CLK_AT(6);
    // ---------------------------------------------------------------- phase C.2: outputs, one wave per channel
    const int nwords = (T + 31) >> 5;
    uint32_t* bmA = scr + wave * 4 * nwords;       // side 0 (large) bitmap of the current channel
    uint32_t* bmB = bmA + nwords;                  // side 1 (small)
    uint32_t* pfxA = bmB + nwords;                 // exclusive prefix popcount per word, side 0
    uint32_t* pfxB = pfxA + nwords;                //                                     side 1
    const int wpl = (nwords + 63) >> 6;            // bitmap words per lane
    for (int lch = wave; lch < 32; lch += 4) {
        const int ch = 32 * q + lch;
        for (int i = 0; i < wpl; i++) {
            const int w = lane * wpl + i;
            if (w < nwords) { bmA[w] = 0u; bmB[w] = 0u; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // both sides of the channel move through the stages together (mark -> prefix -> emit): three LDS round trips per channel
        // instead of six, the two sides' loads and atomics in flight at the same time
        uint32_t cv[2][3];                                            // candidate (bits << 16 | t)
        bool sel[2][3], fast[2];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            uint32_t* bm = side == 0 ? bmA : bmB;
            const int list = lch * 2 + side;
            const int n = __builtin_amdgcn_readfirstlane((int)cnt[list]);
            const uint32_t kt = (uint32_t)__builtin_amdgcn_readfirstlane((int)kthr[list]);
            fast[side] = kt != 0u;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                cv[side][i] = 0u;
                sel[side][i] = false;
                const int g = lane + 64 * i;
                if (fast[side] && g < n) {
                    cv[side][i] = cand[list * KS_STRIDE + g];
                    const uint32_t kx = ((order_key(cv[side][i] >> 16, side) << 14) | (0x3FFFu - (cv[side][i] & 0x3FFFu))) + 1u;
                    sel[side][i] = kx >= kt;
                    if (sel[side][i]) atomicOr(&bm[(cv[side][i] & 0xFFFFu) >> 5], 1u << (cv[side][i] & 31u));
                }
            }
            // threshold guess missed, list overflow or k beyond what the lists hold: this (channel, side) goes to the exact
            // slow selection of k_select_fix_kernel (which also ORs its bits into the channel's bitmap)
            if (!fast[side] && lane == 0) a.todo[atomicAdd(a.todo_cnt, 1u)] = (uint32_t)((bh * KD + ch) * 2 + side);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // exclusive prefix popcount per bitmap word -> rank of a selected token = its position in the sorted list
        {
            uint32_t c = 0u;                                          // side 0 in the low half, side 1 in the high half
            for (int i = 0; i < wpl; i++) {
                const int w = lane * wpl + i;
                if (w < nwords) c += (uint32_t)__popc(bmA[w]) | ((uint32_t)__popc(bmB[w]) << 16);
            }
            uint32_t base = wave_incl_scan_u32(c) - c;
            for (int i = 0; i < wpl; i++) {
                const int w = lane * wpl + i;
                if (w < nwords) {
                    pfxA[w] = base & 0xFFFFu;
                    pfxB[w] = base >> 16;
                    base += (uint32_t)__popc(bmA[w]) | ((uint32_t)__popc(bmB[w]) << 16);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const uint32_t* bm = side == 0 ? bmA : bmB;
            const uint32_t* pfx = side == 0 ? pfxA : pfxB;
            const int64_t lbase = ((bh * KD + ch) * 2 + (side == 0 ? 1 : 0)) * (int64_t)a.kcap + a.o_off;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                if (sel[side][i]) {
                    const uint32_t t = cv[side][i] & 0xFFFFu;
                    const uint32_t r = pfx[t >> 5] + (uint32_t)__popc(bm[t >> 5] & ((1u << (t & 31u)) - 1u));
                    a.oidx[lbase + r] = (uint16_t)(t + a.tok_base);
                    a.oval[lbase + r] = (uint16_t)(cv[side][i] >> 16);
                }
            }
        }
        // the channel's outlier bitmap (both sides), tile-major for the main kernel: [bh][tile][channel] x 2 words
        const int ntiles = (T + 63) >> 6;
        for (int tile = lane; tile < ntiles; tile += 64) {
            const uint32_t w0 = bmA[2 * tile] | bmB[2 * tile];
            const uint32_t w1 = (2 * tile + 1 < nwords) ? (bmA[2 * tile + 1] | bmB[2 * tile + 1]) : 0u;
            *(uint2*)&a.obits[((bh * ntiles + tile) * KD + ch) * 2] = make_uint2(w0, w1);
        }
        __builtin_amdgcn_wave_barrier();
    }
    KS_CLK_AT(7);
}

// The lists k_select_kernel could not decide (a handful per launch on ordinary data, all of them with option kselect_slow):
// exact selection over the channel's whole column, one WORKGROUP per list -- the kernel's duration is the latency of one list,
// so the list is spread over 256 threads (one wave per list took 75 us for 5 lists).  The column's order keys go to LDS once;
// bisection on the 16-bit key (17 rounds: per-thread count over its keys, wave reduction, one LDS atomic per wave, one barrier);
// then one pass in token order (thread = a contiguous run of tokens, block scan of the per-thread counts) that takes everything
// above the threshold value plus the first `need` ties, writes the sorted list and ORs the bits into the channel's bitmap.
__global__ __launch_bounds__(256) void k_select_fix_kernel(SelArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t fkeys[];      // [T] order keys + 1 (0 = beyond T)
    __shared__ int fcnt[2];
    __shared__ uint32_t fscan[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ntodo = *a.todo_cnt;
    const int T = a.T, k = a.k, ntiles = (T + 63) >> 6;
    const int per = (T + 255) >> 8;                                   // tokens per thread in the ordered pass
    for (uint32_t w = blockIdx.x; w < ntodo; w += gridDim.x) {
        const uint32_t e = a.todo[w];
        const int side = e & 1, ch = (e >> 1) & 127;
        const int64_t bh = e >> 8;
        const uint16_t* xc = a.x + bh * (int64_t)T * KD + ch;       // + t * 128
        const int64_t lbase = ((bh * KD + ch) * 2 + (side == 0 ? 1 : 0)) * (int64_t)a.kcap + a.o_off;
        uint32_t* ob = a.obits + ((bh * ntiles) * KD + ch) * 2;     // + tile * 256 words
        __syncthreads();                                             // (the previous list's keys are dead)
        for (int t = tid; t < per * 256; t += 256) fkeys[t] = (t < T) ? order_key(xc[(int64_t)t * KD], side) + 1u : 0u;
        if (tid == 0) { fcnt[0] = 0; fcnt[1] = 0; }
        __syncthreads();
        auto count_ge = [&](uint32_t thr, int slot) {               // block-wide #{keys >= thr}; fcnt[slot] must be 0 on entry
            int c = 0;
            for (int t = tid; t < per * 256; t += 256) c += (fkeys[t] >= thr) ? 1 : 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
            if (lane == 0) atomicAdd(&fcnt[slot], c);
            if (tid == 0) fcnt[slot ^ 1] = 0;                        // the other slot is free: ready for the next call
            __syncthreads();
            const int tot = fcnt[slot];
            return tot;
        };
        uint32_t lo_b = 1u, hi_b = 0x10000u;
        int slot = 0;
        for (int it = 0; it < 17; it++) {
            const uint32_t mid = lo_b + ((hi_b - lo_b + 1u) >> 1);
            const int c = count_ge(mid, slot);
            slot ^= 1;
            if (c >= k) lo_b = mid; else hi_b = mid - 1u;
        }
        const uint32_t vstar = lo_b;
        const int need = k - count_ge(vstar + 1u, slot);             // how many of the ties at the threshold value are taken
        slot ^= 1;
        // ordered pass: thread tid owns tokens [tid * per, (tid + 1) * per)
        int n_gt = 0, n_eq = 0;
        for (int i = 0; i < per; i++) {
            const uint32_t kk = fkeys[tid * per + i];
            n_gt += (kk > vstar) ? 1 : 0;
            n_eq += (kk == vstar) ? 1 : 0;
        }
        // block exclusive scan of (n_gt, n_eq) packed 16 + 16 bits (each total <= T <= 16384)
        const uint32_t mine = (uint32_t)n_gt | ((uint32_t)n_eq << 16);
        const uint32_t incl = wave_incl_scan_u32(mine);
        if (lane == 63) fscan[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int q = 0; q < wave; q++) base += fscan[q];
        const uint32_t excl = base + incl - mine;
        const int gt_before = (int)(excl & 0xFFFFu), eq_before = (int)(excl >> 16);
        int take_eq = need - eq_before;                               // ties this thread still takes (its first ones)
        take_eq = take_eq < 0 ? 0 : (take_eq > n_eq ? n_eq : take_eq);
        int pos = gt_before + (eq_before < need ? eq_before : need);
        uint32_t bits = 0u;
        for (int i = 0; i < per; i++) {
            const int t = tid * per + i;
            const uint32_t kk = fkeys[t];
            bool sel = kk > vstar;
            if (kk == vstar && take_eq > 0) { sel = true; take_eq--; }
            if (sel) {
                a.oidx[lbase + pos] = (uint16_t)(t + a.tok_base);
                a.oval[lbase + pos] = xc[(int64_t)t * KD];
                pos++;
                bits |= 1u << (t & 31);
            }
            if (((t & 31) == 31 || i == per - 1) && bits) {           // the 32-token word ends here (or the thread's run does)
                atomicOr(&ob[(t >> 6) * (KD * 2) + ((t >> 5) & 1)], bits);
                bits = 0u;
            }
        }
    }
}

// ================================================================================================ main
// float(half HI ? high : low of w) + addend (the convert is exact: one rounding)
template <int HI>
__device__ __forceinline__ float add_mix(uint32_t w, float one, float addend) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(one), "v"(addend));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(one), "v"(addend));
    return r;
}
// half-word masks of token j (0..15) from D = (16 outlier bits of the even channel) | (16 bits of the odd channel) << 16
__device__ __forceinline__ uint32_t mask_of(uint32_t D, int j) {
    const short2v d = __builtin_bit_cast(short2v, D);
    const short sh = (short)(15 - j);
    short2v t = d << (short2v){sh, sh};
    t = t >> (short2v){15, 15};
    return __builtin_bit_cast(uint32_t, t);
}
// flag bits (one per code) -> BITS bits per code, 16 / BITS codes -> 16 bits
template <int BITS>
__device__ __forceinline__ uint32_t spread_half(uint32_t f) {
    if (BITS == 2) {
        uint32_t x = f & 0xFFu;
        x = (x | (x << 4)) & 0x0F0Fu;
        x = (x | (x << 2)) & 0x3333u;
        x = (x | (x << 1)) & 0x5555u;
        return x * 3u;
    } else {
        uint32_t x = f & 0xFu;
        x = (x | (x << 6)) & 0x0303u;
        x = (x | (x << 3)) & 0x1111u;
        return x * 15u;
    }
}

struct MainArgs {
    const uint16_t* x;       // [BH][T][128]
    const uint32_t* obits;   // [BH][T/64][128][2] or null (no outliers)
    const float* omean;      // [BH][128] (only with obits)
    int T, tiles_per_slab, nslab;
    uint32_t* code;          // [BH][128][ldc] words
    void* scale;             // [BH][128][lds]
    void* mn;
    int64_t ldc, lds;
    int t_off;               // token offset of this call inside the payload rows (multiple of 64)
    float* gpart;            // [BH][nslab][128][128] partial Gram matrices (blocks on / above the block diagonal), or null
    const uint32_t* only_if; // null, or [BH] words: heads whose word is 0 are skipped
};

// Mode-1 (fp32 simulated arithmetic) tile on packed registers: everything that is exact in fp16 stays packed (min / max
// with the outlier halves masked to +-inf, error = x - dequant), the quotient is a reciprocal multiply with a 1e-5 tie
// guard (exact division redone for the block of a lane that raises it), codes are packed by an fp32 Horner chain along T.
template <int BITS, int G, typename ST>
__device__ __forceinline__ void tile_fast(const uint32_t (&xr)[64], uint32_t mA0, uint32_t mA1, uint32_t mB0, uint32_t mB1,
                                          float meanA, float meanB, uint32_t (&ew)[64], uint32_t (&cwA)[64 * BITS / 32],
                                          uint32_t (&cwB)[64 * BITS / 32], float (&scA)[64 / G], float (&mnA)[64 / G],
                                          float (&scB)[64 / G], float (&mnB)[64 / G]) {
    constexpr int LEVELS = (1 << BITS) - 1;
    constexpr int HC = 16 / BITS;            // codes per 16-bit half word: one Horner chain
    constexpr float TIE = 0.49999f;
    const uint32_t PINF = 0x7C007C00u, NINF = 0xFC00FC00u;
    // D[w]: outlier bits of tokens 16w .. 16w+15, even channel in the low half, odd channel in the high half
    const uint32_t D[4] = {(mA0 & 0xFFFFu) | (mB0 << 16), (mA0 >> 16) | (mB0 & 0xFFFF0000u),
                           (mA1 & 0xFFFFu) | (mB1 << 16), (mA1 >> 16) | (mB1 & 0xFFFF0000u)};
#pragma unroll
    for (int gi = 0; gi < 64 / G; gi++) {
        // ---- group min / max over the non-outliers (packed, exact)
        uint32_t lo2 = PINF, hi2 = NINF;
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int tk = gi * G + i;
            const uint32_t m = mask_of(D[tk >> 4], tk & 15);
            lo2 = pkmin16(lo2, vbfi(m, PINF, xr[tk]));
            hi2 = pkmax16(hi2, vbfi(m, NINF, xr[tk]));
        }
        float loA = h2f_bits((uint16_t)(lo2 & 0xFFFFu)), loB = h2f_bits((uint16_t)(lo2 >> 16));
        float hiA = h2f_bits((uint16_t)(hi2 & 0xFFFFu)), hiB = h2f_bits((uint16_t)(hi2 >> 16));
        uint32_t gA, gB;                     // the group's outlier bits, per channel
        if (G == 64) { gA = mA0 | mA1; gB = mB0 | mB1; }
        else { gA = gi == 0 ? mA0 : mA1; gB = gi == 0 ? mB0 : mB1; }
        // the fill value (fp32 row mean, compress_function.py:279-283) takes part in min / max when the group holds an outlier
        loA = fmin_raw(loA, gA ? meanA : INFINITY); hiA = fmax_raw(hiA, gA ? meanA : -INFINITY);
        loB = fmin_raw(loB, gB ? meanB : INFINITY); hiB = fmax_raw(hiB, gB ? meanB : -INFINITY);
        const float qsA = div_rn(hiA - loA, (float)LEVELS), qsB = div_rn(hiB - loB, (float)LEVELS);
        const float invA = (qsA != 0.0f) ? __builtin_amdgcn_rcpf(qsA) : 0.0f, invB = (qsB != 0.0f) ? __builtin_amdgcn_rcpf(qsB) : 0.0f;
        scA[gi] = qsA; mnA[gi] = loA; scB[gi] = qsB; mnB[gi] = loB;
        const float2v inv2 = {invA, invB}, qs2 = {qsA, qsB}, mn2 = {loA, loB};
        // code of a filled position: quant(mean)
        float cmA = 0.f, cmB = 0.f;
        if (gA | gB) {
            const float ca = (meanA - loA) * invA, cb = (meanB - loB) * invB;
            cmA = rintf(ca); cmB = rintf(cb);
            if (fabsf(ca - cmA) > TIE) cmA = (qsA != 0.0f) ? rintf(div_rn(meanA - loA, qsA)) : 0.0f;
            if (fabsf(cb - cmB) > TIE) cmB = (qsB != 0.0f) ? rintf(div_rn(meanB - loB, qsB)) : 0.0f;
            cmA = __builtin_amdgcn_fmed3f(cmA, 0.0f, (float)LEVELS);
            cmB = __builtin_amdgcn_fmed3f(cmB, 0.0f, (float)LEVELS);
        }
        const uint32_t repA = (uint32_t)cmA * (0xFFFFu / (uint32_t)LEVELS), repB = (uint32_t)cmB * (0xFFFFu / (uint32_t)LEVELS);
        // ---- quantize / pack / error, one 16-bit half word (HC tokens) at a time
#pragma unroll
        for (int hb = 0; hb < G / HC; hb++) {
            const int tb = gi * G + hb * HC;                 // first token of the block
            float2v rq[HC];
            float dmax = 0.0f;
#pragma unroll
            for (int j = 0; j < HC; j++) {
                const float2v t = {add_mix<0>(xr[tb + j], 1.0f, -loA), add_mix<1>(xr[tb + j], 1.0f, -loB)};
                const float2v c = t * inv2;
                const float2v rr = {rintf(c.x), rintf(c.y)};
                const float2v d = c - rr;
                rq[j] = rr;
                asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(dmax) : "v"(dmax), "v"(d.x), "v"(d.y));
            }
            if (dmax > TIE) {                                // within 1e-5 of a rounding tie: redo by exact division
#pragma unroll
                for (int j = 0; j < HC; j++) {
                    const float xa = h2f_bits((uint16_t)(xr[tb + j] & 0xFFFFu)), xb = h2f_bits((uint16_t)(xr[tb + j] >> 16));
                    rq[j].x = (qsA != 0.0f) ? rintf(div_rn(xa - loA, qsA)) : 0.0f;
                    rq[j].y = (qsB != 0.0f) ? rintf(div_rn(xb - loB, qsB)) : 0.0f;
                }
            }
            float2v hn = {0.0f, 0.0f};
#pragma unroll
            for (int j = HC - 1; j >= 0; j--) {
                rq[j].x = __builtin_amdgcn_fmed3f(rq[j].x, 0.0f, (float)LEVELS);
                rq[j].y = __builtin_amdgcn_fmed3f(rq[j].y, 0.0f, (float)LEVELS);
                hn = hn * (float)(1 << BITS) + rq[j];        // exact: < 2^16
            }
            uint32_t hwA = (uint32_t)hn.x, hwB = (uint32_t)hn.y;
            const uint32_t fA = ((tb < 32 ? mA0 >> tb : mA1 >> (tb - 32))) & ((1u << HC) - 1u);
            const uint32_t fB = ((tb < 32 ? mB0 >> tb : mB1 >> (tb - 32))) & ((1u << HC) - 1u);
            if (fA) { const uint32_t sm = spread_half<BITS>(fA); hwA = (repA & sm) | (hwA & ~sm); }
            if (fB) { const uint32_t sm = spread_half<BITS>(fB); hwB = (repB & sm) | (hwB & ~sm); }
            const int hw = tb / HC;                          // half-word index inside the tile
            if ((hw & 1) == 0) { cwA[hw >> 1] = hwA; cwB[hw >> 1] = hwB; }
            else { cwA[hw >> 1] |= hwA << 16; cwB[hw >> 1] |= hwB << 16; }
            // error = x - fp16(code * scale + mn) (mul then add, unfused, like the reference), 0 at the outlier positions
#pragma unroll
            for (int j = 0; j < HC; j++) {
                const float2v dq = rq[j] * qs2 + mn2;        // -ffp-contract=off: v_pk_mul_f32, v_pk_add_f32
                const uint32_t dw = f2h2_bits(dq.x, dq.y);
                uint32_t e2;
                asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e2) : "v"(xr[tb + j]), "v"(dw));
                ew[tb + j] = vbfi(mask_of(D[(tb + j) >> 4], (tb + j) & 15), 0u, e2);
            }
        }
    }
}

__host__ __device__ constexpr int blk_index(int I, int J) {  // upper-triangular block (I <= J) -> 0..9
    return I * 4 - (I * (I - 1)) / 2 + (J - I);
}

// Which 32x32 blocks of the (block-upper-triangular) Gram matrix a wave accumulates, and which operand sets it needs.
// w0: (0,0) (0,1) (1,1)   w1: (2,2) (2,3) (3,3)   w2: (0,2) (0,3)   w3: (1,2) (1,3)     -- 10 operand-set reads per k-step
template <int W> struct WaveBlocks;
template <> struct WaveBlocks<0> { static constexpr int n = 3; static constexpr int I[3] = {0, 0, 1}, J[3] = {0, 1, 1}; static constexpr int need = 0x3; };
template <> struct WaveBlocks<1> { static constexpr int n = 3; static constexpr int I[3] = {2, 2, 3}, J[3] = {2, 3, 3}; static constexpr int need = 0xC; };
template <> struct WaveBlocks<2> { static constexpr int n = 2; static constexpr int I[3] = {0, 0, 0}, J[3] = {2, 3, 3}; static constexpr int need = 0xD; };
template <> struct WaveBlocks<3> { static constexpr int n = 2; static constexpr int I[3] = {1, 1, 1}, J[3] = {2, 3, 3}; static constexpr int need = 0xE; };

template <int W, bool TR>
__device__ __forceinline__ void gram_tile(const uint16_t* etile, int lane, float16_t (&acc)[3]) {
    typedef WaveBlocks<W> WB;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        half8_t f[4];
#pragma unroll
        for (int S = 0; S < 4; S++)
            if (WB::need & (1 << S)) f[S] = load_operand<TR>(etile, 16 * ks, S, lane);
#pragma unroll
        for (int b = 0; b < WB::n; b++)
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[WB::I[b]], f[WB::J[b]], acc[b], 0, 0, 0);
    }
}

template <int W>
__device__ __forceinline__ void gram_store(float* __restrict__ gp, int lane, const float16_t (&acc)[3], float* __restrict__ scratch) {
    typedef WaveBlocks<W> WB;
    const int x31 = lane & 31, kg = lane >> 5;
    // C layout of the 32x32 MFMA: lane l, reg q -> row (q & 3) + 8 (q >> 2) + 4 (l >> 5), col l & 31
#pragma unroll
    for (int b = 0; b < WB::n; b++) {
#pragma unroll
        for (int q = 0; q < 16; q++)
            gp[(32 * WB::I[b] + (q & 3) + 8 * (q >> 2) + 4 * kg) * KD + 32 * WB::J[b] + x31] = acc[b][q];
        if (WB::I[b] != WB::J[b]) {
            // the mirrored block G(J, I) = G(I, J)^T, transposed through LDS (scratch [32][33] floats of this wave) so that it is
            // stored as rows too: the solve then reads whole rows of G and never a column
#pragma unroll
            for (int q = 0; q < 16; q++) scratch[((q & 3) + 8 * (q >> 2) + 4 * kg) * 33 + x31] = acc[b][q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = 2 * q + kg;                       // row of the mirrored block = column of the block
                gp[(32 * WB::J[b] + row) * KD + 32 * WB::I[b] + x31] = scratch[x31 * 33 + row];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// grid (nslab, BH), 256 threads = 4 waves.  A round = 4 tiles of the slab, one per wave: the wave quantizes its tile
// (registers), stores the payload and the error, puts the error tile into LDS; after a barrier every wave adds, for all four
// tiles, ITS blocks of G = E^T E (the 10 blocks of the upper triangle are split 3 / 3 / 2 / 2 over the waves: 48 accumulator
// registers instead of 160, which is what lets two workgroups share a CU, i.e. two waves per SIMD -- one wave's vector
// arithmetic then overlaps the other's loads, LDS traffic and matrix-core work; with one wave per SIMD those costs simply
// added up: 350 us of loads + 165 arithmetic + 150 Gram + 150 payload stores + 100 error store).
// Phase clocks (profiles/r5_kmain_phase_clocks.md; tools/exp_kmain_clk.py): compiled in only with -DGEAR_KF_CLK=1 (k_main_kernel) or
// =2 (k_qpass_kernel) -- `make -C gear_amd/csrc CXXFLAGS+=-DGEAR_KF_CLK=1` --, thread 0 of the first 4096 workgroups stores s_memtime
// at the marked places, gear_debug_kf_clk copies the table out.  Not part of the shipped library.
#ifdef GEAR_KF_CLK
__device__ unsigned long long kf_clk_buf[8 * 4096];
#define KF_CLK_AT(k) do { if (threadIdx.x == 0) { const unsigned bid_ = blockIdx.y * gridDim.x + blockIdx.x; if (bid_ < 4096) kf_clk_buf[bid_ * 8 + (k)] = __builtin_readcyclecounter(); } } while (0)
#define KF_CLK_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define KF_CLK_AT(k) do { } while (0)
#define KF_CLK_WAIT() do { } while (0)
#endif
#if defined(GEAR_KF_CLK) && GEAR_KF_CLK == 1
#define KM_CLK(k) KF_CLK_AT(k)
#define KM_WAIT() KF_CLK_WAIT()
#else
#define KM_CLK(k) do { } while (0)
#define KM_WAIT() do { } while (0)
#endif
#if defined(GEAR_KF_CLK) && GEAR_KF_CLK == 2
#define KQ_CLK(k) KF_CLK_AT(k)
#define KQ_WAIT() KF_CLK_WAIT()
#else
#define KQ_CLK(k) do { } while (0)
#define KQ_WAIT() do { } while (0)
#endif
template <int BITS, int MODE, int G, typename ST, bool FAST, bool LR, bool TR>
__global__ __launch_bounds__(256, 2) void k_main_kernel(MainArgs a) {
    constexpr int CPW = 32 / BITS;
    constexpr int NW = 64 / CPW;             // code words per channel and tile
    constexpr int NG = 64 / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slab = blockIdx.x;
    const int64_t bh = blockIdx.y;
    if (a.only_if && a.only_if[bh] == 0u) return;
    const int T = a.T, ntiles = T >> 6;
    uint16_t* etiles = (uint16_t*)smem;                         // [4][64][ET_PITCH]
    uint16_t* etile = etiles + wave * 64 * ET_PITCH;            // this wave's error tile

    float16_t acc[3];
    if (LR) {
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[b][q] = 0.0f;
    }
    float meanA = 0.f, meanB = 0.f;
    if (a.obits) {
        const float2 mm = *(const float2*)&a.omean[bh * KD + 2 * lane];
        meanA = mm.x; meanB = mm.y;
    }
    const int tile_lo = slab * a.tiles_per_slab, tile_hi = min(ntiles, tile_lo + a.tiles_per_slab);

    uint32_t xr[64];
    uint4 mk = make_uint4(0, 0, 0, 0);
    auto load_tile = [&](int tile) {
        const uint32_t* xw = (const uint32_t*)(a.x + (bh * T + (int64_t)tile * 64) * KD) + lane;
#pragma unroll
        for (int i = 0; i < 64; i++) xr[i] = xw[i * 64];
        mk = make_uint4(0, 0, 0, 0);
        if (a.obits) mk = *(const uint4*)&a.obits[((bh * ntiles + tile) * KD + 2 * lane) * 2];
    };
    if (tile_lo + wave < tile_hi) load_tile(tile_lo + wave);
#pragma unroll 1
    for (int t0 = tile_lo; t0 < tile_hi; t0 += 4) {
        const int tile = t0 + wave;
        KM_CLK(0);
        KM_WAIT();
        KM_CLK(1);
        if (tile < tile_hi) {
            uint32_t ew[64], cwA[NW], cwB[NW];
            float scA[NG], mnA[NG], scB[NG], mnB[NG];
            if (FAST) tile_fast<BITS, G, ST>(xr, mk.x, mk.y, mk.z, mk.w, meanA, meanB, ew, cwA, cwB, scA, mnA, scB, mnB);
            else tile_generic<BITS, MODE, G, ST>(xr, mk.x, mk.y, mk.z, mk.w, meanA, meanB, ew, cwA, cwB, scA, mnA, scB, mnB);
            KM_CLK(2);
            // ---- payload stores: channel-major rows, this tile's words / groups at the token offset
            const int tok = a.t_off + tile * 64;
            uint32_t* cA = a.code + (bh * KD + 2 * lane) * a.ldc + tok / CPW;
            uint32_t* cB = cA + a.ldc;
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                *(uint4*)(cA + w) = make_uint4(cwA[w], cwA[w + 1], cwA[w + 2], cwA[w + 3]);
                *(uint4*)(cB + w) = make_uint4(cwB[w], cwB[w + 1], cwB[w + 2], cwB[w + 3]);
            }
            ST* sA = (ST*)a.scale + (bh * KD + 2 * lane) * a.lds + tok / G;
            ST* nA = (ST*)a.mn + (bh * KD + 2 * lane) * a.lds + tok / G;
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                st_st<ST>(sA + gi, scA[gi]);
                st_st<ST>(nA + gi, mnA[gi]);
                st_st<ST>(sA + a.lds + gi, scB[gi]);
                st_st<ST>(nA + a.lds + gi, mnB[gi]);
            }
            if (LR) {   // error tile -> LDS (row = token, conflict-free 4-byte stores)
#pragma unroll
                for (int i = 0; i < 64; i++) ((uint32_t*)(etile + i * ET_PITCH))[lane] = ew[i];
            }
        }
        // the registers of x are free: the loads of the wave's next tile fly during the Gram phase
        KM_CLK(3);
        if (tile + 4 < tile_hi) load_tile(tile + 4);
        KM_CLK(4);
        if (LR) {
            __syncthreads();
            KM_CLK(5);
            const int nt = min(4, tile_hi - t0);
            for (int tt = 0; tt < nt; tt++) {
                const uint16_t* et = etiles + tt * 64 * ET_PITCH;
                if (wave == 0) gram_tile<0, TR>(et, lane, acc);
                else if (wave == 1) gram_tile<1, TR>(et, lane, acc);
                else if (wave == 2) gram_tile<2, TR>(et, lane, acc);
                else gram_tile<3, TR>(et, lane, acc);
            }
            KM_CLK(6);
            __syncthreads();
            KM_CLK(7);
        }
    }
    if (!LR) return;
    float* gp = a.gpart + (bh * a.nslab + slab) * (int64_t)(KD * KD);
    float* scratch = (float*)etile;                  // (this wave's error tile is dead: the last round's barrier has passed)
    if (wave == 0) gram_store<0>(gp, lane, acc, scratch);
    else if (wave == 1) gram_store<1>(gp, lane, acc, scratch);
    else if (wave == 2) gram_store<2>(gp, lane, acc, scratch);
    else gram_store<3>(gp, lane, acc, scratch);
}

// ================================================================================================ solve
// grid (BH): G = sum of the slabs' partial Gram matrices, then the solve of lowrank_solve.h.
// P_out head bh lives at (bh / p_inner) * p_outer_stride + (bh % p_inner) * 128 * r elements.
// upper: the matrices hold the 32x32 blocks on and above the block diagonal only (kone.hip adds nothing else): an element below is
// read from its mirror image -- G[e][d], contiguous over the threads of a wave like the direct read.
template <int RP>
__global__ __launch_bounds__(256, 2) void k_solve_kernel(const float* __restrict__ gpart, int nslab, int loop,
                                                         const float* __restrict__ P0, int r, float* __restrict__ Wout,
                                                         void* __restrict__ P_out, int out_f16, int64_t p_inner,
                                                         int64_t p_outer_stride, int upper) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Pa = (float*)smem;
    float* Pb = Pa + GS_GD * RP;
    double* Md = (double*)(Pb + GS_GD * RP);
    double* Rinv = Md + RP * RP;
    const int64_t bh = blockIdx.x;
    const int tid = threadIdx.x;
    const float* gp = gpart + bh * nslab * (int64_t)(KD * KD);
    // thread (d = tid / 2, h = tid & 1): columns e(i) = (i & 31) + 64 (i >> 5) + 32 h of row d, summed over the slabs (the partial
    // matrices are complete: k_main_kernel stores the mirrored blocks too)
    float greg[64];
    {
        const int d = tid >> 1, h = tid & 1;
#pragma unroll
        for (int i = 0; i < 64; i++) greg[i] = 0.0f;
        if (upper) {
            const int dlow = d & ~31;
            for (int s = 0; s < nslab; s++) {
                const float* g = gp + s * (int64_t)(KD * KD);
#pragma unroll
                for (int i = 0; i < 64; i++) {
                    const int e = (i & 31) + 64 * (i >> 5) + 32 * h;
                    greg[i] += (e < dlow) ? g[e * KD + d] : g[d * KD + e];
                }
            }
        } else {
            for (int s = 0; s < nslab; s++) {
                const float4* r0 = (const float4*)(gp + s * (int64_t)(KD * KD) + d * KD + 32 * h);
                const float4* r1 = r0 + 16;                       // + 64 columns
#pragma unroll
                for (int v = 0; v < 8; v++) {
                    const float4 a = r0[v], b = r1[v];
                    greg[4 * v] += a.x; greg[4 * v + 1] += a.y; greg[4 * v + 2] += a.z; greg[4 * v + 3] += a.w;
                    greg[32 + 4 * v] += b.x; greg[32 + 4 * v + 1] += b.y; greg[32 + 4 * v + 2] += b.z; greg[32 + 4 * v + 3] += b.w;
                }
            }
        }
    }
    const int64_t po = (bh / p_inner) * p_outer_stride + (bh % p_inner) * (int64_t)(KD * r);
    gram_solve_phase2<RP, true>(nullptr, Pa, Pb, Md, Rinv, P0 + bh * KD * r, r, loop, Wout + bh * KD * RP,
                                out_f16 ? (void*)((uint16_t*)P_out + po) : (void*)((float*)P_out + po), out_f16, greg);
}


// ================================================================================================ Q pass by recomputation
// Q' = E W without an error matrix in HBM: the wave loads its 64-token tile of x exactly as k_main_kernel does (lane = channel
// pair), reads back the codes / scale / mn that kernel stored (channel-major rows: 16 contiguous bytes per channel and tile at
// 2 bits) and the outlier bitmap, rebuilds E = x - fp16(dequant), 0 at the outliers (the same arithmetic, so the same bits),
// puts the tile into LDS and multiplies by W on the matrix cores like lr_qpass_tm_mfma_kernel (W as fp16 head + remainder).
// ~10 VALU instructions per element pair against 2 bytes per element written by k_main_kernel and read back here.
constexpr int QP_TPW = 16;    // tiles per workgroup of the recomputing Q pass
struct QpArgs {
    const uint16_t* x;       // [BH][T][128]
    const uint32_t* obits;   // [BH][T/64][128][2] or null
    int T;
    const uint32_t* code;    // as MainArgs
    const void* scale;
    const void* mn;
    int64_t ldc, lds;
    int t_off;
    const float* W;          // [BH][128][RP]
    int r;
    uint16_t* Q;             // [BH][q_tcap][r] fp16, tokens from q_toff
    int q_tcap, q_toff;
};

template <int BITS, int MODE, int G, typename ST, int RP>
__global__ __launch_bounds__(256, 3) void k_qpass_kernel(QpArgs a) {
    constexpr int CPW = 32 / BITS;
    constexpr int NW = 64 / CPW;
    constexpr int NG = 64 / G;
    constexpr uint32_t CMASK = (1u << BITS) - 1u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* etiles = (uint16_t*)smem;                                   // [4][32][ET_PITCH]: half a tile per wave at a time
    uint16_t* Ah = etiles + 4 * 32 * ET_PITCH;                            // [16][RP][8]: W as fp16, [k / 8][m][k % 8]
    uint16_t* Al = Ah + 16 * RP * 8;                                      // w - fp16(w)
    ST* smz = (ST*)(Al + 16 * RP * 8);                                    // [2][128][QP_TPW * NG]: scale, mn of the workgroup's tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t bh = blockIdx.y;
    const int T = a.T, ntiles = T >> 6;
    uint16_t* etile = etiles + wave * 32 * ET_PITCH;
    {   // W -> LDS as fp16 head + remainder (the KD * RP / 256 loads of a thread issued together)
        constexpr int NWL = (KD * RP + 255) / 256;
        float wv[NWL];
#pragma unroll
        for (int u = 0; u < NWL; u++) {
            const int idx = tid + 256 * u;
            wv[u] = a.W[bh * KD * RP + min(idx, KD * RP - 1)];
        }
#pragma unroll
        for (int u = 0; u < NWL; u++) {
            const int idx = tid + 256 * u;
            if (idx < KD * RP) {
                const int k = idx / RP, m = idx % RP;
                const uint16_t hi = f2h_bits(wv[u]);
                const int pos = ((k >> 3) * RP + m) * 8 + (k & 7);
                Ah[pos] = hi;
                Al[pos] = f2h_bits(wv[u] - h2f_bits(hi));
            }
        }
    }
    const int tile_lo = blockIdx.x * QP_TPW, tile_hi = min(ntiles, tile_lo + QP_TPW);
    // scale / mn of the workgroup's tiles through LDS, read once as whole sectors: fetched per tile they are 4 bytes per channel
    // row at a 256-byte pitch, and the x stream turns the L2 over in microseconds -- every one of those reads then cost a
    // 64-byte sector from HBM (1 GB per launch instead of 0.07)
    {
        constexpr int NV = QP_TPW * NG;                                   // values per (array, channel)
        const int arr = tid >> 7, ch = tid & 127;
        const ST* src = (const ST*)(arr ? a.mn : a.scale) + (bh * KD + ch) * a.lds + (a.t_off + tile_lo * 64) / G;
        const int nv = (tile_hi - tile_lo) * NG;
        // (all NV loads in flight at once, from clamped addresses: `i < nv ? src[i] : 0` compiled to NV conditional loads with a
        // full wait after each -- 16 serialized memory round trips at the head of every workgroup)
        ST vals[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) vals[i] = src[min(i, nv - 1)];
#pragma unroll
        for (int i = 0; i < NV; i++) smz[(arr * KD + ch) * NV + i] = i < nv ? vals[i] : (ST)0;
    }
    __syncthreads();
    const uint32_t* crow = a.code + (bh * KD + 2 * lane) * a.ldc;

    uint32_t xr[64], cw[2][NW];
    uint4 mk = make_uint4(0, 0, 0, 0);
    float sc[2][NG], zp[2][NG];
    auto load_tile = [&](int tile) {
        const uint32_t* xw = (const uint32_t*)(a.x + (bh * T + (int64_t)tile * 64) * KD) + lane;
#pragma unroll
        for (int i = 0; i < 64; i++) xr[i] = xw[i * 64];
        mk = make_uint4(0, 0, 0, 0);
        if (a.obits) mk = *(const uint4*)&a.obits[((bh * ntiles + tile) * KD + 2 * lane) * 2];
        const int tok = a.t_off + tile * 64;
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                const uint4 c = *(const uint4*)(crow + h * a.ldc + tok / CPW + w);
                cw[h][w] = c.x; cw[h][w + 1] = c.y; cw[h][w + 2] = c.z; cw[h][w + 3] = c.w;
            }
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                sc[h][gi] = ld_st<ST>(smz + (2 * lane + h) * (QP_TPW * NG) + (tile - tile_lo) * NG + gi);
                zp[h][gi] = ld_st<ST>(smz + (KD + 2 * lane + h) * (QP_TPW * NG) + (tile - tile_lo) * NG + gi);
            }
        }
    };
    const int n = lane & 31, kg = lane >> 5;
    union U { uint4 u; half8_t h; };
    if (tile_lo + wave < tile_hi) load_tile(tile_lo + wave);
#pragma unroll 1
    for (int tile = tile_lo + wave; tile < tile_hi; tile += 4) {
        KQ_CLK(0);
        KQ_WAIT();
        KQ_CLK(1);
        // ---- E half tile (32 tokens) -> LDS (row = token) -> matrix cores, twice
        const uint32_t D[4] = {(mk.x & 0xFFFFu) | (mk.z << 16), (mk.x >> 16) | (mk.z & 0xFFFF0000u),
                               (mk.y & 0xFFFFu) | (mk.w << 16), (mk.y >> 16) | (mk.w & 0xFFFF0000u)};
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int t2 = 0; t2 < 32; t2++) {
                const int tk = 32 * half + t2;
                const int gi = tk / G;
                const int qa = (int)((cw[0][tk / CPW] >> (BITS * (tk % CPW))) & CMASK), qb = (int)((cw[1][tk / CPW] >> (BITS * (tk % CPW))) & CMASK);
                const float da = dequant_one<MODE>(qa, sc[0][gi], zp[0][gi]), db = dequant_one<MODE>(qb, sc[1][gi], zp[1][gi]);
                const uint32_t dw = f2h2_bits(da, db);
                uint32_t e2;
                asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e2) : "v"(xr[tk]), "v"(dw));
                ((uint32_t*)(etile + t2 * ET_PITCH))[lane] = vbfi(mask_of(D[tk >> 4], tk & 15), 0u, e2);
            }
            if (half == 0) KQ_CLK(2); else KQ_CLK(4);
            if (half == 1 && tile + 4 < tile_hi) load_tile(tile + 4);      // the next tile's loads fly during the matrix-core phase
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
                b.u = *(const uint4*)(etile + n * ET_PITCH + 16 * ks + 8 * kg);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, b.h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, b.h, acc, 0, 0, 0);
            }
            const int64_t token = (int64_t)tile * 64 + 32 * half + n;
#pragma unroll
            for (int qb = 0; qb < (RP + 7) / 8; qb++) {   // register block qb holds rank columns 8 qb + 4 kg + (0..3)
                const int c0 = 8 * qb + 4 * kg;
                if (c0 >= RP) continue;
                if (a.r == RP) {
                    uint2 v;
                    v.x = (uint32_t)f2h_bits(acc[4 * qb]) | ((uint32_t)f2h_bits(acc[4 * qb + 1]) << 16);
                    v.y = (uint32_t)f2h_bits(acc[4 * qb + 2]) | ((uint32_t)f2h_bits(acc[4 * qb + 3]) << 16);
                    *(uint2*)(a.Q + (bh * a.q_tcap + a.q_toff + token) * (int64_t)RP + c0) = v;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (c0 + i < a.r) a.Q[(bh * (int64_t)a.q_tcap + a.q_toff + token) * a.r + c0 + i] = f2h_bits(acc[4 * qb + i]);
                }
            }
            __builtin_amdgcn_wave_barrier();              // the half tile is read before it is overwritten
            if (half == 0) KQ_CLK(3); else KQ_CLK(5);
        }
    }
}

template <int BITS, int MODE, int G, typename ST>
void launch_qpass(const QpArgs& a, int64_t BH, int RP, hipStream_t st) {
    const int ntiles = a.T / 64;
    const dim3 grid((unsigned)((ntiles + QP_TPW - 1) / QP_TPW), (unsigned)BH);
    const size_t shmem = (size_t)4 * 32 * ET_PITCH * 2 + (size_t)2 * 16 * RP * 8 * 2 + (size_t)2 * KD * QP_TPW * (64 / G) * sizeof(ST);
#define KQ_GO(RPV)                                                                                                     \
    do {                                                                                                               \
        auto kfn = k_qpass_kernel<BITS, MODE, G, ST, RPV>;                                                             \
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);           \
        hipLaunchKernelGGL(kfn, grid, dim3(256), shmem, st, a);                                                        \
    } while (0)
    if (RP == 4) KQ_GO(4); else if (RP == 8) KQ_GO(8); else KQ_GO(16);
#undef KQ_GO
}

double inv_norm_cdf(double p) {  // Acklam's rational approximation (relative error 1.2e-9), 0 < p < 1
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                               1.383577518672690e+02,  -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                               6.680131188771972e+01,  -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00,  2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    const double plow = 0.02425;
    if (p < plow) {
        double q = sqrt(-2 * log(p));
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    if (p > 1 - plow) {
        double q = sqrt(-2 * log(1 - p));
        return -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    double q = p - 0.5, rr = q * q;
    return (((((a[0] * rr + a[1]) * rr + a[2]) * rr + a[3]) * rr + a[4]) * rr + a[5]) * q /
           (((((b[0] * rr + b[1]) * rr + b[2]) * rr + b[3]) * rr + b[4]) * rr + 1);
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct KfWs {       // workspace carve-up
    size_t obits, omean, gpart, W, todo, kone, err, total;
    int nslab, tiles_per_slab;
};
// one: the single-read kernel (kone.hip) does selection + dense part + Gram; its Gram matrix is ONE matrix per head
KfWs kf_workspace(int64_t BH, int T, int k, int rank, bool one = false) {
    KfWs w;
    const int ntiles = T / 64;
    // slabs: enough workgroups to fill the chip (>= ~2 per CU) without drowning the solve in partial Gram traffic
    int nslab = 1;
    while (nslab < 4 && BH * nslab < 2048 && ntiles / (nslab * 2) >= 8) nslab *= 2;       // (8 slabs for the 80 heads of 70B / 8: 0.406 -> 0.435 ms)
    if (gear_options().kfused_nslab > 0) {                                               // (A/B runs)
        nslab = gear_options().kfused_nslab;
        while (nslab > 1 && ntiles / nslab < 4) nslab /= 2;
    }
    if (one) nslab = 1;
    w.nslab = nslab;
    w.tiles_per_slab = (ntiles + nslab - 1) / nslab;
    const int RP = rank <= 4 ? 4 : (rank <= 8 ? 8 : 16);
    size_t off = 0;
    w.obits = off; off += align256(k > 0 ? (size_t)BH * ntiles * KD * 8 : 0);
    w.omean = off; off += align256(k > 0 ? (size_t)BH * KD * 4 : 0);
    w.gpart = off; off += align256(rank > 0 ? (size_t)BH * nslab * KD * KD * 4 : 0);
    w.W = off;     off += align256(rank > 0 ? (size_t)BH * KD * RP * 4 : 0);
    w.todo = off;  off += align256(k > 0 ? 256 + (size_t)BH * 256 * 4 : 0);   // counter (first 256 bytes) + list ids
    w.kone = off;  off += align256(one ? gear_kone_workspace(BH, T, k) : 0);
    // the error matrix k_dense_kernel writes for the Q pass (option kfused_eout)
    w.err = off;   off += align256(rank > 0 && gear_options().kfused_eout > 0 ? (size_t)BH * T * KD * 2 : 0);
    w.total = off + 256;
    return w;
}

template <int BITS, int MODE, int G, typename ST>
void launch_main(const MainArgs& a, int64_t BH, bool fast, bool lr, bool tr, hipStream_t st) {
    const size_t shmem = lr ? (size_t)4 * 64 * ET_PITCH * 2 : 0;
    const dim3 grid((unsigned)a.nslab, (unsigned)BH);
#define KF_GO(FASTV, LRV, TRV)                                                                                         \
    do {                                                                                                               \
        auto kfn = k_main_kernel<BITS, MODE, G, ST, FASTV, LRV, TRV>;                                                  \
        if (shmem > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL(kfn, grid, dim3(256), shmem, st, a);                                                        \
    } while (0)
    if constexpr (MODE == 1) {
        if (fast) {
            if (!lr) KF_GO(true, false, false);
            else if (tr) KF_GO(true, true, true);
            else KF_GO(true, true, false);
            return;
        }
    }
    if (!lr) KF_GO(false, false, false);
    else if (tr) KF_GO(false, true, true);
    else KF_GO(false, true, false);
#undef KF_GO
}

}  // namespace
#ifdef GEAR_KS_CLK
extern "C" int gear_debug_ks_clk(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ks_clk_buf), sizeof(unsigned long long) * 8 * 4096);
}
#endif
#ifdef GEAR_KF_CLK
extern "C" int gear_debug_kf_clk(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kf_clk_buf), sizeof(unsigned long long) * 8 * 4096);
}
#endif

// The per-head solve on partial Gram matrices, for callers outside this file (lowrank_gram.hip: the V-side / K^T Gram kernels
// hand over [BH][nslab][128][128] complete (mirrored) matrices exactly as k_main_kernel does).
static int ksolve_launch(const float* gpart, int nslab, int loop, const float* P0, int r, int64_t BH, float* Wout, void* P_out,
                         int out_f16, int64_t p_inner, int64_t p_outer_stride, int upper, hipStream_t st) {
    const int RP = r <= 4 ? 4 : (r <= 8 ? 8 : 16);
    const size_t shmem = gram_solve_lds_bytes(RP) - (size_t)GS_GD * GS_GP * 4;      // no G in LDS: it lives in registers
#define KF_SOLVE(RPV)                                                                                                    \
    do {                                                                                                                 \
        auto kfn = k_solve_kernel<RPV>;                                                                                  \
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);             \
        hipLaunchKernelGGL(kfn, dim3((unsigned)BH), dim3(256), shmem, st, gpart, nslab, loop, P0, r, Wout, P_out,        \
                           out_f16, p_inner, p_outer_stride, upper);                                                     \
    } while (0)
    if (RP == 4) KF_SOLVE(4); else if (RP == 8) KF_SOLVE(8); else KF_SOLVE(16);
#undef KF_SOLVE
    return 0;
}
int gear_ksolve_launch(const float* gpart, int nslab, int loop, const float* P0, int r, int64_t BH, float* Wout, void* P_out,
                       int out_f16, int64_t p_inner, int64_t p_outer_stride, hipStream_t st) {
    return ksolve_launch(gpart, nslab, loop, P0, r, BH, Wout, P_out, out_f16, p_inner, p_outer_stride, 0, st);
}

extern "C" size_t gear_compress_key_fused_workspace(int64_t BH, int T, int k, int rank) {
    if (BH <= 0 || T <= 0 || T % 64) return 0;
    const size_t a = kf_workspace(BH, T, k, rank).total;
    const size_t b = gear_kone_workspace(BH, T, k) ? kf_workspace(BH, T, k, rank, true).total : 0;
    return a > b ? a : b;
}

extern "C" int gear_compress_key_fused(const void* x, int64_t BH, int T, int group, int bits, int mode, int k, void* code,
                                       void* scale, void* mn, int64_t ldc, int64_t lds, int t_off, int rank, int loop,
                                       const void* P0, void* P_out, int64_t p_inner, int64_t p_outer_stride, void* Q_out,
                                       int q_tcap, int q_toff, void* oidx, void* oval, int kcap, int o_off, int variant,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    GEAR_CHECK_ARG(x && code && scale && mn && workspace, "gear_compress_key_fused: null pointer");
    GEAR_CHECK_ARG(BH > 0 && BH <= 65535, "gear_compress_key_fused: BH must be in [1, 65535] (got %lld)", (long long)BH);
    GEAR_CHECK_ARG(T >= 64 && T % 64 == 0 && T <= 16384, "gear_compress_key_fused: T must be a multiple of 64 in [64, 16384] (got %d)", T);
    GEAR_CHECK_ARG(group == 64 || group == 32, "gear_compress_key_fused: group must be 32 or 64 (got %d)", group);
    GEAR_CHECK_ARG(bits == 2 || bits == 4, "gear_compress_key_fused: bits must be 2 or 4 (got %d)", bits);
    GEAR_CHECK_ARG(mode == GEAR_MODE_FP16_STEPWISE || mode == GEAR_MODE_FP32, "gear_compress_key_fused: bad mode");
    GEAR_CHECK_ARG(k >= 0 && 2 * k <= T, "gear_compress_key_fused: need 0 <= 2k <= T (k = %d, T = %d)", k, T);
    GEAR_CHECK_ARG(t_off >= 0 && t_off % 64 == 0, "gear_compress_key_fused: t_off must be a multiple of 64");
    GEAR_CHECK_ARG(ldc * (32 / bits) >= t_off + T && lds * group >= t_off + T, "gear_compress_key_fused: row pitch too small");
    GEAR_CHECK_ARG((ldc * 4) % 16 == 0, "gear_compress_key_fused: code row pitch must be a multiple of 16 bytes");
    GEAR_CHECK_ARG(rank >= 0 && rank <= 16, "gear_compress_key_fused: rank must be in [0, 16]");
    if (rank > 0) {
        GEAR_CHECK_ARG(loop >= 1 && P0 && P_out && Q_out, "gear_compress_key_fused: low-rank needs loop >= 1, P0, P_out, Q_out");
        GEAR_CHECK_ARG(p_inner >= 1 && q_tcap >= q_toff + T, "gear_compress_key_fused: bad factor geometry");
    }
    if (k > 0) GEAR_CHECK_ARG(oidx && oval && kcap >= o_off + k && t_off + T <= 65536 && T <= 16384, "gear_compress_key_fused: bad outlier geometry");
    // the single-read kernel (kone.hip) replaces select + main with option kfused_one = 1 wherever its plan fits.  It is NOT the
    // default: measured 1.85 ms against the chain's 1.00 ms up to the Gram matrices at bench size (profiles/r6_kone.md: the LDS holds
    // two slabs per CU for the ~50 us that three exchanges take).  Variant bits 8 / 32 (measurement hooks of the chain's kernels),
    // 2 and 64 keep the chain
    const int one_opt = gear_options().kfused_one;
    const bool one = one_opt > 0 && !(variant & (8 | 32 | 64 | 2)) && !gear_options().kselect_slow && !gear_options().kfused_generic &&
                     gear_kone_supported(BH, T, group, bits, mode, k);
    const KfWs ws = kf_workspace(BH, T, k, rank, one);
    GEAR_CHECK_ARG(workspace_bytes >= ws.total, "gear_compress_key_fused: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    uint32_t* obits = k > 0 ? (uint32_t*)(base + ws.obits) : nullptr;
    float* omean = k > 0 ? (float*)(base + ws.omean) : nullptr;
    float* gpart = rank > 0 ? (float*)(base + ws.gpart) : nullptr;
    float* Wws = rank > 0 ? (float*)(base + ws.W) : nullptr;
    const uint32_t* only_if = nullptr;
    bool upper_gram = one;                   // the Gram matrices hold the upper 32x32 blocks only (k_solve_kernel mirrors on load)
    void* eq = nullptr;                      // the error matrix k_dense_kernel has written for the Q pass (option kfused_eout), if any
    if (one) {
        const int rc = gear_kone_launch(x, BH, T, group, bits, k, code, scale, mn, ldc, lds, t_off, obits, oidx, oval, kcap, o_off, gpart,
                                        base + ws.kone, st);
        if (rc != 0) return rc;
        if (k == 0) goto after_main;
        only_if = gear_kone_headfail(base + ws.kone, BH, T, k);       // the chain below redoes the heads whose guess failed
    }

    if (k > 0 && !(variant & 8)) {
        SelArgs sa;
        sa.x = (const uint16_t*)x; sa.BH = BH; sa.T = T; sa.k = k;
        sa.only_if = only_if;
        // candidates per side and row: k + 5 sqrt(k) + 8 expected.  Measured on the 7B / 4k tensor (k = 40): a target of 56 / 64 /
        // 72 / 80 / 92 gives 542 / 591 / 598 / 610 / 628 us of select + 673 / 101 / 79 / 77 / 74 us of fix kernel (the guess is validated by the counts; lists hold KS_CAP)
        const double target = k + 5.0 * sqrt((double)k) + 8.0;
        double p = target / (double)T;
        if (p > 0.5) p = 0.5;
        sa.zthr = (target > 0.6 * KS_CAP || (variant & 2) || gear_options().kselect_slow || one) ? 1e30f : (float)(-inv_norm_cdf(p));   // huge z: no candidates -> slow exact path
        sa.sstride = max(1, T / 1024);
        sa.rlen = 1.0f / (float)T;
        sa.obits = obits; sa.omean = omean; sa.oidx = (uint16_t*)oidx; sa.oval = (uint16_t*)oval;
        sa.kcap = kcap; sa.o_off = o_off; sa.tok_base = t_off;
        sa.todo_cnt = (uint32_t*)(base + ws.todo);
        sa.todo = sa.todo_cnt + 64;
        if (hipMemsetAsync(sa.todo_cnt, 0, 4, st) != hipSuccess) { gear_set_error("gear_compress_key_fused: memset failed"); return -2; }
        const int nwords = (T + 31) / 32;
        const size_t scr_words = (size_t)max(max(16 * 16 * 4, 256 * KS_B), 4 * 4 * nwords);
        const size_t shmem = ((size_t)64 * KS_STRIDE + 128 + scr_words) * 4;
        hipLaunchKernelGGL(k_select_kernel, dim3((unsigned)(4 * BH)), dim3(256), shmem, st, sa);
        GEAR_CHECK_LAUNCH("gear_compress_key_fused(select)");
        const unsigned fix_grid = (unsigned)(BH * 256 < 2048 ? BH * 256 : 2048);   // workgroups loop over the to-do list
        hipLaunchKernelGGL(k_select_fix_kernel, dim3(fix_grid), dim3(256), (size_t)((T + 255) / 256 * 256) * 4, st, sa);
        GEAR_CHECK_LAUNCH("gear_compress_key_fused(select fix)");
        if (variant & 32) return 0;
    }
    {
    MainArgs ma;
    ma.x = (const uint16_t*)x; ma.obits = obits; ma.omean = omean; ma.T = T;
    ma.tiles_per_slab = ws.tiles_per_slab; ma.nslab = ws.nslab;
    ma.code = (uint32_t*)code; ma.scale = scale; ma.mn = mn; ma.ldc = ldc; ma.lds = lds; ma.t_off = t_off;
    ma.gpart = gpart;                        // no error matrix in HBM: the Q pass rebuilds it (k_qpass_kernel)
    ma.only_if = only_if;
    const bool fast = (variant & 1) == 0 && !gear_options().kfused_generic, lr = rank > 0;
    // fp32 arithmetic: the slab kernel of kone.hip (k_dense_kernel: LDS-resident 256-token slabs, outliers substituted, mask-free
    // dense part, Gram in registers across the slabs) in place of k_main_kernel; variant bit 128 / option kfused_main keep k_main_kernel
    bool dense_done = false;
    void* eout = (rank > 0 && gear_options().kfused_eout > 0 && !only_if) ? (void*)(base + ws.err) : nullptr;
    if (mode == GEAR_MODE_FP32 && fast && !(variant & (128 | 4)) && !gear_options().kfused_main && !gear_options().kfused_no_tr) {
        const int rc = gear_kdense_launch(x, obits, omean, BH, T, group, bits, code, scale, mn, ldc, lds, t_off, gpart, eout, ws.nslab, only_if, st);
        if (rc < 0) return rc;
        dense_done = rc == 0;
    }
    if (dense_done) eq = eout;
    upper_gram = upper_gram || dense_done;
    if (!dense_done) {
    const bool tr = (variant & 4) == 0 && !gear_options().kfused_no_tr;
#define KF_DISPATCH(B, M, GG, STT) launch_main<B, M, GG, STT>(ma, BH, fast, lr, tr, st)
    if (mode == GEAR_MODE_FP32) {
        if (bits == 2) { if (group == 64) KF_DISPATCH(2, 1, 64, float); else KF_DISPATCH(2, 1, 32, float); }
        else { if (group == 64) KF_DISPATCH(4, 1, 64, float); else KF_DISPATCH(4, 1, 32, float); }
    } else {
        if (bits == 2) { if (group == 64) KF_DISPATCH(2, 0, 64, uint16_t); else KF_DISPATCH(2, 0, 32, uint16_t); }
        else { if (group == 64) KF_DISPATCH(4, 0, 64, uint16_t); else KF_DISPATCH(4, 0, 32, uint16_t); }
    }
#undef KF_DISPATCH
    GEAR_CHECK_LAUNCH("gear_compress_key_fused(main)");
    }
    }
after_main:
    if (variant & 16) return 0;
    if (rank > 0) {
        const int RP = rank <= 4 ? 4 : (rank <= 8 ? 8 : 16);
        ksolve_launch(gpart, ws.nslab, loop, (const float*)P0, rank, BH, Wws, P_out, 1, p_inner, p_outer_stride, upper_gram ? 1 : 0, st);
        GEAR_CHECK_LAUNCH("gear_compress_key_fused(solve)");
        if (eq)                      // the error matrix is in the workspace: the token-major MFMA Q pass of the V chain reads it
            return gear_lr_qpass_tm_launch(eq, Wws, BH, T, rank, Q_out, 1, q_tcap, q_toff, st);
        QpArgs qa;
        qa.x = (const uint16_t*)x; qa.obits = obits; qa.T = T;
        qa.code = (const uint32_t*)code; qa.scale = scale; qa.mn = mn; qa.ldc = ldc; qa.lds = lds; qa.t_off = t_off;
        qa.W = Wws; qa.r = rank; qa.Q = (uint16_t*)Q_out; qa.q_tcap = q_tcap; qa.q_toff = q_toff;
#define KQ_DISPATCH(B, M, GG, STT) launch_qpass<B, M, GG, STT>(qa, BH, RP, st)
        if (mode == GEAR_MODE_FP32) {
            if (bits == 2) { if (group == 64) KQ_DISPATCH(2, 1, 64, float); else KQ_DISPATCH(2, 1, 32, float); }
            else { if (group == 64) KQ_DISPATCH(4, 1, 64, float); else KQ_DISPATCH(4, 1, 32, float); }
        } else {
            if (bits == 2) { if (group == 64) KQ_DISPATCH(2, 0, 64, uint16_t); else KQ_DISPATCH(2, 0, 32, uint16_t); }
            else { if (group == 64) KQ_DISPATCH(4, 0, 64, uint16_t); else KQ_DISPATCH(4, 0, 32, uint16_t); }
        }
#undef KQ_DISPATCH
        GEAR_CHECK_LAUNCH("gear_compress_key_fused(Q pass)");
        return 0;
    }
    return 0;
}

// ================================================================================================ V side, in place
// V [B][H][T][128] token-major -> V payload written at token row t_off of tensors with tcap token rows per head (the
// streaming cache): the row compressor (rows = tokens across the H heads) with its output geometry, then the Gram-matrix
// power iteration of the error and the Q pass with the factor row pitch.  Three launches + the solve inside the Gram kernel.
extern "C" size_t gear_compress_value_fused_workspace(int64_t B, int H, int T, int rank) {
    if (B <= 0 || H <= 0 || T <= 0) return 0;
    const size_t RP = rank <= 4 ? 4 : (rank <= 8 ? 8 : 16);
    return (rank > 0 ? (size_t)B * H * T * KD * 2 + 256 + gear_lowrank_gram_workspace(B * H, T, (int)RP) : 0) + 512;
}

extern "C" int gear_compress_value_fused(const void* x, int64_t B, int H, int T, int group, int bits, int mode, int k,
                                         void* code, void* scale, void* mn, int tcap, int t_off, int rank, int loop,
                                         const void* P0, void* P_out, int64_t p_inner, int64_t p_outer_stride, void* Q_out,
                                         int q_tcap, int q_toff, void* oidx, void* oval, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    GEAR_CHECK_ARG(x && code && scale && mn && workspace, "gear_compress_value_fused: null pointer");
    GEAR_CHECK_ARG(B > 0 && H > 0 && T > 0 && tcap >= t_off + T && t_off >= 0, "gear_compress_value_fused: bad shape");
    GEAR_CHECK_ARG(workspace_bytes >= gear_compress_value_fused_workspace(B, H, T, rank), "gear_compress_value_fused: workspace too small");
    GEAR_CHECK_ARG(rank == 0 || (P0 && P_out && Q_out && loop >= 1 && q_tcap >= q_toff + T && p_inner >= 1),
                   "gear_compress_value_fused: bad factor geometry");
    const int cpw = 32 / (bits ? bits : 2);
    const int sel = mode == GEAR_MODE_FP16_STEPWISE ? 2 : 4;       // bytes per scale / mn element
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    uint16_t* err = rank > 0 ? (uint16_t*)base : nullptr;
    void* lrws = rank > 0 ? (void*)(base + (((size_t)B * H * T * KD * 2 + 255) & ~(size_t)255)) : nullptr;
    // payload row t of head h of batch b: element offset ((b * H + h) * tcap + t_off + t) * 128
    char* code_o = (char*)code + (size_t)t_off * KD / cpw * 4;
    char* scale_o = (char*)scale + (size_t)t_off * KD / group * sel;
    char* mn_o = (char*)mn + (size_t)t_off * KD / group * sel;
    // sparse part: oidx / oval [B][tcap][2k] rows t_off ..
    uint16_t* oi = k > 0 ? (uint16_t*)oidx : nullptr;
    uint16_t* ov = k > 0 ? (uint16_t*)oval : nullptr;
    // one launch over all batch entries: row (b, t) -> payload row b * (H * tcap) ... + t_off + t, sparse list row b * tcap + t_off + t
    const int rc = gear_compress_rows_geom(x, B * T, T, (int64_t)H * T * KD, KD, H, KD, (int64_t)T * KD, (int64_t)H * tcap * KD, KD,
                                           (int64_t)tcap * KD, tcap, group, bits, mode, k, code_o, scale_o, mn_o, err,
                                           oi ? oi + (int64_t)t_off * (2 * k) : nullptr, ov ? ov + (int64_t)t_off * (2 * k) : nullptr,
                                           nullptr, stream);
    if (rc != 0 || rank == 0) return rc;
    return gear_lowrank_gram_ex(err, 0, B * H, T, rank, loop, P0, P_out, p_inner, p_outer_stride, Q_out, q_tcap, q_toff,
                                GEAR_DTYPE_F16, lrws, (hipStream_t)stream);
}

// gear_compress_value_fused for ONE HEAD SHARD of a tensor whose token rows span the heads of several ranks: the outlier selection
// of every row comes from outside (gear_vsel_candidates -> all-gather -> gear_vsel_thresholds: the k smallest / largest of the
// FULL row, compress_function.py:297-333), everything else -- fill, quantize, pack, error, low-rank step, output geometry -- is
// gear_compress_value_fused's.  col0 = global column of this rank's first element (rank * H_local * 128); thr uint32 [B*T][2],
// fill float [B*T].  Unused list slots: index 0xFFFF, value 0.
extern "C" int gear_compress_value_sharded(const void* x, int64_t B, int H, int T, int group, int bits, int mode, int k, void* code,
                                           void* scale, void* mn, int tcap, int t_off, int rank, int loop, const void* P0, void* P_out,
                                           int64_t p_inner, int64_t p_outer_stride, void* Q_out, int q_tcap, int q_toff, void* oidx,
                                           void* oval, int col0, const void* thr, const void* fill, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    GEAR_CHECK_ARG(x && code && scale && mn && workspace && oidx && oval && thr && fill, "gear_compress_value_sharded: null pointer");
    GEAR_CHECK_ARG(B > 0 && H > 0 && T > 0 && tcap >= t_off + T && t_off >= 0 && k > 0, "gear_compress_value_sharded: bad shape");
    GEAR_CHECK_ARG(workspace_bytes >= gear_compress_value_fused_workspace(B, H, T, rank), "gear_compress_value_sharded: workspace too small");
    GEAR_CHECK_ARG(rank == 0 || (P0 && P_out && Q_out && loop >= 1 && q_tcap >= q_toff + T && p_inner >= 1),
                   "gear_compress_value_sharded: bad factor geometry");
    const int cpw = 32 / (bits ? bits : 2);
    const int sel = mode == GEAR_MODE_FP16_STEPWISE ? 2 : 4;
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    uint16_t* err = rank > 0 ? (uint16_t*)base : nullptr;
    void* lrws = rank > 0 ? (void*)(base + (((size_t)B * H * T * KD * 2 + 255) & ~(size_t)255)) : nullptr;
    char* code_o = (char*)code + (size_t)t_off * KD / cpw * 4;
    char* scale_o = (char*)scale + (size_t)t_off * KD / group * sel;
    char* mn_o = (char*)mn + (size_t)t_off * KD / group * sel;
    const int rc = gear_compress_rows_ext(x, B * T, T, (int64_t)H * T * KD, KD, H, KD, (int64_t)T * KD, (int64_t)H * tcap * KD, KD,
                                          (int64_t)tcap * KD, tcap, group, bits, mode, k, col0, thr, fill, code_o, scale_o, mn_o, err,
                                          (uint16_t*)oidx + (int64_t)t_off * (2 * k), (uint16_t*)oval + (int64_t)t_off * (2 * k), stream);
    if (rc != 0 || rank == 0) return rc;
    return gear_lowrank_gram_ex(err, 0, B * H, T, rank, loop, P0, P_out, p_inner, p_outer_stride, Q_out, q_tcap, q_toff,
                                GEAR_DTYPE_F16, lrws, (hipStream_t)stream);
}
