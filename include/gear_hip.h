/*
 * gear_hip.h -- C ABI of libgear_hip.so: the MI355X (gfx950) KV-cache compress / decompress hot path.
 *
 * This is the drop-in boundary.  The reference exposes this path through Python
 * functions backed by Triton kernels and one pybind11 CUDA extension (`kivi_gemv`,
 * cuda_supported_gear/quant/csrc/pybind.cpp:5-8, gemv_cuda.h:13-21) that take torch::Tensor -- i.e. it has
 * no C ABI of its own.  Each entry point below names the reference interface it replaces; the Python
 * mirror of the reference's operator API (gear_amd/quant/new_pack.py, gear_amd/quant/matmul.py, ...) binds
 * these symbols with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; buffers are caller-allocated, contiguous,
 *     16-byte aligned; nothing here allocates or synchronises;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - return value: 0 = ok, < 0 = error (argument check or launch failure); gear_last_error() returns a
 *     thread-local message for the last failing call;
 *   - fp16 tensors are IEEE binary16; "mode" selects the arithmetic:
 *       GEAR_MODE_FP16_STEPWISE (0): torch-eager fp16 semantics of cuda_supported_gear/quant/new_pack.py
 *                                    (:237-240, :273-278) -- packed payload bit-exact; scale/mn are fp16;
 *       GEAR_MODE_FP32          (1): the simulated path's fp32 arithmetic
 *                                    (GenerationBench/.../Simulated/compress_function.py:14-33, :116-125);
 *                                    scale/mn are float32;
 *   - bits in {2,4} (8 also accepted by the quantizers), fpi = 32/bits codes per int32 word, LSB first
 *     (new_pack.py:104, :148-153).
 */
#ifndef GEAR_HIP_H
#define GEAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEAR_MODE_FP16_STEPWISE 0
#define GEAR_MODE_FP32 1

#define GEAR_DTYPE_F16 0
#define GEAR_DTYPE_F32 1

/* Library / build identification. */
const char* gear_last_error(void);
int gear_abi_version(void);
/* Run-time options: switches that select an alternative, equally exact code path (used by the tests to reach the
 * paths ordinary inputs do not).  Names: "attn_generic", "lowrank_generic", "rows_hist_only", "rows_wg_only", "rows_v1", "rows_masked",
 * "kfused_generic", "kselect_slow", "kfused_no_tr", "gram_fused", "gram_nstg", "decomp_general", "decomp_rpb", "kfused_nslab", "attn_fold" (decode attention: 1 = the merge of the partial results inside the partial launch; off by default, measured
 * no faster), "kfused_main" (1 = k_main_kernel instead of k_dense_kernel for fp32 arithmetic), "kfused_eout" (1 = k_dense_kernel writes the error matrix, the Q pass reads it), "kfused_one" (the
 * single-read K kernel, csrc/kone.hip: 1 = wherever it applies; off by default -- measured slower than the chain), and for the decode
 * attention "attn_gqa_group" (one workgroup per KV head serves its 2 / 4 / 8 query heads), "attn_win_chunk" (fp16 window as one more
 * chunk of the split: 0 = on the vector short-chunk kernel, 1 = always, -1 = never), "attn_keep_chunk_index", "attn_mfma" (the matrix-core variant of the short-chunk kernel: 0 = for grouped-query
 * shapes, 1 = always, -1 = never) -- measured alternatives of round 5, all off / automatic by default (gear_amd/csrc/common.h says
 * what each measured).  Each is also read once from the environment (GEAR_<NAME>) when the library is first used.  Returns 0, or -1
 * for an unknown name. */
int gear_set_option(const char* name, int value);

/* ---- a1 / a2 / a3(V): group quantize + bit-pack along the last dim ------------------------------------
 * Replaces triton_quantize_and_pack_along_last_dim (cuda_supported_gear/quant/new_pack.py:217-250),
 * ..._witherror (:253-288, with ALL columns packed -- reference defect B1) and quant_and_pack_vcache (:30-48).
 *   x     : fp16 [rows, L]           code : int32 [rows, L/fpi]
 *   scale, mn : [rows, L/group]  fp16 (mode 0) / float32 (mode 1)
 *   err   : optional fp16 [rows, L] = x - dequant (mode 0: fp16-stepwise, new_pack.py:277-278;
 *           mode 1: fp16(x - fp16(dequant))), NULL to skip.
 * Requires L % group == 0, group a power of two in [16, 1024], L % 16 == 0.
 * Zero-range groups (reference: NaN, defect B6) produce code 0, scale 0.
 */
int gear_quant_pack_lastdim(const void* x, int64_t rows, int L, int group, int bits, int mode, void* code,
                            void* scale, void* mn, void* err, void* stream);

/* ---- a3(K): token-major K tile, groups of `group` consecutive tokens per channel, packed along T --------
 * Replaces quant_and_pack_kcache (new_pack.py:8-27).
 *   x : fp16 [bh, T, D]   code : int32 [bh, T/fpi, D]   scale, mn : [bh, T/group, D]   err : optional fp16 [bh, T, D]
 * Requires D % 8 == 0, T % group == 0, group % fpi == 0.
 */
int gear_quant_pack_k(const void* x, int64_t bh, int T, int D, int group, int bits, int mode, void* code, void* scale,
                      void* mn, void* err, void* stream);

/* ---- a3: unpack + dequantize to fp16 -----------------------------------------------------------------------
 * Replace unpack_and_dequant_vcache (new_pack.py:69-83) and unpack_and_dequant_kcache (:51-66).
 */
int gear_unpack_dequant_lastdim(const void* code, const void* scale, const void* mn, int64_t rows, int L, int group,
                                int bits, int mode, void* out, void* stream);
int gear_unpack_dequant_k(const void* code, const void* scale, const void* mn, int64_t bh, int T, int D, int group,
                          int bits, int mode, void* out, void* stream);

/* ---- a6: decompress-into-attention GEMV ("outer dim" layout) ---------------------------------------------
 * Replaces kivi_gemv.gemv_forward_cuda_outer_dim (gemv_cuda.cu:518-564; kernels :264-434) AND the
 * transpose().contiguous() re-layout that cuda_bmm_fA_qB_outer performs on every call
 * (cuda_supported_gear/quant/matmul.py:205, :215-216): this entry point consumes the layout the Python
 * operator receives.
 *   out[ba, n] = sum_k a[ba, k] * (scale[bw, k, n/group] * code[bw, k, n] + zero[bw, k, n/group]),  bw = ba / n_rep
 *   a    : fp16 [BA, K]                     qB : int32 [BW, K, N/fpi]   (BW = BA / n_rep)
 *   scale, zero : [BW, K, N/group]          out : fp16 [BA, N]
 *   ldq / lds: row pitch (in int32 words / in scale elements) of qB and scale/zero rows; pass 0 for dense.
 *   n_rep : query heads per KV head (1 = MHA; the reference's `mqa` flag, gemv_cuda.cu:276-279).
 *   workspace: device scratch of at least gear_gemv_outer_workspace(...) bytes (split-K partial sums).
 * fp32 accumulation, one fp16 rounding at the end (gemv_cuda.cu:343-345).  bits in {2,4}; group % fpi == 0.
 */
size_t gear_gemv_outer_workspace(int64_t BA, int K, int N, int bits);
int gear_gemv_outer(const void* a, const void* qB, const void* scale, const void* zero, int64_t BA, int n_rep, int K,
                    int N, int group, int bits, int mode, int64_t ldq, int64_t lds, void* out, void* workspace,
                    size_t workspace_bytes, void* stream);

/* The same GEMV on the reference extension's OWN argument layout -- K innermost, what cuda_bmm_fA_qB_outer produces with its
 * per-call transpose(1, 2).contiguous() (matmul.py:205, :215-216) and hands to kivi_gemv.gemv_forward_cuda_outer_dim
 * (gemv_cuda.h:13-21: _in_feats, _kernel, _scaling_factors, _zeros, bit, group_size, nh, mqa):
 *   in_feats fp16 [BA, K]      kernel int32 [BW, N / fpi, K]      scaling_factors, zeros [BW, N / group, K]      out fp16 [BA, N]
 *   n_rep = 1 (one kernel row block per query head), nh with mqa, or any divisor of the query heads (GQA).
 * For a maintainer who keeps matmul.py line by line and swaps only the extension call; gear_gemv_outer (above) is the entry that
 * makes the re-layout unnecessary.  Same arithmetic, no workspace. */
int gear_gemv_outer_dim(const void* in_feats, const void* kernel, const void* scaling_factors, const void* zeros, int64_t BA,
                        int n_rep, int K, int N, int group, int bits, int mode, void* out, void* stream);

/* ---- a7: dequant GEMV + low-rank correction in two launches -------------------------------------------------------------
 * Replaces matmul_withlrap (cuda_supported_gear/modeling_llamagear.py:54-111): cuda_bmm_fA_qB_outer followed by ~8 eager matmul /
 * permute / slice-assign launches for the prefill factors (pbase[0], qbase[0]) and the per-block factors stacked on a leading
 * dim (pbase[1], qbase[1], :71-85 / :87-108).  The GEMV kernel of gear_gemv_outer writes fp32 partial sums; ONE epilogue kernel
 * reduces them, adds the low-rank terms in fp32 and rounds to fp16 once.
 *   kind 0 "key":   K = head_dim, N = Tp + nbuf * blk tokens;  P0 [BW, Tp, r], Q0 [BW, K, r];  P1 [nbuf, BW, blk, r], Q1 [nbuf, BW, K, r]
 *   kind 1 "value": K = Tp + nbuf * blk tokens, N = head_dim;  P0 [BW, N, r], Q0 [BW, Tp, r];  P1 [nbuf, BW, N, r], Q1 [nbuf, BW, blk, r]
 *   (BW = BA / n_rep KV heads; all factors fp16; nbuf == 0: prefill factors only; dense row pitches.) */
size_t gear_gemv_outer_lrap_workspace(int64_t BA, int K, int N, int bits);
int gear_gemv_outer_lrap(const void* a, const void* qB, const void* scale, const void* zero, int64_t BA, int n_rep, int K, int N,
                         int group, int bits, int mode, int kind, const void* P0, const void* Q0, int Tp, const void* P1,
                         const void* Q1, int nbuf, int blk, int r, void* out, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- a11 (+ a9): one pass per row -- outlier top-k select, mean fill, group quantize, pack, error -------------
 * Replaces gears_channelQ / gears_tokenQ (GenerationBench/.../Simulated/compress_function.py:261-333) and, with
 * k == 0, the fake_groupwise_*_asymmetric_quantization functions (:7-67, :100-160), producing a REAL payload.
 * A row is the unit of outlier selection: a token across all heads (V: nseg = H segments of D) or a channel
 * across all tokens (K in the K^T layout [B,H,D,T]: one segment of T).
 *   row r starts at element (r / rows_inner) * outer_stride + (r % rows_inner) * inner_stride; its element j
 *   lives at + (j / seglen) * seg_stride + j % seglen.          len = nseg * seglen <= 16384.
 *   x fp16; code/scale/mn/err use the same geometry with the last dim divided by fpi / group / 1.
 *   k  : outliers per side per row (compress_function.py:265-267 / :300-303); 0 disables the sparse part.
 *   oidx uint16 [n_rows, 2k] (index within the row; first k = the k smallest, then the k largest, each sorted by
 *   index), oval fp16 [n_rows, 2k] (original values), omean float [n_rows] (optional).  Ties across the selection
 *   boundary: lower index first (torch.topk leaves it implementation-defined).
 *   err: optional fp16, x - fp16(dequant) with 0 at outlier positions.
 */
int gear_compress_rows(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride,
                       int nseg, int seglen, int64_t seg_stride, int group, int bits, int mode, int k, void* code,
                       void* scale, void* mn, void* err, void* oidx, void* oval, void* omean, void* stream);

/* ---- a5 / a11 (K side) fused: token-major K -> complete K payload, no transpose, no K^T intermediates ------------------
 * Replaces, for head_dim 128, the chain  key.transpose(2, 3).contiguous()  (cuda_supported_gear/modeling_llamagear.py:268,
 * :403) -> gears_channelQ (GenerationBench/.../Simulated/compress_function.py:261-296) -> fake_poweriteration_group
 * (:69-98) / key_compression (modeling_llamagear.py:23-38):  per-channel outlier selection over the T tokens, mean fill,
 * group quantization along T, bit-packing into the channel-major K^T payload layout, the fp16 error and its rank-r
 * power iteration -- in four launches (select, fused quantize + pack + Gram on the matrix cores, per-head solve, Q pass)
 * that read x three times and never write the error matrix: the Q pass rebuilds E = x - dequant from x, the stored codes /
 * scale / mn and the outlier bitmap (the round-1 chain moved 12.6 bytes per element through HBM, this 6.9).
 *   x     fp16 [BH, T, 128] token-major                       T % 64 == 0, 64 <= T <= 16384; group in {32, 64}; bits in {2, 4}
 *   code  int32 [BH, 128, ldc]   words  t_off / fpi ..  of every channel row are written (ldc = row pitch in words)
 *   scale, mn [BH, 128, lds]     fp16 (mode 0) / float (mode 1), groups t_off / group ..
 *   k     outliers per side per channel row (0: none).  oidx / oval uint16 / fp16 [BH, 128, 2, kcap]: side slot 0 = the k
 *         smallest, slot 1 = the k largest, each ascending by token, written at list position o_off ..; the stored token
 *         index is t_off + t.  Ties at the selection boundary: lower token first (as gear_compress_rows).
 *   rank  0: no low-rank part.  P0 float [BH, 128, rank]; P_out fp16, head bh at element offset
 *         (bh / p_inner) * p_outer_stride + (bh % p_inner) * 128 * rank;  Q_out fp16 [BH, q_tcap, rank], rows q_toff ..
 *   variant: bit 0 = element-by-element tile arithmetic, bit 1 = always the exact slow selection, bit 2 = 16-bit LDS reads
 *         instead of the transposing ones for the matrix-core operands (cross-checks); measurement hooks (bench.py times
 *         the kernels of the chain one by one with them): bit 3 = reuse the selection a previous identical call left in the
 *         workspace, bit 4 = return after the fused quantize + Gram kernel, bit 5 = return after the selection
 *   workspace: gear_compress_key_fused_workspace(BH, T, k, rank) bytes of device scratch.
 * With t_off / ldc / lds / q_tcap / kcap the call appends a block to a pre-allocated streaming cache in place.
 */
size_t gear_compress_key_fused_workspace(int64_t BH, int T, int k, int rank);
int gear_compress_key_fused(const void* x, int64_t BH, int T, int group, int bits, int mode, int k, void* code, void* scale,
                            void* mn, int64_t ldc, int64_t lds, int t_off, int rank, int loop, const void* P0, void* P_out,
                            int64_t p_inner, int64_t p_outer_stride, void* Q_out, int q_tcap, int q_toff, void* oidx,
                            void* oval, int kcap, int o_off, int variant, void* workspace, size_t workspace_bytes,
                            void* stream);

/* V side, written in place (the streaming cache): V [B, H, T, 128] token-major -> payload rows t_off .. t_off + T of tensors
 * with `tcap` token rows per head: code int32 [B, H, tcap, 128/fpi], scale / mn [B, H, tcap, 128/group], sparse lists
 * oidx / oval [B, tcap, 2k] (gears_tokenQ semantics: a row = one token across the H heads,
 * GenerationBench/.../Simulated/compress_function.py:297-333), factors as gear_compress_key_fused.  The row compressor with an
 * output geometry, then the Gram-matrix power iteration and the Q pass. */
size_t gear_compress_value_fused_workspace(int64_t B, int H, int T, int rank);
int gear_compress_value_fused(const void* x, int64_t B, int H, int T, int group, int bits, int mode, int k, void* code,
                              void* scale, void* mn, int tcap, int t_off, int rank, int loop, const void* P0, void* P_out,
                              int64_t p_inner, int64_t p_outer_stride, void* Q_out, int q_tcap, int q_toff, void* oidx,
                              void* oval, void* workspace, size_t workspace_bytes, void* stream);

/* ---- head-sharded V rows: the outlier selection of a row that spans the heads of several GPUs ------------------------------
 * gears_tokenQ selects the k smallest and k largest values of a token's row over ALL heads (compress_function.py:297-333, two
 * torch.topk over H * D columns) and fills them with the row mean.  With the heads spread over `world` ranks:
 *   gear_vsel_candidates   this rank's part of every row (same row geometry as gear_compress_rows: n_rows rows of nseg segments of
 *                          seglen contiguous fp16 elements) -> cand uint32 [n_rows][2k + 2]: per side its k best elements as global
 *                          composites (16-bit order key << 16 | 0xFFFF - global column: ties -> lower column), then the exact
 *                          fp64 sum of the local part (bit pattern, two words).  col0 = global column of the rank's first element;
 *                          a full row has at most 65535 elements.
 *   (the ranks all-gather cand -> cand_all [world][n_rows][2k + 2]; 4 (2k + 2) bytes per row and rank)
 *   gear_vsel_thresholds   -> thr uint32 [n_rows][2] (the k-th largest composite, large side then small side), fill float [n_rows]
 *                          (mean of the full row of row_len_total elements; mode 0: rounded to fp16 like the row kernels' fill)
 *   gear_compress_value_sharded = gear_compress_value_fused with that selection instead of its own: a rank stores the outliers
 *                          that fall into its heads (0 .. k per side and row; unused list slots: index 0xFFFF, value 0).
 * The concatenated shard payloads are the unsharded payload (mode 0: bit for bit; mode 1: the unsharded kernel's fill is an fp32
 * tree sum, here it is the correctly rounded mean -- the last place of the fill can differ, the selection cannot).
 * The reference has no multi-GPU path; this keeps its single-GPU semantics under head sharding.
 * Range of the "shard payloads == unsharded payload, bit for bit" guarantee: |x| < 2^12 -- there the fp64 row sums behind the fill
 * value are exact and hence independent of how the row is split; larger magnitudes can move a fill by one unit in the last place.
 */
int gear_vsel_candidates(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride, int nseg,
                         int seglen, int64_t seg_stride, int k, int col0, void* cand, void* stream);
int gear_vsel_thresholds(const void* cand_all, int world, int64_t n_rows, int k, int64_t row_len_total, int mode, void* thr,
                         void* fill, void* stream);
int gear_compress_value_sharded(const void* x, int64_t B, int H, int T, int group, int bits, int mode, int k, void* code, void* scale,
                                void* mn, int tcap, int t_off, int rank, int loop, const void* P0, void* P_out, int64_t p_inner,
                                int64_t p_outer_stride, void* Q_out, int q_tcap, int q_toff, void* oidx, void* oval, int col0,
                                const void* thr, const void* fill, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a12 (KCVT variants): ONE quantization group per row -------------------------------------------------------------
 * Replaces fake_groupwise_channel_asymmetric_quantization_new(key, bits, seq_len) and
 * fake_groupwise_token_asymmetric_quantization(value, bits, num_head * sep_dim) of the KCVT / GEAR-KCVT / GEARL-KCVT branches
 * (GenerationBench/.../Simulated/compress_function.py:441-452, :496-525, :555-582), optionally around the row's sparse outliers
 * (gears_channelQ / gears_tokenQ with that group size).  Geometry as gear_compress_rows; len = nseg * seglen <= 16384, any
 * length.  oidx: uint16 [n_rows, 2k] as gear_compress_rows writes it (k == 0: none).
 *   y   fp16, same geometry: quantize -> dequantize, outliers keep their original value
 *   err optional fp16 x - y (0 at the outliers)
 */
int gear_quant_rows_whole(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride, int nseg,
                          int seglen, int64_t seg_stride, int bits, int mode, const void* oidx, int k, void* y, void* err,
                          void* stream);

/* ---- a11 / a12 with a sequence length that is not a multiple of the group ------------------------------------------------
 * gears_channelQ / gears_tokenQ (GenerationBench/.../Simulated/compress_function.py:261-333) on rows of ANY length: the k
 * smallest / k largest of the WHOLE row are selected inside the kernel (ties: lower index first) and replaced by the whole
 * row's mean for the quantization; fake_groupwise_channel_asymmetric_quantization_cluster then quantizes the first
 * floor(len / group) * group elements in groups of `group` consecutive elements and leaves the tail untouched (:107-122).
 * Geometry as gear_compress_rows; len = nseg * seglen <= 16384, at most 2048 groups per row, 2k <= len.
 *   y   fp16, same geometry: quantize -> dequantize of the prefix, outliers and tail keep their original value
 *   err optional fp16 x - y (0 at the outliers and in the tail)
 */
int gear_quant_rows_ragged(const void* x, int64_t n_rows, int rows_inner, int64_t outer_stride, int64_t inner_stride, int nseg,
                           int seglen, int64_t seg_stride, int group, int bits, int mode, int k, void* y, void* err,
                           void* stream);

/* ---- a4 / a10: low-rank power iteration ---------------------------------------------------------------------
 * Replaces headwise_lrap (cuda_supported_gear/quant/new_pack.py:291-311) and fake_poweriteration_group
 * (compress_function.py:69-98).  for i < loop: [last: P = orth(P)] Q = E P [last: Q = orth(Q)] P = E^T Q.
 *   E  : [bh, S, Dm] (transposed == 0) or its transpose [bh, Dm, S] (transposed == 1); fp16 or float32
 *   P0 : float32 [bh, Dm, r] initial basis (the reference draws it with torch.rand on the CPU generator; the
 *        caller supplies it so that runs are reproducible);   P_out [bh, Dm, r], Q_out [bh, S, r] fp16 / float32.
 *   1 <= r <= 16.  Per-(batch, head) bases (reference defect B3 not reproduced).
 */
size_t gear_lowrank_workspace(int64_t bh, int S, int Dm, int r);
int gear_lowrank(const void* E, int e_dtype, int transposed, int64_t bh, int S, int Dm, int r, int loop,
                 const void* P0, void* P_out, void* Q_out, int out_dtype, void* workspace, size_t workspace_bytes,
                 void* stream);

/* ---- decompress to fp16: dequant + low-rank + sparse restore ------------------------------------------------
 * out = fp16(fp16(dequant) + Q P^T), outliers restored from their stored fp16 value (compress_function.py:204-220
 * assembles the simulated result the same way).  Geometry as gear_compress_rows.
 *   kind 0: V rows (row = (b, t), rows_inner = T, nseg = H, seglen = D)   kind 1: K^T rows (row = (bh, d),
 *   rows_inner = D, one segment of T).  P fp16 [BH, D, r], Q fp16 [BH, T, r] (r == 0: none).
 * Limits (GEAR_ERR_ARG otherwise): the rows of one outer index must span fewer than 2^31 elements (the kernel adds 32-bit lane
 * offsets to a 64-bit base per workgroup), strides multiples of the group size.
 */
int gear_decompress_rows(const void* code, const void* scale, const void* mn, int64_t n_rows, int rows_inner,
                         int64_t outer_stride, int64_t inner_stride, int nseg, int seglen, int64_t seg_stride, int group,
                         int bits, int mode, int kind, const void* P, const void* Q, int r, int T, int D,
                         const void* oidx, const void* oval, int k, void* out, void* stream);

/* ---- batched fp16 transpose [bh, R, C] -> [bh, C, R] ------------------------------------------------------------
 * Replaces the caller-side key_states.transpose(2, 3).contiguous() of the attention hook
 * (cuda_supported_gear/modeling_llamagear.py:268, :403).  R, C multiples of 8.
 */
int gear_transpose_f16(const void* x, int64_t bh, int R, int C, void* y, void* stream);

/* ---- a6 + a7 (+ outliers): single-token decode attention over the compressed cache ---------------------------
 * One pass over the packed bytes replaces cuda_bmm_fA_qB_outer for K and V (cuda_supported_gear/quant/matmul.py:178),
 * the low-rank bmm chains of matmul_withlrap (modeling_llamagear.py:64-108), the fp32 softmax (:313) and the merge with
 * the fp16 residual window (:256-261, :329-333); also applies the sparse outlier term (not stored by the reference's
 * fused path).  head_dim must be 128.
 *   q      fp16 [B, Hq, 128]
 *   K payload (channel-major, as gear_compress_rows produces for K^T): kcode int32 [B*Hkv, 128, ldk], kscale / kmn
 *     [B*Hkv, 128, lsk], kP fp16 [B*Hkv, 128, rk] (channel side), kQ fp16 [B*Hkv, tf_k, rk] (token side),
 *     koidx / koval uint16 / fp16 [B*Hkv, 128, 2*kk] (token index, each half ascending) -- factors / outliers may be NULL
 *   V payload (token-major): vcode int32 [B*Hkv, tcap_v, 128/fpi], vscale / vmn [B*Hkv, tcap_v, 128/group], vP fp16
 *     [B*Hkv, 128, rv], vQ fp16 [B*Hkv, tf_v, rv], voidx / voval [B, tcap_v, 2*kv] (column Hkv-head*128 + d)
 *   kwin / vwin fp16 [B*Hkv, W, 128]: the W <= 128 most recent, still uncompressed tokens (NULL when W == 0)
 *   T compressed tokens (multiple of fpi); scores are scaled by qscale (1/sqrt(128)); softmax in fp32.
 *   out fp16 [B, Hq, 128]; lse optional float [B, Hq] (log-sum-exp of the scaled scores).
 */
size_t gear_attn_decode_workspace(int B, int Hq, int T, int bits);
int gear_attn_decode(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP, const void* kQ,
                     const void* koidx, const void* koval, const void* vcode, const void* vscale, const void* vmn,
                     const void* vP, const void* vQ, const void* voidx, const void* voval, const void* kwin,
                     const void* vwin, int B, int Hq, int Hkv, int D, int T, int W, int ldk, int lsk, int tcap_v, int tf_k,
                     int tf_v, int group, int bits, int mode, int rk, int rv, int kk, int kv, float qscale, void* out,
                     void* lse, void* workspace, size_t workspace_bytes, void* stream);

/* ---- the uncompressed baseline: single-token attention over an fp16 cache --------------------------------------
 * What the reference's harness runs as model "None" beside gearl / KIVI (cuda_supported_gear/test.py:41-62: HF attention
 * over the fp16 past_key_values) -- same flash-decoding split, same merge kernel and same grouped-query mapping as
 * gear_attn_decode, so that bench.py can put a compressed and an uncompressed time per token side by side.
 *   q fp16 [B, Hq, 128]; k, v fp16 [B, Hkv, tcap, 128], tokens [0, T) valid, 0 < T <= min(tcap, 8320); qscale = 1/sqrt(128)
 *   out fp16 [B, Hq, 128]; lse optional float [B, Hq]; workspace: gear_attn_decode_workspace(B, Hq, T, any bits).
 */
int gear_attn_decode_f16(const void* q, const void* k, const void* v, int B, int Hq, int Hkv, int D, int T, int tcap,
                         float qscale, void* out, void* lse, void* workspace, size_t workspace_bytes, void* stream);

/* ---- decode attention with per-segment low-rank factors (the streaming cache of the attention hook) ----------
 * As gear_attn_decode, but the channel-side factors kP / vP are [nseg, B*Hkv, 128, r]: tokens [0, seg0) use set 0 (the
 * prefill block, cuda_supported_gear/modeling_llamagear.py:402-434), every following `seglen` tokens their own set (the
 * 64-token decode blocks stacked on a leading buffer dim in the reference, :276-286).  seglen == 0: one segment.
 * wcap: row pitch (tokens) of kwin / vwin, >= W (a pre-allocated window buffer).
 */
int gear_attn_decode_seg(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP,
                         const void* kQ, const void* koidx, const void* koval, const void* vcode, const void* vscale,
                         const void* vmn, const void* vP, const void* vQ, const void* voidx, const void* voval,
                         const void* kwin, const void* vwin, int B, int Hq, int Hkv, int D, int T, int W, int ldk, int lsk,
                         int tcap_v, int tf_k, int tf_v, int group, int bits, int mode, int rk, int rv, int kk, int kv,
                         int seg0, int seglen, int wcap, float qscale, void* out, void* lse, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- per-token glue of a decode step (one launch each) ---------------------------------------------------------
 * gear_rope_append: qkv fp16 [B, (Hq + 2 Hkv) * 128] (fused q/k/v projection of the new token) -> RoPE at `pos` on q and k
 *   (modeling_llamagear.py:203-205), q_out fp16 [B, Hq, 128], k / v written to slot `slot` of the fp16 residual window
 *   kwin / vwin [B*Hkv, W, 128] (the torch.cat of :256, :316).
 * gear_add_rmsnorm: res_out = res_in + delta (delta may be NULL), y = weight * rmsnorm(res_out)   (LlamaRMSNorm + residual)
 * gear_silu_mul: out = silu(gate) * up for gate_up = [B, 2 I]                                         (LlamaMLP)
 */
int gear_rope_append(const void* qkv, int B, int Hq, int Hkv, int D, int pos, float theta, void* q_out, void* kwin,
                     void* vwin, int slot, int W, void* stream);
int gear_add_rmsnorm(const void* res_in, const void* delta, const void* weight, int64_t rows, int H, float eps,
                     void* res_out, void* y, void* stream);
int gear_silu_mul(const void* gate_up, int64_t B, int I, void* out, void* stream);

/* ---- device-side decode state (hipGraph-replayable token step) ---------------------------------------------------
 * state = int32[4] {pos, slot, T, W} in device memory: position of the token being processed, its slot in the fp16 window,
 * number of compressed tokens, number of window tokens INCLUDING that token (W = slot + 1).  The _dyn entry points read the per-token scalars from
 * it so that a captured graph of the whole token step can be replayed unchanged; gear_decode_state_advance() moves it on
 * by one token (pos, slot, W += 1).  gear_attn_decode_dyn plans its grid for `T` = the cache capacity; chunks beyond the
 * current length exit at once.  With dyn_state == NULL they behave exactly like the static entry points.
 */
int gear_attn_decode_dyn(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP,
                         const void* kQ, const void* koidx, const void* koval, const void* vcode, const void* vscale,
                         const void* vmn, const void* vP, const void* vQ, const void* voidx, const void* voval,
                         const void* kwin, const void* vwin, int B, int Hq, int Hkv, int D, int T, int W, int ldk, int lsk,
                         int tcap_v, int tf_k, int tf_v, int group, int bits, int mode, int rk, int rv, int kk, int kv,
                         int seg0, int seglen, int wcap, const void* dyn_state, float qscale, void* out, void* lse,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Chunk index of sorted outlier lists (the build's own addition to the sparse payload; the reference's fused path stores no
 * outliers at all, modeling_llamagear.py cache slots 11-12 / 15-16 are None).  oidx: uint16 [n_lists][k], every list
 * ascending; out: uint8 [n_lists][n_bounds], out[l][b] = first position of list l whose index is >= b * step.
 *   K payload: lists = (b, h, d, side), step 128 (tokens), n_bounds = ceil(T / 128) + 1
 *   V payload: lists = (b, t, side),    step 128 (columns = one KV head), n_bounds = Hkv + 1
 * gear_attn_decode_idx = gear_attn_decode with the two index tables (either may be NULL): a 128-token chunk then finds its
 * outliers with two byte loads per list instead of a binary search (contexts <= 8192 tokens; ignored otherwise). */
int gear_outlier_chunk_index(const void* oidx, int64_t n_lists, int k, int step, int n_bounds, void* out, void* stream);
int gear_attn_decode_idx(const void* q, const void* kcode, const void* kscale, const void* kmn, const void* kP, const void* kQ,
                         const void* koidx, const void* koval, const void* kochunk, const void* vcode, const void* vscale,
                         const void* vmn, const void* vP, const void* vQ, const void* voidx, const void* voval,
                         const void* vochunk, const void* kwin, const void* vwin, int B, int Hq, int Hkv, int D, int T, int W,
                         int ldk, int lsk, int tcap_v, int tf_k, int tf_v, int group, int bits, int mode, int rk, int rv,
                         int kk, int kv, float qscale, void* out, void* lse, void* workspace, size_t workspace_bytes,
                         void* stream);
/* ---- the streaming cache as one view --------------------------------------------------------------------------------
 * A pre-allocated cache of one layer (or of several layers riding in the batch dimension) that blocks are appended to in place
 * by gear_compress_key_fused / gear_compress_value_fused.  tcap = token capacity of every per-token tensor.
 *   K outlier lists koidx / koval [B*Hkv, 128, 2, kk_cap]: of every (channel, side) list the first kk0 + kkb * ((T - seg0) / seglen)
 *   entries are valid (kk0 from the prompt segment, kkb appended per block; later blocks hold later tokens, so the lists stay
 *   sorted); V lists voidx / voval [B, tcap, 2 kv].  Optional accelerators, all derived from the lists:
 *   kochunk uint8 [B*Hkv*128*2][nbk_pitch] / vochunk uint8 [B*tcap*2][Hkv+1]   chunk indices (gear_outlier_chunk_index_ex)
 *   ktile uint32 [B*Hkv][nck][ktile_cap] + kcnt int32 [B*Hkv][nck], vtile uint32 [B*Hkv][nblk][vtile_cap] + vcnt   sparse tiles
 *   (gear_cache_tiles_build): the outlier corrections value - dequant of a 128-token chunk / 64-token block as a flat list.
 */
typedef struct gear_cache_view {
    const void *kcode, *kscale, *kmn, *kP, *kQ, *koidx, *koval;
    const void *vcode, *vscale, *vmn, *vP, *vQ, *voidx, *voval;
    const void *kwin, *vwin;
    const void *kochunk, *vochunk;
    void *ktile, *kcnt, *vtile, *vcnt;
    int B, Hkv, D, tcap, ldk, lsk, group, bits, mode, rk, rv;
    int kk_cap, kk0, kkb, kv, seg0, seglen, wcap, nbk_pitch;
    int ktile_cap, nck, vtile_cap, nblk;
} gear_cache_view;

/* Single-token decode attention over a cache view: gear_attn_decode_dyn's arithmetic (segment factors, fp16 window of W <= 128
 * tokens with row pitch wcap, optional device-side {pos, slot, T, W}), outliers through the tiles when present (no search, no
 * dependent loads), else through the chunk indices, else by binary search in the lists. */
int gear_attn_decode_cache(const gear_cache_view* c, const void* q, int Hq, int T, int W, const void* dyn_state, float qscale,
                           void* out, void* lse, void* workspace, size_t workspace_bytes, void* stream);
/* (Re)build the sparse tiles of K chunks [k_chunk0, k_chunk1) (128 tokens each) and V blocks [v_blk0, v_blk1) (64 tokens each) of a
 * cache that holds T compressed tokens: called after the prompt and after every appended block (the chunk the block lies in). */
int gear_cache_tiles_build(const gear_cache_view* c, int T, int k_chunk0, int k_chunk1, int v_blk0, int v_blk1, void* stream);
/* ---- the decode-time block boundary in ONE launch -------------------------------------------------------------------------
 * Replaces, for the 64-token block the attention hook compresses whenever its fp16 window is full
 * (cuda_supported_gear/modeling_llamagear.py:265-286 key_compression of the block, :335-378 value_compression; with a sparsity
 * in the config also gears_channelQ / gears_tokenQ of GenerationBench/.../Simulated/compress_function.py:261-333 on the block),
 * the sequence gear_compress_key_fused + gear_compress_value_fused + gear_outlier_chunk_index_ex + gear_cache_tiles_build (~10
 * launches): one wave per (layer*batch, head, K | V) tile of 64 x 128 fp16 reads the window (c->kwin / c->vwin, row pitch wcap),
 * selects the outliers (K: kkb per side and channel over the 64 tokens; V: kv per side and token row ACROSS the heads, found by
 * "row duty" waves of the same launch and handed to the V tiles through sync_ws), quantizes (fp16-stepwise arithmetic, mode 0),
 * packs, keeps the error tile in LDS and runs the rank-r power iteration on it (token side: G' = E E^T is 64 x 64), then writes
 * codes / scale / mn / lists / chunk-index bytes / sparse tiles / factors at token offset t_off of the view's tensors in place.
 *   c       the cache view with B = layers * batch; mode must be 0, D 128, Hkv <= 64, kkb <= 16; the tensors behind the const
 *           pointers are WRITTEN (token rows t_off .. t_off + 63, K list positions o_off .. o_off + kkb - 1, tile of the chunk /
 *           block the tokens lie in)
 *   P0k / P0v float [B*Hkv, 128, rk | rv] initial bases; kP_out / vP_out fp16, head bh at element offset
 *           (bh / p_inner) * outer_stride + (bh % p_inner) * 128 * r (the per-segment factor tensors of the cache)
 *   sync_ws gear_compress_block_workspace(B, Hkv) bytes of device memory, ZEROED ONCE by the caller and then left alone (row
 *           flags carry a per-call tag); the word at gear_compress_block_status_ptr(sync_ws) becomes non-zero if a V tile ever
 *           gave up waiting for its rows (never on a healthy device; the call then does not hang, its V outliers are wrong).
 * Payload (codes, scale, mn, outlier sets and values, chunk index, tile contents) is bit-identical to the chain's; the factors
 * span the same subspaces (Q P^T within fp16 rounding of the chain's). */
size_t gear_compress_block_workspace(int64_t B, int Hkv);
int gear_compress_block(const gear_cache_view* c, int t_off, int o_off, int loop, const void* P0k, const void* P0v, void* kP_out,
                        void* vP_out, int64_t p_inner, int64_t kp_outer_stride, int64_t vp_outer_stride, void* sync_ws,
                        size_t sync_ws_bytes, void* stream);
const void* gear_compress_block_status_ptr(const void* sync_ws);

/* gear_compress_key_fused's single-read kernel (csrc/kone.hip; fp32 arithmetic, prompt-size tensors) exchanges outlier candidates
 * between the workgroups of a head inside ONE launch; every wait of that exchange is bounded.  Returns how many waits ran into their
 * bound since the library was loaded -- 0 on a healthy device; anything else means the payload of that call is invalid (the device
 * could not co-schedule a head's workgroups).  Replaces nothing in the reference (it has no in-kernel exchange).  Synchronises the
 * device.  -1: the query itself failed. */
int gear_kone_timeouts(void);
/* Heads of single-read-kernel calls whose outlier threshold guess failed and which the exact kernel chain redid (cumulative since the
 * library was loaded; 0 on ordinary data; results are exact either way).  Synchronises the device.  -1: the query failed. */
int gear_kone_fallback_heads(void);
/* Chunk index over lists that live inside larger tensors (the streaming cache keeps its tables up to date block by block):
 * lists (o, i), o < n_outer, i < inner, list id = o * outer_pitch + first + i; list `id` is oidx + id * list_stride with k valid
 * entries; out[id * out_pitch + b] = first position whose index is >= b * step, b < n_bounds.
 *   kochunk of a gear_cache_view: the prompt segment's entries of the K lists (k = kk0, list_stride = kk_cap, pitch nbk_pitch)
 *   vochunk: one row of H + 1 bounds per (token row, side) of the V lists. */
int gear_outlier_chunk_index_ex(const void* oidx, int64_t n_outer, int64_t inner, int64_t outer_pitch, int64_t first, int k,
                                int list_stride, int step, int n_bounds, void* out, int out_pitch, void* stream);
int gear_rope_append_dyn(const void* qkv, int B, int Hq, int Hkv, int D, const void* dyn_state, float theta, void* q_out,
                         void* kwin, void* vwin, int W, void* stream);
int gear_decode_state_advance(void* state, void* stream);

/* ---- fp16 GEMV for the decode token step: y[b, n] = sum_k x[b, k] W[n, k], 1 <= B <= 4, K % 8 == 0 -------------------
 * (torch.nn.functional.linear for one token; used by FastGearDecoder for the q/k/v, o, gate/up, down and lm_head GEMVs)
 */
int gear_gemv_f16(const void* x, const void* W, int B, int K, int N, void* y, void* stream);
/* y = res_in + W x (the residual-stream update after o_proj / down_proj); y may alias res_in, not x */
int gear_gemv_f16_add(const void* x, const void* W, int B, int K, int N, const void* res_in, void* y, void* stream);

/* Fused token-step projections (same 1 <= B <= 4, K % 8 == 0).  They fold the glue launches of a decoder layer
 * (cuda_supported_gear/modeling_llamagear.py:502-560: input_layernorm / post_attention_layernorm, :193-205 q/k/v + rotary,
 * LlamaMLP act_fn(gate) * up) into the weight stream:
 *   gear_gemv_f16_norm : v = x + delta (delta may be NULL; otherwise res_out [B,K] receives v and must not alias x/delta);
 *                        y = W . (norm_w * v) * rsqrt(mean(v^2) + eps)  (norm_w NULL: already folded into W's columns).   swiglu = 1: W rows are interleaved
 *                        (gate_0, up_0, gate_1, up_1, ...), swiglu = 2: blocked [N/2 gate rows | N/2 up rows] (gate_proj / up_proj
 *                        as the modules hold them); y [B, N/2] = fp16(silu(gate)) * up.
 *   gear_gemv_qkv_rope : the same prologue, W = [q heads | k heads | v heads] x 128 rows; RoPE (HF rotate_half, fp16 op by
 *                        op) on q and k at `pos`; q -> q_out [B,Hq,128]; k, v -> window slot `slot` of kwin / vwin
 *                        [B*Hkv, wcap, 128].  dyn_state != NULL: pos and slot are read from the device state
 *                        {pos, slot, T, W} (hipGraph replay).
 */
int gear_gemv_f16_norm(const void* x, const void* delta, const void* norm_w, float eps, const void* W, int B, int K, int N,
                       int swiglu, void* res_out, void* y, void* stream);
int gear_gemv_qkv_rope(const void* x, const void* delta, const void* norm_w, float eps, const void* Wqkv, int B, int K,
                       int Hq, int Hkv, int D, int pos, int slot, int wcap, float theta, const void* dyn_state,
                       void* res_out, void* q_out, void* kwin, void* vwin, void* stream);

/* ---- the exchange step of the head-sharded decode path (the reference has no distributed code) -------
 * With the KV heads split across ranks (one process per GPU) the only data any rank needs from the others is their slice of
 * the attention output in front of o_proj (cuda_supported_gear/modeling_llamagear.py:478-482 merges the heads).  Instead of a
 * collective call per layer per token, every rank owns an "exchange area" of uncached device memory that its peers map with
 * hipIpc (xGMI peer memory; the same calls work between two processes on one GPU, which is how the tests run it) and
 * gear_xchg_allgather is ONE single-workgroup launch that stores the local slice into every area, raises a flag, waits for
 * the peers' flags and copies the gathered rows out -- nothing in it needs the host, so it lives inside the hipGraph of the
 * token step.
 *   gear_xchg_bytes     size of an area for `world` ranks and slices of bytes_per_rank (0: world out of range, max 64)
 *   gear_xchg_alloc     uncached, zeroed, exportable; gear_xchg_free releases it (after the peers closed their mappings)
 *   gear_xchg_export    64-byte handle to send to the peers (any byte transport); gear_xchg_open maps a peer's area in this
 *                       process, gear_xchg_close unmaps it
 *   gear_xchg_allgather src [rows, row_bytes] of this rank -> out [rows, world, row_bytes] (rank r's bytes at slot r of every
 *                       row); peers = DEVICE array of `world` area pointers as mapped in this process (peers[rank] = the own
 *                       area); row_bytes % 16 == 0.  Every rank must issue the same sequence of calls.  A rank whose peers do
 *                       not show up within 3 s sets bit 0 of *status (device int32, may be NULL) and returns garbage rather
 *                       than hang. */
size_t gear_xchg_bytes(int world, size_t bytes_per_rank);
int gear_xchg_alloc(size_t bytes, void** ptr);
int gear_xchg_free(void* ptr);
int gear_xchg_export(const void* ptr, void* handle64);
int gear_xchg_open(const void* handle64, void** ptr);
int gear_xchg_close(void* ptr);
int gear_xchg_allgather(const void* src, int rows, size_t row_bytes, int world, int rank, const void* peers, void* out,
                        void* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEAR_HIP_H */
