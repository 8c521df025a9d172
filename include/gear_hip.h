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

#ifdef __cplusplus
}
#endif
#endif /* GEAR_HIP_H */
