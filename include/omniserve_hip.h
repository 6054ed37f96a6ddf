/*
 * omniserve_hip.h -- C ABI of libomniserve_hip.so, the MI355X (gfx950) implementation of the
 * OmniServe quantized-inference hot path (QServe W4A8KV4 / LServe kernels).
 *
 * This is the drop-in boundary.  Upstream exposes these operations as pybind11 torch-extension
 * modules `omniserve_backend.*` (kernels/setup.py:156-333); every entry point below cites the
 * upstream function it replaces.  All pointers are DEVICE pointers unless stated, tensors are
 * dense row-major with the shapes given, fp16 values are IEEE binary16 passed as `void*`,
 * `stream` is a hipStream_t (NULL = default stream).  Every call only enqueues work on `stream`
 * (graph-capture safe: no allocation, no synchronisation) and returns 0 on success or a negative
 * errno-style code (-22 invalid argument, -12 workspace too small, -5 launch failure).
 *
 * The host-side mirror of the upstream Python API (same module / function names and positional
 * arguments) lives in omniserve_amd/backend/ and binds this ABI through ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 */
#ifndef OMNISERVE_HIP_H
#define OMNISERVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version, bumped on any signature change. */
int omni_abi_version(void);

/* ----------------------------------------------------------------------------------------------
 * W4A8 / W8A8 GEMM         out[m,n] = fp16( epilogue( sum_k A[m,k] * W[n,k] ) )
 * -------------------------------------------------------------------------------------------- */

/* Bytes of int32 split-K scratch for an (M,N,K) problem.  omni_gemm_workspace_bytes: the plain entry points (which return
 * OMNI_ENOMEM when their plan splits K and the scratch is missing or smaller; 0 when no plan splits K).
 * omni_gemm_partial_workspace_bytes: the slab-only forms (omni_*_gemm_partial[_f16], M <= 512: at least one M x N int32 slab
 * even when K stays whole).  Both are the maximum over the three GEMM flavours and over every setting of
 * omni_gemm_set_midm_override (mode and forced split), so a buffer sized once stays large enough.  Callers keep one scratch
 * buffer per device/stream of at least this size (not persistent state: contents are dead after the call).
 * ABI 4 (omni_abi_version): ABI 3 had one query for both (which pinned an M x N slab for plain calls at M = 129..512). */
size_t omni_gemm_workspace_bytes(int M, int N, int K);
size_t omni_gemm_partial_workspace_bytes(int M, int N, int K);

/* Tuning / introspection hooks (tests and bench sweeps; not used by the serving path):
 * override the decode-shape plan (waves per workgroup: 1 or 4; K splits), 0/0 restores the
 * heuristic; query the plan the library would use. */
void omni_gemm_set_plan_override(int waves, int sk);
/* The mid-M kernel (M = 33 .. 128, csrc/qgemm_midm.h; the regime of gemm_cuda.cu:621-655's 32x64x128 / 64x64x64 tiles):
 * mode -1 = the planner's heuristic, 0 = never, 1 = wherever its shape conditions hold (N % 128 == 0, K % 256 == 0);
 * sk > 0 forces its K split.  Per calling thread, like the override above. */
void omni_gemm_set_midm_override(int mode, int sk);
void omni_gemm_get_plan(int M, int N, int K, int kalign, int* mb, int* waves, int* sk);
/* 1 when the row-kernel-free decode forms below (omni_*_gemm_silu for gate_up [2*inter, hidden], omni_*_gemm_partial_f16
 * for o [hidden, attn_dim] and down [hidden, inter]) accept a layer of these dimensions at M rows, 0 otherwise: the plan
 * conditions those entry points check at launch, for a driver that picks its fusion level up front.
 * mode: 0 per-channel W4A8, 1 per-group W4A8, 2 W8A8.  (Nothing upstream: fused extension) */
int omni_gemm_rowfree_ok(int M, int hidden, int attn_dim, int inter, int mode);

/* Replaces omniserve_backend.qgemm_w4a8_per_chn.gemm_forward_cuda
 *   (kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:601-657, kernel :308-599).
 * in_feats int8 [M,K]; qweight int8 [N,K/2] in the QServe 32x32-tile nibble layout
 * (omniserve/modeling/layers/quantized_linear/w4a8_linear.py:296-327), codes 0..15 used unsigned;
 * wscales,w_szs fp16 [N]; ascales,a_ssums fp16 [M]; out fp16, row m at out + m*out_row_stride
 * elements.  out = h( f32(acc)*wscales[n]*ascales[m] - w_szs[n]*a_ssums[m] ).
 * Requires N % 64 == 0, K % 64 == 0, M >= 1. */
int omni_w4a8_per_chn_gemm(const void* in_feats, const void* qweight, const void* wscales,
                           const void* ascales, const void* w_szs, const void* a_ssums,
                           void* out_feats, int M, int N, int K, int64_t out_row_stride,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Replaces omniserve_backend.qgemm_w4a8_per_group.gemm_forward_cuda
 *   (kernels/csrc/qgemm/w4a8_per_group/gemm_cuda.cu:635-707, dequant :276-331).
 * zeros, scales_i8: int8 [K/128, N] in the per-32-channel permuted order of
 * w4a8_linear.py:236-282.  w8 = int8( (u4 * s2 + z2) mod 256 ) with the reference's 32-bit-word
 * multiply; out = h( f32(acc) * (wscales[n]*ascales[m]) ).  Requires K % 128 == 0, N % 64 == 0. */
int omni_w4a8_per_group_gemm(const void* in_feats, const void* qweight, const void* zeros,
                             const void* scales_i8, const void* wscales, const void* ascales,
                             void* out_feats, int M, int N, int K, int64_t out_row_stride,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Replaces omniserve_backend.qgemm_w8a8.w8a8_gemm_forward_cuda
 *   (kernels/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu).  weight int8 [N,K] row-major.
 * out = h( f32(acc) * (wscales[n]*ascales[m]) ).  Requires N % 64 == 0, K % 64 == 0. */
int omni_w8a8_gemm(const void* in_feats, const void* weight, const void* wscales,
                   const void* ascales, void* out_feats, int M, int N, int K,
                   int64_t out_row_stride, void* workspace, size_t workspace_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Activation quantisation, norms, activation  (tokens x hidden, contiguous)
 * -------------------------------------------------------------------------------------------- */

/* Replaces omniserve_backend.fused_kernels.invoke_quant (tensor-scale overload)
 *   (kernels/csrc/fused_kernels.cu:57-94,235-250): per-token amax -> scale=h(amax/127),
 *   q = rni_sat_s8(x * (127/amax)). */
int omni_quant(void* out_i8, const void* in_f16, void* scale_f16, int tokens, int hidden,
               void* stream);

/* Replaces omniserve_backend.fused_kernels.invoke_quant_fuse_sum (tensor overload)
 *   (fused_kernels.cu:97-142,255-271): as omni_quant plus sum = h(sum_i x_i) (f32 tree). */
int omni_quant_fuse_sum(void* out_i8, const void* in_f16, void* sum_f16, void* scale_f16,
                        int tokens, int hidden, void* stream);

/* Replaces omniserve_backend.layernorm_ops.rms_norm (use_quant = false)
 *   (kernels/csrc/layernorm_kernels.cu:335-365,409-430). */
int omni_rms_norm(void* out_f16, const void* in_f16, const void* weight_f16, float eps,
                  int tokens, int hidden, void* stream);

/* Replaces omniserve_backend.layernorm_ops.rms_norm_general (use_per_token_quant = true)
 *   (layernorm_kernels.cu:58-191,432-469): y=(x-mean)*rsqrt(mean(x^2)+eps)*gamma, per-token
 *   int8 quantisation with scale=h(amax/127). */
int omni_rms_norm_general(void* out_i8, const void* in_f16, const void* weight_f16,
                          void* scale_f16, float eps, int tokens, int hidden, void* stream);

/* Replaces omniserve_backend.layernorm_ops.rms_norm_general_fuse_sum (per-token)
 *   (layernorm_kernels.cu:194-331,471-513): as above plus sum = h(sum_i h(y_i)) with the
 *   reference's fp16 per-thread accumulation. */
int omni_rms_norm_general_fuse_sum(void* out_i8, const void* in_f16, const void* weight_f16,
                                   void* sum_f16, void* scale_f16, float eps, int tokens,
                                   int hidden, void* stream);

/* Replaces omniserve_backend.activation_ops.silu_and_mul
 *   (kernels/csrc/activation_kernels.cu:10-30,84-97): in fp16 [tokens, 2d] -> out [tokens, d]. */
int omni_silu_and_mul(void* out_f16, const void* in_f16, int tokens, int d, void* stream);

/* bf16 / fp32 element types of the four functions above (csrc/row_dtypes.hip).  Upstream dispatches them over float, half
 * and bfloat16 (VLLM_DISPATCH_FLOATING_TYPES, kernels/csrc/dispatch_utils.h:7-14; launches fused_kernels.cu:212-272,
 * layernorm_kernels.cu:408-513, activation_kernels.cu:84-97); the entry points above are the `half` case.  dtype: 1 = bf16,
 * 2 = fp32 (0 = fp16 is rejected here: use the entry points above).  in / weight / out (rms_norm, silu_and_mul) are of that
 * type; scales and sums stay fp16 (`at::Half` upstream), quantised outputs int8.  sum_f16_or_null selects the _fuse_sum form.
 * Rounding points follow T as upstream: the general norm rounds y, its running maximum and its per-thread running sum to T. */
int omni_quant_dt(void* out_i8, const void* in, void* sum_f16_or_null, void* scale_f16, int tokens, int hidden, int dtype,
                  void* stream);
int omni_rms_norm_general_dt(void* out_i8, const void* in, const void* weight, void* sum_f16_or_null, void* scale_f16,
                             float eps, int tokens, int hidden, int dtype, void* stream);
int omni_rms_norm_dt(void* out, const void* in, const void* weight, float eps, int tokens, int hidden, int dtype, void* stream);
int omni_silu_and_mul_dt(void* out, const void* in, int tokens, int d, int dtype, void* stream);

/* ---- Overloads of the same three modules that the Llama W4A8 / W8A8 model code does not call --------
 * (csrc/offpath.hip).  `float scale` arguments bound to `at::Half` parameters upstream are rounded to fp16
 * inside.  Rows: hidden (d) % 8 == 0; the three norms additionally need hidden <= 16256 and, where the
 * reference launches min(hidden,1024) threads without rounding to 32, hidden % 32 == 0 below 1024. */

/* Replaces fused_kernels.invoke_quant(out, input, at::Half scale) and
 *   invoke_quant_fuse_sum(out, input, at::Half input_sum, at::Half scale) (its scalar input_sum is unused)
 *   (kernels/csrc/fused_kernels.cu:88-93,136-141,202-216,238-253): q = rni_sat(x / scale). */
int omni_quant_static(void* out_i8, const void* in_f16, float scale, int tokens, int hidden, void* stream);

/* Replaces fused_kernels.invoke_dequant (fused_kernels.cu:45-55,184-200): out = h(f32(acc) * scale);
 *   row strides in elements (in % 4 == 0, out % 8 == 0). */
int omni_dequant(void* out_f16, const void* in_i32, float scale, int tokens, int hidden, long long in_stride,
                 long long out_stride, void* stream);

/* Replaces fused_kernels.invoke_dequant_add_residual, both overloads (fused_kernels.cu:24-43,145-182):
 *   out = h(fma(f32(acc), s, f32(residual))), s = token_scale_f16[token] if given, else `scale`. */
int omni_dequant_add_residual(void* out_f16, const void* in_i32, const void* residual_f16,
                              const void* token_scale_f16, float scale, int tokens, int hidden, void* stream);

/* Replaces layernorm_ops.rms_norm(use_quant = true) (layernorm_kernels.cu:335-365,411-430):
 *   q = rni_sat((x * rsqrt(mean(x^2) + eps)) * w). */
int omni_rms_norm_quant(void* out_i8, const void* in_f16, const void* weight_f16, float eps, int tokens, int hidden,
                        void* stream);

/* Replaces layernorm_ops.rms_norm_general(use_per_token_quant = false) (layernorm_kernels.cu:58-196,455-466):
 *   y as in omni_rms_norm_general, q = rni_sat(f32(h(y)) * f32(scaling_f16[0])). */
int omni_rms_norm_general_static(void* out_i8, const void* in_f16, const void* weight_f16, const void* scaling_f16,
                                 float eps, int tokens, int hidden, void* stream);

/* Replaces layernorm_ops.invoke_dequant_add_residual_rms_norm_quant, both overloads
 *   (layernorm_kernels.cu:370-409,515-561): residual <- h(diff), diff = fma(f32(acc), s, f32(residual));
 *   q = rni_sat((f32(h(diff)) * rsqrt(mean(diff^2) + eps)) * gamma); s per token if token_scale_f16 is given. */
int omni_dequant_add_residual_rms_norm_quant(void* out_i8, const void* in_i32, void* residual_f16,
                                             const void* gamma_f16, const void* token_scale_f16, float scale,
                                             float eps, int tokens, int hidden, void* stream);

/* Replaces activation_ops.gelu_new (kind 0) / gelu_fast (kind 1) (activation_kernels.cu:186-213), fp16. */
int omni_gelu(void* out_f16, const void* in_f16, int kind, int tokens, int d, void* stream);

/* Replaces activation_ops.invoke_dequant_silu_and_mul_quant, both overloads (activation_kernels.cu:31-82,100-131):
 *   t = silu(f32(gate) * scale_gate) * (f32(up) * scale_up), in int32 [tokens, 2d].
 *   token_scale_f32 == NULL (and tmp_f32 == NULL): q = rni_sat(t / scale_out);
 *   otherwise tmp_f32 [tokens, d] = t, token_scale_f32[token] = amax/127, q = rni_sat((127/amax) * t). */
int omni_dequant_silu_and_mul_quant(void* out_i8, const void* in_i32, float scale_gate, float scale_up,
                                    float scale_out, void* token_scale_f32, void* tmp_f32, int tokens, int d,
                                    void* stream);

/* ---- Fused extensions (opt-in, NOT part of the reference API) --------------
 * Bit-identical to the two reference calls they replace; they exist because at MI355X speeds a
 * decode step is bound by the number of dependent kernels, not by bytes. */

/* In the four fused entry points below that take `sum_f16`, NULL means "no row sum": the W8A8 / per-group layers call
 * the non-summing reference functions (rms_norm_general, invoke_quant), and the sum costs a second ordered reduction. */

/* residual += delta (fp16 add, as the torch `x + proj` between the reference's calls,
 * llama_w4a8_unpad.py:425,437), then omni_rms_norm_general_fuse_sum on the updated residual. */
int omni_add_rms_norm_general_fuse_sum(void* out_i8, void* residual_f16, const void* delta_f16,
                                       const void* weight_f16, void* sum_f16, void* scale_f16,
                                       float eps, int tokens, int hidden, void* stream);

/* Deferred split-K epilogue for decode shapes (M <= 512; above 128 rows on the 128 x 256 tile with K slices over grid.y):
 * omni_w4a8_per_chn_gemm_partial writes only the
 * int32 partial sums slab[sk][M][N] (*sk_out = number of slabs, host int), and
 * omni_splitk_add_rms_norm_general_fuse_sum consumes them:
 *   residual += fp16( epilogue( sum_k slab[k] ) )   -- exactly omni_w4a8_per_chn_gemm's output, then
 *   omni_rms_norm_general_fuse_sum(residual).  ascales_in / a_ssums_in are the scales and sums of
 * the GEMM's int8 input (they must not alias the sum / scale outputs). */
int omni_w4a8_per_chn_gemm_partial(const void* in_feats, const void* qweight, void* slab_i32,
                                   size_t slab_bytes, int M, int N, int K, int* sk_out, void* stream);
int omni_splitk_add_rms_norm_general_fuse_sum(void* out_i8, void* residual_f16, const void* slab_i32, int sk,
                                              const void* wscales_f16, const void* ascales_in_f16,
                                              const void* w_szs_f16, const void* a_ssums_in_f16,
                                              const void* weight_f16, void* sum_f16, void* scale_f16,
                                              float eps, int tokens, int hidden, void* stream);

/* The same deferral for the W8A8 GEMM of the LServe models (llama_w8a8_unpad.py:382-427: o_proj / down_proj followed
 * by the residual add and rms_norm_general): slabs from omni_w8a8_gemm_partial, epilogue h(f32(acc) * (wscales[n] *
 * ascales[m])) applied by the consumer. */
int omni_w8a8_gemm_partial(const void* in_feats, const void* weight, void* slab_i32, size_t slab_bytes, int M, int N,
                           int K, int* sk_out, void* stream);
/* ... and for the per-group W4A8 GEMM (same epilogue formula as W8A8: w4a8_per_group/gemm_cuda.cu:620-627), so that the
 * g128 models get the deferred o_proj / down_proj epilogue as well */
int omni_w4a8_per_group_gemm_partial(const void* in_feats, const void* qweight, const void* zeros, const void* scales_i8,
                                     void* slab_i32, size_t slab_bytes, int M, int N, int K, int* sk_out, void* stream);
int omni_splitk_w8_add_rms_norm_general_fuse_sum(void* out_i8, void* residual_f16, const void* slab_i32, int sk,
                                                 const void* wscales_f16, const void* ascales_in_f16,
                                                 const void* weight_f16, void* sum_f16, void* scale_f16, float eps,
                                                 int tokens, int hidden, void* stream);
/* ... and for the LAST decoder layer: its down projection's slabs consumed by the model's final norm --
 * residual += fp16(epilogue(sum_k slab[k])), out_f16 = omni_rms_norm(residual) (llama_w4a8_unpad.py:484).  w_szs_f16 and
 * a_ssums_in_f16 both NULL: the W8A8 / per-group epilogue; both given: the per-channel W4A8 one. */
int omni_splitk_add_rms_norm(void* out_f16, void* residual_f16, const void* slab_i32, int sk, const void* wscales_f16,
                             const void* ascales_in_f16, const void* w_szs_f16, const void* a_ssums_in_f16,
                             const void* weight_f16, float eps, int tokens, int hidden, void* stream);

/* Decode attention with the flash-decoding merge fused into the following activation quantisation
 * (llama_w4a8_unpad.py:354): omni_kv4_decode_attention_partial = omni_kv4_decode_attention without its
 * merge step (*nsplit_out = S, partials stay in `workspace`: f32 [B,Hq,S,2] then f32 [B,Hq,S,128]);
 * omni_attn_merge_quant_fuse_sum = merge + omni_quant_fuse_sum of the [B, Hq*128] attention output. */
int omni_kv4_decode_attention_partial(const void* q_f16, const void* k_f16, const void* v_f16,
                                      int64_t q_stride, int64_t kv_stride, const void* kv_pointers_i64,
                                      const void* lengths_i32, int batch, int max_blocks, int num_heads,
                                      int num_kv_heads, int head_dim, int tokens_per_block, int max_context,
                                      const void* rope_cos_sin_f32, int rope_max_pos, void* workspace,
                                      size_t workspace_bytes, int* nsplit_out, void* stream);
int omni_attn_merge_quant_fuse_sum(void* out_i8, const void* part_ml_f32, const void* part_o_f32, int nsplit,
                                   void* sum_f16, void* scale_f16, int batch, int num_heads, void* stream);

/* ---- fused extension, round 3: the decode layer without its two quantiser row kernels ------------------------------
 * Upstream runs  gate_up GEMM -> silu_and_mul -> invoke_quant_fuse_sum -> down GEMM  and  attention ->
 * invoke_quant_fuse_sum -> o GEMM  (llama_w4a8_unpad.py:354-360,431-436; activation.py:54-64).  A per-token quantiser
 * needs the row maximum before any code can be produced, which is what forces a row kernel (one workgroup per token, a
 * pure latency chain at decode batch sizes) between two weight-streaming GEMVs.  Here the PRODUCER of the fp16
 * activation also raises the row maxima of |x| in `amax_slots_u32` -- uint32 [rows][8] holding f32 bit patterns,
 * raised with integer atomicMax (exact, order independent; the caller zeroes the buffer before the producer runs) --
 * and the CONSUMING projection quantises its own K-slices on the fly:
 *   omni_w4a8_per_{chn,group}_gemm_silu   gemm_forward_cuda + silu_and_mul: act fp16 [M, N/2] (gate = output channels
 *       [0, N/2), up = [N/2, N)), bit-identical to the two reference calls; raises amax_slots.  M <= 16.
 *   omni_attn_merge_f16_amax              the merge step of omni_kv4_decode_attention as a wide kernel: fp16 [B, Hq*128]
 *       (the values omni_kv4_decode_attention returns); raises amax_slots.  Carries an armed L2 prefetch.
 *   omni_w4a8_per_{chn,group}_gemm_partial_f16   omni_w4a8_per_*_gemm_partial on the codes
 *       invoke_quant[_fuse_sum](act) would produce -- code = rni_sat(x * (127 / amax)), fused_kernels.cu:126-131 --
 *       computed inside the GEMV; rider workgroups of the same launch replay the reference's ordered row sum
 *       (fused_kernels.cu:108-127) and write sum_f16 (NULL: not wanted) / scale_f16 = h(amax / 127) for the consumer of
 *       the slabs (omni_splitk_[w8_]add_rms_norm_general_fuse_sum).  M <= 16, K <= 16384.
 * Codes, scales, sums and slabs are bit-identical to the reference call sequence (tests/test_rowfree_gpu.py). */
int omni_w4a8_per_chn_gemm_silu(const void* in_feats, const void* qweight, const void* wscales, const void* ascales,
                                const void* w_szs, const void* a_ssums, void* act_f16, void* amax_slots_u32, int M, int N,
                                int K, void* stream);
int omni_w4a8_per_group_gemm_silu(const void* in_feats, const void* qweight, const void* zeros, const void* scales_i8,
                                  const void* wscales, const void* ascales, void* act_f16, void* amax_slots_u32, int M,
                                  int N, int K, void* stream);
int omni_w4a8_per_chn_gemm_partial_f16(const void* act_f16, const void* amax_slots_u32, const void* qweight,
                                       void* slab_i32, size_t slab_bytes, void* sum_f16, void* scale_f16, int M, int N,
                                       int K, int* sk_out, void* stream);
int omni_w4a8_per_group_gemm_partial_f16(const void* act_f16, const void* amax_slots_u32, const void* qweight,
                                         const void* zeros, const void* scales_i8, void* slab_i32, size_t slab_bytes,
                                         void* sum_f16, void* scale_f16, int M, int N, int K, int* sk_out, void* stream);
int omni_attn_merge_f16_amax(void* out_f16, const void* part_ml_f32, const void* part_o_f32, int nsplit,
                             void* amax_slots_u32, int batch, int num_heads, void* stream);
/* The W8A8 forms (LServe models: w8a8_gemm_forward_cuda -> silu_and_mul -> invoke_quant -> w8a8_gemm_forward_cuda,
 * llama_w8a8_unpad.py:94-111; attention -> invoke_quant -> o_proj, :329-335): no zero-point term, so no row sums --
 * the riders of omni_w8a8_gemm_partial_f16 write scale_f16 only (any K).  Bit-identical to the reference sequence. */
int omni_w8a8_gemm_silu(const void* in_feats, const void* weight, const void* wscales, const void* ascales,
                        void* act_f16, void* amax_slots_u32, int M, int N, int K, void* stream);
int omni_w8a8_gemm_partial_f16(const void* act_f16, const void* amax_slots_u32, const void* weight, void* slab_i32,
                               size_t slab_bytes, void* scale_f16, int M, int N, int K, int* sk_out, void* stream);

/* The same split for the LServe decode attention (retrieval / streaming heads, optional page list; KV4 pages when both
 * scale pointers are NULL, per-tensor KV8 pages otherwise): partials only, finished by omni_attn_merge_quant_fuse_sum. */
int omni_kv_decode_attention_fine_grained_partial(
    const void* q_f16, const void* k_f16, const void* v_f16, int64_t q_stride, int64_t kv_stride,
    const void* kv_scale_quant_orig_f32, const void* kv_scale_orig_quant_f32,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, const void* lengths_i32,
    const void* dynamic_sparse_page_idx_i32, int num_dynamic_pages, int tokens_per_sub_chunk, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int tokens_per_block, int sink_tokens, int local_tokens,
    int sink_blocks, int local_blocks, int max_context, const void* rope_cos_sin_f32, int rope_max_pos,
    void* workspace, size_t workspace_bytes, int* nsplit_out, void* stream);

/* Greedy-sampling helper of the decode runner (not a reference kernel: the reference's sampler is torch code,
 * omniserve/modeling/layers/sampler.py): out_i64[r] = index of the first maximum of logits fp16 [rows, cols]
 * (torch.argmax semantics).  workspace >= omni_argmax_workspace_bytes(rows). */
size_t omni_argmax_workspace_bytes(int rows);
int omni_argmax_f16(void* out_i64, const void* logits_f16, int64_t row_stride, int rows, int cols, void* workspace,
                    size_t workspace_bytes, void* stream);

/* Embedding lookup of the decode drivers (torch.nn.Embedding upstream): out fp16 [rows, cols] = table[idx[r], :],
 * idx int64 [rows] on the device; cols % 8 == 0; ids outside [0, table_rows) leave their row untouched. */
int omni_gather_rows_f16(void* out_f16, const void* table_f16, const void* idx_i64, int rows, int cols,
                         int64_t table_rows, void* stream);
/* First launch of a decode step in ONE kernel (decode drivers; not a reference kernel): the embedding lookup of
 * omni_gather_rows_f16 + lengths_i32[0 .. n_lengths) += 1 (the drivers' `lengths.add_(1)`) + zero_u32[0 .. zero_words) = 0 (the
 * step's row-maximum slots of the row-kernel-free layer).  n_lengths / zero_words may be 0 (then the pointer may be NULL). */
int omni_decode_step_begin(void* out_f16, const void* table_f16, const void* idx_i64, int rows, int cols,
                           int64_t table_rows, void* lengths_i32, int n_lengths, void* zero_u32, long long zero_words,
                           void* stream);

/* Fused extension: the NEXT decode-attention launch of the calling thread (omni_kv4_decode_attention[_partial],
 * omni_kv4_decode_attention_fine_grained, omni_kv8_decode_attention_per_tensor, omni_kv_decode_attention_fine_grained_partial;
 * one shot) takes the current token's q / k / v rows from the qkv projection's int32 split-K slabs
 * (omni_*_gemm_partial: slab [sk][M][N]) and applies the projection's epilogue in its first load trip -- the values are the
 * fp16 the GEMM would have stored, so attention output and appended cache rows are bit-identical; the slab epilogue launch
 * between the projection and the attention disappears.  col_q / col_k / col_v: first output channel of the three blocks of a
 * row (the reference's fused qkv layout: 0, Hq*128, (Hq+Hkv)*128).  w_szs / a_ssums: both given = per-channel W4A8 epilogue,
 * both NULL = W8A8 / per-group.  The q / k / v pointers of the launch are then only validated, not read.  slab NULL disarms. */
int omni_decode_arm_qkv_slabs(const void* slab_i32, int sk, int M, int N, int col_q, int col_k, int col_v,
                              const void* wscales_f16, const void* ascales_f16, const void* w_szs_f16,
                              const void* a_ssums_f16);

/* omni_silu_and_mul followed by omni_quant_fuse_sum without materialising the fp16 product
 * (activation.py:54-64 calls them back to back).  in fp16 [tokens, 2d] -> out int8 [tokens, d]. */
int omni_silu_mul_quant_fuse_sum(void* out_i8, const void* in_f16, void* sum_f16, void* scale_f16,
                                 int tokens, int d, void* stream);

/* ----------------------------------------------------------------------------------------------
 * KV4 paged cache (page = int4 data [H_kv][tpb][Dh/2] | fp16 scale [H_kv][tpb] | fp16 zero [H_kv][tpb],
 *   omniserve/worker/cache_engine.py:73-88, kernels/csrc/fused_attention/common/kvCacheUtils.h:53-164)
 * Block tables are int64 [B, 2, max_blocks] of raw device pointers (K row, then V row), exactly
 * what omniserve/utils/block_table_utils.py:62-93 builds.
 * RoPE uses a host-built table rope_cos_sin f32 [max_pos][Dh/2][2] = {cos, sin}(pos*scale/base^(2i/Dh)).
 * -------------------------------------------------------------------------------------------- */

/* Replaces omniserve_backend.fused_attention_*.compute_padding_offsets
 *   (kernels/csrc/fused_attention/common/input_metadata_helper.cu:16-50).
 * out[tok] = b*max_len - cu_seqlens[b]; cu_seqlens int32 [batch+1] (device). */
int omni_compute_padding_offsets(void* out_i32, const void* cu_seqlens_i32, int batch,
                                 int max_len, int total_tokens, void* stream);

/* Replaces omniserve_backend.fused_attention_fine_grained_dense.apply_bias_rope_update_kv_cache
 *   for the dense (all heads retrieval) KV4+zeros configuration
 *   (fine_grained_common/update_kv_cache.cu:27-124, applyBiasRopeUpdateKVCache.h:101-503).
 * qkv fp16 [tokens, (Hq+2Hkv)*Dh] unpadded; neox RoPE applied to q and k IN PLACE; post-RoPE k
 * and v quantised per (token, kv head) and written to the pages.  seq_lens int32 [B];
 * padding_offsets int32 [tokens]; max_seq_len = padded per-sequence length used by
 * padding_offsets.  rope_max_pos = rows in rope_cos_sin.  max_position_embeddings is the
 * reference's cyclic_kv_cache_len (update_kv_cache.cu:75): tokens below len - that are not stored. */
int omni_kv4_prefill_write(void* qkv_f16, const void* seq_lens_i32, const void* padding_offsets_i32,
                           const void* kv_pointers_i64, int tokens, int batch, int max_blocks,
                           int num_heads, int num_kv_heads, int head_dim, int max_seq_len,
                           int tokens_per_block, const void* rope_cos_sin_f32, int rope_max_pos,
                           int max_position_embeddings, void* stream);

/* Tuning hook: force the number of KV splits of omni_kv4_decode_attention (0 = heuristic). */
void omni_kv4_decode_set_split_override(int nsplit);

/* Scratch bytes for omni_kv4_decode_attention (split-KV partials). */
size_t omni_kv4_decode_workspace_bytes(int batch, int num_heads, int head_dim, int max_context);

/* Replaces omniserve_backend.fused_attention_pure_dense.single_query_attention (KV4 + zeros)
 *   (fused_attention_pure_dense/fused_attention.cpp:150-240; kernel
 *   decoderMaskedMultiheadAttentionTemplate.hpp:743-2222).
 * q fp16 [B,Hq,Dh] with row stride q_stride elements, k,v fp16 [B,Hkv,Dh] with row stride
 * kv_stride (strided views of the fused qkv buffer); lengths int32 [B] = context length
 * INCLUDING the current token (tlen = len-1 is the RoPE position and the append slot).
 * Applies neox RoPE to q and k, quantises + appends k,v of the current token to the pages,
 * attends over the cached history (fp16 dequant) plus the un-quantised current token, writes
 * out fp16 [B,Hq,Dh] contiguous.  max_context bounds lengths[] (sizes the KV split). */
int omni_kv4_decode_attention(void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16,
                              int64_t q_stride, int64_t kv_stride, const void* kv_pointers_i64,
                              const void* lengths_i32, int batch, int max_blocks, int num_heads,
                              int num_kv_heads, int head_dim, int tokens_per_block, int max_context,
                              const void* rope_cos_sin_f32, int rope_max_pos, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Prefill attention (varlen, causal; dense heads and token-streaming heads)
 * Replaces the un-vendored block_sparse_attn.flash_attn_varlen_func / token_streaming_attn_func the
 * reference calls for every prefill (omniserve/modeling/layers/ctx_attn/ctx_attn_func.py:39-45,68-73).
 * q fp16 [Lq, Hq, 128], k,v fp16 [Lk, Hkv, 128] with token strides q/k/v_stride (elements; heads contiguous),
 * out fp16 [Lq, Hq, 128] contiguous; cu_seqlens int32 [B+1]; head_mask_type int32 [Hq] (0 dense, <0
 * streaming) and streaming_info int32 [2*Hq] = (sink, local) per head, or both NULL for all-dense.
 * mask = causal (bottom-right aligned) AND (dense OR k_pos < sink OR q_pos - k_pos < local).
 * -------------------------------------------------------------------------------------------- */
int omni_prefill_attention(void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16,
                           int64_t q_stride, int64_t k_stride, int64_t v_stride,
                           const void* cu_seqlens_q_i32, const void* cu_seqlens_k_i32, int batch,
                           int max_seqlen_q, int num_heads, int num_kv_heads, int head_dim, int causal,
                           const void* head_mask_type_i32, const void* streaming_info_i32, void* stream);
/* Replaces block_sparse_attn.block_streaming_attn_func (imported at ctx_attn_func.py:3-7, wrapped at :47-59; no caller upstream):
 * omni_prefill_attention (causal) with streaming_info = (sink, local) per q head counted in blocks of 128 tokens --
 * mask = causal AND (dense OR k_pos / 128 < sink OR q_pos / 128 - k_pos / 128 < local).  Both tables are required. */
int omni_prefill_attention_block_streaming(void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16,
                                           int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                           const void* cu_seqlens_q_i32, const void* cu_seqlens_k_i32, int batch,
                                           int max_seqlen_q, int num_heads, int num_kv_heads, int head_dim,
                                           const void* head_mask_type_i32, const void* streaming_info_i32, void* stream);
/* Tuning / test hook: 0 = 16 query rows per wave (16x16x32 MFMA, default), 1 = 32 rows per wave (32x32x16 MFMA), 2 = the 16-row
 * form on the 8-wave ping-pong schedule (waves 4-7 one segment behind waves 0-3; same results within the attention tolerance). */
void omni_prefill_set_variant(int variant);

/* Tuning hook: over how many XCDs the query tiles of one kv head are spread when streaming heads are present
 * (head_mask_type != NULL): 1, 2, 4 or 8; 0 = the default (8).  Results do not depend on it. */
void omni_prefill_set_xcd_split(int w);

/* ----------------------------------------------------------------------------------------------
 * LServe dynamic sparsity: K statistics in the page tail and the page selector.
 * K page of a retrieval pool with H_r heads: int4 data | fp16 scale [H_r][tpb] | fp16 zero [H_r][tpb]
 *   | fp16 kmax [tpb/sub][H_r][128] | fp16 kmin [tpb/sub][H_r][128].
 * -------------------------------------------------------------------------------------------- */

/* Replaces omniserve_backend.fused_attention_ctx_pool.paged_min_max_pool
 *   (sparse_utils/ContextPool/context_pool_kernel.cu:145-213): per sequence, pooled head r
 *   (input head pooling_heads_idx[r]) and sub-chunk of `pooling_size` tokens, the elementwise max / min of
 *   the post-RoPE keys k fp16 [L, Hin, 128] are written into the K page tails.  kv_row_bytes = bytes of one
 *   token row of one head in the page (64: KV4, 128: per-tensor KV8) = the reference's
 *   size_per_retrieval_token / num_pool_heads; it locates the statistics behind data + 4 B/token-head tail. */
int omni_kv_min_max_pool(const void* k_f16, const void* kv_pointers_i64, const void* cu_seqlens_i32,
                         const void* pooling_heads_idx_i32, int batch, int max_blocks, int num_input_heads,
                         int num_pool_heads, int head_dim, int kv_row_bytes, int max_seqlen, int pooling_size,
                         int page_size, void* stream);

/* Replaces omniserve_backend.fused_attention_selector.single_query_page_selector
 *   (sparse_utils/KVPageSelector/fused_kv_page_selector.cpp:262-334): for retrieval heads,
 *   out[b,h,c] = fp16( sum_d max(q_d*kmax[c,d], q_d*kmin[c,d]) ) over the sub-chunks c of the history
 *   (q rotated at position lengths[b]-1); out fp16 [B,Hq,padded_sub_chunks] must be zero filled. */
int omni_kv_page_selector(void* out_f16, const void* q_f16, int64_t q_stride, const void* kv_pointers_i64,
                          const void* retrieval_head_flags_i32, const void* head_rank_table_i32,
                          const void* lengths_i32, int batch, int max_blocks, int num_heads, int num_kv_heads,
                          int num_retrieval_kv_heads, int head_dim, int kv_row_bytes, int tokens_per_block,
                          int tokens_per_sub_chunk, int padded_sub_chunks, const void* rope_cos_sin_f32,
                          int rope_max_pos, void* stream);

/* Page choice between the selector and the sparse decode attention (torch code upstream,
 *   omniserve/modeling/layers/decoding_attention.py:132-142): per (sequence, q head) row of selector scores
 *   (fp16, `subs_per_page` consecutive sub-chunk scores per page, rows `head_stride` elements apart)
 *   out_i32[row] = the k pages of [0, total_pages-1) with the largest max-over-sub-chunks score, sorted by
 *   descending score (ties: lower page first), followed by total_pages-1 (the newest page): k+1 entries per row.
 *   Limits: k <= 1024, total_pages <= 14337 (a 917 K-token history at 64 tokens per page). */
int omni_select_topk_pages(void* out_i32, const void* scores_f16, int64_t head_stride, int heads_total,
                           int subs_per_page, int total_pages, int k, void* stream);

/* ---- LServe fine-grained head classes ----------------------------------------
 * Every kv head is a retrieval head (retrieval_head_flags[h] != 0: whole history in the retrieval pool)
 * or a streaming head (sink + local window, kept in a ring of sink_blocks + local_blocks pages of the
 * streaming pool); head_rank_table[h] = index of the head inside its pool's pages, which hold
 * num_retrieval_kv_heads / num_streaming_kv_heads heads.  Page tables: retrieval_kv_pointers i64
 * [B,2,retrieval_blocks], streaming_kv_pointers i64 [B,2,streaming_blocks].
 *
 * omni_kv4_prefill_write_fine_grained replaces
 *   omniserve_backend.fused_attention_fine_grained_dense.apply_bias_rope_update_kv_cache with streaming
 *   heads (fine_grained_common/update_kv_cache.h:16-43, applyBiasRopeUpdateKVCache.h:296-311): a
 *   streaming head stores token pos only if pos < sink_tokens or pos >= len - local_tokens. */
int omni_kv4_prefill_write_fine_grained(
    void* qkv_f16, const void* seq_lens_i32, const void* padding_offsets_i32, const void* retrieval_kv_pointers_i64,
    const void* streaming_kv_pointers_i64, const void* retrieval_head_flags_i32, const void* head_rank_table_i32,
    int tokens, int batch, int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads,
    int num_retrieval_kv_heads, int num_streaming_kv_heads, int head_dim, int max_seq_len, int tokens_per_block,
    int sink_tokens, int local_tokens, int sink_blocks, int local_blocks, const void* rope_cos_sin_f32,
    int rope_max_pos, int max_position_embeddings, void* stream);

/* ---- tensor parallelism (SURVEY.md 8 row e; nothing upstream: the reference hard-codes tp_size = 1) ------------------
 * BASELINE configs[4] shards the row-parallel projections (o_proj, down_proj) over the GPUs of a node and sums their fp16
 * [tokens, hidden] outputs after each.  Besides torch.distributed's all-reduce (RCCL) the library has its own collective
 * over peer-mapped memory (csrc/tp_comm.h): every rank allocates ONE fine-grained buffer (omni_tp_alloc: two data slots
 * [+ a gather region for the two-shot form] followed by 64 flag words, zeroed), exports it (omni_tp_ipc_handle, 64 bytes, exchanged by the host over any
 * bootstrap channel) and maps the peers' (omni_tp_ipc_open).  A rank writes its partial projection into its own slot
 * (the projection's output pointer IS the slot), then one consumer launch both synchronises with the peers (flag words,
 * epochs in device memory: HIP-graph capturable, no host involvement) and reads their slots directly over xGMI:
 *   omni_tp_allreduce_f16                      out = sum over ranks (rank order, f32 accumulate, one rounding);
 *   omni_tp_add_rms_norm_general_fuse_sum      residual += that sum, then omni_rms_norm_general[_fuse_sum] (sum_f16 NULL:
 *                                              no row sum) -- all-reduce + residual add + norm + quant in one launch.
 * peer_data / peer_flags: HOST arrays of `world` device pointers (rank p's data / flag words as mapped into this
 * process; entry `rank` is the caller's own buffer).  slot_offset_elems: fp16 offset of the slot of THIS call; slots must
 * alternate from call to call (tp_comm.h explains why two are enough).  Every rank issues the same sequence of calls.
 * Two forms, bit-identical (same rank-order f32 sums, one rounding): ONE SHOT -- every rank reads every peer's whole slot,
 * (world - 1) x payload per rank over the fabric: right for small payloads; TWO SHOTS (ABI 4) -- rank r reduces chunk r
 * (1 / world of the vector; whole rows for the fused norm) into its GATHER region (gather_offset_elems: fp16 offset inside the
 * rank's data buffer, room for ceil(count / world) elements rounded up to 8 (to whole rows); < 0 = there is none), the ranks
 * meet a second time, then everyone reads the reduced chunks from their owners: 2 (world - 1) / world x payload per rank.
 * algo: 0 = two shots from 4 MiB of payload on more than two ranks (the second rendezvous costs ~16 us: csrc/tp_comm.hip has the
 * crossover arithmetic), 1 = one shot, 2 = two shots. */
int omni_tp_alloc(size_t bytes, void** ptr_out);
int omni_tp_free(void* ptr);
int omni_tp_ipc_handle(void* ptr, void* handle64);
int omni_tp_ipc_open(const void* handle64, void** ptr_out);
int omni_tp_ipc_close(void* ptr);
int omni_tp_allreduce_f16(void* out_f16, const void* const* peer_data, void* const* peer_flags, int rank, int world,
                          long long slot_offset_elems, long long count, long long gather_offset_elems, int algo, void* stream);
int omni_tp_add_rms_norm_general_fuse_sum(void* out_i8, void* residual_f16, const void* const* peer_data,
                                          void* const* peer_flags, int rank, int world, long long slot_offset_elems,
                                          const void* weight_f16, void* sum_f16, void* scale_f16, float eps, int tokens,
                                          int hidden, long long gather_offset_elems, int algo, void* stream);

/* ---- fused extension (nothing upstream): L2 weight prefetch --------------------------
 * The decode step alternates bandwidth-bound GEMVs with latency-bound row kernels (norm / quant, one workgroup per
 * token) during which HBM idles.  omni_prefetch_arm_gemm describes the NEXT decode-shape GEMM (M <= 128; mode 0 =
 * W4A8 per-channel, 1 = per-group, 2 = W8A8; deferred != 0 for omni_w4a8_per_chn_gemm_partial) and arms a one-shot
 * descriptor: the next decode-size (< 1024 tokens) launch of omni_quant[_fuse_sum], omni_rms_norm_general[_fuse_sum],
 * omni_add_rms_norm_general_fuse_sum, omni_silu_mul_quant_fuse_sum, omni_splitk_add_rms_norm_general_fuse_sum,
 * omni_attn_merge_quant_fuse_sum or omni_attn_merge_f16_amax carries `blocks` extra workgroups that pull up to budget_bytes of that GEMM's packed
 * weights (the head of every wave's weight stream, laid out by the GEMM's own plan) into the XCD-private L2s.
 * weight == NULL, blocks <= 0 or budget_bytes <= 0 disarms.  A pure performance hint: results never depend on it.
 * Load policy, per call (no process-wide switch): the decode-shape GEMM this thread enqueues next ON THAT WEIGHT TENSOR
 * reads its weights with plain loads (they sit in L2); every other decode-shape GEMM streams its weights with non-temporal
 * loads.  mode | 0x10: the gate_up form with the SiLU epilogue (omni_*_gemm_silu); mode | 0x20: that GEMM keeps its
 * non-temporal loads.  Evaluated at enqueue time (HIP-graph capturable); state is per enqueueing thread. */
int omni_prefetch_arm_gemm(const void* weight, int M, int N, int K, int mode, int deferred, int64_t budget_bytes,
                           int blocks);

/* Fused extension (round 6): a (residual add + rms_norm_general[_fuse_sum]) -> (decode-shape W4A8 GEMM) pair as ONE launch with
 * heterogeneous workgroups (csrc/norm_gemv_fused.h): the reference's input_layernorm -> qkv_proj and post_attention_layernorm ->
 * gate_up_proj (+ silu_and_mul) edges (omniserve/modeling/models/llama_w4a8_unpad.py:410-432).  Workgroups [0, M) are the rows
 * (the row kernels' own body); the others are the GEMM's 64-channel tiles, which request their WHOLE weight part first and then
 * wait for the rows on sync_u32[0] (agent-scope polls, bounded: a give-up stores NaN outputs and sets *err_u32 = 1).
 * Bit-identical to omni_splitk_[w8_]add_rms_norm_general_fuse_sum / omni_add_rms_norm_general_fuse_sum /
 * omni_rms_norm_general[_fuse_sum] followed by omni_w4a8_per_*_gemm / omni_w4a8_per_*_gemm_silu.
 *   rows' source: slab_i32 != NULL: residual += h(epilogue(sum of sk slabs)) (p_* = the producing GEMM's scales / zero terms and
 *     its input's scales / sums); else delta_f16 != NULL: residual += delta; else the residual as it is.
 *   amax_slots_u32 != NULL: the gate_up form (out_f16 = act [M, N/2], raises the row-maximum slots); NULL: out_f16 [M, N].
 *   sync_u32: 32 words the CALLER ZEROES before every launch (e.g. omni_decode_step_begin's zero list); err_u32: sticky.
 *   clk_u64: NULL, or [grid][8] wall-clock marks (timeline probe).  M <= 16, K % 256 == 0, K <= 4096, N % 64 (128) == 0.
 * omni_norm_gemm_fused_ok: 1 when the shape is accepted AND every workgroup of its grid is resident at once on this device
 * (the hand-off cannot deadlock whatever the dispatch order); 0: use the two launches. */
int omni_norm_gemm_fused_ok(int M, int N, int K, int mode, int silu);
int omni_w4a8_per_chn_norm_gemm_fused(void* codes_i8, void* residual_f16, const void* slab_i32, int sk, const void* delta_f16,
                                      const void* p_wscales_f16, const void* p_ascales_f16, const void* p_wszs_f16,
                                      const void* p_asums_f16, const void* gamma_f16, void* sum_f16, void* scale_f16, float eps,
                                      const void* qweight, const void* wscales_f16, const void* w_szs_f16, void* out_f16,
                                      long long out_row_stride, void* amax_slots_u32, void* sync_u32, void* err_u32, int M, int N,
                                      int K, void* clk_u64, void* stream);
int omni_w4a8_per_group_norm_gemm_fused(void* codes_i8, void* residual_f16, const void* slab_i32, int sk, const void* delta_f16,
                                        const void* p_wscales_f16, const void* p_ascales_f16, const void* gamma_f16,
                                        void* sum_f16, void* scale_f16, float eps, const void* qweight, const void* zeros_i8,
                                        const void* scales_i8, const void* wscales_f16, void* out_f16, long long out_row_stride,
                                        void* amax_slots_u32, void* sync_u32, void* err_u32, int M, int N, int K, void* clk_u64,
                                        void* stream);

/* Fused extension (round 4): omni_kv4_decode_attention_partial + omni_attn_merge_f16_amax as ONE launch.  Every split
 * workgroup writes its partial through, takes a ticket of its (sequence, head group), and the last arriver merges the splits
 * (the merge kernels' arithmetic), stores the fp16 [B, Hq * 128] output and raises the row maxima; an armed L2 prefetch rides
 * on extra z slices of the grid.  tickets_u32: >= batch * num_heads / 4 words, zero on entry and zero again when the launch has
 * finished (keep one buffer per stream).  batch <= 16, num_heads % 4 == 0.  With a single-split plan (short contexts) the two
 * launches run instead.  Bit-identical to the two-launch form. */
int omni_kv4_decode_attention_f16_amax(void* out_f16, void* amax_slots_u32, const void* q_f16, const void* k_f16,
                                       const void* v_f16, int64_t q_stride, int64_t kv_stride, const void* kv_pointers_i64,
                                       const void* lengths_i32, int batch, int max_blocks, int num_heads, int num_kv_heads,
                                       int head_dim, int tokens_per_block, int max_context, const void* rope_cos_sin_f32,
                                       int rope_max_pos, void* workspace, size_t workspace_bytes, void* tickets_u32,
                                       size_t tickets_words, void* stream);

/* omni_kv4_decode_attention_fine_grained replaces
 *   omniserve_backend.fused_attention_fine_grained_dense.single_query_attention
 *     (fused_attention_fine_grained/dense_attention/fused_attention.h:18-46) and, with
 *   dynamic_sparse_page_idx_i32 != NULL, omniserve_backend.fused_attention_fine_grained_sparse
 *     .single_query_attention (sparse_attention/fused_attention.h): retrieval q head h of sequence b attends
 *   only the num_dynamic_pages pages dynamic_sparse_page_idx[b,h,:] (the last one being the newest page) and
 *   the appended key is folded into the page's min/max statistics (tokens_per_sub_chunk).  Streaming heads
 *   attend min(sink+local-1, len-1) cached tokens through the ring.  Workspace as omni_kv4_decode_attention. */
int omni_kv4_decode_attention_fine_grained(
    void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16, int64_t q_stride, int64_t kv_stride,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, const void* lengths_i32,
    const void* dynamic_sparse_page_idx_i32, int num_dynamic_pages, int tokens_per_sub_chunk, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int tokens_per_block, int sink_tokens, int local_tokens,
    int sink_blocks, int local_blocks, int max_context, const void* rope_cos_sin_f32, int rope_max_pos,
    void* workspace, size_t workspace_bytes, void* stream);

/* ---- per-tensor KV8 (LServe's published `w8a8kv8 per_tensor` configuration) ------
 * K or V page of a pool with H heads: int8 data [H][tpb][128] | fp16 scale slot [H][tpb] | fp16 zero slot [H][tpb]
 *   (| K statistics as above).  Static scales, no zero points (omniserve/engine/arg_utils.py:500-503):
 *   code = cvt.rni.sat.s8(kv_scale_orig_quant[i] * x), value = fp16(kv_scale_quant_orig[i] * code), i = 0 (K), 1 (V)
 *   (common/decoderMaskedMultiheadAttentionUtils.h:2041-2048,2086-2093).  Both scale arrays are DEVICE fp32 [2],
 *   as the reference's Python wrappers pass them (decoding_attention.py:202-210, ctx_update_kv.py:59-67).
 *
 * omni_kv8_prefill_write_per_tensor replaces
 *   omniserve_backend.fused_attention_per_tensor_dense.apply_bias_rope_update_kv_cache
 *   (fused_attention_per_tensor/per_tensor_common/update_kv_cache.h:16-44; the writer also leaves each row's own
 *   fp16(absmax/127) in the scale slot, applyBiasRopeUpdateKVCache.h:387-414). */
int omni_kv8_prefill_write_per_tensor(
    void* qkv_f16, const void* kv_scale_orig_quant_f32, const void* seq_lens_i32, const void* padding_offsets_i32,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, int tokens, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int max_seq_len, int tokens_per_block, int sink_tokens,
    int local_tokens, int sink_blocks, int local_blocks, const void* rope_cos_sin_f32, int rope_max_pos,
    int max_position_embeddings, void* stream);

/* omni_kv8_decode_attention_per_tensor replaces
 *   omniserve_backend.fused_attention_per_tensor_dense.single_query_attention
 *     (fused_attention_per_tensor/dense_attention/fused_attention.h:18-46) and, with a page list,
 *   omniserve_backend.fused_attention_per_tensor_sparse.single_query_attention
 *     (sparse_attention/fused_attention.h:18-50).  Semantics of omni_kv4_decode_attention_fine_grained on
 *   KV8 pages; the appended token is stored with kv_scale_orig_quant and no tail is written
 *   (dense_attention/decoderMaskedMultiheadAttentionTemplate.hpp:1377,2162). */
int omni_kv8_decode_attention_per_tensor(
    void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16, int64_t q_stride, int64_t kv_stride,
    const void* kv_scale_quant_orig_f32, const void* kv_scale_orig_quant_f32,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, const void* lengths_i32,
    const void* dynamic_sparse_page_idx_i32, int num_dynamic_pages, int tokens_per_sub_chunk, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int tokens_per_block, int sink_tokens, int local_tokens,
    int sink_blocks, int local_blocks, int max_context, const void* rope_cos_sin_f32, int rope_max_pos,
    void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNISERVE_HIP_H */
