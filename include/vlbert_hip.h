/* libvlbert_hip.so -- C ABI of the MI355X (gfx950) VL-BERT pre-training hot path.
 *
 * The drop-in boundary: plain pointers + sizes + a hipStream_t, no torch / ATen
 * types.  The caller (a thin PyTorch-ROCm wrapper, `vl-bert_amd/_lib.py`) owns every buffer and
 * passes `tensor.data_ptr()` and `torch.cuda.current_stream().cuda_stream`; kernels are enqueued on
 * that stream, never synchronise, and keep no global state besides a thread-local error string
 * (re-entrant per stream, like the reference's ops which enqueue on at::cuda::getCurrentCUDAStream(),
 * common/lib/roi_pooling/cuda/ROIAlign_cuda.cu:273).
 *
 * Conventions
 *   - return 0 on success, <0 on error (VLB_ERR_ARG = -1 bad argument, VLB_ERR_HIP = -2 launch
 *     failure); `vlb_last_error()` returns the message (the reference raises RuntimeError through
 *     AT_ASSERTM / AT_ERROR / THCudaCheck, ROIAlign_cuda.cu:263-264,297 -- the Python wrapper
 *     re-raises RuntimeError with this text).
 *   - "bf16" buffers are raw bfloat16 bits (uint16_t); leading dimensions are in ELEMENTS.
 *   - dropout: `drop_p` in [0,1); `seed` is a DEVICE pointer to one uint32 (so a captured hipGraph
 *     re-reads it on replay); `tag` identifies the dropout site.  Forward and backward of a site
 *     must pass the same (seed value, tag).  drop_p == 0 disables dropout (seed may be NULL).
 *   - empty problems (rows == 0) return 0 immediately, as the reference does
 *     (ROIAlign_cuda.cu:278-281).
 *
 * Each entry point cites the reference code it replaces (paths relative to jackroos/VL-BERT).
 */
#ifndef VLBERT_HIP_H
#define VLBERT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vlb_stream_t; /* == hipStream_t */

const char* vlb_last_error(void);
int vlb_version(void);
/* 16-bit storage type of this build: 0 = bfloat16 (libvlbert_hip.so, the default), 1 = IEEE fp16 (libvlbert_hip_f16.so: every
 * "bf16" in the names below then means fp16 -- the reference's own mixed precision, Apex O2 with a static loss scale,
 * pretrain/function/train.py:345-352; the host scales the loss and passes 1/scale as grad_scale to the optimizer entry points). */
int vlb_act_dtype(void);
/* returns the CU count of `device` and copies its gcnArchName ("gfx950:...") into name */
int vlb_device_info(int device, char* name, int cap);

/* ---- GEMM -----------------------------------------------------------------------------------
 * C[M,N] (+)= A[M,K] * B[N,K]^T, bf16 operands, fp32 accumulate, fused epilogue
 *   v = acc (+ bias[n]) ; act: 0 none | 1 erf-GELU (pre-activation optionally stored to `pre`)
 *   | 2 ReLU | 3 v *= gelu'(aux[m,n]) | 4 erf-GELU with gelu'(v) stored to `pre` (the derivative shares the
 *   forward's exponential, so the backward epilogue becomes act 5) | 5 v *= aux[m,n] | 6 tanh (BertPooler,
 *   modeling.py:430-436) ; dropout(v) ; v += res[m,n] ;
 *   out_mode: 0 store bf16 | 1 store fp32 | 2 fp32 atomicAdd with split-K (`splitk` <= 0: auto) |
 *             3 fp32 accumulate C += (single K pass, no atomics).
 * K % 64 == 0; lda/ldb % 8 == 0; ldc/ldaux/ldpre/ldres % 4 == 0.
 * Replaces nn.Linear / torch.matmul (cuBLAS) + the separate bias/GELU/dropout/residual kernels of
 * external/pytorch_pretrained_bert/modeling.py:291-300 (Q,K,V), :330-333 (BertSelfOutput),
 * :362-364 (BertIntermediate + gelu :114-120), :375-378 (BertOutput), :448-452, :469
 * (MLM transform / tied decoder), common/visual_linguistic_bert.py:482-502 (MVRC head),
 * common/fast_rcnn.py:105-109 (obj_downsample Linear+ReLU) and their autograd backward. */
int vlb_gemm_nt_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                     const float* bias, int act, const void* aux, long ldaux, void* pre, long ldpre,
                     const void* res, long ldres, float drop_p, const uint32_t* seed, uint32_t tag,
                     int out_mode, int splitk, vlb_stream_t stream);
/* vlb_gemm_nt_bf16 with (a) the "LayerNorm residual": res = the fp16 PRE-LayerNorm rows of the sublayer whose output is the
 * residual, res_stats = [M][2] fp32 (mean, rstd) of that LayerNorm, res_gamma / res_beta = its parameters; the term added is
 * (res - mean) * rstd * gamma + beta evaluated in fp32 (needs act 0, out_mode 0); (b) out_f16 != 0: the 16-bit result is stored as
 * IEEE fp16.  With both, BertSelfOutput / BertOutput (modeling.py:329-333,374-378) keep their residual stream out of bf16. */
int vlb_gemm_nt_bf16_ex(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                        const float* bias, int act, const void* aux, long ldaux, void* pre, long ldpre,
                        const void* res, long ldres, const float* res_stats, const float* res_gamma, const float* res_beta,
                        int out_f16, float drop_p, const uint32_t* seed, uint32_t tag, int out_mode, int splitk,
                        vlb_stream_t stream);


/* bf16 C[M,N] = A[M,K] B[N,K]^T, no epilogue, for few output tiles and a very long K (tied-decoder dgrad at small batch):
 * slab split-K through `workspace` (vlb_wgrad_workspace_floats(M, N, K) floats) + a reduce that converts to bf16. */
int vlb_gemm_nt_bf16_splitk(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                            float* workspace, long workspace_floats, vlb_stream_t stream);
/* Run-time tuning knob of the GEMM dispatcher (the environment variables VLB_GEMM_P8* give the defaults): name = "p8_mode"
 * (0: 128x128 kernels only | 1: cost model | 4 / 5: force 256- / 320-row tiles), "p8_keepb", "p8_group", "p8_min_tiles",
 * "tn8_mode" (weight gradients: 0 = 128x128 TN kernel only, 1 = large-tile core where it applies); "ln_fwd_rows" (rows per wave of
 * vlb_layernorm_fwd: 1 / 2 / 4, 0 = by size), "ln_bwd4" (vlb_layernorm_bwd: 0 8-column kernel, 1 4-column, 2 two rows in flight). */
int vlb_gemm_set_option(const char* name, int value);



/* Weight-gradient form: C[M,N] (fp32) += A[M,K] B[N,K]^T for few output tiles and a very long K (K = padded
 * row count of the activations; autograd's `grad_output.t().mm(input)` behind every nn.Linear).  Split-K
 * through fp32 workspace slabs + a streaming reduce (no atomics); splits chosen by an internal cost model, capped
 * by the workspace the caller provides (vlb_wgrad_workspace_floats gives the amount the model would like). */
long vlb_wgrad_workspace_floats(int M, int N, int K);
int vlb_wgrad_nt_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                      float* workspace, long workspace_floats, vlb_stream_t stream);

/* Same product taken straight from the row-major tensors the forward/backward passes already hold:
 * C[Mo,No] (fp32) += A[R,Mo]^T * B[R,No] (reduction over the R rows; LDS transpose reads, no transposed copies);
 * colsum[Mo] += column sums of A when non-NULL (the bias gradient).  lda/ldb % 8 == 0, 16-byte aligned operands.
 * accumulate == 0 overwrites C instead (first micro-batch of an optimizer step: no zero fill, no read-modify-write). */
int vlb_wgrad_tn_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int R, int Mo, int No,
                      float* colsum, float* workspace, long workspace_floats, int accumulate, vlb_stream_t stream);

/* out[c][r] = in[r][c] (bf16, out leading dim ldo >= R); colsum[c] += sum_r in[r][c] if non-NULL
 * (bias gradients).  Feeds the weight-gradient GEMMs (autograd's `grad.t().mm(input)`). */
int vlb_transpose_bf16(const void* in, long ldi, void* out, long ldo, int R, int C, float* colsum, vlb_stream_t stream);
/* n transposes in one launch: desc (device) = n x {in ptr, ldi, out ptr, ldo, R, C} as int64; tile_start (device, n+1
 * int32) = running count of 64x64 tiles, tile_start[n] == total_tiles.  Used for the per-step refresh of the
 * transposed bf16 weight copies read by the dgrad GEMMs (autograd's `grad_output.mm(weight)`). */
int vlb_transpose_batched_bf16(const int64_t* desc, const int32_t* tile_start, int n, int total_tiles, vlb_stream_t stream);

/* ---- BertLayerNorm (modeling.py:222-235; eps inside the sqrt, biased variance) --------------
 * fwd: y = LN(x)*gamma+beta, stats[row] = (mean, rstd) (may be NULL).
 * bwd: dx (bf16), dx_drop = dropout-masked dx (bf16, gradient entering the preceding dense layer),
 *      dx_acc (fp32 atomicAdd) -- any may be NULL; dgamma/dbeta are accumulated.
 *      dy is bf16, or fp32 when dy_f32 != 0; H % 8 == 0, row strides % 8 == 0.
 *      x_f16 != 0: the rows of x are IEEE fp16 (the pre-LayerNorm sums written by vlb_gemm_nt_bf16_ex(out_f16)), else bf16.
 *      workspace: NULL (dgamma/dbeta by direct fp32 atomics) or vlb_layernorm_bwd_workspace_floats(H) floats of
 *      scratch (per-workgroup partial sums stored without atomics, column-summed by a 2nd kernel). */
/* Overflow guard of the fp16 residual stream: every vlb_layernorm_fwd raises a sticky per-device flag when a row's statistics are not
 * finite (an fp16 pre-LayerNorm sum beyond 65504 became inf in the producing GEMM's epilogue; bit 0 = fp16 rows, bit 1 = bf16 rows).
 * vlb_nonfinite_status returns the flag (>= 0; clears it when reset != 0), negative on error; it synchronises the device.  The
 * reference has no counterpart (fp32 activations; its fp16 path relies on apex loss scaling, common/trainer.py:119-127). */
int vlb_nonfinite_status(int reset);
int vlb_layernorm_fwd(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, float* stats,
                      int rows, int H, float eps, int x_f16, vlb_stream_t stream);
int vlb_layernorm_bwd(const void* dy, long lddy, int dy_f32, const void* x, long ldx, const float* stats,
                      const float* gamma, void* dx, long lddx, void* dx_drop, long lddd, float drop_p,
                      const uint32_t* seed, uint32_t tag, float* dx_acc, long ldacc, float* dgamma, float* dbeta,
                      float* workspace, int rows, int H, int x_f16, vlb_stream_t stream);
long vlb_layernorm_bwd_workspace_floats(int H);
/* vlb_layernorm_bwd with the dgamma / dbeta finalize DEFERRED: the per-workgroup partial vectors stay in `workspace` (required, and
 * not to be reused until consumed); vlb_ln_param_finalize_batch adds the column sums of up to 32 such workspaces (entry i: ws[i] with
 * nslab[i] = vlb_layernorm_bwd_slabs(rows of that call) > 0 partial vectors) into dgamma[i] / dbeta[i] in ONE launch.  A training
 * step runs 26 LayerNorm backwards whose parameter gradients are only read by the optimizer / a bucket's all-reduce.
 * vlb_layernorm_bwd_slabs == 0: the call was small enough to add its sums directly (nothing to finalize). */
int vlb_layernorm_bwd_deferred(const void* dy, long lddy, int dy_f32, const void* x, long ldx, const float* stats,
                               const float* gamma, void* dx, long lddx, void* dx_drop, long lddd, float drop_p,
                               const uint32_t* seed, uint32_t tag, float* dx_acc, long ldacc, float* dgamma, float* dbeta,
                               float* workspace, int rows, int H, int x_f16, vlb_stream_t stream);
int vlb_layernorm_bwd_slabs(int rows);
int vlb_ln_param_finalize_batch(int n, const float* const* ws, const int* nslab, float* const* dgamma, float* const* dbeta, int H,
                                vlb_stream_t stream);

/* ---- BertSelfAttention core (modeling.py:300-316) --------------------------------------------
 * qkv [B*S, 3H] bf16 (q | k | v, head h at columns h*64), mask [B,S] fp32 (1 attend / 0 -> -10000),
 * ctx [B*S, H] bf16, lse [B, heads, S] fp32 (row log-sum-exp, saved for backward).
 * S <= 128, head dim 64.  bwd recomputes the probabilities; writes dqkv [B*S, 3H] bf16. */
int vlb_attention_fwd(const void* qkv, const float* mask, void* ctx, float* lse, int B, int S, int H, int nh,
                      float drop_p, const uint32_t* seed, uint32_t tag, vlb_stream_t stream);
int vlb_attention_bwd(const void* qkv, const float* mask, const void* ctx, const float* lse, const void* dctx,
                      void* dqkv, int B, int S, int H, int nh, float drop_p, const uint32_t* seed, uint32_t tag,
                      vlb_stream_t stream);

/* ---- embedding side (common/visual_linguistic_bert.py:173-241, common/fast_rcnn.py:136-187) ---
 * vlb_seq_layout: masks (uint8 [B,T], [B,R]) -> code [B,S] (kind<<16 | src index; kind 0 pad,
 *   1 text, 2 object, 3 END), text_len[B], nobj[B], text_rows [B,T], obj_rows [B,R] (-1 if masked),
 *   attn_mask [B,S] fp32.  S >= T+R+1.  (replaces the .item()/.nonzero()/boolean-mask scatter
 *   host-synchronising code at visual_linguistic_bert.py:200-235.) */
int vlb_seq_layout(const uint8_t* text_mask, const uint8_t* obj_mask, int B, int T, int R, int S, int32_t* code,
                   int32_t* text_len, int32_t* nobj, int32_t* text_rows, int32_t* obj_rows, float* attn_mask,
                   vlb_stream_t stream);
/* boxes [B*R, ldbox] fp32 (x1,y1,x2,y2, 2048 features; x1 <= -1.5 marks padding), im_info [B, ldinfo >= 2] = (width, height, ...)
 * per image -- the reference's datasets emit 5 columns for pre-training / VCR (conceptual_captions.py:138, vcr.py:377) and 4 for VQA
 * (vqa/data/datasets/vqa.py:217), so the row stride is an argument; mvrc_ops int64 [B*R] (1 -> use mask_emb[2048] instead of the feature) -> out bf16 [B*R, 4096]
 * = dropout(coordinate_embeddings || feature)  (common/utils/bbox.py:33-65, fast_rcnn.py:165-175,
 * pretrain/modules/resnet_vlbert_for_pretraining.py:114-117). */
int vlb_obj_prep_fwd(const float* boxes, long ldbox, const float* im_info, long ldinfo, const int64_t* mvrc_ops,
                     const float* mask_emb, void* out, int B, int R, float drop_p, const uint32_t* seed, uint32_t tag,
                     vlb_stream_t stream);
/* x[r, :] = 0 where boxes[r*ldbox] <= -1.5 (padded box): the zero rows pad_sequence leaves in obj_reps (common/fast_rcnn.py:176-186) */
int vlb_zero_padded_rows_bf16(void* x, long ld, const float* boxes, long ldbox, int rows, int H, vlb_stream_t stream);
/* dst[c] += sum over rows with sel[row]==1 of src[row][c] * dropout_mask(row*row_elems + col_off + c) */
int vlb_masked_colsum(const void* src, long lds, const int64_t* sel, int rows, int C, float* dst, float drop_p,
                      const uint32_t* seed, uint32_t tag, uint32_t row_elems, uint32_t col_off, vlb_stream_t stream);
/* fused embedding forward: word/visual/linguistic sum + position + token-type -> LayerNorm -> dropout.
 * text_vis / obj_vis / obj_ling are bf16 with element strides (batch, position); when obj_ling_idx
 * (int64 [B,R]) is non-NULL obj_ling is a table [n,H] indexed by it.  Saves `pre` (bf16 pre-LN sum)
 * and stats for backward. */
int vlb_embed_fwd(const int32_t* code, const int32_t* text_len, const int64_t* text_ids, const int64_t* text_type,
                  const void* word_emb, const void* pos_emb, const void* type_emb, const void* end_emb,
                  const void* text_vis, long tv_sb, long tv_st, const void* obj_vis, long ov_sb, long ov_sr,
                  const void* obj_ling, long ol_sb, long ol_sr, const int64_t* obj_ling_idx, const float* gamma,
                  const float* beta, void* pre, float* stats, void* out, int B, int T, int R, int S, int H, int V,
                  int P, float eps, float drop_p, const uint32_t* seed, uint32_t tag, vlb_stream_t stream);
/* backward of the above: accumulates (fp32 atomics) into the embedding-table gradients and writes the
 * visual-part gradients (d_text_vis: per token, or per sample when dtv_st == 0; d_obj_vis per object;
 * d_obj_ling dense, or the [2,H] table gradient in obj_ling_idx mode).
 * text_vis_zeroed != 0: the caller has zeroed d_text_vis, which lets several workgroups share one sample in the
 * per-sample (dtv_st == 0) mode (their partial sums are then added atomically); 0 keeps one workgroup per sample. */
int vlb_embed_bwd(const void* dy, const void* pre, const float* stats, const float* gamma, const int32_t* code,
                  const int32_t* text_len, const int64_t* text_ids, const int64_t* text_type,
                  const int64_t* obj_ling_idx, float* d_word, float* d_pos, float* d_type, float* d_end,
                  float* d_gamma, float* d_beta, float* d_text_vis, long dtv_sb, long dtv_st, float* d_obj_vis,
                  long dov_sb, long dov_sr, float* d_obj_ling, long dol_sb, long dol_sr, int B, int T, int R, int S,
                  int H, int V, int P, float drop_p, const uint32_t* seed, uint32_t tag, int text_vis_zeroed,
                  vlb_stream_t stream);
/* out[i] = src[idx[i]] (rows of H bf16; idx < 0 -> zero row): text/object split of
 * visual_linguistic_bert.py:146-166 */
int vlb_gather_rows(const void* src, const int32_t* idx, void* out, int n, int H, vlb_stream_t stream);
/* out[idx[i]] = src[i] for idx[i] >= 0: inverse of vlb_gather_rows through the same index list */
int vlb_scatter_rows(const void* src, const int32_t* idx, void* out, int n, int H, vlb_stream_t stream);
/* MLM head compaction (BertOnlyMLMHead, modeling.py:439-482 + F.cross_entropy(ignore_index=-1),
 * resnet_vlbert_for_pretraining.py:176-178): the unlabelled text positions contribute nothing to the loss and get exactly zero
 * d(logits) rows, so the head runs on the labelled rows only.  vlb_mlm_compact lists them (stable order): sel_pos[k] = position in
 * [0, n), sel_src[k] = src_rows[position], labels_c[k] = label (all -1 beyond the count), count0 / count1 = labelled rows among
 * positions < n_split / >= n_split; *overflow is set when more than `cap` rows carry a label (the excess is dropped: an error for
 * the caller).  vlb_ce_fwd_bwd_compact = vlb_ce_fwd_bwd on such rows with the two group means (loss_out0 over count0 rows, then
 * loss_out1 over count1 rows). */
int vlb_mlm_compact(const int64_t* labels, const int32_t* src_rows, int n, int n_split, int V, int cap, int32_t* sel_pos,
                    int32_t* sel_src, int64_t* labels_c, float* count0, float* count1, int32_t* overflow, vlb_stream_t stream);
int vlb_ce_fwd_bwd_compact(void* logits, long ld, int rows, int V, const int64_t* labels_c, const float* count0, const float* count1,
                           float gscale, float* loss_out0, float* loss_out1, vlb_stream_t stream);
/* inverse of the split: dX[b,s] = (s<T ? d_text[b,s] : 0) + (row is object j ? d_obj[b,j] : 0) */
int vlb_head_grad_combine(const void* d_text, const void* d_obj, const int32_t* code, void* dx, int B, int T, int R,
                          int S, int H, vlb_stream_t stream);
/* out_bf16 = (y > 0) ? g_f32 : 0   (ReLU backward of obj_downsample, fast_rcnn.py:108) */
int vlb_relu_bwd_cast(const float* g, const void* y, void* out, long n, vlb_stream_t stream);
/* out = dg * gelu'(u)  (bf16; backward of the erf-GELU at modeling.py:114-120 where it is not fused
 * into a GEMM epilogue: BertPredictionHeadTransform, modeling.py:448-452) */
int vlb_dgelu_mul(const void* dg, const void* u, void* out, long n, vlb_stream_t stream);
/* out = a * b (bf16, n % 8 == 0): the same backward when the forward GEMM saved gelu'(u) (act 4) instead of u */
int vlb_mul_bf16(const void* a, const void* b, void* out, long n, vlb_stream_t stream);
/* out = dy * (1 - y^2): backward of the pooler's tanh (y = the saved bf16 output), n % 4 == 0 */
int vlb_tanh_bwd(const void* dy, const void* y, void* out, long n, vlb_stream_t stream);

/* ---- losses, forward+backward fused, gradient written in place over the bf16 logits ----------
 * MLM: F.cross_entropy(ignore_index=-1) (resnet_vlbert_for_pretraining.py:176-178).
 * MVRC: soft_cross_entropy (common/utils/misc.py:124-151).  `counts` = 1 device float (n_valid),
 * `loss_out` is accumulated (+=).  gscale multiplies the gradient (1/grad-accumulation steps).
 * logits_copy (optional) receives the untouched logits for API parity / metrics. */
int vlb_ce_fwd_bwd(void* logits, long ld, int rows, int V, const int64_t* labels, float* counts, float gscale,
                   float* loss_out, void* logits_copy, long ldcopy, vlb_stream_t stream);
int vlb_soft_ce_fwd_bwd(void* logits, long ld, int rows, int C, const float* target, long ldt, float* tsum,
                        float* counts, float gscale, float* loss_out, void* logits_copy, long ldcopy,
                        vlb_stream_t stream);

/* VQA answer loss (vqa/modules/resnet_vlbert_for_vqa.py:226): binary_cross_entropy_with_logits(logits[rows,A], label) * A =
 * (1/rows) sum of the element losses; logits (bf16, row stride ld, columns >= A zeroed) are overwritten IN PLACE by
 * gscale * d(loss)/d(logits) = gscale * w * (sigmoid(x) - y) / rows; loss_out is accumulated (+=); logits_copy (optional) keeps the logits.
 * w = pos_weight where y > 0.5, else 1: the element `weight` of the VCR answer loss (vcr/modules/resnet_vlbert_for_vcr.py:333-341; 1 = VQA). */
int vlb_bce_logits_fwd_bwd(void* logits, long ld, int rows, int A, const float* label, long ldl, float gscale, float pos_weight,
                           float* loss_out, void* logits_copy, long ldcopy, vlb_stream_t stream);
/* y[i] = x[i] * keep(seed, tag, i) / (1 - p): elementwise dropout with the library's counter RNG (the classifier dropouts of the
 * VQA head, resnet_vlbert_for_vqa.py:57-77); the same call on the gradient is its backward. */
int vlb_dropout_bf16(const void* x, void* y, long n, float drop_p, const uint32_t* seed, uint32_t tag, vlb_stream_t stream);

/* ---- optimizer over flat buffers (common/nlp/bert/optimization.py:129-187,
 *      torch.nn.utils.clip_grad_norm_ at common/trainer.py:139-145) ---------------------------
 * state: DEVICE float[8] = {lr, beta1, beta2, eps, weight_decay, step, max_norm, sumsq}.
 * vlb_sumsq_f32 accumulates sum(g^2) into *out (point it at &state[7]); vlb_adamw_step applies
 * clip coef = min(1, max_norm/(sqrt(sumsq)*grad_scale+1e-6)) * grad_scale, updates p/m/v, writes
 * the bf16 copy, then increments step and zeroes sumsq. */
/* zero n ranges of one fp32 buffer in one launch: ranges (device) = n x {start, length} int64, block_start (device, n+1
 * int32) = running count of 1024-float blocks, block_start[n] == total_blocks.  Used for the gradients that are accumulated by
 * atomics (biases, LayerNorm parameters, small tables) when the GEMM weight gradients are written with accumulate = 0. */
/* x[0..n) *= alpha: the 1 / world average of an all-reduced gradient handed back through autograd (parallel.DistributedDataParallel) */
int vlb_scale_f32(float* x, long n, float alpha, vlb_stream_t stream);
int vlb_zero_ranges_f32(float* base, const int64_t* ranges, const int32_t* block_start, int n, int total_blocks, vlb_stream_t stream);
/* dst[dst_start + i] = src[src_start + i] over n ranges {src_start, dst_start, length} in one launch (block_start: running count of
 * 1024-float blocks, n + 1 entries): pack / unpack of the fp32-read parameters (biases, LayerNorm gamma / beta) that the sharded
 * data-parallel optimizer replicates after its owner-only AdamW -- replaces the implicit "every rank updates everything" of DDP,
 * pretrain/function/train.py:89-90 */
int vlb_copy_ranges_f32(const float* src, float* dst, const int64_t* ranges, const int32_t* block_start, int n, int total_blocks,
                        vlb_stream_t stream);

int vlb_sumsq_f32(const float* g, long n, float* out, vlb_stream_t stream);
/* same sum with a fixed summation order (per-block partials in `partials[partials_len]`, then one block): bit-identical on every
 * data-parallel rank for identical gradients, so the clip coefficient -- and therefore the replicas' parameters -- cannot drift. */
int vlb_sumsq_f32_det(const float* g, long n, float* partials, int partials_len, float* out, vlb_stream_t stream);
int vlb_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float* state, float grad_scale,
                   vlb_stream_t stream);
/* the same two steps on a bf16 gradient image: the wire format of the data-parallel exchange (DDP all-reduce of
 * pretrain/function/train.py:89-90 carried out in bf16 over RCCL); the reduced gradient is consumed as it arrived. */
int vlb_sumsq_bf16_det(const void* g_bf16, long n, float* partials, int partials_len, float* out, vlb_stream_t stream);
int vlb_adamw_step_gbf16(float* p, const void* g_bf16, float* m, float* v, void* p_bf16, long n, float* state, float grad_scale,
                         vlb_stream_t stream);
/* Sharded optimizer of the data-parallel path (replaces the REPLICATED clip + AdamW that DDP implies, common/trainer.py:139-153 after
 * pretrain/function/train.py:89-90): a rank owns one slice of every gradient bucket, the reduce-scatter leaves the reduced slices in a
 * compact image `g` (fp32, or bf16 when g_is_bf16), and clip + AdamW run over those slices only.  ranges: DEVICE int64 [n][3] =
 * {offset into p / m / v, offset into the compact images, length}; block_start: DEVICE int32 [n + 1] = running count of
 * `chunk`-element blocks (chunk a multiple of 1024, total_blocks <= partials_len).  vlb_sumsq_ranges_det adds the slices' sum of
 * squares into *out in a fixed order (the caller all-reduces it over the ranks); vlb_adamw_step_ranges is vlb_adamw_step over the
 * slices, the bf16 copy of the updated parameters written to p_bf16_compact (nullable) at the compact offsets -- the image the
 * weight all-gather distributes -- then step += 1, sumsq = 0. */
int vlb_sumsq_ranges_det(const void* g, int g_is_bf16, const int64_t* ranges, const int32_t* block_start, int n, int total_blocks, int chunk,
                         float* partials, int partials_len, float* out, vlb_stream_t stream);
int vlb_adamw_step_ranges(float* p, const void* g, int g_is_bf16, float* m, float* v, void* p_bf16_compact, const int64_t* ranges,
                          const int32_t* block_start, int n, int total_blocks, int chunk, float* state, float grad_scale, vlb_stream_t stream);
/* SGD with momentum (torch.optim.SGD(lr, momentum, weight_decay), dampening 0, no Nesterov: the optimiser of
 * vcr/function/train.py:124-128) in one pass:  d = coef * g + weight_decay * p ;  buf = momentum * buf + d ;  p -= lr * buf ;
 * p_bf16 (optional) = bf16(p).  coef = grad_scale * min(1, max_norm / (sqrt(*sumsq) * grad_scale + 1e-6)) when `sumsq` (device,
 * e.g. from vlb_sumsq_f32) is given and max_norm > 0 -- clip_grad_norm_ of common/trainer.py:139-145 -- else grad_scale. */
int vlb_sgd_momentum_step(float* p, const float* g, float* momentum_buf, void* p_bf16, long n, float lr, float momentum,
                          float weight_decay, const float* sumsq, float max_norm, float grad_scale, vlb_stream_t stream);
/* lr schedule evaluated on the device from state[5] (steps taken): state[0] = base_lr * lambda(step + 1), matching the
 * reference's scheduler.step() -> optimizer.step() order (common/trainer.py:131-135).  kind 0 = ConstantLRSchedule,
 * 1 = WarmupConstantSchedule, 2 = WarmupLinearSchedule (common/nlp/bert/optimization.py:27-62).  Call before
 * vlb_adamw_step; graph-replayable (no host value changes between steps). */
int vlb_lr_schedule_step(float* state, int kind, float base_lr, float warmup_steps, float t_total, vlb_stream_t stream);
int vlb_cast_f32_bf16(const float* in, void* out, long n, vlb_stream_t stream);
int vlb_cast_bf16_f32(const void* in, float* out, long n, vlb_stream_t stream);
int vlb_rng_advance(uint32_t* seed, vlb_stream_t stream);

/* ---- end-to-end vision path: NHWC bf16 convolution support (common/backbone/resnet/resnet.py:75-199 Bottleneck /
 *      ResNet.forward, common/fast_rcnn.py:55-100,144-156 backbone + RoI head) ---------------------------------------
 * A feature map is a row-major [N*H*W, C] bf16 matrix.  1x1 convolutions call vlb_gemm_nt_bf16 on it directly; 3x3
 * convolutions call it on the vlb_im2col_nhwc_bf16 image ([rows, KH*KW*C], tap-major); frozen BatchNorm
 * (common/fast_rcnn.py:88-100, eval mode :122-126) is folded: y = conv(x, w*scale) + shift, scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale.  GEMM act 7 = relu(acc + bias + res) (Bottleneck tail, resnet.py:112-116), act 8 =
 * (acc [+ res]) where aux > 0 (ReLU backward).
 * vlb_conv_weight_prepare: master w fp32 [O, taps, I] (taps = KH*KW, row-major ky,kx) -> wf bf16 [O, kf] (forward / wgrad
 *   layout, kf >= taps*I, padding untouched), wb bf16 [I, taps*O] with mirrored taps (dgrad operand: dX = im2col(dY) . wb^T
 *   for stride 1, padding = dilation*(KH-1)/2), scale/shift fp32 [O] (any of wb/scale/shift may be NULL; gamma NULL = no BN).
 * vlb_conv_wgrad_finalize: g[O, kreal] (+)= scale[o] * dwf[O, kf][:, :kreal]. */
int vlb_conv_weight_prepare(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                            float eps, void* wf, void* wb, float* scale, float* shift, int O, int I, int taps, int kf,
                            vlb_stream_t stream);
/* n convolutions in one launch: desc (device) = n x 13 int64 {w, gamma, beta, mean, var, wf, wb, scale, shift, O, I, taps, kf} (pointers
 * as integers, 0 = NULL), block_start (device, n int32) = index of each convolution's first 1024-element block, total_blocks = their sum. */
int vlb_conv_weight_prepare_batched(const int64_t* desc, const int32_t* block_start, int n, int total_blocks, float eps,
                                    vlb_stream_t stream);
int vlb_conv_wgrad_finalize(const float* dwf, const float* scale, float* g, int O, int kreal, int kf, int accumulate,
                            vlb_stream_t stream);
/* implicit 3x3 convolution, stride 1, padding = dilation: y[N*H*W, O] = epi(im2col(x) . w^T) without the im2col image in HBM
 * (the gather runs in the GEMM's LDS-DMA address generator).  x: NHWC bf16 [N*H*W, C], C % 64 == 0; w: [O, 9*C] tap-major
 * (vlb_conv_weight_prepare's wf, or its wb for the data gradient); act 0 bias | 2 bias+ReLU | 8 x (aux > 0); zero16: >= 16 B
 * of device zeros (source of out-of-image taps). */
int vlb_conv3x3_nhwc_bf16(const void* x, int N, int H, int W, int C, int dil, const void* w, long ldw, void* y, long ldy,
                          int O, const float* bias, int act, const void* aux, long ldaux, const void* zero16,
                          vlb_stream_t stream);
/* weight gradient of the same convolution without the im2col image: dW[O, 9*C] fp32 (+)= dy[N*H*W, O]^T . im2col(x); the TN
 * GEMM gathers the shifted pixels of x itself.  C % 128 == 0; workspace as vlb_wgrad_tn_bf16 (vlb_wgrad_workspace_floats(O, 9C, rows)). */
int vlb_conv3x3_wgrad_tn_bf16(const void* dy, long lddy, const void* x, int N, int H, int W, int C, int dil, float* dW,
                              long lddw, int O, const float* rowscale, float* workspace, long workspace_floats, int accumulate,
                              vlb_stream_t stream);
/* vlb_wgrad_tn_bf16 with output row m scaled by rowscale[m] (folded frozen-BatchNorm: dW_master = scale[o] * dW_folded) in the
 * slab reduce -- no separate finalize pass.  rowscale (also optional in vlb_conv3x3_wgrad_tn_bf16) needs
 * workspace >= splits * Mo * round4(No) floats, at least Mo * round4(No). */
/* n <= 4 weight gradients over the SAME R rows in one launch (the four Linear layers of a transformer block): dW_i[Mo_i, No_i] (+)=
 * A_i[R, Mo_i]^T B_i[R, No_i], colsum_i[Mo_i] += column sums of A_i (entries / the array may be NULL).  Arrays are HOST arrays of n
 * entries.  workspace >= 2 * sum_i Mo_i * round4(No_i) floats lets the grouped large-tile kernel run (otherwise n single calls). */
int vlb_wgrad_tn_group_bf16(int n, const void* const* A, const long* lda, const void* const* B, const long* ldb, float* const* C,
                            const long* ldc, int R, const int* Mo, const int* No, float* const* colsum, float* workspace,
                            long workspace_floats, int accumulate, vlb_stream_t stream);
int vlb_wgrad_tn_rowscale_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int R, int Mo, int No,
                               const float* rowscale, float* workspace, long workspace_floats, int accumulate, vlb_stream_t stream);
/* Table-driven weight gradients (round 4): n products  C_i[Mo_i, No_i] (+)= rowscale_i[m] * (A_i[R_i, Mo_i]^T B_i[R_i, No_i])  (+ colsum_i),
 * each with its own row count R_i (a multiple of 128, >= 256; operands zero-padded to it), as ONE launch of full-K 256 x 256 work
 * items on the large-tile core: no K slices, no slabs, no reduce pass.  The descriptor table is built once on the HOST for a fixed
 * set of device buffers (_pack: returns the number of work items, 0 when a product is outside what the kernel covers), copied to the
 * device by the caller and replayed every step (_launch).  Replaces the per-convolution dW = dY^T X products of autograd in a ResNet
 * stage (common/backbone/resnet/resnet.py:98-118) -- vision.VisionStack defers the 1x1-convolution gradients of a stage into one. */
long vlb_wgrad_tn_table_desc_bytes(void);
int vlb_wgrad_tn_table_pack(int n, const void* const* A, const long* lda, const void* const* B, const long* ldb, float* const* C,
                            const long* ldc, const int* R, const int* Mo, const int* No, float* const* colsum,
                            const float* const* rowscale, int accumulate, void* host_out, long host_bytes);
int vlb_wgrad_tn_table_launch(const void* desc_dev, int n, int nitems, vlb_stream_t stream);
/* Measurement hook of the weight-gradient core (tools/tn8_probe.py; not used by the training path): dev = NULL (off) or a device
 * table of 4 x uint64 per workgroup (256 workgroups at most) -- every later launch of the core stamps {shader cycles, 100 MHz ticks}
 * at each workgroup's entry and exit, i.e. the clock the chip sustains under it and every workgroup's residency. */
int vlb_tn8_set_stamps(void* dev);
int vlb_im2col_nhwc_bf16(const void* x, void* col, long ldcol, int N, int H, int W, int C, int KH, int KW, int stride,
                         int pad, int dil, vlb_stream_t stream);
/* stem (resnet.py:137-141): fp32 NCHW image -> [N*OH*OW, ldcol] bf16, column (ky*KW+kx)*Cin + c, zero padded to ldcol */
int vlb_im2col_image_f32(const float* img, void* col, int ldcol, int N, int Cin, int H, int W, int KH, int KW, int stride,
                         int pad, vlb_stream_t stream);
/* raw-pixel masking of the masked regions of an e2e batch (pretrain/data/datasets/conceptual_captions.py:201-206): for every box slot
 * with mvrc_ops == 1, img[n, :, int(y1):int(y2)+1, int(x1):int(x2)+1] = 0; img fp32 NCHW, boxes [N, R, ldb] (x1, y1, x2, y2, ...) */
int vlb_mask_image_boxes_f32(float* img, int N, int C, int H, int W, const float* boxes, long ldb, int R, const int64_t* mvrc_ops,
                             vlb_stream_t stream);
int vlb_maxpool3x3s2_nhwc(const void* x, void* y, int N, int H, int W, int C, vlb_stream_t stream);
/* stride-2 1x1 convolutions (caffe-style stride_in_1x1, resnet.py:79): y = x[:, ::2, ::2, :]; its adjoint writes all of dx */
int vlb_subsample2_nhwc(const void* x, void* y, int N, int H, int W, int C, vlb_stream_t stream);
int vlb_upsample2_zero_nhwc(const void* dy, void* dx, int N, int H, int W, int C, vlb_stream_t stream);
/* ROIAlign on NHWC bf16 features (same arithmetic as vlb_roi_align_fwd/bwd below).  RoI k = image k / boxes_per_image,
 * (x1,y1,x2,y2) = boxes[k*ldbox + 0..3] -- the rows common/fast_rcnn.py:145-149 assembles; boxes with x1 <= -1.5 are
 * padding (pretrain/data/collate_batch.py:39): zero output, no gradient.  out / dout: [K, ph, pw, C] bf16.
 * Backward zeroes dfeat (fp32 [N,H,W,C]) and scatters with atomics. */
int vlb_roi_align_nhwc_fwd(const void* feat, const float* boxes, long ldbox, int boxes_per_image, void* out, int K, int C,
                           int H, int W, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                           vlb_stream_t stream);
int vlb_roi_align_nhwc_bwd(const void* dout, const float* boxes, long ldbox, int boxes_per_image, float* dfeat, int K, int N,
                           int C, int H, int W, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                           vlb_stream_t stream);
/* The same backward as a gather over the feature map (the form the engine uses): the bilinear weights factorise into a per-RoI
 * [bins_h, H] and [bins_w, W] matrix (built in `workspace` by a first kernel from the forward's sampling arithmetic), one workgroup
 * per feature pixel sums its RoIs' contributions -- no atomics, no zero-fill; every element of dx_bf16 [N,H,W,C] and / or dx_f32 is
 * written.  act (optional, bf16 [N,H,W,C]): dx_bf16 = 0 where act <= 0 (the ReLU in front of ROIAlign, i.e. vlb_relu_mask_cast
 * folded in).  K = N * boxes_per_image; C % 4 == 0; workspace: 16-byte aligned, vlb_roi_align_gather_workspace_bytes(K, ...). */
long vlb_roi_align_gather_workspace_bytes(int K, int H, int W, int pooled_h, int pooled_w);
int vlb_roi_align_nhwc_bwd_gather(const void* dout, const float* boxes, long ldbox, int boxes_per_image, const void* act,
                                  void* dx_bf16, float* dx_f32, void* workspace, long workspace_bytes, int N, int C, int H, int W,
                                  int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, vlb_stream_t stream);
/* dz (bf16) = g (fp32) where y (bf16) > 0 else 0; n % 8 == 0 */
int vlb_relu_mask_cast(const float* g, const void* y, void* dz, long n, vlb_stream_t stream);
/* AvgPool2d(14)+Flattener (common/fast_rcnn.py:80-84): y [K,P,C] bf16 -> out[k*ld + col0 + c] fp32 (the feature slots of
 * the padded box rows; rows whose column pad_col (>= 0) holds x1 <= -1.5 are padding and get zeros); backward: dz[k,p,c] = (y>0) * keep(k*drop_row_elems + drop_col0 + c) * dfeat[k*lddf + c] / P, with
 * the input-dropout mask of obj_downsample (common/fast_rcnn.py:106) regenerated from (seed, tag); padded boxes get 0.
 * segm (optional, fp32 [K, P]): VCR's per-pixel object mask, multiplied into the RoI-head output before pooling (:152-156). */
int vlb_avgpool_rows_fwd(const void* y, float* out, long ld, int col0, int pad_col, int K, int P, int C, const float* segm,
                         vlb_stream_t stream);
int vlb_avgpool_rows_bwd(const void* dfeat, long lddf, const void* y, const float* boxes, long ldbox, void* dz, int K, int P,
                         int C, float drop_p, const uint32_t* seed, uint32_t tag, uint32_t drop_row_elems, uint32_t drop_col0,
                         const float* segm, vlb_stream_t stream);

/* ---- ROIAlign (common/lib/roi_pooling: vision.cpp:6-11, ROIAlign.h:11-45) --------------------
 * NCHW fp32, rois [K,5] = (batch_idx, x1, y1, x2, y2); same math as
 * cuda/ROIAlign_cuda.cu:15-122 (forward) and :125-254 (backward, atomic scatter into a
 * zero-initialised grad_input).  Exposed to Python under the reference's own names
 * roi_align_forward / roi_align_backward by `vl-bert_amd/common/lib/roi_pooling/C_ROIPooling.py`. */
int vlb_roi_align_fwd(const float* input, const float* rois, float* output, int num_rois, int channels, int height,
                      int width, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                      vlb_stream_t stream);
int vlb_roi_align_bwd(const float* grad_output, const float* rois, float* grad_input, int num_rois, int batch,
                      int channels, int height, int width, int pooled_h, int pooled_w, float spatial_scale,
                      int sampling_ratio, vlb_stream_t stream);


/* ---------------------------------------------------------------------------------------------------------------
 * fp32 compute path of the encoder (csrc/f32_path.hip; host side vl-bert_amd/encoder_f32.py): the reference's fp32 configurations
 * (TRAIN.FP16: false, cfgs/pretrain/base_prec_4x16G_fp32.yaml:112, cfgs/vqa/large_4x16G_fp32.yaml:108) -- per layer
 * external/pytorch_pretrained_bert/modeling.py:268-397 with every tensor in fp32.  The fp32 products run on the bf16 matrix cores
 * by operand splitting (x = h + m, 3 MFMAs per product, fp32 accumulation: ~2^-16 relative per product).
 *
 * vlb_gemm_nt_f32: C[M,N] (+)= epilogue(alpha * A[M,K] . B[N,K]^T), batched over nb1 x nb2 (element strides s?1 / s?2).
 *   epilogue: + bias[n] (row i1 * sBias1: a Linear bias, or the per-sample additive attention mask), activation epi (0 none |
 *   1 erf-GELU with GELU' -> pre | 2 ReLU | 3 x aux | 4 tanh | 5 keep where aux > 0), dropout(drop_p, counter RNG, element m*N+n),
 *   + res.  atomic != 0: C += by atomicAdd over splitk K slices (weight gradients), no epilogue.  K % 32 == 0 (zero-pad),
 *   N % 4 == 0, leading dimensions / strides % 4 == 0, 16-byte aligned pointers.
 * vlb_transpose_f32: dst[c][r] = src[r][c], rows R..Rp of the result zero (the padded reduction dimension of the GEMM above);
 *   colsum += column sums of src (bias gradients; unbatched).
 * vlb_layernorm_f32_fwd / _bwd: BertLayerNorm (modeling.py:222-235) on fp32 rows; bwd writes dx and / or the dropout-masked dx_drop
 *   (element row*H+col, the mask of the GEMM that produced the row) and accumulates dgamma / dbeta.
 * vlb_softmax_f32_fwd / _bwd: softmax over the first S of Sp <= 256 columns; scores carry 1/sqrt(d); mask01 (nullable, [samples][S],
 *   1 = attend) adds the reference's (1 - mask) * -10000 to the keys of sample row / rows_per_sample; p = the
 *   probabilities (0 in the padding), pd = dropout(p); bwd: dpd <- p * (dp - sum dp p) in place, dp = dropout mask applied to dpd. */
int vlb_gemm_nt_f32(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K, int nb1, int nb2,
                    long sA1, long sA2, long sB1, long sB2, long sC1, long sC2, const float* bias, long sBias1, float alpha, int epi,
                    const float* aux, long ldaux, float* pre, long ldpre, const float* res, long ldres, float drop_p,
                    const uint32_t* seed, uint32_t tag, int atomic, int splitk, vlb_stream_t stream);
/* C[Mo,No] (+)= alpha * A[R,Mo]^T . B[R,No]: the same split products with the reduction over the ROWS of two row-major fp32 operands --
 * weight gradients dW = dY^T X (autograd's grad_output.t().mm(input) behind every nn.Linear, external/pytorch_pretrained_bert/modeling.py:
 * 268-397 in fp32) and the attention products P^T dO / dS^T Q without transposed copies.  No % 4 == 0; operand rows are read in whole 8-column chunks (lda >= Mo rounded up to 8, ldb likewise);
 * atomic: accumulate into C (splitk K slices allowed), colsum (nullable, unbatched): += column sums of A (the bias gradient). */
int vlb_gemm_tn_f32(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int R, int Mo, int No, int nb1, int nb2,
                    long sA1, long sA2, long sB1, long sB2, long sC1, long sC2, float alpha, int atomic, int splitk, float* colsum,
                    vlb_stream_t stream);
int vlb_transpose_f32(const float* src, long lds, float* dst, long ldd, int R, int C, int Rp, int nb1, int nb2, long sS1, long sS2,
                      long sD1, long sD2, float* colsum, vlb_stream_t stream);
int vlb_layernorm_f32_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy, float* stats, int rows,
                          int H, float eps, vlb_stream_t stream);
int vlb_layernorm_f32_bwd(const float* dy, long lddy, const float* x, long ldx, const float* stats, const float* gamma, float* dx,
                          long lddx, float* dx_drop, long lddd, float drop_p, const uint32_t* seed, uint32_t tag, float* dgamma,
                          float* dbeta, int rows, int H, vlb_stream_t stream);
int vlb_softmax_f32_fwd(const float* s, const float* mask01, int rows_per_sample, float* p, float* pd, int rows, int S, int Sp, float drop_p,
                        const uint32_t* seed, uint32_t tag, vlb_stream_t stream);
int vlb_softmax_f32_bwd(const float* p, float* dpd, int rows, int S, int Sp, float drop_p, const uint32_t* seed, uint32_t tag,
                        vlb_stream_t stream);

/* ---- data-parallel gradient exchange for a C / C++ host: RCCL over xGMI (csrc/comm.hip) ---------------------------------------
 * Replaces the gradient all-reduce that torch DistributedDataParallel / apex DDP issue for the reference
 * (pretrain/function/train.py:89-90,353-354; vqa/function/train.py:327; vcr/function/train.py:330): one SUM over the ranks of every
 * gradient bucket per optimizer step; the 1/world average is `grad_scale` of the optimizer entry points.  One communicator = one rank
 * = one GPU (hipSetDevice first).  RCCL is bound at run time (dlopen; the copy a torch process already holds is reused, VLB_RCCL_PATH
 * overrides), so the library loads without it.  Collectives are enqueued on `stream` and never synchronise.
 * dtype: 0 = fp32, 1 = the library's 16-bit type (the wire image written by vlb_cast_f32_bf16).
 *   vlb_comm_unique_id     : rank 0 fills 128 opaque bytes and distributes them through the host's own channel
 *   vlb_comm_init          : collective over `world` ranks -> *comm
 *   vlb_comm_allreduce_bucket      : buf[count] <- SUM over ranks, in place (the all-reduce exchange)
 *   vlb_comm_reduce_scatter_bucket : out[count / world] <- this rank's slice of the SUM of buf[count]   } the two halves of the sharded
 *   vlb_comm_allgather_bucket      : out[count] <- the ranks' in[count / world], rank order             } optimizer (ZeRO-1 style)
 *   vlb_comm_finalize      : destroys the communicator (NULL is a no-op)
 * The Python host of this repo issues the same collectives through torch.distributed on the same slices (vl-bert_amd/parallel.py). */
int vlb_comm_unique_id(void* id128);
int vlb_comm_init(int rank, int world, const void* id128, void** comm);
int vlb_comm_allreduce_bucket(void* comm, void* buf, long count, int dtype, vlb_stream_t stream);
int vlb_comm_reduce_scatter_bucket(void* comm, const void* buf, void* out, long count, int dtype, vlb_stream_t stream);
int vlb_comm_allgather_bucket(void* comm, const void* in, void* out, long count, int dtype, vlb_stream_t stream);
int vlb_comm_finalize(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* VLBERT_HIP_H */
