/* astts.h -- C ABI of libastts.so, the MI355X (gfx950) implementation of the AutoStyle-TTS
 * inference hot path.  Plain C: pointers, sizes, integer status codes.  No torch types, no
 * exceptions across the boundary.
 *
 * Conventions
 *   - every function returns ASTTS_OK (0) or a negative ASTTS_ERR_* code; the message for the
 *     calling thread's last failure is astts_last_error_string().
 *   - pointers are DEVICE pointers owned by the caller unless the name ends in _host.
 *   - every launch takes the HIP stream to enqueue on (astts_stream_t == hipStream_t); no
 *     function in a launch path allocates, frees or synchronises (graph-capturable).
 *   - handles are opaque and thread-compatible: one stream per handle at a time.
 *
 * What each group replaces in the reference (paths under /root/reference):
 *   astts_knn_*      MilvusClient.search on the COSINE collection
 *                    milvus/search_embeddings.py:15-22, src/search_milvus.py:140-147,
 *                    milvus/search_json.py:232-259 (+ _ab_bio / _ab_text variants),
 *                    milvus/RAG.py:368-395, milvus/search.py:159-186
 *   astts_op_*       the tensor operators executed inside cosyvoice.inference_tts_with_st /
 *                    inference_zero_shot / inference_vc (tts_with_rag.py:195,133,141;
 *                    tts_with_style_and_timbre.py:93,47,57): acoustic-transformer, flow-matching
 *                    decoder and HiFT vocoder arithmetic (third-party CosyVoice, not vendored).
 */
#ifndef ASTTS_H_
#define ASTTS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASTTS_ABI_VERSION 5

#define ASTTS_OK 0
#define ASTTS_ERR_INVALID (-1)     /* bad argument (null pointer, size, dtype, k, ...)            */
#define ASTTS_ERR_HIP (-2)         /* HIP runtime failure; message carries hipGetErrorString      */
#define ASTTS_ERR_UNSUPPORTED (-3) /* valid in the reference API, not implemented by this build   */
#define ASTTS_ERR_WORKSPACE (-4)   /* workspace too small / misaligned                            */
#define ASTTS_ERR_RANGE (-5)       /* data not representable (e.g. bank value overflows fp16)     */

typedef void* astts_stream_t; /* hipStream_t */

int astts_abi_version(void);
const char* astts_last_error_string(void);

/* Bench-only launch profiler (bench.py's roofline leg): while a kind is enabled, every launch of that
 * kind is bracketed by HIP events on its own stream and its algorithmic work (flops for GEMM /
 * attention kinds, bytes for streaming kinds) is accumulated.  _read synchronises those events,
 * returns the sums since the last read and resets.  Process-global, not thread-safe, off by default. */
#define ASTTS_PROF_GEMM_TILE 0   /* gemm_tile  (flow estimator / vocoder contractions): work = flops  */
#define ASTTS_PROF_GEMM_SKINNY 1 /* LM decode GEMVs, M <= 32 (lm_gemv / gemm_skinny16): work = weight bytes streamed */
#define ASTTS_PROF_ATTN_FLASH 2  /* attn_mha_flash: work = flops                                      */
#define ASTTS_PROF_ATTN_DECODE 3 /* LM decode attention (lm_attn / attn_relpos_decode): work = KV + position bytes read */
#define ASTTS_PROF_KINDS 4
int astts_prof_enable(int32_t kind, int32_t on, int32_t max_launches);
int astts_prof_read(int32_t kind, double* ms_sum, int64_t* launches, double* work_sum, int64_t* dropped);
/* One 64-thread workgroup that busy-waits `microseconds` on `stream`.  HIP multiplexes its streams onto a few
 * hardware queues; two streams that share one never overlap.  The pipelined synthesis uses this probe to pick streams
 * that really run concurrently (astts/synth/model.py: PipelinedSynth). */
int astts_stream_spin(int32_t microseconds, astts_stream_t stream);
/* `count` dependent busy-wait launches (`blocks` workgroups, `microseconds` each): a launch-chain stand-in for stream probing */
int astts_stream_chain(int32_t count, int32_t microseconds, int32_t blocks, astts_stream_t stream);
/* A stream restricted to the CUs whose bit is set (bit i of mask[i / 32] = CU i; hipExtStreamCreateWithCUMask): CU partitions
 * between concurrent stages.  The caller destroys it with astts_stream_destroy once nothing is in flight on it. */
int astts_stream_create_cu_mask(const uint32_t* mask, int32_t n_words, astts_stream_t* out);
int astts_stream_destroy(astts_stream_t stream);
/* Self-test of the cross-lane exchanges of csrc/xlane.h (DPP modifiers, v_permlane16_swap / v_permlane32_swap) against
 * __shfl_xor on pseudo-random values: every butterfly offset and the composed 64-lane sums / maxima must agree bit for bit.
 * *mismatches = number of differing words (0 = the decode-step reductions compute what their shuffle form computed).
 * Synchronises the stream. */
int astts_selftest_xlane(int32_t* mismatches, astts_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Style-bank kNN.  Replaces MilvusClient.search(collection, data=[vec], anns_field="vector",
 * metric_type="COSINE", limit=k) -- milvus/search_embeddings.py:15-22.
 *
 * Result definition (identical to oracle/knn.py): score(q,n) = <q,b_n>/(|q||b_n|) (COSINE; <q,b_n> for IP, |q - b_n|^2 for L2)
 * evaluated in fp64, hits ordered by (closest first, row index ascending).  Returned ids are ROW INDICES
 * into the bank (the reference's pk restarts per speaker and is not unique, RAG.py:507); ids are
 * bit-exact w.r.t. that definition: an fp16-MFMA scan proposes candidates, every candidate is
 * re-scored in fp64, and a query whose candidate set cannot be certified complete (error bound
 * vs. the gap to the best non-candidate) is re-run through an exact fp64 scan on the GPU.
 * ------------------------------------------------------------------------------------------ */
#define ASTTS_DTYPE_F16 1
#define ASTTS_DTYPE_F32 2

#define ASTTS_METRIC_COSINE 0 /* score = cosine similarity, larger = closer (the reference's collection) */
#define ASTTS_METRIC_IP 1     /* score = inner product, larger = closer */
#define ASTTS_METRIC_L2 2     /* score = SQUARED Euclidean distance (Milvus' convention), smaller = closer */

#define ASTTS_KNN_MAX_K 1024  /* k > 32 runs ceil(k / 32) selection passes over one scan */
#define ASTTS_KNN_FORCE_EXACT 1 /* flags: send every query through the exact fp64 scan */

typedef struct astts_knn astts_knn_t;

/* Builds the HBM-resident bank from `bank` ([n,d] row-major device memory of `dtype`).  The
 * handle keeps its own copy (fp16 scan plane, padded to a multiple of 64 columns; the exact
 * plane for re-scoring; fp64 row norms).  Synchronises `stream` before returning. */
int astts_knn_create(const void* bank, int64_t n, int32_t d, int32_t dtype, int32_t metric,
                     astts_stream_t stream, astts_knn_t** out);
int astts_knn_destroy(astts_knn_t* h);
/* n, d, and whether the scan plane is a lossless image of the bank (1) or a rounded one (0). */
int astts_knn_info(const astts_knn_t* h, int64_t* n, int32_t* d, int32_t* scan_plane_exact);
/* Bytes of workspace astts_knn_search needs for up to `nq` queries and `k` hits. */
size_t astts_knn_workspace_bytes(const astts_knn_t* h, int32_t nq, int32_t k);
/* queries: fp32 [nq,d] device.  out_idx: int64 [nq,k], out_score: fp32 [nq,k] (cosine
 * similarity, larger = closer; rows beyond min(k,n) hits get idx -1 / score -inf).
 * workspace: 256-byte aligned device memory of at least astts_knn_workspace_bytes(h,nq,k). */
int astts_knn_search(astts_knn_t* h, const float* queries, int32_t nq, int32_t k,
                     int64_t* out_idx, float* out_score, void* workspace, size_t workspace_bytes,
                     int32_t flags, astts_stream_t stream);
/* The same search with the fp64 cosines as well (out_score64 fp64 [nq,k], may be NULL): what a bank-SHARDED search merges
 * on -- W ranks each return the top-k of their rows, and the k-way merge must order candidates of different ranks exactly
 * as the oracle orders them (fp64 score descending, global row ascending); fp32-rounded scores could tie where fp64 do not. */
int astts_knn_search_f64(astts_knn_t* h, const float* queries, int32_t nq, int32_t k,
                         int64_t* out_idx, float* out_score, double* out_score64, void* workspace, size_t workspace_bytes,
                         int32_t flags, astts_stream_t stream);
/* The general form: `row_mask` (uint8 [n] when mask_stride == 0, else one row of mask_stride >= n bytes per query; NULL = every
 * row) restricts the search to the rows whose byte is non-zero -- a Milvus `filter` expression evaluated by the host
 * (milvus/search_json.py:246-252 passes filter=None).  Hits are ordered closest first under the handle's metric
 * (COSINE / IP: score descending; L2: squared distance ascending), ties by row index ascending; out_score / out_score64 hold the
 * metric's own value; rows beyond the hits that exist get idx -1 and score -inf (+inf for L2).  1 <= k <= ASTTS_KNN_MAX_K. */
int astts_knn_search_masked(astts_knn_t* h, const float* queries, int32_t nq, int32_t k,
                            int64_t* out_idx, float* out_score, double* out_score64, const uint8_t* row_mask, int64_t mask_stride,
                            void* workspace, size_t workspace_bytes, int32_t flags, astts_stream_t stream);
/* Number of queries of the last search on `workspace` that took the exact-scan fallback.
 * Copies one word back and synchronises `stream` (diagnostics; not a launch-path call). */
int astts_knn_last_fallbacks(const astts_knn_t* h, const void* workspace, astts_stream_t stream,
                             int32_t* n_fallback_host);

/* Profiling aid for bench.py (roofline of the scan kernel): while enabled, every search brackets its
 * knn_scan launch(es) with HIP events on the search stream.  _read synchronises, returns the summed
 * scan time and launch count since the last read, and resets.  Not for production launch paths. */
int astts_knn_profile_enable(astts_knn_t* h, int32_t on);
int astts_knn_profile_read(astts_knn_t* h, double* scan_ms_sum, int64_t* scan_launches);

/* ------------------------------------------------------------------------------------------
 * Synthesis operators.  Replace the tensor arithmetic executed inside the reference's
 * cosyvoice.inference_tts_with_st / inference_zero_shot / inference_vc calls (tts_with_rag.py:195,
 * 133,141; tts_with_style_and_timbre.py:93,47,57).  That arithmetic lives in the authors' private
 * CosyVoice fork (not vendored): these follow the published CosyVoice-300M architecture.
 * Activations: fp32, channels-last [B, T, C] row-major.  Weights: fp16 packed by
 * astts_op_pack_weight.  All launches are asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------ */
#define ASTTS_ACT_NONE 0
#define ASTTS_ACT_RELU 1
#define ASTTS_ACT_SILU 2
#define ASTTS_ACT_GELU 3
#define ASTTS_ACT_MISH 4
#define ASTTS_ACT_ELU 5
#define ASTTS_ACT_TANH 6
#define ASTTS_ACT_LEAKY 7

/* fp32 [n, taps, cin] -> fp16 [n_pad, taps, cin_pad], zero padded (n_pad % 128 == 0, cin_pad % 64 == 0). */
int astts_op_pack_weight(const float* src, void* dst_f16, int32_t n, int32_t taps, int32_t cin,
                         int32_t n_pad, int32_t cin_pad, astts_stream_t stream);
/* Implicit GEMM (nn.Linear / nn.Conv1d / phase-decomposed nn.ConvTranspose1d):
 *   out[m, n] = act(sum_{tap,c} x[b*t_in + t*stride + tap*dil - pad, c] * w[n, tap, c] + bias[n]) * alpha
 *               * row_scale[m] + residual[m, n],      m = b*t_out + t, zero outside [0, t_in). */
int astts_op_gemm(const float* x, const void* w_f16, const float* bias, const float* residual,
                  const float* row_scale, float* out, int64_t m, int32_t n, int32_t cin, int32_t cin_pad,
                  int32_t taps, int32_t lda, int32_t ldc, int32_t ldr, int32_t t_in, int32_t t_out,
                  int32_t stride, int32_t dil, int32_t pad, int32_t act, float alpha, float slope,
                  astts_stream_t stream);
/* Same contraction with fp16 activation input and/or output (x_f16 / out_f16 != 0; lda / ldc then count halfs).
 * A producer whose only consumers are MFMA operands writes fp16, halving its store and the consumer's load. */
int astts_op_gemm_ex(const void* x, int32_t x_f16, const void* w_f16, const float* bias, const float* residual,
                     const float* row_scale, void* out, int32_t out_f16, int64_t m, int32_t n, int32_t cin, int32_t cin_pad,
                     int32_t taps, int32_t lda, int32_t ldc, int32_t ldr, int32_t t_in, int32_t t_out,
                     int32_t stride, int32_t dil, int32_t pad, int32_t act, float alpha, float slope,
                     astts_stream_t stream);
/* astts_op_gemm_ex for RAGGED batches (one vocoder pass over utterances of different lengths: HiFTGenerator.inference behind
 * /root/reference/tts_with_rag.py:195, which the reference runs one utterance at a time): input time steps at or beyond
 * in_lens[batch row] (int32 [m / t_out]) are read as zero -- each row convolves as a sequence of its own length with the
 * convolution's zero padding behind it.  Output rows beyond a row's own length hold unspecified values. */
int astts_op_gemm_lens(const void* x, int32_t x_f16, const void* w_f16, const float* bias, const float* residual,
                       const float* row_scale, void* out, int32_t out_f16, int64_t m, int32_t n, int32_t cin, int32_t cin_pad,
                       int32_t taps, int32_t lda, int32_t ldc, int32_t ldr, int32_t t_in, int32_t t_out,
                       int32_t stride, int32_t dil, int32_t pad, int32_t act, float alpha, float slope, const int32_t* in_lens,
                       astts_stream_t stream);
/* Latency-sized GEMM for batches of a few hundred rows (the wide decode engine: one nn.Linear of TransformerLM's decoder layers for
 * every row of a 33 .. 256-row decode batch, /root/reference/tts_with_rag.py:195 -> cosyvoice llm.inference, one launch per
 * projection): out[m, n] = act(x[m, :] . w[n, :] + bias[n]) + residual[m, n].  x fp32 or fp16 [m, lda], w fp16 [>= n rows, k]
 * (k a multiple of 64, K contiguous), out fp32 or fp16 [m, ldc].  out2 (or NULL): columns >= n_split go to out2[m, n - n_split]
 * (row stride ldc2, fp16 when out2_f16: the K | V half of a q | k | v projection lands in the KV cache).  Every operand fragment
 * of a workgroup is requested before its first MFMA (one memory round trip); a row's sums do not depend on the other rows of the
 * launch nor on timing. */
int astts_op_gemm_rows(const void* x, int32_t x_f16, const void* w_f16, const float* bias, const float* residual, void* out,
                       int32_t out_f16, void* out2, int32_t out2_f16, int32_t m, int32_t n, int32_t n_split, int32_t k, int32_t lda,
                       int32_t ldc, int32_t ldc2, int32_t ldr, int32_t act, astts_stream_t stream);
/* Decode-sized GEMM (m <= 32, weight-bandwidth bound) with the fusions that take whole launches out of
 * an LM decode step: optional row gather (x row of output row i = x[gather[i]], i.e. an embedding lookup),
 * optional LayerNorm(gamma, beta, eps) over the cin inputs of every row applied while loading, and an
 * optional second destination for the output columns >= n_split (out2[m*ldc2 + n - n_split], fp32 or fp16). */
int astts_op_gemm_fused(const float* x, const int32_t* gather, const float* ln_gamma, const float* ln_beta, float ln_eps,
                        const void* w_f16, const float* bias, const float* residual, float* out, void* out2, int32_t out2_f16,
                        int32_t m, int32_t n, int32_t n_split, int32_t cin, int32_t cin_pad, int32_t lda, int32_t ldc,
                        int32_t ldc2, int32_t ldr, int32_t act, float alpha, float slope, astts_stream_t stream);
/* Row-complete GEMM with the residual add and the NEXT LayerNorm fused into the epilogue (flow-decoder transformer blocks:
 * attention-out and FFN-out projections, n == 256 == one workgroup's column range):
 *   out[m, :] = x[m, :] @ w^T + bias + residual[m, :]      (fp32, the new residual stream)
 *   ln_out[m, :] = LayerNorm(out[m, :]) * gamma + beta     (fp16, operand of the next projection)
 * x fp16 [m, lda], cin == cin_pad (multiple of 64); everything 16-byte aligned.  Replaces a GEMM launch + a LayerNorm launch.
 * Measured on the flow decoder (5504 rows): 16.6 us against 8.6-12 us + 5 us for the 128x64 / 64x64 tiles followed by a
 * LayerNorm launch -- a 32-row x 256-column workgroup streams the whole weight (0.25-0.5 MB) and only 172 of them exist, so the
 * flow engine keeps the two launches; the operator stays available for wider row counts.
 * (diffusers BasicTransformerBlock: `hidden = attn(norm1(hidden)) + hidden; hidden = ff(norm3(hidden)) + hidden`, [EXT]). */
/* Tile choice of the LDS-DMA ring GEMM (fp16 activations): -1 = by shape (default; env ASTTS_GEMM_RING overrides), 0 = ring
 * kernel off (register-staged tiles), 1 = 128x128 two-stage, 2 = 128x64 two-stage, 3 = 64x64 four-stage, 4 = 256x256 two-stage
 * with eight waves on one barrier per K tile, 5 = the same tile on the eight-phase schedule (what "by shape" picks for large GEMMs: the
 * embedder's projections, the kNN scan of >= 64 queries; results bit-identical to 4).  Process-global
 * test / tuning switch: the parity tests run the benchmark's projection shapes through every tile. */
int astts_op_gemm_set_ring_mode(int32_t mode);
int astts_op_gemm_ln(const void* x_f16, const void* w_f16, const float* bias, const float* residual, float* out,
                     const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out_f16, int64_t m, int32_t n, int32_t cin,
                     int32_t cin_pad, int32_t lda, int32_t ldc, int32_t ldr, int32_t ld_ln, astts_stream_t stream);
/* Same, with a workspace that enables split-K for deep, narrow shapes (K >= 2048 onto <= 2048 columns, no gather / LayerNorm:
 * the FFN-out projection of a decode step): 4 K slices per column block, the last slice to arrive adds the partial sums in
 * slice order, so results are reproducible.  The workspace (astts_op_gemm_fused_workspace_bytes(), 256-byte aligned) must
 * be ZERO on first use and is left zeroed where it matters; launches sharing it must be ordered on one stream. */
size_t astts_op_gemm_fused_workspace_bytes(void);
int astts_op_gemm_fused_ws(const float* x, const int32_t* gather, const float* ln_gamma, const float* ln_beta, float ln_eps,
                        const void* w_f16, const float* bias, const float* residual, float* out, void* out2, int32_t out2_f16,
                        int32_t m, int32_t n, int32_t n_split, int32_t cin, int32_t cin_pad, int32_t lda, int32_t ldc,
                        int32_t ldc2, int32_t ldr, int32_t act, float alpha, float slope, void* workspace, size_t workspace_bytes,
                           astts_stream_t stream);
int astts_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t c,
                       int32_t ldx, int32_t ldy, float eps, astts_stream_t stream);
/* _ex forms: out_f16 != 0 writes fp16 (ldy in halfs) for outputs whose only consumers are MFMA operands. */
int astts_op_layernorm_ex(const float* x, const float* gamma, const float* beta, void* y, int32_t out_f16, int64_t rows,
                          int32_t c, int32_t ldx, int32_t ldy, float eps, astts_stream_t stream);
/* relu_scale > 0: y = relu_scale * max(LayerNorm(x), 0) -- the LM input embedding's LayerNorm -> ReLU -> * sqrt(d) in one launch. */
int astts_op_layernorm_relu(const float* x, const float* gamma, const float* beta, void* y, int32_t out_f16, int64_t rows,
                            int32_t c, int32_t ldx, int32_t ldy, float eps, float relu_scale, astts_stream_t stream);
int astts_op_groupnorm_ex(const float* x, const int32_t* lens, const float* gamma, const float* beta,
                          const float* add_bc, void* y, int32_t out_f16, int32_t b, int32_t t, int32_t c, int32_t groups,
                          float eps, int32_t act_mish, void* workspace, size_t workspace_bytes, astts_stream_t stream);
size_t astts_op_groupnorm_workspace_bytes(int32_t b, int32_t t, int32_t groups);
/* y = act(GroupNorm(x over the first lens[b] rows)) (+ add_bc[b, c]); rows >= lens[b] are written as 0. */
int astts_op_groupnorm(const float* x, const int32_t* lens, const float* gamma, const float* beta,
                       const float* add_bc, float* y, int32_t b, int32_t t, int32_t c, int32_t groups, float eps,
                       int32_t act_mish, void* workspace, size_t workspace_bytes, astts_stream_t stream);
#define ASTTS_EL_SNAKE 0
#define ASTTS_EL_LEAKY 1
#define ASTTS_EL_ADD 2
#define ASTTS_EL_MUL_ROWMASK 3
#define ASTTS_EL_ADD_BC 4
#define ASTTS_EL_SCALE 5
#define ASTTS_EL_CFG_EULER 6
#define ASTTS_EL_MISH 7
#define ASTTS_EL_SILU 8
#define ASTTS_EL_CLAMP 9
#define ASTTS_EL_TANH 10
#define ASTTS_EL_ELU 11
#define ASTTS_EL_RELU_SCALE 12
int astts_op_elementwise(int32_t op, const float* x, const float* z, const float* p0, const int32_t* lens, float* y,
                         int64_t total, int32_t t, int32_t c, float s, float s2, astts_stream_t stream);
int astts_op_embedding(const float* table, const int32_t* ids, float* y, int64_t rows, int32_t c, int32_t ldy,
                       int32_t vocab, float scale, astts_stream_t stream);
int astts_op_interp_linear(const float* x, float* y, int32_t b, int32_t t_in, int32_t t_out, int32_t c,
                           astts_stream_t stream);
/* ragged form: row b is resampled from in_lens[b] to out_lens[b] steps (rows padded to t_in / t_out); NULL = uniform. */
int astts_op_interp_linear_ex(const float* x, float* y, int32_t b, int32_t t_in, int32_t t_out, int32_t c,
                              const int32_t* in_lens, const int32_t* out_lens, astts_stream_t stream);
int astts_op_time_embedding(const float* t, float* y, int32_t b, int32_t dim, float scale, astts_stream_t stream);
/* espnet relative-position attention, head dim 64; tq == 1 selects the KV-cache decode kernel.
 * ld* = time-step strides, *_bs = batch strides (elements): batch-major and time-major layouts both work. */
int astts_op_attn_relpos(const float* q, const float* k, const float* v, const float* pos, const float* bias_u,
                         const float* bias_v, const int32_t* lens, float* out, int32_t b, int32_t h, int32_t tq,
                         int32_t tk, int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldp, int64_t q_bs, int64_t k_bs,
                         int64_t o_bs, int32_t q_pos0, int32_t pos_center, int32_t causal, float scale,
                         astts_stream_t stream);
/* _ex: K/V (e.g. a KV cache) and/or the position table may be fp16 (ldk / k_bs / ldp then count halfs).
 * key_start[b] (or NULL): first valid key of row b -- rows of a ragged batch are LEFT-padded to a common length
 * (relative positions make that exact) and keys < key_start[b] are masked. */
int astts_op_attn_relpos_ex(const float* q, const void* k, const void* v, int32_t kv_f16, const void* pos, int32_t pos_f16,
                            const float* bias_u, const float* bias_v, const int32_t* lens, const int32_t* key_start, float* out,
                            int32_t b, int32_t h, int32_t tq, int32_t tk, int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldp,
                            int64_t q_bs, int64_t k_bs, int64_t o_bs, int32_t q_pos0, int32_t pos_center, int32_t causal,
                            float scale, astts_stream_t stream);
/* masked multi-head attention (flash-style MFMA), head dim 64. */
int astts_op_attn_mha(const float* q, const float* k, const float* v, const int32_t* lens, float* out, int32_t b,
                      int32_t h, int32_t t, int32_t ldq, int32_t ldk, int32_t ldo, float scale, astts_stream_t stream);
int astts_op_attn_mha_ex(const void* q, const void* k, const void* v, int32_t in_f16, const int32_t* lens, void* out,
                         int32_t out_f16, int32_t b, int32_t h, int32_t t, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                         astts_stream_t stream);
size_t astts_op_nsf_source_workspace_bytes(int32_t b, int32_t tm);
int astts_op_nsf_source(const float* f0, const float* phase0, const float* noise, const float* lin_w, const float* lin_b,
                        float* out, int32_t b, int32_t tm, int32_t upsample, int32_t n_harm_plus1, float sample_rate,
                        float sine_amp, float noise_std, float voiced_threshold, void* workspace, size_t workspace_bytes,
                        astts_stream_t stream);
int astts_op_stft16(const float* x, float* y, int32_t b, int64_t n_samples, astts_stream_t stream);
/* ragged batches: lens int32 [b] = samples of each row (a multiple of 4, >= 16; NULL: n_samples).  A row is transformed as a signal of
 * its own length (the reflection at its end mirrors its own last samples); frames behind lens[b] / 4 are zero. */
int astts_op_stft16_lens(const float* x, float* y, int32_t b, int64_t n_samples, const int32_t* lens, astts_stream_t stream);
int astts_op_istft16(const float* y, float* wav, int32_t b, int64_t frames, float mag_clip, float audio_limit,
                     astts_stream_t stream);
/* ragged batches: frame_lens int32 [b] = frames of each row (NULL: frames); a row's overlap-add sees its own frames only, samples behind
 * 4 (frame_lens[b] - 1) are zero. */
int astts_op_istft16_lens(const float* y, float* wav, int32_t b, int64_t frames, float mag_clip, float audio_limit, const int32_t* frame_lens,
                          astts_stream_t stream);
/* Frontend signal processing on the GPU (the reference does both on the host inside CosyVoice's frontend
 * [EXT], reached from load_wav / inference_* at tts_with_rag.py:180-195):
 * polyphase resampler  y[f * up + p] = sum_j kern[p][j] * x[f * down + j - width]  (kern [up][2 width + down]: the Hann-windowed
 * sinc table of astts.audio.resample, zero outside its support; x zero beyond its ends), and the log-mel spectrogram
 * (reflect padding (n_fft - hop) / 2, window[n_fft], magnitude DFT, mel_fb [n_mels][n_fft / 2 + 1], log(max(., log_floor)))
 * -> out [b, frames, n_mels], frames = (n_samples + 2 pad - n_fft) / hop + 1. */
int astts_op_resample_poly(const float* x, const float* kern, float* y, int32_t b, int64_t n_in, int64_t n_out, int32_t up, int32_t down,
                           int32_t width, astts_stream_t stream);
int astts_op_mel_spectrogram(const float* wav, const float* window, const float* mel_fb, float* out, int32_t b, int64_t n_samples,
                             int32_t n_fft, int32_t hop, int32_t n_mels, float log_floor, astts_stream_t stream);
/* Whisper's log-mel features (the input of the speech tokenizer behind frontend._extract_speech_token [EXT], reached from
 * inference_tts_with_st at tts_with_rag.py:195): frames centred on multiples of `hop` (reflect padding n_fft / 2, the last frame dropped:
 * n_samples / hop frames), periodic-Hann window[n_fft], POWER spectrum, mel_fb [n_mels][n_fft / 2 + 1], log10(max(., 1e-10)), floor at
 * the utterance's maximum - 8, (x + 4) / 4 -> out [b, n_mels, n_samples / hop].  workspace: astts_op_whisper_log_mel_workspace_bytes(b). */
size_t astts_op_whisper_log_mel_workspace_bytes(int32_t b);
int astts_op_whisper_log_mel(const float* wav, const float* window, const float* mel_fb, float* out, int32_t b, int64_t n_samples,
                             int32_t n_fft, int32_t hop, int32_t n_mels, void* workspace, size_t workspace_bytes, astts_stream_t stream);
/* Kaldi's fbank (compute-fbank-feats), the input of the speaker-embedding network behind frontend._extract_spk_embedding [EXT] (reached
 * from inference_tts_with_st at tts_with_rag.py:195): frames of frame_len samples every hop samples without padding (1 + (n - frame_len) /
 * hop of them), each multiplied by `scale`, freed of its DC offset, pre-emphasised (x[i] -= preemph x[i-1]; x[0] *= 1 - preemph), multiplied
 * by window[frame_len] (Povey) and zero-padded to n_fft (<= 512); power spectrum, mel_fb [n_mels][n_fft / 2 + 1], log(max(., log_floor))
 * -> out [b, frames, n_mels]. */
int astts_op_kaldi_fbank(const float* wav, const float* window, const float* mel_fb, float* out, int32_t b, int64_t n_samples, int32_t frame_len,
                         int32_t hop, int32_t n_fft, int32_t n_mels, float scale, float preemph, float log_floor, astts_stream_t stream);
/* The learned half of the frontend: glue operators of the CAM++ speaker network (campplus.onnx [EXT]) and the
 * speech tokenizer (speech_tokenizer_v1.onnx [EXT]) that CosyVoice(model_dir) loads (tts_with_rag.py:159) and runs on every prompt
 * (tts_with_rag.py:179-195); their contractions run on astts_op_gemm_* / astts_op_attn_mha_ex, the quantiser's arg-min on
 * astts_knn_* with ASTTS_METRIC_L2 (csrc/ops_frontend.hip; host: astts/frontend_nets.py; definition: oracle/frontend_nets.py).
 * affine_act: y[r, k] = act(x[r, k] * scale[k] + shift[k]) (eval-mode BatchNorm + ReLU in front of a convolution; scale / shift NULL
 *   = identity; act 0 none, 1 relu; x / y fp32 or fp16 with row strides ldx / ldy).
 * freq_unfold: y[b, fo, t, kf * c + k] = x[b, fo * sf + kf - (nkf - 1) / 2, t, k] (0 outside the input rows) as fp16: the nkf = 3
 *   frequency rows of a 3 x 3 convolution window side by side, so that the convolution is a 3-tap astts_op_gemm_ex over time.
 * ftc_to_tfc: [b, f, t, c] -> [b, t, f * c].
 * cam_context: ctx[b, s, k] = mean_t h[b, t, k] + mean over segment s (seg_len frames, the last one shorter) of h[b, t, k].
 * cam_gate: out[b, t, k] = y[b, t, k] * sigmoid(m[b, t / seg_len, k]) with output row stride ldo (a column block of the dense
 *   block's concatenation buffer).
 * stats_pool: out[b, k] = mean_t x[b, t, k], out[b, c + k] = unbiased standard deviation.
 * l2_normalize: y[r, :] = x[r, :] / max(|x[r, :]|, eps).
 * sub_time_mean: x[b, t, k] -= mean_t x[b, t, k] in place (upstream's frontend removes the fbank's mean over time before campplus). */
int astts_op_affine_act(const void* x, int32_t x_f16, int64_t ldx, const float* scale, const float* shift, void* y, int32_t y_f16,
                        int64_t ldy, int64_t rows, int32_t c, int32_t act, astts_stream_t stream);
int astts_op_freq_unfold(const void* x, int32_t x_f16, void* y_f16, int32_t b, int32_t f_in, int32_t t, int32_t c, int32_t f_out, int32_t sf,
                         int32_t nkf, astts_stream_t stream);
int astts_op_ftc_to_tfc(const float* x, float* y, int32_t b, int32_t f, int32_t t, int32_t c, astts_stream_t stream);
int astts_op_cam_context(const void* h, int32_t h_f16, int64_t ldh, float* ctx, int32_t b, int32_t t, int32_t c, int32_t seg_len,
                         astts_stream_t stream);
int astts_op_cam_gate(const float* y, const float* m, float* out, int64_t ldo, int32_t b, int32_t t, int32_t c, int32_t seg_len,
                      astts_stream_t stream);
int astts_op_stats_pool(const float* x, int64_t ldx, float* out, int32_t b, int32_t t, int32_t c, astts_stream_t stream);
int astts_op_l2_normalize(const float* x, float* y, int64_t rows, int32_t c, float eps, astts_stream_t stream);
int astts_op_sub_time_mean(float* x, int32_t b, int32_t t, int32_t c, astts_stream_t stream);
/* Repetition-aware sampling with injected uniforms [b, 2] (definition: csrc/ops_audio.hip, mirrored by oracle/synth.py::ras_sample).
 * ignore_eos: bit 0 = EOS may not be produced at this step (with eos_min_rows: per row, while hist_len < eos_min_rows[b]); bit 1 = the
 * policy inside that window: 0 mask, 1 reject (astts_lm_config_t.eos_policy). */
int astts_op_ras_sample(const float* logits, const int32_t* history, const float* uniforms, int32_t* out_tokens,
                        int32_t b, int32_t vocab, int32_t hist_len, int32_t hist_ld, int32_t top_k, float top_p,
                        int32_t win_size, float tau_r, int32_t eos_id, int32_t ignore_eos, astts_stream_t stream);
int astts_op_ras_sample_ex(const float* logits, int32_t* history, const float* uniforms, int32_t* out_tokens, int32_t b,
                           int32_t vocab, int32_t hist_len, int32_t hist_ld, int32_t top_k, float top_p, int32_t win_size,
                           float tau_r, int32_t eos_id, int32_t ignore_eos, const int32_t* eos_min_rows, const int32_t* forced,
                           astts_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Acoustic-transformer decode engine: the autoregressive loop of TransformerLM.inference (one speech
 * token per step) issued from C++ with no host synchronisation -- 5 launches per layer and step.  Batches of <= 8 rows
 * take the decode-step kernels of csrc/lm_step.hip (73 launches per step), wider ones the operator chain (74);
 * ASTTS_LM_ENGINE=v1|v2 forces one of them.
 * All pointers are device pointers that must outlive the handle (weights packed by
 * astts_op_pack_weight; fp32 biases / norms / tables).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t d, heads, ffn, layers;
    int32_t vocab_out;    /* speech_vocab + 1 (EOS logit) */
    int32_t speech_vocab; /* EOS id */
    int32_t pos_center, pos_ld; /* relative-position tables: row (rel + pos_center), row stride pos_ld floats */
    int32_t top_k, ras_win;
    float top_p, ras_tau, eps;
    int32_t kv_f16, pos_f16; /* KV cache / position tables stored as fp16 */
    int32_t ln_folded;       /* 1: the scale / shift of norm1, norm2 and after_norm are folded into wqkv, w1 and the head
                              * (W' = W diag(gamma), b' = b + W beta, done by the host at load); the n*_g / n*_b / after_*
                              * arrays then hold ones / zeros and the step kernels normalise without reading them */
    int32_t eos_policy;      /* what "EOS may not be produced yet" means: 0 = mask (the EOS logit is removed before the softmax),
                              * 1 = reject (upstream TransformerLM.sampling_ids [EXT]: sample again until the token is not EOS; EOS
                              * keeps its probability and its place in the nucleus) -- see astts_op_ras_sample */
} astts_lm_config_t;
typedef struct {
    const float* speech_emb;                 /* [speech_vocab, d] */
    const void* embed_w; const float* embed_b;       /* llm.embed.out.0 */
    const float* embed_ln_g; const float* embed_ln_b; /* llm.embed.out.1 */
    const float* after_g; const float* after_b;       /* llm.after_norm */
    const void* head_w; const float* head_b;         /* llm_decoder */
    const float* embed_table;                /* optional [speech_vocab, d] fp32 = speech_emb x embed_w^T + embed_b, formed once at
                                              * load: the decode step then gathers its rows instead of running the projection
                                              * (one launch per step less); NULL: the projection runs every step */
} astts_lm_globals_t;
typedef struct {
    const float* n1_g; const float* n1_b;
    const void* wqkv; const float* bqkv;             /* [3d, d]: q | k | v */
    const void* wo; const float* bo;
    const float* n2_g; const float* n2_b;
    const void* w1; const float* b1;
    const void* w2; const float* b2;
    const float* pos; const float* bias_u; const float* bias_v;
} astts_lm_layer_t;
typedef struct astts_lm astts_lm_t;
int astts_lm_create(const astts_lm_config_t* cfg, const astts_lm_globals_t* globals, const astts_lm_layer_t* layers,
                    astts_lm_t** out);
int astts_lm_destroy(astts_lm_t* h);
/* rows per decode call: <= 32 rows take the decode-step kernels (csrc/lm_step.hip), 33 .. ASTTS_LM_MAX_ROWS the wide engine (one plain GEMM
 * per projection for all rows: the weights are read once per token; needs the fp16 cache / tables and globals.embed_table) */
#define ASTTS_LM_MAX_ROWS 256
size_t astts_lm_workspace_bytes(const astts_lm_t* h, int32_t b);
/* logits0 [b, vocab_out]: logits of the last prefix position; kv_cache[l]: fp32 [t_max, b, 2d] (time-major,
 * rows < pos0 filled by the prefill; fp16 when cfg.kv_f16); uniforms [n_steps, b, 2]; forced_tokens [b, n_steps] or NULL;
 * key_start int32 [b] or NULL (left-padded ragged prefixes); tokens_out int32 [b, n_steps]; logits_out [b, n_steps, vocab_out] or NULL.  The EOS logit is masked for the
 * first eos_min_steps steps (pass n_steps for fixed-length decoding), or per row for eos_min_rows[b] steps when that
 * device array is given (ragged batches); rows keep decoding after an EOS -- the caller
 * truncates at the first EOS id (== speech_vocab). */
int astts_lm_decode(astts_lm_t* h, const float* logits0, void* const* kv_cache, const int32_t* key_start, int32_t t_max,
                    int32_t b, int32_t pos0, int32_t n_steps, const float* uniforms, const int32_t* forced_tokens, int32_t eos_min_steps,
                    const int32_t* eos_min_rows, int32_t* tokens_out, float* logits_out, void* workspace, size_t workspace_bytes,
                    astts_stream_t stream);
/* Steps [s_begin, s_end) of that n_steps decode (stream=True of /root/reference/tts_for_dialog.py:188, vc_from_dir.py:18: upstream's
 * LM thread hands tokens to token2wav hop by hop).  The ranges of one decode are issued in order on ONE stream with the same
 * kv_cache, tokens_out (the sampler's history) and workspace (it carries the logits from one range to the next). */
int astts_lm_decode_range(astts_lm_t* h, const float* logits0, void* const* kv_cache, const int32_t* key_start, int32_t t_max,
                          int32_t b, int32_t pos0, int32_t n_steps, int32_t s_begin, int32_t s_end, const float* uniforms,
                          const int32_t* forced_tokens, int32_t eos_min_steps, const int32_t* eos_min_rows, int32_t* tokens_out,
                          float* logits_out, void* workspace, size_t workspace_bytes, astts_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Query-embedder operators: what a Llama-3.2 decoder block needs besides the GEMM family.
 * Replace the arithmetic of get_embedding() / generate_emotion_label() -- transformers' LlamaModel behind
 * src/search_milvus.py:75-108 and milvus/search_json.py:154-198,214-221.  Host side: astts/llm/embedder.py.
 * ------------------------------------------------------------------------------------------ */
/* y = w * (x * rsqrt(mean(x^2) + eps)); x fp32 [rows, ldx], y fp32 or fp16 [rows, ldy] */
int astts_op_rmsnorm(const float* x, const float* w, void* y, int32_t out_f16, int64_t rows, int32_t c, int32_t ldx, int32_t ldy,
                     float eps, astts_stream_t stream);
/* rotate-half RoPE in place on `heads` heads of head_dim columns starting at x (fp16 [b*t, ld]); cos / sin fp32
 * [>= pos0 + t, head_dim / 2] (llama3-scaled frequencies, built by the host exactly as transformers does) */
int astts_op_rope_llama(void* x_f16, const float* cos_tab, const float* sin_tab, int32_t b, int32_t t, int32_t heads, int32_t ld,
                        int32_t head_dim, int32_t pos0, astts_stream_t stream);
/* causal grouped-query attention, head_dim 128: q [b, t, heads*128], k / v [b, t, kv_heads*128] fp16 strided views
 * (row strides ldq / ldk in halfs), lens int32 [b] valid tokens (right padding) or NULL, out fp16 [b, t, ldo] */
int astts_op_attn_causal_gqa(const void* q_f16, const void* k_f16, const void* v_f16, const int32_t* lens, void* out_f16, int32_t b,
                             int32_t t, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                             astts_stream_t stream);
/* the same attention on the matrix cores (v_mfma_f32_32x32x16_f16; round 5), generalised for the generation path of
 * /root/reference/milvus/search_json.py:178-188 (model.generate, greedy): explicit element strides of batch row / time step
 * (sqb, sqt: q; skb, skt: k and v; sob, sot: out) so that a time-major KV cache is read in place; query i has key index
 * pos0 + i and attends keys [key_start[b], min(pos0 + i + 1, lens[b])) (either mask array may be NULL); head_dim 128 */
int astts_op_attn_gqa(const void* q_f16, const void* k_f16, const void* v_f16, const int32_t* lens, const int32_t* key_start, void* out_f16,
                      int32_t b, int32_t tq, int32_t tk, int32_t pos0, int32_t heads, int32_t kv_heads, int32_t head_dim, int64_t sqb, int64_t sqt,
                      int64_t skb, int64_t skt, int64_t sob, int64_t sot, float scale, astts_stream_t stream);
/* astts_op_rope_llama with the row order stated (time_major: row = t * b_count + b, else b * t_count + t) and a per-row position
 * shift (int32 [b] or NULL): position = max(pos0 + t - shift[b], 0) -- left-padded prompts keep transformers' positions */
int astts_op_rope_llama_ex(void* x_f16, const float* cos_tab, const float* sin_tab, const int32_t* shift, int32_t b, int32_t t, int32_t time_major,
                           int32_t heads, int32_t ld, int32_t head_dim, int32_t pos0, astts_stream_t stream);
/* out[row] = argmax over x[row, 0 .. n) (ties: the lowest index, as torch.argmax): the greedy step's token, on the device */
int astts_op_argmax_rows(const float* x, int32_t* out, int32_t rows, int32_t n, int64_t ld, astts_stream_t stream);
/* out = silu(gate) * up on a fused projection gate_up fp16 [rows, ldg] = gate[f] | up[f] */
int astts_op_swiglu(const void* gate_up_f16, void* out_f16, int64_t rows, int32_t f, int32_t ldg, int32_t ldo, astts_stream_t stream);
/* out[b, c] = mean over the first lens[b] (NULL: t) tokens of x fp32 [b, t, c] */
int astts_op_mean_pool(const float* x, const int32_t* lens, float* out, int32_t b, int32_t t, int32_t c, astts_stream_t stream);

/* LayerNorm (scale / shift folded into the weights by the caller) + q|k|v projection + masked multi-head attention of one
 * transformer block of the flow estimator in one launch (csrc/ops_tfm_fused.hip): x fp32 [b, t, c] -> out fp16
 * [b, t, heads*64].  wqkv_frag: the q | k | v weight [3*heads*64, c] re-ordered by astts_op_tfm_pack_frag (from the row-major
 * astts_op_pack_weight image) into MFMA fragment order; bias fp32 [3*heads*64] or NULL, lens int32 [b] or NULL.  Serves
 * c == 256, t <= 384 (astts_op_tfm_attn_fused_supported); otherwise ASTTS_ERR_UNSUPPORTED and the caller runs
 * astts_op_layernorm_ex + astts_op_gemm_ex + astts_op_attn_mha_ex on the row-major weight. */
int astts_op_tfm_pack_frag(const void* w_f16, void* out_f16, int32_t rows, int32_t k, astts_stream_t stream);
int astts_op_tfm_attn_fused_supported(int32_t c, int32_t heads, int32_t t);
int astts_op_tfm_attn_fused(const float* x, const void* wqkv_frag_f16, const float* bias, const int32_t* lens, void* out_f16, int32_t b,
                            int32_t heads, int32_t t, int32_t c, float eps, float scale, astts_stream_t stream);
/* Same, plus an L2 prefetch of up to three ranges (the NEXT launch's weights -- every block has its own, cold in L2): touched one
 * 128-byte line per thread while the attention phase runs. */
int astts_op_tfm_attn_fused_pf(const float* x, const void* wqkv_frag_f16, const float* bias, const int32_t* lens, void* out_f16, int32_t b,
                               int32_t heads, int32_t t, int32_t c, float eps, float scale, const void* const* pf_ptrs,
                               const uint32_t* pf_bytes, int32_t n_pf, astts_stream_t stream);

/* The feed-forward half of the same block in one launch: out = x' + W2 gelu(W1 LayerNorm(x') + b1) + b2, x / out fp32 [m, c]
 * (out may alias x).  LayerNorm scale / shift folded into w1 / b1 by the caller; the weights in fragment order
 * (astts_op_tfm_pack_frag of the row-major [hidden, c] and [c, hidden] images); exact-erf GELU as ASTTS_ACT_GELU.
 * attn_f16 == NULL: x' = x.  Otherwise the attention's output projection and residual run as a prologue of the same launch:
 * x' = x + attn Wo^T + bo with attn fp16 [m, k0] (k0 = 256 or 512), wo_frag the [c, k0] weight in fragment order, bo fp32 [c] or
 * NULL; x' is never written to memory.  Serves c == 256, hidden a multiple of 256 up to 4096
 * (astts_op_tfm_ffn_fused_supported); otherwise ASTTS_ERR_UNSUPPORTED and the caller runs astts_op_layernorm_ex +
 * astts_op_gemm_ex. */
int astts_op_tfm_ffn_fused_supported(int32_t c, int32_t hidden);
int astts_op_tfm_ffn_fused(const float* x, const void* w1_frag_f16, const float* b1, const void* w2_frag_f16, const float* b2, float* out,
                           int64_t m, int32_t c, int32_t hidden, float eps, const void* attn_f16, const void* wo_frag_f16,
                           const float* bo, int32_t k0, astts_stream_t stream);
/* Same, plus an L2 prefetch of one range (the next launch's weights). */
int astts_op_tfm_ffn_fused_pf(const float* x, const void* w1_frag_f16, const float* b1, const void* w2_frag_f16, const float* b2, float* out,
                              int64_t m, int32_t c, int32_t hidden, float eps, const void* attn_f16, const void* wo_frag_f16,
                              const float* bo, int32_t k0, const void* pf_ptr, uint32_t pf_bytes, astts_stream_t stream);

/* ---- HiFT resblock convolution, LDS-staged (csrc/ops_conv_lds.hip): y = conv1d_same(snake_alpha(x)) + bias + res on
 * channels-last [b, l, c] activations, c -> c channels (128 or 256), odd taps, dilation dil, zero padding dil*(taps-1)/2 on
 * both sides.  x fp32 or fp16 (x_f16); alpha fp32 [c] or NULL (no activation: snake(x) = x + sin^2(alpha x) / (alpha + 1e-9));
 * w_frag: the Conv1d weight re-ordered by astts_op_conv_pack_frag from the astts_op_pack_weight image [c, taps, c]; bias /
 * res (fp32 [b, l, c]) optional.  Outputs, either or both: y (fp32 or fp16) and acc (fp32): acc = (acc_add ? acc : 0) +
 * acc_scale * y -- the mean over the parallel resblocks.  Outputs may not alias x (workgroups read halo rows of their
 * neighbours); res may alias y.  astts_op_conv1d_snake_supported tells whether a shape is served. */
int astts_op_conv_pack_frag(const void* w_f16, void* out_f16, int32_t rows, int32_t taps, int32_t k, astts_stream_t stream);
int astts_op_conv1d_snake_supported(int32_t c, int32_t taps, int32_t dil);
int astts_op_conv1d_snake(const void* x, int32_t x_f16, const float* alpha, const void* w_frag_f16, const float* bias, const float* res,
                          void* y, int32_t y_f16, float* acc, float acc_scale, int32_t acc_add, int32_t b, int32_t l, int32_t c,
                          int32_t taps, int32_t dil, astts_stream_t stream);
/* the same over a RAGGED batch: lens int32 [b] = frames of each sequence (NULL: l).  Frames at or beyond a sequence's length are read as
 * zero (it convolves as if alone, zero padding behind it) and are not written; tiles wholly behind the end are skipped. */
int astts_op_conv1d_snake_lens(const void* x, int32_t x_f16, const float* alpha, const void* w_frag_f16, const float* bias, const float* res,
                               void* y, int32_t y_f16, float* acc, float acc_scale, int32_t acc_add, int32_t b, int32_t l, int32_t c,
                               int32_t taps, int32_t dil, const int32_t* lens, astts_stream_t stream);

/* ---- ResnetBlock1D convolutions with the GroupNorm + Mish passes folded in (csrc/ops_resnet_conv.hip): out = conv1d_same(x') +
 * bias [+ res'] on channels-last fp32 x [b, t, cin] -> out [b, t, 256], 1 or 3 taps, cin = 256 or 512 (the staging transform: 256 only).
 *   x'   = x, or (in_stats given) mask * (mish(GroupNorm(x; in_gamma, in_beta)) + in_add[b]) applied while the input tile is staged;
 *   res' = (res given) mask * mish(GroupNorm(res; res_gamma, res_beta)) added in the epilogue;
 *   out_stats (optional): (count, mean, M2) of THIS convolution's output per (sequence, 32-frame tile, group) over the valid frames
 *   -- what a later call takes as in_stats / res_stats (astts_op_resnet_conv_stats_floats(b, t) floats).
 * mask = (frame < lens[b]); lens NULL = all frames.  A ResNet block is three calls: conv 1 (out_stats), conv 2 (in_stats = those,
 * out_stats), 1x1 conv (res = conv 2's output, res_stats).  w_frag: astts_op_conv_pack_frag image. */
size_t astts_op_resnet_conv_stats_floats(int32_t b, int32_t t);
int astts_op_resnet_conv_supported(int32_t cin, int32_t cout, int32_t groups, int32_t taps);
int astts_op_resnet_conv(const float* x, const void* w_frag_f16, const float* bias, float* out, const float* in_stats, const float* in_gamma,
                         const float* in_beta, const float* in_add, const float* res, const float* res_stats, const float* res_gamma,
                         const float* res_beta, float* out_stats, const int32_t* lens, int32_t b, int32_t t, int32_t cin, int32_t taps,
                         float eps, astts_stream_t stream);
/* Same, plus an L2 prefetch of one range (the next launch's weights). */
int astts_op_resnet_conv_pf(const float* x, const void* w_frag_f16, const float* bias, float* out, const float* in_stats, const float* in_gamma,
                            const float* in_beta, const float* in_add, const float* res, const float* res_stats, const float* res_gamma,
                            const float* res_beta, float* out_stats, const int32_t* lens, int32_t b, int32_t t, int32_t cin, int32_t taps,
                            float eps, const void* pf_ptr, uint32_t pf_bytes, astts_stream_t stream);

/* ---- flow-matching solver engine: the reference's hot loop #3: ConditionalCFM.solve_euler
 * -> ConditionalDecoder.forward (cosyvoice/flow/flow_matching.py + decoder.py [EXT], behind
 * CosyVoice.inference_tts_with_st, tts_with_rag.py:195).  n_steps Euler steps of the U-Net estimator with
 * classifier-free guidance, ~520 kernel launches per step, issued from C++ with no host synchronisation
 * (a Python host spends as long enqueueing them as the GPU spends running them). */
typedef struct {            /* packed fp16 weight image (astts_op_pack_weight) + fp32 bias */
    const void* w; const float* bias;
    int32_t n, cin, cin_pad, taps;
} astts_weight_t;
typedef struct {            /* ResnetBlock1D: conv3 -> GN -> Mish (+ time proj) -> conv3 -> GN -> Mish, + 1x1 res conv */
    astts_weight_t c1, mlp, c2, res;
    const float *g1_w, *g1_b, *g2_w, *g2_b;
    const void *c1_frag, *c2_frag, *res_frag;   /* astts_op_conv_pack_frag images of c1 / c2 / res for astts_op_resnet_conv (256 -> 256
                                                 * blocks), or NULL: the five-launch path */
} astts_flow_resnet_t;
typedef struct {            /* BasicTransformerBlock: LN -> qkv -> attention -> out (+x) -> LN -> GELU FFN (+x) */
    const float *n1_w, *n1_b, *n3_w, *n3_b;
    astts_weight_t qkv, wo, w1, w2;
    const void* qkv_frag;   /* qkv.w in fragment order (astts_op_tfm_pack_frag) for the fused attention kernel, or NULL: unfused path */
    const void *w1_frag, *w2_frag;   /* w1.w (LayerNorm n3 folded in: n3_w / n3_b are then ones / zeros) and w2.w in fragment order for
                                      * astts_op_tfm_ffn_fused, or NULL: unfused path */
    const void* wo_frag;             /* wo.w in fragment order: the output projection runs inside astts_op_tfm_ffn_fused; or NULL */
} astts_flow_tfm_t;
#define ASTTS_FLOW_RESAMPLE_NONE 0      /* mid block */
#define ASTTS_FLOW_RESAMPLE_CONV 1      /* conv k=3 (last down / up block) */
#define ASTTS_FLOW_RESAMPLE_DOWN 2      /* conv k=3 stride 2 */
#define ASTTS_FLOW_RESAMPLE_UP 3        /* ConvTranspose1d k=4 stride 2 pad 1, phase-decomposed weight [2*C, 2, C] */
typedef struct {
    astts_flow_resnet_t res;
    const astts_flow_tfm_t* tfm; int32_t n_tfm;
    astts_weight_t resample; int32_t resample_kind;
} astts_flow_block_t;
typedef struct {
    int32_t mel, channels, heads, groups, time_in, time_dim;
    int32_t n_down, n_mid, n_up;
    astts_weight_t t1, t2, fin_c, fin_p;
    const float *fin_g_w, *fin_g_b;
} astts_flow_config_t;
typedef struct astts_flow astts_flow_t;
int astts_flow_create(const astts_flow_config_t* cfg, const astts_flow_block_t* down, const astts_flow_block_t* mid,
                      const astts_flow_block_t* up, astts_flow_t** out);
int astts_flow_destroy(astts_flow_t* h);
size_t astts_flow_workspace_bytes(const astts_flow_t* h, int32_t b, int32_t t);
/* x [b, t, mel]: in = the noise z, out = the solved mel (rows beyond lens stay 0).  mu, cond [b, t, mel], spk [b, mel]
 * (conditioning of the guided half; the unguided half of the internal 2b batch sees zeros).  lens int32 [b] or NULL
 * (every row uses all t frames: the length masks are then not launched).  t_host / dt_host: HOST arrays [n_steps] of the
 * step times and step sizes. */
int astts_flow_solve(astts_flow_t* h, float* x, const float* mu, const float* spk, const float* cond, const int32_t* lens,
                     int32_t b, int32_t t, int32_t n_steps, const float* t_host, const float* dt_host, float cfg_rate,
                     void* workspace, size_t workspace_bytes, astts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ASTTS_H_ */
