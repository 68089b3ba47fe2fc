/*
 * madeleine_amd.h -- C ABI of libmadeleine_amd.so (gfx950 / MI355X).
 *
 * The reference (mahmoodlab/MADELEINE) has no FFI / operator registry: its hot path is plain
 * PyTorch modules.  This header is therefore the boundary the reference's *call sites* would bind
 * if the path were native; every entry point names the reference code it replaces (file:line are
 * relative to the reference checkout).  The Python mirror of the reference classes
 * (madeleine_amd/{model,abmil,loss,trainer}.py) calls these through ctypes with raw device pointers
 * (see INTEGRATION.md for the binding stub).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the CALLER allocates all
 *     inputs, outputs and workspaces.  Kernels never allocate, free or keep global state; they are
 *     re-entrant.
 *   - launches are asynchronous on the caller-provided hipStream_t (passed as void*); no internal
 *     synchronisation.
 *   - return value: 0 on success; a positive hipError_t if a launch failed; a negative MDL_E_* code
 *     for argument validation.  Nothing throws across the boundary.
 *   - fp32 values everywhere.  Contractions come in three engines: the *_split entry points (the DEFAULT of the Python
 *     mirror: every fp32 operand as an fp16 hi + lo image under a power-of-two scale, three v_mfma_f32_32x32x16_f16 products
 *     per logical product, fp32 accumulation -- error below an fp32 fmaf chain), the plain entry points (exact fp32,
 *     v_mfma_f32_32x32x2_f32; MADELEINE_GEMM=fp32) and the *_bf16 entry points at the end of this header (the reference's
 *     `precision: bfloat16` mode: bf16 activation storage and bf16 MFMA operands, fp32 accumulation / epilogues / parameters).
 *
 * Entry points by row of SURVEY.md section 8:  A2 mdl_abmil_gate_*  |  A3 mdl_abmil_pool_*  |  A2+A3 fused backward
 * mdl_abmil_attnpool_bwd  |  L1 mdl_infonce_*  |  G0-G3 mdl_got_*  |  N1 mdl_linear_*, mdl_ln_gelu_drop_*  |  bf16 mode *_bf16.
 *
 * Internal activation layout ("head-major"): the reference interleaves heads as channel j = e*H + c
 * (rearrange 'b t (e c) -> b t e c', Model.py:396).  Our encoder emits the same numbers with the
 * channel axis permuted to j' = c*512 + e by permuting the rows of pre_attn.8 / LayerNorm 9 and the
 * columns of token_projector / projector on the host (madeleine_amd/model.py) -- zero-cost, and every
 * kernel below then reads contiguous 2 KiB head rows.  H (heads) <= 8, hidden = 512 fixed
 * (Model.py:71).
 */
#ifndef MADELEINE_AMD_H
#define MADELEINE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDL_HIDDEN 512
#define MDL_MAX_HEADS 8

#define MDL_OK 0
#define MDL_E_ARG (-1)      /* bad size / null pointer */
#define MDL_E_ALIGN (-2)    /* pointer not 16-byte aligned */
#define MDL_E_UNSUPPORTED (-3)

/* library / build info: returns a static string "madeleine_amd <ver> gfx950" */
const char* mdl_version(void);

/* ABI revision of this header: bumped whenever an entry point's argument list changes.  A binding compares
 * mdl_abi_version() with the MDL_ABI_VERSION it was written against BEFORE calling anything else, so that a stale
 * shared object fails loudly instead of being called with shifted arguments. */
#define MDL_ABI_VERSION 24
int mdl_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * A2 -- gated attention scores.  Replaces BatchedABMIL.forward (madeleine/models/abmil.py:41-68)
 * for all H heads at once (the reference loops H module instances, Model.py:406-409).
 *
 *   s[t,c] = bc[c] + sum_j wc[c,j] * drop(tanh(E[t,c,:].Wa[c,j,:] + ba[c,j])) * drop(sigmoid(E[t,c,:].Wb[c,j,:] + bb[c,j]))
 *
 * E      [T, H*512]  head-major token embeddings (row stride ldE floats, >= H*512)
 * Wa,Wb  [H,512,512] torch Linear layout [out,in];  ba,bb [H,512];  wc [H,512];  bc [H]
 * scores [T,H]       raw (pre-softmax) attention, == raw_attention of Model.py:406-411
 * act_a, act_b [T,H,512]  tanh / sigmoid activations BEFORE dropout, saved for backward (may be NULL
 *                    when no backward is needed: inference)
 * dropout: p_drop in [0,1). keep_a/keep_b (uint8 [T,H,512], 1 = keep) inject explicit masks (parity
 *          tests); when NULL and p_drop > 0 a counter-based hash of (seed, element index) decides.
 *          p_drop == 0 (module.eval()) => identity.
 * ws     workspace of mdl_abmil_gate_fwd_ws_bytes(T,H) bytes (K-major copy of the weights + score partials per
 *        128-wide j tile).
 */
int64_t mdl_abmil_gate_fwd_ws_bytes(int64_t T, int H);
int mdl_abmil_gate_fwd(const float* E, int64_t ldE, const float* Wa, const float* ba, const float* Wb,
                       const float* bb, const float* wc, const float* bc, float* scores, float* act_a,
                       float* act_b, int64_t T, int H, float p_drop, uint64_t seed,
                       const uint8_t* keep_a, const uint8_t* keep_b, void* ws, void* stream);

/* Backward of the above.  d_scores [T,H] incoming gradient.
 * dE [T,H*512] (row stride ldE): gradient wrt the token embeddings through the gates; written when
 *    accumulate == 0, added to the existing contents when accumulate != 0.
 * dWa,dWb [H,512,512]; dba,dbb,dwc [H,512]; dbc [H] (may be NULL)  -- all overwritten.
 * ws: mdl_abmil_gate_bwd_ws_bytes(T,H) bytes: d(za)|d(zb) [T+16,H,1024] (computed once, then both GEMMs are pure
 *     LDS-DMA contractions) + split-K slabs + column-sum partials. */
int64_t mdl_abmil_gate_bwd_ws_bytes(int64_t T, int H);
int mdl_abmil_gate_bwd(const float* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                       const float* act_a, const float* act_b, const float* d_scores, float* dE,
                       int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc,
                       float* dbc, int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                       const uint8_t* keep_b, void* ws, void* stream);

/* Materialises the keep-mask the two calls above derive from (seed, p_drop) when keep_a/keep_b are NULL:
 * keep uint8 [T,H,512], which = 0 (tanh branch) or 1 (sigmoid branch).  For reproducibility / tests. */
int mdl_abmil_gate_dropout_mask(uint8_t* keep, int64_t T, int H, int which, float p_drop, uint64_t seed,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * A3 -- softmax over patches + weighted pooling.  Replaces F.softmax(A, dim=1) (abmil.py:55) and
 * `(embeddings * attention).sum(dim=1)` (Model.py:416-417) without materialising E*attention.
 *
 *   pooled[b,c,e] = sum_t softmax_t(scores[b,:,c])[t] * E[b,t,c,e]
 *
 * Bags: dense (cu_seqlens == NULL: bag b = tokens [b*N, (b+1)*N)) or ragged (cu_seqlens int64
 * [n_bags+1] device array of token offsets into the packed E / scores; max_len = longest bag).
 * An empty bag pools to 0.
 * E [T,H*512] (stride ldE), scores [T,H], pooled [n_bags,H*512],
 * stat_m, stat_l [n_bags,H]: softmax max / sum-of-exp per bag and head (saved for backward).
 * ws: mdl_abmil_pool_ws_bytes(n_bags,max_len,H).
 */
int64_t mdl_abmil_pool_ws_bytes(int64_t n_bags, int64_t max_len, int H);
int mdl_abmil_pool_fwd(const float* E, int64_t ldE, const float* scores, float* pooled, float* stat_m,
                       float* stat_l, int64_t n_bags, int64_t N, const int64_t* cu_seqlens,
                       int64_t max_len, int H, void* ws, void* stream);

/* Backward.  d_pooled [n_bags,H*512].  Outputs: dE [T,H*512] (stride ldE; written or accumulated as
 * above) and d_scores [T,H] (gradient wrt the RAW scores through the softmax; written, or added to
 * existing contents when accumulate_scores != 0 -- used when raw attention also feeds a loss). */
int mdl_abmil_pool_bwd(const float* E, int64_t ldE, const float* scores, const float* pooled,
                       const float* stat_m, const float* stat_l, const float* d_pooled, float* dE,
                       int accumulate, float* d_scores, int accumulate_scores, int64_t n_bags, int64_t N,
                       const int64_t* cu_seqlens, int64_t max_len, int H, void* stream);

/* Weighted pooling WITHOUT the softmax: pooled[b,c,:] = sum_t weights[t,c] E[t,c,:] -- the 'relu' / 'leaky_relu' / 'sigmoid'
 * attention activations of BatchedABMIL (madeleine/models/abmil.py:56-61) pooled as Model.py:416-417 does.  weights [T,H] are the
 * ACTIVATED scores (the elementwise activation and its derivative stay with the caller); same geometry arguments and workspace
 * as mdl_abmil_pool_fwd; scratch_m / scratch_l [n_bags,H] are overwritten.  Backward: dE[t,c,:] (+)= weights[t,c] d_pooled[b,c,:],
 * d_weights[t,c] = <E[t,c,:], d_pooled[b,c,:]>. */
int mdl_abmil_wpool_fwd(const float* E, int64_t ldE, const float* weights, float* pooled, float* scratch_m, float* scratch_l,
                        int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H, void* ws, void* stream);
int mdl_abmil_wpool_bwd(const float* E, int64_t ldE, const float* weights, const float* d_pooled, float* dE, int accumulate,
                        float* d_weights, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H, void* stream);
int mdl_abmil_wpool_fwd_bf16(const uint16_t* E, int64_t ldE, const float* weights, float* pooled, float* scratch_m, float* scratch_l,
                             int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H, void* ws, void* stream);
int mdl_abmil_wpool_bwd_bf16(const uint16_t* E, int64_t ldE, const float* weights, const float* d_pooled, uint16_t* dE, int accumulate,
                             float* d_weights, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H,
                             void* stream);

/* Views: the intra-modality path pools two random half-bags per bag with the raw scores
 * re-softmaxed over the subset (Model.py:419-440).  A view is a DENSE bag of N tokens restricted to the index list
 * token_idx int32 [n_idx] (the same list for every bag: logical token i of bag b is row b*N + token_idx[i]); the kernels
 * gather the 8-KiB token rows in place -- no index_select copies of E.  ws: mdl_abmil_pool_ws_bytes(n_bags, n_idx, H).
 * The backward ACCUMULATES into dE (rows of the view) and/or d_scores; either may be NULL: d_scores == NULL is a dE-only pass
 * that does not read E (dE[t,c,:] += w[t,c] d_pooled[b,c,:]). */
int mdl_abmil_pool_view_fwd(const float* E, int64_t ldE, const float* scores, float* pooled, float* stat_m, float* stat_l,
                            int64_t n_bags, int64_t N, const int32_t* token_idx, int64_t n_idx, int H, void* ws, void* stream);
int mdl_abmil_pool_view_bwd(const float* E, int64_t ldE, const float* scores, const float* pooled, const float* stat_m,
                            const float* stat_l, const float* d_pooled, float* dE, float* d_scores, int64_t n_bags, int64_t N,
                            const int32_t* token_idx, int64_t n_idx, int H, void* stream);

/* Fused backward of A2 + A3 ("abmil_attnpool_bwd").  Call sequence:
 *   1. mdl_abmil_pool_bwd(..., dE = NULL, ...)   -> d_scores only (one read of E; dE may be NULL in that call)
 *   2. mdl_abmil_attnpool_bwd(...)               -> dE, dWa, dWb, dba, dbb, dwc, dbc
 * Step 2 = mdl_abmil_gate_bwd whose dX epilogue adds the pooling term  w[t,c] * d_pooled[bag(t), c, :]
 * (w = exp(scores - stat_m) / stat_l) while writing dE once -- instead of the pooling backward writing dE and the gate
 * backward reading it back to accumulate (saves one write + one read of |E|).  row_bag int32 [T] gives the bag of
 * every token row (ragged bags); row_bag == NULL means dense bags of N tokens (bag = t / N).  ws as mdl_abmil_gate_bwd.
 * accumulate != 0: dE already holds another consumer's gradient of E (the token_projector's dX, Model.py:140) and the epilogue
 * adds to it -- the read-modify-write rides under the MFMA-bound contraction instead of a separate 3 x |E| add pass. */
int mdl_abmil_attnpool_bwd(const float* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                           const float* act_a, const float* act_b, const float* d_scores, float* dE, int accumulate,
                           float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc, int64_t T, int H, float p_drop,
                           uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b, const float* scores,
                           const float* stat_m, const float* stat_l, const float* d_pooled, const int32_t* row_bag,
                           int64_t N, void* ws, void* stream);

/* Profiling form of mdl_abmil_attnpool_bwd: `phases` bit 0 = the HBM-bound dz pass (+ the bias / wc column-sum reduction), bit 1 = the
 * MFMA-bound dX / dW contractions (+ the dW slab reduction); 3 = what mdl_abmil_attnpool_bwd runs.  Calling it with 1 and then 2 on
 * the same arguments and workspace gives the same results and lets the caller time the two halves with its own events (bench.py's
 * roofline_mfma counts the contractions only). */
int mdl_abmil_attnpool_bwd_phases(const float* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                  const float* act_a, const float* act_b, const float* d_scores, float* dE, int accumulate,
                                  float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc, int64_t T, int H,
                                  float p_drop, uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b,
                                  const float* scores, const float* stat_m, const float* stat_l, const float* d_pooled,
                                  const int32_t* row_bag, int64_t N, void* ws, void* stream, int phases);
int mdl_abmil_attnpool_bwd_phases_bf16(const uint16_t* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                       const uint16_t* act_a, const uint16_t* act_b, const float* d_scores, uint16_t* dE,
                                       int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc,
                                       int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b,
                                       const float* scores, const float* stat_m, const float* stat_l, const float* d_pooled,
                                       const int32_t* row_bag, int64_t N, void* ws, void* stream, int phases);

/* ------------------------------------------------------------------------------------------------
 * N1 -- fused LayerNorm -> GELU(erf) -> Dropout of the pre-attention MLP.  Replaces the
 * nn.LayerNorm / nn.GELU / nn.Dropout(0.1) triple that follows each Linear (madeleine/models/Model.py:352-354,
 * :356-358, :360-362) and its autograd, in one HBM pass each way.
 * x, y, dy, dx [rows, W] contiguous; W in {256, 512, 2048}; gamma, beta [W]; mean, rstd [rows] (saved for
 * backward); LayerNorm: biased variance, rstd = 1/sqrt(var + eps).  Dropout as in the gate kernels: keep (uint8
 * [rows,W]) injects a mask, else a counter hash of (seed, element index) -- regenerated in backward.
 * dgamma, dbeta [W] are overwritten.  ws: mdl_ln_gelu_drop_bwd_ws_bytes(rows, W).
 * bias [W] (may be NULL): the bias of the preceding Linear, added to x before the LayerNorm (y = f(x + bias)) so the
 * GEMM runs bias-free; dbias [W] (may be NULL) receives its gradient = the column sums of dx, saving the separate
 * reduction pass over dx that autograd's Linear backward would launch.
 */
int mdl_ln_gelu_drop_fwd(const float* x, const float* bias, const float* gamma, const float* beta, float* y,
                         float* mean, float* rstd, int64_t rows, int W, float eps, float p_drop, uint64_t seed,
                         const uint8_t* keep, void* stream);
int64_t mdl_ln_gelu_drop_bwd_ws_bytes(int64_t rows, int W);
int mdl_ln_gelu_drop_bwd(const float* x, const float* bias, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, const float* dy, float* dx, float* dgamma, float* dbeta, float* dbias,
                         int64_t rows, int W, float p_drop, uint64_t seed, const uint8_t* keep, void* ws,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * L1 -- InfoNCE with in-batch negatives.  Replaces InfoNCE.info_nce, negative_keys=None branch
 * (madeleine/utils/loss.py:92,111-127) for S independent problems in one launch (the reference
 * calls it once per stain from trainer.py:33).
 *
 * Q,P [S,Kmax,D] padded row blocks (problem s uses rows [0,cnt[s])); cnt int32 [S] device array.
 * loss [S]: mean-reduced cross entropy of Q_hat P_hat^T / temperature against the diagonal
 *           (symmetric != 0: 0.5*rows + 0.5*columns, loss.py:120-123).  cnt[s] == 0 => loss 0.
 * ws: mdl_infonce_ws_bytes(S,Kmax,D): normalised rows, inverse norms, logits, LSEs (kept for bwd).
 */
int64_t mdl_infonce_ws_bytes(int S, int Kmax, int D);
/* row_loss (NULL, or [S,Kmax]): the per-sample losses of reduction='none' (loss.py:58 -> F.cross_entropy(..., reduction); symmetric:
 * 0.5 CE_i + 0.5 CE'_i); rows >= cnt[s] are zeroed. */
int mdl_infonce_fwd(const float* Q, const float* P, const int32_t* cnt, float* loss, float* row_loss, int S, int Kmax,
                    int D, float temperature, int symmetric, void* ws, void* stream);
/* Q,P: the forward's inputs (the fused path keeps no normalised copies).  d_loss [S] incoming gradient of the mean losses -- or, when
 * d_row_loss [S,Kmax] != NULL, the gradients of the per-sample losses (d_loss is then ignored and may be NULL); dQ,dP [S,Kmax,D] (rows
 * >= cnt[s] are zeroed).  ws must be the forward's.
 * Default: staged launches (normalise | similarity | log-sum-exp | loss ; coefficients | products | normalize backward), all stains per
 * launch.  MADELEINE_INFONCE_FUSED=1 with Kmax <= 256 and D <= 512: ONE launch forward and ONE backward producing the same bits
 * (measured slower on MI355X: the problem is latency-bound and fusing gives up wave-level parallelism; csrc/infonce.hip). */
int mdl_infonce_bwd(const float* Q, const float* P, const float* d_loss, const float* d_row_loss, const int32_t* cnt, float* dQ,
                    float* dP, int S, int Kmax, int D, float temperature, int symmetric, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N1 -- every Linear of the encoder as an exact-fp32 contraction in hand-written kernels: the three
 * Linears of the pre-attention MLP (madeleine/models/Model.py:351, :355, :359; bias = NULL there: the bias and its gradient
 * are handled by mdl_ln_gelu_drop_*), the token_projector Linear(2048, 128) (Model.py:140) and the slide projector
 * Linear(2048, 512) on the pooled embeddings (Model.py:145).
 * X [T,K] (row stride ldx), W [N,K] contiguous (torch's Linear.weight), bias [N] or NULL, Y [T,N] (row stride ldy).
 * Supported (MDL_E_UNSUPPORTED otherwise):
 *   T > 256 : N % 256 == 0 with K % 32 == 0 -- matrix-core tile engine, 128 x 256 tile; dX (output width K) runs the
 *             ragged-column variant of the tile when K % 256 != 0 (config 5: K = 768 + 32 stain channels);
 *             N % 256 == 128 with K % 256 == 0 -- the 256 x 128 "tall" geometry (token_projector);
 *   T <= 256: K % 4 == 0, N % 4 == 0 -- LDS-tiled fp32 FMA kernel (the slide projector's 64 rows).
 *   mdl_linear_fwd : Y = X W^T (+ bias)
 *   mdl_linear_bwd : dW [N,K] = dY^T X  (always);  dX [T,K] = dY W  when dX != NULL;  dbias [N] = column sums of dY when != NULL
 */
int64_t mdl_linear_fwd_ws_bytes(int64_t T, int N, int K);
int mdl_linear_fwd(const float* X, int64_t ldx, const float* W, const float* bias, float* Y, int64_t ldy, int64_t T, int N, int K,
                   void* ws, void* stream);
int64_t mdl_linear_bwd_ws_bytes(int64_t T, int N, int K);
int mdl_linear_bwd(const float* X, int64_t ldx, const float* W, const float* dY, int64_t ldy, float* dX, int64_t lddx,
                   float* dW, float* dbias, int64_t T, int N, int K, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * G0-G3 -- Graph Optimal Transport.  Replaces GOT (madeleine/utils/loss.py:278-302) =
 * cost_matrix_batch_torch (:162-176) + global-threshold ReLU (:288-292) + IPOT Wasserstein
 * (:179-207, 30 iterations, beta .5) + Gromov-Wasserstein (:236-275, 5 x 20 IPOT iterations,
 * beta .1) with cos_batch_torch intra costs (:210-233).  Called from trainer.py:42-45.
 *
 * V,Q [k,n,d]: the (already sub-sampled, loss.py:281-284) token sets of k cases; n <= 512, d <= 128
 * (n <= 256 -- what trainer.py:44 produces -- keeps the transport plans in registers / one matrix-core panel; 256 < n <= 512 keeps
 * them in the workspace and walks 256 x 256 output blocks).
 * out [2]: out[0] = sum_b WD_b, out[1] = sum_b GWD_b   (GOT returns out[1] + out[0], a SUM over cases).
 * Thresholds thr = min + .1 (max - min): the reference takes min/max over the WHOLE batch tensor for each of
 * the three cost tensors (cross, intra-V, intra-Q).  minmax_out float[6] (may be NULL) receives this call's
 * extrema (cross min,max | Cs min,max | Ct min,max).  When minmax_in (float[6], device) is non-NULL those
 * values are used INSTEAD -- the data-parallel driver all-gathers per-rank extrema so that every rank
 * thresholds with the global-batch values nn.DataParallel would see.
 * ws: mdl_got_ws_bytes(k,n,d) bytes: cost matrices + the per-iteration plans T_t and scaling vectors that
 * the reverse sweep replays (the same tensors the reference's autograd tape holds).  The backward calls
 * must receive the forward's workspace untouched.
 * ABI 23: for 128 < n <= 256 and 2 * (cases of the launch) <= compute units of the device, every IPOT sweep runs as TWO workgroups per
 * case and branch (row halves) that exchange their column sums once per iteration through 8-byte {value, tag} granules in the workspace
 * (device-scope stores / polls; csrc/got_impl.inc, Xch); MADELEINE_GOT_NOSPLIT=1 keeps the one-workgroup sweeps.  Same results up to the
 * association of the column sums; bit-reproducible.  The workspace's global region ends with four floats {pass generation (uint32 bits),
 * exchange time-out flag (0 = none; non-zero voids the pass), 0, 0}, followed by the 64 bytes of padding mdl_got_ws_bytes adds.
 * A voided pass says so in its results: out[0..1] of the forward and dV / dQ of the backward are NaN when the flag is up (a sweep waited
 * ~1 s for a partner workgroup that never became resident -- the device was shared with work the launch did not know about, e.g. a
 * second training process; run those with MADELEINE_GOT_NOSPLIT=1).  The flag is cleared by the extrema kernel of every forward.
 */
int64_t mdl_got_ws_bytes(int k, int n, int d);
int mdl_got_fwd(const float* V, const float* Q, float* out, float* minmax_out, const float* minmax_in,
                int k, int n, int d, void* ws, void* stream);
/* Only the first stage of the forward (normalise, raw costs, extrema): minmax_out [6] for this batch.  The
 * data-parallel driver calls it on every rank, reduces min/max over ranks, then runs mdl_got_fwd with minmax_in. */
int mdl_got_extrema(const float* V, const float* Q, float* minmax_out, int k, int n, int d, void* ws, void* stream);
/* Backward: d_out [2] = incoming gradients of (WD sum, GWD sum); dV,dQ [k,n,d] are written.  Gradient flows
 * through every unrolled IPOT iteration, the GW outer loop, the ReLU masks and the threshold extrema (to the
 * arg-min / arg-max elements, split evenly over ties like torch's min()/max() backward). */
int mdl_got_bwd(const float* V, const float* Q, const float* d_out, float* dV, float* dQ, int k, int n, int d,
                void* ws, void* stream);
/* The same in two steps, for the data-parallel path: _begin runs the reverse sweeps and reports the gradient
 * wrt the six threshold extrema in d_minmax [6] (may be NULL); the caller sums it over ranks; _finish routes
 * d_minmax_total [6] (NULL = this call's own) to the local elements that attain the extrema and writes dV,dQ. */
int mdl_got_bwd_begin(const float* d_out, float* d_minmax, int k, int n, int d, void* ws, void* stream);
int mdl_got_bwd_finish(const float* V, const float* Q, float* dV, float* dQ, const float* d_minmax_total,
                       int k, int n, int d, void* ws, void* stream);
/* Several GOT problems in ONE launch sequence (round 4).  The reference calls GOT once per stain (trainer.py:40-49); a data-parallel
 * rank used to run the stains' chains on one HIP stream each, and a process has four hardware queues: the first stream that is not
 * ours (a prefetcher's, RCCL's) made two chains share a queue and run one after the other.  Here every launch covers all problems
 * (a workgroup finds its problem from blockIdx), on the caller's stream alone.  1 <= np <= MDL_GOT_MAX_BATCH, every problem
 * non-empty (k >= 1, n >= 1) with n <= 256 (MDL_E_UNSUPPORTED above), one d for all; arrays of np entries: V[p], Q[p] [k[p], n[p], d],
 * ws[p] = mdl_got_ws_bytes(k[p], n[p], d) kept from _fwd_multi to _bwd_finish_multi, out[p] [2], minmax / d_minmax entries [6].
 * The kernels of the largest problem's size class run every problem: results of a problem equal its single-problem call bit for
 * bit when it is in that class itself, and to fp32 rounding otherwise (another reduction order). */
#define MDL_GOT_MAX_BATCH 4
int mdl_got_extrema_multi(int np, const float* const* V, const float* const* Q, float* const* minmax_out, const int* k, const int* n,
                          int d, void* const* ws, void* stream);
int mdl_got_fwd_multi(int np, const float* const* V, const float* const* Q, float* const* out, const float* const* minmax_in,
                      const int* k, const int* n, int d, void* const* ws, void* stream);
int mdl_got_bwd_begin_multi(int np, const float* const* d_out, float* const* d_minmax, const int* k, const int* n, int d,
                            void* const* ws, void* stream);
int mdl_got_bwd_finish_multi(int np, const float* const* V, const float* const* Q, float* const* dV, float* const* dQ,
                             const float* const* d_minmax_total, const int* k, const int* n, int d, void* const* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 mode -- the reference's `precision: bfloat16` runs (torch autocast around the forward,
 * madeleine/utils/trainer.py:101-103, scripts/launch_pretrain_withStainEncodings.sh): activations are STORED in
 * bf16 (uint16_t bit patterns of torch.bfloat16: E, the gate activations, dE, the LayerNorm input/output), the
 * contractions run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, every epilogue / reduction / statistic is
 * fp32, parameters and their gradients stay fp32.  Same argument meaning as the fp32 entry points above; ldE counts
 * bf16 elements and must be a multiple of 8.  mdl_abmil_gate_bwd_bf16 requires ldE == H*512 (it transposes E).
 * The fp32 entry points remain the parity path (1e-3 rel of the reference's fp32 results); this mode is held to
 * the reference-under-autocast accuracy (tests/test_bf16_gpu.py).
 */
int mdl_ln_gelu_drop_fwd_bf16(const uint16_t* x, const float* bias, const float* gamma, const float* beta, uint16_t* y,
                              float* mean, float* rstd, int64_t rows, int W, float eps, float p_drop, uint64_t seed,
                              const uint8_t* keep, void* stream);
int mdl_ln_gelu_drop_bwd_bf16(const uint16_t* x, const float* bias, const float* gamma, const float* beta, const float* mean,
                              const float* rstd, const uint16_t* dy, uint16_t* dx, float* dgamma, float* dbeta, float* dbias,
                              int64_t rows, int W, float p_drop, uint64_t seed, const uint8_t* keep, void* ws, void* stream);
/* Grouped LayerNorm-GELU-Dropout (ABI 24) for the engines whose GEMM epilogue adds no bias (exact fp32: no suffix; bf16): rows
 * [cu_groups[g], cu_groups[g + 1]) of group g (int64 [G + 1] on the device, G <= 2048, rows of a group contiguous) take the bias row
 * group_bias[g][W] -- the preceding Linear's bias + the bag's stain-encoding row times the encoding columns of the first weight
 * (Model.py:125-132, :351: [x | e_g] W^T = x Wx^T + e_g We^T), so the [T, D + 32] concatenation never exists.  The backward returns
 * dgroup_bias [G][W] = per-group column sums of dx (merged in a fixed order).  W in {256, 512, 1024}.
 * ws of the backward: mdl_ln_gelu_drop_bwd_groups_ws_bytes(rows, W, G). */
int mdl_ln_gelu_drop_fwd_groups(const float* x, const float* group_bias, const float* gamma, const float* beta, float* y, float* mean,
                                float* rstd, int64_t rows, int W, float eps, float p_drop, uint64_t seed, const uint8_t* keep,
                                const int64_t* cu_groups, int G, void* stream);
int mdl_ln_gelu_drop_fwd_groups_bf16(const uint16_t* x, const float* group_bias, const float* gamma, const float* beta, uint16_t* y,
                                     float* mean, float* rstd, int64_t rows, int W, float eps, float p_drop, uint64_t seed,
                                     const uint8_t* keep, const int64_t* cu_groups, int G, void* stream);
int mdl_ln_gelu_drop_bwd_groups(const float* x, const float* group_bias, const float* gamma, const float* beta, const float* mean,
                                const float* rstd, const float* dy, float* dx, float* dgamma, float* dbeta, float* dgroup_bias,
                                int64_t rows, int W, float p_drop, uint64_t seed, const uint8_t* keep, const int64_t* cu_groups, int G,
                                void* ws, void* stream);
int mdl_ln_gelu_drop_bwd_groups_bf16(const uint16_t* x, const float* group_bias, const float* gamma, const float* beta,
                                     const float* mean, const float* rstd, const uint16_t* dy, uint16_t* dx, float* dgamma,
                                     float* dbeta, float* dgroup_bias, int64_t rows, int W, float p_drop, uint64_t seed,
                                     const uint8_t* keep, const int64_t* cu_groups, int G, void* ws, void* stream);
int mdl_abmil_pool_fwd_bf16(const uint16_t* E, int64_t ldE, const float* scores, float* pooled, float* stat_m,
                            float* stat_l, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H,
                            void* ws, void* stream);
int mdl_abmil_pool_bwd_bf16(const uint16_t* E, int64_t ldE, const float* scores, const float* pooled, const float* stat_m,
                            const float* stat_l, const float* d_pooled, uint16_t* dE, int accumulate, float* d_scores,
                            int accumulate_scores, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len,
                            int H, void* stream);
int mdl_abmil_pool_view_fwd_bf16(const uint16_t* E, int64_t ldE, const float* scores, float* pooled, float* stat_m, float* stat_l,
                                 int64_t n_bags, int64_t N, const int32_t* token_idx, int64_t n_idx, int H, void* ws, void* stream);
int mdl_abmil_pool_view_bwd_bf16(const uint16_t* E, int64_t ldE, const float* scores, const float* pooled, const float* stat_m,
                                 const float* stat_l, const float* d_pooled, uint16_t* dE, float* d_scores, int64_t n_bags,
                                 int64_t N, const int32_t* token_idx, int64_t n_idx, int H, void* stream);
/* N1 Linears in the bf16 mode (madeleine/models/Model.py:351, :355, :359 pre-attention MLP, :140 token_projector under
 * `precision: bfloat16`, trainer.py:101-103): X, Y, dY, dX bf16; W [N,K], bias [N], dW, dbias fp32.  v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation; dW through the ds_read_b64_tr_b16 "TN" engine (no transposed copies of X / dY).
 * Supported (mdl_linear_bf16_supported; MDL_E_UNSUPPORTED otherwise): N % 128 == 0 and K % 32 == 0 (forward and
 * backward).  Leading dimensions multiples of 8.  Same argument meaning as mdl_linear_fwd / mdl_linear_bwd. */
int mdl_linear_bf16_supported(int64_t N, int64_t K, int backward);
int64_t mdl_linear_fwd_bf16_ws_bytes(int64_t T, int64_t N, int64_t K);
int mdl_linear_fwd_bf16(const uint16_t* X, int64_t ldx, const float* W, const float* bias, uint16_t* Y, int64_t ldy, int64_t T,
                        int64_t N, int64_t K, void* ws, void* stream);
int64_t mdl_linear_bwd_bf16_ws_bytes(int64_t T, int64_t N, int64_t K);
int mdl_linear_bwd_bf16(const uint16_t* X, int64_t ldx, const float* W, const uint16_t* dY, int64_t lddy, uint16_t* dX, int64_t lddx,
                        float* dW, float* dbias, int64_t T, int64_t N, int64_t K, void* ws, void* stream);

int64_t mdl_abmil_gate_fwd_bf16_ws_bytes(int64_t T, int H);
int mdl_abmil_gate_fwd_bf16(const uint16_t* E, int64_t ldE, const float* Wa, const float* ba, const float* Wb,
                            const float* bb, const float* wc, const float* bc, float* scores, uint16_t* act_a,
                            uint16_t* act_b, int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                            const uint8_t* keep_b, void* ws, void* stream);
int64_t mdl_abmil_gate_bwd_bf16_ws_bytes(int64_t T, int H);
int mdl_abmil_gate_bwd_bf16(const uint16_t* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                            const uint16_t* act_a, const uint16_t* act_b, const float* d_scores, uint16_t* dE,
                            int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc,
                            int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b,
                            void* ws, void* stream);
int mdl_abmil_attnpool_bwd_bf16(const uint16_t* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                const uint16_t* act_a, const uint16_t* act_b, const float* d_scores, uint16_t* dE,
                                int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc, int64_t T,
                                int H, float p_drop, uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b,
                                const float* scores, const float* stat_m, const float* stat_l, const float* d_pooled,
                                const int32_t* row_bag, int64_t N, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Split-fp16 contraction engine (round 3; csrc/split_engine.hpp): the fp32 contractions of the path -- nn.Linear of
 * madeleine/models/Model.py:351, :355, :359, :140 and the gate products of madeleine/models/abmil.py:49-52 -- with fp32-level accuracy
 * on v_mfma_f32_32x32x16_f16: every operand value as two fp16 planes hi + lo of (power-of-two scale) x value, every product as
 * ah bh + ah bl + al bh accumulated in fp32 (measured error below a plain fp32 fmaf chain, tools/micro/split_lab.hip).
 * SPLIT IMAGE of X [rows, K], K % 32 == 0: uint16 [rows][K / 32][2][32] -- the hi and the lo plane of every 32-column block side
 * by side (128 B = the bytes of the fp32 block); row stride rsb bytes (>= 4 K, % 16 == 0).  scale: device float[2] = {scale, absmax}.
 *   mdl_split_image   : image of a fp32 tensor (exact absmax -> scale with max |scale x| in [2^13, 2^14)), + pad_rows zero rows
 *   mdl_split_gemm_nt : C [M,N] (+)= sum_k A[m][k] B[n][k] (+ bias[n])   A, B images with K columns (rows m, n)
 *   mdl_split_gemm_tn : out [N][Mi] = sum_t B[t][n] A[t][m]              A, B images with T rows (contraction = rows); B must be
 *                       followed by >= 32 all-zero rows; ws = mdl_split_gemm_tn_ws_bytes (token-split slabs, reduced in fixed order)
 */
/* A3 on the image of E (the split GEMM mode keeps E as the image its LayerNorm kernel wrote, nothing else): mdl_abmil_pool_fwd with
 * E_img = split image [T][H*512] (row stride e_rsb bytes, scale e_scale[0]); mdl_abmil_pool_dscores_img = the score gradients of
 * mdl_abmil_pool_bwd (d_scores (+)= ...; the dE term is part of mdl_abmil_attnpool_bwd_split's dX epilogue). */
int mdl_abmil_pool_fwd_img(const void* E_img, int64_t e_rsb, const float* e_scale, const float* scores, float* pooled, float* stat_m,
                           float* stat_l, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H, void* ws,
                           void* stream);
int mdl_abmil_pool_dscores_img(const void* E_img, int64_t e_rsb, const float* e_scale, const float* scores, const float* pooled,
                               const float* stat_m, const float* stat_l, const float* d_pooled, float* d_scores, int accumulate_scores,
                               int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H, void* stream);
/* Dispatch timer of the A3 forward (measurement only; bench.py's roofline.achieved).  mdl_pool_timer_arm(slot), 0 <= slot < 64 (negative:
 * disarm): the NEXT mdl_abmil_pool_fwd / _bf16 / _img call on this process launches its two kernels through hipExtLaunchKernel with
 * start / stop events on the caller's stream, i.e. the begin / end timestamps of the dispatches themselves (what rocprofv3
 * --kernel-trace reports) instead of the distance between two markers in a busy stream.  Nothing waits at launch;
 * mdl_pool_timer_read(slot, ms[3]) blocks until that slot's launches finished and returns {pool_partial, pool_combine, start of the
 * first to end of the second} in milliseconds (MDL_E_ARG for a slot that was never used).  Process-wide state, one training thread. */
int mdl_pool_timer_arm(int slot);
int mdl_pool_timer_read(int slot, float* ms);
/* A HIP stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask): mask = n_words x 32 bits, bit i = CU i may run
 * this stream's kernels.  For partitioning the chip between a matrix-core-bound and an HBM-bound chain that run concurrently (the dW
 * products of the backward beside the LayerNorm / dz passes).  *stream_out receives the hipStream_t; mdl_stream_destroy releases it. */
int mdl_stream_create_cu_mask(uint32_t n_words, const uint32_t* mask, void** stream_out);
int mdl_stream_destroy(void* stream);
int mdl_split_image(const float* X, int64_t ldx, int64_t rows, int K, void* img, int64_t rsb, int64_t pad_rows, float* scale,
                    void* stream);
/* ROW-SCALED image (round 4): row r of X is scaled by its own power of two s_r (max_k |s_r X[r][k]| in [2^13, 2^14); an all-zero
 * row has no scale: its image row is zero and row_inv = 0 -- do NOT divide by row_inv) -- for the tensor whose rows a CALLER controls, the patch features (Model.py:113, :351): with one common scale a patch 2^20 times
 * larger than the others would cost them their low bits (fp32 nn.Linear has no such coupling between rows).  row_inv (device
 * float[rows]) receives 1 / s_r (0 for an all-zero row: it contributes nothing to any product), scale = {1, max |X|}; scale may be NULL
 * (no bookkeeping launches: the image is ONE kernel -- the weights' images, built every forward; pass a constant {1, .} as its scale).  Consumers: mdl_split_gemm_nt(A = this image, a_row_mul = row_inv) -- the row
 * factor is undone in the epilogue, exactly; mdl_split_gemm_tn pairs it with a gradient image written with row_mul = row_inv
 * (mdl_ln_gelu_drop_bwd_split), so that the row factors cancel inside the contraction over rows. */
int mdl_split_image_rows(const float* X, int64_t ldx, int64_t rows, int K, void* img, int64_t rsb, int64_t pad_rows, float* row_inv,
                         float* scale, void* stream);
/* row_gate (device float[ceil(M / 256)], may be NULL; only with accumulate != 0 and bias == NULL): output tiles whose entry is 0 are
 * skipped -- mdl_split_tile_absmax(X, ...) fills it with the per-256-row maxima of |X| for A = image(X), and chunk_max (float
 * [ceil(rows / 32)], may be NULL) with the per-32-row maxima that mdl_split_gemm_tn's b_chunk_max takes. */
int mdl_split_tile_absmax(const float* X, int64_t ldx, int64_t rows, int K, float* gate, float* chunk_max, void* stream);
/* terms (round 4; 3 everywhere by default, anything but 2 / 3 is MDL_E_ARG): 3 = ah bh + ah bl + al bh, the fp32-class product.  2 = the
 * operand that is NOT a gradient enters with its hi plane only (11 significant bits): mdl_split_gemm_nt drops ah bl (B = the weight of
 * dX = dY W), mdl_split_gemm_tn drops al bh (A = the activations of dW = dY^T X), mdl_abmil_attnpool_bwd_split does both for the gate's
 * dX / dW -- 4 instead of 6 MFMA sets per 32-k block, results at ~2^-12 relative.  Only ever selected for BACKWARD products, by
 * madeleine_amd.functional.set_gradient_terms(2); the forward (every value the losses see) always runs with 3.
 * a_row_mul (device float[M], may be NULL): row m of the product is multiplied by a_row_mul[m] before bias / accumulation (the
 * 1 / s_r of a row-scaled A image, or the s_r that undoes a row_mul folded into a gradient image).  b_col_mul (device float[N], N % 4
 * == 0, may be NULL): the same per output column = per row of a row-scaled B image (a weight matrix W [N, K]: nn.Linear's rows). */
int mdl_split_gemm_nt(const void* A, int64_t a_rsb, const float* a_scale, const void* B, int64_t b_rsb, const float* b_scale, float* C,
                      int64_t ldc, int64_t M, int N, int K, const float* bias, int accumulate, float* absmax_out, const float* row_gate,
                      const float* a_row_mul, const float* b_col_mul, int terms, void* stream);
/* (round 5) The same product with a bias row per GROUP of output rows: C[m][:] += group_bias[row_group[m]][:]; group_bias [G][N]
 * contiguous fp32, row_group device int32 [M] with values in [0, G).  MADELEINE's stain encoding (Model.py:125-132) concatenates ONE
 * 32-vector per bag to every patch feature before the first Linear (:351): [x | e_g] W^T = x Wx^T + e_g We^T, so the concat never has to
 * exist -- the bag's row e_g We^T enters here, and mdl_ln_gelu_drop_bwd_split_groups returns its gradient. */
int mdl_split_gemm_nt_group_bias(const void* A, int64_t a_rsb, const float* a_scale, const void* B, int64_t b_rsb, const float* b_scale,
                                 float* C, int64_t ldc, int64_t M, int N, int K, const float* bias, const float* a_row_mul,
                                 const float* b_col_mul, const float* group_bias, const int32_t* row_group, int terms, void* stream);
int64_t mdl_split_gemm_tn_ws_bytes(int64_t T, int Mi, int N);
/* b_chunk_max (float [ceil(T / 32)], may be NULL): per-32-row maxima of |X| for B = image(X) (mdl_split_tile_absmax) -- chunks whose
 * entry is 0 are skipped (the token_projector's dW: the local loss reads the first <= 256 tokens of a bag, the rest of d_tok is zero). */
int mdl_split_gemm_tn(const void* A, int64_t a_rsb, const float* a_scale, int Mi, const void* B, int64_t b_rsb, const float* b_scale,
                      int N, float* out, int64_t T, const float* b_chunk_max, void* ws, int terms, void* stream);

/* N1 producers that write split images directly (csrc/preattn_act.hip): mdl_ln_gelu_drop_fwd / _bwd with the output tensor as an
 * image -- the pre-attention activations and their gradients are consumed by contractions only, so no fp32 copy is written (the
 * forward also writes y when y != NULL: E, which the pooling kernels read).  Scales from rigorous bounds (no pass over the data):
 *   forward  |y|  <= (max|gamma| sqrt(W-1) + max|beta|) / (1-p)
 *   backward |dx| <= rstd_max 1.13 max|gamma| max|dy| (2 + sqrt(W)) / (1-p);  dy_absmax = device float with max |dy| (from the producing
 *            kernel's epilogue) or NULL (one extra pass over dy).  The dx image is followed by 32 zero rows; ws as mdl_ln_gelu_drop_bwd. */
/* row_mul / rstd_max (forward, both may be NULL): rstd_max (device float) receives max_r rstd[r] * row_mul[r] (row_mul NULL: max rstd) --
 * handed to the backward it replaces that call's pass over rstd. */
int mdl_ln_gelu_drop_fwd_split(const float* x, const float* bias, const float* gamma, const float* beta, float* y, void* img, float* scale,
                               float* mean, float* rstd, int64_t rows, int W, float eps, float p_drop, uint64_t seed, const uint8_t* keep,
                               const float* row_mul, float* rstd_max, void* stream);
/* row_mul (device float[rows], may be NULL): the dx image holds row_mul[r] dx[r][:] (bound through max_r rstd[r] row_mul[r]); dgamma,
 * dbeta, dbias are unaffected.  For the first pre_attn block, whose input image is row-scaled (mdl_split_image_rows): row_mul = row_inv.
 * rstd_max (device float, may be NULL): the forward's max_r rstd[r] * row_mul[r] for THE SAME row_mul; with it and dy_absmax the call
 * issues no bookkeeping launch but the bound kernel. */
int mdl_ln_gelu_drop_bwd_split(const float* x, const float* bias, const float* gamma, const float* beta, const float* mean,
                               const float* rstd, const float* dy, const float* dy_absmax, void* dx_img, float* dx_scale, float* dgamma,
                               float* dbeta, float* dbias, int64_t rows, int W, float p_drop, uint64_t seed, const uint8_t* keep,
                               const float* row_mul, const float* rstd_max, void* ws, void* stream);
/* (round 5) mdl_ln_gelu_drop_bwd_split for a pre-LN tensor whose rows carry a per-group bias row (see mdl_split_gemm_nt_group_bias): the
 * rows of a group are contiguous, cu_groups = device int64 [G + 1] row offsets, G <= 2048.  Additionally dgroup_bias [G][W] = the sum
 * of dx over the rows of each group.  ws: mdl_ln_gelu_drop_bwd_groups_ws_bytes(rows, W, G). */
int64_t mdl_ln_gelu_drop_bwd_groups_ws_bytes(int64_t rows, int W, int G);
int mdl_ln_gelu_drop_bwd_split_groups(const float* x, const float* bias, const float* gamma, const float* beta, const float* mean,
                                      const float* rstd, const float* dy, const float* dy_absmax, void* dx_img, float* dx_scale,
                                      float* dgamma, float* dbeta, float* dbias, int64_t rows, int W, float p_drop, uint64_t seed,
                                      const uint8_t* keep, const float* row_mul, const float* rstd_max, const int64_t* cu_groups, int G,
                                      float* dgroup_bias, void* ws, void* stream);

/* A2 on the split engine (csrc/abmil_gate_split.hip): mdl_abmil_gate_fwd / mdl_abmil_attnpool_bwd(_phases) with E given as a split
 * image (rows of e_rsb bytes holding the H*512 head-major channels, scale e_scale) -- everything else (parameters, scores, saved
 * activations, gradients, dropout, pooling term, `accumulate`) as in the fp32 entry points.  scores == NULL in the backward: no
 * pooling term (plain gate backward).  dE_absmax (device float, may be NULL, zeroed by the caller) is raised to max |dE|.
 * phases of mdl_abmil_attnpool_bwd_split (bit mask, 1 .. 15): 1 = the dz pass (+ bias / wc column sums), 2 = both contractions,
 * 4 = the dX contraction alone, 8 = the dW contraction alone (dWa, dWb) -- the two contractions only read what the dz pass wrote into
 * `ws`, so a caller may queue 8 on another stream behind an event recorded after 1 while 4 and the rest of the backward proceed
 * (measured null on MI355X; the Python mirror issues 3, functional.attnpool_bwd_split_raw(phases=...) issues any
 * sequence -- tests/test_split_gpu.py holds 1, 4, 8 == 3 bit for bit). */
int64_t mdl_abmil_gate_fwd_split_ws_bytes(int64_t T, int H);
int mdl_abmil_gate_fwd_split(const void* E_img, int64_t e_rsb, const float* e_scale, const float* Wa, const float* ba, const float* Wb,
                             const float* bb, const float* wc, const float* bc, float* scores, float* act_a, float* act_b, int64_t T,
                             int H, float p_drop, uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b, void* ws, void* stream);
int64_t mdl_abmil_gate_bwd_split_ws_bytes(int64_t T, int H);
int mdl_abmil_attnpool_bwd_split(const void* E_img, int64_t e_rsb, const float* e_scale, const float* Wa, const float* Wb, const float* wc,
                                 const float* act_a, const float* act_b, const float* d_scores, float* dE, int64_t ldE, int accumulate,
                                 float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc, int64_t T, int H, float p_drop,
                                 uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b, const float* scores, const float* stat_m,
                                 const float* stat_l, const float* d_pooled, const int32_t* row_bag, int64_t N, float* dE_absmax, void* ws,
                                 void* stream, int phases, int terms);

#ifdef __cplusplus
}
#endif
#endif /* MADELEINE_AMD_H */
