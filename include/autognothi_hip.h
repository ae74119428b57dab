/*
 * autognothi_hip.h — C ABI of the MI355X (gfx950) masked-forward / Shapley hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every device pointer
 * is caller-owned HBM (the Python host allocates it through torch), `stream` is a hipStream_t
 * passed as void* (0 = default stream); all calls are asynchronous on that stream unless noted.
 * Every function returns AG_OK (0) or a negative AG_ERR_* code; ag_last_error() gives the text.
 *
 * The reference (gszfwsb/AutoGnothi) is Python-only, so there is no pre-existing FFI: each entry
 * point below replaces the stock-PyTorch implementation of the reference symbol cited next to it
 * (file:line relative to the reference repo).  INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add to call them from recipes/<kind>.py / models/shapley.py.
 *
 * Storage dtypes (AG_BF16 / AG_F32) select what GEMM operands and inter-kernel activations are
 * stored in; accumulation, LayerNorm statistics and soft-max are fp32 in
 * both modes; the residual stream itself is kept in the storage dtype.  AG_F32 runs the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) and is the mode the
 * 1e-4 Shapley-value parity criterion is checked in; AG_BF16 is the throughput mode.
 */
#ifndef AUTOGNOTHI_HIP_H_
#define AUTOGNOTHI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AG_ABI_VERSION 6   /* 2: device-side row counts are an explicit `d_rows` argument (was a thread-local mode, ag_dynamic_rows);
                            * 3: ag_gemm_ex + the fused training kernels + ag_gemm_resid_split (additions only);
                            * 4: ag_gemm_ws (additions only);
                            * 5: ag_cls_last_attention_rows (the kernel-level entry of the K / V-free last layer), ag_gemm_ex_group,
                            *    ag_colsum_bf16_group (additions only);
                            * 6: ag_gemm_last_plan (a diagnostic; addition only) */

enum { AG_OK = 0, AG_ERR_INVALID = -1, AG_ERR_HIP = -2, AG_ERR_UNSUPPORTED = -3 };
enum { AG_F32 = 0, AG_BF16 = 1 };
/* attention mask semantics: ViT multiplies key logits by the 0/1 mask (reference
 * models/vanilla_vit.py:446-450); BERT adds (1-mask)*finfo(f32).min (models/vanilla_bert.py:520-523). */
enum { AG_MASK_VIT_MUL = 0, AG_MASK_BERT_ADD = 1 };
/* GEMM epilogues */
enum {
    AG_EPI_BIAS = 0,        /* C = A·Wᵀ + b                     -> storage dtype            */
    AG_EPI_BIAS_GELU = 1,   /* C = gelu_erf(A·Wᵀ + b)           -> storage dtype            */
    AG_EPI_BIAS_RESID = 2,  /* C = A·Wᵀ + b + R   (R, C: storage dtype = the residual stream) */
    AG_EPI_BIAS_F32 = 3,    /* C = A·Wᵀ + b                     -> fp32                     */
    AG_EPI_BIAS_TANH = 4,   /* C = tanh(A·Wᵀ + b)               -> storage dtype (pooler)   */
    AG_EPI_BIAS_GELU_ADD = 5 /* C = R + gelu_erf(A·Wᵀ + b)  (side-network ladder add, reference models/ltt_vit.py:431) */
};

int ag_abi_version(void);
const char* ag_last_error(void);
/* number of CUs / gfx arch of `device`; both out-params optional.  Synchronous. */
int ag_device_info(int device, int* cu_count, char* arch, size_t arch_len);

/* ------------------------------------------------------------------------------------------------
 * Device-resident MT19937 (the engine behind torch's CPU generator) and the mask samplers.
 * d_state: AG_MT_STATE_BYTES of HBM holding {mt[624], pos}.  The stream of 32-bit draws is
 * bit-identical to at::mt19937; fp32 uniforms are (x & 0xFFFFFF) * 2^-24, as torch.rand.
 * ---------------------------------------------------------------------------------------------- */
#define AG_MT_STATE_BYTES 2560 /* 624 words + pos, padded */
/* == torch.manual_seed(seed) on the CPU generator (reference utils/tools.py:33-43 set_seed). */
int ag_mt19937_seed(void* d_state, uint32_t seed, void* stream);
/* import / export the 624-word state + position (0..624; 624 = twist on next draw), e.g. from
 * torch.get_rng_state() so the device stream continues the host generator.  Export synchronises. */
int ag_mt19937_import(void* d_state, const uint32_t* h_mt624, int pos, void* stream);
int ag_mt19937_export(const void* d_state, uint32_t* h_mt624, int* pos, void* stream);
/* advance the state by n draws without producing them (a rank of a row-sharded run steps over the other ranks' draws and
 * stays on the one stream the unsharded reference loop consumes). */
int ag_mt19937_skip(void* d_state, int64_t n, void* stream);
/* raw draws (testing / other samplers): out[n] u32, advances the state. */
int ag_mt19937_raw(void* d_state, uint32_t* d_out, int64_t n, void* stream);

/* reference models/shapley.py:56-79 mask_shapley_new(n_mask_samples, n_players) (+ _torch_choice
 * :131-135).  d_prefix = the fp32 exclusive-prefix table of the size prior (P-1 entries, :65-67,:132).
 * Outputs (either may be NULL): d_mask_i64 [n, P] int64 0/1 — the reference's return value;
 * d_mask_bits [n, ceil((P+1)/32)] uint32 — the T-wide key mask with the always-on CLS bit 0
 * prepended (recipes/vanilla_vit.py:219-224 _fw_xs_preprocess), bit t of word t/32.
 * Consumes n/2*P + n/2 draws in the reference's order.  d_scratch: >= n/2*(P+1) uint32. */
int ag_mask_shapley_new(void* d_state, int n_mask_samples, int n_players, const float* d_prefix,
                        int64_t* d_mask_i64, uint32_t* d_mask_bits, uint32_t* d_scratch, void* stream);
/* Rows [row_lo, row_hi) (even bounds) of the call ag_mask_shapley_new(n_mask_samples_total, ...): what a rank of a row-sharded
 * run needs.  The draws of the other rows are stepped over (twists only), the state advances by the whole call, so every rank
 * of a run stays on the one stream of the unsharded reference loop and the union of the ranks' rows is bit-identical to the
 * single call.  Outputs hold (row_hi - row_lo) rows; d_scratch: >= (row_hi - row_lo)/2 * (P+1) uint32. */
int ag_mask_shapley_new_rows(void* d_state, int n_mask_samples_total, int row_lo, int row_hi, int n_players,
                             const float* d_prefix, int64_t* d_mask_i64, uint32_t* d_mask_bits, uint32_t* d_scratch,
                             void* stream);
/* reference models/shapley.py:109-115 mask_purely_uniform(batch, n_features).
 * d_scratch: >= batch*(P+1) uint32. */
int ag_mask_purely_uniform(void* d_state, int batch, int n_players, int64_t* d_mask_i64,
                           uint32_t* d_mask_bits, uint32_t* d_scratch, void* stream);
/* Rows [row_lo, row_hi) of the call ag_mask_purely_uniform(batch_total, ...) for a rank of a row-sharded run (the reference draws
 * rand(B,P) first and rand(B,1) after it: a rank's rows are two spans of the stream); the state advances by the whole call.
 * Outputs hold (row_hi - row_lo) rows; d_scratch: >= (row_hi - row_lo) * (P+1) uint32. */
int ag_mask_purely_uniform_rows(void* d_state, int batch_total, int row_lo, int row_hi, int n_players, int64_t* d_mask_i64,
                                uint32_t* d_mask_bits, uint32_t* d_scratch, void* stream);
/* recipes/<kind>.py _fw_xs_preprocess: int64 [R,P] 0/1 mask -> [R, ceil((P+1)/32)] bits with CLS prepended. */
int ag_pack_mask(const int64_t* d_mask_i64, int rows, int n_players, uint32_t* d_mask_bits, void* stream);
/* scripts/measure_faithfulness.py:225-251 _get_perturbed_samples for `n_attr` attribution vectors
 * at once: d_attr [n_attr, P] fp32; stops = linspace(0,P,steps) as int64; mask i flips the `stops[i]`
 * highest-attribution players of an all-`mask_base` row.  Ties: the reference ranks with np.argsort's default kind, which
 * is not stable — its order inside a group of equal attributions depends on the host's numpy build (fixture
 * tests/golden/perturbed_ties.npz) — so the result is only defined up to the choice inside a tie group; here the higher
 * index ranks first (= stable ascending argsort, reversed).  d_stops [steps] int64, d_mask_i64 [n_attr, steps, P]. */
int ag_perturbed_masks(const float* d_attr, int n_attr, int n_players, int steps, int mask_base,
                       int64_t* d_stops, int64_t* d_mask_i64, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Building-block kernels (also used one by one by the parity tests).
 * ---------------------------------------------------------------------------------------------- */
/* fp32 -> storage dtype conversion (weights packing). */
int ag_cast_f32(const float* d_src, void* d_dst, int64_t n, int dtype, void* stream);
/* Weight packing of a Linear that consumes a LayerNorm (LayerNorm fold of ag_gemm / ag_gemm_ws; reference models/vanilla_vit.py:369,373
 * feed LayerNorm outputs straight into Linear layers): w [N,K], b [N] (may be NULL), gamma / beta [K] fp32 ->
 *   w_out[n,k] = store(w[n,k] * gamma[k])  (storage dtype),  b_out[n] = b[n] + sum_k w[n,k] * beta[k],
 *   s_out[n] = sum_k w_out[n,k]  (of the ROUNDED weights: the mean term of the fold then cancels exactly).  One wave per row, fixed order. */
int ag_pack_folded_linear(const float* d_w, const float* d_b, const float* d_gamma, const float* d_beta, int N, int K,
                          void* d_w_out, int dtype, float* d_b_out, float* d_s_out, void* stream);

/* Device-side row counts (`d_rows`).  ag_gemm, ag_gemm_resid_ln, ag_layernorm, ag_gather_rows, ag_side_mlp, ag_side_linear and
 * ag_bert_layers_forward_packed take a `const int* d_rows` (device pointer, may be NULL).  NULL: the host-side row count (M / rows /
 * n / N) is exact.  Non-NULL: the host-side count is an UPPER BOUND that sizes the launch; the kernel reads the actual count from
 * d_rows[0] when it runs, builds its (XCD-aware) tile order from it, and surplus workgroups exit at once.  This is how the packed
 * token count of a pruned BERT forward stays on the device: no device->host read, no synchronisation, graph-capturable. */

/* torch.nn.LayerNorm over the last dim (reference call sites models/vanilla_vit.py:353-362,:205;
 * models/vanilla_bert.py:323,:559,:603).  x [rows, H] of dtype x_dtype (AG_F32 / AG_BF16) with row
 * stride ldx (elements); statistics in fp32; y_store (storage dtype, stride H) and/or y_f32
 * (stride H) may be NULL. */
int ag_layernorm(const void* d_x, int x_dtype, int64_t ldx, int rows, int H, const float* d_gamma, const float* d_beta,
                 float eps, void* d_y_store, float* d_y_f32, int dtype, const int* d_rows, void* stream);

/* C[M,N] = epilogue(A[M,K] · W[N,K]ᵀ + bias[N]) — every nn.Linear on the path (q/k/v fused into
 * one [3H,H] weight: models/vanilla_vit.py:422-424; :477; :491; :510).  A, W in storage dtype
 * (A row stride lda, W dense [N,K]); bias fp32.  For AG_EPI_BIAS_RESID, R is in the storage dtype with
 * row stride ldr and row index ((m / T) / resid_share) * T + m % T  (T = rows_per_seq; resid_share > 1 lets
 * the K masked copies of one input share the layer-0 residual).  K % 64 == 0 (bf16) / K % 32 == 0
 * (fp32) required; N, M arbitrary.
 * LayerNorm folding (optional; only where ag_gemm_supports_ln_fold() says so — the large-M bf16 kernel):
 *   row statistics are slab-major partial sums: stats[s][m] = (sum, sum of squares) of features [256 s, 256 s + 256)
 *   of row m, S = ceil(H / 256) slabs of 2·M floats (AG_ROW_STATS_FLOATS).  d_ln_stats [ceil(K/256), M, 2] describe
 *   the A rows, d_ln_colsum [N] = sum_k W[n,k]: the result becomes rstd[m]*(A·Wᵀ - mean[m]*colsum[n]) + bias[n],
 *   i.e. Linear(LayerNorm(A)) when W is pre-scaled by gamma and bias carries beta·Wᵀ (reference
 *   models/vanilla_vit.py:369,373 feed LayerNorm outputs straight into Linear layers).  d_stats_out
 *   [ceil(N/256), M, 2] receives the statistics of the (bf16-rounded) rows this call writes, for the next folded
 *   consumer: each element is written once by the tile that owns it (no zero fill needed, no atomics) and the consumer
 *   adds the slabs in a fixed order, so results are bit-reproducible from run to run. */
#define AG_ROW_STATS_FLOATS(M, H) ((size_t)(((H) + 255) / 256) * (size_t)(M) * 2)
int ag_gemm(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
            const void* d_R, int64_t ldr, int rows_per_seq, int resid_share,
            int M, int N, int K, int epilogue, int dtype,
            const float* d_ln_stats, const float* d_ln_colsum, float ln_eps, float* d_stats_out, const int* d_rows, void* stream);
int ag_gemm_supports_ln_fold(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, int epilogue, int dtype);
/* Post-LN residual without the LayerNorm pass (BERT, models/vanilla_bert.py:556-560 and :600-604: hidden = LayerNorm(dense(x) + input)
 * where `input` is itself the previous LayerNorm's output):  C = A·Wᵀ + bias + LayerNorm(Rpre)[m, n], bf16, large-M kernel only
 * (ag_gemm_resid_ln_supported).  Rpre [M, N] (row stride ldr) holds the PRE-LayerNorm rows the previous GEMM wrote, d_r_stats
 * [ceil(N/256), M, 2] their slab statistics (that GEMM's d_stats_out), ln_g / ln_b [N] the LayerNorm's parameters: the epilogue adds
 * (Rpre - mean[m]) * rstd[m] * ln_g[n] + ln_b[n] in fp32.  d_stats_out [ceil(N/256), M, 2] (required) receives the statistics of the
 * rows written — which are again pre-LN rows: their LayerNorm is folded into the next GEMMs (ag_gemm's d_ln_stats) and into the next
 * residual (this call).  Together: a post-LN block chain in which no LayerNorm output is ever written or read. */
int ag_gemm_resid_ln(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                     const void* d_Rpre, int64_t ldr, const float* d_r_stats, const float* d_ln_g, const float* d_ln_b, float ln_eps,
                     int M, int N, int K, float* d_stats_out, const int* d_rows, void* stream);
int ag_gemm_resid_ln_supported(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr);
/* ag_gemm with the bias + residual epilogue (identity residual rows: rows_per_seq = resid_share = 1; bf16) for launches whose LAST
 * round of 256 x 256 tiles leaves most CUs idle — the reference's own batch sizes: one input x K = 32 masks of ViT-base is 75 tiles on
 * 256 CUs, four inputs are one round and 41 tiles (`ViTOutput.dense` + residual, models/vanilla_vit.py:498-504; `BertOutput.dense`,
 * models/vanilla_bert.py:596-604).  The rows of the full rounds run as in ag_gemm; the rows of the tail round are computed as several
 * contraction ranges side by side in one launch (fp32 partial tiles in d_scratch) and finished by a row kernel that adds the ranges in
 * order, the bias and the residual, stores bf16 and writes the same slab statistics (d_stats_out, optional) as the GEMM epilogue.
 * Deterministic; equal to ag_gemm to fp32 rounding of the sums (another summation order).
 * ag_gemm_resid_split_scratch_bytes: bytes of d_scratch for this shape on the current device, 0 = the shape does not split (use ag_gemm). */
size_t ag_gemm_resid_split_scratch_bytes(int M, int N, int K);
int ag_gemm_resid_split(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc, const void* d_R,
                        int64_t ldr, int M, int N, int K, float* d_stats_out, void* d_scratch, size_t scratch_bytes, void* stream);
/* ag_gemm for launches that UNDER-FILL the chip (round 5): the masked forward at the reference's own batch sizes — one to four inputs x
 * K masks (experiments/vit_base_imagenette_vanilla/.hparams.json:53-54; scripts/measure_faithfulness.py:195-218 runs one image at a
 * time) — and the 8-64 masked rows per GPU of an 8-way shard of BASELINE configs 4 / 5: M = 1.5-12 k token rows, where the persistent
 * 256 x 256 kernel works in rounds of 256 tiles and a Linear of 75 or 300 tiles wastes most of a round.  Same contract as ag_gemm (bf16:
 * AG_EPI_BIAS / AG_EPI_BIAS_GELU with optional LayerNorm fold, AG_EPI_BIAS_RESID with optional row statistics; everything else is
 * passed through to ag_gemm), but the call is PLANNED: a cost model picks among
 *     route 0  ag_gemm as it is;                         route 1  ag_gemm_resid_split;
 *     route 2  128 x 128 units of ag_gemm_ex's kernel with the forward's epilogue in the GEMM;
 *     route 3  128 x 128 units x `splits` contraction ranges into fp32 slabs in d_scratch + one row kernel (bias + residual + statistics);
 * Row statistics travel as slab-major partial sums as in ag_gemm, over slabs of 256 columns (routes 0, 1, 3) or of 128 columns (route 2):
 * stats_in_cols says which layout d_ln_stats has (route 0 reads 256 only), out_cols_ok which layouts the consumer of d_stats_out can
 * read (bit 0: 256, bit 1: 128), *stats_out_cols (host, may be NULL) returns the one written.  route < 0: planned; >= 0: that route
 * (splits: route 3's split count, 0 = planned) — the parity tests pin every route.  d_scratch: ag_gemm_ws_scratch_bytes() bytes.
 * Results equal ag_gemm's to fp32 rounding of the sums (routes 2, 3 add in another order); deterministic, no atomics. */
size_t ag_gemm_ws_scratch_bytes(int M, int N, int K, int epilogue);
int ag_gemm_ws(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
               const void* d_R, int64_t ldr, int rows_per_seq, int resid_share, int M, int N, int K, int epilogue, int dtype,
               const float* d_ln_stats, int stats_in_cols, const float* d_ln_colsum, float ln_eps,
               float* d_stats_out, int out_cols_ok, int* stats_out_cols, const int* d_rows, int route, int splits,
               void* d_scratch, size_t scratch_bytes, void* stream);
/* ag_gemm_resid_ln, planned like ag_gemm_ws: the persistent kernel (route 0) or 128 x 128 units x `splits` contraction ranges + the row
 * kernel (route 3), which recomputes LayerNorm(Rpre) from the pre-LN rows and their slab statistics exactly as the persistent kernel's
 * epilogue does.  Serves the token-pruned BERT forward at the reference's batch sizes (a few thousand packed rows: 24-48 tiles of 256 x 256
 * for the N = 768 Linears).  m_expected: with a device-side row count (d_rows), the rows the planner should price (0: M). */
int ag_gemm_resid_ln_ws(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                        const void* d_Rpre, int64_t ldr, const float* d_r_stats, const float* d_ln_g, const float* d_ln_b, float ln_eps,
                        int M, int N, int K, float* d_stats_out, const int* d_rows, int m_expected, int route, int splits,
                        void* d_scratch, size_t scratch_bytes, void* stream);
/* The MLP half of a NARROW transformer layer (the LTT ladder's side layers: hidden width h <= 128, reference
 * models/ltt_vit.py:383-394 / models/ltt_bert.py:440-455 instantiate VanillaViTLayer / VanillaBertLayer at s_attn_hidden_size) in ONE
 * kernel, bf16:   post_ln = 0 (ViT, models/vanilla_vit.py:373-376):  out = x + fc2(gelu(fc1(LN(x))))     (ln_g NULL: no LN)
 *                 post_ln = 1 (BERT, models/vanilla_bert.py:576-577,:601-603):  out = LN(x + fc2(gelu(fc1(x))))
 * x [M, h] (row stride ldx), w1 [I, h], w2 [h, I] bf16, biases / LN parameters fp32, out [M, h] (row stride ldo).  Both weight
 * matrices stay in LDS, the I-wide intermediate never leaves registers.  ag_side_mlp_supported: h in {32, 64, 96, 128},
 * I % 32 == 0, weights fit the LDS. */
int ag_side_mlp_supported(int h, int I, int dtype);
int ag_side_mlp(const void* d_x, int64_t ldx, int M, int h, int I, const void* d_w1, const float* d_b1, const void* d_w2,
                const float* d_b2, const float* d_ln_g, const float* d_ln_b, float ln_eps, int post_ln, void* d_out, int64_t ldo,
                const int* d_rows, void* stream);
/* One Linear of a narrow layer with its neighbours fused (bf16; the attention half of the ladder's side layers):
 *     out[M, N] = LN_post( resid + W . LN_pre(x) + b )       each of LN_pre (gamma/beta over h), resid [M,N], LN_post (over N) optional
 * = LN1 + QKV (ViT, models/vanilla_vit.py:369,:437-441), QKV (BERT), out-proj + residual (ViT :372,:477), out-proj + residual +
 * attention-output LayerNorm (BERT, models/vanilla_bert.py:557-559).  x [M, h], W [N, h]; h in {32,64,96,128}, N % 32 == 0,
 * N <= 384 (LN_post: N <= 128). */
int ag_side_linear_supported(int h, int N, int post_ln, int dtype);
int ag_side_linear(const void* d_x, int64_t ldx, int M, int h, int N, const void* d_w, const float* d_b,
                   const float* d_pre_g, const float* d_pre_b, const void* d_resid, int64_t ldr,
                   const float* d_post_g, const float* d_post_b, float ln_eps, void* d_out, int64_t ldo, const int* d_rows, void* stream);
/* row statistics (layout above) of a bf16 [rows,H] tensor -> d_stats [ceil(H/256), rows, 2]. */
int ag_row_stats_bf16(const void* d_x, int64_t ldx, int rows, int H, float* d_stats, void* stream);

/* Fused masked multi-head attention (reference models/vanilla_vit.py:436-465,
 * models/vanilla_bert.py:503-537): per row r and head h, softmax(mask_op(Q·Kᵀ/sqrt(d)))·V.
 * d_qkv: storage dtype [R_src, T, 3H] (q | k | v column blocks, heads contiguous inside each);
 * row r reads source row r / qkv_share (layer-0 sharing of q/k/v across the masks of one input).
 * d_mask_bits [R, ceil(T/32)]; d_ctx storage dtype [R, T, H].  head_dim must be 64.
 * n_query: only queries [0, n_query) of each row are computed/written (0 = all T; the surrogate's
 * last layer only needs the CLS query).
 * bf16, head_dim 64: ViT-mode rows of 193-200 tokens with at least three (row, head) items per CU run as ONE K/V request stream per
 * CU (a persistent workgroup per CU, three LDS images; AG_ATTN_STREAM3=0 turns it off), every other shape as one workgroup per item;
 * the two give bit-identical results. */
int ag_masked_attention(const void* d_qkv, const uint32_t* d_mask_bits, void* d_ctx, int R, int T, int H,
                        int heads, int qkv_share, int mask_mode, int n_query, int dtype, void* stream);

/* reference models/vanilla_vit.py:242-253 + :279-284: Conv2d(k=s=patch) patch embedding as
 * im2col (this call) + ag_gemm + ag_vit_assemble.  d_img fp32 [B,Cin,px,px] -> d_cols storage
 * dtype [B*(px/patch)^2, Cin*patch*patch] in (c, ph, pw) order. */
int ag_vit_im2col(const float* d_img, int B, int Cin, int px, int patch, void* d_cols, int dtype, void* stream);
/* h0[b,0,:] = cls + pos[0]; h0[b,1+p,:] = patch_emb[b,p,:] + pos[1+p]   (fp32 [B,T,H]). */
int ag_vit_assemble(const float* d_patch_emb, const float* d_cls, const float* d_pos, int B, int P, int H,
                    float* d_h0, void* stream);
/* reference models/vanilla_bert.py:307-325: LN(word[ids] + type[0] + pos[t]); ids int64 [B,T].
 * Writes fp32 h0 [B,T,H] and (optionally) the storage-dtype copy. */
int ag_bert_embed(const int64_t* d_ids, int B, int T, int H, const float* d_word, int vocab, const float* d_type0,
                  const float* d_pos, const float* d_gamma, const float* d_beta, float eps,
                  float* d_h0, void* d_h0_store, int dtype, void* stream);

/* softmax(x[rows, C]) in place / out of place, fp32 (the nn.Softmax heads, models/vanilla_vit.py:55). */
int ag_softmax_rows(const float* d_x, float* d_y, int rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Shapley reductions.
 * ---------------------------------------------------------------------------------------------- */
/* reference models/shapley.py:82-93 normalize_shapley_explanation fused with the
 * `[:, 1:, :].permute(0, 2, 1)` of models/vanilla_vit.py:129: pred fp32 [B,T,C] (T = P+1 rows
 * INCLUDING the CLS row), grand [B,C], null [1,C] -> phi [B,C,P].  normalize=0 only drops CLS and
 * permutes.  Token-axis sums are wavefront shuffle reductions. */
int ag_shapley_normalize(const float* d_pred, const float* d_grand, const float* d_null, int B, int T, int C,
                         int normalize, float* d_phi, void* stream);
/* its backward: dpred[b,t,c] = (t>0 ? dphi[b,c,t-1] : 0) - (normalize ? sum_p dphi[b,c,p] / T : 0). */
int ag_shapley_normalize_bwd(const float* d_dphi, int B, int T, int C, int normalize, float* d_dpred, void* stream);
/* reference models/shapley.py:82-93 normalize_shapley_explanation exactly as declared there (the drop-in for direct callers
 * of that function): pred [B,T,C] -> out [B,T,C] = pred + ((grand - null) - sum_t pred) / T over all T rows; and its adjoint
 * dpred = dout - sum_t dout / T. */
int ag_shapley_normalize_rows(const float* d_pred, const float* d_grand, const float* d_null, int B, int T, int C,
                              float* d_out, void* stream);
int ag_shapley_normalize_rows_bwd(const float* d_dout, int B, int T, int C, float* d_dpred, void* stream);
/* reference models/shapley.py:9-53 loss_shapley_new: mask bits [B*K, ceil((P+1)/32)] (CLS bit
 * ignored), v0 [1,C], v_s [B*K,C], phi [B,C,P] -> loss (1 float, device) and optional dphi [B,C,P]
 * (d loss / d phi).  d_scratch: >= B*K*C floats. */
int ag_shapley_loss(const uint32_t* d_mask_bits, const float* d_v0, const float* d_vs, const float* d_phi,
                    int B, int K, int P, int C, float* d_loss, float* d_dphi, float* d_scratch, void* stream);
/* reference models/shapley.py:96-106 loss_logits_kl_divergence(ref, current) on [B,C] fp32 ->
 * loss (1 float) and optional d loss / d current [B,C]. */
int ag_kl_loss(const float* d_ref, const float* d_cur, int B, int C, float* d_loss, float* d_dcur, void* stream);

/* Monte-Carlo permutation Shapley (reference scripts/preview_text_shapley.py:62-153): v [reps, P+1, C] fp32 = surrogate
 * outputs along each permutation's nested-mask chain, rank [reps, P] int32 = position of each player in its
 * permutation -> sv [C, P] (mean marginal contribution of the sharpened value log(p/(1-p+1e-6)), p = softmax(v)),
 * v0 / vn [C] = that value at the empty / full coalition of the last permutation (as the reference returns them). */
int ag_mc_shapley_reduce(const float* d_v, const int* d_rank, int reps, int P, int C, float* d_sv, float* d_v0,
                         float* d_vn, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole masked forward: the composite the recipes' fw_surrogate / fw_classifier / fw_explainer call.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ag_layer_weights {
    const void* w_qkv;   /* storage dtype [3H, H]  (query | key | value rows)                */
    const float* b_qkv;  /* [3H]                                                             */
    const void* w_o;     /* [H, H]   attention.output.dense                                  */
    const float* b_o;
    const void* w_fc1;   /* [I, H]   intermediate.dense                                      */
    const float* b_fc1;
    const void* w_fc2;   /* [H, I]   output.dense                                            */
    const float* b_fc2;
    const float* ln1_g;  /* ViT layernorm_before / BERT attention.output.LayerNorm; NULL = Identity */
    const float* ln1_b;
    const float* ln2_g;  /* ViT layernorm_after  / BERT output.LayerNorm                     */
    const float* ln2_b;
    /* optional (bf16): LayerNorm-folded projections — weights pre-scaled by gamma, bias' = b + W·beta,
     * colsum[n] = sum_k W'[n,k].  NULL = run the LayerNorm kernel.  Which LayerNorm feeds the projection:
     *   ViT (pre-LN):   qkv <- this layer's layernorm_before (ln1), fc1 <- this layer's layernorm_after (ln2);
     *   BERT (post-LN): qkv <- the PREVIOUS layer's output.LayerNorm (ln2 of layer l-1; unused in layer 0),
     *                   fc1 <- this layer's attention.output.LayerNorm (ln1).  Used by the token-pruned forward, whose
     *                   packed layers then keep only pre-LN rows + row statistics (ag_gemm_resid_ln).              */
    const void* w_qkv_ln; const float* b_qkv_ln; const float* s_qkv_ln;
    const void* w_fc1_ln; const float* b_fc1_ln; const float* s_fc1_ln;
} ag_layer_weights;

typedef struct ag_encoder_desc {
    int kind;        /* AG_MASK_VIT_MUL (pre-LN ViT block) or AG_MASK_BERT_ADD (post-LN BERT block) */
    int dtype;       /* AG_F32 / AG_BF16 */
    int T, H, I, heads;
    float ln_eps;
    int n_layers;
    const ag_layer_weights* layers; /* host array [n_layers] of device pointers */
} ag_encoder_desc;

/* bytes of workspace ag_encoder_forward needs for R rows. */
size_t ag_encoder_workspace_bytes(const ag_encoder_desc* desc, int R);
/* Run the encoder stack (reference models/vanilla_vit.py:315-320 / models/vanilla_bert.py:362-367)
 * over R rows that share B = R / share distinct inputs: d_h0 (storage dtype, [B,T,H]) are the
 * embeddings of the distinct inputs (layer 0's LN/QKV are computed once per input and its residual is
 * shared); d_mask_bits [R, ceil(T/32)].  Output d_h (storage dtype, [R,T,H]) = last layer's hidden
 * states (before any final LN).  cls_only_last != 0 computes the last layer's attention/out-proj/MLP
 * for token 0 only (legal when only h[:,0] is consumed: surrogate/classifier heads) — then only
 * d_h[r,0,:] is defined. */
int ag_encoder_forward(const ag_encoder_desc* desc, const void* d_h0, int R, int share,
                       const uint32_t* d_mask_bits, void* d_h, int cls_only_last,
                       void* d_workspace, size_t workspace_bytes, void* stream);
/* The same forward for callers that run the layers one call at a time (the LTT ladder taps the stream after every backbone
 * layer, reference models/ltt_vit.py:423-436): d_row_stats [AG_ROW_STATS_FLOATS(R*T, H)] floats, caller-owned, carries the LayerNorm-fold row
 * statistics from one call to the next.  stats_in_ready: they describe d_h0 (written by the previous call with
 * want_stats_out that reported *stats_written = 1, same R, share == 1); want_stats_out: the last layer's fc2 accumulates the
 * statistics of d_h into them when the shapes fold (ViT, bf16, large GEMMs) and *stats_written (host int) says whether it
 * did: pass that value as the next call's stats_in_ready. */
int ag_encoder_forward_chained(const ag_encoder_desc* desc, const void* d_h0, int R, int share,
                               const uint32_t* d_mask_bits, void* d_h, int cls_only_last,
                               void* d_workspace, size_t workspace_bytes, float* d_row_stats, int stats_in_ready,
                               int want_stats_out, int* stats_written, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward-pass building blocks (fp32) for explainer / surrogate training — what torch.autograd does
 * for the reference in scripts/train_explainer.py:184-198 and scripts/train_surrogate.py:145-147.
 * Linear backward reuses ag_gemm:  dX = ag_gemm(dY, Wᵀ),  dW = ag_gemm(dYᵀ, Xᵀ)  (K dims zero-padded
 * to a multiple of 32 by the caller), db = ag_colsum_f32(dY).
 * ---------------------------------------------------------------------------------------------- */
/* dst[c, r] = src[r, c]; src row stride lds, dst row stride ldd (>= rows; padding untouched). */
int ag_transpose_f32(const float* d_src, int rows, int cols, int64_t lds, float* d_dst, int64_t ldd, void* stream);
/* The same with a bf16 destination (mixed-precision training: the dW operands dYᵀ, Xᵀ are transposed and rounded in one
 * pass); every element of dst [cols, ldd] is written, zeros beyond `rows`. */
int ag_transpose_f32_bf16(const float* d_src, int rows, int cols, int64_t lds, void* d_dst, int64_t ldd, void* stream);
/* Both bf16 operand forms of an fp32 matrix in one pass: d_plain [rows, cols] (dense) and d_dst_t [cols, ldd] (as above). */
int ag_cast_transpose_f32_bf16(const float* d_src, int rows, int cols, int64_t lds, void* d_plain, void* d_dst_t, int64_t ldd,
                               void* stream);
/* The same pass with the exact-erf GELU applied on the way (mixed-precision training, fc1 -> fc2): d_plain / d_dst_t hold gelu(u);
 * the fp32 GELU output is never written (the backward recomputes from u).  cols % 4 == 0, ldd % 4 == 0. */
int ag_gelu_cast_transpose_f32_bf16(const float* d_u, int rows, int cols, void* d_plain, void* d_dst_t, int64_t ldd, void* stream);
/* ... and its backward: du = dy * gelu'(u) as both bf16 operand forms, and (d_du != NULL) in fp32 for the bias sums. */
int ag_gelu_bwd_cast_transpose_f32_bf16(const float* d_u, const float* d_dy, int rows, int cols, float* d_du, void* d_plain,
                                        void* d_dst_t, int64_t ldd, void* stream);
/* out[n] (+)= sum_m x[m, n]  (bias gradients). */
int ag_colsum_f32(const float* d_x, int M, int N, int64_t ldx, float* d_out, int accumulate, void* stream);
/* exact-erf GELU (nn.GELU default) and its derivative: du = dy * gelu'(u). */
int ag_gelu_f32(const float* d_u, float* d_y, int64_t n, void* stream);
int ag_gelu_bwd_f32(const float* d_u, const float* d_dy, float* d_du, int64_t n, void* stream);
/* dx = dy * (1 - y^2)  (BERT pooler tanh). */
int ag_tanh_bwd_f32(const float* d_y, const float* d_dy, float* d_dx, int64_t n, void* stream);
int ag_add_f32(const float* d_a, const float* d_b, float* d_y, int64_t n, void* stream);
/* inverted dropout with a counter-based keep decision hash(seed, index): y = keep ? x/(1-p) : 0.
 * Calling it on dy with the same (p, seed) is the backward (nn.Dropout, p = hidden_dropout_prob). */
int ag_dropout_f32(const float* d_x, float* d_y, int64_t n, float p, uint32_t seed, void* stream);
/* y = resid + dropout(x)  (models/vanilla_vit.py:372,:512-513 / vanilla_bert.py:558-559: hidden = residual + dropout(dense)). */
int ag_dropout_add_f32(const float* d_x, const float* d_resid, float* d_y, int64_t n, float p, uint32_t seed, void* stream);
/* dx = y * (dy - sum_c y*dy)  for y = softmax(x) rows. */
int ag_softmax_rows_bwd(const float* d_y, const float* d_dy, float* d_dx, int rows, int C, void* stream);
/* LayerNorm backward: dx [rows,H]; dgamma/dbeta [H] (= or += when accumulate); gamma may be NULL (ones).
 * d_scratch: >= 256*2*H floats. */
int ag_layernorm_bwd(const float* d_x, const float* d_gamma, const float* d_dy, int rows, int H, float eps,
                     float* d_dx, float* d_dgamma, float* d_dbeta, int accumulate, float* d_scratch, void* stream);
/* ... with an addend: dx = d_add + LayerNorm backward (the gradient arriving over the residual branch; d_add may be NULL). */
int ag_layernorm_bwd_add(const float* d_x, const float* d_gamma, const float* d_dy, const float* d_add, int rows, int H, float eps,
                         float* d_dx, float* d_dgamma, float* d_dbeta, int accumulate, float* d_scratch, void* stream);
/* fp32 masked attention forward WITH attention-probability dropout (training); p_drop = 0 is the plain
 * forward.  Same layout as ag_masked_attention (AG_F32). */
int ag_masked_attention_train(const float* d_qkv, const uint32_t* d_mask_bits, float* d_ctx, int R, int T, int H,
                              int heads, int mask_mode, float p_drop, uint32_t seed, void* stream);
/* its backward: dqkv [R,T,3H] from dctx, recomputing the probabilities (two passes, no atomics).
 * d_stats: R*heads*T*3 floats of scratch. */
int ag_masked_attention_bwd(const float* d_qkv, const uint32_t* d_mask_bits, const float* d_ctx, const float* d_dctx,
                            float* d_dqkv, float* d_stats, int R, int T, int H, int heads, int mask_mode,
                            float p_drop, uint32_t seed, void* stream);
/* The same gradients on the matrix cores for mixed-precision training (head dim 64, T <= 256): Q, K, V, dO are rounded to
 * bf16 operands, accumulation, soft-max statistics and I/O stay fp32; same dropout decisions (p_drop, seed) as
 * ag_masked_attention_train.  No statistics scratch.  Reference: torch.autograd through models/vanilla_vit.py:436-465 /
 * models/vanilla_bert.py:503-537 under autocast(bf16). */
/* Training forward of the same mode: ctx [R,T,H] fp32 from bf16 MFMA operands, dropout as ag_masked_attention_train. */
int ag_masked_attention_train_mixed(const float* d_qkv, const uint32_t* d_mask_bits, float* d_ctx, int R, int T, int H, int heads,
                                    int mask_mode, float p_drop, uint32_t seed, void* stream);
int ag_masked_attention_bwd_mixed(const float* d_qkv, const uint32_t* d_mask_bits, const float* d_ctx, const float* d_dctx,
                                  float* d_dqkv, int R, int T, int H, int heads, int mask_mode, float p_drop, uint32_t seed,
                                  void* stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM for under-filled launches (round 4): the training step's Linear forward / dX / dW on B*T ~ 1-1.6 k rows (what torch.autograd
 * runs for every nn.Linear in scripts/train_explainer.py:183-198, scripts/train_duo_explainer.py:180-198) and the masked forward
 * at the reference's own batch sizes.  bf16 operands, fp32 accumulation:
 *     C[M,N] = sum_kc A(m,kc) * B(n,kc)        Kc = contraction length
 * with each operand read IN PLACE in either storage order (no transposed or re-cast copies):
 *     a_col = 0: A stored [M, Kc], row stride lda       a_col = 1: A stored [Kc, M], row stride lda
 *     b_col = 0: B stored [N, Kc] (torch [out,in])      b_col = 1: B stored [Kc, N]
 *   Linear forward  Y  = X . W^T :  (A, B) = (X [M,K],  W [N,K]),  (a_col, b_col) = (0, 0)
 *   Linear dX       dX = dY . W  :  (A, B) = (dY [M,N], W [N,K] read as [Kc=N, K]),  (0, 1)
 *   Linear dW       dW = dY^T . X:  (A, B) = (dY [M,N] read as [Kc=M, N], X [M,K] read as [Kc=M, K]),  (1, 1); the result is [N, K]
 * Work = 128 x 128 output tiles x `splits` contraction ranges, one workgroup each, so that a product with few tiles still covers
 * the chip.  splits > 1 (epilogue AG_EX_SLABS only): unit s stores its fp32 partial tile into d_slabs [splits][M][N] (dense) with
 * plain stores; the consumer (ag_rows_finish, ag_rows_ln_bwd, ag_slab_reduce, ...) adds the slabs in slab order — no atomics, no
 * zero fill, bit-reproducible.  ag_gemm_ex_splits() is the split count the launch heuristics recommend for a shape.
 * Epilogues with splits == 1:
 *   AG_EX_STORE      C = acc + bias                     -> c_dtype (AG_BF16 / AG_F32), row stride ldc
 *   AG_EX_GELU_DUAL  C = bf16(acc + bias), out2 = gelu(C)   (fc1: the backward needs the pre-activation, fc2 the activation)
 *   AG_EX_GELU_BWD   C = bf16(acc * gelu'(aux))         aux = bf16 pre-activation [M, ld_aux]   (fc2's dX feeding fc1's backward)
 * (bf16 GELU forms: |gelu - erf form| <= 1.7e-5, as ag_gemm's bf16 mode.)  Requirements: N, lda, ldb multiples of 8; Kc % 8 == 0 unless
 * both operands are stored [Kc, .]; M % 8 == 0 when a_col; ldc % 4 == 0; 16-byte aligned pointers. */
enum { AG_EX_STORE = 0, AG_EX_GELU_DUAL = 1, AG_EX_GELU_BWD = 2, AG_EX_SLABS = 3 };
int ag_gemm_ex(const void* d_A, int64_t lda, int a_col, const void* d_B, int64_t ldb, int b_col, int M, int N, int Kc,
               int epilogue, const float* d_bias, void* d_C, int64_t ldc, int c_dtype, const void* d_aux, int64_t ld_aux,
               void* d_out2, int64_t ld_out2, int splits, float* d_slabs, void* stream);
int ag_gemm_ex_splits(int M, int N, int Kc);
/* `count` products C_i = A_i . B_i of ONE operand order (a_col, b_col as in ag_gemm_ex) with the plain store epilogue and no bias, C_i
 * [M_i, N_i] in c_dtype (AG_F32 / AG_BF16), as ONE launch per 8 products: the backward of the bf16 training step hands the dW products
 * of two layers (8 Linears: torch.autograd's grad_weight of every nn.Linear in scripts/train_explainer.py:197, train_duo_explainer.py:197)
 * to one launch whose 128 x 128 units cover the chip, where each product alone needed contraction ranges + a slab reduction.  The arrays
 * are HOST arrays of `count` entries.  Requirements per product as ag_gemm_ex. */
int ag_gemm_ex_group(int count, const void* const* d_A, const int64_t* lda, const void* const* d_B, const int64_t* ldb, const int* M,
                     const int* N, const int* Kc, void* const* d_C, const int64_t* ldc, int a_col, int b_col, int c_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row kernels of the bf16 training step (round 4, csrc/train_fused.hip): the consumers of ag_gemm_ex's split-K slabs.  Each adds
 * `splits` fp32 slabs [splits][M][H] (slab stride in floats; splits = 1: a plain fp32 [M,H] tensor) in slab order and does, in the
 * same pass over the row, what the reference's autograd graph does between two Linear layers (models/vanilla_vit.py:364-377,
 * models/vanilla_bert.py:410-427, :556-560, :600-604).  One wave per row, H % 4 == 0, H <= 1024, fp32 arithmetic.  The dropout keep
 * decision is the counter hash of ag_dropout_f32 on the element index m*H + c.
 * ---------------------------------------------------------------------------------------------- */
/* t = resid + dropout(x + bias);  z = LayerNorm(t) (gamma NULL: z = t).  Outputs (each optional): d_t_out fp32 (the residual
 * stream / the LayerNorm input the backward needs), d_z_f32, d_z_bf16 (the next GEMM's operand). */
int ag_rows_finish(const float* d_x, int splits, int64_t slab_stride, const float* d_bias, float p_drop, uint32_t seed,
                   const float* d_resid, float* d_t_out, const float* d_gamma, const float* d_beta, float eps,
                   float* d_z_f32, void* d_z_bf16, int M, int H, void* stream);
/* dy = x_slabs (+ d_dy_add);  dx = LayerNorm backward of dy at input rows d_x (d_x NULL: dx = dy)  (+ d_add: the gradient arriving
 * over the residual branch).  Outputs (each optional): d_dx fp32; d_dx_bf16 = bf16(dropout'(dx; p_drop, seed)) — the dY operand of
 * the Linear below; d_dgamma / d_dbeta [H]; d_dbias [H] = column sums of dropout'(dx) (that Linear's bias gradient); `accumulate`
 * adds to the three instead of storing.  d_scratch: ag_rows_ln_bwd_scratch_floats(M, H) floats (per-block partials, folded in block
 * order: bit-reproducible). */
size_t ag_rows_ln_bwd_scratch_floats(int M, int H);
int ag_rows_ln_bwd(const float* d_dy, int splits, int64_t slab_stride, const float* d_dy_add, const float* d_x, const float* d_gamma,
                   float eps, const float* d_add, float* d_dx, void* d_dx_bf16, float p_drop, uint32_t seed, float* d_dgamma,
                   float* d_dbeta, float* d_dbias, int accumulate, float* d_scratch, int M, int H, void* stream);
/* dst[n] (+)= sum of slabs (a dW product split over the rows); n, slab_stride multiples of 4. */
int ag_slab_reduce(const float* d_slabs, int splits, int64_t slab_stride, int64_t n, float* d_dst, int accumulate, void* stream);
/* out[N] (+)= column sums of a bf16 [M,N] matrix (the bias gradient of a dY that a GEMM / attention epilogue produced);
 * N, ldx multiples of 8.  One launch, a fixed summation tree (bit-reproducible); d_scratch is unused (kept in the signature, may be NULL;
 * ag_colsum_bf16_scratch_floats returns 0). */
size_t ag_colsum_bf16_scratch_floats(int M, int N);
int ag_colsum_bf16(const void* d_x, int M, int N, int64_t ldx, float* d_out, int accumulate, float* d_scratch, void* stream);
/* the same column sums (same row walk and summation tree: the same bits) for `count` matrices in one launch per 16 (HOST arrays):
 * the bias gradients of the Linears whose dW products ag_gemm_ex_group computes */
int ag_colsum_bf16_group(int count, const void* const* d_x, const int* M, const int* N, const int64_t* ldx, float* const* d_out, void* stream);
/* `count` fp32 -> bf16 conversions (h_dst_dtype[i] = AG_BF16) or fp32 copies (AG_F32) in ONE launch per 96 segments: every
 * weight of a model after the optimiser step, q | k | v landing side by side in their fused buffer.  HOST arrays of DEVICE pointers
 * (passed to the kernel by value: no table copy, graph-capturable); segments 16-byte aligned. */
int ag_cast_f32_many(const float* const* h_src, void* const* h_dst, const int64_t* h_n, const int* h_dst_dtype, int count, void* stream);
/* The same launch with every element scaled on the way: the gradient exchange of N > 1 ranks packs the gradients a bucket holds —
 * each a tensor of its own — into the bucket's flat send buffer (fp32, or bf16 for the compressed exchange), weighted by this rank's
 * share of the global batch, in ONE launch (the reference's single process has no such step: scripts/train_explainer.py:197-198 is
 * loss.backward(); optimizer.step()).  autognothi_amd/distributed.py GradBucketReducer. */
int ag_pack_f32_many(const float* const* h_src, void* const* h_dst, const int64_t* h_n, const int* h_dst_dtype, int count, float scale,
                     void* stream);
/* Dropout salt of the bf16 training step: mixed into the seed of every dropout decision of ag_rows_finish, ag_rows_ln_bwd,
 * ag_masked_attention_*_bf16 / _mixed, ag_dropout_f32 and ag_dropout_add_f32 (seed ^ salt * 0x9E3779B1).  A hipGraph-captured step has
 * its per-site seeds frozen into the kernel arguments; setting a new salt before each replay (stream-ordered, outside the graph)
 * gives it fresh keep patterns.  0 — the value until this is called — leaves seeds as given. */
int ag_set_dropout_salt(uint32_t salt, void* stream);
/* dst[m, c] = c < cols_src ? src[m, c] : 0 for c < cols_dst; dst fp32 or bf16 (dst_dtype).  Pads the C-wide output / gradient of the
 * explainer's last Linear (C = 10 / 2 classes, models/vanilla_vit.py:92-100) to the 16 columns ag_gemm_ex wants, or strips them. */
int ag_pad_cols_f32(const float* d_src, int64_t ld_src, int cols_src, void* d_dst, int64_t ld_dst, int cols_dst, int dst_dtype, int M,
                    void* stream);
/* Masked attention of the bf16 training step on the matrix cores (the kernel of ag_masked_attention_*_mixed with bf16 I/O):
 * qkv [R,T,3H] bf16 -> ctx [R,T,H] bf16, attention-probability dropout (p_drop, seed) as ag_masked_attention_train; backward:
 * dqkv [R,T,3H] bf16 from d_dctx given as `dslabs` fp32 slabs [dslabs][R*T][H] (ag_gemm_ex's split-K partials of the
 * out-projection dX), added in slab order on load.  head_dim 64, T <= 256. */
int ag_masked_attention_train_bf16(const void* d_qkv, const uint32_t* d_mask_bits, void* d_ctx, int R, int T, int H, int heads,
                                   int mask_mode, float p_drop, uint32_t seed, void* stream);
int ag_masked_attention_bwd_bf16(const void* d_qkv, const uint32_t* d_mask_bits, const float* d_dctx, int dslabs, int64_t dslab_stride,
                                 void* d_dqkv, int R, int T, int H, int heads, int mask_mode, float p_drop, uint32_t seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * In-library kernel timing (used by bench.py for the roofline block): when enabled, every launch of
 * an instrumented kernel class is bracketed by hipEvents on the launch stream.  ag_profile_collect
 * synchronises those events, returns the totals of one class since the last collect and clears it.
 * Classes: 0..4 = ag_gemm by epilogue (AG_EPI_*), 8 = ag_masked_attention, 9 = ag_layernorm, 10 = ag_gemm_ex.
 * ---------------------------------------------------------------------------------------------- */
#define AG_PROF_ATTENTION 8
#define AG_PROF_LAYERNORM 9
#define AG_PROF_GEMM_EX 10   /* ag_gemm_ex */
/* number of kernels this library has launched in this process (diagnostics: launches per training step in bench.py). */
int64_t ag_launch_count(void);
/* (diagnostic) how the persistent large-M kernel behind ag_gemm / ag_gemm_resid_ln / ag_gemm_ws was last launched by the calling thread:
 * resident workgroups, the first half-height unit (= the tile count when the launch had no half-height tail) and the number of 256^2
 * tiles that ran as two 128-row units each in the last round (0: none).  Returns 1 if such a launch happened on this thread, else 0.
 * The reference has no counterpart (torch picks its GEMM kernels out of sight): parity tests use it to prove which schedule they compared. */
int ag_gemm_last_plan(int* grid, int* half_from, int* ntail);
int ag_profile_enable(int on);
int ag_profile_collect(int kernel_class, double* total_ms, double* total_flops, double* total_bytes, int64_t* launches);

/* ------------------------------------------------------------------------------------------------
 * BERT token pruning.  reference models/vanilla_bert.py:523 adds (1 - mask) * finfo.min to the scores, so a masked
 * key's soft-max weight is exactly 0 in every layer; the heads read the CLS row only (:73-76) and CLS is never
 * masked: hidden states of masked tokens are dead rows.  ag_bert_encoder_forward_pruned = ag_encoder_forward(...,
 * cls_only_last = 1) for the BERT kind, computing layer 0 on every token (its LN/QKV are shared by the K masks) and
 * layers 1.. on the packed visible tokens only (mask-free varlen attention).  Same output contract (token 0 of
 * every row of d_h [R,T,H]); d_packed_rows_out (optional, DEVICE int) receives the number of visible tokens.  Nothing is
 * read back to the host: the packed section's launches are sized for the upper bound R*T and clamp to the device-side count
 * (their d_rows argument), so the call is asynchronous and graph-capturable.
 * Building blocks: ag_seq_compact_plan (cu_seqlens [R+1] by popcount + scan, packed-row -> source-row table [<= R*T]),
 * ag_gather_rows (dst[i,:] = src[index[i],:]), ag_masked_attention_varlen (rows = token ranges of cu_seqlens).
 * ---------------------------------------------------------------------------------------------- */
int ag_seq_compact_plan(const uint32_t* d_mask_bits, int R, int T, int* d_cu_seqlens, int* d_tok_src, void* stream);
int ag_gather_rows(const void* d_src, int64_t ld_src, const int* d_index, void* d_dst, int64_t ld_dst, int n, int H,
                   int dtype, const int* d_rows, void* stream);
int ag_masked_attention_varlen(const void* d_qkv, const int* d_cu_seqlens, void* d_ctx, int R, int t_max, int H,
                               int heads, int cls_only, int dtype, void* stream);
/* desc's BERT layers on packed rows: x / out [N, H] (visible tokens of R sequences, ranges in cu_seqlens [R+1]), every packed
 * token a visible key, all tokens processed.  Workspace as ag_encoder_workspace_bytes(desc, R).  out must not alias x.
 * d_rows (optional): N is an upper bound, the packed row count is read from d_rows[0] on the device (= d_cu_seqlens[R]). */
int ag_bert_layers_forward_packed(const ag_encoder_desc* desc, const void* d_x, const int* d_cu_seqlens, int R, int N,
                                  void* d_out, void* d_workspace, size_t workspace_bytes, const int* d_rows, void* stream);
int ag_bert_encoder_forward_pruned(const ag_encoder_desc* desc, const void* d_h0, int R, int share,
                                   const uint32_t* d_mask_bits, void* d_h, void* d_workspace, size_t workspace_bytes,
                                   int* d_packed_rows_out, void* stream);
/* Experiment / test knobs (AG_GEMM_*, AG_SIDE_MLP, AG_BERT_LN_FOLD, ...) are read from the environment ONCE per process, never on
 * the launch path; a test that changes one calls this to have them read again. */
int ag_reload_knobs(void);
/* How many CUs the persistent large-M GEMM may take on a given stream (n_cu > 0 registers, n_cu <= 0 forgets; default: every CU).  The
 * kernel launches one resident workgroup per CU; a stream registered with n_cu = 8 c launches 8 c workgroups — c on every XCD, an equal
 * number on every shader engine when c is a multiple of 4 — and leaves the other CUs to whatever else is queued on the device.  Used by
 * the explainer training epoch (scripts/common.TrainPartition): the K-mask target forward of the NEXT batches on a second stream with
 * 24-28 of every XCD's 32 CUs while the explainer's own, under-filled step runs on the caller's stream (the reference runs the two back
 * to back: scripts/train_explainer.py:153-198).  Also what a stream created with hipExtStreamCreateWithCUMask has to be registered with. */
int ag_set_stream_cus(void* stream, int n_cu);

/* The last layer's attention of a CLS-only ViT forward WITHOUT its key / value projection (csrc/cls_last.hip; the encoder takes this
 * path from AG_LAST_KV_SKIP rows up).  Restates reference models/vanilla_vit.py:436-465 (Q / K / V Linear, scores / 8, scores * mask,
 * soft-max, P V) for the ONE query per row and head that the heads read (models/vanilla_vit.py:51-56: hidden[:, 0]), with the layer's
 * LayerNorm (models/vanilla_vit.py:369) folded:  d_h [R*T, H] bf16 = the residual stream entering the layer, d_stats its row statistics
 * ((sum, sumsq) per slab of `cols` = 256 or 128 columns, slab-major: ag_row_stats_bf16), d_mask_bits [R, ceil(T/32)] key bits,
 * d_q [R, H] bf16 = the CLS queries (already projected, bias included), w_kv_ln [2H, H] bf16 = the gamma-folded key | value rows of the
 * fused projection and b_kv_ln [2H] their folded biases (b + W beta) -> d_ctx + r * ctx_row_stride (elements): the merged-heads
 * attention output [H] of row r's CLS token.  ag_cls_last_supported: bf16, H = 768 / 1024, heads * 64 == H, 2 <= T <= 256. */
int ag_cls_last_is_supported(int T, int H, int heads, int dtype);
size_t ag_cls_last_workspace_bytes(int R, int H, int heads);
int ag_cls_last_attention_rows(const void* d_h, const float* d_stats, int cols, const uint32_t* d_mask_bits, const void* d_q,
                               const void* w_kv_ln, const float* b_kv_ln, float ln_eps, void* d_ctx, int64_t ctx_row_stride, int R, int T,
                               int H, int heads, void* d_scratch, size_t scratch_bytes, void* stream);

/* Measurement aid (bench.py): what this board's matrix cores sustain when a kernel issues nothing but
 * v_mfma_f32_16x16x32_bf16 from registers on every SIMD (two waves each) for `iters` x 16 instructions per
 * wave — the power-capped MFMA ceiling — and the effective shader clock during it (s_memtime ticks against
 * the constant 100 MHz s_memrealtime).  zero_operands is a flag word: bit 0 runs the same loop on all-zero operands (no
 * data-dependent switching power), bit 1 issues v_mfma_f32_32x32x16_bf16 instead (same FLOPs per iteration).
 * Synchronous.  No reference counterpart. */
int ag_probe_mfma(int iters, int zero_operands, double* tflops, double* shader_ghz, void* stream);
/* One wave idling for `microseconds` on `stream`: two of them on two streams take one kernel's time when the streams run beside each
 * other and two when HIP has put them on one hardware queue (the two-stream training epoch checks its second stream with it). */
int ag_probe_spin(int microseconds, void* stream);

/* Measurement aid: what the global -> LDS feed of one CU sustains when a workgroup issues nothing but the ring GEMM's staging requests
 * (32 `global_load_lds_dwordx4` pieces of 16 rows x 64 B per K=32 half-step, 4-slot ring, counted vmcnt) from `waves` (4 / 8 / 16) waves.
 * d_A: [panels*256, ld_bytes] bytes, d_W: [768, ld_bytes] bytes, half_steps <= ld_bytes / 64.  flags: bit 0 one s_barrier per half-step,
 * bit 1 every workgroup reads the same (L2-resident) panels, bit 2 plain loads into registers instead of LDS-DMA.  Synchronous. */
int ag_probe_dma(int waves, int half_steps, int flags, const void* d_A, const void* d_W, int64_t ld_bytes, int panels,
                 double* bytes_per_clk_per_cu, double* gbytes_per_s, double* shader_ghz, void* stream);

/* Measurement aid: what the output path of a CU sustains for 256 x 256 bf16 output tiles of a row-major [rows, ld_bytes / 2] matrix written by
 * `grid` 8-wave workgroups and nothing else.  shape 0: 8 rows x 128 B per store instruction (the large-M GEMM's epilogue: a wave owns a
 * 64-column strip), 1: 2 rows x 512 B, 2: 4 rows x 256 B; flags bit 0: non-temporal stores.  Synchronous. */
int ag_probe_store(int shape, int flags, void* d_C, int64_t ld_bytes, int rows, int grid, double* bytes_per_clk_per_cu, double* gbytes_per_s,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUTOGNOTHI_HIP_H_ */
