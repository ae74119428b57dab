// encoder.cpp — the masked transformer stack as one C call (host-side orchestration of the HIP
// kernels on the caller's stream; no allocation, no synchronisation — graph-capturable).
//
// reference models/vanilla_vit.py:315-320 + :364-377 (pre-LN ViT block) and
// models/vanilla_bert.py:362-367 + :410-427 (post-LN BERT block), run for R = B*K masked rows.
//
// What is MI355X-specific rather than a restatement of the reference loop:
//   * the K masked copies of one input share everything that does not depend on the mask: the
//     embeddings, layer 0's LayerNorm + QKV projection (computed for B rows, not R) and its
//     residual; divergence starts at layer 0's masked soft-max.
//   * q/k/v are one [3H,H] GEMM; bias+GELU, bias+residual are GEMM epilogues; the score tensor is
//     never materialised; the residual stream lives in the storage dtype so a LayerNorm output is at
//     once the next GEMM's operand and (BERT) the next residual.
//   * with cls_only_last the last layer's attention / out-projection / MLP run on the CLS token only
//     (the only row the surrogate/classifier heads read, models/vanilla_vit.py:54).
#include "common.h"
#include <stdlib.h>

namespace {

struct Ws {
    char* xs;      // [M, H]   LayerNorm output (GEMM operand)
    char* qkv;     // [M, 3H]
    char* ctx;     // [M, H]
    char* inter;   // [M, I]
    char* hx;      // [M, H]   residual stream after the attention block
    char* ha;      // [M, H]   BERT: post-attention LayerNorm output
    float* st1;    // [S, M, 2]  per-256-column partial (sum, sumsq) of the rows entering LN1 (folded into the QKV GEMM), S = ceil(H/256)
    float* st2;    // [S, M, 2]  same for LN2 (folded into fc1)
    int* idx;      // [R+1 + M] token pruning plan
    float* split;  // fp32 partial tiles of a split fc2 (ag_gemm_resid_split; NULL when the shape does not split)
    size_t split_bytes;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t carve(const ag_encoder_desc* d, int R, char* base, Ws* ws) {
    const size_t es = dtype_size(d->dtype);
    const size_t M = (size_t)R * d->T;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return base ? base + o : nullptr; };
    char* xs = take(M * d->H * es);
    char* qkv = take(M * 3 * d->H * es);
    char* ctx = take(M * d->H * es);
    char* inter = take(M * d->I * es);
    char* hx = take(M * d->H * es);
    char* ha = d->kind == AG_MASK_BERT_ADD ? take(M * d->H * es) : nullptr;
    const size_t S = ((size_t)d->H + 127) / 128;   // (slabs of 256 columns, or of 128: the 128-tile producers of ag_gemm_ws)
    char* st1 = take(S * M * 2 * sizeof(float));
    char* st2 = take(S * M * 2 * sizeof(float));
    char* idx = take((M + (size_t)R + 1) * sizeof(int));   // token pruning: cu_seqlens [R+1] + packed-row sources [M]
    // under-filled fc2 (the reference's own batch sizes: one to four inputs x K masks): the tail round's rows as contraction ranges side by side
    // ... and the fp32 slabs of the planned 128-tile routes (ag_gemm_ws): the widest of both residual Linears
    size_t split_bytes = 0;
    if (d->dtype == AG_BF16 && M <= 0x7FFFFFFF) {
        // (all tokens, and the R CLS rows of a cls_only_last layer)
        const size_t b1 = ag_gemm_ws_scratch_bytes((int)M, d->H, d->I, AG_EPI_BIAS_RESID), b2 = ag_gemm_ws_scratch_bytes((int)M, d->H, d->H, AG_EPI_BIAS_RESID);
        const size_t b3 = ag_gemm_ws_scratch_bytes(R, d->H, d->I, AG_EPI_BIAS_RESID), b4 = ag_gemm_ws_scratch_bytes(R, d->H, d->H, AG_EPI_BIAS_RESID);
        split_bytes = b1 > b2 ? b1 : b2;
        split_bytes = split_bytes > b3 ? split_bytes : b3;
        split_bytes = split_bytes > b4 ? split_bytes : b4;
    }
    char* split = split_bytes ? take(split_bytes) : nullptr;
    if (ws) { ws->split = (float*)split; ws->split_bytes = split_bytes; }
    if (ws) ws->idx = (int*)idx;
    if (ws) { ws->xs = xs; ws->qkv = qkv; ws->ctx = ctx; ws->inter = inter; ws->hx = hx; ws->ha = ha; ws->st1 = (float*)st1; ws->st2 = (float*)st2; }
    return off;
}

#define TRY(expr) do { int _rc = (expr); if (_rc != AG_OK) return _rc; } while (0)

}  // namespace

extern "C" size_t ag_encoder_workspace_bytes(const ag_encoder_desc* desc, int R) {
    if (!desc || R <= 0) return 0;
    return carve(desc, R, nullptr, nullptr);
}

// chain: caller-owned row statistics [ceil(H/256), R*T, 2] handed from one call to the next (a model that runs the layers one call at
// a time, e.g. to tap the stream after each: the LTT ladder): stats_in_ready = they describe d_h0 (written by the previous
// call's last fc2), want_stats_out = the last layer's fc2 accumulates the statistics of d_h into them.
static int encoder_forward_impl(const ag_encoder_desc* d, const void* d_h0, int R, int share,
                                const uint32_t* d_mask_bits, void* d_h, int cls_only_last,
                                void* d_workspace, size_t workspace_bytes, float* chain_stats, bool stats_in_ready,
                                bool want_stats_out, int* stats_written, void* stream) {
    AG_REQUIRE(d && d_h0 && d_mask_bits && d_h && d_workspace, "ag_encoder_forward: null pointer");
    AG_REQUIRE(d->kind == AG_MASK_VIT_MUL || d->kind == AG_MASK_BERT_ADD, "ag_encoder_forward: bad kind %d", d->kind);
    AG_REQUIRE(d->dtype == AG_BF16 || d->dtype == AG_F32, "ag_encoder_forward: bad dtype %d", d->dtype);
    AG_REQUIRE(R > 0 && share >= 1 && R % share == 0, "ag_encoder_forward: R=%d must be a positive multiple of share=%d", R, share);
    AG_REQUIRE(d->n_layers >= 1 && d->layers, "ag_encoder_forward: no layers");
    AG_REQUIRE(workspace_bytes >= ag_encoder_workspace_bytes(d, R), "ag_encoder_forward: workspace too small (%zu < %zu)",
               workspace_bytes, ag_encoder_workspace_bytes(d, R));
    Ws ws;
    carve(d, R, (char*)d_workspace, &ws);
    const int T = d->T, H = d->H, I = d->I, dt = d->dtype;
    const size_t es = dtype_size(dt);
    const int M = R * T;
    const bool vit = d->kind == AG_MASK_VIT_MUL;
    static const bool side_off = getenv("AG_SIDE_MLP") && atoi(getenv("AG_SIDE_MLP")) == 0;
    const bool side_mlp = !side_off && ag_side_mlp_supported(H, I, dt);
    const bool side_qkv = !side_off && ag_side_linear_supported(H, 3 * H, 0, dt);
    const bool side_lin_o = !side_off && ag_side_linear_supported(H, H, 1, dt);
    if (chain_stats) ws.st1 = chain_stats;
    if (stats_written) *stats_written = 0;
    const char* h_in = (const char*)d_h0;  // residual stream entering the layer (storage dtype)
    int in_share = share;                  // how many rows share one h_in sequence
    // ws.st1 holds the row statistics of h_in (written by the previous fc2 / the previous call): their slab width (256 / 128 columns;
    // 0: not there)
    int fmt1 = (chain_stats && stats_in_ready && share == 1) ? 256 : 0;
    hipStream_t hs = (hipStream_t)stream;
    const int* const dyn = nullptr;        // every row count of this entry is exact on the host
    const bool bf = dt == AG_BF16;
    // Every Linear is PLANNED (ag_ws_plan, gemm_tn.hip): the persistent 256^2 kernel where the launch fills it, 128^2 units (epilogue in the
    // GEMM, or split-K slabs + a row kernel) where it would leave most of a round idle — the reference's own batch sizes and the 8-GPU
    // shards.  A producer of row statistics and the folded consumer that reads them are planned together: the 128-tile producer writes
    // 128-column slabs, which only the 128-tile consumer reads.
    auto plan_w = [&](int M_, int N_, int K_, int64_t lda_, int64_t ldc_, int epi, bool fold_in, int in_cols) {
        return ag_ws_plan(M_, N_, K_, lda_, ldc_, 0, epi, dt, false, fold_in, in_cols, false, 0, 1);
    };
    auto plan_r = [&](int M_, int N_, int K_, int64_t lda_, int64_t ldc_, int64_t ldr_, bool stats_out, int cols_ok, int share_) {
        AgWsPlan pl = ag_ws_plan(M_, N_, K_, lda_, ldc_, ldr_, AG_EPI_BIAS_RESID, dt, false, false, 0, stats_out, cols_ok, share_);
        if (pl.valid && pl.scratch_bytes > ws.split_bytes)     // (a caller's workspace from another shape: the routes that need no scratch)
            pl = ag_ws_plan(M_, N_, K_, lda_, ldc_, ldr_, AG_EPI_BIAS_RESID, dt, false, false, 0, stats_out, cols_ok, share_, AG_WS_GEMM);
        return pl;
    };
    auto run = [&](const AgWsPlan& pl, const void* A, int64_t lda_, const void* W, const float* b, void* C, int64_t ldc_, const void* Rr,
                   int64_t ldr_, int tq, int sh, int M_, int N_, int K_, int epi, const float* st_in, int in_cols, const float* colsum,
                   float* st_out) {
        return ag_gemm_ws_run(pl, A, lda_, W, b, C, ldc_, Rr, ldr_, tq, sh, M_, N_, K_, epi, dt, st_in, in_cols, colsum, d->ln_eps, st_out,
                              nullptr, ws.split, ws.split_bytes, hs);
    };
    // a statistics producer (bias + residual) and its folded consumer (bias / bias + GELU), planned together over the slab widths the
    // pair can agree on; false: no valid pair (the LayerNorm stays a kernel)
    auto plan_pair = [&](int Mp, int Np, int Kp, int64_t ldap, int64_t ldcp, int64_t ldrp, int sharep, int Mc, int Nc, int Kc_, int64_t ldac,
                         int64_t ldcc, int epic, bool only256, AgWsPlan* prod, AgWsPlan* cons, int* cols) {
        double best = 1e30;
        for (int c = 256; c >= 128; c -= 128) {
            if (c == 128 && only256) continue;
            const AgWsPlan a = plan_r(Mp, Np, Kp, ldap, ldcp, ldrp, true, c == 256 ? 1 : 2, sharep);
            const AgWsPlan b = plan_w(Mc, Nc, Kc_, ldac, ldcc, epic, true, c);
            if (a.valid && b.valid && a.cost_us + b.cost_us < best) { best = a.cost_us + b.cost_us; *prod = a; *cons = b; *cols = c; }
        }
        return best < 1e30;
    };
    static AgKnob k_trim("AG_LAST_Q_TRIM");     // 0: the last layer projects queries for every token (A/B, parity tests)
    for (int l = 0; l < d->n_layers; ++l) {
        const ag_layer_weights& w = d->layers[l];
        const bool last_cls = cls_only_last && (l == d->n_layers - 1);
        const int Min = (R / in_share) * T;  // distinct rows entering this layer
        const int Mo = last_cls ? R : M;     // rows processed after attention: all tokens, or only token 0 of each row
        const int Tq = last_cls ? 1 : T;
        const int64_t ld_tok = last_cls ? (int64_t)T * H : H;  // stride between processed rows inside [R,T,H] buffers
        const bool trim_off = (int)k_trim.get(1) == 0;
        const bool chain_out = chain_stats && want_stats_out && l + 1 == d->n_layers;   // the consumer is the next CALL (same shape)
        // LayerNorm folding (ViT, bf16): the LN kernels disappear; row statistics come from the producing GEMM's epilogue (or
        // ag_row_stats_bf16 for the embeddings) and are applied in the consumer's.
        // -- attention input: ViT LN1(h_in) (pre-LN; Identity for explainer_attn.0) / BERT h_in itself --
        const int cols1 = fmt1 ? fmt1 : 256;   // (no statistics yet: ag_row_stats_bf16 writes 256-column slabs)
        AgWsPlan pq{}, pq_trim{};
        const bool can_fold1 = vit && bf && w.ln1_g && w.w_qkv_ln && !side_qkv;
        if (can_fold1) pq = plan_w(Min, 3 * H, H, H, 3 * H, AG_EPI_BIAS, true, cols1);
        const bool fold1 = can_fold1 && pq.valid;
        if (fold1 && last_cls && in_share == 1 && !trim_off) pq_trim = plan_w(Min, 2 * H, H, H, 3 * H, AG_EPI_BIAS, true, cols1);
        // ... and with only the CLS query read, the keys and values need not exist either (cls_last.hip): s_k = x_k . (W_k^T q) + b_k . q and
        // o = W_v (sum_k p_k x_k) + b_v — one pass over the layer's input rows instead of the K / V projection and its attention launch
        // AG_LAST_KV_SKIP = the rows from which the path is taken (default 64: below, its five short launches cost more than the projection of a few
        // thousand token rows — one input x 32 masks: 16.55 -> 16.45 k fwd/s, four inputs: 26.9 -> 27.2 k); 0: never (A/B, parity tests), 1: always
        static AgKnob k_kvskip("AG_LAST_KV_SKIP");
        const int kvskip_rows = (int)k_kvskip.get(64);
        const size_t q_bytes = align_up((size_t)R * H * es);
        const bool kv_skip = fold1 && pq_trim.valid && kvskip_rows > 0 && R >= kvskip_rows && ag_cls_last_supported(T, H, d->heads, dt) &&
                             (size_t)M * 3 * H * es >= q_bytes + ag_cls_last_scratch_bytes(R, H, d->heads);
        if (kv_skip) {
            if (!fmt1) TRY(ag_row_stats_bf16(h_in, H, Min, H, ws.st1, stream));
            TRY(ag_layernorm(h_in, dt, (int64_t)T * H, R, H, w.ln1_g, w.ln1_b, d->ln_eps, ws.xs, nullptr, dt, dyn, stream));
            // the CLS queries, compact [R, H], at the head of the (otherwise unused) qkv buffer; the scratch of the path behind them
            TRY(ag_gemm(ws.xs, H, w.w_qkv, w.b_qkv, ws.qkv, H, nullptr, 0, 0, 0, R, H, H, AG_EPI_BIAS, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
            TRY(ag_cls_last_attention(h_in, ws.st1, cols1, d_mask_bits, ws.qkv, (const char*)w.w_qkv_ln + (size_t)H * H * es, w.b_qkv_ln + H, d->ln_eps,
                                      ws.ctx, (int64_t)T * H, R, T, H, d->heads, ws.qkv + q_bytes, (size_t)M * 3 * H * es - q_bytes, hs));
        } else if (fold1 && pq_trim.valid) {
            // the last layer's attention reads the CLS query only (cls_only_last): keys and values of every token (the [H, 3H) rows of the
            // fused projection), queries of the R CLS rows — a third of this layer's QKV product is never computed.  The CLS rows are
            // normalised by the LayerNorm kernel (strided: one row per sequence) and projected with the unfolded query rows.
            if (!fmt1) TRY(ag_row_stats_bf16(h_in, H, Min, H, ws.st1, stream));
            TRY(run(pq_trim, h_in, H, (const char*)w.w_qkv_ln + (size_t)H * H * es, w.b_qkv_ln + H, ws.qkv + (size_t)H * es, 3 * H, nullptr, 0, 0, 0,
                    Min, 2 * H, H, AG_EPI_BIAS, ws.st1, cols1, w.s_qkv_ln + H, nullptr));
            TRY(ag_layernorm(h_in, dt, (int64_t)T * H, R, H, w.ln1_g, w.ln1_b, d->ln_eps, ws.xs, nullptr, dt, dyn, stream));
            TRY(ag_gemm(ws.xs, H, w.w_qkv, w.b_qkv, ws.qkv, (int64_t)T * 3 * H, nullptr, 0, 0, 0, R, H, H, AG_EPI_BIAS, dt,
                        nullptr, nullptr, 0.f, nullptr, dyn, stream));
        } else if (fold1) {
            if (!fmt1) TRY(ag_row_stats_bf16(h_in, H, Min, H, ws.st1, stream));
            TRY(run(pq, h_in, H, w.w_qkv_ln, w.b_qkv_ln, ws.qkv, 3 * H, nullptr, 0, 0, 0, Min, 3 * H, H, AG_EPI_BIAS, ws.st1, cols1, w.s_qkv_ln,
                    nullptr));
        } else if (side_qkv) {
            // narrow layer (LTT ladder): (LN1 +) QKV in one register-resident kernel
            TRY(ag_side_linear(h_in, H, Min, H, 3 * H, w.w_qkv, w.b_qkv, vit ? w.ln1_g : nullptr, vit ? w.ln1_b : nullptr,
                               nullptr, 0, nullptr, nullptr, d->ln_eps, ws.qkv, 3 * H, dyn, stream));
        } else {
            const char* att_in = h_in;
            if (vit && w.ln1_g) {
                TRY(ag_layernorm(h_in, dt, H, Min, H, w.ln1_g, w.ln1_b, d->ln_eps, ws.xs, nullptr, dt, dyn, stream));
                att_in = ws.xs;
            }
            TRY(run(plan_w(Min, 3 * H, H, H, 3 * H, AG_EPI_BIAS, false, 0), att_in, H, w.w_qkv, w.b_qkv, ws.qkv, 3 * H, nullptr, 0, 0, 0, Min,
                    3 * H, H, AG_EPI_BIAS, nullptr, 0, nullptr, nullptr));
        }
        fmt1 = 0;
        if (!kv_skip) TRY(ag_masked_attention(ws.qkv, d_mask_bits, ws.ctx, R, T, H, d->heads, in_share, d->kind, last_cls ? 1 : 0, dt, stream));

        // -- out-projection + residual(h_in) -> hx (compact [Mo,H]) --
        // narrow layer: out-proj + residual (+ BERT's attention-output LayerNorm) fused; needs row-aligned residual rows
        const bool side_o = side_lin_o && in_share == 1 && !last_cls;
        bool ln1_done = false;
        // ViT: LN2 folded into fc1 when the out-projection can hand on the statistics of what it writes
        AgWsPlan po{}, pf1{};
        int cols2 = 0;
        const bool fold2 = vit && bf && w.w_fc1_ln && !side_mlp && !side_o &&
                           plan_pair(Mo, H, H, ld_tok, H, ld_tok, in_share, Mo, I, H, H, I, AG_EPI_BIAS_GELU, false, &po, &pf1, &cols2);
        if (side_o) {
            const bool post = !vit && w.ln1_g;
            TRY(ag_side_linear(ws.ctx, H, Mo, H, H, w.w_o, w.b_o, nullptr, nullptr, h_in, H, post ? w.ln1_g : nullptr,
                               post ? w.ln1_b : nullptr, d->ln_eps, post ? ws.ha : ws.hx, H, dyn, stream));
            ln1_done = post;
        } else {
            if (!fold2) po = plan_r(Mo, H, H, ld_tok, H, ld_tok, false, 0, in_share);
            TRY(run(po, ws.ctx, ld_tok, w.w_o, w.b_o, ws.hx, H, h_in, ld_tok, Tq, in_share, Mo, H, H, AG_EPI_BIAS_RESID, nullptr, 0, nullptr,
                    fold2 ? ws.st2 : nullptr));
        }
        if (vit) {
            if (side_mlp && !fold2) {
                // narrow layer (LTT ladder): LN2 + fc1 + GELU + fc2 + residual in one register-resident kernel
                TRY(ag_side_mlp(ws.hx, H, Mo, H, I, w.w_fc1, w.b_fc1, w.w_fc2, w.b_fc2, w.ln2_g, w.ln2_b, d->ln_eps, 0, d_h, ld_tok, dyn, stream));
            } else {
            if (fold2) {
                TRY(run(pf1, ws.hx, H, w.w_fc1_ln, w.b_fc1_ln, ws.inter, I, nullptr, 0, 0, 0, Mo, I, H, AG_EPI_BIAS_GELU, ws.st2, cols2, w.s_fc1_ln,
                        nullptr));
            } else {
                TRY(ag_layernorm(ws.hx, dt, H, Mo, H, w.ln2_g, w.ln2_b, d->ln_eps, ws.xs, nullptr, dt, dyn, stream));
                TRY(run(plan_w(Mo, I, H, H, I, AG_EPI_BIAS_GELU, false, 0), ws.xs, H, w.w_fc1, w.b_fc1, ws.inter, I, nullptr, 0, 0, 0, Mo, I, H,
                        AG_EPI_BIAS_GELU, nullptr, 0, nullptr, nullptr));
            }
            // h_out = fc2(inter) + hx -> d_h (strided to token 0 when cls-only); will the NEXT layer fold its LN1?  then this fc2 hands on
            // the statistics of what it writes, in the slab width that layer's QKV projection reads
            AgWsPlan p2{}, pq_next{};
            int cols_next = 0;
            bool next_fold1 = false;
            if (bf && (l + 1 < d->n_layers || chain_out) && !last_cls) {
                const ag_layer_weights& wn = chain_out ? d->layers[l] : d->layers[l + 1];
                if (wn.ln1_g && wn.w_qkv_ln && !side_qkv) {
                    // (the last layer of a CLS-only forward projects keys and values only: see above)
                    const bool next_trim = !chain_out && cls_only_last && l + 2 == d->n_layers && !trim_off;
                    next_fold1 = plan_pair(Mo, H, I, I, ld_tok, H, 1, M, next_trim ? 2 * H : 3 * H, H, H, 3 * H, AG_EPI_BIAS,
                                           chain_stats != nullptr, &p2, &pq_next, &cols_next);
                    if (!next_fold1 && next_trim)
                        next_fold1 = plan_pair(Mo, H, I, I, ld_tok, H, 1, M, 3 * H, H, H, 3 * H, AG_EPI_BIAS, chain_stats != nullptr, &p2, &pq_next,
                                               &cols_next);
                }
            }
            if (!next_fold1) p2 = plan_r(Mo, H, I, I, ld_tok, H, false, 0, 1);
            TRY(run(p2, ws.inter, I, w.w_fc2, w.b_fc2, d_h, ld_tok, ws.hx, H, 1, 1, Mo, H, I, AG_EPI_BIAS_RESID, nullptr, 0, nullptr,
                    next_fold1 ? ws.st1 : nullptr));
            fmt1 = next_fold1 ? cols_next : 0;
            if (chain_out && stats_written) *stats_written = next_fold1 ? 1 : 0;
            }
        } else {
            const char* a = ws.hx;  // explainer_attn.0: attention.output.LayerNorm = Identity (models/vanilla_bert.py:107,:550-553)
            if (ln1_done) {
                a = ws.ha;
            } else if (w.ln1_g) {
                TRY(ag_layernorm(ws.hx, dt, H, Mo, H, w.ln1_g, w.ln1_b, d->ln_eps, ws.ha, nullptr, dt, dyn, stream));
                a = ws.ha;
            }
            AG_REQUIRE(w.ln2_g, "ag_encoder_forward: BERT output.LayerNorm missing in layer %d", l);
            if (side_mlp) {   // narrow layer (LTT ladder): fc1 + GELU + fc2 + residual + LN2 in one kernel
                char* dst = last_cls ? ws.ctx : (char*)d_h;
                TRY(ag_side_mlp(a, H, Mo, H, I, w.w_fc1, w.b_fc1, w.w_fc2, w.b_fc2, w.ln2_g, w.ln2_b, d->ln_eps, 1, dst, H, dyn, stream));
                if (last_cls) {
                    hipError_t e = hipMemcpy2DAsync(d_h, (size_t)T * H * es, ws.ctx, (size_t)H * es, (size_t)H * es, (size_t)R,
                                                    hipMemcpyDeviceToDevice, hs);
                    if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipMemcpy2DAsync: %s", hipGetErrorString(e));
                }
                h_in = (const char*)d_h;
                in_share = 1;
                continue;
            }
            TRY(run(plan_w(Mo, I, H, H, I, AG_EPI_BIAS_GELU, false, 0), a, H, w.w_fc1, w.b_fc1, ws.inter, I, nullptr, 0, 0, 0, Mo, I, H,
                    AG_EPI_BIAS_GELU, nullptr, 0, nullptr, nullptr));
            char* pre = (a == ws.hx) ? ws.ha : ws.hx;
            TRY(run(plan_r(Mo, H, I, I, H, H, false, 0, 1), ws.inter, I, w.w_fc2, w.b_fc2, pre, H, a, H, 1, 1, Mo, H, I, AG_EPI_BIAS_RESID, nullptr, 0,
                    nullptr, nullptr));
            if (last_cls) {
                // LayerNorm the compact [R,H] rows, then scatter them to token 0 of d_h
                TRY(ag_layernorm(pre, dt, H, Mo, H, w.ln2_g, w.ln2_b, d->ln_eps, ws.ctx, nullptr, dt, dyn, stream));
                hipError_t e = hipMemcpy2DAsync(d_h, (size_t)T * H * es, ws.ctx, (size_t)H * es, (size_t)H * es, (size_t)R,
                                                hipMemcpyDeviceToDevice, hs);
                if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipMemcpy2DAsync: %s", hipGetErrorString(e));
            } else {
                TRY(ag_layernorm(pre, dt, H, Mo, H, w.ln2_g, w.ln2_b, d->ln_eps, d_h, nullptr, dt, dyn, stream));
            }
        }
        h_in = (const char*)d_h;
        in_share = 1;
    }
    return AG_OK;
}

extern "C" int ag_encoder_forward(const ag_encoder_desc* d, const void* d_h0, int R, int share,
                                  const uint32_t* d_mask_bits, void* d_h, int cls_only_last,
                                  void* d_workspace, size_t workspace_bytes, void* stream) {
    return encoder_forward_impl(d, d_h0, R, share, d_mask_bits, d_h, cls_only_last, d_workspace, workspace_bytes, nullptr, false, false, nullptr, stream);
}

extern "C" int ag_encoder_forward_chained(const ag_encoder_desc* d, const void* d_h0, int R, int share,
                                          const uint32_t* d_mask_bits, void* d_h, int cls_only_last,
                                          void* d_workspace, size_t workspace_bytes, float* d_row_stats, int stats_in_ready,
                                          int want_stats_out, int* stats_written, void* stream) {
    AG_REQUIRE(d_row_stats && stats_written, "ag_encoder_forward_chained: null statistics buffer / flag");
    return encoder_forward_impl(d, d_h0, R, share, d_mask_bits, d_h, cls_only_last, d_workspace, workspace_bytes, d_row_stats,
                                stats_in_ready != 0, want_stats_out != 0, stats_written, stream);
}


// ---- BERT with token pruning ----------------------------------------------------------------------------------
// reference models/vanilla_bert.py:523: scores + (1 - mask) * finfo.min  =>  a masked key's soft-max weight is exactly
// 0 in every layer and every head.  The classifier reads the CLS row only (models/vanilla_bert.py:73-76), CLS is never
// masked, so the hidden states of masked tokens are never read by anything that reaches the output: they are dead
// rows.  Layer 0 still runs on all tokens (its LN/QKV are shared by the K masks of an input); after it the visible
// tokens of every row are packed (cu_seqlens) and layers 1.. run on the packed rows with a mask-free varlen
// attention; the last layer's out-projection / MLP run on the CLS rows only.  On Shapley-kernel masks half the
// players are off on average: half the GEMM rows, a quarter of the attention.
// Output contract = ag_encoder_forward(cls_only_last = 1): d_h [R,T,H] with token 0 of every row defined.
// The packed row count never leaves the device: launches are sized for the upper bound R*T and clamp to it at run time
// (their d_rows argument), so this entry neither synchronises nor allocates and can be captured into a hipGraph.
extern "C" int ag_bert_encoder_forward_pruned(const ag_encoder_desc* d, const void* d_h0, int R, int share,
                                              const uint32_t* d_mask_bits, void* d_h, void* d_workspace, size_t workspace_bytes,
                                              int* d_packed_rows_out, void* stream) {
    AG_REQUIRE(d && d_h0 && d_mask_bits && d_h && d_workspace, "ag_bert_encoder_forward_pruned: null pointer");
    AG_REQUIRE(d->kind == AG_MASK_BERT_ADD, "ag_bert_encoder_forward_pruned: only the additive (BERT) mask prunes exactly");
    AG_REQUIRE(d->n_layers >= 1 && d->layers, "ag_bert_encoder_forward_pruned: no layers");
    if (d->n_layers == 1) {
        if (d_packed_rows_out)   // (a 32-bit fill: asynchronous and legal under stream capture, unlike a copy from the stack)
            AG_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)d_packed_rows_out, R * d->T, 1, (hipStream_t)stream));
        return ag_encoder_forward(d, d_h0, R, share, d_mask_bits, d_h, 1, d_workspace, workspace_bytes, stream);
    }
    AG_REQUIRE(workspace_bytes >= ag_encoder_workspace_bytes(d, R), "ag_bert_encoder_forward_pruned: workspace too small");
    Ws ws;
    carve(d, R, (char*)d_workspace, &ws);
    const int T = d->T, H = d->H, I = d->I, dt = d->dtype;
    const size_t es = dtype_size(dt);
    hipStream_t hs = (hipStream_t)stream;
    // layer 0 on every token (shared LN/QKV), full output into d_h
    ag_encoder_desc d0 = *d;
    d0.n_layers = 1;
    TRY(ag_encoder_forward(&d0, d_h0, R, share, d_mask_bits, d_h, 0, d_workspace, workspace_bytes, stream));
    // plan + pack.  The packed row count N lives on the device (cu[R]); nothing is read back: the launches of the packed
    // section are sized for the upper bound R*T and get d_rows = cu + R, their kernels clamp to N.
    int* cu = ws.idx;
    int* tok_src = ws.idx + R + 1;
    TRY(ag_seq_compact_plan(d_mask_bits, R, T, cu, tok_src, stream));
    const int N = R * T;                   // upper bound of the packed rows
    const int* dN = cu + R;
    if (d_packed_rows_out) AG_HIP_CHECK(hipMemcpyAsync(d_packed_rows_out, dN, sizeof(int), hipMemcpyDeviceToDevice, hs));
    // LayerNorm-free packed section (bf16, ring-kernel shapes, folded weights present): from layer 1's out-projection on, the
    // stream holds PRE-LayerNorm rows h plus their slab statistics; LN(h) is folded into the GEMMs that consume it (QKV / fc1:
    // ag_gemm's d_ln_stats) and recomputed inside the epilogues that add it as the residual (ag_gemm_resid_ln).  No LayerNorm
    // output is written or read between layer 1's attention and the last layer's CLS rows: 2 of the 7 launches per layer
    // (and their 2 x N x H round trips) disappear.
    static const bool fold_off = getenv("AG_BERT_LN_FOLD") && atoi(getenv("AG_BERT_LN_FOLD")) == 0;
    bool fold = !fold_off && dt == AG_BF16 && d->n_layers >= 3 &&
                ag_gemm_supports_ln_fold(R * T, 3 * H, H, H, 3 * H, 0, AG_EPI_BIAS, dt) &&
                ag_gemm_supports_ln_fold(R * T, I, H, H, I, 0, AG_EPI_BIAS_GELU, dt) &&
                ag_gemm_supports_ln_fold(R * T, H, H, H, H, H, AG_EPI_BIAS_RESID, dt) &&
                ag_gemm_resid_ln_supported(R * T, H, H, H, H, H) && ag_gemm_resid_ln_supported(R * T, H, I, I, H, H);
    for (int l = 1; fold && l < d->n_layers; ++l) {
        const ag_layer_weights& w = d->layers[l];
        if (!w.ln1_g || !w.ln2_g || !w.w_fc1_ln || (l >= 2 && !w.w_qkv_ln)) fold = false;
    }
    const int* dyn = dN;                   // d_rows of the packed section's launches (NULL again for the last layer's CLS rows)
    // The Linears of the LayerNorm-free packed section are PLANNED (ag_ws_plan): launches sized for the bound N = R*T, priced for the rows
    // expected (Shapley-kernel masks leave about half of the players visible) — at the reference's batch sizes (one to four sequences x K
    // masks: 2-8 k packed rows) the N = 768 Linears are 24-48 tiles of 256 x 256 and run as 128 x 128 units x contraction ranges instead
    const int m_exp = R > (int)(0.55 * N) ? R : (int)(0.55 * N);
    auto plan_w = [&](int N_, int K_, int epi, bool fold_in) {
        return ag_ws_plan(N, N_, K_, K_, N_, 0, epi, dt, true, fold_in, 256, false, 0, 1, -1, 0, m_exp);
    };
    auto run = [&](const AgWsPlan& pl, const void* A, int64_t lda_, const void* W, const float* b, void* C, int64_t ldc_, const void* Rr, int64_t ldr_,
                   int N_, int K_, int epi, const float* st_in, const float* colsum, float* st_out) {
        return ag_gemm_ws_run(pl, A, lda_, W, b, C, ldc_, Rr, ldr_, 1, 1, N, N_, K_, epi, dt, st_in, 256, colsum, d->ln_eps, st_out, dN, ws.split,
                              ws.split_bytes, hs);
    };
    char* x = ws.xs;                       // packed stream entering the layer
    TRY(ag_gather_rows(d_h, H, tok_src, x, H, N, H, dt, dyn, stream));
    char* xn = (char*)d_h;                 // d_h is free again (only token 0 of each row is defined at exit): ping-pong
    bool x_pre = false;                    // fold: x holds pre-LN rows of the previous layer's output.LayerNorm, ws.st2 their statistics
    for (int l = 1; l < d->n_layers; ++l) {
        const ag_layer_weights& w = d->layers[l];
        const bool last = l == d->n_layers - 1;
        AG_REQUIRE(w.ln2_g, "ag_bert_encoder_forward_pruned: BERT output.LayerNorm missing in layer %d", l);
        const ag_layer_weights& wp = d->layers[l - 1];
        if (x_pre)
            TRY(run(plan_w(3 * H, H, AG_EPI_BIAS, true), x, H, w.w_qkv_ln, w.b_qkv_ln, ws.qkv, 3 * H, nullptr, 0, 3 * H, H, AG_EPI_BIAS, ws.st2, w.s_qkv_ln,
                    nullptr));
        else
            TRY(ag_gemm(x, H, w.w_qkv, w.b_qkv, ws.qkv, 3 * H, nullptr, 0, 0, 0, N, 3 * H, H, AG_EPI_BIAS, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
        TRY(ag_masked_attention_varlen(ws.qkv, cu, ws.ctx, R, T, H, d->heads, last ? 1 : 0, dt, stream));
        if (fold && !last) {
            // h1 = ctx Wo^T + bo + (l == 1 ? x : LN2_prev(x)) with statistics; inter = gelu(LN1(h1) W1^T + b1) folded;
            // h2 = inter W2^T + b2 + LN1(h1) recomputed, with statistics: the next layer's input
            if (x_pre)
                TRY(ag_gemm_resid_ln_ws(ws.ctx, H, w.w_o, w.b_o, ws.hx, H, x, H, ws.st2, wp.ln2_g, wp.ln2_b, d->ln_eps, N, H, H, ws.st1, dyn, m_exp, -1, 0,
                                        ws.split, ws.split_bytes, stream));
            else
                TRY(run(ag_ws_plan(N, H, H, H, H, H, AG_EPI_BIAS_RESID, dt, true, false, 0, true, 1, 1, -1, 0, m_exp), ws.ctx, H, w.w_o, w.b_o, ws.hx, H, x, H,
                        H, H, AG_EPI_BIAS_RESID, nullptr, nullptr, ws.st1));
            TRY(run(plan_w(I, H, AG_EPI_BIAS_GELU, true), ws.hx, H, w.w_fc1_ln, w.b_fc1_ln, ws.inter, I, nullptr, 0, I, H, AG_EPI_BIAS_GELU, ws.st1, w.s_fc1_ln,
                    nullptr));
            TRY(ag_gemm_resid_ln_ws(ws.inter, I, w.w_fc2, w.b_fc2, xn, H, ws.hx, H, ws.st1, w.ln1_g, w.ln1_b, d->ln_eps, N, H, I, ws.st2, dyn, m_exp, -1, 0,
                                    ws.split, ws.split_bytes, stream));
            char* t = x; x = xn; xn = t;
            x_pre = true;
            continue;
        }
        const char* ctx = ws.ctx;
        const char* res = x;
        int Mo = N;
        if (last) {   // CLS rows only: compact [R,H] copies of the attention output and of the residual; exact row count from here on
            dyn = nullptr;
            TRY(ag_gather_rows(ws.ctx, H, cu, ws.inter, H, R, H, dt, dyn, stream));
            char* rres = ws.inter + (size_t)R * H * es;
            if (x_pre) {   // the residual is LN2_prev of the (pre-LN) CLS rows: R rows, normalised here
                char* tmp = ws.inter + 2 * (size_t)R * H * es;
                TRY(ag_gather_rows(x, H, cu, tmp, H, R, H, dt, dyn, stream));
                TRY(ag_layernorm(tmp, dt, H, R, H, wp.ln2_g, wp.ln2_b, d->ln_eps, rres, nullptr, dt, dyn, stream));
            } else {
                TRY(ag_gather_rows(x, H, cu, rres, H, R, H, dt, dyn, stream));
            }
            ctx = ws.inter; res = rres; Mo = R;
        }
        TRY(ag_gemm(ctx, H, w.w_o, w.b_o, ws.hx, H, res, H, 1, 1, Mo, H, H, AG_EPI_BIAS_RESID, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
        const char* a = ws.hx;
        if (w.ln1_g) {
            TRY(ag_layernorm(ws.hx, dt, H, Mo, H, w.ln1_g, w.ln1_b, d->ln_eps, ws.ha, nullptr, dt, dyn, stream));
            a = ws.ha;
        }
        char* inter = last ? ws.qkv : ws.inter;   // (last: ws.inter holds the gathered CLS rows)
        TRY(ag_gemm(a, H, w.w_fc1, w.b_fc1, inter, I, nullptr, 0, 0, 0, Mo, I, H, AG_EPI_BIAS_GELU, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
        char* pre = (a == ws.hx) ? ws.ha : ws.hx;
        TRY(ag_gemm(inter, I, w.w_fc2, w.b_fc2, pre, H, a, H, 1, 1, Mo, H, I, AG_EPI_BIAS_RESID, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
        if (last) {
            TRY(ag_layernorm(pre, dt, H, Mo, H, w.ln2_g, w.ln2_b, d->ln_eps, ws.ctx, nullptr, dt, dyn, stream));
            hipError_t e = hipMemcpy2DAsync(d_h, (size_t)T * H * es, ws.ctx, (size_t)H * es, (size_t)H * es, (size_t)R, hipMemcpyDeviceToDevice, hs);
            if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipMemcpy2DAsync: %s", hipGetErrorString(e));
        } else {
            TRY(ag_layernorm(pre, dt, H, Mo, H, w.ln2_g, w.ln2_b, d->ln_eps, xn, nullptr, dt, dyn, stream));
            char* t = x; x = xn; xn = t;
        }
    }
    return AG_OK;
}


// BERT layers on PACKED rows (token pruning building block; the LTT-BERT ladder interleaves its map GEMMs between
// such calls): x [N, H] holds the visible tokens of R sequences, cu_seqlens [R+1] their ranges; every packed token is a
// visible key (mask-free varlen attention).  All tokens are processed (no CLS-only shortcut); out [N, H] must not alias x.
extern "C" int ag_bert_layers_forward_packed(const ag_encoder_desc* d, const void* d_x, const int* d_cu_seqlens, int R, int N,
                                             void* d_out, void* d_workspace, size_t workspace_bytes, const int* d_rows, void* stream) {
    const int* const dyn = d_rows;         // N is an upper bound when given: the kernels clamp to *d_rows
    AG_REQUIRE(d && d_x && d_cu_seqlens && d_out && d_workspace, "ag_bert_layers_forward_packed: null pointer");
    AG_REQUIRE(d->kind == AG_MASK_BERT_ADD, "ag_bert_layers_forward_packed: only the additive (BERT) mask prunes exactly");
    AG_REQUIRE(d->n_layers >= 1 && d->layers && R >= 1 && N >= R && N <= R * d->T, "ag_bert_layers_forward_packed: bad shape (R=%d N=%d)", R, N);
    AG_REQUIRE(workspace_bytes >= ag_encoder_workspace_bytes(d, R), "ag_bert_layers_forward_packed: workspace too small");
    AG_REQUIRE(d_out != d_x, "ag_bert_layers_forward_packed: out must not alias x");
    Ws ws;
    carve(d, R, (char*)d_workspace, &ws);
    const int T = d->T, H = d->H, I = d->I, dt = d->dtype;
    const char* x = (const char*)d_x;
    static const bool side_off = getenv("AG_SIDE_MLP") && atoi(getenv("AG_SIDE_MLP")) == 0;
    const bool side_mlp = !side_off && ag_side_mlp_supported(H, I, dt);
    const bool side_lin = !side_off && ag_side_linear_supported(H, 3 * H, 0, dt) && ag_side_linear_supported(H, H, 1, dt);
    // wide layers (the LTT backbone, one layer per call): the attention-output LayerNorm is never written — out-proj emits the
    // statistics of the pre-LN rows, fc1 folds LN1 into its epilogue, fc2 recomputes LN1(h1) as its residual (ag_gemm_resid_ln);
    // the layer's OUTPUT LayerNorm stays a kernel: the caller's taps and the next call read its result
    static const bool fold_off = getenv("AG_BERT_LN_FOLD") && atoi(getenv("AG_BERT_LN_FOLD")) == 0;
    const bool fold_ok = !fold_off && !side_lin && dt == AG_BF16 && ag_gemm_supports_ln_fold(N, I, H, H, I, 0, AG_EPI_BIAS_GELU, dt) &&
                         ag_gemm_supports_ln_fold(N, H, H, H, H, H, AG_EPI_BIAS_RESID, dt) && ag_gemm_resid_ln_supported(N, H, I, I, H, H);
    for (int l = 0; l < d->n_layers; ++l) {
        const ag_layer_weights& w = d->layers[l];
        AG_REQUIRE(w.ln2_g, "ag_bert_layers_forward_packed: BERT output.LayerNorm missing in layer %d", l);
        if (fold_ok && w.ln1_g && w.w_fc1_ln) {
            TRY(ag_gemm(x, H, w.w_qkv, w.b_qkv, ws.qkv, 3 * H, nullptr, 0, 0, 0, N, 3 * H, H, AG_EPI_BIAS, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
            TRY(ag_masked_attention_varlen(ws.qkv, d_cu_seqlens, ws.ctx, R, T, H, d->heads, 0, dt, stream));
            TRY(ag_gemm(ws.ctx, H, w.w_o, w.b_o, ws.hx, H, x, H, 1, 1, N, H, H, AG_EPI_BIAS_RESID, dt, nullptr, nullptr, 0.f, ws.st1, dyn, stream));
            TRY(ag_gemm(ws.hx, H, w.w_fc1_ln, w.b_fc1_ln, ws.inter, I, nullptr, 0, 0, 0, N, I, H, AG_EPI_BIAS_GELU, dt, ws.st1, w.s_fc1_ln, d->ln_eps,
                        nullptr, dyn, stream));
            TRY(ag_gemm_resid_ln(ws.inter, I, w.w_fc2, w.b_fc2, ws.ha, H, ws.hx, H, ws.st1, w.ln1_g, w.ln1_b, d->ln_eps, N, H, I, ws.st2, dyn, stream));
            char* dst_f = (l == d->n_layers - 1) ? (char*)d_out : ws.xs;
            TRY(ag_layernorm(ws.ha, dt, H, N, H, w.ln2_g, w.ln2_b, d->ln_eps, dst_f, nullptr, dt, dyn, stream));
            x = dst_f;
            continue;
        }
        const char* a = ws.hx;
        if (side_lin) {   // narrow layers (LTT ladder): QKV, and out-proj + residual + LayerNorm, as one kernel each
            TRY(ag_side_linear(x, H, N, H, 3 * H, w.w_qkv, w.b_qkv, nullptr, nullptr, nullptr, 0, nullptr, nullptr, d->ln_eps, ws.qkv, 3 * H, dyn, stream));
            TRY(ag_masked_attention_varlen(ws.qkv, d_cu_seqlens, ws.ctx, R, T, H, d->heads, 0, dt, stream));
            TRY(ag_side_linear(ws.ctx, H, N, H, H, w.w_o, w.b_o, nullptr, nullptr, x, H, w.ln1_g, w.ln1_b, d->ln_eps,
                               w.ln1_g ? ws.ha : ws.hx, H, dyn, stream));
            if (w.ln1_g) a = ws.ha;
        } else {
        TRY(ag_gemm(x, H, w.w_qkv, w.b_qkv, ws.qkv, 3 * H, nullptr, 0, 0, 0, N, 3 * H, H, AG_EPI_BIAS, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
        TRY(ag_masked_attention_varlen(ws.qkv, d_cu_seqlens, ws.ctx, R, T, H, d->heads, 0, dt, stream));
        TRY(ag_gemm(ws.ctx, H, w.w_o, w.b_o, ws.hx, H, x, H, 1, 1, N, H, H, AG_EPI_BIAS_RESID, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
        if (w.ln1_g) {
            TRY(ag_layernorm(ws.hx, dt, H, N, H, w.ln1_g, w.ln1_b, d->ln_eps, ws.ha, nullptr, dt, dyn, stream));
            a = ws.ha;
        }
        }
        // the last layer writes the caller's buffer; intermediate ones ping-pong through ws.xs (never an input of this loop)
        char* dst = (l == d->n_layers - 1) ? (char*)d_out : ws.xs;
        if (side_mlp) {
            TRY(ag_side_mlp(a, H, N, H, I, w.w_fc1, w.b_fc1, w.w_fc2, w.b_fc2, w.ln2_g, w.ln2_b, d->ln_eps, 1, dst, H, dyn, stream));
            x = dst;
            continue;
        }
        TRY(ag_gemm(a, H, w.w_fc1, w.b_fc1, ws.inter, I, nullptr, 0, 0, 0, N, I, H, AG_EPI_BIAS_GELU, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
        char* pre = (a == ws.hx) ? ws.ha : ws.hx;
        TRY(ag_gemm(ws.inter, I, w.w_fc2, w.b_fc2, pre, H, a, H, 1, 1, N, H, I, AG_EPI_BIAS_RESID, dt, nullptr, nullptr, 0.f, nullptr, dyn, stream));
        TRY(ag_layernorm(pre, dt, H, N, H, w.ln2_g, w.ln2_b, d->ln_eps, dst, nullptr, dt, dyn, stream));
        x = dst;
    }
    return AG_OK;
}
