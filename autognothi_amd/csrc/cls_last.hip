// cls_last.hip — the LAST layer's attention of a CLS-only ViT forward without its key / value projection.
//
// The heads of the surrogate read the CLS row only (reference models/vanilla_vit.py:51-56), so the last layer needs ONE query per
// row and head (encoder.cpp: cls_only_last).  With x_k = LayerNorm(h_k) the layer's input rows, q the CLS query of head a and
// (W_k, b_k), (W_v, b_v) that head's 64 rows of the key / value projections (models/vanilla_vit.py:436-465):
//     s_k = q . (W_k x_k + b_k) / 8          = x_k . (W_k^T q) / 8 + (b_k . q) / 8
//     o   = sum_k p_k (W_v x_k + b_v)        = W_v (sum_k p_k x_k) + b_v                    (sum_k p_k = 1)
// — the 2 M H^2 flops of the K / V projection (0.71 TFLOP for ViT-base at 1 536 rows: ~0.5 ms) and the [M, 2H] tensor its attention
// launch stages (0.73 GB: ~0.3 ms) become ONE pass over the layer's input rows h (M H elements), framed by two block-structured
// products that the persistent GEMM does in one round each:
//     Wt[(r, a), :] = gamma (.) W_k,a^T q_r,a       = Qexp[(r, a), :] . (gamma (.) W_k)          (Qexp: q_r masked to head a's 64 columns)
//     O[(r, a), :]  = (gamma (.) W_v) zhat_r,a + b_v'                                            (only columns [64 a, 64 a + 64) are read)
// with the LayerNorm folded as everywhere in the bf16 forward (x_k = rstd_k (h_k - mean_k) (.) gamma + beta, statistics from the producing
// GEMM's epilogue):  s_k = rstd_k (h_k . Wt - mean_k S) + q . b_k',  S = sum_i Wt_i,   zhat = sum_k p_k rstd_k (h_k - mean_k).
// ViT masking (scores * mask): a masked key has logit exactly 0 and keeps its value row.
//
// cls_attend_kernel: one workgroup (4 waves) per row r, all heads at once — a 16-"query" (the heads, padded) x H-wide attention over the
// row's T tokens, K = V = h: 16-token tiles through LDS (double buffered, global -> registers -> LDS), scores on v_mfma_f32_16x16x32_bf16
// with the contraction split over the waves (partials through LDS), online soft-max per head, zhat on v_mfma_f32_16x16x16_bf16 with the
// transposed operand by ds_read_b64_tr_b16, each wave owning H / 4 output features.
#include "common.h"

namespace {

constexpr float NEG_BIG = -3.0e38f;

struct ClsArgs {
    const bf16_t* h;        // [R*T, H] residual stream entering the layer
    const float* stats;     // [nslab][R*T][2] (sum, sumsq) per slab of `cols` columns
    long stats_slab;        // floats between slabs
    int nslab;
    const uint32_t* mask;   // [R, Tw]
    const bf16_t* wt;       // [R*heads, H]
    const bf16_t* q;        // [R, H] CLS queries
    const float* bk;        // [H] folded key bias (b_k + W_k beta)
    bf16_t* z;              // [R*heads, H]
    int R, T, heads, Tw;
    float eps, inv_h;
};

// x[l] op x[l^16] op x[l^32] op x[l^48] on every lane (the four lanes that hold one head's four token groups)
__device__ __forceinline__ float quad_max(float x) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int H>
__global__ __launch_bounds__(256) void cls_attend_kernel(ClsArgs p) {
    constexpr int RS = 2 * H + 16;            // LDS row stride in bytes (388 / 516 dwords: 16 rows x 16 B touch every bank once)
    constexpr int FW = H / 4;                 // output features per wave
    constexpr int NTILE = FW / 16;            // 16-feature MFMA tiles per wave
    constexpr int KS = FW / 32;               // score k-steps per wave
    constexpr int NLOAD = H / 128;            // 16-byte loads per thread and 16-token tile
    constexpr int CPR = H / 8;                // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wt_l = smem;                               // [16][RS]
    char* const x_l = smem + 16 * RS;                      // [2][16][RS]
    float* const sred = reinterpret_cast<float*>(smem + 48 * RS);          // [4][16 heads][16 tokens]
    float2* const tokst = reinterpret_cast<float2*>(sred + 4 * 256);       // [256] (mean, rstd) of every token of the row
    float* const hs = reinterpret_cast<float*>(tokst + 256);               // [16] S_h
    float* const hc = hs + 16;                                             // [16] c_h
    float* const xch = hc + 16;                                            // [4 waves][32] per-wave exchange (alpha | final scale, offset)
    uint32_t* const mwl = reinterpret_cast<uint32_t*>(xch + 128);          // [8] the row's mask words
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    const int T = p.T, heads = p.heads;
    const float c2 = 0.125f * 1.4426950408889634f;
    typedef __attribute__((ext_vector_type(4))) short s16x4;

    // ---- prologue: the row's Wt (heads x H, rows >= heads zero) -> LDS; S_h, c_h
    for (int e = tid; e < 16 * CPR; e += 256) {
        const int row = e / CPR, ch = e - row * CPR;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < heads) v = *reinterpret_cast<const uint4*>(p.wt + ((long)r * heads + row) * H + ch * 8);
        *reinterpret_cast<uint4*>(wt_l + row * RS + ch * 16) = v;
    }
    // tile loader: 16 tokens x H bf16, global -> registers one tile ahead -> LDS (two workgroups per CU take turns on the memory latency).  Rows
    // past T: the last token, finite data whose probability is exactly 0.  (Named registers, not arrays: hipcc kept a register array that lives across
    // the tile loop in scratch memory.  Two register sets alternating by tile parity, for two tiles in flight, were if-converted by hipcc into
    // unconditional loads + selects, i.e. a wait for the loads just issued.)
    uint4 xa0 = {}, xa1 = {}, xa2 = {}, xa3 = {}, xa4 = {}, xa5 = {}, xa6 = {}, xa7 = {};
#define CLS_FOR8(X, n) X(0, n##0) X(1, n##1) X(2, n##2) X(3, n##3) X(4, n##4) X(5, n##5) X(6, n##6) X(7, n##7)
#define CLS_LD(i, reg) if (i < NLOAD) { int t = tl0 + tile_row(i); t = t < T ? t : T - 1; reg = *reinterpret_cast<const uint4*>(p.h + ((long)r * T + t) * H + tile_ch(i) * 8); }
#define CLS_ST(i, reg) if (i < NLOAD) *reinterpret_cast<uint4*>(x_l + (bufs * 16 + tile_row(i)) * RS + tile_ch(i) * 16) = reg;
    const int ntile = (T + 15) >> 4;
    auto tile_row = [&](int i) { return (i * 256 + tid) / CPR; };    // (per-thread element coordinates of the NLOAD chunks: compile-time strides)
    auto tile_ch = [&](int i) { return (i * 256 + tid) % CPR; };
#define CLS_LOAD_TILE(n, tile_)                                                   \
    {                                                                             \
        const int tl0 = ((tile_) < ntile ? (tile_) : ntile - 1) * 16;             \
        CLS_FOR8(CLS_LD, n)                                                       \
    }
#define CLS_STORE_TILE(n, buf_)                                                   \
    {                                                                             \
        const int bufs = (buf_);                                                  \
        CLS_FOR8(CLS_ST, n)                                                       \
    }
    if (tid < 8) mwl[tid] = tid < p.Tw ? p.mask[(long)r * p.Tw + tid] : 0u;
    {   // (mean, rstd) of every token of the row from the slab partial sums, added in slab order: once, here — per tile, in front of the tile's loads,
        // hipcc waited for them with vmcnt(0): one full memory round trip per tile for the whole workgroup (171 us per launch)
        int t = tid < T ? tid : T - 1;
        const long m = (long)r * T + t;
        float sx = 0.f, sq = 0.f;
        for (int s = 0; s < p.nslab; ++s) {
            const float2 w = *reinterpret_cast<const float2*>(p.stats + s * p.stats_slab + 2 * m);
            sx += w.x; sq += w.y;
        }
        const float mean = sx * p.inv_h;
        tokst[tid] = make_float2(mean, rsqrtf(fmaxf(sq * p.inv_h - mean * mean, 0.f) + p.eps));
    }
    CLS_LOAD_TILE(xa, 0)
    __syncthreads();
    {   // S_h = sum_i Wt[h][i] (the bf16 values the score product reads), c_h = q_h . b_k'_h: 16 lanes per head
        const int head = tid >> 4, part = tid & 15;
        float s = 0.f, c = 0.f;
        for (int i = part; i < H; i += 16) s += bf16_to_f32(*reinterpret_cast<const bf16_t*>(wt_l + head * RS + i * 2));
        if (head < heads)
            for (int d = part; d < 64; d += 16) c += bf16_to_f32(p.q[(long)r * H + head * 64 + d]) * p.bk[head * 64 + d];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
        if (part == 0) { hs[head] = s; hc[head] = c; }
    }
    // this lane in the soft-max phase: head lh = lane & 15, tokens 4 tg .. 4 tg + 3 of the tile (tg = lane >> 4)
    const int lh = lane & 15, tg = lane >> 4;
    float m_run = NEG_BIG, l_run = 0.f, mb_run = 0.f;   // per head (identical m_run on the head's four lanes; l / mbar partial per lane)
    f32x4_t zacc[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; ++i) zacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float* const xw = xch + wave * 32;

    for (int it = 0; it < ntile; ++it) {
        const int buf = it & 1, t0 = it * 16;
        CLS_STORE_TILE(xa, buf)
        __syncthreads();                                   // tile `it` (and, first time, S_h / c_h / the token statistics) visible; sred of the previous tile fully read
        CLS_LOAD_TILE(xa, it + 1)                          // (unconditional: past the end the last tile again)
        // ---- scores: this wave's H/4 slice of the contraction
        {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const char* ap = wt_l + lh * RS + (wave * FW + 8 * tg) * 2;
            const char* bp = x_l + (buf * 16 + lh) * RS + (wave * FW + 8 * tg) * 2;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 a = *reinterpret_cast<const uint4*>(ap + ks * 64);
                const uint4 b = *reinterpret_cast<const uint4*>(bp + ks * 64);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
            }
            // C layout: lane holds heads 4 tg + i (rows) of token lh (column) -> sred[wave][head][token]
#pragma unroll
            for (int i = 0; i < 4; ++i) sred[wave * 256 + (4 * tg + i) * 16 + lh] = acc[i];
        }
        __syncthreads();
        // ---- soft-max of head lh over tokens t0 + 4 tg .. + 3 (every wave, redundantly: each needs the probabilities as its A operand)
        float pr[4];
        float alpha;
        {
            float4 raw = *reinterpret_cast<const float4*>(sred + lh * 16 + 4 * tg);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 u = *reinterpret_cast<const float4*>(sred + w * 256 + lh * 16 + 4 * tg);
                raw.x += u.x; raw.y += u.y; raw.z += u.z; raw.w += u.w;
            }
            const float rv[4] = {raw.x, raw.y, raw.z, raw.w};
            const float S = hs[lh], c = hc[lh];
            float a[4], rs[4], mn[4];
            float tmax = NEG_BIG;
            const uint32_t mword = mwl[(t0 >> 5) & 7] >> (t0 & 31);     // (a 16-token tile lies inside one mask word)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + 4 * tg + j;
                const float2 st = tokst[t < 256 ? t : 255];
                mn[j] = st.x; rs[j] = st.y;
                const bool vis = (mword >> (4 * tg + j)) & 1u;
                float s = vis ? fmaf(st.y, fmaf(-st.x, S, rv[j]), c) : 0.f;     // ViT: a masked key keeps logit 0
                a[j] = t < T ? s * c2 : NEG_BIG;
                tmax = fmaxf(tmax, a[j]);
            }
            tmax = quad_max(tmax);
            const float m_new = fmaxf(m_run, tmax);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);            // first tile: exp2(-3e38 - finite) = 0
            m_run = m_new;
            float ps = 0.f, pm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pv = __builtin_amdgcn_exp2f(a[j] - m_new);
                ps += pv;
                pr[j] = pv * rs[j];
                pm = fmaf(pr[j], mn[j], pm);
            }
            l_run = fmaf(l_run, alpha, ps);
            mb_run = fmaf(mb_run, alpha, pm);
            if (tg == 0) xw[lh] = alpha;
        }
        // ---- zhat accumulation: Z[head][feature] = alpha Z + sum_tokens P'[head][token] h[token][feature], features of this wave
        {
            const float4 al = *reinterpret_cast<const float4*>(xw + 4 * tg);   // alpha of heads 4 tg .. + 3 (this wave wrote them: in order)
            s16x4 pa;
            {
                const uint32_t lo = pack_bf16x2(pr[0], pr[1]), hi = pack_bf16x2(pr[2], pr[3]);
                pa = __builtin_bit_cast(s16x4, make_uint2(lo, hi));
            }
            const int li = lane & 15, tq = li >> 2, tp = li & 3;
            const char* vp = x_l + (buf * 16 + 4 * tg + tq) * RS + (wave * FW + 4 * tp) * 2;
#pragma unroll
            for (int nt = 0; nt < NTILE; ++nt) {
                const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + nt * 32));
                f32x4_t z = zacc[nt];
                z[0] *= al.x; z[1] *= al.y; z[2] *= al.z; z[3] *= al.w;
                zacc[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, vb, z, 0, 0, 0);
            }
        }
    }
    // ---- zhat = (Z - mbar) / l per head; the head's l and mbar are spread over its four lanes
    {
        float l = l_run, mb = mb_run;
        {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(l), __float_as_uint(l), false, false);
            l = __uint_as_float(a[0]) + __uint_as_float(a[1]);
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
            l = __uint_as_float(b[0]) + __uint_as_float(b[1]);
            const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(mb), __float_as_uint(mb), false, false);
            mb = __uint_as_float(c[0]) + __uint_as_float(c[1]);
            const auto d = __builtin_amdgcn_permlane32_swap(__float_as_uint(mb), __float_as_uint(mb), false, false);
            mb = __uint_as_float(d[0]) + __uint_as_float(d[1]);
        }
        const float inv = 1.0f / l;
        if (tg == 0) { xw[lh] = inv; xw[16 + lh] = mb; }
        const float4 iv = *reinterpret_cast<const float4*>(xw + 4 * tg);
        const float4 mv = *reinterpret_cast<const float4*>(xw + 16 + 4 * tg);
        const float ivv[4] = {iv.x, iv.y, iv.z, iv.w}, mvv[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
        for (int nt = 0; nt < NTILE; ++nt) {
            const int f = wave * FW + nt * 16 + lh;         // C layout: column = feature, rows = heads 4 tg + i
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int head = 4 * tg + i;
                if (head < heads) p.z[((long)r * heads + head) * H + f] = f32_to_bf16((zacc[nt][i] - mvv[i]) * ivv[i]);
            }
        }
    }
}

// Qexp[(r, a), j] = q[r, j] for j in head a's 64 columns, else 0
__global__ __launch_bounds__(256) void expand_heads_kernel(const bf16_t* __restrict__ q, bf16_t* __restrict__ qe, int R, int heads, int H) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;         // one 16-byte chunk (8 columns) each
    const int cpr = H / 8;
    const long row = e / cpr;
    if (row >= (long)R * heads) return;
    const int ch = (int)(e - row * cpr);
    const int r = (int)(row / heads), a = (int)(row - (long)r * heads);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((ch * 8) / 64 == a) v = *reinterpret_cast<const uint4*>(q + (long)r * H + ch * 8);
    *reinterpret_cast<uint4*>(qe + row * H + ch * 8) = v;
}

// dst[i][j] = src[j][i], n x n bf16, 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_sq_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int n) {
    __shared__ bf16_t tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = src[(long)(by + i) * n + bx + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) dst[(long)(bx + i) * n + by + tx] = tile[tx][i];
}

// ctx[r, 0, 64 a + d] = O[(r, a), 64 a + d]
__global__ __launch_bounds__(256) void gather_heads_kernel(const bf16_t* __restrict__ o, bf16_t* __restrict__ ctx, long ctx_row_stride, int R, int heads, int H) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;         // one 16-byte chunk of a CLS row
    const int cpr = H / 8;
    const long r = e / cpr;
    if (r >= R) return;
    const int ch = (int)(e - r * cpr);
    const int a = (ch * 8) / 64;
    *reinterpret_cast<uint4*>(ctx + r * ctx_row_stride + ch * 8) = *reinterpret_cast<const uint4*>(o + (r * heads + a) * H + ch * 8);
}

}  // namespace

bool ag_cls_last_supported(int T, int H, int heads, int dtype) {
    return dtype == AG_BF16 && (H == 768 || H == 1024) && heads * 64 == H && heads <= 16 && T >= 2 && T <= 256;
}

// bytes of scratch: Qexp | Wt | Z | O ([R*heads, H] bf16 each) + the transposed key block [H, H]
size_t ag_cls_last_scratch_bytes(int R, int H, int heads) {
    return 4 * (((size_t)R * heads * H * 2 + 255) & ~(size_t)255) + (size_t)H * H * 2 + 256;
}

// d_h [R*T, H] bf16 with row statistics d_stats (slabs of `cols` columns), d_q [R, H] the CLS queries, w_kv_ln the gamma-folded key | value
// rows of the fused projection ([2H, H]), b_kv_ln their folded biases ([2H]) -> the attention output of the CLS rows, written to
// d_ctx + r * ctx_row_stride (elements)
int ag_cls_last_attention(const void* d_h, const float* d_stats, int cols, const uint32_t* d_mask_bits, const void* d_q, const void* w_kv_ln,
                          const float* b_kv_ln, float ln_eps, void* d_ctx, int64_t ctx_row_stride, int R, int T, int H, int heads,
                          void* d_scratch, size_t scratch_bytes, hipStream_t s) {
    AG_REQUIRE(ag_cls_last_supported(T, H, heads, AG_BF16) && (cols == 256 || cols == 128), "ag_cls_last_attention: unsupported shape");
    AG_REQUIRE(scratch_bytes >= ag_cls_last_scratch_bytes(R, H, heads), "ag_cls_last_attention: scratch too small");
    const size_t blk = ((size_t)R * heads * H * 2 + 255) & ~(size_t)255;
    char* base = (char*)d_scratch;
    bf16_t* qe = (bf16_t*)base; bf16_t* wt = (bf16_t*)(base + blk); bf16_t* z = (bf16_t*)(base + 2 * blk); bf16_t* o = (bf16_t*)(base + 3 * blk);
    bf16_t* wkt = (bf16_t*)(base + 4 * blk);
    const int Mh = R * heads;
    {
        const long chunks = (long)Mh * (H / 8);
        hipLaunchKernelGGL(expand_heads_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const bf16_t*)d_q, qe, R, heads, H);
        hipLaunchKernelGGL(transpose_sq_kernel, dim3(H / 32, H / 32), dim3(256), 0, s, (const bf16_t*)w_kv_ln, wkt, H);
        AG_LAUNCH_CHECK();
    }
    // Wt = Qexp . (gamma (.) W_k): NT product against the transposed key block
    int rc = ag_gemm(qe, H, wkt, nullptr, wt, H, nullptr, 0, 0, 0, Mh, H, H, AG_EPI_BIAS, AG_BF16, nullptr, nullptr, 0.f, nullptr, nullptr, s);
    if (rc != AG_OK) return rc;
    ClsArgs a;
    a.h = (const bf16_t*)d_h; a.stats = d_stats; a.stats_slab = 2L * R * T; a.nslab = (H + cols - 1) / cols;
    a.mask = d_mask_bits; a.wt = wt; a.q = (const bf16_t*)d_q; a.bk = b_kv_ln; a.z = z;
    a.R = R; a.T = T; a.heads = heads; a.Tw = (T + 31) / 32; a.eps = ln_eps; a.inv_h = 1.0f / (float)H;
    {
        const int RS = 2 * H + 16;
        const int lds = 48 * RS + 4 * 256 * 4 + 256 * 8 + 32 * 4 + 4 * 32 * 4 + 32;
        static bool attr_set[16][2] = {};
        int dev = 0;
        AG_HIP_CHECK(hipGetDevice(&dev));
        AG_REQUIRE(dev >= 0 && dev < 16, "ag_cls_last_attention: device index %d", dev);
        // (accounted with the attention launches it replaces: 2 x 2 T H flops per row and head, one pass over the layer's input rows)
        AgProfScope prof(AG_PROF_ATTENTION, 4.0 * R * (double)heads * T * H, ((double)R * T * H + 2.0 * R * heads * H) * 2.0, s);
        if (H == 768) {
            if (!attr_set[dev][0]) { AG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cls_attend_kernel<768>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr_set[dev][0] = true; }
            hipLaunchKernelGGL(cls_attend_kernel<768>, dim3(R), dim3(256), lds, s, a);
        } else {
            if (!attr_set[dev][1]) { AG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cls_attend_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr_set[dev][1] = true; }
            hipLaunchKernelGGL(cls_attend_kernel<1024>, dim3(R), dim3(256), lds, s, a);
        }
        AG_LAUNCH_CHECK();
    }
    // O = Z . (gamma (.) W_v)^T + b_v'
    rc = ag_gemm(z, H, (const char*)w_kv_ln + (size_t)H * H * 2, b_kv_ln + H, o, H, nullptr, 0, 0, 0, Mh, H, H, AG_EPI_BIAS, AG_BF16, nullptr, nullptr, 0.f,
                 nullptr, nullptr, s);
    if (rc != AG_OK) return rc;
    {
        const long chunks = (long)R * (H / 8);
        hipLaunchKernelGGL(gather_heads_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, o, (bf16_t*)d_ctx, (long)ctx_row_stride, R, heads, H);
        AG_LAUNCH_CHECK();
    }
    return AG_OK;
}

// C ABI (include/autognothi_hip.h): the kernel-level entry of the path, for the parity tests
extern "C" int ag_cls_last_is_supported(int T, int H, int heads, int dtype) { return ag_cls_last_supported(T, H, heads, dtype) ? 1 : 0; }
extern "C" size_t ag_cls_last_workspace_bytes(int R, int H, int heads) { return ag_cls_last_scratch_bytes(R, H, heads); }
extern "C" int ag_cls_last_attention_rows(const void* d_h, const float* d_stats, int cols, const uint32_t* d_mask_bits, const void* d_q,
                                          const void* w_kv_ln, const float* b_kv_ln, float ln_eps, void* d_ctx, int64_t ctx_row_stride, int R,
                                          int T, int H, int heads, void* d_scratch, size_t scratch_bytes, void* stream) {
    return ag_cls_last_attention(d_h, d_stats, cols, d_mask_bits, d_q, w_kv_ln, b_kv_ln, ln_eps, d_ctx, ctx_row_stride, R, T, H, heads, d_scratch,
                                 scratch_bytes, (hipStream_t)stream);
}
