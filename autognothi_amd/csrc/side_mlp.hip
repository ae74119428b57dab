// side_mlp.hip — the MLP half of a NARROW transformer layer in ONE kernel (bf16):
//     ViT block  (pre-LN):   out = x + fc2(gelu(fc1(LN(x))))
//     BERT block (post-LN):  out = LN(x + fc2(gelu(fc1(x))))
// for hidden widths h <= 128 — the ladder side network of the LTT recipes (h = 96, I = 384: reference
// models/ltt_vit.py:383-394 builds the side layers as VanillaViTLayer(hidden=s_attn_hidden_size), models/ltt_bert.py:440-455
// as VanillaBertLayer; the layer math is models/vanilla_vit.py:373-376,:491-492,:510-512 / models/vanilla_bert.py:576-577,
// :601-603).  On 300 k rows of 96 features each of the three kernels it replaces (LayerNorm, fc1+GELU, fc2+residual) is a
// launch that moves 58-290 MB for a few MFLOP per row: latency- and HBM-bound at 90-165 us apiece, for 4.4 GFLOP/us of work.
//
// Structure: row-local, register-resident.  Both weight matrices live in LDS for the whole launch (h = 96, I = 384:
// 2 x 74 KB + padding = 152 KiB of the 160 KiB); one workgroup per CU, whose 16 waves each take 32-row chunks
// and never talk to one another: no barrier after the weight staging.  Everything is computed TRANSPOSED
//     F^T[I x rows] = gelu(W1[I x h] . X^T[h x rows] + b1),        OUT^T[h x rows] = W2[h x I] . F^T + b2
// so that (1) the activations are the MFMA B operand, whose lane layout (lane = row, 8 consecutive features) is a plain
// 16-byte load of a row segment, (2) a 16x16 fp32 result tile of fc1 — feature index on the registers, row on the lane — is,
// after GELU and bf16 packing, directly the B operand of fc2: two consecutive result tiles make one 32-deep k-step, in the
// permuted k order (4q+r | 16+4q+r) that the W2 image in LDS is pre-arranged for.  The 384-wide intermediate never exists
// outside registers; HBM sees x once and out once.
// LayerNorm statistics: a row's 96 features sit in the four lanes {c, c+16, c+32, c+48}: v_permlane16/32_swap sums
// (quad_rows_sum), no LDS.
#include "common.h"

namespace {

constexpr int NT = 512;                  // 8 waves, 2 per SIMD, up to 256 VGPRs each: room to keep a slab's weight fragments in
                                         // flight (at 16 waves / 128 VGPRs hipcc funnels every fragment through one register
                                         // quad, read -> wait -> two MFMAs -> read ..., and spills)
constexpr int ROWS_PER_WAVE = 32;        // two 16-row blocks share every weight fragment read
constexpr int TILE_ROWS = (NT / 64) * ROWS_PER_WAVE;

struct SideArgs {
    const bf16_t* x; long ldx;           // [M, h], row stride in elements
    bf16_t* out; long ldo;
    const bf16_t* w1; const float* b1;   // [I, h], [I]
    const bf16_t* w2; const float* b2;   // [h, I], [h]
    const float* g; const float* be;     // LayerNorm gamma / beta [h]
    float eps;
    int M, h, I, post_ln;
    const int* dyn;                      // ag_dynamic_rows
};

__device__ __forceinline__ f32x4_t mfma(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// H = hidden width (template: fragment counts are register array sizes), I runtime (multiple of 32)
template <int H>
__global__ __launch_bounds__(NT, 2) void side_mlp_kernel(SideArgs p) {
    constexpr int KS = H / 32;           // k-steps of fc1
    constexpr int OB = H / 16;           // 16-feature output blocks of fc2
    constexpr int W1_ROWB = H * 2 + 16;  // padded LDS rows: 16 consecutive rows hit 16 different 16-byte bank groups
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int I = p.I;
    const int W2_ROWB = I * 2 + 16;
    char* const lw1 = smem;                                   // [I][W1_ROWB]
    char* const lw2 = lw1 + (size_t)I * W1_ROWB;              // [H][W2_ROWB], k order permuted inside every 32-block
    float* const lb1 = reinterpret_cast<float*>(lw2 + (size_t)H * W2_ROWB);   // [I]
    float* const lb2 = lb1 + I;                               // [H]
    float* const lg = lb2 + H;                                // [H]
    float* const lbe = lg + H;                                // [H]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int M = ag_dyn_clamp(p.M, p.dyn);

    // ---- stage the weights once per workgroup ----
    for (int i = tid; i < I * (H / 8); i += NT) {             // W1: 16-byte pieces, row-major
        const int n = i / (H / 8), pc = i - n * (H / 8);
        *reinterpret_cast<uint4*>(lw1 + (size_t)n * W1_ROWB + pc * 16) = *reinterpret_cast<const uint4*>(p.w1 + (size_t)n * H + pc * 8);
    }
    // W2 image: inside every 32-wide block of the contraction index the 8 values lane group q needs for one k-step —
    // features {4q..4q+3} of the even 16-block and of the odd 16-block — are made contiguous: dst[8q + j] = src[(j<4 ? 4q+j : 16+4q+j-4)]
    for (int i = tid; i < H * (I / 4); i += NT) {             // 8-byte pieces (4 features)
        const int o = i / (I / 4), pc = i - o * (I / 4);       // pc: 4-feature piece of the source row
        const int blk = pc >> 3, in = pc & 7;                  // 32-block, piece inside it (0..7)
        const int half = in >> 2, qq = in & 3;                 // source: half (even/odd 16-block), 4-group qq
        const uint2 v = *reinterpret_cast<const uint2*>(p.w2 + (size_t)o * I + pc * 4);
        *reinterpret_cast<uint2*>(lw2 + (size_t)o * W2_ROWB + (blk * 32 + qq * 8 + half * 4) * 2) = v;
    }
    for (int i = tid; i < I; i += NT) lb1[i] = p.b1 ? p.b1[i] : 0.f;
    for (int i = tid; i < H; i += NT) {
        lb2[i] = p.b2 ? p.b2[i] : 0.f;
        lg[i] = p.g ? p.g[i] : 1.f;
        lbe[i] = p.be ? p.be[i] : 0.f;
    }
    __syncthreads();

    // 32-row chunks are dealt to the waves of the whole grid round-robin (waves never synchronise after the staging), so the
    // chip is balanced to within one chunk per wave whatever M is
    const int nchunks = (M + ROWS_PER_WAVE - 1) / ROWS_PER_WAVE;
    const float inv_h = 1.0f / (float)H;
    for (int chunk = blockIdx.x * (NT / 64) + wave; chunk < nchunks; chunk += gridDim.x * (NT / 64)) {
        const int row0 = chunk * ROWS_PER_WAVE;
        // ---- load this wave's 2 x 16 rows as B fragments: lane (q, c) holds x[row c][32 s + 8 q .. + 8], s < KS ----
        uint4 xf[2][KS];
        int rowc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            int r = row0 + rb * 16 + c;
            rowc[rb] = r < M ? r : M - 1;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                xf[rb][s] = *reinterpret_cast<const uint4*>(p.x + (size_t)rowc[rb] * p.ldx + s * 32 + q * 8);
        }
        // ---- pre-LN (ViT): normalise the fragments in place ----
        if (!p.post_ln && p.g) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float v[KS][8];
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const uint32_t w[4] = {xf[rb][s].x, xf[rb][s].y, xf[rb][s].z, xf[rb][s].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[s][2 * e] = __uint_as_float(w[e] << 16);
                        v[s][2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
                        sum += v[s][2 * e] + v[s][2 * e + 1];
                    }
                }
                const float mean = quad_rows_sum(sum) * inv_h;
                float sq = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[s][e] - mean; sq = fmaf(d, d, sq); }
                const float rstd = rsqrtf(quad_rows_sum(sq) * inv_h + p.eps);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int k0 = s * 32 + q * 8;
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = fmaf((v[s][e] - mean) * rstd, lg[k0 + e], lbe[k0 + e]);
                    xf[rb][s] = make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7]));
                }
            }
        }
        // ---- fc1 -> GELU -> fc2, one 32-feature slab of the intermediate at a time ----
        f32x4_t acc[2][OB];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) acc[rb][ob] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // Software pipeline, pinned with scheduling fences: the W2 fragments of slab sl are requested before its GELU (their LDS
        // latency rides under the VALU work), the W1 fragments of slab sl + 1 before its fc2 MFMAs.
        auto load_w1 = [&](uint4 (&a)[2][KS], int sl) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_)
                    a[hb][s_] = *reinterpret_cast<const uint4*>(lw1 + (size_t)(sl * 32 + hb * 16 + c) * W1_ROWB + (s_ * 32 + q * 8) * 2);
        };
        uint4 a1[2][KS];
        load_w1(a1, 0);
        const int nsl = I / 32;
        for (int sl = 0; sl < nsl; ++sl) {
            f32x4_t d[2][2];     // [row block][even / odd 16-feature block]
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const float4 bias = *reinterpret_cast<const float4*>(lb1 + sl * 32 + hb * 16 + 4 * q);   // features n0 + 4q + r
                d[0][hb] = f32x4_t{bias.x, bias.y, bias.z, bias.w};
                d[1][hb] = d[0][hb];
            }
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    d[0][hb] = mfma(a1[hb][s_], xf[0][s_], d[0][hb]);
                    d[1][hb] = mfma(a1[hb][s_], xf[1][s_], d[1][hb]);
                }
            uint4 a2[OB];
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
                a2[ob] = *reinterpret_cast<const uint4*>(lw2 + (size_t)(ob * 16 + c) * W2_ROWB + (sl * 32 + q * 8) * 2);
            __builtin_amdgcn_sched_barrier(0);
            uint4 ff[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const f32x2_t g0 = fast_gelu2(f32x2_t{d[rb][0][0], d[rb][0][1]}), g1 = fast_gelu2(f32x2_t{d[rb][0][2], d[rb][0][3]});
                const f32x2_t g2 = fast_gelu2(f32x2_t{d[rb][1][0], d[rb][1][1]}), g3 = fast_gelu2(f32x2_t{d[rb][1][2], d[rb][1][3]});
                ff[rb] = make_uint4(pack_bf16x2(g0.x, g0.y), pack_bf16x2(g1.x, g1.y), pack_bf16x2(g2.x, g2.y), pack_bf16x2(g3.x, g3.y));
            }
            if (sl + 1 < nsl) load_w1(a1, sl + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                acc[0][ob] = mfma(a2[ob], ff[0], acc[0][ob]);
                acc[1][ob] = mfma(a2[ob], ff[1], acc[1][ob]);
            }
        }
        // ---- epilogue: + b2 + residual (x in the accumulator layout: row c, features 16 ob + 4 q + r), post-LN (BERT), store ----
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float o[OB][4];
            float sum = 0.f;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                const int f0 = ob * 16 + 4 * q;
                const float4 xr = load4_as_f32(p.x + (size_t)rowc[rb] * p.ldx + f0);
                const float4 b2 = *reinterpret_cast<const float4*>(lb2 + f0);
                o[ob][0] = acc[rb][ob][0] + b2.x + xr.x; o[ob][1] = acc[rb][ob][1] + b2.y + xr.y;
                o[ob][2] = acc[rb][ob][2] + b2.z + xr.z; o[ob][3] = acc[rb][ob][3] + b2.w + xr.w;
                sum += (o[ob][0] + o[ob][1]) + (o[ob][2] + o[ob][3]);
            }
            if (p.post_ln) {
                const float mean = quad_rows_sum(sum) * inv_h;
                float sq = 0.f;
#pragma unroll
                for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float dd = o[ob][e] - mean; sq = fmaf(dd, dd, sq); }
                const float rstd = rsqrtf(quad_rows_sum(sq) * inv_h + p.eps);
#pragma unroll
                for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int f = ob * 16 + 4 * q + e;
                        o[ob][e] = fmaf((o[ob][e] - mean) * rstd, lg[f], lbe[f]);
                    }
            }
            const int r = row0 + rb * 16 + c;
            if (r < M) {
#pragma unroll
                for (int ob = 0; ob < OB; ++ob)
                    *reinterpret_cast<uint2*>(p.out + (size_t)r * p.ldo + ob * 16 + 4 * q) =
                        make_uint2(pack_bf16x2(o[ob][0], o[ob][1]), pack_bf16x2(o[ob][2], o[ob][3]));
            }
        }
    }
}

// ---- one Linear of a narrow layer with its neighbours fused -------------------------------------------------------------
//     out[M, N] = LN_post( resid + W . LN_pre(x) + b )          (each of LN_pre, resid, LN_post optional)
// N <= 384 features out of h <= 128: the attention half of a side layer — LN1 + QKV (ViT, models/vanilla_vit.py:369,:437-441),
// QKV alone (BERT), out-proj + residual (ViT :372,:477) and out-proj + residual + LN1 (BERT, models/vanilla_bert.py:557-559).
// Same transposed, register-resident scheme as the MLP kernel; W (<= 80 KB) stays in LDS, two workgroups per CU.  The rows of W
// are laid out in LDS so that an MFMA lane ends up with 8 CONSECUTIVE output features of a row (fragment row i of 16-block ob
// holds feature 32 (ob >> 1) + 8 (i >> 2) + 4 (ob & 1) + (i & 3)): residual loads and output stores are 16 bytes per lane.
struct LinArgs {
    const bf16_t* x; long ldx;
    const bf16_t* w; const float* b;     // [N, h], [N]
    const bf16_t* resid; long ldr;       // [M, N] or NULL
    bf16_t* out; long ldo;
    const float* g0; const float* b0;    // LN_pre gamma / beta [h] or NULL
    const float* g1; const float* b1;    // LN_post gamma / beta [N] or NULL (N <= 128)
    float eps;
    int M, h, N;
    const int* dyn;
};

template <int H, bool POST>
__global__ __launch_bounds__(NT, 2) void side_linear_kernel(LinArgs p) {
    constexpr int KS = H / 32;
    constexpr int W_ROWB = H * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = p.N;
    char* const lw = smem;                                              // [N][W_ROWB], rows in fragment order
    float* const lb = reinterpret_cast<float*>(lw + (size_t)N * W_ROWB);   // [N] bias, feature order
    float* const lg0 = lb + N;                                          // [H]
    float* const lb0 = lg0 + H;                                         // [H]
    float* const lg1 = lb0 + H;                                         // [N]
    float* const lb1 = lg1 + N;                                         // [N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int M = ag_dyn_clamp(p.M, p.dyn);
    for (int i = tid; i < N * (H / 8); i += NT) {
        const int dr = i / (H / 8), pc = i - dr * (H / 8);             // destination row = 16-block ob, fragment row fi
        const int ob = dr >> 4, fi = dr & 15;
        const int feat = 32 * (ob >> 1) + 8 * (fi >> 2) + 4 * (ob & 1) + (fi & 3);
        *reinterpret_cast<uint4*>(lw + (size_t)dr * W_ROWB + pc * 16) = *reinterpret_cast<const uint4*>(p.w + (size_t)feat * H + pc * 8);
    }
    for (int i = tid; i < N; i += NT) {
        lb[i] = p.b ? p.b[i] : 0.f;
        lg1[i] = p.g1 ? p.g1[i] : 1.f;
        lb1[i] = p.b1 ? p.b1[i] : 0.f;
    }
    for (int i = tid; i < H; i += NT) {
        lg0[i] = p.g0 ? p.g0[i] : 1.f;
        lb0[i] = p.b0 ? p.b0[i] : 0.f;
    }
    __syncthreads();
    const int nchunks = (M + ROWS_PER_WAVE - 1) / ROWS_PER_WAVE;
    const float inv_h = 1.0f / (float)H, inv_n = 1.0f / (float)N;
    for (int chunk = blockIdx.x * (NT / 64) + wave; chunk < nchunks; chunk += gridDim.x * (NT / 64)) {
        const int row0 = chunk * ROWS_PER_WAVE;
        uint4 xf[2][KS];
        int rowc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int r = row0 + rb * 16 + c;
            rowc[rb] = r < M ? r : M - 1;
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
                xf[rb][s_] = *reinterpret_cast<const uint4*>(p.x + (size_t)rowc[rb] * p.ldx + s_ * 32 + q * 8);
        }
        if (p.g0) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float v[KS][8];
                float sum = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    const uint32_t w_[4] = {xf[rb][s_].x, xf[rb][s_].y, xf[rb][s_].z, xf[rb][s_].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[s_][2 * e] = __uint_as_float(w_[e] << 16);
                        v[s_][2 * e + 1] = __uint_as_float(w_[e] & 0xFFFF0000u);
                        sum += v[s_][2 * e] + v[s_][2 * e + 1];
                    }
                }
                const float mean = quad_rows_sum(sum) * inv_h;
                float sq = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float dd = v[s_][e] - mean; sq = fmaf(dd, dd, sq); }
                const float rstd = rsqrtf(quad_rows_sum(sq) * inv_h + p.eps);
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    const int k0 = s_ * 32 + q * 8;
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = fmaf((v[s_][e] - mean) * rstd, lg0[k0 + e], lb0[k0 + e]);
                    xf[rb][s_] = make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7]));
                }
            }
        }
        // 32 output features (two 16-blocks) at a time: lane (q, c) ends with features 32 g + 8 q + {0..7} of row c
        const int ngrp = N / 32;
        float rsum[2] = {0.f, 0.f};
        auto group = [&](const int g, float (&kp)[2][8]) {     // kp: where POST keeps this group's values (static index at the call)
            f32x4_t d[2][2];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const float4 bias = *reinterpret_cast<const float4*>(lb + g * 32 + 8 * q + 4 * hb);
                d[0][hb] = f32x4_t{bias.x, bias.y, bias.z, bias.w};
                d[1][hb] = d[0][hb];
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    const uint4 a = *reinterpret_cast<const uint4*>(lw + (size_t)((2 * g + hb) * 16 + c) * W_ROWB + (s_ * 32 + q * 8) * 2);
                    d[0][hb] = mfma(a, xf[0][s_], d[0][hb]);
                    d[1][hb] = mfma(a, xf[1][s_], d[1][hb]);
                }
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float o[8] = {d[rb][0][0], d[rb][0][1], d[rb][0][2], d[rb][0][3], d[rb][1][0], d[rb][1][1], d[rb][1][2], d[rb][1][3]};
                const int f0 = g * 32 + 8 * q;
                if (p.resid) {
                    const uint4 rv = *reinterpret_cast<const uint4*>(p.resid + (size_t)rowc[rb] * p.ldr + f0);
                    const uint32_t w_[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[2 * e] += __uint_as_float(w_[e] << 16);
                        o[2 * e + 1] += __uint_as_float(w_[e] & 0xFFFF0000u);
                    }
                }
                if (POST) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { kp[rb][e] = o[e]; rsum[rb] += o[e]; }
                } else {
                    const int r = row0 + rb * 16 + c;
                    if (r < M)
                        *reinterpret_cast<uint4*>(p.out + (size_t)r * p.ldo + f0) =
                            make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
                }
            }
        };
        if (POST) {
            float k0[2][8], k1[2][8], k2[2][8], k3[2][8];     // named (statically indexed) so that they stay in registers
            group(0, k0);
            if (ngrp > 1) group(1, k1);
            if (ngrp > 2) group(2, k2);
            if (ngrp > 3) group(3, k3);
            float mean[2], rstd[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                mean[rb] = quad_rows_sum(rsum[rb]) * inv_n;
                float sq = 0.f;
                auto add_sq = [&](const float (&k)[2][8]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float dd = k[rb][e] - mean[rb]; sq = fmaf(dd, dd, sq); }
                };
                add_sq(k0);
                if (ngrp > 1) add_sq(k1);
                if (ngrp > 2) add_sq(k2);
                if (ngrp > 3) add_sq(k3);
                rstd[rb] = rsqrtf(quad_rows_sum(sq) * inv_n + p.eps);
            }
            auto emit = [&](const int g, const float (&k)[2][8]) {
                const int f0 = g * 32 + 8 * q;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = fmaf((k[rb][e] - mean[rb]) * rstd[rb], lg1[f0 + e], lb1[f0 + e]);
                    const int r = row0 + rb * 16 + c;
                    if (r < M)
                        *reinterpret_cast<uint4*>(p.out + (size_t)r * p.ldo + f0) =
                            make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7]));
                }
            };
            emit(0, k0);
            if (ngrp > 1) emit(1, k1);
            if (ngrp > 2) emit(2, k2);
            if (ngrp > 3) emit(3, k3);
        } else {
            float unused[2][8];
            for (int g = 0; g < ngrp; ++g) group(g, unused);
        }
    }
}

size_t lin_lds_bytes(int h, int N) { return (size_t)N * (h * 2 + 16) + (size_t)(3 * N + 2 * h) * sizeof(float); }

template <int H, bool POST>
int launch_lin_p(const LinArgs& a, hipStream_t s) {
    static size_t attr_bytes = 0;
    const size_t lds = lin_lds_bytes(a.h, a.N);
    if (lds > attr_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(side_linear_kernel<H, POST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(side_linear): %s", hipGetErrorString(e));
        attr_bytes = lds;
    }
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int chunks = ceil_div(a.M, ROWS_PER_WAVE), want = ceil_div(chunks, NT / 64);
    const int grid = want < 256 * per_cu ? want : 256 * per_cu;
    hipLaunchKernelGGL((side_linear_kernel<H, POST>), dim3(grid), dim3(NT), lds, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

template <int H>
int launch_lin(const LinArgs& a, hipStream_t s) {
    return a.g1 ? launch_lin_p<H, true>(a, s) : launch_lin_p<H, false>(a, s);
}

// ---- the ladder's map: a wide stream into a narrow one -------------------------------------------------------------------
//     out[M, N] = resid + gelu(W[N, K] . x + b)        N <= 128 features out of K = 768 (reference models/ltt_vit.py:431
// `side = side + gelu(map(hidden))`, models/ltt_bert.py:492).  HBM-bound: the 768-wide backbone stream is read once; W (147 KB for
// 96 x 768) fills the LDS of a CU for the whole launch; a wave walks K in 32-deep steps with its 32 rows' fragments loaded straight
// from global memory (8 loads in flight), 6 weight fragments from LDS and 12 MFMAs per step.
struct MapArgs {
    const bf16_t* x; long ldx;
    const bf16_t* w; const float* b;
    const bf16_t* resid; long ldr;
    bf16_t* out; long ldo;
    int M, N, K, gelu;
    const int* dyn;
};

template <int NOUT>
__global__ __launch_bounds__(NT, 2) void side_map_kernel(MapArgs p) {
    constexpr int OB = NOUT / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, W_ROWB = K * 2 + 16;
    char* const lw = smem;                                                  // [NOUT][W_ROWB], rows in fragment order
    float* const lb = reinterpret_cast<float*>(lw + (size_t)NOUT * W_ROWB);   // [NOUT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int M = ag_dyn_clamp(p.M, p.dyn);
    for (int i = tid; i < NOUT * (K / 8); i += NT) {
        const int dr = i / (K / 8), pc = i - dr * (K / 8);
        const int ob = dr >> 4, fi = dr & 15;
        const int feat = 32 * (ob >> 1) + 8 * (fi >> 2) + 4 * (ob & 1) + (fi & 3);      // lane ends with 8 consecutive features
        *reinterpret_cast<uint4*>(lw + (size_t)dr * W_ROWB + pc * 16) = *reinterpret_cast<const uint4*>(p.w + (size_t)feat * K + pc * 8);
    }
    for (int i = tid; i < NOUT; i += NT) lb[i] = p.b ? p.b[i] : 0.f;
    __syncthreads();
    const int nchunks = (M + ROWS_PER_WAVE - 1) / ROWS_PER_WAVE;
    const int nks = K / 32;
    for (int chunk = blockIdx.x * (NT / 64) + wave; chunk < nchunks; chunk += gridDim.x * (NT / 64)) {
        const int row0 = chunk * ROWS_PER_WAVE;
        const bf16_t* xr[2];
        int rowc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int r = row0 + rb * 16 + c;
            rowc[rb] = r < M ? r : M - 1;
            xr[rb] = p.x + (size_t)rowc[rb] * p.ldx + q * 8;
        }
        f32x4_t acc[2][OB];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
            const float4 bias = *reinterpret_cast<const float4*>(lb + 32 * (ob >> 1) + 8 * q + 4 * (ob & 1));
            acc[0][ob] = f32x4_t{bias.x, bias.y, bias.z, bias.w};
            acc[1][ob] = acc[0][ob];
        }
        // K loop, PF steps of activations (2 x PF KiB per wave) in flight: the stream is HBM-bound
        constexpr int PF = 8;
        uint4 xb[PF][2];
#pragma unroll
        for (int j = 0; j < PF; ++j)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) xb[j][rb] = *reinterpret_cast<const uint4*>(xr[rb] + (j < nks ? j : 0) * 32);
        for (int s0 = 0; s0 < nks; s0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int s_ = s0 + j;
                if (s_ < nks) {
                    const uint4 x0 = xb[j][0], x1 = xb[j][1];
                    const int nx = s_ + PF;
                    if (nx < nks) {
                        xb[j][0] = *reinterpret_cast<const uint4*>(xr[0] + nx * 32);
                        xb[j][1] = *reinterpret_cast<const uint4*>(xr[1] + nx * 32);
                    }
#pragma unroll
                    for (int ob = 0; ob < OB; ++ob) {
                        const uint4 a = *reinterpret_cast<const uint4*>(lw + (size_t)(ob * 16 + c) * W_ROWB + (s_ * 32 + q * 8) * 2);
                        acc[0][ob] = mfma(a, x0, acc[0][ob]);
                        acc[1][ob] = mfma(a, x1, acc[1][ob]);
                    }
                }
            }
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int r = row0 + rb * 16 + c;
#pragma unroll
            for (int g = 0; g < OB / 2; ++g) {
                float o[8] = {acc[rb][2 * g][0], acc[rb][2 * g][1], acc[rb][2 * g][2], acc[rb][2 * g][3],
                              acc[rb][2 * g + 1][0], acc[rb][2 * g + 1][1], acc[rb][2 * g + 1][2], acc[rb][2 * g + 1][3]};
                if (p.gelu) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) { const f32x2_t gg = fast_gelu2(f32x2_t{o[e], o[e + 1]}); o[e] = gg.x; o[e + 1] = gg.y; }
                }
                const int f0 = g * 32 + 8 * q;
                if (p.resid) {
                    const uint4 rv = *reinterpret_cast<const uint4*>(p.resid + (size_t)rowc[rb] * p.ldr + f0);
                    const uint32_t w_[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[2 * e] += __uint_as_float(w_[e] << 16);
                        o[2 * e + 1] += __uint_as_float(w_[e] & 0xFFFF0000u);
                    }
                }
                if (r < M)
                    *reinterpret_cast<uint4*>(p.out + (size_t)r * p.ldo + f0) =
                        make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
            }
        }
    }
}

size_t map_lds_bytes(int N, int K) { return (size_t)N * (K * 2 + 16) + (size_t)N * sizeof(float); }

template <int NOUT>
int launch_map(const MapArgs& a, hipStream_t s) {
    static size_t attr_bytes = 0;
    const size_t lds = map_lds_bytes(a.N, a.K);
    if (lds > attr_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(side_map_kernel<NOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(side_map): %s", hipGetErrorString(e));
        attr_bytes = lds;
    }
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int chunks = ceil_div(a.M, ROWS_PER_WAVE), want = ceil_div(chunks, NT / 64);
    const int grid = want < 256 * per_cu ? want : 256 * per_cu;
    hipLaunchKernelGGL(side_map_kernel<NOUT>, dim3(grid), dim3(NT), lds, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

size_t side_lds_bytes(int h, int I) {
    return (size_t)I * (h * 2 + 16) + (size_t)h * (I * 2 + 16) + (size_t)(I + 3 * h) * sizeof(float);
}

template <int H>
int launch_side(const SideArgs& a, hipStream_t s) {
    static size_t attr_bytes = 0;
    const size_t lds = side_lds_bytes(a.h, a.I);
    if (lds > attr_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(side_mlp_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(side_mlp): %s", hipGetErrorString(e));
        attr_bytes = lds;
    }
    const int tiles = ceil_div(a.M, TILE_ROWS);
    const int grid = tiles < 256 ? tiles : 256;      // one workgroup per CU (the weights fill its LDS), walking tiles
    hipLaunchKernelGGL(side_mlp_kernel<H>, dim3(grid), dim3(NT), lds, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

}  // namespace

extern "C" int ag_side_mlp_supported(int h, int I, int dtype) {
    return (dtype == AG_BF16 && (h == 32 || h == 64 || h == 96 || h == 128) && I >= 32 && I % 32 == 0 &&
            side_lds_bytes(h, I) <= 160 * 1024) ? 1 : 0;
}

extern "C" int ag_side_mlp(const void* d_x, int64_t ldx, int M, int h, int I, const void* d_w1, const float* d_b1, const void* d_w2,
                           const float* d_b2, const float* d_ln_g, const float* d_ln_b, float ln_eps, int post_ln, void* d_out,
                           int64_t ldo, const int* d_rows, void* stream) {
    if (M == 0) return AG_OK;
    AG_REQUIRE(d_x && d_w1 && d_w2 && d_out && M > 0, "ag_side_mlp: null pointer");
    AG_REQUIRE(ag_side_mlp_supported(h, I, AG_BF16), "ag_side_mlp: h=%d I=%d unsupported (bf16, h in {32,64,96,128}, I %% 32 == 0, "
               "both weight matrices must fit the 160 KiB LDS)", h, I);
    AG_REQUIRE(ldx % 8 == 0 && ldo % 4 == 0, "ag_side_mlp: rows must be 16-byte (x) / 8-byte (out) aligned");
    AG_REQUIRE(!post_ln || (d_ln_g && d_ln_b), "ag_side_mlp: post-LN needs gamma and beta");
    SideArgs a;
    a.x = (const bf16_t*)d_x; a.ldx = ldx; a.out = (bf16_t*)d_out; a.ldo = ldo;
    a.w1 = (const bf16_t*)d_w1; a.b1 = d_b1; a.w2 = (const bf16_t*)d_w2; a.b2 = d_b2;
    a.g = d_ln_g; a.be = d_ln_b; a.eps = ln_eps; a.M = M; a.h = h; a.I = I; a.post_ln = post_ln;
    a.dyn = d_rows;
    hipStream_t s = (hipStream_t)stream;
    AgProfScope prof(AG_EPI_BIAS_GELU, 4.0 * M * (double)h * I, (double)M * h * 2.0 * 2.0 + 4.0 * h * I, s, d_rows, (double)M);
    switch (h) {
        case 32: return launch_side<32>(a, s);
        case 64: return launch_side<64>(a, s);
        case 96: return launch_side<96>(a, s);
        default: return launch_side<128>(a, s);
    }
}

extern "C" int ag_side_linear_supported(int h, int N, int post_ln, int dtype) {
    return (dtype == AG_BF16 && (h == 32 || h == 64 || h == 96 || h == 128) && N >= 32 && N % 32 == 0 && N <= 384 &&
            (!post_ln || N <= 128) && lin_lds_bytes(h, N) <= 160 * 1024) ? 1 : 0;
}

extern "C" int ag_side_linear(const void* d_x, int64_t ldx, int M, int h, int N, const void* d_w, const float* d_b,
                              const float* d_pre_g, const float* d_pre_b, const void* d_resid, int64_t ldr,
                              const float* d_post_g, const float* d_post_b, float ln_eps, void* d_out, int64_t ldo, const int* d_rows,
                              void* stream) {
    if (M == 0) return AG_OK;
    AG_REQUIRE(d_x && d_w && d_out && M > 0, "ag_side_linear: null pointer");
    AG_REQUIRE(ag_side_linear_supported(h, N, d_post_g != nullptr, AG_BF16), "ag_side_linear: h=%d N=%d unsupported", h, N);
    AG_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && (!d_resid || ldr % 8 == 0), "ag_side_linear: rows must be 16-byte aligned");
    AG_REQUIRE((d_pre_g == nullptr) == (d_pre_b == nullptr) && (d_post_g == nullptr) == (d_post_b == nullptr), "ag_side_linear: gamma without beta");
    LinArgs a;
    a.x = (const bf16_t*)d_x; a.ldx = ldx; a.w = (const bf16_t*)d_w; a.b = d_b; a.resid = (const bf16_t*)d_resid; a.ldr = ldr;
    a.out = (bf16_t*)d_out; a.ldo = ldo; a.g0 = d_pre_g; a.b0 = d_pre_b; a.g1 = d_post_g; a.b1 = d_post_b; a.eps = ln_eps;
    a.M = M; a.h = h; a.N = N; a.dyn = d_rows;
    hipStream_t s = (hipStream_t)stream;
    AgProfScope prof(d_resid ? AG_EPI_BIAS_RESID : AG_EPI_BIAS, 2.0 * M * (double)h * N,
                     (double)M * (h + N + (d_resid ? N : 0)) * 2.0 + 2.0 * h * N, s, d_rows, (double)M);
    switch (h) {
        case 32: return launch_lin<32>(a, s);
        case 64: return launch_lin<64>(a, s);
        case 96: return launch_lin<96>(a, s);
        default: return launch_lin<128>(a, s);
    }
}

// wide -> narrow Linear with (GELU and) an additive residual, weights resident in LDS: selected by ag_gemm for the ladder's map
bool ag_side_map_eligible(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, int epilogue, bool has_resid) {
    return M >= 2048 && (N == 32 || N == 64 || N == 96 || N == 128) && K % 32 == 0 && K >= 128 && lda % 8 == 0 && ldc % 8 == 0 &&
           (!has_resid || ldr % 8 == 0) && (epilogue == AG_EPI_BIAS_GELU || epilogue == AG_EPI_BIAS_GELU_ADD) &&
           map_lds_bytes(N, K) <= 160 * 1024;
}

int ag_side_map(const void* d_x, int64_t ldx, const void* d_w, const float* d_b, const void* d_resid, int64_t ldr, void* d_out,
                int64_t ldo, int M, int N, int K, int gelu, const int* d_rows, hipStream_t s) {
    MapArgs a;
    a.x = (const bf16_t*)d_x; a.ldx = ldx; a.w = (const bf16_t*)d_w; a.b = d_b; a.resid = (const bf16_t*)d_resid; a.ldr = ldr;
    a.out = (bf16_t*)d_out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.gelu = gelu; a.dyn = d_rows;
    switch (N) {
        case 32: return launch_map<32>(a, s);
        case 64: return launch_map<64>(a, s);
        case 96: return launch_map<96>(a, s);
        default: return launch_map<128>(a, s);
    }
}
