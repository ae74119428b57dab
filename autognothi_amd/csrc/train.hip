// train.hip — backward-pass building blocks for explainer / surrogate training (fp32).
//
// The training consumers run on B inputs (not B*K rows): reference scripts/train_explainer.py:184-198
// (explainer forward+backward once per batch) and scripts/train_surrogate.py:145-147.  The Linear
// backward GEMMs reuse ag_gemm (dX = dY·W as an NT GEMM against the transposed weight, dW = dYᵀ·X
// against transposed activations); this file adds what autograd needs around them: transposes, bias
// (column) sums, GELU / LayerNorm / soft-max / tanh backward, dropout, and the masked-attention
// backward (two recompute passes, no atomics, deterministic).
#include "common.h"

int ag_set_salt_train(uint32_t salt, hipStream_t s);

namespace {
AG_DEFINE_DROPOUT_SALT(set_salt_here)

// ---- 2-D transpose: dst[c][r] = src[r][c]; dst row stride ldd >= R (caller zero-fills padding) ----
__global__ void transpose_kernel(const float* __restrict__ src, int R, int Cc, int64_t lds_, float* __restrict__ dst, int64_t ldd) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = by + i, c = bx + tx;
        tile[i][tx] = (r < R && c < Cc) ? src[(int64_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = bx + i, r = by + tx;
        if (c < Cc && r < R) dst[(int64_t)c * ldd + r] = tile[tx][i];
    }
}

// the same with a bf16 destination (mixed-precision dW operands: transpose and cast in one pass); the WHOLE padded
// destination [Cc, ldd] is written (zeros beyond R), so the caller needs no memset
// `plain` (optional): the un-transposed bf16 copy [R, Cc] from the same pass — a mixed-precision Linear needs x and x^T (dy and
// dy^T, W and W^T) as bf16 operands: one launch instead of a cast plus a transpose
__global__ void transpose_bf16_kernel(const float* __restrict__ src, int R, int Cc, int64_t lds_, bf16_t* __restrict__ dst, int64_t ldd,
                                      bf16_t* __restrict__ plain) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = by + i, c = bx + tx;
        const bool in = r < R && c < Cc;
        const float v = in ? src[(int64_t)r * lds_ + c] : 0.f;
        tile[i][tx] = v;
        if (plain && in) plain[(int64_t)r * Cc + c] = (bf16_t)(pack_bf16x2(v, 0.f) & 0xFFFFu);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = bx + i, r = by + tx;
        if (c < Cc && r < ldd) dst[(int64_t)c * ldd + r] = (bf16_t)(pack_bf16x2(tile[tx][i], 0.f) & 0xFFFFu);
    }
}

// The same pass over 64 x 64 tiles with 16-byte loads and 8-byte stores, and optionally an activation on the way (ACT):
//   0  v = src                               (operand forms of x, dY, W)
//   1  v = gelu(src)                         (fc2's operand: the fp32 GELU output is never written; backward recomputes from src)
//   2  v = src * gelu'(aux)                  (fc1's incoming gradient: f32_out (optional) receives it in fp32 for the bias sums)
// Needs cols % 4 == 0, 16-byte aligned rows, ldd % 4 == 0 (every Linear width here); the 32 x 32 kernel above covers the rest.
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
template <int ACT>
__global__ __launch_bounds__(256) void cast_transpose64_kernel(const float* __restrict__ src, const float* __restrict__ aux, int R, int Cc,
                                                               int64_t lds_, bf16_t* __restrict__ dst, int64_t ldd,
                                                               bf16_t* __restrict__ plain, float* __restrict__ f32_out) {
    __shared__ float tile[64][65];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64, t = threadIdx.x;
    const int lc = (t & 15) * 4, lr = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = lr + i * 16, r = by + rr, c = bx + lc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < Cc) {
            v = *reinterpret_cast<const float4*>(src + (int64_t)r * lds_ + c);
            if (ACT == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
            if (ACT == 2) {
                const float4 u = *reinterpret_cast<const float4*>(aux + (int64_t)r * lds_ + c);
                v.x *= gelu_grad(u.x); v.y *= gelu_grad(u.y); v.z *= gelu_grad(u.z); v.w *= gelu_grad(u.w);
                if (f32_out) *reinterpret_cast<float4*>(f32_out + (int64_t)r * Cc + c) = v;
            }
            if (plain) *reinterpret_cast<uint2*>(plain + (int64_t)r * Cc + c) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        }
        tile[rr][lc] = v.x; tile[rr][lc + 1] = v.y; tile[rr][lc + 2] = v.z; tile[rr][lc + 3] = v.w;
    }
    __syncthreads();
    const int sr = (t & 15) * 4, sc = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = sc + i * 16, c = bx + cc, r = by + sr;
        if (c < Cc && r < ldd)    // rows beyond R were staged as zeros: the whole padded destination is written
            *reinterpret_cast<uint2*>(dst + (int64_t)c * ldd + r) = make_uint2(pack_bf16x2(tile[sr][cc], tile[sr + 1][cc]),
                                                                              pack_bf16x2(tile[sr + 2][cc], tile[sr + 3][cc]));
    }
}

// ---- column sums: out[n] (+)= sum_m x[m][n] (bias gradients) ----
// One launch, no atomics, nothing to zero: a block owns CPB columns for ALL rows (256 threads = CPB/4 float4 lanes x RG row
// groups), every row group walks its rows in order and the RG partials are added in a fixed tree through LDS, so the sums are
// bit-reproducible (the previous version combined row slabs with float atomics behind a hipMemsetAsync).
template <int CPB>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int M, int N, int64_t ldx, float* __restrict__ out, int accumulate,
                                                     int vec_ok) {
    constexpr int VL = CPB / 4, RG = 256 / VL;
    __shared__ float4 part[RG][VL];
    const int vl = threadIdx.x % VL, rg = threadIdx.x / VL;
    const int c = blockIdx.x * CPB + vl * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_ok && c + 3 < N) {
        int m = rg;
        for (; m + 3 * RG < M; m += 4 * RG) {          // four independent loads in flight per thread
            const float4 a0 = *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + c);
            const float4 a1 = *reinterpret_cast<const float4*>(x + (int64_t)(m + RG) * ldx + c);
            const float4 a2 = *reinterpret_cast<const float4*>(x + (int64_t)(m + 2 * RG) * ldx + c);
            const float4 a3 = *reinterpret_cast<const float4*>(x + (int64_t)(m + 3 * RG) * ldx + c);
            s.x += (a0.x + a1.x) + (a2.x + a3.x); s.y += (a0.y + a1.y) + (a2.y + a3.y);
            s.z += (a0.z + a1.z) + (a2.z + a3.z); s.w += (a0.w + a1.w) + (a2.w + a3.w);
        }
        for (; m < M; m += RG) {
            const float4 a0 = *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + c);
            s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
        }
    } else if (c < N) {                                 // ragged last columns / rows not 16-byte aligned: scalar
        for (int m = rg; m < M; m += RG) {
            const float* r = x + (int64_t)m * ldx + c;
            s.x += r[0];
            if (c + 1 < N) s.y += r[1];
            if (c + 2 < N) s.z += r[2];
            if (c + 3 < N) s.w += r[3];
        }
    }
    part[rg][vl] = s;
    __syncthreads();
#pragma unroll
    for (int h = RG / 2; h >= 1; h >>= 1) {
        if (rg < h) {
            const float4 o = part[rg + h][vl];
            float4 t = part[rg][vl];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            part[rg][vl] = t;
        }
        __syncthreads();
    }
    if (rg == 0 && c < N) {
        const float4 t = part[0][vl];
        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c + j < N) out[c + j] = accumulate ? out[c + j] + v[j] : v[j];
    }
}

__global__ void gelu_fwd_kernel(const float* __restrict__ u, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = gelu_erf(u[i]);
}
// d/du [0.5 u (1 + erf(u/sqrt2))] = 0.5 (1 + erf(u/sqrt2)) + u * exp(-u^2/2) / sqrt(2 pi)
__global__ void gelu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dy, float* __restrict__ du, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = u[i];
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        du[i] = dy[i] * (cdf + x * 0.3989422804014327f * expf(-0.5f * x * x));
    }
}
__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dx[i] = dy[i] * (1.0f - y[i] * y[i]);
}
// y = a + b (elementwise), in place allowed
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}

// inverted dropout (torch.nn.Dropout semantics): y = keep ? x/(1-p) : 0 ; applying it to dy gives dx
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p, uint32_t seed) {
    seed = ag_salted(seed);
    const float sc = 1.0f / (1.0f - p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = keep_elem(seed, (uint64_t)i, p) ? x[i] * sc : 0.f;
}
// y = resid + dropout(x): the transformer block's "hidden = residual + dropout(dense(...))" in one pass (p = 0: a plain add)
__global__ void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ resid, float* __restrict__ y, int64_t n, float p,
                                   uint32_t seed) {
    seed = ag_salted(seed);
    const float sc = 1.0f / (1.0f - p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = resid[i] + ((p == 0.f || keep_elem(seed, (uint64_t)i, p)) ? x[i] * sc : 0.f);
}

// soft-max backward on rows: dx = y * (dy - sum(y*dy))
__global__ void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int rows, int Cc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < Cc; c += 64) s += y[(int64_t)row * Cc + c] * dy[(int64_t)row * Cc + c];
    s = wave_sum(s);
    for (int c = lane; c < Cc; c += 64) dx[(int64_t)row * Cc + c] = y[(int64_t)row * Cc + c] * (dy[(int64_t)row * Cc + c] - s);
}

// LayerNorm backward.  One wave per row, the row held in registers (NC = H/64 columns per lane, one read of x and dy);
// dgamma/dbeta partials are carried in registers across the wave's rows and folded once per wave through LDS into
// part[block][2][H].  NC = 0: any H (<= 4096), columns re-read from L2 and LDS atomics per row.
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ dy,
                                                     int rows, int H, float eps, float* __restrict__ dx, float* __restrict__ part,
                                                     const float* __restrict__ addend) {
    extern __shared__ float sacc[];  // [2][H] per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) sacc[i] = 0.f;
    __syncthreads();
    if (NC > 0) {
        float gv[NC > 0 ? NC : 1], ag[NC > 0 ? NC : 1], ab[NC > 0 ? NC : 1];
#pragma unroll
        for (int i = 0; i < NC; ++i) { gv[i] = g ? g[lane + 64 * i] : 1.f; ag[i] = 0.f; ab[i] = 0.f; }
        for (int row = blockIdx.x * nw + wave; row < rows; row += gridDim.x * nw) {
            const float* xr = x + (int64_t)row * H;
            const float* dr = dy + (int64_t)row * H;
            float xv[NC > 0 ? NC : 1], dv[NC > 0 ? NC : 1];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NC; ++i) { xv[i] = xr[lane + 64 * i]; dv[i] = dr[lane + 64 * i]; s += xv[i]; }
            const float mean = wave_sum(s) / (float)H;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NC; ++i) { xv[i] -= mean; sq += xv[i] * xv[i]; }
            const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
            float a = 0.f, b = 0.f;  // mean(dy*g), mean(dy*g*xhat)
#pragma unroll
            for (int i = 0; i < NC; ++i) { xv[i] *= rstd; const float dg = dv[i] * gv[i]; a += dg; b += dg * xv[i]; }
            a = wave_sum(a) / (float)H; b = wave_sum(b) / (float)H;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const float r_ = rstd * (dv[i] * gv[i] - a - xv[i] * b);   // (+ the gradient arriving over the residual branch)
                dx[(int64_t)row * H + lane + 64 * i] = addend ? r_ + addend[(int64_t)row * H + lane + 64 * i] : r_;
                ag[i] = fmaf(dv[i], xv[i], ag[i]);
                ab[i] += dv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {   // LDS atomics: waves of one block only, once per wave
            atomicAdd(&sacc[lane + 64 * i], ag[i]);
            atomicAdd(&sacc[H + lane + 64 * i], ab[i]);
        }
    } else {
        for (int row = blockIdx.x * nw + wave; row < rows; row += gridDim.x * nw) {
            const float* xr = x + (int64_t)row * H;
            const float* dr = dy + (int64_t)row * H;
            float s = 0.f;
            for (int c = lane; c < H; c += 64) s += xr[c];
            const float mean = wave_sum(s) / (float)H;
            float sq = 0.f;
            for (int c = lane; c < H; c += 64) { const float d = xr[c] - mean; sq += d * d; }
            const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
            float a = 0.f, b = 0.f;
            for (int c = lane; c < H; c += 64) {
                const float xh = (xr[c] - mean) * rstd, dg = dr[c] * (g ? g[c] : 1.f);
                a += dg; b += dg * xh;
            }
            a = wave_sum(a) / (float)H; b = wave_sum(b) / (float)H;
            for (int c = lane; c < H; c += 64) {
                const float xh = (xr[c] - mean) * rstd, dg = dr[c] * (g ? g[c] : 1.f);
                const float r_ = rstd * (dg - a - xh * b);
                dx[(int64_t)row * H + c] = addend ? r_ + addend[(int64_t)row * H + c] : r_;
                atomicAdd(&sacc[c], dr[c] * xh);
                atomicAdd(&sacc[H + c], dr[c]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) part[(int64_t)blockIdx.x * 2 * H + i] = sacc[i];
}
// block = 64 columns x 4 partial lanes: each lane walks a quarter of the per-block partials, LDS folds the four
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int H, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int accumulate) {
    __shared__ float sa[4][64], sb[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float a = 0.f, b = 0.f;
    if (c < H)
        for (int i = q; i < nblocks; i += 4) { a += part[(int64_t)i * 2 * H + c]; b += part[(int64_t)i * 2 * H + H + c]; }
    sa[q][threadIdx.x & 63] = a; sb[q][threadIdx.x & 63] = b;
    __syncthreads();
    if (q == 0 && c < H) {
        a = (sa[0][threadIdx.x] + sa[1][threadIdx.x]) + (sa[2][threadIdx.x] + sa[3][threadIdx.x]);
        b = (sb[0][threadIdx.x] + sb[1][threadIdx.x]) + (sb[2][threadIdx.x] + sb[3][threadIdx.x]);
        dgamma[c] = accumulate ? dgamma[c] + a : a;
        dbeta[c] = accumulate ? dbeta[c] + b : b;
    }
}

// ---- masked attention backward (fp32), head_dim 64, T <= 1024 ---------------------------------------
// qkv [R,T,3H] (row r reads source row r: no sharing in training), dctx [R,T,H], ctx [R,T,H].
// Pass A (thread per query i): m_i, l_i, D_i = dO_i.O_i, and dQ_i.   Pass B (thread per key j): dK_j, dV_j.
// s_ij = q_i.k_j/8, ViT: s*=mask_j, BERT: masked j excluded.  p-dropout: kept entries scaled by 1/(1-p).
struct AttnBwdArgs {
    const float* qkv; const uint32_t* mask; const float* ctx; const float* dctx;
    float* dqkv; float* stats;  // stats [R*heads*T][3] = (m, l, D)
    int R, T, H, heads, mode, Tw; float pdrop; uint32_t seed;
    // bf16 I/O (ag_masked_attention_*_bf16): qkv / outputs as bf16, dctx = sum of `dslabs` fp32 slabs (a split-K GEMM's partials)
    const bf16_t* qkv16; bf16_t* out16; int dslabs; long dslab_stride;
};
template <int AHD>
__device__ __forceinline__ float dot_hd(const float* a, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < AHD; d += 4) {
        const float4 x = *reinterpret_cast<const float4*>(a + d), y = *reinterpret_cast<const float4*>(b + d);
        s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
    }
    return s;
}
template <int AHD>
__global__ __launch_bounds__(256) void attn_bwd_query_kernel(AttnBwdArgs p) {
    const float inv_sqrt_d = 1.0f / sqrtf((float)AHD);
    __shared__ __attribute__((aligned(16))) float sK[64 * AHD];
    __shared__ __attribute__((aligned(16))) float sV[64 * AHD];
    const int row = blockIdx.x / p.heads, head = blockIdx.x % p.heads, tid = threadIdx.x;
    const long ts = (long)3 * p.H;
    const float* base = p.qkv + (long)row * p.T * ts + (long)head * AHD;
    const uint32_t* mrow = p.mask + (long)row * p.Tw;
    const float keep_sc = 1.0f / (1.0f - p.pdrop);
    for (int q0 = 0; q0 < p.T; q0 += blockDim.x) {
        const int i = q0 + tid;
        const bool valid = i < p.T;
        float q[AHD], dO[AHD], dq[AHD];
        const float* qp = base + (long)(valid ? i : 0) * ts;
        const float* dop = p.dctx + ((long)row * p.T + (valid ? i : 0)) * p.H + (long)head * AHD;
        const float* op = p.ctx + ((long)row * p.T + (valid ? i : 0)) * p.H + (long)head * AHD;
        float D = 0.f;
#pragma unroll
        for (int d = 0; d < AHD; ++d) { q[d] = qp[d]; dO[d] = dop[d]; dq[d] = 0.f; D = fmaf(dO[d], op[d], D); }
        // pass 1: soft-max statistics
        float m = -3.0e38f, l = 0.f;
        for (int k0 = 0; k0 < p.T; k0 += 64) {
            __syncthreads();
            for (int c = tid; c < 64 * (AHD / 4); c += blockDim.x) {
                const int r = c / (AHD / 4), ch = c % (AHD / 4);
                float4 kv = make_float4(0, 0, 0, 0);
                if (k0 + r < p.T) kv = *reinterpret_cast<const float4*>(base + (long)(k0 + r) * ts + p.H + ch * 4);
                *reinterpret_cast<float4*>(sK + r * AHD + ch * 4) = kv;
            }
            __syncthreads();
            const int kn = min(64, p.T - k0);
            for (int kk = 0; kk < kn; ++kk) {
                const int j = k0 + kk;
                const bool on = (mrow[j >> 5] >> (j & 31)) & 1u;
                float s = dot_hd<AHD>(q, sK + kk * AHD) * inv_sqrt_d;
                if (p.mode == AG_MASK_VIT_MUL) s = on ? s : 0.f; else if (!on) continue;
                const float mn = fmaxf(m, s);
                l = l * expf(m - mn) + expf(s - mn);
                m = mn;
            }
        }
        // pass 2: dQ
        for (int k0 = 0; k0 < p.T; k0 += 64) {
            __syncthreads();
            for (int c = tid; c < 64 * (AHD / 4); c += blockDim.x) {
                const int r = c / (AHD / 4), ch = c % (AHD / 4);
                float4 kv = make_float4(0, 0, 0, 0), vv = kv;
                if (k0 + r < p.T) {
                    kv = *reinterpret_cast<const float4*>(base + (long)(k0 + r) * ts + p.H + ch * 4);
                    vv = *reinterpret_cast<const float4*>(base + (long)(k0 + r) * ts + 2 * p.H + ch * 4);
                }
                *reinterpret_cast<float4*>(sK + r * AHD + ch * 4) = kv;
                *reinterpret_cast<float4*>(sV + r * AHD + ch * 4) = vv;
            }
            __syncthreads();
            const int kn = min(64, p.T - k0);
            for (int kk = 0; kk < kn; ++kk) {
                const int j = k0 + kk;
                const bool on = (mrow[j >> 5] >> (j & 31)) & 1u;
                float s = dot_hd<AHD>(q, sK + kk * AHD) * inv_sqrt_d;
                if (p.mode == AG_MASK_VIT_MUL) s = on ? s : 0.f; else if (!on) continue;
                const float pr = expf(s - m) / l;
                float dP = dot_hd<AHD>(dO, sV + kk * AHD);
                if (p.pdrop > 0.f) dP = keep_elem(p.seed, ((uint64_t)(blockIdx.x) * p.T + i) * p.T + j, p.pdrop) ? dP * keep_sc : 0.f;
                float ds = pr * (dP - D) * inv_sqrt_d;
                if (p.mode == AG_MASK_VIT_MUL && !on) ds = 0.f;  // d(s*0)/ds = 0
#pragma unroll
                for (int d = 0; d < AHD; ++d) dq[d] = fmaf(ds, sK[kk * AHD + d], dq[d]);
            }
        }
        if (valid) {
            float* out = p.dqkv + ((long)row * p.T + i) * ts + (long)head * AHD;
#pragma unroll
            for (int d = 0; d < AHD; ++d) out[d] = dq[d];
            float* st = p.stats + ((long)blockIdx.x * p.T + i) * 3;
            st[0] = m; st[1] = l; st[2] = D;
        }
    }
}
template <int AHD>
__global__ __launch_bounds__(256) void attn_bwd_key_kernel(AttnBwdArgs p) {
    const float inv_sqrt_d = 1.0f / sqrtf((float)AHD);
    __shared__ __attribute__((aligned(16))) float sQ[64 * AHD];
    __shared__ __attribute__((aligned(16))) float sO[64 * AHD];
    __shared__ float sS[64 * 3];
    const int row = blockIdx.x / p.heads, head = blockIdx.x % p.heads, tid = threadIdx.x;
    const long ts = (long)3 * p.H;
    const float* base = p.qkv + (long)row * p.T * ts + (long)head * AHD;
    const uint32_t* mrow = p.mask + (long)row * p.Tw;
    const float keep_sc = 1.0f / (1.0f - p.pdrop);
    for (int j0 = 0; j0 < p.T; j0 += blockDim.x) {
        const int j = j0 + tid;
        const bool valid = j < p.T;
        const bool on = valid ? ((mrow[j >> 5] >> (j & 31)) & 1u) : false;
        float k[AHD], v[AHD], dk[AHD], dv[AHD];
        const float* kp = base + (long)(valid ? j : 0) * ts + p.H;
        const float* vp = base + (long)(valid ? j : 0) * ts + 2 * p.H;
#pragma unroll
        for (int d = 0; d < AHD; ++d) { k[d] = kp[d]; v[d] = vp[d]; dk[d] = 0.f; dv[d] = 0.f; }
        for (int i0 = 0; i0 < p.T; i0 += 64) {
            __syncthreads();
            for (int c = tid; c < 64 * (AHD / 4); c += blockDim.x) {
                const int r = c / (AHD / 4), ch = c % (AHD / 4);
                float4 qv = make_float4(0, 0, 0, 0), ov = qv;
                if (i0 + r < p.T) {
                    qv = *reinterpret_cast<const float4*>(base + (long)(i0 + r) * ts + ch * 4);
                    ov = *reinterpret_cast<const float4*>(p.dctx + ((long)row * p.T + i0 + r) * p.H + (long)head * AHD + ch * 4);
                }
                *reinterpret_cast<float4*>(sQ + r * AHD + ch * 4) = qv;
                *reinterpret_cast<float4*>(sO + r * AHD + ch * 4) = ov;
            }
            for (int c = tid; c < 64 * 3; c += blockDim.x) {
                const int r = c / 3;
                sS[c] = (i0 + r < p.T) ? p.stats[((long)blockIdx.x * p.T + i0 + r) * 3 + c % 3] : 0.f;
            }
            __syncthreads();
            const int in = min(64, p.T - i0);
            if (valid && (p.mode == AG_MASK_VIT_MUL || on)) {
                for (int ii = 0; ii < in; ++ii) {
                    const int i = i0 + ii;
                    float s = dot_hd<AHD>(k, sQ + ii * AHD) * inv_sqrt_d;
                    if (p.mode == AG_MASK_VIT_MUL) s = on ? s : 0.f;
                    const float pr = expf(s - sS[ii * 3]) / sS[ii * 3 + 1];
                    float dP = dot_hd<AHD>(v, sO + ii * AHD);
                    float pk = pr;  // weight that multiplied V_j in the forward (after dropout)
                    if (p.pdrop > 0.f) {
                        const bool kp_ = keep_elem(p.seed, ((uint64_t)(blockIdx.x) * p.T + i) * p.T + j, p.pdrop);
                        dP = kp_ ? dP * keep_sc : 0.f;
                        pk = kp_ ? pr * keep_sc : 0.f;
                    }
                    float ds = pr * (dP - sS[ii * 3 + 2]) * inv_sqrt_d;
                    if (p.mode == AG_MASK_VIT_MUL && !on) ds = 0.f;
#pragma unroll
                    for (int d = 0; d < AHD; ++d) { dv[d] = fmaf(pk, sO[ii * AHD + d], dv[d]); dk[d] = fmaf(ds, sQ[ii * AHD + d], dk[d]); }
                }
            }
        }
        if (valid) {
            float* ok = p.dqkv + ((long)row * p.T + j) * ts + p.H + (long)head * AHD;
            float* ov = p.dqkv + ((long)row * p.T + j) * ts + 2 * p.H + (long)head * AHD;
#pragma unroll
            for (int d = 0; d < AHD; ++d) { ok[d] = dk[d]; ov[d] = dv[d]; }
        }
    }
}

int grid_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096); }


// ---- masked attention backward on the matrix cores (mixed-precision training: bf16 operands, fp32 accumulate / I/O) ----
// One workgroup per (row, head), head dim 64, T <= 256.  Q, K, V, dO head slices are converted to bf16 once into four
// swizzled LDS images (the forward's layout: 128-byte rows, 16-byte chunk c of row r at slot c ^ swz(r), so rows can be
// read as MFMA row fragments and, through ds_read_b64_tr_b16, as transposed fragments).  ViT: the K rows of masked keys are
// zero (logit exactly 0, as forward).  D_i = dO_i . O_i per query.
//   pass 1 (a lane owns a QUERY, wave = query block): S^T = K.Q^T per key block -> row max / sum -> LSE_i (base 2);
//   pass 2 (same layout): S^T and dP^T = V.dO^T again, P = exp2(c S - LSE), dS = P (keep/(1-p) dP - D) (ViT: x mask_j);
//           the dS^T accumulator tile is the B operand of dQ^T += K^T.dS^T (K^T by transposed reads) -> dQ, no reduction
//           across waves;
//   pass 3 (a lane owns a KEY, wave = key block): S = Q.K^T and dP = dO.V^T per query block; the P~ and dS tiles are the B
//           operands of dV^T += dO^T.P~ and dK^T += Q^T.dS (dO^T, Q^T by transposed reads) -> dK, dV, no reduction across
//           waves and nothing but LSE / D (2 KiB) crosses LDS.
// Dropout: the forward's counter hash on (workgroup, query, key).  Recomputing S, dP in pass 3 instead of staging dS through
// LDS costs 8 MFMAs per tile pair and keeps the kernel free of cross-wave protocols.
constexpr int BROW = 128;
__device__ __forceinline__ int bswz(int row) {
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1);
}
__device__ __forceinline__ uint4 pack8_bf16(const float4 a, const float4 b) {
    return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}

// FWD: only pass 1 runs and its O' (the training forward on bf16 operands, dropout included) is stored to p.dqkv as ctx [R,T,H].
// IO16: bf16 qkv in, bf16 ctx / dqkv out, dctx summed from fp32 slabs (the bf16-activation training step); else fp32 I/O.
template <int MODE, bool FWD, bool IO16 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_mfma_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    p.seed = ag_salted(p.seed);
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(4))) __bf16 b16x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int row = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
    const int T = p.T, Tp = (T + 31) & ~31, nb = Tp >> 5;
    const int img = Tp * BROW;
    char* iQ = smem; char* iK = smem + img; char* iV = smem + 2 * img; char* iO = smem + 3 * img;
    float* sD = reinterpret_cast<float*>(smem + 4 * img);   // [Tp] dO_i . O_i
    float* sL = sD + Tp;                                    // [Tp] LSE_i in base-2 units of the scaled scores
    const long ts = (long)3 * p.H;
    const float* base = p.qkv + (long)row * T * ts + (long)head * 64;
    const float* dob = p.dctx + (long)row * T * p.H + (long)head * 64;
    const uint32_t* mrow = p.mask + (long)row * p.Tw;
    const float c2 = 0.125f * 1.4426950408889634f;
    const float keep_sc = 1.0f / (1.0f - p.pdrop);
    const bool drop = p.pdrop > 0.f;
    const uint64_t hbase = (uint64_t)blockIdx.x * T;

    // ---- images
    const bf16_t* base16 = IO16 ? p.qkv16 + (long)row * T * ts + (long)head * 64 : nullptr;
    for (int c = tid; c < Tp * 8; c += blockDim.x) {
        const int t = c >> 3, ch = c & 7;
        const int tc = t < T ? t : T - 1;
        const int dst = t * BROW + ((ch ^ bswz(t)) << 4);
        uint4 kv;
        if (IO16) {
            const bf16_t* src = base16 + (long)tc * ts + ch * 8;
            *reinterpret_cast<uint4*>(iQ + dst) = *reinterpret_cast<const uint4*>(src);
            kv = *reinterpret_cast<const uint4*>(src + p.H);
            *reinterpret_cast<uint4*>(iV + dst) = *reinterpret_cast<const uint4*>(src + 2 * p.H);
        } else {
            const float* src = base + (long)tc * ts + ch * 8;
            *reinterpret_cast<uint4*>(iQ + dst) = pack8_bf16(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4));
            kv = pack8_bf16(*reinterpret_cast<const float4*>(src + p.H), *reinterpret_cast<const float4*>(src + p.H + 4));
            *reinterpret_cast<uint4*>(iV + dst) = pack8_bf16(*reinterpret_cast<const float4*>(src + 2 * p.H), *reinterpret_cast<const float4*>(src + 2 * p.H + 4));
        }
        if (MODE == AG_MASK_VIT_MUL && !((mrow[tc >> 5] >> (tc & 31)) & 1u)) kv = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(iK + dst) = kv;
        uint4 dv = make_uint4(0u, 0u, 0u, 0u);
        if (!FWD && t < T) {
            const float* ds_ = dob + (long)t * p.H + ch * 8;
            float4 d0 = *reinterpret_cast<const float4*>(ds_), d1 = *reinterpret_cast<const float4*>(ds_ + 4);
            if (IO16)
                for (int sl = 1; sl < p.dslabs; ++sl) {   // slab order: bit-reproducible
                    const float4 e0 = *reinterpret_cast<const float4*>(ds_ + sl * p.dslab_stride), e1 = *reinterpret_cast<const float4*>(ds_ + sl * p.dslab_stride + 4);
                    d0.x += e0.x; d0.y += e0.y; d0.z += e0.z; d0.w += e0.w; d1.x += e1.x; d1.y += e1.y; d1.z += e1.z; d1.w += e1.w;
                }
            dv = pack8_bf16(d0, d1);
        }
        *reinterpret_cast<uint4*>(iO + dst) = dv;
    }
    __syncthreads();

    const int lr = lane & 31, lh = lane >> 5;
    // row-fragment offsets (8 consecutive head-dim elements 16ks + 8lh .. of row lr of a 32-row block)
    int roff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) roff[ks] = lr * BROW + (((2 * ks + lh) ^ bswz(lr)) << 4);
    // transposed-fragment offsets (forward's V^T recipe): [d tile][first / second 4-row group]; k-step 1 is +2048 B
    int toff[2][2];
    {
        const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int k0 = 4 * (g >> 1) + tq, k1 = k0 + 8;
            const int chunk = dt * 4 + 2 * (g & 1) + (tp >> 1);
            toff[dt][0] = k0 * BROW + ((chunk ^ bswz(k0)) << 4) + 8 * (tp & 1);
            toff[dt][1] = k1 * BROW + ((chunk ^ bswz(k1)) << 4) + 8 * (tp & 1);
        }
    }
    auto row_frag = [&](const char* image, int blk, int ks) -> bf16x8_t {
        return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(image + blk * 32 * BROW + roff[ks]));
    };
    auto tr_frag = [&](const char* image, int blk, int st, int dt) -> bf16x8_t {
        const char* b = image + blk * 32 * BROW + 2048 * st;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + toff[dt][0]));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + toff[dt][1]));
        return __builtin_shufflevector(__builtin_bit_cast(b16x4, v0), __builtin_bit_cast(b16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto pack_step = [&](const f32x16_t& x, int st) -> bf16x8_t {
        return __builtin_bit_cast(bf16x8_t, make_uint4(pack_bf16x2(x[8 * st + 0], x[8 * st + 1]), pack_bf16x2(x[8 * st + 2], x[8 * st + 3]),
                                                       pack_bf16x2(x[8 * st + 4], x[8 * st + 5]), pack_bf16x2(x[8 * st + 6], x[8 * st + 7])));
    };
    auto zero16 = [](f32x16_t& x) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = 0.f;
    };

    // ================= passes 1 + 2: a lane owns a query =================
    for (int qb = wave; qb < nb; qb += nwaves) {
        const int q = qb * 32 + lr;
        bf16x8_t qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = row_frag(iQ, qb, ks); dof[ks] = row_frag(FWD ? iQ : iO, qb, ks); }
        // scores of one key block, keys in registers: reg i <-> key kb*32 + (i&3) + 8(i>>2) + 4lh; masked / padded -> -inf
        auto scores_t = [&](int kb) -> f32x16_t {
            f32x16_t s; zero16(s);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(iK, kb, ks), qf[ks], s, 0, 0, 0);
            uint32_t vis = 0xFFFFFFFFu;
            if (MODE == AG_MASK_BERT_ADD) vis = mrow[kb];
            const int kvalid = T - kb * 32;
            if (kvalid < 32) vis &= (1u << kvalid) - 1u;
            if (vis != 0xFFFFFFFFu) {   // wave-uniform
                const uint32_t vl = vis >> (4 * lh);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (!((vl >> ((i & 3) + 8 * (i >> 2))) & 1u)) s[i] = -3.0e38f;
            }
            return s;
        };
        // pass 1 = the forward on the SAME bf16 operands: LSE_i, and D_i = dO_i . O'_i with O' = sum_j P~_ij V_j from the
        // matrix cores, so that dP_ij - D_i cancels exactly where it should (an fp32 D against bf16 dP would not)
        float m = -3.0e38f, l = 0.f;
        f32x16_t o0, o1; zero16(o0); zero16(o1);
        for (int kb = 0; kb < nb; ++kb) {
            f32x16_t s = scores_t(kb);
            float bm = s[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) bm = fmaxf(bm, s[i]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(bm), __float_as_uint(bm), false, false);
                bm = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float mn = fmaxf(m, bm);
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * c2);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f((s[i] - mn) * c2);
                acc += pv;
                s[i] = pv;
                if (drop) s[i] = keep_elem(p.seed, (hbase + q) * T + (kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh), p.pdrop) ? pv : 0.f;   // the 1/(1-p) scale is applied in fp32 below: it would not survive the bf16 rounding of the fragment exactly
            }
            l = fmaf(l, alpha, acc);
            m = mn;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8_t pf = pack_step(s, st);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(iV, kb, st, 0), pf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(iV, kb, st, 1), pf, o1, 0, 0, 0);
            }
        }
        {   // the two lane halves hold disjoint key subsets of the same query (same reference max)
            const auto sl = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
            l = __uint_as_float(sl[0]) + __uint_as_float(sl[1]);
        }
        if (FWD) {   // ctx = O' / l with the dropout scale, fp32
            if (q < T) {
                const float sc = keep_sc / l;
                const long oo = ((long)row * T + q) * p.H + (long)head * 64;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = 8 * g4 + 4 * lh;
                    if (IO16) {
                        *reinterpret_cast<uint2*>(p.out16 + oo + d) = make_uint2(pack_bf16x2(o0[4 * g4] * sc, o0[4 * g4 + 1] * sc), pack_bf16x2(o0[4 * g4 + 2] * sc, o0[4 * g4 + 3] * sc));
                        *reinterpret_cast<uint2*>(p.out16 + oo + 32 + d) = make_uint2(pack_bf16x2(o1[4 * g4] * sc, o1[4 * g4 + 1] * sc), pack_bf16x2(o1[4 * g4 + 2] * sc, o1[4 * g4 + 3] * sc));
                    } else {
                        float* out = p.dqkv + oo;
                        *reinterpret_cast<float4*>(out + d) = make_float4(o0[4 * g4] * sc, o0[4 * g4 + 1] * sc, o0[4 * g4 + 2] * sc, o0[4 * g4 + 3] * sc);
                        *reinterpret_cast<float4*>(out + 32 + d) = make_float4(o1[4 * g4] * sc, o1[4 * g4 + 1] * sc, o1[4 * g4 + 2] * sc, o1[4 * g4 + 3] * sc);
                    }
                }
            }
            continue;
        }
        const float lse2 = fmaf(m, c2, __builtin_amdgcn_logf(l));
        float Dq = 0.f;
        {   // regs 4g..4g+3 of lane (query, lh) are head-dim elements 8g + 4lh .. +3 (second tile: +32); dO from its bf16 image
            const char* drow = iO + q * BROW + 8 * lh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint2 a = *reinterpret_cast<const uint2*>(drow + ((g4 ^ bswz(q)) << 4));
                const uint2 b = *reinterpret_cast<const uint2*>(drow + (((4 + g4) ^ bswz(q)) << 4));
                Dq = fmaf(__uint_as_float(a.x << 16), o0[4 * g4], Dq); Dq = fmaf(__uint_as_float(a.x & 0xFFFF0000u), o0[4 * g4 + 1], Dq);
                Dq = fmaf(__uint_as_float(a.y << 16), o0[4 * g4 + 2], Dq); Dq = fmaf(__uint_as_float(a.y & 0xFFFF0000u), o0[4 * g4 + 3], Dq);
                Dq = fmaf(__uint_as_float(b.x << 16), o1[4 * g4], Dq); Dq = fmaf(__uint_as_float(b.x & 0xFFFF0000u), o1[4 * g4 + 1], Dq);
                Dq = fmaf(__uint_as_float(b.y << 16), o1[4 * g4 + 2], Dq); Dq = fmaf(__uint_as_float(b.y & 0xFFFF0000u), o1[4 * g4 + 3], Dq);
            }
            const auto sd = __builtin_amdgcn_permlane32_swap(__float_as_uint(Dq), __float_as_uint(Dq), false, false);
            Dq = (__uint_as_float(sd[0]) + __uint_as_float(sd[1])) * keep_sc / l;
        }
        if (lh == 0) { sL[q] = lse2; sD[q] = Dq; }
        f32x16_t dq0, dq1; zero16(dq0); zero16(dq1);
        for (int kb = 0; kb < nb; ++kb) {
            const f32x16_t s = scores_t(kb);
            f32x16_t dp; zero16(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(iV, kb, ks), dof[ks], dp, 0, 0, 0);
            const uint32_t mvit = MODE == AG_MASK_VIT_MUL ? (mrow[kb] >> (4 * lh)) : 0xFFFFFFFFu;
            f32x16_t ds;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = (i & 3) + 8 * (i >> 2);
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[i], c2, -lse2));
                float dpv = dp[i];
                if (drop) dpv = keep_elem(p.seed, (hbase + q) * T + (kb * 32 + kk + 4 * lh), p.pdrop) ? dpv * keep_sc : 0.f;
                float v = pr * (dpv - Dq);
                if (MODE == AG_MASK_VIT_MUL && !((mvit >> kk) & 1u)) v = 0.f;   // d(s * 0) / ds = 0
                ds[i] = v;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8_t pf = pack_step(ds, st);
                dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(iK, kb, st, 0), pf, dq0, 0, 0, 0);
                dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(iK, kb, st, 1), pf, dq1, 0, 0, 0);
            }
        }
        if (q < T) {   // accumulator regs 4g..4g+3 of lane (query, lh) are head-dim elements 8g + 4lh .. +3 (second tile: +32)
            const long oo = ((long)row * T + q) * ts + (long)head * 64;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 8 * g4 + 4 * lh;
                if (IO16) {
                    *reinterpret_cast<uint2*>(p.out16 + oo + d) = make_uint2(pack_bf16x2(dq0[4 * g4] * 0.125f, dq0[4 * g4 + 1] * 0.125f), pack_bf16x2(dq0[4 * g4 + 2] * 0.125f, dq0[4 * g4 + 3] * 0.125f));
                    *reinterpret_cast<uint2*>(p.out16 + oo + 32 + d) = make_uint2(pack_bf16x2(dq1[4 * g4] * 0.125f, dq1[4 * g4 + 1] * 0.125f), pack_bf16x2(dq1[4 * g4 + 2] * 0.125f, dq1[4 * g4 + 3] * 0.125f));
                } else {
                    float* out = p.dqkv + oo;
                    *reinterpret_cast<float4*>(out + d) = make_float4(dq0[4 * g4] * 0.125f, dq0[4 * g4 + 1] * 0.125f, dq0[4 * g4 + 2] * 0.125f, dq0[4 * g4 + 3] * 0.125f);
                    *reinterpret_cast<float4*>(out + 32 + d) = make_float4(dq1[4 * g4] * 0.125f, dq1[4 * g4 + 1] * 0.125f, dq1[4 * g4 + 2] * 0.125f, dq1[4 * g4 + 3] * 0.125f);
                }
            }
        }
    }
    if (FWD) return;
    __syncthreads();   // every LSE is in LDS

    // ================= pass 3: a lane owns a key =================
    for (int kb = wave; kb < nb; kb += nwaves) {
        const int key = kb * 32 + lr;
        const bool kvalid = key < T;
        const bool kvis = (mrow[key >> 5] >> (key & 31)) & 1u;
        const bool kon = kvalid && (MODE == AG_MASK_VIT_MUL || kvis);   // BERT: a masked key has weight exactly 0
        bf16x8_t kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = row_frag(iK, kb, ks); vf[ks] = row_frag(iV, kb, ks); }
        f32x16_t dv0, dv1, dk0, dk1; zero16(dv0); zero16(dv1); zero16(dk0); zero16(dk1);
        for (int qb = 0; qb < nb; ++qb) {
            f32x16_t s, dp; zero16(s); zero16(dp);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(iQ, qb, ks), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(iO, qb, ks), vf[ks], dp, 0, 0, 0);
            }
            // reg i <-> query qb*32 + (i&3) + 8(i>>2) + 4lh; one 16-query k-step (regs 8st .. 8st+7) at a time
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float ptv8[8], dsv8[8];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g4 = 2 * st + gg;
                    const int q0 = qb * 32 + 8 * g4 + 4 * lh;
                    const float4 ls = *reinterpret_cast<const float4*>(sL + q0), dd = *reinterpret_cast<const float4*>(sD + q0);
                    const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * g4 + e, qi = q0 + e;
                        float pr = __builtin_amdgcn_exp2f(fmaf(s[i], c2, -lsv[e]));
                        if (!kon || qi >= T) pr = 0.f;
                        float ptv = pr, dpv = dp[i];
                        if (drop) {
                            const bool keep = keep_elem(p.seed, (hbase + qi) * T + key, p.pdrop);
                            ptv = keep ? pr : 0.f;             // dV is scaled by 1/(1-p) once, in fp32, at the store
                            dpv = keep ? dpv * keep_sc : 0.f;
                        }
                        float v = pr * (dpv - ddv[e]);
                        if (MODE == AG_MASK_VIT_MUL && !kvis) v = 0.f;
                        ptv8[4 * gg + e] = ptv; dsv8[4 * gg + e] = v;
                    }
                }
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, make_uint4(pack_bf16x2(ptv8[0], ptv8[1]), pack_bf16x2(ptv8[2], ptv8[3]),
                                                                            pack_bf16x2(ptv8[4], ptv8[5]), pack_bf16x2(ptv8[6], ptv8[7])));
                const bf16x8_t sf = __builtin_bit_cast(bf16x8_t, make_uint4(pack_bf16x2(dsv8[0], dsv8[1]), pack_bf16x2(dsv8[2], dsv8[3]),
                                                                            pack_bf16x2(dsv8[4], dsv8[5]), pack_bf16x2(dsv8[6], dsv8[7])));
                dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(iO, qb, st, 0), pf, dv0, 0, 0, 0);
                dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(iO, qb, st, 1), pf, dv1, 0, 0, 0);
                dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(iQ, qb, st, 0), sf, dk0, 0, 0, 0);
                dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(iQ, qb, st, 1), sf, dk1, 0, 0, 0);
            }
        }
        if (kvalid) {
            const long ok_ = ((long)row * T + key) * ts + p.H + (long)head * 64, ov_ = ok_ + p.H;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 8 * g4 + 4 * lh;
                if (IO16) {
                    *reinterpret_cast<uint2*>(p.out16 + ok_ + d) = make_uint2(pack_bf16x2(dk0[4 * g4] * 0.125f, dk0[4 * g4 + 1] * 0.125f), pack_bf16x2(dk0[4 * g4 + 2] * 0.125f, dk0[4 * g4 + 3] * 0.125f));
                    *reinterpret_cast<uint2*>(p.out16 + ok_ + 32 + d) = make_uint2(pack_bf16x2(dk1[4 * g4] * 0.125f, dk1[4 * g4 + 1] * 0.125f), pack_bf16x2(dk1[4 * g4 + 2] * 0.125f, dk1[4 * g4 + 3] * 0.125f));
                    *reinterpret_cast<uint2*>(p.out16 + ov_ + d) = make_uint2(pack_bf16x2(dv0[4 * g4] * keep_sc, dv0[4 * g4 + 1] * keep_sc), pack_bf16x2(dv0[4 * g4 + 2] * keep_sc, dv0[4 * g4 + 3] * keep_sc));
                    *reinterpret_cast<uint2*>(p.out16 + ov_ + 32 + d) = make_uint2(pack_bf16x2(dv1[4 * g4] * keep_sc, dv1[4 * g4 + 1] * keep_sc), pack_bf16x2(dv1[4 * g4 + 2] * keep_sc, dv1[4 * g4 + 3] * keep_sc));
                } else {
                    float* outk = p.dqkv + ok_;
                    float* outv = p.dqkv + ov_;
                    *reinterpret_cast<float4*>(outk + d) = make_float4(dk0[4 * g4] * 0.125f, dk0[4 * g4 + 1] * 0.125f, dk0[4 * g4 + 2] * 0.125f, dk0[4 * g4 + 3] * 0.125f);
                    *reinterpret_cast<float4*>(outk + 32 + d) = make_float4(dk1[4 * g4] * 0.125f, dk1[4 * g4 + 1] * 0.125f, dk1[4 * g4 + 2] * 0.125f, dk1[4 * g4 + 3] * 0.125f);
                    *reinterpret_cast<float4*>(outv + d) = make_float4(dv0[4 * g4] * keep_sc, dv0[4 * g4 + 1] * keep_sc, dv0[4 * g4 + 2] * keep_sc, dv0[4 * g4 + 3] * keep_sc);
                    *reinterpret_cast<float4*>(outv + 32 + d) = make_float4(dv1[4 * g4] * keep_sc, dv1[4 * g4 + 1] * keep_sc, dv1[4 * g4 + 2] * keep_sc, dv1[4 * g4 + 3] * keep_sc);
                }
            }
        }
    }
}

}  // namespace

int ag_set_salt_train(uint32_t salt, hipStream_t s) { return set_salt_here(salt, s); }

extern "C" int ag_transpose_f32(const float* d_src, int rows, int cols, int64_t lds, float* d_dst, int64_t ldd, void* stream) {
    AG_REQUIRE(d_src && d_dst && rows >= 0 && cols >= 0 && lds >= cols && ldd >= rows, "ag_transpose_f32: bad arguments");
    if (rows == 0 || cols == 0) return AG_OK;
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, lds, d_dst, ldd);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
static bool ct64_ok(const void* src, int cols, int64_t lds, int64_t ldd) {
    return cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && ((uintptr_t)src % 16) == 0;
}
template <int ACT>
static void launch_ct64(const float* src, const float* aux, int rows, int cols, int64_t lds, bf16_t* dst, int64_t ldd, bf16_t* plain,
                        float* f32_out, hipStream_t s) {
    hipLaunchKernelGGL(cast_transpose64_kernel<ACT>, dim3(ceil_div(cols, 64), ceil_div((int)ldd, 64)), dim3(256), 0, s, src, aux, rows, cols, lds,
                       dst, ldd, plain, f32_out);
}
extern "C" int ag_transpose_f32_bf16(const float* d_src, int rows, int cols, int64_t lds, void* d_dst, int64_t ldd, void* stream) {
    AG_REQUIRE(d_src && d_dst && rows >= 0 && cols >= 0 && lds >= cols && ldd >= rows, "ag_transpose_f32_bf16: bad arguments");
    if (rows == 0 || cols == 0) return AG_OK;
    if (ct64_ok(d_src, cols, lds, ldd)) launch_ct64<0>(d_src, nullptr, rows, cols, lds, (bf16_t*)d_dst, ldd, nullptr, nullptr, (hipStream_t)stream);
    else hipLaunchKernelGGL(transpose_bf16_kernel, dim3(ceil_div(cols, 32), ceil_div(ldd, 32)), dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, lds,
                            (bf16_t*)d_dst, ldd, (bf16_t*)nullptr);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_cast_transpose_f32_bf16(const float* d_src, int rows, int cols, int64_t lds, void* d_plain, void* d_dst_t, int64_t ldd,
                                          void* stream) {
    AG_REQUIRE(d_src && d_plain && d_dst_t && rows >= 0 && cols >= 0 && lds >= cols && ldd >= rows, "ag_cast_transpose_f32_bf16: bad arguments");
    if (rows == 0 || cols == 0) return AG_OK;
    if (ct64_ok(d_src, cols, lds, ldd)) launch_ct64<0>(d_src, nullptr, rows, cols, lds, (bf16_t*)d_dst_t, ldd, (bf16_t*)d_plain, nullptr, (hipStream_t)stream);
    else hipLaunchKernelGGL(transpose_bf16_kernel, dim3(ceil_div(cols, 32), ceil_div(ldd, 32)), dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, lds,
                            (bf16_t*)d_dst_t, ldd, (bf16_t*)d_plain);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_gelu_cast_transpose_f32_bf16(const float* d_u, int rows, int cols, void* d_plain, void* d_dst_t, int64_t ldd, void* stream) {
    AG_REQUIRE(d_u && d_plain && d_dst_t && rows >= 0 && cols >= 0 && ldd >= rows, "ag_gelu_cast_transpose_f32_bf16: bad arguments");
    if (rows == 0 || cols == 0) return AG_OK;
    AG_REQUIRE(ct64_ok(d_u, cols, cols, ldd), "ag_gelu_cast_transpose_f32_bf16: cols and ldd must be multiples of 4, rows 16-byte aligned");
    launch_ct64<1>(d_u, nullptr, rows, cols, cols, (bf16_t*)d_dst_t, ldd, (bf16_t*)d_plain, nullptr, (hipStream_t)stream);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_gelu_bwd_cast_transpose_f32_bf16(const float* d_u, const float* d_dy, int rows, int cols, float* d_du, void* d_plain,
                                                   void* d_dst_t, int64_t ldd, void* stream) {
    AG_REQUIRE(d_u && d_dy && d_plain && d_dst_t && rows >= 0 && cols >= 0 && ldd >= rows, "ag_gelu_bwd_cast_transpose_f32_bf16: bad arguments");
    if (rows == 0 || cols == 0) return AG_OK;
    AG_REQUIRE(ct64_ok(d_dy, cols, cols, ldd) && ((uintptr_t)d_u % 16) == 0 && (!d_du || ((uintptr_t)d_du % 16) == 0),
               "ag_gelu_bwd_cast_transpose_f32_bf16: cols and ldd must be multiples of 4, rows 16-byte aligned");
    launch_ct64<2>(d_dy, d_u, rows, cols, cols, (bf16_t*)d_dst_t, ldd, (bf16_t*)d_plain, d_du, (hipStream_t)stream);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_colsum_f32(const float* d_x, int M, int N, int64_t ldx, float* d_out, int accumulate, void* stream) {
    AG_REQUIRE(d_x && d_out && M >= 0 && N >= 1, "ag_colsum_f32: bad arguments");
    const int vec_ok = ((ldx % 4) == 0 && ((uintptr_t)d_x % 16) == 0) ? 1 : 0;   // float4 row segments
    // 16-column blocks (64-byte row segments) unless 32-column ones (whole 128-byte lines) already give the chip >= 64 blocks
    if (N >= 64 * 32) hipLaunchKernelGGL(colsum_kernel<32>, dim3(ceil_div(N, 32)), dim3(256), 0, (hipStream_t)stream, d_x, M, N, ldx, d_out, accumulate, vec_ok);
    else hipLaunchKernelGGL(colsum_kernel<16>, dim3(ceil_div(N, 16)), dim3(256), 0, (hipStream_t)stream, d_x, M, N, ldx, d_out, accumulate, vec_ok);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_gelu_f32(const float* d_u, float* d_y, int64_t n, void* stream) {
    AG_REQUIRE(d_u && d_y && n >= 0, "ag_gelu_f32: bad arguments");
    if (n) hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_u, d_y, n);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_gelu_bwd_f32(const float* d_u, const float* d_dy, float* d_du, int64_t n, void* stream) {
    AG_REQUIRE(d_u && d_dy && d_du && n >= 0, "ag_gelu_bwd_f32: bad arguments");
    if (n) hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_u, d_dy, d_du, n);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_tanh_bwd_f32(const float* d_y, const float* d_dy, float* d_dx, int64_t n, void* stream) {
    AG_REQUIRE(d_y && d_dy && d_dx && n >= 0, "ag_tanh_bwd_f32: bad arguments");
    if (n) hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_y, d_dy, d_dx, n);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_add_f32(const float* d_a, const float* d_b, float* d_y, int64_t n, void* stream) {
    AG_REQUIRE(d_a && d_b && d_y && n >= 0, "ag_add_f32: bad arguments");
    if (n) hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_a, d_b, d_y, n);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_dropout_f32(const float* d_x, float* d_y, int64_t n, float p, uint32_t seed, void* stream) {
    AG_REQUIRE(d_x && d_y && n >= 0 && p >= 0.f && p < 1.f, "ag_dropout_f32: bad arguments");
    if (n) hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_x, d_y, n, p, seed);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_dropout_add_f32(const float* d_x, const float* d_resid, float* d_y, int64_t n, float p, uint32_t seed, void* stream) {
    AG_REQUIRE(d_x && d_resid && d_y && n >= 0 && p >= 0.f && p < 1.f, "ag_dropout_add_f32: bad arguments");
    if (n) hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_x, d_resid, d_y, n, p, seed);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_softmax_rows_bwd(const float* d_y, const float* d_dy, float* d_dx, int rows, int C, void* stream) {
    AG_REQUIRE(d_y && d_dy && d_dx && rows >= 0 && C >= 1, "ag_softmax_rows_bwd: bad arguments");
    if (rows) hipLaunchKernelGGL(softmax_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, d_y, d_dy, d_dx, rows, C);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_layernorm_bwd(const float* d_x, const float* d_gamma, const float* d_dy, int rows, int H, float eps,
                                float* d_dx, float* d_dgamma, float* d_dbeta, int accumulate, float* d_scratch, void* stream) {
    return ag_layernorm_bwd_add(d_x, d_gamma, d_dy, nullptr, rows, H, eps, d_dx, d_dgamma, d_dbeta, accumulate, d_scratch, stream);
}
extern "C" int ag_layernorm_bwd_add(const float* d_x, const float* d_gamma, const float* d_dy, const float* d_add, int rows, int H, float eps,
                                    float* d_dx, float* d_dgamma, float* d_dbeta, int accumulate, float* d_scratch, void* stream) {
    AG_REQUIRE(d_x && d_dy && d_dx && d_scratch && rows >= 0 && H >= 1 && H <= 4096, "ag_layernorm_bwd: bad arguments");
    if (rows == 0) return AG_OK;
    // >= 4 rows per wave; scratch: nblocks*2*H floats (<= 256*2*H).  (More, smaller blocks were measured: the row pass does not
    // get faster and the partial-sum reduction gets slower: 26.6 -> 41 us for the pair at 1 576 x 768.)
    const int nblocks = rows / 16 + 1 < 256 ? rows / 16 + 1 : 256;
    void (*kern)(const float*, const float*, const float*, int, int, float, float*, float*, const float*) = ln_bwd_kernel<0>;
    switch (H % 64 == 0 ? H / 64 : 0) {   // the hidden sizes of the shipped configurations keep the row in registers
        case 3: kern = ln_bwd_kernel<3>; break;      // 192 (ViT-tiny)
        case 12: kern = ln_bwd_kernel<12>; break;    // 768
        case 16: kern = ln_bwd_kernel<16>; break;    // 1024
        default: break;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), (size_t)2 * H * 4, (hipStream_t)stream, d_x, d_gamma, d_dy, rows, H, eps, d_dx, d_scratch, d_add);
    AG_LAUNCH_CHECK();
    if (d_dgamma && d_dbeta) {
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(H, 64)), dim3(256), 0, (hipStream_t)stream, d_scratch, nblocks, H, d_dgamma, d_dbeta, accumulate);
        AG_LAUNCH_CHECK();
    }
    return AG_OK;
}
extern "C" int ag_masked_attention_train_mixed(const float* d_qkv, const uint32_t* d_mask_bits, float* d_ctx, int R, int T, int H, int heads,
                                               int mask_mode, float p_drop, uint32_t seed, void* stream) {
    AG_REQUIRE(d_qkv && d_mask_bits && d_ctx, "ag_masked_attention_train_mixed: null pointer");
    AG_REQUIRE(heads > 0 && H == heads * 64 && R >= 0 && T >= 1 && T <= 256 && p_drop >= 0.f && p_drop < 1.f,
               "ag_masked_attention_train_mixed: needs head_dim 64 and T <= 256 (T=%d, H=%d, heads=%d)", T, H, heads);
    AG_REQUIRE(mask_mode == AG_MASK_VIT_MUL || mask_mode == AG_MASK_BERT_ADD, "ag_masked_attention_train_mixed: bad mask mode %d", mask_mode);
    if (R == 0) return AG_OK;
    AttnBwdArgs a;
    a.qkv = d_qkv; a.mask = d_mask_bits; a.ctx = nullptr; a.dctx = nullptr; a.dqkv = d_ctx; a.stats = nullptr;
    a.qkv16 = nullptr; a.out16 = nullptr; a.dslabs = 1; a.dslab_stride = 0;
    a.R = R; a.T = T; a.H = H; a.heads = heads; a.mode = mask_mode; a.Tw = (T + 31) / 32; a.pdrop = p_drop; a.seed = seed;
    const int Tp = (T + 31) & ~31;
    const size_t lds = (size_t)4 * Tp * BROW + (size_t)2 * Tp * sizeof(float);
    void (*kern)(AttnBwdArgs) = mask_mode == AG_MASK_VIT_MUL ? attn_bwd_mfma_kernel<AG_MASK_VIT_MUL, true> : attn_bwd_mfma_kernel<AG_MASK_BERT_ADD, true>;
    if (lds > 64 * 1024)
        AG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(R * heads), dim3(512), lds, (hipStream_t)stream, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_masked_attention_bwd_mixed(const float* d_qkv, const uint32_t* d_mask_bits, const float* d_ctx, const float* d_dctx,
                                             float* d_dqkv, int R, int T, int H, int heads, int mask_mode, float p_drop, uint32_t seed,
                                             void* stream) {
    AG_REQUIRE(d_qkv && d_mask_bits && d_ctx && d_dctx && d_dqkv, "ag_masked_attention_bwd_mixed: null pointer");
    AG_REQUIRE(heads > 0 && H == heads * 64 && R >= 0 && T >= 1 && T <= 256 && p_drop >= 0.f && p_drop < 1.f,
               "ag_masked_attention_bwd_mixed: needs head_dim 64 and T <= 256 (T=%d, H=%d, heads=%d)", T, H, heads);
    AG_REQUIRE(mask_mode == AG_MASK_VIT_MUL || mask_mode == AG_MASK_BERT_ADD, "ag_masked_attention_bwd_mixed: bad mask mode %d", mask_mode);
    if (R == 0) return AG_OK;
    AttnBwdArgs a;
    a.qkv = d_qkv; a.mask = d_mask_bits; a.ctx = d_ctx; a.dctx = d_dctx; a.dqkv = d_dqkv; a.stats = nullptr;
    a.qkv16 = nullptr; a.out16 = nullptr; a.dslabs = 1; a.dslab_stride = 0;
    a.R = R; a.T = T; a.H = H; a.heads = heads; a.mode = mask_mode; a.Tw = (T + 31) / 32; a.pdrop = p_drop; a.seed = seed;
    const int Tp = (T + 31) & ~31;
    const size_t lds = (size_t)4 * Tp * BROW + (size_t)2 * Tp * sizeof(float);
    void (*kern)(AttnBwdArgs) = mask_mode == AG_MASK_VIT_MUL ? attn_bwd_mfma_kernel<AG_MASK_VIT_MUL, false> : attn_bwd_mfma_kernel<AG_MASK_BERT_ADD, false>;
    if (lds > 64 * 1024)
        AG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(R * heads), dim3(512), lds, (hipStream_t)stream, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
// bf16-activation training step: qkv [R,T,3H] bf16 -> ctx [R,T,H] bf16 (forward), and dqkv [R,T,3H] bf16 from dctx given as
// `dslabs` fp32 partial slabs [dslabs][R*T][H] (the split-K out-projection dX of ag_gemm_ex), summed in slab order on load.
static int attn_mixed16(bool fwd, const void* d_qkv, const uint32_t* d_mask_bits, const float* d_dctx, int dslabs, int64_t dslab_stride,
                        void* d_out, int R, int T, int H, int heads, int mask_mode, float p_drop, uint32_t seed, void* stream) {
    AG_REQUIRE(d_qkv && d_mask_bits && d_out && (fwd || (d_dctx && dslabs >= 1)), "ag_masked_attention_*_bf16: null pointer");
    AG_REQUIRE(heads > 0 && H == heads * 64 && R >= 0 && T >= 1 && T <= 256 && p_drop >= 0.f && p_drop < 1.f,
               "ag_masked_attention_*_bf16: needs head_dim 64 and T <= 256 (T=%d, H=%d, heads=%d)", T, H, heads);
    AG_REQUIRE(mask_mode == AG_MASK_VIT_MUL || mask_mode == AG_MASK_BERT_ADD, "ag_masked_attention_*_bf16: bad mask mode %d", mask_mode);
    if (R == 0) return AG_OK;
    AttnBwdArgs a;
    a.qkv = nullptr; a.mask = d_mask_bits; a.ctx = nullptr; a.dctx = d_dctx; a.dqkv = nullptr; a.stats = nullptr;
    a.R = R; a.T = T; a.H = H; a.heads = heads; a.mode = mask_mode; a.Tw = (T + 31) / 32; a.pdrop = p_drop; a.seed = seed;
    a.qkv16 = (const bf16_t*)d_qkv; a.out16 = (bf16_t*)d_out; a.dslabs = dslabs; a.dslab_stride = (long)dslab_stride;
    const int Tp = (T + 31) & ~31;
    const size_t lds = (size_t)4 * Tp * BROW + (size_t)2 * Tp * sizeof(float);
    void (*kern)(AttnBwdArgs);
    if (fwd) kern = mask_mode == AG_MASK_VIT_MUL ? attn_bwd_mfma_kernel<AG_MASK_VIT_MUL, true, true> : attn_bwd_mfma_kernel<AG_MASK_BERT_ADD, true, true>;
    else kern = mask_mode == AG_MASK_VIT_MUL ? attn_bwd_mfma_kernel<AG_MASK_VIT_MUL, false, true> : attn_bwd_mfma_kernel<AG_MASK_BERT_ADD, false, true>;
    if (lds > 64 * 1024)
        AG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(R * heads), dim3(512), lds, (hipStream_t)stream, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_masked_attention_train_bf16(const void* d_qkv, const uint32_t* d_mask_bits, void* d_ctx, int R, int T, int H, int heads,
                                              int mask_mode, float p_drop, uint32_t seed, void* stream) {
    return attn_mixed16(true, d_qkv, d_mask_bits, nullptr, 0, 0, d_ctx, R, T, H, heads, mask_mode, p_drop, seed, stream);
}
extern "C" int ag_masked_attention_bwd_bf16(const void* d_qkv, const uint32_t* d_mask_bits, const float* d_dctx, int dslabs,
                                            int64_t dslab_stride, void* d_dqkv, int R, int T, int H, int heads, int mask_mode, float p_drop,
                                            uint32_t seed, void* stream) {
    return attn_mixed16(false, d_qkv, d_mask_bits, d_dctx, dslabs, dslab_stride, d_dqkv, R, T, H, heads, mask_mode, p_drop, seed, stream);
}
extern "C" int ag_masked_attention_bwd(const float* d_qkv, const uint32_t* d_mask_bits, const float* d_ctx, const float* d_dctx,
                                       float* d_dqkv, float* d_stats, int R, int T, int H, int heads, int mask_mode,
                                       float p_drop, uint32_t seed, void* stream) {
    AG_REQUIRE(d_qkv && d_mask_bits && d_ctx && d_dctx && d_dqkv && d_stats, "ag_masked_attention_bwd: null pointer");
    AG_REQUIRE(heads > 0 && H % heads == 0 && R >= 0 && T >= 1 && p_drop >= 0.f && p_drop < 1.f, "ag_masked_attention_bwd: bad arguments");
    if (R == 0) return AG_OK;
    AttnBwdArgs a;
    a.qkv = d_qkv; a.mask = d_mask_bits; a.ctx = d_ctx; a.dctx = d_dctx; a.dqkv = d_dqkv; a.stats = d_stats;
    a.qkv16 = nullptr; a.out16 = nullptr; a.dslabs = 1; a.dslab_stride = 0;
    a.R = R; a.T = T; a.H = H; a.heads = heads; a.mode = mask_mode; a.Tw = (T + 31) / 32; a.pdrop = p_drop; a.seed = seed;
    const dim3 grid(R * heads), block(256);
    hipStream_t hs = (hipStream_t)stream;
    switch (H / heads) {
        case 8: hipLaunchKernelGGL(attn_bwd_query_kernel<8>, grid, block, 0, hs, a); break;
        case 16: hipLaunchKernelGGL(attn_bwd_query_kernel<16>, grid, block, 0, hs, a); break;
        case 32: hipLaunchKernelGGL(attn_bwd_query_kernel<32>, grid, block, 0, hs, a); break;
        case 64: hipLaunchKernelGGL(attn_bwd_query_kernel<64>, grid, block, 0, hs, a); break;
        default: return ag_fail(AG_ERR_INVALID, "ag_masked_attention_bwd: head_dim %d not built (8, 16, 32, 64)", H / heads);
    }
    AG_LAUNCH_CHECK();
    switch (H / heads) {
        case 8: hipLaunchKernelGGL(attn_bwd_key_kernel<8>, grid, block, 0, hs, a); break;
        case 16: hipLaunchKernelGGL(attn_bwd_key_kernel<16>, grid, block, 0, hs, a); break;
        case 32: hipLaunchKernelGGL(attn_bwd_key_kernel<32>, grid, block, 0, hs, a); break;
        default: hipLaunchKernelGGL(attn_bwd_key_kernel<64>, grid, block, 0, hs, a); break;
    }
    AG_LAUNCH_CHECK();
    return AG_OK;
}
