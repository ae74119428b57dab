// train.hip — backward-pass building blocks for explainer / surrogate training (fp32).
//
// The training consumers run on B inputs (not B*K rows): reference scripts/train_explainer.py:184-198
// (explainer forward+backward once per batch) and scripts/train_surrogate.py:145-147.  The Linear
// backward GEMMs reuse ag_gemm (dX = dY·W as an NT GEMM against the transposed weight, dW = dYᵀ·X
// against transposed activations); this file adds what autograd needs around them: transposes, bias
// (column) sums, GELU / LayerNorm / soft-max / tanh backward, dropout, and the masked-attention
// backward (two recompute passes, no atomics, deterministic).
#include "common.h"

namespace {

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
__global__ void transpose_bf16_kernel(const float* __restrict__ src, int R, int Cc, int64_t lds_, bf16_t* __restrict__ dst, int64_t ldd) {
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
        if (c < Cc && r < ldd) dst[(int64_t)c * ldd + r] = (bf16_t)(pack_bf16x2(tile[tx][i], 0.f) & 0xFFFFu);
    }
}

// ---- column sums: out[n] (+)= sum_m x[m][n] ; block = 64 columns x one row slab (gridDim.y slabs), 4 row lanes ----
// One slab: plain store / read-modify-write.  Several slabs (tall inputs: a 64-column block alone would walk all M rows
// on one CU): partial sums are combined with float atomics into an output the launcher zeroed (or that accumulates).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int M, int N, int64_t ldx, float* __restrict__ out, int accumulate) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int m_lo = blockIdx.y * rows_per, m_hi = min(M, m_lo + rows_per);
    float s = 0.f;
    if (c < N) for (int m = m_lo + w; m < m_hi; m += 4) s += x[(int64_t)m * ldx + c];
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < N) {
        const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (gridDim.y > 1) atomicAdd(out + c, t);
        else out[c] = accumulate ? out[c] + t : t;
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
    const float sc = 1.0f / (1.0f - p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = keep_elem(seed, (uint64_t)i, p) ? x[i] * sc : 0.f;
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
                                                     int rows, int H, float eps, float* __restrict__ dx, float* __restrict__ part) {
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
                dx[(int64_t)row * H + lane + 64 * i] = rstd * (dv[i] * gv[i] - a - xv[i] * b);
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
                dx[(int64_t)row * H + c] = rstd * (dg - a - xh * b);
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

}  // namespace

extern "C" int ag_transpose_f32(const float* d_src, int rows, int cols, int64_t lds, float* d_dst, int64_t ldd, void* stream) {
    AG_REQUIRE(d_src && d_dst && rows >= 0 && cols >= 0 && lds >= cols && ldd >= rows, "ag_transpose_f32: bad arguments");
    if (rows == 0 || cols == 0) return AG_OK;
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, lds, d_dst, ldd);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_transpose_f32_bf16(const float* d_src, int rows, int cols, int64_t lds, void* d_dst, int64_t ldd, void* stream) {
    AG_REQUIRE(d_src && d_dst && rows >= 0 && cols >= 0 && lds >= cols && ldd >= rows, "ag_transpose_f32_bf16: bad arguments");
    if (rows == 0 || cols == 0) return AG_OK;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(ceil_div(cols, 32), ceil_div(ldd, 32)), dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, lds,
                       (bf16_t*)d_dst, ldd);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_colsum_f32(const float* d_x, int M, int N, int64_t ldx, float* d_out, int accumulate, void* stream) {
    AG_REQUIRE(d_x && d_out && M >= 0 && N >= 1, "ag_colsum_f32: bad arguments");
    int slabs = M / 128;                       // >= 128 rows per slab; enough blocks to cover the chip
    const int want = 1024 / ceil_div(N, 64);
    if (slabs > want) slabs = want;
    if (slabs < 1) slabs = 1;
    if (slabs > 1 && !accumulate) AG_HIP_CHECK(hipMemsetAsync(d_out, 0, (size_t)N * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64), slabs), dim3(256), 0, (hipStream_t)stream, d_x, M, N, ldx, d_out, accumulate);
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
extern "C" int ag_softmax_rows_bwd(const float* d_y, const float* d_dy, float* d_dx, int rows, int C, void* stream) {
    AG_REQUIRE(d_y && d_dy && d_dx && rows >= 0 && C >= 1, "ag_softmax_rows_bwd: bad arguments");
    if (rows) hipLaunchKernelGGL(softmax_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, d_y, d_dy, d_dx, rows, C);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
extern "C" int ag_layernorm_bwd(const float* d_x, const float* d_gamma, const float* d_dy, int rows, int H, float eps,
                                float* d_dx, float* d_dgamma, float* d_dbeta, int accumulate, float* d_scratch, void* stream) {
    AG_REQUIRE(d_x && d_dy && d_dx && d_scratch && rows >= 0 && H >= 1 && H <= 4096, "ag_layernorm_bwd: bad arguments");
    if (rows == 0) return AG_OK;
    const int nblocks = rows / 16 + 1 < 256 ? rows / 16 + 1 : 256;  // >= 4 rows per wave; scratch: nblocks*2*H floats (<= 256*2*H)
    void (*kern)(const float*, const float*, const float*, int, int, float, float*, float*) = ln_bwd_kernel<0>;
    switch (H % 64 == 0 ? H / 64 : 0) {   // the hidden sizes of the shipped configurations keep the row in registers
        case 3: kern = ln_bwd_kernel<3>; break;      // 192 (ViT-tiny)
        case 12: kern = ln_bwd_kernel<12>; break;    // 768
        case 16: kern = ln_bwd_kernel<16>; break;    // 1024
        default: break;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), (size_t)2 * H * 4, (hipStream_t)stream, d_x, d_gamma, d_dy, rows, H, eps, d_dx, d_scratch);
    AG_LAUNCH_CHECK();
    if (d_dgamma && d_dbeta) {
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(H, 64)), dim3(256), 0, (hipStream_t)stream, d_scratch, nblocks, H, d_dgamma, d_dbeta, accumulate);
        AG_LAUNCH_CHECK();
    }
    return AG_OK;
}
extern "C" int ag_masked_attention_bwd(const float* d_qkv, const uint32_t* d_mask_bits, const float* d_ctx, const float* d_dctx,
                                       float* d_dqkv, float* d_stats, int R, int T, int H, int heads, int mask_mode,
                                       float p_drop, uint32_t seed, void* stream) {
    AG_REQUIRE(d_qkv && d_mask_bits && d_ctx && d_dctx && d_dqkv && d_stats, "ag_masked_attention_bwd: null pointer");
    AG_REQUIRE(heads > 0 && H % heads == 0 && R >= 0 && T >= 1 && p_drop >= 0.f && p_drop < 1.f, "ag_masked_attention_bwd: bad arguments");
    if (R == 0) return AG_OK;
    AttnBwdArgs a;
    a.qkv = d_qkv; a.mask = d_mask_bits; a.ctx = d_ctx; a.dctx = d_dctx; a.dqkv = d_dqkv; a.stats = d_stats;
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
