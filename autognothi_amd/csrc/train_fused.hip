// train_fused.hip — the row kernels between the GEMMs of the bf16 training step (round 4).
//
// Every Linear of the step runs on ag_gemm_ex (gemm_tn.hip); where that GEMM is split over the contraction its result arrives as
// `splits` fp32 partial slabs [splits][M][H].  The kernels here are the consumers: they add the slabs in slab order (bit-reproducible)
// and do, in the same pass over the row, everything the reference's autograd graph does between two Linear layers
// (reference models/vanilla_vit.py:364-377 pre-LN block, models/vanilla_bert.py:410-427 post-LN block; backward: torch.autograd in
// scripts/train_explainer.py:183-198):
//   ag_rows_finish   t = resid + dropout(x + bias);  z = LayerNorm(t)          -> t fp32 (residual stream), z fp32 and / or bf16
//   ag_rows_ln_bwd   dy = x_slabs (+ dy_add);  dx = LayerNorm'(dy) (+ add)      -> dx fp32, bf16(dropout'(dx)) = the dY operand of the
//                    Linear below, column partials of dgamma / dbeta / that Linear's bias gradient
//   ag_slab_reduce   dst (+)= sum of slabs (a dW product split over the rows)
//   ag_colsum_bf16   bias gradient of a bf16 dY that a GEMM or attention epilogue produced
//   ag_cast_f32_many every trainable weight fp32 -> bf16 after the optimiser step, ONE launch (q | k | v land fused)
// One wave per row, the row in registers (H <= 1024), fp32 arithmetic throughout.
#include "common.h"

int ag_set_salt_fused(uint32_t salt, hipStream_t s);
int ag_set_salt_train(uint32_t salt, hipStream_t s);   // train.hip

namespace {
AG_DEFINE_DROPOUT_SALT(set_salt_here)

constexpr int MAXV = 4;   // float4 vectors per lane: H <= 1024

__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// v[i] = sum over slabs of the NV float4 vectors of one row, ADDED IN SLAB ORDER (bit-reproducible) with the loads of four slabs in
// flight at a time (a runtime-length loop of load -> add serialises one memory round trip per slab: 6 slabs x 3 vectors took the row
// kernels 20+ us at 1 576 rows)
template <int NV>
__device__ __forceinline__ void slab_sum(const float* __restrict__ slabs, int splits, long stride, long base, int lane, int H, float4 (&v)[NV]) {
    const float* p0 = slabs + base;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < H ? *reinterpret_cast<const float4*>(p0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int k = 1;
    for (; k + 3 < splits; k += 4) {
        float4 y[4][NV];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                y[j][i] = c < H ? *reinterpret_cast<const float4*>(p0 + (long)(k + j) * stride + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NV; ++i) add4(v[i], y[j][i]);
    }
    for (; k < splits; ++k) {
        float4 y[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            y[i] = c < H ? *reinterpret_cast<const float4*>(p0 + (long)k * stride + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) add4(v[i], y[i]);
    }
}

struct RowsArgs {
    const float* slabs; int splits; long slab_stride;
    const float* bias;
    float pdrop; uint32_t seed;
    const float* resid;
    float* t_out;
    const float* gamma; const float* beta; float eps;
    float* z_f32; bf16_t* z_bf16;
    int M, H;
};

template <int NV>
__global__ __launch_bounds__(256) void rows_finish_kernel(RowsArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.M) return;
    p.seed = ag_salted(p.seed);
    const long base = (long)row * p.H;
    float4 v[NV];
    bool on[NV];
    const float sc = 1.0f / (1.0f - p.pdrop);
    slab_sum<NV>(p.slabs, p.splits, p.slab_stride, base, lane, p.H, v);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        on[i] = c < p.H;
        if (!on[i]) continue;
        float4 x = v[i];
        if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + c);
            x.x += b.x; x.y += b.y; x.z += b.z; x.w += b.w;
        }
        if (p.pdrop > 0.f) {
            const uint64_t i0 = (uint64_t)(base + c);
            x.x = keep_elem(p.seed, i0, p.pdrop) ? x.x * sc : 0.f;
            x.y = keep_elem(p.seed, i0 + 1, p.pdrop) ? x.y * sc : 0.f;
            x.z = keep_elem(p.seed, i0 + 2, p.pdrop) ? x.z * sc : 0.f;
            x.w = keep_elem(p.seed, i0 + 3, p.pdrop) ? x.w * sc : 0.f;
        }
        if (p.resid) {
            const float4 r = *reinterpret_cast<const float4*>(p.resid + base + c);
            x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
        }
        if (p.t_out) *reinterpret_cast<float4*>(p.t_out + base + c) = x;
        v[i] = x;
    }
    if (p.gamma) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = wave_sum(s) / (float)p.H;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (on[i]) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        const float rstd = rsqrtf(wave_sum(sq) / (float)p.H + p.eps);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (on[i]) {
                const int c = (i * 64 + lane) * 4;
                const float4 g = *reinterpret_cast<const float4*>(p.gamma + c), b = *reinterpret_cast<const float4*>(p.beta + c);
                v[i].x = fmaf(v[i].x * rstd, g.x, b.x); v[i].y = fmaf(v[i].y * rstd, g.y, b.y);
                v[i].z = fmaf(v[i].z * rstd, g.z, b.z); v[i].w = fmaf(v[i].w * rstd, g.w, b.w);
            }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (on[i]) {
            const int c = (i * 64 + lane) * 4;
            if (p.z_f32) *reinterpret_cast<float4*>(p.z_f32 + base + c) = v[i];
            if (p.z_bf16) *reinterpret_cast<uint2*>(p.z_bf16 + base + c) = make_uint2(pack_bf16x2(v[i].x, v[i].y), pack_bf16x2(v[i].z, v[i].w));
        }
}

struct LnBwdArgs {
    const float* slabs; int splits; long slab_stride;
    const float* dy_add;     // joins dy BEFORE the LayerNorm backward (post-LN blocks)
    const float* x;          // the LayerNorm's input rows; NULL: no LayerNorm (dx = dy)
    const float* gamma; float eps;
    const float* add;        // joins dx AFTER it (pre-LN blocks: the residual branch)
    float* dx;
    bf16_t* dx_bf16; float pdrop; uint32_t seed;
    float* part;             // [blocks][3][H]: dgamma, dbeta, colsum(dropout'(dx)) partials of the block's rows
    int M, H;
};

// blocks of 4 waves; a wave walks rows blockIdx*4 + wave, + gridDim*4, ...; column partials are carried in registers over the wave's
// rows, folded through LDS in wave order and stored per block (no atomics: the reduce kernel adds blocks in order)
template <int NV>
__global__ __launch_bounds__(256) void rows_ln_bwd_kernel(LnBwdArgs p) {
    __shared__ float4 sacc[4][3][MAXV * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    p.seed = ag_salted(p.seed);
    float4 ag[NV], ab[NV], ac[NV], gv[NV];
    bool on[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        on[i] = c < p.H;
        ag[i] = ab[i] = ac[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gv[i] = (on[i] && p.gamma) ? *reinterpret_cast<const float4*>(p.gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    const float sc = 1.0f / (1.0f - p.pdrop);
    const float invH = 1.0f / (float)p.H;
    for (int row = blockIdx.x * 4 + wave; row < p.M; row += gridDim.x * 4) {
        const long base = (long)row * p.H;
        float4 dv[NV], xv[NV];
        float s = 0.f;
        slab_sum<NV>(p.slabs, p.splits, p.slab_stride, base, lane, p.H, dv);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!on[i]) continue;
            float4 d = dv[i];
            if (p.dy_add) {
                const float4 y = *reinterpret_cast<const float4*>(p.dy_add + base + c);
                d.x += y.x; d.y += y.y; d.z += y.z; d.w += y.w;
            }
            dv[i] = d;
            if (p.x) {
                xv[i] = *reinterpret_cast<const float4*>(p.x + base + c);
                s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
            }
        }
        float4 r[NV];
        if (p.x) {
            const float mean = wave_sum(s) * invH;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (on[i]) {
                    xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
                    sq += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
                }
            const float rstd = rsqrtf(wave_sum(sq) * invH + p.eps);
            float a = 0.f, b = 0.f;   // mean(dy g), mean(dy g xhat)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                xv[i].x *= rstd; xv[i].y *= rstd; xv[i].z *= rstd; xv[i].w *= rstd;
                const float g0 = dv[i].x * gv[i].x, g1 = dv[i].y * gv[i].y, g2 = dv[i].z * gv[i].z, g3 = dv[i].w * gv[i].w;
                if (on[i]) {
                    a += (g0 + g1) + (g2 + g3);
                    b += (g0 * xv[i].x + g1 * xv[i].y) + (g2 * xv[i].z + g3 * xv[i].w);
                }
            }
            a = wave_sum(a) * invH; b = wave_sum(b) * invH;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                r[i].x = rstd * (dv[i].x * gv[i].x - a - xv[i].x * b); r[i].y = rstd * (dv[i].y * gv[i].y - a - xv[i].y * b);
                r[i].z = rstd * (dv[i].z * gv[i].z - a - xv[i].z * b); r[i].w = rstd * (dv[i].w * gv[i].w - a - xv[i].w * b);
                ag[i].x = fmaf(dv[i].x, xv[i].x, ag[i].x); ag[i].y = fmaf(dv[i].y, xv[i].y, ag[i].y);
                ag[i].z = fmaf(dv[i].z, xv[i].z, ag[i].z); ag[i].w = fmaf(dv[i].w, xv[i].w, ag[i].w);
                ab[i].x += dv[i].x; ab[i].y += dv[i].y; ab[i].z += dv[i].z; ab[i].w += dv[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) r[i] = dv[i];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (!on[i]) continue;
            const int c = (i * 64 + lane) * 4;
            if (p.add) {
                const float4 y = *reinterpret_cast<const float4*>(p.add + base + c);
                r[i].x += y.x; r[i].y += y.y; r[i].z += y.z; r[i].w += y.w;
            }
            if (p.dx) *reinterpret_cast<float4*>(p.dx + base + c) = r[i];
            float4 d = r[i];
            if (p.pdrop > 0.f) {
                const uint64_t i0 = (uint64_t)(base + c);
                d.x = keep_elem(p.seed, i0, p.pdrop) ? d.x * sc : 0.f;
                d.y = keep_elem(p.seed, i0 + 1, p.pdrop) ? d.y * sc : 0.f;
                d.z = keep_elem(p.seed, i0 + 2, p.pdrop) ? d.z * sc : 0.f;
                d.w = keep_elem(p.seed, i0 + 3, p.pdrop) ? d.w * sc : 0.f;
            }
            if (p.dx_bf16) *reinterpret_cast<uint2*>(p.dx_bf16 + base + c) = make_uint2(pack_bf16x2(d.x, d.y), pack_bf16x2(d.z, d.w));
            ac[i].x += d.x; ac[i].y += d.y; ac[i].z += d.z; ac[i].w += d.w;
        }
    }
    if (!p.part) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        sacc[wave][0][i * 64 + lane] = ag[i];
        sacc[wave][1][i * 64 + lane] = ab[i];
        sacc[wave][2][i * 64 + lane] = ac[i];
    }
    __syncthreads();
    // 256 threads fold the four waves' partials in wave order: thread -> (kind, vector) pairs
    for (int idx = threadIdx.x; idx < 3 * NV * 64; idx += 256) {
        const int kind = idx / (NV * 64), vi = idx - kind * (NV * 64);
        const int c = vi * 4;
        if (c >= p.H) continue;
        float4 t = sacc[0][kind][vi];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 o = sacc[w][kind][vi];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        *reinterpret_cast<float4*>(p.part + ((long)blockIdx.x * 3 + kind) * p.H + c) = t;
    }
}

// out_k[c] (+)= sum over blocks of part[b][k][c], k = 0..2 (any out may be NULL).  grid = (ceil(H / 32), 3 kinds); a block = 32 columns x 8
// partial lanes, each lane walks every 8th partial with four loads in flight; the eight lanes are folded in a fixed order
__global__ __launch_bounds__(256) void part3_reduce_kernel(const float* __restrict__ part, int nblocks, int H, float* o0, float* o1, float* o2,
                                                           int accumulate) {
    __shared__ float sa[8][32];
    const int kind = blockIdx.y;
    float* out = kind == 0 ? o0 : (kind == 1 ? o1 : o2);
    if (!out) return;
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < H) {
        const float* base = part + (long)kind * H + c;
        const long st = (long)3 * H;
        int i = q;
        for (; i + 24 < nblocks; i += 32) {
            a0 += base[(long)i * st]; a1 += base[(long)(i + 8) * st]; a2 += base[(long)(i + 16) * st]; a3 += base[(long)(i + 24) * st];
        }
        for (; i < nblocks; i += 8) a0 += base[(long)i * st];
    }
    sa[q][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && c < H) {
        const float t = ((sa[0][cl] + sa[1][cl]) + (sa[2][cl] + sa[3][cl])) + ((sa[4][cl] + sa[5][cl]) + (sa[6][cl] + sa[7][cl]));
        out[c] = accumulate ? out[c] + t : t;
    }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int splits, long stride, long n4, float* __restrict__ dst,
                                                          int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 x = reinterpret_cast<const float4*>(slabs)[i];
        for (int s = 1; s < splits; ++s) {
            const float4 y = *reinterpret_cast<const float4*>(slabs + (long)s * stride + i * 4);
            x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        }
        if (accumulate) {
            const float4 y = reinterpret_cast<const float4*>(dst)[i];
            x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        }
        reinterpret_cast<float4*>(dst)[i] = x;
    }
}

// column sums of a bf16 [M, N] matrix (N % 8 == 0) in ONE launch: a block owns 32 columns (4 chunks of 8 = 16 bytes) for all rows, its
// 256 threads are 4 chunks x 64 row groups; row group g walks rows g, g + 64, ... in order and the 64 partials of a column are added
// in group order by one thread: a fixed summation tree, bit-reproducible, no scratch.  (The first version spread the rows over
// blocks and folded them in a second launch: two dependent launches of 8 us each for 5-10 MB of input.)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, int M, int N, long ldx, float* __restrict__ out, int accumulate) {
    __shared__ float sacc[64][4][8];
    const int cg = threadIdx.x & 3, rg = threadIdx.x >> 2;
    const int c = blockIdx.x * 32 + cg * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        int m = rg;
        for (; m + 192 < M; m += 256) {          // four independent loads in flight per thread
            uint4 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const uint4*>(x + (long)(m + 64 * j) * ldx + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[0] += __uint_as_float(u[j].x << 16); a[1] += __uint_as_float(u[j].x & 0xFFFF0000u);
                a[2] += __uint_as_float(u[j].y << 16); a[3] += __uint_as_float(u[j].y & 0xFFFF0000u);
                a[4] += __uint_as_float(u[j].z << 16); a[5] += __uint_as_float(u[j].z & 0xFFFF0000u);
                a[6] += __uint_as_float(u[j].w << 16); a[7] += __uint_as_float(u[j].w & 0xFFFF0000u);
            }
        }
        for (; m < M; m += 64) {
            const uint4 u = *reinterpret_cast<const uint4*>(x + (long)m * ldx + c);
            a[0] += __uint_as_float(u.x << 16); a[1] += __uint_as_float(u.x & 0xFFFF0000u);
            a[2] += __uint_as_float(u.y << 16); a[3] += __uint_as_float(u.y & 0xFFFF0000u);
            a[4] += __uint_as_float(u.z << 16); a[5] += __uint_as_float(u.z & 0xFFFF0000u);
            a[6] += __uint_as_float(u.w << 16); a[7] += __uint_as_float(u.w & 0xFFFF0000u);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sacc[rg][cg][j] = a[j];
    __syncthreads();
    if (threadIdx.x < 32) {
        const int col = blockIdx.x * 32 + threadIdx.x;
        if (col < N) {
            float t = 0.f;
            for (int g = 0; g < 64; ++g) t += sacc[g][threadIdx.x >> 3][threadIdx.x & 7];
            out[col] = accumulate ? out[col] + t : t;
        }
    }
}

// the same column sums for several matrices in one launch (round 6: the bias gradients of the Linears whose dW products go out as one
// grouped launch): matrix i owns blocks [first[i], first[i + 1]) of 32 columns each
constexpr int COLSUM_GROUP_MAX = 16;
struct ColsumGroup {
    const bf16_t* x[COLSUM_GROUP_MAX]; float* out[COLSUM_GROUP_MAX]; long ldx[COLSUM_GROUP_MAX];
    int M[COLSUM_GROUP_MAX], N[COLSUM_GROUP_MAX], first[COLSUM_GROUP_MAX + 1]; int count;
};
__global__ __launch_bounds__(256) void colsum_bf16_group_kernel(const ColsumGroup g) {
    __shared__ float sacc[64][4][8];
    int i = 0;
#pragma unroll
    for (int j = 1; j < COLSUM_GROUP_MAX; ++j) i += (j < g.count && (int)blockIdx.x >= g.first[j]) ? 1 : 0;
    const bf16_t* __restrict__ x = g.x[i];
    float* __restrict__ out = g.out[i];
    const int M = g.M[i], N = g.N[i], blk = (int)blockIdx.x - g.first[i];
    const long ldx = g.ldx[i];
    const int cg = threadIdx.x & 3, rg = threadIdx.x >> 2;
    const int c = blk * 32 + cg * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < N) {          // (the row walk and the summation tree of colsum_bf16_kernel: the same bits)
        int m = rg;
        for (; m + 192 < M; m += 256) {
            uint4 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const uint4*>(x + (long)(m + 64 * j) * ldx + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[0] += __uint_as_float(u[j].x << 16); a[1] += __uint_as_float(u[j].x & 0xFFFF0000u);
                a[2] += __uint_as_float(u[j].y << 16); a[3] += __uint_as_float(u[j].y & 0xFFFF0000u);
                a[4] += __uint_as_float(u[j].z << 16); a[5] += __uint_as_float(u[j].z & 0xFFFF0000u);
                a[6] += __uint_as_float(u[j].w << 16); a[7] += __uint_as_float(u[j].w & 0xFFFF0000u);
            }
        }
        for (; m < M; m += 64) {
            const uint4 u = *reinterpret_cast<const uint4*>(x + (long)m * ldx + c);
            a[0] += __uint_as_float(u.x << 16); a[1] += __uint_as_float(u.x & 0xFFFF0000u);
            a[2] += __uint_as_float(u.y << 16); a[3] += __uint_as_float(u.y & 0xFFFF0000u);
            a[4] += __uint_as_float(u.z << 16); a[5] += __uint_as_float(u.z & 0xFFFF0000u);
            a[6] += __uint_as_float(u.w << 16); a[7] += __uint_as_float(u.w & 0xFFFF0000u);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sacc[rg][cg][j] = a[j];
    __syncthreads();
    if (threadIdx.x < 32) {
        const int col = blk * 32 + threadIdx.x;
        if (col < N) {
            float t = 0.f;
            for (int gr = 0; gr < 64; ++gr) t += sacc[gr][threadIdx.x >> 3][threadIdx.x & 7];
            out[col] = t;
        }
    }
}

constexpr int CAST_MAX = 96;
struct CastTable {
    const float* src[CAST_MAX];
    void* dst[CAST_MAX];
    int first_block[CAST_MAX + 1];   // segment s owns blocks [first_block[s], first_block[s+1]); a block converts 4096 elements
    int64_t n[CAST_MAX];
    unsigned char f32[CAST_MAX];     // destination dtype: 0 bf16, 1 fp32 (a plain copy: fused q | k | v biases)
    int count;
    float scale;                     // every element times this (1: as it is; ag_pack_f32_many: a rank's share of the global batch)
};
__global__ __launch_bounds__(256) void cast_many_kernel(const CastTable t) {
    int lo = 0, hi = t.count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= t.first_block[mid]) lo = mid; else hi = mid;
    }
    const float* src = t.src[lo];
    const int64_t n = t.n[lo];
    const float sc = t.scale;        // (x * 1.0f is x: one code path)
    const int64_t i0 = ((int64_t)(blockIdx.x - t.first_block[lo]) * 256 + threadIdx.x) * 16;
    if (t.f32[lo]) {
        float* dstf = reinterpret_cast<float*>(t.dst[lo]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = i0 + 4 * j;
            if (i + 4 <= n) {
                const float4 a = *reinterpret_cast<const float4*>(src + i);
                *reinterpret_cast<float4*>(dstf + i) = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
            } else for (int64_t k = i; k < n && k < i + 4; ++k) dstf[k] = src[k] * sc;
        }
        return;
    }
    bf16_t* dst = reinterpret_cast<bf16_t*>(t.dst[lo]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t i = i0 + 8 * j;
        if (i + 8 <= n) {
            const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
            *reinterpret_cast<uint4*>(dst + i) = make_uint4(pack_bf16x2(a.x * sc, a.y * sc), pack_bf16x2(a.z * sc, a.w * sc),
                                                            pack_bf16x2(b.x * sc, b.y * sc), pack_bf16x2(b.z * sc, b.w * sc));
        } else {
            for (int64_t k = i; k < n && k < i + 8; ++k) dst[k] = f32_to_bf16(src[k] * sc);
        }
    }
}

// dst[m][c] = c < cols_src ? src[m][c] : 0 for c < cols_dst (fp32 or bf16 destination): pads the C-wide head output / gradient
// (C = 10 or 2 classes) to the 16 columns the GEMM wants, or strips the padding again
template <typename TD>
__global__ void pad_cols_kernel(const float* __restrict__ src, long ld_s, int cols_src, TD* __restrict__ dst, long ld_d, int cols_dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / cols_dst;
        const int c = (int)(i - m * cols_dst);
        const float v = c < cols_src ? src[m * ld_s + c] : 0.f;
        Store<TD>::store(dst + m * ld_d + c, v);
    }
}

}  // namespace

int ag_set_salt_fused(uint32_t salt, hipStream_t s) { return set_salt_here(salt, s); }

extern "C" int ag_set_dropout_salt(uint32_t salt, void* stream) {
    if (ag_set_salt_fused(salt, (hipStream_t)stream) != AG_OK || ag_set_salt_train(salt, (hipStream_t)stream) != AG_OK)
        return ag_fail(AG_ERR_HIP, "ag_set_dropout_salt: hipMemcpyToSymbolAsync failed");
    return AG_OK;
}

extern "C" int ag_pad_cols_f32(const float* d_src, int64_t ld_src, int cols_src, void* d_dst, int64_t ld_dst, int cols_dst, int dst_dtype,
                               int M, void* stream) {
    AG_REQUIRE(d_src && d_dst && M >= 0 && cols_src > 0 && cols_dst > 0 && ld_src >= cols_src && ld_dst >= cols_dst, "ag_pad_cols_f32: bad arguments");
    AG_REQUIRE(dst_dtype == AG_F32 || dst_dtype == AG_BF16, "ag_pad_cols_f32: bad dtype %d", dst_dtype);
    if (M == 0) return AG_OK;
    const long n = (long)M * cols_dst;
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    const int cs = cols_src < cols_dst ? cols_src : cols_dst;
    if (dst_dtype == AG_F32)
        hipLaunchKernelGGL(pad_cols_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_src, (long)ld_src, cs, (float*)d_dst, (long)ld_dst, cols_dst, n);
    else
        hipLaunchKernelGGL(pad_cols_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_src, (long)ld_src, cs, (bf16_t*)d_dst, (long)ld_dst, cols_dst, n);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_rows_finish(const float* d_x, int splits, int64_t slab_stride, const float* d_bias, float p_drop, uint32_t seed,
                              const float* d_resid, float* d_t_out, const float* d_gamma, const float* d_beta, float eps,
                              float* d_z_f32, void* d_z_bf16, int M, int H, void* stream) {
    if (M == 0) return AG_OK;
    AG_REQUIRE(d_x && splits >= 1 && M > 0 && H > 0, "ag_rows_finish: bad arguments");
    AG_REQUIRE(H % 4 == 0 && H <= 256 * MAXV, "ag_rows_finish: H=%d must be a multiple of 4 and <= %d", H, 256 * MAXV);
    AG_REQUIRE(splits == 1 || slab_stride % 4 == 0, "ag_rows_finish: slab stride must be a multiple of 4 floats");
    AG_REQUIRE(!d_gamma || d_beta, "ag_rows_finish: gamma without beta");
    AG_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "ag_rows_finish: p_drop=%f", (double)p_drop);
    RowsArgs a{d_x, splits, (long)slab_stride, d_bias, p_drop, seed, d_resid, d_t_out, d_gamma, d_beta, eps, d_z_f32, (bf16_t*)d_z_bf16, M, H};
    const dim3 grid(ceil_div(M, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (ceil_div(H, 256)) {
        case 1: hipLaunchKernelGGL(rows_finish_kernel<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(rows_finish_kernel<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(rows_finish_kernel<3>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(rows_finish_kernel<4>, grid, block, 0, s, a); break;
    }
    AG_LAUNCH_CHECK();
    return AG_OK;
}

static int ln_bwd_blocks(int M) { return M < 4 * 256 ? ceil_div(M, 4) : 256; }
extern "C" size_t ag_rows_ln_bwd_scratch_floats(int M, int H) {
    const int blocks = ln_bwd_blocks(M);
    return (size_t)(blocks > 0 ? blocks : 1) * 3 * (size_t)H;
}

extern "C" int ag_rows_ln_bwd(const float* d_dy, int splits, int64_t slab_stride, const float* d_dy_add, const float* d_x,
                              const float* d_gamma, float eps, const float* d_add, float* d_dx, void* d_dx_bf16, float p_drop,
                              uint32_t seed, float* d_dgamma, float* d_dbeta, float* d_dbias, int accumulate, float* d_scratch, int M,
                              int H, void* stream) {
    if (M == 0) return AG_OK;
    AG_REQUIRE(d_dy && splits >= 1 && M > 0 && H > 0, "ag_rows_ln_bwd: bad arguments");
    AG_REQUIRE(H % 4 == 0 && H <= 256 * MAXV, "ag_rows_ln_bwd: H=%d must be a multiple of 4 and <= %d", H, 256 * MAXV);
    AG_REQUIRE(splits == 1 || slab_stride % 4 == 0, "ag_rows_ln_bwd: slab stride must be a multiple of 4 floats");
    AG_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "ag_rows_ln_bwd: p_drop=%f", (double)p_drop);
    const bool want_part = d_dgamma || d_dbeta || d_dbias;
    AG_REQUIRE(!want_part || d_scratch, "ag_rows_ln_bwd: column sums need d_scratch (ag_rows_ln_bwd_scratch_floats)");
    AG_REQUIRE(!(d_dgamma || d_dbeta) || d_x, "ag_rows_ln_bwd: dgamma / dbeta without a LayerNorm (d_x == NULL)");
    const int blocks = ln_bwd_blocks(M);
    LnBwdArgs a{d_dy, splits, (long)slab_stride, d_dy_add, d_x, d_gamma, eps, d_add, d_dx, (bf16_t*)d_dx_bf16, p_drop, seed,
                want_part ? d_scratch : nullptr, M, H};
    hipStream_t s = (hipStream_t)stream;
    switch (ceil_div(H, 256)) {
        case 1: hipLaunchKernelGGL(rows_ln_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(rows_ln_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(rows_ln_bwd_kernel<3>, dim3(blocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(rows_ln_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, a); break;
    }
    AG_LAUNCH_CHECK();
    if (want_part) {
        hipLaunchKernelGGL(part3_reduce_kernel, dim3(ceil_div(H, 32), 3), dim3(256), 0, s, d_scratch, blocks, H, d_dgamma, d_dbeta, d_dbias, accumulate);
        AG_LAUNCH_CHECK();
    }
    return AG_OK;
}

extern "C" int ag_slab_reduce(const float* d_slabs, int splits, int64_t slab_stride, int64_t n, float* d_dst, int accumulate, void* stream) {
    if (n == 0) return AG_OK;
    AG_REQUIRE(d_slabs && d_dst && splits >= 1 && n > 0, "ag_slab_reduce: bad arguments");
    AG_REQUIRE(n % 4 == 0 && slab_stride % 4 == 0 && ((uintptr_t)d_dst % 16) == 0 && ((uintptr_t)d_slabs % 16) == 0,
               "ag_slab_reduce: n and the slab stride must be multiples of 4 floats, pointers 16-byte aligned");
    const long n4 = n / 4;
    const int blocks = (int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_slabs, splits, (long)slab_stride, n4, d_dst, accumulate);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" size_t ag_colsum_bf16_scratch_floats(int M, int N) { (void)M; (void)N; return 0; }   // (ABI 3 kept the query; no scratch since the one-launch form)

extern "C" int ag_colsum_bf16(const void* d_x, int M, int N, int64_t ldx, float* d_out, int accumulate, float* d_scratch, void* stream) {
    (void)d_scratch;
    AG_REQUIRE(d_x && d_out && M >= 0 && N > 0, "ag_colsum_bf16: bad arguments");
    AG_REQUIRE(N % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)d_x % 16) == 0, "ag_colsum_bf16: N and ldx must be multiples of 8, x 16-byte aligned");
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(ceil_div(N, 32)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_x, M, N, (long)ldx, d_out, accumulate);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_colsum_bf16_group(int count, const void* const* d_x, const int* M, const int* N, const int64_t* ldx, float* const* d_out, void* stream) {
    AG_REQUIRE(count >= 0 && (count == 0 || (d_x && M && N && ldx && d_out)), "ag_colsum_bf16_group: bad arguments");
    for (int base = 0; base < count; base += COLSUM_GROUP_MAX) {
        ColsumGroup g{};
        int n = 0, blocks = 0;
        for (int i = base; i < count && i < base + COLSUM_GROUP_MAX; ++i) {
            AG_REQUIRE(d_x[i] && d_out[i] && M[i] >= 0 && N[i] > 0, "ag_colsum_bf16_group: bad matrix %d", i);
            AG_REQUIRE(N[i] % 8 == 0 && ldx[i] % 8 == 0 && ((uintptr_t)d_x[i] % 16) == 0, "ag_colsum_bf16_group: matrix %d: N and ldx must be multiples of 8, x 16-byte aligned", i);
            g.x[n] = (const bf16_t*)d_x[i]; g.out[n] = d_out[i]; g.ldx[n] = (long)ldx[i]; g.M[n] = M[i]; g.N[n] = N[i];
            g.first[n] = blocks;
            blocks += ceil_div(N[i], 32);
            ++n;
        }
        if (n == 0) continue;
        g.first[n] = blocks;
        g.count = n;
        hipLaunchKernelGGL(colsum_bf16_group_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
        AG_LAUNCH_CHECK();
    }
    return AG_OK;
}

static int cast_many_impl(const float* const* h_src, void* const* h_dst, const int64_t* h_n, const int* h_dst_dtype, int count, float scale,
                          void* stream);
extern "C" int ag_cast_f32_many(const float* const* h_src, void* const* h_dst, const int64_t* h_n, const int* h_dst_dtype, int count,
                                void* stream) {
    return cast_many_impl(h_src, h_dst, h_n, h_dst_dtype, count, 1.0f, stream);
}
extern "C" int ag_pack_f32_many(const float* const* h_src, void* const* h_dst, const int64_t* h_n, const int* h_dst_dtype, int count,
                                float scale, void* stream) {
    return cast_many_impl(h_src, h_dst, h_n, h_dst_dtype, count, scale, stream);
}
static int cast_many_impl(const float* const* h_src, void* const* h_dst, const int64_t* h_n, const int* h_dst_dtype, int count, float scale,
                          void* stream) {
    AG_REQUIRE(count >= 0 && (count == 0 || (h_src && h_dst && h_n && h_dst_dtype)), "ag_cast_f32_many: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < count; base += CAST_MAX) {
        CastTable t;
        t.scale = scale;
        t.count = count - base < CAST_MAX ? count - base : CAST_MAX;
        int blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            AG_REQUIRE(h_src[base + i] && h_dst[base + i] && h_n[base + i] >= 0, "ag_cast_f32_many: null segment %d", base + i);
            AG_REQUIRE(((uintptr_t)h_src[base + i] % 16) == 0 && ((uintptr_t)h_dst[base + i] % 16) == 0,
                       "ag_cast_f32_many: segment %d is not 16-byte aligned", base + i);
            AG_REQUIRE(h_dst_dtype[base + i] == AG_BF16 || h_dst_dtype[base + i] == AG_F32, "ag_cast_f32_many: segment %d: bad dtype", base + i);
            t.src[i] = h_src[base + i]; t.dst[i] = h_dst[base + i]; t.n[i] = h_n[base + i]; t.f32[i] = h_dst_dtype[base + i] == AG_F32;
            t.first_block[i] = blocks;
            blocks += (int)((h_n[base + i] + 4095) / 4096);
        }
        t.first_block[t.count] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(cast_many_kernel, dim3(blocks), dim3(256), 0, s, t);
        AG_LAUNCH_CHECK();
    }
    return AG_OK;
}
