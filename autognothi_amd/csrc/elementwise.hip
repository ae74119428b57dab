// elementwise.hip — the HBM-bound pieces of the masked forward: LayerNorm, embeddings, soft-max
// heads, dtype casts.  One wave per row, 16-byte accesses, fp32 statistics.
#include "common.h"

namespace {

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        Store<T>::store(dst + i, v.x); Store<T>::store(dst + i + 1, v.y);
        Store<T>::store(dst + i + 2, v.z); Store<T>::store(dst + i + 3, v.w);
    }
    if (i < n) for (int64_t j = i; j < n && j < i + 4; ++j) Store<T>::store(dst + j, src[j]);
}

// torch.nn.LayerNorm: biased variance, y = (x-mean)*rsqrt(var+eps)*g + b.  One wave per row; the row
// is held in registers (H <= 64*4*MAXV floats) so x is read from HBM exactly once.
constexpr int LN_MAXV = 8;  // supports H up to 2048
template <typename XT, typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const XT* __restrict__ x, int64_t ldx, int rows, int H,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        float eps, T* __restrict__ ys, float* __restrict__ yf,
                                                        const int* __restrict__ dyn) {
    rows = ag_dyn_clamp(rows, dyn);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const XT* xr = x + (int64_t)row * ldx;
    float4 v[LN_MAXV];
    float sum = 0.f;
    const int nv = H >> 2;  // 4-element groups per row
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            v[i] = load4_as_f32(xr + c * 4);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + bb * bb) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float4 gv = *reinterpret_cast<const float4*>(g + c * 4);
            const float4 bv = *reinterpret_cast<const float4*>(b + c * 4);
            float4 o;
            o.x = (v[i].x - mean) * rstd * gv.x + bv.x; o.y = (v[i].y - mean) * rstd * gv.y + bv.y;
            o.z = (v[i].z - mean) * rstd * gv.z + bv.z; o.w = (v[i].w - mean) * rstd * gv.w + bv.w;
            if (yf) *reinterpret_cast<float4*>(yf + (int64_t)row * H + c * 4) = o;
            if (ys) {
                T* yp = ys + (int64_t)row * H + c * 4;
                if (sizeof(T) == 2) {
                    *reinterpret_cast<uint2*>(yp) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
                } else {
                    *reinterpret_cast<float4*>(yp) = o;
                }
            }
        }
    }
}

// bf16 -> bf16 rows of 8*k elements, k <= 128: half a wave per row, 16-byte loads and stores (the 8-byte groups of the generic
// kernel leave the memory pipeline with twice the instructions per byte)
__global__ __launch_bounds__(256) void layernorm_bf16x8_kernel(const bf16_t* __restrict__ x, int64_t ldx, int rows, int H,
                                                               const float* __restrict__ g, const float* __restrict__ b,
                                                               float eps, bf16_t* __restrict__ ys, float* __restrict__ yf,
                                                               const int* __restrict__ dyn) {
    rows = ag_dyn_clamp(rows, dyn);
    const int lane = threadIdx.x & 63, sub = lane & 31;
    const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool live = row < rows;
    const bf16_t* xr = x + (int64_t)(live ? row : 0) * ldx;
    const int nc = H >> 3;       // 8-element chunks per row (<= 128)
    uint4 raw[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = sub + i * 32;
        raw[i] = make_uint4(0u, 0u, 0u, 0u);
        if (c < nc) {
            raw[i] = *reinterpret_cast<const uint4*>(xr + c * 8);
            const uint32_t w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xFFFF0000u);
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = sub + i * 32;
        if (c < nc) {
            const uint32_t w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __uint_as_float(w[e] << 16) - mean, bb = __uint_as_float(w[e] & 0xFFFF0000u) - mean;
                sq += a * a + bb * bb;
            }
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq / (float)H + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = sub + i * 32;
        if (c < nc) {
            const uint32_t w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
            const float4 g0 = *reinterpret_cast<const float4*>(g + c * 8), g1 = *reinterpret_cast<const float4*>(g + c * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(b + c * 8), b1 = *reinterpret_cast<const float4*>(b + c * 8 + 4);
            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[2 * e] = (__uint_as_float(w[e] << 16) - mean) * rstd * gv[2 * e] + bv[2 * e];
                o[2 * e + 1] = (__uint_as_float(w[e] & 0xFFFF0000u) - mean) * rstd * gv[2 * e + 1] + bv[2 * e + 1];
            }
            if (yf) {
                *reinterpret_cast<float4*>(yf + (int64_t)row * H + c * 8) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(yf + (int64_t)row * H + c * 8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
            if (ys) *reinterpret_cast<uint4*>(ys + (int64_t)row * H + c * 8) =
                        make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        }
    }
}

// Narrow rows (H <= 128: the 96-wide LTT side network): 8 lanes per row, 8 rows per wave — a 96-element row would leave 40
// of a wave's 64 lanes idle in the kernel above.  Reductions are three xor-shuffles inside the 8-lane group.
template <typename XT, typename T>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const XT* __restrict__ x, int64_t ldx, int rows, int H,
                                                               const float* __restrict__ g, const float* __restrict__ b,
                                                               float eps, T* __restrict__ ys, float* __restrict__ yf,
                                                               const int* __restrict__ dyn) {
    rows = ag_dyn_clamp(rows, dyn);
    const int lane = threadIdx.x & 63, sub = lane & 7;
    const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const bool live = row < rows;
    const XT* xr = x + (int64_t)(live ? row : 0) * ldx;
    const int nv = H >> 2;
    float4 v[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = sub + i * 8;
        if (c < nv) {
            v[i] = load4_as_f32(xr + c * 4);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = sub + i * 8;
        if (c < nv) {
            const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + bb * bb) + (cc * cc + d * d);
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq / (float)H + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = sub + i * 8;
        if (c < nv) {
            const float4 gv = *reinterpret_cast<const float4*>(g + c * 4);
            const float4 bv = *reinterpret_cast<const float4*>(b + c * 4);
            float4 o;
            o.x = (v[i].x - mean) * rstd * gv.x + bv.x; o.y = (v[i].y - mean) * rstd * gv.y + bv.y;
            o.z = (v[i].z - mean) * rstd * gv.z + bv.z; o.w = (v[i].w - mean) * rstd * gv.w + bv.w;
            if (yf) *reinterpret_cast<float4*>(yf + (int64_t)row * H + c * 4) = o;
            if (ys) {
                T* yp = ys + (int64_t)row * H + c * 4;
                if (sizeof(T) == 2) *reinterpret_cast<uint2*>(yp) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
                else *reinterpret_cast<float4*>(yp) = o;
            }
        }
    }
}

// im2col for Conv2d(k = s = patch): cols[(b*gh + py)*gw + px][c*patch*patch + iy*patch + ix]
template <typename T>
__global__ void im2col_kernel(const float* __restrict__ img, int B, int C, int px, int patch, T* __restrict__ cols) {
    const int g = px / patch;
    const int kdim = C * patch * patch;
    const int64_t total = (int64_t)B * g * g * kdim / 4;  // 4 consecutive ix per thread (patch % 4 == 0)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4;
        const int k = (int)(e % kdim);
        const int64_t prow = e / kdim;
        const int pxi = (int)(prow % g), pyi = (int)((prow / g) % g), bi = (int)(prow / ((int64_t)g * g));
        const int ix = k % patch, iy = (k / patch) % patch, c = k / (patch * patch);
        const float4 v = *reinterpret_cast<const float4*>(img + (((int64_t)bi * C + c) * px + (pyi * patch + iy)) * px + pxi * patch + ix);
        T* o = cols + e;
        Store<T>::store(o, v.x); Store<T>::store(o + 1, v.y); Store<T>::store(o + 2, v.z); Store<T>::store(o + 3, v.w);
    }
}

__global__ void vit_assemble_kernel(const float* __restrict__ pe, const float* __restrict__ cls, const float* __restrict__ pos,
                                    int B, int P, int H, float* __restrict__ h0) {
    const int T = P + 1;
    const int64_t total = (int64_t)B * T * H / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4;
        const int hcol = (int)(e % H);
        const int64_t tok = e / H;
        const int t = (int)(tok % T), b = (int)(tok / T);
        const float4 pv = *reinterpret_cast<const float4*>(pos + (int64_t)t * H + hcol);
        float4 v;
        if (t == 0) v = *reinterpret_cast<const float4*>(cls + hcol);
        else v = *reinterpret_cast<const float4*>(pe + ((int64_t)b * P + (t - 1)) * H + hcol);
        v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
        *reinterpret_cast<float4*>(h0 + e) = v;
    }
}

// BERT embeddings: one wave per token: gather word row + type[0] + pos[t], LayerNorm in registers.
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* __restrict__ ids, int B, int Tn, int H,
                                                         const float* __restrict__ word, int vocab,
                                                         const float* __restrict__ type0, const float* __restrict__ pos,
                                                         const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                         float* __restrict__ h0, T* __restrict__ hs) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tok >= B * Tn) return;
    const int t = tok % Tn;
    int64_t id = ids[tok];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* wr = word + id * H;
    float4 v[LN_MAXV];
    float sum = 0.f;
    const int nv = H >> 2;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float4 a = *reinterpret_cast<const float4*>(wr + c * 4);
            const float4 ty = *reinterpret_cast<const float4*>(type0 + c * 4);
            const float4 po = *reinterpret_cast<const float4*>(pos + (int64_t)t * H + c * 4);
            // reference order: (word + type) then += pos  (models/vanilla_bert.py:319-322)
            v[i].x = (a.x + ty.x) + po.x; v[i].y = (a.y + ty.y) + po.y; v[i].z = (a.z + ty.z) + po.z; v[i].w = (a.w + ty.w) + po.w;
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + bb * bb) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float4 gv = *reinterpret_cast<const float4*>(g + c * 4);
            const float4 bv = *reinterpret_cast<const float4*>(b + c * 4);
            float4 o;
            o.x = (v[i].x - mean) * rstd * gv.x + bv.x; o.y = (v[i].y - mean) * rstd * gv.y + bv.y;
            o.z = (v[i].z - mean) * rstd * gv.z + bv.z; o.w = (v[i].w - mean) * rstd * gv.w + bv.w;
            *reinterpret_cast<float4*>(h0 + (int64_t)tok * H + c * 4) = o;
            if (hs) {
                T* yp = hs + (int64_t)tok * H + c * 4;
                if (sizeof(T) == 2) *reinterpret_cast<uint2*>(yp) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
                else *reinterpret_cast<float4*>(yp) = o;
            }
        }
    }
}

// (sum, sumsq) of each row of the bf16 stream — LayerNorm statistics for the folded GEMM epilogue when the
// producing GEMM did not emit them (layer 0's embeddings).  Same slab-major partial-sum layout as the GEMM producer
// (gemm_big.hip): stats[s][row] covers columns [256 s, 256 s + 256); one loop iteration of the wave = one slab.
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ x, int64_t ldx, int rows, int H, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (int64_t)row * ldx;
    const int64_t slab = 2 * (int64_t)rows;
    for (int c0 = 0, s_i = 0; c0 < H; c0 += 256, ++s_i) {
        float s = 0.f, q = 0.f;
        const int c = c0 + lane * 4;
        if (c < H) {
            const float4 v = load4_as_f32(xr + c);
            s = (v.x + v.y) + (v.z + v.w);
            q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        s = wave_sum(s); q = wave_sum(q);
        if (lane == 0) { stats[s_i * slab + 2 * (int64_t)row] = s; stats[s_i * slab + 2 * (int64_t)row + 1] = q; }
    }
}

__global__ void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float m = -3.0e38f;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[(int64_t)row * C + c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(x[(int64_t)row * C + c] - m);
    s = wave_sum(s);
    for (int c = lane; c < C; c += 64) y[(int64_t)row * C + c] = expf(x[(int64_t)row * C + c] - m) / s;
}

// ---- token pruning (BERT additive mask): pack the visible tokens of every row ---------------------------------
// counts[r] = popcount(mask row r) with the CLS bit forced on and bits at or beyond T ignored, so that the packed row count
// cu[R] lies in [R, R*T] by construction (nothing downstream has to validate a device-side count); cu = exclusive scan; one workgroup.
__global__ __launch_bounds__(1024) void seq_scan_kernel(const uint32_t* __restrict__ mask, int R, int Tw, int T, int* __restrict__ cu) {
    __shared__ int part[1024];
    const uint32_t last = (T & 31) ? ((1u << (T & 31)) - 1u) : 0xFFFFFFFFu;   // bits at or beyond T do not count
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (R + nt - 1) / nt;
    const int r0 = tid * per, r1 = min(R, r0 + per);
    int sum = 0;
    for (int r = r0; r < r1; ++r)
        for (int w = 0; w < Tw; ++w) sum += __popc((mask[(long)r * Tw + w] | (w == 0 ? 1u : 0u)) & (w == Tw - 1 ? last : 0xFFFFFFFFu));
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {   // Hillis-Steele inclusive scan of the per-thread sums
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int r = r0; r < r1; ++r) {
        cu[r] = run;
        for (int w = 0; w < Tw; ++w) run += __popc((mask[(long)r * Tw + w] | (w == 0 ? 1u : 0u)) & (w == Tw - 1 ? last : 0xFFFFFFFFu));
    }
    if (tid == nt - 1) cu[R] = part[nt - 1];
}
// tok_src[cu[r] + rank(t)] = r*T + t for every visible token t of row r; one wave per row
__global__ __launch_bounds__(256) void seq_index_kernel(const uint32_t* __restrict__ mask, int R, int T, int Tw,
                                                        const int* __restrict__ cu, int* __restrict__ tok_src) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    int base = cu[row];
    for (int w0 = 0; w0 < Tw; w0 += 2) {           // 64 tokens per step: lane = token
        const int t = w0 * 32 + lane;
        uint32_t word = (w0 + (lane >> 5)) < Tw ? mask[(long)row * Tw + w0 + (lane >> 5)] : 0u;
        if (w0 + (lane >> 5) == 0) word |= 1u;      // CLS is always visible (as in the scan): cu[R] is in [R, R*T] whatever the caller packed
        const bool on = t < T && ((word >> (lane & 31)) & 1u);
        const unsigned long long ball = __ballot(on);
        if (on) tok_src[base + __popcll(ball & ((1ull << lane) - 1ull))] = row * T + t;
        base += __popcll(ball);
    }
}
// dst[i, :] = src[idx[i], :]  (rows of H elements of `es` bytes, 16-byte vectors)
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, long ld_src_b, const int* __restrict__ idx,
                                                          char* __restrict__ dst, long ld_dst_b, int n, int row_bytes,
                                                          const int* __restrict__ dyn) {
    n = ag_dyn_clamp(n, dyn);
    const int vec = row_bytes >> 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n * vec; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / vec), v = (int)(i % vec);
        *reinterpret_cast<uint4*>(dst + (long)r * ld_dst_b + v * 16) =
            *reinterpret_cast<const uint4*>(src + (long)idx[r] * ld_src_b + v * 16);
    }
}

}  // namespace

// Linear(LayerNorm(x)) folded for the GEMM epilogue (PackedFoldedLinear, engine.py): one wave per output row n:
//   w'[n,k] = store(w[n,k] * gamma[k]),   b'[n] = b[n] + sum_k w[n,k] * beta[k],   s[n] = sum_k float(w'[n,k])  (of the ROUNDED w')
template <typename T>
__global__ __launch_bounds__(256) void fold_ln_pack_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int N, int K, T* __restrict__ w_out,
                                                           float* __restrict__ b_out, float* __restrict__ s_out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const float* wr = w + (long)n * K;
    float dot = 0.f, sum = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float x = wr[k];
        dot = fmaf(x, beta[k], dot);
        const float y = x * gamma[k];
        Store<T>::store(w_out + (long)n * K + k, y);
        sum += Store<T>::load(w_out + (long)n * K + k);
    }
    dot = wave_sum(dot); sum = wave_sum(sum);
    if (lane == 0) { b_out[n] = (b ? b[n] : 0.f) + dot; s_out[n] = sum; }
}

extern "C" int ag_pack_folded_linear(const float* d_w, const float* d_b, const float* d_gamma, const float* d_beta, int N, int K,
                                     void* d_w_out, int dtype, float* d_b_out, float* d_s_out, void* stream) {
    AG_REQUIRE(d_w && d_gamma && d_beta && d_w_out && d_b_out && d_s_out && N > 0 && K > 0, "ag_pack_folded_linear: bad arguments");
    AG_REQUIRE(dtype == AG_BF16 || dtype == AG_F32, "ag_pack_folded_linear: bad dtype %d", dtype);
    const dim3 grid(ceil_div(N, 4)), block(256);
    if (dtype == AG_BF16)
        hipLaunchKernelGGL(fold_ln_pack_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, d_w, d_b, d_gamma, d_beta, N, K, (bf16_t*)d_w_out, d_b_out, d_s_out);
    else
        hipLaunchKernelGGL(fold_ln_pack_kernel<float>, grid, block, 0, (hipStream_t)stream, d_w, d_b, d_gamma, d_beta, N, K, (float*)d_w_out, d_b_out, d_s_out);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_cast_f32(const float* d_src, void* d_dst, int64_t n, int dtype, void* stream) {
    AG_REQUIRE(d_src && d_dst && n >= 0, "ag_cast_f32: bad arguments");
    if (n == 0) return AG_OK;
    const int blocks = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 + 1 : 2048);
    if (dtype == AG_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_src, (bf16_t*)d_dst, n);
    else if (dtype == AG_F32) hipLaunchKernelGGL(cast_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_src, (float*)d_dst, n);
    else return ag_fail(AG_ERR_INVALID, "ag_cast_f32: bad dtype %d", dtype);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_layernorm(const void* d_x, int x_dtype, int64_t ldx, int rows, int H, const float* d_gamma, const float* d_beta,
                            float eps, void* d_y_store, float* d_y_f32, int dtype, const int* d_rows, void* stream) {
    if (rows == 0) return AG_OK;
    AG_REQUIRE(d_x && d_gamma && d_beta && (d_y_store || d_y_f32), "ag_layernorm: null pointer");
    AG_REQUIRE(H % 4 == 0 && H <= 64 * 4 * LN_MAXV && ldx % 4 == 0, "ag_layernorm: H=%d unsupported (multiple of 4, <= %d)", H, 64 * 4 * LN_MAXV);
    if (rows == 0) return AG_OK;
    const int blocks = ceil_div(rows, 4);
    hipStream_t s = (hipStream_t)stream;
    AG_REQUIRE(x_dtype == AG_F32 || x_dtype == AG_BF16, "ag_layernorm: bad x_dtype %d", x_dtype);
    const int* dyn = d_rows;
    AgProfScope prof(AG_PROF_LAYERNORM, 0.0, (double)rows * H * ((double)dtype_size(x_dtype) + (d_y_store ? (double)dtype_size(dtype) : 0.0) + (d_y_f32 ? 4.0 : 0.0)), s,
                     dyn, (double)rows);
    if (dtype != AG_BF16 && dtype != AG_F32) return ag_fail(AG_ERR_INVALID, "ag_layernorm: bad dtype %d", dtype);
    const bool narrow = H <= 128;                      // 8 lanes per row, 32 rows per block
    const int nblk = narrow ? ceil_div(rows, 32) : blocks;
#define AG_LN_LAUNCH(XT_, T_, X_, Y_)                                                                                              \
    do {                                                                                                                           \
        if (narrow) hipLaunchKernelGGL((layernorm_narrow_kernel<XT_, T_>), dim3(nblk), dim3(256), 0, s, X_, ldx, rows, H, d_gamma, d_beta, eps, Y_, d_y_f32, dyn); \
        else hipLaunchKernelGGL((layernorm_kernel<XT_, T_>), dim3(nblk), dim3(256), 0, s, X_, ldx, rows, H, d_gamma, d_beta, eps, Y_, d_y_f32, dyn);          \
    } while (0)
    if (x_dtype == AG_F32) {
        const float* x = (const float*)d_x;
        if (dtype == AG_BF16) AG_LN_LAUNCH(float, bf16_t, x, (bf16_t*)d_y_store);
        else AG_LN_LAUNCH(float, float, x, (float*)d_y_store);
    } else {
        const bf16_t* x = (const bf16_t*)d_x;
        static const bool wide_off = getenv("AG_LN_WIDE") && atoi(getenv("AG_LN_WIDE")) == 0;
        if (dtype == AG_BF16 && !narrow && !wide_off && H % 8 == 0 && H <= 1024 && ldx % 8 == 0)
            hipLaunchKernelGGL(layernorm_bf16x8_kernel, dim3(ceil_div(rows, 8)), dim3(256), 0, s, x, ldx, rows, H, d_gamma, d_beta, eps, (bf16_t*)d_y_store, d_y_f32, dyn);
        else if (dtype == AG_BF16) AG_LN_LAUNCH(bf16_t, bf16_t, x, (bf16_t*)d_y_store);
        else AG_LN_LAUNCH(bf16_t, float, x, (float*)d_y_store);
    }
#undef AG_LN_LAUNCH
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_vit_im2col(const float* d_img, int B, int Cin, int px, int patch, void* d_cols, int dtype, void* stream) {
    AG_REQUIRE(d_img && d_cols, "ag_vit_im2col: null pointer");
    AG_REQUIRE(patch > 0 && px % patch == 0 && patch % 4 == 0 && px % 4 == 0, "ag_vit_im2col: px=%d patch=%d unsupported", px, patch);
    if (B == 0) return AG_OK;
    const int64_t total = (int64_t)B * px * px * Cin / 4;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == AG_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, d_img, B, Cin, px, patch, (bf16_t*)d_cols);
    else if (dtype == AG_F32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, s, d_img, B, Cin, px, patch, (float*)d_cols);
    else return ag_fail(AG_ERR_INVALID, "ag_vit_im2col: bad dtype %d", dtype);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_vit_assemble(const float* d_patch_emb, const float* d_cls, const float* d_pos, int B, int P, int H,
                               float* d_h0, void* stream) {
    AG_REQUIRE(d_patch_emb && d_cls && d_pos && d_h0 && H % 4 == 0, "ag_vit_assemble: bad arguments");
    if (B == 0) return AG_OK;
    const int64_t total = (int64_t)B * (P + 1) * H / 4;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_patch_emb, d_cls, d_pos, B, P, H, d_h0);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_bert_embed(const int64_t* d_ids, int B, int T, int H, const float* d_word, int vocab, const float* d_type0,
                             const float* d_pos, const float* d_gamma, const float* d_beta, float eps,
                             float* d_h0, void* d_h0_store, int dtype, void* stream) {
    AG_REQUIRE(d_ids && d_word && d_type0 && d_pos && d_gamma && d_beta && d_h0, "ag_bert_embed: null pointer");
    AG_REQUIRE(H % 4 == 0 && H <= 64 * 4 * LN_MAXV, "ag_bert_embed: H=%d unsupported", H);
    if (B == 0) return AG_OK;
    const int blocks = ceil_div((int64_t)B * T, 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == AG_BF16) hipLaunchKernelGGL(bert_embed_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, d_ids, B, T, H, d_word, vocab, d_type0, d_pos, d_gamma, d_beta, eps, d_h0, (bf16_t*)d_h0_store);
    else if (dtype == AG_F32) hipLaunchKernelGGL(bert_embed_kernel<float>, dim3(blocks), dim3(256), 0, s, d_ids, B, T, H, d_word, vocab, d_type0, d_pos, d_gamma, d_beta, eps, d_h0, (float*)d_h0_store);
    else return ag_fail(AG_ERR_INVALID, "ag_bert_embed: bad dtype %d", dtype);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_softmax_rows(const float* d_x, float* d_y, int rows, int C, void* stream) {
    AG_REQUIRE(d_x && d_y && rows >= 0 && C >= 1, "ag_softmax_rows: bad arguments");
    if (rows == 0) return AG_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, d_x, d_y, rows, C);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_row_stats_bf16(const void* d_x, int64_t ldx, int rows, int H, float* d_stats, void* stream) {
    AG_REQUIRE(d_x && d_stats && rows >= 0 && H % 4 == 0 && ldx % 4 == 0, "ag_row_stats_bf16: bad arguments");
    if (rows == 0) return AG_OK;
    hipLaunchKernelGGL(row_stats_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_x, ldx, rows, H, d_stats);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_seq_compact_plan(const uint32_t* d_mask_bits, int R, int T, int* d_cu_seqlens, int* d_tok_src, void* stream) {
    AG_REQUIRE(d_mask_bits && d_cu_seqlens && d_tok_src && R >= 1 && T >= 1, "ag_seq_compact_plan: bad arguments");
    const int Tw = (T + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(seq_scan_kernel, dim3(1), dim3(1024), 0, s, d_mask_bits, R, Tw, T, d_cu_seqlens);
    AG_LAUNCH_CHECK();
    hipLaunchKernelGGL(seq_index_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, d_mask_bits, R, T, Tw, d_cu_seqlens, d_tok_src);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_gather_rows(const void* d_src, int64_t ld_src, const int* d_index, void* d_dst, int64_t ld_dst, int n, int H,
                              int dtype, const int* d_rows, void* stream) {
    AG_REQUIRE(d_src && d_index && d_dst && n >= 0 && H >= 1, "ag_gather_rows: bad arguments");
    AG_REQUIRE(dtype == AG_BF16 || dtype == AG_F32, "ag_gather_rows: bad dtype %d", dtype);
    const size_t es = dtype_size(dtype);
    AG_REQUIRE((H * es) % 16 == 0 && (ld_src * es) % 16 == 0 && (ld_dst * es) % 16 == 0, "ag_gather_rows: rows must be 16-byte multiples");
    if (n == 0) return AG_OK;
    const long total = (long)n * ((H * es) >> 4);
    const int grid = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)d_src, (long)(ld_src * es), d_index,
                       (char*)d_dst, (long)(ld_dst * es), n, (int)(H * es), d_rows);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
