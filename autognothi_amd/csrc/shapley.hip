// shapley.hip — Shapley-value reductions: efficiency normalisation, the FastSHAP-style regression
// loss and its gradient, and the surrogate's KL loss.  All are tiny, HBM/latency-bound reductions;
// token/player-axis sums are wavefront shuffle reductions, results are deterministic (no atomics).
//
// reference models/shapley.py:82-93 (normalize_shapley_explanation), :9-53 (loss_shapley_new),
// :96-106 (loss_logits_kl_divergence); models/vanilla_vit.py:125-129 (normalise, drop CLS, permute).
#include "common.h"

namespace {

// one block per batch item: stage pred[b] ([T,C]) in LDS, per class: token-sum by wave shuffles,
// then write phi[b,c,:] coalesced along players.
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ pred, const float* __restrict__ grand,
                                                        const float* __restrict__ null, int T, int C, int normalize,
                                                        float* __restrict__ phi) {
    extern __shared__ float sp[];  // [T*C] + [C] corrections
    float* corr = sp + T * C;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float* pb = pred + (int64_t)b * T * C;
    for (int i = tid; i < T * C; i += blockDim.x) sp[i] = pb[i];
    __syncthreads();
    for (int c = wave; c < C; c += nw) {
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += sp[t * C + c];
        s = wave_sum(s);
        if (lane == 0) corr[c] = normalize ? ((grand[(int64_t)b * C + c] - null[c]) - s) / (float)T : 0.f;
    }
    __syncthreads();
    const int P = T - 1;
    float* ob = phi + (int64_t)b * C * P;
    for (int i = tid; i < C * P; i += blockDim.x) {
        const int c = i / P, p = i - c * P;
        ob[i] = sp[(p + 1) * C + c] + corr[c];
    }
}

__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float* __restrict__ dphi, int T, int C, int normalize,
                                                            float* __restrict__ dpred) {
    extern __shared__ float sc[];  // [C]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int P = T - 1;
    const float* gb = dphi + (int64_t)b * C * P;
    for (int c = wave; c < C; c += nw) {
        float s = 0.f;
        for (int p = lane; p < P; p += 64) s += gb[c * P + p];
        s = wave_sum(s);
        if (lane == 0) sc[c] = normalize ? s / (float)T : 0.f;
    }
    __syncthreads();
    float* ob = dpred + (int64_t)b * T * C;
    for (int i = tid; i < T * C; i += blockDim.x) {
        const int t = i / C, c = i - t * C;
        ob[i] = (t > 0 ? gb[c * P + (t - 1)] : 0.f) - sc[c];
    }
}

// the reference function as it stands (models/shapley.py:82-93): out[b,t,c] = pred[b,t,c] + ((grand[b,c] - null[c]) - sum_t pred[b,t,c]) / T
// over ALL T rows (no CLS drop, no permute); bwd != 0 computes its adjoint: dpred = dout - sum_t dout / T.
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ pred, const float* __restrict__ grand,
                                                             const float* __restrict__ null, int T, int C, int bwd,
                                                             float* __restrict__ out) {
    extern __shared__ float sc[];  // [C]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float* pb = pred + (int64_t)b * T * C;
    for (int c = wave; c < C; c += nw) {
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += pb[t * C + c];
        s = wave_sum(s);
        if (lane == 0) sc[c] = bwd ? -s / (float)T : ((grand[(int64_t)b * C + c] - null[c]) - s) / (float)T;
    }
    __syncthreads();
    float* ob = out + (int64_t)b * T * C;
    for (int i = tid; i < T * C; i += blockDim.x) ob[i] = pb[i] + sc[i % C];
}

// diff[b,k,c] = v0[c] + sum_p bit(b,k,p+1) * phi[b,c,p] - v_s[b*K+k, c]; one wave per (b,k,c)
__global__ __launch_bounds__(256) void loss_diff_kernel(const uint32_t* __restrict__ bits, const float* __restrict__ v0,
                                                        const float* __restrict__ vs, const float* __restrict__ phi,
                                                        int B, int K, int P, int C, int Tw, float* __restrict__ diff) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= (int64_t)B * K * C) return;
    const int c = (int)(w % C);
    const int64_t bk = w / C;
    const int b = (int)(bk / K);
    const uint32_t* mb = bits + bk * Tw;
    const float* ph = phi + ((int64_t)b * C + c) * P;
    float s = 0.f;
    for (int p = lane; p < P; p += 64) {
        const int t = p + 1;
        if ((mb[t >> 5] >> (t & 31)) & 1u) s += ph[p];
    }
    s = wave_sum(s);
    if (lane == 0) diff[w] = (v0[c] + s) - vs[bk * C + c];
}

// single block: loss = P * mean(diff^2), fixed summation order
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ diff, int64_t n, int P, float* __restrict__ loss) {
    __shared__ float part[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += diff[i] * diff[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
        *loss = (float)P * (t / (float)n);
    }
}

// dphi[b,c,p] = (2P/(BKC)) * sum_k bit(b,k,p+1) * diff[b,k,c]
__global__ void loss_grad_kernel(const uint32_t* __restrict__ bits, const float* __restrict__ diff, int B, int K, int P, int C,
                                 int Tw, float* __restrict__ dphi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C * P) return;
    const int p = (int)(i % P);
    const int c = (int)((i / P) % C);
    const int b = (int)(i / ((int64_t)P * C));
    const int t = p + 1;
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const int64_t bk = (int64_t)b * K + k;
        if ((bits[bk * Tw + (t >> 5)] >> (t & 31)) & 1u) s += diff[bk * C + c];
    }
    dphi[i] = s * (2.0f * (float)P / (float)((int64_t)B * K * C));
}

// kl_div(log_softmax(ref), softmax(cur), batchmean); d/dcur: t_j * ((log t_j - ls_j) - sum_i t_i (log t_i - ls_i)) / B
__global__ __launch_bounds__(256) void kl_kernel(const float* __restrict__ ref, const float* __restrict__ cur, int B, int C,
                                                 float* __restrict__ loss, float* __restrict__ dcur) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float acc = 0.f;
    for (int b = wave; b < B; b += nw) {
        const float* r = ref + (int64_t)b * C;
        const float* q = cur + (int64_t)b * C;
        float mr = -3.0e38f, mq = -3.0e38f;
        for (int c = lane; c < C; c += 64) { mr = fmaxf(mr, r[c]); mq = fmaxf(mq, q[c]); }
        mr = wave_max(mr); mq = wave_max(mq);
        float sr = 0.f, sq = 0.f;
        for (int c = lane; c < C; c += 64) { sr += expf(r[c] - mr); sq += expf(q[c] - mq); }
        const float lr = logf(wave_sum(sr)), lq = logf(wave_sum(sq));
        float row = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float lt = (q[c] - mq) - lq, ls = (r[c] - mr) - lr;
            row += expf(lt) * (lt - ls);
        }
        row = wave_sum(row);
        if (dcur) for (int c = lane; c < C; c += 64) {
            const float lt = (q[c] - mq) - lq, ls = (r[c] - mr) - lr;
            dcur[(int64_t)b * C + c] = expf(lt) * ((lt - ls) - row) / (float)B;
        }
        acc += row;
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < nw; ++i) t += part[i];
        *loss = t / (float)B;
    }
}

// ---- Monte-Carlo permutation Shapley reduction (reference scripts/preview_text_shapley.py:112-132, :135-153) ----
// v [reps, P+1, C]: surrogate outputs along each permutation's nested-mask chain (row i = the first i players of the
// permutation visible).  value f = log(p / (1 - p + 1e-6)) with p = softmax over classes of the (already soft-maxed)
// outputs — the reference's "sharpening", quirk included.  Player q sits at position rank[r][q] of permutation r, so
// its marginal contribution there is f(v[r, rank+1]) - f(v[r, rank]); sv[c, q] = mean over r.  Thread per (q, c):
// no atomics, deterministic.
__device__ __forceinline__ float mc_value(const float* __restrict__ row, int C, int c) {
    float m = -3.0e38f;
    for (int k = 0; k < C; ++k) m = fmaxf(m, row[k]);
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += expf(row[k] - m);
    const float p = expf(row[c] - m) / s;
    return logf(p / (1.0f - p + 1e-6f));
}
__global__ void mc_shapley_kernel(const float* __restrict__ v, const int* __restrict__ rank, int reps, int P, int C,
                                  float* __restrict__ sv, float* __restrict__ v0, float* __restrict__ vn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) {   // the cached end points: of the LAST permutation, as the reference returns them (:128-129)
        const float* last = v + (long)(reps - 1) * (P + 1) * C;
        v0[i] = mc_value(last, C, i);
        vn[i] = mc_value(last + (long)P * C, C, i);
    }
    if (i >= P * C) return;
    const int q = i / C, c = i % C;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        const int j = rank[(long)r * P + q];
        const float* base = v + ((long)r * (P + 1) + j) * C;
        acc += mc_value(base + C, C, c) - mc_value(base, C, c);
    }
    sv[(long)c * P + q] = acc / (float)reps;
}

}  // namespace

extern "C" int ag_shapley_normalize(const float* d_pred, const float* d_grand, const float* d_null, int B, int T, int C,
                                    int normalize, float* d_phi, void* stream) {
    AG_REQUIRE(d_pred && d_phi && (!normalize || (d_grand && d_null)), "ag_shapley_normalize: null pointer");
    AG_REQUIRE(T >= 2 && C >= 1 && (size_t)(T * C + C) * 4 <= 64 * 1024, "ag_shapley_normalize: T*C=%d too large for the LDS image", T * C);
    if (B == 0) return AG_OK;
    hipLaunchKernelGGL(normalize_kernel, dim3(B), dim3(256), (size_t)(T * C + C) * 4, (hipStream_t)stream, d_pred, d_grand, d_null, T, C, normalize, d_phi);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_shapley_normalize_bwd(const float* d_dphi, int B, int T, int C, int normalize, float* d_dpred, void* stream) {
    AG_REQUIRE(d_dphi && d_dpred && T >= 2 && C >= 1 && C <= 4096, "ag_shapley_normalize_bwd: bad arguments");
    if (B == 0) return AG_OK;
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(B), dim3(256), (size_t)C * 4, (hipStream_t)stream, d_dphi, T, C, normalize, d_dpred);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_shapley_normalize_rows(const float* d_pred, const float* d_grand, const float* d_null, int B, int T, int C,
                                         float* d_out, void* stream) {
    AG_REQUIRE(d_pred && d_out && d_grand && d_null && T >= 1 && C >= 1 && C <= 4096, "ag_shapley_normalize_rows: bad arguments");
    if (B == 0) return AG_OK;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(B), dim3(256), (size_t)C * 4, (hipStream_t)stream, d_pred, d_grand, d_null, T, C, 0, d_out);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_shapley_normalize_rows_bwd(const float* d_dout, int B, int T, int C, float* d_dpred, void* stream) {
    AG_REQUIRE(d_dout && d_dpred && T >= 1 && C >= 1 && C <= 4096, "ag_shapley_normalize_rows_bwd: bad arguments");
    if (B == 0) return AG_OK;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(B), dim3(256), (size_t)C * 4, (hipStream_t)stream, d_dout, (const float*)nullptr,
                       (const float*)nullptr, T, C, 1, d_dpred);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_shapley_loss(const uint32_t* d_mask_bits, const float* d_v0, const float* d_vs, const float* d_phi,
                               int B, int K, int P, int C, float* d_loss, float* d_dphi, float* d_scratch, void* stream) {
    AG_REQUIRE(d_mask_bits && d_v0 && d_vs && d_phi && d_loss && d_scratch, "ag_shapley_loss: null pointer");
    AG_REQUIRE(B >= 1 && K >= 1 && P >= 1 && C >= 1, "ag_shapley_loss: bad shape");
    const int Tw = (P + 1 + 31) / 32;
    const int64_t n = (int64_t)B * K * C;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_diff_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, d_mask_bits, d_v0, d_vs, d_phi, B, K, P, C, Tw, d_scratch);
    AG_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, s, d_scratch, n, P, d_loss);
    AG_LAUNCH_CHECK();
    if (d_dphi) {
        const int64_t m = (int64_t)B * C * P;
        hipLaunchKernelGGL(loss_grad_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, s, d_mask_bits, d_scratch, B, K, P, C, Tw, d_dphi);
        AG_LAUNCH_CHECK();
    }
    return AG_OK;
}

extern "C" int ag_kl_loss(const float* d_ref, const float* d_cur, int B, int C, float* d_loss, float* d_dcur, void* stream) {
    AG_REQUIRE(d_ref && d_cur && d_loss && B >= 1 && C >= 1, "ag_kl_loss: bad arguments");
    hipLaunchKernelGGL(kl_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_ref, d_cur, B, C, d_loss, d_dcur);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_mc_shapley_reduce(const float* d_v, const int* d_rank, int reps, int P, int C, float* d_sv, float* d_v0,
                                    float* d_vn, void* stream) {
    AG_REQUIRE(d_v && d_rank && d_sv && d_v0 && d_vn && reps >= 1 && P >= 1 && C >= 1, "ag_mc_shapley_reduce: bad arguments");
    const int n = P * C > C ? P * C : C;
    hipLaunchKernelGGL(mc_shapley_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, d_v, d_rank, reps, P, C, d_sv, d_v0, d_vn);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
