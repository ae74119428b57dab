// sampler.hip — device-resident MT19937 + the reference's mask samplers (bit-exact).
//
// reference models/shapley.py:56-79 (mask_shapley_new), :109-115 (mask_purely_uniform),
// :131-135 (_torch_choice); recipes/vanilla_vit.py:219-224 (_fw_xs_preprocess, CLS column);
// scripts/measure_faithfulness.py:225-251 (_get_perturbed_samples).
//
// The reference draws its uniforms from torch's *CPU* generator (at::mt19937) and then copies the
// int64 masks host->device every batch (scripts/train_explainer.py:153-155).  Here the generator
// state lives in HBM and the whole sampler runs on the stream: one workgroup regenerates the MT
// state block by block — the recurrence mt[i] = mt[i+397] ^ f(mt[i], mt[i+1]) only looks 397 words
// back, so each 624-word twist is three data-parallel phases ([0,227), [227,454), [454,624)) — and
// the same workgroup turns the tempered words into mask bits.  The 32-bit stream, the
// (x & 0xFFFFFF) * 2^-24 float mapping and the draw order (U1[h,P] first, then u2[h]) are exactly
// torch's, so masks are bit-identical to the reference for the same seed / generator state.
#include "common.h"
#include <string.h>

namespace {

constexpr int MT_N = 624, MT_M = 397;
struct MtState { uint32_t mt[MT_N]; int pos; int pad[15]; };
static_assert(sizeof(MtState) == AG_MT_STATE_BYTES, "AG_MT_STATE_BYTES mismatch");

__device__ __forceinline__ uint32_t mt_mix(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7FFFFFFFu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
    return y;
}
// whole-block twist of the LDS-resident state (blockDim.x >= 227)
__device__ void mt_twist(uint32_t* mt) {
    const int t = threadIdx.x;
    uint32_t v = 0;
    if (t < MT_N - MT_M) v = mt_mix(mt[t], mt[t + 1], mt[t + MT_M]);
    __syncthreads();
    if (t < MT_N - MT_M) mt[t] = v;
    __syncthreads();
    const int i2 = t + (MT_N - MT_M);
    if (t < MT_N - MT_M) v = mt_mix(mt[i2], mt[i2 + 1], mt[i2 - (MT_N - MT_M)]);
    __syncthreads();
    if (t < MT_N - MT_M) mt[i2] = v;
    __syncthreads();
    const int i3 = t + 2 * (MT_N - MT_M);
    if (i3 < MT_N - 1) v = mt_mix(mt[i3], mt[i3 + 1], mt[i3 - (MT_N - MT_M)]);
    __syncthreads();
    if (i3 < MT_N - 1) mt[i3] = v;
    __syncthreads();
    if (t == 0) mt[MT_N - 1] = mt_mix(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
    __syncthreads();
}

// Cooperative stream reader: draw(j) for j = 0..count-1 handed to `sink(j, tempered_word)` by the
// thread that owns word j; all threads of the block must call this together.
template <typename Sink>
__device__ void mt_stream(uint32_t* mt, int& pos, int64_t count, Sink sink) {
    int64_t done = 0;
    while (done < count) {
        if (pos >= MT_N) { mt_twist(mt); pos = 0; }
        const int take = (int)min((int64_t)(MT_N - pos), count - done);
        for (int i = threadIdx.x; i < take; i += blockDim.x) sink(done + i, mt_temper(mt[pos + i]));
        pos += take;
        done += take;
    }
}

// step over n draws: twists only
__device__ void mt_skip(uint32_t* mt, int& pos, int64_t n) {
    while (n > 0) {
        if (pos >= MT_N) { mt_twist(mt); pos = 0; }
        const int64_t take = min((int64_t)(MT_N - pos), n);
        pos += (int)take;
        n -= take;
    }
}

__global__ void mt_seed_kernel(MtState* st, uint32_t seed) {
    if (threadIdx.x == 0) {
        uint32_t x = seed;
        st->mt[0] = x;
        for (int i = 1; i < MT_N; ++i) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i; st->mt[i] = x; }
        st->pos = MT_N;  // torch: first draw after manual_seed twists
    }
}

__device__ void load_state(const MtState* st, uint32_t* mt, int& pos) {
    for (int i = threadIdx.x; i < MT_N; i += blockDim.x) mt[i] = st->mt[i];
    pos = st->pos;
    __syncthreads();
}
__device__ void store_state(MtState* st, const uint32_t* mt, int pos) {
    __syncthreads();
    for (int i = threadIdx.x; i < MT_N; i += blockDim.x) st->mt[i] = mt[i];
    if (threadIdx.x == 0) st->pos = pos;
}

__global__ __launch_bounds__(256) void mt_raw_kernel(MtState* st, uint32_t* out, int64_t n) {
    __shared__ uint32_t mt[MT_N];
    int pos;
    load_state(st, mt, pos);
    mt_stream(mt, pos, n, [&](int64_t j, uint32_t w) { out[j] = w; });
    store_state(st, mt, pos);
}

// fast-forward by n draws: only the twists, nothing tempered or stored (row-sharded ranks step over the draws that belong to
// other ranks so that every rank stays on the ONE stream an unsharded run would consume)
__global__ __launch_bounds__(256) void mt_skip_kernel(MtState* st, int64_t n) {
    __shared__ uint32_t mt[MT_N];
    int pos;
    load_state(st, mt, pos);
    mt_skip(mt, pos, n);
    store_state(st, mt, pos);
}

__device__ __forceinline__ float u24(uint32_t w) { return (float)(w & 0xFFFFFFu) * 5.9604644775390625e-08f; }

// Phase 1 of both samplers: one workgroup streams the h*P + h tempered words into the caller's
// scratch (U1 row-major first, then u2) and writes the advanced state back.
struct SamplerArgs {
    MtState* st;
    int h, P, paired;
    float inv_p;           // fp32(1/P) rounded from the double quotient, as torch casts the python scalar
    const float* prefix;   // [P-1] (shapley) or nullptr (purely uniform)
    int64_t* mask_i64;     // [n, P] or nullptr
    uint32_t* mask_bits;   // [n, Tw] or nullptr
    uint32_t* scratch;     // [h*P + h] u32
    int Tw;
    // row-sharded call (ag_mask_shapley_new_rows): this rank's h rows are rows [h_lo, h_lo + h) of a call with h_total sampled
    // rows; the draws of the other rows are stepped over so that the state advances by the FULL call (0 / h: unsharded)
    int h_lo, h_total;
};

__global__ __launch_bounds__(256) void sampler_draw_kernel(SamplerArgs a) {
    __shared__ uint32_t mt[MT_N];
    int pos;
    load_state(a.st, mt, pos);
    uint32_t* sc = a.scratch;
    const int64_t own_u1 = (int64_t)a.h * a.P, after = a.h_total - a.h_lo - a.h;
    // the reference draws U1[h_total, P] row-major first, then u2[h_total] (models/shapley.py:69,:133)
    mt_skip(mt, pos, (int64_t)a.h_lo * a.P);
    mt_stream(mt, pos, own_u1, [&](int64_t j, uint32_t w) { sc[j] = w; });
    mt_skip(mt, pos, after * a.P + a.h_lo);
    mt_stream(mt, pos, (int64_t)a.h, [&](int64_t j, uint32_t w) { sc[own_u1 + j] = w; });
    mt_skip(mt, pos, after);
    store_state(a.st, mt, pos);
}

// Phase 2: one wave per sampled row i: threshold, compare, emit pair (2i, 2i+1) / single row.
__global__ __launch_bounds__(256) void sampler_emit_kernel(SamplerArgs a) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= a.h) return;
    const uint32_t* u1 = a.scratch + (int64_t)i * a.P;
    const float u2 = u24(a.scratch[(int64_t)a.h * a.P + i]);
    float thr;
    if (a.prefix) {
        // _torch_choice: position = max(count(u2 >= prefix[j]) - 1, 0); thr = fp32(1/P) * fp32(position)
        int cnt = 0;
        for (int j = lane; j < a.P - 1; j += 64) cnt += (u2 >= a.prefix[j]) ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        const int position = max(cnt - 1, 0);
        thr = a.inv_p * (float)position;
    } else {
        thr = u2;  // mask_purely_uniform: rand(B,P) > rand(B,1)
    }
    const int rows = a.paired ? 2 : 1;
    const int r0 = i * rows;
    const int T = a.P + 1;
    for (int w = 0; w < a.Tw; ++w) {
        const int t = w * 32 + (lane & 31);       // token index (0 = CLS)
        bool on = false;
        if (t >= 1 && t < T) on = u24(u1[t - 1]) > thr;
        if (lane < 32) {
            if (a.mask_i64 && t >= 1 && t < T) {
                a.mask_i64[(int64_t)r0 * a.P + (t - 1)] = on ? 1 : 0;
                if (a.paired) a.mask_i64[(int64_t)(r0 + 1) * a.P + (t - 1)] = on ? 0 : 1;
            }
        }
        const bool valid = (t >= 1 && t < T);
        const unsigned long long b_on = __ballot(lane < 32 && (t == 0 || (valid && on)));
        const unsigned long long b_off = __ballot(lane < 32 && (t == 0 || (valid && !on)));
        if (lane == 0 && a.mask_bits) {
            a.mask_bits[(int64_t)r0 * a.Tw + w] = (uint32_t)b_on;
            if (a.paired) a.mask_bits[(int64_t)(r0 + 1) * a.Tw + w] = (uint32_t)b_off;
        }
    }
}

__global__ void pack_mask_kernel(const int64_t* m, int rows, int P, uint32_t* bits, int Tw) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    for (int w = 0; w < Tw; ++w) {
        const int t = w * 32 + (lane & 31);
        bool on = false;
        if (lane < 32) {
            if (t == 0) on = true;
            else if (t <= P) on = m[(int64_t)r * P + (t - 1)] != 0;
        }
        const unsigned long long b = __ballot(on);
        if (lane == 0) bits[(int64_t)r * Tw + w] = (uint32_t)b;
    }
}

// rank[p] = #{q : attr[q] > attr[p]} + #{q > p : attr[q] == attr[p]}  (descending, ties: higher index first)
__global__ void perturbed_kernel(const float* attr, int P, int steps, int base, int64_t* stops, int64_t* masks) {
    extern __shared__ float sa[];
    const int a = blockIdx.x;
    const float* av = attr + (int64_t)a * P;
    for (int p = threadIdx.x; p < P; p += blockDim.x) sa[p] = av[p];
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        const float v = sa[p];
        int rank = 0;
        for (int q = 0; q < P; ++q) rank += (sa[q] > v || (sa[q] == v && q > p)) ? 1 : 0;
        for (int s = 0; s < steps; ++s) {
            // np.linspace(0, P, steps, dtype=int64): start + s*step in float64, truncated
            long stop;
            if (steps == 1) stop = 0;
            else if (s == steps - 1) stop = P;
            else stop = (long)((double)s * ((double)P / (double)(steps - 1)));
            if (a == 0 && p == 0) stops[s] = stop;
            masks[((int64_t)a * steps + s) * P + p] = (rank < stop) ? (base ^ 1) : base;
        }
    }
}

}  // namespace

extern "C" int ag_mt19937_seed(void* d_state, uint32_t seed, void* stream) {
    AG_REQUIRE(d_state, "ag_mt19937_seed: null state");
    hipLaunchKernelGGL(mt_seed_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (MtState*)d_state, seed);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_mt19937_import(void* d_state, const uint32_t* h_mt624, int pos, void* stream) {
    AG_REQUIRE(d_state && h_mt624 && pos >= 0 && pos <= MT_N, "ag_mt19937_import: bad arguments");
    MtState tmp;
    memset(&tmp, 0, sizeof(tmp));
    memcpy(tmp.mt, h_mt624, sizeof(tmp.mt));
    tmp.pos = pos;
    AG_HIP_CHECK(hipMemcpyAsync(d_state, &tmp, sizeof(tmp), hipMemcpyHostToDevice, (hipStream_t)stream));
    AG_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));  // tmp is a stack object
    return AG_OK;
}

extern "C" int ag_mt19937_export(const void* d_state, uint32_t* h_mt624, int* pos, void* stream) {
    AG_REQUIRE(d_state && h_mt624 && pos, "ag_mt19937_export: null pointer");
    MtState tmp;
    AG_HIP_CHECK(hipMemcpyAsync(&tmp, d_state, sizeof(tmp), hipMemcpyDeviceToHost, (hipStream_t)stream));
    AG_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    memcpy(h_mt624, tmp.mt, sizeof(tmp.mt));
    *pos = tmp.pos;
    return AG_OK;
}

extern "C" int ag_mt19937_raw(void* d_state, uint32_t* d_out, int64_t n, void* stream) {
    AG_REQUIRE(d_state && d_out && n >= 0, "ag_mt19937_raw: bad arguments");
    if (n == 0) return AG_OK;
    hipLaunchKernelGGL(mt_raw_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (MtState*)d_state, d_out, n);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_mt19937_skip(void* d_state, int64_t n, void* stream);
static int run_sampler(void* d_state, int h, int P, int paired, const float* prefix, int64_t* mi, uint32_t* mb,
                       uint32_t* scratch, void* stream, int h_lo = 0, int h_total = -1) {
    if (h_total < 0) h_total = h;
    if (h == 0) return h_total > 0 ? ag_mt19937_skip(d_state, (int64_t)h_total * P + h_total, stream) : AG_OK;
    SamplerArgs a;
    a.h_lo = h_lo; a.h_total = h_total;
    a.st = (MtState*)d_state; a.h = h; a.P = P; a.inv_p = (float)(1.0 / (double)P); a.paired = paired; a.prefix = prefix;
    a.mask_i64 = mi; a.mask_bits = mb; a.scratch = scratch; a.Tw = (P + 1 + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sampler_draw_kernel, dim3(1), dim3(256), 0, s, a);
    AG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sampler_emit_kernel, dim3(ceil_div(h, 4)), dim3(256), 0, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_mt19937_skip(void* d_state, int64_t n, void* stream) {
    AG_REQUIRE(d_state && n >= 0, "ag_mt19937_skip: bad arguments");
    if (n == 0) return AG_OK;
    hipLaunchKernelGGL(mt_skip_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (MtState*)d_state, n);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_mask_shapley_new(void* d_state, int n_mask_samples, int n_players, const float* d_prefix,
                                   int64_t* d_mask_i64, uint32_t* d_mask_bits, uint32_t* d_scratch, void* stream) {
    AG_REQUIRE(d_state && d_prefix && d_scratch, "ag_mask_shapley_new: null pointer");
    AG_REQUIRE(n_mask_samples >= 0 && n_mask_samples % 2 == 0, "ag_mask_shapley_new: n_mask_samples=%d must be even (reference models/shapley.py:62)", n_mask_samples);
    AG_REQUIRE(n_players >= 2, "ag_mask_shapley_new: n_players=%d", n_players);
    return run_sampler(d_state, n_mask_samples / 2, n_players, 1, d_prefix, d_mask_i64, d_mask_bits, d_scratch, stream);
}

extern "C" int ag_mask_shapley_new_rows(void* d_state, int n_mask_samples_total, int row_lo, int row_hi, int n_players,
                                        const float* d_prefix, int64_t* d_mask_i64, uint32_t* d_mask_bits, uint32_t* d_scratch,
                                        void* stream) {
    AG_REQUIRE(d_state && d_prefix && d_scratch, "ag_mask_shapley_new_rows: null pointer");
    AG_REQUIRE(n_mask_samples_total >= 0 && n_mask_samples_total % 2 == 0 && row_lo % 2 == 0 && row_hi % 2 == 0 &&
               0 <= row_lo && row_lo <= row_hi && row_hi <= n_mask_samples_total,
               "ag_mask_shapley_new_rows: rows [%d, %d) of %d must be even-aligned (paired rows 2i / 2i+1 stay together)",
               row_lo, row_hi, n_mask_samples_total);
    AG_REQUIRE(n_players >= 2, "ag_mask_shapley_new_rows: n_players=%d", n_players);
    return run_sampler(d_state, (row_hi - row_lo) / 2, n_players, 1, d_prefix, d_mask_i64, d_mask_bits, d_scratch, stream,
                       row_lo / 2, n_mask_samples_total / 2);
}

extern "C" int ag_mask_purely_uniform(void* d_state, int batch, int n_players, int64_t* d_mask_i64,
                                      uint32_t* d_mask_bits, uint32_t* d_scratch, void* stream) {
    AG_REQUIRE(d_state && d_scratch && batch >= 0 && n_players >= 1, "ag_mask_purely_uniform: bad arguments");
    return run_sampler(d_state, batch, n_players, 0, nullptr, d_mask_i64, d_mask_bits, d_scratch, stream);
}

extern "C" int ag_mask_purely_uniform_rows(void* d_state, int batch_total, int row_lo, int row_hi, int n_players, int64_t* d_mask_i64,
                                           uint32_t* d_mask_bits, uint32_t* d_scratch, void* stream) {
    AG_REQUIRE(d_state && d_scratch && n_players >= 1 && 0 <= row_lo && row_lo <= row_hi && row_hi <= batch_total,
               "ag_mask_purely_uniform_rows: bad arguments (rows [%d, %d) of %d)", row_lo, row_hi, batch_total);
    return run_sampler(d_state, row_hi - row_lo, n_players, 0, nullptr, d_mask_i64, d_mask_bits, d_scratch, stream, row_lo, batch_total);
}

extern "C" int ag_pack_mask(const int64_t* d_mask_i64, int rows, int n_players, uint32_t* d_mask_bits, void* stream) {
    if (rows == 0) return AG_OK;
    AG_REQUIRE(d_mask_i64 && d_mask_bits && rows >= 0 && n_players >= 1, "ag_pack_mask: bad arguments");
    if (rows == 0) return AG_OK;
    hipLaunchKernelGGL(pack_mask_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, d_mask_i64, rows,
                       n_players, d_mask_bits, (n_players + 1 + 31) / 32);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

extern "C" int ag_perturbed_masks(const float* d_attr, int n_attr, int n_players, int steps, int mask_base,
                                  int64_t* d_stops, int64_t* d_mask_i64, void* stream) {
    AG_REQUIRE(d_attr && d_stops && d_mask_i64, "ag_perturbed_masks: null pointer");
    AG_REQUIRE(n_attr >= 1 && n_players >= 1 && steps >= 1 && (mask_base == 0 || mask_base == 1), "ag_perturbed_masks: bad arguments");
    const int st = steps < n_players ? steps : n_players;  // steps = min(P, steps)
    hipLaunchKernelGGL(perturbed_kernel, dim3(n_attr), dim3(256), n_players * sizeof(float), (hipStream_t)stream,
                       d_attr, n_players, st, mask_base, d_stops, d_mask_i64);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
