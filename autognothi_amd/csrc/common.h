// common.h — shared device/host helpers for the gfx950 hot-path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <string>

#include "../../include/autognothi_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ---- error plumbing (host) ----------------------------------------------------------------
void ag_set_error(const std::string& msg);
int ag_fail(int code, const char* fmt, ...);
int ag_stream_cus(hipStream_t s);   // capi.cpp: CUs of a CU-masked stream registered with ag_set_stream_cus (0: not registered)
#define AG_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return ag_fail(AG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
extern std::atomic<long long> g_ag_launch_count;   // capi.cpp: kernels this library has launched (ag_launch_count)
#define AG_LAUNCH_CHECK()                      \
    do {                                       \
        g_ag_launch_count.fetch_add(1, std::memory_order_relaxed); \
        AG_HIP_CHECK(hipGetLastError());       \
    } while (0)
#define AG_REQUIRE(cond, ...)                                                                      \
    do {                                                                                           \
        if (!(cond)) return ag_fail(AG_ERR_INVALID, __VA_ARGS__);                                  \
    } while (0)

// ---- bf16 <-> f32 -------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// two fp32 -> packed bf16x2 in ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-preserving)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Store;
template <> struct Store<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Store<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4 consecutive stream elements -> fp32
__device__ __forceinline__ float4 load4_as_f32(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4_as_f32(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}

// counter-based dropout keep decision shared by forward and backward (murmur3 finaliser of (seed, index))
__device__ __forceinline__ bool keep_elem(uint32_t seed, uint64_t idx, float p) {
    uint32_t h = (uint32_t)idx * 0x9E3779B1u ^ (uint32_t)(idx >> 32) * 0x85EBCA77u ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (float)(h >> 8) * 5.9604644775390625e-08f >= p;
}

// Dropout salt of the bf16 training step (ag_set_dropout_salt): mixed into the seed of every dropout decision its kernels make, so
// that a hipGraph-captured step — whose per-site seeds are frozen into the kernel arguments — draws fresh keep patterns at every
// replay.  0 (the value in eager runs) leaves the seeds as given.  One copy per translation unit that defines the setter below.
#define AG_DEFINE_DROPOUT_SALT(setter)                                                                                \
    __device__ uint32_t g_ag_drop_salt = 0u;                                                                          \
    __device__ __forceinline__ uint32_t ag_salted(uint32_t seed) { return seed ^ (g_ag_drop_salt * 0x9E3779B1u); }    \
    /* the salt travels BY VALUE in a kernel argument (a copy from host memory would be read when the copy executes,  \
       not when it is issued: a host many replays ahead of the GPU would overwrite a staging slot still waiting) */     \
    __global__ void ag_set_salt_kernel(uint32_t salt) { g_ag_drop_salt = salt; }                                      \
    int setter(uint32_t salt, hipStream_t s) {                                                                        \
        hipLaunchKernelGGL(ag_set_salt_kernel, dim3(1), dim3(1), 0, s, salt);                                         \
        return hipGetLastError() == hipSuccess ? AG_OK : AG_ERR_HIP;                                                  \
    }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// bf16-mode GELU with ONE transcendental:  with a = min(|x|, 9) and Q(a) = 1 - Phi(a),
//     x Phi(x) = max(x, 0) - a Q(a),        Q(a) ~= 0.5 * exp2(a * (c0 + c1 a + c2 a^2 + c3 a^3))
// (log2(2Q) is smooth: a cubic times a is a minimax fit to |gelu err| <= 1.7e-5 on the whole line against the erf form the
// reference uses, nn.GELU() in models/vanilla_vit.py:491; the degree-8 erf polynomial this replaces had 9.3e-5 and cost 16
// VALU per element).  6 plain VALU + v_exp_f32 per element: the fc1 epilogue runs with the matrix cores idle, where a
// transcendental costs about four plain instructions.  The clamp keeps the cubic negative; the 0.5 is the -1 in the last FMA.
__device__ __forceinline__ float fast_gelu(float x) {
    const float a = fminf(fabsf(x), 9.0f);
    float p = fmaf(0.003938046284019947f, a, -0.044971074908971786f);
    p = fmaf(p, a, -0.46572810411453247f);
    p = fmaf(p, a, -1.1492576599121094f);
    const float e = __builtin_amdgcn_exp2f(fmaf(p, a, -1.0f));
    return fmaf(-a, e, fmaxf(x, 0.0f));
}

// the same on two elements at a time: the Horner chain and the last FMA run as packed fp32 instructions (two elements per
// issue slot in an epilogue that has no MFMAs to share the slots with)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ f32x2_t fast_gelu2(f32x2_t x) {
    f32x2_t a, r;
    a.x = fminf(fabsf(x.x), 9.0f); a.y = fminf(fabsf(x.y), 9.0f);
    r.x = fmaxf(x.x, 0.0f); r.y = fmaxf(x.y, 0.0f);
    const f32x2_t c3 = {0.003938046284019947f, 0.003938046284019947f}, c2 = {-0.044971074908971786f, -0.044971074908971786f},
                  c1 = {-0.46572810411453247f, -0.46572810411453247f}, c0 = {-1.1492576599121094f, -1.1492576599121094f},
                  m1 = {-1.0f, -1.0f};
    f32x2_t p = __builtin_elementwise_fma(c3, a, c2);
    p = __builtin_elementwise_fma(p, a, c1);
    p = __builtin_elementwise_fma(p, a, c0);
    const f32x2_t q = __builtin_elementwise_fma(p, a, m1);
    f32x2_t e;
    e.x = __builtin_amdgcn_exp2f(q.x); e.y = __builtin_amdgcn_exp2f(q.y);
    return __builtin_elementwise_fma(-a, e, r);
}

// the same on eight pairs (a lane's 16 outputs of one token in the large-M GEMM epilogue), written stage by stage: every packed
// FMA of the Horner chain has seven independent ones between itself and its consumer, so no wait state is paid between
// dependent packed instructions (fast_gelu2 pair after pair cost one s_nop per chain link where the scheduler ran out of filler)
__device__ __forceinline__ void fast_gelu2x8(f32x2_t (&lo)[4], f32x2_t (&hi)[4]) {
    const f32x2_t c3 = {0.003938046284019947f, 0.003938046284019947f}, c2 = {-0.044971074908971786f, -0.044971074908971786f},
                  c1 = {-0.46572810411453247f, -0.46572810411453247f}, c0 = {-1.1492576599121094f, -1.1492576599121094f},
                  m1 = {-1.0f, -1.0f};
    f32x2_t a[8], r[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x2_t x = (i & 1) ? hi[i >> 1] : lo[i >> 1];
        a[i].x = fminf(fabsf(x.x), 9.0f); a[i].y = fminf(fabsf(x.y), 9.0f);
        r[i].x = fmaxf(x.x, 0.0f); r[i].y = fmaxf(x.y, 0.0f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(c3, a[i], c2);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], a[i], c1);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], a[i], c0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = __builtin_elementwise_fma(q[i], a[i], m1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { q[i].x = __builtin_amdgcn_exp2f(q[i].x); q[i].y = __builtin_amdgcn_exp2f(q[i].y); }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x2_t g = __builtin_elementwise_fma(-a[i], q[i], r[i]);
        if (i & 1) hi[i >> 1] = g; else lo[i >> 1] = g;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// x[l] + x[l^16] + x[l^32] + x[l^48] on every lane (the four lanes that hold one accumulator row in the 16x16 MFMA layout),
// by the gfx950 row / half swaps: v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows
// of the second, v_permlane32_swap the upper half of the first with the lower half of the second — with both operands the
// same value the two results are x and its partner, so one swap + one add per level (a __shfl_xor is a ds_bpermute with
// address arithmetic and an LDS round trip).  All four lanes end with the same bits: (x0 + x1) + (x2 + x3).
__device__ __forceinline__ float quad_rows_sum(float x) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Experiment / test knobs: an environment variable read ONCE (and again after ag_reload_knobs()), never on the launch path.
extern int g_ag_knob_epoch;   // capi.cpp
struct AgKnob {
    const char* name;
    int epoch;
    bool set;
    double val;
    const char* str;
    explicit AgKnob(const char* n) : name(n), epoch(0), set(false), val(0.0), str(nullptr) {}
    void sync() {
        if (epoch == g_ag_knob_epoch) return;
        str = getenv(name);
        set = str != nullptr;
        val = str ? atof(str) : 0.0;
        epoch = g_ag_knob_epoch;
    }
    double get(double dflt) { sync(); return set ? val : dflt; }
    bool is_set() { sync(); return set; }
};

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t dtype_size(int dtype) { return dtype == AG_BF16 ? 2 : 4; }

// ---- optional per-launch event timing (capi.cpp) -------------------------------------------
struct AgProfScope {
    int idx;
    hipStream_t stream;
    // d_rows != NULL: flops / bytes were computed for rows_upper rows while the launch runs on *d_rows of them (a device-side
    // row count): the totals are scaled by (actual rows / rows_upper) when collected
    AgProfScope(int kernel_class, double flops, double bytes, hipStream_t s, const int* d_rows = nullptr, double rows_upper = 0.0);
    ~AgProfScope();
};

// d_rows (ABI 2: an explicit argument of ag_gemm / ag_gemm_resid_ln / ag_layernorm / ag_gather_rows / ag_side_*): device
// pointer to the actual row count of the launch (NULL = the host-side count is exact).
__device__ __forceinline__ int ag_dyn_clamp(int rows, const int* dyn) {
    if (dyn) { const int d = *dyn; rows = d < rows ? d : rows; }
    return rows;
}

// side_mlp.hip: wide -> narrow Linear (+ GELU, + additive residual) with the weights resident in LDS
bool ag_side_map_eligible(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, int epilogue, bool has_resid);
int ag_side_map(const void* d_x, int64_t ldx, const void* d_w, const float* d_b, const void* d_resid, int64_t ldr, void* d_out,
                int64_t ldo, int M, int N, int K, int gelu, const int* d_rows, hipStream_t s);

// cls_last.hip — the last layer's attention of a CLS-only ViT forward without its key / value projection
bool ag_cls_last_supported(int T, int H, int heads, int dtype);
size_t ag_cls_last_scratch_bytes(int R, int H, int heads);
int ag_cls_last_attention(const void* d_h, const float* d_stats, int cols, const uint32_t* d_mask_bits, const void* d_q, const void* w_kv_ln,
                          const float* b_kv_ln, float ln_eps, void* d_ctx, int64_t ctx_row_stride, int R, int T, int H, int heads,
                          void* d_scratch, size_t scratch_bytes, hipStream_t s);

// gemm_big.hip
bool ag_gemm_big_eligible(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, int epilogue);
int ag_gemm_big(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                const void* d_R, int64_t ldr, int rows_per_seq, int resid_share, int M, int N, int K, int epilogue,
                const float* d_ln_stats, const float* d_ln_colsum, float ln_eps, float* d_stats_out, const int* d_rows, hipStream_t s);
// the plan of ag_gemm_resid_split for a shape on the current device (false: the shape does not split): rows [0, m1) as full rounds of
// the persistent kernel, rows [m1, m1 + m2) as `splits` contraction ranges side by side
bool ag_resid_split_plan(int M, int N, int K, int* m1, int* m2, int* splits);
int ag_device_cus();   // CUs of the current device, rounded down to a multiple of the 8 XCDs
// gemm_big.hip: does the persistent 256^2 kernel run `tiles` tiles on `n_cu` workgroups with a half-height last round?  (the planner prices it)
bool ag_big_half_tail(int tiles, int n_cu, int* half_from, int* ntail, int* grid);

// gemm_tn.hip — the route of one masked-forward GEMM (ag_gemm_ws; csrc/encoder.cpp plans whole layers with it).
//   AG_WS_GEMM       ag_gemm as it is (the persistent 256^2 kernel when the shape fills it, else the 128 / 64-tile kernel)
//   AG_WS_BIG_SPLIT  ag_gemm_resid_split (bias + residual, identity residual rows)
//   AG_WS_EX         128^2 units, one per workgroup, epilogue in the GEMM (bias [+ LayerNorm fold] [+ GELU], or bias + residual
//                    [+ row statistics over 128-column slabs])
//   AG_WS_EX_SLABS   128^2 units x `splits` contraction ranges -> fp32 slabs in the scratch + one row kernel (bias + residual
//                    [+ row statistics over 256-column slabs])
enum { AG_WS_GEMM = 0, AG_WS_BIG_SPLIT = 1, AG_WS_EX = 2, AG_WS_EX_SLABS = 3 };
struct AgWsPlan {
    int route, splits;
    int stats_out_cols;     // columns per slab of the row statistics this route writes (256 / 128; 0: none asked for)
    size_t scratch_bytes;   // of d_scratch (fp32 slabs / partial tiles)
    double cost_us;         // the planner's estimate
    bool valid;
};
// fold_in: the call reads d_ln_stats (slabs of `stats_in_cols` columns); stats_out: it writes d_stats_out, and `out_cols_ok` says which
// slab widths its consumer can read (bit 0: 256, bit 1: 128); route < 0: the cheapest valid route, else that route (valid = false when
// it cannot serve the call)
// (m_expected: with a device-side row count, the rows the cost model should price — M is then the bound that sizes and checks the launch)
AgWsPlan ag_ws_plan(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, int epilogue, int dtype, bool dyn_rows, bool fold_in,
                    int stats_in_cols, bool stats_out, int out_cols_ok, int resid_share, int route = -1, int splits = 0, int m_expected = 0);
// C = A.W^T + bias + LayerNorm(Rpre) (ag_gemm_resid_ln) as 128^2 units x `splits` contraction ranges + the row kernel; d_r_stats /
// d_stats_out in 256-column slabs
int ag_gemm_resid_ln_slabs(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc, const void* d_Rpre, int64_t ldr,
                           const float* d_r_stats, const float* d_ln_g, const float* d_ln_b, float ln_eps, int M, int N, int K, float* d_stats_out,
                           const int* d_rows, int splits, void* d_scratch, size_t scratch_bytes, hipStream_t s);
int ag_gemm_ws_run(const AgWsPlan& plan, const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                   const void* d_R, int64_t ldr, int rows_per_seq, int resid_share, int M, int N, int K, int epilogue, int dtype,
                   const float* d_ln_stats, int stats_in_cols, const float* d_ln_colsum, float ln_eps, float* d_stats_out,
                   const int* d_rows, void* d_scratch, size_t scratch_bytes, hipStream_t s);
