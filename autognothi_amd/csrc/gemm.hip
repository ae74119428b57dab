// gemm.hip — C[M,N] = epilogue(A[M,K] · W[N,K]^T + bias) on gfx950 MFMA.
//
// Replaces every nn.Linear on the masked-forward path (reference models/vanilla_vit.py:422-424,
// :477, :491, :510; models/vanilla_bert.py:488-490, :557, :576, :601).
//
// Structure ("128x128x128B, one barrier per K slice"):
//   * block = 256 threads = 4 waves (2 along M x 2 along N); block tile 128(M) x 128(N) (or 64 x 64); each wave
//     owns 64x64 = 4x4 MFMA 16x16 sub-tiles (64 accumulator VGPRs).
//   * K is walked in 128-byte slices per row (64 bf16 / 32 fp32), so both storage dtypes use the
//     same staging code: each tile row is 8 x 16-B chunks, staged global->LDS with
//     global_load_lds_dwordx4 (no VGPR round trip), through a 2- or 4-slot LDS ring (counted vmcnt).
//   * LDS image is lane-linear (what LDS-DMA requires); the bank-conflict swizzle
//     slot = chunk ^ (row & 7) is applied on the SOURCE address and again on the ds_read_b128.
//   * operands are swapped (MFMA "A" = weight rows, "B" = activation rows) so a lane ends up with
//     4 consecutive output features of one token: bias / residual / output are 8- or 16-byte
//     vector accesses along N.
//   * bf16: v_mfma_f32_16x16x32_bf16 (one per 16-B fragment pair); fp32: v_mfma_f32_16x16x4_f32
//     (four per 16-B fragment pair; exact fp32 fma chain) — the AG_F32 parity mode.
//   * blockIdx -> tile map is XCD-aware: the 8 XCDs get contiguous tile ranges, N-tiles fastest,
//     so the blocks that share one A row-panel run on one XCD and hit its L2.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int ROWB = 128;                      // bytes of K per tile row
// block tile BT x BT (128, or 64 when a launch would otherwise leave most CUs idle): 4 waves as 2 x 2, wave tile BT/2
constexpr int NTHREADS = 256;

struct GemmArgs {
    const char* A; long lda_b;  // byte strides
    const char* W; long ldw_b;
    const float* bias;
    char* C; long ldc;          // element stride
    const void* R; long ldr;  // residual, storage dtype
    int T, share;
    int M, N, K;
    long kbytes;          // K * sizeof(T): 16-B chunks at or beyond it are read from `zeros` instead
    const char* zeros;    // >= 16 B of zeros in HBM
    const int* dyn;       // ag_dynamic_rows(): M is an upper bound, the actual row count is read here (NULL: M is exact)
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// LDS-DMA through inline asm on purpose (as in gemm_big.hip): with the builtin hipcc sees "LDS written by a pending VMEM
// op" and drains vmcnt to 0 in front of the first ds_read of every K slice, which would turn the NST-slot ring below into a
// one-slice-deep one.  The only waits on these loads are the counted s_waitcnt vmcnt(N) of the main loop; the "memory"
// clobber keeps LDS accesses from moving across.  M0 (compiler-reserved) carries the wave-uniform LDS address and is restored.
__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}

// stage one 128-row x 128-byte operand tile; rows beyond `rows_total` are clamped (their products
// only reach accumulators that the epilogue never stores).
template <int BT>
__device__ __forceinline__ void stage_tile(const char* base, long ld_b, int row0, int rows_total, long kbyte0,
                                           char* lds_tile, int wave, int lane, long kbytes, const char* zeros) {
    const int r_in = lane >> 3, slot = lane & 7;
    constexpr int RW = BT / 4;   // rows staged by one wave
#pragma unroll
    for (int i = 0; i < RW / 8; ++i) {
        const int row = wave * RW + i * 8 + r_in;
        int grow = row0 + row;
        grow = grow < rows_total ? grow : rows_total - 1;
        const int chunk = slot ^ (row & 7);
        const long kb = kbyte0 + chunk * 16;   // K tail (K*sizeof(T) not a multiple of 128 B): the missing chunks are zeros
        glds16(kb < kbytes ? base + (long)grow * ld_b + kb : zeros, lds_tile + (wave * RW + i * 8) * ROWB);
    }
}

__device__ __forceinline__ uint4 lds_frag(const char* lds_tile, int row, int chunk) {
    return *reinterpret_cast<const uint4*>(lds_tile + row * ROWB + ((chunk ^ (row & 7)) << 4));
}

// leave the `newer` most recently issued stages (LPS LDS-DMA loads per wave each) in flight: the immediate must be a literal
template <int LPS>
__device__ __forceinline__ void wait_stages(int newer) {
    switch (newer) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPS) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * LPS) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * LPS) : "memory"); break;
    }
}

// NST-slot LDS ring, NST-1 K slices in flight (counted vmcnt, one barrier per slice).  These launches are the ones too small
// for the 256x256 ring kernel (a few hundred to a few thousand rows: the explainer's training step, heads).
template <typename T, int EPI, int BT, int NST>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(GemmArgs pin) {
    GemmArgs p = pin;
    p.M = ag_dyn_clamp(p.M, p.dyn);    // the grid was sized for the upper bound: surplus workgroups leave at once
    constexpr int BM = BT, BN = BT, TILE_BYTES = BT * ROWB, NS = BT / 32, WT = BT / 2;   // NS sub-tiles per wave and dim
    constexpr int LPS = BT / 16;       // LDS-DMA loads per wave and K slice (A + W)
    static_assert((NST - 2) * LPS <= 63 && NST >= 2 && NST <= 8, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    if ((int)blockIdx.x >= nwg) return;
    // bijective XCD remap (blocks b and b+8 share an XCD under round-robin dispatch)
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

    f32x4_t acc[NS][NS];  // [n sub-tile][m sub-tile]
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((p.kbytes + ROWB - 1) / ROWB);
    // LDS: NST slots of [A tile | W tile]
    auto stage = [&](int kt, int slot) {
        char* dst = smem + slot * 2 * TILE_BYTES;
        stage_tile<BT>(p.A, p.lda_b, m0, p.M, (long)kt * ROWB, dst, wave, lane, p.kbytes, p.zeros);
        stage_tile<BT>(p.W, p.ldw_b, n0, p.N, (long)kt * ROWB, dst + TILE_BYTES, wave, lane, p.kbytes, p.zeros);
    };
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_)
        if (s_ < nk) stage(s_, s_);

    const int frow = lane & 15, fq = lane >> 4;
    int slot = 0, slot_in = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // slice kt of this wave has landed (newer slices stay in flight); behind the barrier so have everyone's, and every
        // wave has finished reading the slot of slice kt-1, which the next request overwrites
        wait_stages<LPS>(min(nk - 1 - kt, NST - 2));
        __syncthreads();
        if (kt + NST - 1 < nk) stage(kt + NST - 1, slot_in);
        const char* tA = smem + slot * 2 * TILE_BYTES;
        const char* tW = tA + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 fw[NS], fx[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                fw[s] = lds_frag(tW, wn * WT + s * 16 + frow, kk * 4 + fq);
                fx[s] = lds_frag(tA, wm * WT + s * 16 + frow, kk * 4 + fq);
            }
#pragma unroll
            for (int sn = 0; sn < NS; ++sn)
#pragma unroll
                for (int sm = 0; sm < NS; ++sm) Mma<T>::run(fw[sn], fx[sm], acc[sn][sm]);
        }
        slot = slot + 1 == NST ? 0 : slot + 1;
        slot_in = slot_in + 1 == NST ? 0 : slot_in + 1;
    }

    // ---- epilogue: lane holds n = nb + (lane>>4)*4 + {0..3}, m = mb + (lane&15) per sub-tile ----
    constexpr bool OUT_F32 = (EPI == AG_EPI_BIAS_F32 || sizeof(T) == 4);
    constexpr bool HAS_R = (EPI == AG_EPI_BIAS_RESID || EPI == AG_EPI_BIAS_GELU_ADD);
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && (!HAS_R || (p.ldr & 3) == 0);
#pragma unroll
    for (int sm = 0; sm < NS; ++sm) {
        const int m = m0 + wm * WT + sm * 16 + frow;
        if (m >= p.M) continue;
        long rrow = 0;
        if (HAS_R) {
            const int seq = m / p.T, t = m - seq * p.T;
            rrow = (long)(seq / p.share) * p.T + t;
        }
#pragma unroll
        for (int sn = 0; sn < NS; ++sn) {
            const int n = n0 + wn * WT + sn * 16 + fq * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[sn][sm][0], acc[sn][sm][1], acc[sn][sm][2], acc[sn][sm][3]};
            float rres[4] = {0.f, 0.f, 0.f, 0.f};   // AG_EPI_BIAS_GELU_ADD: residual joins after the activation
            if (vec_ok) {
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (HAS_R) {
                    const float4 rv = load4_as_f32(reinterpret_cast<const T*>(p.R) + rrow * p.ldr + n);
                    if (EPI == AG_EPI_BIAS_RESID) { v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
                    else { rres[0] = rv.x; rres[1] = rv.y; rres[2] = rv.z; rres[3] = rv.w; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n + j < p.N) {
                        if (p.bias) v[j] += p.bias[n + j];
                        if (EPI == AG_EPI_BIAS_RESID) v[j] += Store<T>::load(reinterpret_cast<const T*>(p.R) + rrow * p.ldr + n + j);
                        if (EPI == AG_EPI_BIAS_GELU_ADD) rres[j] = Store<T>::load(reinterpret_cast<const T*>(p.R) + rrow * p.ldr + n + j);
                    }
                }
            }
            if (EPI == AG_EPI_BIAS_GELU || EPI == AG_EPI_BIAS_GELU_ADD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (sizeof(T) == 2 ? fast_gelu(v[j]) : gelu_erf(v[j])) + rres[j];  // bf16 mode: |gelu err| <= 1e-4
            }
            if (EPI == AG_EPI_BIAS_TANH) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = tanhf(v[j]);
            }
            if (OUT_F32) {
                float* cp = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
                if (vec_ok) {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (n + j < p.N) cp[j] = v[j];
                }
            } else {
                bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n;
                if (vec_ok) {
                    *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (n + j < p.N) cp[j] = f32_to_bf16(v[j]);
                }
            }
        }
    }
}

template <typename T, int EPI, int BT, int NST>
int launch_bt(const GemmArgs& a, hipStream_t s) {
    static bool attr_set = false;
    constexpr int LDS = NST * 2 * BT * ROWB;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, EPI, BT, NST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ceil_div(a.M, BT) * ceil_div(a.N, BT);
    hipLaunchKernelGGL((gemm_kernel<T, EPI, BT, NST>), dim3(tiles), dim3(NTHREADS), LDS, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

template <typename T, int EPI>
int launch(const GemmArgs& a, hipStream_t s) {
    // 128^2 tiles unless they would leave most of the 256 CUs (2 workgroups each) without work: the training steps run
    // on a few images (M = B*T ~ 1.5 k rows) and their dW GEMMs have N x K outputs of a few dozen 128^2 tiles.
    // Ring depth.  64^2 tiles: four slots of 16 KiB.  Back to back on L2-warm operands (tools/gemm_small.py) two slots — more
    // workgroups per CU — win by 5-10 % on short K and lose 12-27 % on K >= 2048; inside the training step, where every operand was
    // written by the previous kernel, four slots win for every K (step: ViT-base 18.57 -> 18.03 ms, duo BERT-base 11.44 -> 10.87 ms
    // against two slots everywhere).  The 8-slot variant (one workgroup per CU) lost everywhere and is gone; 128^2 tiles keep two
    // slots of 32 KiB (two workgroups per CU).  AG_GEMM_NST = 2 / 4 overrides (dev knob, read once).
    const long tiles128 = (long)ceil_div(a.M, 128) * ceil_div(a.N, 128);
    static const int bt_env = getenv("AG_GEMM_BT") ? atoi(getenv("AG_GEMM_BT")) : 0;
    static const int nst_env = getenv("AG_GEMM_NST") ? atoi(getenv("AG_GEMM_NST")) : 0;
    if (bt_env == 64 || (bt_env == 0 && tiles128 < 384)) {
        if (nst_env == 2) return launch_bt<T, EPI, 64, 2>(a, s);
        return launch_bt<T, EPI, 64, 4>(a, s);
    }
    return launch_bt<T, EPI, 128, 2>(a, s);
}

template <typename T>
int dispatch(int epi, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case AG_EPI_BIAS: return launch<T, AG_EPI_BIAS>(a, s);
        case AG_EPI_BIAS_GELU: return launch<T, AG_EPI_BIAS_GELU>(a, s);
        case AG_EPI_BIAS_RESID: return launch<T, AG_EPI_BIAS_RESID>(a, s);
        case AG_EPI_BIAS_F32: return launch<T, AG_EPI_BIAS_F32>(a, s);
        case AG_EPI_BIAS_TANH: return launch<T, AG_EPI_BIAS_TANH>(a, s);
        case AG_EPI_BIAS_GELU_ADD: return launch<T, AG_EPI_BIAS_GELU_ADD>(a, s);
        default: return ag_fail(AG_ERR_INVALID, "ag_gemm: unknown epilogue %d", epi);
    }
}

}  // namespace

extern "C" int ag_gemm(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                       const void* d_R, int64_t ldr, int rows_per_seq, int resid_share, int M, int N, int K,
                       int epilogue, int dtype, const float* d_ln_stats, const float* d_ln_colsum, float ln_eps,
                       float* d_stats_out, const int* d_rows, void* stream) {
    if (M == 0) return AG_OK;   // empty row sets (an empty torch tensor has a null data pointer) are legal no-ops
    AG_REQUIRE(d_A && d_W && d_C, "ag_gemm: null pointer");
    AG_REQUIRE(M >= 0 && N > 0 && K > 0, "ag_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    AG_REQUIRE(dtype == AG_BF16 || dtype == AG_F32, "ag_gemm: bad dtype %d", dtype);
    const size_t es = dtype_size(dtype);
    AG_REQUIRE((K * es) % 16 == 0, "ag_gemm: K=%d must be a multiple of %d for this dtype", K, (int)(16 / es));
    AG_REQUIRE((lda * es) % 16 == 0, "ag_gemm: lda=%ld rows must be 16-byte aligned", (long)lda);
    const bool has_r = epilogue == AG_EPI_BIAS_RESID || epilogue == AG_EPI_BIAS_GELU_ADD;
    AG_REQUIRE(!has_r || (d_R && rows_per_seq > 0 && resid_share > 0),
               "ag_gemm: residual epilogue needs R, rows_per_seq and resid_share");
    if (M == 0) return AG_OK;
    GemmArgs a;
    a.A = (const char*)d_A; a.lda_b = (long)lda * es;
    a.W = (const char*)d_W; a.ldw_b = (long)K * es;
    a.bias = d_bias; a.C = (char*)d_C; a.ldc = ldc;
    a.R = d_R; a.ldr = ldr; a.T = rows_per_seq > 0 ? rows_per_seq : 1; a.share = resid_share > 0 ? resid_share : 1;
    a.M = M; a.N = N; a.K = K; a.kbytes = (long)K * es;
    static char* zero_chunk = nullptr;   // source of the K-tail chunks (one per process; never freed)
    if (!zero_chunk) {
        AG_HIP_CHECK(hipMalloc((void**)&zero_chunk, 256));
        AG_HIP_CHECK(hipMemset(zero_chunk, 0, 256));
    }
    a.zeros = zero_chunk;
    a.dyn = d_rows;
    hipStream_t s = (hipStream_t)stream;
    // algorithmic work of this launch: 2*M*N*K flops; bytes = A + W + C (+R) each touched once
    const double out_es = (epilogue == AG_EPI_BIAS_F32) ? 4.0 : (double)es;
    AgProfScope prof(epilogue, 2.0 * M * (double)N * K,
                     (double)M * K * es + (double)N * K * es + (double)M * N * out_es + (has_r ? (double)M * N * es : 0.0), s,
                     d_rows, (double)M);
    static const bool force_small = getenv("AG_GEMM_SMALL") != nullptr;
    const bool big = dtype == AG_BF16 && !force_small && epilogue != AG_EPI_BIAS_GELU_ADD && ag_gemm_big_eligible(M, N, K, lda, ldc, ldr, epilogue);
    AG_REQUIRE(big || (!d_ln_stats && !d_stats_out), "ag_gemm: LayerNorm folding is only available on the large-M bf16 path "
               "(check ag_gemm_supports_ln_fold first)");
    AG_REQUIRE(!d_ln_stats || d_ln_colsum, "ag_gemm: ln_stats given without ln_colsum");
    // the LTT ladder's map (a 768-wide stream into a 96-wide one, + GELU + additive residual): HBM-bound, weights stay in LDS
    static const bool map_off = getenv("AG_SIDE_MLP") && atoi(getenv("AG_SIDE_MLP")) == 0;
    // (resid_share <= 1: the residual row of output row m is row m whatever rows_per_seq says)
    if (dtype == AG_BF16 && !map_off && resid_share <= 1 &&
        ag_side_map_eligible(M, N, K, lda, ldc, ldr, epilogue, epilogue == AG_EPI_BIAS_GELU_ADD))
        return ag_side_map(d_A, lda, d_W, d_bias, epilogue == AG_EPI_BIAS_GELU_ADD ? d_R : nullptr, ldr, d_C, ldc, M, N, K, 1, d_rows, s);
    if (big)
        return ag_gemm_big(d_A, lda, d_W, d_bias, d_C, ldc, d_R, ldr, rows_per_seq, resid_share, M, N, K, epilogue,
                           d_ln_stats, d_ln_colsum, ln_eps, d_stats_out, d_rows, s);
    return dtype == AG_BF16 ? dispatch<bf16_t>(epilogue, a, s) : dispatch<float>(epilogue, a, s);
}

extern "C" int ag_gemm_supports_ln_fold(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, int epilogue, int dtype) {
    static const bool force_small = getenv("AG_GEMM_SMALL") != nullptr;
    // consumer side (ln_stats) exists for the bias / bias+gelu epilogues, producer side (stats_out) for bias+residual
    const bool epi_ok = epilogue == AG_EPI_BIAS || epilogue == AG_EPI_BIAS_GELU || epilogue == AG_EPI_BIAS_RESID;
    return (dtype == AG_BF16 && !force_small && epi_ok && ag_gemm_big_eligible(M, N, K, lda, ldc, ldr, epilogue)) ? 1 : 0;
}
