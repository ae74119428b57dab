// gemm_tn.hip — bf16 GEMM for UNDER-FILLED launches: the explainer / surrogate training step (M = B*T ~ 1-1.6 k rows:
// reference scripts/train_explainer.py:183-198, scripts/train_duo_explainer.py:180-198, where torch.autograd runs the three
// GEMMs of every nn.Linear) and the masked forward at the reference's own batch sizes.
//
//     C[M,N] = sum_kc  A(m,kc) * B(n,kc)            contraction length Kc, fp32 accumulate on v_mfma_f32_16x16x32_bf16
//
// Each operand is read IN PLACE in either storage order, so that no transposed / re-cast copy of an activation, a gradient or
// a weight is ever made:
//     a_col = 0:  A stored [M, Kc] (contraction contiguous)        a_col = 1:  A stored [Kc, M]
//     b_col = 0:  B stored [N, Kc] (the torch [out,in] weight)     b_col = 1:  B stored [Kc, N]
//   forward   Y  = X  . W^T      A = X  [M,K]       B = W  [N,K]         (0,0)  "NT"
//   dX        dX = dY . W        A = dY [M,N]       B = W  [N,K] = [Kc,K] (0,1)  "NN"
//   dW        dW = dY^T . X      A = dY [M,N]=[Kc,N] B = X [M,K] = [Kc,K] (1,1)  "TN"   (output [N,K])
// A "col" operand is staged as a [64 kc][128] LDS image (256-byte rows, 16-byte chunk c of row r at c ^ swz(r)) and its MFMA
// fragments are read with ds_read_b64_tr_b16 (the hardware transposing read); a "row" operand as [128][64 kc] with ds_read_b128.
// Both images are filled by LDS-DMA in whole 128-byte lines (asm, counted vmcnt, as gemm.hip).
//
// Work decomposition: 128 x 128 output tiles x `splits` contraction ranges = one workgroup each (two per CU), so that a
// 1 576 x 768 x 3 072 product is 78 x 4 = 312 units instead of 78.  splits > 1: every unit stores its fp32 partial tile into
// slab s of d_slabs [splits][M][N] with plain stores; the consumer (train_fused.hip: the residual / LayerNorm / cast kernels
// that follow every such GEMM anyway) adds the slabs in slab order: no atomics, no zero fill, bit-reproducible.
//
// Epilogues (splits == 1): bias + store (bf16 / fp32), bias + GELU with BOTH the pre-activation and the activation stored
// (fc1: the backward needs the former, fc2 the latter), and  acc * gelu'(U)  (the dX GEMM of fc2 feeding fc1's backward).
#include "common.h"

namespace {

constexpr int BT = 128;               // block tile (both dims)
constexpr int KS = 64;                // contraction elements per step
constexpr int TILE_BYTES = BT * KS * 2;
constexpr int NTHREADS = 256;

struct ExArgs {
    const char* A; long lda_b;        // byte strides
    const char* B; long ldb_b;
    int M, N, Kc;
    const float* bias;
    char* C; long ldc;                // element stride
    const bf16_t* aux; long ld_aux;   // GELU_BWD: pre-activation U [M, ld_aux]
    bf16_t* out2; long ld_out2;       // GELU_DUAL: gelu(pre) [M, ld_out2]
    float* slabs;                     // [splits][M][N]
    int splits;
    int lds_epilogue;                 // 1: the tile leaves through LDS as whole row segments; 0: direct stores from the accumulator layout
};

// source of every staged 16-byte chunk that lies outside the matrix (contraction tail, ragged columns)
__device__ __attribute__((aligned(16))) const uint32_t g_ex_zero_chunk[4] = {0u, 0u, 0u, 0u};

// (see gemm.hip: through asm so that hipcc neither tracks the LDS write nor drains vmcnt in front of the next ds_read)
__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}

// One operand tile of one contraction step.  `x0`: the tile's first non-contraction index (row of a "row" operand, column of a
// "col" operand), `xn` its extent in the matrix, `kc0` the step's first contraction index.
template <bool COL>
__device__ __forceinline__ void stage_operand(const char* base, long ld_b, int x0, int xn, int kc0, int Kc, char* tile, int wave,
                                              int lane, const char* zeros) {
    if (!COL) {
        // [128 rows][128 B]: 8 rows per instruction, chunk c of row r at slot c ^ (r & 7); rows beyond xn are clamped (their
        // products only reach accumulators that are never stored), contraction chunks beyond Kc come from zeros
        const int r_in = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + r_in;
            int g = x0 + row;
            g = g < xn ? g : xn - 1;
            const int kc = kc0 + ((slot ^ (row & 7)) << 3);
            glds16(kc < Kc ? base + (long)g * ld_b + (long)kc * 2 : zeros, tile + (wave * 32 + i * 8) * 128);
        }
    } else {
        // [64 kc][256 B]: 4 rows per instruction, chunk c of row r at slot c ^ (((r & 3) << 2) | ((r >> 2) & 3)): conflict-free
        // for the transposing reads below; contraction rows beyond Kc and columns beyond xn are zeros
        const int r_in = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 16 + i * 4 + r_in;          // row & 3 == r_in, (row >> 2) & 3 == i
            const int col = x0 + ((pc ^ ((r_in << 2) | i)) << 3);
            const int kc = kc0 + row;
            glds16((kc < Kc && col < xn) ? base + (long)kc * ld_b + (long)col * 2 : zeros, tile + (wave * 16 + i * 4) * 256);
        }
    }
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 b16x4;

// per-lane fragment addressing of one operand for the wave's 64-wide range starting at `wb`
template <bool COL>
struct FragAddr {
    int off[4][2];
    __device__ __forceinline__ void init(int wb, int lane) {
        if (!COL) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = wb + s * 16 + (lane & 15);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) off[s][kk] = row * 128 + (((kk * 4 + (lane >> 4)) ^ (row & 7)) << 4);
            }
        } else {
            // lane 4q+p of 16-lane group g supplies the address of row 8g + 4 hlf + q, columns 4p .. 4p+3 of the 16-column
            // sub-tile; the group's lane i receives column i of those four rows: [hlf 0 | hlf 1] = contraction 8g .. 8g+7
            const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const int row = 8 * g + 4 * hlf + q;
                    const int chunk = (wb >> 3) + 2 * s + (p >> 1);
                    off[s][hlf] = row * 256 + ((chunk ^ ((q << 2) | ((2 * g + hlf) & 3))) << 4) + 8 * (p & 1);
                }
        }
    }
    __device__ __forceinline__ bf16x8_t load(const char* tile, int s, int kk) const {
        if (!COL) {
            return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(tile + off[s][kk]));
        } else {
            const char* b = tile + kk * 8192;   // 32 contraction rows of 256 B
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + off[s][0]));
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + off[s][1]));
            return __builtin_shufflevector(__builtin_bit_cast(b16x4, v0), __builtin_bit_cast(b16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
};

template <int LPS>
__device__ __forceinline__ void wait_stages(int newer) {
    switch (newer) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory"); break;
    }
}

// d/dx [x Phi(x)] = Phi(x) + x phi(x) with the one-exponential tail of fast_gelu (common.h): Q(a) = 1 - Phi(a), a = min(|x|, 9)
__device__ __forceinline__ float fast_gelu_grad(float x) {
    const float a = fminf(fabsf(x), 9.0f);
    float p = fmaf(0.003938046284019947f, a, -0.044971074908971786f);
    p = fmaf(p, a, -0.46572810411453247f);
    p = fmaf(p, a, -1.1492576599121094f);
    const float q = __builtin_amdgcn_exp2f(fmaf(p, a, -1.0f));
    const float cdf = x >= 0.f ? 1.0f - q : q;
    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * x * x);
    return fmaf(x, pdf, cdf);
}

enum { E_STORE_BF16 = 0, E_STORE_F32 = 1, E_GELU_DUAL = 2, E_GELU_BWD = 3, E_SLABS = 4 };

template <bool AC, bool BC, int EPI, int NST>
__global__ __launch_bounds__(NTHREADS, NST == 2 ? 2 : 1) void gemm_ex_kernel(ExArgs p) {
    constexpr int LPS = 8;   // LDS-DMA instructions per wave and step (4 per operand)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + BT - 1) / BT, tiles_m = (p.M + BT - 1) / BT;
    const int nunits = tiles_m * tiles_n * p.splits;
    // bijective XCD remap: units b, b + 8, ... share an XCD under round-robin dispatch and get consecutive work
    const int b = blockIdx.x, xcd = b & 7, q_ = nunits >> 3, r_ = nunits & 7;
    const int unit = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + (b >> 3);
    const int tile = unit / p.splits, sp = unit - tile * p.splits;
    const int m0 = (tile / tiles_n) * BT, n0 = (tile % tiles_n) * BT;
    const int nk_all = (p.Kc + KS - 1) / KS;
    const int k_lo = (int)((long)nk_all * sp / p.splits), k_hi = (int)((long)nk_all * (sp + 1) / p.splits);
    const int nk = k_hi - k_lo;

    f32x4_t acc[4][4];   // [n sub-tile][m sub-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int kt, int slot) {
        char* dst = smem + slot * 2 * TILE_BYTES;
        const char* zeros = reinterpret_cast<const char*>(g_ex_zero_chunk);
        stage_operand<AC>(p.A, p.lda_b, m0, p.M, (k_lo + kt) * KS, p.Kc, dst, wave, lane, zeros);
        stage_operand<BC>(p.B, p.ldb_b, n0, p.N, (k_lo + kt) * KS, p.Kc, dst + TILE_BYTES, wave, lane, zeros);
    };
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_)
        if (s_ < nk) stage(s_, s_);

    FragAddr<AC> fa;   // M side = MFMA "B" operand
    FragAddr<BC> fb;   // N side = MFMA "A" operand
    fa.init(wm * 64, lane);
    fb.init(wn * 64, lane);

    int slot = 0, slot_in = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        wait_stages<LPS>(min(nk - 1 - kt, NST - 2));
        __syncthreads();
        if (kt + NST - 1 < nk) stage(kt + NST - 1, slot_in);
        const char* tA = smem + slot * 2 * TILE_BYTES;
        const char* tB = tA + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fn[4], fm[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { fn[s] = fb.load(tB, s, kk); fm[s] = fa.load(tA, s, kk); }
#pragma unroll
            for (int sn = 0; sn < 4; ++sn)
#pragma unroll
                for (int sm = 0; sm < 4; ++sm)
                    acc[sn][sm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[sn], fm[sm], acc[sn][sm], 0, 0, 0);
        }
        slot = slot + 1 == NST ? 0 : slot + 1;
        slot_in = slot_in + 1 == NST ? 0 : slot_in + 1;
    }

    // ---- epilogue: lane holds n = nb + (lane>>4)*4 + {0..3}, m = mb + (lane&15) per sub-tile (N % 4 == 0) ----
    const int frow = lane & 15, fq = lane >> 4;
    float* slab = EPI == E_SLABS ? p.slabs + (long)sp * p.M * p.N : nullptr;
    // the values of one sub-tile after the epilogue's arithmetic (bias, gelu'): v[4]; E_GELU_DUAL also gives the activation
    auto finish4 = [&](int sm, int sn, int m, int n, float (&v)[4]) {
        v[0] = acc[sn][sm][0]; v[1] = acc[sn][sm][1]; v[2] = acc[sn][sm][2]; v[3] = acc[sn][sm][3];
        if (EPI == E_SLABS) return;
        if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (EPI == E_GELU_BWD) {
            const float4 u = load4_as_f32(p.aux + (long)m * p.ld_aux + n);
            v[0] *= fast_gelu_grad(u.x); v[1] *= fast_gelu_grad(u.y); v[2] *= fast_gelu_grad(u.z); v[3] *= fast_gelu_grad(u.w);
        }
    };
    // the activation of the bf16-ROUNDED pre-activation: what the backward (which only has the rounded one) differentiates
    auto act_of = [](const uint2 pk) {
        const float u0 = __uint_as_float(pk.x << 16), u1 = __uint_as_float(pk.x & 0xFFFF0000u);
        const float u2 = __uint_as_float(pk.y << 16), u3 = __uint_as_float(pk.y & 0xFFFF0000u);
        return make_uint2(pack_bf16x2(fast_gelu(u0), fast_gelu(u1)), pack_bf16x2(fast_gelu(u2), fast_gelu(u3)));
    };
    constexpr bool F32_OUT = EPI == E_SLABS || EPI == E_STORE_F32;
    if (!p.lds_epilogue) {
        // direct stores from the accumulator layout: 16 rows x 32 B (bf16) / 64 B (fp32) per instruction (the parity reference of the
        // staged form below, AG_GEMM_EX_EPI=0)
#pragma unroll
        for (int sm = 0; sm < 4; ++sm) {
            const int m = m0 + wm * 64 + sm * 16 + frow;
            if (m >= p.M) continue;
#pragma unroll
            for (int sn = 0; sn < 4; ++sn) {
                const int n = n0 + wn * 64 + sn * 16 + fq * 4;
                if (n >= p.N) continue;
                float v[4];
                finish4(sm, sn, m, n, v);
                if (EPI == E_SLABS) {
                    *reinterpret_cast<float4*>(slab + (long)m * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (EPI == E_STORE_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) = pk;
                    if (EPI == E_GELU_DUAL) *reinterpret_cast<uint2*>(p.out2 + (long)m * p.ld_out2 + n) = act_of(pk);
                }
            }
        }
        return;
    }
    // Staged form: the tile goes through the (now idle) ring in two rounds of 64 rows — round r holds rows
    // {wm*64 + r*32 + [0, 32)} of both wave rows as a row-major plane with padded rows — and leaves as whole 256 B (bf16) / 512 B
    // (fp32) row segments, 16 B per lane: full cache lines instead of 32 / 64 B pieces of them.
    constexpr int ES = F32_OUT ? 4 : 2;
    constexpr int RB = BT * ES + 16;             // plane row stride in bytes (padding: the 16 rows of a store land on different banks)
    constexpr int PLANE = 64 * RB;
    constexpr int CPR = BT * ES / 16;            // 16-byte chunks per row
    constexpr int RPP = NTHREADS / CPR;          // rows per copy-out pass
    char* out_base = F32_OUT ? (EPI == E_SLABS ? reinterpret_cast<char*>(slab) : p.C) : p.C;
    const long out_ld_b = (EPI == E_SLABS ? (long)p.N : p.ldc) * ES;
    __syncthreads();                             // every wave has read its last fragments: the ring is free
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r) __syncthreads();                  // round 0 has been copied out
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int sm = 2 * r + s2;
            const int lrow = wm * 32 + s2 * 16 + frow;
            const int m = m0 + wm * 64 + sm * 16 + frow;
            const int mc = m < p.M ? m : p.M - 1;          // (rows beyond M are never copied out; clamp what the epilogue may read)
#pragma unroll
            for (int sn = 0; sn < 4; ++sn) {
                const int nl = wn * 64 + sn * 16 + fq * 4;
                const int n = n0 + nl < p.N ? n0 + nl : p.N - 4;
                float v[4];
                finish4(sm, sn, mc, n, v);
                char* dst = smem + lrow * RB + nl * ES;
                if (F32_OUT) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    *reinterpret_cast<uint2*>(dst) = pk;
                    if (EPI == E_GELU_DUAL) *reinterpret_cast<uint2*>(dst + PLANE) = act_of(pk);
                }
            }
        }
        __syncthreads();
        const int chunk = tid % CPR, r0 = tid / CPR;
        const int ncol = n0 + chunk * (16 / ES);
        if (ncol < p.N) {
#pragma unroll
            for (int ps = 0; ps < 64 / RPP; ++ps) {
                const int lrow = ps * RPP + r0;
                const int m = m0 + (lrow >> 5) * 64 + r * 32 + (lrow & 31);
                if (m >= p.M) continue;
                const uint4 val = *reinterpret_cast<const uint4*>(smem + lrow * RB + chunk * 16);
                *reinterpret_cast<uint4*>(out_base + (long)m * out_ld_b + (long)n0 * ES + chunk * 16) = val;
                if (EPI == E_GELU_DUAL) {
                    const uint4 v2 = *reinterpret_cast<const uint4*>(smem + PLANE + lrow * RB + chunk * 16);
                    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out2) + ((long)m * p.ld_out2 + n0) * 2 + chunk * 16) = v2;
                }
            }
        }
    }
}

template <bool AC, bool BC, int EPI, int NST>
int launch_nst(const ExArgs& a, hipStream_t s) {
    constexpr int LDS = NST * 2 * TILE_BYTES;
    static bool attr_set[16] = {};
    int dev = 0;
    AG_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 16 && !attr_set[dev]) {
        AG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ex_kernel<AC, BC, EPI, NST>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set[dev] = true;
    }
    const int units = ceil_div(a.M, BT) * ceil_div(a.N, BT) * a.splits;
    hipLaunchKernelGGL((gemm_ex_kernel<AC, BC, EPI, NST>), dim3(units), dim3(NTHREADS), LDS, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

// Ring depth: two slots of 32 KiB, two workgroups per CU.  Measured against one workgroup per CU with three / four slots (two / three
// contraction steps in flight) on the 12 Linear products of a ViT-base block at 1 576 rows: 179 us against 201 / 209 us.
template <bool AC, bool BC, int EPI>
int launch_ex(const ExArgs& a, hipStream_t s) {
    return launch_nst<AC, BC, EPI, 2>(a, s);
}

template <bool AC, bool BC>
int dispatch_epi(int e, const ExArgs& a, hipStream_t s) {
    switch (e) {
        case E_STORE_BF16: return launch_ex<AC, BC, E_STORE_BF16>(a, s);
        case E_STORE_F32: return launch_ex<AC, BC, E_STORE_F32>(a, s);
        case E_GELU_DUAL: return launch_ex<AC, BC, E_GELU_DUAL>(a, s);
        case E_GELU_BWD: return launch_ex<AC, BC, E_GELU_BWD>(a, s);
        default: return launch_ex<AC, BC, E_SLABS>(a, s);
    }
}

}  // namespace

extern "C" int ag_gemm_ex_splits(int M, int N, int Kc) {
    // Units = tiles x splits should cover the 256 CUs (two workgroups each) about once: measured on the Linear products of a
    // ViT-base / BERT-base block at 1 576 / 1 024 rows the best split puts 430-470 units in flight (78 tiles x 6, 108 x 4, 144 x 3,
    // 36 x 6 of 24 steps); more only adds slab traffic.  A unit keeps at least four contraction steps (256 elements), so that its
    // prologue and epilogue stay the minor part.
    if (M <= 0 || N <= 0 || Kc <= 0) return 1;
    const int tiles = ceil_div(M, BT) * ceil_div(N, BT);
    const int nk = ceil_div(Kc, KS);
    int s = (448 + tiles / 2) / tiles;
    const int smax = nk / 4 > 0 ? nk / 4 : 1;
    s = s < smax ? s : smax;
    s = s > 8 ? 8 : s;
    return s < 1 ? 1 : s;
}

extern "C" int ag_gemm_ex(const void* d_A, int64_t lda, int a_col, const void* d_B, int64_t ldb, int b_col, int M, int N, int Kc,
                          int epilogue, const float* d_bias, void* d_C, int64_t ldc, int c_dtype, const void* d_aux, int64_t ld_aux,
                          void* d_out2, int64_t ld_out2, int splits, float* d_slabs, void* stream) {
    if (M == 0 || N == 0) return AG_OK;
    AG_REQUIRE(d_A && d_B, "ag_gemm_ex: null operand");
    AG_REQUIRE(M > 0 && N > 0 && Kc > 0, "ag_gemm_ex: bad shape M=%d N=%d Kc=%d", M, N, Kc);
    AG_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "ag_gemm_ex: N and the operand row strides must be multiples of 8 "
               "(M=%d N=%d Kc=%d lda=%ld ldb=%ld)", M, N, Kc, (long)lda, (long)ldb);
    // (a row-stored operand is staged in 16-byte chunks along the contraction; a column-stored one row by row: any Kc)
    AG_REQUIRE((a_col && b_col) || Kc % 8 == 0, "ag_gemm_ex: Kc=%d must be a multiple of 8 unless both operands are stored [Kc, .]", Kc);
    AG_REQUIRE(!a_col || M % 8 == 0, "ag_gemm_ex: a transposed A needs M %% 8 == 0 (M=%d)", M);
    AG_REQUIRE((a_col == 0 || a_col == 1) && (b_col == 0 || b_col == 1) && !(a_col && !b_col),
               "ag_gemm_ex: operand orders (a_col, b_col) in {(0,0), (0,1), (1,1)}");
    AG_REQUIRE(((uintptr_t)d_A % 16) == 0 && ((uintptr_t)d_B % 16) == 0, "ag_gemm_ex: operands must be 16-byte aligned");
    AG_REQUIRE(splits >= 1 && splits <= 64, "ag_gemm_ex: splits=%d", splits);
    int epi;
    if (epilogue == AG_EX_SLABS) {
        AG_REQUIRE(d_slabs && ((uintptr_t)d_slabs % 16) == 0, "ag_gemm_ex: AG_EX_SLABS needs d_slabs [splits][M][N]");
        epi = E_SLABS;
    } else {
        AG_REQUIRE(splits == 1, "ag_gemm_ex: splits > 1 needs AG_EX_SLABS");
        AG_REQUIRE(d_C && ldc % 4 == 0 && ((uintptr_t)d_C % 16) == 0, "ag_gemm_ex: C must be 16-byte aligned with ldc %% 4 == 0");
        AG_REQUIRE(!d_bias || ((uintptr_t)d_bias % 16) == 0, "ag_gemm_ex: bias must be 16-byte aligned");
        if (epilogue == AG_EX_STORE) {
            AG_REQUIRE(c_dtype == AG_BF16 || c_dtype == AG_F32, "ag_gemm_ex: bad c_dtype %d", c_dtype);
            epi = c_dtype == AG_BF16 ? E_STORE_BF16 : E_STORE_F32;
        } else if (epilogue == AG_EX_GELU_DUAL) {
            AG_REQUIRE(c_dtype == AG_BF16 && d_out2 && ld_out2 % 4 == 0, "ag_gemm_ex: AG_EX_GELU_DUAL stores bf16 C and bf16 out2");
            epi = E_GELU_DUAL;
        } else if (epilogue == AG_EX_GELU_BWD) {
            AG_REQUIRE(c_dtype == AG_BF16 && d_aux && ld_aux % 4 == 0, "ag_gemm_ex: AG_EX_GELU_BWD needs the bf16 pre-activation in d_aux");
            epi = E_GELU_BWD;
        } else {
            return ag_fail(AG_ERR_INVALID, "ag_gemm_ex: unknown epilogue %d", epilogue);
        }
    }
    ExArgs a;
    a.A = (const char*)d_A; a.lda_b = (long)lda * 2;
    a.B = (const char*)d_B; a.ldb_b = (long)ldb * 2;
    a.M = M; a.N = N; a.Kc = Kc;
    a.bias = d_bias; a.C = (char*)d_C; a.ldc = ldc;
    a.aux = (const bf16_t*)d_aux; a.ld_aux = ld_aux;
    a.out2 = (bf16_t*)d_out2; a.ld_out2 = ld_out2;
    a.slabs = d_slabs; a.splits = splits;
    static AgKnob epi_knob("AG_GEMM_EX_EPI");
    // (the staged epilogue stores 16-byte chunks: 8 bf16 / 4 fp32 columns; row strides must keep them aligned)
    a.lds_epilogue = epi_knob.get(1) != 0 && (epi == E_SLABS || ldc % 8 == 0) && (epi != E_GELU_DUAL || ld_out2 % 8 == 0);
    hipStream_t s = (hipStream_t)stream;
    const double out_b = epi == E_SLABS ? 4.0 * splits : (epi == E_STORE_F32 ? 4.0 : (epi == E_GELU_DUAL ? 4.0 : 2.0));
    AgProfScope prof(AG_PROF_GEMM_EX, 2.0 * M * (double)N * Kc,
                     2.0 * ((double)M * Kc + (double)N * Kc) + out_b * (double)M * N + (epi == E_GELU_BWD ? 2.0 * (double)M * N : 0.0), s);
    if (!a_col && !b_col) return dispatch_epi<false, false>(epi, a, s);
    if (!a_col && b_col) return dispatch_epi<false, true>(epi, a, s);
    return dispatch_epi<true, true>(epi, a, s);
}
