// gemm_big.hip — 256x256 block-tile bf16 GEMM variants for the large-M GEMMs of the masked forward
// (M = R*T ~ 1e5 rows).  Same contract as gemm.hip (C = epi(A[M,K] W[N,K]^T + b)); selected by
// ag_gemm when M and N are large enough to fill 256-wide tiles.
//
// Why 256^2: the 128^2 kernel needs (128+128)*128 B of L2->LDS traffic per 2.1 MFLOP, i.e. 64 B/clk/CU at
// full MFMA rate — more than an XCD L2 delivers (~56 B/clk/CU); 256^2 halves that, and halves the
// LDS-DMA instructions issued per MFMA (the dominant issue-slot cost next to the MFMAs).
//
// Variant RING (default): K is walked in 32-element (64-byte) half-steps through a 4-slot LDS ring
// (slot = A[256 x 64B] + W[256 x 64B] = 32 KiB).  global_load_lds for half-step j+4 is issued while
// j is computed; a counted s_waitcnt vmcnt(8) (never 0 in the loop) + one raw s_barrier per half-step
// publish slot j+1..; loads stay in flight ACROSS barriers.  8 waves = 2(M) x 4(N), wave tile
// 128(M) x 64(N) = 8x4 MFMA 16x16x32 sub-tiles (128 accumulator VGPRs), 1 workgroup per CU.
// LDS image: 16-row x 64-B sub-tiles (one LDS-DMA piece each), 16-B slot XOR-swizzled by
// f((row>>2)&3) = {0,2,3,1} (applied on the source address and on the ds_read_b128) — conflict-free
// for the 16x16x32 operand read.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BT = 256;                 // block tile (M and N)
constexpr int HROWB = 64;               // bytes of K per row per half-step (32 bf16)
constexpr int HALF_OP_BYTES = BT * HROWB;       // 16 KiB per operand per half-step
constexpr int SLOT_BYTES = 2 * HALF_OP_BYTES;   // 32 KiB
constexpr int NSLOT = 4;
constexpr int NT = 512;

struct BigArgs {
    const char* A; long lda_b;
    const char* W; long ldw_b;
    const float* bias;
    char* C; long ldc;
    const bf16_t* R; long ldr;
    int T, share;
    int M, N, K;
    int nt_store;  // outputs far larger than the 256 MiB Infinity Cache: stream them past the caches
};

__device__ __forceinline__ void glds16b(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz4(int q) { return (0x1320 >> (q * 4)) & 3; }  // {0,2,3,1}[q]

// stage one operand's 256 x 64-B half-step: 16 pieces of 16 rows; wave w takes pieces 2w, 2w+1.
__device__ __forceinline__ void stage_half(const char* base, long ld_b, int row0, int rows_total, long kbyte0,
                                           char* lds_half, int wave, int lane) {
    const int r_in = lane >> 2;                         // row within the 16-row piece
    const int chunk = (lane & 3) ^ swz4((lane >> 4) & 3);  // (row>>2)&3 == (lane>>4)&3 inside a piece
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        int grow = row0 + piece * 16 + r_in;
        grow = grow < rows_total ? grow : rows_total - 1;
        glds16b(base + (long)grow * ld_b + kbyte0 + chunk * 16, lds_half + piece * 1024);
    }
}

__device__ __forceinline__ uint4 frag_half(const char* lds_half, int row16base, int lane) {
    const int r = lane & 15, c = lane >> 4;
    return *reinterpret_cast<const uint4*>(lds_half + (row16base + r) * HROWB + ((c ^ swz4((r >> 2) & 3)) << 4));
}

template <int EPI, int ABL = 0>  // ABL (dev ablations): 1 = no epilogue stores, 2 = no refill loads, 3 = no MFMA
__global__ __launch_bounds__(NT, 2) void gemm_ring_kernel(BigArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_n = (p.N + BT - 1) / BT, tiles_m = (p.M + BT - 1) / BT;
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int m0 = (wg / tiles_n) * BT, n0 = (wg % tiles_n) * BT;

    f32x4_t acc[4][8];  // [n sub-tile][m sub-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nh = p.K / 32;  // half-steps
    // prologue: half-steps 0..3 into slots 0..3 (4 glds per wave per half-step)
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        if (j < nh) {
            stage_half(p.A, p.lda_b, m0, p.M, (long)j * HROWB, smem + j * SLOT_BYTES, wave, lane);
            stage_half(p.W, p.ldw_b, n0, p.N, (long)j * HROWB, smem + j * SLOT_BYTES + HALF_OP_BYTES, wave, lane);
        }
    }

    for (int j = 0; j < nh; ++j) {
        // Retire this wave's loads of half-step j.  Issued so far: 0..3 (prologue) and j'+3 at every
        // iteration j' in [1, j), so at most `ahead` later half-steps (4 loads each) may stay in flight.
        const int ahead = min(nh - 1 - j, j == 0 ? 3 : 2);
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // raw barrier (no vmcnt(0) drain): publishes slot j&3; also proves every wave finished its
        // ds_reads of half-step j-1 (its MFMAs consumed them), so slot (j-1)&3 is free to refill.
        asm volatile("s_barrier" ::: "memory");

        const char* sA = smem + (j & 3) * SLOT_BYTES;
        const char* sW = sA + HALF_OP_BYTES;
        uint4 fw[4], fx[8];
#pragma unroll
        for (int s = 0; s < 4; ++s) fw[s] = frag_half(sW, wn * 64 + s * 16, lane);
#pragma unroll
        for (int s = 0; s < 8; ++s) fx[s] = frag_half(sA, wm * 128 + s * 16, lane);
        if (ABL != 2 && j >= 1 && j + 3 < nh) {
            char* dst = smem + ((j + 3) & 3) * SLOT_BYTES;
            stage_half(p.A, p.lda_b, m0, p.M, (long)(j + 3) * HROWB, dst, wave, lane);
            stage_half(p.W, p.ldw_b, n0, p.N, (long)(j + 3) * HROWB, dst + HALF_OP_BYTES, wave, lane);
        }
        if (ABL == 3) {
#pragma unroll
            for (int sn = 0; sn < 4; ++sn) asm volatile("" ::"v"(fw[sn].x), "v"(fw[sn].y), "v"(fw[sn].z), "v"(fw[sn].w));
#pragma unroll
            for (int sm = 0; sm < 8; ++sm) asm volatile("" ::"v"(fx[sm].x), "v"(fx[sm].y), "v"(fx[sm].z), "v"(fx[sm].w));
        } else {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int sn = 0; sn < 4; ++sn)
#pragma unroll
            for (int sm = 0; sm < 8; ++sm)
                acc[sn][sm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[sn]),
                                                                      __builtin_bit_cast(bf16x8_t, fx[sm]), acc[sn][sm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        }
    }
    if (ABL == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j2 = 0; j2 < 8; ++j2) asm volatile("" ::"v"(acc[i][j2][0]), "v"(acc[i][j2][1]), "v"(acc[i][j2][2]), "v"(acc[i][j2][3]));
        return;
    }

    // ---- epilogue ----
    // The accumulator layout gives a lane 4 consecutive output features of one token (8 bytes of bf16),
    // i.e. 32-byte row segments per store instruction.  Instead each wave transposes its 128x64 tile
    // through its private 16 KiB of the (now idle) ring, 32 rows at a time, and stores whole 128-byte
    // rows: 8 lanes x 16 B per row, 8 rows (1 KiB of full cache lines) per store instruction.
    const int frow = lane & 15, fq = lane >> 4;
    constexpr bool OUT_F32 = (EPI == AG_EPI_BIAS_F32);
    constexpr int SROW = 144;  // staged row: 128 B + 16 B pad (16-B aligned reads, <=2-way write conflicts)
    if (!OUT_F32) asm volatile("s_barrier" ::: "memory");  // every wave is done reading the ring
    char* stg = smem + wave * 16384;
    const int nw0 = n0 + wn * 64;               // first output column of this wave
    const bool full_cols = nw0 + 64 <= p.N;     // N % 8 == 0 guaranteed by eligibility
#pragma unroll
    for (int sm = 0; sm < 8; ++sm) {
        const int m = m0 + wm * 128 + sm * 16 + frow;
        long rrow = 0;
        if (EPI == AG_EPI_BIAS_RESID && m < p.M) {
            const int seq = m / p.T, t = m - seq * p.T;
            rrow = (long)(seq / p.share) * p.T + t;
        }
#pragma unroll
        for (int sn = 0; sn < 4; ++sn) {
            const int n = nw0 + sn * 16 + fq * 4;
            float v[4] = {acc[sn][sm][0], acc[sn][sm][1], acc[sn][sm][2], acc[sn][sm][3]};
            const bool inb = (m < p.M) && (n < p.N);
            if (inb) {
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (EPI == AG_EPI_BIAS_RESID) {
                    const float4 rv = load4_as_f32(p.R + rrow * p.ldr + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
            }
            if (EPI == AG_EPI_BIAS_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fast_gelu(v[e]);
            }
            if (EPI == AG_EPI_BIAS_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
            }
            if (OUT_F32) {
                if (inb) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                *reinterpret_cast<uint2*>(stg + ((sm & 1) * 16 + frow) * SROW + sn * 32 + fq * 8) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
        if (!OUT_F32 && (sm & 1)) {
            // 32 staged rows ready (this wave's own LDS ops complete in order): 4 x (8 rows x 128 B)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = i * 8 + (lane >> 3), ch = lane & 7;
                const uint4 val = *reinterpret_cast<const uint4*>(stg + rr * SROW + ch * 16);
                const int mm = m0 + wm * 128 + (sm - 1) * 16 + rr;
                if (mm < p.M && (full_cols || nw0 + ch * 8 < p.N)) {
                    uint4* dstp = reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (long)mm * p.ldc + nw0 + ch * 8);
                    if (p.nt_store) {
                        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(u32x4{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4*>(dstp));
                    } else *dstp = val;
                }
            }
        }
    }
}

template <int EPI>
int launch_ring(const BigArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT_BYTES);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(gemm_ring): %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ceil_div(a.M, BT) * ceil_div(a.N, BT);
    static const int abl = getenv("AG_GEMM_ABL") ? atoi(getenv("AG_GEMM_ABL")) : 0;
    if (abl && EPI == AG_EPI_BIAS) {  // dev ablations (timing only; results are wrong)
        static bool aset = false;
        if (!aset) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<AG_EPI_BIAS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<AG_EPI_BIAS, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<AG_EPI_BIAS, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<AG_EPI_BIAS, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT_BYTES);
            aset = true;
        }
        if (abl == 1) hipLaunchKernelGGL((gemm_ring_kernel<AG_EPI_BIAS, 1>), dim3(tiles), dim3(NT), NSLOT * SLOT_BYTES, s, a);
        else if (abl == 2) hipLaunchKernelGGL((gemm_ring_kernel<AG_EPI_BIAS, 2>), dim3(tiles), dim3(NT), NSLOT * SLOT_BYTES, s, a);
        else if (abl == 4) hipLaunchKernelGGL((gemm_ring_kernel<AG_EPI_BIAS, 4>), dim3(tiles), dim3(NT), NSLOT * SLOT_BYTES, s, a);
        else hipLaunchKernelGGL((gemm_ring_kernel<AG_EPI_BIAS, 3>), dim3(tiles), dim3(NT), NSLOT * SLOT_BYTES, s, a);
        AG_LAUNCH_CHECK();
        return AG_OK;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<EPI>), dim3(tiles), dim3(NT), NSLOT * SLOT_BYTES, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}


// =================================================================================================
// Variant TILE<BM,BN,WM,WN,NSLOT>: K walked in 64-element steps (128-byte rows = whole cache lines per
// LDS-DMA piece: 8 rows x 128 B), NSLOT-deep LDS ring of [A: BM x 128B | W: BN x 128B] slots, one raw
// barrier per K step, loads for step t+NSLOT-1 issued while t is computed, counted vmcnt.
//   <256,256,2,4,2>: wave tile 128x64, 128 KiB LDS, classic double buffer (vmcnt(0) once per 64-K step)
//   <256,128,4,2,3>: wave tile  64x64, 144 KiB LDS, one K step stays in flight across each barrier
// =================================================================================================
template <int BM, int BN, int WM, int WN, int NS, int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_tile_kernel(BigArgs p) {
    constexpr int ROWB = 128;
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, SLOT = A_BYTES + W_BYTES;
    constexpr int TM = BM / WM, TN = BN / WN;       // wave tile
    constexpr int SM_ = TM / 16, SN_ = TN / 16;     // sub-tiles per wave
    constexpr int PA = BM / 64, PW = BN / 64;       // 8-row pieces per wave per operand
    constexpr int G = PA + PW;                      // glds per wave per K step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

    f32x4_t acc[SN_][SM_];
#pragma unroll
    for (int i = 0; i < SN_; ++i)
#pragma unroll
        for (int j = 0; j < SM_; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int r_in = lane >> 3, slot = lane & 7;
    auto stage = [&](int t, char* dst) {
        const long kb = (long)t * ROWB;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (wave * PA + i) * 8 + r_in;
            int grow = m0 + row; grow = grow < p.M ? grow : p.M - 1;
            glds16b(p.A + (long)grow * p.lda_b + kb + ((slot ^ (row & 7)) << 4), dst + (wave * PA + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int row = (wave * PW + i) * 8 + r_in;
            int grow = n0 + row; grow = grow < p.N ? grow : p.N - 1;
            glds16b(p.W + (long)grow * p.ldw_b + kb + ((slot ^ (row & 7)) << 4), dst + A_BYTES + (wave * PW + i) * 1024);
        }
    };
    auto wait_inflight = [&](int tiles_ahead) {  // leave `tiles_ahead` K steps (G loads each) in flight
        if (tiles_ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (tiles_ahead == 1) { if (G == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else { if (G == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
    };

    const int nk = p.K / 64;
#pragma unroll
    for (int t = 0; t < NS; ++t)
        if (t < nk) stage(t, smem + t * SLOT);

    const int frow = lane & 15, fq = lane >> 4;
    int cur = 0;  // slot of step t
    for (int t = 0; t < nk; ++t) {
        wait_inflight(min(nk - 1 - t, t == 0 ? NS - 1 : NS - 2));
        asm volatile("s_barrier" ::: "memory");
        // slot of step t-1 is free now (every wave consumed it before arriving): refill with t+NS-1
        if (t >= 1 && t + NS - 1 < nk) {
            int prev = cur - 1; if (prev < 0) prev += NS;
            stage(t + NS - 1, smem + prev * SLOT);
        }
        const char* tA = smem + cur * SLOT;
        const char* tW = tA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 fw[SN_], fx[SM_];
#pragma unroll
            for (int s = 0; s < SN_; ++s) {
                const int row = wn * TN + s * 16 + frow;
                fw[s] = *reinterpret_cast<const uint4*>(tW + row * ROWB + (((kk * 4 + fq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int s = 0; s < SM_; ++s) {
                const int row = wm * TM + s * 16 + frow;
                fx[s] = *reinterpret_cast<const uint4*>(tA + row * ROWB + (((kk * 4 + fq) ^ (row & 7)) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int sn = 0; sn < SN_; ++sn)
#pragma unroll
                for (int sm = 0; sm < SM_; ++sm)
                    acc[sn][sm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[sn]),
                                                                          __builtin_bit_cast(bf16x8_t, fx[sm]), acc[sn][sm], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
    }

    constexpr bool OUT_F32 = (EPI == AG_EPI_BIAS_F32);
#pragma unroll
    for (int sm = 0; sm < SM_; ++sm) {
        const int m = m0 + wm * TM + sm * 16 + frow;
        if (m >= p.M) continue;
        long rrow = 0;
        if (EPI == AG_EPI_BIAS_RESID) {
            const int seq = m / p.T, tt = m - seq * p.T;
            rrow = (long)(seq / p.share) * p.T + tt;
        }
#pragma unroll
        for (int sn = 0; sn < SN_; ++sn) {
            const int n = n0 + wn * TN + sn * 16 + fq * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[sn][sm][0], acc[sn][sm][1], acc[sn][sm][2], acc[sn][sm][3]};
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (EPI == AG_EPI_BIAS_RESID) {
                const float4 rv = load4_as_f32(p.R + rrow * p.ldr + n);
                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            }
            if (EPI == AG_EPI_BIAS_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fast_gelu(v[e]);
            }
            if (EPI == AG_EPI_BIAS_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
            }
            if (OUT_F32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int NS, int EPI>
int launch_tile(const BigArgs& a, hipStream_t s) {
    constexpr int LDS = NS * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_kernel<BM, BN, WM, WN, NS, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(gemm_tile): %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ceil_div(a.M, BM) * ceil_div(a.N, BN);
    hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, WM, WN, NS, EPI>), dim3(tiles), dim3(NT), LDS, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}


// =================================================================================================
// Variant RING2 (software-pipelined ring): as RING, but the fragments of half-step j+1 are read into a
// second register set while the MFMAs of half-step j run, and the 4 LDS-DMA refills are spread between
// MFMA groups — so after each barrier the matrix pipe restarts immediately instead of waiting for
// 12 ds_read_b128 + 4 LDS-DMA issues per wave (the two waves of a SIMD run in lockstep behind the shared
// barrier, so nothing else hides that bubble).  K must be a multiple of 64 (even number of half-steps).
// =================================================================================================
template <int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_ring2_kernel(BigArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_n = (p.N + BT - 1) / BT, tiles_m = (p.M + BT - 1) / BT;
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int m0 = (wg / tiles_n) * BT, n0 = (wg % tiles_n) * BT;

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nh = p.K / 32;
    // per-lane source rows of this wave's 4 pieces (A: 2, W: 2) and the swizzled source chunk
    const int r_in = lane >> 2;
    const int chunkb = ((lane & 3) ^ swz4((lane >> 4) & 3)) * 16;
    const char* srcA[2];
    const char* srcW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int ga = m0 + (wave * 2 + i) * 16 + r_in; ga = ga < p.M ? ga : p.M - 1;
        int gw = n0 + (wave * 2 + i) * 16 + r_in; gw = gw < p.N ? gw : p.N - 1;
        srcA[i] = p.A + (long)ga * p.lda_b + chunkb;
        srcW[i] = p.W + (long)gw * p.ldw_b + chunkb;
    }
    auto stage_piece = [&](int j, int which) {  // which: 0,1 = A pieces, 2,3 = W pieces
        char* dst = smem + (j & 3) * SLOT_BYTES + (which >= 2 ? HALF_OP_BYTES : 0) + (wave * 2 + (which & 1)) * 1024;
        const char* src = (which >= 2 ? srcW[which & 1] : srcA[which & 1]) + (long)j * HROWB;
        glds16b(src, dst);
    };
    // fragment read offsets inside a slot (constant per lane)
    const int fr = lane & 15, fc = lane >> 4;
    const int foff = fr * HROWB + ((fc ^ swz4((fr >> 2) & 3)) << 4);
    const int offW = HALF_OP_BYTES + (wn * 64) * HROWB + foff;
    const int offA = (wm * 128) * HROWB + foff;
    auto read_frags = [&](int j, uint4 (&fw)[4], uint4 (&fx)[8]) {
        const char* slot = smem + (j & 3) * SLOT_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) fw[s] = *reinterpret_cast<const uint4*>(slot + offW + s * 16 * HROWB);
#pragma unroll
        for (int s = 0; s < 8; ++s) fx[s] = *reinterpret_cast<const uint4*>(slot + offA + s * 16 * HROWB);
    };

#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
        if (j < nh) { stage_piece(j, 0); stage_piece(j, 1); stage_piece(j, 2); stage_piece(j, 3); }

    uint4 fwA[4], fxA[8], fwB[4], fxB[8];
    // half-step 0 landed (12 younger loads may be in flight) -> publish -> first fragment set
    if (nh >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    read_frags(0, fwA, fxA);

    // one pipelined half-step: wait for j+1, barrier, prefetch its fragments into (fwN,fxN), refill slot
    // (j-1)&3 with half-step j+3 (its readers all passed this barrier), MFMAs of j on (fwC,fxC).
#define AG_RING2_STEP(J, fwC, fxC, fwN, fxN)                                                                   \
    {                                                                                                          \
        const int j_ = (J);                                                                                    \
        if (j_ + 1 < nh) {                                                                                     \
            /* issued so far: 0..3 and j'+3 for j' in [1, j_) -> younger than j_+1: j_+2 (and j_+3 when j_==0) */ \
            const int ahead = min(nh - 2 - j_, j_ == 0 ? 2 : 1);                                               \
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                   \
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                              \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
            asm volatile("s_barrier" ::: "memory");                                                            \
            read_frags(j_ + 1, fwN, fxN);                                                                      \
        }                                                                                                      \
        const bool refill = (j_ >= 1) && (j_ + 3 < nh);                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int sn = 0; sn < 4; ++sn) {                                                     \
            _Pragma("unroll") for (int sm = 0; sm < 8; ++sm)                                                   \
                acc[sn][sm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fwC[sn]),  \
                                                                      __builtin_bit_cast(bf16x8_t, fxC[sm]), acc[sn][sm], 0, 0, 0); \
            if (refill) stage_piece(j_ + 3, sn);                                                               \
        }                                                                                                      \
        __builtin_amdgcn_s_setprio(0);                                                                         \
    }

    for (int j = 0; j < nh; j += 2) {
        AG_RING2_STEP(j, fwA, fxA, fwB, fxB)
        AG_RING2_STEP(j + 1, fwB, fxB, fwA, fxA)
    }
#undef AG_RING2_STEP

    // ---- epilogue ----
    const int frow = lane & 15, fq = lane >> 4;
    constexpr bool OUT_F32 = (EPI == AG_EPI_BIAS_F32);
#pragma unroll
    for (int sm = 0; sm < 8; ++sm) {
        const int m = m0 + wm * 128 + sm * 16 + frow;
        if (m >= p.M) continue;
        long rrow = 0;
        if (EPI == AG_EPI_BIAS_RESID) {
            const int seq = m / p.T, t = m - seq * p.T;
            rrow = (long)(seq / p.share) * p.T + t;
        }
#pragma unroll
        for (int sn = 0; sn < 4; ++sn) {
            const int n = n0 + wn * 64 + sn * 16 + fq * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[sn][sm][0], acc[sn][sm][1], acc[sn][sm][2], acc[sn][sm][3]};
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (EPI == AG_EPI_BIAS_RESID) {
                const float4 rv = load4_as_f32(p.R + rrow * p.ldr + n);
                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            }
            if (EPI == AG_EPI_BIAS_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fast_gelu(v[e]);
            }
            if (EPI == AG_EPI_BIAS_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
            }
            if (OUT_F32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
    }
}

template <int EPI>
int launch_ring2(const BigArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring2_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT_BYTES);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(gemm_ring2): %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ceil_div(a.M, BT) * ceil_div(a.N, BT);
    hipLaunchKernelGGL((gemm_ring2_kernel<EPI>), dim3(tiles), dim3(NT), NSLOT * SLOT_BYTES, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

template <int EPI>
int launch_variant(int variant, const BigArgs& a, hipStream_t s) {
    switch (variant) {
        case 1: return launch_tile<256, 256, 2, 4, 2, EPI>(a, s);
        case 2: return launch_tile<256, 128, 4, 2, 3, EPI>(a, s);
        case 3: return launch_ring2<EPI>(a, s);
        default: return launch_ring<EPI>(a, s);
    }
}

}  // namespace (reopened below)
namespace {
}  // namespace

// Eligibility: bf16, vectorisable epilogue, K a multiple of 32 with at least 4 half-steps.
bool ag_gemm_big_eligible(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, int epilogue) {
    return M >= 1024 && N >= 256 && (N % 8) == 0 && K % 32 == 0 && K >= 128 && (lda % 8) == 0 && (ldc % 8) == 0 &&
           (epilogue != AG_EPI_BIAS_RESID || (ldr % 4) == 0);
}

int ag_gemm_big(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                const void* d_R, int64_t ldr, int rows_per_seq, int resid_share, int M, int N, int K, int epilogue,
                hipStream_t s) {
    BigArgs a;
    a.A = (const char*)d_A; a.lda_b = (long)lda * 2;
    a.W = (const char*)d_W; a.ldw_b = (long)K * 2;
    a.bias = d_bias; a.C = (char*)d_C; a.ldc = ldc; a.R = (const bf16_t*)d_R; a.ldr = ldr;
    a.T = rows_per_seq > 0 ? rows_per_seq : 1; a.share = resid_share > 0 ? resid_share : 1;
    a.M = M; a.N = N; a.K = K;
    static const int nt_env = getenv("AG_GEMM_NT") ? atoi(getenv("AG_GEMM_NT")) : -1;
    a.nt_store = nt_env >= 0 ? nt_env : ((double)M * N * 2.0 > 192.0 * 1024 * 1024);
    static const int env_variant = getenv("AG_GEMM_VARIANT") ? atoi(getenv("AG_GEMM_VARIANT")) : 0;
    const int variant = (K % 64 == 0) ? env_variant : 0;
    switch (epilogue) {
        case AG_EPI_BIAS: return launch_variant<AG_EPI_BIAS>(variant, a, s);
        case AG_EPI_BIAS_GELU: return launch_variant<AG_EPI_BIAS_GELU>(variant, a, s);
        case AG_EPI_BIAS_RESID: return launch_variant<AG_EPI_BIAS_RESID>(variant, a, s);
        case AG_EPI_BIAS_F32: return launch_variant<AG_EPI_BIAS_F32>(variant, a, s);
        case AG_EPI_BIAS_TANH: return launch_variant<AG_EPI_BIAS_TANH>(variant, a, s);
        default: return ag_fail(AG_ERR_INVALID, "ag_gemm_big: unknown epilogue %d", epilogue);
    }
}
