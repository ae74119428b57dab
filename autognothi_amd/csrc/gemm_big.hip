// gemm_big.hip — 256x256 block-tile bf16 GEMM for the large-M GEMMs of the masked forward
// (M = R*T ~ 1e5 rows).  Same contract as gemm.hip (C = epi(A[M,K] W[N,K]^T + b)); selected by
// ag_gemm when M and N are large enough to fill 256-wide tiles.
//
// Why 256^2: the 128^2 kernel needs (128+128)*128 B of L2->LDS traffic per 2.1 MFLOP, i.e. 64 B/clk/CU at
// full MFMA rate — more than an XCD L2 delivers (~56 B/clk/CU); 256^2 halves that, and halves the
// LDS-DMA instructions issued per MFMA (the dominant issue-slot cost next to the MFMAs).
//
// K is walked in 32-element (64-byte) half-steps through a 4-slot LDS ring
// (slot = A[256 x 64B] + W[256 x 64B] = 32 KiB).  global_load_lds for half-step j+4 is issued while
// j is computed; a counted s_waitcnt vmcnt(8) (never 0 in the loop) + one raw s_barrier per half-step
// publish slot j+1..; loads stay in flight ACROSS barriers.  8 waves = 2(M) x 4(N), wave tile
// 128(M) x 64(N) = 8x4 MFMA 16x16x32 sub-tiles (128 accumulator VGPRs), 1 workgroup per CU.
// LDS image: 16-row x 64-B sub-tiles (one LDS-DMA piece each), 16-B slot XOR-swizzled by
// f((row>>2)&3) = {0,2,3,1} (applied on the source address and on the ds_read_b128) — conflict-free
// for the 16x16x32 operand read.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int BT = 256;                 // block tile (M and N)
constexpr int HROWB = 64;               // bytes of K per row per half-step (32 bf16)
constexpr int HALF_OP_BYTES = BT * HROWB;       // 16 KiB per operand per half-step
constexpr int SLOT_BYTES = 2 * HALF_OP_BYTES;   // 32 KiB
constexpr int NSLOT = 4;
constexpr int NT = 512;
// LDS tail behind the ring: what a tile's epilogue needs from memory besides the residual, fetched while the ring fills for the
// first time (kernel start) instead of at the start of the epilogue, where the matrix cores are idle and an L2 round trip is paid
// in full: one (mean, rstd) pair per tile row (LayerNorm-fold consumer / residual-LayerNorm variant) and three per-column
// constants (bias | folded column sums or residual-LN gamma | residual-LN beta)
constexpr int TAIL_STAT = 0, TAIL_C0 = BT * 8, TAIL_C1 = TAIL_C0 + BT * 4, TAIL_C2 = TAIL_C1 + BT * 4, TAIL_BYTES = TAIL_C2 + BT * 4;
constexpr int LDS_BYTES = NSLOT * SLOT_BYTES + TAIL_BYTES;

// loads the compiler does not see (no s_waitcnt of its own on them: its vmcnt bookkeeping does not know the LDS-DMA loads
// issued after these, so a wait of its making would drain the whole prologue): waited for by the counted wait that the
// prologue needs anyway, and tied to it with gload_landed()
__device__ __forceinline__ float gload_f32_async(const float* ptr) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
typedef float f32x2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v_t gload_f32x2_async(const float* ptr) {
    f32x2v_t v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ void gload_landed(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void gload_landed(f32x2v_t& v) { asm volatile("" : "+v"(v)); }

struct BigArgs {
    const char* A; long lda_b;
    const char* W; long ldw_b;
    const float* bias;
    char* C; long ldc;
    const bf16_t* R; long ldr;
    int T, share;
    int M, N, K;
    // LayerNorm folding (ViT pre-LN, bf16 mode): consumer side  out = rstd[m]*(acc - mean[m]*ln_s[n]) + bias[n]
    // with (sum, sumsq) of row m in ln_stats[2m..]; producer side accumulates (sum, sumsq) of the rows it writes.
    // Row statistics travel as per-256-column partial sums, slab-major: stats[s][m] = (sum, sumsq) of columns
    // [256 s, 256 s + 256) of row m, S = ceil(H / 256) slabs of 2 M floats.  Every element is written exactly once by the
    // producing tile (plain stores: no zero fill, no atomics) and the consumer adds the slabs in a fixed order, so the
    // folded LayerNorm is bit-reproducible from run to run.
    const float* ln_stats; const float* ln_s; float ln_eps; float ln_inv_h;
    float* stats_out;
    long stats_slab;   // floats between slabs (= 2 M)
    int ln_nslab;      // slabs of ln_stats (= ceil(K / 256); residual-LayerNorm variant: ceil(N / 256))
    // residual-LayerNorm variant (BERT post-LN, VAR = 3): the residual operand R holds PRE-LayerNorm rows h, ln_stats their
    // statistics; the epilogue adds LN(h)[m, n] = (h - mean[m]) * rstd[m] * rln_g[n] + rln_b[n] instead of h
    const float* rln_g; const float* rln_b;
    unsigned long long* dbg;  // diagnostic build only
    const int* dyn;  // ag_dynamic_rows(): actual row count (NULL: M is exact)
    int ngrp;      // N-tiles per tile-order group (>= 1)
    int nt_store;  // outputs far larger than the 256 MiB Infinity Cache: stream them past the caches
    // contraction ranges side by side (gemm_stream_kernel, fp32-output epilogue only; ag_gemm_resid_split): `nbatch` products of the
    // same shape in one launch, product z reading A / W `bk_b` bytes further along K and writing `bc` output elements further
    int nbatch; long bk_b; long bc;
    // half-height tail (gemm_stream_kernel<..., HT = true>): units [0, half_from) are whole 256^2 tiles (complete rounds of the resident
    // workgroups); the `ntail` tiles left run as 128-row halves, two units each, all in the last round
    int half_from, ntail;
};

// LDS-DMA issued through inline asm ON PURPOSE: hipcc does not count an asm load in its s_waitcnt
// bookkeeping.  With the builtin it sees "LDS written by a pending VMEM op" and puts s_waitcnt vmcnt(0) in
// front of the first ds_read after every barrier, which drains the whole ring each half-step and turns the
// 3-deep prefetch into a 1-deep one.  Here the only waits on these loads are the counted s_waitcnt vmcnt(N)
// written by hand in the main loop; the "memory" clobber keeps the compiler from moving LDS accesses across.
// M0 carries the wave-uniform LDS byte address and is restored (it is compiler-reserved).
__device__ __forceinline__ void glds16b(const char* gsrc, char* lds_wave_base) {
    const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}

// one LDS-DMA piece: 64 lanes x 16 B from (SGPR base + per-lane byte offset) to LDS at M0 (+ lane*16)
__device__ __forceinline__ void glds16b_s(const char* sbase, uint32_t voff, uint32_t lds_off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}

// this wave's four pieces of one half-step (A piece 0, W piece 0, A piece 1, W piece 1) in one statement: M0 saved and
// restored once, no compiler-scheduled instructions in between
__device__ __forceinline__ void glds16b_s_x4(const char* baseA, const char* baseW, uint32_t oA0, uint32_t oW0, uint32_t oA1,
                                             uint32_t oW1, uint32_t ldsA0, uint32_t ldsW0) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
                 "s_add_u32 m0, %7, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, %8, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(oA0), "v"(oW0), "v"(oA1), "v"(oW1), "s"(baseA), "s"(baseW), "s"(ldsA0), "s"(ldsW0) : "memory", "scc");
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

// ---- epilogue of one wave's 128(M) x 64(N) accumulator tile --------------------------------------------
// The accumulator layout gives a lane 4 consecutive output features of one token (8 bytes of bf16),
// i.e. 32-byte row segments per store instruction.  Instead each wave transposes its tile through a private
// piece of the (now idle) ring, 32 rows at a time, and stores whole 128-byte rows: 8 lanes x 16 B per row,
// 8 rows (1 KiB of full cache lines) per store instruction.
// Everything the tile needs from memory besides the residual (bias, LayerNorm column sums, row statistics) is
// requested up front in one batch and the residual rows are prefetched one 16-row block ahead: the epilogue
// runs with the matrix cores idle, so every exposed L2 round trip in it is paid in full.
// Out-of-range rows / columns are clamped for the loads and masked at the stores; no divergent branches.
template <int EPI, bool LNF, bool STATS, bool RLN = false, int LAYOUT = 0>
__device__ __forceinline__ void wave_epilogue(const BigArgs& p, f32x4_t (&acc)[4][8], const int mw0, const int nw0,
                                              char* stg, const int lane, const char* smem_base, const int tile_n,
                                              const char* tail_at = nullptr, const long c_ofs = 0, const int m_lim = -1, const bool idle = false) {
    // m_lim: rows of the matrix this TILE may write (a half-height tile of gemm_stream_kernel ends 128 rows after its origin: the rows
    // behind belong to another unit); idle: a wave none of whose rows the tile owns — it only keeps the workgroup's barriers
    const int Mlim = m_lim >= 0 ? m_lim : p.M;
    if (idle) {
        if (EPI != AG_EPI_BIAS_F32) asm volatile("s_barrier" ::: "memory");
        if (STATS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }
    const int frow = lane & 15, fq = lane >> 4;
    constexpr bool OUT_F32 = (EPI == AG_EPI_BIAS_F32);
    constexpr bool RESID = (EPI == AG_EPI_BIAS_RESID);
    // staged row: 128 B + 16 B pad (16-B aligned reads; writes conflict-free, the whole-row reads of two consecutive rows meet in four banks).
    // (Round 5: 128-byte rows with the main loop's chunk swizzle c ^ ((r >> 1) & 7) instead — both accesses conflict-free, SQ_LDS_BANK_CONFLICT of the
    // QKV / fc1 launches 57 k / 85 k cycles per CU -> the residual class's 14 k — measured SLOWER, same box: QKV 811 -> 817 us, fc1 1 355 -> 1 358: four
    // per-lane staging addresses instead of one base + literal offsets cost more than the conflicts, which are not on the epilogue's critical path.)
    constexpr int SROW = 144;
    // this wave's 128 x (sum, sumsq) partials, behind the 32 staged rows (4.6 KB) of its LDS piece.  LAYOUT 0: 16 KiB per wave from
    // the start of the ring; LAYOUT 1 (gemm_stream_kernel: step slot 0 already holds the NEXT tile's first step image): 8 KiB per
    // wave in step slot 1 — waves 0-3 in the A half (32 KiB ..), waves 4-7 in the W half (96 KiB ..)
    constexpr int STAT_OFF = LAYOUT ? 4608 : 8192;
    constexpr int WPIECE = LAYOUT ? 8192 : 16384, GPIECE = LAYOUT ? 65536 : 65536, GBASE = LAYOUT ? 32768 : 0;
    const bool full_cols = nw0 + 64 <= p.N;     // N % 8 == 0 guaranteed by eligibility
    const bool edge_tile = !full_cols || mw0 + 128 > Mlim;   // (wave-uniform) some of this wave's outputs lie outside the matrix

    // ---- row and column constants: parked in the LDS tail by the kernel's prologue (tile_constants_*) ----
    const char* const tail = tail_at ? tail_at : smem_base + NSLOT * SLOT_BYTES;   // (gemm_stream_kernel alternates between two tails)
    const float2* const stat_lds = reinterpret_cast<const float2*>(tail + TAIL_STAT);
    const int stg_off = (int)(stg - smem_base) - GBASE;
    const int wave_id = (stg_off / GPIECE) * 4 + (stg_off % GPIECE) / WPIECE;
    constexpr bool ROWST = LNF || RLN;   // this tile's rows need (mean, rstd): of the A rows (fold) or of the residual rows
    float4 bv[4], sv[4];
    float4 gv[RLN ? 4 : 1], btv[RLN ? 4 : 1];
#pragma unroll
    for (int sn = 0; sn < 4; ++sn) {
        const int cl = (wave_id & 3) * 64 + sn * 16 + fq * 4;          // tile-local column of this lane's four outputs
        bv[sn] = *reinterpret_cast<const float4*>(tail + TAIL_C0 + cl * 4);
        sv[sn] = LNF ? *reinterpret_cast<const float4*>(tail + TAIL_C1 + cl * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (RLN) {
            gv[sn] = *reinterpret_cast<const float4*>(tail + TAIL_C1 + cl * 4);
            btv[sn] = *reinterpret_cast<const float4*>(tail + TAIL_C2 + cl * 4);
        }
    }
    int ncl[4];
#pragma unroll
    for (int sn = 0; sn < 4; ++sn) {
        const int n = nw0 + sn * 16 + fq * 4;
        ncl[sn] = n < p.N ? n : p.N - 4;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // every wave is done reading the ring (the staging buffers reuse it)
    if (!OUT_F32) asm volatile("s_barrier" ::: "memory");
    // residual row of output row m: ((m / T) / share) * T + (m % T); walked incrementally (m advances by 16)
    int r_t = 0, r_seq_rem = 0;
    long r_base = 0;  // (seq / share) * T
    uint2 rnext[4];
    auto resid_issue = [&](uint2 (&dst)[4], bool valid) {   // rows past M are never stored: read row 0 for them
        const long row = valid ? r_base + r_t : 0;
#pragma unroll
        for (int sn = 0; sn < 4; ++sn)
            dst[sn] = *reinterpret_cast<const uint2*>(p.R + row * p.ldr + ncl[sn]);
    };
    auto resid_advance = [&]() {
        r_t += 16;
        while (r_t >= p.T) {
            r_t -= p.T;
            if (++r_seq_rem == p.share) { r_seq_rem = 0; r_base += p.T; }
        }
    };
    if (RESID) {
        const int m = mw0 + frow;
        const int seq = m / p.T;
        r_t = m - seq * p.T;
        const int sq = seq / p.share;
        r_seq_rem = seq - sq * p.share;
        r_base = (long)sq * p.T;
        resid_issue(rnext, m < Mlim);
    }

#pragma unroll
    for (int sm = 0; sm < 8; ++sm) {
        const int m = mw0 + sm * 16 + frow;
        uint2 rcur[4];
        if (RESID) {
#pragma unroll
            for (int sn = 0; sn < 4; ++sn) rcur[sn] = rnext[sn];
            if (sm < 7) {
                resid_advance();
                resid_issue(rnext, m + 16 < Mlim);
            }
        }
        float ln_mean = 0.f, ln_rstd = 1.f;
        if (ROWST) {
            const float2 mr = stat_lds[(wave_id >> 2) * 128 + sm * 16 + frow];
            ln_mean = mr.x; ln_rstd = mr.y;
        }
        f32x2_t rs2 = {0.f, 0.f}, rq2 = {0.f, 0.f};  // producer side: stats of the bf16-rounded values this lane writes
        // LayerNorm fold + bias of all four column groups first, as PACKED fp32 FMAs (two elements per issue slot: this code runs with the
        // matrix cores idle, every VALU slot of it is tile time): rstd * (acc - mean * s) + b  =  fma(rstd, fma(-mean, s, acc), b), the
        // contraction hipcc made of the scalar form (bit-identical)
        f32x2_t vlo[4], vhi[4];
#pragma unroll
        for (int sn = 0; sn < 4; ++sn) {
            vlo[sn] = f32x2_t{acc[sn][sm][0], acc[sn][sm][1]}; vhi[sn] = f32x2_t{acc[sn][sm][2], acc[sn][sm][3]};
            const f32x2_t blo = {bv[sn].x, bv[sn].y}, bhi = {bv[sn].z, bv[sn].w};
            if (LNF) {
                const f32x2_t nm2 = {-ln_mean, -ln_mean}, rs2 = {ln_rstd, ln_rstd};
                vlo[sn] = __builtin_elementwise_fma(rs2, __builtin_elementwise_fma(nm2, f32x2_t{sv[sn].x, sv[sn].y}, vlo[sn]), blo);
                vhi[sn] = __builtin_elementwise_fma(rs2, __builtin_elementwise_fma(nm2, f32x2_t{sv[sn].z, sv[sn].w}, vhi[sn]), bhi);
            } else {
                vlo[sn] += blo; vhi[sn] += bhi;
            }
        }
        if (EPI == AG_EPI_BIAS_GELU) fast_gelu2x8(vlo, vhi);   // the eight pairs stage by stage: no dependent back-to-back packed FMAs
#pragma unroll
        for (int sn = 0; sn < 4; ++sn) {
            const int n = nw0 + sn * 16 + fq * 4;
            float v[4] = {vlo[sn].x, vlo[sn].y, vhi[sn].x, vhi[sn].y};
            const bool inb = (m < Mlim) && (n < p.N);
            if (RESID && RLN) {   // residual = LayerNorm of the stored pre-LN row (never materialised)
                constexpr int SI = RLN ? 1 : 0;    // (gv / btv have one element in the other instantiations)
                const float nm = -ln_mean;
                const float h0 = (__uint_as_float(rcur[sn].x << 16) + nm) * ln_rstd, h1 = (__uint_as_float(rcur[sn].x & 0xFFFF0000u) + nm) * ln_rstd;
                const float h2 = (__uint_as_float(rcur[sn].y << 16) + nm) * ln_rstd, h3 = (__uint_as_float(rcur[sn].y & 0xFFFF0000u) + nm) * ln_rstd;
                v[0] += fmaf(h0, gv[sn * SI].x, btv[sn * SI].x); v[1] += fmaf(h1, gv[sn * SI].y, btv[sn * SI].y);
                v[2] += fmaf(h2, gv[sn * SI].z, btv[sn * SI].z); v[3] += fmaf(h3, gv[sn * SI].w, btv[sn * SI].w);
            } else if (RESID) {
                v[0] += __uint_as_float(rcur[sn].x << 16); v[1] += __uint_as_float(rcur[sn].x & 0xFFFF0000u);
                v[2] += __uint_as_float(rcur[sn].y << 16); v[3] += __uint_as_float(rcur[sn].y & 0xFFFF0000u);
            }
            if (EPI == AG_EPI_BIAS_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
            }
            if (OUT_F32) {
                if (inb) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + c_ofs + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                *reinterpret_cast<uint2*>(stg + ((sm & 1) * 16 + frow) * SROW + sn * 32 + fq * 8) = pk;
                if (STATS) {   // two partial sums per lane (even / odd element pairs) on packed instructions, folded once per row
                    f32x2_t ra = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xFFFF0000u)};
                    f32x2_t rb = {__uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xFFFF0000u)};
                    if (edge_tile) { const float keep = inb ? 1.f : 0.f; const f32x2_t k2 = {keep, keep}; ra *= k2; rb *= k2; }
                    rs2 += ra + rb;
                    rq2 = __builtin_elementwise_fma(rb, rb, __builtin_elementwise_fma(ra, ra, rq2));
                }
            }
        }
        if (STATS) {  // lanes frow + 16*fq hold parts of row m: combine the 4 column groups; this wave's 64-column partial of
            // the row goes to LDS (combined with the other three column waves after the loop)
            float row_s = rs2.x + rs2.y, row_q = rq2.x + rq2.y;
            row_s = quad_rows_sum(row_s); row_q = quad_rows_sum(row_q);
            if (fq == 0) *reinterpret_cast<float2*>(stg + STAT_OFF + (sm * 16 + frow) * 8) = make_float2(row_s, row_q);
        }
        if (!OUT_F32 && (sm & 1)) {
            // 32 staged rows ready (this wave's own LDS ops complete in order): 4 x (8 rows x 128 B).  (Stores after every 16 rows instead:
            // built and measured in round 4, same box — no gain, fc1 + GELU +0.6 %.)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = i * 8 + (lane >> 3), ch = lane & 7;
                const uint4 val = *reinterpret_cast<const uint4*>(stg + rr * SROW + ch * 16);
                const int mm = mw0 + (sm - 1) * 16 + rr;
                if (mm < Mlim && (full_cols || nw0 + ch * 8 < p.N)) {
                    uint4* dstp = reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (long)mm * p.ldc + nw0 + ch * 8);
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    if (p.nt_store) {
                        __builtin_nontemporal_store(u32x4{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4*>(dstp));
                    } else *dstp = val;
                }
            }
        }
    }
    if (STATS) {
        // the four column waves (wn = 0..3) of this row half have each left 128 row partials in their LDS pieces: add them
        // in wave order and store ONE (sum, sumsq) per row and column tile.  Wave wn finishes rows [32 wn, 32 wn + 32).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int wave = wave_id, wn = wave & 3;
        if (lane < 32) {
            const int row = wn * 32 + lane;
            const char* half = smem_base + GBASE + (wave >> 2) * GPIECE + STAT_OFF + row * 8;
            float2 t = *reinterpret_cast<const float2*>(half);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float2 u = *reinterpret_cast<const float2*>(half + w * WPIECE);
                t.x += u.x; t.y += u.y;
            }
            const int m = mw0 + row;
            if (m < Mlim) *reinterpret_cast<float2*>(p.stats_out + (long)tile_n * p.stats_slab + 2 * (long)m) = t;
        }
    }
}

#ifdef AG_REF_KERNELS   // the round-1/2 ring kernel: parity reference of the shipped kernel (libautognothi_hip_ref.so: tests only)
// VAR: 0 plain, 1 LayerNorm-folded consumer (ln_stats / ln_s), 2 row-statistics producer (stats_out), 3 producer whose
// residual is the LayerNorm of the stored pre-LN rows (BERT post-LN: ln_stats / rln_g / rln_b describe R)
template <int EPI, int VAR = 0, bool DBG = false>
__global__ __launch_bounds__(NT, 2) void gemm_ring_kernel(BigArgs pin) {
    BigArgs p = pin;
    p.M = ag_dyn_clamp(p.M, p.dyn);    // grid sized for the upper bound: workgroups beyond the actual tiles leave at once
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_n = (p.N + BT - 1) / BT, tiles_m = (p.M + BT - 1) / BT;
    const int nwg = tiles_m * tiles_n;
    if ((int)blockIdx.x >= nwg) return;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    // Tile order: N-tiles are walked in groups of `ngrp` columns, a whole group for every M-panel before the next
    // group, so the weight rows an XCD needs at a time (ngrp x 256 x K) stay resident in its 4 MiB L2 instead of being
    // re-fetched from the Infinity Cache for every few M-panels (fc1: W = 4.7 MB; measured fabric traffic 1.54 GB vs
    // 0.78 GB algorithmic before).  A panels are then fetched once per group.
    int tm, tn;
    {
        const int ngrp = p.ngrp;
        const int full = (tiles_n / ngrp) * ngrp * tiles_m;          // tiles inside complete groups
        if (wg < full) {
            const int g = wg / (ngrp * tiles_m), rem = wg - g * (ngrp * tiles_m);
            tm = rem / ngrp; tn = g * ngrp + rem % ngrp;
        } else {                                                      // the ragged last group
            const int w = tiles_n % ngrp, rem = wg - full;
            tm = rem / w; tn = (tiles_n / ngrp) * ngrp + rem % w;
        }
    }
    const int m0 = tm * BT, n0 = tn * BT;

    f32x4_t acc[4][8];  // [n sub-tile][m sub-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define AG_MARK(idx_)                                                                                     \
    if (DBG && (blockIdx.x == 0 || blockIdx.x == 777) && lane == 0) {                                          \
        unsigned long long t_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
        p.dbg[(((blockIdx.x ? 1 : 0) * 8 + wave) * 128 + (idx_)) * 8] = t_;                                    \
    }
    AG_MARK(120)
    const int nh = p.K / 32;  // half-steps
    // refills: SGPR tile base (+ jn*64 B on the scalar unit) + a loop-invariant per-lane byte offset (rows clamped at the
    // matrix edge): no address arithmetic on the vector unit inside the loop
    uint32_t offA[2], offW[2];
    {
        const int r_in = lane >> 2, chunk = (lane & 3) ^ swz4((lane >> 4) & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ra = (wave * 2 + i) * 16 + r_in, rw = ra;
            ra = m0 + ra < p.M ? ra : p.M - 1 - m0;
            rw = n0 + rw < p.N ? rw : p.N - 1 - n0;
            offA[i] = (uint32_t)(ra * (int)p.lda_b + chunk * 16);
            offW[i] = (uint32_t)(rw * (int)p.ldw_b + chunk * 16);
        }
    }
    const char* tileA = p.A + (long)m0 * p.lda_b;
    const char* tileW = p.W + (long)n0 * p.ldw_b;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t ldsA_w = lds0 + wave * 2048, ldsW_w = ldsA_w + HALF_OP_BYTES;   // this wave's piece pair inside a slot
    auto refill4 = [&](int jn, int slot) {  // this wave's A pieces 2w, 2w+1 and W pieces 2w, 2w+1 of half-step jn into ring slot `slot`
        glds16b_s_x4(tileA + jn * HROWB, tileW + jn * HROWB, offA[0], offW[0], offA[1], offW[1],
                     ldsA_w + slot * SLOT_BYTES, ldsW_w + slot * SLOT_BYTES);
    };
    // ---- tile constants: requested BEFORE the ring's first fill (so that they are the oldest loads in flight) and parked in the
    // LDS tail once the prologue's counted wait has let them land; the epilogue then needs nothing from memory but the residual
    constexpr bool ROWST = (VAR == 1 || VAR == 3);
    float c_a = 0.f, c_b = 0.f;
    f32x2v_t c_st[4] = {f32x2v_t{0.f, 0.f}, f32x2v_t{0.f, 0.f}, f32x2v_t{0.f, 0.f}, f32x2v_t{0.f, 0.f}};
    {
        int n = n0 + (tid & 255);
        n = n < p.N ? n : p.N - 1;
        if (tid < 256) {                                   // (wave-uniform: waves 0-3 / 4-7)
            if (p.bias) c_a = gload_f32_async(p.bias + n);
            if (VAR == 3) c_b = gload_f32_async(p.rln_b + n);
        } else {
            if (VAR == 1) c_a = gload_f32_async(p.ln_s + n);
            if (VAR == 3) c_a = gload_f32_async(p.rln_g + n);
        }
        if (ROWST && lane < 32) {   // wave (wm, wn) finishes rows [32 wn, 32 wn + 32) of its half: up to four slabs (H <= 1024) at once
            int m = m0 + wm * 128 + wn * 32 + lane;
            m = m < p.M ? m : p.M - 1;
            const float* sp = p.ln_stats + 2 * (long)m;
#pragma unroll
            for (int s_i = 0; s_i < 4; ++s_i) c_st[s_i] = gload_f32x2_async(sp + (s_i < p.ln_nslab ? s_i : 0) * p.stats_slab);
        }
    }
    // prologue: half-steps 0..3 into slots 0..3 (4 pieces per wave per half-step)
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
        if (j < nh) refill4(j, j);

    // ---- main loop: two wave groups half a step out of phase ---------------------------------------
    // Waves w and w+4 share a SIMD.  Group 0 (waves 0-3) and group 1 (waves 4-7) run the same program, but
    // group 1 starts one barrier late, so in every barrier-to-barrier phase one wave of each SIMD issues
    // MFMAs while its partner reads fragments from LDS and issues the LDS-DMA refills:
    //   global barrier #  2j        2j+1       2j+2        2j+3
    //   group 0          | reads(j) | MFMA(j)  | reads(j+1) | MFMA(j+1)
    //   group 1          | MFMA(j-1)| reads(j) | MFMA(j)    | reads(j+1)
    // Slot j is complete before barrier #2j (each wave waits, counted, for its own LDS-DMA pieces of it:
    // group 0 before its "a" barrier, group 1 before its "b" barrier of step j-1); slot (j-1)&3 is refilled
    // with half-step j+3 only after barrier #2j, when both groups have finished reading it.
    const int grp = wave >> 2;
    auto wait_ahead = [&](int ahead) {  // leave `ahead` newer half-steps (4 loads each) of this wave in flight
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    AG_MARK(121)
    wait_ahead(min(nh - 1, 3));                     // slot 0 (this wave's pieces) before barrier #0
    {   // everything older than slot 0 has landed too: the tile constants.  Into the LDS tail (read in the epilogue, many barriers on)
        gload_landed(c_a); gload_landed(c_b);
        char* const tail = smem + NSLOT * SLOT_BYTES;
        if (tid < 256) {
            *reinterpret_cast<float*>(tail + TAIL_C0 + (tid & 255) * 4) = c_a;
            if (VAR == 3) *reinterpret_cast<float*>(tail + TAIL_C2 + (tid & 255) * 4) = c_b;
        } else if (VAR == 1 || VAR == 3) {
            *reinterpret_cast<float*>(tail + TAIL_C1 + (tid & 255) * 4) = c_a;
        }
        if (ROWST) {
#pragma unroll
            for (int s_i = 0; s_i < 4; ++s_i) gload_landed(c_st[s_i]);
            if (lane < 32) {
                float sx = c_st[0].x, sq = c_st[0].y;   // slabs added in slab order: bit-reproducible
#pragma unroll
                for (int s_i = 1; s_i < 4; ++s_i) {
                    sx += s_i < p.ln_nslab ? c_st[s_i].x : 0.f; sq += s_i < p.ln_nslab ? c_st[s_i].y : 0.f;
                }
                if (p.ln_nslab > 4) {                   // wider rows (H > 1024: none shipped): plain loads, one round trip per slab
                    int m = m0 + wm * 128 + wn * 32 + lane;
                    m = m < p.M ? m : p.M - 1;
                    for (int s_i = 4; s_i < p.ln_nslab; ++s_i) {
                        const float2 w = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (long)m + s_i * p.stats_slab);
                        sx += w.x; sq += w.y;
                    }
                }
                const float mean = sx * p.ln_inv_h;
                const float rstd = rsqrtf(fmaxf(sq * p.ln_inv_h - mean * mean, 0.f) + p.ln_eps);
                *reinterpret_cast<float2*>(tail + TAIL_STAT + (wm * 128 + wn * 32 + lane) * 8) = make_float2(mean, rstd);
            }
        }
    }
    AG_MARK(122)
    if (grp == 1) asm volatile("s_barrier" ::: "memory");
    // this wave's LDS-DMA pieces: 2 of A, 2 of W per half-step
#define AG_STAMP(slot_)                                                                                   \
        if (DBG && (blockIdx.x == 0 || blockIdx.x == 777) && lane == 0) {                                      \
            unsigned long long t_;                                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
            p.dbg[(((blockIdx.x ? 1 : 0) * 8 + wave) * 128 + j) * 8 + (slot_)] = t_;                           \
        }
    // one half-step; `slot`, `do_refill`, `ahead` are literals at every call site below (the steady state is unrolled by the
    // ring length), so the slot addresses fold into instruction offsets and the loop carries no compare/branch/select
    // overhead: every scalar instruction here sits on the critical read phase of one wave group.
    auto half_step = [&](const int j, const int slot, const bool do_refill, const int ahead) {
        AG_STAMP(0)
        asm volatile("s_barrier" ::: "memory");                        // "a": slot j is complete and visible
        AG_STAMP(1)
        // ---- read phase (the SIMD partner wave is in its MFMA phase) ----
        const char* sA = smem + slot * SLOT_BYTES;
        const char* sW = sA + HALF_OP_BYTES;
        uint4 fw[4], fx[8];
#pragma unroll
        for (int s = 0; s < 4; ++s) fw[s] = frag_half(sW, wn * 64 + s * 16, lane);
#pragma unroll
        for (int s = 0; s < 8; ++s) fx[s] = frag_half(sA, wm * 128 + s * 16, lane);
        if (do_refill) refill4(j + 3, (slot + 3) & 3);                  // into the slot read one half-step ago
        AG_STAMP(2)
        __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0): my fragments are in registers (the builtin,
        asm volatile("" ::: "memory");                                  // so hipcc does not add its own per-MFMA lgkmcnt waits)
        // Slot j+1 must be complete before barrier #2j+2 (group 0's next "a", group 1's "b" below).  Waiting
        // for it here keeps the MFMA phase free of waits; group 0 is one barrier early, which costs nothing:
        // those pieces were issued two iterations ago.  Newer than j+1 at this point: j+2, j+3 (`ahead` of them).
        if (ahead >= 0) wait_ahead(ahead);
        AG_STAMP(3)
        asm volatile("s_barrier" ::: "memory");                        // "b"
        AG_STAMP(4)
        // ---- MFMA phase: nothing but the 32 MFMAs (splitting the refills 2/2 across the phases measured slower) ----
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int sn = 0; sn < 4; ++sn)
#pragma unroll
            for (int sm = 0; sm < 8; ++sm)
                acc[sn][sm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[sn]),
                                                                      __builtin_bit_cast(bf16x8_t, fx[sm]), acc[sn][sm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        AG_STAMP(5)
    };
    if (!DBG && (nh & 3) == 0 && nh >= 8) {
        half_step(0, 0, false, 2);                                   // the prologue already filled slots 0..3
        int j = 1;
        for (; j + 7 <= nh; j += 4) {                                 // steady state: half-steps 1 .. nh-4 request j+3 (<= nh-1), 2 ahead
            half_step(j, 1, true, 2);
            half_step(j + 1, 2, true, 2);
            half_step(j + 2, 3, true, 2);
            half_step(j + 3, 0, true, 2);
        }
        // (nh - 4) % 4 == 0: the loop stops at j == nh - 3; the last three half-steps request nothing and drain the ring
        half_step(j, 1, false, 1);
        half_step(j + 1, 2, false, 0);
        half_step(j + 2, 3, false, -1);
    } else {
        for (int j = 0; j < nh; ++j)
            half_step(j, j & 3, j >= 1 && j + 3 < nh, j + 1 < nh ? min(nh - 2 - j, 2) : -1);
    }
#undef AG_STAMP
    if (grp == 0) asm volatile("s_barrier" ::: "memory");              // pairs group 1's extra first barrier

    // ---- epilogue ----
    AG_MARK(123)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (already true: the last half-step waited for 0)
    wave_epilogue<EPI, VAR == 1, VAR == 2 || VAR == 3, VAR == 3>(p, acc, m0 + wm * 128, n0 + wn * 64, smem + wave * 16384, lane, smem, tn);
    AG_MARK(124)
    if (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    AG_MARK(125)
#undef AG_MARK
}

#endif  // AG_REF_KERNELS

// =====================================================================================================================
// gemm_line_kernel — the same tile, waves, phases and epilogue, fed by WHOLE 128-byte cache lines.
//
// What was measured (round 3, tools/dma_probe.py = ag_probe_dma; experiment builds recorded in profiles/HISTORY.md §9): the ring above stages 16-row x 64-byte
// pieces (K = 32 per half-step).  A 64-byte row segment is half a 128-byte line; the other half is requested one half-step
// later, after 32 KiB of other lines have passed through the CU's 32 KiB vector L1, so EVERY line crosses the L2 -> L1 path
// twice.  With all 256 CUs pulling, that path delivers 28 B/clk/CU of such pieces (a 32 KiB half-step = 1 170 cycles, more
// than the 1 024 cycles its 256 MFMAs take: the kernel was feed-bound, and removing the loads altogether made it 33 % faster)
// against 47 B/clk/CU for 8-row x 128-byte pieces (700 cycles).  So: K is walked in 64-element steps, a step image is
// A[256 x 128 B] + W[256 x 128 B] = 64 KiB, a piece is 8 rows x 128 B (one line per row, each line requested once per CU),
// the ring has two step slots (LDS: 128 KiB + the 5 KiB tail, as before).
//   LDS image: row-major 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7) (applied on the SOURCE
//   address of the LDS-DMA, whose destination is linear, and on the ds_read_b128): the 16 lanes of a fragment read (16 rows,
//   one k-chunk) hit 16 different 16-byte bank groups.  Layout [A slot 0 | A slot 1 | W slot 0 | W slot 1] so that one base
//   VGPR per operand and k-half reaches both slots through the 16-bit ds offset.
//   Phases: unchanged (two wave groups half a phase apart, read phase / MFMA phase of 32 MFMAs, K = 32 per phase pair).
//   Refills: step s+1 is requested during step s into the slot step s-1 was read from, i.e. less than one step ahead, so
//   (1) the requests are placed by group so that every piece has at least one full phase to land: group 0's waves stage A, group
//   1's stage W, eight pieces each; group 0 (which meets its next "a" barrier after an MFMA phase) requests four pieces in its
//   read-lo and four in its read-hi phase and waits after its MFMA-hi phase; group 1 (which meets it after a read phase) requests
//   all eight in read-lo and waits at the end of read-hi; per barrier interval the CU's texture-address unit sees 16 / 32 / 16 / 0
//   pieces;
//   (2) HBM latency is taken by a PREFETCH of the A stream (W stays L2-resident: every CU of an XCD re-reads the same few
//   panels): one LDS-DMA dword per group-0 wave and step touches one dword of each of the wave's 64 A lines of step s+2,
//   pulling them into the XCD's L2 a step before their pieces are requested; it lands in a 256-byte LDS sink (a register
//   destination would be written asynchronously, after the compiler has re-used it) and is the youngest vector-memory
//   operation of the wave when the step's pieces are waited for (vmcnt(1)).  Measured (M = 302 592, same box): sum of the four
//   encoder GEMMs 4 230 us without, 4 105 us with it (fc2 1 279 -> 1 221 us); for both operands 4 262; distances 1 / 3 / 4: 4 301 /
//   4 177 / 4 163.
//   What did NOT work here, each built and measured: requests issued from inside the MFMA phases, spread one per eight MFMAs
//   (+12 %: a vector-memory issue stalls the wave's MFMA issue for much longer than one MFMA's shadow); group 0 requesting all
//   eight pieces in its first read phase for the longest lookahead (+8 %: 32 pieces in one barrier interval); the same number
//   of pieces in every read phase (the last interval's pieces then have no time to land).
constexpr int LROWB = 128;                       // bytes of K per row per step (64 bf16)
constexpr int LOP_BYTES = BT * LROWB;            // 32 KiB per operand per step
constexpr int LW_BASE = 2 * LOP_BYTES;           // W slots behind the two A slots
// what measured best (round 3, profiles/HISTORY.md §9): L2 prefetch ON, for the A stream only (group 0), two steps ahead
constexpr int PF_AHEAD = 2;                      // L2 prefetch distance in steps

// four 1 KiB pieces of one operand (SGPR base + per-lane offsets) to four consecutive LDS kilobytes
__device__ __forceinline__ void glds_x4(const char* base, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, uint32_t lds0) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, %6, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, %6, 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, %6, 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(lds0) : "memory", "scc");
}
// the same four pieces as two piece pairs: pieces 0, 1 from base0 and pieces 2, 3 from base1 (= 16 rows further), per-lane offsets o0 / o1
__device__ __forceinline__ void glds_x4b(const char* base0, const char* base1, uint32_t o0, uint32_t o1, uint32_t lds0) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_add_u32 m0, %5, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_add_u32 m0, %5, 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                 "s_add_u32 m0, %5, 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "v"(o1), "s"(base0), "s"(base1), "s"(lds0) : "memory", "scc");
}
// one piece (the wave-uniform operands are forced into SGPRs: behind a group-dependent branch the compiler no longer proves them uniform)
__device__ __forceinline__ void glds_x1(const char* base_, uint32_t o, uint32_t lds_) {
    const uint64_t bv = (uint64_t)(uintptr_t)base_;
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv);
    const char* base = (const char*)(uintptr_t)bu;
    const uint32_t lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o), "s"(base), "s"(lds) : "memory");
}
// touch one dword per lane (L2 prefetch of that lane's line).  The data must go SOMEWHERE: a plain load's destination VGPR would
// be written whenever the load returns, long after the compiler has given that register to something else (first version of
// this kernel: memory faults from clobbered address registers) — so it is an LDS-DMA into a 256-byte sink behind the tail.
constexpr int PF_SINK = NSLOT * SLOT_BYTES + TAIL_BYTES;
constexpr int LINE_LDS_BYTES = PF_SINK + 256;
__device__ __forceinline__ void l2_touch(const char* base, uint32_t voff, uint32_t lds_sink) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_sink) : "memory");
}

#ifdef AG_REF_KERNELS   // one tile per workgroup on the whole-line feed: A/B + parity reference of the stream kernel (tests only)
template <int EPI, int VAR = 0>
__global__ __launch_bounds__(NT, 2) void gemm_line_kernel(BigArgs pin) {
    BigArgs p = pin;
    p.M = ag_dyn_clamp(p.M, p.dyn);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int grp = wave >> 2, gw = wave & 3;         // wave group (phase offset) and index inside it

    const int tiles_n = (p.N + BT - 1) / BT, tiles_m = (p.M + BT - 1) / BT;
    const int nwg = tiles_m * tiles_n;
    if ((int)blockIdx.x >= nwg) return;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    int tm, tn;
    {   // tile order: as gemm_ring_kernel
        const int ngrp = p.ngrp;
        const int full = (tiles_n / ngrp) * ngrp * tiles_m;
        if (wg < full) {
            const int g = wg / (ngrp * tiles_m), rem = wg - g * (ngrp * tiles_m);
            tm = rem / ngrp; tn = g * ngrp + rem % ngrp;
        } else {
            const int w = tiles_n % ngrp, rem = wg - full;
            tm = rem / w; tn = (tiles_n / ngrp) * ngrp + rem % w;
        }
    }
    const int m0 = tm * BT, n0 = tn * BT;
    if (p.dbg && tid == 0) {      // timeline diagnostic (AG_GEMM_DBG, tools/gemm_timeline.py): workgroup start, 100 MHz clock + hardware id
        unsigned long long t_; uint32_t hw_;
        asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(hw_)::"memory");
        p.dbg[8 * (long)b + 0] = t_; p.dbg[8 * (long)b + 3] = hw_;
    }

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int ns = p.K / 64;                          // steps
    const char* tileA = p.A + (long)m0 * p.lda_b;
    const char* tileW = p.W + (long)n0 * p.ldw_b;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // ---- this wave's pieces: group 0 (waves 0-3) stages A, group 1 (waves 4-7) stages W; wave g of a group owns rows
    // [64 g, 64 g + 64) of its operand = 8 pieces of 8 rows.  Piece q = 2 k + par: rows 64 g + 16 k + 8 par + (lane >> 3); its
    // per-lane source offset is vb[par] + k * 16 rows, so two loop-invariant VGPRs serve all eight pieces (the register budget
    // of this kernel is what decides its structure: 128 accumulators + 48 fragment registers of 256, and a single spilled value
    // inside the loop would make hipcc drain the hand-counted LDS-DMA queue with a vmcnt(0) of its own).  Rows past the matrix
    // edge are clamped to the last valid row by a v_min against `off_max`.
    const bool stA = grp == 0;        // (the other way round — group 1 stages A, all eight pieces two phases ahead — measured +1 %)
    const char* const tileX = stA ? tileA : tileW;
    const int ldx = stA ? (int)p.lda_b : (int)p.ldw_b;
    const int vrows = stA ? min(BT, p.M - m0) : min(BT, p.N - n0);           // valid rows of this tile's operand
    const uint32_t off_max = (uint32_t)((vrows - 1) * ldx + 112);
    const uint32_t d16 = (uint32_t)(16 * ldx);
    uint32_t vb[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = gw * 64 + par * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);               // source chunk that lands at linear position lane & 7
        vb[par] = (uint32_t)(row * ldx + ch * 16);
    }
    const uint32_t ldsX_w = lds0 + (stA ? 0 : LW_BASE) + gw * 8192;         // + slot * LOP_BYTES
    // pieces 4 h .. 4 h + 3 (h = 0, 1) of step `step` into ring slot `slot`.  Interior tiles (all 256 rows valid): the 16-row
    // advance of a piece pair goes into the SGPR base (scalar adds), the two loop-invariant per-lane offsets are used as they
    // are: no vector instruction on the request path.  Edge tiles: per-lane offsets made and clamped here.
    const bool edge = vrows < BT;
    auto refill4 = [&](int step, int slot, int h) {
        const uint32_t lds = ldsX_w + slot * LOP_BYTES + h * 4096;
        if (!edge) {
            const char* b0 = tileX + (long)step * LROWB + (long)(2 * h) * d16;
            glds_x4b(b0, b0 + d16, vb[0], vb[1], lds);
        } else {
            uint32_t d = d16, om = off_max;
            asm volatile("" : "+s"(d), "+s"(om));   // (made here, four short-lived temporaries: not hoisted into registers that do not exist)
            uint32_t o0 = vb[0] + (2 * h) * d, o1 = vb[1] + (2 * h) * d, o2 = vb[0] + (2 * h + 1) * d, o3 = vb[1] + (2 * h + 1) * d;
            o0 = min(o0, om); o1 = min(o1, om); o2 = min(o2, om); o3 = min(o3, om);      // rows past the edge -> the last valid row
            glds_x4(tileX + (long)step * LROWB, o0, o1, o2, o3, lds);
        }
    };
    // piece q (0..7) of step `step` into ring slot `slot`, by itself (issued between MFMAs)
    auto piece1 = [&](int step, int slot, int q) {
        const uint32_t lds = ldsX_w + slot * LOP_BYTES + q * 1024;
        if (!edge) {
            glds_x1(tileX + (long)step * LROWB + (long)(q >> 1) * d16, vb[q & 1], lds);
        } else {
            uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d16), om = (uint32_t)__builtin_amdgcn_readfirstlane((int)off_max);
            asm volatile("" : "+s"(d), "+s"(om));
            glds_x1(tileX + (long)step * LROWB, min(vb[q & 1] + (q >> 1) * d, om), lds);
        }
    };
    // L2 prefetch: lane l of wave g touches the line of row 64 g + l of the wave's operand
    auto prefetch = [&](int step) {
        {   // past the last step: the last step's lines again (an L2 hit) — the waits below COUNT this request (vmcnt(1))
            step = step < ns ? step : ns - 1;
            int lx = ldx;
            uint32_t om = off_max - 112u, ones = ~0u;
            asm volatile("" : "+s"(lx), "+s"(om), "+s"(ones));
            const int ln = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));   // lane id, re-made here: no live register
            const uint32_t po = min((uint32_t)((gw * 64 + ln) * lx), om);
            l2_touch(tileX + (long)step * LROWB, po, lds0 + PF_SINK);
        }
    };
    // fragment read addresses: row r = lane & 15, k-chunk c = lane >> 4 (+ 4 for the upper K half = ^ 64 bytes after the swizzle)
    const int fr = lane & 15, fc = lane >> 4;
    const uint32_t frag_lo = (uint32_t)(fr * LROWB + ((fc ^ ((fr >> 1) & 7)) << 4));
    uint32_t vA = lds0 + frag_lo + wm * (128 * LROWB), vW = lds0 + frag_lo + LW_BASE + wn * (64 * LROWB);
    // (opaque to the optimiser: otherwise it re-associates LW_BASE with the slot offset into constants beyond the 16-bit ds offset
    // field and keeps a separate address register per fragment)
    asm volatile("" : "+v"(vA), "+v"(vW));

    // ---- tile constants (as gemm_ring_kernel): requested first, parked in the LDS tail once the prologue's wait has let them land
    constexpr bool ROWST = (VAR == 1 || VAR == 3);
    float c_a = 0.f, c_b = 0.f;
    f32x2v_t c_st[4] = {f32x2v_t{0.f, 0.f}, f32x2v_t{0.f, 0.f}, f32x2v_t{0.f, 0.f}, f32x2v_t{0.f, 0.f}};
    {
        int n = n0 + (tid & 255);
        n = n < p.N ? n : p.N - 1;
        if (tid < 256) {
            if (p.bias) c_a = gload_f32_async(p.bias + n);
            if (VAR == 3) c_b = gload_f32_async(p.rln_b + n);
        } else {
            if (VAR == 1) c_a = gload_f32_async(p.ln_s + n);
            if (VAR == 3) c_a = gload_f32_async(p.rln_g + n);
        }
        if (ROWST && lane < 32) {
            int m = m0 + wm * 128 + wn * 32 + lane;
            m = m < p.M ? m : p.M - 1;
            const float* sp = p.ln_stats + 2 * (long)m;
#pragma unroll
            for (int s_i = 0; s_i < 4; ++s_i) c_st[s_i] = gload_f32x2_async(sp + (s_i < p.ln_nslab ? s_i : 0) * p.stats_slab);
        }
    }
    // ---- prologue: L2 prefetch of step 2, then steps 0 and 1 into slots 0 and 1
    if (stA) prefetch(2);
    refill4(0, 0, 0); refill4(0, 0, 1);
    refill4(1, 1, 0); refill4(1, 1, 1);                 // (ns >= 2)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    {   // everything older than step 0's pieces has landed too: the tile constants -> LDS tail
        gload_landed(c_a); gload_landed(c_b);
        char* const tail = smem + NSLOT * SLOT_BYTES;
        if (tid < 256) {
            *reinterpret_cast<float*>(tail + TAIL_C0 + (tid & 255) * 4) = c_a;
            if (VAR == 3) *reinterpret_cast<float*>(tail + TAIL_C2 + (tid & 255) * 4) = c_b;
        } else if (VAR == 1 || VAR == 3) {
            *reinterpret_cast<float*>(tail + TAIL_C1 + (tid & 255) * 4) = c_a;
        }
        if (ROWST) {
#pragma unroll
            for (int s_i = 0; s_i < 4; ++s_i) gload_landed(c_st[s_i]);
            if (lane < 32) {
                float sx = c_st[0].x, sq = c_st[0].y;
#pragma unroll
                for (int s_i = 1; s_i < 4; ++s_i) {
                    sx += s_i < p.ln_nslab ? c_st[s_i].x : 0.f; sq += s_i < p.ln_nslab ? c_st[s_i].y : 0.f;
                }
                if (p.ln_nslab > 4) {
                    int m = m0 + wm * 128 + wn * 32 + lane;
                    m = m < p.M ? m : p.M - 1;
                    for (int s_i = 4; s_i < p.ln_nslab; ++s_i) {
                        const float2 w = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (long)m + s_i * p.stats_slab);
                        sx += w.x; sq += w.y;
                    }
                }
                const float mean = sx * p.ln_inv_h;
                const float rstd = rsqrtf(fmaxf(sq * p.ln_inv_h - mean * mean, 0.f) + p.ln_eps);
                *reinterpret_cast<float2*>(tail + TAIL_STAT + (wm * 128 + wn * 32 + lane) * 8) = make_float2(mean, rstd);
            }
        }
    }
    if (grp == 1) asm volatile("s_barrier" ::: "memory");

    // one K = 32 half of step `s` (KH = 0 lower / 1 upper K half) from ring slot SLOT.  What is requested / waited for in it
    // depends on the group and the half (see the header); all of SLOT, KH, REFILL are literals at the call sites.
    auto half = [&](const int s, const int slot, const int kh, const bool refill, const bool last) {
        asm volatile("s_barrier" ::: "memory");                        // "a"
        // requests first (their address temporaries die before the fragments arrive): group 0 four pieces in each half, group 1 all
        // eight in the lower half
        if (refill) {
            if (grp == 0) refill4(s + 1, slot ^ 1, kh);
            else if (kh == 0) { refill4(s + 1, slot ^ 1, 0); refill4(s + 1, slot ^ 1, 1); }
        }
        if (kh == 1 && stA && !last) prefetch(s + PF_AHEAD);
        uint32_t x64 = kh ? 64u : 0u;
        asm volatile("" : "+s"(x64));                                  // (made here, two temporaries: not hoisted into two more registers)
        typedef __attribute__((address_space(3))) const char* lds_cptr;
        typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const u32x4v* lds_u4ptr;
        const lds_cptr pa = (lds_cptr)(uintptr_t)(vA ^ x64) + slot * LOP_BYTES;      // (an LDS address is the byte offset itself)
        const lds_cptr pw = (lds_cptr)(uintptr_t)(vW ^ x64) + slot * LOP_BYTES;
        u32x4v fw[4], fx[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[i] = *(lds_u4ptr)(pw + i * (16 * LROWB));
#pragma unroll
        for (int i = 0; i < 8; ++i) fx[i] = *(lds_u4ptr)(pa + i * (16 * LROWB));
        __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0)
        asm volatile("" ::: "memory");
        // group 1 meets the barrier that opens step s+1 at the end of THIS read phase (upper half): its pieces of step s+1
        // (requested two phases ago) must have landed; the prefetch just issued stays in flight
        if (grp == 1 && kh == 1 && !last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (last step: nothing was requested)
        asm volatile("s_barrier" ::: "memory");                        // "b"
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int sn = 0; sn < 4; ++sn)
#pragma unroll
            for (int sm = 0; sm < 8; ++sm)
                acc[sn][sm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[sn]),
                                                                      __builtin_bit_cast(bf16x8_t, fx[sm]), acc[sn][sm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // group 0 meets that barrier after this MFMA phase (upper half): its 8 pieces of step s+1 had at least a phase to land
        if (grp == 0 && kh == 1 && !last) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    };
    auto step = [&](const int s, const int slot, const bool refill, const bool last) {
        half(s, slot, 0, refill, last);
        half(s, slot, 1, refill, last);
    };
    // ns is even (K % 128 == 0: the launcher's condition), so the walk is straight-line code around one loop — no branch joins
    // with the 128 accumulator registers live, which is what lets the register allocator keep them in place
    if (p.dbg && tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); p.dbg[8 * (long)b + 4] = t_; }
    step(0, 0, false, false);                                          // the prologue requested step 1 already
    int s = 1;
    for (; s + 1 < ns; s += 2) {                                       // steps 1 .. ns-2: each has a successor to request
        step(s, 1, true, false);
        step(s + 1, 0, true, false);
    }
    step(s, 1, false, true);                                           // s == ns - 1: nothing left to request or to wait for
    if (p.dbg && tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); p.dbg[8 * (long)b + 5] = t_; }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");              // pairs group 1's extra first barrier

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_epilogue<EPI, VAR == 1, VAR == 2 || VAR == 3, VAR == 3>(p, acc, m0 + wm * 128, n0 + wn * 64, smem + wave * 16384, lane, smem, tn);
    if (p.dbg) {                  // timeline diagnostic: epilogue issued / this wave's stores acknowledged (waves 0 and 7)
        unsigned long long t1, t2;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        if (tid == 0) { p.dbg[8 * (long)b + 1] = t1; p.dbg[8 * (long)b + 2] = t2; }
    }
}

#endif  // AG_REF_KERNELS

// gemm_stream_kernel — gemm_line_kernel as ONE request stream per CU.
//
// What tools/gemm_timeline.py shows for the whole-line kernel (M = 302 592): a K = 768 tile lives 2.1 us of prologue (cold first step
// image) + 18-19 us of main loop + 3.5-5 us (bias / GELU) or 12-14 us (residual) of epilogue, and its CU then waits 1-2.3 us for the
// successor workgroup: the matrix cores run in 64-72 % of a tile's period.  The epilogue needs the accumulators and cannot overlap
// with the next tile in one workgroup; the prologue and the turn-over can: here a workgroup stays on its CU (grid = CU count, tiles
// b, b + grid, ... — the same tiles on the same XCD as the hardware dispatch would give it) and the LAST step of a tile requests
// step 0 of the NEXT tile into the ring slot that step ns-2 has released, exactly as any step requests its successor; the L2
// prefetch runs two steps ahead across the tile boundary the same way.  When the epilogue starts, the next tile's first step image
// is in LDS (slot 0), so the epilogue's staging pieces live in slot 1 (wave w's piece = the 8 KiB its own next LDS-DMA pieces go
// to); after the epilogue: one barrier, tile constants, step 1's requests, and step 0 computes at once.
// LDS behind the ring: two tails and two raw-partials areas (tiles alternate: the next tile's constants arrive while this tile's epilogue
// reads its own), then the sink of the L2 touches
// Round 6: the UPPER K half's twelve fragments are read from between the lower half's MFMAs into a second fragment set, so the upper
// half's request phase holds requests only (1) — same request schedule, same waits, bit-identical results; -0.3 ... -1.2 % per launch on
// the four encoder GEMMs, same box (profiles/HISTORY.md §12).  0 = every half's fragments in its own request phase (rounds 3-5; A/B via
// tools/build_variant.sh -DAG_STREAM_HOIST_HI=0).  The MFMAs of both halves go through asm with the accumulator TIED (vdst = src2): left
// to itself hipcc picks the untied form for the software-pipelined half, every accumulator moving to another register quad at each MFMA —
// sixteen registers more in flight and spills inside the loop (§12); tied, both fragment sets fit in 211-229 registers.
#ifndef AG_STREAM_HOIST_HI
#define AG_STREAM_HOIST_HI 1
#endif
constexpr int STREAM_TAIL = NSLOT * SLOT_BYTES;                  // + par * TAIL_BYTES
constexpr int STREAM_RAW = STREAM_TAIL + 2 * TAIL_BYTES;         // + par * 8192: 8 waves x 4 slabs x (32 sums | 32 sums of squares)
constexpr int STREAM_SINK = STREAM_RAW + 2 * 8192;
constexpr int STREAM_LDS_BYTES = STREAM_SINK + 256;
// RLDS (bias + residual epilogue with an identity residual row map, share == 1): the RESIDUAL tile comes through LDS.  Read on demand in
// the epilogue (accumulator-layout or whole-line loads into registers) it costs 7 us of a 13 us epilogue with the matrix cores idle —
// what one CU gets out of cold vector loads (profiles/HISTORY.md) — and no placement of those loads changes that.  LDS-DMA pieces do
// twice that rate on cold lines, need no registers, and can start before the main loop ends: the last step requests rows [0, 64) of
// each wave's 128 x 64 sub-tile of the residual into the wave's own 8 KiB piece area of ring slot 0 (where a normal step would put its
// successor), the epilogue requests rows [64, 128) into its area of slot 1 once slot 1 has been read; a half is then added in place
// (ds_read_b64 in the accumulator layout, chunk-swizzled image), read back as whole 128-byte rows and stored; the next tile's step 0
// goes into the slot-0 area as soon as the first half has left it.  Everything is private to the wave.  (L2 touches of the residual lines
// a step ahead of the pieces, from behind the waves' last wait: +6 % on the out-projection — one line per lane is the dearest request shape.)
// HT (round 6, half-height tail): the tiles left over after the last COMPLETE round of the resident workgroups (at most half a round of
// them) run as two 128-row units each, side by side in that last round, instead of as one more whole round of 256^2 tiles on a part of the
// chip: wave group 0 (rows [0, 128) of a tile) works as in any tile — it stages the 128 A rows it needs (waves 0, 1; waves 2, 3 request
// nothing), reads its fragments, computes, runs the epilogue — and wave group 1 (rows [128, 256)) only stages W and keeps the barriers:
// waves w and w + 4 share a SIMD, so every SIMD issues half the MFMAs of a whole tile per step.  Every output element is the same sum in the
// same order as in a whole tile (bit-identical: tests/test_gpu_gemm_ring.py).  Units b >= half_from: tile half_from + (k & 7) + 8 (k >> 4),
// half (k >> 3) & 1 with k = b - half_from — the two halves of a tile on workgroups 8 apart, i.e. on the same XCD.  A half-height unit is
// always the last unit of its workgroup.
template <int EPI, int VAR = 0, bool RLDS = false, bool HT = false>
__global__ __launch_bounds__(NT, 2) void gemm_stream_kernel(BigArgs pin) {
    BigArgs p = pin;
    p.M = __builtin_amdgcn_readfirstlane(ag_dyn_clamp(p.M, p.dyn));   // (a scalar: everything derived from it — tile counts, edges, has_next — stays in SGPRs)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int grp = wave >> 2, gw = wave & 3;         // wave group (phase offset) and index inside it
    const int tiles_n = (p.N + BT - 1) / BT, tiles_m = (p.M + BT - 1) / BT;
    // BATCH (fp32-output instantiation only): p.nbatch products of this shape side by side, unit = (product, tile)
    constexpr bool BATCH = (EPI == AG_EPI_BIAS_F32 && VAR == 0);
    const int nwg_t = tiles_m * tiles_n;
    const int nwg = BATCH ? nwg_t * p.nbatch : nwg_t;
    const int nres = (int)gridDim.x;                  // resident workgroups: a multiple of 8 (or all tiles)
    const int nunits = HT ? p.half_from + 2 * ((p.ntail + 7) & ~7) : nwg;
    int bt = blockIdx.x;
    if (bt >= nunits) return;
    const int ns = p.K / 64;                          // steps per tile (even)
    const bool stA = grp == 0;                        // group 0 stages A, group 1 stages W (see gemm_line_kernel)
    const int ldx = stA ? (int)p.lda_b : (int)p.ldw_b;
    const uint32_t d16 = (uint32_t)(16 * ldx);
    uint32_t vb[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = gw * 64 + par * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
        vb[par] = (uint32_t)(row * ldx + ch * 16);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t ldsX_w = lds0 + (stA ? 0 : LW_BASE) + gw * 8192;         // + slot * LOP_BYTES
    const int fr = lane & 15, fc = lane >> 4;
    const uint32_t frag_lo = (uint32_t)(fr * LROWB + ((fc ^ ((fr >> 1) & 7)) << 4));
    uint32_t vA = lds0 + frag_lo + wm * (128 * LROWB), vW = lds0 + frag_lo + LW_BASE + wn * (64 * LROWB);
    asm volatile("" : "+v"(vA), "+v"(vW));

    // a tile of the stream: scalars only
    // (half: 0 a whole tile, 1 / 2 the lower / upper 128 rows of one; mhi: first row of the matrix the unit does NOT own; skip: nothing to do)
    struct Tile { int m0, n0, tn; const char* x; uint32_t off_max; int edge; long cz; int half, mhi, skip; };     // (whole words only: never copied through memory)
    auto tile_of = [&](int b) {
        int half = 0;
        if (HT && b >= p.half_from) {
            const int k = b - p.half_from;
            b = p.half_from + (k & 7) + ((k >> 4) << 3);
            half = 1 + ((k >> 3) & 1);
        }
        const bool beyond = HT && b >= nwg;           // (the padding of the tail to a multiple of 8 tiles)
        if (beyond) b = nwg - 1;
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        int bz = 0;
        if (BATCH && p.nbatch > 1) { bz = wg / nwg_t; wg -= bz * nwg_t; }
        int tm, tn;
        const int ngrp = p.ngrp;
        const int full = (tiles_n / ngrp) * ngrp * tiles_m;
        if (wg < full) {
            const int g = wg / (ngrp * tiles_m), rem = wg - g * (ngrp * tiles_m);
            tm = rem / ngrp; tn = g * ngrp + rem % ngrp;
        } else {
            const int w = tiles_n % ngrp, rem = wg - full;
            tm = rem / w; tn = (tiles_n / ngrp) * ngrp + rem % w;
        }
        Tile t;
        t.m0 = __builtin_amdgcn_readfirstlane(tm * BT + ((HT && half == 2) ? BT / 2 : 0)); t.n0 = __builtin_amdgcn_readfirstlane(tn * BT); t.tn = __builtin_amdgcn_readfirstlane(tn);
        const int trows = (HT && half) ? BT / 2 : BT;                           // rows of A the unit needs
        t.half = 0; t.mhi = 0; t.skip = 0;
        if constexpr (HT) {
            t.half = __builtin_amdgcn_readfirstlane(half);
            t.mhi = __builtin_amdgcn_readfirstlane(min(p.M, t.m0 + trows));
            t.skip = __builtin_amdgcn_readfirstlane((beyond || t.m0 >= p.M) ? 1 : 0);   // (the upper half of a ragged last row tile of <= 128 rows)
            if (t.skip) { t.m0 = 0; t.mhi = 0; }                                // (addresses stay inside the matrix)
        }
        const int vrows = stA ? min(trows, p.M - t.m0) : min(BT, p.N - t.n0);
        t.x = stA ? p.A + (long)t.m0 * p.lda_b : p.W + (long)t.n0 * p.ldw_b;
        t.cz = 0;
        if (BATCH) {
            bz = __builtin_amdgcn_readfirstlane(bz);
            t.x += (long)bz * p.bk_b;
            t.cz = (long)bz * p.bc;
        }
        t.off_max = (uint32_t)((vrows - 1) * ldx + 112);
        t.edge = vrows < (stA ? trows : BT) ? 1 : 0;
        return t;
    };
    auto refill4 = [&](const Tile& t, int step, int slot, int h) {
        if (HT && t.half && stA && gw >= 2) return;          // rows [128, 256) of a half-height unit's A image: nobody reads them
        const uint32_t lds = ldsX_w + slot * LOP_BYTES + h * 4096;
        if (!t.edge) {
            const char* b0 = t.x + (long)step * LROWB + (long)(2 * h) * d16;
            glds_x4b(b0, b0 + d16, vb[0], vb[1], lds);
        } else {
            uint32_t d = d16, om = t.off_max;
            asm volatile("" : "+s"(d), "+s"(om));
            uint32_t o0 = vb[0] + (2 * h) * d, o1 = vb[1] + (2 * h) * d, o2 = vb[0] + (2 * h + 1) * d, o3 = vb[1] + (2 * h + 1) * d;
            o0 = min(o0, om); o1 = min(o1, om); o2 = min(o2, om); o3 = min(o3, om);
            glds_x4(t.x + (long)step * LROWB, o0, o1, o2, o3, lds);
        }
    };
    auto prefetch = [&](const Tile& t, int step) {     // lane l of wave g touches the line of row 64 g + l of its operand's step image
        int lx = ldx;
        uint32_t om = t.off_max - 112u, ones = ~0u;
        asm volatile("" : "+s"(lx), "+s"(om), "+s"(ones));
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        const uint32_t po = min((uint32_t)((gw * 64 + ln) * lx), om);
        l2_touch(t.x + (long)step * LROWB, po, lds0 + STREAM_SINK);
    };
    // RLDS: four 8-row pieces (j = 4 q .. 4 q + 3) of half h (rows [64 h, 64 h + 64)) of this wave's residual sub-tile -> its piece
    // area of ring slot h.  Image: 128-byte rows, 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7) — the main loop's swizzle: the
    // accumulator-layout ds_read_b64 / ds_write_b64 of 16 rows x 8 bytes then touch every bank once (rows 2j, 2j + 1 share a chunk position
    // and differ in the 128-byte half of the bank space).  (Round 3 keyed the swizzle on r & 7: rows r and r + 8 of every 16-row access met in
    // the same banks — the 1 536 conflict cycles per tile of profiles/r04_a_pmc_summary.)  Lane l fetches what belongs at chunk position l & 7 of
    // row l >> 3 of piece i: logical chunk (l & 7) ^ (4 (i & 1) + (l >> 4)).  Rows past M / columns past N are clamped to valid addresses
    // (never stored).
    auto res4 = [&](const Tile& t, const int h, const int q) {
        uint32_t ones = ~0u;
        int ldr = (int)p.ldr, mlast = p.M - 1, nlast = p.N - 8;
        asm volatile("" : "+s"(ones), "+s"(ldr), "+s"(mlast), "+s"(nlast));
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        const int ch_e = (ln & 7) ^ (ln >> 4);                                  // pieces 0, 2 (image rows 8 i + (l >> 3): key (4 i + (l >> 4)) & 7)
        const int col_e = min(t.n0 + wn * 64 + (ch_e << 3), nlast), col_o = min(t.n0 + wn * 64 + ((ch_e ^ 4) << 3), nlast);
        const int row0 = t.m0 + wm * 128 + h * 64 + q * 32 + (ln >> 3);
        const uint32_t ldsR = lds0 + (stA ? 0 : LW_BASE) + gw * 8192 + h * LOP_BYTES + q * 4096;
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (uint32_t)(min(row0 + 8 * i, mlast) * ldr + ((i & 1) ? col_o : col_e)) * 2u;
        glds_x4(reinterpret_cast<const char*>(p.R), o[0], o[1], o[2], o[3], ldsR);
    };
    // tile constants (bias | LayerNorm column sums or residual-LN gamma | residual-LN beta by column; row-statistics partials by row):
    // LDS-DMA dwords straight into the LDS tail / a raw-partials area, requested at the top of their tile and covered by the counted wait
    // that follows.  (gemm_line_kernel loads them into registers with asynchronous loads the compiler cannot see; in a persistent loop
    // hipcc spills such registers — before the load has landed — and every reload is a vmcnt(0) of its own.)
    constexpr bool ROWST = (VAR == 1 || VAR == 3);
    auto request_constants = [&](const Tile& t, const int par) {
        const uint32_t lds_tail = lds0 + STREAM_TAIL + par * TAIL_BYTES;
        uint32_t ones = ~0u;
        asm volatile("" : "+s"(ones));
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        int n = t.n0 + gw * 64 + ln;
        n = n < p.N ? n : p.N - 1;
        const uint32_t col_off = (uint32_t)n * 4u;
        if (stA) {
            if (p.bias) l2_touch(reinterpret_cast<const char*>(p.bias), col_off, lds_tail + TAIL_C0 + gw * 256);
            if (VAR == 3) l2_touch(reinterpret_cast<const char*>(p.rln_b), col_off, lds_tail + TAIL_C2 + gw * 256);
        } else {
            if (VAR == 1) l2_touch(reinterpret_cast<const char*>(p.ln_s), col_off, lds_tail + TAIL_C1 + gw * 256);
            if (VAR == 3) l2_touch(reinterpret_cast<const char*>(p.rln_g), col_off, lds_tail + TAIL_C1 + gw * 256);
        }
        if (ROWST) {   // lanes 0-31: the sums of rows [32 wn, 32 wn + 32) of this wave's row half, lanes 32-63: their sums of squares; slab s_i
            int m = t.m0 + wm * 128 + wn * 32 + (ln & 31);
            m = m < p.M ? m : p.M - 1;
            const uint32_t row_off = (uint32_t)(2 * m + (ln >> 5)) * 4u;
#pragma unroll
            for (int s_i = 0; s_i < 4; ++s_i)
                l2_touch(reinterpret_cast<const char*>(p.ln_stats + (s_i < p.ln_nslab ? s_i : 0) * p.stats_slab), row_off,
                         lds0 + STREAM_RAW + par * 8192 + (wave * 4 + s_i) * 256);
        }
    };
    auto finish_constants = [&](const Tile& t, const int par) {     // raw partials -> (mean, rstd) per tile row
        if (ROWST) {
            // (tile-invariant inputs re-made from opaque values: hoisted out of the tile loop they would be registers held across the main
            // loop, i.e. spills, i.e. a vmcnt(0) of the compiler's — which waits for step 1's pieces — at every reload)
            int nslab = p.ln_nslab;
            uint32_t ones = ~0u;
            asm volatile("" : "+s"(nslab), "+s"(ones));
            const int ln = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
            if (ln < 32) {
                const float* raw = reinterpret_cast<const float*>(smem + STREAM_RAW + par * 8192 + wave * 1024) + ln;
                float sx = raw[0], sq = raw[32];
#pragma unroll
                for (int s_i = 1; s_i < 4; ++s_i) {
                    const float kx = raw[s_i * 64], kq = raw[s_i * 64 + 32];     // (slabs past nslab hold a copy of slab 0: not added)
                    sx += s_i < nslab ? kx : 0.f; sq += s_i < nslab ? kq : 0.f;
                }
                if (nslab > 4) {
                    int m = t.m0 + wm * 128 + wn * 32 + ln;
                    m = m < p.M ? m : p.M - 1;
                    for (int s_i = 4; s_i < nslab; ++s_i) {
                        const float2 w = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (long)m + s_i * p.stats_slab);
                        sx += w.x; sq += w.y;
                    }
                }
                const float mean = sx * p.ln_inv_h;
                const float rstd = rsqrtf(fmaxf(sq * p.ln_inv_h - mean * mean, 0.f) + p.ln_eps);
                *reinterpret_cast<float2*>(smem + STREAM_TAIL + par * TAIL_BYTES + TAIL_STAT + (wm * 128 + wn * 32 + ln) * 8) = make_float2(mean, rstd);
            }
        }
    };
    if (!p.bias && stA) {   // (no bias: zeros, once)
        *reinterpret_cast<float*>(smem + STREAM_TAIL + TAIL_C0 + (gw * 64 + lane) * 4) = 0.f;
        *reinterpret_cast<float*>(smem + STREAM_TAIL + TAIL_BYTES + TAIL_C0 + (gw * 64 + lane) * 4) = 0.f;
    }

    f32x4_t acc[4][8];
    Tile cur = tile_of(bt);
    if (HT && cur.skip) return;                                      // (workgroup-uniform, before any barrier)
    int par = 0;                                                     // which tail / raw area holds the current tile's constants
    request_constants(cur, 0);
    refill4(cur, 0, 0, 0); refill4(cur, 0, 0, 1);                    // the stream's first step image
    for (;;) {
        const int bn = bt + nres;
        const bool has_next0 = bn < nunits;
        if (p.dbg && tid == 0) {
            unsigned long long t_; uint32_t hw_;
            asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(hw_)::"memory");
            p.dbg[8 * (long)bt + 0] = t_; p.dbg[8 * (long)bt + 3] = hw_;
        }
        // ---- top of a tile: its step 0 and its constants are in LDS (first tile: on their way).  Step 1's requests go out first; the next
        // tile's coordinates are worked out under their latency.  (Steps 0 and 1 were touched into L2 by the previous tile's last two
        // steps, step 2 is touched by step 0 as usual.)
        const Tile nxt0 = tile_of(has_next0 ? bn : bt);               // (no next tile: the prefetch slots re-touch this one)
        const bool has_next = HT ? (has_next0 && !nxt0.skip) : has_next0;     // (a padding unit of the half-height tail: no next tile either)
        const Tile nxt = (HT && has_next0 && nxt0.skip) ? tile_of(bt) : nxt0;
        // a wave none of whose rows this unit owns: no fragment reads, no MFMAs, no epilogue — its requests and its barriers only
        const bool idle = HT && cur.half && grp == 1;
        refill4(cur, 1, 1, 0); refill4(cur, 1, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // all but step 1's pieces (after the first tile: nothing but store acknowledgements)
        finish_constants(cur, par);
        if (grp == 1) asm volatile("s_barrier" ::: "memory");

        // one K = 32 half of step `s` from ring slot SLOT (phases, request placement and waits: gemm_line_kernel).  `rt` / `rstep`:
        // the step image this step requests (its own tile's step s+1, or the next tile's step 0), `pt` / `pstep`: the one it touches.
        // (`last` && RLDS: the step's requests are the first half of the residual tile; nothing is waited for in it — the epilogue counts)
#if AG_STREAM_HOIST_HI
        // (AG_STREAM_HOIST_HI) the upper half's fragments, read during the lower half's MFMA phase
        typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
        u32x4h fwh[4], fxh[8];
#endif
        auto half = [&](const int slot, const int kh, const bool refill, const Tile& rt, const int rstep, const Tile& pt, const int pstep, const bool zero,
                        const bool last) {
            asm volatile("s_barrier" ::: "memory");                        // "a"
            // (the step's requests before the fragment reads.  The reads first — so that the LDS serves them while the requests issue — was
            // built and measured in round 4: bit-identical, same time to +-0.1 %.)
            if (RLDS && last) {
                if (grp == 0) res4(cur, 0, kh);
                else if (kh == 0) { res4(cur, 0, 0); res4(cur, 0, 1); }
            } else if (refill) {
                if (grp == 0) refill4(rt, rstep, slot ^ 1, kh);
                else if (kh == 0) { refill4(rt, rstep, slot ^ 1, 0); refill4(rt, rstep, slot ^ 1, 1); }
            }
            if (kh == 1 && stA) prefetch(pt, pstep);
            uint32_t x64 = kh ? 64u : 0u;
            asm volatile("" : "+s"(x64));
            typedef __attribute__((address_space(3))) const char* lds_cptr;
            typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) const u32x4v* lds_u4ptr;
            const lds_cptr pa = (lds_cptr)(uintptr_t)(vA ^ x64) + slot * LOP_BYTES;
            const lds_cptr pw = (lds_cptr)(uintptr_t)(vW ^ x64) + slot * LOP_BYTES;
            u32x4v fw[4], fx[8];
#if AG_STREAM_HOIST_HI
            if (kh == 0) {
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[i] = *(lds_u4ptr)(pw + i * (16 * LROWB));
#pragma unroll
            for (int i = 0; i < 8; ++i) fx[i] = *(lds_u4ptr)(pa + i * (16 * LROWB));
#if AG_STREAM_HOIST_HI
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) fw[i] = fwh[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) fx[i] = fxh[i];
            }
#endif
            __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0)
            asm volatile("" ::: "memory");
            if (grp == 1 && kh == 1 && !(RLDS && last)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");                        // "b"
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#if AG_STREAM_HOIST_HI
            if (kh == 0) {
                const lds_cptr pa1 = (lds_cptr)(uintptr_t)(vA ^ 64u) + slot * LOP_BYTES;
                const lds_cptr pw1 = (lds_cptr)(uintptr_t)(vW ^ 64u) + slot * LOP_BYTES;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int sn = 0; sn < 4; ++sn) {
                        if (zero) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc[sn][g]) : "v"(fw[sn]), "v"(fx[g]));
                        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[sn][g]) : "v"(fw[sn]), "v"(fx[g]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (g < 4) { fwh[g] = *(lds_u4ptr)(pw1 + g * (16 * LROWB)); fxh[g] = *(lds_u4ptr)(pa1 + g * (16 * LROWB)); }
                    else if (g < 6) { fxh[2 * g - 4] = *(lds_u4ptr)(pa1 + (2 * g - 4) * (16 * LROWB)); fxh[2 * g - 3] = *(lds_u4ptr)(pa1 + (2 * g - 3) * (16 * LROWB)); }
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int g = 0; g < 8; ++g)
#pragma unroll
                    for (int sn = 0; sn < 4; ++sn)
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[sn][g]) : "v"(fw[sn]), "v"(fx[g]));
            }
#else
#pragma unroll
            for (int sn = 0; sn < 4; ++sn)
#pragma unroll
                for (int sm = 0; sm < 8; ++sm)
                    acc[sn][sm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[sn]), __builtin_bit_cast(bf16x8_t, fx[sm]),
                                                                          zero ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[sn][sm], 0, 0, 0);
#endif
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0 && kh == 1 && !(RLDS && last)) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        };
        // step s of this tile: requests step s+1 (or the next tile's step 0), touches step s+2 (or the next tile's step s+2-ns)
        auto step = [&](const int s, const int slot, const bool refill, const bool first, const bool last) {
            const bool own_r = s + 1 < ns, own_p = s + 2 < ns;
            const Tile& rt = own_r ? cur : nxt;
            const Tile& pt = own_p ? cur : nxt;
            const int rstep = own_r ? s + 1 : 0, pstep = own_p ? s + 2 : s + 2 - ns;
            half(slot, 0, refill, rt, rstep, pt, pstep, first, last);    // (first: the tile's accumulators start here, from a zero C operand)
            half(slot, 1, refill, rt, rstep, pt, pstep, false, last);
        };
        if (p.dbg && tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); p.dbg[8 * (long)bt + 4] = t_; }
        if (HT && idle) {
            // a wave of group 1 in a half-height unit (its workgroup's last unit: nothing is requested for a successor): every step's W
            // requests, the waits on them and the barriers of `half`, nothing else.  (Its own copy of the loop: a wave-uniform branch per
            // unit, none inside the working waves' loop.)
            for (int si = 0; si < ns; ++si) {
                const int slot = si & 1;
                asm volatile("s_barrier" ::: "memory");                        // "a", lower half
                if (si > 0 && si + 1 < ns) { refill4(cur, si + 1, slot ^ 1, 0); refill4(cur, si + 1, slot ^ 1, 1); }
                asm volatile("s_barrier" ::: "memory");                        // "b"
                asm volatile("s_barrier" ::: "memory");                        // "a", upper half
                if (!(RLDS && si + 1 == ns)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");                        // "b"
            }
            // (nobody reads an idle wave's accumulators; defined here so that the previous unit's are not carried around the working waves' loop)
#pragma unroll
            for (int sn = 0; sn < 4; ++sn)
#pragma unroll
                for (int g = 0; g < 8; ++g) acc[sn][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        } else {
        step(0, 0, false, true, false);                                    // step 1 was requested above
        int s = 1;
        for (; s + 1 < ns; s += 2) {
            step(s, 1, true, false, false);
            step(s + 1, 0, true, false, false);
        }
        step(s, 1, has_next, false, true);                                              // s == ns - 1: requests the NEXT tile's step 0 into slot 0
        }
#if AG_STREAM_HOIST_HI
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // (the asm MFMAs' results are read by vector instructions next: hipcc's hazard recogniser does not see them)
#endif
        if (p.dbg && tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); p.dbg[8 * (long)bt + 5] = t_; }
        if (grp == 0) asm volatile("s_barrier" ::: "memory");              // pairs group 1's extra first barrier

        // (the next tile's step 0 has landed in slot 0: the last step's waits; the epilogue stages through slot 1.)  The next tile's constants
        // are requested now, into the other tail: they land under the epilogue
        if (has_next) request_constants(nxt, par ^ 1);
        if constexpr (RLDS) {
            constexpr bool STATS = (VAR == 2 || VAR == 3), RLN = (VAR == 3);
            uint32_t ones = ~0u;
            asm volatile("" : "+s"(ones));
            const int le = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));   // (opaque lane id: see below)
            const int frow = le & 15, fq = le >> 4;
            const int mw0 = cur.m0 + wm * 128, nw0 = cur.n0 + wn * 64;
            const bool full_cols = nw0 + 64 <= p.N;
            const int Mt = HT ? cur.mhi : p.M;                                  // rows this unit may write
            const bool edge_tile = !full_cols || mw0 + 128 > Mt;                // (wave-uniform)
            char* const area = smem + (stA ? 0 : LW_BASE) + gw * 8192;          // this wave's piece area of slot 0; slot 1: + LOP_BYTES
            // this wave's 128 x (sum, sumsq) partials: in the raw-partials area of THIS tile's parity (consumed at the top of the tile; the
            // next tile's raw partials arrive in the other one)
            char* const part = smem + STREAM_RAW + par * 8192 + wave * 1024;
            const char* const tail = smem + STREAM_TAIL + par * TAIL_BYTES;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done reading slot 1
            if (idle) {                                                         // (a half-height unit is its workgroup's last: nothing to request either)
                if (STATS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
            res4(cur, 1, 0); res4(cur, 1, 1);                                   // rows [64, 128) of the residual sub-tile -> slot 1
            float4 bv[4], gv[RLN ? 4 : 1], btv[RLN ? 4 : 1];
#pragma unroll
            for (int sn = 0; sn < 4; ++sn) {
                bv[sn] = *reinterpret_cast<const float4*>(tail + TAIL_C0 + (wn * 64 + sn * 16 + fq * 4) * 4);
                if (RLN) {   // residual = LayerNorm of the stored pre-LN rows: gamma, beta by column; (mean, rstd) by row from the tail
                    gv[RLN ? sn : 0] = *reinterpret_cast<const float4*>(tail + TAIL_C1 + (wn * 64 + sn * 16 + fq * 4) * 4);
                    btv[RLN ? sn : 0] = *reinterpret_cast<const float4*>(tail + TAIL_C2 + (wn * 64 + sn * 16 + fq * 4) * 4);
                }
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                    // all but the eight pieces just requested: rows [0, 64) are in LDS
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                char* const img = area + h * LOP_BYTES;
#pragma unroll
                for (int smh = 0; smh < 4; ++smh) {
                    const int sm = 4 * h + smh;
                    const int lrow = smh * 16 + frow;                            // row of the 64-row half image
                    const int m = mw0 + sm * 16 + frow;
                    f32x2_t rs2 = {0.f, 0.f}, rq2 = {0.f, 0.f};
                    float nm = 0.f, ln_rstd = 1.f;
                    if (RLN) {
                        const float2 mr = *reinterpret_cast<const float2*>(tail + TAIL_STAT + (wm * 128 + sm * 16 + frow) * 8);
                        nm = -mr.x; ln_rstd = mr.y;
                    }
#pragma unroll
                    for (int sn = 0; sn < 4; ++sn) {
                        uint2* const at = reinterpret_cast<uint2*>(img + lrow * 128 + (((sn * 2 + (fq >> 1)) ^ ((lrow >> 1) & 7)) << 4) + (fq & 1) * 8);
                        const uint2 rq = *at;
                        float r0_ = __uint_as_float(rq.x << 16), r1_ = __uint_as_float(rq.x & 0xFFFF0000u);
                        float r2_ = __uint_as_float(rq.y << 16), r3_ = __uint_as_float(rq.y & 0xFFFF0000u);
                        if (RLN) {
                            constexpr int SI = RLN ? 1 : 0;
                            r0_ = fmaf((r0_ + nm) * ln_rstd, gv[sn * SI].x, btv[sn * SI].x); r1_ = fmaf((r1_ + nm) * ln_rstd, gv[sn * SI].y, btv[sn * SI].y);
                            r2_ = fmaf((r2_ + nm) * ln_rstd, gv[sn * SI].z, btv[sn * SI].z); r3_ = fmaf((r3_ + nm) * ln_rstd, gv[sn * SI].w, btv[sn * SI].w);
                        }
                        // (acc + bias) + residual: the same sum in the same order as wave_epilogue
                        const float v0 = (acc[sn][sm][0] + bv[sn].x) + r0_, v1 = (acc[sn][sm][1] + bv[sn].y) + r1_;
                        const float v2 = (acc[sn][sm][2] + bv[sn].z) + r2_, v3 = (acc[sn][sm][3] + bv[sn].w) + r3_;
                        const uint2 pk = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                        *at = pk;
                        if (STATS) {   // (the same packed partial sums, in the same order, as wave_epilogue)
                            f32x2_t ra = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xFFFF0000u)};
                            f32x2_t rb = {__uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xFFFF0000u)};
                            if (edge_tile) {
                                const float keep = (m < Mt && nw0 + sn * 16 + fq * 4 < p.N) ? 1.f : 0.f;
                                const f32x2_t k2 = {keep, keep};
                                ra *= k2; rb *= k2;
                            }
                            rs2 += ra + rb;
                            rq2 = __builtin_elementwise_fma(rb, rb, __builtin_elementwise_fma(ra, ra, rq2));
                        }
                    }
                    if (STATS) {
                        float row_s = rs2.x + rs2.y, row_q = rq2.x + rq2.y;
                        row_s = quad_rows_sum(row_s); row_q = quad_rows_sum(row_q);
                        if (fq == 0) *reinterpret_cast<float2*>(part + (sm * 16 + frow) * 8) = make_float2(row_s, row_q);
                    }
                }
                // the half image is now the OUTPUT: read it back as whole 128-byte rows (this wave's own LDS operations complete in order)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rr = i * 8 + (le >> 3), ch = le & 7;
                    const uint4 val = *reinterpret_cast<const uint4*>(img + rr * 128 + ((ch ^ ((rr >> 1) & 7)) << 4));
                    const int mm = mw0 + h * 64 + rr;
                    if (mm < Mt && (full_cols || nw0 + ch * 8 < p.N)) {
                        uint4* dstp = reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (long)mm * p.ldc + nw0 + ch * 8);
                        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                        if (p.nt_store) {
                            __builtin_nontemporal_store(u32x4{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4*>(dstp));
                        } else *dstp = val;
                    }
                }
                if (h == 0) {
                    // the slot-0 area is free again (its rows are in registers / on their way out): the next tile's step 0 goes there now;
                    // then everything older than those eight pieces — the residual's second half, the stores — is waited for
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (has_next) {
                        refill4(nxt, 0, 0, 0); refill4(nxt, 0, 0, 1);
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
            }
            if (STATS) {
                // the four column waves (wn = 0..3) of this row half have each left 128 row partials: add them in wave order and store ONE
                // (sum, sumsq) per row and column tile.  Wave wn finishes rows [32 wn, 32 wn + 32).
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (le < 32) {
                    const int row = wn * 32 + le;
                    const char* half_ = smem + STREAM_RAW + par * 8192 + (wm * 4) * 1024 + row * 8;
                    float2 t = *reinterpret_cast<const float2*>(half_);
#pragma unroll
                    for (int w = 1; w < 4; ++w) {
                        const float2 u = *reinterpret_cast<const float2*>(half_ + w * 1024);
                        t.x += u.x; t.y += u.y;
                    }
                    const int m = mw0 + row;
                    if (m < Mt) *reinterpret_cast<float2*>(p.stats_out + (long)cur.tn * p.stats_slab + 2 * (long)m) = t;
                }
            }
            }   // (!idle)
        } else
        {   // (the lane id re-made from an opaque value: the epilogue's per-lane addresses are tile-invariant, and hoisted out of the tile
            // loop they would be a dozen registers held across the main loop)
            uint32_t ones = ~0u;
            asm volatile("" : "+s"(ones));
            const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
            wave_epilogue<EPI, VAR == 1, VAR == 2 || VAR == 3, VAR == 3, 1>(p, acc, cur.m0 + wm * 128, cur.n0 + wn * 64,
                                                                              smem + 32768 + grp * 65536 + gw * 8192, lane_e, smem, cur.tn,
                                                                              smem + STREAM_TAIL + par * TAIL_BYTES, cur.cz, HT ? cur.mhi : -1, idle);
        }
        if (p.dbg) {
            unsigned long long t1, t2;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
            if (tid == 0) { p.dbg[8 * (long)bt + 1] = t1; p.dbg[8 * (long)bt + 2] = t2; }
        }
        if (!has_next) break;
        // this wave is done with its staging piece (= where its own step-1 pieces go next); the statistics producers also read each
        // other's partials there: they wait for each other.  (The tails alternate: nobody waits for a tail.)
        if (VAR == 2 || VAR == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        par ^= 1;
        bt = bn;
        cur = nxt;
    }
}

}  // namespace

// Half-height tail: `tiles` 256^2 tiles on `n_cu` resident workgroups = `rounds` complete rounds + a rest.  A rest that fits HALF a round
// (padded to a multiple of the 8 XCDs) runs as 128-row halves next to each other instead of as one more round of whole tiles on part of the
// chip.  Measured (tools/r6_gemm_ab.py, same box, whole -> half-height tail; profiles/HISTORY.md §12): a half-height unit takes 0.85-0.9 of a
// whole tile's time, not 0.5 — a step is as long as its LDS-DMA pieces take to land, whatever the MFMAs under it — so the tail pays where it
// is a large part of the launch: no complete round (ViT-base at one input, out-projection, 75 tiles: 21.8 -> 19.5 us; fc2 59.6 -> 50.8) or one
// (fc1 at one input, 300 tiles: 44.5 -> 42.0; the out-projection at four inputs, 297: 44.6 -> 42.9); from two complete rounds on it is
// within +-1 % (QKV at four inputs 85.6 / 86.0, ViT-large one input x 64 masks 75.6 / 76.1), and the benchmarked fc1 (55 rounds + 104 tiles)
// loses 1.2 % to it: AG_GEMM_HALFTAIL_ROUNDS = the most complete rounds a launch may have (default 1).  AG_GEMM_HALFTAIL=0: never (A/B,
// parity tests).
bool ag_big_half_tail(int tiles, int n_cu, int* half_from, int* ntail, int* grid) {
    static AgKnob k_ht("AG_GEMM_HALFTAIL"), k_rounds("AG_GEMM_HALFTAIL_ROUNDS");
    if ((int)k_ht.get(1) == 0 || n_cu < 16 || (n_cu & 7) || tiles <= 0) return false;
    const int rounds = tiles / n_cu, rest = tiles - rounds * n_cu, pad = (rest + 7) & ~7;
    if (rest == 0 || 2 * pad > n_cu || rounds > (int)k_rounds.get(1)) return false;
    *half_from = rounds * n_cu;
    *ntail = rest;
    *grid = rounds > 0 ? n_cu : 2 * pad;
    return true;
}

static thread_local int g_last_plan[4] = {0, 0, 0, 0};      // {seen, grid, half_from, ntail} of this thread's last launch_stream_var
extern "C" int ag_gemm_last_plan(int* grid, int* half_from, int* ntail) {
    if (grid) *grid = g_last_plan[1];
    if (half_from) *half_from = g_last_plan[2];
    if (ntail) *ntail = g_last_plan[3];
    return g_last_plan[0];
}

namespace {

template <int EPI, int VAR, bool RLDS = false>
int launch_stream_var(const BigArgs& a, hipStream_t s) {
    if constexpr (!RLDS && EPI == AG_EPI_BIAS_RESID && (VAR == 0 || VAR == 2 || VAR == 3)) {
        // residual through LDS: identity residual row map (share == 1: every layer but the first) and 32-bit byte offsets into R
        static AgKnob k_rlds("AG_GEMM_RLDS");
        if (a.R && a.share == 1 && (unsigned long long)a.M * (unsigned long long)a.ldr < 0x7FFFFFF0ull && (int)k_rlds.get(1) != 0)
            return launch_stream_var<EPI, VAR, true>(a, s);
    }
    // per device (a process may drive several: the attribute belongs to the device's copy of the code object, and partitions
    // differ in their CU count)
    constexpr int MAX_DEV = 16;
    static bool attr_set[MAX_DEV] = {};
    static int n_cu_of[MAX_DEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ag_fail(AG_ERR_HIP, "gemm_stream: hipGetDevice");
    if (dev < 0 || dev >= MAX_DEV) return ag_fail(AG_ERR_UNSUPPORTED, "gemm_stream: device index %d", dev);
    constexpr bool BATCHED = (EPI == AG_EPI_BIAS_F32 && VAR == 0);
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_stream_kernel<EPI, VAR, RLDS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, STREAM_LDS_BYTES);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(gemm_stream): %s", hipGetErrorString(e));
        if constexpr (!BATCHED) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_stream_kernel<EPI, VAR, RLDS, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, STREAM_LDS_BYTES);
            if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(gemm_stream, half-height tail): %s", hipGetErrorString(e));
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ag_fail(AG_ERR_HIP, "gemm_stream: device properties");
        // one resident workgroup per CU; a multiple of the 8 XCDs keeps a workgroup's tiles on its XCD (a partition with fewer than
        // 8 CUs: every CU)
        const int cu = prop.multiProcessorCount;
        n_cu_of[dev] = cu >= 8 ? (cu & ~7) : (cu > 0 ? cu : 1);
        attr_set[dev] = true;
    }
    int n_cu = n_cu_of[dev];
    {   // a CU-masked stream (ag_set_stream_cus): one resident workgroup per CU of ITS partition (a multiple of the 8 XCDs)
        const int sc = ag_stream_cus(s);
        if (sc > 0 && sc < n_cu) n_cu = sc >= 8 ? (sc & ~7) : sc;
    }
    const int tiles = ceil_div(a.M, BT) * ceil_div(a.N, BT) * (BATCHED ? a.nbatch : 1);
    if constexpr (!BATCHED) {
        // half-height tail (gemm_stream_kernel<..., HT>): an exact row count only (the plan is made from M)
        int half_from = 0, ntail = 0, hgrid = 0;
        if (!a.dyn && ag_big_half_tail(tiles, n_cu, &half_from, &ntail, &hgrid)) {
            BigArgs h = a;
            h.half_from = half_from; h.ntail = ntail;
            hipLaunchKernelGGL((gemm_stream_kernel<EPI, VAR, RLDS, true>), dim3(hgrid), dim3(NT), STREAM_LDS_BYTES, s, h);
            AG_LAUNCH_CHECK();
            g_last_plan[0] = 1; g_last_plan[1] = hgrid; g_last_plan[2] = half_from; g_last_plan[3] = ntail;
            return AG_OK;
        }
    }
    // resident workgroups: every CU — unless the same number of rounds is done by fewer: 297 tiles are two rounds on 256 CUs (the second with 41
    // busy) and two rounds on 152 (both full), and a tile is faster the fewer CUs share an XCD's L2 feed (the forward confined to 224 / 192 CUs:
    // 7 % / 15 % less time per tile, tools/gemm_cus_sweep.py).  Large launches (14+ rounds) come out at every CU.
    int grid = tiles < n_cu ? tiles : n_cu;
    if (tiles > n_cu) {
        static AgKnob k_bal("AG_GEMM_BALANCE");            // 0: always every CU (A/B)
        if ((int)k_bal.get(1) != 0) {
            const int rounds = ceil_div(tiles, n_cu);
            grid = (ceil_div(tiles, rounds) + 7) & ~7;
            if (grid > n_cu) grid = n_cu;
        }
    }
    hipLaunchKernelGGL((gemm_stream_kernel<EPI, VAR, RLDS>), dim3(grid), dim3(NT), STREAM_LDS_BYTES, s, a);
    AG_LAUNCH_CHECK();
    g_last_plan[0] = 1; g_last_plan[1] = grid; g_last_plan[2] = tiles; g_last_plan[3] = 0;
    return AG_OK;
}

#ifdef AG_REF_KERNELS
template <int EPI, int VAR>
int launch_line_var(const BigArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_line_kernel<EPI, VAR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LINE_LDS_BYTES);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(gemm_line): %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ceil_div(a.M, BT) * ceil_div(a.N, BT);
    hipLaunchKernelGGL((gemm_line_kernel<EPI, VAR>), dim3(tiles), dim3(NT), LINE_LDS_BYTES, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
#endif

// The shipped library carries ONE large-M kernel, gemm_stream_kernel (K % 128 == 0: every encoder shape of the base / large
// configurations).  The reference build (-DAG_REF_KERNELS -> libautognothi_hip_ref.so, loaded by the parity tests only) adds the two
// earlier generations behind AG_GEMM_STREAM=0 (gemm_line_kernel) and AG_GEMM_LINE=0 / K % 128 != 0 (gemm_ring_kernel): the stream
// kernel's every epilogue is tested bit for bit against them.
template <int EPI, int VAR>
int launch_ring_var(const BigArgs& a, hipStream_t s) {
#ifdef AG_REF_KERNELS
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<EPI, VAR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return ag_fail(AG_ERR_HIP, "hipFuncSetAttribute(gemm_ring): %s", hipGetErrorString(e));
        attr_set = true;
    }
    static AgKnob k_line("AG_GEMM_LINE");
    static AgKnob k_stream("AG_GEMM_STREAM");
    if (a.K % 128 == 0 && (int)k_line.get(1) != 0 && (int)k_stream.get(1) != 0) return launch_stream_var<EPI, VAR>(a, s);
    if (a.K % 128 == 0 && (int)k_line.get(1) != 0) return launch_line_var<EPI, VAR>(a, s);
    const int tiles = ceil_div(a.M, BT) * ceil_div(a.N, BT);
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, VAR>), dim3(tiles), dim3(NT), LDS_BYTES, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
#else
    if (a.K % 128 != 0) return ag_fail(AG_ERR_UNSUPPORTED, "ag_gemm_big: K=%d is not a multiple of 128 (ag_gemm_big_eligible)", a.K);
    return launch_stream_var<EPI, VAR>(a, s);
#endif
}

template <int EPI>
int launch_ring(const BigArgs& a, hipStream_t s) {
    constexpr bool CAN_FOLD = (EPI == AG_EPI_BIAS || EPI == AG_EPI_BIAS_GELU);
    constexpr bool CAN_STATS = (EPI == AG_EPI_BIAS_RESID);
    if (a.ln_stats) {
        if constexpr (CAN_FOLD) return launch_ring_var<EPI, 1>(a, s);
        else return ag_fail(AG_ERR_INVALID, "ag_gemm: LayerNorm folding is built for the bias and bias+gelu epilogues only");
    }
    if (a.stats_out) {
        if constexpr (CAN_STATS) return launch_ring_var<EPI, 2>(a, s);
        else return ag_fail(AG_ERR_INVALID, "ag_gemm: row statistics are built for the bias+residual epilogue only");
    }
#ifdef AG_REF_KERNELS
    if (a.dbg && EPI == AG_EPI_BIAS) {  // diagnostic (stamped) build, tools/gemm_stamps.py
        static bool dset = false;
        if (!dset) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<AG_EPI_BIAS, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); dset = true; }
        const int tiles = ceil_div(a.M, BT) * ceil_div(a.N, BT);
        hipLaunchKernelGGL((gemm_ring_kernel<AG_EPI_BIAS, 0, true>), dim3(tiles), dim3(NT), LDS_BYTES, s, a);
        AG_LAUNCH_CHECK();
        return AG_OK;
    }
#endif
    return launch_ring_var<EPI, 0>(a, s);
}


}  // namespace

// Eligibility: bf16, vectorisable epilogue, K a multiple of 32 with at least 4 half-steps.
bool ag_gemm_big_eligible(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, int epilogue) {
    // Problems of fewer than 48 tiles (< 1/5 of the CUs busy, each for a whole 256^2 tile's latency) go to the 128 / 64-tile
    // kernel: more, shorter workgroups.  These are the explainer-training GEMMs (M = 8 images x 197 tokens: 7 row panels):
    // training step +11 % (vanilla ViT-base) / +23 % (duo BERT-base) at 8 images; a threshold of 100 would also catch the
    // N = 768 GEMMs of a single-input inference step (75 tiles, K up to 3072), where the ring is 7 % faster.
    // fp32-output launches are the training step's (bf16 operands, fp32 activations): nothing folds a LayerNorm into them and
    // with the counted-vmcnt ring of the 64-tile kernel the break-even moved up: 130 tiles (duo BERT-base step -5 %, ViT -1.5 %).
    static AgKnob k_min_tiles("AG_GEMM_BIG_MIN_TILES");    // (the kernel parity tests pin the ring with it: ag_reload_knobs)
    const int min_tiles = (int)k_min_tiles.get(epilogue == AG_EPI_BIAS_F32 ? 130 : 48);
    if ((long)ceil_div(M, BT) * ceil_div(N, BT) < min_tiles) return false;
#ifdef AG_REF_KERNELS
    constexpr int KMULT = 32;      // (the ring kernel walks K in 32-element half-steps)
#else
    constexpr int KMULT = 128;     // the stream kernel's step pair; other K (ViT-tiny's 192) go to the 128-tile kernel of gemm.hip
#endif
    return M >= 1024 && N >= 256 && (N % 8) == 0 && K % KMULT == 0 && K >= 128 && (lda % 8) == 0 && (ldc % 8) == 0 &&
           (epilogue != AG_EPI_BIAS_RESID || (ldr % 4) == 0);
}

static int run_big(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                   const void* d_R, int64_t ldr, int rows_per_seq, int resid_share, int M, int N, int K, int epilogue,
                   const float* d_ln_stats, const float* d_ln_colsum, float ln_eps, float* d_stats_out,
                   const float* d_rln_g, const float* d_rln_b, const int* d_rows, hipStream_t s,
                   long stats_rows = 0, int nbatch = 1, long ldw = 0, long bk_b = 0, long bc = 0) {
    BigArgs a;
    a.rln_g = d_rln_g; a.rln_b = d_rln_b;
    a.A = (const char*)d_A; a.lda_b = (long)lda * 2;
    a.W = (const char*)d_W; a.ldw_b = (ldw > 0 ? ldw : (long)K) * 2;
    a.nbatch = nbatch; a.bk_b = bk_b; a.bc = bc;
    a.bias = d_bias; a.C = (char*)d_C; a.ldc = ldc; a.R = (const bf16_t*)d_R; a.ldr = ldr;
    a.T = rows_per_seq > 0 ? rows_per_seq : 1; a.share = resid_share > 0 ? resid_share : 1;
    a.M = M; a.N = N; a.K = K;
    a.ln_stats = d_ln_stats; a.ln_s = d_ln_colsum; a.ln_eps = ln_eps; a.ln_inv_h = 1.0f / (float)K; a.stats_out = d_stats_out;
    a.stats_slab = 2L * (stats_rows > 0 ? stats_rows : (long)M); a.ln_nslab = ceil_div(K, BT);   // (stats_rows: this launch covers a row range of a taller output)
    if (d_rln_g) { a.ln_inv_h = 1.0f / (float)N; a.ln_nslab = ceil_div(N, BT); }   // the statistics describe the residual rows [M, N]
    a.dbg = nullptr;
    a.dyn = d_rows;
    static AgKnob k_dbg("AG_GEMM_DBG");
    if (k_dbg.is_set()) {  // diagnostic build: stamps into a lazily allocated device buffer (never in production)
        static unsigned long long* dbuf = nullptr;
        constexpr int DBG_WGS = 32768;      // 8 stamps per workgroup (gemm_line_kernel's timeline); larger launches go unstamped
        if (!dbuf) { (void)hipMalloc((void**)&dbuf, 8L * DBG_WGS * sizeof(unsigned long long)); }
        if (ceil_div(M, BT) * ceil_div(N, BT) <= DBG_WGS) a.dbg = dbuf;
        FILE* f = fopen(k_dbg.str, "w");
        if (f) { fprintf(f, "%p\n", (void*)dbuf); fclose(f); }
    }
    // group width: one group (the plain N-fastest order) unless the weight matrix overflows an XCD's 4 MiB L2 while the
    // A panels are cheap to fetch again (short K): then groups of <= 3.2 MiB of weight rows (fc1 768->3072: 2 groups of 6
    // tiles, measured -1.7 %; splitting fc2's 3 tiles (K = 3072) costs +16 %: its A panels are 1.5 MB each)
    static AgKnob k_ngrp("AG_GEMM_NGRP"), k_wfit("AG_GEMM_WFIT_MB"), k_nt("AG_GEMM_NT");   // (the parity tests toggle them)
    const int ngrp_env = (int)k_ngrp.get(0);
    {
        const int tiles_n = ceil_div(N, BT);
        const double wbytes = (double)N * K * 2.0;
        const double wfit_mb = k_wfit.get(4.0);   // (experiment knob)
        // (round 6, tools/r6_gemm_ab.py at M = 605 184: ViT-large's QKV, 3 072 x 1 024 = 6.3 MB of W, 12 N-tiles, as 2 / 3 / 4 / 6 / 12 tiles
        // per group: 3 240 / 3 119 / 3 209 / **3 075** / 3 127 us — groups of <= 3.2 MB, not 2.5: two groups of 6, where the old rule made three of 4)
        int groups = (wbytes > wfit_mb * 1024 * 1024 && K <= 1024) ? (int)(wbytes / (3.2 * 1024 * 1024) + 0.999) : 1;
        int g = ngrp_env > 0 ? ngrp_env : ceil_div(tiles_n, groups);
        // wide outputs (16 or more N-tiles: none in the encoder): the 32 tiles an XCD runs at a time would be ONE row of
        // tiles (1 A slice + 32 W slices per half-step); groups of 4 columns make them an 8 x 4 block (8 + 4 slices) that
        // walks K in step and shares its slices in L2 (tools/gemm_square.py: 8192^3 935 -> 1385 TFLOP/s, 4096^3 1224 -> 1296)
        if (ngrp_env <= 0 && tiles_n >= 16 && g > 4) g = 4;
        if (g < 1) g = 1;
        if (g > tiles_n) g = tiles_n;
        a.ngrp = g;
    }
    const int nt_env = (int)k_nt.get(-1);
    a.nt_store = nt_env >= 0 ? nt_env : ((double)M * N * 2.0 > 192.0 * 1024 * 1024);
    if (d_rln_g) return launch_ring_var<AG_EPI_BIAS_RESID, 3>(a, s);
    switch (epilogue) {
        case AG_EPI_BIAS: return launch_ring<AG_EPI_BIAS>(a, s);
        case AG_EPI_BIAS_GELU: return launch_ring<AG_EPI_BIAS_GELU>(a, s);
        case AG_EPI_BIAS_RESID: return launch_ring<AG_EPI_BIAS_RESID>(a, s);
        case AG_EPI_BIAS_F32: return launch_ring<AG_EPI_BIAS_F32>(a, s);
        case AG_EPI_BIAS_TANH: return launch_ring<AG_EPI_BIAS_TANH>(a, s);
        default: return ag_fail(AG_ERR_INVALID, "ag_gemm_big: unknown epilogue %d", epilogue);
    }
}

int ag_gemm_big(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                const void* d_R, int64_t ldr, int rows_per_seq, int resid_share, int M, int N, int K, int epilogue,
                const float* d_ln_stats, const float* d_ln_colsum, float ln_eps, float* d_stats_out, const int* d_rows, hipStream_t s) {
    return run_big(d_A, lda, d_W, d_bias, d_C, ldc, d_R, ldr, rows_per_seq, resid_share, M, N, K, epilogue, d_ln_stats, d_ln_colsum,
                   ln_eps, d_stats_out, nullptr, nullptr, d_rows, s);
}

// C = A·Wᵀ + bias + LayerNorm(Rpre) with the LayerNorm recomputed in the epilogue from the pre-LN rows and their slab
// statistics, and the statistics of the rows written handed on (see include/autognothi_hip.h).
extern "C" int ag_gemm_resid_ln(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                                const void* d_Rpre, int64_t ldr, const float* d_r_stats, const float* d_ln_g, const float* d_ln_b,
                                float ln_eps, int M, int N, int K, float* d_stats_out, const int* d_rows, void* stream) {
    if (M == 0) return AG_OK;
    AG_REQUIRE(d_A && d_W && d_C && d_Rpre && d_r_stats && d_ln_g && d_ln_b && d_stats_out, "ag_gemm_resid_ln: null pointer");
    AG_REQUIRE(ag_gemm_resid_ln_supported(M, N, K, lda, ldc, ldr), "ag_gemm_resid_ln: shape M=%d N=%d K=%d is not served by the large-M "
               "bf16 kernel (check ag_gemm_resid_ln_supported first)", M, N, K);
    AgProfScope prof(AG_EPI_BIAS_RESID, 2.0 * M * (double)N * K, ((double)M * K + (double)N * K + 2.0 * (double)M * N) * 2.0, (hipStream_t)stream,
                     d_rows, (double)M);
    return run_big(d_A, lda, d_W, d_bias, d_C, ldc, d_Rpre, ldr, 1, 1, M, N, K, AG_EPI_BIAS_RESID, d_r_stats, nullptr, ln_eps, d_stats_out,
                   d_ln_g, d_ln_b, d_rows, (hipStream_t)stream);
}

extern "C" int ag_gemm_resid_ln_supported(int M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr) {
    static const bool force_small = getenv("AG_GEMM_SMALL") != nullptr;
    return (!force_small && ag_gemm_big_eligible(M, N, K, lda, ldc, ldr, AG_EPI_BIAS_RESID)) ? 1 : 0;
}

// =====================================================================================================================
// ag_gemm_resid_split — C = A·Wᵀ + bias + R for launches whose LAST ROUND of 256² tiles is under-filled.
//
// The persistent kernel gives every CU the tiles b, b + 256, ...: 297 tiles (one input x K = 32 ViT-base masks x 4, N = 768) are a
// full round and a second one with 41 CUs busy — for all 48 steps of fc2's K = 3 072.  (One input x 32 masks: 75 tiles, one round,
// 29 % of the CUs.)  Here the row panels of the full rounds run as before and the rows of the tail round are computed as `splits`
// contraction ranges SIDE BY SIDE in one launch (gemm_stream_kernel's BATCH form: unit = (range, tile), fp32 partial tiles into
// d_scratch [splits][M2][N]): 41 x 6 = 246 units of 8 steps.  A row kernel then adds the partials in range order, bias and the
// residual, rounds to bf16, stores, and emits the (sum, sum of squares) slab statistics of what it stored exactly as the
// GEMM epilogue would (LayerNorm fold of the consumer): no atomics, bit-reproducible from run to run (the sums are taken in
// another order than the unsplit launch's, i.e. equal to it to fp32 rounding, not bit for bit).
// Reference: the Linear of models/vanilla_vit.py:498-504 (ViTOutput: dense + residual) at the reference's own batch sizes.
namespace {

struct SplitPlan { int m1, m2, splits; };

int device_cus() {
    constexpr int MAX_DEV = 16;
    static int n_cu_of[MAX_DEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 0;
    if (!n_cu_of[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        const int cu = prop.multiProcessorCount;
        n_cu_of[dev] = cu >= 8 ? (cu & ~7) : (cu > 0 ? cu : 1);
    }
    return n_cu_of[dev];
}

bool split_plan(int M, int N, int K, int n_cu, SplitPlan* pl) {
    static AgKnob k_split("AG_GEMM_SPLIT");            // 0: never split (A/B, parity tests)
    if ((int)k_split.get(1) == 0 || n_cu <= 0 || K % 128 != 0) return false;
    const int tiles_n = ceil_div(N, BT);
    if (tiles_n > n_cu) return false;
    const int ppr = n_cu / tiles_n;                     // row panels per round
    const int panels = ceil_div(M, BT);
    const int tail_panels = panels % ppr;
    if (tail_panels == 0) return false;
    const int tail_tiles = tail_panels * tiles_n;
    if (tail_tiles * 2 > n_cu) return false;            // more than half a round: the split's fixed costs eat the gain
    const int ns = K / 64;
    int s_max = n_cu / tail_tiles;
    if (s_max > 8) s_max = 8;
    const int m1 = (panels - tail_panels) * BT, m2 = M - m1;
    for (int s = s_max; s >= 2; --s) {
        if (K % (s * 128) != 0 || ns / s < 8) continue;
        // what the split saves: (ns - ns / s) steps of ~1.45 us on the tail round; what it costs: the partial tiles written and read back
        // (m2 x N x (4 s + 4) bytes through the finishing kernel at ~4 TB/s) + ~6 us of extra prologue / epilogue / launch.  Measured
        // (tools/r4_ab.sh split): worth it from a 1.5 x margin on.
        const double gain_us = 1.45 * (ns - ns / s);
        const double cost_us = (double)m2 * N * (4.0 * s + 4.0) / 4.0e6 + 6.0;
        if (gain_us < 1.5 * cost_us) continue;
        pl->m1 = m1; pl->m2 = m2; pl->splits = s;
        return true;
    }
    return false;
}

// a half-wave (32 lanes x 8 columns) per (row, 256-column slab); SPLITS partial rows loaded at once (a runtime-length loop of
// load -> add would pay one memory round trip per range)
template <int SPLITS>
__global__ __launch_bounds__(256) void split_finish_kernel(const float* __restrict__ slabs, long slab_stride, const float* __restrict__ bias,
                                                           const bf16_t* __restrict__ R, long ldr, bf16_t* __restrict__ C, long ldc, int m1, int M, int N,
                                                           float* __restrict__ stats_out, long stats_slab) {
    const int nslab = (N + 255) >> 8;
    const long item = ((long)blockIdx.x * 256 + threadIdx.x) >> 5;       // (row of the tail, slab)
    const int sub = threadIdx.x & 31;
    const int row = (int)(item / nslab), slab = (int)(item - (long)row * nslab);
    if (row >= M - m1) return;                                          // (whole half-waves leave together)
    const int m = m1 + row;
    const int c = slab * 256 + sub * 8;
    const bool on = c < N;                                              // (N % 8 == 0)
    float sum = 0.f, sq = 0.f;
    if (on) {
        const float* p0 = slabs + (long)row * N + c;
        float4 x0[SPLITS], x1[SPLITS];
#pragma unroll
        for (int s = 0; s < SPLITS; ++s) {
            x0[s] = *reinterpret_cast<const float4*>(p0 + s * slab_stride);
            x1[s] = *reinterpret_cast<const float4*>(p0 + s * slab_stride + 4);
        }
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (bias) { b0 = *reinterpret_cast<const float4*>(bias + c); b1 = *reinterpret_cast<const float4*>(bias + c + 4); }
        const uint4 r = *reinterpret_cast<const uint4*>(R + (long)m * ldr + c);
        float4 a0 = x0[0], a1 = x1[0];
#pragma unroll
        for (int s = 1; s < SPLITS; ++s) {                              // in range order: bit-reproducible
            a0.x += x0[s].x; a0.y += x0[s].y; a0.z += x0[s].z; a0.w += x0[s].w;
            a1.x += x1[s].x; a1.y += x1[s].y; a1.z += x1[s].z; a1.w += x1[s].w;
        }
        // (acc + bias) + residual: the order of the GEMM epilogue
        const float v0 = (a0.x + b0.x) + __uint_as_float(r.x << 16), v1 = (a0.y + b0.y) + __uint_as_float(r.x & 0xFFFF0000u);
        const float v2 = (a0.z + b0.z) + __uint_as_float(r.y << 16), v3 = (a0.w + b0.w) + __uint_as_float(r.y & 0xFFFF0000u);
        const float v4 = (a1.x + b1.x) + __uint_as_float(r.z << 16), v5 = (a1.y + b1.y) + __uint_as_float(r.z & 0xFFFF0000u);
        const float v6 = (a1.z + b1.z) + __uint_as_float(r.w << 16), v7 = (a1.w + b1.w) + __uint_as_float(r.w & 0xFFFF0000u);
        const uint4 pk = make_uint4(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3), pack_bf16x2(v4, v5), pack_bf16x2(v6, v7));
        *reinterpret_cast<uint4*>(C + (long)m * ldc + c) = pk;
        const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                   // statistics of the ROUNDED values, as the GEMM epilogue takes them
            const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xFFFF0000u);
            sum += lo + hi; sq += lo * lo + hi * hi;
        }
    }
    if (stats_out) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
        if (sub == 0) *reinterpret_cast<float2*>(stats_out + (long)slab * stats_slab + 2 * (long)m) = make_float2(sum, sq);
    }
}

}  // namespace

// CUs the persistent kernel may take on stream s: the device's, or the stream's own budget (ag_set_stream_cus), a multiple of the 8 XCDs
static int stream_cus_of(hipStream_t s) {
    int n = device_cus();
    const int sc = ag_stream_cus(s);
    if (sc > 0 && sc < n) n = sc >= 8 ? (sc & ~7) : sc;
    return n;
}

bool ag_resid_split_plan(int M, int N, int K, int* m1, int* m2, int* splits) {
    SplitPlan pl;
    if (!ag_gemm_big_eligible(M, N, K, K, N, N, AG_EPI_BIAS_RESID) || !split_plan(M, N, K, device_cus(), &pl)) return false;
    if (m1) *m1 = pl.m1;
    if (m2) *m2 = pl.m2;
    if (splits) *splits = pl.splits;
    return true;
}
int ag_device_cus() { return device_cus(); }

// 0: the shape does not split on the whole device (call ag_gemm).  Otherwise the LARGEST scratch any CU budget of a stream needs (the
// launch plans its rounds for the CUs of ITS stream — a multiple of 8 up to the device's —, which this call does not know)
extern "C" size_t ag_gemm_resid_split_scratch_bytes(int M, int N, int K) {
    SplitPlan pl;
    if (!ag_gemm_big_eligible(M, N, K, K, N, N, AG_EPI_BIAS_RESID) || !split_plan(M, N, K, device_cus(), &pl)) return 0;
    size_t need = (size_t)pl.splits * (size_t)pl.m2 * (size_t)N * sizeof(float);
    for (int n = 8; n < device_cus(); n += 8)
        if (split_plan(M, N, K, n, &pl)) need = std::max(need, (size_t)pl.splits * (size_t)pl.m2 * (size_t)N * sizeof(float));
    return need;
}

extern "C" int ag_gemm_resid_split(const void* d_A, int64_t lda, const void* d_W, const float* d_bias, void* d_C, int64_t ldc,
                                   const void* d_R, int64_t ldr, int M, int N, int K, float* d_stats_out, void* d_scratch,
                                   size_t scratch_bytes, void* stream) {
    AG_REQUIRE(d_A && d_W && d_C && d_R && d_scratch, "ag_gemm_resid_split: null pointer");
    SplitPlan pl;
    hipStream_t s = (hipStream_t)stream;
    AG_REQUIRE(ag_gemm_big_eligible(M, N, K, lda, ldc, ldr, AG_EPI_BIAS_RESID) && split_plan(M, N, K, device_cus(), &pl),
               "ag_gemm_resid_split: M=%d N=%d K=%d does not split (ag_gemm_resid_split_scratch_bytes == 0: call ag_gemm)", M, N, K);
    AG_REQUIRE((N % 8) == 0 && (lda % 8) == 0 && (ldc % 8) == 0 && (ldr % 8) == 0 && ((uintptr_t)d_scratch % 16) == 0, "ag_gemm_resid_split: N, lda, ldc, ldr must be multiples of 8");
    AgProfScope prof(AG_EPI_BIAS_RESID, 2.0 * M * (double)N * K, ((double)M * K + (double)N * K + 2.0 * (double)M * N) * 2.0, s, nullptr, (double)M);
    // the rounds are those of THIS stream's CU budget (the target forward of the two-stream training epoch runs on 192 / 224 CUs: the
    // whole-device plan's "full rounds" are not whole there); a budget on which the tail does not split runs the plain kernel
    const int n_cu = stream_cus_of(s);
    if (n_cu != device_cus() && !split_plan(M, N, K, n_cu, &pl))
        return run_big(d_A, lda, d_W, d_bias, d_C, ldc, d_R, ldr, 1, 1, M, N, K, AG_EPI_BIAS_RESID, nullptr, nullptr, 0.f, d_stats_out,
                       nullptr, nullptr, nullptr, s);
    AG_REQUIRE(scratch_bytes >= (size_t)pl.splits * pl.m2 * N * sizeof(float), "ag_gemm_resid_split: scratch too small");
    if (pl.m1 > 0) {
        int rc = run_big(d_A, lda, d_W, d_bias, d_C, ldc, d_R, ldr, 1, 1, pl.m1, N, K, AG_EPI_BIAS_RESID, nullptr, nullptr, 0.f, d_stats_out,
                         nullptr, nullptr, nullptr, s, /*stats_rows=*/M);
        if (rc != AG_OK) return rc;
    }
    const int kp = K / pl.splits;
    const char* a2 = (const char*)d_A + (size_t)pl.m1 * lda * 2;
    int rc = run_big(a2, lda, d_W, nullptr, d_scratch, N, nullptr, 0, 1, 1, pl.m2, N, kp, AG_EPI_BIAS_F32, nullptr, nullptr, 0.f, nullptr,
                     nullptr, nullptr, nullptr, s, 0, pl.splits, /*ldw=*/K, /*bk_b=*/(long)kp * 2, /*bc=*/(long)pl.m2 * N);
    if (rc != AG_OK) return rc;
    const long items = (long)pl.m2 * ceil_div(N, BT);                   // half-waves
    const dim3 fgrid((unsigned)((items + 7) / 8)), fblock(256);
#define AG_FINISH(S_) hipLaunchKernelGGL(split_finish_kernel<S_>, fgrid, fblock, 0, s, (const float*)d_scratch, (long)pl.m2 * N, d_bias, \
                                         (const bf16_t*)d_R, (long)ldr, (bf16_t*)d_C, (long)ldc, pl.m1, M, N, d_stats_out, 2L * M)
    switch (pl.splits) {
        case 2: AG_FINISH(2); break;
        case 3: AG_FINISH(3); break;
        case 4: AG_FINISH(4); break;
        case 5: AG_FINISH(5); break;
        case 6: AG_FINISH(6); break;
        case 7: AG_FINISH(7); break;
        default: AG_FINISH(8); break;
    }
#undef AG_FINISH
    AG_LAUNCH_CHECK();
    return AG_OK;
}
