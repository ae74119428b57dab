// attention.hip — fused masked multi-head attention for the K-mask surrogate forward.
//
// Replaces reference models/vanilla_vit.py:436-465 (scores * mask, i.e. a masked KEY keeps logit 0
// and still receives soft-max weight) and models/vanilla_bert.py:503-537 (scores + (1-mask)*f32min,
// i.e. a masked key gets exactly zero weight).  The reference materialises the [R,heads,T,T] score
// tensor four times per layer; here scores never leave registers.
//
// bf16 path (throughput mode), one workgroup per (row, head), one wave per 32-query block:
//   * K and V head slices ([T,64] bf16, 128-byte rows) are staged once into LDS (16-B chunks,
//     XOR-swizzled so both the row reads and the transposed reads are bank-conflict free).
//   * S^T = K·Q^T with v_mfma_f32_32x32x16_bf16 ("swapped" product): a lane owns one query column,
//     so the soft-max row reductions are in-register plus one cross-half shuffle.
//   * masking.  ViT (scores * mask): a masked key has logit exactly 0 for every query, so its K row is replaced by zeros at
//     staging (the LDS-DMA source of its chunks is a zero chunk) and the scores need no mask work at all.  BERT fixed-length
//     rows: mask bits (one uint32 per 32-key block) -> -inf, two VALU ops per score, before the running max.  Packed rows
//     hold visible keys only.  Padded keys (>= T): -inf.
//   * online soft-max with a lazily raised reference max (rescale only when a block beats it by 2^8), Q fragments requested
//     before the K/V staging, 16-byte output pieces built by a lane-half swap.
//   * the S^T accumulator tile, converted to bf16, is directly the B operand of the PV product
//     O^T += V^T·P^T (no LDS round trip); V^T fragments come from the row-major V image through
//     ds_read_b64_tr_b16.
// fp32 path (AG_F32 parity mode) and every head dim other than 64: plain VALU kernel, one query per thread,
// K/V tiles broadcast from LDS (attn_valu_kernel<T, D>).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int HD = 64;        // head dim of the MFMA kernel (backbones: 192/3, 768/12, 1024/16)
constexpr int ROWB = 128;     // bytes per K/V row in bf16
constexpr float NEG_BIG = -3.0e38f;
constexpr uint32_t NEG_BIG_BITS = 0xFF61B1E6u;  // bit pattern of -3.0e38f

// 16-B chunk slot swizzle: bijective on 8 consecutive same-parity rows (row reads) and sends rows
// r, r+2 to different 64-B halves (transposed reads) — see DESIGN.md "attention LDS image".
__device__ __forceinline__ int swz(int row) {
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1);
}

struct AttnArgs {
    const char* qkv;  // [R_src, T, 3H]
    const uint32_t* mask;  // [R, Tw]
    char* ctx;        // [R, T, H]
    int R, T, H, heads, share, mode, Tw, Tp, nq;
    float pdrop; uint32_t seed;  // fp32 training forward only
    int dbg;          // dev ablations (AG_ATTN_DBG): 1 = no compute, 2 = no K/V staging, 4 = no stores, 8 = phase stamps
    unsigned long long* stamps;  // [workgroup][wave][4] s_memrealtime (100 MHz) when dbg & 8
    const int* cu;    // packed (token-pruned) sequences: row r owns tokens [cu[r], cu[r+1]) of qkv / ctx, all visible; else null
};

// 16 zero bytes: the LDS-DMA source of every K chunk of a ViT-masked key (see the staging loop)
__device__ __attribute__((aligned(16))) const uint32_t g_zero_chunk[4] = {0u, 0u, 0u, 0u};

// One 32-key block of the online soft-max for one 32-query block.  kbyte = byte offset of the key block inside the K / V
// images (a literal in the unrolled form: it lands in the ds_read offset field).  voff carries the V image base.
//   * MASKOPS (BERT fixed-length rows only): masked logit := -inf, two VALU ops per score.  ViT rows need none: a masked
//     key's K row was zeroed at staging, so its logit is exactly +0.0 and it still competes in the soft-max, as
//     reference vanilla_vit.py:454 (scores * mask) has it.  Packed rows hold visible keys only.
//   * keys >= T (ragged last block, wave-uniform branch): -inf.
//   * the running max is raised only when a block beats it by more than TAU (2^8 in probability): the accumulators are
//     rescaled a handful of times per row instead of every block; l and O always share the same reference max, so the
//     quotient O/l is unchanged (probabilities up to 2^8 in between are exact in bf16's relative precision).
// S^T tile of one 32-key block: four 32x32x16 MFMAs over the 64-wide head dim (K fragments by rows from the K image)
template <typename QF>   // uint4, or a 4 x 32-bit vector (attn_stream3_kernel: whole 128-bit registers, never taken apart)
__device__ __forceinline__ f32x16_t attn_scores(const char* smem, const int kbyte, const int (&koff)[4], const QF (&qf)[4]) {
    f32x16_t s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(smem + koff[ks] + kbyte);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[ks]), s, 0, 0, 0);
    }
    return s;
}

template <int MODE, bool MASKOPS, bool FIRST, bool FENCE = true>
__device__ __forceinline__ void attn_softmax_pv(f32x16_t s, const char* smem, const int kbyte, const int (&voff)[2][2],
                                                f32x16_t& o0, f32x16_t& o1, float& m_run, float& l_run,
                                                const uint32_t mw, const int kvalid, const int lh, const float c2) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(4))) __bf16 b16x4;
    // key of register i: kb*32 + (i&3) + 8*(i>>2) + 4*lh
    if (MASKOPS) {
        const uint32_t mwl = mw >> (4 * lh);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = (i & 3) + 8 * (i >> 2);
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)mwl, kk, 1);
            s[i] = __uint_as_float((__float_as_uint(s[i]) & m) | (NEG_BIG_BITS & ~m));
        }
    }
    if (kvalid < 32) {  // wave-uniform
        const uint32_t vwl = ((1u << kvalid) - 1u) >> (4 * lh);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = (i & 3) + 8 * (i >> 2);
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)vwl, kk, 1);
            s[i] = __uint_as_float((__float_as_uint(s[i]) & m) | (NEG_BIG_BITS & ~m));
        }
    }
    float bmax = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) bmax = fmaxf(bmax, s[i]);
    {   // both lane halves hold the same query: combine their maxima (VALU half-swap, no LDS)
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(bmax), __float_as_uint(bmax), false, false);
        bmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (FIRST) {
        m_run = bmax;       // O = 0, l = 0: nothing to rescale
    } else {
        const float TAU = 8.0f / c2;   // raw-score units
        const bool need = bmax > m_run + TAU;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {   // wave-uniform; rare after the first blocks
            const float m_new = need ? bmax : m_run;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
    }
    const float mc = -m_run * c2;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[i], c2, mc));  // raw v_exp_f32; -inf logits give exactly 0
        s[i] = pv;
        psum += pv;
    }
    l_run += psum;
    // P^T fragments: regs 8st..8st+7 -> k-step st; element j <-> key 16st + 8(j>>2) + 4lh + (j&3)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        uint4 pf;
        pf.x = pack_bf16x2(s[8 * st + 0], s[8 * st + 1]);
        pf.y = pack_bf16x2(s[8 * st + 2], s[8 * st + 3]);
        pf.z = pack_bf16x2(s[8 * st + 4], s[8 * st + 5]);
        pf.w = pack_bf16x2(s[8 * st + 6], s[8 * st + 7]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            // V^T fragments via transposed reads (16-lane group g: d cols 16(g&1)+i, key half g>>1); k-step st is +2048 B
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + voff[dt][0] + kbyte + 2048 * st));
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + voff[dt][1] + kbyte + 2048 * st));
            const bf16x8_t vf = __builtin_shufflevector(__builtin_bit_cast(b16x4, v0), __builtin_bit_cast(b16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
            if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8_t, pf), o0, 0, 0, 0);
            else         o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8_t, pf), o1, 0, 0, 0);
        }
    }
    if (FENCE) __builtin_amdgcn_sched_barrier(0);   // keep the unrolled blocks from being interleaved into a spilling schedule
}

// The LAST key block of a row whose length leaves it at most 8 valid keys (T = 197 = 6 x 32 + 5: ViT).  Register i of lane half lh holds key
// (i & 3) + 8 (i >> 2) + 4 lh of the block: only registers 0-3 can be valid, every other score is -inf, its probability exactly 0 — so the exponent
// arguments, exponentials, sums, conversions and the -inf selects of twelve of the sixteen scores, and the second 16-key step of the PV product (all-zero
// probabilities), are not issued: ~25 vector instructions + 6 MFMAs instead of ~125 + 8 for the block that was the dearest of the seven.  Same values in the
// same order as attn_softmax_pv (adding exact zeros changes nothing): bit-identical outputs.
template <int MODE>
__device__ __forceinline__ void attn_softmax_pv_tail8(const f32x16_t s, const char* smem, const int kbyte, const int (&voff)[2][2],
                                                      f32x16_t& o0, f32x16_t& o1, float& m_run, float& l_run, const int kvalid, const int lh, const float c2) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(4))) __bf16 b16x4;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (4 * lh + i) < kvalid ? s[i] : NEG_BIG;
    float bmax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(bmax), __float_as_uint(bmax), false, false);
        bmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    {
        const float TAU = 8.0f / c2;
        const bool need = bmax > m_run + TAU;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
            const float m_new = need ? bmax : m_run;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
    }
    const float mc = -m_run * c2;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = __builtin_amdgcn_exp2f(fmaf(v[i], c2, mc));
        psum += v[i];
    }
    l_run += psum;
    uint4 pf;
    pf.x = pack_bf16x2(v[0], v[1]); pf.y = pack_bf16x2(v[2], v[3]); pf.z = 0u; pf.w = 0u;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + voff[dt][0] + kbyte));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + voff[dt][1] + kbyte));
        const bf16x8_t vf = __builtin_shufflevector(__builtin_bit_cast(b16x4, v0), __builtin_bit_cast(b16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
        if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8_t, pf), o0, 0, 0, 0);
        else         o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8_t, pf), o1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, bool MASKOPS, bool FIRST, typename QF>
__device__ __forceinline__ void attn_block(const char* smem, const int kbyte, const int (&koff)[4], const int (&voff)[2][2],
                                           const QF (&qf)[4], f32x16_t& o0, f32x16_t& o1, float& m_run, float& l_run,
                                           const uint32_t mw, const int kvalid, const int lh, const float c2) {
    const f32x16_t s = attn_scores(smem, kbyte, koff, qf);
    attn_softmax_pv<MODE, MASKOPS, FIRST>(s, smem, kbyte, voff, o0, o1, m_run, l_run, mw, kvalid, lh, c2);
}

#define ATTN_STAMP(i)                                                                                   \
    if (p.dbg & 8) {                                                                                    \
        unsigned long long t_;                                                                          \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        if (lane == 0) p.stamps[((long)blockIdx.x * 8 + wave) * 4 + (i)] = t_;                          \
    }

// NKB > 0: the key-block loop is unrolled for exactly NKB blocks (ViT's 197 tokens = 7 blocks); NKB = 0: runtime loop.
template <int MODE, bool MASKOPS, int NKB, bool TAIL8 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_bf16_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int row = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
    // fixed-length rows, or (p.cu) this row's packed token range: every packed token is a visible key
    int T = p.T, Tp = p.Tp, nq = p.nq;
    long tok_in = (long)(row / p.share) * p.T, tok_out = (long)row * p.T;
    if (p.cu) {
        const int t0 = p.cu[row];
        T = p.cu[row + 1] - t0;
        Tp = (T + 31) & ~31;
        nq = p.nq < p.T ? p.nq : T;    // CLS-only (nq = 1) or every token of the row
        tok_in = tok_out = t0;
    }
    char* ldsK = smem;
    char* ldsV = smem + Tp * ROWB;
    const long rowstride = (long)3 * p.H * 2;  // bytes per token in qkv
    const char* qbase = p.qkv + tok_in * rowstride + (long)head * HD * 2;
    const char* kbase = qbase + (long)p.H * 2;
    const char* vbase = qbase + (long)2 * p.H * 2;
    // this row's mask words: lane w holds word w (Tw <= 16), fetched once; v_readlane per key block
    const uint32_t mwords = p.cu ? 0xFFFFFFFFu : (lane < p.Tw ? p.mask[(long)row * p.Tw + lane] : 0u);

    ATTN_STAMP(0)
    const int lr = lane & 31, lh = lane >> 5;
    // Q fragments of this wave's first query block (B operand: lane holds Q[q][16ks + 8lh .. +8]) are requested BEFORE the
    // K/V staging so that their HBM round trip rides under it
    uint4 qf[4];
    {
        const int q0 = wave * 32 + lr;
        const int qc0 = q0 < nq ? q0 : T - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const uint4*>(qbase + (long)qc0 * rowstride + (2 * ks + lh) * 16);
    }

    // ---- stage K, V by LDS-DMA: 8-row x 128-B pieces (whole cache lines), the slot swizzle applied on the
    // per-lane SOURCE address so the LDS image stays lane-linear.  Rows >= T are clamped to row T-1 (finite
    // data; those keys get weight exactly 0 below), never out-of-bounds.  ViT: the K chunks of a masked key are
    // fetched from a zero chunk instead, which makes its logit exactly +0.0 for every query.
    {
        const int npieces = Tp >> 3;
        const int r_in = lane >> 3, slot = lane & 7;
        for (int pc = wave; pc < ((p.dbg & 2) ? 0 : npieces); pc += nwaves) {
            const int r = pc * 8 + r_in;
            const int rc = r < T ? r : T - 1;
            const long src = (long)rc * rowstride + ((slot ^ swz(r)) << 4);
            const char* ksrc = kbase + src;
            if (MODE == AG_MASK_VIT_MUL) {
                const uint32_t w = __builtin_amdgcn_readlane(mwords, pc >> 2);
                if (!((w >> (r & 31)) & 1u)) ksrc = reinterpret_cast<const char*>(g_zero_chunk);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ksrc,
                                             (__attribute__((address_space(3))) void*)(ldsK + pc * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + src),
                                             (__attribute__((address_space(3))) void*)(ldsV + pc * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ATTN_STAMP(1)

    const int nkb = Tp >> 5;             // 32-key blocks
    const int nqb = (p.dbg & 1) ? 0 : (nq + 31) >> 5;      // 32-query blocks
    // soft-max in base 2 on the raw scores: p = exp2(s*c - m*c), c = log2(e)/sqrt(64); the 1/sqrt(d) scale
    // (exact power of two) is order-preserving, so the running max is tracked on the raw scores.
    const float c2 = 0.125f * 1.4426950408889634f;

    // per-lane LDS offsets that do not depend on the key block (swz only looks at key bits 1..3)
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = lr * ROWB + (((2 * ks + lh) ^ swz(lr)) << 4);
    int voff[2][2];  // [d tile][first/second 4-key group], V image base included; k-step 1 is +2048 B
    {
        const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int k0 = 4 * (g >> 1) + tq, k1 = k0 + 8;
            const int chunk = dt * 4 + 2 * (g & 1) + (tp >> 1);
            voff[dt][0] = Tp * ROWB + k0 * ROWB + ((chunk ^ swz(k0)) << 4) + 8 * (tp & 1);
            voff[dt][1] = Tp * ROWB + k1 * ROWB + ((chunk ^ swz(k1)) << 4) + 8 * (tp & 1);
        }
    }

    for (int qb = wave; qb < nqb; qb += nwaves) {
        int q = qb * 32 + lr;
        const bool qvalid = q < nq;
        const int qc = qvalid ? q : T - 1;
        if (qb != wave) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[ks] = *reinterpret_cast<const uint4*>(qbase + (long)qc * rowstride + (2 * ks + lh) * 16);
        }

        f32x16_t o0, o1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
        float m_run = NEG_BIG, l_run = 0.f;

        if (NKB > 0) {
            attn_block<MODE, MASKOPS, true>(smem, 0, koff, voff, qf, o0, o1, m_run, l_run,
                                            MASKOPS ? __builtin_amdgcn_readlane(mwords, 0) : 0u, NKB == 1 ? T : 32, lh, c2);
#pragma unroll
            for (int kb = 1; kb < NKB - 1; ++kb)
                attn_block<MODE, MASKOPS, false>(smem, kb * 32 * ROWB, koff, voff, qf, o0, o1, m_run, l_run,
                                                 MASKOPS ? __builtin_amdgcn_readlane(mwords, kb) : 0u, 32, lh, c2);
            if (NKB > 1) {
                constexpr int kb = NKB > 1 ? NKB - 1 : 1;
                if (TAIL8) {   // the short last block (the launcher knows T - 32 (NKB - 1) <= 8): attn_softmax_pv_tail8
                    const f32x16_t sl = attn_scores(smem, kb * 32 * ROWB, koff, qf);
                    attn_softmax_pv_tail8<MODE>(sl, smem, kb * 32 * ROWB, voff, o0, o1, m_run, l_run, T - kb * 32, lh, c2);
                } else {
                    attn_block<MODE, MASKOPS, false>(smem, kb * 32 * ROWB, koff, voff, qf, o0, o1, m_run, l_run,
                                                     MASKOPS ? __builtin_amdgcn_readlane(mwords, kb) : 0u, T - kb * 32, lh, c2);
                }
            }
        } else {
            int ko[4], vo[2][2];   // walked by one key block per iteration
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ko[ks] = koff[ks];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { vo[dt][0] = voff[dt][0]; vo[dt][1] = voff[dt][1]; }
            attn_block<MODE, MASKOPS, true>(smem, 0, ko, vo, qf, o0, o1, m_run, l_run,
                                            MASKOPS ? __builtin_amdgcn_readlane(mwords, 0) : 0u, T, lh, c2);
            for (int kb = 1; kb < nkb; ++kb) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ko[ks] += 32 * ROWB;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) { vo[dt][0] += 32 * ROWB; vo[dt][1] += 32 * ROWB; }
                attn_block<MODE, MASKOPS, false>(smem, 0, ko, vo, qf, o0, o1, m_run, l_run,
                                                 MASKOPS ? __builtin_amdgcn_readlane(mwords, kb) : 0u, T - kb * 32, lh, c2);
            }
        }
        float l_tot;
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
            l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float inv = 1.0f / l_tot;
        ATTN_STAMP(2)
        // accumulator regs 4g..4g+3 of lane (lr, lh) are d rows 8g + 4lh .. +3: a half-swap pairs them into 16-byte
        // pieces (lane half 0: d 16j .. 16j+7, half 1: d 16j+8 .. 16j+15), four 16-byte stores per lane
        uint4 piece[4];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const int g = 2 * (tj & 1);
            const f32x16_t& o = tj < 2 ? o0 : o1;
            const uint32_t x0 = pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv), x1 = pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            const uint32_t y0 = pack_bf16x2(o[4 * g + 4] * inv, o[4 * g + 5] * inv), y1 = pack_bf16x2(o[4 * g + 6] * inv, o[4 * g + 7] * inv);
            const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            piece[tj] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
        if (qvalid && !(p.dbg & 4)) {
            char* out = p.ctx + (tok_out + q) * p.H * 2 + (long)head * HD * 2;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                *reinterpret_cast<uint4*>(out + (tj >> 1) * 64 + (tj & 1) * 32 + lh * 16) = piece[tj];
        }
        if (p.dbg & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        ATTN_STAMP(3)
    }
}

// ---- attn_stream3_kernel: the ViT kernel above as ONE K/V request stream per CU, two wave teams, three LDS images -----------
//
// What the counters say about attn_bf16_kernel at the bench shape (profiles/r04_attention_sq_counters.json): per item and CU the
// memory system needs 4.5 us (321 us per launch with the block bodies removed), the vector issue 3.9 us (four waves per SIMD), and
// the kernel takes 6.4 us — a workgroup stages (4.0 us, nothing to issue), computes, stores and is replaced (1.8 us), and two
// workgroups per CU overlap those phases only pairwise.  The persistent forms of rounds 1-3 kept ONE 8-wave workgroup per CU (two
// waves per SIMD: the block body is then issue-bound) or three smaller ones; what none of them had is both at once: four waves per
// SIMD AND a K/V request in flight at every moment.  Here: grid = CU count, 16 waves = two TEAMS of 8 (7 query blocks + one wave
// that only stages), items j = 0, 1, 2, ... of the workgroup go to team j & 1 and LDS image j % 3 (K 200 rows + V 224 rows =
// 53 KiB; three of them fill the 160 KiB).  An item takes two SLOTS (key blocks 0-3 | key blocks 4-6, normalisation, stores); the
// teams run half an item apart, one workgroup barrier per slot:
//     slot            j-1             j              j+1             j+2
//     team j & 1      item j-2 (2nd)  item j (1st)   item j (2nd)    item j+2 (1st)
//     other team      item j-1 (1st)  item j-1 (2nd) item j+1 (1st)  item j+1 (2nd)
//     LDS-DMA         item j          item j+1       item j+2        item j+3         (issued by the team in its 2nd half)
// The image of item j+2 is the one item j-1 was read from: free at the barrier that opens slot j+1, requested right behind it by the
// team that will read it, waited for (counted: the output stores stay in flight) before the barrier that opens slot j+2.  Every
// LDS-DMA goes through inline asm (hipcc must not see it: a tracked LDS-DMA puts a vmcnt(0) in front of every ds_read behind a
// barrier); the Q fragments of a team's next item are requested at the top of the current one into a second register set (asynchronous
// loads hipcc does not see, whole 128-bit registers that are never taken apart before their wait); the mask words by scalar loads.  Rows >= 200 of the V images are zeroed once (a padded key's
// probability is exactly 0, its V row must only be finite); the last key block reads K rows 200-223 out of the V image (finite bits,
// logits replaced by -inf).
constexpr int S3_KROWS = 200, S3_VROWS = 224;
constexpr int S3_VBASE = S3_KROWS * ROWB;                  // 25 600
constexpr int S3_IMG = (S3_KROWS + S3_VROWS) * ROWB;       // 54 272
constexpr int S3_LDS = 3 * S3_IMG;                         // 162 816 of 163 840
constexpr int S3_PIECES = S3_KROWS / 8;                    // 25 pieces of 8 rows per operand

typedef unsigned int s3_q4 __attribute__((ext_vector_type(4)));   // one Q fragment: a whole 128-bit register
__device__ __forceinline__ void s3_glds(const char* gsrc, uint32_t lds_off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}
__device__ __forceinline__ int s3_lane() {   // the lane id from an opaque value: per-lane constants are re-made per item instead of living across the loop
    uint32_t ones = ~0u;
    asm volatile("" : "+s"(ones));
    return (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
}

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_stream3_kernel(AttnArgs p, int n_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MODE = AG_MASK_VIT_MUL;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int team = wave >> 3, tw = wave & 7;
    const int T = p.T;
    const long rowstride = (long)3 * p.H * 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int grid = (int)gridDim.x;
    const int nj = (n_items - (int)blockIdx.x + grid - 1) / grid;       // items of this workgroup: blockIdx.x + j * grid
    const float c2 = 0.125f * 1.4426950408889634f;

    {   // V rows [200, 224) of the three images: zero, once
        const int t = threadIdx.x;
        if (t < 3 * 192) {
            const int img = t / 192, o = t - img * 192;
            *reinterpret_cast<uint4*>(smem + img * S3_IMG + S3_VBASE + S3_KROWS * ROWB + o * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    auto item_of = [&](int j) { return (int)blockIdx.x + j * grid; };
    // mask words of item j's row by SCALAR loads (lgkmcnt: a vector load would sit in the vmcnt queue between the LDS-DMA pieces, and hipcc waits
    // for a tracked load with a count that drains whatever was issued behind it).  A query wave's two pieces lie in ONE 32-key word
    // (word tw >> 1), the staging wave's pieces 14-24 in words 3-6.  Waited for (lgkmcnt(0)) by the caller before stage().
    auto mask_word_async = [&](int j, int w) -> uint32_t {
        const int row = item_of(j) / p.heads;
        const uint32_t* mp = p.mask + ((long)row * p.Tw + w);
        uint32_t v;
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(mp) : "memory");
        return v;
    };
    // one piece pair (K with the masked keys' chunks from the zero chunk, V) of item j -> image j % 3
    auto stage_pair = [&](int j, int pc, uint32_t word, int ln) {
        const int item = item_of(j), row = item / p.heads, head = item - row * p.heads;
        const char* kb = p.qkv + (long)(row / p.share) * T * rowstride + (long)head * HD * 2 + (long)p.H * 2;
        const char* vb = kb + (long)p.H * 2;
        const uint32_t img = lds0 + (uint32_t)((j % 3) * S3_IMG);
        const int r = pc * 8 + (ln >> 3), slot = ln & 7;
        const int rc = r < T ? r : T - 1;
        const long src = (long)rc * rowstride + ((slot ^ swz(r)) << 4);
        const char* ksrc = kb + src;
        if (MODE == AG_MASK_VIT_MUL && !((word >> (r & 31)) & 1u)) ksrc = reinterpret_cast<const char*>(g_zero_chunk);
        s3_glds(ksrc, img + pc * 1024);
        s3_glds(vb + src, img + S3_VBASE + pc * 1024);
    };
    auto load_q = [&](int j, s3_q4 (&qf)[4]) {                            // B operand of the score product: lane holds Q[q][16 ks + 8 lh .. + 8]
        const int ln = s3_lane();
        const int lr = ln & 31, lh = ln >> 5;
        const int item = item_of(j), row = item / p.heads, head = item - row * p.heads;
        const char* qb = p.qkv + (long)(row / p.share) * T * rowstride + (long)head * HD * 2;
        const int q0 = tw * 32 + lr;
        const int qc = q0 < T ? q0 : T - 1;
        // (asynchronous loads hipcc does not see: a tracked load is waited for at its first use, at the top of the next item, with a count
        // that also drains the previous item's output stores; these are covered by the counted wait in front of the slot barrier and tied to
        // it by q_landed)
        const char* qp = qb + (long)qc * rowstride + lh * 16;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                     "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                     : "=&v"(qf[0]), "=&v"(qf[1]), "=&v"(qf[2]), "=&v"(qf[3]) : "v"(qp) : "memory");
    };
    auto q_landed = [&](s3_q4 (&qf)[4]) { asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3])); };

    // ---- prologue: the team's first item on its way, its Q fragments, the mask words of the item the team stages first inside the loop
    s3_q4 qf[4] = {};
    uint32_t mw[4] = {0u, 0u, 0u, 0u};                                    // query wave: mw[0]; staging wave: words 3-6
    auto request_mask = [&](int j) {
        if (tw < 7) mw[0] = mask_word_async(j, tw >> 1);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) mw[i] = mask_word_async(j, 3 + i);
        }
    };
    auto mask_landed = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(mw[0]), "+s"(mw[1]), "+s"(mw[2]), "+s"(mw[3]) :: "memory");
    };
    auto stage = [&](int j) {                                             // this wave's share of item j: query waves two piece pairs, the staging wave eleven
        const int ln = s3_lane();
        if (tw < 7) {
            stage_pair(j, 2 * tw, mw[0], ln);
            stage_pair(j, 2 * tw + 1, mw[0], ln);
        } else {
#pragma unroll
            for (int pc = 14; pc < S3_PIECES; ++pc) stage_pair(j, pc, mw[(pc >> 2) - 3], ln);
        }
    };
    if (team < nj) {
        request_mask(team);
        mask_landed();
        stage(team);                                                      // (team t stages item t: read by nobody else before t's own barrier)
        if (tw < 7 && tw * 32 < p.nq) load_q(team, qf);
        if (team + 2 < nj) request_mask(team + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    q_landed(qf);
    __syncthreads();                                                      // opens slot 0 (and publishes the zeroed V rows)
    if (team == 1) asm volatile("s_barrier" ::: "memory");               // team 1 has nothing in slot 0

    // (a query wave whose block holds no query — the CLS-only last layer: n_query = 1 — only stages its share, like the team's staging wave: that
    // launch is then bound by its K/V stream alone, 330 -> 165 us)
    if (tw < 7 && tw * 32 < p.nq) {
        for (int j = team; j < nj; j += 2) {
            const bool more = j + 2 < nj;
            const int ln = s3_lane();
            const int lr = ln & 31, lh = ln >> 5;
            int koff[4], voff[2][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) koff[ks] = lr * ROWB + (((2 * ks + lh) ^ swz(lr)) << 4);
            {
                const int g = ln >> 4, li = ln & 15, tq = li >> 2, tp = li & 3;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int k0 = 4 * (g >> 1) + tq, k1 = k0 + 8;
                    const int chunk = dt * 4 + 2 * (g & 1) + (tp >> 1);
                    voff[dt][0] = S3_VBASE + k0 * ROWB + ((chunk ^ swz(k0)) << 4) + 8 * (tp & 1);
                    voff[dt][1] = S3_VBASE + k1 * ROWB + ((chunk ^ swz(k1)) << 4) + 8 * (tp & 1);
                }
            }
            {   // the image's LDS offset goes INTO the per-lane offsets, once per item, and the sums are made opaque: every K / V read of the item is then
                // one VGPR + a literal in the instruction's offset field (with the image base as a scalar hipcc re-made the address of each of the eight
                // transposed V reads of a block with a v_add_u32: 8 of a block's ~85 vector instructions)
                const int ib = (j % 3) * S3_IMG;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { koff[ks] += ib; asm volatile("" : "+v"(koff[ks])); }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    voff[dt][0] += ib; voff[dt][1] += ib;
                    asm volatile("" : "+v"(voff[dt][0]), "+v"(voff[dt][1]));
                }
            }
            f32x16_t o0, o1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
            float m_run = NEG_BIG, l_run = 0.f;
            // the team's next item's Q fragments: requested a whole item ahead into a second register set (16 of the 24 spare registers; requested
            // after the last score product instead — into the registers it releases — the round trip sat in front of the slot barrier: 396 -> 383 us)
            s3_q4 qn[4];
            load_q(more ? j + 2 : j, qn);                                 // (unconditional: a conditional load is a copy of the fragments at the loop edge)
            const bool compute = !(p.dbg & 1), staging = !(p.dbg & 2);   // (AG_ATTN_DBG ablations, wrong results: timing only)
            // ---- first half: key blocks 0-3
            if (compute) {
                attn_block<MODE, false, true>(smem, 0, koff, voff, qf, o0, o1, m_run, l_run, 0u, 32, lh, c2);
#pragma unroll
                for (int kb = 1; kb < 4; ++kb)
                    attn_block<MODE, false, false>(smem, kb * 32 * ROWB, koff, voff, qf, o0, o1, m_run, l_run, 0u, 32, lh, c2);
            }
            asm volatile("s_barrier" ::: "memory");                      // opens slot j+1: the other team has left image (j+2) % 3
            // ---- second half: the team's next item on its way, key blocks 4-6, normalisation, stores
            if (more && staging) { mask_landed(); stage(j + 2); }
            if (compute) {
#pragma unroll
                for (int kb = 4; kb < 6; ++kb)
                    attn_block<MODE, false, false>(smem, kb * 32 * ROWB, koff, voff, qf, o0, o1, m_run, l_run, 0u, 32, lh, c2);
                const f32x16_t s6 = attn_scores(smem, 6 * 32 * ROWB, koff, qf);
                attn_softmax_pv_tail8<MODE>(s6, smem, 6 * 32 * ROWB, voff, o0, o1, m_run, l_run, T - 192, lh, c2);   // (T <= 200: at most 8 valid keys)
            }
            request_mask(j + 4 < nj ? j + 4 : j);
            float l_tot;
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
                l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            const float inv = 1.0f / l_tot;
            uint4 piece[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const int g = 2 * (tj & 1);
                const f32x16_t& o = tj < 2 ? o0 : o1;
                const uint32_t x0 = pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv), x1 = pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                const uint32_t y0 = pack_bf16x2(o[4 * g + 4] * inv, o[4 * g + 5] * inv), y1 = pack_bf16x2(o[4 * g + 6] * inv, o[4 * g + 7] * inv);
                const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                piece[tj] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
            const int q = tw * 32 + lr;
            if (q < p.nq) {
                const int item = item_of(j), row = item / p.heads, head = item - row * p.heads;
                char* out = p.ctx + ((long)row * T + q) * p.H * 2 + (long)head * HD * 2;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
                    *reinterpret_cast<uint4*>(out + (tj >> 1) * 64 + (tj & 1) * 32 + lh * 16) = piece[tj];
            }
            // everything but this item's (at most four) output stores: the pieces of item j+2 and its Q fragments have landed
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            q_landed(qn);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
            asm volatile("s_barrier" ::: "memory");                      // opens slot j+2
        }
    } else {                                                             // the team's staging wave, query waves without a query
        for (int j = team; j < nj; j += 2) {
            const bool more = j + 2 < nj;
            asm volatile("s_barrier" ::: "memory");
            if (more && !(p.dbg & 2)) { mask_landed(); stage(j + 2); }
            request_mask(j + 4 < nj ? j + 4 : j);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    }
    if (team == (nj & 1)) asm volatile("s_barrier" ::: "memory");        // the barrier the other team's last slot pairs with
}

// ---- VALU kernel: one query per thread, 64-key K/V tiles (fp32) broadcast from LDS ----------------------
// T = float, D = 64: AG_F32 parity mode and the training forward (optional dropout).
// Any other head dim (the 8-wide heads of the LTT side network, reference models/ltt_vit.py:383-394: hidden 96
// over 12 heads) in either storage dtype: the contraction is too short for a matrix-core tile to pay.
constexpr int FKT = 64;
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_valu_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) float sK[FKT * D];
    __shared__ __attribute__((aligned(16))) float sV[FKT * D];
    const int tid = threadIdx.x;
    const int row = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
    int Tn = p.T, nq = p.nq;
    long tok_in = (long)(row / p.share) * p.T, tok_out = (long)row * p.T;
    if (p.cu) {   // packed (token-pruned) rows: every packed token is a visible key
        const int t0 = p.cu[row];
        Tn = p.cu[row + 1] - t0;
        nq = p.nq < p.T ? p.nq : Tn;
        tok_in = tok_out = t0;
    }
    const long ts = (long)3 * p.H;  // elements per token
    const T* base = reinterpret_cast<const T*>(p.qkv) + tok_in * ts + (long)head * D;
    const uint32_t* mrow = p.mask + (long)row * p.Tw;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);

    for (int q0 = 0; q0 < nq; q0 += blockDim.x) {
        const int q = q0 + tid;
        const bool qvalid = q < nq;
        float qv[D], o[D];
        const T* qp = base + (long)(qvalid ? q : 0) * ts;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            const float4 t = load4_as_f32(qp + d);
            qv[d] = t.x; qv[d + 1] = t.y; qv[d + 2] = t.z; qv[d + 3] = t.w;
            o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
        }
        float m_run = NEG_BIG, l_run = 0.f;
        for (int k0 = 0; k0 < Tn; k0 += FKT) {
            __syncthreads();
            for (int c = tid; c < FKT * (D / 4); c += blockDim.x) {
                const int r = c / (D / 4), ch = c % (D / 4);
                float4 kv = make_float4(0, 0, 0, 0), vv = kv;
                if (k0 + r < Tn) {
                    kv = load4_as_f32(base + (long)(k0 + r) * ts + p.H + ch * 4);
                    vv = load4_as_f32(base + (long)(k0 + r) * ts + 2 * p.H + ch * 4);
                }
                *reinterpret_cast<float4*>(sK + r * D + ch * 4) = kv;
                *reinterpret_cast<float4*>(sV + r * D + ch * 4) = vv;
            }
            __syncthreads();
            const int kn = min(FKT, Tn - k0);
            for (int kk = 0; kk < kn; ++kk) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < D; d += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(sK + kk * D + d);
                    s = fmaf(qv[d], t.x, s); s = fmaf(qv[d + 1], t.y, s); s = fmaf(qv[d + 2], t.z, s); s = fmaf(qv[d + 3], t.w, s);
                }
                s = s * inv_sqrt_d;
                const int key = k0 + kk;
                const bool on = p.cu ? true : ((mrow[key >> 5] >> (key & 31)) & 1u);
                if (p.mode == AG_MASK_VIT_MUL) s = on ? s : 0.f;
                else if (!on) continue;  // exactly zero weight
                const float m_new = fmaxf(m_run, s);
                const float alpha = expf(m_run - m_new);
                float pv = expf(s - m_new);
                l_run = l_run * alpha + pv;
                m_run = m_new;
                if (p.pdrop > 0.f)  // dropout on the normalised probabilities: dropped weights still count in l
                    pv = keep_elem(p.seed, ((uint64_t)blockIdx.x * p.T + q) * p.T + key, p.pdrop) ? pv / (1.0f - p.pdrop) : 0.f;
#pragma unroll
                for (int d = 0; d < D; d += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(sV + kk * D + d);
                    o[d] = fmaf(pv, t.x, o[d] * alpha); o[d + 1] = fmaf(pv, t.y, o[d + 1] * alpha);
                    o[d + 2] = fmaf(pv, t.z, o[d + 2] * alpha); o[d + 3] = fmaf(pv, t.w, o[d + 3] * alpha);
                }
            }
        }
        if (qvalid) {
            const float inv = 1.0f / l_run;
            T* out = reinterpret_cast<T*>(p.ctx) + (tok_out + q) * p.H + (long)head * D;
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                if (sizeof(T) == 4)
                    *reinterpret_cast<float4*>(out + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
                else
                    *reinterpret_cast<uint2*>(out + d) = make_uint2(pack_bf16x2(o[d] * inv, o[d + 1] * inv), pack_bf16x2(o[d + 2] * inv, o[d + 3] * inv));
            }
        }
    }
}

template <typename T>
int launch_valu(const AttnArgs& a, int head_dim, hipStream_t s) {
    const dim3 grid(a.R * a.heads), block(256);
    switch (head_dim) {
        case 8: hipLaunchKernelGGL((attn_valu_kernel<T, 8>), grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL((attn_valu_kernel<T, 16>), grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL((attn_valu_kernel<T, 32>), grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL((attn_valu_kernel<T, 64>), grid, block, 0, s, a); break;
        default: return ag_fail(AG_ERR_INVALID, "masked attention: head_dim %d not built (8, 16, 32, 64)", head_dim);
    }
    AG_LAUNCH_CHECK();
    return AG_OK;
}

// ---- bf16 MFMA kernel for narrow heads (D = 8 or 16: the LTT side network, reference models/ltt_vit.py:383-394) ----
// Same roles as attn_bf16_kernel (S^T = K.Q^T on 32x32x16, a lane owns a query, P^T reused as the PV B operand), with
// the head dim zero-padded to the 16-wide contraction of ONE MFMA per 32-key block, and O^T = V^T.P^T on a single
// 32-row d tile of which D rows are live.  K is staged row-major ([key][D] bf16), V transposed ([d][key]) by the staging
// threads, so the V^T fragments are two 8-byte reads per lane.  Per (query, key) pair the VALU work drops from ~36
// operations of the scalar kernel to the ~10 of the soft-max.
template <int MODE, int D>
__global__ __launch_bounds__(512) void attn_narrow_bf16_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int row = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
    int T = p.T, Tp = p.Tp, nq = p.nq;
    long tok_in = (long)(row / p.share) * p.T, tok_out = (long)row * p.T;
    if (p.cu) {
        const int t0 = p.cu[row];
        T = p.cu[row + 1] - t0;
        Tp = (T + 31) & ~31;
        nq = p.nq < p.T ? p.nq : T;
        tok_in = tok_out = t0;
    }
    constexpr int KROW = D * 2;                 // bytes per key in the K image
    const int vstride = Tp * 2 + 16;            // bytes per d row of the V^T image (+16: rows land on different banks)
    char* ldsK = smem;
    char* ldsV = smem + p.Tp * KROW;            // (sized for the longest row)
    const long rowstride = (long)3 * p.H * 2;
    const char* qbase = p.qkv + tok_in * rowstride + (long)head * D * 2;
    const char* kbase = qbase + (long)p.H * 2;
    const char* vbase = qbase + (long)2 * p.H * 2;
    for (int k = tid; k < Tp; k += blockDim.x) {     // keys >= T: a finite copy of the last key; masked to -inf below
        const int kc = k < T ? k : T - 1;
#pragma unroll
        for (int c = 0; c < D / 8; ++c) {
            uint4 kv = *reinterpret_cast<const uint4*>(kbase + (long)kc * rowstride + c * 16);
            // ViT: a masked key has logit exactly 0 for every query (scores * mask): zero its K row, no per-score mask work
            if (MODE == AG_MASK_VIT_MUL && !p.cu && !((p.mask[(long)row * p.Tw + (kc >> 5)] >> (kc & 31)) & 1u)) kv = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(ldsK + k * KROW + c * 16) = kv;
            const uint4 v = *reinterpret_cast<const uint4*>(vbase + (long)kc * rowstride + c * 16);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<bf16_t*>(ldsV + (c * 8 + e) * vstride + k * 2) = (bf16_t)((w[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
        }
    }
    __syncthreads();

    const int nkb = Tp >> 5, nqb = (nq + 31) >> 5;
    const int lr = lane & 31, lh = lane >> 5;
    const uint32_t* mrow = p.mask + (long)row * p.Tw;
    const uint32_t mwords = p.cu ? 0xFFFFFFFFu : (lane < p.Tw ? mrow[lane] : 0u);
    const float c2 = 1.4426950408889634f / sqrtf((float)D);
    const bool live = lh * 8 < D;               // this lane half carries real head-dim elements of the QK operands
    const bool vlive = lr < D;                  // this lane's d row of the PV A operand is real

    for (int qb = wave; qb < nqb; qb += nwaves) {
        const int q = qb * 32 + lr;
        const bool qvalid = q < nq;
        const int qc = qvalid ? q : T - 1;
        uint4 qf = make_uint4(0, 0, 0, 0);
        if (live) qf = *reinterpret_cast<const uint4*>(qbase + (long)qc * rowstride + lh * 16);
        f32x16_t o0;
#pragma unroll
        for (int i = 0; i < 16; ++i) o0[i] = 0.f;
        float m_run = NEG_BIG, l_run = 0.f;
        for (int kb = 0; kb < nkb; ++kb) {
            uint4 kf = make_uint4(0, 0, 0, 0);
            if (live) kf = *reinterpret_cast<const uint4*>(ldsK + (kb * 32 + lr) * KROW + lh * 16);
            f32x16_t s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf), s, 0, 0, 0);
            // key of register i: kb*32 + (i&3) + 8*(i>>2) + 4*lh (as in attn_bf16_kernel): mask, ragged tail
            const uint32_t mw = __builtin_amdgcn_readlane(mwords, kb);
            const uint32_t mwl = mw >> (4 * lh);
            const int kvalid = T - kb * 32;
            float bmax = NEG_BIG;
            if (MODE == AG_MASK_BERT_ADD && !p.cu) {   // wave-uniform
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kk = (i & 3) + 8 * (i >> 2);
                    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)mwl, kk, 1);
                    s[i] = __uint_as_float((__float_as_uint(s[i]) & m) | (NEG_BIG_BITS & ~m));
                }
            }
            if (kvalid < 32) {
                const uint32_t vwl = ((1u << kvalid) - 1u) >> (4 * lh);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kk = (i & 3) + 8 * (i >> 2);
                    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)vwl, kk, 1);
                    s[i] = __uint_as_float((__float_as_uint(s[i]) & m) | (NEG_BIG_BITS & ~m));
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) bmax = fmaxf(bmax, s[i]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(bmax), __float_as_uint(bmax), false, false);
                bmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            if (kb == 0) {
                m_run = bmax;
            } else {   // the reference max is raised only when a block beats it by 2^8 (as attn_softmax_pv)
                const bool need = bmax > m_run + 8.0f / c2;
                if (__builtin_amdgcn_ballot_w64(need) != 0) {
                    const float m_new = need ? bmax : m_run;
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                    l_run *= alpha;
                    m_run = m_new;
#pragma unroll
                    for (int i = 0; i < 16; ++i) o0[i] *= alpha;
                }
            }
            const float mc = -m_run * c2;
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[i], c2, mc));
                s[i] = pv;
                psum += pv;
            }
            l_run += psum;
            // P^T fragments: regs 8st..8st+7 -> k-step st; element j <-> key 16st + 8(j>>2) + 4lh + (j&3)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uint4 pf;
                pf.x = pack_bf16x2(s[8 * st + 0], s[8 * st + 1]);
                pf.y = pack_bf16x2(s[8 * st + 2], s[8 * st + 3]);
                pf.z = pack_bf16x2(s[8 * st + 4], s[8 * st + 5]);
                pf.w = pack_bf16x2(s[8 * st + 6], s[8 * st + 7]);
                uint4 vf = make_uint4(0, 0, 0, 0);       // V^T[d = lr][the same 8 keys]
                if (vlive) {
                    const char* vr = ldsV + lr * vstride + (kb * 32 + 16 * st + 4 * lh) * 2;
                    const uint2 a = *reinterpret_cast<const uint2*>(vr), b = *reinterpret_cast<const uint2*>(vr + 16);
                    vf = make_uint4(a.x, a.y, b.x, b.y);
                }
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf), o0, 0, 0, 0);
            }
        }
        float l_tot;
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
            l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float inv = 1.0f / l_tot;
        if (qvalid) {   // accumulator register i of lane (lr = query, lh) is d row (i&3) + 8(i>>2) + 4lh
            char* out = p.ctx + (tok_out + q) * p.H * 2 + (long)head * D * 2;
#pragma unroll
            for (int g4 = 0; g4 < D / 8; ++g4) {
                const int d = 8 * g4 + 4 * lh;
                *reinterpret_cast<uint2*>(out + d * 2) =
                    make_uint2(pack_bf16x2(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv), pack_bf16x2(o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv));
            }
        }
    }
}

template <int D>
int launch_narrow(const AttnArgs& a, int mask_mode, hipStream_t s) {
    const size_t lds = (size_t)a.Tp * D * 2 + (size_t)D * ((size_t)a.Tp * 2 + 16);
    AG_REQUIRE(lds <= 64 * 1024, "masked attention: T=%d too long for the narrow-head LDS image", a.T);
    const int nqb = (a.nq + 31) / 32;
    int nwaves = nqb < 8 ? nqb : 8;
    if (nwaves < 2) nwaves = 2;
    if (mask_mode == AG_MASK_VIT_MUL) hipLaunchKernelGGL((attn_narrow_bf16_kernel<AG_MASK_VIT_MUL, D>), dim3(a.R * a.heads), dim3(nwaves * 64), lds, s, a);
    else hipLaunchKernelGGL((attn_narrow_bf16_kernel<AG_MASK_BERT_ADD, D>), dim3(a.R * a.heads), dim3(nwaves * 64), lds, s, a);
    AG_LAUNCH_CHECK();
    return AG_OK;
}

// dispatch shared by the fixed-length and the packed (varlen) entry points
int run_attention(const AttnArgs& a, int dtype, int hd, int mask_mode, hipStream_t s) {
    if (dtype != AG_BF16 && dtype != AG_F32) return ag_fail(AG_ERR_INVALID, "masked attention: bad dtype %d", dtype);
    if (dtype == AG_BF16 && hd == HD) {
        const size_t lds = (size_t)2 * a.Tp * ROWB;
        AG_REQUIRE(lds <= 160 * 1024, "masked attention: T=%d too long for the single-pass LDS image", a.T);
        const int nqb = (a.nq + 31) / 32;
        int nwaves = nqb < 8 ? nqb : 8;
        static const int waves_env = getenv("AG_ATTN_WAVES") ? atoi(getenv("AG_ATTN_WAVES")) : 0;
        if (waves_env > 0 && nwaves > waves_env) nwaves = waves_env;
        if (nwaves < 4) nwaves = 4;  // waves beyond the query blocks only help staging K/V
        // ViT: masked keys are zeroed K rows (no per-score mask work), 7 key blocks (T = 197) unrolled;
        // BERT fixed-length rows: per-score -inf; packed rows: visible keys only
        // ViT rows of 193-200 tokens (every token a query, or the first n_query), enough items for every CU: one K/V stream per CU (attn_stream3_kernel)
        static AgKnob k_s3("AG_ATTN_STREAM3");
        if (mask_mode == AG_MASK_VIT_MUL && !a.cu && a.Tp == 224 && a.T > 192 && a.T <= S3_KROWS && !(a.dbg & ~3) &&
            (int)k_s3.get(1) != 0) {
            int n_cu = ag_device_cus();
            const int sc = ag_stream_cus(s);
            if (sc > 0 && sc < n_cu) n_cu = sc;
            const long items = (long)a.R * a.heads;
            // items per CU from which the stream pays (tests: 1): n items per workgroup take n + 1 slots of ~2.7 us behind a ~4 us first fill, the
            // workgroup-per-item kernel ~6.4 us per item and CU: from three items on
            static AgKnob k_s3min("AG_ATTN_STREAM3_MIN");
            if (n_cu > 0 && items >= (long)k_s3min.get(3) * n_cu && items < 0x7FFFFFFFL) {
                static bool attr_set[16] = {};
                int dev = 0;
                AG_HIP_CHECK(hipGetDevice(&dev));
                if (dev >= 0 && dev < 16 && !attr_set[dev]) {
                    AG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_stream3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS));
                    attr_set[dev] = true;
                }
                if (dev >= 0 && dev < 16) {
                    hipLaunchKernelGGL(attn_stream3_kernel, dim3(n_cu), dim3(1024), S3_LDS, s, a, (int)items);
                    AG_LAUNCH_CHECK();
                    return AG_OK;
                }
            }
        }
        void (*kern)(AttnArgs);
        if (mask_mode == AG_MASK_VIT_MUL && !a.cu)
            kern = a.Tp == 224 ? (a.T - 192 <= 8 ? attn_bf16_kernel<AG_MASK_VIT_MUL, false, 7, true> : attn_bf16_kernel<AG_MASK_VIT_MUL, false, 7>)
                               : attn_bf16_kernel<AG_MASK_VIT_MUL, false, 0>;
        else if (a.cu) kern = attn_bf16_kernel<AG_MASK_BERT_ADD, false, 0>;
        else kern = attn_bf16_kernel<AG_MASK_BERT_ADD, true, 0>;
        AG_REQUIRE(!(mask_mode == AG_MASK_VIT_MUL && a.cu), "masked attention: packed rows are a BERT-mode path");
        if (lds > 64 * 1024)
            AG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const size_t lds_k = lds;
        if (a.dbg & 8) {   // dev: per-workgroup phase stamps, summarised on stderr
            AttnArgs b = a;
            const size_t nst = (size_t)a.R * a.heads * 8 * 4;
            AG_HIP_CHECK(hipMalloc(&b.stamps, nst * 8));
            AG_HIP_CHECK(hipMemsetAsync(b.stamps, 0, nst * 8, s));
            hipLaunchKernelGGL(kern, dim3(a.R * a.heads), dim3(nwaves * 64), lds_k, s, b);
            AG_HIP_CHECK(hipStreamSynchronize(s));
            unsigned long long* h = (unsigned long long*)malloc(nst * 8);
            AG_HIP_CHECK(hipMemcpy(h, b.stamps, nst * 8, hipMemcpyDeviceToHost));
            double ph[3] = {0, 0, 0}, life = 0; long n = 0; unsigned long long tmin = ~0ull, tmax = 0;
            for (long wg = 0; wg < (long)a.R * a.heads; ++wg) {
                unsigned long long w0 = ~0ull, w3 = 0;
                for (int w = 0; w < nwaves; ++w) {
                    const unsigned long long* t = h + (wg * 8 + w) * 4;
                    if (!t[0] || !t[3]) continue;
                    ph[0] += t[1] - t[0]; ph[1] += t[2] - t[1]; ph[2] += t[3] - t[2]; ++n;
                    if (t[0] < w0) w0 = t[0];
                    if (t[3] > w3) w3 = t[3];
                }
                if (w3) { life += w3 - w0; if (w0 < tmin) tmin = w0; if (w3 > tmax) tmax = w3; }
            }
            fprintf(stderr, "[attn stamps] waves %ld: stage %.2f us | compute %.2f us | store %.2f us | wg life %.2f us | kernel %.1f us\n",
                    n, ph[0] / n * 0.01, ph[1] / n * 0.01, ph[2] / n * 0.01, life / ((double)a.R * a.heads) * 0.01, (tmax - tmin) * 0.01);
            free(h); AG_HIP_CHECK(hipFree(b.stamps));
            return AG_OK;
        }
        hipLaunchKernelGGL(kern, dim3(a.R * a.heads), dim3(nwaves * 64), lds_k, s, a);
    } else if (dtype == AG_F32) {
        return launch_valu<float>(a, hd, s);
    } else if (hd == 8 && !getenv("AG_ATTN_VALU")) {
        return launch_narrow<8>(a, mask_mode, s);
    } else if (hd == 16 && !getenv("AG_ATTN_VALU")) {
        return launch_narrow<16>(a, mask_mode, s);
    } else {
        return launch_valu<bf16_t>(a, hd, s);   // other head dims in bf16 storage
    }
    AG_LAUNCH_CHECK();
    return AG_OK;
}

}  // namespace

extern "C" int ag_masked_attention(const void* d_qkv, const uint32_t* d_mask_bits, void* d_ctx, int R, int T, int H,
                                   int heads, int qkv_share, int mask_mode, int n_query, int dtype, void* stream) {
    if (R == 0) return AG_OK;
    AG_REQUIRE(d_qkv && d_mask_bits && d_ctx, "ag_masked_attention: null pointer");
    AG_REQUIRE(R >= 0 && T > 0 && heads > 0 && H % heads == 0, "ag_masked_attention: H=%d is not a multiple of heads=%d", H, heads);
    const int hd = H / heads;
    AG_REQUIRE(qkv_share >= 1 && R % qkv_share == 0, "ag_masked_attention: R=%d not a multiple of share=%d", R, qkv_share);
    AG_REQUIRE(mask_mode == AG_MASK_VIT_MUL || mask_mode == AG_MASK_BERT_ADD, "ag_masked_attention: bad mask mode %d", mask_mode);
    if (R == 0) return AG_OK;
    AttnArgs a;
    a.qkv = (const char*)d_qkv; a.mask = d_mask_bits; a.ctx = (char*)d_ctx;
    a.R = R; a.T = T; a.H = H; a.heads = heads; a.share = qkv_share; a.mode = mask_mode;
    a.Tw = (T + 31) / 32; a.Tp = a.Tw * 32;
    a.nq = (n_query > 0 && n_query < T) ? n_query : T;
    a.pdrop = 0.f; a.seed = 0; a.cu = nullptr;
    static const int dbg_env = getenv("AG_ATTN_DBG") ? atoi(getenv("AG_ATTN_DBG")) : 0;
    a.dbg = dbg_env; a.stamps = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const double es = dtype == AG_BF16 ? 2.0 : 4.0;
    AgProfScope prof(AG_PROF_ATTENTION, 4.0 * R * (double)a.nq * T * H,
                     ((double)(R / qkv_share) * T * 3 * H + (double)R * a.nq * H) * es, s);
    return run_attention(a, dtype, hd, mask_mode, s);
}

/* Packed (token-pruned) sequences: row r owns tokens [cu[r], cu[r+1]) of qkv [N,3H] / ctx [N,H]; every packed token
 * is a visible key (BERT: an additively masked key has exactly zero weight, so it is simply absent).  cls_only: only
 * the first token of each row is a query.  t_max bounds the row lengths (sizes the LDS image). */
extern "C" int ag_masked_attention_varlen(const void* d_qkv, const int* d_cu_seqlens, void* d_ctx, int R, int t_max, int H,
                                          int heads, int cls_only, int dtype, void* stream) {
    AG_REQUIRE(d_qkv && d_cu_seqlens && d_ctx, "ag_masked_attention_varlen: null pointer");
    AG_REQUIRE(R >= 0 && t_max > 0 && heads > 0 && H % heads == 0, "ag_masked_attention_varlen: bad shape");
    if (R == 0) return AG_OK;
    AttnArgs a;
    a.qkv = (const char*)d_qkv; a.mask = nullptr; a.ctx = (char*)d_ctx;
    a.R = R; a.T = t_max; a.H = H; a.heads = heads; a.share = 1; a.mode = AG_MASK_BERT_ADD;
    a.Tw = (t_max + 31) / 32; a.Tp = a.Tw * 32;
    a.nq = cls_only ? 1 : t_max;
    a.pdrop = 0.f; a.seed = 0; a.cu = d_cu_seqlens; a.dbg = 0; a.stamps = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const double es = dtype == AG_BF16 ? 2.0 : 4.0;
    // work is data dependent: account the dense bound scaled by 1/4 (half the keys x half the queries on Shapley masks)
    AgProfScope prof(AG_PROF_ATTENTION, (cls_only ? 2.0 : 1.0) * R * (double)a.nq * t_max * H,
                     ((double)R * t_max * 3 * H * 0.5 + (double)R * a.nq * H * (cls_only ? 1.0 : 0.5)) * es, s);
    return run_attention(a, dtype, H / heads, AG_MASK_BERT_ADD, s);
}

extern "C" int ag_masked_attention_train(const float* d_qkv, const uint32_t* d_mask_bits, float* d_ctx, int R, int T, int H,
                                         int heads, int mask_mode, float p_drop, uint32_t seed, void* stream) {
    AG_REQUIRE(d_qkv && d_mask_bits && d_ctx, "ag_masked_attention_train: null pointer");
    AG_REQUIRE(R >= 0 && T > 0 && heads > 0 && H % heads == 0 && p_drop >= 0.f && p_drop < 1.f, "ag_masked_attention_train: bad arguments");
    if (R == 0) return AG_OK;
    AttnArgs a;
    a.qkv = (const char*)d_qkv; a.mask = d_mask_bits; a.ctx = (char*)d_ctx;
    a.R = R; a.T = T; a.H = H; a.heads = heads; a.share = 1; a.mode = mask_mode;
    a.Tw = (T + 31) / 32; a.Tp = a.Tw * 32; a.nq = T; a.pdrop = p_drop; a.seed = seed; a.cu = nullptr; a.dbg = 0; a.stamps = nullptr;
    return launch_valu<float>(a, H / heads, (hipStream_t)stream);
}
