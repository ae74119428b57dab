// probe.hip — measurement aid, not part of the data path: what the matrix cores of THIS board sustain.
// The MI355X MFMA peak quoted in the roofline (2.5 PFLOP/s dense bf16) assumes 2.4 GHz.  Under the board
// power cap a kernel that keeps every matrix core busy on random operands runs at a lower shader clock; this
// probe issues nothing but v_mfma_f32_16x16x32_bf16 from registers (two waves per SIMD, 16 independent
// accumulators, operands changing on every instruction), reports the FLOP/s it sustains and the effective
// shader clock (s_memtime ticks per s_memrealtime tick; the latter is a constant 100 MHz).
#include "common.h"
#include <vector>

namespace {

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// two bf16 in [-2, 2) with random mantissas (exponent field 0x3F or 0x3E or 0x3D..), as GEMM operands look
__device__ __forceinline__ uint32_t rand_bf16x2(uint32_t seed, bool zero) {
    if (zero) return 0u;
    const uint32_t h = hash32(seed);
    const uint32_t lo = (h & 0x807Fu) | ((0x7Cu + ((h >> 8) & 3u)) << 7);
    const uint32_t hi = ((h >> 16) & 0x807Fu) | ((0x7Cu + ((h >> 24) & 3u)) << 7);
    return lo | (hi << 16);
}

__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(int iters, int zero, float* sink, unsigned long long* clocks) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = make_uint4(rand_bf16x2(gid * 64 + i * 8 + 0, zero), rand_bf16x2(gid * 64 + i * 8 + 1, zero),
                           rand_bf16x2(gid * 64 + i * 8 + 2, zero), rand_bf16x2(gid * 64 + i * 8 + 3, zero));
        fb[i] = make_uint4(rand_bf16x2(gid * 64 + i * 8 + 4, zero), rand_bf16x2(gid * 64 + i * 8 + 5, zero),
                           rand_bf16x2(gid * 64 + i * 8 + 6, zero), rand_bf16x2(gid * 64 + i * 8 + 7, zero));
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[i]),
                                                                    __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123456.789f) sink[0] = s;   // keeps the accumulators alive
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

// the same measurement with v_mfma_f32_32x32x16_bf16 (2x2 tiles, 4 independent accumulators): same FLOPs per iteration
// per wave as four 16x16x32 rows of the kernel above would be -> iters are scaled by the caller
__global__ __launch_bounds__(512, 2) void mfma_probe32_kernel(int iters, int zero, float* sink, unsigned long long* clocks) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = make_uint4(rand_bf16x2(gid * 64 + i * 8 + 0, zero), rand_bf16x2(gid * 64 + i * 8 + 1, zero),
                           rand_bf16x2(gid * 64 + i * 8 + 2, zero), rand_bf16x2(gid * 64 + i * 8 + 3, zero));
        fb[i] = make_uint4(rand_bf16x2(gid * 64 + i * 8 + 4, zero), rand_bf16x2(gid * 64 + i * 8 + 5, zero),
                           rand_bf16x2(gid * 64 + i * 8 + 6, zero), rand_bf16x2(gid * 64 + i * 8 + 7, zero));
    }
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    unsigned long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[i]),
                                                                        __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 123456.789f) sink[0] = s;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}


// ---- LDS-DMA feed probe: what the global -> LDS path of one CU sustains with nothing else running --------------------------
// The ring GEMM moves 32 KiB per K=32 half-step and CU through 32 `global_load_lds_dwordx4` wave-instructions (16-row x 64-B
// pieces of two row-major operands).  This kernel issues exactly those requests — same addressing, same 4-slot ring, same
// counted vmcnt — from `waves` waves per workgroup (one workgroup per CU) and reports bytes per shader clock and CU.
//   flags bit 0: one s_barrier per half-step (as the GEMM's slot hand-over); bit 1: every workgroup reads the SAME 256-row
//   panels (L2-resident after the first touch) instead of its own (an HBM stream); bit 2: plain global_load_dwordx4 into
//   registers instead of LDS-DMA.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dma_probe_kernel(const char* A, const char* W, long ld_b, int nh, int flags, int panels,
                                                               unsigned long long* clocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PER = 16 / WAVES;                     // pieces of each operand per wave and half-step
    const int panel = (flags & 2) ? 0 : (int)(blockIdx.x % panels);
    const int jmask = (flags & 8) ? 1 : 0x7FFFFFFF;    // bit 3: every workgroup re-reads ITS OWN two half-steps (64 KiB per CU: L2-resident, distinct lines per CU)
    const char* tileA = A + (long)panel * 256 * ld_b;
    const char* tileW = W + (long)(panel % 3) * 256 * ld_b;
    // bit 4: pieces of 8 rows x 128 B (whole cache lines; a piece pair = the 16 rows of a K=64 step) instead of 16 rows x 64 B
    const bool full_lines = flags & 16;
    const int r_in = full_lines ? (lane >> 3) : (lane >> 2), chunk = full_lines ? (lane & 7) : (lane & 3);
    uint32_t off[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i)
        off[i] = full_lines ? (uint32_t)((((wave * PER + i) >> 1) * 16 + ((wave * PER + i) & 1) * 8 + r_in) * (int)ld_b + chunk * 16)
                            : (uint32_t)(((wave * PER + i) * 16 + r_in) * (int)ld_b + chunk * 16);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    u32x4v sink = {0u, 0u, 0u, 0u};
    for (int j = 0; j < nh; ++j) {
        const int slot = j & 3;
        const char* a = tileA + (full_lines ? (long)((j & jmask) >> 1) * 128 + (long)(j & 1) * 128 * ld_b : (long)(j & jmask) * 64);
        const char* w = ((flags & 8) ? tileA + 128 : tileW) + (full_lines ? (long)((j & jmask) >> 1) * 128 + (long)(j & 1) * 128 * ld_b : (long)(j & jmask) * 64);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t la = lds0 + slot * 32768 + (wave * PER + i) * 1024, lw = la + 16384;
            if (flags & 4) {
                u32x4v d0, d1;
                asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %4" : "=&v"(d0), "=&v"(d1) : "v"(off[i]), "s"(a), "s"(w) : "memory");
                sink ^= d0 ^ d1;    // (consumed only after the loop's waits: the xor below is ordered by the final vmcnt(0))
            } else {
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                             "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off[i]), "s"(a), "s"(w), "s"(la), "s"(lw) : "memory");
            }
        }
        // leave the two newest half-steps of this wave in flight (the GEMM's counted wait)
        if (PER == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (PER == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (flags & 1) asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if ((sink.x ^ sink.y ^ sink.z ^ sink.w) == 0x12345678u) clocks[0] = 1;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}


// ---- store probe: what the output path of one CU sustains for the GEMM epilogue's store shapes ------------------------------
// Every workgroup (8 waves) writes `tiles` 256 x 256 bf16 tiles (128 KiB each) of a row-major [rows, ld] matrix, nothing else.
//   shape 0: the ring GEMM's epilogue: a wave owns a 64-column strip; one store = 8 rows x 128 B (8 lanes x 16 B per row)
//   shape 1: a wave owns 32 whole tile rows; one store = 2 rows x 512 B (32 lanes x 16 B per row)
//   shape 2: as 0 with 4 rows x 256 B (a wave owns a 128-column strip)
//   shape 3 / 4: straight from the accumulator layout (no LDS transposition): 16 rows x 4 x 16 B / 16 rows x 64 B per store
//   flags bit 0: non-temporal stores
__global__ __launch_bounds__(512) void store_probe_kernel(char* C, long ld_b, int tiles_per_wg, int tiles_n, int shape, int flags,
                                                          unsigned long long* clocks) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 val = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + wave};
    unsigned long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        char* base = C + (long)tm * 256 * ld_b + (long)tn * 512;
        if (shape == 0) {            // wave (wm, wn): rows wm*128 .. +128, byte columns wn*128 .. +128; 16 stores of 8 rows
            const int wm = wave >> 2, wn = wave & 3;
            char* p0 = base + (long)(wm * 128 + (lane >> 3)) * ld_b + wn * 128 + (lane & 7) * 16;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                u32x4* dst = reinterpret_cast<u32x4*>(p0 + (long)i * 8 * ld_b);
                if (flags & 1) __builtin_nontemporal_store(val, dst); else *dst = val;
            }
        } else if (shape == 1) {     // wave w: rows w*32 .. +32, all 512 bytes; 16 stores of 2 rows
            char* p0 = base + (long)(wave * 32 + (lane >> 5)) * ld_b + (lane & 31) * 16;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                u32x4* dst = reinterpret_cast<u32x4*>(p0 + (long)i * 2 * ld_b);
                if (flags & 1) __builtin_nontemporal_store(val, dst); else *dst = val;
            }
        } else if (shape == 3 || shape == 4) {
            // the 16x16 MFMA accumulator layout with permuted W rows, NO transposition: lane (frow = lane & 15, fq = lane >> 4) owns 32
            // contiguous bytes of row frow (16 output features).  shape 3: two stores of 16 rows x (4 x 16 B at a 32-byte stride);
            // shape 4: after a half swap between fq and fq ^ 2, two stores of 16 rows x 64 contiguous bytes (lanes fq = 0, 2, 1, 3)
            const int wm = wave >> 2, wn = wave & 3, frow = lane & 15, fq = lane >> 4;
            const int col = shape == 3 ? fq * 32 : ((fq & 1) * 32 + (fq >> 1) * 16);
            char* p0 = base + (long)(wm * 128 + frow) * ld_b + wn * 128 + col;
#pragma unroll 4
            for (int i = 0; i < 8; ++i) {
                u32x4* d0 = reinterpret_cast<u32x4*>(p0 + (long)i * 16 * ld_b);
                u32x4* d1 = reinterpret_cast<u32x4*>(p0 + (long)i * 16 * ld_b + (shape == 3 ? 16 : 64));
                if (flags & 1) { __builtin_nontemporal_store(val, d0); __builtin_nontemporal_store(val, d1); } else { *d0 = val; *d1 = val; }
            }
        } else {                     // wave (wm 0..3, wn 0..1): rows wm*64 .. +64, byte columns wn*256 .. +256; 16 stores of 4 rows
            const int wm = wave >> 1, wn = wave & 1;
            char* p0 = base + (long)(wm * 64 + (lane >> 4)) * ld_b + wn * 256 + (lane & 15) * 16;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                u32x4* dst = reinterpret_cast<u32x4*>(p0 + (long)i * 4 * ld_b);
                if (flags & 1) __builtin_nontemporal_store(val, dst); else *dst = val;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

}  // namespace

extern "C" int ag_probe_store(int shape, int flags, void* d_C, int64_t ld_bytes, int rows, int grid, double* bytes_per_clk_per_cu,
                              double* gbytes_per_s, void* stream) {
    AG_REQUIRE(d_C && bytes_per_clk_per_cu && gbytes_per_s && shape >= 0 && shape <= 4 && ld_bytes % 512 == 0 && rows % 256 == 0 && grid > 0,
               "ag_probe_store: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int tiles_n = (int)(ld_bytes / 512), tiles = (rows / 256) * tiles_n;
    const int per_wg = tiles / grid;
    AG_REQUIRE(per_wg >= 1, "ag_probe_store: fewer tiles than workgroups");
    unsigned long long* clocks = nullptr;
    AG_HIP_CHECK(hipMalloc((void**)&clocks, sizeof(unsigned long long) * 2 * grid));
    hipEvent_t e0, e1;
    AG_HIP_CHECK(hipEventCreate(&e0));
    AG_HIP_CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(store_probe_kernel, dim3(grid), dim3(512), 0, s, (char*)d_C, (long)ld_bytes, per_wg, tiles_n, shape, flags, clocks);
    AG_HIP_CHECK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(store_probe_kernel, dim3(grid), dim3(512), 0, s, (char*)d_C, (long)ld_bytes, per_wg, tiles_n, shape, flags, clocks);
    AG_HIP_CHECK(hipEventRecord(e1, s));
    AG_HIP_CHECK(hipEventSynchronize(e1));
    AG_LAUNCH_CHECK();
    float ms = 0.f;
    AG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * grid);
    AG_HIP_CHECK(hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (int i = 0; i < grid; ++i) cyc += (double)h[2 * i];
    const double bytes_wg = (double)per_wg * 131072.0;
    *bytes_per_clk_per_cu = bytes_wg / (cyc / grid);
    *gbytes_per_s = bytes_wg * grid / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(clocks);
    return AG_OK;
}

namespace {
}  // namespace

extern "C" int ag_probe_dma(int waves, int half_steps, int flags, const void* d_A, const void* d_W, int64_t ld_bytes, int panels,
                            double* bytes_per_clk_per_cu, double* gbytes_per_s, double* shader_ghz, void* stream) {
    AG_REQUIRE((waves == 4 || waves == 8 || waves == 16) && half_steps > 0 && d_A && d_W && panels > 0 && bytes_per_clk_per_cu && gbytes_per_s && shader_ghz,
               "ag_probe_dma: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, cus = 0;
    AG_HIP_CHECK(hipGetDevice(&dev));
    AG_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    static AgKnob k_grid("AG_PROBE_GRID");      // (experiment: fewer workgroups than CUs — is the limit per CU or per L2?)
    cus = (int)k_grid.get(cus);
    unsigned long long* clocks = nullptr;
    AG_HIP_CHECK(hipMalloc((void**)&clocks, sizeof(unsigned long long) * 2 * cus));
    const int lds = 4 * 32768;
    auto launch = [&]() {
        if (waves == 4) hipLaunchKernelGGL(dma_probe_kernel<4>, dim3(cus), dim3(256), lds, s, (const char*)d_A, (const char*)d_W, (long)ld_bytes, half_steps, flags, panels, clocks);
        else if (waves == 8) hipLaunchKernelGGL(dma_probe_kernel<8>, dim3(cus), dim3(512), lds, s, (const char*)d_A, (const char*)d_W, (long)ld_bytes, half_steps, flags, panels, clocks);
        else hipLaunchKernelGGL(dma_probe_kernel<16>, dim3(cus), dim3(1024), lds, s, (const char*)d_A, (const char*)d_W, (long)ld_bytes, half_steps, flags, panels, clocks);
    };
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_probe_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_probe_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_probe_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipEvent_t e0, e1;
    AG_HIP_CHECK(hipEventCreate(&e0));
    AG_HIP_CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) launch();
    AG_HIP_CHECK(hipEventRecord(e0, s));
    launch();
    AG_HIP_CHECK(hipEventRecord(e1, s));
    AG_HIP_CHECK(hipEventSynchronize(e1));
    AG_LAUNCH_CHECK();
    float ms = 0.f;
    AG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * cus);
    AG_HIP_CHECK(hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < cus; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double bytes_wg = (double)half_steps * 32768.0;
    *shader_ghz = real > 0 ? cyc / real * 0.1 : 0.0;
    *bytes_per_clk_per_cu = bytes_wg / (cyc / cus);
    *gbytes_per_s = bytes_wg * cus / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(clocks);
    return AG_OK;
}

namespace {
}  // namespace

extern "C" int ag_probe_mfma(int iters, int zero_operands, double* tflops, double* shader_ghz, void* stream) {
    AG_REQUIRE(iters > 0 && tflops && shader_ghz, "ag_probe_mfma: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, cus = 0;
    AG_HIP_CHECK(hipGetDevice(&dev));
    AG_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = cus;   // one 8-wave workgroup per CU: two waves per SIMD
    float* sink = nullptr;
    unsigned long long* clocks = nullptr;
    AG_HIP_CHECK(hipMalloc((void**)&sink, sizeof(float)));
    AG_HIP_CHECK(hipMalloc((void**)&clocks, sizeof(unsigned long long) * 2 * grid));
    hipEvent_t e0, e1;
    AG_HIP_CHECK(hipEventCreate(&e0));
    AG_HIP_CHECK(hipEventCreate(&e1));
    // warm the clocks / reach the sustained power state, then measure
    // zero_operands: bit 0 = all-zero operands, bit 1 = v_mfma_f32_32x32x16_bf16 instead of 16x16x32 (same FLOPs per iteration)
    void (*kern)(int, int, float*, unsigned long long*) = (zero_operands & 2) ? mfma_probe32_kernel : mfma_probe_kernel;
    const int zero = zero_operands & 1;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, s, iters, zero, sink, clocks);
    AG_HIP_CHECK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, s, iters, zero, sink, clocks);
    AG_HIP_CHECK(hipEventRecord(e1, s));
    AG_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    AG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * grid);
    AG_HIP_CHECK(hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < grid; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    *shader_ghz = real > 0 ? cyc / real * 0.1 : 0.0;   // ticks per 10 ns -> GHz
    const double flops = (double)grid * 8 * iters * 16 * (2.0 * 16 * 16 * 32);
    *tflops = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(sink); (void)hipFree(clocks);
    return AG_OK;
}

// One wave that does nothing for `microseconds` (s_memrealtime: the 100 MHz constant clock): the yes/no probe of whether two HIP streams
// run their kernels BESIDE each other (two such kernels, one per stream, take one kernel's time) or behind each other (two kernels' time) —
// scripts/common.TrainPartition checks its second stream with it before an epoch relies on the overlap.
__global__ void spin_kernel(int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < (long long)ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int ag_probe_spin(int microseconds, void* stream) {
    AG_REQUIRE(microseconds > 0 && microseconds <= 100000, "ag_probe_spin: 1 .. 100000 us");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, microseconds * 100);
    AG_LAUNCH_CHECK();
    return AG_OK;
}
