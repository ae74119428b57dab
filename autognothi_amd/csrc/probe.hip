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
