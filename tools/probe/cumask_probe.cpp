// Dev probe: which physical CUs (XCC, SE, CU of HW_REG_HW_ID / HW_REG_XCC_ID) a stream created with hipExtStreamCreateWithCUMask runs on,
// for a few mask patterns.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probe/cumask_probe.cpp -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("CUs %d\n", ncu);
    const int n = 8192;
    uint32_t* d;
    hipMalloc(&d, n * 2 * sizeof(uint32_t));
    std::vector<uint32_t> h(n * 2);
    auto run = [&](const char* name, std::vector<uint32_t> mask) {
        hipStream_t s;
        hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: stream creation failed: %s\n", name, hipGetErrorString(e)); return; }
        hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, s, d, 2000);   // 20 us at 100 MHz
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, n * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
        std::map<int, std::set<int>> per_xcc;
        for (int i = 0; i < n; ++i) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
            const int cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        int tot = 0;
        printf("%-28s", name);
        for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
        printf("  total %d\n", tot);
        if (per_xcc.size() && tot <= 40) {
            for (auto& kv : per_xcc) { printf("    xcc%d:", kv.first); for (int v : kv.second) printf(" se%d.cu%d", v >> 8, v & 0xF); printf("\n"); }
        }
        hipStreamDestroy(s);
    };
    const int words = (ncu + 31) / 32;
    run("no mask", {});
    { std::vector<uint32_t> m(words, 0xFFFFFFFFu); run("all ones", m); }
    { std::vector<uint32_t> m(words, 0u); m[0] = 0xFFFFFFFFu; run("bits 0..31", m); }
    { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < ncu; i += 8) m[i / 32] |= 1u << (i % 32); run("every 8th bit", m); }
    { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < 8; ++i) m[0] |= 1u << i; run("bits 0..7", m); }
    { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < ncu; ++i) if (i < 224) m[i / 32] |= 1u << (i % 32); run("bits 0..223", m); }
    { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < ncu; ++i) if (i >= 224) m[i / 32] |= 1u << (i % 32); run("bits 224..255", m); }
    { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < ncu; ++i) if ((i / 8) % 8 == 7) m[i / 32] |= 1u << (i % 32); run("(i/8)%8==7", m); }
    return 0;
}
