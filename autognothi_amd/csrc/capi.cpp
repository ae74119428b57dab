// capi.cpp — error plumbing and device queries of the C ABI (host code).
#include <stdarg.h>
#include <string.h>

#include "common.h"
#include <mutex>
#include <unordered_map>

static thread_local std::string g_last_error;

void ag_set_error(const std::string& msg) { g_last_error = msg; }

int ag_fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

extern "C" int ag_abi_version(void) { return AG_ABI_VERSION; }

std::atomic<long long> g_ag_launch_count{0};   // (launches come from any host thread: relaxed increments)
extern "C" int64_t ag_launch_count(void) { return (int64_t)g_ag_launch_count.load(std::memory_order_relaxed); }

// How many CUs the persistent large-M GEMM may take on a stream (ag_set_stream_cus: the two-stream training epoch, CU-masked streams).
// Everything else launches ordinary grids and needs no hint.
static std::mutex g_stream_cus_mu;
static std::unordered_map<void*, int> g_stream_cus;
extern "C" int ag_set_stream_cus(void* stream, int n_cu) {
    std::lock_guard<std::mutex> lk(g_stream_cus_mu);
    if (n_cu > 0) g_stream_cus[stream] = n_cu; else g_stream_cus.erase(stream);
    return AG_OK;
}
int ag_stream_cus(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_stream_cus_mu);
    if (g_stream_cus.empty()) return 0;
    auto it = g_stream_cus.find((void*)s);
    return it == g_stream_cus.end() ? 0 : it->second;
}

int g_ag_knob_epoch = 1;
extern "C" int ag_reload_knobs(void) {
    ++g_ag_knob_epoch;
    return AG_OK;
}
extern "C" const char* ag_last_error(void) { return g_last_error.c_str(); }

extern "C" int ag_device_info(int device, int* cu_count, char* arch, size_t arch_len) {
    hipDeviceProp_t prop;
    AG_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return AG_OK;
}

// ---- per-launch event timing -------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfRec { int cls; double flops, bytes; hipEvent_t e0, e1; const int* dyn; double rows_upper; };
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
std::mutex g_prof_mu;
hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

AgProfScope::AgProfScope(int kernel_class, double flops, double bytes, hipStream_t s, const int* d_rows, double rows_upper) : idx(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{kernel_class, flops, bytes, take_event(), take_event(), d_rows, rows_upper};
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
    idx = (int)g_recs.size() - 1;
}
AgProfScope::~AgProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_recs[idx].e1, stream);
}

extern "C" int ag_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return AG_OK;
}

extern "C" int ag_profile_collect(int kernel_class, double* total_ms, double* total_flops, double* total_bytes, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    std::vector<ProfRec> keep;
    for (auto& r : g_recs) {
        if (r.cls != kernel_class) { keep.push_back(r); continue; }
        AG_HIP_CHECK(hipEventSynchronize(r.e1));
        float t = 0.f;
        AG_HIP_CHECK(hipEventElapsedTime(&t, r.e0, r.e1));
        double scale = 1.0;
        if (r.dyn) {   // the launch ran on *dyn rows of the rows_upper it was sized (and priced) for
            int actual = 0;
            AG_HIP_CHECK(hipMemcpy(&actual, r.dyn, sizeof(int), hipMemcpyDeviceToHost));
            if ((double)actual < r.rows_upper) scale = (double)actual / r.rows_upper;
        }
        ms += t; fl += r.flops * scale; by += r.bytes * scale; ++n;
        g_pool.push_back(r.e0); g_pool.push_back(r.e1);
    }
    g_recs.swap(keep);
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (total_bytes) *total_bytes = by;
    if (launches) *launches = n;
    return AG_OK;
}
