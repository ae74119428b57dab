// capi.cpp — error plumbing and device queries of the C ABI (host code).
#include <stdarg.h>
#include <string.h>

#include "common.h"

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
