// C-ABI plumbing shared by all entry points: version, error string, device queries.
#include "pp_common.h"

#include <cstring>

namespace pp {

static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

}  // namespace pp

extern "C" {

int pp_abi_version(void) { return PP_ABI_VERSION; }

const char* pp_last_error(void) { return pp::g_last_error; }

const char* pp_status_string(int status) {
    switch (status) {
        case PP_OK: return "PP_OK";
        case PP_ERR_INVALID_ARG: return "PP_ERR_INVALID_ARG";
        case PP_ERR_UNSUPPORTED: return "PP_ERR_UNSUPPORTED";
        case PP_ERR_HIP: return "PP_ERR_HIP";
        case PP_ERR_WORKSPACE: return "PP_ERR_WORKSPACE";
        default: return "PP_ERR_UNKNOWN";
    }
}

int pp_device_cu_count(void) {
    int dev = 0;
    PP_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    return prop.multiProcessorCount;
}

}  // extern "C"
