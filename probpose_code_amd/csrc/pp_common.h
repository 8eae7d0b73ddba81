// Shared host-side helpers of libprobpose_mi355x.so: status codes, thread-local error
// string, launch checking. gfx950 only -- no dual paths, no CUDA shims.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/probpose_mi355x.h"

namespace pp {

void set_error(const char* fmt, ...);
int option(const char* name);  // explicit dev switches (pp_set_option); the library never reads the environment
// Diagnostics (pp_launch_count): every kernel launch is tallied under its source file's name ("pp_winograd.hip") and, where a file holds
// several kernel families, under an explicit tag as well ("linear_dma_tile"). Host side, a few string compares per launch.
void count_launch(const char* file_or_tag);

inline int fail(int code, const char* what) {
    set_error("%s", what);
    return code;
}

// Argument check at the head of every entry point. It also drops whatever error an EARLIER runtime call of this
// thread (the caller's, e.g. a device probe of the framework around us) left in HIP's sticky per-thread slot, so that
// PP_LAUNCH_CHECK reports this entry point's own launches only.
#define PP_REQUIRE(cond, code, msg)          \
    do {                                     \
        (void)hipGetLastError();             \
        if (!(cond)) {                       \
            ::pp::set_error("%s", (msg));    \
            return (code);                   \
        }                                    \
    } while (0)

#define PP_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) {                                                        \
            ::pp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                            __FILE__, __LINE__);                                        \
            return PP_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

// Called after every kernel launch: surfaces launch-configuration errors without syncing.
#define PP_LAUNCH_CHECK()                  \
    do {                                   \
        ::pp::count_launch(__FILE__);      \
        PP_HIP_CHECK(hipGetLastError());   \
    } while (0)
#define PP_LAUNCH_CHECK_AS(tag)            \
    do {                                   \
        ::pp::count_launch(tag);           \
        PP_LAUNCH_CHECK();                 \
    } while (0)

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

}  // namespace pp
