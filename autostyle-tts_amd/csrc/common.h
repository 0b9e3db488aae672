// common.h -- shared host-side helpers for libastts.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/astts.h"

namespace astts {

void set_error(const char* fmt, ...);

// bench-only launch profiler (runtime.hip); prof_begin returns true when the launch is being timed
bool prof_begin(int kind, hipStream_t st, double work);
void prof_end(int kind, hipStream_t st);
// kernel-level variant: a reserved (start, stop) pair for hipExtLaunchKernelGGL; false when the kind is not being profiled
bool prof_events(int kind, double work, hipEvent_t* e0, hipEvent_t* e1);

#define ASTTS_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            astts::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                             __FILE__, __LINE__);                                          \
            return ASTTS_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define ASTTS_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            astts::set_error(__VA_ARGS__);                                                 \
            return (code);                                                                 \
        }                                                                                  \
    } while (0)

// launch-path variant: checks the launch itself, never synchronises
#define ASTTS_CHECK_LAUNCH()                                                               \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess) {                                                            \
            astts::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),    \
                             __FILE__, __LINE__);                                          \
            return ASTTS_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

}  // namespace astts
