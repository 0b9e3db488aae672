// common.h -- shared host-side helpers for libastts.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>

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

// the kNN scan as one GEMM on the LDS-DMA ring kernel (ops_gemm.hip), row panels of one bank tile first
// (blockmax given: the scores leave scaled -- out = dot * col_scale (+ row_qs * col_bias) -- with the maximum of every (query, 64-row
// block) beside them; ring kernels only: ASTTS_ERR_INVALID when the shape is outside them)
int gemm_scan(const _Float16* queries, const _Float16* bank, float* out, int32_t qg, int64_t n, int32_t dp, int32_t ldc, hipStream_t st,
              const float* col_scale = nullptr, const float* col_bias = nullptr, const float* row_qs = nullptr, float* blockmax = nullptr,
              int32_t bm_ld = 0);

// an integer experiment switch from the environment (`dflt` when unset).  A set variable is reported once on stderr: these switches
// change which kernel form runs (same results), and a stray one in a user's environment should not go unnoticed (runtime.hip)
int exp_env_int(const char* name, int dflt);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one exp, one reciprocal and five FMAs instead of the
// branchy library erff -- the exact-erf GELU of the transformer FFN is evaluated 5.6M times per projection.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // v_rcp_f32 (1 ulp); __frcp_rn is a ten-instruction IEEE division
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

}  // namespace astts
