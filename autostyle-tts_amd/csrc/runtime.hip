// runtime.hip -- error reporting, ABI version, and the bench-only launch profiler of libastts.so
#include "common.h"
#include "xlane.h"

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

namespace astts {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- launch profiler: HIP events on the launch stream around every launch of an enabled kind,
// plus the algorithmic work (flops or bytes) the caller attributes to the launch.
struct ProfKind {
    bool on = false;
    std::vector<hipEvent_t> ev;  // pairs
    size_t used = 0;
    double work = 0.0;
    int64_t dropped = 0;
};
static ProfKind g_prof[ASTTS_PROF_KINDS];
static bool g_prof_any = false;
static std::mutex g_prof_mu;   // launches of one kind come from several host threads in the pipelined benchmark

static thread_local size_t g_pair[ASTTS_PROF_KINDS];   // the event pair the calling thread's prof_begin reserved

bool prof_begin(int kind, hipStream_t st, double work) {
    if (!g_prof_any) return false;
    ProfKind& p = g_prof[kind];
    if (!p.on) return false;
    size_t idx;
    {
        std::lock_guard<std::mutex> lock(g_prof_mu);
        if (p.used + 2 > p.ev.size()) {
            ++p.dropped;
            return false;
        }
        idx = p.used;
        p.used += 2;
        p.work += work;
    }
    g_pair[kind] = idx;
    if (hipEventRecord(p.ev[idx], st) != hipSuccess) {
        (void)hipEventRecord(p.ev[idx + 1], st);      // keep the pair well-formed
        return false;
    }
    return true;
}

// Kernel-level timing (hipExtLaunchKernelGGL start/stop events: the dispatch's own begin / end timestamps, as rocprofv3 reads
// them, with no extra packet between dependent launches -- an hipEventRecord pair around a 4 us kernel adds ~3 us to it).
bool prof_events(int kind, double work, hipEvent_t* e0, hipEvent_t* e1) {
    if (!g_prof_any) return false;
    ProfKind& p = g_prof[kind];
    if (!p.on) return false;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (p.used + 2 > p.ev.size()) {
        ++p.dropped;
        return false;
    }
    *e0 = p.ev[p.used];
    *e1 = p.ev[p.used + 1];
    p.used += 2;
    p.work += work;
    return true;
}

int exp_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    static std::mutex mu;
    static std::vector<std::string> seen;
    {
        std::lock_guard<std::mutex> lock(mu);
        bool known = false;
        for (const auto& s : seen) known = known || s == name;
        if (!known) {
            seen.emplace_back(name);
            fprintf(stderr, "astts: experiment switch %s=%s is set in the environment (default %d)\n", name, e, dflt);
        }
    }
    return atoi(e);
}

void prof_end(int kind, hipStream_t st) {
    ProfKind& p = g_prof[kind];
    (void)hipEventRecord(p.ev[g_pair[kind] + 1], st);
}

}  // namespace astts

using namespace astts;

namespace astts {
// wall_clock64 ticks at 100 MHz on gfx9
// csrc/xlane.h against __shfl_xor: slots 0..5 = the six butterfly offsets, 6 = 64-lane sum (steps 32..1), 7 = maximum over the
// lanes with equal (lane & 7) (steps 8, 16, 32), 8 = sum in ascending steps 1, 2, 4 (as the decode attention's score reduction)
__global__ void xlane_selftest(unsigned seed, int* mismatches) {
    unsigned h = (threadIdx.x + blockIdx.x * blockDim.x) * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float v = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * (1.0f + (float)(h >> 28));
    asm volatile("" : "+v"(v));     // a value, not an expression: the shuffle form's first add must not be contracted with this product
    float ref[9], got[9];
    ref[0] = __shfl_xor(v, 1, 64); got[0] = lane_xor<1>(v);
    ref[1] = __shfl_xor(v, 2, 64); got[1] = lane_xor<2>(v);
    ref[2] = __shfl_xor(v, 4, 64); got[2] = lane_xor<4>(v);
    ref[3] = __shfl_xor(v, 8, 64); got[3] = lane_xor<8>(v);
    ref[4] = __shfl_xor(v, 16, 64); got[4] = lane_xor<16>(v);
    ref[5] = __shfl_xor(v, 32, 64); got[5] = lane_xor<32>(v);
    float s = v;
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    ref[6] = s; got[6] = wave_sum_desc(v);
    float m = v;
    for (int off = 8; off <= 32; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    ref[7] = m; got[7] = xmax<32>(xmax<16>(xmax<8>(v)));
    float t = v;
    t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
    ref[8] = t; got[8] = xadd<4>(xadd<2>(xadd<1>(v)));
#pragma unroll
    for (int i = 0; i < 9; ++i)
        if (__builtin_bit_cast(unsigned, ref[i]) != __builtin_bit_cast(unsigned, got[i])) atomicAdd(mismatches + i, 1);
}

__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace astts

extern "C" {

int astts_abi_version(void) { return ASTTS_ABI_VERSION; }

const char* astts_last_error_string(void) { return astts::g_err; }

int astts_prof_enable(int32_t kind, int32_t on, int32_t max_launches) {
    ASTTS_REQUIRE(kind >= 0 && kind < ASTTS_PROF_KINDS, ASTTS_ERR_INVALID, "astts_prof_enable: kind=%d", kind);
    ProfKind& p = g_prof[kind];
    if (on) {
        const size_t want = (size_t)(max_launches > 0 ? max_launches : 4096) * 2;
        while (p.ev.size() < want) {
            hipEvent_t e;
            ASTTS_CHECK_HIP(hipEventCreate(&e));
            p.ev.push_back(e);
        }
    }
    p.on = on != 0;
    p.used = 0;
    p.work = 0.0;
    p.dropped = 0;
    g_prof_any = false;
    for (auto& k : g_prof) g_prof_any = g_prof_any || k.on;
    return ASTTS_OK;
}

int astts_prof_read(int32_t kind, double* ms_sum, int64_t* launches, double* work_sum, int64_t* dropped) {
    ASTTS_REQUIRE(kind >= 0 && kind < ASTTS_PROF_KINDS, ASTTS_ERR_INVALID, "astts_prof_read: kind=%d", kind);
    ASTTS_REQUIRE(ms_sum && launches && work_sum, ASTTS_ERR_INVALID, "astts_prof_read: null argument");
    ProfKind& p = g_prof[kind];
    double sum = 0.0;
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        ASTTS_CHECK_HIP(hipEventSynchronize(p.ev[i + 1]));
        float ms = 0.f;
        ASTTS_CHECK_HIP(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
        sum += ms;
    }
    *ms_sum = sum;
    *launches = (int64_t)(p.used / 2);
    *work_sum = p.work;
    if (dropped) *dropped = p.dropped;
    p.used = 0;
    p.work = 0.0;
    p.dropped = 0;
    return ASTTS_OK;
}

int astts_stream_spin(int32_t microseconds, astts_stream_t stream) {
    ASTTS_REQUIRE(microseconds >= 0 && microseconds <= 100000, ASTTS_ERR_INVALID, "astts_stream_spin: microseconds=%d", microseconds);
    hipLaunchKernelGGL(astts::spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

/* A stream whose kernels run on a subset of the CUs (hipExtStreamCreateWithCUMask): bit i of mask[i / 32] enables CU i.
 * Experiments with CU partitions between the pipeline's stages (scripts/cu_mask_probe.py); destroyed by astts_stream_destroy. */
int astts_stream_create_cu_mask(const uint32_t* mask, int32_t n_words, astts_stream_t* out) {
    ASTTS_REQUIRE(mask && out && n_words >= 1 && n_words <= 32, ASTTS_ERR_INVALID, "astts_stream_create_cu_mask: bad arguments");
    hipStream_t st;
    ASTTS_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask));
    *out = (astts_stream_t)st;
    return ASTTS_OK;
}

int astts_stream_destroy(astts_stream_t stream) {
    ASTTS_REQUIRE(stream, ASTTS_ERR_INVALID, "astts_stream_destroy: null stream (the default stream is not the caller's to destroy)");
    ASTTS_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return ASTTS_OK;
}

int astts_selftest_xlane(int32_t* mismatches, astts_stream_t stream) {
    ASTTS_REQUIRE(mismatches, ASTTS_ERR_INVALID, "astts_selftest_xlane: null pointer");
    int* d = nullptr;
    ASTTS_CHECK_HIP(hipMalloc(&d, 9 * sizeof(int)));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d, 0, 9 * sizeof(int), st);
    for (unsigned seed = 1; e == hipSuccess && seed <= 4; ++seed) {
        hipLaunchKernelGGL(astts::xlane_selftest, dim3(64), dim3(512), 0, st, seed * 7919u, d);
        e = hipGetLastError();
    }
    int h[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) {
        astts::set_error("astts_selftest_xlane: %s", hipGetErrorString(e));
        return ASTTS_ERR_HIP;
    }
    int total = 0;
    for (int i = 0; i < 9; ++i) total += h[i];
    if (total)      // left in the error string for whoever looks (the call itself succeeds: the count is the result)
        astts::set_error("astts_selftest_xlane: mismatches per check (xor 1, 2, 4, 8, 16, 32, sum 32..1, max 8..32, sum 1..4): %d %d %d %d %d %d %d %d %d",
                         h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
    *mismatches = total;
    return ASTTS_OK;
}

/* `count` dependent busy-wait kernels of `microseconds` each on `stream`, one workgroup of `threads` threads per launch x `blocks`
 * workgroups: a stand-in for a launch chain (LM decode) when probing which streams really run side by side -- two streams
 * whose hardware queues sit on one command-processor pipe overlap long kernels but take turns at every kernel boundary. */
int astts_stream_chain(int32_t count, int32_t microseconds, int32_t blocks, astts_stream_t stream) {
    ASTTS_REQUIRE(count >= 1 && count <= 100000 && microseconds >= 0 && microseconds <= 100000 && blocks >= 1 && blocks <= 65536,
                  ASTTS_ERR_INVALID, "astts_stream_chain: count=%d microseconds=%d blocks=%d", count, microseconds, blocks);
    for (int i = 0; i < count; ++i) hipLaunchKernelGGL(astts::spin_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
