// decode_chain.hip -- the LM decode step as a stand-alone launch chain (no torch, no Python): what do the
// decode-step kernels of csrc/lm_step.hip cost per launch, eager vs hipGraph replay, one chain vs two concurrent ones?
//   hipcc -O3 --offload-arch=gfx950 decode_chain.hip -o decode_chain            (timing)
//   hipcc -O3 --offload-arch=gfx950 -DLM_STAMPS decode_chain.hip -o decode_chain_stamps   (in-kernel s_memrealtime stamps)
// Shapes: CosyVoice-300M LM body (d 1024, 16 heads, FFN 4096, 14 layers), batch B (default 8), ~185 prefix keys.
#include "../../autostyle-tts_amd/csrc/lm_step.hip"

#include <algorithm>
#include <chrono>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace astts {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
}
int exp_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }   // the library's experiment switches: plain env here
bool prof_begin(int, hipStream_t, double) { return false; }   // the library's bench-only launch profiler: off here
void prof_end(int, hipStream_t) {}
bool prof_events(int, double, hipEvent_t*, hipEvent_t*) { return false; }
}  // namespace astts
using namespace astts;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__global__ void fill_f16(_Float16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (_Float16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale, float offset) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffff) / 32768.0f - 1.0f) * scale + offset;
    }
}
__global__ void set_state(LmStep* s, int step, int pos0) { s->step = step; s->pos0 = pos0; s->eos_min = 1 << 30; }

template <typename T>
static T* dalloc(size_t n) {
    T* p;
    CK(hipMalloc(&p, n * sizeof(T)));
    return p;
}
static _Float16* rand16(size_t n, unsigned seed, float scale) {
    _Float16* p = dalloc<_Float16>(n);
    hipLaunchKernelGGL(fill_f16, dim3(1024), dim3(256), 0, 0, p, n, seed, scale);
    return p;
}
static float* rand32(size_t n, unsigned seed, float scale, float offset = 0.f) {
    float* p = dalloc<float>(n);
    hipLaunchKernelGGL(fill_f32, dim3(256), dim3(256), 0, 0, p, n, seed, scale, offset);
    return p;
}

struct Layer {
    _Float16 *wqkv, *wo, *w1, *w2, *pos, *kv;
    float *bqkv, *bo, *b1, *b2, *n1g, *n1b, *n2g, *n2b, *u, *v;
};
struct Model {
    int d = 1024, heads = 16, ffn = 4096, layers = 14, vocab = 4097, center = 2048, tmax = getenv("DC_TMAX") ? atoi(getenv("DC_TMAX")) : 512;
    std::vector<Layer> L;
    _Float16* head;
    float *head_b, *ag, *ab;
};
struct Ctx {   // per decode chain
    int b;
    float *x0, *x1, *q, *lg;
    _Float16 *ao, *ff;
    float *part_o, *part_ml;
    std::vector<_Float16*> kv;
    LmStep* st;
    unsigned long long* stamps = nullptr;
};

static Model make_model(int b) {
    Model m;
    unsigned s = 1;
    for (int l = 0; l < m.layers; ++l) {
        Layer y;
        y.wqkv = rand16((size_t)3 * m.d * m.d, s++, 0.03f);
        y.wo = rand16((size_t)m.d * m.d, s++, 0.03f);
        y.w1 = rand16((size_t)m.ffn * m.d, s++, 0.03f);
        y.w2 = rand16((size_t)m.d * m.ffn, s++, 0.015f);
        y.pos = rand16((size_t)(2 * m.center + 1) * m.d, s++, 0.5f);
        y.bqkv = rand32(3 * m.d, s++, 0.1f); y.bo = rand32(m.d, s++, 0.1f); y.b1 = rand32(m.ffn, s++, 0.1f); y.b2 = rand32(m.d, s++, 0.1f);
        y.n1g = rand32(m.d, s++, 0.1f, 1.f); y.n1b = rand32(m.d, s++, 0.1f); y.n2g = rand32(m.d, s++, 0.1f, 1.f); y.n2b = rand32(m.d, s++, 0.1f);
        y.u = rand32(m.d, s++, 0.1f); y.v = rand32(m.d, s++, 0.1f);
        m.L.push_back(y);
    }
    if (getenv("DC_HOTW")) {   // every layer reads layer 0's weight images: what an ideal L2 prefetch of the next launch's weights would buy
        const int nhot = atoi(getenv("DC_HOTW"));      // 1: all four projections; 2: only wo (2 MB)
        for (int l = 1; l < m.layers; ++l) {
            m.L[l].wo = m.L[0].wo;
            if (nhot == 1) { m.L[l].wqkv = m.L[0].wqkv; m.L[l].w1 = m.L[0].w1; m.L[l].w2 = m.L[0].w2; }
        }
    }
    m.head = rand16((size_t)4224 * m.d, s++, 0.03f);
    m.head_b = rand32(4224, s++, 0.1f); m.ag = rand32(m.d, s++, 0.1f, 1.f); m.ab = rand32(m.d, s++, 0.1f);
    return m;
}
static Ctx make_ctx(const Model& m, int b, unsigned seed) {
    Ctx c;
    c.b = b;
    c.x0 = rand32((size_t)b * m.d, seed + 1, 1.f); c.x1 = rand32((size_t)b * m.d, seed + 2, 1.f);
    c.q = rand32((size_t)b * m.d, seed + 3, 1.f); c.lg = rand32((size_t)b * m.vocab, seed + 4, 1.f);
    c.ao = rand16((size_t)b * m.d, seed + 5, 1.f); c.ff = rand16((size_t)b * m.ffn, seed + 6, 1.f);
    c.part_o = rand32((size_t)b * m.heads * 2 * 64, seed + 7, 1.f); c.part_ml = rand32((size_t)b * m.heads * 2 * 2, seed + 8, 1.f, 2.f);
    for (int l = 0; l < m.layers; ++l) c.kv.push_back(rand16((size_t)m.tmax * b * 2 * m.d, seed + 10 + l, 1.f));
    c.st = dalloc<LmStep>(1);
    return c;
}

// ---- synthetic "render" background (DC_BG=1): 256 workgroups x 512 threads of arithmetic, ~25 us per launch, back to back on
// its own stream, with a chosen LDS footprint (DC_BG_LDS_KB) and register footprint (DC_BG_REGS=1: ~256 VGPRs per wave): how much
// of the decode chain's slow-down inside the pipeline is co-residency?
template <int NREG>
__global__ __launch_bounds__(512, NREG > 64 ? 2 : 4) void bg_kernel(const float* __restrict__ in, int iters, float* sink) {
    extern __shared__ float bg_lds[];
    if (threadIdx.x == 0) bg_lds[0] = 1.0f;
    float r[NREG];
#pragma unroll
    for (int k = 0; k < NREG; ++k) r[k] = in[(threadIdx.x + k * 64) & 8191];
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < NREG; ++k) r[k] = r[k] * 1.0001f + r[(k + 1) % NREG];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NREG; ++k) s += r[k];
    if (s == 12345.678f) *sink = s;
}

static int g_ksplit = 1, g_lnplain = 0;
static KvLayout g_lay = {0, 0, 0, 0};
static int g_nostate = 0, g_pos = 305;   // DC_NOSTATE=1: positions as kernel arguments (no device-side step state)
// one decode step: 14 x (QKV, attention, Wo, W1, W2) + head.  Returns the number of launches.
static int enqueue_step(const Model& m, Ctx& c, hipStream_t st, int slot0) {
    int slot = slot0;
    float* x = c.x0;
    float* y = c.x1;
    const int d = m.d, b = c.b;
    auto G = [&]() {
        GemvArgs a;
        memset(&a, 0, sizeof(a));
        a.st = g_nostate ? nullptr : c.st; a.pos = g_pos; a.m = b; a.ln_eps = 1e-5f; a.stamps = c.stamps; a.stamp_slot = slot++;
        return a;
    };
    for (int l = 0; l < m.layers; ++l) {
        const Layer& L = m.L[l];
        GemvArgs a = G();   // LN1 + QKV -> q, K|V into the cache
        a.x = x; a.ldx = d; a.ln_g = L.n1g; a.ln_b = L.n1b; if (g_lnplain) { a.ln_g = nullptr; a.ln_b = nullptr; a.ln_plain = 1; } a.w = L.wqkv; a.bias = L.bqkv; a.out = c.q; a.ldo = d;
        a.kv = c.kv[l]; a.n_split = d; a.kv_t = g_lay.t; a.kv_b = g_lay.b; a.kv_h = g_lay.h; a.kv_v = g_lay.v; a.n = 3 * d; a.k = d; a.kpad = d;
        if (lm_gemv_launch(a, st)) exit(2);
        AttnArgs t;
        memset(&t, 0, sizeof(t));
        t.q = c.q; t.kv = c.kv[l]; t.postab = L.pos; t.bias_u = L.u; t.bias_v = L.v; t.out = c.ao; t.st = g_nostate ? nullptr : c.st; t.pos = g_pos; t.b = b; t.h = m.heads;
        t.ldq = d; t.ldo = d; t.ldp = d; t.center = m.center; t.d = d; t.kv_t = g_lay.t; t.kv_b = g_lay.b; t.kv_h = g_lay.h; t.kv_v = g_lay.v; t.scale = 0.125f; t.stamps = c.stamps; t.stamp_slot = slot++;
        t.ksplit = g_ksplit; t.part_o = c.part_o; t.part_ml = c.part_ml;
        if (lm_attn_launch(t, st)) exit(2);
        a = G();            // Wo + residual
        a.x = c.ao; a.x_mode = 1; a.ldx = d;
        if (g_ksplit == 2) { a.x = c.part_o; a.x2 = c.part_ml; a.x_mode = 2; } a.w = L.wo; a.bias = L.bo; a.res = x; a.ldr = d; a.out = y; a.ldo = d; a.n = d; a.k = d; a.kpad = d;
        if (lm_gemv_launch(a, st)) exit(2);
        a = G();            // LN2 + W1 + ReLU -> fp16 hidden
        a.x = y; a.ldx = d; a.ln_g = L.n2g; a.ln_b = L.n2b; if (g_lnplain) { a.ln_g = nullptr; a.ln_b = nullptr; a.ln_plain = 1; } a.w = L.w1; a.bias = L.b1; a.out16 = c.ff; a.ldo16 = m.ffn; a.relu = 1;
        a.n = m.ffn; a.k = d; a.kpad = d;
        if (lm_gemv_launch(a, st)) exit(2);
        a = G();            // W2 + residual
        a.x = c.ff; a.x_mode = 1; a.ldx = m.ffn; a.w = L.w2; a.bias = L.b2; a.res = y; a.ldr = d; a.out = x; a.ldo = d; a.n = d; a.k = m.ffn; a.kpad = m.ffn;
        if (lm_gemv_launch(a, st)) exit(2);
    }
    GemvArgs a = G();       // after_norm + head; advances the step
    a.x = x; a.ldx = d; a.ln_g = m.ag; a.ln_b = m.ab; if (g_lnplain) { a.ln_g = nullptr; a.ln_b = nullptr; a.ln_plain = 1; } a.w = m.head; a.bias = m.head_b; a.out = c.lg; a.ldo = m.vocab; a.n = m.vocab; a.k = d; a.kpad = d;
    a.advance = 1;
    if (lm_gemv_launch(a, st)) exit(2);
    if (g_nostate) g_pos = g_pos >= 440 ? 185 : g_pos + 1;
    return slot - slot0;
}

static double time_ms(hipStream_t st, const std::function<void()>& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char** argv) {
    const int b = argc > 1 ? atoi(argv[1]) : 8;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    const int pos0 = 185;
    if (getenv("DC_KSPLIT")) g_ksplit = atoi(getenv("DC_KSPLIT"));
    g_lay = getenv("DC_KVHM") && atoi(getenv("DC_KVHM")) ? KvLayout::head_major(16, getenv("DC_TMAX") ? atoi(getenv("DC_TMAX")) : 512) : KvLayout::time_major(b, 1024);
    printf("cache layout: %s\n", g_lay.t == 64 ? "head-major" : "time-major");
    if (getenv("DC_NOSTATE")) g_nostate = 1;
    if (getenv("DC_LNPLAIN")) g_lnplain = 1;
    lm_step_set_attrs();
    Model m = make_model(b);
    Ctx c0 = make_ctx(m, b, 100), c1 = make_ctx(m, b, 200);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    auto reset = [&](Ctx& c, hipStream_t st) { hipLaunchKernelGGL(set_state, dim3(1), dim3(1), 0, st, c.st, 0, pos0); };
#ifdef LM_STAMPS
    {
        const int nk = 14 * 5 + 1;
        c0.stamps = dalloc<unsigned long long>((size_t)nk * 1024 * 8);
        hipLaunchKernelGGL(set_state, dim3(1), dim3(1), 0, s0, c0.st, 120, pos0);
        for (int w = 0; w < 3; ++w) {   // warm, then the measured step (stamps of the last one survive)
            CK(hipMemsetAsync(c0.stamps, 0, (size_t)nk * 1024 * 8 * 8, s0));
            enqueue_step(m, c0, s0, 0);
        }
        CK(hipStreamSynchronize(s0));
        std::vector<unsigned long long> h((size_t)nk * 1024 * 8);
        CK(hipMemcpy(h.data(), c0.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        const char* names[5] = {"qkv", "attn", "wo", "w1", "w2"};
        unsigned long long prev_end = 0, first = 0;
        printf("per kernel (10 ns ticks -> us): gap = first block start - previous kernel's last stamp; then per stamp avg / max over blocks, relative to the kernel's first block start\n");
        for (int kx = 0; kx < nk; ++kx) {
            unsigned long long t0 = ~0ull;
            int nb = 0;
            for (int blk = 0; blk < 1024; ++blk) {
                const unsigned long long* s = &h[((size_t)kx * 1024 + blk) * 8];
                if (s[0]) { t0 = std::min(t0, s[0]); ++nb; }
            }
            if (!nb) continue;
            if (!first) first = t0;
            const bool attn = kx < 70 && kx % 5 == 1;
            const int ns = attn ? 4 : 6;
            unsigned long long last = 0;
            std::string line;
            char buf[128];
            for (int i = 0; i < ns; ++i) {
                double sum = 0; unsigned long long mx = 0; int cnt = 0;
                for (int blk = 0; blk < 1024; ++blk) {
                    const unsigned long long* s = &h[((size_t)kx * 1024 + blk) * 8];
                    if (s[0] && s[i]) { sum += (double)(s[i] - t0); mx = std::max(mx, s[i] - t0); ++cnt; last = std::max(last, s[i]); }
                }
                snprintf(buf, sizeof buf, "  s%d %.2f/%.2f", i, cnt ? sum / cnt / 100.0 : 0.0, mx / 100.0);
                line += buf;
            }
            if (kx < 10 || kx >= 65)
                printf("%-5s L%-2d blocks %4d  gap %.2f us |%s | start-to-last %.2f us\n", kx == 70 ? "head" : names[kx % 5], kx / 5, nb,
                       prev_end ? (double)((long long)t0 - (long long)prev_end) / 100.0 : 0.0, line.c_str(), (last - t0) / 100.0);
            prev_end = last;
        }
        printf("whole step (first stamp to last stamp): %.1f us\n", (prev_end - first) / 100.0);
        return 0;
    }
#endif
    // ---- each operator alone: a dependent chain of the same operator over the 14 layers' (cold) weights, per-launch period
    if (getenv("DC_OPS")) {
        hipStream_t st = s0;
        Ctx& c = c0;
        const int d = m.d;
        auto G = [&]() { GemvArgs a; memset(&a, 0, sizeof(a)); a.st = g_nostate ? nullptr : c.st; a.pos = 305; a.m = b; a.ln_eps = 1e-5f; return a; };
        auto run = [&](const char* name, const std::function<void(int)>& op) {
            hipLaunchKernelGGL(set_state, dim3(1), dim3(1), 0, st, c.st, getenv("DC_STEP") ? atoi(getenv("DC_STEP")) : 120, pos0);
            for (int i = 0; i < 28; ++i) op(i % 14);
            CK(hipStreamSynchronize(st));
            const int n = 14 * 40;
            const double t = time_ms(st, [&] { for (int i = 0; i < n; ++i) op(i % 14); });
            printf("  %-28s %.2f us per launch (incl. ~1.45 us boundary)\n", name, t * 1e3 / n);
        };
        printf("B=%d, single operators (step %d: %d keys):\n", b, getenv("DC_STEP") ? atoi(getenv("DC_STEP")) : 120, pos0 + 1 + (getenv("DC_STEP") ? atoi(getenv("DC_STEP")) : 120));
        run("qkv  (LN, n=3072, k=1024)", [&](int l) { const Layer& L = m.L[l]; GemvArgs a = G(); a.x = (l & 1) ? c.x1 : c.x0; a.ldx = d; a.ln_g = L.n1g; a.ln_b = L.n1b; if (g_lnplain) { a.ln_g = nullptr; a.ln_b = nullptr; a.ln_plain = 1; } a.w = L.wqkv; a.bias = L.bqkv; a.out = c.q; a.ldo = d;
            a.kv = c.kv[l]; a.n_split = d; a.kv_t = g_lay.t; a.kv_b = g_lay.b; a.kv_h = g_lay.h; a.kv_v = g_lay.v; a.n = 3 * d; a.k = d; a.kpad = d; if (lm_gemv_launch(a, st)) exit(2); });
        run("attn", [&](int l) { const Layer& L = m.L[l]; AttnArgs t; memset(&t, 0, sizeof(t));
            t.q = c.q; t.kv = c.kv[l]; t.postab = L.pos; t.bias_u = L.u; t.bias_v = L.v; t.out = c.ao; t.st = g_nostate ? nullptr : c.st; t.pos = 305; t.b = b; t.h = m.heads;
            t.ldq = d; t.ldo = d; t.ldp = d; t.center = m.center; t.d = d; t.kv_t = g_lay.t; t.kv_b = g_lay.b; t.kv_h = g_lay.h; t.kv_v = g_lay.v; t.scale = 0.125f; t.ksplit = g_ksplit; t.part_o = c.part_o; t.part_ml = c.part_ml; if (lm_attn_launch(t, st)) exit(2); });
        run("wo   (f16 x, n=1024, k=1024)", [&](int l) { const Layer& L = m.L[l]; GemvArgs a = G(); a.x = c.ao; a.x_mode = 1; a.ldx = d; if (g_ksplit == 2) { a.x = c.part_o; a.x2 = c.part_ml; a.x_mode = 2; } a.w = L.wo; a.bias = L.bo; a.res = c.x0; a.ldr = d; a.out = c.x1; a.ldo = d; a.n = d; a.k = d; a.kpad = d;
            if (lm_gemv_launch(a, st)) exit(2); });
        run("w1   (LN, n=4096, k=1024)", [&](int l) { const Layer& L = m.L[l]; GemvArgs a = G(); a.x = c.x1; a.ldx = d; a.ln_g = L.n2g; a.ln_b = L.n2b; if (g_lnplain) { a.ln_g = nullptr; a.ln_b = nullptr; a.ln_plain = 1; } a.w = L.w1; a.bias = L.b1; a.out16 = c.ff; a.ldo16 = m.ffn; a.relu = 1;
            a.n = m.ffn; a.k = d; a.kpad = d; if (lm_gemv_launch(a, st)) exit(2); });
        run("w2   (f16 x, n=1024, k=4096)", [&](int l) { const Layer& L = m.L[l]; GemvArgs a = G(); a.x = c.ff; a.x_mode = 1; a.ldx = m.ffn; a.w = L.w2; a.bias = L.b2; a.res = c.x1; a.ldr = d; a.out = c.x0; a.ldo = d; a.n = d; a.k = m.ffn; a.kpad = m.ffn;
            if (lm_gemv_launch(a, st)) exit(2); });
        run("head (LN, n=4097, k=1024)", [&](int l) { GemvArgs a = G(); a.x = c.x0; a.ldx = d; a.ln_g = m.ag; a.ln_b = m.ab; if (g_lnplain) { a.ln_g = nullptr; a.ln_b = nullptr; a.ln_plain = 1; } a.w = m.head; a.bias = m.head_b; a.out = c.lg; a.ldo = m.vocab; a.n = m.vocab; a.k = d; a.kpad = d;
            if (lm_gemv_launch(a, st)) exit(2); });
        return 0;
    }
    if (getenv("DC_BG")) {
        const size_t lds = (size_t)(getenv("DC_BG_LDS_KB") ? atoi(getenv("DC_BG_LDS_KB")) : 150) * 1024;
        const bool regs = getenv("DC_BG_REGS") && atoi(getenv("DC_BG_REGS"));
        const int wgs = getenv("DC_BG_WGS") ? atoi(getenv("DC_BG_WGS")) : 256;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bg_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bg_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        float* bin = rand32(8192, 77, 1.f);
        float* sink = dalloc<float>(1);
        hipStream_t sbg;
        CK(hipStreamCreateWithFlags(&sbg, hipStreamNonBlocking));
        const int nreg = getenv("DC_BG_NREG") ? atoi(getenv("DC_BG_NREG")) : (regs ? 192 : 32);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bg_kernel<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bg_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bg_kernel<160>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        auto one = [&]() {
            if (nreg >= 192) hipLaunchKernelGGL(bg_kernel<192>, dim3(wgs), dim3(512), lds, sbg, bin, 14, sink);
            else if (nreg >= 160) hipLaunchKernelGGL(bg_kernel<160>, dim3(wgs), dim3(512), lds, sbg, bin, 17, sink);
            else if (nreg >= 128) hipLaunchKernelGGL(bg_kernel<128>, dim3(wgs), dim3(512), lds, sbg, bin, 21, sink);
            else if (nreg >= 96) hipLaunchKernelGGL(bg_kernel<96>, dim3(wgs), dim3(512), lds, sbg, bin, 28, sink);
            else hipLaunchKernelGGL(bg_kernel<32>, dim3(wgs), dim3(512), lds, sbg, bin, 90, sink);
        };
        for (int i = 0; i < 10; ++i) one();
        CK(hipStreamSynchronize(sbg));
        const double each = time_ms(sbg, [&] { for (int i = 0; i < 50; ++i) one(); }) * 1e3 / 50;
        reset(c0, s0);
        for (int i = 0; i < 20; ++i) enqueue_step(m, c0, s0, 0);
        CK(hipStreamSynchronize(s0));
        reset(c0, s0);
        const int n_bg = (int)(steps * 1500.0 / each) + 100;          // enough to outlast the chain even at 1.5 ms per step
        std::thread th([&] { for (int i = 0; i < n_bg; ++i) one(); });
        std::this_thread::sleep_for(std::chrono::milliseconds(3));
        auto w0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) enqueue_step(m, c0, s0, 0);
        CK(hipStreamSynchronize(s0));
        auto w1 = std::chrono::steady_clock::now();
        th.join();
        const bool still = hipStreamQuery(sbg) == hipErrorNotReady;
        CK(hipStreamSynchronize(sbg));
        printf("B=%d chain beside a background of %d x 512-thread workgroups, %zu KB LDS, %d live values per lane, %.1f us per launch (%s): %.1f us per step\n", b, wgs,
               lds >> 10, nreg, each, still ? "outlasted the chain" : "ENDED EARLY", std::chrono::duration<double, std::micro>(w1 - w0).count() / steps);
        return 0;
    }
    // ---- eager
    reset(c0, s0);
    for (int i = 0; i < 20; ++i) enqueue_step(m, c0, s0, 0);
    CK(hipStreamSynchronize(s0));
    reset(c0, s0);
    int nl = 0;
    double host_eager = 0.0;
    double ms = time_ms(s0, [&] {
        auto h0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) nl = enqueue_step(m, c0, s0, 0);
        host_eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    });
    printf("B=%d eager:            %.1f us per step (%d launches, %.2f us per launch), keys %d..%d; HOST time of the enqueue loop: %.1f us per step = %.2f us per launch\n", b, ms * 1e3 / steps, nl, ms * 1e3 / steps / nl, pos0, pos0 + steps,
           host_eager / steps, host_eager / steps / nl);
    // ---- graph, R steps per replay
    for (int R : {1, 8}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        auto t0 = std::chrono::steady_clock::now();
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < R; ++r) enqueue_step(m, c0, s0, 0);
        CK(hipStreamEndCapture(s0, &g));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        auto t2 = std::chrono::steady_clock::now();
        reset(c0, s0);
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        reset(c0, s0);
        const int reps = steps / R;
        double host_us = 0.0;
        ms = time_ms(s0, [&] {
            auto h0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s0));
            host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
        });
        printf("B=%d graph R=%d:        %.1f us per step (capture %.2f ms, instantiate %.2f ms); HOST time of the launch loop: %.1f us per step = %.2f us per kernel node\n", b, R, ms * 1e3 / (reps * R),
               std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), host_us / (reps * R), host_us / (reps * R) / nl);
        if (R == 8) {
            // two chains, each with its own buffers, graph and stream, launched from two host threads
            hipGraph_t g1;
            hipGraphExec_t ge1;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
            for (int r = 0; r < R; ++r) enqueue_step(m, c1, s1, 0);
            CK(hipStreamEndCapture(s1, &g1));
            CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
            reset(c0, s0); reset(c1, s1);
            CK(hipDeviceSynchronize());
            auto w0 = std::chrono::steady_clock::now();
            std::thread th0([&] { for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0)); });
            std::thread th1([&] { for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge1, s1)); CK(hipStreamSynchronize(s1)); });
            th0.join(); th1.join();
            auto w1 = std::chrono::steady_clock::now();
            printf("B=%d graph R=8, TWO concurrent chains: %.1f us per step of each chain (wall)\n", b,
                   std::chrono::duration<double, std::micro>(w1 - w0).count() / (reps * R));
            // the same two chains eager from two host threads
            reset(c0, s0); reset(c1, s1);
            CK(hipDeviceSynchronize());
            w0 = std::chrono::steady_clock::now();
            std::thread e0([&] { for (int i = 0; i < steps; ++i) enqueue_step(m, c0, s0, 0); CK(hipStreamSynchronize(s0)); });
            std::thread e1([&] { for (int i = 0; i < steps; ++i) enqueue_step(m, c1, s1, 0); CK(hipStreamSynchronize(s1)); });
            e0.join(); e1.join();
            w1 = std::chrono::steady_clock::now();
            printf("B=%d eager, TWO concurrent chains:      %.1f us per step of each chain (wall)\n", b,
                   std::chrono::duration<double, std::micro>(w1 - w0).count() / steps);
            // N chains (3, 4, 6), each on its own stream and host thread: how far does chain-level concurrency scale?
            std::vector<Ctx> cs;
            std::vector<hipStream_t> ss(8);
            // DC_CUMASK=1: chain i runs on the CUs whose index is congruent to i modulo the number of chains of the 2-chain test
            // (even / odd); DC_CUMASK=2: contiguous halves.  Do two chains stack their workgroups on the same CUs otherwise?
            const int cumask = getenv("DC_CUMASK") ? atoi(getenv("DC_CUMASK")) : 0;
            for (int i = 0; i < 8; ++i) {
                cs.push_back(make_ctx(m, b, 300 + 100 * i));
                if (cumask && i < 2) {
                    uint32_t mask[8];
                    for (int w = 0; w < 8; ++w) {
                        if (cumask == 1) mask[w] = i == 0 ? 0x55555555u : 0xaaaaaaaau;
                        else if (cumask == 3) {      // both chains share the DC_CUMASK_N highest mask bits (bit b = CU b / 8 of XCC b % 8: cu_census)
                            const int n = getenv("DC_CUMASK_N") ? atoi(getenv("DC_CUMASK_N")) : 64;
                            mask[w] = 0u;
                            for (int bit = 0; bit < 32; ++bit) if (w * 32 + bit >= 256 - n) mask[w] |= 1u << bit;
                        }
                        else mask[w] = ((w < 4) == (i == 0)) ? 0xffffffffu : 0u;
                    }
                    CK(hipExtStreamCreateWithCUMask(&ss[i], 8, mask));
                } else {
                    CK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
                }
            }
            for (int rep = 0; rep < 2; ++rep)
            for (int nc : {2, 3, 4, 5, 6, 8}) {
                for (int i = 0; i < nc; ++i) reset(cs[i], ss[i]);
                CK(hipDeviceSynchronize());
                w0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int i = 0; i < nc; ++i)
                    th.emplace_back([&, i] { for (int k = 0; k < steps; ++k) enqueue_step(m, cs[i], ss[i], 0); CK(hipStreamSynchronize(ss[i])); });
                for (auto& t : th) t.join();
                w1 = std::chrono::steady_clock::now();
                const double us = std::chrono::duration<double, std::micro>(w1 - w0).count() / steps;
                printf("B=%d eager, %d concurrent chains: %.1f us per step of each chain (wall) = %.1f us per chain-step\n", b, nc, us, us / nc);
            }
        }
    }
    return 0;
}
