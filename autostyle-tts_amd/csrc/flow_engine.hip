// flow_engine.hip -- Euler solver of the flow-matching decoder, host side in C++.
//
// The reference's hot loop #3: ConditionalCFM.solve_euler -> ConditionalDecoder.forward
// (cosyvoice/flow/flow_matching.py, cosyvoice/flow/decoder.py [EXT]; reached from tts_with_rag.py:195).
// One solve = n_steps x (2 down + 12 mid + 2 up) U-Net blocks, each a ResnetBlock1D and 4 transformer
// blocks: ~520 launches per step.  A Python host needs as long to enqueue them (~10 us each) as the GPU
// needs to run them, which both caps the stage and starves the other pipeline threads of the GIL; this
// engine issues the same operator sequence (astts_op_gemm_ex / groupnorm_ex / layernorm_ex /
// attn_mha_ex / elementwise: bit-identical results to the operator-by-operator path) from C++.
//
// Data layout: activations [2b, t, C] row-major (time-major rows, channels contiguous), rows [0, b) the
// guided half, rows [b, 2b) the unguided half (mu / spk / cond read as zero).  The residual stream is
// fp32; tensors consumed only by MFMA operands (GroupNorm-1 / LayerNorm / qkv / attention / GELU
// outputs) are fp16.
#include "common.h"

#include <vector>

struct astts_flow {
    astts_flow_config_t cfg;
    struct Block {
        astts_flow_resnet_t res;
        std::vector<astts_flow_tfm_t> tfm;
        astts_weight_t resample;
        int kind;
    };
    std::vector<Block> down, mid, up;
};

namespace astts {

// h[r, :] = [x | mu | spk | cond] for the guided half, [x | 0 | 0 | 0] for the unguided half; 0 beyond lens.
__global__ __launch_bounds__(256) void flow_pack_input(const float* __restrict__ x, const float* __restrict__ mu,
                                                      const float* __restrict__ spk, const float* __restrict__ cond,
                                                      const int* __restrict__ lens2, float* __restrict__ out, int b, int t,
                                                      int mel) {
    const int c4 = 4 * mel;
    const int64_t total = (int64_t)2 * b * t * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4);
        const int64_t row = i / c4;
        const int bb = (int)(row / t), tt = (int)(row % t);
        const int src_b = bb < b ? bb : bb - b;
        const int part = c / mel, cc = c % mel;
        float v = 0.0f;
        if (tt < lens2[bb]) {
            const int64_t o = ((int64_t)src_b * t + tt) * mel + cc;
            if (part == 0) v = x[o];
            else if (bb < b) v = part == 1 ? mu[o] : (part == 2 ? spk[(int64_t)src_b * mel + cc] : cond[o]);
        }
        out[i] = v;
    }
}

// out[b, t, 0:C] = a[b * a_bs + t * C + c] (first t steps of a longer tensor), out[b, t, C:2C] = skip; 0 beyond lens.
__global__ __launch_bounds__(256) void flow_concat_skip(const float* __restrict__ a, int64_t a_bs, const float* __restrict__ skip,
                                                       const int* __restrict__ lens, float* __restrict__ out, int b2, int t, int c) {
    const int c4 = c / 4;
    const int64_t total = (int64_t)b2 * t * 2 * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % (2 * c4));
        const int64_t row = i / (2 * c4);
        const int bb = (int)(row / t), tt = (int)(row % t);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tt < lens[bb]) {
            if (cc < c4) v = reinterpret_cast<const float4*>(a + (int64_t)bb * a_bs + (int64_t)tt * c)[cc];
            else v = reinterpret_cast<const float4*>(skip + row * c)[cc - c4];
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// lens arrays of the 2b batch at both time resolutions (+ the per-step time value broadcast)
__global__ void flow_fill_lens(const int* __restrict__ lens, int* __restrict__ lens_full, int* __restrict__ lens_half, int b, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * b) {
        const int l = lens ? lens[i < b ? i : i - b] : t;
        lens_full[i] = l;
        lens_half[i] = (l + 1) / 2;
    }
}

__global__ void flow_fill_time(float* __restrict__ tv, float value, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tv[i] = value;
}

static inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : g);
}

}  // namespace astts

using namespace astts;

#define RUN(expr)                          \
    do {                                   \
        const int rc_ = (expr);            \
        if (rc_ != ASTTS_OK) return rc_;   \
    } while (0)

namespace {

constexpr int FLOW_MAX_STEPS = 32;   // Euler steps whose time embeddings / projections are precomputed (the reference runs 10)

struct Buffers {
    float *xin, *r1, *r2, *r3, *pool[6], *temb0, *temb1, *temb2, *tproj, *tv, *d, *up;   // temb* / tproj / tv: all Euler steps at once
    _Float16 *r1h, *n16, *qkv16, *a16, *f16;
    int *lens_full, *lens_half;
    void* gn_ws;
    size_t gn_ws_bytes;
    float *rstat1, *rstat2;   // astts_op_resnet_conv: per-tile GroupNorm statistics of conv 1 / conv 2
};

// one carve-up serves astts_flow_workspace_bytes (base == nullptr) and astts_flow_solve
size_t carve(const astts_flow* h, int b, int t, char* base, Buffers* B) {
    const astts_flow_config_t& c = h->cfg;
    const size_t b2 = 2 * (size_t)b, rows = b2 * t, C = c.channels, hd = (size_t)c.heads * 64;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? base + o : nullptr;
        o = align_up(o + bytes, 256);
        return p;
    };
    const size_t in_ch = 4 * (size_t)c.mel > 2 * C ? 4 * (size_t)c.mel : 2 * C;
    Buffers tmp;
    Buffers& X = B ? *B : tmp;
    X.xin = (float*)take(sizeof(float) * rows * in_ch);
    X.r1 = (float*)take(sizeof(float) * rows * C);
    X.r2 = (float*)take(sizeof(float) * rows * C);
    X.r3 = (float*)take(sizeof(float) * rows * C);
    for (int i = 0; i < 6; ++i) X.pool[i] = (float*)take(sizeof(float) * rows * C);
    X.up = (float*)take(sizeof(float) * b2 * ((size_t)t + 4) * 2 * C);
    X.r1h = (_Float16*)take(sizeof(_Float16) * rows * C);
    X.n16 = (_Float16*)take(sizeof(_Float16) * rows * C);
    X.qkv16 = (_Float16*)take(sizeof(_Float16) * rows * 3 * hd);
    X.a16 = (_Float16*)take(sizeof(_Float16) * rows * hd);
    X.f16 = (_Float16*)take(sizeof(_Float16) * rows * 4 * C);
    // the time path does not depend on x: it is evaluated for ALL Euler steps before the first one (FLOW_MAX_STEPS rows blocks)
    const size_t n_res = h->down.size() + h->mid.size() + h->up.size();
    X.temb0 = (float*)take(sizeof(float) * FLOW_MAX_STEPS * b2 * c.time_in);
    X.temb1 = (float*)take(sizeof(float) * FLOW_MAX_STEPS * b2 * c.time_dim);
    X.temb2 = (float*)take(sizeof(float) * FLOW_MAX_STEPS * b2 * c.time_dim);
    X.tproj = (float*)take(sizeof(float) * n_res * FLOW_MAX_STEPS * b2 * C);
    X.tv = (float*)take(sizeof(float) * FLOW_MAX_STEPS * b2);
    X.d = (float*)take(sizeof(float) * rows * c.mel);
    X.lens_full = (int*)take(sizeof(int) * b2);
    X.lens_half = (int*)take(sizeof(int) * b2);
    X.gn_ws_bytes = astts_op_groupnorm_workspace_bytes((int)b2, t, c.groups);
    X.gn_ws = take(X.gn_ws_bytes > 16 ? X.gn_ws_bytes : 16);
    X.rstat1 = (float*)take(sizeof(float) * astts_op_resnet_conv_stats_floats((int32_t)b2, t));
    X.rstat2 = (float*)take(sizeof(float) * astts_op_resnet_conv_stats_floats((int32_t)b2, t));
    return o;
}

struct Ctx {
    const astts_flow* h;
    Buffers B;
    int b2, T;
    bool full;
    astts_stream_t st;

    int gemm(const void* x, int x16, const astts_weight_t& w, const float* residual, void* out, int out16, int64_t m,
             int t_in, int t_out, int stride, int pad, int act) const {
        return astts_op_gemm_ex(x, x16, w.w, w.bias, residual, nullptr, out, out16, m, w.n, w.cin, w.cin_pad, w.taps, w.cin, w.n,
                                residual ? w.n : 0, t_in, t_out, stride, 1, pad, act, 1.0f, 0.1f, st);
    }
    int linear(const void* x, int x16, const astts_weight_t& w, const float* residual, void* out, int out16, int64_t m, int act) const {
        return gemm(x, x16, w, residual, out, out16, m, (int)m, (int)m, 1, 0, act);
    }
    int mask(float* x, const int* lens, int t, int c) const {
        if (full) return ASTTS_OK;
        return astts_op_elementwise(ASTTS_EL_MUL_ROWMASK, x, nullptr, nullptr, lens, x, (int64_t)b2 * t * c, t, c, 0.0f, 0.0f, st);
    }

    // ResnetBlock1D on x [b2, t, cin] (already masked) -> out [b2, t, C]
    // `tproj` [b2, C]: this block's time projection for the current Euler step (precomputed for all steps by astts_flow_solve)
    // `next_w` / `next_bytes`: the weight image the launch after this block reads first (its first transformer block's q|k|v image)
    int resnet(const astts_flow_resnet_t& r, const float* tproj, const float* x, const int* lens, int t, float* out,
               const void* next_w = nullptr, uint32_t next_bytes = 0) const {
        const int C = h->cfg.channels, G = h->cfg.groups;
        const int64_t rows = (int64_t)b2 * t;
        if (r.c1_frag && r.c2_frag && r.res_frag && astts_op_resnet_conv_supported(r.c1.cin, r.c1.n, G, r.c1.taps) &&
            astts_op_resnet_conv_supported(r.c2.cin, r.c2.n, G, r.c2.taps) && astts_op_resnet_conv_supported(r.res.cin, r.res.n, G, r.res.taps)) {
            // three launches, GroupNorm + Mish folded into the convolutions (ops_resnet_conv.hip)
            // every launch requests the NEXT one's (cold) weights into L2 while it runs
            auto wbytes = [](const astts_weight_t& w) { return (uint32_t)((size_t)w.n * w.taps * w.cin_pad * 2); };
            RUN(astts_op_resnet_conv_pf(x, r.c1_frag, r.c1.bias, B.r1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                        B.rstat1, lens, b2, t, r.c1.cin, r.c1.taps, 1e-5f, r.c2_frag, wbytes(r.c2), st));
            RUN(astts_op_resnet_conv_pf(B.r1, r.c2_frag, r.c2.bias, B.r2, B.rstat1, r.g1_w, r.g1_b, tproj, nullptr, nullptr, nullptr, nullptr,
                                        B.rstat2, lens, b2, t, r.c2.cin, r.c2.taps, 1e-5f, r.res_frag, wbytes(r.res), st));
            return astts_op_resnet_conv_pf(x, r.res_frag, r.res.bias, out, nullptr, nullptr, nullptr, nullptr, B.r2, B.rstat2, r.g2_w, r.g2_b,
                                           nullptr, lens, b2, t, r.res.cin, r.res.taps, 1e-5f, next_w, next_bytes, st);
        }
        RUN(gemm(x, 0, r.c1, nullptr, B.r1, 0, rows, t, t, 1, 1, ASTTS_ACT_NONE));
        RUN(astts_op_groupnorm_ex(B.r1, lens, r.g1_w, r.g1_b, tproj, B.r1h, 1, b2, t, C, G, 1e-5f, 1, B.gn_ws, B.gn_ws_bytes, st));
        RUN(gemm(B.r1h, 1, r.c2, nullptr, B.r2, 0, rows, t, t, 1, 1, ASTTS_ACT_NONE));
        RUN(astts_op_groupnorm_ex(B.r2, lens, r.g2_w, r.g2_b, nullptr, B.r3, 0, b2, t, C, G, 1e-5f, 1, B.gn_ws, B.gn_ws_bytes, st));
        return gemm(x, 0, r.res, B.r3, out, 0, rows, t, t, 1, 0, ASTTS_ACT_NONE);     // res_conv(x) + h
    }

    // BasicTransformerBlock: x (in p) -> result back in p, q is scratch of the same size
    // `next_w` / `next_bytes`: the weight image the launch AFTER this block reads first (the next block's q|k|v image, or the next
    // ResNet's first convolution): prefetched into L2 by the feed-forward launch
    int tfm(const astts_flow_tfm_t& w, float* p, float* q, const int* lens, int t, const void* next_w = nullptr, uint32_t next_bytes = 0) const {
        const int C = h->cfg.channels, heads = h->cfg.heads, hd = heads * 64;
        const int64_t rows = (int64_t)b2 * t;
        // ASTTS_FLOW_UNFUSED_ATTN_MIN_ROWS=n (experiment): at >= n rows the attention half runs as LayerNorm + one q|k|v GEMM + the flash kernel
        static const int64_t unfused_min = exp_env_int("ASTTS_FLOW_UNFUSED_ATTN_MIN_ROWS", 0);
        const bool fused_attn = w.qkv_frag && astts_op_tfm_attn_fused_supported(C, heads, t) && !(unfused_min > 0 && rows >= unfused_min);
        if (fused_attn) {      // LayerNorm + q|k|v + attention in one launch (ops_tfm_fused.hip)
            // the feed-forward launch that follows streams 1.25 MB of weights no launch has touched since the last Euler step:
            // requested into L2 from here
            const void* pf[3] = {w.wo_frag, w.w1_frag, w.w2_frag};
            const uint32_t pfb[3] = {(uint32_t)((size_t)C * hd * 2), (uint32_t)((size_t)w.w1.n * C * 2), (uint32_t)((size_t)C * w.w1.n * 2)};
            RUN(astts_op_tfm_attn_fused_pf(p, w.qkv_frag, w.qkv.bias, lens, B.a16, b2, heads, t, C, 1e-5f, 0.125f, pf, pfb,
                                           (w.wo_frag && w.w1_frag && w.w2_frag) ? 3 : 0, st));
        } else {
            RUN(astts_op_layernorm_ex(p, w.n1_w, w.n1_b, B.n16, 1, rows, C, C, C, 1e-5f, st));
            RUN(linear(B.n16, 1, w.qkv, nullptr, B.qkv16, 1, rows, ASTTS_ACT_NONE));
            RUN(astts_op_attn_mha_ex(B.qkv16, B.qkv16 + hd, B.qkv16 + 2 * hd, 1, lens, B.a16, 1, b2, heads, t, 3 * hd, 3 * hd, hd,
                                     0.125f, st));
        }
        const bool ffn = w.w1_frag && w.w2_frag && astts_op_tfm_ffn_fused_supported(C, w.w1.n);
        // output projection + residual + LayerNorm + W1 + GELU + W2 + residual in one launch (in place on p); faster than the
        // separate projection at every row count measured (5 504: 20.2 vs 26.8 us, 11 008: 36.6 vs 45.0, 44 032: 116 vs 128)
        if (ffn && w.wo_frag && (hd == 256 || hd == 512))
            return astts_op_tfm_ffn_fused_pf(p, w.w1_frag, w.w1.bias, w.w2_frag, w.w2.bias, p, rows, C, w.w1.n, 1e-5f, B.a16, w.wo_frag,
                                             w.wo.bias, hd, next_w, next_bytes, st);
        RUN(linear(B.a16, 1, w.wo, p, q, 0, rows, ASTTS_ACT_NONE));
        if (ffn)      // LayerNorm + W1 + GELU + W2 + residual: one launch
            return astts_op_tfm_ffn_fused(q, w.w1_frag, w.w1.bias, w.w2_frag, w.w2.bias, p, rows, C, w.w1.n, 1e-5f, nullptr, nullptr, nullptr, 0, st);
        RUN(astts_op_layernorm_ex(q, w.n3_w, w.n3_b, B.n16, 1, rows, C, C, C, 1e-5f, st));
        RUN(linear(B.n16, 1, w.w1, nullptr, B.f16, 1, rows, ASTTS_ACT_GELU));
        return linear(B.f16, 1, w.w2, q, p, 0, rows, ASTTS_ACT_NONE);
    }
};

}  // namespace

extern "C" {

int astts_flow_create(const astts_flow_config_t* cfg, const astts_flow_block_t* down, const astts_flow_block_t* mid,
                      const astts_flow_block_t* up, astts_flow_t** out) {
    ASTTS_REQUIRE(cfg && down && mid && up && out, ASTTS_ERR_INVALID, "astts_flow_create: null argument");
    ASTTS_REQUIRE(cfg->mel >= 1 && cfg->channels >= 64 && cfg->channels % 64 == 0 && cfg->heads >= 1 && cfg->groups >= 1 &&
                      cfg->channels % cfg->groups == 0, ASTTS_ERR_INVALID, "astts_flow_create: bad sizes");
    ASTTS_REQUIRE(cfg->n_down >= 1 && cfg->n_down <= 4 && cfg->n_up == cfg->n_down && cfg->n_mid >= 0, ASTTS_ERR_INVALID,
                  "astts_flow_create: n_down=%d n_mid=%d n_up=%d", cfg->n_down, cfg->n_mid, cfg->n_up);
    int n_half = 0;
    for (int i = 0; i < cfg->n_down; ++i) n_half += down[i].resample_kind == ASTTS_FLOW_RESAMPLE_DOWN;
    ASTTS_REQUIRE(n_half <= 1, ASTTS_ERR_UNSUPPORTED, "astts_flow_create: more than one stride-2 level is not implemented");
    ASTTS_REQUIRE(down[cfg->n_down - 1].resample_kind == ASTTS_FLOW_RESAMPLE_CONV && up[cfg->n_up - 1].resample_kind == ASTTS_FLOW_RESAMPLE_CONV,
                  ASTTS_ERR_UNSUPPORTED, "astts_flow_create: the last down / up block must end in the k=3 convolution");
    astts_flow* h = new astts_flow();
    h->cfg = *cfg;
    auto copy = [](const astts_flow_block_t* src, int n, std::vector<astts_flow::Block>& dst) {
        for (int i = 0; i < n; ++i) {
            astts_flow::Block blk;
            blk.res = src[i].res;
            blk.tfm.assign(src[i].tfm, src[i].tfm + src[i].n_tfm);
            blk.resample = src[i].resample;
            blk.kind = src[i].resample_kind;
            dst.push_back(blk);
        }
    };
    copy(down, cfg->n_down, h->down);
    copy(mid, cfg->n_mid, h->mid);
    copy(up, cfg->n_up, h->up);
    *out = h;
    return ASTTS_OK;
}

int astts_flow_destroy(astts_flow_t* h) {
    delete h;
    return ASTTS_OK;
}

size_t astts_flow_workspace_bytes(const astts_flow_t* h, int32_t b, int32_t t) {
    if (!h || b < 1 || t < 1) return 0;
    return carve(h, b, t, nullptr, nullptr);
}

int astts_flow_solve(astts_flow_t* h, float* x, const float* mu, const float* spk, const float* cond, const int32_t* lens,
                     int32_t b, int32_t t, int32_t n_steps, const float* t_host, const float* dt_host, float cfg_rate,
                     void* workspace, size_t workspace_bytes, astts_stream_t stream) {
    ASTTS_REQUIRE(h && x && mu && spk && cond && t_host && dt_host && workspace, ASTTS_ERR_INVALID, "astts_flow_solve: null argument");
    ASTTS_REQUIRE(b >= 1 && t >= 2 && n_steps >= 1, ASTTS_ERR_INVALID, "astts_flow_solve: b=%d t=%d n_steps=%d", b, t, n_steps);
    ASTTS_REQUIRE(workspace_bytes >= astts_flow_workspace_bytes(h, b, t) && ((uintptr_t)workspace & 255) == 0,
                  ASTTS_ERR_WORKSPACE, "astts_flow_solve: workspace too small or misaligned");
    const astts_flow_config_t& c = h->cfg;
    hipStream_t st = (hipStream_t)stream;
    Ctx k;
    k.h = h;
    k.b2 = 2 * b;
    k.T = t;
    k.full = lens == nullptr;
    k.st = stream;
    carve(h, b, t, (char*)workspace, &k.B);
    const Buffers& B = k.B;
    const int b2 = k.b2, C = c.channels, mel = c.mel;
    const int t_half = (t - 1) / 2 + 1;     // conv k=3, stride 2, pad 1

    hipLaunchKernelGGL(flow_fill_lens, dim3((b2 + 63) / 64), dim3(64), 0, st, lens, B.lens_full, B.lens_half, b, t);
    ASTTS_CHECK_LAUNCH();

    // ---- the time path of every Euler step, hoisted out of the dependent chain (it depends on the schedule only): time
    // embedding -> MLP -> Mish (every ResnetBlock1D applies Mish before its own projection) for all n_steps x 2B rows in one
    // pass, then ONE projection GEMM per ResNet block over all steps: ~20 launches per solve instead of 21 per step.
    // Solves with more than FLOW_MAX_STEPS steps evaluate it chunk by chunk (the workspace holds FLOW_MAX_STEPS steps).
    for (int s = 0; s < n_steps; ++s) {
      const int s0 = s - s % FLOW_MAX_STEPS, sl = s - s0;
      const int64_t trows = (int64_t)(n_steps - s0 < FLOW_MAX_STEPS ? n_steps - s0 : FLOW_MAX_STEPS) * b2;
      if (sl == 0) {
        for (int j = 0; j < (int)(trows / b2); ++j) {
            hipLaunchKernelGGL(flow_fill_time, dim3((b2 + 63) / 64), dim3(64), 0, st, B.tv + (size_t)j * b2, t_host[s0 + j], b2);
            ASTTS_CHECK_LAUNCH();
        }
        RUN(astts_op_time_embedding(B.tv, B.temb0, (int)trows, c.time_in, 1000.0f, st));
        RUN(k.linear(B.temb0, 0, c.t1, nullptr, B.temb1, 0, trows, ASTTS_ACT_SILU));
        RUN(k.linear(B.temb1, 0, c.t2, nullptr, B.temb2, 0, trows, ASTTS_ACT_NONE));
        RUN(astts_op_elementwise(ASTTS_EL_MISH, B.temb2, nullptr, nullptr, nullptr, B.temb2, trows * c.time_dim, 1, c.time_dim, 0.0f, 0.0f, st));
        size_t rj = 0;
        for (const std::vector<astts_flow::Block>* grp : {&h->down, &h->mid, &h->up})
            for (const astts_flow::Block& blk : *grp)
                RUN(k.linear(B.temb2, 0, blk.res.mlp, nullptr, B.tproj + (rj++) * (size_t)trows * C, 0, trows, ASTTS_ACT_NONE));
      }
      {
        size_t ri = 0;      // ResNet block counter of this estimator pass (down, mid, up order: the order of the table above)
        auto tproj_of = [&]() { return B.tproj + ((ri++) * (size_t)trows + (size_t)sl * b2) * C; };
        hipLaunchKernelGGL(flow_pack_input, dim3(grid_for((int64_t)b2 * t * 4 * mel)), dim3(256), 0, st, x, mu, spk, cond,
                           B.lens_full, B.xin, b, t, mel);
        ASTTS_CHECK_LAUNCH();

        // buffer pool: cur / other ping-pong for the residual stream, the rest become skips
        float* pool[6];
        for (int i = 0; i < 6; ++i) pool[i] = B.pool[i];
        int n_free = 6;
        auto grab = [&]() { return pool[--n_free]; };
        float* cur = grab();
        float* other = grab();
        float* skips[4];
        int skip_t[4];
        const int* skip_lens[4];
        int n_skips = 0;
        const float* block_in = B.xin;
        int tt = t;
        const int* L = B.lens_full;

        const uint32_t qkv_bytes = 3u * (uint32_t)c.heads * 64u * (uint32_t)C * 2u;     // one block's q|k|v image
        for (size_t i = 0; i < h->down.size(); ++i) {
            const astts_flow::Block& blk = h->down[i];
            RUN(k.resnet(blk.res, tproj_of(), block_in, L, tt, other, blk.tfm.empty() ? nullptr : blk.tfm[0].qkv_frag, qkv_bytes));     // never in place: every GEMM block reads whole input rows
            { float* sw = cur; cur = other; other = sw; }
            for (size_t ti = 0; ti < blk.tfm.size(); ++ti) {
                const bool more = ti + 1 < blk.tfm.size();
                RUN(k.tfm(blk.tfm[ti], cur, other, L, tt, more ? blk.tfm[ti + 1].qkv_frag : nullptr,
                          more ? (uint32_t)(3u * (uint32_t)c.heads * 64u * (uint32_t)C * 2u) : 0u));
            }
            RUN(k.mask(cur, L, tt, C));
            skips[n_skips] = cur;
            skip_t[n_skips] = tt;
            skip_lens[n_skips] = L;
            ++n_skips;
            float* nxt = other;
            other = grab();
            if (blk.kind == ASTTS_FLOW_RESAMPLE_DOWN) {
                RUN(k.gemm(cur, 0, blk.resample, nullptr, nxt, 0, (int64_t)b2 * t_half, tt, t_half, 2, 1, ASTTS_ACT_NONE));
                tt = t_half;
                L = B.lens_half;
            } else {
                RUN(k.gemm(cur, 0, blk.resample, nullptr, nxt, 0, (int64_t)b2 * tt, tt, tt, 1, 1, ASTTS_ACT_NONE));
            }
            cur = nxt;
            RUN(k.mask(cur, L, tt, C));
            block_in = cur;
        }
        for (size_t mi = 0; mi < h->mid.size(); ++mi) {
            const astts_flow::Block& blk = h->mid[mi];
            RUN(k.resnet(blk.res, tproj_of(), cur, L, tt, other, blk.tfm.empty() ? nullptr : blk.tfm[0].qkv_frag, qkv_bytes));
            float* sw = cur; cur = other; other = sw;
            for (size_t ti = 0; ti < blk.tfm.size(); ++ti) {
                const bool more = ti + 1 < blk.tfm.size();
                // after the last transformer block of a mid block comes the next mid block's first convolution
                // (its fragment-order image when that ResNet block takes the three-launch path, else the row-major one)
                const astts_flow_resnet_t* nr = (!more && mi + 1 < h->mid.size()) ? &h->mid[mi + 1].res : nullptr;
                const astts_weight_t* nc = nr ? &nr->c1 : nullptr;
                const void* nw = !nr ? nullptr
                                 : (nr->c1_frag && nr->c2_frag && nr->res_frag && astts_op_resnet_conv_supported(nc->cin, nc->n, c.groups, nc->taps) &&
                                    astts_op_resnet_conv_supported(nr->c2.cin, nr->c2.n, c.groups, nr->c2.taps) &&
                                    astts_op_resnet_conv_supported(nr->res.cin, nr->res.n, c.groups, nr->res.taps)) ? nr->c1_frag : nc->w;
                RUN(k.tfm(blk.tfm[ti], cur, other, L, tt, more ? blk.tfm[ti + 1].qkv_frag : nw,
                          more ? (uint32_t)(3u * (uint32_t)c.heads * 64u * (uint32_t)C * 2u)
                               : (nc ? (uint32_t)((size_t)nc->n * nc->taps * nc->cin_pad * 2) : 0u)));
            }
            RUN(k.mask(cur, L, tt, C));
        }
        const float* up_src = cur;          // [b2, up_t >= skip t, C] with batch stride up_bs
        int64_t up_bs = (int64_t)tt * C;
        for (size_t i = 0; i < h->up.size(); ++i) {
            const astts_flow::Block& blk = h->up[i];
            --n_skips;
            tt = skip_t[n_skips];
            L = skip_lens[n_skips];
            hipLaunchKernelGGL(flow_concat_skip, dim3(grid_for((int64_t)b2 * tt * 2 * (C / 4))), dim3(256), 0, st, up_src, up_bs,
                               skips[n_skips], L, B.xin, b2, tt, C);
            ASTTS_CHECK_LAUNCH();
            RUN(k.resnet(blk.res, tproj_of(), B.xin, L, tt, cur, blk.tfm.empty() ? nullptr : blk.tfm[0].qkv_frag, qkv_bytes));
            for (size_t ti = 0; ti < blk.tfm.size(); ++ti) {
                const bool more = ti + 1 < blk.tfm.size();
                RUN(k.tfm(blk.tfm[ti], cur, other, L, tt, more ? blk.tfm[ti + 1].qkv_frag : nullptr,
                          more ? (uint32_t)(3u * (uint32_t)c.heads * 64u * (uint32_t)C * 2u) : 0u));
            }
            RUN(k.mask(cur, L, tt, C));
            if (blk.kind == ASTTS_FLOW_RESAMPLE_UP) {
                // phase-decomposed ConvTranspose1d: [b2 * (tt + 1), 2C] == [b2, 2 (tt + 1), C]; output step j is row 1 + j
                RUN(k.gemm(cur, 0, blk.resample, nullptr, B.up, 0, (int64_t)b2 * (tt + 1), tt, tt + 1, 1, 1, ASTTS_ACT_NONE));
                up_src = B.up + C;
                up_bs = (int64_t)(tt + 1) * 2 * C;
            } else {
                RUN(k.gemm(cur, 0, blk.resample, nullptr, other, 0, (int64_t)b2 * tt, tt, tt, 1, 1, ASTTS_ACT_NONE));
                float* sw = cur; cur = other; other = sw;
                up_src = cur;
                up_bs = (int64_t)tt * C;
            }
        }
        // final block at the full resolution (tt == t here: the last up block restores it)
        float* fin = const_cast<float*>(up_src);
        RUN(k.mask(fin, B.lens_full, t, C));
        RUN(k.gemm(fin, 0, c.fin_c, nullptr, B.r1, 0, (int64_t)b2 * t, t, t, 1, 1, ASTTS_ACT_NONE));
        RUN(astts_op_groupnorm_ex(B.r1, B.lens_full, c.fin_g_w, c.fin_g_b, nullptr, B.r3, 0, b2, t, C, c.groups, 1e-5f, 1, B.gn_ws,
                                  B.gn_ws_bytes, st));
        RUN(k.gemm(B.r3, 0, c.fin_p, nullptr, B.d, 0, (int64_t)b2 * t, t, t, 1, 0, ASTTS_ACT_NONE));
        RUN(k.mask(B.d, B.lens_full, t, mel));
        // x += dt * ((1 + r) d_cond - r d_uncond)
        RUN(astts_op_elementwise(ASTTS_EL_CFG_EULER, x, B.d, nullptr, nullptr, x, (int64_t)b * t * mel, t, mel, dt_host[s], cfg_rate, st));
      }
    }
    return ASTTS_OK;
}

}  // extern "C"
