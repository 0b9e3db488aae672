// ops_resnet_conv.hip -- the three convolutions of a ResnetBlock1D of the flow-matching estimator (the reference's hot loop #3,
// ConditionalDecoder's ResnetBlock1D = Block1D(conv3 -> GroupNorm -> Mish) x 2 + 1x1 residual conv, behind
// cosyvoice.inference_tts_with_st, tts_with_rag.py:195) with the two GroupNorm + Mish passes folded INTO them.
//
// Why: a ResNet block was five dependent launches -- conv (12.6 us), GroupNorm + Mish (8.7), conv (13.2), GroupNorm + Mish (8.7),
// 1x1 conv + residual (12.6) -- and a GroupNorm launch is nothing but launch floor (4 us) + one pass over 5.6 MB.  GroupNorm needs
// whole-sequence statistics, so it cannot simply ride on a GEMM's operand load; split in two it can:
//   * statistics: the convolution that PRODUCES the tensor reduces its own output tile (exact two-pass mean / M2 in registers:
//     a wave's 32 x 32 tile is exactly one group's channels) and leaves one (count, mean, M2) triple per (sequence, tile, group);
//   * normalise + Mish (+ time-embedding add, + length mask): done by the CONSUMER -- while it stages its input tile (conv 2: once
//     per element, the staged tile serves all output channels) or in its epilogue on the residual operand (1x1 conv) -- after
//     merging the <= 22 triples of its sequence in a fixed order (Chan's update: deterministic, no atomics).
// Three launches per block instead of five; nothing else changes (same masks, same statistics over the valid frames only).
//
// Kernel: C = 256 -> 256 channels, 1 or 3 taps.  Workgroup = 32 frames of one sequence x all 256 output channels, 8 waves, wave w =
// output channels 32 w .. 32 w + 31 (= group w).  The (32 + halo) x 256 input tile is staged once in LDS as fp16; every wave
// streams its own 32-column weight slice (fragment order, astts_op_conv_pack_frag) through registers one tap ahead: no weight
// byte is fetched twice by a workgroup.  Grid = ceil(T / 32) x sequences (176 workgroups at 16 x 344).
// Tried and dropped: 64 frames x 128 channels per workgroup of four waves (192 workgroups, half the weight stream each): the flow
// solve got 1.2 ms SLOWER -- as with the split feed-forward form, the weight stream is not what a launch waits for.
#include "common.h"
#include "xlane.h"

namespace astts {

static constexpr int RC_C = 256;           // output channels (8 groups of 32); input channels CIN = 256 or 512 (the up blocks' concat)

__device__ __forceinline__ float rc_mish(float x) {      // as ops_norm_elem.hip: x n / (n + 2), n = e^x (e^x + 2)
    const float e = __expf(fminf(x, 20.0f));
    const float n = e * (e + 2.0f);
    return x > 20.0f ? x : x * n * __frcp_rn(n + 2.0f);
}

__device__ __forceinline__ float rc_wsum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct RconvArgs {
    const float* x;           // [b][t][cin] fp32 input (conv 1 / 1x1: the block input, already masked; conv 2: conv 1's raw output)
    const _Float16* w;        // [taps][8][cin / 16][64][8] fp16 fragment order
    const float* bias;        // [256]
    float* out;               // [b][t][256] fp32
    // staging transform (conv 2): x <- mask * (mish(GroupNorm(x; in_stats, in_gamma, in_beta)) + in_add[b])
    const float* in_stats;    // [b][ntile][8][3] (count, mean, M2) triples of x, or null: x is used as it is
    const float* in_gamma;
    const float* in_beta;
    const float* in_add;      // [b][256] or null
    // epilogue residual (1x1 conv): out += mask * mish(GroupNorm(res; res_stats, res_gamma, res_beta))
    const float* res;         // [b][t][256] fp32 or null
    const float* res_stats;
    const float* res_gamma;
    const float* res_beta;
    float* out_stats;         // [b][ntile][8][3]: triples of THIS convolution's output over the valid frames, or null
    const int* lens;          // [b] valid frames or null (all)
    int t, taps;
    float eps;
    const char* pf;           // L2 prefetch of the next launch's weights (one range; see tfm_attn_fused), or null
    unsigned pf_bytes;
};

// merge the per-tile triples of sequence bb, group g, in tile order -> (mean, rstd)
__device__ __forceinline__ void rc_merge(const float* stats, int bb, int ntile, int g, float eps, float* mean_out, float* rstd_out) {
    float n = 0.0f, mean = 0.0f, m2 = 0.0f;
    const float* p = stats + ((int64_t)bb * ntile * 8 + g) * 3;
    for (int i = 0; i < ntile; ++i) {
        const float nb = p[(int64_t)i * 24], mb = p[(int64_t)i * 24 + 1], qb = p[(int64_t)i * 24 + 2];
        if (nb > 0.0f) {
            const float nn = n + nb, d = mb - mean;
            mean += d * (nb / nn);
            m2 += qb + d * d * (n * nb / nn);
            n = nn;
        }
    }
    *mean_out = mean;
    *rstd_out = rsqrtf((n > 0.0f ? m2 / n : 0.0f) + eps);
}

// The same merge by one WAVE: lane i holds tile i's triple (loaded by the caller: ONE round trip instead of one per tile), the tiles are
// folded in tile order through v_readlane -- the same operations in the same order as rc_merge, hence the same bits.  <= 64 tiles.
__device__ __forceinline__ void rc_merge_lanes(float nb_l, float mb_l, float qb_l, int ntile, float eps, float* mean_out, float* rstd_out) {
    float n = 0.0f, mean = 0.0f, m2 = 0.0f;
    for (int i = 0; i < ntile; ++i) {
        const float nb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nb_l), i));
        const float mb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mb_l), i));
        const float qb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qb_l), i));
        if (nb > 0.0f) {
            const float nn = n + nb, d = mb - mean;
            mean += d * (nb / nn);
            m2 += qb + d * d * (n * nb / nn);
            n = nn;
        }
    }
    *mean_out = mean;
    *rstd_out = rsqrtf((n > 0.0f ? m2 / n : 0.0f) + eps);
}

// SHARE: the form for grids of two rounds of workgroups and more (long-form batches).  A workgroup is one latency chain -- rows in,
// statistics merge, staging transform, three taps of 16 MFMAs per wave, epilogue: 13 % MFMA-busy -- so two of them per CU cover each
// other's prologue and epilogue: the weights come in HALF units (8 fragments = 32 VGPRs, the next half requested under this one's
// MFMAs) instead of whole units two deep (128 VGPRs), which fits the wave into 128 registers.  Same operations in the same order:
// bit-identical outputs.
template <int CIN, bool SHARE = false>
// Leading parameters = what the first loads need: preloaded into SGPRs by the command processor (-amdgpu-kernarg-preload-count,
// csrc/Makefile; a by-value struct is not), the struct carries the rest.
__global__ __launch_bounds__(512, SHARE ? 2 : 1) void rconv_lds(const float* p_x, const _Float16* p_w, const float* p_in_stats, const int* p_lens, int p_t, int p_taps,
                                                    float p_eps, RconvArgs a_in) {
    RconvArgs a = a_in;
    a.x = p_x; a.w = p_w; a.in_stats = p_in_stats; a.lens = p_lens; a.t = p_t; a.taps = p_taps; a.eps = p_eps;
    extern __shared__ __attribute__((aligned(16))) _Float16 rc_smem[];
    __shared__ float s_in[8][2], s_res[8][2];
    constexpr int RS = CIN + 8;                       // halfs per staged row
    constexpr int KC = CIN / 256;                     // 256-channel slices per tap
    constexpr int TPR = CIN / 4;                      // threads per staged row (one float4 each)
    constexpr int RPP = 512 / TPR;                    // rows per pass
    constexpr int NPASS = (34 + RPP - 1) / RPP;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int bb = blockIdx.y, t0 = blockIdx.x * 32;
    const int ntile = (a.t + 31) >> 5;               // = gridDim.x, from a preloaded argument (an implicit argument is a scalar load + wait)
    const int halo = (a.taps - 1) / 2;
    const int sr = 32 + 2 * halo;
    const int64_t seq = (int64_t)bb * a.t;

    // ---- Prologue loads, in the order their consumers run and with nothing conditional between them (a load inside an `if` block is
    // waited for at the end of the block: the bounds-checked rows were NPASS dependent round trips, the tile-by-tile statistics merge
    // one per tile, and the L2 prefetch loop held everything behind an HBM miss).  (1) the rows, frame index clamped into the sequence and zeroed
    // afterwards; (2) two units of weights; (3) GroupNorm partials: wave w owns group w, lane i tile i (through a selected pointer when
    // absent); (4) the per-channel parameters of the staging transform; (5) the prefetch lines, nobody waits for.
    // TPR threads per row (one float4 each), RPP rows per pass; a thread keeps one column group (4 channels)
    const int col = (tid % TPR) * 4, r0 = tid / TPR;
    float4 v[NPASS];
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
        const int t = min(max(t0 - halo + r0 + RPP * u, 0), a.t - 1);
        v[u] = *reinterpret_cast<const float4*>(a.x + (seq + t) * CIN + col);
    }
    constexpr int WFN = SHARE ? 8 : 16;               // fragments per weight buffer: half a unit / a unit
    half8 wf[2][WFN];
    const _Float16* wbase = a.w + ((int64_t)wid * (CIN / 16) * 64 + lane) * 8;
    auto load_unit = [&](int u, half8 (&dst)[WFN]) {  // unit = tap * KC + slice; SHARE: u counts half units
        const int uu = SHARE ? u >> 1 : u;
        const int tap = uu / KC, kc = uu - tap * KC;
        const _Float16* p = wbase + ((int64_t)tap * 8 * (CIN / 16) + kc * 16 + (SHARE ? (u & 1) * 8 : 0)) * 512;
#pragma unroll
        for (int ks = 0; ks < WFN; ++ks) dst[ks] = *reinterpret_cast<const half8*>(p + (int64_t)ks * 512);
    };
    const int nunits = a.taps * KC * (SHARE ? 2 : 1);
    load_unit(0, wf[0]);
    load_unit(nunits > 1 ? 1 : 0, wf[1]);
    // (everything above needs preloaded arguments only; what follows needs the struct -- res_stats, in_gamma ... -- whose scalar load has
    // had the time of ~50 load issues to arrive)
    const bool in_gn = CIN == RC_C && a.in_stats != nullptr;      // the staging transform exists for 256-channel inputs only
    const bool lanes_ok = ntile <= 64;
    float st_in[3] = {0.f, 0.f, 0.f}, st_res[3] = {0.f, 0.f, 0.f};
    {
        const int ti = min(lane, ntile - 1);
        const float* pi = a.in_stats ? a.in_stats + (((int64_t)bb * ntile + ti) * 8 + wid) * 3 : a.x;
        const float* pr = a.res_stats ? a.res_stats + (((int64_t)bb * ntile + ti) * 8 + wid) * 3 : a.x;
#pragma unroll
        for (int j = 0; j < 3; ++j) {     // (the compiler sinks these into the blocks that use them: a second, short round trip)
            st_in[j] = pi[j];
            st_res[j] = pr[j];
        }
    }
    float4 ga, be, ad;
    {
        const float* pg = in_gn ? a.in_gamma + col : a.x;
        const float* pb = in_gn ? a.in_beta + col : a.x;
        const float* pa = in_gn && a.in_add ? a.in_add + (int64_t)bb * RC_C + col : a.x;
        ga = *reinterpret_cast<const float4*>(pg);
        be = *reinterpret_cast<const float4*>(pb);
        ad = *reinterpret_cast<const float4*>(pa);
        if (!(in_gn && a.in_add)) ad = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    unsigned pf_keep[2];
    {   // workgroups of one XCD (linear ids congruent mod 8) split the range, one 128-byte line per thread; the first two lines of a
        // thread, as straight-line loads behind everything above (a counted wait never reaches them; a range beyond 1 024 lines per slot --
        // none today -- is left to the next launch itself: the prefetch is a hint)
        const unsigned lin = blockIdx.y * ntile + blockIdx.x, nwg = ntile * gridDim.y;
        const unsigned slot = lin >> 3, nslots = max((nwg + 7) >> 3, 1u);
        const unsigned lines = a.pf ? (a.pf_bytes + 127) >> 7 : 0u;
        const unsigned per = (lines + nslots - 1) / nslots;
        const char* pfb = a.pf ? a.pf : reinterpret_cast<const char*>(a.x);
#pragma unroll
        for (unsigned k = 0; k < 2; ++k) {
            const unsigned i = tid + k * 512, ln = slot * per + i;
            prefetch_line(pfb + (i < per && ln < lines ? (size_t)ln << 7 : (size_t)0), pf_keep[k]);      // (xlane.h)
        }
    }
    const int len = a.lens ? min(a.lens[bb], a.t) : a.t;     // (here, not at the top: its scalar load's wait covers every scalar load in flight)
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
        const int r = r0 + RPP * u;
        const int t = t0 - halo + r;
        if (!(r < sr && t >= 0 && t < a.t)) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (lanes_ok) {
        float mo, ro;
        if (a.in_stats) {
            rc_merge_lanes(st_in[0], st_in[1], st_in[2], ntile, a.eps, &mo, &ro);
            if (lane == 0) { s_in[wid][0] = mo; s_in[wid][1] = ro; }
        }
        if (a.res_stats) {
            rc_merge_lanes(st_res[0], st_res[1], st_res[2], ntile, a.eps, &mo, &ro);
            if (lane == 0) { s_res[wid][0] = mo; s_res[wid][1] = ro; }
        }
    } else {
        if (a.in_stats && tid < 8) rc_merge(a.in_stats, bb, ntile, tid, a.eps, &s_in[tid][0], &s_in[tid][1]);
        if (a.res_stats && tid >= 64 && tid < 72) rc_merge(a.res_stats, bb, ntile, tid - 64, a.eps, &s_res[tid - 64][0], &s_res[tid - 64][1]);
    }
    if (in_gn) __syncthreads();                       // the merged statistics are in LDS
    {
        const float mean = in_gn ? s_in[(col >> 5) & 7][0] : 0.0f, rstd = in_gn ? s_in[(col >> 5) & 7][1] : 1.0f;
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            const int r = r0 + RPP * u;
            const int t = t0 - halo + r;
            if (r < sr) {
                float4 o = v[u];
                if (in_gn) {
                    if (t >= 0 && t < len)
                        o = make_float4(rc_mish((o.x - mean) * rstd * ga.x + be.x) + ad.x, rc_mish((o.y - mean) * rstd * ga.y + be.y) + ad.y,
                                        rc_mish((o.z - mean) * rstd * ga.z + be.z) + ad.z, rc_mish((o.w - mean) * rstd * ga.w + be.w) + ad.w);
                    else
                        o = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                half4 h4;
                h4[0] = (_Float16)o.x; h4[1] = (_Float16)o.y; h4[2] = (_Float16)o.z; h4[3] = (_Float16)o.w;
                *reinterpret_cast<half4*>(rc_smem + (size_t)r * RS + col) = h4;
            }
        }
    }
    __syncthreads();

    // ---- (tap, slice) units: A fragments from LDS (row-shifted), B fragments from registers, next unit's weights on their way
    float16v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const _Float16* arow = rc_smem + (size_t)c * RS + 8 * hh;
    auto compute_unit = [&](int u, const half8 (&w)[WFN]) {
        const int uu = SHARE ? u >> 1 : u;
        const int tap = uu / KC, kc = uu - tap * KC;
        const _Float16* ap = arow + (size_t)tap * RS + kc * 256 + (SHARE ? (u & 1) * 128 : 0);
        constexpr int AB = SHARE ? 4 : 8;             // A fragments read ahead of their MFMAs
#pragma unroll
        for (int ks0 = 0; ks0 < WFN; ks0 += AB) {
            half8 af[AB];
#pragma unroll
            for (int ks = 0; ks < AB; ++ks) af[ks] = *reinterpret_cast<const half8*>(ap + 16 * (ks0 + ks));
#pragma unroll
            for (int ks = 0; ks < AB; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], w[ks0 + ks], acc, 0, 0, 0);
        }
    };
    for (int u = 0; u < nunits; u += 2) {
        compute_unit(u, wf[0]);
        if (u + 2 < nunits) load_unit(u + 2, wf[0]);
        if (u + 1 < nunits) {
            compute_unit(u + 1, wf[1]);
            if (u + 3 < nunits) load_unit(u + 3, wf[1]);
        }
    }

    // ---- epilogue: element e = frame t0 + (e & 3) + 8 (e >> 2) + 4 hh, the lane's channel is 32 wid + c (group wid)
    const int f = wid * 32 + c;
    const float bias = a.bias ? a.bias[f] : 0.0f;
    float rmean = 0.0f, rrstd = 1.0f, rg = 1.0f, rb = 0.0f;
    if (a.res) {
        rmean = s_res[wid][0]; rrstd = s_res[wid][1];
        rg = a.res_gamma[f]; rb = a.res_beta[f];
    }
    float rv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int t = t0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        rv[e] = (a.res && t < len) ? a.res[(seq + t) * RC_C + f] : 0.0f;
    }
    float val[16];
    float s = 0.0f;
    int nval = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int t = t0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        val[e] = acc[e] + bias;
        if (t < len) {
            s += val[e];
            ++nval;
        }
    }
    if (a.out_stats) {                                // this tile's (count, mean, M2) over the valid frames: exact two-pass
        const float cnt = rc_wsum((float)nval);
        s = rc_wsum(s);
        const float mean = cnt > 0.0f ? s / cnt : 0.0f;
        float q = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int t = t0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (t < len) {
                const float d = val[e] - mean;
                q += d * d;
            }
        }
        q = rc_wsum(q);
        if (lane == 0) {
            float* p = a.out_stats + (((int64_t)bb * ntile + blockIdx.x) * 8 + wid) * 3;
            p[0] = cnt; p[1] = mean; p[2] = q;
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int t = t0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (t < a.t) {
            float o = val[e];
            if (a.res && t < len) o += rc_mish((rv[e] - rmean) * rrstd * rg + rb);
            a.out[(seq + t) * RC_C + f] = o;
        }
    }
    prefetch_keep(pf_keep[0]);
    prefetch_keep(pf_keep[1]);
}

}  // namespace astts

using namespace astts;

extern "C" {

/* floats of one statistics buffer of astts_op_resnet_conv for b sequences of t frames */
size_t astts_op_resnet_conv_stats_floats(int32_t b, int32_t t) { return (size_t)b * ((t + 31) / 32) * 8 * 3; }

/* 1 when astts_op_resnet_conv serves this shape: 256 or 512 -> 256 channels in 8 groups of 32, 1 or 3 taps */
int astts_op_resnet_conv_supported(int32_t cin, int32_t cout, int32_t groups, int32_t taps) {
    return (cin == RC_C || cin == 2 * RC_C) && cout == RC_C && groups == 8 && (taps == 1 || taps == 3) ? 1 : 0;
}

int astts_op_resnet_conv(const float* x, const void* w_frag_f16, const float* bias, float* out, const float* in_stats, const float* in_gamma,
                         const float* in_beta, const float* in_add, const float* res, const float* res_stats, const float* res_gamma,
                         const float* res_beta, float* out_stats, const int32_t* lens, int32_t b, int32_t t, int32_t cin, int32_t taps,
                         float eps, astts_stream_t stream) {
    return astts_op_resnet_conv_pf(x, w_frag_f16, bias, out, in_stats, in_gamma, in_beta, in_add, res, res_stats, res_gamma, res_beta, out_stats,
                                   lens, b, t, cin, taps, eps, nullptr, 0, stream);
}

int astts_op_resnet_conv_pf(const float* x, const void* w_frag_f16, const float* bias, float* out, const float* in_stats, const float* in_gamma,
                            const float* in_beta, const float* in_add, const float* res, const float* res_stats, const float* res_gamma,
                            const float* res_beta, float* out_stats, const int32_t* lens, int32_t b, int32_t t, int32_t cin, int32_t taps,
                            float eps, const void* pf_ptr, uint32_t pf_bytes, astts_stream_t stream) {
    ASTTS_REQUIRE(x && w_frag_f16 && out && x != out, ASTTS_ERR_INVALID, "astts_op_resnet_conv: null / aliased pointer");
    ASTTS_REQUIRE(astts_op_resnet_conv_supported(cin, RC_C, 8, taps), ASTTS_ERR_UNSUPPORTED,
                  "astts_op_resnet_conv: cin=%d taps=%d (256 or 512 input channels, 1 or 3 taps)", cin, taps);
    ASTTS_REQUIRE(!(in_stats && cin != RC_C), ASTTS_ERR_UNSUPPORTED, "astts_op_resnet_conv: the GroupNorm staging transform needs 256 input channels");
    ASTTS_REQUIRE(b >= 1 && t >= 1 && (!in_stats || (in_gamma && in_beta)) && (!res || (res_stats && res_gamma && res_beta)),
                  ASTTS_ERR_INVALID, "astts_op_resnet_conv: b=%d t=%d or a GroupNorm operand without its statistics / scale / shift", b, t);
    ASTTS_REQUIRE((((uintptr_t)x | (uintptr_t)w_frag_f16 | (uintptr_t)out | (uintptr_t)in_gamma | (uintptr_t)in_beta | (uintptr_t)in_add) & 15) == 0,
                  ASTTS_ERR_INVALID, "astts_op_resnet_conv: operands must be 16-byte aligned");
    RconvArgs a{x, (const _Float16*)w_frag_f16, bias, out, in_stats, in_gamma, in_beta, in_add, res, res_stats, res_gamma, res_beta, out_stats,
                lens, t, taps, eps, nullptr, 0u};
    static const bool pf_on = !(getenv("ASTTS_TFM_PREFETCH") && atoi(getenv("ASTTS_TFM_PREFETCH")) == 0);
    if (pf_ptr && pf_on) {
        a.pf = (const char*)pf_ptr;
        a.pf_bytes = pf_bytes;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(32 + 2) * (cin + 8) * sizeof(_Float16);
    const bool prof = prof_begin(ASTTS_PROF_GEMM_TILE, st, 2.0 * (double)b * t * cin * RC_C * taps);
    // two workgroups per CU once the grid is two rounds of 256 and more (ASTTS_RCONV_SHARE=0 / 1 forces one form: A/B)
    static const int share_env = [] { const char* e = getenv("ASTTS_RCONV_SHARE"); return e ? atoi(e) : -1; }();
    const bool share = cin == RC_C && (share_env >= 0 ? share_env != 0 : (int64_t)((t + 31) / 32) * b >= 512);   // (512 input channels: 136 registers, one workgroup per CU either way)
    const dim3 grid((unsigned)((t + 31) / 32), b);
    if (cin == RC_C && share) hipLaunchKernelGGL((rconv_lds<256, true>), grid, dim3(512), lds, st, a.x, a.w, a.in_stats, a.lens, a.t, a.taps, a.eps, a);
    else if (cin == RC_C) hipLaunchKernelGGL((rconv_lds<256, false>), grid, dim3(512), lds, st, a.x, a.w, a.in_stats, a.lens, a.t, a.taps, a.eps, a);
    else hipLaunchKernelGGL((rconv_lds<512, false>), grid, dim3(512), lds, st, a.x, a.w, a.in_stats, a.lens, a.t, a.taps, a.eps, a);
    if (prof) prof_end(ASTTS_PROF_GEMM_TILE, st);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
