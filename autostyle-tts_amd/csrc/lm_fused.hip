// lm_fused.hip -- decode engine v3: TWO launches per transformer layer instead of five.
//
// Why ("What bounds the pipelined step", item 8): inside the stream pipeline the step time follows the NUMBER of
// decode launches, not their duration (10 % fewer decode steps: step -6.8 %; decode kernels 16 % slower at the same count: nothing).
// A layer of engine v2 is five launches because each operator needs every column of the previous one (QKV -> attention -> out-proj ->
// FFN-in -> FFN-out).  Two of those all-to-all edges disappear when the sums that cross workgroups are taken the tensor-parallel way:
//
//   lm_attn_block   workgroup = (head, 2 rows): LayerNorm -> q | k | v of ITS head -> K / V into the cache -> attention over the cache
//                   -> the out-projection's partial sum over ITS 64 input features, for all d outputs
//   lm_ffn_block    workgroup = 64 hidden features: LayerNorm -> FFN-in (its 64 features) -> ReLU -> the FFN-out projection's partial sum
//                   over ITS 64 hidden features, for all d outputs
//
// The partial sums of the 16 heads / 64 hidden slices meet in the residual stream itself: it is kept as 64-bit FIXED-POINT accumulators
// (2^-32 resolution) and every workgroup adds its partial with device-scope integer atomics.  Integer addition is associative, so the result
// does not depend on the order the workgroups arrive in: bit-reproducible, and a row's result does not depend on the batch width.  (fp32
// atomics would make every run different; a counter + last-arriver reduction costs a launch's worth of latency.)  Price: 64 adds per address
// and FFN launch = ~2 us at 8 rows (scripts/micro/atomic_probe.hip).  Three accumulator buffers rotate: a kernel reads the previous one's,
// adds into its own and clears the next one's.
//
// Arithmetic otherwise as engine v2 (lm_step.hip): fp16 MFMA operands, fp32 accumulation, fp32 LayerNorm / softmax, relative-position
// scores ((q + u) . k + (q + v) . p(i - j)) / sqrt(64), keys [key_start[row], pos].  What differs is only where fp32 sums are rounded.
#include "lm_step.h"
#include "xlane.h"

#include <hip/hip_ext.h>

namespace astts {

static constexpr float FX_SCALE = 4294967296.0f;          // 2^32
static constexpr float FX_INV = 1.0f / 4294967296.0f;
__device__ __forceinline__ long long to_fx(float v) { return __float2ll_rn(v * FX_SCALE); }
__device__ __forceinline__ float from_fx(long long a) { return __ll2float_rn(a) * FX_INV; }
__device__ __forceinline__ void fx_add(long long* p, float v) { atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)to_fx(v)); }

// One wave stages one row: fixed-point or fp32 input -> (layer 0: LayerNorm -> ReLU -> * pre_scale) -> optional fp32 copy (the residual
// operand) -> LayerNorm -> fp16 row in LDS.  d <= 1024, d % 64 == 0.  Lane l holds elements 4 l + 256 i .. + 3.
__device__ __forceinline__ void fused_stage_row(const long long* xin_row, const float* tab_row, const float* pre_g, const float* pre_b,
                                                float pre_scale, const float* ln_g, const float* ln_b, int ln_plain, float eps, int d, int lane,
                                                float* res_row, _Float16* out_row) {
    float4 v[4];
    const int nv = (d + 255) >> 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + i * 256;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nv && k < d) {
            if (xin_row) {
                const ulonglong2 a0 = *reinterpret_cast<const ulonglong2*>(xin_row + k), a1 = *reinterpret_cast<const ulonglong2*>(xin_row + k + 2);
                v[i] = make_float4(from_fx((long long)a0.x), from_fx((long long)a0.y), from_fx((long long)a1.x), from_fx((long long)a1.y));
            } else {
                v[i] = *reinterpret_cast<const float4*>(tab_row + k);
            }
        }
    }
    const float inv_d = 1.0f / (float)d;
    if (!xin_row && pre_g) {      // the LM's input embedding: two-pass LayerNorm -> ReLU -> * sqrt(d) (as lm_gemv's pre-transform)
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = wave_sum_desc(s) * inv_d;
        float qq = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane * 4 + i * 256;
            if (i < nv && k < d) {
                const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float rstd = rsqrtf(wave_sum_desc(qq) * inv_d + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane * 4 + i * 256;
            if (i < nv && k < d) {
                const float4 pg = *reinterpret_cast<const float4*>(pre_g + k), pb = *reinterpret_cast<const float4*>(pre_b + k);
                v[i].x = pre_scale * fmaxf((v[i].x - mean) * rstd * pg.x + pb.x, 0.0f);
                v[i].y = pre_scale * fmaxf((v[i].y - mean) * rstd * pg.y + pb.y, 0.0f);
                v[i].z = pre_scale * fmaxf((v[i].z - mean) * rstd * pg.z + pb.z, 0.0f);
                v[i].w = pre_scale * fmaxf((v[i].w - mean) * rstd * pg.w + pb.w, 0.0f);
            }
        }
    }
    if (res_row) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane * 4 + i * 256;
            if (i < nv && k < d) *reinterpret_cast<float4*>(res_row + k) = v[i];
        }
    }
    float s1 = 0.0f, s2 = 0.0f;                     // single-pass statistics (as lm_gemv)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s1 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        s2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    s1 = wave_sum_desc(s1);
    s2 = wave_sum_desc(s2);
    const float mean = s1 * inv_d;
    const float rstd = rsqrtf(fmaxf(s2 * inv_d - mean * mean, 0.0f) + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + i * 256;
        if (i < nv && k < d) {
            float4 o;
            o.x = (v[i].x - mean) * rstd; o.y = (v[i].y - mean) * rstd; o.z = (v[i].z - mean) * rstd; o.w = (v[i].w - mean) * rstd;
            if (!ln_plain && ln_g) {
                const float4 g = *reinterpret_cast<const float4*>(ln_g + k), bb = *reinterpret_cast<const float4*>(ln_b + k);
                o.x = o.x * g.x + bb.x; o.y = o.y * g.y + bb.y; o.z = o.z * g.z + bb.z; o.w = o.w * g.w + bb.w;
            }
            half4 h4;
            h4[0] = (_Float16)o.x; h4[1] = (_Float16)o.y; h4[2] = (_Float16)o.z; h4[3] = (_Float16)o.w;
            *reinterpret_cast<half4*>(out_row + k) = h4;
        }
    }
}

__device__ __forceinline__ float fdot8(const float (&q)[8], half8 k) {
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += q[e] * (float)k[e];
    return s;
}

static constexpr int FA_ROWS = 2;      // rows per workgroup of lm_attn_block
static constexpr int FA_U = 4;         // key passes per chunk (32 keys per pass and row)

// ------------------------------------------------------------------------------------------------------------------ attention block
__global__ __launch_bounds__(512, 2) void lm_attn_block(FAttnArgs a) {
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int head = blockIdx.x, r0 = blockIdx.y * FA_ROWS;
    const int d = a.d, lines = d >> 6, xs = d + 8;
    _Float16* sX = reinterpret_cast<_Float16*>(fa_smem);                       // [2][d + 8] normalised rows
    float* sRes = reinterpret_cast<float*>(sX + FA_ROWS * xs);                 // [2][d] fp32 residual rows
    float* red = sRes + FA_ROWS * d;                                           // [8 waves][12 tiles][2 rows][16]
    float* sQ = red + 8 * 12 * 32;                                             // [2][64]
    _Float16* sO = reinterpret_cast<_Float16*>(sQ + FA_ROWS * 64);             // [2][64 + 8]
    float* s_m = reinterpret_cast<float*>(sO + FA_ROWS * 72);                  // [8]
    float* s_l = s_m + 8;                                                      // [8]
    float* s_o = s_l + 8;                                                      // [8][64]

    // ---- weights of the first six tiles (q0..q3, k0, k1): issued before anything waits.  Tile t covers rows part * d + head * 64 +
    // (t % 4) * 16 .. + 15 of wqkv (part = t / 4); a wave walks K lines wid, wid + 8 (lane (c, g): bytes [32 g, 32 g + 32) of each line)
    auto wrow = [&](int t) { return a.wqkv + ((int64_t)(t >> 2) * d + head * 64 + (t & 3) * 16 + c) * d + g * 16; };
    half8 fb[6][2][2];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int line = wid + i * 8;
            if (line < lines) {
                fb[t][i][0] = *reinterpret_cast<const half8*>(wrow(t) + line * 64);
                fb[t][i][1] = *reinterpret_cast<const half8*>(wrow(t) + line * 64 + 8);
            }
        }
    // ---- rows: waves 0 and 1 stage one row each
    if (wid < FA_ROWS) {
        const int row = r0 + wid;
        if (row < a.b) {
            const long long* xr = a.xin ? a.xin + (int64_t)row * d : nullptr;
            const float* tr = a.xin ? nullptr : a.table + (int64_t)a.tok[row] * d;
            fused_stage_row(xr, tr, a.pre_g, a.pre_b, a.pre_scale, a.ln_g, a.ln_b, a.ln_plain, a.eps, d, lane, sRes + wid * d, sX + wid * xs);
        } else {
            for (int k = lane; k < d; k += 64) sX[wid * xs + k] = (_Float16)0.0f;
        }
    }
    __syncthreads();
    // ---- q | k | v of this head for the two rows: 12 tiles of 16 columns in two batches of six; every wave takes its K lines of every tile
    const _Float16* arow = sX + (c < FA_ROWS ? c : 0) * xs + g * 16;
    float4v acc[6];
    auto run_batch = [&](int t0) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int line = wid + i * 8;
            if (line < lines) {
                const half8 fa0 = *reinterpret_cast<const half8*>(arow + line * 64);
                const half8 fa1 = *reinterpret_cast<const half8*>(arow + line * 64 + 8);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, fb[t][i][0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa1, fb[t][i][1], acc[t], 0, 0, 0);
                }
            }
        }
        if (g == 0) {          // rows 0 and 1 of the 16 x 16 tile: elements 0 and 1 of the lanes with g == 0
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                red[((wid * 12 + t0 + t) * 2 + 0) * 16 + c] = acc[t][0];
                red[((wid * 12 + t0 + t) * 2 + 1) * 16 + c] = acc[t][1];
            }
        }
    };
    run_batch(0);
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int line = wid + i * 8;
            if (line < lines) {
                fb[t][i][0] = *reinterpret_cast<const half8*>(wrow(6 + t) + line * 64);
                fb[t][i][1] = *reinterpret_cast<const half8*>(wrow(6 + t) + line * 64 + 8);
            }
        }
    run_batch(6);
    // the out-projection's weights of this wave (tiles wid, wid + 8, ...; K = this head's 64 input features): requested now, they land
    // while the attention runs
    const int otiles = d >> 4;
    half8 fo[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = wid + j * 8;
        if (t < otiles) {
            const _Float16* wp = a.wo + ((int64_t)t * 16 + c) * d + head * 64 + g * 16;
            fo[j][0] = *reinterpret_cast<const half8*>(wp);
            fo[j][1] = *reinterpret_cast<const half8*>(wp + 8);
        }
    }
    __syncthreads();
    // sum over the waves (fixed order), bias; q stays in LDS, k and v go to the cache
    if (tid < 12 * 32) {
        const int t = tid >> 5, row = (tid >> 4) & 1, col = tid & 15;
        float v = 0.0f;
        const int lw = lines < 8 ? lines : 8;              // waves that held a K line
        for (int w = 0; w < lw; ++w) v += red[((w * 12 + t) * 2 + row) * 16 + col];
        const int part = t >> 2, hc = (t & 3) * 16 + col;
        if (a.bqkv) v += a.bqkv[part * d + head * 64 + hc];
        if (part == 0) {
            sQ[row * 64 + hc] = v;
        } else if (r0 + row < a.b) {
            a.kv[(int64_t)a.pos * a.kv_t + (int64_t)(r0 + row) * a.kv_b + (int64_t)head * a.kv_h + (part == 2 ? a.kv_v : 0) + hc] = (_Float16)v;
        }
    }
    if (!(a.dbg & 8)) __threadfence();       // this token's K / V rows are in L2 (and this CU's L1 holds no older copy) before any wave of the workgroup reads keys
    __syncthreads();
    // ---- attention: waves 0..3 take row 0, waves 4..7 row 1; 8 lanes per key (8 dims each), 32 keys per pass and row
    {
        const int rl = tid >> 8, tl = tid & 255, sub = tl & 7, kg = tl >> 3;
        const int row = r0 + rl;
        const int qpos = a.pos, len = qpos + 1;
        const int ks0 = row < a.b ? (a.kstart ? min(a.kstart[row], len - 1) : 0) : len;
        const int kend = row < a.b && !(a.dbg & 2) ? len : 0;
        const int64_t trow = a.kv_t;
        const _Float16* kb = a.kv + (int64_t)(row < a.b ? row : 0) * a.kv_b + (int64_t)head * a.kv_h + sub * 8;
        const _Float16* vb = kb + a.kv_v;
        const _Float16* pb = a.postab + (int64_t)a.center * a.ldp + head * 64 + sub * 8;
        float qu[8], qv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = sQ[rl * 64 + sub * 8 + e];
            qu[e] = (x + a.bias_u[head * 64 + sub * 8 + e]) * a.scale;
            qv[e] = (x + a.bias_v[head * 64 + sub * 8 + e]) * a.scale;
        }
        float m_run = -INFINITY, l_run = 0.0f;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.0f;
        for (int j0 = ks0; j0 < kend; j0 += 32 * FA_U) {
            half8 kk[FA_U], pp[FA_U], vv[FA_U];
#pragma unroll
            for (int u = 0; u < FA_U; ++u) {
                const int j = j0 + u * 32 + kg;
                if (j < kend) {
                    kk[u] = *reinterpret_cast<const half8*>(kb + (int64_t)j * trow);
                    pp[u] = *reinterpret_cast<const half8*>(pb + (int64_t)(qpos - j) * a.ldp);
                    vv[u] = *reinterpret_cast<const half8*>(vb + (int64_t)j * trow);
                }
            }
            float s[FA_U];
            float m_new = m_run;
#pragma unroll
            for (int u = 0; u < FA_U; ++u) {
                const int j = j0 + u * 32 + kg;
                float t = j < kend ? fdot8(qu, kk[u]) + fdot8(qv, pp[u]) : 0.0f;
                t = xadd<1>(t);
                t = xadd<2>(t);
                t = xadd<4>(t);
                s[u] = j < kend ? t : -INFINITY;
                m_new = fmaxf(m_new, s[u]);
            }
            m_new = xmax<32>(xmax<16>(xmax<8>(m_new)));
            if (m_new > -INFINITY) {
                const float sc_old = m_run == -INFINITY ? 0.0f : __expf(m_run - m_new);
                l_run *= sc_old;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] *= sc_old;
#pragma unroll
                for (int u = 0; u < FA_U; ++u) {
                    const int j = j0 + u * 32 + kg;
                    if (j < kend) {
                        const float p = __expf(s[u] - m_new);
                        l_run += p;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += p * (float)vv[u][e];
                    }
                }
                m_run = m_new;
            }
        }
        l_run = xadd<32>(xadd<16>(xadd<8>(l_run)));
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = xadd<32>(xadd<16>(xadd<8>(o[e])));
        if (lane < 8) {
            *reinterpret_cast<float4*>(&s_o[wid * 64 + lane * 8]) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(&s_o[wid * 64 + lane * 8 + 4]) = make_float4(o[4], o[5], o[6], o[7]);
            if (lane == 0) {
                s_l[wid] = l_run;
                s_m[wid] = m_run;
            }
        }
    }
    __syncthreads();
    if (tid < FA_ROWS * 64) {            // merge the four waves of a row (fixed order), normalise, fp16 operand of the out-projection
        const int rl = tid >> 6, dd = tid & 63, w0 = rl * 4;
        float mx = s_m[w0];
#pragma unroll
        for (int w = 1; w < 4; ++w) mx = fmaxf(mx, s_m[w0 + w]);
        float tot = 0.0f, l = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = s_m[w0 + w] == -INFINITY ? 0.0f : __expf(s_m[w0 + w] - mx);
            tot += s_o[(w0 + w) * 64 + dd] * sc;
            l += s_l[w0 + w] * sc;
        }
        sO[rl * 72 + dd] = (_Float16)(l > 0.0f ? tot / l : 0.0f);
    }
    __syncthreads();
    // ---- the out-projection's partial sum over this head's 64 features, added into the fixed-point residual stream
    {
        const _Float16* ao = sO + (c < FA_ROWS ? c : 0) * 72 + g * 16;
        const half8 fa0 = *reinterpret_cast<const half8*>(ao), fa1 = *reinterpret_cast<const half8*>(ao + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = wid + j * 8;
            if (t < otiles) {
                float4v y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = 0.0f;
                y = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, fo[j][0], y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa1, fo[j][1], y, 0, 0, 0);
                if (g == 0) {
                    const int n = t * 16 + c;
#pragma unroll
                    for (int e = 0; e < FA_ROWS; ++e) {
                        if (r0 + e < a.b && !(a.dbg & 1)) {
                            float v = y[e];
                            if (head == 0) v += sRes[e * d + n] + (a.bo ? a.bo[n] : 0.0f);     // the residual and the bias enter once
                            fx_add(a.xout + (int64_t)(r0 + e) * d + n, v);
                        }
                    }
                }
            }
        }
    }
    // ---- clear this workgroup's share of the buffer the NEXT kernel accumulates into
    {
        const int nwg = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int total = a.b * d, per = (total + nwg - 1) / nwg;
        for (int i = tid; i < per; i += 512) {
            const int idx = id * per + i;
            if (idx < total && !(a.dbg & 4)) a.xzero[idx] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ feed-forward block
// Workgroup = 64 hidden features (4 column tiles of W1; K line `blockIdx.x` of W2).  MT row tiles of 16 (b <= 32).
template <int MT>
__global__ __launch_bounds__(512, 2) void lm_ffn_block(FFfnArgs a) {
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char ff_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int j = blockIdx.x, d = a.d, lines = d >> 6, xs = d + 8, M = a.b;
    _Float16* sX = reinterpret_cast<_Float16*>(ff_smem);                       // [16 MT][d + 8]
    float* red = reinterpret_cast<float*>(sX + 16 * MT * xs);                  // [8 waves][4 tiles][MT][4][64]
    _Float16* sH = reinterpret_cast<_Float16*>(red + 8 * 4 * MT * 256);        // [16 MT][64 + 8]
    // ---- every weight this workgroup needs, requested up front: W1 rows 64 j .. 64 j + 63 (K lines wid, wid + 8), W2 K line j of the
    // output tiles wid, wid + 8, ...
    half8 f1[4][2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int line = wid + i * 8;
            if (line < lines) {
                const _Float16* wp = a.w1 + ((int64_t)j * 64 + t * 16 + c) * d + g * 16 + line * 64;
                f1[t][i][0] = *reinterpret_cast<const half8*>(wp);
                f1[t][i][1] = *reinterpret_cast<const half8*>(wp + 8);
            }
        }
    const int otiles = d >> 4;
    half8 f2[8][2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int t = wid + q * 8;
        if (t < otiles) {
            const _Float16* wp = a.w2 + ((int64_t)t * 16 + c) * a.ffn + j * 64 + g * 16;
            f2[q][0] = *reinterpret_cast<const half8*>(wp);
            f2[q][1] = *reinterpret_cast<const half8*>(wp + 8);
        }
    }
    // ---- rows: wave w stages rows w, w + 8, ... (LayerNorm 2 -> fp16)
    for (int mr = wid; mr < 16 * MT; mr += 8) {
        if (mr < M) {
            fused_stage_row(a.xin + (int64_t)mr * d, nullptr, nullptr, nullptr, 0.0f, a.ln_g, a.ln_b, a.ln_plain, a.eps, d, lane, nullptr, sX + mr * xs);
        } else {
            for (int k = lane; k < d; k += 64) sX[mr * xs + k] = (_Float16)0.0f;
        }
    }
    __syncthreads();
    // ---- hidden features: 4 tiles x MT row tiles, K split over the waves
    {
        float4v acc[4][MT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][m][e] = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int line = wid + i * 8;
            if (line < lines) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const _Float16* ar = sX + (m * 16 + c) * xs + g * 16 + line * 64;
                    const half8 fa0 = *reinterpret_cast<const half8*>(ar), fa1 = *reinterpret_cast<const half8*>(ar + 8);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, f1[t][i][0], acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa1, f1[t][i][1], acc[t][m], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[(((wid * 4 + t) * MT + m) * 4 + e) * 64 + lane] = acc[t][m][e];
    }
    __syncthreads();
    // sum over the waves (fixed order), bias, ReLU -> fp16 hidden tile [rows][64]
    for (int o = tid; o < 4 * MT * 256; o += 512) {
        const int t = o / (MT * 256), m = (o / 256) % MT, e = (o >> 6) & 3, ln = o & 63;
        float v = 0.0f;
        const int lw = lines < 8 ? lines : 8;
        for (int w = 0; w < lw; ++w) v += red[(((w * 4 + t) * MT + m) * 4 + e) * 64 + ln];
        const int row = m * 16 + (ln >> 4) * 4 + e, hc = t * 16 + (ln & 15);
        if (a.b1) v += a.b1[j * 64 + hc];
        sH[row * 72 + hc] = (_Float16)fmaxf(v, 0.0f);
    }
    __syncthreads();
    // ---- the FFN-out projection's partial sum over these 64 hidden features, added into the fixed-point residual stream
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const _Float16* ah = sH + (m * 16 + c) * 72 + g * 16;
        const half8 fa0 = *reinterpret_cast<const half8*>(ah), fa1 = *reinterpret_cast<const half8*>(ah + 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int t = wid + q * 8;
            if (t < otiles) {
                float4v y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = 0.0f;
                y = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, f2[q][0], y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa1, f2[q][1], y, 0, 0, 0);
                const int n = t * 16 + c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + g * 4 + e;
                    if (row < M && !(a.dbg & 1)) {
                        float v = y[e];
                        if (j == 0) v += from_fx(a.xin[(int64_t)row * d + n]) + (a.b2 ? a.b2[n] : 0.0f);   // the residual and the bias enter once
                        fx_add(a.xout + (int64_t)row * d + n, v);
                    }
                }
            }
        }
    }
    {
        const int nwg = gridDim.x, total = M * d, per = (total + nwg - 1) / nwg;
        for (int i = tid; i < per; i += 512) {
            const int idx = j * per + i;
            if (idx < total && !(a.dbg & 4)) a.xzero[idx] = 0;
        }
    }
}

// fixed-point residual stream -> fp32 rows (the output head's input)
__global__ __launch_bounds__(256) void lm_fx_to_f32(const long long* __restrict__ x, float* __restrict__ y, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] = from_fx(x[i]);
}

static size_t fattn_lds(int d) {
    return (size_t)FA_ROWS * (d + 8) * 2 + (size_t)FA_ROWS * d * 4 + 8 * 12 * 32 * 4 + FA_ROWS * 64 * 4 + FA_ROWS * 72 * 2 + 16 * 4 + 8 * 64 * 4 + 64;
}
static size_t fffn_lds(int d, int mt) { return (size_t)16 * mt * (d + 8) * 2 + (size_t)8 * 4 * mt * 256 * 4 + (size_t)16 * mt * 72 * 2 + 64; }

int lm_attn_block_launch(const FAttnArgs& a, hipStream_t st) {
    ASTTS_REQUIRE(a.b >= 1 && a.b <= 32 && a.d >= 64 && a.d <= 1024 && (a.d & 63) == 0 && a.heads * 64 == a.d, ASTTS_ERR_UNSUPPORTED,
                  "lm_attn_block: b=%d d=%d heads=%d (d <= 1024, head dim 64)", a.b, a.d, a.heads);
    static std::once_flag attr;
    std::call_once(attr, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_attn_block), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); });
    // algorithmic bytes (bench profiler): this layer's q | k | v and out-projection weights + the K / V / position rows, each read once
    hipEvent_t e0, e1;
    const double bytes = 4.0 * a.d * a.d * 2.0 + (double)(a.pos + 1) * a.d * 2.0 * (2.0 * a.b + 1.0);
    const dim3 grid(a.heads, (a.b + FA_ROWS - 1) / FA_ROWS);
    if (prof_events(ASTTS_PROF_GEMM_SKINNY, bytes, &e0, &e1))
        hipExtLaunchKernelGGL(lm_attn_block, grid, dim3(512), (uint32_t)fattn_lds(a.d), st, e0, e1, 0, a);
    else
        hipLaunchKernelGGL(lm_attn_block, grid, dim3(512), fattn_lds(a.d), st, a);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int lm_ffn_block_launch(const FFfnArgs& a, hipStream_t st) {
    ASTTS_REQUIRE(a.b >= 1 && a.b <= 32 && a.d >= 64 && a.d <= 1024 && (a.d & 63) == 0 && a.ffn >= 64 && (a.ffn & 63) == 0, ASTTS_ERR_UNSUPPORTED,
                  "lm_ffn_block: b=%d d=%d ffn=%d", a.b, a.d, a.ffn);
    static std::once_flag attr;
    std::call_once(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_ffn_block<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_ffn_block<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int mt = a.b <= 16 ? 1 : 2;
    hipEvent_t e0, e1;
    const bool prof = prof_events(ASTTS_PROF_GEMM_SKINNY, 2.0 * a.d * a.ffn * 2.0, &e0, &e1);
    const dim3 grid(a.ffn / 64);
    if (mt == 1) {
        if (prof) hipExtLaunchKernelGGL(lm_ffn_block<1>, grid, dim3(512), (uint32_t)fffn_lds(a.d, 1), st, e0, e1, 0, a);
        else hipLaunchKernelGGL(lm_ffn_block<1>, grid, dim3(512), fffn_lds(a.d, 1), st, a);
    } else {
        if (prof) hipExtLaunchKernelGGL(lm_ffn_block<2>, grid, dim3(512), (uint32_t)fffn_lds(a.d, 2), st, e0, e1, 0, a);
        else hipLaunchKernelGGL(lm_ffn_block<2>, grid, dim3(512), fffn_lds(a.d, 2), st, a);
    }
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int lm_fx_to_f32_launch(const long long* x, float* y, int n, hipStream_t st) {
    hipLaunchKernelGGL(lm_fx_to_f32, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0, st, x, y, n);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // namespace astts
