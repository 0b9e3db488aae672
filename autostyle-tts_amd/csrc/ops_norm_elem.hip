// ops_norm_elem.hip -- normalisation, element-wise and data-movement operators of the synthesis
// path (gfx950).  All activations are fp32, channels-last [B, T, C]: every wave touches whole
// contiguous rows (coalesced), row reductions are wavefront (64-lane) shuffles.
//
// The reference runs these inside cosyvoice (nn.LayerNorm, nn.GroupNorm+Mish of matcha Block1D,
// Snake / leaky-relu of HiFT, F.interpolate of the length regulator, the CFM Euler update, ...).
#include "common.h"

#include <cstdlib>

namespace astts {

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// mish(x) = x tanh(softplus(x)); with e = exp(x): tanh(log(1 + e)) = ((1 + e)^2 - 1) / ((1 + e)^2 + 1) = n / (n + 2),
// n = e (e + 2) -- one exp and one reciprocal instead of exp + log1p + tanh (the GroupNorm kernels were ALU-bound on it)
__device__ __forceinline__ float mishf(float x) {
    const float e = __expf(fminf(x, 20.0f));
    const float n = e * (e + 2.0f);
    return x > 20.0f ? x : x * n * __frcp_rn(n + 2.0f);
}

// ------------------------------------------------------------------ LayerNorm: one wave per row
// The row is read once into registers (c <= 2048, c % 4 == 0: 16-byte loads), statistics by 64-lane shuffles,
// output fp32 or fp16 (fp16 when the only consumer is an MFMA operand).
template <typename OutT>
__global__ __launch_bounds__(256) void layernorm_rows(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, OutT* __restrict__ y,
                                                      int64_t rows, int c, int ldx, int ldy, float eps, float relu_scale) {
    // relu_scale > 0: y = relu_scale * max(LayerNorm(x), 0) (the LM's input embedding: LayerNorm -> ReLU -> * sqrt(d))
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    OutT* yr = y + row * ldy;
    if ((c & 3) == 0 && c <= 2048 && (ldx & 3) == 0 && (ldy & 3) == 0 && ((uintptr_t)x & 15) == 0) {
        constexpr int MAXV = 8;
        float4 v[MAXV];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int k = lane * 4 + i * 256;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < c) v[i] = *reinterpret_cast<const float4*>(xr + k);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wave_sum_f32(s) / (float)c;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int k = lane * 4 + i * 256;
            if (k < c) {
                const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float rstd = rsqrtf(wave_sum_f32(q) / (float)c + eps);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int k = lane * 4 + i * 256;
            if (k < c) {
                const float4 ga = *reinterpret_cast<const float4*>(gamma + k);
                const float4 be = *reinterpret_cast<const float4*>(beta + k);
                float o0 = (v[i].x - mean) * rstd * ga.x + be.x, o1 = (v[i].y - mean) * rstd * ga.y + be.y;
                float o2 = (v[i].z - mean) * rstd * ga.z + be.z, o3 = (v[i].w - mean) * rstd * ga.w + be.w;
                if (relu_scale > 0.0f) {
                    o0 = relu_scale * fmaxf(o0, 0.0f); o1 = relu_scale * fmaxf(o1, 0.0f);
                    o2 = relu_scale * fmaxf(o2, 0.0f); o3 = relu_scale * fmaxf(o3, 0.0f);
                }
                if constexpr (sizeof(OutT) == 2) {
                    half4 h4;
                    h4[0] = (_Float16)o0; h4[1] = (_Float16)o1; h4[2] = (_Float16)o2; h4[3] = (_Float16)o3;
                    *reinterpret_cast<half4*>(yr + k) = h4;
                } else {
                    *reinterpret_cast<float4*>(yr + k) = make_float4(o0, o1, o2, o3);
                }
            }
        }
        return;
    }
    float s = 0.0f;
    for (int k = lane; k < c; k += 64) s += xr[k];
    const float mean = wave_sum_f32(s) / (float)c;
    float v = 0.0f;
    for (int k = lane; k < c; k += 64) {
        const float d = xr[k] - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum_f32(v) / (float)c + eps);
    for (int k = lane; k < c; k += 64) {
        float o = (xr[k] - mean) * rstd * gamma[k] + beta[k];
        if (relu_scale > 0.0f) o = relu_scale * fmaxf(o, 0.0f);
        yr[k] = (OutT)o;
    }
}

// ------------------------------------------------------------------ GroupNorm on [B, T, C]
// pass 1: per (b, chunk of 64 rows): sum / sum of squares per group over the valid rows
static constexpr int GN_ROWS = 64;

__global__ __launch_bounds__(256) void groupnorm_stats(const float* __restrict__ x, const int* __restrict__ lens,
                                                       float* __restrict__ partial, int t, int c, int groups,
                                                       int nchunks) {
    extern __shared__ float sh[];  // [2][c]
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int len = lens ? min(lens[b], t) : t;
    const int t0 = chunk * GN_ROWS;
    const int t1 = min(t0 + GN_ROWS, len);
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float s1 = 0.0f, s2 = 0.0f;
        const float* p = x + ((int64_t)b * t + t0) * c + ch;
        for (int r = t0; r < t1; ++r, p += c) {
            const float v = *p;
            s1 += v;
            s2 += v * v;
        }
        sh[ch] = s1;
        sh[c + ch] = s2;
    }
    __syncthreads();
    const int cpg = c / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        float s1 = 0.0f, s2 = 0.0f;
        for (int k = 0; k < cpg; ++k) {
            s1 += sh[g * cpg + k];
            s2 += sh[c + g * cpg + k];
        }
        float* o = partial + (((int64_t)b * nchunks + chunk) * groups + g) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

// pass 2: y = act(gn(x)) * mask(t < len) + add_bc[b, c].  Thread = 4 consecutive channels (per-channel scale and
// shift folded once), loop over the chunk's rows: no per-element division, 16-byte accesses when c % 4 == 0.
template <typename OutT>
__global__ __launch_bounds__(256) void groupnorm_apply(const float* __restrict__ x, const int* __restrict__ lens,
                                                       const float* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ add_bc, OutT* __restrict__ y, int t,
                                                       int c, int groups, int nchunks, float eps, int act_mish) {
    extern __shared__ float sh[];  // [2][groups]: mean, rstd
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int len = lens ? min(lens[b], t) : t;
    const int cpg = c / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < nchunks; ++k) {
            const float* p = partial + (((int64_t)b * nchunks + k) * groups + g) * 2;
            s1 += p[0];
            s2 += p[1];
        }
        const double cnt = (double)len * cpg;
        const double mean = cnt > 0 ? s1 / cnt : 0.0;
        double var = cnt > 0 ? s2 / cnt - mean * mean : 0.0;
        if (var < 0) var = 0;
        sh[g] = (float)mean;
        sh[groups + g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int t0 = chunk * GN_ROWS;
    const int t1 = min(t0 + GN_ROWS, t);
    if ((c & 3) == 0 && (cpg & 3) == 0) {
        const int cv = c >> 2;                       // float4 columns
        const int rstep = 256 / cv > 0 ? 256 / cv : 1;
        for (int col = threadIdx.x % cv, r0 = threadIdx.x / cv; col < cv && r0 < rstep; col += 256) {
            const int ch = col * 4;
            const int g = ch / cpg;
            const float mean = sh[g], rstd = sh[groups + g];
            const float4 ga = *reinterpret_cast<const float4*>(gamma + ch);
            const float4 be = *reinterpret_cast<const float4*>(beta + ch);
            const float4 sc = make_float4(rstd * ga.x, rstd * ga.y, rstd * ga.z, rstd * ga.w);
            const float4 sf = make_float4(be.x - mean * sc.x, be.y - mean * sc.y, be.z - mean * sc.z, be.w - mean * sc.w);
            float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
            if (add_bc) ad = *reinterpret_cast<const float4*>(add_bc + (int64_t)b * c + ch);
            for (int r = t0 + r0; r < t1; r += rstep) {
                const int64_t off = ((int64_t)b * t + r) * c + ch;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < len) {
                    const float4 v = *reinterpret_cast<const float4*>(x + off);
                    o = make_float4(v.x * sc.x + sf.x, v.y * sc.y + sf.y, v.z * sc.z + sf.z, v.w * sc.w + sf.w);
                    if (act_mish) o = make_float4(mishf(o.x), mishf(o.y), mishf(o.z), mishf(o.w));
                    o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                }
                if constexpr (sizeof(OutT) == 2) {
                    half4 h4;
                    h4[0] = (_Float16)o.x; h4[1] = (_Float16)o.y; h4[2] = (_Float16)o.z; h4[3] = (_Float16)o.w;
                    *reinterpret_cast<half4*>(y + off) = h4;
                } else {
                    *reinterpret_cast<float4*>(y + off) = o;
                }
            }
        }
        return;
    }
    const int64_t n = (int64_t)(t1 - t0) * c;
    const int64_t base = ((int64_t)b * t + t0) * c;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const int ch = (int)(i % c);
        const int r = t0 + (int)(i / c);
        float v = 0.0f;
        if (r < len) {
            const int g = ch / cpg;
            v = (x[base + i] - sh[g]) * sh[groups + g] * gamma[ch] + beta[ch];
            if (act_mish) v = mishf(v);
            if (add_bc) v += add_bc[(int64_t)b * c + ch];
        }
        y[base + i] = (OutT)v;
    }
}

// single-kernel GroupNorm: one block per (batch row, group) keeps its [len x cpg] tile in LDS -- the input is read
// once, statistics are the exact two-pass form, one launch instead of two.  Used when the tile fits (<= 150 KB).
static constexpr int GNF_NT = 512;    // threads per (row, group) tile (1024 measured 0.2 ms slower per flow solve)
template <typename OutT>
__global__ __launch_bounds__(GNF_NT) void groupnorm_fused(const float* __restrict__ x, const int* __restrict__ lens,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ add_bc, OutT* __restrict__ y, int t,
                                                       int c, int groups, float eps, int act_mish) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [len][cpg]
    __shared__ float red[2 * (GNF_NT / 64)];
    __shared__ float s_mean, s_rstd;
    const int b = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int len = lens ? min(lens[b], t) : t;
    const int cpg = c / groups, cv = cpg >> 2;                      // float4 columns per row
    const int nvec = len * cv;
    const float* xb = x + (int64_t)b * t * c + g * cpg;
    float s = 0.0f;
    // eight loads per thread in flight (a load -> LDS store -> add loop of runtime length is not pipelined by the compiler: the
    // 5-6 iterations of a [344 x 32] tile were 5-6 dependent memory round trips, half of this kernel's 10 us)
    constexpr int GU = 8;
    for (int i0 = tid; i0 < nvec; i0 += GNF_NT * GU) {
        float4 v[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = i0 + u * GNF_NT;
            if (i < nvec) {
                const int r = i / cv, q = i - r * cv;       // (cv is 8 for the estimator: a shift; the loads below do not wait for it)
                v[u] = *reinterpret_cast<const float4*>(xb + (int64_t)r * c + q * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = i0 + u * GNF_NT;
            if (i < nvec) {
                *reinterpret_cast<float4*>(tile + (size_t)i * 4) = v[u];
                s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            }
        }
    }
    s = wave_sum_f32(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.0f;
        for (int w = 0; w < GNF_NT / 64; ++w) tot += red[w];
        s_mean = nvec > 0 ? tot / (float)(nvec * 4) : 0.0f;
    }
    __syncthreads();
    const float mean = s_mean;
    float q2 = 0.0f;
    for (int i = tid; i < nvec; i += GNF_NT) {
        const float4 v = *reinterpret_cast<const float4*>(tile + (size_t)i * 4);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        q2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    q2 = wave_sum_f32(q2);
    if (lane == 0) red[GNF_NT / 64 + wid] = q2;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.0f;
        for (int w = 0; w < GNF_NT / 64; ++w) tot += red[GNF_NT / 64 + w];
        s_rstd = rsqrtf((nvec > 0 ? tot / (float)(nvec * 4) : 0.0f) + eps);
    }
    __syncthreads();
    const float rstd = s_rstd;
    OutT* yb = y + (int64_t)b * t * c + g * cpg;
    const int total = t * cv;                                       // rows beyond len are written as zeros
    // When the thread count is a multiple of the float4 columns per row (cv = 8 for 32-channel groups) a thread keeps ONE column
    // for the whole loop: its scale / shift / broadcast-add vectors are loaded once (as loads inside the loop they were a
    // dependent L1 round trip per iteration) and the row index advances by a constant (no division per element).
    const bool fixed_col = (GNF_NT % cv) == 0;
    const int q_fix = tid % cv, r_fix = tid / cv, r_step = GNF_NT / cv;
    float4 ga_f = make_float4(1.f, 1.f, 1.f, 1.f), be_f = make_float4(0.f, 0.f, 0.f, 0.f), ad_f = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fixed_col) {
        const int ch = g * cpg + q_fix * 4;
        ga_f = *reinterpret_cast<const float4*>(gamma + ch);
        be_f = *reinterpret_cast<const float4*>(beta + ch);
        if (add_bc) ad_f = *reinterpret_cast<const float4*>(add_bc + (int64_t)b * c + ch);
    }
    // gridDim.z workgroups per (row, group) tile: every one stages the whole tile and takes the statistics (the loads come out
    // of L2), each writes its share of the frames -- 128 tiles on 256 CUs spend 3.8 of their 8.9 us in this loop
    const int rows_per = (t + (int)gridDim.z - 1) / (int)gridDim.z;
    const int row_lo = (int)blockIdx.z * rows_per, row_hi = min(t, row_lo + rows_per);
    int r_run = r_fix + (fixed_col ? (row_lo / r_step) * r_step : 0);
    for (int i = fixed_col ? tid + (row_lo / r_step) * r_step * cv : tid; i < total; i += GNF_NT, r_run += r_step) {
        const int r = fixed_col ? r_run : i / cv, q = fixed_col ? q_fix : i - (i / cv) * cv;
        const int ch = g * cpg + q * 4;
        if (r < row_lo) continue;
        if (r >= row_hi) {
            if (fixed_col) break;                     // rows only grow along a thread's walk
            continue;
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < len) {
            const float4 v = *reinterpret_cast<const float4*>(tile + (size_t)i * 4);
            const float4 ga = fixed_col ? ga_f : *reinterpret_cast<const float4*>(gamma + ch);
            const float4 be = fixed_col ? be_f : *reinterpret_cast<const float4*>(beta + ch);
            o = make_float4((v.x - mean) * rstd * ga.x + be.x, (v.y - mean) * rstd * ga.y + be.y,
                            (v.z - mean) * rstd * ga.z + be.z, (v.w - mean) * rstd * ga.w + be.w);
            if (act_mish) o = make_float4(mishf(o.x), mishf(o.y), mishf(o.z), mishf(o.w));
            if (add_bc) {
                const float4 ad = fixed_col ? ad_f : *reinterpret_cast<const float4*>(add_bc + (int64_t)b * c + ch);
                o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
            }
        }
        OutT* op = yb + (int64_t)r * c + q * 4;
        if constexpr (sizeof(OutT) == 2) {
            half4 h4;
            h4[0] = (_Float16)o.x; h4[1] = (_Float16)o.y; h4[2] = (_Float16)o.z; h4[3] = (_Float16)o.w;
            *reinterpret_cast<half4*>(op) = h4;
        } else {
            *reinterpret_cast<float4*>(op) = o;
        }
    }
}

// ------------------------------------------------------------------ element-wise family
enum ElemOp : int {
    EL_SNAKE = 0,      // y = x + sin^2(alpha_c x) / (alpha_c + 1e-9)          (p0 = alpha[c])
    EL_LEAKY = 1,      // y = x > 0 ? x : slope x
    EL_ADD = 2,        // y = x + s * z
    EL_MUL_ROWMASK = 3,// y = x * (t < len[b])
    EL_ADD_BC = 4,     // y = x + v[b, c]
    EL_SCALE = 5,      // y = s * x
    EL_CFG_EULER = 6,  // y = x + dt * ((1 + r) * z_cond - r * z_uncond)   z = [2B,...]: cond rows then uncond rows
    EL_MISH = 7,
    EL_SILU = 8,
    EL_CLAMP = 9,      // y = clamp(x, -s, s)
    EL_TANH = 10,
    EL_ELU = 11,
    EL_RELU_SCALE = 12,  // y = s * max(x, 0)
};

struct ElemArgs {
    const float* x;
    const float* z;
    const float* p0;
    const int* lens;
    float* y;
    int64_t total;  // B*T*C
    int t, c;
    float s, s2;
    int op;
};

__global__ __launch_bounds__(256) void elementwise(ElemArgs a) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.total; i += stride) {
        const float x = a.x[i];
        float y;
        switch (a.op) {
            case EL_SNAKE: {
                const float al = a.p0[i % a.c];
                const float sn = __sinf(al * x);
                y = x + sn * sn / (al + 1e-9f);
                break;
            }
            case EL_LEAKY: y = x > 0.0f ? x : x * a.s; break;
            case EL_ADD: y = x + a.s * a.z[i]; break;
            case EL_MUL_ROWMASK: {
                const int64_t row = i / a.c;
                const int b = (int)(row / a.t), tt = (int)(row % a.t);
                y = (tt < a.lens[b]) ? x : 0.0f;
                break;
            }
            case EL_ADD_BC: {
                const int64_t row = i / a.c;
                y = x + a.p0[(row / a.t) * a.c + (i % a.c)];
                break;
            }
            case EL_SCALE: y = a.s * x; break;
            case EL_CFG_EULER: y = x + a.s * ((1.0f + a.s2) * a.z[i] - a.s2 * a.z[i + a.total]); break;
            case EL_MISH: y = mishf(x); break;
            case EL_SILU: y = x / (1.0f + __expf(-x)); break;
            case EL_CLAMP: y = fminf(fmaxf(x, -a.s), a.s); break;
            case EL_TANH: y = tanhf(x); break;
            case EL_ELU: y = x > 0.0f ? x : (__expf(x) - 1.0f); break;
            case EL_RELU_SCALE: y = a.s * fmaxf(x, 0.0f); break;
            default: y = x;
        }
        a.y[i] = y;
    }
}

// ------------------------------------------------------------------ embedding gather: y[r, :] = table[ids[r], :] * scale
__global__ __launch_bounds__(256) void embedding_rows(const float* __restrict__ table, const int* __restrict__ ids,
                                                      float* __restrict__ y, int64_t rows, int c, int ldy,
                                                      int vocab, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* src = table + (int64_t)id * c;
    float* dst = y + row * ldy;
    for (int k = lane; k < c; k += 64) dst[k] = src[k] * scale;
}

// ------------------------------------------------------------------ linear interpolation along T (F.interpolate, align_corners=False)
// in_lens / out_lens (nullable): per-row valid lengths of a ragged batch -- row b is resampled from its own in_lens[b]
// steps to its own out_lens[b] steps (exactly F.interpolate on that row alone); outputs beyond out_lens[b] are 0.
__global__ __launch_bounds__(256) void interp_linear_rows(const float* __restrict__ x, float* __restrict__ y, int b,
                                                          int t_in_max, int t_out_max, int c,
                                                          const int* __restrict__ in_lens, const int* __restrict__ out_lens) {
    const int64_t total = (int64_t)b * t_out_max * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % c);
        const int64_t row = i / c;
        const int bb = (int)(row / t_out_max), to = (int)(row % t_out_max);
        const int t_in = in_lens ? min(in_lens[bb], t_in_max) : t_in_max;
        const int t_out = out_lens ? min(out_lens[bb], t_out_max) : t_out_max;
        if (to >= t_out || t_in < 1) {
            y[i] = 0.0f;
            continue;
        }
        const float scale = (float)t_in / (float)t_out;
        float src = ((float)to + 0.5f) * scale - 0.5f;
        if (src < 0.0f) src = 0.0f;
        int i0 = (int)src;
        if (i0 > t_in - 1) i0 = t_in - 1;
        const int i1 = min(i0 + 1, t_in - 1);
        const float w1 = src - (float)i0;
        const float* p = x + (int64_t)bb * t_in_max * c + ch;
        y[i] = p[(int64_t)i0 * c] * (1.0f - w1) + p[(int64_t)i1 * c] * w1;
    }
}

// ------------------------------------------------------------------ sinusoidal timestep embedding (matcha SinusoidalPosEmb, scale 1000)
__global__ void time_embedding(const float* __restrict__ t, float* __restrict__ y, int b, int dim, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= b * half) return;
    const int bb = i / half, k = i % half;
    const float f = __expf(-logf(10000.0f) * (float)k / (float)(half - 1));
    const float arg = scale * t[bb] * f;
    y[(int64_t)bb * dim + k] = sinf(arg);
    y[(int64_t)bb * dim + half + k] = cosf(arg);
}

}  // namespace astts

using namespace astts;

static inline int grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" {

int astts_op_layernorm_relu(const float* x, const float* gamma, const float* beta, void* y, int32_t out_f16, int64_t rows,
                            int32_t c, int32_t ldx, int32_t ldy, float eps, float relu_scale, astts_stream_t stream) {
    ASTTS_REQUIRE(x && gamma && beta && y && rows >= 1 && c >= 1 && relu_scale >= 0.0f, ASTTS_ERR_INVALID, "astts_op_layernorm: bad argument");
    if (out_f16)
        hipLaunchKernelGGL((layernorm_rows<_Float16>), dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x,
                           gamma, beta, (_Float16*)y, rows, c, ldx, ldy, eps, relu_scale);
    else
        hipLaunchKernelGGL((layernorm_rows<float>), dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                           beta, (float*)y, rows, c, ldx, ldy, eps, relu_scale);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_layernorm_ex(const float* x, const float* gamma, const float* beta, void* y, int32_t out_f16, int64_t rows,
                          int32_t c, int32_t ldx, int32_t ldy, float eps, astts_stream_t stream) {
    return astts_op_layernorm_relu(x, gamma, beta, y, out_f16, rows, c, ldx, ldy, eps, 0.0f, stream);
}

int astts_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t c,
                       int32_t ldx, int32_t ldy, float eps, astts_stream_t stream) {
    return astts_op_layernorm_ex(x, gamma, beta, y, 0, rows, c, ldx, ldy, eps, stream);
}

size_t astts_op_groupnorm_workspace_bytes(int32_t b, int32_t t, int32_t groups) {
    return (size_t)b * cdiv(t, GN_ROWS) * groups * 2 * sizeof(float);
}

int astts_op_groupnorm(const float* x, const int32_t* lens, const float* gamma, const float* beta,
                       const float* add_bc, float* y, int32_t b, int32_t t, int32_t c, int32_t groups, float eps,
                       int32_t act_mish, void* workspace, size_t workspace_bytes, astts_stream_t stream) {
    return astts_op_groupnorm_ex(x, lens, gamma, beta, add_bc, y, 0, b, t, c, groups, eps, act_mish, workspace, workspace_bytes, stream);
}

int astts_op_groupnorm_ex(const float* x, const int32_t* lens, const float* gamma, const float* beta,
                          const float* add_bc, void* y, int32_t out_f16, int32_t b, int32_t t, int32_t c, int32_t groups,
                          float eps, int32_t act_mish, void* workspace, size_t workspace_bytes, astts_stream_t stream) {
    ASTTS_REQUIRE(x && gamma && beta && y && workspace, ASTTS_ERR_INVALID, "astts_op_groupnorm: null pointer");
    ASTTS_REQUIRE(b >= 1 && t >= 1 && c >= 1 && groups >= 1 && c % groups == 0 && groups <= 256 && c <= 8192,
                  ASTTS_ERR_INVALID, "astts_op_groupnorm: bad shape b=%d t=%d c=%d groups=%d", b, t, c, groups);
    const int nch = (int)cdiv(t, GN_ROWS);
    ASTTS_REQUIRE(workspace_bytes >= astts_op_groupnorm_workspace_bytes(b, t, groups), ASTTS_ERR_WORKSPACE,
                  "astts_op_groupnorm: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int cpg = c / groups;
    const size_t tile_bytes = (size_t)t * cpg * sizeof(float);
    if ((cpg & 3) == 0 && (c & 3) == 0 && tile_bytes <= 150 * 1024 && ((uintptr_t)x & 15) == 0) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&groupnorm_fused<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&groupnorm_fused<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            attr_set = true;
        }
        const unsigned gz = (int64_t)b * groups * 2 <= 256 ? 2u : 1u;      // half the chip idle otherwise: two workgroups per tile share the write-out
        if (out_f16)
            hipLaunchKernelGGL((groupnorm_fused<_Float16>), dim3(b, groups, gz), dim3(GNF_NT), tile_bytes, st, x, lens, gamma, beta, add_bc,
                               (_Float16*)y, t, c, groups, eps, act_mish);
        else
            hipLaunchKernelGGL((groupnorm_fused<float>), dim3(b, groups, gz), dim3(GNF_NT), tile_bytes, st, x, lens, gamma, beta, add_bc,
                               (float*)y, t, c, groups, eps, act_mish);
        ASTTS_CHECK_LAUNCH();
        return ASTTS_OK;
    }
    hipLaunchKernelGGL(groupnorm_stats, dim3(b, nch), dim3(256), 2 * c * sizeof(float), st, x, lens, (float*)workspace,
                       t, c, groups, nch);
    ASTTS_CHECK_LAUNCH();
    if (out_f16)
        hipLaunchKernelGGL((groupnorm_apply<_Float16>), dim3(b, nch), dim3(256), 2 * groups * sizeof(float), st, x, lens,
                           (const float*)workspace, gamma, beta, add_bc, (_Float16*)y, t, c, groups, nch, eps, act_mish);
    else
        hipLaunchKernelGGL((groupnorm_apply<float>), dim3(b, nch), dim3(256), 2 * groups * sizeof(float), st, x, lens,
                           (const float*)workspace, gamma, beta, add_bc, (float*)y, t, c, groups, nch, eps, act_mish);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_elementwise(int32_t op, const float* x, const float* z, const float* p0, const int32_t* lens, float* y,
                         int64_t total, int32_t t, int32_t c, float s, float s2, astts_stream_t stream) {
    ASTTS_REQUIRE(x && y && total >= 1 && c >= 1 && t >= 1, ASTTS_ERR_INVALID, "astts_op_elementwise: bad argument");
    ASTTS_REQUIRE(op >= EL_SNAKE && op <= EL_RELU_SCALE, ASTTS_ERR_INVALID, "astts_op_elementwise: op=%d", op);
    ASTTS_REQUIRE(!((op == EL_ADD || op == EL_CFG_EULER) && !z), ASTTS_ERR_INVALID, "astts_op_elementwise: z is null");
    ASTTS_REQUIRE(!((op == EL_SNAKE || op == EL_ADD_BC) && !p0), ASTTS_ERR_INVALID, "astts_op_elementwise: p0 is null");
    ASTTS_REQUIRE(!(op == EL_MUL_ROWMASK && !lens), ASTTS_ERR_INVALID, "astts_op_elementwise: lens is null");
    ElemArgs a{x, z, p0, lens, y, total, t, c, s, s2, op};
    hipLaunchKernelGGL(elementwise, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_embedding(const float* table, const int32_t* ids, float* y, int64_t rows, int32_t c, int32_t ldy,
                       int32_t vocab, float scale, astts_stream_t stream) {
    ASTTS_REQUIRE(table && ids && y && rows >= 1 && c >= 1 && vocab >= 1, ASTTS_ERR_INVALID, "astts_op_embedding: bad argument");
    hipLaunchKernelGGL(embedding_rows, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, table, ids, y,
                       rows, c, ldy, vocab, scale);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_interp_linear_ex(const float* x, float* y, int32_t b, int32_t t_in, int32_t t_out, int32_t c,
                              const int32_t* in_lens, const int32_t* out_lens, astts_stream_t stream) {
    ASTTS_REQUIRE(x && y && b >= 1 && t_in >= 1 && t_out >= 1 && c >= 1, ASTTS_ERR_INVALID, "astts_op_interp_linear: bad argument");
    hipLaunchKernelGGL(interp_linear_rows, dim3(grid_for((int64_t)b * t_out * c)), dim3(256), 0, (hipStream_t)stream, x,
                       y, b, t_in, t_out, c, in_lens, out_lens);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_interp_linear(const float* x, float* y, int32_t b, int32_t t_in, int32_t t_out, int32_t c,
                           astts_stream_t stream) {
    return astts_op_interp_linear_ex(x, y, b, t_in, t_out, c, nullptr, nullptr, stream);
}

int astts_op_time_embedding(const float* t, float* y, int32_t b, int32_t dim, float scale, astts_stream_t stream) {
    ASTTS_REQUIRE(t && y && b >= 1 && dim >= 4 && dim % 2 == 0, ASTTS_ERR_INVALID, "astts_op_time_embedding: bad argument");
    const int total = b * (dim / 2);
    hipLaunchKernelGGL(time_embedding, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, y, b, dim, scale);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
