// ops_llm.hip -- operators of the retrieval path's query embedder: the Llama-3.2-3B decoder
// whose mean-pooled last hidden state is one half of a style-bank query (/root/reference/src/search_milvus.py:75-108)
// and whose greedy continuation is the emotion label (/root/reference/milvus/search_json.py:154-198).
// The dense contractions go through the GEMM family of ops_gemm.hip (fp16 activations: the LDS-DMA ring kernel); this
// file holds what a Llama block needs besides them:
//   rmsnorm_rows       x * rsqrt(mean(x^2) + eps) * w, fp32 residual stream in, fp16 MFMA operand out (one wave per row)
//   rope_llama         rotate-half RoPE on the q and k heads of a fused q|k|v buffer, in place (fp32 math on fp16 data)
//   attn_causal_gqa    causal grouped-query attention at head dimension 128 (the flash kernel of ops_attention.hip is
//                      built around 64): 32 queries x 8 lanes per workgroup, K / V tiles staged in LDS, online softmax.
//                      Prompts are <= 512 tokens (src/search_milvus.py:92): per layer ~1 % of the projection flops.
//   swiglu             silu(gate) * up on the fused gate|up projection
//   mean_pool          masked mean over the tokens of each text (the embedding itself)
#include "common.h"

namespace astts {

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename OutT>
__global__ __launch_bounds__(256) void rmsnorm_rows(const float* __restrict__ x, const float* __restrict__ w, OutT* __restrict__ y,
                                                    int64_t rows, int c, int ldx, int ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    OutT* yr = y + row * ldy;
    float ss = 0.0f;
    if ((c & 3) == 0 && (ldx & 3) == 0 && ((uintptr_t)x & 15) == 0) {
        for (int k = lane * 4; k < c; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + k);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (int k = lane; k < c; k += 64) ss += xr[k] * xr[k];
    }
    const float r = rsqrtf(wave_sum64(ss) / (float)c + eps);
    for (int k = lane; k < c; k += 64) yr[k] = (OutT)(w[k] * (xr[k] * r));     // w * (x * rstd): the order transformers uses
}

// x: fp16 [b*t][ld], heads laid out [head][head_dim] from column 0; cos / sin: fp32 [>= pos0 + t][head_dim / 2]
// (the table's two halves are equal: emb = cat(freqs, freqs)).  out[i] = x[i] cos_i - x[i + h] sin_i, out[i + h] = x[i + h] cos_i + x[i] sin_i
__global__ __launch_bounds__(256) void rope_llama(_Float16* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn,
                                                  int64_t rows, int t, int heads, int ld, int head_dim, int pos0) {
    const int half = head_dim >> 1;
    const int64_t total = rows * heads * half;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % half);
        const int64_t rh = i / half;
        const int hd = (int)(rh % heads);
        const int64_t row = rh / heads;
        const int pos = pos0 + (int)(row % t);
        _Float16* p = x + row * ld + hd * head_dim + j;
        const float a = (float)p[0], b = (float)p[half];
        const float c = cs[(int64_t)pos * half + j], s = sn[(int64_t)pos * half + j];
        p[0] = (_Float16)(a * c - b * s);
        p[half] = (_Float16)(b * c + a * s);
    }
}

// ---- causal GQA attention, head_dim 128.  Workgroup = (32 queries, head, batch row); thread (qi = tid >> 3, sub = tid & 7)
// owns dims [16 sub, 16 sub + 16) of query qi.  Keys come in tiles of 32 through LDS (fp16); only tiles at or below the
// diagonal are visited.
static constexpr int GA_Q = 32, GA_K = 32, GA_D = 128;

__global__ __launch_bounds__(256) void attn_causal_gqa(const _Float16* __restrict__ q, const _Float16* __restrict__ k,
                                                       const _Float16* __restrict__ v, const int* __restrict__ lens,
                                                       _Float16* __restrict__ out, int t, int heads, int kv_heads, int ldq, int ldk,
                                                       int ldo, float scale) {
    __shared__ __attribute__((aligned(16))) _Float16 sk[GA_K][GA_D + 8];
    __shared__ __attribute__((aligned(16))) _Float16 sv[GA_K][GA_D + 8];
    const int tid = threadIdx.x, qi = tid >> 3, sub = tid & 7;
    const int q0 = blockIdx.x * GA_Q, head = blockIdx.y, b = blockIdx.z;
    const int kvh = head / (heads / kv_heads);
    const int len = lens ? min(lens[b], t) : t;
    const int qrow = q0 + qi;
    const bool qvalid = qrow < len;
    float qf[16];
    {
        const _Float16* qp = q + ((int64_t)b * t + min(qrow, t - 1)) * ldq + head * GA_D + sub * 16;
        const half8 a = *reinterpret_cast<const half8*>(qp), c = *reinterpret_cast<const half8*>(qp + 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qf[e] = (float)a[e] * scale;
            qf[8 + e] = (float)c[e] * scale;
        }
    }
    float m_run = -INFINITY, l_run = 0.0f, o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.0f;
    const int kend = min(len, q0 + GA_Q);                   // causal: no key beyond the tile's last query
    for (int k0 = 0; k0 < kend; k0 += GA_K) {
        __syncthreads();
        {   // stage 32 keys x 128 dims of K and V: thread -> (key = tid >> 3, 16 dims)
            const int kr = min(k0 + qi, t - 1);
            const _Float16* kp = k + ((int64_t)b * t + kr) * ldk + kvh * GA_D + sub * 16;
            const _Float16* vp = v + ((int64_t)b * t + kr) * ldk + kvh * GA_D + sub * 16;
            *reinterpret_cast<half8*>(&sk[qi][sub * 16]) = *reinterpret_cast<const half8*>(kp);
            *reinterpret_cast<half8*>(&sk[qi][sub * 16 + 8]) = *reinterpret_cast<const half8*>(kp + 8);
            *reinterpret_cast<half8*>(&sv[qi][sub * 16]) = *reinterpret_cast<const half8*>(vp);
            *reinterpret_cast<half8*>(&sv[qi][sub * 16 + 8]) = *reinterpret_cast<const half8*>(vp + 8);
        }
        __syncthreads();
        const int nk = min(GA_K, kend - k0);
        for (int j = 0; j < nk; ++j) {
            const half8 ka = *reinterpret_cast<const half8*>(&sk[j][sub * 16]), kb = *reinterpret_cast<const half8*>(&sk[j][sub * 16 + 8]);
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qf[e] * (float)ka[e] + qf[8 + e] * (float)kb[e];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if (k0 + j > qrow) continue;                     // causal mask (uniform over the 8 lanes of a query)
            const float m_new = fmaxf(m_run, s);
            const float sc = m_run == -INFINITY ? 0.0f : __expf(m_run - m_new);
            const float p = __expf(s - m_new);
            l_run = l_run * sc + p;
            const half8 va = *reinterpret_cast<const half8*>(&sv[j][sub * 16]), vb = *reinterpret_cast<const half8*>(&sv[j][sub * 16 + 8]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = o[e] * sc + p * (float)va[e];
                o[8 + e] = o[8 + e] * sc + p * (float)vb[e];
            }
            m_run = m_new;
        }
    }
    if (qrow < t) {
        _Float16* op = out + ((int64_t)b * t + qrow) * ldo + head * GA_D + sub * 16;
        const float inv = (qvalid && l_run > 0.0f) ? 1.0f / l_run : 0.0f;
        half8 a, c;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] = (_Float16)(o[e] * inv);
            c[e] = (_Float16)(o[8 + e] * inv);
        }
        *reinterpret_cast<half8*>(op) = a;
        *reinterpret_cast<half8*>(op + 8) = c;
    }
}

// gu: fp16 [rows][2f] = gate | up  ->  out fp16 [rows][f] = silu(gate) * up
__global__ __launch_bounds__(256) void swiglu_rows(const _Float16* __restrict__ gu, _Float16* __restrict__ out, int64_t rows, int f, int ldg, int ldo) {
    const int64_t total = rows * (f >> 3);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / (f >> 3);
        const int c = (int)(i - row * (f >> 3)) << 3;
        const half8 g = *reinterpret_cast<const half8*>(gu + row * ldg + c), u = *reinterpret_cast<const half8*>(gu + row * ldg + f + c);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (float)g[e];
            o[e] = (_Float16)(x / (1.0f + __expf(-x)) * (float)u[e]);
        }
        *reinterpret_cast<half8*>(out + row * ldo + c) = o;
    }
}

// x: fp32 [b][t][c] -> out [b][c] = mean over the first lens[b] (or t) tokens
__global__ __launch_bounds__(256) void mean_pool(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ out, int t, int c) {
    const int b = blockIdx.y;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= c) return;
    const int n = lens ? min(lens[b], t) : t;
    const float* p = x + (int64_t)b * t * c + col;
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc += p[(int64_t)i * c];          // fixed order: the same bits on every run
    out[(int64_t)b * c + col] = n > 0 ? acc / (float)n : 0.0f;
}

}  // namespace astts

using namespace astts;

extern "C" {

int astts_op_rmsnorm(const float* x, const float* w, void* y, int32_t out_f16, int64_t rows, int32_t c, int32_t ldx, int32_t ldy,
                     float eps, astts_stream_t stream) {
    ASTTS_REQUIRE(x && w && y, ASTTS_ERR_INVALID, "astts_op_rmsnorm: null pointer");
    ASTTS_REQUIRE(rows >= 1 && c >= 1 && ldx >= c && ldy >= c, ASTTS_ERR_INVALID, "astts_op_rmsnorm: bad shape rows=%lld c=%d", (long long)rows, c);
    const dim3 grid((unsigned)cdiv(rows, 4));
    if (out_f16)
        hipLaunchKernelGGL((rmsnorm_rows<_Float16>), grid, dim3(256), 0, (hipStream_t)stream, x, w, (_Float16*)y, rows, c, ldx, ldy, eps);
    else
        hipLaunchKernelGGL((rmsnorm_rows<float>), grid, dim3(256), 0, (hipStream_t)stream, x, w, (float*)y, rows, c, ldx, ldy, eps);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_rope_llama(void* x_f16, const float* cos_tab, const float* sin_tab, int32_t b, int32_t t, int32_t heads, int32_t ld,
                        int32_t head_dim, int32_t pos0, astts_stream_t stream) {
    ASTTS_REQUIRE(x_f16 && cos_tab && sin_tab, ASTTS_ERR_INVALID, "astts_op_rope_llama: null pointer");
    ASTTS_REQUIRE(b >= 1 && t >= 1 && heads >= 1 && head_dim >= 2 && (head_dim & 1) == 0 && ld >= heads * head_dim && pos0 >= 0, ASTTS_ERR_INVALID,
                  "astts_op_rope_llama: bad shape b=%d t=%d heads=%d head_dim=%d ld=%d", b, t, heads, head_dim, ld);
    const int64_t total = (int64_t)b * t * heads * (head_dim / 2);
    const unsigned blocks = (unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(rope_llama, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (_Float16*)x_f16, cos_tab, sin_tab, (int64_t)b * t, t, heads,
                       ld, head_dim, pos0);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_attn_causal_gqa(const void* q_f16, const void* k_f16, const void* v_f16, const int32_t* lens, void* out_f16, int32_t b,
                             int32_t t, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                             astts_stream_t stream) {
    ASTTS_REQUIRE(q_f16 && k_f16 && v_f16 && out_f16, ASTTS_ERR_INVALID, "astts_op_attn_causal_gqa: null pointer");
    ASTTS_REQUIRE(head_dim == GA_D, ASTTS_ERR_UNSUPPORTED, "astts_op_attn_causal_gqa: head_dim=%d (built for 128; 64 is astts_op_attn_mha)", head_dim);
    ASTTS_REQUIRE(b >= 1 && t >= 1 && heads >= 1 && kv_heads >= 1 && heads % kv_heads == 0 && (ldq & 7) == 0 && (ldk & 7) == 0 && (ldo & 7) == 0 &&
                      ((uintptr_t)q_f16 & 15) == 0 && ((uintptr_t)k_f16 & 15) == 0 && ((uintptr_t)v_f16 & 15) == 0 && ((uintptr_t)out_f16 & 15) == 0,
                  ASTTS_ERR_INVALID, "astts_op_attn_causal_gqa: bad shape / alignment b=%d t=%d heads=%d kv_heads=%d", b, t, heads, kv_heads);
    hipLaunchKernelGGL(attn_causal_gqa, dim3((unsigned)cdiv(t, GA_Q), heads, b), dim3(256), 0, (hipStream_t)stream, (const _Float16*)q_f16,
                       (const _Float16*)k_f16, (const _Float16*)v_f16, lens, (_Float16*)out_f16, t, heads, kv_heads, ldq, ldk, ldo, scale);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_swiglu(const void* gate_up_f16, void* out_f16, int64_t rows, int32_t f, int32_t ldg, int32_t ldo, astts_stream_t stream) {
    ASTTS_REQUIRE(gate_up_f16 && out_f16, ASTTS_ERR_INVALID, "astts_op_swiglu: null pointer");
    ASTTS_REQUIRE(rows >= 1 && f >= 8 && (f & 7) == 0 && ldg >= 2 * f && ldo >= f && (ldg & 7) == 0 && (ldo & 7) == 0 &&
                      ((uintptr_t)gate_up_f16 & 15) == 0 && ((uintptr_t)out_f16 & 15) == 0,
                  ASTTS_ERR_INVALID, "astts_op_swiglu: bad shape rows=%lld f=%d", (long long)rows, f);
    const int64_t total = rows * (f / 8);
    const unsigned blocks = (unsigned)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    hipLaunchKernelGGL(swiglu_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)gate_up_f16, (_Float16*)out_f16, rows, f, ldg, ldo);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_mean_pool(const float* x, const int32_t* lens, float* out, int32_t b, int32_t t, int32_t c, astts_stream_t stream) {
    ASTTS_REQUIRE(x && out, ASTTS_ERR_INVALID, "astts_op_mean_pool: null pointer");
    ASTTS_REQUIRE(b >= 1 && t >= 1 && c >= 1, ASTTS_ERR_INVALID, "astts_op_mean_pool: bad shape b=%d t=%d c=%d", b, t, c);
    hipLaunchKernelGGL(mean_pool, dim3((unsigned)cdiv(c, 256), b), dim3(256), 0, (hipStream_t)stream, x, lens, out, t, c);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
