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


// ---- causal GQA attention on the matrix cores, head_dim 128 (round 5; the VALU kernel above stays as the second implementation the
// tests compare with).  The skeleton of attn_mha_flash (ops_attention.hip) at twice the head dimension:
//   workgroup = 4 waves x 32 queries of one (head, batch row); keys in tiles of 64 through LDS (K row-major, V transposed);
//   S^T = K Q^T with v_mfma_f32_32x32x16_f16 (8 k-steps over the 128 dims) so that the QUERY sits on the lane: the online softmax is
//   lane-local (one cross-half exchange) in the log2 domain, P^T feeds O^T += V^T P^T straight from the accumulator registers (the
//   k order inside a step follows the accumulator layout: element j of lane half h is key 16 s + 8 (j >> 2) + 4 h + (j & 3)).
// Generalised for the generation path: explicit (batch, time) strides (time-major KV cache), `pos0` = key index of query 0 (a decode
// step's single query attends the whole cache), `key_start` (left-padded prompts: keys before a row's first token are masked), `lens`
// (right-padded batches: keys at or beyond are masked, queries at or beyond produce zeros).  The three query heads of a KV group are
// separate workgroups (K / V tiles come from L2 the second and third time: prompts are <= 512 tokens).
static constexpr int GM_KT = 64;    // keys per staged tile
static constexpr int GM_KS = 136;   // halfs per K row in LDS (128 + 8): 68 dwords, 16 lanes x 16 B cover the 64 banks once
static constexpr int GM_VS = 68;    // halfs per V^T row (as FA_VS: the 32 dims a half-wave reads with ds_read_b64 start on 32 different even banks)
static constexpr int GM_OS = 136;   // halfs per output row of the epilogue transpose (aliases the K / V^T images)

struct GqaArgs {
    const _Float16* q;
    const _Float16* k;
    const _Float16* v;
    const int* lens;
    const int* key_start;
    _Float16* out;
    int64_t sqb, sqt, skb, skt, sob, sot;   // element strides of batch row / time step
    int tq, tk, pos0, heads, kv_heads;
    float scale;
};

__global__ __launch_bounds__(256) void attn_gqa_mfma(GqaArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[GM_KT * GM_KS + GA_D * GM_VS];      // 17 408 + 17 408 bytes
    _Float16* ks = smem;
    _Float16* vt = smem + GM_KT * GM_KS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int kvh = head / (a.heads / a.kv_heads);
    const int qblk = blockIdx.x * 128, q0 = qblk + wid * 32;
    const int len = a.lens ? min(a.lens[b], a.tk) : a.tk;
    const int kstart = a.key_start ? max(a.key_start[b], 0) : 0;
    const int kend = min(len, a.pos0 + min(qblk + 128, a.tq));       // causal: no key beyond the block's last query
    const _Float16* qp = a.q + (int64_t)b * a.sqb + head * GA_D;
    const _Float16* kp = a.k + (int64_t)b * a.skb + kvh * GA_D;
    const _Float16* vp = a.v + (int64_t)b * a.skb + kvh * GA_D;

    // Q fragments (B operand of S^T = K Q^T): lane (c, hh) holds Q[q0 + c][16 s + 8 hh + j] * scale * log2(e)
    half8 qf[8];
    {
        const _Float16* qr = qp + (int64_t)min(q0 + c, a.tq - 1) * a.sqt;
        const float sc = a.scale * 1.44269504088896341f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const half8 x = *reinterpret_cast<const half8*>(qr + 16 * s + 8 * hh);
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[s][i] = (_Float16)((float)x[i] * sc);
        }
    }
    float16v ot[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[dt][e] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    // staging: K -- key row skey, 32 dims from sd0; V -- key PAIR vkp, 16 dims from vd0, written transposed as whole dwords
    const int skey = tid >> 2, sd0 = (tid & 3) * 32;
    const int vkp = tid >> 3, vd0 = (tid & 7) * 16;
    half8 rk[4], rv[4];
    auto prefetch = [&](int j0) {
        const int j = min(j0 + skey, a.tk - 1);                       // clamped; masked in the scores
        const _Float16* p = kp + (int64_t)j * a.skt + sd0;
#pragma unroll
        for (int i = 0; i < 4; ++i) rk[i] = *reinterpret_cast<const half8*>(p + 8 * i);
        const int jv0 = min(j0 + 2 * vkp, a.tk - 1), jv1 = min(j0 + 2 * vkp + 1, a.tk - 1);
        const _Float16* p0 = vp + (int64_t)jv0 * a.skt + vd0;
        const _Float16* p1 = vp + (int64_t)jv1 * a.skt + vd0;
        rv[0] = *reinterpret_cast<const half8*>(p0);
        rv[1] = *reinterpret_cast<const half8*>(p0 + 8);
        rv[2] = *reinterpret_cast<const half8*>(p1);
        rv[3] = *reinterpret_cast<const half8*>(p1 + 8);
    };
    const int jfirst = (kstart / GM_KT) * GM_KT;                      // tiles wholly before the row's first token are skipped
    if (jfirst < kend) prefetch(jfirst);
    for (int j0 = jfirst; j0 < kend; j0 += GM_KT) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<half8*>(&ks[skey * GM_KS + sd0 + 8 * i]) = rk[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            half2v pr;
            pr[0] = rv[i >> 3][i & 7];
            pr[1] = rv[2 + (i >> 3)][i & 7];
            *reinterpret_cast<half2v*>(&vt[(vd0 + i) * GM_VS + 2 * vkp]) = pr;
        }
        __syncthreads();
        if (j0 + GM_KT < kend) prefetch(j0 + GM_KT);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int jb = j0 + sub * 32;
            if (jb >= kend || jb > a.pos0 + q0 + 31) break;           // wave-uniform: beyond this wave's last query
            float16v st;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = 0.0f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const half8 kf = *reinterpret_cast<const half8*>(&ks[(sub * 32 + c) * GM_KS + 16 * s + 8 * hh]);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st, 0, 0, 0);
            }
            // masks only where they can bite (wave-uniform test): the ragged end, the diagonal, the left padding
            if (jb + 32 > len || jb + 31 > a.pos0 + q0 || jb < kstart) {
                const int qpos = a.pos0 + q0 + c;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = jb + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    st[e] = (key < len && key <= qpos && key >= kstart) ? st[e] : -INFINITY;
                }
            }
            float mloc = st[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mloc = fmaxf(mloc, st[e]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            if (__builtin_amdgcn_ballot_w64(mloc > m_run) != 0) {
                const float m_new = fmaxf(m_run, mloc);
                const float alpha = m_run == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ot[dt][e] *= alpha;
            }
            half8 pf[2];
            const float m_use = m_run == -INFINITY ? 0.0f : m_run;    // a query that has seen no valid key yet: every p is exp2(-inf) = 0
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(st[e] - m_use);
                l_run += p;
                pf[e >> 3][e & 7] = (_Float16)p;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const _Float16* vrow = &vt[(dt * 32 + c) * GM_VS];
                    const int key0 = sub * 32 + 16 * s + 4 * hh;
                    const half4 lo = *reinterpret_cast<const half4*>(vrow + key0);
                    const half4 hi = *reinterpret_cast<const half4*>(vrow + key0 + 8);
                    half8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], ot[dt], 0, 0, 0);
                }
            }
        }
    }
    // O^T (dims in registers, query on the lane) -> this wave's slice of the (now idle) LDS -> whole 256-byte rows
    l_run += __shfl_xor(l_run, 32, 64);
    const bool qvalid = a.pos0 + q0 + c < len;                        // a padded query (right padding) produces zeros
    const float inv = (qvalid && l_run > 0.0f) ? 1.0f / l_run : 0.0f;
    __syncthreads();                                                  // every wave is done with the K / V^T images
    _Float16* so = smem + wid * 32 * GM_OS;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 o4;
#pragma unroll
            for (int i = 0; i < 4; ++i) o4[i] = (_Float16)(ot[dt][4 * g + i] * inv);
            *reinterpret_cast<half4*>(&so[c * GM_OS + dt * 32 + 8 * g + 4 * hh]) = o4;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 4 + (lane >> 4), col = (lane & 15) * 8;
        const int qi = q0 + r;
        if (qi < a.tq)
            *reinterpret_cast<half8*>(a.out + (int64_t)b * a.sob + (int64_t)qi * a.sot + head * GA_D + col) =
                *reinterpret_cast<const half8*>(&so[r * GM_OS + col]);
    }
}

// rope_llama with an explicit row -> (time step, batch row) map and a per-row position shift (left-padded prompts of the generation
// path: the first real token of row b sits at time step shift[b] and must get position 0, as transformers rotates it)
__global__ __launch_bounds__(256) void rope_llama_ex(_Float16* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn,
                                                     const int* __restrict__ shift, int64_t rows, int t, int bsz, int time_major, int heads,
                                                     int ld, int head_dim, int pos0) {
    const int half = head_dim >> 1;
    const int64_t total = rows * heads * half;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % half);
        const int64_t rh = i / half;
        const int hd = (int)(rh % heads);
        const int64_t row = rh / heads;
        const int ti = time_major ? (int)(row / bsz) : (int)(row % t);
        const int bi = time_major ? (int)(row % bsz) : (int)(row / t);
        const int pos = max(pos0 + ti - (shift ? shift[bi] : 0), 0);
        _Float16* p = x + row * ld + hd * head_dim + j;
        const float a = (float)p[0], b = (float)p[half];
        const float c = cs[(int64_t)pos * half + j], s = sn[(int64_t)pos * half + j];
        p[0] = (_Float16)(a * c - b * s);
        p[half] = (_Float16)(b * c + a * s);
    }
}

// greedy step: out[row] = index of the largest of x[row][0 .. n) (ties: the lowest index, as torch.argmax); one workgroup per row
__global__ __launch_bounds__(256) void argmax_rows(const float* __restrict__ x, int* __restrict__ out, int n, int64_t ld) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const float* xr = x + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = xr[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        out[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
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

int astts_op_attn_gqa(const void* q_f16, const void* k_f16, const void* v_f16, const int32_t* lens, const int32_t* key_start, void* out_f16,
                      int32_t b, int32_t tq, int32_t tk, int32_t pos0, int32_t heads, int32_t kv_heads, int32_t head_dim, int64_t sqb, int64_t sqt,
                      int64_t skb, int64_t skt, int64_t sob, int64_t sot, float scale, astts_stream_t stream) {
    ASTTS_REQUIRE(q_f16 && k_f16 && v_f16 && out_f16, ASTTS_ERR_INVALID, "astts_op_attn_gqa: null pointer");
    ASTTS_REQUIRE(head_dim == GA_D, ASTTS_ERR_UNSUPPORTED, "astts_op_attn_gqa: head_dim=%d (built for 128; 64 is astts_op_attn_mha)", head_dim);
    ASTTS_REQUIRE(b >= 1 && tq >= 1 && tk >= 1 && pos0 >= 0 && pos0 + tq <= tk && heads >= 1 && kv_heads >= 1 && heads % kv_heads == 0, ASTTS_ERR_INVALID,
                  "astts_op_attn_gqa: bad shape b=%d tq=%d tk=%d pos0=%d heads=%d kv_heads=%d", b, tq, tk, pos0, heads, kv_heads);
    ASTTS_REQUIRE(((sqb | sqt | skb | skt | sob | sot) & 7) == 0 && ((uintptr_t)q_f16 & 15) == 0 && ((uintptr_t)k_f16 & 15) == 0 &&
                      ((uintptr_t)v_f16 & 15) == 0 && ((uintptr_t)out_f16 & 15) == 0,
                  ASTTS_ERR_INVALID, "astts_op_attn_gqa: strides must be multiples of 8 halfs and pointers 16-byte aligned");
    GqaArgs a{(const _Float16*)q_f16, (const _Float16*)k_f16, (const _Float16*)v_f16, lens, key_start, (_Float16*)out_f16,
              sqb, sqt, skb, skt, sob, sot, tq, tk, pos0, heads, kv_heads, scale};
    hipLaunchKernelGGL(attn_gqa_mfma, dim3((unsigned)cdiv(tq, 128), heads, b), dim3(256), 0, (hipStream_t)stream, a);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_rope_llama_ex(void* x_f16, const float* cos_tab, const float* sin_tab, const int32_t* shift, int32_t b, int32_t t, int32_t time_major,
                           int32_t heads, int32_t ld, int32_t head_dim, int32_t pos0, astts_stream_t stream) {
    ASTTS_REQUIRE(x_f16 && cos_tab && sin_tab, ASTTS_ERR_INVALID, "astts_op_rope_llama_ex: null pointer");
    ASTTS_REQUIRE(b >= 1 && t >= 1 && heads >= 1 && head_dim >= 2 && (head_dim & 1) == 0 && ld >= heads * head_dim && pos0 >= 0, ASTTS_ERR_INVALID,
                  "astts_op_rope_llama_ex: bad shape b=%d t=%d heads=%d head_dim=%d ld=%d", b, t, heads, head_dim, ld);
    const int64_t total = (int64_t)b * t * heads * (head_dim / 2);
    const unsigned blocks = (unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(rope_llama_ex, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (_Float16*)x_f16, cos_tab, sin_tab, shift, (int64_t)b * t, t, b,
                       time_major, heads, ld, head_dim, pos0);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_argmax_rows(const float* x, int32_t* out, int32_t rows, int32_t n, int64_t ld, astts_stream_t stream) {
    ASTTS_REQUIRE(x && out, ASTTS_ERR_INVALID, "astts_op_argmax_rows: null pointer");
    ASTTS_REQUIRE(rows >= 1 && n >= 1 && ld >= n, ASTTS_ERR_INVALID, "astts_op_argmax_rows: bad shape rows=%d n=%d", rows, n);
    hipLaunchKernelGGL(argmax_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, out, n, ld);
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
