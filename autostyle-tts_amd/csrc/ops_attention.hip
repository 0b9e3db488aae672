// ops_attention.hip -- attention operators of the synthesis path (gfx950), head dim 64.
//
//   attn_relpos         espnet relative-position attention (text encoder, token encoder, LM prefill):
//                       score(i,j) = ((q_i+u).k_j + (q_i+v).p_{i-j}) / sqrt(dh), optional causal mask.
//                       LDS-tiled fp32 VALU kernel (exact fp32; sequences here are <= ~500).
//   attn_relpos_decode  one new query per (batch, head) against the KV cache (LM decode step).
//   attn_mha_flash      plain masked MHA of the flow estimator's transformer blocks: flash-style,
//                       fp16 MFMA 32x32x16 with fp32 softmax/accumulate.  S^T = K Q^T keeps the
//                       query on the lane, so the online softmax is lane-local and P^T feeds the
//                       second MFMA straight from the accumulator registers (no LDS round trip).
//
// Replaces cosyvoice RelPositionMultiHeadedAttention / diffusers Attention inside the reference's
// CosyVoice calls (tts_with_rag.py:195); third-party code, restated from the published architecture.
#include "common.h"

#include <cstdlib>
#include <cstring>

namespace astts {

static constexpr int DH = 64;

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_add_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct RelPosArgs {
    const float* q;    // [B, Tq, ldq]  (+ head*64)
    const void* k;     // [B, Tk, ldk]  fp32, or fp16 when kv_f16 (the KV cache)
    const void* v;     // [B, Tk, ldk]
    const void* pos;   // [2*pos_center+1 rows][H*64]: row (rel + pos_center) holds linear_pos(pe(rel)); fp32 or fp16
    const float* bias_u;
    const float* bias_v;
    const int* lens;   // [B] valid keys (null -> Tk)
    float* out;        // [B, Tq, ldo]
    int b, h, tq, tk, ldq, ldk, ldo, ldp;
    int64_t q_bs, k_bs, o_bs;  // batch strides (elements): batch-major [B,T,*] or time-major [T,B,*] both work
    int q_pos0;        // absolute position of query row 0 (Tk - Tq for cached decode)
    int pos_center;
    int causal;
    float scale;
    int kv_f16, pos_f16;
    int len_all;       // decode kernel: number of valid keys when lens == null (same for every batch row)
    const int* kstart; // [B] first valid key (left-padded rows: keys < kstart are masked) or null
};

template <typename T>
__device__ __forceinline__ float4 ld4(const T* p) {
    if constexpr (sizeof(T) == 2) {
        const half4 h = *reinterpret_cast<const half4*>(p);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}

static constexpr int RP_QB = 16;  // query rows per block
static constexpr int RP_KB = 64;  // keys per tile
static constexpr int RP_LD = 65;  // padded LDS row

template <typename KVT, typename PT>
__global__ __launch_bounds__(256) void attn_relpos(RelPosArgs a) {
    __shared__ float sk[RP_KB * RP_LD];
    __shared__ float sv[RP_KB * DH];
    __shared__ float sp[(RP_KB + RP_QB - 1) * RP_LD];
    __shared__ float squ[RP_QB][DH];
    __shared__ float sqv[RP_QB][DH];
    __shared__ float sprob[4][RP_KB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int i0 = qt * RP_QB;
    const int len = a.lens ? min(a.lens[b], a.tk) : a.tk;
    const int ks0 = a.kstart ? max(a.kstart[b], 0) : 0;
    const float* qb = a.q + (int64_t)b * a.q_bs + head * DH;
    const KVT* kb = reinterpret_cast<const KVT*>(a.k) + (int64_t)b * a.k_bs + head * DH;
    const KVT* vb = reinterpret_cast<const KVT*>(a.v) + (int64_t)b * a.k_bs + head * DH;
    const PT* posb = reinterpret_cast<const PT*>(a.pos);
    // q + u, q + v for the block's query rows
    for (int e = tid; e < RP_QB * DH; e += 256) {
        const int r = e >> 6, d = e & 63;
        const int i = min(i0 + r, a.tq - 1);
        const float qv = qb[(int64_t)i * a.ldq + d];
        squ[r][d] = (qv + a.bias_u[head * DH + d]) * a.scale;
        sqv[r][d] = (qv + a.bias_v[head * DH + d]) * a.scale;
    }
    float m_run[4], l_run[4], o_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m_run[r] = -INFINITY;
        l_run[r] = 0.0f;
        o_run[r] = 0.0f;
    }
    int kmax = len;  // keys needed by this block
    if (a.causal) kmax = min(len, a.q_pos0 + i0 + RP_QB);
    for (int j0 = (ks0 / RP_KB) * RP_KB; j0 < kmax; j0 += RP_KB) {
        __syncthreads();
        for (int e = tid; e < RP_KB * DH; e += 256) {
            const int r = e >> 6, d = e & 63;
            const int j = j0 + r;
            const bool ok = j < len;
            sk[r * RP_LD + d] = ok ? (float)kb[(int64_t)j * a.ldk + d] : 0.0f;
            sv[r * DH + d] = ok ? (float)vb[(int64_t)j * a.ldk + d] : 0.0f;
        }
        // relative positions needed: rel = (q_pos0 + i) - j, i in [i0, i0+QB), j in [j0, j0+KB)
        const int rel_min = a.q_pos0 + i0 - (j0 + RP_KB - 1);
        for (int e = tid; e < (RP_KB + RP_QB - 1) * DH; e += 256) {
            const int r = e >> 6, d = e & 63;
            const int row = rel_min + r + a.pos_center;
            sp[r * RP_LD + d] = (float)posb[(int64_t)row * a.ldp + head * DH + d];
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wid * 4 + rr;
            const int i_abs = a.q_pos0 + i0 + r;
            const int j = j0 + lane;
            const float* kr = &sk[lane * RP_LD];
            const float* pr = &sp[(i_abs - j - rel_min) * RP_LD];
            float s = 0.0f;
#pragma unroll 16
            for (int d = 0; d < DH; ++d) s += squ[r][d] * kr[d] + sqv[r][d] * pr[d];
            const bool valid = j >= ks0 && j < len && (!a.causal || j <= i_abs);
            s = valid ? s : -INFINITY;
            const float m_new = fmaxf(m_run[rr], wave_max_f32(s));
            const float alpha = (m_new == -INFINITY) ? 1.0f : __expf(m_run[rr] - m_new);
            const float p = valid ? __expf(s - m_new) : 0.0f;
            l_run[rr] = l_run[rr] * alpha + wave_add_f32(p);
            m_run[rr] = m_new;
            sprob[wid][lane] = p;
            // LDS write -> read by the same wave: wave-synchronous, make the write visible
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float o = o_run[rr] * alpha;
#pragma unroll 16
            for (int jj = 0; jj < RP_KB; ++jj) o += sprob[wid][jj] * sv[jj * DH + lane];
            o_run[rr] = o;
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = i0 + wid * 4 + rr;
        if (i < a.tq) {
            const float inv = l_run[rr] > 0.0f ? 1.0f / l_run[rr] : 0.0f;
            a.out[(int64_t)b * a.o_bs + (int64_t)i * a.ldo + head * DH + lane] = o_run[rr] * inv;
        }
    }
}

// one (batch, head) per block (512 threads); query = the single new position q_pos0 (== tk - 1 for the LM step).
// 16 lanes share one key: lane sub = lane & 15 owns dims [4 sub, 4 sub + 4), so every wave instruction reads four
// whole K (or V, or position) rows -- fully coalesced.  32 key groups x DK keys each per pass keep up to
// 2*DK 8/16-byte loads in flight per lane: the ~400 cached keys of a step take two passes instead of 25.
static constexpr int DG = 32;  // key groups per block
static constexpr int DK = 8;   // keys per group and pass

template <typename KVT, typename PT>
__global__ __launch_bounds__(512) void attn_relpos_decode(RelPosArgs a) {
    extern __shared__ float sc[];  // [tk] scores, then [DG][64] partial outputs
    __shared__ float redm[8], reds[8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int sub = lane & 15, grp = tid >> 4;
    const int head = blockIdx.x, b = blockIdx.y;
    const int len = a.lens ? min(a.lens[b], a.len_all) : a.len_all;
    const int ks0 = a.kstart ? max(min(a.kstart[b], len - 1), 0) : 0;
    const float* qb = a.q + (int64_t)b * a.q_bs + head * DH;  // tq == 1
    const KVT* kb = reinterpret_cast<const KVT*>(a.k) + (int64_t)b * a.k_bs + head * DH + 4 * sub;
    const KVT* vb = reinterpret_cast<const KVT*>(a.v) + (int64_t)b * a.k_bs + head * DH + 4 * sub;
    const PT* pb = reinterpret_cast<const PT*>(a.pos) + head * DH + 4 * sub;
    float4 qu, qv;
    {
        const float4 x = *reinterpret_cast<const float4*>(qb + 4 * sub);
        const float4 u = *reinterpret_cast<const float4*>(a.bias_u + head * DH + 4 * sub);
        const float4 v = *reinterpret_cast<const float4*>(a.bias_v + head * DH + 4 * sub);
        qu = make_float4((x.x + u.x) * a.scale, (x.y + u.y) * a.scale, (x.z + u.z) * a.scale, (x.w + u.w) * a.scale);
        qv = make_float4((x.x + v.x) * a.scale, (x.y + v.y) * a.scale, (x.z + v.z) * a.scale, (x.w + v.w) * a.scale);
    }
    float mloc = -INFINITY;
    for (int j0 = ks0; j0 < len; j0 += DG * DK) {
        float4 kk[DK], pp[DK];
#pragma unroll
        for (int u = 0; u < DK; ++u) {
            const int j = min(j0 + u * DG + grp, len - 1);  // clamped: always a valid row, masked below
            kk[u] = ld4<KVT>(kb + (int64_t)j * a.ldk);
            pp[u] = ld4<PT>(pb + (int64_t)(a.q_pos0 - j + a.pos_center) * a.ldp);
        }
#pragma unroll
        for (int u = 0; u < DK; ++u) {
            const int j = j0 + u * DG + grp;
            float s = qu.x * kk[u].x + qu.y * kk[u].y + qu.z * kk[u].z + qu.w * kk[u].w + qv.x * pp[u].x + qv.y * pp[u].y +
                      qv.z * pp[u].z + qv.w * pp[u].w;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (j < len) {
                if (sub == 0) sc[j] = s;
                mloc = fmaxf(mloc, s);
            }
        }
    }
    mloc = wave_max_f32(mloc);
    if (lane == 0) redm[wid] = mloc;
    __syncthreads();
    float m = redm[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, redm[w]);
    float sloc = 0.0f;
    for (int j = ks0 + tid; j < len; j += 512) {
        const float p = __expf(sc[j] - m);
        sc[j] = p;
        sloc += p;
    }
    sloc = wave_add_f32(sloc);
    if (lane == 0) reds[wid] = sloc;
    __syncthreads();
    float l = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) l += reds[w];
    // out[d] = sum_j p_j v[j][d]: key group g takes keys j = g (mod DG)
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = ks0 + grp; j0 < len; j0 += DG * DK) {
        float4 vv[DK];
        float pw[DK];
#pragma unroll
        for (int u = 0; u < DK; ++u) {
            const int j = j0 + u * DG;
            const int jc = min(j, len - 1);
            vv[u] = ld4<KVT>(vb + (int64_t)jc * a.ldk);
            pw[u] = j < len ? sc[jc] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < DK; ++u) {
            o.x += pw[u] * vv[u].x; o.y += pw[u] * vv[u].y; o.z += pw[u] * vv[u].z; o.w += pw[u] * vv[u].w;
        }
    }
    float* part = sc + a.tk;
    *reinterpret_cast<float4*>(part + grp * DH + 4 * sub) = o;
    __syncthreads();
    if (tid < DH) {
        float tot = 0.0f;
#pragma unroll
        for (int g2 = 0; g2 < DG; ++g2) tot += part[g2 * DH + tid];
        a.out[(int64_t)b * a.o_bs + head * DH + tid] = l > 0.0f ? tot / l : 0.0f;
    }
}

// attn_relpos_rows: the decode step of a WIDE batch (33 .. 256 rows through one launch: the wide decode engine) over an fp16 KV cache and
// an fp16 position table.  At these sizes the step is the cache's HBM stream (128 rows x 220 keys x 4 KB = 114 MB per layer), and
// attn_relpos_decode's (row, head) workgroups read it in 128-byte pieces half a megabyte apart, in two dependent phases (scores, then
// values).  Here a workgroup owns HG heads of one row: 8 lanes share a (key, head) pair with 8 dims each, so one wave instruction reads
// 16-byte pieces that cover HG x 128 CONTIGUOUS bytes of a key's row (512 B / 1 KB), and K, the position row and V of a key are
// requested together: one pass over the keys with a running (max, sum, output) per (wave, key slot, head) -- the flash-decoding
// recurrence -- merged through LDS in a fixed order at the end.  A row's result does not depend on the other rows of the launch.
template <int HG>
__global__ __launch_bounds__(512, 4) void attn_relpos_rows(RelPosArgs a) {
    constexpr int KPI = 8 / HG;                  // keys per wave instruction (HG = 8: 1, HG = 4: 2)
    constexpr int NSLOT = 8 * KPI;               // key slots of the workgroup: slot s takes keys ks0 + s, ks0 + s + NSLOT, ...
    constexpr int DKR = 4;                       // keys per slot and pass (3 x 16-byte loads each in flight; two workgroups per CU)
    __shared__ float s_m[NSLOT][HG], s_l[NSLOT][HG];
    __shared__ __attribute__((aligned(16))) float s_o[NSLOT][HG][DH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int sub = lane & 7, hq = (lane >> 3) % HG, kq = lane / (8 * HG);
    const int slot = wid * KPI + kq;
    const int head = blockIdx.x * HG + hq, b = blockIdx.y;
    const int len = a.lens ? min(a.lens[b], a.len_all) : a.len_all;
    const int ks0 = a.kstart ? max(min(a.kstart[b], len - 1), 0) : 0;
    const _Float16* kb = reinterpret_cast<const _Float16*>(a.k) + (int64_t)b * a.k_bs + head * DH + 8 * sub;
    const _Float16* vb = reinterpret_cast<const _Float16*>(a.v) + (int64_t)b * a.k_bs + head * DH + 8 * sub;
    const _Float16* pb = reinterpret_cast<const _Float16*>(a.pos) + head * DH + 8 * sub;
    float qu[8], qv[8];
    {
        const float* qb = a.q + (int64_t)b * a.q_bs + head * DH + 8 * sub;
        const float* up = a.bias_u + head * DH + 8 * sub;
        const float* vp = a.bias_v + head * DH + 8 * sub;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const float4 x = *reinterpret_cast<const float4*>(qb + 4 * h2);
            const float4 u = *reinterpret_cast<const float4*>(up + 4 * h2);
            const float4 v = *reinterpret_cast<const float4*>(vp + 4 * h2);
            qu[4 * h2] = (x.x + u.x) * a.scale; qu[4 * h2 + 1] = (x.y + u.y) * a.scale; qu[4 * h2 + 2] = (x.z + u.z) * a.scale; qu[4 * h2 + 3] = (x.w + u.w) * a.scale;
            qv[4 * h2] = (x.x + v.x) * a.scale; qv[4 * h2 + 1] = (x.y + v.y) * a.scale; qv[4 * h2 + 2] = (x.z + v.z) * a.scale; qv[4 * h2 + 3] = (x.w + v.w) * a.scale;
        }
    }
    float m = -INFINITY, l = 0.0f, o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = 0.0f;
    for (int j0 = ks0 + slot; j0 < len; j0 += NSLOT * DKR) {
        half8 kk[DKR], pp[DKR], vv[DKR];
#pragma unroll
        for (int u = 0; u < DKR; ++u) {
            const int j = min(j0 + u * NSLOT, len - 1);            // clamped: always a valid row, masked below
            kk[u] = *reinterpret_cast<const half8*>(kb + (int64_t)j * a.ldk);
            pp[u] = *reinterpret_cast<const half8*>(pb + (int64_t)(a.q_pos0 - j + a.pos_center) * a.ldp);
            vv[u] = *reinterpret_cast<const half8*>(vb + (int64_t)j * a.ldk);
        }
        __builtin_amdgcn_sched_barrier(0);      // the pass's 12 loads are ONE batch: no use may be scheduled between them
        float sc[DKR];
        float mp = -INFINITY;
#pragma unroll
        for (int u = 0; u < DKR; ++u) {
            float t = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) t += qu[d] * (float)kk[u][d] + qv[d] * (float)pp[u][d];
            t += __shfl_xor(t, 4, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 1, 64);
            sc[u] = j0 + u * NSLOT < len ? t : -INFINITY;
            mp = fmaxf(mp, sc[u]);
        }
        const float mn = fmaxf(m, mp);                             // finite: the pass's first key exists
        const float resc = m == -INFINITY ? 0.0f : __expf(m - mn);
        l *= resc;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] *= resc;
#pragma unroll
        for (int u = 0; u < DKR; ++u) {
            const float pw = sc[u] == -INFINITY ? 0.0f : __expf(sc[u] - mn);
            l += pw;
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] += pw * (float)vv[u][d];
        }
        m = mn;
    }
    if (sub == 0) {
        s_m[slot][hq] = m;
        s_l[slot][hq] = l;
    }
    *reinterpret_cast<float4*>(&s_o[slot][hq][8 * sub]) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(&s_o[slot][hq][8 * sub + 4]) = make_float4(o[4], o[5], o[6], o[7]);
    __syncthreads();
    if (tid < HG * DH) {
        const int h2 = tid >> 6, d = tid & 63;
        float mm = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < NSLOT; ++s2) mm = fmaxf(mm, s_m[s2][h2]);
        float lt = 0.0f, ot = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < NSLOT; ++s2) {
            const float w = s_m[s2][h2] == -INFINITY ? 0.0f : __expf(s_m[s2][h2] - mm);
            lt += s_l[s2][h2] * w;
            ot += s_o[s2][h2][d] * w;
        }
        a.out[(int64_t)b * a.o_bs + (blockIdx.x * HG + h2) * DH + d] = lt > 0.0f ? ot / lt : 0.0f;
    }
}

// ------------------------------------------------------------------ MFMA flash attention
struct MhaArgs {
    const void* q;   // fp32 or fp16 (IN16), [B, T, ldq]
    const void* k;
    const void* v;
    const int* lens;  // [B] valid keys
    void* out;        // fp32 or fp16 (OUT16)
    int b, h, t, ldq, ldk, ldo;
    float scale;
};

static constexpr int FA_KT = 64;  // keys per staged tile (two 32-key MFMA sub-tiles)
static constexpr int FA_KS = 72;  // halfs per K row in LDS (64 dims + 8): conflict-free b128 reads
static constexpr int FA_VS = 68;  // halfs per V^T row in LDS: 34 dwords, so the 32 dims (rows) a half-wave reads with ds_read_b64
                                  // start on 32 different even banks (34 c mod 64 = 2 (17 c mod 32)): conflict-free; the half2
                                  // scatter of the staging writes is 2-way (free on ds_write_b32).  72 + a key XOR swizzle made the
                                  // writes conflict-free and left the reads 2-way: 43 % of the kernel's LDS cycles (r02 PMC pass)

template <bool IN16>
__device__ __forceinline__ void load8(const void* base, int64_t off, float* dst) {
    if constexpr (IN16) {
        const half8 hv = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(base) + off);
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = (float)hv[i];
    } else {
        const float* p = reinterpret_cast<const float*>(base) + off;
        const float4 a0 = *reinterpret_cast<const float4*>(p);
        const float4 a1 = *reinterpret_cast<const float4*>(p + 4);
        dst[0] = a0.x; dst[1] = a0.y; dst[2] = a0.z; dst[3] = a0.w; dst[4] = a1.x; dst[5] = a1.y; dst[6] = a1.z; dst[7] = a1.w;
    }
}

template <bool IN16>
__device__ __forceinline__ half8 load8h(const void* base, int64_t off) {
    if constexpr (IN16) {
        return *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(base) + off);
    } else {
        float t[8];
        load8<false>(base, off, t);
        half8 h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (_Float16)t[i];
        return h;
    }
}

// Block = 4 waves x 32 queries.  Per 64-key tile: barrier, registers -> LDS, barrier, issue the NEXT tile's global
// loads, then two 32-key sub-tiles of {S^T = K Q^T (4 MFMA), lane-local online softmax, O^T += V^T P^T (4 MFMA)}.
template <bool IN16, bool OUT16>
__global__ __launch_bounds__(256) void attn_mha_flash(MhaArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 ks[FA_KT * FA_KS];
    __shared__ __attribute__((aligned(16))) _Float16 vt[DH * FA_VS];
    __shared__ float so[4][32][DH + 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wid * 32;
    const int len = a.lens ? min(a.lens[b], a.t) : a.t;
    const int64_t qb = (int64_t)b * a.t * a.ldq + head * DH;
    const int64_t kb = (int64_t)b * a.t * a.ldk + head * DH;

    // Q fragments (B operand of S^T = K Q^T): lane (c, hh) holds Q[q0+c][16s + 8hh + j] * scale * log2(e)
    half8 qf[4];
    {
        const int qi = min(q0 + c, a.t - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float x[8];
            load8<IN16>(a.q, qb + (int64_t)qi * a.ldq + 16 * s + 8 * hh, x);
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[s][i] = (_Float16)(x[i] * (a.scale * 1.44269504088896341f));
        }
    }
    float16v ot[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        ot[0][e] = 0.0f;
        ot[1][e] = 0.0f;
    }
    float m_run = -INFINITY, l_run = 0.0f;
    // staging coordinates.  K: key row skey (0..63), 16 dims from sd0.  V: key PAIR vkp (keys 2 vkp, 2 vkp + 1), 8 dims from
    // vd0 -- the transposed image V^T[dim][key] is written as whole dwords (two keys of one dim; a 2-byte scatter of one key
    // per thread was 8-way conflicted).  Row stride FA_VS = 68 halfs: see its definition.
    // Round 5: the staging WRITES were the kernel's bank conflicts (24.6 % of its LDS cycles, profiles/r04_pmc_flow_lds.txt; the fragment
    // reads are conflict-free by the strides above).  K: a thread owned 16 consecutive dims of one key -- the 16 lanes of a ds_write_b128
    // cycle then wrote rows 0..3, and 36 r mod 64 puts rows 0 and 2 on the same banks.  Now a thread writes ONE 16-byte chunk (tid & 7) of
    // rows kr and kr + 32, and consecutive 8-lane groups take rows 0, 8, 1, 9, ...: 36 * 8 = 32 (mod 64), so the two rows of a cycle
    // cover banks 0..31 and 32..63.  V^T: a thread owned dims 8 j .. 8 j + 7 of a key pair; the eight j of a wave landed on banks
    // 16 j (mod 64): j and j + 4 collided.  Now it owns dims 4 j .. 4 j + 3 and 32 + 4 j .. 32 + 4 j + 3: 34 * 4 j = 8 j (mod 64), the
    // key pairs vkp = 0..7 of a wave fill the eight banks between -- all 64 banks once per ds_write_b32.
    const int kc8 = (tid & 7) * 8, kg = tid >> 3;
    const int kr = (kg & 1) * 8 + ((kg >> 1) & 7) + (kg >> 4) * 16;       // 0..31, lane groups alternate between rows r and r + 8
    const int vkp = tid >> 3, vd0 = (tid & 7) * 4;
    // Two register sets: the loads of tile j + 2 are issued while tile j is computed, so a tile's K / V rows have two tiles of
    // compute (and four barriers) to arrive -- one tile of compute (~0.5 us at T = 344) did not cover an L2 round trip, and
    // the kernel ran at the pace of its global loads (18 us for 6 tiles).
    half8 rkA[2], rvA[2], rkB[2], rvB[2];
    auto load4h = [&](const void* base, int64_t off, half8& dst, int at) {      // 4 consecutive values -> dst[at .. at + 3]
        if constexpr (IN16) {
            const half4 h = *reinterpret_cast<const half4*>(reinterpret_cast<const _Float16*>(base) + off);
            dst[at] = h[0]; dst[at + 1] = h[1]; dst[at + 2] = h[2]; dst[at + 3] = h[3];
        } else {
            const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
            dst[at] = (_Float16)f.x; dst[at + 1] = (_Float16)f.y; dst[at + 2] = (_Float16)f.z; dst[at + 3] = (_Float16)f.w;
        }
    };
    auto prefetch = [&](int j0, half8 (&rk)[2], half8 (&rv)[2]) {
        const int ja = min(j0 + kr, len - 1), jb = min(j0 + kr + 32, len - 1);         // clamped; keys >= len are masked in the scores
        rk[0] = load8h<IN16>(a.k, kb + (int64_t)ja * a.ldk + kc8);
        rk[1] = load8h<IN16>(a.k, kb + (int64_t)jb * a.ldk + kc8);
        const int jv0 = min(j0 + 2 * vkp, len - 1), jv1 = min(j0 + 2 * vkp + 1, len - 1);
        load4h(a.v, kb + (int64_t)jv0 * a.ldk + vd0, rv[0], 0);
        load4h(a.v, kb + (int64_t)jv0 * a.ldk + 32 + vd0, rv[0], 4);
        load4h(a.v, kb + (int64_t)jv1 * a.ldk + vd0, rv[1], 0);
        load4h(a.v, kb + (int64_t)jv1 * a.ldk + 32 + vd0, rv[1], 4);
    };
    auto stage = [&](const half8 (&rk)[2], const half8 (&rv)[2]) {
        *reinterpret_cast<half8*>(&ks[kr * FA_KS + kc8]) = rk[0];
        *reinterpret_cast<half8*>(&ks[(kr + 32) * FA_KS + kc8]) = rk[1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            half2v pr;
            pr[0] = rv[0][i];
            pr[1] = rv[1][i];
            *reinterpret_cast<half2v*>(&vt[((i >> 2) * 32 + vd0 + (i & 3)) * FA_VS + 2 * vkp]) = pr;
        }
    };
    auto compute_tile = [&](int j0) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int jb = j0 + sub * 32;
            if (jb >= len) break;   // block-uniform
            // S^T[key][query] = sum_d K[key][d] Q[query][d]
            float16v st;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half8 kf = *reinterpret_cast<const half8*>(&ks[(sub * 32 + c) * FA_KS + 16 * s + 8 * hh]);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st, 0, 0, 0);
            }
            // scores arrive in the log2 domain (Q carries scale * log2 e).  Keys past len only exist in the last
            // sub-tile (block-uniform test); exp2(-inf - m) is 0 and every query sees at least one valid key here.
            if (jb + 32 > len) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = jb + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    st[e] = key < len ? st[e] : -INFINITY;
                }
            }
            float mloc = st[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mloc = fmaxf(mloc, st[e]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            // rescale only when some query's running maximum moves (wave-uniform branch): after the first
            // few tiles it rarely does, and the 32 accumulator multiplies are a third of the softmax VALU work
            if (__builtin_amdgcn_ballot_w64(mloc > m_run) != 0) {
                const float m_new = fmaxf(m_run, mloc);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // first tile: exp2(-inf) = 0
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    ot[0][e] *= alpha;
                    ot[1][e] *= alpha;
                }
            }
            half8 pf[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(st[e] - m_run);
                l_run += p;                          // per lane half; the two halves are added once at the end
                pf[e >> 3][e & 7] = (_Float16)p;
            }
            // O^T[d][query] += V^T[d][key] P^T[key][query]; key order inside a k-step follows the
            // accumulator layout: element j of lane half hh is key 16s + 8(j>>2) + 4hh + (j&3)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const _Float16* vrow = &vt[(dt * 32 + c) * FA_VS];
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
    };
    if (len > 0) prefetch(0, rkA, rvA);
    if (len > FA_KT) prefetch(FA_KT, rkB, rvB);
    for (int j0 = 0; j0 < len; j0 += 2 * FA_KT) {
        __syncthreads();
        stage(rkA, rvA);
        __syncthreads();
        if (j0 + 2 * FA_KT < len) prefetch(j0 + 2 * FA_KT, rkA, rvA);
        compute_tile(j0);
        if (j0 + FA_KT >= len) break;                  // block-uniform
        __syncthreads();
        stage(rkB, rvB);
        __syncthreads();
        if (j0 + 3 * FA_KT < len) prefetch(j0 + 3 * FA_KT, rkB, rvB);
        compute_tile(j0 + FA_KT);
    }
    // O^T (dims in registers, query on the lane) -> LDS transpose -> coalesced rows
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int d = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            so[wid][c][d] = ot[dt][e] * inv;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 32; ++r) {
        const int qi = q0 + r;
        if (qi < a.t) {
            const int64_t o = ((int64_t)b * a.t + qi) * a.ldo + head * DH + lane;
            if constexpr (OUT16)
                reinterpret_cast<_Float16*>(a.out)[o] = (_Float16)so[wid][r][lane];
            else
                reinterpret_cast<float*>(a.out)[o] = so[wid][r][lane];
        }
    }
}

// ------------------------------------------------------------------------------------------
// attn_relpos_mfma: espnet relative-position attention for full sequences (encoders, LM prefill) on the matrix cores.
//   score[i][j] = ((q_i + u) . k_j + (q_i + v) . p_{a_i - j}) * scale,   a_i = q_pos0 + i (absolute position of query i)
// Block = 4 waves x 32 queries, 64-key tiles staged in LDS exactly as attn_mha_flash (S^T = K (Q+u)^T: query on the
// lane, online softmax lane-local, P^T feeds the PV MFMA from the accumulator registers).  The position term: for a
// 32-key sub-tile a wave needs rel = a_i - j over 63 consecutive values; G^T[rel][query] = P[rel] (Q+v)^T is two 32-row
// MFMA blocks (the window slides by 32 per sub-tile, so one block is carried over), written to a per-wave LDS tile and
// read back skewed -- lane c takes G[c - key + 31][c], address 33 c + const: conflict-free -- and added to S^T.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ half8 ld8h(const T* p) {
    if constexpr (sizeof(T) == 2) {
        return *reinterpret_cast<const half8*>(p);
    } else {
        const float4 a0 = *reinterpret_cast<const float4*>(p);
        const float4 a1 = *reinterpret_cast<const float4*>(p + 4);
        half8 h;
        h[0] = (_Float16)a0.x; h[1] = (_Float16)a0.y; h[2] = (_Float16)a0.z; h[3] = (_Float16)a0.w;
        h[4] = (_Float16)a1.x; h[5] = (_Float16)a1.y; h[6] = (_Float16)a1.z; h[7] = (_Float16)a1.w;
        return h;
    }
}

template <typename KVT, typename PT>
__global__ __launch_bounds__(256) void attn_relpos_mfma(RelPosArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 ks[FA_KT * FA_KS];
    __shared__ __attribute__((aligned(16))) _Float16 vt[DH * FA_VS];
    __shared__ float gbuf[4][64 * 33];        // per wave: G^T rows (rel) x 32 queries (+1 pad); reused for the output transpose
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0b = blockIdx.x * 128, q0 = q0b + wid * 32;
    const int len = a.lens ? min(a.lens[b], a.tk) : a.tk;
    const int ks0 = a.kstart ? max(a.kstart[b], 0) : 0;
    const float* qb = a.q + (int64_t)b * a.q_bs + head * DH;
    const KVT* kb = reinterpret_cast<const KVT*>(a.k) + (int64_t)b * a.k_bs + head * DH;
    const KVT* vb = reinterpret_cast<const KVT*>(a.v) + (int64_t)b * a.k_bs + head * DH;
    const PT* posb = reinterpret_cast<const PT*>(a.pos) + head * DH;
    const int prow_max = 2 * a.pos_center;
    const float sc = a.scale * 1.44269504088896341f;      // scores in the log2 domain

    // (Q + u), (Q + v) fragments: lane (c, hh) holds dims 16 s + 8 hh + j of query q0 + c
    half8 quf[4], qvf[4];
    {
        const int qi = min(q0 + c, a.tq - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int d0 = 16 * s + 8 * hh;
            const float* qp = qb + (int64_t)qi * a.ldq + d0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float qv = qp[i];
                quf[s][i] = (_Float16)((qv + a.bias_u[head * DH + d0 + i]) * sc);
                qvf[s][i] = (_Float16)((qv + a.bias_v[head * DH + d0 + i]) * sc);
            }
        }
    }
    const int a0w = a.q_pos0 + q0;            // absolute position of this wave's query 0
    float16v ot[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        ot[0][e] = 0.0f;
        ot[1][e] = 0.0f;
    }
    float m_run = -INFINITY, l_run = 0.0f;
    int kmax = len;                           // keys this block needs (block-uniform)
    if (a.causal) kmax = min(len, a.q_pos0 + q0b + 128);
    const int jstart = (ks0 / FA_KT) * FA_KT;

    const int skey = tid >> 2, sd0 = (tid & 3) * 16;   // K: one key row, 16 dims; V: a key pair, 8 dims (see attn_mha_flash)
    const int vkp = tid >> 3, vd0 = (tid & 7) * 8;
    half8 rk[2], rv[2];
    auto prefetch = [&](int j0) {
        const int j = min(j0 + skey, max(len - 1, 0));
        const int64_t off = (int64_t)j * a.ldk + sd0;
        rk[0] = ld8h<KVT>(kb + off);
        rk[1] = ld8h<KVT>(kb + off + 8);
        const int jv0 = min(j0 + 2 * vkp, max(len - 1, 0)), jv1 = min(j0 + 2 * vkp + 1, max(len - 1, 0));
        rv[0] = ld8h<KVT>(vb + (int64_t)jv0 * a.ldk + vd0);
        rv[1] = ld8h<KVT>(vb + (int64_t)jv1 * a.ldk + vd0);
    };
    // position rows of one 32-row block: table row (rel + center) for rel = rlo + c, clamped into the table
    auto load_p = [&](int rlo, half8 (&pf)[4]) {
        int row = rlo + c + a.pos_center;
        row = row < 0 ? 0 : (row > prow_max ? prow_max : row);
        const PT* pr = posb + (int64_t)row * a.ldp + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) pf[s] = ld8h<PT>(pr + 16 * s);
    };
    auto g_block = [&](const half8 (&pf)[4]) {
        float16v g;
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) g = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[s], qvf[s], g, 0, 0, 0);
        return g;
    };
    float* gw = gbuf[wid];
    if (jstart < kmax) prefetch(jstart);
    // rel window of sub-tile jb: [a0w - jb - 31, a0w - jb + 32); hi block of the first sub-tile, then one new lo block each
    half8 pcur[4], pnext[4];
    float16v g_hi;
    {
        load_p(a0w - jstart - 31 + 32, pcur);
        g_hi = g_block(pcur);
        load_p(a0w - jstart - 31, pcur);
    }
    for (int j0 = jstart; j0 < kmax; j0 += FA_KT) {
        __syncthreads();
        *reinterpret_cast<half8*>(&ks[skey * FA_KS + sd0]) = rk[0];
        *reinterpret_cast<half8*>(&ks[skey * FA_KS + sd0 + 8]) = rk[1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            half2v pr;
            pr[0] = rv[0][i];
            pr[1] = rv[1][i];
            *reinterpret_cast<half2v*>(&vt[(vd0 + i) * FA_VS + 2 * vkp]) = pr;
        }
        __syncthreads();
        if (j0 + FA_KT < kmax) prefetch(j0 + FA_KT);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int jb = j0 + sub * 32;
            // every wave runs both sub-tiles of a staged tile (the carried G block must slide in step); pcur holds the
            // lo block's position rows of this sub-tile, the next sub-tile's are requested now
            load_p(a0w - (jb + 32) - 31, pnext);
            const float16v g_lo = g_block(pcur);
            float16v st;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half8 kf = *reinterpret_cast<const half8*>(&ks[(sub * 32 + c) * FA_KS + 16 * s + 8 * hh]);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, quf[s], st, 0, 0, 0);
            }
            // G^T -> LDS (row = rel - rmin, col = query), skewed read: key row kk of query c needs rel index c - kk + 31
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * hh;
                gw[rr * 33 + c] = g_lo[e];
                gw[(32 + rr) * 33 + c] = g_hi[e];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int ai = a0w + c;
            float mloc = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kk = (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int j = jb + kk;
                const float sv = st[e] + gw[(c - kk + 31) * 33 + c];
                const bool valid = j >= ks0 && j < len && (!a.causal || j <= ai);
                st[e] = valid ? sv : -INFINITY;
                mloc = fmaxf(mloc, st[e]);
            }
            __builtin_amdgcn_wave_barrier();
            g_hi = g_lo;
#pragma unroll
            for (int s = 0; s < 4; ++s) pcur[s] = pnext[s];
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            if (__builtin_amdgcn_ballot_w64(mloc > m_run) != 0) {
                const float m_new = fmaxf(m_run, mloc);
                const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    ot[0][e] *= alpha;
                    ot[1][e] *= alpha;
                }
            }
            half8 pf[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = (m_run == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(st[e] - m_run);   // a query may have no valid key yet
                l_run += p;
                pf[e >> 3][e & 7] = (_Float16)p;
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const _Float16* vrow = &vt[(dt * 32 + c) * FA_VS];
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
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
    // O^T (dims in registers, query on the lane) -> LDS transpose (the wave's G tile is free now) -> coalesced rows
    float* so = gw;                            // [32 queries][65]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int d = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            so[c * 65 + d] = ot[dt][e] * inv;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 32; ++r) {
        const int qi = q0 + r;
        if (qi < a.tq) a.out[(int64_t)b * a.o_bs + (int64_t)qi * a.ldo + head * DH + lane] = so[r * 65 + lane];
    }
}

}  // namespace astts

using namespace astts;

extern "C" {

int astts_op_attn_relpos_ex(const float* q, const void* k, const void* v, int32_t kv_f16, const void* pos, int32_t pos_f16,
                            const float* bias_u, const float* bias_v, const int32_t* lens, const int32_t* key_start, float* out,
                            int32_t b, int32_t h, int32_t tq, int32_t tk, int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldp,
                            int64_t q_bs, int64_t k_bs, int64_t o_bs, int32_t q_pos0, int32_t pos_center, int32_t causal,
                            float scale, astts_stream_t stream) {
    ASTTS_REQUIRE(q && k && v && pos && bias_u && bias_v && out, ASTTS_ERR_INVALID, "astts_op_attn_relpos: null pointer");
    ASTTS_REQUIRE(b >= 1 && h >= 1 && tq >= 1 && tk >= 1, ASTTS_ERR_INVALID, "astts_op_attn_relpos: bad shape");
    ASTTS_REQUIRE(q_pos0 + tq - 1 <= pos_center && tk - 1 <= pos_center + q_pos0, ASTTS_ERR_INVALID,
                  "astts_op_attn_relpos: position table too small (center %d, q_pos0 %d, tq %d, tk %d)", pos_center, q_pos0, tq, tk);
    RelPosArgs a{q, k, v, pos, bias_u, bias_v, lens, out, b, h, tq, tk, ldq, ldk, ldo, ldp, q_bs, k_bs, o_bs, q_pos0, pos_center,
                 causal, scale, kv_f16 ? 1 : 0, pos_f16 ? 1 : 0, tk, key_start};
    hipStream_t st = (hipStream_t)stream;
    const int variant = (kv_f16 ? 2 : 0) | (pos_f16 ? 1 : 0);
    if (tq == 1) {
        a.tk = (tk + 3) & ~3;  // keeps the partial-output area 16-byte aligned (keys are bounded by lens)
        const size_t lds = ((size_t)a.tk + DG * DH) * sizeof(float);
        ASTTS_REQUIRE(lds <= 60 * 1024, ASTTS_ERR_INVALID, "astts_op_attn_relpos: tk=%d too long for the decode kernel", tk);
        const int esz = kv_f16 ? 2 : 4;
        const bool prof = prof_begin(ASTTS_PROF_ATTN_DECODE, st, (double)b * h * tk * DH * esz * 2.0);
        // wide batches over an fp16 cache and position table: contiguous 16-byte pieces, one pass over the keys (attn_relpos_rows);
        // ASTTS_ATTN_DECODE=v1 keeps the (row, head) kernel
        static const bool rows_off = getenv("ASTTS_ATTN_DECODE") && !strcmp(getenv("ASTTS_ATTN_DECODE"), "v1");
        const bool al16 = ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)pos & 15) == 0 && ((uintptr_t)q & 15) == 0 &&
                          (ldk & 7) == 0 && (ldp & 7) == 0 && (k_bs & 7) == 0 && (q_bs & 3) == 0;
        if (b > 32 && variant == 3 && (h % 4) == 0 && al16 && !rows_off) {
            // (always four heads per workgroup: the partition of a row's keys, hence its sums, must not depend on the batch size)
            static const int hg_env = getenv("ASTTS_ATTN_ROWS_HG") ? atoi(getenv("ASTTS_ATTN_ROWS_HG")) : 4;
            if (hg_env == 8 && (h % 8) == 0) hipLaunchKernelGGL((attn_relpos_rows<8>), dim3(h / 8, b), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((attn_relpos_rows<4>), dim3(h / 4, b), dim3(512), 0, st, a);
            if (prof) prof_end(ASTTS_PROF_ATTN_DECODE, st);
            ASTTS_CHECK_LAUNCH();
            return ASTTS_OK;
        }
        const dim3 grid(h, b);
        switch (variant) {
            case 0: hipLaunchKernelGGL((attn_relpos_decode<float, float>), grid, dim3(512), lds, st, a); break;
            case 1: hipLaunchKernelGGL((attn_relpos_decode<float, _Float16>), grid, dim3(512), lds, st, a); break;
            case 2: hipLaunchKernelGGL((attn_relpos_decode<_Float16, float>), grid, dim3(512), lds, st, a); break;
            default: hipLaunchKernelGGL((attn_relpos_decode<_Float16, _Float16>), grid, dim3(512), lds, st, a); break;
        }
        if (prof) prof_end(ASTTS_PROF_ATTN_DECODE, st);
    } else {
        // the tile loader reads rel in [q_pos0+i0-(j0+63), q_pos0+i0+15-j0]; keep that inside the table
        ASTTS_REQUIRE(pos_center >= tk + RP_KB + RP_QB && pos_center >= q_pos0 + tq + RP_QB, ASTTS_ERR_INVALID,
                      "astts_op_attn_relpos: pos_center %d must exceed tk/tq by the tile margin", pos_center);
        static const bool valu_env = getenv("ASTTS_RELPOS_VALU") != nullptr;
        const bool al_ok = ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)pos & 15) == 0 &&
                           (ldk & 7) == 0 && (ldp & 7) == 0 && (k_bs & 7) == 0;
        if (!valu_env && al_ok) {
            const dim3 gridm((tq + 127) / 128, h, b);
            switch (variant) {
                case 0: hipLaunchKernelGGL((attn_relpos_mfma<float, float>), gridm, dim3(256), 0, st, a); break;
                case 1: hipLaunchKernelGGL((attn_relpos_mfma<float, _Float16>), gridm, dim3(256), 0, st, a); break;
                case 2: hipLaunchKernelGGL((attn_relpos_mfma<_Float16, float>), gridm, dim3(256), 0, st, a); break;
                default: hipLaunchKernelGGL((attn_relpos_mfma<_Float16, _Float16>), gridm, dim3(256), 0, st, a); break;
            }
            ASTTS_CHECK_LAUNCH();
            return ASTTS_OK;
        }
        const dim3 grid((tq + RP_QB - 1) / RP_QB, h, b);
        switch (variant) {
            case 0: hipLaunchKernelGGL((attn_relpos<float, float>), grid, dim3(256), 0, st, a); break;
            case 1: hipLaunchKernelGGL((attn_relpos<float, _Float16>), grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL((attn_relpos<_Float16, float>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((attn_relpos<_Float16, _Float16>), grid, dim3(256), 0, st, a); break;
        }
    }
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_attn_relpos(const float* q, const float* k, const float* v, const float* pos, const float* bias_u,
                         const float* bias_v, const int32_t* lens, float* out, int32_t b, int32_t h, int32_t tq,
                         int32_t tk, int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldp, int64_t q_bs, int64_t k_bs,
                         int64_t o_bs, int32_t q_pos0, int32_t pos_center, int32_t causal, float scale,
                         astts_stream_t stream) {
    return astts_op_attn_relpos_ex(q, k, v, 0, pos, 0, bias_u, bias_v, lens, nullptr, out, b, h, tq, tk, ldq, ldk, ldo, ldp, q_bs, k_bs,
                                   o_bs, q_pos0, pos_center, causal, scale, stream);
}

int astts_op_attn_mha_ex(const void* q, const void* k, const void* v, int32_t in_f16, const int32_t* lens, void* out,
                         int32_t out_f16, int32_t b, int32_t h, int32_t t, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                         astts_stream_t stream) {
    ASTTS_REQUIRE(q && k && v && out, ASTTS_ERR_INVALID, "astts_op_attn_mha: null pointer");
    ASTTS_REQUIRE(b >= 1 && h >= 1 && t >= 1, ASTTS_ERR_INVALID, "astts_op_attn_mha: bad shape");
    const int al = in_f16 ? 7 : 3;
    ASTTS_REQUIRE((ldq & al) == 0 && (ldk & al) == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 &&
                      ((uintptr_t)v & 15) == 0,
                  ASTTS_ERR_INVALID, "astts_op_attn_mha: q/k/v must be 16-byte aligned with 16-byte row strides");
    MhaArgs a{q, k, v, lens, out, b, h, t, ldq, ldk, ldo, scale};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((t + 127) / 128, h, b);
    const bool prof = prof_begin(ASTTS_PROF_ATTN_FLASH, st, 4.0 * (double)b * h * (double)t * t * DH);
    if (in_f16 && out_f16)
        hipLaunchKernelGGL((attn_mha_flash<true, true>), grid, dim3(256), 0, st, a);
    else if (in_f16)
        hipLaunchKernelGGL((attn_mha_flash<true, false>), grid, dim3(256), 0, st, a);
    else if (out_f16)
        hipLaunchKernelGGL((attn_mha_flash<false, true>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((attn_mha_flash<false, false>), grid, dim3(256), 0, st, a);
    if (prof) prof_end(ASTTS_PROF_ATTN_FLASH, st);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_attn_mha(const float* q, const float* k, const float* v, const int32_t* lens, float* out, int32_t b,
                      int32_t h, int32_t t, int32_t ldq, int32_t ldk, int32_t ldo, float scale, astts_stream_t stream) {
    return astts_op_attn_mha_ex(q, k, v, 0, lens, out, 0, b, h, t, ldq, ldk, ldo, scale, stream);
}

}  // extern "C"
