// ops_gemm.hip -- fp16-MFMA implicit GEMM with fp32 activations / fp32 accumulate (gfx950).
//
// One kernel family serves every dense contraction of the synthesis path (the arithmetic the
// reference runs inside cosyvoice: nn.Linear, nn.Conv1d incl. dilation/stride, nn.ConvTranspose1d
// after phase decomposition):
//
//   out[m, n] = epilogue( sum_{tap, c} X[src_row(m, tap), c] * W[n, tap*cin_pad + c] )
//   m = b*t_out + t,   src_row = b*t_in + t*stride + tap*dil - pad   (zero outside [0, t_in))
//
// X: fp32 [B*t_in, lda] channels-last, W: fp16 [n_pad, taps*cin_pad] (K contiguous, zero padded:
// cin_pad multiple of 64, n_pad multiple of 128), out: fp32 [B*t_out, ldc].
// epilogue: (+bias[n]) -> activation -> *alpha -> *row_scale[m] -> +residual[m, n]
//
// gemm_tile:     block tile (WM*TM*32) x (WN*TN*32) x 32, 4 waves, v_mfma_f32_32x32x16_f16, fp32->fp16
//                conversion while staging through LDS (80-byte padded rows: conflict-free b128 reads),
//                register prefetch of the next K tile under the MFMAs of the current one.  The tile
//                shape is picked per call so that the grid covers the 256 CUs.
// gemm_skinny16: M <= 32 (LM decode steps, conditioning MLPs): weight-bandwidth bound.  One block =
//                16 output columns, 8 waves that split K line by line, weights streamed straight to
//                VGPRs (v_mfma_f32_16x16x32_f16), LDS reduction.  Optional fusions that remove whole
//                launches from the decode step: LayerNorm of the input rows in the prologue, row
//                gather (embedding lookup), and a second output for the columns >= n_split (the K|V
//                half of a fused QKV projection lands directly in the KV cache).
#include "common.h"

#include <cstdlib>
#include <mutex>

namespace astts {

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_GELU = 3, ACT_MISH = 4, ACT_ELU = 5, ACT_TANH = 6, ACT_LEAKY = 7 };

__device__ __forceinline__ float apply_act(float x, int act, float slope) {
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_MISH: {   // n / (n + 2) form of tanh(softplus(x)), as ops_norm_elem.hip
            const float e = __expf(fminf(x, 20.0f));
            const float n = e * (e + 2.0f);
            return x > 20.0f ? x : x * n * __frcp_rn(n + 2.0f);
        }
        case ACT_ELU: return x > 0.0f ? x : (__expf(x) - 1.0f);
        case ACT_TANH: return tanhf(x);
        case ACT_LEAKY: return x > 0.0f ? x : x * slope;
        default: return x;
    }
}

struct GemmArgs {
    const float* x;
    const _Float16* w;
    const float* bias;       // [n] or null
    const float* residual;   // [m, ldr] or null
    const float* row_scale;  // [m] or null
    float* out;
    int64_t m;               // B * t_out
    int n, cin, cin_pad, taps;
    int lda, ldc, ldr;
    int t_in, t_out, stride, dil, pad;
    int act;
    float alpha, slope;
    // gemm_skinny16 extras
    const int* gather;       // [m] row indices into x (embedding lookup) or null
    const float* ln_gamma;   // LayerNorm over the cin inputs of every row, applied while loading (or null)
    const float* ln_beta;
    float ln_eps;
    float* out2;             // columns >= n_split go to out2[m*ldc2 + n - n_split] (or null)
    int ldc2, n_split;
    int x_f16, out_f16;      // activations in / out as fp16 (ld* are then in halfs)
    int out2_f16;            // skinny kernel: the split destination (KV cache) is fp16
    // skinny kernel split-K (ksplit > 1): partial sums [n/16][ksplit][MT*256] + one arrival counter per column block
    float* sk_part;
    unsigned* sk_cnt;
    int ksplit;
    // ragged batches (gemm_tile only): input time steps at or beyond in_lens[batch row] read as zero, as if each row were a sequence of
    // its own length with the convolution's zero padding behind it (null: every row has t_in steps)
    const int* in_lens;
    // gemm_ring tile order: 0 = the N tiles of one row panel first (blocks that share an ACTIVATION panel sit on one XCD: the flow's
    // projections), 1 = the row panels of one N tile first (blocks that share a WEIGHT tile are neighbours on one XCD and ask its L2 for
    // the same lines at the same time: the kNN scan, where the "weights" are the 1.2 GB bank and the activations 256 queries)
    int m_first;
    // the kNN scan's epilogue (gemm_scan): y[m][n] = acc * col_scale[n] (+ row_qs[m] * col_bias[n]) is what leaves, and the largest y of
    // every (row, 64-column block) goes to blockmax[m * bm_ld + n / 64] -- the selection then reads c blocks per query, not the row
    const float* col_scale;
    const float* col_bias;
    const float* row_qs;
    float* blockmax;
    int bm_ld;
};

#ifdef EPI_DBG_LOCAL
#define EPI_ROW(m) ((m) - m0)   // microbenchmark only: every block stores to the first rows (no HBM traffic)
#else
#define EPI_ROW(m) (m)
#endif
// Fast epilogue for interior blocks (all BM x BN outputs valid, 16-byte aligned rows, no row scale): the activation,
// the output type and the residual are compile-time, so the per-vector work is one ds_read_b128, the arithmetic and one
// store, with every residual load of the slab issued before the first store.  The general epilogue below spends
// ~1.5k VALU instructions per wave on predicates and the activation switch -- 2-3x the whole main loop of the small-K
// projections of the flow decoder (measured: 13 of 22 us on 5504 x 1536 x 256).
template <int TM, int TN, int ACT, bool OUT16, bool RES>
__device__ __forceinline__ void tile_epilogue_fast(const GemmArgs& a, float16v (&acc)[TM][TN], float* slab_base, int64_t m0, int n0,
                                                   int wm, int wn, int wid, int lane) {
    constexpr int EPI_W = 32 * TN + 4;
    constexpr int VPR = 8 * TN;            // float4 vectors per slab row
    constexpr int RPI = 64 / VPR;          // rows per wave-instruction
    constexpr int NIT = 32 / RPI;
    const int r = lane & 31, h = lane >> 5;
    const int vq = lane % VPR, vr = lane / VPR;
    float* slab = slab_base + wid * 32 * EPI_W;
    const int nb = n0 + wn * TN * 32 + vq * 4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + nb);
    const float alpha = a.alpha, slope = a.slope;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t mrow = m0 + (wm * TM + i) * 32 + vr;
        float4 res[NIT];
        if constexpr (RES) {
            const float* rp = a.residual + mrow * a.ldr + nb;
#pragma unroll
            for (int it = 0; it < NIT; ++it) res[it] = *reinterpret_cast<const float4*>(rp + (int64_t)it * RPI * a.ldr);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_W + j * 32 + r] = acc[i][j][e];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const float4* srow = reinterpret_cast<const float4*>(slab + vr * EPI_W + vq * 4);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float4 v = srow[it * RPI * (EPI_W / 4)];
            if constexpr (ACT == ACT_GELU) {
                v.x = gelu_erf_fast(v.x + bias4.x) * alpha;
                v.y = gelu_erf_fast(v.y + bias4.y) * alpha;
                v.z = gelu_erf_fast(v.z + bias4.z) * alpha;
                v.w = gelu_erf_fast(v.w + bias4.w) * alpha;
            } else {
                v.x = apply_act(v.x + bias4.x, ACT, slope) * alpha;
                v.y = apply_act(v.y + bias4.y, ACT, slope) * alpha;
                v.z = apply_act(v.z + bias4.z, ACT, slope) * alpha;
                v.w = apply_act(v.w + bias4.w, ACT, slope) * alpha;
            }
            if constexpr (RES) {
                v.x += res[it].x; v.y += res[it].y; v.z += res[it].z; v.w += res[it].w;
            }
            const int64_t o = EPI_ROW(mrow + it * RPI) * a.ldc + nb;
            if constexpr (OUT16) {
                half4 h4;
                h4[0] = (_Float16)v.x; h4[1] = (_Float16)v.y; h4[2] = (_Float16)v.z; h4[3] = (_Float16)v.w;
                *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(a.out) + o) = h4;
            } else {
                *reinterpret_cast<float4*>(a.out + o) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Epilogue of the row-complete tile (block = 32 rows x all n = 128 TN columns, 4 waves side by side in N): the new residual
// stream value  out = acc + bias + residual  (fp32) AND LayerNorm(out) * gamma + beta (fp16, the next GEMM's operand) leave
// in one pass -- the separate LayerNorm launch (5 us of launch floor for 8 MB of traffic, 1280 of them per flow solve) and
// its re-read of `out` disappear.  Exact two-pass statistics: row sums, then squared deviations, each reduced over the 16
// lanes that share a row and across the 4 waves through LDS (fixed order).
template <int TN>
__device__ __forceinline__ void tile_epilogue_ln(const GemmArgs& a, float16v (&acc)[TN], float* slab_base, float* stat, int64_t m0, int wn,
                                                 int wid, int lane) {
    constexpr int EPI_W = 32 * TN + 4;
    constexpr int VPR = 8 * TN, RPI = 64 / VPR, NIT = 32 / RPI;
    const int r = lane & 31, h = lane >> 5;
    const int vq = lane % VPR, vr = lane / VPR;
    float* slab = slab_base + wid * 32 * EPI_W;
    const int nb = wn * TN * 32 + vq * 4;
    const int ncols = 4 * 32 * TN;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + nb);
    const float4 ga = *reinterpret_cast<const float4*>(a.ln_gamma + nb);
    const float4 be = *reinterpret_cast<const float4*>(a.ln_beta + nb);
    float4 res[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int64_t m = m0 + it * RPI + vr;
        res[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.residual && m < a.m) res[it] = *reinterpret_cast<const float4*>(a.residual + m * a.ldr + nb);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_W + j * 32 + r] = acc[j][e];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float4 v[NIT];
    const float4* srow = reinterpret_cast<const float4*>(slab + vr * EPI_W + vq * 4);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        float4 t = srow[it * RPI * (EPI_W / 4)];
        t.x = (t.x + bias4.x) * a.alpha + res[it].x;
        t.y = (t.y + bias4.y) * a.alpha + res[it].y;
        t.z = (t.z + bias4.z) * a.alpha + res[it].z;
        t.w = (t.w + bias4.w) * a.alpha + res[it].w;
        v[it] = t;
        const int64_t m = m0 + it * RPI + vr;
        if (m < a.m) *reinterpret_cast<float4*>(a.out + EPI_ROW(m) * a.ldc + nb) = t;
        float s1 = (t.x + t.y) + (t.z + t.w);
#pragma unroll
        for (int off = 1; off < VPR; off <<= 1) s1 += __shfl_xor(s1, off, 64);
        if (vq == 0) stat[wid * 32 + it * RPI + vr] = s1;
    }
    __syncthreads();
    float mean[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + vr;
        mean[it] = ((stat[row] + stat[32 + row]) + (stat[64 + row] + stat[96 + row])) / (float)ncols;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const float dx = v[it].x - mean[it], dy = v[it].y - mean[it], dz = v[it].z - mean[it], dw = v[it].w - mean[it];
        float s2 = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
        for (int off = 1; off < VPR; off <<= 1) s2 += __shfl_xor(s2, off, 64);
        if (vq == 0) stat[wid * 32 + it * RPI + vr] = s2;
    }
    __syncthreads();
    _Float16* lnout = reinterpret_cast<_Float16*>(a.out2);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + vr;
        const float var = ((stat[row] + stat[32 + row]) + (stat[64 + row] + stat[96 + row])) / (float)ncols;
        const float rstd = rsqrtf(var + a.ln_eps);
        const int64_t m = m0 + row;
        if (m < a.m) {
            half4 h4;
            h4[0] = (_Float16)((v[it].x - mean[it]) * rstd * ga.x + be.x);
            h4[1] = (_Float16)((v[it].y - mean[it]) * rstd * ga.y + be.y);
            h4[2] = (_Float16)((v[it].z - mean[it]) * rstd * ga.z + be.z);
            h4[3] = (_Float16)((v[it].w - mean[it]) * rstd * ga.w + be.w);
            *reinterpret_cast<half4*>(lnout + EPI_ROW(m) * a.ldc2 + nb) = h4;
        }
    }
}

// Epilogue of the kNN scan (GemmArgs::blockmax): the slab transpose of tile_epilogue, then per 16-byte vector the score
// y = acc * col_scale (+ row_qs * col_bias: L2's -|b|^2 / 2 on the query's scale) with exactly the arithmetic the selection kernels
// applied to the raw dot products before (knn.hip, knn_select_body), the store, and the maximum over the wave's 64 columns of the row:
// the 16 lanes that hold one slab row reduce it with four lane exchanges and leave it in bm_lds[tile row][wave column]; the kernel
// stores the tile's maxima behind a barrier, one 8- or 16-byte vector per row (tile_store_blockmax: a 4-byte store per (row, block)
// straight from here was 1/4 of the epilogue's cost).  Columns past n take no part; NaNs never win an fmaxf.
template <int TM, int TN, int WN, bool INTERIOR>
__device__ __forceinline__ void tile_epilogue_knn_t(const GemmArgs& a, float16v (&acc)[TM][TN], float* slab_base, float* bm_lds, int64_t m0,
                                                    int n0, int wm, int wn, int wid, int lane) {
    constexpr int EPI_W = 32 * TN + 4, VPR = 8 * TN, RPI = 64 / VPR, NIT = 32 / RPI;
    const int r = lane & 31, h = lane >> 5;
    const int vq = lane % VPR, vr = lane / VPR;
    float* slab = slab_base + wid * 32 * EPI_W;
    const int nb = n0 + wn * 64 + vq * 4;
    float cs[4], cb[4];
    bool ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ok[e] = INTERIOR || nb + e < a.n;
        cs[e] = ok[e] ? a.col_scale[nb + e] : 0.0f;
        cb[e] = ok[e] && a.col_bias ? a.col_bias[nb + e] : 0.0f;
    }
    const bool has_bias = a.col_bias != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_W + j * 32 + r] = acc[i][j][e];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float qs[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int64_t m = m0 + (wm * TM + i) * 32 + it * RPI + vr;
            qs[it] = has_bias && (INTERIOR || m < a.m) ? a.row_qs[m] : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + vr;
            const int64_t m = m0 + (wm * TM + i) * 32 + row;
            const float4 v = *reinterpret_cast<const float4*>(&slab[row * EPI_W + vq * 4]);
            float y[4] = {v.x * cs[0], v.y * cs[1], v.z * cs[2], v.w * cs[3]};
            if (has_bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaf(qs[it], cb[e], y[e]);
            }
            float mx;
            if constexpr (INTERIOR) {
                mx = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
            } else {
                mx = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ok[e]) mx = fmaxf(mx, y[e]);
            }
#pragma unroll
            for (int off = 1; off < VPR; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            if (vq == 0) bm_lds[((wm * TM + i) * 32 + row) * WN + wn] = mx;
            if (INTERIOR) {
                *reinterpret_cast<float4*>(a.out + m * a.ldc + nb) = make_float4(y[0], y[1], y[2], y[3]);
            } else if (m < a.m) {
                float* op = a.out + m * a.ldc + nb;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ok[e]) op[e] = y[e];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int TM, int TN, int WN>
__device__ __forceinline__ void tile_epilogue_knn(const GemmArgs& a, float16v (&acc)[TM][TN], float* slab_base, float* bm_lds, int64_t m0,
                                                  int n0, int wm, int wn, int wid, int lane, bool interior) {
    if constexpr (TN == 2) {
        if (interior && (a.ldc & 3) == 0 && ((uintptr_t)a.out & 15) == 0) tile_epilogue_knn_t<TM, TN, WN, true>(a, acc, slab_base, bm_lds, m0, n0, wm, wn, wid, lane);
        else tile_epilogue_knn_t<TM, TN, WN, false>(a, acc, slab_base, bm_lds, m0, n0, wm, wn, wid, lane);
    }
}

// the tile's block maxima, bm_lds[BM rows][WN blocks], to blockmax[m][n0 / 64 ..): one vector per row (bm_ld and n0 / 64 are multiples of WN)
template <int BM, int WN>
__device__ __forceinline__ void tile_store_blockmax(const GemmArgs& a, const float* bm_lds, int64_t m0, int n0, int tid, int nthreads) {
    __syncthreads();
    for (int row = tid; row < BM; row += nthreads) {
        const int64_t m = m0 + row;
        if (m >= a.m) continue;
        float* dst = a.blockmax + m * a.bm_ld + (n0 >> 6);
        if (n0 + 64 * WN <= a.n + 63 && (a.bm_ld % WN) == 0) {      // every block of the tile has a column below n
            if constexpr (WN == 4) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(bm_lds + row * 4);
            else if constexpr (WN == 2) *reinterpret_cast<float2*>(dst) = *reinterpret_cast<const float2*>(bm_lds + row * 2);
        } else {
            for (int w = 0; w < WN; ++w)
                if (n0 + 64 * w < a.n) dst[w] = bm_lds[row * WN + w];
        }
    }
}

// Epilogue shared by the tile kernels.  The accumulator layout (column on the lane, 16 rows in registers) would cost
// 16*TM*TN four-byte stores per lane; instead each wave transposes one 32 x (32*TN) slab at a time through its own
// LDS region (slab_base: 4 x 32 x EPI_W floats holding no live data) and stores whole 16-byte vectors
// (store-issue bound otherwise: 2-3x the main loop).
template <int TM, int TN>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& a, float16v (&acc)[TM][TN], float* slab_base, int64_t m0, int n0,
                                              int wm, int wn, int wid, int lane, bool interior) {
    constexpr int EPI_W = 32 * TN + 4;
    if (interior && !a.row_scale && (a.ldc & 3) == 0 && ((uintptr_t)a.out & 15) == 0 && ((uintptr_t)a.bias & 15) == 0 &&
        (!a.residual || ((a.ldr & 3) == 0 && ((uintptr_t)a.residual & 15) == 0))) {
        if (a.act == ACT_NONE && !a.residual && a.out_f16) return tile_epilogue_fast<TM, TN, ACT_NONE, true, false>(a, acc, slab_base, m0, n0, wm, wn, wid, lane);
        if (a.act == ACT_NONE && !a.residual && !a.out_f16) return tile_epilogue_fast<TM, TN, ACT_NONE, false, false>(a, acc, slab_base, m0, n0, wm, wn, wid, lane);
        if (a.act == ACT_NONE && a.residual && !a.out_f16) return tile_epilogue_fast<TM, TN, ACT_NONE, false, true>(a, acc, slab_base, m0, n0, wm, wn, wid, lane);
        if (a.act == ACT_GELU && !a.residual && a.out_f16) return tile_epilogue_fast<TM, TN, ACT_GELU, true, false>(a, acc, slab_base, m0, n0, wm, wn, wid, lane);
    }
    const int r = lane & 31, h = lane >> 5;
    float* slab = slab_base + wid * 32 * EPI_W;
    constexpr int VPR = 8 * TN;            // float4 vectors per slab row
    constexpr int RPI = 64 / VPR;          // rows per wave-instruction
    const int vq = lane % VPR, vr = lane / VPR;
    _Float16* out16 = reinterpret_cast<_Float16*>(a.out);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_W + j * 32 + r] = acc[i][j][e];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int nb = n0 + wn * TN * 32 + vq * 4;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) {
            bias4.x = nb < a.n ? a.bias[nb] : 0.f;
            bias4.y = nb + 1 < a.n ? a.bias[nb + 1] : 0.f;
            bias4.z = nb + 2 < a.n ? a.bias[nb + 2] : 0.f;
            bias4.w = nb + 3 < a.n ? a.bias[nb + 3] : 0.f;
        }
#pragma unroll 1  // rolled on purpose: the activation switch must appear once, not 32 times (I-cache: 148 KB -> ~12 KB)
        for (int rr = 0; rr < 32; rr += RPI) {
            const int row = rr + vr;
            const int64_t m = m0 + (wm * TM + i) * 32 + row;
            float4 v = *reinterpret_cast<const float4*>(&slab[row * EPI_W + vq * 4]);
            if (m < a.m && nb < a.n) {
                v.x = apply_act(v.x + bias4.x, a.act, a.slope) * a.alpha;
                v.y = apply_act(v.y + bias4.y, a.act, a.slope) * a.alpha;
                v.z = apply_act(v.z + bias4.z, a.act, a.slope) * a.alpha;
                v.w = apply_act(v.w + bias4.w, a.act, a.slope) * a.alpha;
                if (a.row_scale) {
                    const float rs = a.row_scale[m];
                    v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
                }
                const bool full = nb + 4 <= a.n;
                if (a.residual) {
                    const float* rp = a.residual + m * a.ldr + nb;
                    if (full && (a.ldr & 3) == 0) {
                        const float4 r4 = *reinterpret_cast<const float4*>(rp);
                        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                    } else {
                        v.x += rp[0];
                        if (nb + 1 < a.n) v.y += rp[1];
                        if (nb + 2 < a.n) v.z += rp[2];
                        if (nb + 3 < a.n) v.w += rp[3];
                    }
                }
                if (a.out_f16) {
                    _Float16* op = out16 + EPI_ROW(m) * a.ldc + nb;
                    if (full && (a.ldc & 3) == 0) {
                        half4 h4;
                        h4[0] = (_Float16)v.x; h4[1] = (_Float16)v.y; h4[2] = (_Float16)v.z; h4[3] = (_Float16)v.w;
                        *reinterpret_cast<half4*>(op) = h4;
                    } else {
                        op[0] = (_Float16)v.x;
                        if (nb + 1 < a.n) op[1] = (_Float16)v.y;
                        if (nb + 2 < a.n) op[2] = (_Float16)v.z;
                        if (nb + 3 < a.n) op[3] = (_Float16)v.w;
                    }
                } else {
                    float* op = a.out + EPI_ROW(m) * a.ldc + nb;
                    if (full && (a.ldc & 3) == 0) {
                        *reinterpret_cast<float4*>(op) = v;
                    } else {
                        op[0] = v.x;
                        if (nb + 1 < a.n) op[1] = v.y;
                        if (nb + 2 < a.n) op[2] = v.z;
                        if (nb + 3 < a.n) op[3] = v.w;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// A16: the activations arrive as fp16 (written by a producer whose only consumers are MFMA operands:
// LayerNorm / GroupNorm / attention / a previous GEMM), so staging is a plain 16-byte copy.
// BKT: K elements staged per iteration (32 / 64 / 128).  These GEMMs run at ~1-2 blocks per CU, so the
// bytes in flight that hide HBM/L2 latency must come from inside the block: a wide K tile keeps
// (BM + BN) * BKT * 2 bytes of loads outstanding per block and halves / quarters the barrier count.
template <int WM, int WN, int TM, int TN, bool A16, int BKT>
__global__ __launch_bounds__(256) void gemm_tile(GemmArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int BK = BKT;
    constexpr int LDS_ROW = BKT + 8;              // halfs per staged row: (2*BKT + 16) bytes = odd multiple of 16
    constexpr int CPR = BKT / 8;                  // 8-element chunks per row
    constexpr int A_CH = (BM * CPR + 255) / 256;
    constexpr int B_CH = (BN * CPR + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    constexpr int STAGE_HALFS = 2 * (BM + BN) * LDS_ROW;
    constexpr int EPI_W = 32 * TN + 4;           // padded fp32 row of a wave's 32 x (32*TN) output slab
    static_assert(4 * 32 * EPI_W * 4 <= STAGE_HALFS * 2, "epilogue slab must fit in the staging LDS");
    __shared__ __attribute__((aligned(16))) _Float16 smem[STAGE_HALFS];
    _Float16* sa0 = smem;                         // [2][BM * LDS_ROW]
    _Float16* sb0 = smem + 2 * BM * LDS_ROW;      // [2][BN * LDS_ROW]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int ktot = a.taps * a.cin_pad;
    const int nkt = ktot / BK;
    const _Float16* x16 = reinterpret_cast<const _Float16*>(a.x);
    const bool vec_ok = A16 ? ((a.lda & 7) == 0 && ((uintptr_t)a.x & 15) == 0) : ((a.lda & 3) == 0 && ((uintptr_t)a.x & 15) == 0);

    int a_row[A_CH], a_seg[A_CH], a_t[A_CH], a_lim[A_CH];
    int64_t a_base[A_CH];
    bool a_live[A_CH];
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
        const int id = tid + c * 256;
        a_row[c] = id / CPR;
        a_seg[c] = (id % CPR) * 8;
        const int64_t m = m0 + a_row[c];
        a_live[c] = (id < BM * CPR) && m < a.m;
        const int64_t b = a_live[c] ? m / a.t_out : 0;
        const int t = a_live[c] ? (int)(m - b * a.t_out) : 0;
        a_t[c] = t * a.stride - a.pad;
        a_base[c] = b * a.t_in;
        a_lim[c] = (a.in_lens && a_live[c]) ? min(a.in_lens[b], a.t_in) : a.t_in;
    }
    int b_row[B_CH], b_seg[B_CH];
    bool b_live[B_CH];
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
        const int id = tid + c * 256;
        b_row[c] = id / CPR;
        b_seg[c] = (id % CPR) * 8;
        b_live[c] = id < BN * CPR;
    }

    half8 ra[A_CH];
    half8 rb[B_CH];

    // Every load of a tile is UNCONDITIONAL on the usual shapes (aligned rows, channels a multiple of 8): the address of a chunk that lies
    // in the padding / outside the sequence is replaced by the tensor's base and the chunk zeroed afterwards.  A load inside the bounds
    // check is waited for at the end of its block, which made the A_CH + B_CH chunks of a tile as many dependent round trips.
    const bool fast = vec_ok && (a.cin & 7) == 0;
    // (weight rows are padded to whole tiles: a chunk index beyond the tile -- only when BN * CPR is not a multiple of 256 -- re-reads chunk 0)
    auto load_b = [&](int k0) {
#pragma unroll
        for (int c = 0; c < B_CH; ++c)
            rb[c] = *reinterpret_cast<const half8*>(a.w + (int64_t)(n0 + (b_live[c] ? b_row[c] : 0)) * ktot + k0 + (b_live[c] ? b_seg[c] : 0));
    };
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int tap = k0 / a.cin_pad;
        const int c0 = k0 - tap * a.cin_pad;
        if (fast) {
            bool okc[A_CH];
            if constexpr (A16) {
#pragma unroll
                for (int c = 0; c < A_CH; ++c) {
                    const int ts = a_t[c] + tap * a.dil;
                    const int ch = c0 + a_seg[c];
                    okc[c] = a_live[c] && ts >= 0 && ts < a_lim[c] && ch < a.cin;
                    const int64_t off = okc[c] ? (a_base[c] + ts) * (int64_t)a.lda + ch : 0;
                    ra[c] = *reinterpret_cast<const half8*>(x16 + off);
                }
                load_b(k0);
#pragma unroll
                for (int c = 0; c < A_CH; ++c)
                    if (!okc[c]) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) ra[c][j] = (_Float16)0.0f;
                    }
            } else {
                float4 v0[A_CH], v1[A_CH];
#pragma unroll
                for (int c = 0; c < A_CH; ++c) {
                    const int ts = a_t[c] + tap * a.dil;
                    const int ch = c0 + a_seg[c];
                    okc[c] = a_live[c] && ts >= 0 && ts < a_lim[c] && ch < a.cin;
                    const int64_t off = okc[c] ? (a_base[c] + ts) * (int64_t)a.lda + ch : 0;
                    v0[c] = *reinterpret_cast<const float4*>(a.x + off);
                    v1[c] = *reinterpret_cast<const float4*>(a.x + off + 4);
                }
                load_b(k0);
#pragma unroll
                for (int c = 0; c < A_CH; ++c) {
                    if (!okc[c]) v0[c] = v1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    ra[c][0] = (_Float16)v0[c].x; ra[c][1] = (_Float16)v0[c].y; ra[c][2] = (_Float16)v0[c].z; ra[c][3] = (_Float16)v0[c].w;
                    ra[c][4] = (_Float16)v1[c].x; ra[c][5] = (_Float16)v1[c].y; ra[c][6] = (_Float16)v1[c].z; ra[c][7] = (_Float16)v1[c].w;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < A_CH; ++c) {
                const int ts = a_t[c] + tap * a.dil;
                const bool ok = a_live[c] && ts >= 0 && ts < a_lim[c];
                const int ch = c0 + a_seg[c];
                const int64_t off = (a_base[c] + ts) * (int64_t)a.lda + ch;
                if constexpr (A16) {
                    if (ok && vec_ok && ch + 8 <= a.cin) {
                        ra[c] = *reinterpret_cast<const half8*>(x16 + off);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) ra[c][j] = (ok && ch + j < a.cin) ? x16[off + j] : (_Float16)0.0f;
                    }
                } else {
                    float tmp[8];
                    if (ok && vec_ok && ch + 8 <= a.cin) {
                        const float4 v0 = *reinterpret_cast<const float4*>(a.x + off);
                        const float4 v1 = *reinterpret_cast<const float4*>(a.x + off + 4);
                        tmp[0] = v0.x; tmp[1] = v0.y; tmp[2] = v0.z; tmp[3] = v0.w; tmp[4] = v1.x; tmp[5] = v1.y; tmp[6] = v1.z; tmp[7] = v1.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) tmp[j] = (ok && ch + j < a.cin) ? a.x[off + j] : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) ra[c][j] = (_Float16)tmp[j];
                }
            }
            load_b(k0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int c = 0; c < A_CH; ++c)
            if (tid + c * 256 < BM * CPR) *reinterpret_cast<half8*>(&sa0[buf * BM * LDS_ROW + a_row[c] * LDS_ROW + a_seg[c]]) = ra[c];
#pragma unroll
        for (int c = 0; c < B_CH; ++c)
            if (b_live[c]) *reinterpret_cast<half8*>(&sb0[buf * BN * LDS_ROW + b_row[c] * LDS_ROW + b_seg[c]]) = rb[c];
    };

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const _Float16* sa = sa0 + buf * BM * LDS_ROW;
        const _Float16* sb = sb0 + buf * BN * LDS_ROW;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const half8*>(&sa[((wm * TM + i) * 32 + r) * LDS_ROW + ks * 16 + h * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const half8*>(&sb[((wn * TN + j) * 32 + r) * LDS_ROW + ks * 16 + h * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    tile_epilogue<TM, TN>(a, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, wid, lane, m0 + BM <= a.m && n0 + BN <= a.n);
}

// ------------------------------------------------------------------------------------------
// gemm_ring: plain GEMM (taps == 1) on fp16 activations, operands staged global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers) into a STAGES-deep ring of 64-wide K tiles.
//
// Why: the flow decoder's projections are small-K GEMMs (K = 256..1024 on 5.5k..11k rows) at 1-2 blocks
// per CU; with register staging one K tile (16 KB) is in flight per block and the kernels sit at ~20 us
// where their bytes and flops need ~5.  Here (STAGES - 1) tiles are in flight per block across the
// barriers: a counted s_waitcnt vmcnt(N) + a raw s_barrier per K tile (a __syncthreads() would drain
// the DMA queue), nothing else touches vmcnt inside the loop.
//
// LDS image of one stage: (BM + BN) rows of 128 B (64 halfs of K).  One wave-instruction writes 1 KiB =
// 8 consecutive rows, lane l -> row l / 8, 16-byte slot l % 8: the destination is lane-linear by
// construction, so the bank swizzle goes on the SOURCE address: slot s of row r holds K chunk
// s ^ ((r >> 1) & 7), and the fragment reads apply the same XOR (16 lanes = 16 rows of one chunk then
// cover all 64 banks once).
// ------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// Logical tile L -> (row tile bx, column tile by) of the ring kernels: GROUPS of `gm` row tiles, column-major inside a group, so that
// the tiles one XCD multiplies at the same time (32 consecutive L for the one-block-per-CU 256 x 256 tile) form a gm x (32 / gm)
// rectangle and share gm + 32 / gm operand panels in that XCD's L2 instead of 1 + 32.  gm = 0: one group of all row tiles (row tile
// fastest: GemmArgs::m_first).
__device__ __forceinline__ void ring_tile_of(int L, int gx, int gy, int gm, int& bx, int& by) {
    if (gm <= 0 || gm > gx) gm = gx;
    const int per = gm * gy;
    const int grp = L / per, idx = L - grp * per;
    const int first = grp * gm;
    const int gsz = gx - first < gm ? gx - first : gm;
    by = idx / gsz;
    bx = first + idx - by * gsz;
}

template <int TM, int TN, int STAGES, int WN = 2, int NW = 4>      // NW waves as (NW / WN) x WN; wave tile (32 TM) x (32 TN)
__global__ __launch_bounds__(NW * 64) void gemm_ring(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (it cannot parse the LDS-DMA builtin)
    constexpr int WM = NW / WN;
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE_BYTES = ROWS * 128;
    constexpr int IPW = ROWS / (8 * NW);           // DMA instructions per wave and stage (8 rows each)
    constexpr int EPI_W = 32 * TN + 4;
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    static_assert(ROWS % (8 * NW) == 0, "whole DMA instructions per wave");
    static_assert(NW * 32 * EPI_W * 4 <= STAGES * STAGE_BYTES, "epilogue slabs must fit in the ring");
    static_assert((STAGES - 2) * IPW <= 63, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order (1-D launch): the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with
    // its own L2.  Workgroup w becomes logical tile L = (its XCD's contiguous range) + w / 8, and L walks groups of 4 (256 x 256)
    // or 8 row panels column by column (ring_tile_of): the blocks that run on one XCD at the same time share a few activation AND
    // weight panels in its L2.  PMC on 8192^3 with the 256 x 256 tile, one row panel at a time -> groups of 4: TCC hit rate 48 % -> 80 %,
    // FETCH_SIZE 4.45 GB -> 1.7 GB per launch, +4 % (profiles/r06_ring_group_ab.log, r06_pmc_ring8.txt).
    const int gy = (a.n + BN - 1) / BN;
    int bx, by;
    {
        const int nwg = gridDim.x, w = blockIdx.x;
        const int xcd = w & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
        ring_tile_of(L, (int)((a.m + BM - 1) / BM), gy, a.m_first ? 0 : (BM >= 256 ? 4 : 8), bx, by);
    }
    const int64_t m0 = (int64_t)bx * BM;
    const int n0 = by * BN;
    const int ktot = a.cin_pad;
    const int nkt = ktot / 64;
    const _Float16* x16 = reinterpret_cast<const _Float16*>(a.x);

    // per-lane source of this wave's IPW row groups at K tile 0
    const _Float16* src[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int row = 8 * (wid + NW * i) + (lane >> 3);
        const int slot = lane & 7;
        if (row < BM) {
            const int c = slot ^ ((row >> 1) & 7);
            int64_t m = m0 + row;
            if (m >= a.m) m = a.m - 1;             // rows past the end: any valid address, never stored
            src[i] = x16 + m * a.lda + c * 8;
        } else {
            const int lr = row - BM;
            const int c = slot ^ ((lr >> 1) & 7);
            int nn = n0 + lr;                      // packed weights are padded to whole tiles; a raw [n][K] matrix (the kNN bank) is not
            if (nn >= a.n) nn = a.n - 1;
            src[i] = a.w + (int64_t)nn * ktot + c * 8;
        }
    }
    auto issue = [&](int kt) {
        unsigned char* dst = ring + (kt % STAGES) * STAGE_BYTES + wid * 1024;
#ifndef RING_SKIP_LOAD
#pragma unroll
        for (int i = 0; i < IPW; ++i)
            __builtin_amdgcn_global_load_lds(src[i] + kt * 64, (__attribute__((address_space(3))) void*)(dst + i * (NW * 1024)), 16, 0, 0);
#endif
    };

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // fragment byte offsets inside a stage (row part; the K chunk is XORed in per k-step)
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int lr = (wm * TM + i) * 32 + r;
        a_off[i] = lr * 128;
        a_sw[i] = (lr >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int lr = (wn * TN + j) * 32 + r;
        b_off[j] = BM * 128 + lr * 128;
        b_sw[j] = (lr >> 1) & 7;
    }

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nkt) issue(s);

    for (int kt = 0; kt < nkt; ++kt) {
        // K tile kt has landed once at most the later tiles' DMAs are outstanding
        const int rem = nkt - kt;
        if (rem >= STAGES - 1) wait_vmcnt<(STAGES - 2) * IPW>();
        else if (rem == 2) wait_vmcnt<IPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // every wave's part of tile kt is in LDS; tile kt-1 is no longer read
        if (kt + STAGES - 1 < nkt) issue(kt + STAGES - 1);
        const unsigned char* st = ring + (kt % STAGES) * STAGE_BYTES;
#ifndef RING_SKIP_MFMA
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8 fa[TM], fb[TN];
            const int c = ks * 2 + h;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const half8*>(st + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const half8*>(st + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
#endif
    }
    __syncthreads();                               // all fragment reads done (and no DMA outstanding): the ring becomes slab space
#ifndef RING_SKIP_EPI
    if constexpr (WN == 4 && TM == 1 && NW == 4) {
        if (a.ln_gamma) {      // block = whole output rows (n == BN): residual add + LayerNorm of the result fused in
            tile_epilogue_ln<TN>(a, acc[0], reinterpret_cast<float*>(ring), reinterpret_cast<float*>(ring + 4 * 32 * EPI_W * 4), m0, wn, wid, lane);
            return;
        }
    }
    if (a.blockmax) {
        if constexpr (TN == 2 && (WN == 2 || WN == 4) && NW * 32 * EPI_W * 4 + BM * WN * 4 <= STAGES * STAGE_BYTES) {
            float* bm_lds = reinterpret_cast<float*>(ring + NW * 32 * EPI_W * 4);      // behind the waves' slabs
            tile_epilogue_knn<TM, TN, WN>(a, acc, reinterpret_cast<float*>(ring), bm_lds, m0, n0, wm, wn, wid, lane, m0 + BM <= a.m && n0 + BN <= a.n);
            tile_store_blockmax<BM, WN>(a, bm_lds, m0, n0, tid, NW * 64);
        }
    } else {
        tile_epilogue<TM, TN>(a, acc, reinterpret_cast<float*>(ring), m0, n0, wm, wn, wid, lane, m0 + BM <= a.m && n0 + BN <= a.n);
    }
#else
    if (acc[0][0][0] == 123.0f) a.out[0] = 1.0f;
#endif
#endif
}

// ------------------------------------------------------------------------------------------
// gemm_ring8: the 256 x 256 tile of gemm_ring<4, 2, 2, 4, 8> on the CDNA guide's EIGHT-PHASE schedule (cdna_hip_programming.md, "The
// 256^2 8-phase template"): instead of one barrier, 24 fragment reads and 32 MFMAs per K tile and wave, a K tile is four PHASES of
// (fragment reads of ONE accumulator quadrant's operands + one sub-tile's LDS-DMA, barrier, 8 MFMAs, barrier), and the two wave groups
// of a SIMD (waves w and w + 4: the two row halves of the tile) run ONE BARRIER APART, so that one group's MFMAs cover the other's
// reads and DMA issue.  Same LDS image per stage as gemm_ring (512 rows of 128 B, swizzle on the DMA source and on the reads), two
// stages = 128 KB = eight sub-tile slots.
//
// Sub-tiles (16 KB = 2 DMA instructions per wave) are cut by what a PHASE reads, not by halves of the tile:
//   A0 = row tiles 0, 1 of both wave rows (rows 0-63, 128-191)    A1 = row tiles 2, 3 (rows 64-127, 192-255)
//   B0 = column tile 0 of all four wave columns (32 of every 64)   B1 = column tile 1
// and form ONE sequence S_j, j = 4 t + (0: A0, 1: B0, 2: B1, 3: A1) over the K tiles t.  Phase P = 4 t + p multiplies quadrant
//   p0: (A0, B0) after reading both    p1: (A0, B1) after reading B1    p2: (A1, B1) after reading A1    p3: (A1, B0), B0 kept in registers
// so S_j is read for the first time in phase j (A0) or j - 1 and for the LAST time no later than phase j.
//
// Depth.  The DMA runs as far ahead as the slots allow: phase P requests S_(P+6) into the slot of S_(P-2), whose last read is two
// phases back (the restaging distance the guide asks for with staggered groups) -- five sub-tiles (80 KB) in flight after the request,
// four after the wait (measured: no faster than three ahead -- the kernel is not bound by bytes in flight -- and never slower).
//
// Ordering of the LDS-DMA data (nothing but the issuing wave's counted vmcnt followed by a barrier the READER has passed orders it):
// what phase P + 1 reads first -- everything up to S_(P+2); up to S_(P+1) when P + 1 is a p3 -- is waited for by every wave before global
// barrier 2 (P + 1): the leading group at the end of its phase P, the trailing group (one barrier behind) in the read section of ITS
// phase P, with the same count in both places: vmcnt(2 x the sub-tiles requested behind the needed one).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gemm_ring8(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BN = 256, STAGE_BYTES = 512 * 128, EPI_W = 32 * 2 + 4;
    static_assert(8 * 32 * EPI_W * 4 <= 2 * STAGE_BYTES, "epilogue slabs must fit in the ring");
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the group branches below)
    const int wm = wid >> 2, wn = wid & 3;           // wave tile: rows [128 wm, +128), columns [64 wn, +64)
    const int r = lane & 31, h = lane >> 5;
    const int gy = (a.n + BN - 1) / BN;
    int bx, by;
    {
        const int nwg = gridDim.x, w = blockIdx.x;
        const int xcd = w & 7, q = nwg >> 3, rr = nwg & 7;
        const int L = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (w >> 3);
        ring_tile_of(L, (int)((a.m + BM - 1) / BM), gy, a.m_first ? 0 : 4, bx, by);
    }
    const int64_t m0 = (int64_t)bx * BM;
    const int n0 = by * BN;
    const int ktot = a.cin_pad;
    const int nkt = ktot / 64;
    const _Float16* x16 = reinterpret_cast<const _Float16*>(a.x);

    // ---- DMA: sub-tile s (0 A0, 1 B0, 2 B1, 3 A1) = 16 row groups of 8 rows; wave w takes groups w and w + 8.
    //   A sub-tiles: group g -> tile rows 128 (g / 8) + 64 (s == 3) + 8 (g % 8)
    //   B sub-tiles: group g -> tile columns 64 (g / 4) + 32 (s == 2) + 8 (g % 4)
    const _Float16* src[4][2];
    int dst[4][2];                                   // byte offset of the 8-row group inside a stage
#pragma unroll
    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = wid + 8 * i;
            const bool is_a = sb == 0 || sb == 3;
            const int row0 = is_a ? 128 * (g >> 3) + (sb == 3 ? 64 : 0) + 8 * (g & 7) : 64 * (g >> 2) + (sb == 2 ? 32 : 0) + 8 * (g & 3);
            const int row = row0 + (lane >> 3);      // this lane's row of the group, 16-byte slot lane & 7
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            if (is_a) {
                int64_t m = m0 + row;
                if (m >= a.m) m = a.m - 1;
                src[sb][i] = x16 + m * a.lda + c * 8;
                dst[sb][i] = row0 * 128;
            } else {
                int nn = n0 + row;
                if (nn >= a.n) nn = a.n - 1;
                src[sb][i] = a.w + (int64_t)nn * ktot + c * 8;
                dst[sb][i] = (BM + row0) * 128;
            }
        }
    auto issue = [&](int kt, int sb) {               // (sb is a compile-time constant at every call)
        unsigned char* base = ring + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds(src[sb][i] + kt * 64, (__attribute__((address_space(3))) void*)(base + dst[sb][i]), 16, 0, 0);
    };

    float16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    int a_off[4], a_sw[4], b_off[2], b_sw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lr = wm * 128 + i * 32 + r;
        a_off[i] = lr * 128;
        a_sw[i] = (lr >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = wn * 64 + j * 32 + r;
        b_off[j] = BM * 128 + lr * 128;
        b_sw[j] = (lr >> 1) & 7;
    }
    half8 fa[2][4], fb[2][4];                        // two row tiles x 4 k-steps of the current A sub-tile; both column tiles x 4 k-steps
    auto read_a = [&](const unsigned char* st, int i0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fa[i][ks] = *reinterpret_cast<const half8*>(st + a_off[i0 + i] + (((ks * 2 + h) ^ a_sw[i0 + i]) << 4));
    };
    auto read_b = [&](const unsigned char* st, int j) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[j][ks] = *reinterpret_cast<const half8*>(st + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
    };
    auto mma = [&](int i0, int j) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][ks], fb[j][ks], acc[i0 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // the counted wait in front of the NEXT phase's reads.  rem = sub-tiles of the sequence behind this phase's own index P
    // (4 nkt - 1 - P); requested so far: through S_(P + min(6, rem)); needed: through S_(P + min(need, rem)).
    auto wait_for = [&](int rem, int need) {
        const int left = (rem < 6 ? rem : 6) - (rem < need ? rem : need);
        if (left >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (left == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (left == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const bool trail = wm == 1;                      // the trailing group: one barrier behind, waits in its read section
    const int last = 4 * nkt - 1;

    // ---- prologue: S_0 .. S_5 (K tile 0 whole, A0 and B0 of K tile 1); S_0, S_1 must have landed before phase 0
    issue(0, 0);
    issue(0, 1);
    issue(0, 2);
    issue(0, 3);
    if (nkt > 1) {
        issue(1, 0);
        issue(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                    // (global barrier 0: every wave's part of S_0, S_1 is in LDS)
    if (trail) __builtin_amdgcn_s_barrier();

#define RING8_PHASE(STEADY, P, READS, ISSUE_KT, SB, I0, J, NEED)                                              \
    do {                                                                                                      \
        const int rem = last - (4 * kt + (P));                                                                \
        READS;                                                                                                \
        if (STEADY || rem >= 6) issue(ISSUE_KT, SB);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (trail) {                                                                                          \
            if (STEADY) wait_vmcnt<2 * (6 - (NEED))>();                                                       \
            else wait_for(rem, NEED);                                                                         \
        }                                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        mma(I0, J);                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (!trail) {                                                                                         \
            if (STEADY) wait_vmcnt<2 * (6 - (NEED))>();                                                       \
            else wait_for(rem, NEED);                                                                         \
        }                                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                         \
    } while (0)
#define RING8_TILE(STEADY)                                                                                                                     \
    do {                                                                                                                                       \
        const unsigned char* st = ring + (kt & 1) * STAGE_BYTES;                                                                               \
        RING8_PHASE(STEADY, 0, (read_b(st, 0), __builtin_amdgcn_sched_barrier(0), read_a(st, 0)), kt + 1, 2, 0, 0, 2); /* requests B1(kt + 1); p1 reads B1(kt) */ \
        RING8_PHASE(STEADY, 1, read_b(st, 1), kt + 1, 3, 0, 1, 2);                                  /* requests A1(kt + 1); p2 reads A1(kt) */  \
        RING8_PHASE(STEADY, 2, read_a(st, 2), kt + 2, 0, 2, 1, 1);                                  /* requests A0(kt + 2); p3 reads nothing */ \
        RING8_PHASE(STEADY, 3, (void)0, kt + 2, 1, 2, 0, 2);                                        /* requests B0(kt + 2); p0 reads A0, B0(kt + 1) */ \
    } while (0)

    int kt = 0;
    for (; kt < nkt - 2; ++kt) RING8_TILE(true);     // steady state: every phase requests, every wait is a constant
    for (; kt < nkt; ++kt) RING8_TILE(false);        // the last two K tiles: requests run out, the counts shrink
#undef RING8_TILE
#undef RING8_PHASE
    if (!trail) __builtin_amdgcn_s_barrier();        // (the barrier the trailing group took up front)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // all fragment reads done, no DMA outstanding: the ring becomes slab space
    if (a.blockmax) {
        static_assert(8 * 32 * EPI_W * 4 + BM * 4 * 4 <= 2 * STAGE_BYTES, "block maxima behind the slabs");
        float* bm_lds = reinterpret_cast<float*>(ring + 8 * 32 * EPI_W * 4);
        tile_epilogue_knn<4, 2, 4>(a, acc, reinterpret_cast<float*>(ring), bm_lds, m0, n0, wm, wn, wid, lane, m0 + BM <= a.m && n0 + BN <= a.n);
        tile_store_blockmax<BM, 4>(a, bm_lds, m0, n0, tid, 512);
    } else {
        tile_epilogue<4, 2>(a, acc, reinterpret_cast<float*>(ring), m0, n0, wm, wn, wid, lane, m0 + BM <= a.m && n0 + BN <= a.n);
    }
#endif
}

// ------------------------------------------------------------------------------------------
// M <= 32, plain (taps == 1): one block = 16 output columns; 8 waves split K by 64-element lines.
// Lane (c = lane & 15, g = lane >> 4) owns bytes [32g, 32g+32) of row c in every 128-byte weight
// line: two MFMA 16x16x32 k-steps of 8 halfs each (same permuted k order for X and W).
// ------------------------------------------------------------------------------------------
// Structure: (1) every wave issues ALL its weight loads first (up to 8 lines = 16 x 16 B per lane in
// flight), (2) while they fly, wave w stages input rows w, w+8, ... into LDS as fp16 -- gathered,
// LayerNorm'ed (two-pass, fp32) and converted once per block instead of once per line --,
// (3) barrier, MFMAs with A fragments from LDS, (4) cross-wave reduction + epilogue.
static constexpr int SK_WAVES = 8;
static constexpr int SK_LINES = 8;  // weight lines prefetched per wave and pass

template <int MT, int XV>
__global__ __launch_bounds__(512) void gemm_skinny16(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KS = a.ksplit > 1 ? a.ksplit : 1;              // split-K: block (x, y) reduces K slice y of column block x
    const int kfull = a.cin_pad;
    const int ktot = kfull / KS;                             // K elements of this block's slice
    const int k_lo = (int)blockIdx.y * ktot;
    const int lines = ktot >> 6;
    const int M = (int)a.m;
    const int xs = ktot + 8;                                 // LDS row stride in halfs (16-byte skew)
    _Float16* sx = reinterpret_cast<_Float16*>(sk_smem);     // [M][xs]
    float* red = reinterpret_cast<float*>(sk_smem + (((size_t)M * xs * 2 + 15) & ~(size_t)15));  // [8][MT][4][64]

    // Load ORDER matters: vmcnt retires in issue order, so whatever is needed first must be issued first.
    //   (0) this wave's first input row (L2-resident, needed for the LayerNorm right away)
    //   (1) the wave's weight lines (HBM, consumed after the barrier)
    //   (2) the epilogue operands (bias / residual, consumed last)
    // XV == 4: rows of <= 1024 inputs live in registers (prefetched, single-pass LayerNorm).  XV == 16 (K up to 4096,
    // the FFN-out GEMM): rows are staged in rolled 16-byte pieces (no LayerNorm on that path; it falls back to scalar).
    const bool al_ok = (a.lda & 3) == 0 && ((uintptr_t)a.x & 15) == 0 && (a.cin & 3) == 0;
    const bool vec_ok = XV == 4 && al_ok && ktot <= 1024;
    const bool vec_wide = XV != 4 && al_ok && !a.ln_gamma;
    constexpr int MAXV = 4;
    const int nv = (ktot + 255) >> 8;
    float4 v0[MAXV];
    if (vec_ok && wid < M) {
        const int64_t src = a.gather ? (int64_t)a.gather[wid] : (int64_t)wid;
        const float* xr = a.x + src * a.lda + k_lo;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int k = lane * 4 + i * 256;
            v0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nv && k_lo + k < a.cin) v0[i] = *reinterpret_cast<const float4*>(xr + k);
        }
    }
    const _Float16* wrow = a.w + (int64_t)(n0 + c) * kfull + k_lo + g * 16;
    half8 fb[SK_LINES][2];
#pragma unroll
    for (int i = 0; i < SK_LINES; ++i) {
        const int line = wid + i * SK_WAVES;
        if (line < lines) {
            fb[i][0] = *reinterpret_cast<const half8*>(wrow + line * 64);
            fb[i][1] = *reinterpret_cast<const half8*>(wrow + line * 64 + 8);
        }
    }
    float e_bias = 0.0f, e_res = 0.0f;
    if (tid < MT * 256) {
        const int t = tid / 256, e = (tid >> 6) & 3, ln = tid & 63;
        const int n = n0 + (ln & 15), m = t * 16 + (ln >> 4) * 4 + e;
        if (n < a.n && m < M) {
            if (a.bias) e_bias = a.bias[n];
            if (a.residual) e_res = a.residual[(int64_t)m * a.ldr + n];
        }
    }
    // stage x rows (gather + LayerNorm + fp16): wave w owns rows w, w+8, ...
    auto stage_row = [&](float4 (&v)[MAXV], _Float16* dst) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            s1 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            s2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        float mean = 0.0f, rstd = 1.0f;
        if (a.ln_gamma) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                s1 += __shfl_xor(s1, off, 64);
                s2 += __shfl_xor(s2, off, 64);
            }
            mean = s1 / (float)a.cin;
            const float var = fmaxf(s2 / (float)a.cin - mean * mean, 0.0f);
            rstd = rsqrtf(var + a.ln_eps);
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int k = lane * 4 + i * 256;
            if (i < nv && k < ktot) {
                float4 o = v[i];
                if (a.ln_gamma && k < a.cin) {
                    const float4 ga = *reinterpret_cast<const float4*>(a.ln_gamma + k);
                    const float4 be = *reinterpret_cast<const float4*>(a.ln_beta + k);
                    o.x = (o.x - mean) * rstd * ga.x + be.x;
                    o.y = (o.y - mean) * rstd * ga.y + be.y;
                    o.z = (o.z - mean) * rstd * ga.z + be.z;
                    o.w = (o.w - mean) * rstd * ga.w + be.w;
                }
                half4 h4;
                h4[0] = (_Float16)o.x; h4[1] = (_Float16)o.y; h4[2] = (_Float16)o.z; h4[3] = (_Float16)o.w;
                *reinterpret_cast<half4*>(dst + k) = h4;
            }
        }
    };
    if (vec_ok && wid < M) stage_row(v0, sx + (size_t)wid * xs);   // the prefetched row
    for (int mr = vec_ok ? wid + SK_WAVES : wid; mr < M; mr += SK_WAVES) {
        const int64_t src = a.gather ? (int64_t)a.gather[mr] : (int64_t)mr;
        const float* xr = a.x + src * a.lda + k_lo;          // (LayerNorm / scalar paths below only run unsplit: k_lo == 0)
        _Float16* dst = sx + (size_t)mr * xs;
        if (vec_wide) {
#pragma unroll 8  // 8 independent 16-byte loads in flight per lane (a 4096-wide row = 2 round trips)
            for (int k = lane * 4; k < ktot; k += 256) {
                float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k_lo + k < a.cin) v4 = *reinterpret_cast<const float4*>(xr + k);
                half4 h4;
                h4[0] = (_Float16)v4.x; h4[1] = (_Float16)v4.y; h4[2] = (_Float16)v4.z; h4[3] = (_Float16)v4.w;
                *reinterpret_cast<half4*>(dst + k) = h4;
            }
        } else if (vec_ok) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int k = lane * 4 + i * 256;
                v0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nv && k_lo + k < a.cin) v0[i] = *reinterpret_cast<const float4*>(xr + k);
            }
            stage_row(v0, dst);
        } else {
            float mean = 0.0f, rstd = 1.0f;
            if (a.ln_gamma) {
                float s = 0.0f;
                for (int k = lane; k < a.cin; k += 64) s += xr[k];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
                mean = s / (float)a.cin;
                float vv = 0.0f;
                for (int k = lane; k < a.cin; k += 64) {
                    const float d = xr[k] - mean;
                    vv += d * d;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) vv += __shfl_xor(vv, off, 64);
                rstd = rsqrtf(vv / (float)a.cin + a.ln_eps);
            }
            for (int k = lane; k < ktot; k += 64) {
                float v = 0.0f;
                if (k < a.cin) {
                    v = xr[k];
                    if (a.ln_gamma) v = (v - mean) * rstd * a.ln_gamma[k] + a.ln_beta[k];
                }
                dst[k] = (_Float16)v;
            }
        }
    }
    __syncthreads();

    float4v acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.0f;
    const _Float16* arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int mr = t * 16 + c;
        if (mr >= M) mr = M - 1;
        arow[t] = sx + (size_t)mr * xs + g * 16;
    }
    // (3) MFMAs; further passes only when a wave owns more than SK_LINES lines (K > 4096)
    for (int pass = 0;; ++pass) {
#pragma unroll
        for (int i = 0; i < SK_LINES; ++i) {
            const int line = wid + (pass * SK_LINES + i) * SK_WAVES;
            if (line < lines) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const half8 fa0 = *reinterpret_cast<const half8*>(arow[t] + line * 64);
                    const half8 fa1 = *reinterpret_cast<const half8*>(arow[t] + line * 64 + 8);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, fb[i][0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa1, fb[i][1], acc[t], 0, 0, 0);
                }
            }
        }
        if (wid + (pass + 1) * SK_LINES * SK_WAVES >= lines) break;
#pragma unroll
        for (int i = 0; i < SK_LINES; ++i) {
            const int line = wid + ((pass + 1) * SK_LINES + i) * SK_WAVES;
            if (line < lines) {
                fb[i][0] = *reinterpret_cast<const half8*>(wrow + line * 64);
                fb[i][1] = *reinterpret_cast<const half8*>(wrow + line * 64 + 8);
            }
        }
    }
    // (4) reduction + epilogue
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[((wid * MT + t) * 4 + e) * 64 + lane] = acc[t][e];
    __syncthreads();
    // MT*256 outputs <= 512 threads: one element each (threads past that only take part in the barriers)
    const bool owner = tid < MT * 256;
    const int o = tid;
    const int t = o / 256, e = (o >> 6) & 3, ln = o & 63;
    float v = 0.0f;
    if (owner) {
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) v += red[((w * MT + t) * 4 + e) * 64 + ln];
    }
    if (KS > 1) {
        // split-K: publish this slice's partial sums; the block that arrives LAST at the column block's counter adds the
        // KS slices in slice order (a fixed order: the result does not depend on which block that is) and runs the
        // epilogue.  No block ever waits for another one.
        // The XCDs' L2 caches are not coherent with each other, and this is ordinary (coarse-grained) device memory:
        // an sc1 load may still hit a stale line in the reader's own L2 (seen as wrong tokens once two decode chains ran
        // concurrently and workgroups stopped landing on the same XCD every launch).  So: partial sums are written
        // through (sc1 stores) and acknowledged before the barrier; ONE wave per block then bumps the arrival counter,
        // and only the last block -- one per column block -- pays an agent-scope acquire (buffer_inv: its XCD's L2 and
        // this CU's L1) before the partials are read back.  A __threadfence() in every wave of every block (write-back +
        // invalidate x 2048 per launch, under the weight stream) measured 2x slower than no split at all.
        float* part = a.sk_part + (size_t)blockIdx.x * KS * (MT * 256);
        if (owner) __hip_atomic_store(part + (size_t)blockIdx.y * (MT * 256) + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Every storing wave drains its own sc1 stores BEFORE the barrier: s_barrier alone does not wait for vmcnt, and the
        // compiler emits no wait here (checked in the ISA: store -> s_barrier -> atomic).  Inline asm so no pass can drop it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // every wave's partial sums are acknowledged; `red` reads done
        int* s_last = reinterpret_cast<int*>(red);
        if (tid == 0) {
            // Release side: the partial sums are sc1 (write-through) stores already acknowledged by every wave of the block
            // (the explicit s_waitcnt vmcnt(0) before the barrier above) -- what an agent-scope release fence adds on top, buffer_wbl2 for
            // dirty L2 lines, has nothing of ours to write back and costs 4 ms per 250-step decode.
#ifdef SPLITK_RELEASE_FENCE
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
            const unsigned old = __hip_atomic_fetch_add(&a.sk_cnt[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old == (unsigned)(KS - 1);
            if (last) {
                __hip_atomic_store(&a.sk_cnt[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *s_last = last;
        }
        __syncthreads();
        if (!*s_last) return;
        v = 0.0f;
        if (owner)
            for (int k = 0; k < KS; ++k) v += __hip_atomic_load(part + (size_t)k * (MT * 256) + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (owner) {
        const int n = n0 + (ln & 15);
        const int m = t * 16 + (ln >> 4) * 4 + e;
        if (n < a.n && m < M) {
            v = apply_act(v + e_bias, a.act, a.slope) * a.alpha;
            if (a.row_scale) v *= a.row_scale[m];
            v += e_res;
            if (a.out2 && n >= a.n_split) {
                if (a.out2_f16)
                    reinterpret_cast<_Float16*>(a.out2)[(int64_t)m * a.ldc2 + (n - a.n_split)] = (_Float16)v;
                else
                    a.out2[(int64_t)m * a.ldc2 + (n - a.n_split)] = v;
            } else {
                a.out[(int64_t)m * a.ldc + n] = v;
            }
        }
    }
}

// gemm_rows: 33 .. 256 rows (the wide decode engine's projections: every row of a big decode batch through ONE launch per
// projection).  At these sizes a GEMM is a latency problem, not a flop problem (128 x 1024 x 1024 = 0.1 us of MFMA, 2 MB of weights):
// the 64 x 64 ring tile leaves 32 workgroups walking K serially (11.8 us per launch measured in the engine).  Here a workgroup owns
// (16 RT) rows x (16 CT) columns, its 8 waves take the 64-element K lines round robin, and EVERY operand fragment a wave needs
// (PL lines x (RT + CT) x 32 bytes per lane) is requested before the first MFMA: one memory round trip, then
// v_mfma_f32_16x16x32_f16 on registers, an LDS reduction over the waves in wave order (a fixed order: results do not depend on
// timing) and the epilogue.  A row's sums do not depend on the other rows of the launch.  Column tiles are the fast block index:
// workgroups that share a weight tile differ by a multiple of the tile count and land on the same XCD's L2 when that is a multiple of 8.
// Columns >= n_split may go to a second destination (a.out2: the K | V half of a q | k | v projection lands in the KV cache as fp16).
// (A LayerNorm prologue -- fp32 rows normalised in registers while the weights are in flight -- was built, parity-green and slower than
// a LayerNorm launch of its own: fp32 rows double the activation bytes every column tile re-reads.  EXPERIMENTS.md M.)
template <int RT, int CT, int PL, bool A32>
__global__ __launch_bounds__(512) void gemm_rows(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gr_red[];    // [8][RT * CT][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int ncol = (a.n + 16 * CT - 1) / (16 * CT);
    const int n0 = (int)(blockIdx.x % (unsigned)ncol) * 16 * CT, m0 = (int)(blockIdx.x / (unsigned)ncol) * 16 * RT;
    const int M = (int)a.m;
    const int lines = a.cin_pad >> 6;
    const _Float16* wrow[CT];
#pragma unroll
    for (int u = 0; u < CT; ++u) wrow[u] = a.w + (int64_t)min(n0 + 16 * u + c, a.n - 1) * a.cin_pad + g * 16;
    const char* arow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
        arow[t] = reinterpret_cast<const char*>(a.x) + ((int64_t)min(m0 + 16 * t + c, M - 1) * a.lda + g * 16) * (A32 ? 4 : 2);
    float4v acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][u][e] = 0.0f;
    // epilogue operands of this thread's outputs: requested behind the first pass's fragments, consumed last
    constexpr int NOUT = RT * CT / 2;                        // outputs per thread (RT * CT * 256 over 512 threads)
    float e_bias[NOUT], e_res[NOUT];
    for (int pass = 0; pass * PL * 8 < lines; ++pass) {
        half8 fb[CT][PL][2];
        half8 fa[A32 ? 1 : RT][A32 ? 1 : PL][2];
        float4 fx[A32 ? RT : 1][A32 ? PL : 1][4];
        // Unconditional loads from a clamped line, zeroed afterwards when the line does not exist: `if (on) x = load` compiles to a
        // branch per load with an s_waitcnt between them (EXPERIMENTS.md G) -- the whole point here is ONE batch of loads.
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int line = wid + (pass * PL + i) * 8;
            const int lc = line < lines ? line : 0;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if constexpr (A32) {
                    const float* p = reinterpret_cast<const float*>(arow[t]) + lc * 64;
#pragma unroll
                    for (int j = 0; j < 4; ++j) fx[t][i][j] = *reinterpret_cast<const float4*>(p + 4 * j);
                } else {
                    const _Float16* p = reinterpret_cast<const _Float16*>(arow[t]) + lc * 64;
                    fa[t][i][0] = *reinterpret_cast<const half8*>(p);
                    fa[t][i][1] = *reinterpret_cast<const half8*>(p + 8);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int line = wid + (pass * PL + i) * 8;
            const int lc = line < lines ? line : 0;
#pragma unroll
            for (int u = 0; u < CT; ++u) {
                fb[u][i][0] = *reinterpret_cast<const half8*>(wrow[u] + lc * 64);
                fb[u][i][1] = *reinterpret_cast<const half8*>(wrow[u] + lc * 64 + 8);
            }
        }
        if (pass == 0) {
            const float* bp = a.bias ? a.bias : reinterpret_cast<const float*>(a.w);          // (a valid address either way; the value is
            const float* rp = a.residual ? a.residual : reinterpret_cast<const float*>(a.w);  // dropped below when the operand is absent)
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                const int o = tid + j * 512;
                const int tile = o >> 8, e = (o >> 6) & 3, ln = o & 63;
                const int n = min(n0 + 16 * (tile % CT) + (ln & 15), a.n - 1), m = min(m0 + 16 * (tile / CT) + (ln >> 4) * 4 + e, M - 1);
                e_bias[j] = bp[a.bias ? n : 0];
                e_res[j] = rp[a.residual ? (int64_t)m * a.ldr + n : 0];
            }
        }
        if (pass * PL * 8 + PL * 8 > lines) {       // a ragged last pass: lines that do not exist count as zeros (the B side suffices)
#pragma unroll
            for (int i = 0; i < PL; ++i)
                if (wid + (pass * PL + i) * 8 >= lines) {
#pragma unroll
                    for (int u = 0; u < CT; ++u)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int q = 0; q < 8; ++q) fb[u][i][j][q] = (_Float16)0.0f;
                }
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                half8 x0, x1;
                if constexpr (A32) {
                    float4 v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = fx[t][i][j];
                    }
                    x0[0] = (_Float16)v[0].x; x0[1] = (_Float16)v[0].y; x0[2] = (_Float16)v[0].z; x0[3] = (_Float16)v[0].w;
                    x0[4] = (_Float16)v[1].x; x0[5] = (_Float16)v[1].y; x0[6] = (_Float16)v[1].z; x0[7] = (_Float16)v[1].w;
                    x1[0] = (_Float16)v[2].x; x1[1] = (_Float16)v[2].y; x1[2] = (_Float16)v[2].z; x1[3] = (_Float16)v[2].w;
                    x1[4] = (_Float16)v[3].x; x1[5] = (_Float16)v[3].y; x1[6] = (_Float16)v[3].z; x1[7] = (_Float16)v[3].w;
                } else {
                    x0 = fa[t][i][0];
                    x1 = fa[t][i][1];
                }
#pragma unroll
                for (int u = 0; u < CT; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x0, fb[u][i][0], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, fb[u][i][1], acc[t][u], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) gr_red[((wid * (RT * CT) + t * CT + u) * 4 + e) * 64 + lane] = acc[t][u][e];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        const int o = tid + j * 512;
        const int tile = o >> 8, e = (o >> 6) & 3, ln = o & 63;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += gr_red[((w * (RT * CT) + tile) * 4 + e) * 64 + ln];
        const int n = n0 + 16 * (tile % CT) + (ln & 15), m = m0 + 16 * (tile / CT) + (ln >> 4) * 4 + e;
        if (n < a.n && m < M) {
            v = apply_act(v + (a.bias ? e_bias[j] : 0.0f), a.act, a.slope) * a.alpha + (a.residual ? e_res[j] : 0.0f);
            if (a.out2 && n >= a.n_split) {
                if (a.out2_f16) reinterpret_cast<_Float16*>(a.out2)[(int64_t)m * a.ldc2 + (n - a.n_split)] = (_Float16)v;
                else a.out2[(int64_t)m * a.ldc2 + (n - a.n_split)] = v;
            } else if (a.out_f16) {
                reinterpret_cast<_Float16*>(a.out)[(int64_t)m * a.ldc + n] = (_Float16)v;
            } else {
                a.out[(int64_t)m * a.ldc + n] = v;
            }
        }
    }
}

template <int RT, int CT, int PL, bool A32>
static void launch_rows(const GemmArgs& a, hipStream_t st) {
    static std::once_flag attr;
    const size_t lds = (size_t)8 * RT * CT * 4 * 64 * sizeof(float);
    std::call_once(attr, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows<RT, CT, PL, A32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    const unsigned grid = (unsigned)(cdiv(a.m, 16 * RT) * cdiv(a.n, 16 * CT));
    hipLaunchKernelGGL((gemm_rows<RT, CT, PL, A32>), dim3(grid), dim3(512), lds, st, a);
}

// picks the workgroup tile of gemm_rows: the fattest one (fewest re-reads of the activation rows) that still gives every CU a workgroup
template <int PL, bool A32>
static void launch_rows_pl(const GemmArgs& a, hipStream_t st) {
    auto blocks = [&](int rt, int ct) { return cdiv(a.m, 16 * rt) * cdiv(a.n, 16 * ct); };
    if constexpr (PL <= 2) {
        if (blocks(4, 2) >= 256) return launch_rows<4, 2, PL, A32>(a, st);
    }
    if constexpr (PL <= 4) {
        if (blocks(2, 2) >= 256) return launch_rows<2, 2, PL, A32>(a, st);
    }
    launch_rows<2, 1, PL, A32>(a, st);
}

static size_t skinny_lds_bytes(int m, int cin_pad, int mt) {
    return (((size_t)m * (cin_pad + 8) * 2 + 15) & ~(size_t)15) + (size_t)SK_WAVES * mt * 4 * 64 * sizeof(float);
}

// fp32 [n, taps, cin] (conv weight already permuted so that cin is innermost) -> fp16 [n_pad, taps, cin_pad]
__global__ void pack_weight_f16(const float* __restrict__ src, _Float16* __restrict__ dst, int n, int taps,
                                int cin, int n_pad, int cin_pad) {
    const int64_t total = (int64_t)n_pad * taps * cin_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cin_pad);
        const int64_t rest = i / cin_pad;
        const int tp = (int)(rest % taps);
        const int nn = (int)(rest / taps);
        float v = 0.0f;
        if (nn < n && c < cin) v = src[((int64_t)nn * taps + tp) * cin + c];
        dst[i] = (_Float16)v;
    }
}

template <int WM, int WN, int TM, int TN, int BKT>
static void launch_tile(const GemmArgs& a, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const dim3 grid((unsigned)cdiv(a.m, BM), (unsigned)cdiv(a.n, BN));
    if (a.x_f16)
        hipLaunchKernelGGL((gemm_tile<WM, WN, TM, TN, true, BKT>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((gemm_tile<WM, WN, TM, TN, false, BKT>), grid, dim3(256), 0, st, a);
}

static int g_ring_mode_override = -1;   // -1: rule below / ASTTS_GEMM_RING; 0: ring kernel off; 1..3: force that tile

static int launch_gemm(const GemmArgs& a, hipStream_t st) {
    const bool plain = a.taps == 1 && a.stride == 1 && a.pad == 0 && a.t_in == a.t_out && !a.in_lens;   // (row lengths: the tile kernel masks them)
    if (a.m <= 32 && plain && !a.x_f16 && !a.out_f16) {
        // the block keeps its m rows of x as an fp16 image in LDS: rows are taken in chunks that fit 160 KB
        // (only m > 16 with K > 2048 needs two passes, e.g. the FFN-out projection of a 32-row decode group)
        // split-K (a.sk_part given): deep, narrow GEMMs (the FFN-out projection: K = 4096 onto 1024 columns = 64 column
        // blocks) are cut into 4 K slices so that 256 workgroups stream the weights instead of 64
        const int lines_tot = a.cin_pad >> 6;
        int ksplit = 1;
        static const bool nosplit_env = getenv("ASTTS_NO_SPLITK") != nullptr;
        if (a.sk_part && !nosplit_env && !a.gather && !a.ln_gamma && lines_tot >= 32 && lines_tot % 32 == 0 && (a.n + 15) / 16 <= 128) ksplit = 4;
        const int kslice = a.cin_pad / ksplit;
        int rows = (int)a.m;
        while (rows > 1 && skinny_lds_bytes(rows, kslice, rows <= 16 ? 1 : 2) > 160 * 1024) rows = rows > 16 ? 16 : rows / 2;
        if (skinny_lds_bytes(rows, kslice, 1) > 160 * 1024) {
            set_error("astts_op_gemm: cin_pad=%d does not fit the skinny kernel's LDS image", a.cin_pad);
            return ASTTS_ERR_INVALID;
        }
        static bool attr_set = false;
        if (!attr_set) {  // allow > 64 KiB of dynamic LDS (one-off, outside any captured region in practice)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny16<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny16<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny16<1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny16<2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        const dim3 grid((a.n + 15) / 16, ksplit);
        const bool small_k = kslice <= 1024;   // row of <= 4 float4 per lane: a quarter of the staging registers
        for (int r0 = 0; r0 < (int)a.m; r0 += rows) {
            GemmArgs c = a;
            c.ksplit = ksplit;
            c.m = (int)a.m - r0 < rows ? (int)a.m - r0 : rows;
            if (c.gather) c.gather += r0; else c.x += (int64_t)r0 * a.lda;
            if (c.residual) c.residual += (int64_t)r0 * a.ldr;
            if (c.row_scale) c.row_scale += r0;
            c.out += (int64_t)r0 * a.ldc;
            if (c.out2) c.out2 = a.out2_f16 ? (float*)((_Float16*)a.out2 + (int64_t)r0 * a.ldc2) : a.out2 + (int64_t)r0 * a.ldc2;
            const int mt = c.m <= 16 ? 1 : 2;
            const size_t lds = skinny_lds_bytes((int)c.m, kslice, mt);
            const bool prof = prof_begin(ASTTS_PROF_GEMM_SKINNY, st, (double)a.n * a.cin_pad * 2.0);
            if (mt == 1 && small_k)
                hipLaunchKernelGGL((gemm_skinny16<1, 4>), grid, dim3(512), lds, st, c);
            else if (mt == 1)
                hipLaunchKernelGGL((gemm_skinny16<1, 16>), grid, dim3(512), lds, st, c);
            else if (small_k)
                hipLaunchKernelGGL((gemm_skinny16<2, 4>), grid, dim3(512), lds, st, c);
            else
                hipLaunchKernelGGL((gemm_skinny16<2, 16>), grid, dim3(512), lds, st, c);
            if (prof) prof_end(ASTTS_PROF_GEMM_SKINNY, st);
        }
        ASTTS_CHECK_LAUNCH();
        return ASTTS_OK;
    }
    ASTTS_REQUIRE(!a.gather && !a.ln_gamma && !a.out2, ASTTS_ERR_INVALID,
                  "astts_op_gemm_fused: gather / LayerNorm / split output need m <= 32 and a plain (non-conv) GEMM");
    const bool prof = prof_begin(ASTTS_PROF_GEMM_TILE, st, 2.0 * (double)a.m * a.n * a.cin * a.taps);
    auto blocks = [&](int bm, int bn) { return cdiv(a.m, bm) * cdiv(a.n, bn); };
    // fp16 activations, plain GEMM, whole 64-wide K tiles: the LDS-DMA ring kernel
    static const int ring_env0 = [] { const char* e = getenv("ASTTS_GEMM_RING"); return e ? atoi(e) : -1; }();
    const int ring_env = g_ring_mode_override >= 0 ? g_ring_mode_override : ring_env0;   // astts_op_gemm_set_ring_mode (tests)
    if (plain && a.x_f16 && a.cin == a.cin_pad && (a.lda & 7) == 0 && ((uintptr_t)a.x & 15) == 0 && a.m >= 64 && a.n > 32 &&
        ring_env != 0) {
        static bool ring_attr = false;
        if (!ring_attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring<2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring<1, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            ring_attr = true;
        }
        // Tile choice, measured IN SITU (scripts/flow_only.py; the back-to-back micro-benchmark runs on L2-hot weights and ranks
        // the deep-K tiles differently).  Wide outputs (K = 256: four K tiles) take two-stage rings -- 48 KB of LDS puts three
        // blocks on a CU and the overlap of one block's epilogue with its neighbours' loads beats a deeper ring (16 -> 12 us);
        // the n <= 256 projections (K = 512 / 1024 from cold weights) keep the 4-stage 64x64 ring.
        //   1: 128x128 x2 (once there are >= 4-8 tiles per CU)   2: 128x64 x2 (wide outputs)   3: 64x64 x4 (n <= 256)   4: 256x256 x2, 8 waves
        const int64_t b128 = blocks(128, 128);
        int mode;
        // (deep K with enough tiles -- the kNN scan as a GEMM, K = 6144: 128 x 128 measured 727 us per 256-query search against 797 for
        // 128 x 64 and 906 for 64 x 64, profiles/r05_knn_gemm_ab.log)
        // 4: 256 x 256 with eight waves (two stages = 128 KB, one block per CU).  The ring kernels are bound by the bytes a CU can keep in
        // flight (latency x rate, against the LDS left beside the tile being multiplied): 128 x 128 holds 2 blocks x 32 KB for 2.1 MFLOP
        // each, 256 x 256 holds 64 KB for 8.4 MFLOP.  scripts/ring_shapes.py, TFLOP/s 128 x 128 -> 256 x 256: 15 360 x 3072 x 5120 (the
        // embedder's q|k|v) 807 -> 1025, 15 360 x 8192 x 3072 832 -> 1037, 23 680 x 1024 x 4096 700 -> 916, the kNN scan 256 x 6144 x 100k
        // 639 -> 796; needs more than a round of blocks (86 blocks: 349 against 548) and deep K (the flow's K = 256 projections at 44 032
        // rows: 532 against 463 back to back, but the 64-sequence flow solve measured 123.8 ms with it and 121.0 without).
        // Round 6: the rule's 256 x 256 tile runs on the eight-phase schedule (gemm_ring8: bit-identical results, +8-10 % -- alternating
        // A/B in one process, scripts/ring_ab.py, profiles/r06_ring8_ab.log: 8192^3 1066 -> 1171, 15 360 x 3072 x 5120 1005 -> 1087,
        // the kNN scan 783 -> 861 TFLOP/s on N(0, 1) operands; on zero-filled operands 1425 -> 1667, i.e. what is left is the clock the
        // chip holds under fp16 MFMA load on real data (1.71 GHz measured, GRBM_GUI_ACTIVE), not the schedule).
        // Round 6, with the eight-phase kernel and the grouped tile order (alternating A/B of tiles 1 / 2 / 5 on 17 embedder, prefill and
        // frontend shapes, profiles/r06_ring_rule_ab.log): what decides for the 256 x 256 tile is how full its LAST round of 256 blocks is
        // -- 180 or 235 blocks (one round, 0.70 / 0.92 full) win by 10-19 % at K >= 3072, 264, 300 or 516 blocks (0.52 - 0.67) lose by 7-47 % --
        // and shallow K needs a fuller round (K = 1024-1280: wins at 0.92, even at 0.87, loses at 0.70); below it 128 x 128 beats 128 x 64
        // on every deep-K shape with >= 256 tiles (the K = 256 projections of the flow keep their in-situ rule).
        const int64_t b256 = blocks(256, 256);
        const int64_t fill_pct = b256 * 100 / (cdiv(b256, 256) * 256);          // how full the rounds of 256 x 256 tiles are
        if (a.n > 128 && ((a.cin_pad >= 2048 && fill_pct >= 68) || (a.cin_pad >= 1024 && fill_pct >= 85))) mode = 4;
        else if (b128 >= (a.n >= 512 ? 2048 : 1024) || (a.cin_pad >= 4096 && b128 >= 512) || (a.cin_pad >= 1024 && a.n >= 512 && b128 >= 256)) mode = 1;
        else if (a.n >= 512 && blocks(128, 64) >= 160) mode = 2;
        else mode = 3;
        if (mode == 4) mode = 5;            // the rule's 256 x 256 tile is the eight-phase kernel; a FORCED 4 keeps the one-barrier form (A/B, tests)
        if (ring_env > 0) mode = ring_env;
        if (a.blockmax && mode != 4 && mode != 5) mode = 1;     // (the block maxima are per 64 columns of a wave: the tiles with TN = 2)
        if (mode == 5 && a.n > 128) {       // 256 x 256 on the eight-phase schedule (gemm_ring8)
            static std::once_flag attr85;
            std::call_once(attr85, [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring8), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128);
            });
            hipLaunchKernelGGL(gemm_ring8, dim3((unsigned)(cdiv(a.m, 256) * cdiv(a.n, 256))), dim3(512), 2 * 512 * 128, st, a);
        } else
        if (mode == 4 && a.n > 128) {       // 256 x 256, eight waves, two stages (128 KB): twice the flops per byte in flight
            static std::once_flag attr8;
            std::call_once(attr8, [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring<4, 2, 2, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128);
            });
            hipLaunchKernelGGL((gemm_ring<4, 2, 2, 4, 8>), dim3((unsigned)(cdiv(a.m, 256) * cdiv(a.n, 256))), dim3(512), 2 * 512 * 128, st, a);
        } else if ((mode == 1 || mode == 4) && a.n > 64) {
            hipLaunchKernelGGL((gemm_ring<2, 2, 2>), dim3((unsigned)(cdiv(a.m, 128) * cdiv(a.n, 128))), dim3(256), 2 * 256 * 128, st, a);
        } else if (mode == 2 || mode == 1) {
            hipLaunchKernelGGL((gemm_ring<2, 1, 2>), dim3((unsigned)(cdiv(a.m, 128) * cdiv(a.n, 64))), dim3(256), 2 * 192 * 128, st, a);
        } else {
            hipLaunchKernelGGL((gemm_ring<1, 1, 4>), dim3((unsigned)(cdiv(a.m, 64) * cdiv(a.n, 64))), dim3(256), 4 * 128 * 128, st, a);
        }
        if (prof) prof_end(ASTTS_PROF_GEMM_TILE, st);
        ASTTS_CHECK_LAUNCH();
        return ASTTS_OK;
    }
    if (a.blockmax) {          // (gemm_scan checks the shape; what is left is a ring path switched off: astts_op_gemm_set_ring_mode(0))
        set_error("gemm_scan: the block-maximum epilogue needs the ring kernels");
        return ASTTS_ERR_INVALID;
    }
    const int64_t want = 384;  // >= 1.5 blocks per CU
    // K tile: cin_pad is a multiple of 64, so 64 never straddles a tap; 128 needs cin_pad % 128 == 0
    const bool k128 = (a.cin_pad % 128) == 0 && a.taps * a.cin_pad >= 256;
    if (a.n <= 32) {
        launch_tile<4, 1, 1, 1, 64>(a, st);
    } else if (a.n > 64 && blocks(128, 128) >= want) {
        launch_tile<2, 2, 2, 2, 32>(a, st);   // measured: the 128x128 tile is fastest at BK=32 (2+ blocks/CU, small K)
    } else if (blocks(128, 64) >= want) {
        launch_tile<2, 2, 2, 1, 32>(a, st);
    } else if (k128) {
        launch_tile<2, 2, 1, 1, 128>(a, st);
    } else {
        launch_tile<2, 2, 1, 1, 64>(a, st);
    }
    if (prof) prof_end(ASTTS_PROF_GEMM_TILE, st);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

// the kNN scan as one GEMM (knn.hip): scores[q][n] = <query q, bank row n>, fp16 operands, fp32 out [qg][ldc]; row panels of one bank
// tile first (GemmArgs::m_first), so that the bank is fetched from HBM once however many 128-query panels there are
int gemm_scan(const _Float16* queries, const _Float16* bank, float* out, int32_t qg, int64_t n, int32_t dp, int32_t ldc, hipStream_t st,
              const float* col_scale, const float* col_bias, const float* row_qs, float* blockmax, int32_t bm_ld) {
    GemmArgs a{(const float*)queries, bank, nullptr, nullptr, nullptr, out, qg, (int)n, dp, dp, 1,
               dp, ldc, 0, qg, qg, 1, 1, 0, ASTTS_ACT_NONE, 1.0f, 0.1f,
               nullptr, nullptr, nullptr, 0.0f, nullptr, 0, 0, 1, 0, 0};
    a.m_first = 1;
    if (blockmax) {
        // (only the ring kernels carry that epilogue: their launch conditions, restated -- the caller takes the plain scan otherwise)
        if (!(qg >= 64 && n > 64 && (dp & 63) == 0 && ((uintptr_t)queries & 15) == 0 && col_scale && (!col_bias || row_qs))) {
            set_error("gemm_scan: shape outside the ring kernels (qg=%d n=%lld dp=%d)", qg, (long long)n, dp);
            return ASTTS_ERR_INVALID;
        }
        a.col_scale = col_scale;
        a.col_bias = col_bias;
        a.row_qs = row_qs;
        a.blockmax = blockmax;
        a.bm_ld = bm_ld;
    }
    return launch_gemm(a, st);
}

}  // namespace astts

using namespace astts;

static int check_gemm_args(const char* who, const float* x, const void* w, float* out, int64_t m, int32_t n, int32_t cin,
                           int32_t cin_pad, int32_t taps, int32_t t_in, int32_t t_out, int32_t stride, int32_t dil, int32_t act) {
    ASTTS_REQUIRE(x && w && out, ASTTS_ERR_INVALID, "%s: null pointer", who);
    ASTTS_REQUIRE((const void*)x != (const void*)out, ASTTS_ERR_INVALID,
                  "%s: out aliases x (every workgroup reads whole input rows: an in-place GEMM races)", who);
    ASTTS_REQUIRE(m >= 1 && n >= 1 && cin >= 1 && taps >= 1, ASTTS_ERR_INVALID, "%s: bad shape m=%lld n=%d cin=%d taps=%d",
                  who, (long long)m, n, cin, taps);
    ASTTS_REQUIRE(cin_pad >= cin && cin_pad % 64 == 0, ASTTS_ERR_INVALID,
                  "%s: cin_pad=%d must be a multiple of 64 and >= cin=%d", who, cin_pad, cin);
    ASTTS_REQUIRE(t_in >= 1 && t_out >= 1 && m % t_out == 0 && stride >= 1 && dil >= 1, ASTTS_ERR_INVALID,
                  "%s: bad conv geometry t_in=%d t_out=%d stride=%d dil=%d", who, t_in, t_out, stride, dil);
    ASTTS_REQUIRE(act >= ACT_NONE && act <= ACT_LEAKY, ASTTS_ERR_INVALID, "%s: act=%d", who, act);
    return ASTTS_OK;
}

extern "C" {

int astts_op_gemm_set_ring_mode(int32_t mode) {
    ASTTS_REQUIRE(mode >= -1 && mode <= 5, ASTTS_ERR_INVALID, "astts_op_gemm_set_ring_mode: mode=%d (-1 auto, 0 off, 1..5 tile)", mode);
    g_ring_mode_override = mode;
    return ASTTS_OK;
}

int astts_op_pack_weight(const float* src, void* dst_f16, int32_t n, int32_t taps, int32_t cin,
                         int32_t n_pad, int32_t cin_pad, astts_stream_t stream) {
    ASTTS_REQUIRE(src && dst_f16, ASTTS_ERR_INVALID, "astts_op_pack_weight: null pointer");
    ASTTS_REQUIRE(n >= 1 && taps >= 1 && cin >= 1 && n_pad >= n && cin_pad >= cin, ASTTS_ERR_INVALID,
                  "astts_op_pack_weight: bad shape n=%d taps=%d cin=%d n_pad=%d cin_pad=%d", n, taps, cin, n_pad, cin_pad);
    const int64_t total = (int64_t)n_pad * taps * cin_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_f16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                       (_Float16*)dst_f16, n, taps, cin, n_pad, cin_pad);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_gemm(const float* x, const void* w_f16, const float* bias, const float* residual,
                  const float* row_scale, float* out, int64_t m, int32_t n, int32_t cin, int32_t cin_pad,
                  int32_t taps, int32_t lda, int32_t ldc, int32_t ldr, int32_t t_in, int32_t t_out,
                  int32_t stride, int32_t dil, int32_t pad, int32_t act, float alpha, float slope,
                  astts_stream_t stream) {
    const int rc = check_gemm_args("astts_op_gemm", x, w_f16, out, m, n, cin, cin_pad, taps, t_in, t_out, stride, dil, act);
    if (rc != ASTTS_OK) return rc;
    GemmArgs a{x, (const _Float16*)w_f16, bias, residual, row_scale, out, m, n, cin, cin_pad, taps,
               lda, ldc, ldr, t_in, t_out, stride, dil, pad, act, alpha, slope,
               nullptr, nullptr, nullptr, 0.0f, nullptr, 0, 0, 0, 0, 0};
    return launch_gemm(a, (hipStream_t)stream);
}

int astts_op_gemm_ex(const void* x, int32_t x_f16, const void* w_f16, const float* bias, const float* residual,
                     const float* row_scale, void* out, int32_t out_f16, int64_t m, int32_t n, int32_t cin, int32_t cin_pad,
                     int32_t taps, int32_t lda, int32_t ldc, int32_t ldr, int32_t t_in, int32_t t_out,
                     int32_t stride, int32_t dil, int32_t pad, int32_t act, float alpha, float slope,
                     astts_stream_t stream) {
    const int rc = check_gemm_args("astts_op_gemm_ex", (const float*)x, w_f16, (float*)out, m, n, cin, cin_pad, taps, t_in, t_out, stride, dil, act);
    if (rc != ASTTS_OK) return rc;
    GemmArgs a{(const float*)x, (const _Float16*)w_f16, bias, residual, row_scale, (float*)out, m, n, cin, cin_pad, taps,
               lda, ldc, ldr, t_in, t_out, stride, dil, pad, act, alpha, slope,
               nullptr, nullptr, nullptr, 0.0f, nullptr, 0, 0, x_f16 ? 1 : 0, out_f16 ? 1 : 0, 0};
    return launch_gemm(a, (hipStream_t)stream);
}

int astts_op_gemm_lens(const void* x, int32_t x_f16, const void* w_f16, const float* bias, const float* residual,
                       const float* row_scale, void* out, int32_t out_f16, int64_t m, int32_t n, int32_t cin, int32_t cin_pad,
                       int32_t taps, int32_t lda, int32_t ldc, int32_t ldr, int32_t t_in, int32_t t_out,
                       int32_t stride, int32_t dil, int32_t pad, int32_t act, float alpha, float slope, const int32_t* in_lens,
                       astts_stream_t stream) {
    const int rc = check_gemm_args("astts_op_gemm_lens", (const float*)x, w_f16, (float*)out, m, n, cin, cin_pad, taps, t_in, t_out, stride, dil, act);
    if (rc != ASTTS_OK) return rc;
    GemmArgs a{(const float*)x, (const _Float16*)w_f16, bias, residual, row_scale, (float*)out, m, n, cin, cin_pad, taps,
               lda, ldc, ldr, t_in, t_out, stride, dil, pad, act, alpha, slope,
               nullptr, nullptr, nullptr, 0.0f, nullptr, 0, 0, x_f16 ? 1 : 0, out_f16 ? 1 : 0, 0};
    a.in_lens = in_lens;
    return launch_gemm(a, (hipStream_t)stream);
}

int astts_op_gemm_rows(const void* x, int32_t x_f16, const void* w_f16,
                       const float* bias, const float* residual, void* out, int32_t out_f16, void* out2, int32_t out2_f16, int32_t m, int32_t n,
                       int32_t n_split, int32_t k, int32_t lda, int32_t ldc, int32_t ldc2, int32_t ldr, int32_t act, astts_stream_t stream) {
    ASTTS_REQUIRE(x && w_f16 && out, ASTTS_ERR_INVALID, "astts_op_gemm_rows: null pointer");
    ASTTS_REQUIRE(m >= 1 && m <= 4096 && n >= 1 && k >= 64 && (k % 64) == 0, ASTTS_ERR_UNSUPPORTED,
                  "astts_op_gemm_rows: m=%d n=%d k=%d (1 <= m <= 4096 rows, k a multiple of 64)", m, n, k);
    ASTTS_REQUIRE((lda % (x_f16 ? 8 : 4)) == 0 && (((uintptr_t)x | (uintptr_t)w_f16) & 15) == 0 && lda >= k && ldc >= (out2 ? n_split : n) &&
                      (!residual || ldr >= n),
                  ASTTS_ERR_INVALID, "astts_op_gemm_rows: operands must be 16-byte aligned (lda=%d), ld* >= the row's width", lda);
    ASTTS_REQUIRE(!out2 || (n_split >= 1 && n_split < n && ldc2 >= n - n_split), ASTTS_ERR_INVALID,
                  "astts_op_gemm_rows: split output needs 1 <= n_split=%d < n=%d and ldc2=%d >= n - n_split", n_split, n, ldc2);
    ASTTS_REQUIRE(act >= ACT_NONE && act <= ACT_LEAKY, ASTTS_ERR_INVALID, "astts_op_gemm_rows: act=%d", act);
    GemmArgs a{(const float*)x, (const _Float16*)w_f16, bias, residual, nullptr, (float*)out, m, n, k, k, 1,
               lda, ldc, ldr, m, m, 1, 1, 0, act, 1.0f, 0.1f,
               nullptr, nullptr, nullptr, 0.0f, (float*)out2, ldc2, n_split, x_f16 ? 1 : 0, out_f16 ? 1 : 0, out2_f16 ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const bool prof = prof_begin(ASTTS_PROF_GEMM_TILE, st, 2.0 * (double)m * n * k);
    const int per_wave = (k / 64 + 7) / 8;      // K lines per wave: one pass when the fragments fit the registers
    if (x_f16) {
        if (per_wave <= 2) launch_rows_pl<2, false>(a, st);
        else if (per_wave <= 4) launch_rows_pl<4, false>(a, st);
        else launch_rows_pl<8, false>(a, st);
    } else {
        if (per_wave <= 2) launch_rows_pl<2, true>(a, st);
        else launch_rows_pl<4, true>(a, st);
    }
    if (prof) prof_end(ASTTS_PROF_GEMM_TILE, st);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_gemm_ln(const void* x_f16, const void* w_f16, const float* bias, const float* residual, float* out,
                     const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out_f16, int64_t m, int32_t n, int32_t cin,
                     int32_t cin_pad, int32_t lda, int32_t ldc, int32_t ldr, int32_t ld_ln, astts_stream_t stream) {
    const int rc = check_gemm_args("astts_op_gemm_ln", (const float*)x_f16, w_f16, out, m, n, cin, cin_pad, 1, (int32_t)m, (int32_t)m, 1, 1,
                                   ASTTS_ACT_NONE);
    if (rc != ASTTS_OK) return rc;
    ASTTS_REQUIRE(ln_gamma && ln_beta && ln_out_f16, ASTTS_ERR_INVALID, "astts_op_gemm_ln: null pointer");
    ASTTS_REQUIRE(n == 256 && cin == cin_pad, ASTTS_ERR_UNSUPPORTED,
                  "astts_op_gemm_ln: n=%d cin=%d (a workgroup owns whole output rows: n must be 256, cin a multiple of 64)", n, cin);
    ASTTS_REQUIRE((lda & 7) == 0 && (ldc & 3) == 0 && (ld_ln & 3) == 0 && (!residual || (ldr & 3) == 0) &&
                      (((uintptr_t)x_f16 | (uintptr_t)out | (uintptr_t)ln_out_f16 | (uintptr_t)residual | (uintptr_t)bias |
                        (uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0,
                  ASTTS_ERR_INVALID, "astts_op_gemm_ln: operands must be 16-byte aligned with 16-byte row strides");
    GemmArgs a{(const float*)x_f16, (const _Float16*)w_f16, bias, residual, nullptr, out, m, n, cin, cin_pad, 1,
               lda, ldc, ldr, (int32_t)m, (int32_t)m, 1, 1, 0, ASTTS_ACT_NONE, 1.0f, 0.1f,
               nullptr, ln_gamma, ln_beta, ln_eps, (float*)ln_out_f16, ld_ln, 0, 1, 0, 1};
    hipStream_t st = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring<1, 2, 3, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 288 * 128);
        attr = true;
    }
    const bool prof = prof_begin(ASTTS_PROF_GEMM_TILE, st, 2.0 * (double)m * n * cin);
    hipLaunchKernelGGL((gemm_ring<1, 2, 3, 4>), dim3((unsigned)cdiv(m, 32)), dim3(256), 3 * 288 * 128, st, a);
    if (prof) prof_end(ASTTS_PROF_GEMM_TILE, st);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

size_t astts_op_gemm_fused_workspace_bytes(void) {
    return 1024 + (size_t)128 * 4 * 512 * sizeof(float);      // arrival counters + [128 column blocks][4 slices][512] partial sums
}

int astts_op_gemm_fused_ws(const float* x, const int32_t* gather, const float* ln_gamma, const float* ln_beta, float ln_eps,
                           const void* w_f16, const float* bias, const float* residual, float* out, void* out2, int32_t out2_f16,
                           int32_t m, int32_t n, int32_t n_split, int32_t cin, int32_t cin_pad, int32_t lda, int32_t ldc,
                           int32_t ldc2, int32_t ldr, int32_t act, float alpha, float slope, void* workspace, size_t workspace_bytes,
                           astts_stream_t stream) {
    const int rc = check_gemm_args("astts_op_gemm_fused", x, w_f16, out, m, n, cin, cin_pad, 1, 1, 1, 1, 1, act);
    if (rc != ASTTS_OK) return rc;
    ASTTS_REQUIRE(m <= 32, ASTTS_ERR_INVALID, "astts_op_gemm_fused: m=%d > 32", m);
    ASTTS_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), ASTTS_ERR_INVALID, "astts_op_gemm_fused: gamma/beta");
    ASTTS_REQUIRE(!out2 || (n_split > 0 && n_split < n), ASTTS_ERR_INVALID, "astts_op_gemm_fused: n_split=%d", n_split);
    ASTTS_REQUIRE(!workspace || (workspace_bytes >= astts_op_gemm_fused_workspace_bytes() && ((uintptr_t)workspace & 255) == 0),
                  ASTTS_ERR_WORKSPACE, "astts_op_gemm_fused: workspace too small or misaligned");
    GemmArgs a{x, (const _Float16*)w_f16, bias, residual, nullptr, out, m, n, cin, cin_pad, 1,
               lda, ldc, ldr, m, m, 1, 1, 0, act, alpha, slope,
               gather, ln_gamma, ln_beta, ln_eps, (float*)out2, ldc2, n_split, 0, 0, out2_f16 ? 1 : 0,
               workspace ? reinterpret_cast<float*>((char*)workspace + 1024) : nullptr, reinterpret_cast<unsigned*>(workspace), 0};
    return launch_gemm(a, (hipStream_t)stream);
}

int astts_op_gemm_fused(const float* x, const int32_t* gather, const float* ln_gamma, const float* ln_beta, float ln_eps,
                        const void* w_f16, const float* bias, const float* residual, float* out, void* out2, int32_t out2_f16,
                        int32_t m, int32_t n, int32_t n_split, int32_t cin, int32_t cin_pad, int32_t lda, int32_t ldc,
                        int32_t ldc2, int32_t ldr, int32_t act, float alpha, float slope, astts_stream_t stream) {
    return astts_op_gemm_fused_ws(x, gather, ln_gamma, ln_beta, ln_eps, w_f16, bias, residual, out, out2, out2_f16, m, n, n_split, cin,
                                  cin_pad, lda, ldc, ldc2, ldr, act, alpha, slope, nullptr, 0, stream);
}

}  // extern "C"
