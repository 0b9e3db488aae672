// knn.hip -- brute-force kNN (COSINE / IP / L2) over an HBM-resident style bank (gfx950 / MI355X).
//
// Replaces MilvusClient.search on the COSINE collection of the reference
// (/root/reference/milvus/search_embeddings.py:15-22, /root/reference/src/search_milvus.py:140-147).
// Result definition = oracle/knn.py: fp64 score, order (closer first, row asc).  COSINE is what the reference's collection
// uses; IP and L2 (squared distance, Milvus' convention) are the other two metrics of the MilvusClient surface, and L2 with
// k = 1 is the arg-min of the speech tokenizer's vector quantiser (astts/frontend_nets.py).
// Internally every metric is a score S to MAXIMISE: cos, <q,b>, -|q-b|^2; the scan proposes with
// T = <q,b>/|b| (COSINE), <q,b> (IP), <q,b> - |b|^2/2 (L2: the same order as -|q-b|^2 for one query).
// Optional row mask (a Milvus `filter` evaluated by the host, or the rows already returned by earlier passes of a
// k > 32 search): masked rows are never candidates, in the approximate and in the exact path alike.
//
// Pipeline (five launches on one stream, no host sync, no allocation):
//   1 knn_prep_queries     fp32 queries -> power-of-two scaled fp16 image + padded fp32 copy + fp64 norms
//   2 knn_scan             fp16 MFMA (32x32x16) scan of the whole bank: S[q][n] ~ <q,b_n>/|b_n|  (HBM-bound)
//   3 knn_select           per query: top-C candidates of S by (score desc, row asc)
//   4 knn_rescore_finalize fp64 cosine of every candidate (one wave each), order by it, emit top-k and
//                          CERTIFY the candidate set: kth exact score > best possible score of any
//                          non-candidate (+ error bound); otherwise queue the query for the exact path
//     (exact path: inside knn_rescore_finalize -- a query that cannot be certified is re-scanned in fp64 by its own block)
//
// HBM layout: the SCAN plane is stored pre-tiled in the order the scan consumes it -- fp16
// [N/32 row tiles][Dp/64 lines][4 k-steps][64 lanes][8 halfs]: one (row tile, line) block is a contiguous 4 KB, one
// wave instruction reads a contiguous 1 KB (lane (r, h) of k-step i holds bank[32*rt + r][64*line + 32*h + 8*i .. +8]),
// so the whole scan is sequential streaming instead of 32 rows 2*Dp bytes apart.  Dp = D rounded up to 64, rows
// padded to a multiple of 32 with zeros.  The EXACT plane (fp64 re-score / exact path) stays row-major [N][Dp]:
// fp16 when the bank is fp16-exact, else fp32.  fp64 row norms [N]; fp32 inverse norms [N].
#include "common.h"
#include "toplist.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace astts {

static constexpr int kWave = 64;
static constexpr int kScanThreads = 256;
static constexpr int kMaxQPerPass = 256;

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// fp64 dot of a padded fp32 query row with a padded bank row (fp16 or fp32), one wave.
// Lane l accumulates elements 8*(64*s + l) .. +7 for s = 0,1,...; then a symmetric butterfly,
// so every lane returns the same value and identical rows give identical results.
template <typename RowT>
__device__ __forceinline__ double wave_dot64(const float* __restrict__ q, const RowT* __restrict__ row,
                                             int dp, int lane) {
    // (the element pairs of SIX steps are requested before the first FMA: as a rolled loop every step of 512 elements was a
    // dependent global round trip -- twelve in a row for a 6144-wide row, most of the re-score's time; the sums are taken in the same order)
    double acc = 0.0;
    constexpr int U = 6;
    for (int k0 = lane * 8; k0 < dp; k0 += U * kWave * 8) {
        float qv[U][8], b[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(k0 + u * kWave * 8, dp - 8);          // clamped: a valid address, unused past the end
            const float4 q0 = *reinterpret_cast<const float4*>(q + k);
            const float4 q1 = *reinterpret_cast<const float4*>(q + k + 4);
            qv[u][0] = q0.x; qv[u][1] = q0.y; qv[u][2] = q0.z; qv[u][3] = q0.w;
            qv[u][4] = q1.x; qv[u][5] = q1.y; qv[u][6] = q1.z; qv[u][7] = q1.w;
            if constexpr (sizeof(RowT) == 2) {
                const half8 hb = *reinterpret_cast<const half8*>(row + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) b[u][j] = (float)hb[j];
            } else {
                const float4 b0 = *reinterpret_cast<const float4*>(row + k);
                const float4 b1 = *reinterpret_cast<const float4*>(row + k + 4);
                b[u][0] = b0.x; b[u][1] = b0.y; b[u][2] = b0.z; b[u][3] = b0.w;
                b[u][4] = b1.x; b[u][5] = b1.y; b[u][6] = b1.z; b[u][7] = b1.w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k0 + u * kWave * 8 < dp) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = fma((double)qv[u][j], (double)b[u][j], acc);
            }
    }
    return wave_sum_f64(acc);
}

__device__ __forceinline__ double cos_from_parts(double dot, double qn, double bn) {
    double c = dot / (qn * bn);
    return isfinite(c) ? c : 0.0;
}

// fp64 squared distance sum_i (q_i - b_i)^2, one wave; the summation tree of wave_dot64 (identical rows give identical results,
// a row equal to the query gives exactly 0)
template <typename RowT>
__device__ __forceinline__ double wave_dist64(const float* __restrict__ q, const RowT* __restrict__ row, int dp, int lane) {
    double acc = 0.0;
    for (int k = lane * 8; k < dp; k += kWave * 8) {
        float4 q0 = *reinterpret_cast<const float4*>(q + k);
        float4 q1 = *reinterpret_cast<const float4*>(q + k + 4);
        const float qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        float b[8];
        if constexpr (sizeof(RowT) == 2) {
            half8 hb = *reinterpret_cast<const half8*>(row + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (float)hb[j];
        } else {
            float4 b0 = *reinterpret_cast<const float4*>(row + k);
            float4 b1 = *reinterpret_cast<const float4*>(row + k + 4);
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w;
            b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double df = (double)qq[j] - (double)b[j];
            acc = fma(df, df, acc);
        }
    }
    return wave_sum_f64(acc);
}

// The same two sums with the QUERY in LDS (the small-bank finishing kernel stages it once per workgroup: sixteen waves no longer fetch the
// same 4 dp bytes each) and ALL of the row's pieces requested before the first FMA (dp <= 8192: at most 16 steps; fp32 rows six steps at
// a time).  Same products, same order: bit-identical to wave_dot64 / wave_dist64.
template <typename RowT, bool DIST>
__device__ __forceinline__ double wave_sum64_ldsq(const float* q_lds, const RowT* __restrict__ row, int dp, int lane) {
    double acc = 0.0;
    constexpr int U = sizeof(RowT) == 2 ? 16 : 6;
    typedef typename std::conditional<sizeof(RowT) == 2, half8, float4>::type Piece;      // 8 fp16 values, or 4 of the 8 fp32 values
    for (int k0 = lane * 8; k0 < dp; k0 += U * kWave * 8) {
        Piece pa[U], pb[U];                          // (pb: the second four fp32 values; unused for fp16 rows)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(k0 + u * kWave * 8, dp - 8);
            pa[u] = *reinterpret_cast<const Piece*>(row + k);
            if constexpr (sizeof(RowT) != 2) pb[u] = *reinterpret_cast<const Piece*>(row + k + 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * kWave * 8;
            if (k < dp) {
                const float4 q0 = *reinterpret_cast<const float4*>(q_lds + k);
                const float4 q1 = *reinterpret_cast<const float4*>(q_lds + k + 4);
                const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                float bv[8];
                if constexpr (sizeof(RowT) == 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bv[j] = (float)pa[u][j];
                } else {
                    bv[0] = pa[u].x; bv[1] = pa[u].y; bv[2] = pa[u].z; bv[3] = pa[u].w;
                    bv[4] = pb[u].x; bv[5] = pb[u].y; bv[6] = pb[u].z; bv[7] = pb[u].w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (DIST) {
                        const double df = (double)qv[j] - (double)bv[j];
                        acc = fma(df, df, acc);
                    } else {
                        acc = fma((double)qv[j], (double)bv[j], acc);
                    }
                }
            }
        }
    }
    return wave_sum_f64(acc);
}

// the exact score S (larger = closer) of query row `q` against bank row `row` under `metric`, in two steps: the wave-wide sum
// (<q, b> or -|q - b|^2), then the part that needs the norms (COSINE)
template <typename RowT>
__device__ __forceinline__ double exact_raw(int metric, const float* __restrict__ q, const RowT* __restrict__ row, int dp, int lane) {
    return metric == ASTTS_METRIC_L2 ? -wave_dist64<RowT>(q, row, dp, lane) : wave_dot64<RowT>(q, row, dp, lane);
}
__device__ __forceinline__ double exact_finish(int metric, double raw, double qn, double bn) {
    return metric == ASTTS_METRIC_COSINE ? cos_from_parts(raw, qn, bn) : raw;
}
template <typename RowT>
__device__ __forceinline__ double exact_score(int metric, const float* __restrict__ q, const RowT* __restrict__ row, int dp, int lane,
                                              double qn, double bn) {
    return exact_finish(metric, exact_raw<RowT>(metric, q, row, dp, lane), qn, bn);
}
// what the caller sees: cosine, inner product, squared distance
__device__ __forceinline__ double user_score(int metric, double s) { return metric == ASTTS_METRIC_L2 ? -s : s; }

// ------------------------------------------------------------------------------------------
// bank construction: one wave per row -- copy/convert into the padded planes, fp64 norm,
// exactness + range flags
// ------------------------------------------------------------------------------------------
template <typename SrcT>
__global__ void knn_build_bank(const SrcT* __restrict__ src, int64_t n, int d, int dp,
                               _Float16* __restrict__ scan_tiled, _Float16* __restrict__ plane16,
                               float* __restrict__ plane32, double* __restrict__ norm64,
                               float* __restrict__ inv_norm, int* __restrict__ flags /* [0]=inexact, [1]=overflow, [2..3]=max norm (fp64 bits) */,
                               int metric, float* __restrict__ bias /* L2: -|b|^2 / 2 */) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
    if (row >= n) return;
    const SrcT* s = src + row * (int64_t)d;
    double acc = 0.0;
    bool inexact = false, overflow = false;
    for (int k = lane; k < dp; k += kWave) {
        float v = (k < d) ? (float)s[k] : 0.0f;
        _Float16 h = (_Float16)v;
        float back = (float)h;
        if (back != v) inexact = true;
        if (!isfinite(v)) overflow = true;     // (finite values beyond fp16's range make the bank inexact: rescaled per row below)
        plane16[row * (int64_t)dp + k] = h;
        if (plane32) plane32[row * (int64_t)dp + k] = v;
        {   // tiled scan image
            const int64_t rt = row >> 5;
            const int r = (int)(row & 31), line = k >> 6, kk = k & 63;
            const int hh = kk >> 5, i = (kk & 31) >> 3, j = kk & 7;
            scan_tiled[((rt * (dp >> 6) + line) << 11) + i * 512 + (hh * 32 + r) * 8 + j] = h;
        }
        acc = fma((double)v, (double)v, acc);
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) {
        double nrm = sqrt(acc);
        norm64[row] = nrm;
        inv_norm[row] = metric != ASTTS_METRIC_COSINE ? 1.0f : (nrm > 0.0 ? (float)(1.0 / nrm) : 0.0f);
        if (bias) bias[row] = (float)(-0.5 * acc);
        if (isfinite(nrm)) atomicMax(reinterpret_cast<unsigned long long*>(flags + 2), (unsigned long long)__double_as_longlong(nrm));
    }
    if (__any(inexact) && lane == 0) atomicOr(&flags[0], 1);
    if (__any(overflow) && lane == 0) atomicOr(&flags[1], 1);
}

// A bank that is NOT fp16-exact (an fp32 upload) gets its approximate planes re-written with a power-of-two scale per row, as
// the queries do: max|v| of every row lands in [2^13, 2^14), so the fp16 image keeps an 11-bit significand for every element
// that matters -- cosine is scale invariant, and rows of norm ~1e-2 (elements below fp16's normal range, 6e-5) would otherwise
// go subnormal or flush to zero in the scan and drop out of the candidate lists while the error bound still certified the
// query.  The scale is folded into inv_norm (exactly: a power of two); the exact plane (plane32) and the fp64 norms keep the
// original values.  Elements more than 2^37 below their row's maximum still flush: |error| <= sqrt(dp) * 2^-38 on the cosine
// scale, inside the 2^-20 term of the bound.
__global__ void knn_rescale_rows(const float* __restrict__ plane32, int64_t n, int dp, _Float16* __restrict__ scan_tiled,
                                 _Float16* __restrict__ plane16, const double* __restrict__ norm64, float* __restrict__ inv_norm, int metric) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* s = plane32 + row * (int64_t)dp;
    float mx = 0.0f;
    for (int k = lane; k < dp; k += kWave) mx = fmaxf(mx, fabsf(s[k]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    int e = 0;
    if (mx > 0.0f) frexpf(mx, &e);            // mx = m * 2^e, m in [0.5, 1)
    const int sh = mx > 0.0f ? 14 - e : 0;
    const float scale = ldexpf(1.0f, sh);
    for (int k = lane; k < dp; k += kWave) {
        const _Float16 h = (_Float16)(s[k] * scale);
        plane16[row * (int64_t)dp + k] = h;
        const int64_t rt = row >> 5;
        const int r = (int)(row & 31), line = k >> 6, kk = k & 63;
        const int hh = kk >> 5, i = (kk & 31) >> 3, j = kk & 7;
        scan_tiled[((rt * (dp >> 6) + line) << 11) + i * 512 + (hh * 32 + r) * 8 + j] = h;
    }
    if (lane == 0) {
        const double nrm = norm64[row];
        inv_norm[row] = metric != ASTTS_METRIC_COSINE ? ldexpf(1.0f, -sh) : (nrm > 0.0 ? (float)ldexp(1.0 / nrm, -sh) : 0.0f);
    }
}

// ------------------------------------------------------------------------------------------
// 1. query preparation: one block per query row; block 0 also clears the fallback counter
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_prep_queries(const float* __restrict__ q, int nq, int d, int dp,
                                                        _Float16* __restrict__ qh,
                                                        float* __restrict__ qf,
                                                        double* __restrict__ qn64,
                                                        float* __restrict__ qscale,
                                                        int* __restrict__ nflag,
                                                        _Float16* __restrict__ qrow) {
    __shared__ float smax[4];
    __shared__ double ssum[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (row == 0 && tid == 0 && nflag) *nflag = 0;
    if (row >= nq) {  // zero rows completing the last 32-query tile
        const int qt = row >> 5, r = row & 31;
        _Float16* tb = qh + (int64_t)qt * (dp >> 6) * 2048;
        half8 zero8;
#pragma unroll
        for (int j = 0; j < 8; ++j) zero8[j] = (_Float16)0.0f;
        for (int k = tid * 8; k < dp; k += 2048) {
            const int line = k >> 6, kk = k & 63;
            *reinterpret_cast<half8*>(&tb[((int64_t)line << 11) + ((kk & 31) >> 3) * 512 + ((kk >> 5) * 32 + r) * 8]) = zero8;
        }
        return;
    }
    const float* s = q + (int64_t)row * d;
    float* of = qf + (int64_t)row * dp;
    float mx = 0.0f;
    double acc = 0.0;
    // eight consecutive elements per thread and step: they are one 16-byte group of the tiled fp16 image, so the row is
    // read once (kept in registers across the block-wide max) and every store is a whole vector.  dp is a multiple of 64.
    constexpr int MAXG = 4;                        // groups per thread: covers dp <= 8192; longer rows loop again below
    const bool vec = ((uintptr_t)s & 15) == 0 && (d & 3) == 0;
    float vreg[MAXG][8];
    const int ngroups = dp >> 3;
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
        const int g8 = (tid + gi * 256) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) vreg[gi][j] = 0.0f;
        if (tid + gi * 256 < ngroups) {
            if (vec && g8 + 8 <= d) {
                const float4 a0 = *reinterpret_cast<const float4*>(s + g8);
                const float4 a1 = *reinterpret_cast<const float4*>(s + g8 + 4);
                vreg[gi][0] = a0.x; vreg[gi][1] = a0.y; vreg[gi][2] = a0.z; vreg[gi][3] = a0.w;
                vreg[gi][4] = a1.x; vreg[gi][5] = a1.y; vreg[gi][6] = a1.z; vreg[gi][7] = a1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) vreg[gi][j] = (g8 + j < d) ? s[g8 + j] : 0.0f;
            }
            *reinterpret_cast<float4*>(of + g8) = make_float4(vreg[gi][0], vreg[gi][1], vreg[gi][2], vreg[gi][3]);
            *reinterpret_cast<float4*>(of + g8 + 4) = make_float4(vreg[gi][4], vreg[gi][5], vreg[gi][6], vreg[gi][7]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mx = fmaxf(mx, fabsf(vreg[gi][j]));
                acc = fma((double)vreg[gi][j], (double)vreg[gi][j], acc);
            }
        }
    }
    for (int k = MAXG * 2048 + tid; k < dp; k += 256) {      // rows longer than 8192
        float v = (k < d) ? s[k] : 0.0f;
        of[k] = v;
        mx = fmaxf(mx, fabsf(v));
        acc = fma((double)v, (double)v, acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    acc = wave_sum_f64(acc);
    if (lane == 0) {
        smax[wid] = mx;
        ssum[wid] = acc;
    }
    __syncthreads();
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    const double tot = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
    // power-of-two scale that puts max|q| in [2^13, 2^14): exact in fp32, keeps fp16 well inside
    // its normal range so the only rounding is the 11-bit significand
    float scale = 1.0f;
    if (mx > 0.0f && isfinite(mx)) {
        int e;
        frexpf(mx, &e);  // mx = m * 2^e, m in [0.5,1)
        scale = ldexpf(1.0f, 14 - e);
    }
    {   // tiled fp16 image, same (row tile, line, k-step, lane) order as the bank's scan plane
        const int qt = row >> 5, r = row & 31;
        _Float16* tb = qh + (int64_t)qt * (dp >> 6) * 2048;
#pragma unroll
        for (int gi = 0; gi < MAXG; ++gi) {
            if (tid + gi * 256 < ngroups) {
                const int k = (tid + gi * 256) * 8;
                const int line = k >> 6, kk = k & 63;
                half8 hv;
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(vreg[gi][j] * scale);
                *reinterpret_cast<half8*>(&tb[((int64_t)line << 11) + ((kk & 31) >> 3) * 512 + ((kk >> 5) * 32 + r) * 8]) = hv;
                if (qrow) *reinterpret_cast<half8*>(&qrow[(int64_t)row * dp + k]) = hv;      // row-major copy for the GEMM scan
            }
        }
        for (int k = MAXG * 2048 + tid; k < dp; k += 256) {
            float v = (k < d) ? s[k] : 0.0f;
            const int line = k >> 6, kk = k & 63;
            tb[((int64_t)line << 11) + ((kk & 31) >> 3) * 512 + ((kk >> 5) * 32 + r) * 8 + (kk & 7)] = (_Float16)(v * scale);
            if (qrow) qrow[(int64_t)row * dp + k] = (_Float16)(v * scale);
        }
    }
    if (tid == 0) {
        qn64[row] = sqrt(tot);
        qscale[row] = scale;
    }
}

// ------------------------------------------------------------------------------------------
// 2. MFMA scan.  D[q][n] = sum_k Qh[q][k] * B[n][k]; A operand = 32 queries, B operand = 32 bank rows.
// Lane l = (r = l & 31, h = l >> 5) owns 64 contiguous bytes of row r of its tile in every 128-byte
// line: bytes [64h, 64h+64).  Those are 4 MFMA k-steps of 8 halfs each.  The k order inside a line is
// a permutation of the natural one, identical for A and B, which a dot product does not see.
// Block = 4 waves that split the block's K range line by line and reduce through LDS.
// Query rows past the group's last query are clamped to it (their results are never read).
// ------------------------------------------------------------------------------------------
// DIRECT (one tile pair per wave only): the query fragments come straight from the caller's fp32 rows [nq][dp] (dp == d, 16-byte
// aligned), rounded to fp16 as they are -- no preparation launch in front of the scan (a 12 MB bank is launch-bound: three dependent
// launches of ~5 us each).  Without the power-of-two pre-scale elements below fp16's normal range lose relative precision: the
// certification bound of that path carries the extra term (knn_rescore_body, `direct`); block (0, 0) clears the fallback counter.
template <int QT, int RT, bool DIRECT = false>
__global__ __launch_bounds__(kScanThreads) void knn_scan(
    const _Float16* __restrict__ bank, const _Float16* __restrict__ qh,
    const float* __restrict__ inv_norm, float* __restrict__ s_part, int64_t n, int dp, int nld,
    int qpad, int nq_group, int lines_per_split, const float* __restrict__ bias, const float* __restrict__ qscale_g,
    const float* __restrict__ qdirect, int* __restrict__ nflag_clear) {
    if (DIRECT && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *nflag_clear = 0;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [3][QT*RT*16][64]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * (32 * RT);
    const int total_lines = dp >> 6;
    const int line_begin = blockIdx.y * lines_per_split;
    int line_end = line_begin + lines_per_split;
    if (line_end > total_lines) line_end = total_lines;

    float16v acc[QT][RT];
#pragma unroll
    for (int a = 0; a < QT; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    // tiled planes: block (tile, line) = 2048 halfs; k-step i of lane l at i*512 + l*8
    const _Float16* bptr[RT];
#pragma unroll
    for (int b = 0; b < RT; ++b) bptr[b] = bank + ((int64_t)(blockIdx.x * RT + b) * total_lines << 11) + lane * 8;
    const _Float16* aptr[QT];
#pragma unroll
    for (int a = 0; a < QT; ++a) aptr[a] = qh + ((int64_t)a * total_lines << 11) + lane * 8;

    if constexpr (QT * RT == 1 && DIRECT) {
        half8 b0[4], b1[4];
        float4 q0[8], q1[8];
        const float* qp = qdirect + (int64_t)min(r, nq_group - 1) * dp + 32 * h;
        auto ld = [&](int line, half8 (&bf)[4], float4 (&qf)[8]) {
            const int lc = min(line, line_end - 1);
            const int64_t koff = (int64_t)lc << 11;
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = *reinterpret_cast<const half8*>(bptr[0] + koff + i * 512);
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const float4*>(qp + lc * 64 + 4 * i);
        };
        auto mm = [&](const half8 (&bf)[4], const float4 (&qf)[8]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                half8 af;
                af[0] = (_Float16)qf[2 * i].x; af[1] = (_Float16)qf[2 * i].y; af[2] = (_Float16)qf[2 * i].z; af[3] = (_Float16)qf[2 * i].w;
                af[4] = (_Float16)qf[2 * i + 1].x; af[5] = (_Float16)qf[2 * i + 1].y; af[6] = (_Float16)qf[2 * i + 1].z; af[7] = (_Float16)qf[2 * i + 1].w;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[i], acc[0][0], 0, 0, 0);
            }
        };
        int line = line_begin + wid;
        if (line < line_end) ld(line, b0, q0);
        for (; line < line_end; line += 8) {
            ld(line + 4, b1, q1);
            mm(b0, q0);
            if (line + 4 < line_end) {
                ld(line + 8, b0, q0);
                mm(b1, q1);
            }
        }
    } else if constexpr (QT * RT == 1) {
        // one tile pair per wave: the next line's fragments are requested before this line's MFMAs (two lines in flight per wave; the
        // loop is unrolled by two so that neither set is ever copied)
        half8 b0[4], a0[4], b1[4], a1[4];
        auto ld = [&](int line, half8 (&bf)[4], half8 (&af)[4]) {
            const int64_t koff = (int64_t)min(line, line_end - 1) << 11;       // clamped: a valid address, the result unused past the end
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = *reinterpret_cast<const half8*>(bptr[0] + koff + i * 512);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const half8*>(aptr[0] + koff + i * 512);
        };
        auto mm = [&](const half8 (&bf)[4], const half8 (&af)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[i], acc[0][0], 0, 0, 0);
        };
        int line = line_begin + wid;
        if (line < line_end) ld(line, b0, a0);
        for (; line < line_end; line += 8) {
            ld(line + 4, b1, a1);
            mm(b0, a0);
            if (line + 4 < line_end) {
                ld(line + 8, b0, a0);
                mm(b1, a1);
            }
        }
    } else
    for (int line = line_begin + wid; line < line_end; line += 4) {
        const int64_t koff = (int64_t)line << 11;
        half8 bf[RT][4];
        half8 af[QT][4];
#pragma unroll
        for (int b = 0; b < RT; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                bf[b][i] = *reinterpret_cast<const half8*>(bptr[b] + koff + i * 512);
#pragma unroll
        for (int a = 0; a < QT; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[a][i] = *reinterpret_cast<const half8*>(aptr[a] + koff + i * 512);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int a = 0; a < QT; ++a)
#pragma unroll
                for (int b = 0; b < RT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a][i], bf[b][i], acc[a][b], 0, 0, 0);
    }

    // cross-wave reduction (waves 1..3 -> LDS -> wave 0)
    if (wid > 0) {
        float* dst = red + (size_t)(wid - 1) * (QT * RT * 16 * 64);
#pragma unroll
        for (int a = 0; a < QT; ++a)
#pragma unroll
            for (int b = 0; b < RT; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[((a * RT + b) * 16 + i) * 64 + lane] = acc[a][b][i];
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int a = 0; a < QT; ++a)
#pragma unroll
            for (int b = 0; b < RT; ++b) {
                const int64_t col = row0 + b * 32 + r;  // bank row = MFMA column
                const float inv = (col < n) ? inv_norm[col] : 0.0f;
                // L2: the proposal score is qscale * (<q,b> - |b|^2 / 2); the constant rides on K slice 0
                const float bcol = (bias != nullptr && blockIdx.y == 0 && col < n) ? bias[col] : 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = acc[a][b][i];
#pragma unroll
                    for (int w = 0; w < 3; ++w)
                        v += red[(size_t)w * (QT * RT * 16 * 64) + ((a * RT + b) * 16 + i) * 64 + lane];
                    const int qrow = a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;  // MFMA row = query
                    if (col < nld && qrow < nq_group)
                        s_part[((size_t)blockIdx.y * qpad + qrow) * nld + col] = bias ? fmaf(DIRECT ? 1.0f : qscale_g[qrow], bcol, v * inv) : v * inv;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// 3. selection: top-C of a score row by (score desc, row asc).  One block (4 waves) per query.
// The K-split partial planes are summed while a 2048-score tile is staged in LDS (all loads of a
// tile are independent, so their latency overlaps); each wave then filters its quarter of the tile
// against its current c-th best and inserts the few survivors.
// ------------------------------------------------------------------------------------------
// Fast path: the segment (<= 8192 scores, K-split planes summed) is staged in LDS once; the c-th largest score is found by a
// 3-pass radix select on order-preserving keys (LDS histograms), the <= 64 entries at or above it are gathered and one wave
// sorts them by (score desc, row asc).  Ties that push the gather past 64 entries fall back to the chunked sorted-list path
// (each wave filters its share of the segment against its current c-th best and inserts the survivors).  The insertion path
// alone took 36 us on 1000 scores -- most of a config-2 search.
static constexpr int kSelSeg = 8192;      // host: seg_len <= kSelSeg

__device__ __forceinline__ unsigned sel_key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);       // larger float <=> larger unsigned key (-0 < +0 is harmless)
}

static constexpr int kSelThreads = 1024;
// body of the selection for query q, segment segy; the c candidates go to cand_idx / cand_s [0, c) (global memory: the
// stand-alone kernel; LDS: the fused select + re-score kernel)
__device__ __forceinline__ void knn_select_staged(const float* seg, int nl, int64_t seg0, int c, int* cand_idx, float* cand_s,
                                                  const uint8_t* __restrict__ mask);
__device__ __forceinline__ void knn_select_body(const float* __restrict__ s_part, int ksplit, int qpad, int nld, int64_t n_all, int c,
                                                int seg_len, int q, int segy, int* cand_idx, float* cand_s,
                                                const float* __restrict__ inv_norm, const float* __restrict__ bias, float qs,
                                                const uint8_t* __restrict__ mask, float* seg /* LDS [kSelSeg], the caller's */) {
    const int tid = threadIdx.x;
    const size_t plane = (size_t)qpad * nld;
    const float* base = s_part + (size_t)q * nld;
    // this block's segment of the row: [seg0, n)
    const int64_t seg0 = (int64_t)segy * seg_len;
    const int64_t n = (seg0 + seg_len < n_all) ? seg0 + seg_len : n_all;
    const int nl = (int)(n - seg0);
    if (ksplit == 1) {
        // one score plane (large banks, the GEMM scan): the segment's <= 8 scores of this thread (and their 1 / |b|) in ONE batch of
        // unconditional loads (clamped index) -- as a rolled loop every score was a dependent global round trip in front of its LDS store
        static_assert(kSelSeg == 8 * kSelThreads, "staging batch");
        float v[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t at = seg0 + min(tid + u * kSelThreads, nl - 1);
            v[u] = base[at];
            w[u] = inv_norm ? inv_norm[at] : 1.0f;
        }
        if (bias) {         // (the GEMM scan leaves raw dot products: L2's constant is added here)
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = fmaf(qs, bias[seg0 + min(tid + u * kSelThreads, nl - 1)] , v[u] * w[u]);
        } else if (inv_norm) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] *= w[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * kSelThreads;
            if (i < nl) seg[i] = v[u];
        }
    } else
    for (int i = tid; i < nl; i += kSelThreads) {
        // K-split partial planes (up to ~100 for a small bank): eight independent loads in flight per thread
        const float* pp = base + seg0 + i;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int ks = 0;
        for (; ks + 16 <= ksplit; ks += 16) {          // sixteen loads in flight (a 1000-row bank: 16 K slices = one round trip)
            float w16[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w16[u] = pp[(size_t)(ks + u) * plane];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] += w16[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] += w16[8 + u];
        }
        for (; ks + 8 <= ksplit; ks += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] += pp[(size_t)(ks + u) * plane];
        }
        for (; ks < ksplit; ++ks) v[0] += pp[(size_t)ks * plane];
        const float sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        float sc = inv_norm ? sum * inv_norm[seg0 + i] : sum;      // the GEMM scan leaves raw dot products
        if (bias) sc = fmaf(qs, bias[seg0 + i], sc);
        seg[i] = sc;
    }
    if (mask) {             // masked rows leave the ranking (a filter, or rows an earlier pass of a k > 32 search returned)
        __syncthreads();
        for (int i = tid; i < nl; i += kSelThreads)
            if (!mask[seg0 + i]) seg[i] = -INFINITY;
    }
    knn_select_staged(seg, nl, seg0, c, cand_idx, cand_s, mask);
}

// the selection proper: top c of seg[0, nl) (LDS, written by every thread's own stores: the first barrier below publishes them) by
// (score desc, position asc); candidate j is row seg0 + position
__device__ __forceinline__ void knn_select_staged(const float* seg, int nl, int64_t seg0, int c, int* cand_idx, float* cand_s,
                                                  const uint8_t* __restrict__ mask) {
    __shared__ float sh_s[kSelThreads];
    __shared__ int sh_i[kSelThreads];
    __shared__ unsigned hist[2048];
    __shared__ int s_sel_bin, s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        s_cnt = 0;
        s_sel_bin = 2048;       // (nothing to gather unless the histogram walk finds the bin)
    }
    __syncthreads();
    TopList<float> tl;
    tl.init();
    bool fast = true;
    if (nl <= kSelThreads && c <= 16) {
        // a small bank (one score per thread) and a 16-entry list: three rounds of one-wave sorts -- 16 waves keep their best 16 of 64,
        // four waves the best 16 of those 4 x 16, one wave the best 16 of the last 64 -- instead of the histogram walk (min / max,
        // 2048-bin histogram, prefix walk, gather, sort: eight barriers on a 1 000-score row).  The same total order (score desc, row asc).
        __syncthreads();
        {
            const bool valid = tid < nl;
            tl.seed(valid ? seg[tid] : -INFINITY, (int)(seg0 + tid), valid, lane);
            if (lane < 16) {
                sh_s[wid * 16 + lane] = tl.s;
                sh_i[wid * 16 + lane] = tl.idx;
            }
        }
        __syncthreads();
        if (wid < 4) {
            const int vi = sh_i[wid * 64 + lane];
            tl.seed(sh_s[wid * 64 + lane], vi, vi != kNoIdx, lane);
            if (lane < 16) {
                sh_s[512 + wid * 16 + lane] = tl.s;
                sh_i[512 + wid * 16 + lane] = tl.idx;
            }
        }
        __syncthreads();
        if (wid == 0) {
            const int vi = sh_i[512 + lane];
            tl.seed(sh_s[512 + lane], vi, vi != kNoIdx, lane);
        }
        if (tid < c) {
            int id = (tl.idx == kNoIdx) ? -1 : tl.idx;
            if (id >= 0 && mask && !mask[id]) id = -1;
            cand_idx[tid] = id;
            cand_s[tid] = tl.s;
        }
        return;
    }
    if (nl > 64) {
        // ONE histogram pass over 2048 LINEAR bins of [min, max] of the segment (round 5).  The 3-pass radix select on the float bits
        // that stood here put cosine scores -- a narrow band around zero on a large bank -- into a handful of bins per pass: up to 8 192
        // LDS atomics on the same few addresses, serialised (225 us of a 740 us 256-query search against a 100k bank went into this
        // kernel, profiles/r05_knn_q256_kernel_stats.csv).  Linear bins spread the band (a Gaussian's densest bin of 2048 over +-4 sigma
        // holds ~13 of 8 192 scores), and only the TOP of the histogram is walked: the bin b* in which the c-th largest score falls;
        // everything in bins >= b* is gathered (a superset of the top c: the map score -> bin is monotone) and sorted by one wave.
        __shared__ float s_mn[kSelThreads / 64], s_mx[kSelThreads / 64];
        __shared__ int s_nr[kSelThreads / 64];
        float mn = INFINITY, mx = -INFINITY;
        int nr = 0;                 // scores that take part in the ranking (masked rows and NaNs sit at -inf and do not)
        for (int i = tid; i < nl; i += kSelThreads) {
            const float v = seg[i];
            if (v > -INFINITY) ++nr;
            if (v > -INFINITY && v < INFINITY) {
                mn = fminf(mn, v);
                mx = fmaxf(mx, v);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off, 64));
            mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            nr += __shfl_xor(nr, off, 64);
        }
        if (lane == 0) {
            s_mn[wid] = mn;
            s_mx[wid] = mx;
            s_nr[wid] = nr;
        }
        for (int i = tid; i < 2048; i += kSelThreads) hist[i] = 0u;
        __syncthreads();
        nr = 0;
#pragma unroll
        for (int w = 0; w < kSelThreads / 64; ++w) {
            mn = fminf(mn, s_mn[w]);
            mx = fmaxf(mx, s_mx[w]);
            nr += s_nr[w];
        }
        const float bscale = mx > mn ? 2047.0f / (mx - mn) : 0.0f;
        auto bin_of = [&](float v) -> int {
            if (!(v > -INFINITY)) return 0;
            if (!(v < INFINITY)) return 2047;
            const int bq = (int)((v - mn) * bscale);                      // monotone in v (fp subtraction, product, truncation all are)
            return bq < 0 ? 0 : (bq > 2047 ? 2047 : bq);
        };
        const int remaining = c < nr ? c : nr;
        for (int i = tid; i < nl; i += kSelThreads)
            if (seg[i] > -INFINITY) atomicAdd(&hist[bin_of(seg[i])], 1u);
        __syncthreads();
        if (wid == 0) {
            constexpr int per = 32;                          // 2048 bins / 64 lanes
            unsigned local = 0u;
            for (int j = 0; j < per; ++j) local += hist[lane * per + j];
            unsigned incl = local;                           // sum over lanes >= lane
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned tv = __shfl_down(incl, off, 64);
                if (lane + off < 64) incl += tv;
            }
            const unsigned above = incl - local;
            if (remaining > 0 && above < (unsigned)remaining && (unsigned)remaining <= incl) {
                unsigned acc = above;
                for (int j = per - 1; j >= 0; --j) {
                    const unsigned hcount = hist[lane * per + j];
                    if (acc + hcount >= (unsigned)remaining) {
                        s_sel_bin = lane * per + j;
                        break;
                    }
                    acc += hcount;
                }
            }
        }
        __syncthreads();
        const int bstar = s_sel_bin;
        for (int i = tid; i < nl; i += kSelThreads) {
            if (seg[i] > -INFINITY && bin_of(seg[i]) >= bstar) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < 64) {
                    sh_s[pos] = seg[i];
                    sh_i[pos] = (int)(seg0 + i);
                }
            }
        }
        __syncthreads();
        fast = s_cnt <= 64;
    } else {
        if (tid < nl) {
            sh_s[tid] = seg[tid];
            sh_i[tid] = (int)(seg0 + tid);
        }
        if (tid == 0) s_cnt = nl;
        __syncthreads();
    }
    if (fast) {
        if (wid == 0) tl.seed(sh_s[lane], sh_i[lane], lane < s_cnt, lane);   // the sort fixes the order whatever the gather order was
    } else {
        __syncthreads();
        bool seeded = false;
        for (int b0 = wid * 64; b0 < nl; b0 += kSelThreads) {
            const int li = b0 + lane;
            const bool valid = li < nl;
            const float x = valid ? seg[li] : -INFINITY;
            if (!seeded) {
                tl.seed(x, (int)(seg0 + li), valid, lane);
                seeded = true;
            } else {
                tl.offer(x, (int)(seg0 + li), valid, lane, c);
            }
        }
        merge_lists<float>(tl, sh_s, sh_i, c);
    }
    if (tid < c) {
        int id = (tl.idx == kNoIdx) ? -1 : tl.idx;
        if (id >= 0 && mask && !mask[id]) id = -1;          // fewer than c allowed rows in the segment
        cand_idx[tid] = id;
        cand_s[tid] = tl.s;
    }
}

__global__ __launch_bounds__(kSelThreads) void knn_select(const float* __restrict__ s_part, int ksplit, int qpad, int nld, int64_t n_all,
                                                          int c, int seg_len, int* __restrict__ cand_idx, float* __restrict__ cand_s,
                                                          const float* __restrict__ inv_norm, const float* __restrict__ bias,
                                                          const float* __restrict__ qscale_g, const uint8_t* __restrict__ mask,
                                                          int64_t mask_stride) {
    __shared__ float seg[kSelSeg];
    const size_t o = ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64;
    knn_select_body(s_part, ksplit, qpad, nld, n_all, c, seg_len, blockIdx.x, blockIdx.y, cand_idx + o, cand_s + o, inv_norm, bias,
                    bias ? qscale_g[blockIdx.x] : 0.0f, mask ? mask + (int64_t)blockIdx.x * mask_stride : nullptr, seg);
}

// merge the per-segment candidate lists of one query (one wave) into the final top-C; the lists of sixteen segments are requested
// together (as one load per segment in front of its offer, every segment cost a global round trip: 21.6 us for 13 segments)
__global__ __launch_bounds__(64) void knn_select_merge(const int* __restrict__ seg_idx, const float* __restrict__ seg_s,
                                                       int nseg, int c, int* __restrict__ cand_idx,
                                                       float* __restrict__ cand_s) {
    const int q = blockIdx.x, lane = threadIdx.x;
    TopList<float> tl;
    tl.init();
    for (int s0 = 0; s0 < nseg; s0 += 16) {
        int vi[16];
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int sg = s0 + u < nseg ? s0 + u : nseg - 1;
            const size_t o = ((size_t)q * nseg + sg) * 64 + lane;
            const bool live = lane < c && s0 + u < nseg;
            vi[u] = live ? seg_idx[o] : -1;
            v[u] = live ? seg_s[o] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (s0 + u >= nseg) break;
            if (s0 + u == 0)
                tl.seed(v[u], vi[u], vi[u] >= 0, lane);
            else
                tl.offer(v[u], vi[u], vi[u] >= 0, lane, c);
        }
    }
    if (lane < c) {
        cand_idx[q * 64 + lane] = (tl.idx == kNoIdx) ? -1 : tl.idx;
        cand_s[q * 64 + lane] = tl.s;
    }
}

// Selection over a LONG score row for a large query group (the GEMM scan: one plane, >= 64 queries): ONE block per query streams
// the row against the c-th best score so far.  The first 8192-score tile goes through the histogram selection of knn_select_body and
// seeds the list; after it the row is taken in spans of NV scores per thread whose loads are all requested before the first compare
// (no barrier inside a span -- a __syncthreads() drains vmcnt, which exposed the load latency once per tile in the tile-at-a-time form:
// 68 us), survivors (strictly above the c-th best at the start of the span: on a random row ~c x span / rows-so-far of them) go to an LDS
// pool through an atomic cursor, and behind ONE barrier per span wave 0 inserts the pool into its sorted list.  A span that overflows the
// pool (an ascending row) is redone tile by tile with the histogram selection.  The result is the top c by (score desc, row asc), the
// same set and order knn_select + knn_select_merge produce, from 256 blocks instead of 256 x 13 + 256 (119 + 22 us ->
// profiles/r06_knn_q256_kernel_stats.csv).
static constexpr int kStreamPool = 512;
template <int NV, bool EXTRA>   // scores per thread and span (NV x 1024 scores per span); EXTRA: a row mask and / or L2's per-row constant
__global__ __launch_bounds__(kSelThreads) void knn_select_stream(const float* __restrict__ s_plane, int nld, int64_t n, int c,
                                                                 int* __restrict__ cand_idx, float* __restrict__ cand_s,
                                                                 const float* __restrict__ inv_norm, const float* __restrict__ bias,
                                                                 const float* __restrict__ qscale_g, const uint8_t* __restrict__ mask,
                                                                 int64_t mask_stride) {
    __shared__ float seg[kSelSeg];
    __shared__ int f_ci[64];
    __shared__ float f_cs[64];
    __shared__ float pool_s[kStreamPool];
    __shared__ int pool_i[kStreamPool];
    __shared__ int s_pool_n;
    __shared__ float s_tau;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* base = s_plane + (size_t)q * nld;
    const uint8_t* mq = EXTRA && mask ? mask + (int64_t)q * mask_stride : nullptr;
    if (!EXTRA) bias = nullptr;
    const float qs = bias ? qscale_g[q] : 0.0f;
    TopList<float> tl;           // wave 0's: the best c so far
    tl.init();
    if (tid == 0) {
        s_pool_n = 0;
        s_tau = -INFINITY;
    }
    // one 8192-score tile through the histogram selection, its c candidates offered to the list (all threads; ends behind a barrier)
    auto tile_by_histogram = [&](int t) {
        knn_select_body(s_plane, 1, 0, nld, n, c, kSelSeg, q, t, f_ci, f_cs, inv_norm, bias, qs, mq, seg);
        __syncthreads();
        if (wid == 0) {
            const int ci = lane < c ? f_ci[lane] : -1;
            tl.offer(lane < c ? f_cs[lane] : -INFINITY, ci, ci >= 0, lane, c);
            if (lane == c - 1) s_tau = tl.idx == kNoIdx ? -INFINITY : tl.s;
        }
        __syncthreads();
    };
    tile_by_histogram(0);
    constexpr int64_t kSpan = (int64_t)NV * kSelThreads;
    static_assert(kSpan % kSelSeg == 0, "a span is whole tiles");
    for (int64_t s0 = kSelSeg; s0 < n; s0 += kSpan) {
        const int64_t left = n - s0;
        const float tau = s_tau;            // (written before the barrier that ended the previous span)
        float v[NV], w[NV];
        unsigned live = 0u;
#pragma unroll
        for (int u = 0; u < NV; ++u) {      // every load of the span, with exactly the arithmetic of knn_select_body's staging below
            const int64_t i = (int64_t)tid + u * kSelThreads;
            const int64_t at = s0 + (i < left ? i : left - 1);
            v[u] = base[at];
            w[u] = inv_norm ? inv_norm[at] : 1.0f;
            if (i < left && (!mq || mq[at])) live |= 1u << u;
        }
        if (bias) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int64_t i = (int64_t)tid + u * kSelThreads;
                v[u] = fmaf(qs, bias[s0 + (i < left ? i : left - 1)], v[u] * w[u]);
            }
        } else if (inv_norm) {
#pragma unroll
            for (int u = 0; u < NV; ++u) v[u] *= w[u];
        }
        // survivors: strictly above the c-th best so far (rows arrive in ascending order, so a tie loses to the row already listed)
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (((live >> u) & 1u) && v[u] > tau) {
                const int pos = atomicAdd(&s_pool_n, 1);
                if (pos < kStreamPool) {
                    pool_s[pos] = v[u];
                    pool_i[pos] = (int)(s0 + tid + u * kSelThreads);
                }
            }
        __syncthreads();
        const int cnt = s_pool_n;
        if (cnt == 0) continue;             // (nobody writes the cursor before the next span's barrier unless it has a survivor)
        if (cnt <= kStreamPool) {
            if (wid == 0) {
                for (int b0 = 0; b0 < cnt; b0 += 64) {
                    const int li = b0 + lane;
                    const bool ok = li < cnt;
                    tl.offer(ok ? pool_s[li] : -INFINITY, ok ? pool_i[li] : kNoIdx, ok, lane, c);
                }
                if (lane == c - 1) s_tau = tl.idx == kNoIdx ? -INFINITY : tl.s;
                if (lane == 0) s_pool_n = 0;
            }
            __syncthreads();
        } else {                            // more survivors than the pool holds: this span again, tile by tile
            __syncthreads();                // (every thread has read the cursor)
            if (tid == 0) s_pool_n = 0;
            const int t_end = (int)(((s0 + kSpan < n ? s0 + kSpan : n) + kSelSeg - 1) / kSelSeg);
            for (int t = (int)(s0 / kSelSeg); t < t_end; ++t) tile_by_histogram(t);
        }
    }
    if (wid == 0 && lane < c) {
        cand_idx[q * 64 + lane] = (tl.idx == kNoIdx) ? -1 : tl.idx;
        cand_s[q * 64 + lane] = tl.s;
    }
}

// Selection for a large query group whose scan left BLOCK MAXIMA beside the scores (gemm_scan's epilogue: the largest score of
// every 64 bank rows, per query).  The c blocks with the largest maxima -- by (maximum desc, block asc) -- hold the top c scores: each of
// them has a score >= T = the c-th largest maximum, every other block's scores are <= T, and a score equal to T in an unlisted block
// sits at a higher row than the T-scores of the listed ones.  So the block reads its query's n / 64 maxima (25 KB of a 400 KB row at
// 100k), selects c of them with the selection every segment goes through, puts the listed blocks in ascending order (positions then
// ascend with the row index: ties), gathers their c x 64 scores and selects again.  Same candidates, same order as knn_select_stream /
// knn_select + merge; no row mask (a mask changes the maxima): masked and multi-pass searches keep the streaming form.
// (body: the c candidates -- rows and scores -- go to out_ci / out_cs [0, c), global memory or LDS)
__device__ __forceinline__ void knn_select_blocks_body(const float* __restrict__ s_plane, int nld, int64_t n, int c,
                                                       const float* __restrict__ bmax, int bm_ld, int nblk, int q, int* out_ci,
                                                       float* out_cs, float* seg /* LDS [kSelSeg], the caller's */) {
    __shared__ int f_ci[64], f_blk[64];
    __shared__ float f_cs[64];
    __shared__ int s_nb;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < nblk; i += kSelThreads) seg[i] = bmax[(size_t)q * bm_ld + i];
    knn_select_staged(seg, nblk, 0, c, f_ci, f_cs, nullptr);
    __syncthreads();
    if (wid == 0) {
        const int b = lane < c ? f_ci[lane] : -1;
        TopList<float> tl;
        tl.seed(-(float)b, b, b >= 0, lane);        // ascending block index (< 2^24: exact as a float)
        f_blk[lane] = tl.idx;
        const int nb = __popcll(__ballot(tl.idx != kNoIdx));
        if (lane == 0) s_nb = nb;
    }
    __syncthreads();
    const int nl = s_nb * 64;
    const float* row = s_plane + (size_t)q * nld;
    for (int p = tid; p < nl; p += kSelThreads) {
        const int64_t at = (int64_t)f_blk[p >> 6] * 64 + (p & 63);
        seg[p] = at < n ? row[at] : -INFINITY;
    }
    knn_select_staged(seg, nl, 0, c, f_ci, f_cs, nullptr);
    __syncthreads();
    if (tid < c) {
        const int p = f_ci[tid];
        out_ci[tid] = p >= 0 ? f_blk[p >> 6] * 64 + (p & 63) : -1;
        out_cs[tid] = f_cs[tid];
    }
}

// ------------------------------------------------------------------------------------------
// 4. fp64 re-score of the candidates (16 waves, one candidate each per round), then wave 0 orders
// them by the exact score, emits the top-k and certifies the candidate set.
// ------------------------------------------------------------------------------------------
// body for query q; cand_idx / cand_s: this query's candidates [0, c) (global memory or LDS)
template <typename RowT>
__device__ __forceinline__ void knn_rescore_body(
    const int q, const float* __restrict__ qf, const double* __restrict__ qn64, const float* __restrict__ qscale,
    const RowT* __restrict__ plane, const double* __restrict__ norm64, int64_t n, int dp, int c, int k,
    const int* cand_idx, const float* cand_s, double err_bound,
    int force_exact, int64_t* __restrict__ out_idx, float* __restrict__ out_score, double* __restrict__ out_score64,
    int* __restrict__ nflag, int* __restrict__ flagged, int metric, double bmax, const uint8_t* __restrict__ mask,
    int out_ld, int out_off, int direct = 0, bool have_pre = false, float4 pre0 = float4{0.f, 0.f, 0.f, 0.f},
    float4 pre1 = float4{0.f, 0.f, 0.f, 0.f}, const float* q_lds = nullptr) {
    // q_lds (with have_pre): the query sits in LDS, thread t staged elements [8 t, 8 t + 8) from pre0 / pre1 (zeros beyond dp)
    // out_*: row q starts at q * out_ld + out_off (a k > 32 search emits 32 hits per pass into its [nq, k] result)
    // direct: no preparation launch ran (knn_scan<.., DIRECT>): qf is the caller's query matrix (dp == d, dp % 128 == 0), the fp16
    // image was taken without a pre-scale (qscale = 1), and the query norm is formed here -- 16 waves, one slice each, summed in wave order
    __shared__ double sh_cos[64];
    __shared__ double sh_bn[64];
    __shared__ double sh_qq[16];
    __shared__ float sh_qmax[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double qn = direct ? 0.0 : qn64[q];
    const int64_t ob = (int64_t)q * out_ld + out_off;
    float qmax = 0.0f;
    if (direct) {
        const int per = dp >> 4;
        double a = 0.0;
        if (have_pre ? tid * 8 < dp : lane * 8 < per) {      // (pre: thread t holds elements [8 t, 8 t + 8); else wave w's slice of dp / 16)
            const float* qp = qf + (int64_t)q * dp + wid * per + lane * 8;
            const float4 a0 = have_pre ? pre0 : *reinterpret_cast<const float4*>(qp), a1 = have_pre ? pre1 : *reinterpret_cast<const float4*>(qp + 4);
            a = fma((double)a0.x, (double)a0.x, a); a = fma((double)a0.y, (double)a0.y, a);
            a = fma((double)a0.z, (double)a0.z, a); a = fma((double)a0.w, (double)a0.w, a);
            a = fma((double)a1.x, (double)a1.x, a); a = fma((double)a1.y, (double)a1.y, a);
            a = fma((double)a1.z, (double)a1.z, a); a = fma((double)a1.w, (double)a1.w, a);
            // (NaN-propagating maximum: a non-finite element must end in the exact path too)
            const float m8 = fmaxf(fmaxf(fmaxf(fabsf(a0.x), fabsf(a0.y)), fmaxf(fabsf(a0.z), fabsf(a0.w))),
                                   fmaxf(fmaxf(fabsf(a1.x), fabsf(a1.y)), fmaxf(fabsf(a1.z), fabsf(a1.w))));
            qmax = isfinite(a) ? m8 : INFINITY;
        }
        a = wave_sum_f64(a);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) qmax = fmaxf(qmax, __shfl_xor(qmax, off, 64));
        if (lane == 0) {
            sh_qq[wid] = a;
            sh_qmax[wid] = qmax;
        }
    }
    for (int ci = wid; ci < c; ci += 16) {
        const int idx = cand_idx[ci];
        double raw = -INFINITY;
        const double bn = idx >= 0 ? norm64[idx] : 0.0;      // (requested with the row: the ranking below does not wait for it again)
        if (idx >= 0) {
            if (q_lds)
                raw = metric == ASTTS_METRIC_L2 ? -wave_sum64_ldsq<RowT, true>(q_lds, plane + (int64_t)idx * dp, dp, lane)
                                                : wave_sum64_ldsq<RowT, false>(q_lds, plane + (int64_t)idx * dp, dp, lane);
            else
                raw = exact_raw<RowT>(metric, qf + (int64_t)q * dp, plane + (int64_t)idx * dp, dp, lane);
        }
        if (lane == 0) {
            sh_cos[ci] = raw;
            sh_bn[ci] = bn;
        }
    }
    __syncthreads();
    if (direct) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            t += sh_qq[w];
            qmax = fmaxf(qmax, sh_qmax[w]);
        }
        qn = sqrt(t);
    }
    // direct: an element beyond fp16's range became +-inf in the scan's query image -- its scores are inf / NaN and bound nothing
    const bool q_overflow = direct && !(qmax <= 65504.0f);
    __shared__ int s_exact;
    if (wid == 0) {
    const bool valid = lane < c;
    const int idx = valid ? cand_idx[lane] : -1;
    const bool live = valid && idx >= 0;
    const double cs = live ? exact_finish(metric, sh_cos[lane], qn, sh_bn[lane]) : -INFINITY;
    const float ap = live ? cand_s[lane] : INFINITY;
    // rank among the candidates: every lane reads all c (score, row) pairs back from LDS -- uniform addresses, all reads in flight at
    // once (as __shfl of a double and an int this loop was three dependent ds_bpermute round trips per candidate: 3.5 us of a 17 us kernel)
    __shared__ double sh_fin[64];
    __shared__ int sh_fid[64];
    if (valid) {
        sh_fin[lane] = cs;
        sh_fid[lane] = idx;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    int rank = 0;
#pragma unroll 16
    for (int j = 0; j < c; ++j) {
        const double sj = sh_fin[j];
        const int ij = sh_fid[j];
        if (ij >= 0 && j != lane && better<double>(sj, ij, cs, idx)) ++rank;
    }
    // hits that exist: fewer live candidates than the list holds means EVERY allowed row is a candidate (each segment returns
    // its best c rows, masked ones last and dropped; the merge keeps the best c of the union)
    const int n_live = __popcll(__ballot(live));
    const int kk = k < n_live ? k : n_live;
    const double no_hit = metric == ASTTS_METRIC_L2 ? INFINITY : -INFINITY;
    if (live && rank < kk) {
        out_idx[ob + rank] = idx;
        out_score[ob + rank] = (float)user_score(metric, cs);
        if (out_score64) out_score64[ob + rank] = user_score(metric, cs);
    }
    if (lane >= kk && lane < k) {  // fewer than k (allowed) rows in the bank
        out_idx[ob + lane] = -1;
        out_score[ob + lane] = (float)no_hit;
        if (out_score64) out_score64[ob + lane] = no_hit;
    }
    float tau = ap;  // smallest approximate score among the candidates bounds every non-candidate
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tau = fminf(tau, __shfl_xor(tau, off, 64));
    double kth = (live && rank == kk - 1) ? cs : -INFINITY;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) kth = fmax(kth, __shfl_xor(kth, off, 64));
    if (lane == 0) {
        bool certified;
        if (n <= (int64_t)c || n_live < c) {
            certified = true;  // every (allowed) row is a candidate
        } else if (metric == ASTTS_METRIC_COSINE) {
            const double denom = (direct ? 1.0 : (double)qscale[q]) * qn;
            const double tau_cos = denom > 0.0 ? (double)tau / denom : INFINITY;
            // direct: query elements below fp16's normal range were rounded with an ABSOLUTE error of 2^-25 each (no pre-scale):
            // |sum_i dq_i b_i| <= 2^-25 sqrt(dp) |b|, i.e. 2^-25 sqrt(dp) / |q| on the cosine scale
            const double extra = direct ? ldexp(sqrt((double)dp), -25) / qn : 0.0;
            certified = isfinite(tau_cos) && (kth > tau_cos + err_bound + extra);
        } else {
            // IP / L2: the proposal score is T = <q,b> (- |b|^2 / 2), its error |q||b| * err_bound <= |q| * max|b| * err_bound
            // (+ the fp32 rounding of the L2 constant and of the sum); the k-th exact hit in T units: L2  T = (|q|^2 - d^2) / 2
            const double tau_t = (double)tau / (direct ? 1.0 : (double)qscale[q]);
            const double kth_t = metric == ASTTS_METRIC_L2 ? 0.5 * (qn * qn + kth) : kth;
            const double err = qn * bmax * err_bound + (metric == ASTTS_METRIC_L2 ? (qn * bmax + bmax * bmax) * 4.8e-7 : 0.0) +
                               (direct ? ldexp(sqrt((double)dp), -25) * bmax : 0.0);
            certified = isfinite(tau_t) && isfinite(kth_t) && (kth_t > tau_t + err);
        }
        s_exact = (!certified || force_exact || q_overflow) ? 1 : 0;
        if (s_exact) {
            atomicAdd(nflag, 1);     // astts_knn_last_fallbacks
            (void)flagged;
        }
    }
    }
    __syncthreads();
    if (!s_exact) return;
    // ---- exact path (rare by construction): fp64 cosine against every row with the same wave_dot64 as the re-score (both
    // paths return identical scores), exact top-k; this query's 16 waves stream the whole bank.  (Was a fifth launch whose
    // blocks normally exited at once: 4.5 us of launch floor on a 38 us search.)
    __shared__ double ex_s[1024];
    __shared__ int ex_i[1024];
    const float* qrow = qf + (int64_t)q * dp;
    TopList<double> tl;
    tl.init();
    for (int64_t base = (int64_t)wid * 64; base < n; base += 16 * 64) {
        double mine = -INFINITY;
        const int64_t lim = (n - base) < 64 ? (n - base) : 64;
        const bool allowed = lane < lim && (!mask || mask[base + lane]);
        const unsigned long long todo = __ballot(allowed);
        for (int j = 0; j < lim; ++j) {
            if (!((todo >> j) & 1ull)) continue;              // (wave-uniform)
            const int64_t row = base + j;
            const double csx = exact_score<RowT>(metric, qrow, plane + row * (int64_t)dp, dp, lane, qn, norm64[row]);
            if (lane == j) mine = csx;
        }
        tl.offer(mine, (int)(base + lane), allowed, lane, k);
    }
    merge_lists<double>(tl, ex_s, ex_i, k);
    if (tid < k) {
        const bool ok = tl.idx != kNoIdx;
        const double no_hit = metric == ASTTS_METRIC_L2 ? INFINITY : -INFINITY;
        out_idx[ob + tid] = ok ? tl.idx : -1;
        out_score[ob + tid] = ok ? (float)user_score(metric, tl.s) : (float)no_hit;
        if (out_score64) out_score64[ob + tid] = ok ? user_score(metric, tl.s) : no_hit;
    }
}

template <typename RowT>
__global__ __launch_bounds__(1024) void knn_rescore_finalize(
    const float* __restrict__ qf, const double* __restrict__ qn64, const float* __restrict__ qscale,
    const RowT* __restrict__ plane, const double* __restrict__ norm64, int64_t n, int dp, int c, int k,
    const int* __restrict__ cand_idx, const float* __restrict__ cand_s, double err_bound,
    int force_exact, int64_t* __restrict__ out_idx, float* __restrict__ out_score, double* __restrict__ out_score64,
    int* __restrict__ nflag, int* __restrict__ flagged, int metric, double bmax, const uint8_t* __restrict__ mask, int64_t mask_stride,
    int out_ld, int out_off) {
    const int q = blockIdx.x;
    knn_rescore_body<RowT>(q, qf, qn64, qscale, plane, norm64, n, dp, c, k, cand_idx + q * 64, cand_s + q * 64, err_bound, force_exact,
                           out_idx, out_score, out_score64, nflag, flagged, metric, bmax, mask ? mask + (int64_t)q * mask_stride : nullptr,
                           out_ld, out_off);
}

// rows a pass of a k > 32 search has returned leave the per-query mask before the next pass
__global__ void knn_mask_out(const int64_t* __restrict__ out_idx, int out_ld, int out_off, int kp, uint8_t* __restrict__ mask, int64_t n) {
    const int q = blockIdx.x;
    if ((int)threadIdx.x < kp) {
        const int64_t id = out_idx[(int64_t)q * out_ld + out_off + threadIdx.x];
        if (id >= 0) mask[(int64_t)q * n + id] = 0;
    }
}
// per-query masks of a k > 32 search: the caller's row mask (one for all queries, or one per query) or all ones
__global__ void knn_mask_init(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int64_t src_stride, int64_t n) {
    const int q = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[(int64_t)q * n + i] = src ? (src[(int64_t)q * src_stride + i] ? 1 : 0) : 1;
}

// Selection + re-score in one launch when a query's score row is one segment (N <= 8192) and all queries fit one pass: the
// candidates stay in LDS (a config-2 search is four dependent launches of 5-9 us each, mostly launch floor: one less).
template <typename RowT>
__global__ __launch_bounds__(1024) void knn_select_rescore(
    const float* __restrict__ s_part, int ksplit, int qpad, int nld, int seg_len, const float* __restrict__ inv_norm,
    const float* __restrict__ qf, const double* __restrict__ qn64, const float* __restrict__ qscale,
    const RowT* __restrict__ plane, const double* __restrict__ norm64, int64_t n, int dp, int c, int k, double err_bound,
    int force_exact, int64_t* __restrict__ out_idx, float* __restrict__ out_score, double* __restrict__ out_score64,
    int* __restrict__ nflag, int* __restrict__ flagged, int metric, double bmax, const float* __restrict__ bias,
    const uint8_t* __restrict__ mask, int64_t mask_stride, int direct) {
    __shared__ int f_ci[64];
    __shared__ float f_cs[64];
    __shared__ __attribute__((aligned(16))) float seg[kSelSeg];      // the selection's score segment, then (direct) the query
    const int q = blockIdx.x;
    const uint8_t* mq = mask ? mask + (int64_t)q * mask_stride : nullptr;
    // direct (dp <= 8192 = 8 floats per thread): the whole query is requested before the selection -- thread t elements [8 t, 8 t + 8) --
    // and goes to LDS behind it: the fp64 dots read it from there, and the sixteen waves' loads are their candidate rows only
    float4 pre0 = float4{0.f, 0.f, 0.f, 0.f}, pre1 = pre0;
    if (direct && (int)threadIdx.x * 8 < dp) {
        const float* qp = qf + (int64_t)q * dp + threadIdx.x * 8;
        pre0 = *reinterpret_cast<const float4*>(qp);
        pre1 = *reinterpret_cast<const float4*>(qp + 4);
    }
    knn_select_body(s_part, ksplit, qpad, nld, n, c, seg_len, q, 0, f_ci, f_cs, inv_norm, bias, bias ? (direct ? 1.0f : qscale[q]) : 0.0f, mq, seg);
    __syncthreads();
    if (direct) {
        *reinterpret_cast<float4*>(seg + threadIdx.x * 8) = pre0;
        *reinterpret_cast<float4*>(seg + threadIdx.x * 8 + 4) = pre1;
        __syncthreads();
    }
    knn_rescore_body<RowT>(q, qf, qn64, qscale, plane, norm64, n, dp, c, k, f_ci, f_cs, err_bound, force_exact, out_idx, out_score,
                           out_score64, nflag, flagged, metric, bmax, mq, k, 0, direct, direct != 0, pre0, pre1, direct ? seg : nullptr);
}

// the same pair for a large query group behind the GEMM scan: selection from the block maxima + fp64 re-score, one block per query
template <typename RowT>
__global__ __launch_bounds__(1024) void knn_blocks_rescore(
    const float* __restrict__ s_plane, int nld, const float* __restrict__ bmax, int bm_ld, int nblk,
    const float* __restrict__ qf, const double* __restrict__ qn64, const float* __restrict__ qscale,
    const RowT* __restrict__ plane, const double* __restrict__ norm64, int64_t n, int dp, int c, int k, double err_bound,
    int force_exact, int64_t* __restrict__ out_idx, float* __restrict__ out_score, double* __restrict__ out_score64,
    int* __restrict__ nflag, int* __restrict__ flagged, int metric, double bmax_norm) {
    __shared__ int c_ci[64];
    __shared__ float c_cs[64];
    __shared__ float seg[kSelSeg];
    const int q = blockIdx.x;
    knn_select_blocks_body(s_plane, nld, n, c, bmax, bm_ld, nblk, q, c_ci, c_cs, seg);
    __syncthreads();
    knn_rescore_body<RowT>(q, qf, qn64, qscale, plane, norm64, n, dp, c, k, c_ci, c_cs, err_bound, force_exact, out_idx, out_score,
                           out_score64, nflag, flagged, metric, bmax_norm, nullptr, k, 0);
}

}  // namespace astts

// ==========================================================================================
// host side
// ==========================================================================================
using namespace astts;

struct astts_knn {
    int64_t n = 0;
    int d = 0, dp = 0, nld = 0;
    int metric = 0;
    bool exact16 = true;          // scan plane is a lossless image of the bank
    _Float16* scan = nullptr;     // tiled scan plane [ceil(n/128)*4 row tiles][dp/64][4][64][8]
    _Float16* plane16 = nullptr;  // [n][dp] row-major (exact plane when exact16)
    float* plane32 = nullptr;     // [n][dp], only when !exact16
    double* norm64 = nullptr;     // [n]
    float* inv_norm = nullptr;    // [n]  COSINE: 1 / |b| (x the row's power-of-two scale); IP / L2: that scale alone
    float* bias = nullptr;        // [n]  L2 only: -|b|^2 / 2
    double bmax = 0.0;            // largest row norm (IP / L2 certification works on the absolute scale)
    double err_bound = 0.0;
    // bench-only profiling (astts_knn_profile_*)
    bool profile = false;
    std::vector<hipEvent_t> ev;  // pairs (start, stop)
    size_t ev_used = 0;
};

namespace {

struct KnnPlan {
    int qt, rt, ksplit, lines_per_split, tiles, qpad, c, nseg, seg_len;
    bool gemm;       // query groups of >= 64: the scan is a plain GEMM on the ring kernel (MFMA-side regime)
    size_t off_qrow;
    size_t off_nflag, off_flagged, off_qh, off_qf, off_qn, off_qscale, off_spart, off_cidx, off_cs, off_sidx, off_ss, off_mask, off_bmax, total;
    int nblk, bm_ld; // 64-row blocks of the bank; block maxima beside the GEMM scan's scores when nblk <= 8192 (one selection segment)
    int passes;      // k > 32: ceil(k / 32) selection + re-score passes over ONE scan, per chunk of <= 256 queries
    bool direct_ok;  // shape allows the two-launch form (scan straight from the caller's fp32 queries + fused select / re-score)
};

static constexpr int kPassK = 32;         // hits per pass (the certified top-k kernel keeps k <= 32 of a 64-entry candidate list)

KnnPlan make_plan(const astts_knn* h, int nq, int k) {
    KnnPlan p{};
    p.passes = 1;
    if (k > kPassK) {        // multi-pass search: chunks of <= 256 queries, 32 hits per pass, per-query masks of the rows already returned
        p.passes = (int)cdiv(k, kPassK);
        if (nq > kMaxQPerPass) nq = kMaxQPerPass;
        k = kPassK;
    }
    const int qgroup = nq < kMaxQPerPass ? nq : kMaxQPerPass;
    p.qt = qgroup <= 32 ? 1 : qgroup <= 64 ? 2 : qgroup <= 128 ? 4 : 8;
    p.qpad = p.qt * 32;
    // row tiles per wave: ONE for up to 32 queries -- the scan is an HBM stream and what it needs is waves in flight (48 VGPRs: eight waves
    // per SIMD), not reuse of the query fragments (L2 hits): 100k x 6144, Q = 8: 336 us per search with four tiles per wave (782 blocks of
    // 160 VGPRs), 277 with one; 100k x 768: 75.8 -> 57.6.  Two tiles for 33 .. 63 queries on a large bank (342 against 383 us at Q = 48).
    p.rt = (h->n >= 16384 && p.qt == 2) ? 2 : 1;
    p.tiles = (int)cdiv(h->n, 32 * p.rt);
    const int total_lines = h->dp / 64;
    // K split: a small bank needs it to fill the chip at all (32 tiles x 16 slices); a mid-sized one gets enough slices for ~six blocks per CU
    int ks = (int)cdiv(p.tiles >= 256 ? 1536 : 512, p.tiles);
    int ks_max = total_lines / 4;
    if (ks_max < 1) ks_max = 1;
    if (ks > ks_max) ks = ks_max;
    if (ks < 1) ks = 1;
    static const int ks_env = [] { const char* e = getenv("ASTTS_KNN_KSPLIT"); return e ? atoi(e) : 0; }();
    if (ks_env > 0) ks = ks_env < ks_max ? ks_env : ks_max;  // tuning override
    p.lines_per_split = (int)cdiv(total_lines, ks);
    p.ksplit = (int)cdiv(total_lines, p.lines_per_split);
    static const bool nogemm_env = getenv("ASTTS_KNN_NO_GEMM") != nullptr;
    // ... once the GEMM grid fills the chip with 64 x 64 tiles; a small bank keeps the K-split scan (16 blocks x 96 K tiles would crawl).
    // (Round 6: counted in 64-row query tiles, not 128 -- the speech tokenizer's quantiser, 4096 codes x 1280 against groups of 240
    // frames, took the register-streaming scan with eight query tiles per wave at 95 us per group; as a GEMM: see DESIGN section 3.)
    p.gemm = !nogemm_env && qgroup >= 64 && cdiv(qgroup, 64) * cdiv(h->n, 64) >= 256;
    if (p.gemm) {        // one score plane; a tail group of < 64 queries runs the register-streaming scan unsplit
        p.ksplit = 1;
        p.lines_per_split = total_lines;
    }
    p.c = k <= 8 ? 16 : 64;
    // selection segments: one block per (query, 8192-score segment), at most 64 segments per query
    p.nseg = (int)cdiv(h->n, 8192);       // segments of <= kSelSeg scores (staged in LDS by knn_select)
    if (p.nseg < 1) p.nseg = 1;           // (astts_knn_create bounds n so that nseg <= 1024)
    p.seg_len = (int)align_up((size_t)cdiv(h->n, p.nseg), 64);
    p.nseg = (int)cdiv(h->n, p.seg_len);
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + bytes, 256);
        return at;
    };
    p.off_nflag = take(256);
    p.off_flagged = take(sizeof(int) * (size_t)nq * p.passes);
    p.off_qh = take(sizeof(_Float16) * (align_up((size_t)nq, 32) + 256) * h->dp);  // whole 32-query tiles (+ one group's tail)
    p.off_qf = take(sizeof(float) * (size_t)nq * h->dp);
    p.off_qrow = take(p.gemm ? sizeof(_Float16) * (size_t)nq * h->dp : 16);
    p.off_qn = take(sizeof(double) * (size_t)nq);
    p.off_qscale = take(sizeof(float) * (size_t)nq);
    p.off_spart = take(sizeof(float) * (size_t)p.ksplit * p.qpad * h->nld);
    p.off_cidx = take(sizeof(int) * (size_t)nq * 64);
    p.off_cs = take(sizeof(float) * (size_t)nq * 64);
    p.off_sidx = take(sizeof(int) * (size_t)kMaxQPerPass * p.nseg * 64);
    p.off_ss = take(sizeof(float) * (size_t)kMaxQPerPass * p.nseg * 64);
    p.off_mask = take(p.passes > 1 ? (size_t)nq * (size_t)h->n : 16);
    p.nblk = (int)cdiv(h->n, 64);
    p.bm_ld = (int)align_up((size_t)p.nblk, 4);         // (the GEMM stores a tile's four maxima of a row as one vector)
    p.off_bmax = take(p.gemm && p.nblk <= kSelSeg ? sizeof(float) * (size_t)kMaxQPerPass * p.bm_ld : 16);
    p.total = o;
    // small bank, one query tile: no preparation launch (the caller's pointer alignment is checked at the call)
    p.direct_ok = p.passes == 1 && nq <= 32 && p.nseg == 1 && !p.gemm && p.qt == 1 && p.rt == 1 && h->dp == h->d && (h->dp & 127) == 0 &&
                  h->dp <= 8192;      // (the finishing kernel stages the query in its 8192-float segment buffer)
    return p;
}

template <int QT, int RT, bool DIRECT = false>
int launch_scan(const astts_knn* h, const KnnPlan& p, const _Float16* qh, int nq_group, float* spart,
                const float* qscale_g, hipStream_t st, const float* qdirect = nullptr, int* nflag_clear = nullptr) {
    dim3 grid(p.tiles, p.ksplit);
    size_t lds = (size_t)3 * QT * RT * 16 * 64 * sizeof(float);
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_scan<QT, RT, DIRECT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return ASTTS_ERR_HIP;
            }
            once = true;
        }
    }
    hipLaunchKernelGGL((knn_scan<QT, RT, DIRECT>), grid, dim3(kScanThreads), lds, st, h->scan, qh,
                       h->inv_norm, spart, h->n, h->dp, h->nld, p.qpad, nq_group, p.lines_per_split, h->bias, qscale_g, qdirect, nflag_clear);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // namespace

extern "C" {

int astts_knn_create(const void* bank, int64_t n, int32_t d, int32_t dtype, int32_t metric,
                     astts_stream_t stream, astts_knn_t** out) {
    ASTTS_REQUIRE(out != nullptr, ASTTS_ERR_INVALID, "astts_knn_create: out is null");
    *out = nullptr;
    ASTTS_REQUIRE(bank != nullptr, ASTTS_ERR_INVALID, "astts_knn_create: bank is null");
    ASTTS_REQUIRE(n >= 1 && n <= 0x7fffffff - 4096, ASTTS_ERR_INVALID,
                  "astts_knn_create: n=%lld out of range", (long long)n);
    ASTTS_REQUIRE(n <= (int64_t)1024 * kSelSeg, ASTTS_ERR_UNSUPPORTED,
                  "astts_knn_create: n=%lld rows (selection covers at most 1024 segments of %d scores)", (long long)n, kSelSeg);
    ASTTS_REQUIRE(d >= 1 && d <= (1 << 20), ASTTS_ERR_INVALID, "astts_knn_create: d=%d out of range", d);
    ASTTS_REQUIRE(dtype == ASTTS_DTYPE_F16 || dtype == ASTTS_DTYPE_F32, ASTTS_ERR_INVALID,
                  "astts_knn_create: dtype %d (want ASTTS_DTYPE_F16|F32)", dtype);
    ASTTS_REQUIRE(metric == ASTTS_METRIC_COSINE || metric == ASTTS_METRIC_IP || metric == ASTTS_METRIC_L2, ASTTS_ERR_INVALID,
                  "astts_knn_create: metric %d (want ASTTS_METRIC_COSINE|IP|L2)", metric);
    hipStream_t st = (hipStream_t)stream;
    astts_knn* h = new astts_knn();
    h->n = n;
    h->d = d;
    h->dp = (int)align_up((size_t)d, 64);
    h->nld = (int)align_up((size_t)n, 128);
    h->metric = metric;
    int* flags = nullptr;
    float* p32 = nullptr;
    auto fail = [&](int code) {
        if (flags) (void)hipFree(flags);
        if (p32) (void)hipFree(p32);
        astts_knn_destroy(h);
        return code;
    };
#define KNN_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));                        \
            return fail(ASTTS_ERR_HIP);                                                      \
        }                                                                                    \
    } while (0)
    const size_t scan_rows = align_up((size_t)n, 128);  // every row tile a scan block may touch exists
    KNN_TRY(hipMalloc(&h->scan, sizeof(_Float16) * scan_rows * h->dp));
    KNN_TRY(hipMemsetAsync(h->scan, 0, sizeof(_Float16) * scan_rows * h->dp, st));
    KNN_TRY(hipMalloc(&h->plane16, sizeof(_Float16) * (size_t)n * h->dp));
    KNN_TRY(hipMalloc(&h->norm64, sizeof(double) * (size_t)n));
    KNN_TRY(hipMalloc(&h->inv_norm, sizeof(float) * (size_t)n));
    if (metric == ASTTS_METRIC_L2) KNN_TRY(hipMalloc(&h->bias, sizeof(float) * (size_t)n));
    KNN_TRY(hipMalloc(&flags, 4 * sizeof(int)));
    KNN_TRY(hipMemsetAsync(flags, 0, 4 * sizeof(int), st));
    const int rows_per_block = 4;
    dim3 grid((unsigned)cdiv(n, rows_per_block));
    if (dtype == ASTTS_DTYPE_F32) {
        KNN_TRY(hipMalloc(&p32, sizeof(float) * (size_t)n * h->dp));
        hipLaunchKernelGGL((knn_build_bank<float>), grid, dim3(256), 0, st, (const float*)bank, n, d,
                           h->dp, h->scan, h->plane16, p32, h->norm64, h->inv_norm, flags, metric, h->bias);
    } else {
        hipLaunchKernelGGL((knn_build_bank<_Float16>), grid, dim3(256), 0, st, (const _Float16*)bank,
                           n, d, h->dp, h->scan, h->plane16, (float*)nullptr, h->norm64, h->inv_norm, flags, metric, h->bias);
    }
    KNN_TRY(hipGetLastError());
    int hf[4] = {0, 0, 0, 0};
    KNN_TRY(hipMemcpyAsync(hf, flags, sizeof(hf), hipMemcpyDeviceToHost, st));
    KNN_TRY(hipStreamSynchronize(st));
    memcpy(&h->bmax, &hf[2], sizeof(double));
    if (hf[1]) {
        set_error("astts_knn_create: bank holds non-finite values");
        return fail(ASTTS_ERR_RANGE);
    }
    h->exact16 = (hf[0] == 0);
    if (!h->exact16) {
        h->plane32 = p32;  // keep the fp32 image for exact re-scoring
        p32 = nullptr;
        // approximate planes re-written with a per-row power-of-two scale (see knn_rescale_rows)
        hipLaunchKernelGGL(knn_rescale_rows, grid, dim3(256), 0, st, h->plane32, n, h->dp, h->scan, h->plane16, h->norm64, h->inv_norm, metric);
        KNN_TRY(hipGetLastError());
        KNN_TRY(hipStreamSynchronize(st));
    }
    (void)hipFree(flags);
    flags = nullptr;
    if (p32) {
        (void)hipFree(p32);
        p32 = nullptr;
    }
#undef KNN_TRY
    // Error bound of the fp16 scan on the cosine scale (DESIGN.md "certification"):
    //   query rounded to fp16 (11-bit significand, power-of-two pre-scale): 2^-11 (Cauchy-Schwarz)
    //   fp32 accumulation of dp exact products inside the MFMA chain + cross-wave/ksplit adds: 2*dp*2^-24
    //   bank rounded to fp16 when it is not fp16-exact: 2^-11
    //   inv_norm rounding, final scaling: 2^-20
    h->err_bound = ldexp(1.0, -11) + 2.0 * (double)h->dp * ldexp(1.0, -24) + ldexp(1.0, -20) +
                   (h->exact16 ? 0.0 : ldexp(1.0, -11));
    *out = h;
    return ASTTS_OK;
}

int astts_knn_destroy(astts_knn_t* h) {
    if (!h) return ASTTS_OK;
    if (h->scan) (void)hipFree(h->scan);
    if (h->plane16) (void)hipFree(h->plane16);
    if (h->plane32) (void)hipFree(h->plane32);
    if (h->norm64) (void)hipFree(h->norm64);
    if (h->inv_norm) (void)hipFree(h->inv_norm);
    if (h->bias) (void)hipFree(h->bias);
    for (auto& e : h->ev) (void)hipEventDestroy(e);
    delete h;
    return ASTTS_OK;
}

int astts_knn_info(const astts_knn_t* h, int64_t* n, int32_t* d, int32_t* scan_plane_exact) {
    ASTTS_REQUIRE(h != nullptr, ASTTS_ERR_INVALID, "astts_knn_info: handle is null");
    if (n) *n = h->n;
    if (d) *d = h->d;
    if (scan_plane_exact) *scan_plane_exact = h->exact16 ? 1 : 0;
    return ASTTS_OK;
}

size_t astts_knn_workspace_bytes(const astts_knn_t* h, int32_t nq, int32_t k) {
    if (!h || nq < 1 || k < 1 || k > ASTTS_KNN_MAX_K) return 0;
    return make_plan(h, nq, k).total;
}

int astts_knn_search(astts_knn_t* h, const float* queries, int32_t nq, int32_t k, int64_t* out_idx,
                     float* out_score, void* workspace, size_t workspace_bytes, int32_t flags,
                     astts_stream_t stream) {
    return astts_knn_search_masked(h, queries, nq, k, out_idx, out_score, nullptr, nullptr, 0, workspace, workspace_bytes, flags, stream);
}

int astts_knn_search_f64(astts_knn_t* h, const float* queries, int32_t nq, int32_t k, int64_t* out_idx,
                         float* out_score, double* out_score64, void* workspace, size_t workspace_bytes, int32_t flags,
                         astts_stream_t stream) {
    return astts_knn_search_masked(h, queries, nq, k, out_idx, out_score, out_score64, nullptr, 0, workspace, workspace_bytes, flags, stream);
}

}  // extern "C"

namespace {

// One chunk of queries (all of them when k <= 32; <= 256 when k > 32): preparation, ONE scan per query group, then `passes` rounds of
// selection + fp64 re-score that each emit `kp` <= 32 hits per query into out[q * out_ld + done ..] -- between rounds the rows just
// returned leave the chunk's per-query masks, so round r + 1 ranks what is left (same certification, same exact path).
int knn_search_chunk(astts_knn* h, const KnnPlan& p, const float* queries, int nq, int k, int64_t* out_idx, float* out_score,
                     double* out_score64, const uint8_t* row_mask, int64_t mask_stride, char* ws, int flags, bool clear_flag,
                     hipStream_t st) {
    astts_stream_t stream = (astts_stream_t)st;
    int* nflag = (int*)(ws + p.off_nflag);
    int* flagged = (int*)(ws + p.off_flagged);
    _Float16* qh = (_Float16*)(ws + p.off_qh);
    float* qf = (float*)(ws + p.off_qf);
    _Float16* qrow = p.gemm ? (_Float16*)(ws + p.off_qrow) : nullptr;
    double* qn = (double*)(ws + p.off_qn);
    float* qscale = (float*)(ws + p.off_qscale);
    float* spart = (float*)(ws + p.off_spart);
    int* cidx = (int*)(ws + p.off_cidx);
    float* cs = (float*)(ws + p.off_cs);
    int* sidx = (int*)(ws + p.off_sidx);
    float* ss = (float*)(ws + p.off_ss);
    const int force = (flags & ASTTS_KNN_FORCE_EXACT) ? 1 : 0;
    const bool multi = p.passes > 1;
#define KNN_RESCORE_D(KERNEL, GRID, ...)                                                                                            \
    do {                                                                                                                            \
        if (h->exact16) {                                                                                                           \
            const _Float16* PLANE = h->plane16;                                                                                     \
            hipLaunchKernelGGL((KERNEL<_Float16>), GRID, dim3(1024), 0, st, __VA_ARGS__);                                          \
        } else {                                                                                                                    \
            const float* PLANE = h->plane32;                                                                                        \
            hipLaunchKernelGGL((KERNEL<float>), GRID, dim3(1024), 0, st, __VA_ARGS__);                                             \
        }                                                                                                                           \
        ASTTS_CHECK_LAUNCH();                                                                                                       \
    } while (0)
    const uint8_t* mask = row_mask;
    int64_t mstride = mask_stride;
    if (multi) {
        uint8_t* pm = (uint8_t*)(ws + p.off_mask);
        hipLaunchKernelGGL(knn_mask_init, dim3((unsigned)(cdiv(h->n, 256 * 16) < 1024 ? cdiv(h->n, 256 * 16) : 1024), nq), dim3(256), 0, st,
                           pm, row_mask, mask_stride, h->n);
        ASTTS_CHECK_LAUNCH();
        mask = pm;
        mstride = h->n;
    }

    static const bool no_direct = getenv("ASTTS_KNN_NO_DIRECT") != nullptr;       // A/B: the three-launch form for small banks
    static const bool no_stream = getenv("ASTTS_KNN_NO_STREAM_SELECT") != nullptr;   // A/B: per-segment selection + merge for large query groups
    static const bool no_blocks = getenv("ASTTS_KNN_NO_BLOCK_MAX") != nullptr;       // A/B: the GEMM scan without its block-maximum epilogue
    static const bool n_first = exp_env_int("ASTTS_KNN_GEMM_N_FIRST", 0) != 0;       // A/B: the projections' tile order (bank read once per panel)
    const bool direct = p.direct_ok && !no_direct && (((uintptr_t)queries) & 15) == 0;
    if (direct) {
        // two launches: the scan reads the fp32 queries itself; selection + fp64 re-score + certification in one kernel
        const bool prof = h->profile && h->ev_used + 2 <= h->ev.size();
        if (prof) ASTTS_CHECK_HIP(hipEventRecord(h->ev[h->ev_used], st));
        const int rc = launch_scan<1, 1, true>(h, p, nullptr, nq, spart, nullptr, st, queries, nflag);
        if (rc != ASTTS_OK) return rc;
        if (prof) {
            ASTTS_CHECK_HIP(hipEventRecord(h->ev[h->ev_used + 1], st));
            h->ev_used += 2;
        }
        const uint8_t* mask_g = mask;
        KNN_RESCORE_D(knn_select_rescore, dim3(nq), spart, p.ksplit, p.qpad, h->nld, p.seg_len, (const float*)nullptr, queries, qn, qscale,
                      PLANE, h->norm64, h->n, h->dp, p.c, k, h->err_bound, force, out_idx, out_score, out_score64, nflag, flagged, h->metric,
                      h->bmax, (const float*)nullptr, mask_g, mstride, 1);
        return ASTTS_OK;
    }
    hipLaunchKernelGGL(knn_prep_queries, dim3((unsigned)align_up((size_t)nq, 32)), dim3(256), 0, st, queries, nq, h->d, h->dp,
                       qh, qf, qn, qscale, clear_flag ? nflag : nullptr, qrow);
    ASTTS_CHECK_LAUNCH();

    // (PLANE: the exact plane in its own type -- fp16 when the bank is fp16-exact, else fp32)
#define KNN_RESCORE(KERNEL, GRID, ...)                                                                                              \
    do {                                                                                                                            \
        if (h->exact16) {                                                                                                           \
            const _Float16* PLANE = h->plane16;                                                                                     \
            hipLaunchKernelGGL((KERNEL<_Float16>), GRID, dim3(1024), 0, st, __VA_ARGS__);                                          \
        } else {                                                                                                                    \
            const float* PLANE = h->plane32;                                                                                        \
            hipLaunchKernelGGL((KERNEL<float>), GRID, dim3(1024), 0, st, __VA_ARGS__);                                             \
        }                                                                                                                           \
        ASTTS_CHECK_LAUNCH();                                                                                                       \
    } while (0)

    int finished = 0;           // leading queries whose hits a fused selection + re-score launch has already written
    // query groups of <= 256; behind the GEMM scan EQUAL groups (300 queries = 150 + 150, not 256 + a tail of 44 that falls back to the
    // register-streaming scan with eight query tiles per wave: 3.1 ms of a 3.5 ms search at 100k x 6144)
    const int gstep = p.gemm ? (int)cdiv(nq, cdiv(nq, kMaxQPerPass)) : kMaxQPerPass;
    for (int q0 = 0; q0 < nq; q0 += gstep) {
        const int qg = (nq - q0) < gstep ? (nq - q0) : gstep;
        const _Float16* qh_g = qh + (size_t)q0 * h->dp;
        int rc;
        const bool prof = h->profile && h->ev_used + 2 <= h->ev.size();
        if (prof) ASTTS_CHECK_HIP(hipEventRecord(h->ev[h->ev_used], st));
        const bool as_gemm = p.gemm && qg >= 64;
        // block maxima beside the scores (and the scores scaled by the GEMM's epilogue): unmasked single-pass searches
        const bool use_blocks = as_gemm && !no_blocks && p.nblk <= kSelSeg && p.nseg > 1 && !multi && !mask && !n_first;
        if (as_gemm) {
            // S[q][n] = <q, b_n> as one GEMM: activations = this group's queries (row-major fp16), "weights" = the bank's
            // row-major fp16 plane [n][dp]; the LDS-DMA ring kernel runs it at 400+ TFLOP/s where the register-streaming scan
            // (built for the HBM-bound small-Q regime) re-reads the query tile from L2 per bank tile.  1 / |b_n| (and L2's
            // constant) are applied by the selection kernel.
            if (n_first)
                rc = astts_op_gemm_ex(qrow + (size_t)q0 * h->dp, 1, h->plane16, nullptr, nullptr, nullptr, spart, 0, qg, (int32_t)h->n,
                                      h->dp, h->dp, 1, h->dp, h->nld, 0, qg, qg, 1, 1, 0, ASTTS_ACT_NONE, 1.0f, 0.1f, stream);
            else if (use_blocks)
                rc = gemm_scan(qrow + (size_t)q0 * h->dp, h->plane16, spart, qg, h->n, h->dp, h->nld, st, h->inv_norm, h->bias, qscale + q0,
                               (float*)(ws + p.off_bmax), p.bm_ld);
            else
                rc = gemm_scan(qrow + (size_t)q0 * h->dp, h->plane16, spart, qg, h->n, h->dp, h->nld, st);
        } else
        switch (p.qt * 10 + p.rt) {
            case 11: rc = launch_scan<1, 1>(h, p, qh_g, qg, spart, qscale + q0, st); break;
            case 21: rc = launch_scan<2, 1>(h, p, qh_g, qg, spart, qscale + q0, st); break;
            case 22: rc = launch_scan<2, 2>(h, p, qh_g, qg, spart, qscale + q0, st); break;
            case 41: rc = launch_scan<4, 1>(h, p, qh_g, qg, spart, qscale + q0, st); break;
            case 81: rc = launch_scan<8, 1>(h, p, qh_g, qg, spart, qscale + q0, st); break;
            default:
                set_error("astts_knn_search: no scan variant for qt=%d rt=%d", p.qt, p.rt);
                return ASTTS_ERR_INVALID;
        }
        if (rc != ASTTS_OK) return rc;
        if (prof) {
            ASTTS_CHECK_HIP(hipEventRecord(h->ev[h->ev_used + 1], st));
            h->ev_used += 2;
        }
        const float* sel_inv = as_gemm && !use_blocks ? h->inv_norm : nullptr;
        const float* sel_bias = as_gemm && !use_blocks ? h->bias : nullptr;
        const int sel_ks = as_gemm ? 1 : p.ksplit;
        const uint8_t* mask_g = mask ? mask + (int64_t)q0 * mstride : nullptr;
        if (p.nseg == 1 && nq <= kMaxQPerPass && !multi) {      // one segment, one query group: selection + re-score in one launch
            KNN_RESCORE(knn_select_rescore, dim3(nq), spart, sel_ks, p.qpad, h->nld, p.seg_len, sel_inv, qf, qn, qscale,
                        PLANE, h->norm64, h->n, h->dp, p.c, k, h->err_bound, force,
                        out_idx, out_score, out_score64, nflag, flagged, h->metric, h->bmax, sel_bias, mask_g, mstride, 0);
            return ASTTS_OK;
        }
        // (a k > 32 search re-ranks the SAME score plane once per pass, so its groups finish before the next group's scan overwrites it)
        const int rounds = multi ? p.passes : 1;
        for (int r = 0; r < rounds; ++r) {
            if (p.nseg == 1) {
                hipLaunchKernelGGL(knn_select, dim3(qg, 1), dim3(kSelThreads), 0, st, spart, sel_ks, p.qpad, h->nld,
                                   h->n, p.c, p.seg_len, cidx + (size_t)q0 * 64, cs + (size_t)q0 * 64, sel_inv, sel_bias, qscale + q0,
                                   mask_g, mstride);
                ASTTS_CHECK_LAUNCH();
            } else if (use_blocks) {                // the scan left block maxima: c blocks of 64 scores per query instead of the row,
                                                    // and the fp64 re-score behind the selection in the same launch (use_blocks: !multi)
                KNN_RESCORE(knn_blocks_rescore, dim3(qg), spart, h->nld, (const float*)(ws + p.off_bmax), p.bm_ld, p.nblk,
                            qf + (size_t)q0 * h->dp, qn + q0, qscale + q0, PLANE, h->norm64, h->n, h->dp, p.c, k, h->err_bound, force,
                            out_idx + (size_t)q0 * k, out_score + (size_t)q0 * k, out_score64 ? out_score64 + (size_t)q0 * k : nullptr,
                            nflag, flagged + q0, h->metric, h->bmax);
                finished = q0 + qg;                 // (groups come in order and only the last can be a tail below 64 queries)
            } else if (as_gemm && !no_stream) {     // a long row per query, >= 64 queries: one streaming block per query
                if (sel_bias || mask_g)     // (a third / fourth load per score: half the span keeps the registers)
                    hipLaunchKernelGGL((knn_select_stream<8, true>), dim3(qg), dim3(kSelThreads), 0, st, spart, h->nld, h->n, p.c,
                                       cidx + (size_t)q0 * 64, cs + (size_t)q0 * 64, sel_inv, sel_bias, qscale + q0, mask_g, mstride);
                else
                    hipLaunchKernelGGL((knn_select_stream<16, false>), dim3(qg), dim3(kSelThreads), 0, st, spart, h->nld, h->n, p.c,
                                       cidx + (size_t)q0 * 64, cs + (size_t)q0 * 64, sel_inv, sel_bias, qscale + q0, mask_g, mstride);
                ASTTS_CHECK_LAUNCH();
            } else {
                hipLaunchKernelGGL(knn_select, dim3(qg, p.nseg), dim3(kSelThreads), 0, st, spart, sel_ks, p.qpad, h->nld,
                                   h->n, p.c, p.seg_len, sidx, ss, sel_inv, sel_bias, qscale + q0, mask_g, mstride);
                ASTTS_CHECK_LAUNCH();
                hipLaunchKernelGGL(knn_select_merge, dim3(qg), dim3(64), 0, st, sidx, ss, p.nseg, p.c,
                                   cidx + (size_t)q0 * 64, cs + (size_t)q0 * 64);
                ASTTS_CHECK_LAUNCH();
            }
            if (!multi) break;
            const int done = r * kPassK;
            const int kp = (k - done) < kPassK ? (k - done) : kPassK;
            // this group's queries only: grid offset through the pointer arguments
            KNN_RESCORE(knn_rescore_finalize, dim3(qg), qf + (size_t)q0 * h->dp, qn + q0, qscale + q0,
                        PLANE, h->norm64, h->n, h->dp, p.c, kp,
                        cidx + (size_t)q0 * 64, cs + (size_t)q0 * 64, h->err_bound, force, out_idx + (size_t)q0 * k, out_score + (size_t)q0 * k,
                        out_score64 ? out_score64 + (size_t)q0 * k : nullptr, nflag, flagged + (size_t)r * nq + q0, h->metric, h->bmax,
                        mask_g, mstride, k, done);
            if (r + 1 < rounds) {
                hipLaunchKernelGGL(knn_mask_out, dim3(qg), dim3(64), 0, st, out_idx + (size_t)q0 * k, k, done, kp,
                                   (uint8_t*)(ws + p.off_mask) + (int64_t)q0 * h->n, h->n);
                ASTTS_CHECK_LAUNCH();
            }
        }
    }
    if (multi || finished == nq) return ASTTS_OK;
    {       // (the queries no fused launch has finished: all of them, or the tail group)
        const int f = finished;
        KNN_RESCORE(knn_rescore_finalize, dim3(nq - f), qf + (size_t)f * h->dp, qn + f, qscale + f, PLANE, h->norm64,
                    h->n, h->dp, p.c, k, cidx + (size_t)f * 64, cs + (size_t)f * 64, h->err_bound, force, out_idx + (size_t)f * k,
                    out_score + (size_t)f * k, out_score64 ? out_score64 + (size_t)f * k : nullptr, nflag, flagged + f, h->metric, h->bmax,
                    mask ? mask + (int64_t)f * mstride : nullptr, mstride, k, 0);
    }
#undef KNN_RESCORE
    return ASTTS_OK;
}

}  // namespace

extern "C" {

int astts_knn_search_masked(astts_knn_t* h, const float* queries, int32_t nq, int32_t k, int64_t* out_idx,
                            float* out_score, double* out_score64, const uint8_t* row_mask, int64_t mask_stride,
                            void* workspace, size_t workspace_bytes, int32_t flags, astts_stream_t stream) {
    ASTTS_REQUIRE(h != nullptr, ASTTS_ERR_INVALID, "astts_knn_search: handle is null");
    ASTTS_REQUIRE(queries && out_idx && out_score, ASTTS_ERR_INVALID, "astts_knn_search: null pointer argument");
    ASTTS_REQUIRE(nq >= 1, ASTTS_ERR_INVALID, "astts_knn_search: nq=%d", nq);
    ASTTS_REQUIRE(k >= 1 && k <= ASTTS_KNN_MAX_K, ASTTS_ERR_INVALID,
                  "astts_knn_search: k=%d (1..%d)", k, ASTTS_KNN_MAX_K);
    ASTTS_REQUIRE(row_mask == nullptr || mask_stride == 0 || mask_stride >= h->n, ASTTS_ERR_INVALID,
                  "astts_knn_search: mask_stride %lld (0 = one mask for every query, else >= n = %lld)", (long long)mask_stride, (long long)h->n);
    ASTTS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 255) == 0, ASTTS_ERR_WORKSPACE,
                  "astts_knn_search: workspace must be 256-byte aligned");
    const KnnPlan p = make_plan(h, nq, k);
    ASTTS_REQUIRE(workspace_bytes >= p.total, ASTTS_ERR_WORKSPACE,
                  "astts_knn_search: workspace %zu < required %zu", workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    if (p.passes == 1)
        return knn_search_chunk(h, p, queries, nq, k, out_idx, out_score, out_score64, row_mask, mask_stride, (char*)workspace, flags, true, st);
    for (int q0 = 0; q0 < nq; q0 += kMaxQPerPass) {       // k > 32: chunks of <= 256 queries share the workspace, in stream order
        const int qc = (nq - q0) < kMaxQPerPass ? (nq - q0) : kMaxQPerPass;
        const int rc = knn_search_chunk(h, p, queries + (size_t)q0 * h->d, qc, k, out_idx + (size_t)q0 * k, out_score + (size_t)q0 * k,
                                        out_score64 ? out_score64 + (size_t)q0 * k : nullptr,
                                        row_mask ? row_mask + (int64_t)q0 * mask_stride : nullptr, mask_stride, (char*)workspace, flags,
                                        q0 == 0, st);
        if (rc != ASTTS_OK) return rc;
    }
    return ASTTS_OK;
}

int astts_knn_profile_enable(astts_knn_t* h, int32_t on) {
    ASTTS_REQUIRE(h != nullptr, ASTTS_ERR_INVALID, "astts_knn_profile_enable: handle is null");
    if (on && h->ev.empty()) {
        h->ev.resize(2 * 8192);
        for (auto& e : h->ev) ASTTS_CHECK_HIP(hipEventCreate(&e));
    }
    h->profile = on != 0;
    h->ev_used = 0;
    return ASTTS_OK;
}

int astts_knn_profile_read(astts_knn_t* h, double* scan_ms_sum, int64_t* scan_launches) {
    ASTTS_REQUIRE(h && scan_ms_sum && scan_launches, ASTTS_ERR_INVALID, "astts_knn_profile_read: null argument");
    double sum = 0.0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        ASTTS_CHECK_HIP(hipEventSynchronize(h->ev[i + 1]));
        float ms = 0.f;
        ASTTS_CHECK_HIP(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        sum += ms;
    }
    *scan_ms_sum = sum;
    *scan_launches = (int64_t)(h->ev_used / 2);
    h->ev_used = 0;
    return ASTTS_OK;
}

int astts_knn_last_fallbacks(const astts_knn_t* h, const void* workspace, astts_stream_t stream,
                             int32_t* n_fallback_host) {
    ASTTS_REQUIRE(h && workspace && n_fallback_host, ASTTS_ERR_INVALID,
                  "astts_knn_last_fallbacks: null argument");
    hipStream_t st = (hipStream_t)stream;
    ASTTS_CHECK_HIP(hipMemcpyAsync(n_fallback_host, workspace, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ASTTS_CHECK_HIP(hipStreamSynchronize(st));
    return ASTTS_OK;
}

}  // extern "C"
