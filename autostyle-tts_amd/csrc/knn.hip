// knn.hip -- brute-force cosine kNN over an HBM-resident style bank (gfx950 / MI355X).
//
// Replaces MilvusClient.search on the COSINE collection of the reference
// (/root/reference/milvus/search_embeddings.py:15-22, /root/reference/src/search_milvus.py:140-147).
// Result definition = oracle/knn.py: fp64 cosine, order (score desc, row asc).
//
// Pipeline (all on one stream, no host sync, no allocation):
//   1 knn_prep_queries   fp32 queries -> power-of-two scaled fp16 image + padded fp32 copy + fp64 norms
//   2 knn_scan           fp16 MFMA (32x32x16) scan of the whole bank: S[q][n] ~ <q,b_n>/|b_n|  (HBM-bound)
//   3 knn_select         per query: top-C candidates of S by (score desc, row asc)
//   4 knn_rescore        fp64 cosine of every candidate (one wave per candidate)
//   5 knn_finalize       order candidates by the fp64 score, emit top-k, CERTIFY the candidate set:
//                        kth exact score > best possible score of any non-candidate (+ error bound),
//                        otherwise queue the query for the exact path
//   6 knn_exact_scan     fp64 cosine of queued queries against every row   (normally zero work)
//   7 knn_exact_select   exact top-k for queued queries                     (normally zero work)
//
// HBM layout: scan plane fp16 [N][Dp] row-major, Dp = D rounded up to 64 (zero filled) so every row
// is a whole number of 128-byte lines; exact plane = the scan plane when the bank is fp16-exact,
// else fp32 [N][Dp]; fp64 row norms [N]; fp32 inverse norms [N].
#include "common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace astts {

static constexpr int kWave = 64;
static constexpr int kScanThreads = 256;
static constexpr int kQTile = 32;      // queries per MFMA tile
static constexpr int kMaxQPerPass = 256;

struct KnnCand64 {
    double s;
    int idx;
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool better(T sa, int ia, T sb, int ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

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
    double acc = 0.0;
    for (int k = lane * 8; k < dp; k += kWave * 8) {
        float4 q0 = *reinterpret_cast<const float4*>(q + k);
        float4 q1 = *reinterpret_cast<const float4*>(q + k + 4);
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
        acc = fma((double)q0.x, (double)b[0], acc);
        acc = fma((double)q0.y, (double)b[1], acc);
        acc = fma((double)q0.z, (double)b[2], acc);
        acc = fma((double)q0.w, (double)b[3], acc);
        acc = fma((double)q1.x, (double)b[4], acc);
        acc = fma((double)q1.y, (double)b[5], acc);
        acc = fma((double)q1.z, (double)b[6], acc);
        acc = fma((double)q1.w, (double)b[7], acc);
    }
    return wave_sum_f64(acc);
}

__device__ __forceinline__ double cos_from_parts(double dot, double qn, double bn) {
    double c = dot / (qn * bn);
    return isfinite(c) ? c : 0.0;
}

// ------------------------------------------------------------------------------------------
// bank construction
// ------------------------------------------------------------------------------------------
// one wave per row: copy/convert into the padded planes, fp64 norm, exactness + range flags
template <typename SrcT>
__global__ void knn_build_bank(const SrcT* __restrict__ src, int64_t n, int d, int dp,
                               _Float16* __restrict__ plane16, float* __restrict__ plane32,
                               double* __restrict__ norm64, float* __restrict__ inv_norm,
                               int* __restrict__ flags /* [0]=inexact, [1]=overflow */) {
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
        if (!isfinite(back) || !isfinite(v)) overflow = true;
        plane16[row * (int64_t)dp + k] = h;
        if (plane32) plane32[row * (int64_t)dp + k] = v;
        acc = fma((double)v, (double)v, acc);
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) {
        double nrm = sqrt(acc);
        norm64[row] = nrm;
        inv_norm[row] = nrm > 0.0 ? (float)(1.0 / nrm) : 0.0f;
    }
    if (__any(inexact) && lane == 0) atomicOr(&flags[0], 1);
    if (__any(overflow) && lane == 0) atomicOr(&flags[1], 1);
}

// ------------------------------------------------------------------------------------------
// 1. query preparation: one block per (padded) query row
// ------------------------------------------------------------------------------------------
__global__ void knn_prep_queries(const float* __restrict__ q, int nq, int d, int dp,
                                 _Float16* __restrict__ qh, float* __restrict__ qf,
                                 double* __restrict__ qn64, float* __restrict__ qscale) {
    __shared__ float smax[4];
    __shared__ double ssum[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    _Float16* oh = qh + (int64_t)row * dp;
    float* of = qf + (int64_t)row * dp;
    if (row >= nq) {  // padding rows of the last 32-query tile
        for (int k = tid; k < dp; k += blockDim.x) {
            oh[k] = (_Float16)0.0f;
            of[k] = 0.0f;
        }
        if (tid == 0) {
            qn64[row] = 0.0;
            qscale[row] = 1.0f;
        }
        return;
    }
    const float* s = q + (int64_t)row * d;
    float mx = 0.0f;
    double acc = 0.0;
    for (int k = tid; k < d; k += blockDim.x) {
        float v = s[k];
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
    double tot = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
    // power-of-two scale that puts max|q| in [2^13, 2^14): exact in fp32, keeps fp16 well inside
    // its normal range so the only rounding is the 11-bit significand
    float scale = 1.0f;
    if (mx > 0.0f && isfinite(mx)) {
        int e;
        frexpf(mx, &e);  // mx = m * 2^e, m in [0.5,1)
        scale = ldexpf(1.0f, 14 - e);
    }
    for (int k = tid; k < dp; k += blockDim.x) {
        float v = (k < d) ? s[k] : 0.0f;
        oh[k] = (_Float16)(v * scale);
        of[k] = v;
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
// ------------------------------------------------------------------------------------------
template <int QT, int RT>
__global__ __launch_bounds__(kScanThreads) void knn_scan(
    const _Float16* __restrict__ bank, const _Float16* __restrict__ qh,
    const float* __restrict__ inv_norm, float* __restrict__ s_part, int64_t n, int dp, int nld,
    int qpad, int lines_per_split) {
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

    const _Float16* bptr[RT];
#pragma unroll
    for (int b = 0; b < RT; ++b) {
        int64_t row = row0 + b * 32 + r;
        if (row >= n) row = n - 1;  // clamp: valid memory, masked at the store
        bptr[b] = bank + row * (int64_t)dp + h * 32;
    }
    const _Float16* aptr[QT];
#pragma unroll
    for (int a = 0; a < QT; ++a) aptr[a] = qh + (int64_t)(a * 32 + r) * dp + h * 32;

    for (int line = line_begin + wid; line < line_end; line += 4) {
        const int koff = line * 64;
        half8 bf[RT][4];
        half8 af[QT][4];
#pragma unroll
        for (int b = 0; b < RT; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                bf[b][i] = *reinterpret_cast<const half8*>(bptr[b] + koff + i * 8);
#pragma unroll
        for (int a = 0; a < QT; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[a][i] = *reinterpret_cast<const half8*>(aptr[a] + koff + i * 8);
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
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = acc[a][b][i];
#pragma unroll
                    for (int w = 0; w < 3; ++w)
                        v += red[(size_t)w * (QT * RT * 16 * 64) + ((a * RT + b) * 16 + i) * 64 + lane];
                    const int qrow = a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;  // MFMA row = query
                    if (col < nld)
                        s_part[((size_t)blockIdx.y * qpad + qrow) * nld + col] = v * inv;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// 3. selection: top-C of a score row by (score desc, row asc).  One block per query.
// Each wave keeps a sorted list distributed over its lanes (lane i = i-th best) and inserts by
// ballot; wave 0 then merges the four lists.
// ------------------------------------------------------------------------------------------
template <typename T>
struct TopList {
    T s;
    int idx;
    __device__ __forceinline__ void init() {
        s = -INFINITY;
        idx = 0x7fffffff;
    }
    // insert (xs, xi) into the wave-distributed sorted list of length c
    __device__ __forceinline__ void insert(T xs, int xi, int lane, int c) {
        const bool mine_better = better<T>(s, idx, xs, xi) && lane < c;
        const int pos = __popcll(__ballot(mine_better));
        T ups = __shfl_up(s, 1, 64);
        int upi = __shfl_up(idx, 1, 64);
        if (lane == pos) {
            s = xs;
            idx = xi;
        } else if (lane > pos) {
            s = ups;
            idx = upi;
        }
    }
    __device__ __forceinline__ void offer(T vs, int vi, bool valid, int lane, int c) {
        T ws = __shfl(s, c - 1, 64);
        int wi = __shfl(idx, c - 1, 64);
        unsigned long long mask = __ballot(valid && better<T>(vs, vi, ws, wi));
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            T xs = __shfl(vs, src, 64);
            int xi = __shfl(vi, src, 64);
            insert(xs, xi, lane, c);
        }
    }
};

template <typename T, typename LoadFn>
__device__ __forceinline__ void block_select(LoadFn load, int64_t n, int c, T* sh_s, int* sh_i,
                                             TopList<T>& out) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nw = blockDim.x >> 6;
    TopList<T> tl;
    tl.init();
    for (int64_t base = (int64_t)wid * 64; base < n; base += (int64_t)nw * 64) {
        const int64_t i = base + lane;
        const bool valid = i < n;
        T v = valid ? load(i) : (T)-INFINITY;
        tl.offer(v, (int)i, valid, lane, c);
    }
    sh_s[wid * 64 + lane] = tl.s;
    sh_i[wid * 64 + lane] = tl.idx;
    __syncthreads();
    if (wid == 0) {
        for (int w = 1; w < nw; ++w) {
            T v = sh_s[w * 64 + lane];
            int vi = sh_i[w * 64 + lane];
            tl.offer(v, vi, lane < c && vi != 0x7fffffff, lane, c);
        }
    }
    out = tl;
}

__global__ __launch_bounds__(256) void knn_select(const float* __restrict__ s_part, int ksplit,
                                                  int qpad, int nld, int64_t n, int c,
                                                  int* __restrict__ cand_idx,
                                                  float* __restrict__ cand_s) {
    __shared__ float sh_s[256];
    __shared__ int sh_i[256];
    const int q = blockIdx.x;
    auto load = [&](int64_t i) {
        float v = 0.0f;
        for (int ks = 0; ks < ksplit; ++ks) v += s_part[((size_t)ks * qpad + q) * nld + i];
        return v;
    };
    TopList<float> tl;
    block_select<float>(load, n, c, sh_s, sh_i, tl);
    if (threadIdx.x < c) {
        cand_idx[q * 64 + threadIdx.x] = (tl.idx == 0x7fffffff) ? -1 : tl.idx;
        cand_s[q * 64 + threadIdx.x] = tl.s;
    }
}

// ------------------------------------------------------------------------------------------
// 4. fp64 re-score: one wave per (query, candidate)
// ------------------------------------------------------------------------------------------
template <typename RowT>
__global__ __launch_bounds__(64) void knn_rescore(const float* __restrict__ qf,
                                                  const double* __restrict__ qn64,
                                                  const RowT* __restrict__ plane,
                                                  const double* __restrict__ norm64, int dp, int c,
                                                  const int* __restrict__ cand_idx,
                                                  double* __restrict__ cand_cos) {
    const int q = blockIdx.y, ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= c) return;
    const int idx = cand_idx[q * 64 + ci];
    if (idx < 0) {
        if (lane == 0) cand_cos[q * 64 + ci] = -INFINITY;
        return;
    }
    double dot = wave_dot64<RowT>(qf + (int64_t)q * dp, plane + (int64_t)idx * dp, dp, lane);
    if (lane == 0) cand_cos[q * 64 + ci] = cos_from_parts(dot, qn64[q], norm64[idx]);
}

// ------------------------------------------------------------------------------------------
// 5. finalize: order by fp64 score, emit top-k, certify, queue uncertified queries
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void knn_finalize(const int* __restrict__ cand_idx,
                                                   const float* __restrict__ cand_s,
                                                   const double* __restrict__ cand_cos,
                                                   const double* __restrict__ qn64,
                                                   const float* __restrict__ qscale, int64_t n,
                                                   int c, int k, double err_bound, int force_exact,
                                                   int64_t* __restrict__ out_idx,
                                                   float* __restrict__ out_score,
                                                   int* __restrict__ nflag,
                                                   int* __restrict__ flagged) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const bool valid = lane < c;
    int idx = valid ? cand_idx[q * 64 + lane] : -1;
    double cs = (valid && idx >= 0) ? cand_cos[q * 64 + lane] : -INFINITY;
    float ap = (valid && idx >= 0) ? cand_s[q * 64 + lane] : INFINITY;
    const bool live = valid && idx >= 0;
    int rank = 0;
    for (int j = 0; j < c; ++j) {
        double sj = __shfl(cs, j, 64);
        int ij = __shfl(idx, j, 64);
        if (ij >= 0 && j != lane && better<double>(sj, ij, cs, idx)) ++rank;
    }
    const int kk = (int64_t)k < n ? k : (int)n;  // hits that exist
    if (live && rank < kk) {
        out_idx[(int64_t)q * k + rank] = idx;
        out_score[(int64_t)q * k + rank] = (float)cs;
    }
    if (lane >= kk && lane < k) {  // fewer than k rows in the bank
        out_idx[(int64_t)q * k + lane] = -1;
        out_score[(int64_t)q * k + lane] = -INFINITY;
    }
    // certification
    float tau = ap;  // min approx over live candidates
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tau = fminf(tau, __shfl_xor(tau, off, 64));
    double kth = (live && rank == kk - 1) ? cs : -INFINITY;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) kth = fmax(kth, __shfl_xor(kth, off, 64));
    if (lane == 0) {
        bool certified;
        if (n <= (int64_t)c) {
            certified = true;  // every row is a candidate
        } else {
            const double denom = (double)qscale[q] * qn64[q];
            const double tau_cos = denom > 0.0 ? (double)tau / denom : INFINITY;
            certified = isfinite(tau_cos) && (kth > tau_cos + err_bound);
        }
        if (!certified || force_exact) {
            int slot = atomicAdd(nflag, 1);
            flagged[slot] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------
// 6./7. exact path for queued queries
// ------------------------------------------------------------------------------------------
template <typename RowT>
__global__ __launch_bounds__(256) void knn_exact_scan(const float* __restrict__ qf,
                                                      const double* __restrict__ qn64,
                                                      const RowT* __restrict__ plane,
                                                      const double* __restrict__ norm64, int64_t n,
                                                      int dp, int nld, const int* __restrict__ nflag,
                                                      const int* __restrict__ flagged,
                                                      double* __restrict__ s64) {
    const int nf = *nflag;
    if (nf == 0) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;
    if (row >= n) return;
    const RowT* rp = plane + row * (int64_t)dp;
    const double bn = norm64[row];
    for (int f = 0; f < nf; ++f) {
        const int q = flagged[f];
        double dot = wave_dot64<RowT>(qf + (int64_t)q * dp, rp, dp, lane);
        if (lane == 0) s64[(size_t)f * nld + row] = cos_from_parts(dot, qn64[q], bn);
    }
}

__global__ __launch_bounds__(256) void knn_exact_select(const double* __restrict__ s64, int nld,
                                                        int64_t n, int k,
                                                        const int* __restrict__ nflag,
                                                        const int* __restrict__ flagged,
                                                        int64_t* __restrict__ out_idx,
                                                        float* __restrict__ out_score) {
    __shared__ double sh_s[256];
    __shared__ int sh_i[256];
    const int f = blockIdx.x;
    if (f >= *nflag) return;
    const int q = flagged[f];
    const double* row = s64 + (size_t)f * nld;
    auto load = [&](int64_t i) { return row[i]; };
    TopList<double> tl;
    block_select<double>(load, n, k, sh_s, sh_i, tl);
    if (threadIdx.x < k) {
        const bool ok = tl.idx != 0x7fffffff;
        out_idx[(int64_t)q * k + threadIdx.x] = ok ? tl.idx : -1;
        out_score[(int64_t)q * k + threadIdx.x] = ok ? (float)tl.s : -INFINITY;
    }
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
    _Float16* plane16 = nullptr;  // [n][dp]
    float* plane32 = nullptr;     // [n][dp], only when !exact16
    double* norm64 = nullptr;     // [n]
    float* inv_norm = nullptr;    // [n]
    double err_bound = 0.0;
    // bench-only profiling (astts_knn_profile_*)
    bool profile = false;
    std::vector<hipEvent_t> ev;  // pairs (start, stop)
    size_t ev_used = 0;
};

namespace {

struct KnnPlan {
    int qt, rt, ksplit, lines_per_split, tiles, qpad, c;
    size_t off_nflag, off_flagged, off_qh, off_qf, off_qn, off_qscale, off_spart, off_cidx, off_cs,
        off_ccos, off_s64, total;
};

KnnPlan make_plan(const astts_knn* h, int nq, int k) {
    KnnPlan p{};
    const int qgroup = nq < kMaxQPerPass ? nq : kMaxQPerPass;
    p.qt = qgroup <= 32 ? 1 : qgroup <= 64 ? 2 : qgroup <= 128 ? 4 : 8;
    p.qpad = p.qt * 32;
    // rows per wave tile: reuse each query fragment over several bank tiles once the bank is large
    p.rt = (h->n >= 16384) ? (p.qt == 1 ? 4 : p.qt == 2 ? 2 : 1) : 1;
    p.tiles = (int)cdiv(h->n, 32 * p.rt);
    const int total_lines = h->dp / 64;
    int ks = (int)cdiv(512, p.tiles);
    int ks_max = total_lines / 4;
    if (ks_max < 1) ks_max = 1;
    if (ks > ks_max) ks = ks_max;
    if (ks < 1) ks = 1;
    p.lines_per_split = (int)cdiv(total_lines, ks);
    p.ksplit = (int)cdiv(total_lines, p.lines_per_split);
    p.c = k <= 8 ? 16 : 64;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + bytes, 256);
        return at;
    };
    const int nqpad_all = (int)align_up((size_t)nq, 32) + 32 * 8;  // room for the last group's tile padding
    p.off_nflag = take(256);
    p.off_flagged = take(sizeof(int) * (size_t)nq);
    p.off_qh = take(sizeof(_Float16) * (size_t)nqpad_all * h->dp);
    p.off_qf = take(sizeof(float) * (size_t)nqpad_all * h->dp);
    p.off_qn = take(sizeof(double) * (size_t)nqpad_all);
    p.off_qscale = take(sizeof(float) * (size_t)nqpad_all);
    p.off_spart = take(sizeof(float) * (size_t)p.ksplit * p.qpad * h->nld);
    p.off_cidx = take(sizeof(int) * (size_t)nq * 64);
    p.off_cs = take(sizeof(float) * (size_t)nq * 64);
    p.off_ccos = take(sizeof(double) * (size_t)nq * 64);
    p.off_s64 = take(sizeof(double) * (size_t)nq * h->nld);
    p.total = o;
    return p;
}

template <int QT, int RT>
int launch_scan(const astts_knn* h, const KnnPlan& p, const _Float16* qh, float* spart,
                hipStream_t st) {
    dim3 grid(p.tiles, p.ksplit);
    size_t lds = (size_t)3 * QT * RT * 16 * 64 * sizeof(float);
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_scan<QT, RT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return ASTTS_ERR_HIP;
            }
            once = true;
        }
    }
    hipLaunchKernelGGL((knn_scan<QT, RT>), grid, dim3(kScanThreads), lds, st, h->plane16, qh,
                       h->inv_norm, spart, h->n, h->dp, h->nld, p.qpad, p.lines_per_split);
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
    ASTTS_REQUIRE(n >= 1 && n <= 0x7fffffff - 1024, ASTTS_ERR_INVALID,
                  "astts_knn_create: n=%lld out of range", (long long)n);
    ASTTS_REQUIRE(d >= 1 && d <= (1 << 20), ASTTS_ERR_INVALID, "astts_knn_create: d=%d out of range", d);
    ASTTS_REQUIRE(dtype == ASTTS_DTYPE_F16 || dtype == ASTTS_DTYPE_F32, ASTTS_ERR_INVALID,
                  "astts_knn_create: dtype %d (want ASTTS_DTYPE_F16|F32)", dtype);
    if (metric != ASTTS_METRIC_COSINE) {
        set_error("astts_knn_create: metric %d not implemented (COSINE only, as the reference collection)", metric);
        return ASTTS_ERR_UNSUPPORTED;
    }
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
    KNN_TRY(hipMalloc(&h->plane16, sizeof(_Float16) * (size_t)n * h->dp));
    KNN_TRY(hipMalloc(&h->norm64, sizeof(double) * (size_t)n));
    KNN_TRY(hipMalloc(&h->inv_norm, sizeof(float) * (size_t)n));
    KNN_TRY(hipMalloc(&flags, 2 * sizeof(int)));
    KNN_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(int), st));
    const int rows_per_block = 4;
    dim3 grid((unsigned)cdiv(n, rows_per_block));
    if (dtype == ASTTS_DTYPE_F32) {
        KNN_TRY(hipMalloc(&p32, sizeof(float) * (size_t)n * h->dp));
        hipLaunchKernelGGL((knn_build_bank<float>), grid, dim3(256), 0, st, (const float*)bank, n, d,
                           h->dp, h->plane16, p32, h->norm64, h->inv_norm, flags);
    } else {
        hipLaunchKernelGGL((knn_build_bank<_Float16>), grid, dim3(256), 0, st, (const _Float16*)bank,
                           n, d, h->dp, h->plane16, (float*)nullptr, h->norm64, h->inv_norm, flags);
    }
    KNN_TRY(hipGetLastError());
    int hf[2] = {0, 0};
    KNN_TRY(hipMemcpyAsync(hf, flags, sizeof(hf), hipMemcpyDeviceToHost, st));
    KNN_TRY(hipStreamSynchronize(st));
    if (hf[1]) {
        set_error("astts_knn_create: bank holds values that are non-finite or overflow fp16");
        return fail(ASTTS_ERR_RANGE);
    }
    h->exact16 = (hf[0] == 0);
    if (!h->exact16) {
        h->plane32 = p32;  // keep the fp32 image for exact re-scoring
        p32 = nullptr;
    }
    (void)hipFree(flags);
    flags = nullptr;
    if (p32) {
        (void)hipFree(p32);
        p32 = nullptr;
    }
#undef KNN_TRY
    // Error bound of the fp16 scan on the cosine scale (see DESIGN.md "certification"):
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
    if (h->plane16) (void)hipFree(h->plane16);
    if (h->plane32) (void)hipFree(h->plane32);
    if (h->norm64) (void)hipFree(h->norm64);
    if (h->inv_norm) (void)hipFree(h->inv_norm);
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
    ASTTS_REQUIRE(h != nullptr, ASTTS_ERR_INVALID, "astts_knn_search: handle is null");
    ASTTS_REQUIRE(queries && out_idx && out_score, ASTTS_ERR_INVALID, "astts_knn_search: null pointer argument");
    ASTTS_REQUIRE(nq >= 1, ASTTS_ERR_INVALID, "astts_knn_search: nq=%d", nq);
    ASTTS_REQUIRE(k >= 1 && k <= ASTTS_KNN_MAX_K, ASTTS_ERR_INVALID,
                  "astts_knn_search: k=%d (1..%d)", k, ASTTS_KNN_MAX_K);
    ASTTS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 255) == 0, ASTTS_ERR_WORKSPACE,
                  "astts_knn_search: workspace must be 256-byte aligned");
    const KnnPlan p = make_plan(h, nq, k);
    ASTTS_REQUIRE(workspace_bytes >= p.total, ASTTS_ERR_WORKSPACE,
                  "astts_knn_search: workspace %zu < required %zu", workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int* nflag = (int*)(ws + p.off_nflag);
    int* flagged = (int*)(ws + p.off_flagged);
    _Float16* qh = (_Float16*)(ws + p.off_qh);
    float* qf = (float*)(ws + p.off_qf);
    double* qn = (double*)(ws + p.off_qn);
    float* qscale = (float*)(ws + p.off_qscale);
    float* spart = (float*)(ws + p.off_spart);
    int* cidx = (int*)(ws + p.off_cidx);
    float* cs = (float*)(ws + p.off_cs);
    double* ccos = (double*)(ws + p.off_ccos);
    double* s64 = (double*)(ws + p.off_s64);

    ASTTS_CHECK_HIP(hipMemsetAsync(nflag, 0, 256, st));
    const int nq_rows = (int)align_up((size_t)nq, 32) + 32 * 8;  // also zero the tile padding rows
    hipLaunchKernelGGL(knn_prep_queries, dim3(nq_rows), dim3(256), 0, st, queries, nq, h->d, h->dp,
                       qh, qf, qn, qscale);
    ASTTS_CHECK_LAUNCH();

    for (int q0 = 0; q0 < nq; q0 += kMaxQPerPass) {
        const int qg = (nq - q0) < kMaxQPerPass ? (nq - q0) : kMaxQPerPass;
        const _Float16* qh_g = qh + (size_t)q0 * h->dp;
        int rc;
        const bool prof = h->profile && h->ev_used + 2 <= h->ev.size();
        if (prof) ASTTS_CHECK_HIP(hipEventRecord(h->ev[h->ev_used], st));
        switch (p.qt * 10 + p.rt) {
            case 11: rc = launch_scan<1, 1>(h, p, qh_g, spart, st); break;
            case 14: rc = launch_scan<1, 4>(h, p, qh_g, spart, st); break;
            case 21: rc = launch_scan<2, 1>(h, p, qh_g, spart, st); break;
            case 22: rc = launch_scan<2, 2>(h, p, qh_g, spart, st); break;
            case 41: rc = launch_scan<4, 1>(h, p, qh_g, spart, st); break;
            case 81: rc = launch_scan<8, 1>(h, p, qh_g, spart, st); break;
            default:
                set_error("astts_knn_search: no scan variant for qt=%d rt=%d", p.qt, p.rt);
                return ASTTS_ERR_INVALID;
        }
        if (rc != ASTTS_OK) return rc;
        if (prof) {
            ASTTS_CHECK_HIP(hipEventRecord(h->ev[h->ev_used + 1], st));
            h->ev_used += 2;
        }
        hipLaunchKernelGGL(knn_select, dim3(qg), dim3(256), 0, st, spart, p.ksplit, p.qpad, h->nld,
                           h->n, p.c, cidx + (size_t)q0 * 64, cs + (size_t)q0 * 64);
        ASTTS_CHECK_LAUNCH();
    }
    if (h->exact16) {
        hipLaunchKernelGGL((knn_rescore<_Float16>), dim3(p.c, nq), dim3(64), 0, st, qf, qn, h->plane16,
                           h->norm64, h->dp, p.c, cidx, ccos);
    } else {
        hipLaunchKernelGGL((knn_rescore<float>), dim3(p.c, nq), dim3(64), 0, st, qf, qn, h->plane32,
                           h->norm64, h->dp, p.c, cidx, ccos);
    }
    ASTTS_CHECK_LAUNCH();
    hipLaunchKernelGGL(knn_finalize, dim3(nq), dim3(64), 0, st, cidx, cs, ccos, qn, qscale, h->n, p.c,
                       k, h->err_bound, (flags & ASTTS_KNN_FORCE_EXACT) ? 1 : 0, out_idx, out_score,
                       nflag, flagged);
    ASTTS_CHECK_LAUNCH();
    dim3 egrid((unsigned)cdiv(h->n, 4));
    if (h->exact16) {
        hipLaunchKernelGGL((knn_exact_scan<_Float16>), egrid, dim3(256), 0, st, qf, qn, h->plane16,
                           h->norm64, h->n, h->dp, h->nld, nflag, flagged, s64);
    } else {
        hipLaunchKernelGGL((knn_exact_scan<float>), egrid, dim3(256), 0, st, qf, qn, h->plane32,
                           h->norm64, h->n, h->dp, h->nld, nflag, flagged, s64);
    }
    ASTTS_CHECK_LAUNCH();
    hipLaunchKernelGGL(knn_exact_select, dim3(nq), dim3(256), 0, st, s64, h->nld, h->n, k, nflag,
                       flagged, out_idx, out_score);
    ASTTS_CHECK_LAUNCH();
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
