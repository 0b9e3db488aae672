// toplist.h -- wave-distributed sorted top-C list (shared by the kNN selection and the LM sampler).
#pragma once
#include "common.h"
#include "xlane.h"

namespace astts {

static constexpr int kNoIdx = 0x7fffffff;

// total order used everywhere: larger score first, ties -> smaller index first
template <typename T>
__device__ __forceinline__ bool better(T sa, int ia, T sb, int ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

// Sorted top-C list distributed over the lanes of one wave: lane i holds the i-th best entry.
template <typename T>
struct TopList {
    T s;
    int idx;
    __device__ __forceinline__ void init() {
        s = -INFINITY;
        idx = kNoIdx;
    }
    // bitonic sort of the 64 per-lane entries, best first
    __device__ __forceinline__ void sort_desc(int lane) {
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                T os = lane_xor_dyn(s, j);          // j is a constant after unrolling: DPP / permlane swap (xlane.h), not ds_bpermute
                int oi = lane_xor_dyn(idx, j);
                const bool up = (lane & k) == 0;     // this block ends best-first
                const bool lower = (lane & j) == 0;  // lower lane of the pair
                const bool other_better = better<T>(os, oi, s, idx);
                const bool take = (up == lower) ? other_better : !other_better;
                if (take) {
                    s = os;
                    idx = oi;
                }
            }
        }
    }
    // insert (xs, xi) into the sorted list of length c (lanes >= c are scratch)
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
    // every lane offers one entry; those that beat the current c-th best are inserted
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
    // first chunk of a wave: the list is empty, so sort the chunk instead of 64 serial inserts
    __device__ __forceinline__ void seed(T vs, int vi, bool valid, int lane) {
        s = valid ? vs : (T)-INFINITY;
        idx = valid ? vi : kNoIdx;
        sort_desc(lane);
    }
};

// merge the per-wave lists (staged in LDS) into wave 0's list
template <typename T>
__device__ __forceinline__ void merge_lists(TopList<T>& tl, T* sh_s, int* sh_i, int c) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nw = blockDim.x >> 6;
    sh_s[wid * 64 + lane] = tl.s;
    sh_i[wid * 64 + lane] = tl.idx;
    __syncthreads();
    if (wid == 0) {
        for (int w = 1; w < nw; ++w) {
            T v = sh_s[w * 64 + lane];
            int vi = sh_i[w * 64 + lane];
            tl.offer(v, vi, lane < c && vi != kNoIdx, lane, c);
        }
    }
}


}  // namespace astts
