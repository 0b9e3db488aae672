// xlane.h -- cross-lane exchange without the LDS crossbar (gfx950).
//
// __shfl_xor compiles to ds_bpermute_b32: an LDS round trip (~100+ cycles) per butterfly step, and the steps of a reduction
// are dependent.  A 64-lane all-reduce is then ~0.3 us of pure latency; a decode GEMV normalises up to four rows per wave
// (two sums each) before its first MFMA.  The exchanges below are VALU operations (DPP modifiers inside a row of 16 lanes,
// v_permlane16_swap / v_permlane32_swap across rows -- both new in gfx950): a few cycles per step.
//
// lane_xor<OFF>(v) returns the value of lane (id ^ OFF) for ANY contents, so op(v, lane_xor<OFF>(v)) is exactly the
// butterfly step op(v, __shfl_xor(v, OFF)) and a reduction built from these steps produces the same bits as the shuffle
// version (fp32 addition is commutative; the pairing per step is the same).  Full waves only (every lane active).
#pragma once
#include <hip/hip_runtime.h>

namespace astts {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// v_permlane16_swap / v_permlane32_swap exchange rows between TWO registers: a = b = v in, (a', b') out with
//   permlane16_swap: a' = [r0 r0 r2 r2], b' = [r1 r1 r3 r3]   (odd rows of a <-> even rows of b; a row is 16 lanes)
//   permlane32_swap: a' = [lo lo],       b' = [hi hi]          (upper half of a <-> lower half of b)
// Inline assembly, not __builtin_amdgcn_permlane*_swap: called with the same value for both operands, hipcc (ROCm 7.2) treats
// the two results as one value (`a' + b'` was compiled to `2 * a'`).  The s_nop covers the VALU-write -> permlane-read hazard
// the compiler would otherwise pad by itself.
template <int ROWS>
__device__ __forceinline__ void row_swap(float v, float& a, float& b) {
    a = v;
    b = v;
    if constexpr (ROWS == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// the lane id itself (mbcnt), not threadIdx.x: the same in a 2-D block or one whose x extent is not a multiple of 64
__device__ __forceinline__ unsigned xlane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int OFF>
__device__ __forceinline__ float lane_xor(float v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "one bit of the lane id");
    if constexpr (OFF == 1) return dpp_mov<0xB1>(v);                    // quad_perm [1, 0, 3, 2]
    else if constexpr (OFF == 2) return dpp_mov<0x4E>(v);               // quad_perm [2, 3, 0, 1]
    else if constexpr (OFF == 4) return dpp_mov<0x141>(dpp_mov<0x1B>(v));   // quad reverse (id ^ 3), then row_half_mirror (id ^ 7)
    else if constexpr (OFF == 8) return dpp_mov<0x128>(v);              // row_ror:8
    else if constexpr (OFF == 16) {
        // lane ^ 16 wants [r1 r0 r3 r2]: b' on even rows, a' on odd rows
        float a, b;
        row_swap<16>(v, a, b);
        return (xlane_id() & 16) ? a : b;
    } else {
        float a, b;
        row_swap<32>(v, a, b);
        return (xlane_id() & 32) ? a : b;
    }
}

// the same exchange for 32-bit integers and doubles (two words), and with the offset as a run-time value that is a constant
// after unrolling (the bitonic sort of toplist.h): the switch folds away
template <int OFF> __device__ __forceinline__ int lane_xor(int v) { return __builtin_bit_cast(int, lane_xor<OFF>(__builtin_bit_cast(float, v))); }
template <int OFF> __device__ __forceinline__ double lane_xor(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const float lo = lane_xor<OFF>(__builtin_bit_cast(float, (unsigned)u)), hi = lane_xor<OFF>(__builtin_bit_cast(float, (unsigned)(u >> 32)));
    return __builtin_bit_cast(double, ((unsigned long long)__builtin_bit_cast(unsigned, hi) << 32) | __builtin_bit_cast(unsigned, lo));
}
template <typename T>
__device__ __forceinline__ T lane_xor_dyn(T v, int off) {
    switch (off) {
        case 1: return lane_xor<1>(v);
        case 2: return lane_xor<2>(v);
        case 4: return lane_xor<4>(v);
        case 8: return lane_xor<8>(v);
        case 16: return lane_xor<16>(v);
        default: return lane_xor<32>(v);
    }
}

// op(v, value of lane ^ OFF) for a COMMUTATIVE op: the row swaps need no select (a' op b' is the same pair on both sides)
template <int OFF, typename Op>
__device__ __forceinline__ float lane_xor_op(float v, Op op) {
    if constexpr (OFF == 16) {
        float a, b;
        row_swap<16>(v, a, b);
        return op(a, b);
    } else if constexpr (OFF == 32) {
        float a, b;
        row_swap<32>(v, a, b);
        return op(a, b);
    } else {
        return op(v, lane_xor<OFF>(v));
    }
}

struct XAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct XMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };

template <int OFF> __device__ __forceinline__ float xadd(float v) { return lane_xor_op<OFF>(v, XAdd()); }   // v + lane(id ^ OFF)
template <int OFF> __device__ __forceinline__ float xmax(float v) { return lane_xor_op<OFF>(v, XMax()); }

// all-reduce over the 64 lanes, butterfly steps 32, 16, ..., 1 (the order of `for (off = 32; off; off >>= 1) v += __shfl_xor(v, off)`)
__device__ __forceinline__ float wave_sum_desc(float v) {
    v = xadd<32>(v); v = xadd<16>(v); v = xadd<8>(v); v = xadd<4>(v); v = xadd<2>(v); v = xadd<1>(v);
    return v;
}


// L2 prefetch of a 128-byte line nobody waits for.  A normal load whose value is unused is deleted; one whose value is "used" (volatile, or
// consumed at the end of the kernel) is waited for -- volatile loads at once, the others by the first s_waitcnt vmcnt(0) the compiler
// places, and that stalls a kernel prologue on an HBM miss.  An asm load is invisible to the compiler's vmcnt bookkeeping: its waits for
// loads issued EARLIER never include this one (the counter is in order; an untracked younger load only makes vmcnt(N) wait for more of the
// older ones).  The destination register must stay reserved until the data has landed: pass it to prefetch_keep() at the end of the kernel.
__device__ __forceinline__ void prefetch_line(const void* p, unsigned& keep) { asm volatile("global_load_dword %0, %1, off" : "=v"(keep) : "v"(p)); }
__device__ __forceinline__ void prefetch_keep(unsigned keep) { asm volatile("" ::"v"(keep)); }

}  // namespace astts
