// lm_step.hip -- decode-step kernels of the acoustic transformer (see lm_step.h for the design rules).
//
//   lm_gemv<MT, FORM, XM, NL>  out[m, n] = epilogue( LN?(x)[m, :] . W[n, :] ),  m <= 16*MT rows (the decode batch)
//   lm_attn                    one new query per (row, head) against the fp16 KV cache, relative-position scores
//
// Arithmetic follows the operator-by-operator step (AcousticLM.step_logits / gemm_skinny16 / attn_relpos_decode): fp16
// MFMA operands, fp32 accumulation, fp32 LayerNorm / softmax.  Only the fp32 summation ORDER of a dot product differs
// between the variants (which K lines a wave owns), never which products are formed.
#include "lm_step.h"
#include "xlane.h"

#include <hip/hip_ext.h>

#include <cstdlib>

namespace astts {

#ifdef LM_STAMPS
#define LM_STAMP(a, i)                                                                                              \
    do {                                                                                                            \
        if ((a).stamps && threadIdx.x == 0)                                                                         \
            (a).stamps[((size_t)(a).stamp_slot * 1024 + (blockIdx.x + blockIdx.y * gridDim.x)) * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define LM_STAMP(a, i) do { } while (0)
#endif

// Wave priority.  A decode-step wave that shares a SIMD with a wave of another stream's kernel (the render stage of an earlier
// batch: co-resident whenever LDS and registers allow) loses the VALU-issue arbitration to it at equal priority (older wave wins):
// beside a background that keeps every SIMD issuing, the decode step went 394 -> 755 us in the chain benchmark
// (scripts/micro/decode_chain.hip, DC_BG=1); with s_setprio 3 at kernel entry: 413 us.  The kernels are a few microseconds long and
// mostly waiting for memory, so the background loses next to nothing.  -DLM_SETPRIO=0 builds without it (A/B).
#ifndef LM_SETPRIO
#define LM_SETPRIO 3
#endif
#define LM_RAISE_PRIO() do { if (LM_SETPRIO) __builtin_amdgcn_s_setprio(LM_SETPRIO); } while (0)

// Kernel arguments: hipcc loads the fields of a by-value argument struct lazily, each s_load_dword right before its first
// use and followed by its own s_waitcnt -- in a kernel with this much control flow that is ~15 SERIALISED scalar-cache
// misses on a cold kernarg segment (measured: 1.4 us from the first instruction to the last up-front load issued, a third
// of the kernel).  Pinning every field into SGPRs in the entry block turns them into a few wide s_loads and ONE wait.
// (one asm statement per group: every separate statement gets its own wait)
__device__ __forceinline__ void pin_args(const GemvArgs& a) {
    asm volatile("" ::"s"(a.x), "s"(a.x2), "s"(a.gather), "s"(a.pre_g), "s"(a.pre_b), "s"(a.pre_out), "s"(a.ln_g), "s"(a.ln_b), "s"(a.w),
                 "s"(a.bias), "s"(a.res), "s"(a.out), "s"(a.out16), "s"(a.kv), "s"(a.st), "s"(a.m), "s"(a.n), "s"(a.k), "s"(a.kpad), "s"(a.ldx),
                 "s"(a.ldr), "s"(a.ldo), "s"(a.ldo16), "s"(a.n_split), "s"(a.kv_t), "s"(a.kv_b), "s"(a.kv_h), "s"(a.kv_v), "s"(a.x_mode), "s"(a.relu), "s"(a.pos), "s"(a.ln_plain),
                 "s"(a.zero), "s"(a.zero_n), "s"(a.advance), "s"(a.stamp_slot), "s"(a.ksplit), "s"(a.stamps), "s"(a.ln_eps), "s"(a.pre_scale)
                 : "memory");     // EVERY field is listed: a field left out is dead on arrival, its SGPR is reused at once, and a write to a register
                                  // whose scalar load is still in flight costs an s_waitcnt lgkmcnt(0) on this cold miss -- it sat in front of the
                                  // weight loads.  The clobber keeps the loads issued before the pin (input rows, weight lines) before it.
}
__device__ __forceinline__ void pin_args(const AttnArgs& a) {
    asm volatile("" ::"s"(a.q), "s"(a.kv), "s"(a.bias_u), "s"(a.bias_v), "s"(a.kstart), "s"(a.out), "s"(a.part_o), "s"(a.part_ml),
                 "s"(a.st), "s"(a.h), "s"(a.ldq), "s"(a.ldo), "s"(a.ldp), "s"(a.kv_t), "s"(a.kv_b), "s"(a.kv_h), "s"(a.kv_v), "s"(a.scale), "s"(a.ksplit), "s"(a.pos));
}

static constexpr int GV_WAVES = 8;

// Weight lines are read once per launch by one workgroup.  -DLM_NT_WEIGHTS loads them with the non-temporal policy (the guide's
// "nt-weights" row); A/B in scripts/micro/decode_chain.hip.
#ifdef LM_NT_WEIGHTS
#define LM_WLOAD(p) __builtin_nontemporal_load(reinterpret_cast<const half8*>(p))
#else
#define LM_WLOAD(p) (*reinterpret_cast<const half8*>(p))
#endif

// sum over the wave, butterfly order 32, 16, ..., 1 -- as VALU lane exchanges (xlane.h), not ds_bpermute round trips: same bits
__device__ __forceinline__ float wsum64(float v) { return wave_sum_desc(v); }

// FORM 0: 16 columns per workgroup, every wave walks whole K lines (MT row tiles of 16).
// FORM 1 ("diagonal", m <= 8): MFMA row i = r + 8h carries x[r][h*Kc + ...], MFMA column j = c + 8h' carries
// W[n0 + c][h'*Kc + ...] (Kc = kpad / 2): the two diagonal 8x8 blocks of the 16x16 product are the two K halves of an
// 8-row x 8-column tile, the off-diagonal blocks are never read.  The block owns 8 columns instead of 16.
// FORM 2 ("halved", m <= 16 MT): the same 8-column weight fragments (column j = c + 8h' carries K half h'), one MFMA per K
// half of the rows: accumulator [t][h] = x[rows of tile t][half h] . W^T, of which the columns of half h are used.  Two (MT
// = 1) or four (MT = 2) MFMAs per weight fragment, half of each product unused -- the matrix pipe is idle in this kernel
// anyway; what it buys is the 8-column grid (twice the workgroups) for 16- and 32-row batches.  A row's sum is formed in
// exactly the order of FORM 1 (per wave: its lines in ascending order per K half; then waves 0..7, half 0 + half 1 each), so
// a row's result does not depend on how many rows share the launch: 8-, 16- and 32-row batches agree bit for bit.
// XM: input form -- 0 fp32 rows (optional gather / embedding pre-transform / LayerNorm), 1 fp16 rows, 2 the split-key
// partials of lm_attn (merged while staging).  NL: weight lines a wave keeps in flight (2 for K <= 1024 at 16 columns: the
// kernel then fits two workgroups per CU, e.g. the 257 workgroups of the output head in one wave of blocks).
// NCT (FORM 0, MT 1 only): column tiles of 16 per workgroup.  NCT = 2 ("wide"): 32 columns, the A fragments of the staged rows feed two
// weight fragments each -- half the workgroups of the 16-column grid for the wide projections (QKV, FFN-in, head) at the same number of
// memory round trips.  Inside the stream pipeline a decode kernel costs the chip (CUs it holds) x (time it holds them); the sums of a
// column are formed exactly as with NCT = 1 (per wave: its K lines in order; then waves 0..7): bit-identical results.
template <int MT, int FORM, int XM, int NL, int NCT = 1>
__global__ __launch_bounds__(512, (NCT == 1 && NL <= 2 && (FORM == 1 || (FORM == 0 && MT == 1 && XM != 2))) ? 4 : 2) void lm_gemv(const void* p_x, const _Float16* p_w, const float* p_x2, const int* p_gather, int p_m, int p_n, int p_k, int p_kpad, int p_ldx, int p_ks,
                                                                                                                GemvArgs a_in) {
    // The first 14 dwords of the kernarg segment (the most the hardware preloads) are what the FIRST global loads need (input rows, weight lines).  As explicit
    // leading parameters they are PRELOADED into SGPRs by the command processor (-amdgpu-kernarg-preload-count, Makefile): the
    // wave issues those loads without waiting for an s_load of its arguments (a cold scalar-cache miss, ~0.5-1 us in a kernel
    // that lasts 3-5).  The struct behind them carries the same fields again (ignored) and everything else.
    LM_RAISE_PRIO();
    // Until the up-front loads are out, `a` holds the leading parameters only (the rest zero).  The struct itself is read through a pointer the
    // compiler cannot see through before that point (LM_LATE_ARGS below): read at the top -- where the scheduler hoists independent scalar
    // loads -- its ~60 destination SGPRs are in flight while the kernel computes its load addresses, the kernel uses every SGPR there is,
    // and the first temporary that lands in one of them costs an s_waitcnt lgkmcnt(0) on a cold scalar miss: in front of the weight loads.
#if !defined(LM_LATE_ARGS) || defined(LM_STAMPS) || !defined(__HIP_DEVICE_COMPILE__)
    GemvArgs a = a_in;
#else
    GemvArgs a{};
#endif
    a.x = p_x; a.w = p_w; a.x2 = p_x2; a.gather = p_gather; a.m = p_m; a.n = p_n; a.k = p_k; a.kpad = p_kpad; a.ldx = p_ldx;
    const int wld = p_kpad;                                   // weight row stride
    if constexpr (XM == 1) {
        if (p_ks > 1) {                                       // K slice blockIdx.y of a split projection (GemvArgs::ksplit; preloaded: gridDim.y
                                                              // is an implicit argument = a scalar load + wait in front of the first loads)
            const int kh = p_kpad >> 1;
            a.x = reinterpret_cast<const _Float16*>(p_x) + (int)blockIdx.y * kh;
            a.w = p_w + (int)blockIdx.y * kh;
            a.k = kh; a.kpad = kh;
        }
    }
    constexpr bool XF16 = XM != 0;
    constexpr bool HALF8 = FORM != 0;                          // K halved over MFMA columns, 8 output columns per workgroup
    constexpr bool DIAG = FORM == 1;
    constexpr int RH = DIAG ? 8 : 16 * MT;                     // LDS row offset of the second K half
    static_assert(NCT == 1 || (FORM == 0 && MT == 1), "column tiles: 16-column form, one row tile");
    constexpr int NACC = FORM == 2 ? 2 * MT : MT * NCT;        // accumulators per wave
    extern __shared__ __attribute__((aligned(16))) char gv_smem[];
    LM_STAMP(a, 0);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    constexpr int NC = HALF8 ? 8 : 16 * NCT;
    const int n0 = blockIdx.x * NC;
    const int M = a.m;
    const int Kc = HALF8 ? (a.kpad >> 1) : a.kpad;          // K elements per MFMA row
    const int lines = Kc >> 6;
    const int xs = Kc + 8;                                    // LDS row stride in halfs (16-byte skew)
    const int lrows = HALF8 ? 2 * RH : M;
    _Float16* sx = reinterpret_cast<_Float16*>(gv_smem);      // [lrows][xs]
    float* red = reinterpret_cast<float*>(gv_smem + (((size_t)lrows * xs * 2 + 15) & ~(size_t)15));  // [8][NACC][4][64]

    // ---- (0) everything this block will need from memory, issued before anything waits
    // weights: lane (c, g) owns bytes [32g, 32g + 32) of its row in every 128-byte line (two 16x16x32 k-steps)
    const _Float16* wrow = HALF8 ? a.w + (int64_t)(n0 + (c & 7)) * wld + (c >> 3) * Kc + g * 16
                                 : a.w + (int64_t)(n0 + c) * wld + g * 16;
    // input row of this wave (fp32 path, K <= 1024: 4 float4 per lane) -- needed first, so issued first
    constexpr int MAXV = 4;
    constexpr int RPW = DIAG ? 1 : 2 * MT;                    // rows a wave stages (rows wid, wid + 8, ...): ALL prefetched up front
    const bool vec_ok = !XF16 && a.k <= 1024 && (a.k & 3) == 0 && (a.ldx & 3) == 0 && ((uintptr_t)a.x & 15) == 0;
    const int nv = (a.kpad + 255) >> 8;
    float4 v0[XF16 ? 1 : RPW][MAXV];
    if constexpr (!XF16) {
        if (vec_ok) {
            // Every load of every row of this wave is UNCONDITIONAL (row index clamped to the last row, K position clamped into the row,
            // zeroed afterwards): loads inside `if (row < m)` / `if (k < K)` blocks, or behind the wait for a gathered row index, are
            // waited for block by block -- the RPW rows of a wave (2 at 16 rows, 4 at 32) were as many dependent round trips.
            auto load_rows = [&](const int64_t (&srcs)[RPW]) {
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const float* xr = reinterpret_cast<const float*>(a.x) + srcs[rr] * a.ldx;
#pragma unroll
                    for (int i = 0; i < MAXV; ++i) v0[rr][i] = *reinterpret_cast<const float4*>(xr + min(lane * 4 + i * 256, a.k - 4));
                }
            };
            int64_t srcs[RPW];
            if (a.gather) {
                int gi[RPW];
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) gi[rr] = a.gather[min(wid + rr * GV_WAVES, M - 1)];
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) srcs[rr] = gi[rr];
                load_rows(srcs);
            } else {
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) srcs[rr] = min(wid + rr * GV_WAVES, M - 1);
                load_rows(srcs);
            }
        }
    }
    // fp16 input (already an MFMA operand image: FFN hidden; XM == 2: the attention partials, merged here): 16-byte
    // pieces of 8 K positions, up to 8 per thread
    // (the halved form takes fp16 rows by LDS-DMA instead: no staging registers, every piece of every row in flight at once)
    constexpr int XP = XM == 2 ? (DIAG ? 2 : 4) : 8;
    const bool x_dma = XM == 1 && FORM == 2 && a.k == a.kpad && (Kc & 511) == 0;
    half8 xh[XM == 1 ? XP : 1];
    float4 po[XM == 2 ? XP : 1][2][2];                         // [piece][split][8 floats]
    float2 pml[XM == 2 ? XP : 1][2];
    const int kp8 = a.kpad >> 3;                              // 16-byte pieces per row
    const int npieces = M * kp8;
    if constexpr (XM == 1) {
        const _Float16* xb = reinterpret_cast<const _Float16*>(a.x);
        if (x_dma) {
            // one wave instruction moves 1 KB = 512 consecutive K positions of one row into its (row, K half) line of the image
            const int ppr = a.kpad >> 9;                     // 1 KB pieces per row
            for (int p = wid; p < M * ppr; p += GV_WAVES) {
                const int row = p / ppr, k0 = (p - row * ppr) << 9;
                const int hh = k0 >= Kc ? 1 : 0;
                __builtin_amdgcn_global_load_lds(xb + (int64_t)row * a.ldx + k0 + lane * 8,
                                                 (__attribute__((address_space(3))) void*)(sx + (size_t)(row + RH * hh) * xs + (k0 - hh * Kc)), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int q = tid + i * 512;
                if (q < npieces) {
                    const int row = q / kp8, kk = (q - row * kp8) << 3;
                    half8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
                    xh[i] = kk < a.k ? *reinterpret_cast<const half8*>(xb + (int64_t)row * a.ldx + kk) : z;
                }
            }
        }
    }
    if constexpr (XM == 2) {
        // x = part_o [m][heads][2][64] fp32 (unnormalised), x2 = part_ml [m][heads][2] (running max, sum)
        const float* pob = reinterpret_cast<const float*>(a.x);
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            // (unconditional, piece index clamped: guarded, the compiler merges this block with the guarded merge + LDS store below and
            // drains every piece before the next one's loads -- and before the weight loads, which it moves behind them)
            const int q = min(tid + i * 512, npieces - 1);
            const int row = q / kp8, kk = (q - row * kp8) << 3;
            const int hd = kk >> 6, dd = kk & 63;
            const int64_t base = ((int64_t)row * (a.k >> 6) + hd) * 2;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                po[i][sp][0] = *reinterpret_cast<const float4*>(pob + (base + sp) * 64 + dd);
                po[i][sp][1] = *reinterpret_cast<const float4*>(pob + (base + sp) * 64 + dd + 4);
                pml[i][sp] = *reinterpret_cast<const float2*>(a.x2 + (base + sp) * 2);
            }
        }
    }
    half8 fb[NCT * NL][2];                                      // [column tile][line in flight]
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            // (unconditional, line clamped: guarded, these loads sit in blocks of their own and the compiler moves them BELOW the argument
            // pin that follows -- whose scalar-load wait, a cold miss, then delays the whole weight stream; a line beyond K is never used)
            const int line = min(wid + i * GV_WAVES, lines - 1);
            fb[ct * NL + i][0] = LM_WLOAD(wrow + (int64_t)ct * 16 * wld + line * 64);
            fb[ct * NL + i][1] = LM_WLOAD(wrow + (int64_t)ct * 16 * wld + line * 64 + 8);
        }
#if defined(LM_LATE_ARGS) && !defined(LM_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    {   // LM_LATE_ARGS: the struct sits behind the 14 leading dwords (4 pointers + 6 ints = 56 bytes) of the kernarg segment
        const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const GemvArgs full = *(const __attribute__((address_space(4))) GemvArgs*)(kp + 56);
        const GemvArgs lead = a;
        a = full;
        a.x = lead.x; a.w = lead.w; a.x2 = lead.x2; a.gather = lead.gather; a.m = lead.m; a.n = lead.n; a.k = lead.k; a.kpad = lead.kpad; a.ldx = lead.ldx;
    }
#endif
    pin_args(a);            // the remaining arguments: ONE wide scalar load + ONE wait, behind the loads issued above
    // LayerNorm parameters of the lanes' k positions (same positions for every row)
    float4 lg[MAXV], lb[MAXV];
    if constexpr (!XF16) {
        if (vec_ok) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int k = lane * 4 + i * 256;
                if (a.ln_g && i < nv && k < a.k) {
                    lg[i] = *reinterpret_cast<const float4*>(a.ln_g + k);
                    lb[i] = *reinterpret_cast<const float4*>(a.ln_b + k);
                }
            }
        }
    }
    // epilogue operands of the thread's output element
    constexpr int NOUT = DIAG ? 64 : (FORM == 2 ? MT * 128 : MT * NCT * 256);
    const bool owner = tid < NOUT;
    int on, om;                                               // output column / row of this thread
    if constexpr (DIAG) {
        on = n0 + (tid & 7);
        om = (tid >> 3) & 7;
    } else if constexpr (FORM == 2) {
        on = n0 + (tid & 7);
        om = tid >> 3;
    } else {
        const int t = tid >> 8, e = (tid >> 6) & 3;              // accumulator t: row tile (NCT 1) or column tile (NCT 2)
        on = n0 + (NCT == 2 ? t * 16 : 0) + (lane & 15);
        om = (NCT == 2 ? 0 : t * 16) + (lane >> 4) * 4 + e;
    }
    const bool live = owner && on < a.n && om < M;
    float e_bias = 0.0f, e_res = 0.0f;
    if (live && blockIdx.y == 0) {                            // (a split projection's second K slice adds neither)
        if (a.bias) e_bias = a.bias[on];
        if (a.res) e_res = a.res[(int64_t)om * a.ldr + on];
    }
    if (a.zero) {                                             // accumulator of a later split launch
        const int nblk = (a.n + NC - 1) / NC;                 // = gridDim.x, without the scalar load + wait an implicit argument costs
        for (int i = (int)blockIdx.x * 512 + tid; i < a.zero_n; i += nblk * 512) a.zero[i] = 0.0f;
    }
    int kv_pos = a.pos;
    if (a.kv && a.st) kv_pos = a.st->pos0 + a.st->step;
    LM_STAMP(a, 1);

    // ---- (1) stage the input rows as an fp16 image in LDS
    auto put4 = [&](int mr, int k, float4 o) {               // 4 consecutive K positions of row mr
        half4 h4;
        h4[0] = (_Float16)o.x; h4[1] = (_Float16)o.y; h4[2] = (_Float16)o.z; h4[3] = (_Float16)o.w;
        if constexpr (HALF8) {
            const int hh = k >= Kc ? 1 : 0;
            *reinterpret_cast<half4*>(sx + (size_t)(mr + RH * hh) * xs + (k - hh * Kc)) = h4;
        } else {
            *reinterpret_cast<half4*>(sx + (size_t)mr * xs + k) = h4;
        }
    };
    auto put8 = [&](int row, int kk, half8 hv) {              // 8 consecutive K positions of row `row`
        if constexpr (HALF8) {
            const int hh = kk >= Kc ? 1 : 0;
            *reinterpret_cast<half8*>(sx + (size_t)(row + RH * hh) * xs + (kk - hh * Kc)) = hv;
        } else {
            *reinterpret_cast<half8*>(sx + (size_t)row * xs + kk) = hv;
        }
    };
    if constexpr (XM == 1) {
        if (x_dma) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces (and its weight lines) have landed
        } else {
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int q = tid + i * 512;
                if (q < npieces) {
                    const int row = q / kp8;
                    put8(row, (q - row * kp8) << 3, xh[i]);
                }
            }
            for (int q0 = XP * 512; q0 < npieces; q0 += XP * 512) {   // rows beyond the prefetched pieces, a batch of loads at a time
#pragma unroll
                for (int i = 0; i < XP; ++i) {
                    const int q = q0 + tid + i * 512;
                    if (q < npieces) {
                        const int row = q / kp8, kk = (q - row * kp8) << 3;
                        half8 z;
#pragma unroll
                        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
                        xh[i] = kk < a.k ? *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(a.x) + (int64_t)row * a.ldx + kk) : z;
                    }
                }
#pragma unroll
                for (int i = 0; i < XP; ++i) {
                    const int q = q0 + tid + i * 512;
                    if (q < npieces) {
                        const int row = q / kp8;
                        put8(row, (q - row * kp8) << 3, xh[i]);
                    }
                }
            }
        }
    } else if constexpr (XM == 2) {
        // merge the two key-range partials of lm_attn: out = (o0 w0 + o1 w1) / (l0 w0 + l1 w1), w_s = exp(m_s - max m)
        auto merge = [&](const float4 (&o)[2][2], const float2 (&ml)[2]) {
            const float mx = fmaxf(ml[0].x, ml[1].x);
            const float w0 = ml[0].x == -INFINITY ? 0.0f : __expf(ml[0].x - mx), w1 = ml[1].x == -INFINITY ? 0.0f : __expf(ml[1].x - mx);
            // a w0 + b w1 with the contraction PINNED (fma(a, w0, b w1)): left to the compiler, which of the two products is fused depends on
            // the surrounding code, i.e. on the kernel variant -- and a row's bits must not depend on the variant (8- vs 16- vs 32-row forms)
            auto mix = [&](float p0, float p1) { return __builtin_fmaf(p0, w0, p1 * w1); };
            const float l = mix(ml[0].y, ml[1].y);
            const float inv = l > 0.0f ? 1.0f / l : 0.0f;
            half8 hv;
            hv[0] = (_Float16)(mix(o[0][0].x, o[1][0].x) * inv); hv[1] = (_Float16)(mix(o[0][0].y, o[1][0].y) * inv);
            hv[2] = (_Float16)(mix(o[0][0].z, o[1][0].z) * inv); hv[3] = (_Float16)(mix(o[0][0].w, o[1][0].w) * inv);
            hv[4] = (_Float16)(mix(o[0][1].x, o[1][1].x) * inv); hv[5] = (_Float16)(mix(o[0][1].y, o[1][1].y) * inv);
            hv[6] = (_Float16)(mix(o[0][1].z, o[1][1].z) * inv); hv[7] = (_Float16)(mix(o[0][1].w, o[1][1].w) * inv);
            return hv;
        };
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int q = tid + i * 512;
            if (q < npieces) {
                const int row = q / kp8;
                put8(row, (q - row * kp8) << 3, merge(po[i], pml[i]));
            }
        }
        for (int q0 = XP * 512; q0 < npieces; q0 += XP * 512) {   // more rows: a batch of loads at a time
            const float* pob = reinterpret_cast<const float*>(a.x);
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int q = min(q0 + tid + i * 512, npieces - 1);      // unconditional, clamped (see the first batch)
                const int row = q / kp8, kk = (q - row * kp8) << 3;
                const int hd = kk >> 6, dd = kk & 63;
                const int64_t base = ((int64_t)row * (a.k >> 6) + hd) * 2;
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    po[i][sp][0] = *reinterpret_cast<const float4*>(pob + (base + sp) * 64 + dd);
                    po[i][sp][1] = *reinterpret_cast<const float4*>(pob + (base + sp) * 64 + dd + 4);
                    pml[i][sp] = *reinterpret_cast<const float2*>(a.x2 + (base + sp) * 2);
                }
            }
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int q = q0 + tid + i * 512;
                if (q < npieces) {
                    const int row = q / kp8;
                    put8(row, (q - row * kp8) << 3, merge(po[i], pml[i]));
                }
            }
        }
    } else {
        // register path: wave w owns rows w, w + 8, ... (the first one was prefetched above)
        auto stage_row = [&](float4 (&v)[MAXV], int mr) {
            const float inv_k = 1.0f / (float)a.k;
#pragma unroll
            for (int i = 0; i < MAXV; ++i)                    // positions beyond K were loaded from a clamped address: zero them
                if (!(i < nv && lane * 4 + i * 256 < a.k)) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.pre_g) {
                // embedding stage: two-pass LayerNorm (as layernorm_rows) -> ReLU -> * pre_scale
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < MAXV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
                const float mean = wsum64(s) * inv_k;
                float qq = 0.0f;
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int k = lane * 4 + i * 256;
                    if (i < nv && k < a.k) {
                        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                        qq += __builtin_fmaf(dx, dx, dy * dy) + __builtin_fmaf(dz, dz, dw * dw);
                    }
                }
                const float rstd = rsqrtf(wsum64(qq) * inv_k + a.ln_eps);
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int k = lane * 4 + i * 256;
                    if (i < nv && k < a.k) {
                        // (parameters loaded here, not with the up-front loads: one kernel per step takes this branch and
                        // eight more float4 of live registers would cost every other variant its second workgroup per CU)
                        const float4 pg = *reinterpret_cast<const float4*>(a.pre_g + k);
                        const float4 pb = *reinterpret_cast<const float4*>(a.pre_b + k);
                        float4 o;
                        o.x = a.pre_scale * fmaxf((v[i].x - mean) * rstd * pg.x + pb.x, 0.0f);
                        o.y = a.pre_scale * fmaxf((v[i].y - mean) * rstd * pg.y + pb.y, 0.0f);
                        o.z = a.pre_scale * fmaxf((v[i].z - mean) * rstd * pg.z + pb.z, 0.0f);
                        o.w = a.pre_scale * fmaxf((v[i].w - mean) * rstd * pg.w + pb.w, 0.0f);
                        v[i] = o;
                        if (a.pre_out && blockIdx.x == 0) *reinterpret_cast<float4*>(a.pre_out + (int64_t)mr * a.ldx + k) = o;
                    }
                }
            }
            float mean = 0.0f, rstd = 1.0f;
            if (a.ln_g || a.ln_plain) {                       // single-pass statistics (as gemm_skinny16)
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    s1 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
                    // (a a + b b with the contraction pinned: which product the compiler fuses may differ between kernel variants, and a
                    // row's bits must not depend on the variant)
                    s2 += __builtin_fmaf(v[i].x, v[i].x, v[i].y * v[i].y) + __builtin_fmaf(v[i].z, v[i].z, v[i].w * v[i].w);
                }
                s1 = wsum64(s1);
                s2 = wsum64(s2);
                mean = s1 / (float)a.k;
                const float var = fmaxf(s2 / (float)a.k - mean * mean, 0.0f);
                rstd = rsqrtf(var + a.ln_eps);
            }
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int k = lane * 4 + i * 256;
                if (i < nv && k < a.kpad) {
                    float4 o = v[i];
                    if (a.ln_g && k < a.k) {
                        o.x = (o.x - mean) * rstd * lg[i].x + lb[i].x;
                        o.y = (o.y - mean) * rstd * lg[i].y + lb[i].y;
                        o.z = (o.z - mean) * rstd * lg[i].z + lb[i].z;
                        o.w = (o.w - mean) * rstd * lg[i].w + lb[i].w;
                    } else if (a.ln_plain && k < a.k) {      // scale and shift are folded into the weights / bias (host, at load)
                        o.x = (o.x - mean) * rstd; o.y = (o.y - mean) * rstd;
                        o.z = (o.z - mean) * rstd; o.w = (o.w - mean) * rstd;
                    }
                    put4(mr, k, o);
                }
            }
        };
        if (vec_ok) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
                if (wid + rr * GV_WAVES < M) stage_row(v0[rr], wid + rr * GV_WAVES);
        } else {
            // general path (any K / alignment; no embedding pre-transform): scalar, two-pass statistics
            for (int mr = wid; mr < M; mr += GV_WAVES) {
                const int64_t src = a.gather ? (int64_t)a.gather[mr] : (int64_t)mr;
                const float* xr = reinterpret_cast<const float*>(a.x) + src * a.ldx;
                float mean = 0.0f, rstd = 1.0f;
                if (a.ln_g || a.ln_plain) {
                    float s = 0.0f;
                    for (int k = lane; k < a.k; k += 64) s += xr[k];
                    mean = wsum64(s) / (float)a.k;
                    float vv = 0.0f;
                    for (int k = lane; k < a.k; k += 64) {
                        const float d = xr[k] - mean;
                        vv += d * d;
                    }
                    rstd = rsqrtf(wsum64(vv) / (float)a.k + a.ln_eps);
                }
                for (int k = lane; k < a.kpad; k += 64) {
                    float v = 0.0f;
                    if (k < a.k) {
                        v = xr[k];
                        if (a.ln_g) v = (v - mean) * rstd * a.ln_g[k] + a.ln_b[k];
                        else if (a.ln_plain) v = (v - mean) * rstd;
                    }
                    if constexpr (HALF8) {
                        const int hh = k >= Kc ? 1 : 0;
                        sx[(size_t)(mr + RH * hh) * xs + (k - hh * Kc)] = (_Float16)v;
                    } else {
                        sx[(size_t)mr * xs + k] = (_Float16)v;
                    }
                }
            }
        }
    }
    LM_STAMP(a, 2);
    __syncthreads();
    LM_STAMP(a, 3);

    // ---- (2) MFMAs: A fragments from LDS, B fragments are the prefetched weight registers
    float4v acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.0f;
    const _Float16* arow[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        int mr;
        if constexpr (FORM == 2) {
            mr = (t >> 1) * 16 + c + RH * (t & 1);            // accumulator (row tile t / 2, K half t % 2)
        } else {
            mr = (NCT == 2 ? 0 : t * 16) + c;                 // NCT 2: both accumulators take the one row tile
            if constexpr (!HALF8) {
                if (mr >= M) mr = M - 1;
            }
        }
        arow[t] = sx + (size_t)mr * xs + g * 16;
    }
    for (int pass = 0;; ++pass) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int line = wid + (pass * NL + i) * GV_WAVES;
            if (line < lines) {
#pragma unroll
                for (int t = 0; t < NACC; ++t) {
                    const half8 fa0 = *reinterpret_cast<const half8*>(arow[t] + line * 64);
                    const half8 fa1 = *reinterpret_cast<const half8*>(arow[t] + line * 64 + 8);
                    const int fi = (NCT == 2 ? t * NL : 0) + i;      // weights: shared by the row tiles, one set per column tile
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, fb[fi][0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa1, fb[fi][1], acc[t], 0, 0, 0);
                }
            }
        }
        if (wid + (pass + 1) * NL * GV_WAVES >= lines) break;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int line = wid + ((pass + 1) * NL + i) * GV_WAVES;
                if (line < lines) {
                    fb[ct * NL + i][0] = LM_WLOAD(wrow + (int64_t)ct * 16 * wld + line * 64);
                    fb[ct * NL + i][1] = LM_WLOAD(wrow + (int64_t)ct * 16 * wld + line * 64 + 8);
                }
            }
    }
    LM_STAMP(a, 4);
    // ---- (3) cross-wave reduction + epilogue
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[((wid * NACC + t) * 4 + e) * 64 + lane] = acc[t][e];
    __syncthreads();
    if (live) {
        float v = 0.0f;
        if constexpr (DIAG) {
            const int r = om, cc = tid & 7;
            const int e = r & 3, la = 16 * (r >> 2) + cc, lb2 = la + 40;      // D[r][c] and D[r + 8][c + 8]
#pragma unroll
            for (int w = 0; w < GV_WAVES; ++w) v += red[(w * 4 + e) * 64 + la] + red[(w * 4 + e) * 64 + lb2];
        } else if constexpr (FORM == 2) {
            const int t = om >> 4, r = om & 15, cc = tid & 7;
            const int e = r & 3, la = 16 * (r >> 2) + cc;                     // D0[r][c] (K half 0) and D1[r][c + 8] (K half 1)
#pragma unroll
            for (int w = 0; w < GV_WAVES; ++w)
                v += red[((w * NACC + 2 * t) * 4 + e) * 64 + la] + red[((w * NACC + 2 * t + 1) * 4 + e) * 64 + la + 8];
        } else {
            const int t = tid >> 8, e = (tid >> 6) & 3;
#pragma unroll
            for (int w = 0; w < GV_WAVES; ++w) v += red[((w * NACC + t) * 4 + e) * 64 + lane];
        }
        v += e_bias;
        if (a.relu) v = fmaxf(v, 0.0f);
        v += e_res;
        if (a.kv && on >= a.n_split) {
            const int dd = (a.n - a.n_split) >> 1, c = on - a.n_split, isv = c >= dd ? 1 : 0, cc = c - isv * dd;
            a.kv[(int64_t)kv_pos * a.kv_t + (int64_t)om * a.kv_b + (int64_t)(cc >> 6) * a.kv_h + (int64_t)isv * a.kv_v + (cc & 63)] = (_Float16)v;
        } else if (XM == 1 && p_ks > 1) {
            unsafeAtomicAdd(a.out + (int64_t)om * a.ldo + on, v);     // the two K slices meet on a zero: a + b == b + a
        } else {
            if (a.out) a.out[(int64_t)om * a.ldo + on] = v;
            if (a.out16) a.out16[(int64_t)om * a.ldo16 + on] = (_Float16)v;
        }
    }
    if (a.advance && a.st && blockIdx.x == 0 && tid == 0) a.st->step = a.st->step + 1;   // no other block of this kernel reads it
    LM_STAMP(a, 5);
}

// ------------------------------------------------------------------------------------------------------------------
// Decode attention, one (row, head, key half) per block.  8 lanes share a key (lane sub = tid & 7 owns dims [8 sub, 8 sub + 8):
// one 16-byte load each of K, V and the position row), 64 keys per pass of the 512 threads, AT_U passes per chunk with
// ALL of a chunk's K, position and V loads issued before the first score is formed: one memory round trip per 256 keys
// (a benchmark step has <= 434 keys = <= 256 per key half: one round trip for the whole kernel; K, then softmax, then V took
// three).  Chunks are combined by the usual running maximum / running sum.  AT_U = 4 (was 8): 48 instead of 96 staging
// registers = two workgroups per CU, which is what 16- and 32-row batches (512 / 1024 workgroups) need; the chunk size is part
// of a row's arithmetic (when the running maximum moves), so it is ONE constant for every batch size.
static constexpr int AT_U = 4;

__device__ __forceinline__ float dot8(const float (&q)[8], half8 k) {
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += q[e] * (float)k[e];
    return s;
}

__global__ __launch_bounds__(512, 4) void lm_attn(const float* p_q, const _Float16* p_kv, const _Float16* p_pos0, const int* p_kstart, int p_kv_t, int p_kv_b,
                                                  int p_kv_h, int p_kv_v, int p_ld, int p_pos, AttnArgs a_in) {
    __shared__ float s_m[8];
    __shared__ float s_l[8];
    __shared__ __attribute__((aligned(16))) float s_o[8][64];
    LM_RAISE_PRIO();
    // Leading parameters: preloaded into SGPRs by the command processor (see lm_gemv; 14 dwords is the most the hardware preloads).  They
    // are everything the QUERY and KEY loads need -- p_ld = ldq | ldp << 16, p_pos = the query's position (-1: device-side step state)
    // -- so those loads go out before the by-value struct (a cold scalar-cache miss, ~1 us: bias vectors, scale, outputs) has arrived.
    const _Float16* pos_row0 = p_pos0;                       // the position table's row of relative position 0 (postab + center * ldp)
    const int ldq = p_ld & 0xffff, ldp = (int)((unsigned)p_ld >> 16);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int sub = tid & 7, kg = tid >> 3;                   // key group 0..63
    const int head = blockIdx.x, bb = blockIdx.y;
    // p_pos >= 0: bits 0-27 the query's absolute position (= index of the newest key), bits 28-29 the key split - 1 (gridDim.z is an
    // implicit kernel argument: reading it is a scalar load whose wait covers the whole struct); < 0: device-side step state, -key split
    int qpos = p_pos & 0x0fffffff;
    int nsplit = ((p_pos >> 28) & 3) + 1;
    if (p_pos < 0) {
        qpos = a_in.st->pos0 + a_in.st->step;
        nsplit = -p_pos;
    }
    const int len = qpos + 1;
    // Load schedule (round 4).  The kernel has THREE kinds of operands -- the row's first valid key, the query / bias vectors, the keys --
    // and only the key addresses depend on another load (the first valid key).  Issued in that order, with every key load of a chunk
    // UNCONDITIONAL (the index is clamped to the newest key; what lies beyond the range is masked when the scores are formed): a load
    // inside an `if (j < kend)` block made the compiler open each block with s_waitcnt vmcnt(0), so the four 64-key groups of a chunk --
    // meant to be one memory round trip -- were four dependent ones, behind a fifth for the query.
    const float* qp = p_q + (int64_t)bb * ldq + head * 64 + sub * 8;
    const float4 x0 = *reinterpret_cast<const float4*>(qp), x1 = *reinterpret_cast<const float4*>(qp + 4);
    // (the first valid key in a uniform branch: batches without left padding pass no table and must not wait for a load here -- the key
    // addresses depend on this value)
    int ks_raw = 0;
    if (p_kstart) ks_raw = p_kstart[bb];
    const int ks_first = min(ks_raw, len - 1);
    // key split (gridDim.z = 2): two workgroups per (row, head) take the two halves of the valid keys (a multiple of 64
    // keys each) and write unnormalised partials; the consumer (lm_gemv XM == 2) merges them while staging its input.
    int ks0 = ks_first, kend = len;
    if (nsplit > 1) {
        const int half = (((len - ks_first + 1) >> 1) + 63) & ~63;
        ks0 = ks_first + (int)blockIdx.z * half;
        kend = min(len, ks0 + half);
    }
    const int64_t trow = p_kv_t;                              // one time step of the cache (KvLayout)
    const _Float16* kb = p_kv + (int64_t)bb * p_kv_b + (int64_t)head * p_kv_h + sub * 8;
    const _Float16* vb = kb + p_kv_v;
    const _Float16* pb = pos_row0 + head * 64 + sub * 8;
    half8 kk[AT_U], pp[AT_U], vv[AT_U];
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int u = 0; u < AT_U; ++u) {
            const int jc = min(j0 + u * 64 + kg, qpos);      // clamped: always a row of the cache
            kk[u] = *reinterpret_cast<const half8*>(kb + (int64_t)jc * trow);
            pp[u] = *reinterpret_cast<const half8*>(pb + (int64_t)(qpos - jc) * ldp);
            vv[u] = *reinterpret_cast<const half8*>(vb + (int64_t)jc * trow);
        }
    };
    load_chunk(ks0);                                          // (a workgroup with an empty key range loads the newest key and uses nothing)
    // ---- now the struct: bias vectors, scale, outputs
    AttnArgs a = a_in;
    a.q = p_q; a.kv = p_kv; a.kstart = p_kstart; a.kv_t = p_kv_t; a.kv_b = p_kv_b; a.kv_h = p_kv_h; a.kv_v = p_kv_v; a.ldq = ldq; a.ldp = ldp;
    pin_args(a);
    LM_STAMP(a, 0);
    const float4 u0 = *reinterpret_cast<const float4*>(a.bias_u + head * 64 + sub * 8), u1 = *reinterpret_cast<const float4*>(a.bias_u + head * 64 + sub * 8 + 4);
    const float4 w0 = *reinterpret_cast<const float4*>(a.bias_v + head * 64 + sub * 8), w1 = *reinterpret_cast<const float4*>(a.bias_v + head * 64 + sub * 8 + 4);
    float qu[8], qv[8];
    {
        const float xs_[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float us_[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        const float ws_[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qu[e] = (xs_[e] + us_[e]) * a.scale;
            qv[e] = (xs_[e] + ws_[e]) * a.scale;
        }
    }
    // every wave keeps its OWN running maximum / sum / output over the keys it sees (no workgroup barrier inside the key
    // loop); the eight waves are merged once at the end.
    float m_run = -INFINITY, l_run = 0.0f;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.0f;
    for (int j0 = ks0; j0 < kend; j0 += 64 * AT_U) {
        if (j0 != ks0) load_chunk(j0);
        LM_STAMP(a, 1);
        float s[AT_U];
        float m_new = m_run;
#pragma unroll
        for (int u = 0; u < AT_U; ++u) {
            const int j = j0 + u * 64 + kg;
            // the 8 lanes of a key group hold partial dots of the same key (an invalid key is invalid on all 8)
            float t = j < kend ? dot8(qu, kk[u]) + dot8(qv, pp[u]) : 0.0f;
            t = xadd<1>(t);
            t = xadd<2>(t);
            t = xadd<4>(t);
            s[u] = j < kend ? t : -INFINITY;
            m_new = fmaxf(m_new, s[u]);
        }
        m_new = xmax<32>(xmax<16>(xmax<8>(m_new)));             // the wave's maximum
        if (m_new > -INFINITY) {                               // wave-uniform; false only while the wave has seen no valid key
            const float sc_old = m_run == -INFINITY ? 0.0f : __expf(m_run - m_new);
            l_run *= sc_old;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= sc_old;
#pragma unroll
            for (int u = 0; u < AT_U; ++u) {
                const int j = j0 + u * 64 + kg;
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
    LM_STAMP(a, 2);
    // sum over the key groups of the wave (lanes with equal `sub`); l_run is per key group and equal on its 8 lanes
    l_run = xadd<32>(xadd<16>(xadd<8>(l_run)));               // steps 8, 16, 32 as before (xlane.h: VALU exchanges, same bits)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = xadd<32>(xadd<16>(xadd<8>(o[e])));
    if (lane < 8) {
        *reinterpret_cast<float4*>(&s_o[wid][lane * 8]) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(&s_o[wid][lane * 8 + 4]) = make_float4(o[4], o[5], o[6], o[7]);
        if (lane == 0) {
            s_l[wid] = l_run;
            s_m[wid] = m_run;
        }
    }
    __syncthreads();
    if (tid < 64) {
        float mx = s_m[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) mx = fmaxf(mx, s_m[w]);
        float tot = 0.0f, l = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {                          // fixed order: the same bits on every run
            const float sc = s_m[w] == -INFINITY ? 0.0f : __expf(s_m[w] - mx);
            tot += s_o[w][tid] * sc;
            l += s_l[w] * sc;
        }
        const float m_run = mx;
        if (nsplit > 1) {
            const int64_t slot = ((int64_t)bb * a.h + head) * 2 + blockIdx.z;
            a.part_o[slot * 64 + tid] = tot;
            if (tid == 0) *reinterpret_cast<float2*>(a.part_ml + slot * 2) = make_float2(m_run, l);
        } else {
            a.out[(int64_t)bb * a.ldo + head * 64 + tid] = (_Float16)(l > 0.0f ? tot / l : 0.0f);
        }
    }
    LM_STAMP(a, 3);
}

// ------------------------------------------------------------------------------------------------------------------
static size_t gemv_lds_bytes(int lrows, int kc, int nacc) {
    return (((size_t)lrows * (kc + 8) * 2 + 15) & ~(size_t)15) + (size_t)GV_WAVES * nacc * 4 * 64 * sizeof(float);
}

// 8-column forms: twice the workgroups.  Measured (scripts/micro/decode_chain.hip, per launch incl. the 1.45 us boundary): it
// pays where the 16-column grid leaves CUs idle (n = 1024: 64 -> 128 workgroups; the deep FFN-out projection 7.5 -> 6.0-6.4 us
// with no split-K hand-off) and costs where the 16-column grid already covers the chip (n = 3072 / 4096: 5.7 -> 8.9 us, every
// workgroup stages the whole input and a CU then ingests 96 KB instead of 64).  The choice depends on the projection's shape
// only, never on the row count: a row's arithmetic is then the same in 8-, 16- and 32-row batches.
static int half8_max_blocks() {
    static const int v = exp_env_int("ASTTS_LM_HALF8_MAX_BLOCKS", 256);
    return v;
}

// bits 0-1: form (0 = 16 columns, 1 = diagonal 8 x 8, 2 = halved 8 columns), bits 2..: row tiles of 16 (forms 0 and 2)
int lm_gemv_variant(const GemvArgs& a) {
    const bool halved = (a.kpad % 128) == 0 && (a.n + 7) / 8 <= half8_max_blocks();
    const int mt = a.m <= 16 ? 1 : 2;
    const int form = !halved ? 0 : (a.m <= 8 ? 1 : 2);
    return form | (mt << 2);
}

// ASTTS_LM_WIDE=1: the wide (32-column) form for 16-column projections with K <= 1024 at <= 16 rows (same sums, half the workgroups)
static bool lm_wide() {
    static const bool v = exp_env_int("ASTTS_LM_WIDE", 0) != 0;
    return v;
}

#define GV_LEAD(a) (a).x, (a).w, (a).x2, (a).gather, (a).m, (a).n, (a).k, (a).kpad, (a).ldx, ((a).ksplit == 2 ? 2 : 1)     // the preloaded leading kernel arguments (14 dwords)
#define AT_LEAD(a) (a).q, (a).kv, (a).postab + (int64_t)(a).center * (a).ldp, (a).kstart, (a).kv_t, (a).kv_b, (a).kv_h, (a).kv_v, (int)((unsigned)(a).ldq | (unsigned)(a).ldp << 16), ((a).st ? -(a).ksplit : (int)((unsigned)(a).pos | (unsigned)((a).ksplit - 1) << 28))

template <int MT, int FORM, int XM>
static void gemv_launch_nl(const GemvArgs& a, dim3 grid, size_t lds, int lines_per_wave, hipStream_t st, bool wide = false) {
    static std::once_flag attr;   // first use (never inside a capture: lm_step_set_attrs() visits every variant up front)
    std::call_once(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_gemv<MT, FORM, XM, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_gemv<MT, FORM, XM, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if constexpr (FORM == 0 && MT == 1 && XM != 2)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_gemv<1, 0, XM, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (grid.x == 0) return;
    if constexpr (FORM == 0 && MT == 1 && XM != 2) {
        if (wide) {              // 32 columns per workgroup (K <= 1024: two lines per wave)
            hipEvent_t e0, e1;
            if (prof_events(ASTTS_PROF_GEMM_SKINNY, (double)a.n * a.kpad * 2.0, &e0, &e1))
                hipExtLaunchKernelGGL((lm_gemv<1, 0, XM, 2, 2>), grid, dim3(512), (uint32_t)lds, st, e0, e1, 0, GV_LEAD(a), a);
            else
                hipLaunchKernelGGL((lm_gemv<1, 0, XM, 2, 2>), grid, dim3(512), lds, st, GV_LEAD(a), a);
            return;
        }
    }
    // bench-only launch profiler: algorithmic bytes of a decode GEMV = its weight image, streamed once
    hipEvent_t e0, e1;
    if (prof_events(ASTTS_PROF_GEMM_SKINNY, (double)a.n * a.kpad * 2.0, &e0, &e1)) {
        if (lines_per_wave <= 2) hipExtLaunchKernelGGL((lm_gemv<MT, FORM, XM, 2>), grid, dim3(512), (uint32_t)lds, st, e0, e1, 0, GV_LEAD(a), a);
        else hipExtLaunchKernelGGL((lm_gemv<MT, FORM, XM, 8>), grid, dim3(512), (uint32_t)lds, st, e0, e1, 0, GV_LEAD(a), a);
        return;
    }
    if (lines_per_wave <= 2) hipLaunchKernelGGL((lm_gemv<MT, FORM, XM, 2>), grid, dim3(512), lds, st, GV_LEAD(a), a);
    else hipLaunchKernelGGL((lm_gemv<MT, FORM, XM, 8>), grid, dim3(512), lds, st, GV_LEAD(a), a);
}

template <int MT, int FORM>
static void gemv_launch_xm(const GemvArgs& a, int xm, dim3 grid, size_t lds, int lpw, hipStream_t st, bool wide = false) {
    if (xm == 0) gemv_launch_nl<MT, FORM, 0>(a, grid, lds, lpw, st, wide);
    else if (xm == 1) gemv_launch_nl<MT, FORM, 1>(a, grid, lds, lpw, st, wide);
    else gemv_launch_nl<MT, FORM, 2>(a, grid, lds, lpw, st);
}

void lm_step_set_attrs() {
    static std::once_flag done;
    std::call_once(done, [] {
        GemvArgs z{};
        for (int xm = 0; xm < 3; ++xm) {   // grid 0: sets the attributes, launches nothing
            gemv_launch_xm<1, 1>(z, xm, dim3(0), 0, 2, nullptr);
            gemv_launch_xm<1, 2>(z, xm, dim3(0), 0, 2, nullptr);
            gemv_launch_xm<2, 2>(z, xm, dim3(0), 0, 2, nullptr);
            gemv_launch_xm<1, 0>(z, xm, dim3(0), 0, 2, nullptr);
            gemv_launch_xm<2, 0>(z, xm, dim3(0), 0, 2, nullptr);
        }
    });
}

int lm_gemv_launch(const GemvArgs& a0, hipStream_t st) {
    ASTTS_REQUIRE(a0.x && a0.w && (a0.out || a0.out16 || a0.kv), ASTTS_ERR_INVALID, "lm_gemv: null pointer");
    ASTTS_REQUIRE(a0.m >= 1 && a0.m <= 32 && a0.n >= 1 && a0.k >= 1 && a0.kpad >= a0.k && (a0.kpad & 63) == 0, ASTTS_ERR_INVALID,
                  "lm_gemv: bad shape m=%d n=%d k=%d kpad=%d", a0.m, a0.n, a0.k, a0.kpad);
    const int xm = a0.x_mode;
    ASTTS_REQUIRE(xm >= 0 && xm <= 2, ASTTS_ERR_INVALID, "lm_gemv: x_mode=%d", xm);
    ASTTS_REQUIRE(xm == 0 || (!a0.gather && !a0.ln_g && !a0.pre_g && (a0.k & 7) == 0 && ((uintptr_t)a0.x & 15) == 0), ASTTS_ERR_INVALID,
                  "lm_gemv: fp16 / partial input takes no gather / LayerNorm and needs 16-byte aligned rows");
    ASTTS_REQUIRE(xm != 1 || (a0.ldx & 7) == 0, ASTTS_ERR_INVALID, "lm_gemv: fp16 input rows must be 16-byte aligned");
    ASTTS_REQUIRE(xm != 2 || (a0.x2 && (a0.k & 63) == 0 && a0.k == a0.kpad), ASTTS_ERR_INVALID, "lm_gemv: attention partials need x2 and k = heads * 64");
    ASTTS_REQUIRE(!a0.pre_g || (a0.k <= 1024 && (a0.k & 3) == 0 && (a0.ldx & 3) == 0 && ((uintptr_t)a0.x & 15) == 0), ASTTS_ERR_INVALID,
                  "lm_gemv: the embedding pre-transform needs k <= 1024 and aligned rows");
    const int ksp = a0.ksplit == 2 ? 2 : 1;
    ASTTS_REQUIRE(a0.ksplit == 0 || a0.ksplit == 1 || (a0.ksplit == 2 && xm == 1 && a0.k == a0.kpad && (a0.kpad & 255) == 0 && a0.out && !a0.out16 && !a0.kv &&
                                                          !a0.relu && !a0.advance), ASTTS_ERR_INVALID,
                  "lm_gemv: ksplit=%d needs fp16 input, k == kpad (a multiple of 256), a fp32 output holding zeros, no relu / out16 / kv", a0.ksplit);
    ASTTS_REQUIRE(!a0.zero || a0.zero_n >= 0, ASTTS_ERR_INVALID, "lm_gemv: zero_n=%d", a0.zero_n);
    lm_step_set_attrs();
    const int kslice = a0.kpad / ksp;                         // K extent of a workgroup
    auto shape_of = [&](int rows, int* form, int* mt, int* kc, size_t* lds) {
        GemvArgs t = a0;
        t.m = rows;
        t.k = t.kpad = kslice;
        const int var = lm_gemv_variant(t);
        *form = var & 3;
        *mt = var >> 2;
        *kc = *form ? kslice / 2 : kslice;
        const int lrows = *form == 1 ? 16 : (*form == 2 ? 32 * *mt : rows);
        *lds = gemv_lds_bytes(lrows, *kc, *form == 2 ? 2 * *mt : *mt);
    };
    // rows are taken in chunks whose fp16 image fits the LDS (only m > 16 with K = 4096 needs two launches)
    int rows = a0.m, form, mt, kc;
    size_t lds;
    for (;;) {
        shape_of(rows, &form, &mt, &kc, &lds);
        if (lds <= 160 * 1024) break;
        ASTTS_REQUIRE(rows > 1, ASTTS_ERR_INVALID, "lm_gemv: kpad=%d does not fit the LDS image", a0.kpad);
        rows = rows > 16 ? 16 : rows / 2;
    }
    ASTTS_REQUIRE(rows == a0.m || (!a0.kv && !a0.pre_g), ASTTS_ERR_INVALID, "lm_gemv: row chunks cannot write the kv cache");
    for (int r0 = 0; r0 < a0.m; r0 += rows) {
        GemvArgs a = a0;
        a.m = a0.m - r0 < rows ? a0.m - r0 : rows;
        if (r0) {
            if (a.gather) a.gather += r0;
            else if (xm == 0) a.x = reinterpret_cast<const float*>(a0.x) + (int64_t)r0 * a0.ldx;
            else if (xm == 1) a.x = reinterpret_cast<const _Float16*>(a0.x) + (int64_t)r0 * a0.ldx;
            else {
                a.x = reinterpret_cast<const float*>(a0.x) + (int64_t)r0 * (a0.k >> 6) * 2 * 64;
                a.x2 = a0.x2 + (int64_t)r0 * (a0.k >> 6) * 2 * 2;
            }
            if (a.res) a.res += (int64_t)r0 * a0.ldr;
            if (a.out) a.out += (int64_t)r0 * a0.ldo;
            if (a.out16) a.out16 += (int64_t)r0 * a0.ldo16;
        }
        a.advance = a0.advance && r0 + rows >= a0.m;
        // the form of the FIRST chunk serves every chunk (a short last chunk must not change the arithmetic of its rows)
        const int lpw = ((kc >> 6) + GV_WAVES - 1) / GV_WAVES;
        const bool wide = form == 0 && mt == 1 && xm != 2 && lpw <= 2 && lm_wide();
        const dim3 grid(wide ? (a.n + 31) / 32 : (a.n + (form ? 7 : 15)) / (form ? 8 : 16), ksp);
        if (wide) lds = gemv_lds_bytes(a.m, kc, 2);      // two accumulators per wave in the cross-wave reduction
        if (form == 1) gemv_launch_xm<1, 1>(a, xm, grid, lds, lpw, st);
        else if (form == 2 && mt == 1) gemv_launch_xm<1, 2>(a, xm, grid, lds, lpw, st);
        else if (form == 2) gemv_launch_xm<2, 2>(a, xm, grid, lds, lpw, st);
        else if (mt == 1) gemv_launch_xm<1, 0>(a, xm, grid, lds, lpw, st, wide);
        else gemv_launch_xm<2, 0>(a, xm, grid, lds, lpw, st);
    }
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int lm_attn_launch(const AttnArgs& a, hipStream_t st) {
    ASTTS_REQUIRE(a.q && a.kv && a.postab && a.bias_u && a.bias_v, ASTTS_ERR_INVALID, "lm_attn: null pointer");
    ASTTS_REQUIRE(a.ldq > 0 && a.ldq < 65536 && a.ldp > 0 && a.ldp < 65536 && (a.st || (a.pos >= 0 && a.pos < (1 << 28))), ASTTS_ERR_INVALID,
                  "lm_attn: ldq=%d ldp=%d (< 65536: they share one preloaded kernel argument) pos=%d", a.ldq, a.ldp, a.pos);
    ASTTS_REQUIRE(a.b >= 1 && a.h >= 1 && a.d == a.h * 64 && (a.ldq & 3) == 0 && (a.ldp & 7) == 0, ASTTS_ERR_INVALID,
                  "lm_attn: bad shape b=%d h=%d d=%d", a.b, a.h, a.d);
    ASTTS_REQUIRE(a.kv_t > 0 && ((a.kv_t | a.kv_b | a.kv_h | a.kv_v) & 7) == 0, ASTTS_ERR_INVALID,
                  "lm_attn: cache strides (%d, %d, %d, %d) must be multiples of 8 halfs (KvLayout)", a.kv_t, a.kv_b, a.kv_h, a.kv_v);
    ASTTS_REQUIRE(a.ksplit == 1 ? a.out != nullptr : (a.ksplit == 2 && a.part_o && a.part_ml), ASTTS_ERR_INVALID,
                  "lm_attn: ksplit=%d needs %s", a.ksplit, a.ksplit == 1 ? "out" : "the partial buffers (ksplit 1 or 2)");
    // algorithmic bytes: the fp16 K and V rows of every (row, head) + the position rows of every head, each read once
    const double keys = (double)((a.st ? 0 : a.pos) + 1);
    hipEvent_t e0, e1;
    if (prof_events(ASTTS_PROF_ATTN_DECODE, keys * a.d * 2.0 * (2.0 * a.b + 1.0), &e0, &e1))
        hipExtLaunchKernelGGL(lm_attn, dim3(a.h, a.b, a.ksplit), dim3(512), 0, st, e0, e1, 0, AT_LEAD(a), a);
    else
        hipLaunchKernelGGL(lm_attn, dim3(a.h, a.b, a.ksplit), dim3(512), 0, st, AT_LEAD(a), a);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // namespace astts
