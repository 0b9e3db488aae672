// lm_step.h -- the decode-step kernels of the acoustic transformer (launchers; kernels in lm_step.hip).
//
// One autoregressive step of the reference's hot loop #2 (TransformerLM.inference behind
// tts_with_rag.py:195) is a chain of ~73 small DEPENDENT launches, each bound by latency, not bandwidth
// (25 MB of weights per layer = 4 us of HBM time against 5 kernel boundaries).  These kernels are built for that regime:
//   * every global load a block needs is issued in its first instructions (weights, input rows, LayerNorm
//     parameters, epilogue operands), so a block pays ONE memory round trip before its MFMAs;
//   * nothing step-dependent is a kernel ARGUMENT: the step index lives in device memory (LmStep) and is advanced by
//     the last kernel of the step, so ONE captured hipGraph replays for every step of a decode call;
//   * 8-row batches (the benchmark batch) put two K halves on the diagonal blocks of one 16x16 MFMA, so a block owns
//     8 output columns: twice the workgroups per projection and no split-K hand-off on the deep FFN-out projection.
#pragma once
#include "common.h"

namespace astts {

struct LmStep {      // device-resident step state of one decode call
    int step;        // index s of the token being sampled (0 .. n_steps-1); the head kernel of a step increments it
    int pos0;        // absolute position of the first decoded token's transformer input (= prefix length)
    int eos_min;     // EOS masked while step < eos_min
    int pad;
};

struct GemvArgs {
    const void* x;            // x_mode 0: fp32 [m][ldx]; 1: fp16 [m][ldx]; 2: lm_attn's part_o [m][k/64][2][64] fp32
    const float* x2;          // x_mode 2: lm_attn's part_ml [m][k/64][2][2] (running max, sum)
    const int* gather;        // row indices into x (embedding lookup) or null; fp32 x only
    const float* pre_g;       // optional FIRST transform of the staged row (fp32 x, k <= 1024): x = pre_scale * relu(LayerNorm(x;
    const float* pre_b;       //   pre_g, pre_b)) -- the LM's input embedding (LayerNorm -> ReLU -> * sqrt(d)); block 0 writes the
    float* pre_out;           //   transformed rows to pre_out [m][ldx] (the residual stream a later kernel adds to)
    const float* ln_g;        // LayerNorm over the k inputs while staging (fp32 x only) or null
    const float* ln_b;
    const _Float16* w;        // [n_pad][kpad] fp16, K contiguous
    const float* bias;        // [n] or null
    const float* res;         // fp32 [m][ldr] or null
    float* out;               // fp32 [m][ldo] or null
    _Float16* out16;          // fp16 [m][ldo16] or null (both may be given)
    _Float16* kv;             // columns >= n_split go to the KV cache: column c = n - n_split of row r lands at
                              //   kv[(pos0 + step) * kv_t + r * kv_b + ((c % d) / 64) * kv_h + (c / d) * kv_v + c % 64],  d = (n - n_split) / 2
    LmStep* st;               // optional device-side step state (kv row = pos0 + step; advanced by block 0 when `advance`); null: `pos`
    unsigned long long* stamps;  // micro-benchmark builds only (LM_STAMPS)
    float ln_eps, pre_scale;
    int m, n, k, kpad, ldx, ldr, ldo, ldo16, n_split;
    int kv_t, kv_b, kv_h, kv_v;  // cache strides in halfs: time step, row, head, K -> V (see KvLayout)
    int x_mode, relu, advance, stamp_slot;
    int ln_plain;             // LayerNorm without scale / shift (folded into w / bias at load); ln_g must then be null
    int pos;                  // kv row when st is null
    int ksplit;               // 2: the launch carries TWO workgroups per column block, each over one half of K (fp16 input only, k == kpad); both
                              //   add their sum into `out` with ONE fp32 atomic each (slice 0 carries bias + residual).  `out` must hold zeros at
                              //   launch; two addends on a zero commute exactly, so the result does not depend on which arrives first
    float* zero;              // optional: the launch clears zero[0 .. zero_n) (the accumulator of a later split launch) -- no reader in between
    int zero_n;
};

struct AttnArgs {
    const float* q;           // [b][ldq] fp32 (the q third of the QKV projection)
    const _Float16* kv;       // key j of (row r, head hd): kv[j * kv_t + r * kv_b + hd * kv_h + 0..63], its value kv_v halfs further
    const _Float16* postab;   // [2*center+1][ldp] position projections of this layer
    const float* bias_u;
    const float* bias_v;
    const int* kstart;        // [b] first valid key (left padding) or null
    _Float16* out;            // [b][ldo] fp16 (ksplit == 1)
    float* part_o;            // ksplit == 2: [b][h][2][64] unnormalised partial outputs ...
    float* part_ml;           //              [b][h][2][2] ... with their running max and sum (merged by lm_gemv x_mode 2)
    const LmStep* st;         // optional device-side step state; null: `pos`
    unsigned long long* stamps;
    int b, h, ldq, ldo, ldp, center, d;
    int kv_t, kv_b, kv_h, kv_v;  // cache strides in halfs (KvLayout)
    float scale;
    int ksplit, stamp_slot;
    int pos;                  // absolute position of the query (= index of the newest key) when st is null
};

// The two cache layouts the step kernels are used with (any strides work: multiples of 8 halfs).
//   time-major [t][b][2d] (k | v): what the prefill GEMM writes -- one 128-byte piece per (key, row, head), b * 4d bytes apart
//   head-major [b][h][2][t_max][64]: a (row, head)'s keys and values are two contiguous streams -- what lm_attn wants at long
//   context (32 rows x ~1 000 keys: the time-major pieces are 128 KB apart and the kernel stalls at ~4 TB/s)
struct KvLayout {
    int t, b, h, v;
    static KvLayout time_major(int rows, int d) { return {rows * 2 * d, 2 * d, 64, d}; }
    static KvLayout head_major(int heads, int t_max) { return {64, heads * 2 * t_max * 64, 2 * t_max * 64, t_max * 64}; }
};

// launchers (no allocation, no synchronisation; graph-capturable).  Return ASTTS_OK or ASTTS_ERR_*.
int lm_gemv_launch(const GemvArgs& a, hipStream_t st);
int lm_attn_launch(const AttnArgs& a, hipStream_t st);
void lm_step_set_attrs();   // one-off hipFuncSetAttribute calls (outside any capture)
// the kernel variant lm_gemv_launch picks: bits 0-1 = form (0: 16 columns, 1: diagonal 8 x 8 for m <= 8, 2: halved 8 columns),
// bits 2.. = row tiles of 16 (1 or 2)
int lm_gemv_variant(const GemvArgs& a);


}  // namespace astts
