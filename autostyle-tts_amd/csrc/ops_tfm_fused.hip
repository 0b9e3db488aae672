// ops_tfm_fused.hip -- LayerNorm + q|k|v projection + masked multi-head attention of one BasicTransformerBlock of the
// flow-matching estimator in ONE launch (the reference's hot loop #3, SURVEY.md 3.1: ConditionalDecoder's transformer
// blocks behind cosyvoice.inference_tts_with_st, tts_with_rag.py:195; 48 of the 56 blocks of an estimator pass run at
// T = 344 frames x 16 sequences).
//
// Why: as three launches (layernorm_rows 5.2 us, gemm_ring q|k|v 12.3 us, attn_mha_flash 16.4 us + 3 x 1.45 us of kernel
// boundary) this part of a block takes 38 us for 8.2 GFLOP; each of them is bound by latency (a dependent kernel over a few MB
// of fresh data does not finish under ~5 us on this chip) and by workgroup granularity (384 attention workgroups on 256 CUs
// cost as much as 512), not by MFMA or HBM time.  Fused, the normalised rows, Q, K and V never leave the CU.
//
// Workgroup = (query half, head, sequence): 2 x 8 x 16 = 256 workgroups of 8 waves, one per CU.
//   phase 1  for every 32-frame chunk of the sequence: 512 threads load the fp32 rows, normalise them (two-pass statistics over
//            the 256 channels; the LayerNorm scale / shift are folded into the projection weights by the host) and stage them as
//            fp16 in LDS (double buffered); each wave owns one 32-feature tile of this head's K | V | Q projection, its
//            weights live in registers (64 VGPRs) for the whole phase; K goes to LDS row-major, V transposed, Q (only for the
//            chunks of this workgroup's query half) row-major.  Both workgroups of a (head, sequence) project all keys:
//            1.33x the projection flops, no exchange.
//   phase 2  each wave takes one 32-query tile: S^T = K Q^T, online softmax in the log2 domain, O^T += V^T P^T exactly as
//            attn_mha_flash (ops_attention.hip), but K and V^T are read from LDS where phase 1 left them: no staging, no
//            barrier, no global load in the key loop.
// LDS: K [TKP][72] + V^T [64][TKP + 4] + A [2][32][264] + Q [192][64] halfs = 151 KB at T = 344 (TKP = 352), 159.5 KB at TKP = 384: T <= 384.
#include "common.h"
#include "xlane.h"

namespace astts {

static constexpr int TF_C = 256;           // channels of the estimator's transformer blocks
static constexpr int TF_DH = 64;
static constexpr int TF_KS = 72;           // halfs per K / Q row in LDS (conflict-free ds_read_b128, as FA_KS)
static constexpr int TF_AS = TF_C + 8;     // halfs per staged A row
static constexpr int TF_QROWS = 192;       // queries per workgroup (6 tiles of 32)
static constexpr int TF_QS = 64;           // halfs per Q row in LDS: unpadded, 16-byte pieces XOR-swizzled by the row (tf_q)
static constexpr int TF_MAX_T = 384;       // 24 kHz, config 2: 749 frames -> 375 after the down block

// Q rows (and the output tile transposed through them) are touched a handful of times per workgroup, K rows once per key tile:
// the 8-half row padding goes to K only, Q rows are 128 bytes with piece p of row r stored at piece p ^ (r & 7) (two-way
// conflicts at worst) -- 3 KB less LDS, which is what lets T = 384 keys (K + V^T = 105 KB) fit beside staging and Q.
__device__ __forceinline__ int tf_q(int row, int col) { return row * TF_QS + ((((col >> 3) ^ row) & 7) << 3) + (col & 7); }

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k + 15), result on every lane: four row rotations on the VALU.
// (__shfl_xor compiles to ds_bpermute_b32: an LDS round trip of ~100 cycles per step, eight steps per staged chunk.)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    return v;
}

struct TfmAttnArgs {
    const float* x;          // [b, t, 256] fp32 residual stream
    const _Float16* w;       // q | k | v weight (LayerNorm scale folded in) in fragment order: [3 * heads * 2 tiles][16][64][8] fp16
    const float* bias;       // [3 * heads * 64] fp32 (W beta, + the projection's own bias if any) or null
    const int* lens;         // [b] valid frames or null
    _Float16* out;           // [b, t, heads * 64] fp16
    int b, heads, t;
    float eps, scale;
    int balance;             // split the key range of the tiles owned by waves 4 / 5 with the otherwise idle waves
    const char* pf[3];       // L2 prefetch of the NEXT launch's weights (cold otherwise: every block has its own): up to three ranges,
    unsigned pf_bytes[4];    // touched one 128-byte line per thread by the workgroups of each XCD at the start of phase 2.  ([3] is unused but
                             // READ: the compiler fetches the four dwords with one wide scalar load at kernel entry; a dead fourth dword's SGPR
                             // is reused at once, and writing it under the in-flight load costs an s_waitcnt lgkmcnt(0) in front of the first loads)
};

// The kernel body: workgroup L of 2 * heads * b.  Every wave returns from it (no early exit, so that a caller can continue in the
// same launch: the one-launch block kernel measured in DESIGN.md did); seq_out / unit_out: this workgroup's sequence and its index
// (head * 2 + query half) among the sequence's.
// FULL: one workgroup per (head, sequence) takes BOTH query halves, one after the other -- K and V are projected once instead of twice,
// the weights and the sequence's rows are fetched and normalised once (batches of more than 16 sequences, where the 2 x heads x b
// workgroups of the half form need several rounds on the 256 CUs anyway).  LDS has no room for the second half's Q rows: phase 1 parks them
// in the rows of the OUTPUT tensor they will be replaced by (same size, same owner), and they come back into the Q buffer between the two
// attention passes.  Every row goes through the same arithmetic in the same order as in the half form: the results are bit-identical.
template <bool FULL>
__device__ __forceinline__ void tfm_attn_body(const TfmAttnArgs& a, int* seq_out, int* unit_out) {
    extern __shared__ __attribute__((aligned(16))) _Float16 tf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    // Workgroup -> (sequence, head, query half).  Hardware deals workgroups round-robin over the 8 XCDs (id mod 8) and every XCD
    // has its own L2: with the plain (x, y, z) order the 16 workgroups of a sequence sit on all 8 XCDs and every L2 fetches the whole
    // activation tensor (8 x 5.6 MB of fabric traffic per launch at 16 x 344 x 256 fp32: the kernel ran at that pace).  When the
    // batch is a multiple of 8, sequence b goes to XCD b mod 8: one L2 fetches its rows once for all 16 workgroups.
    int qs, head, b;
    {
        const int units = FULL ? a.heads : 2 * a.heads;  // workgroups per sequence
        const int L = blockIdx.x;
        int hq;
        if ((a.b & 7) == 0) {
            const int xcd = L & 7, slot = L >> 3;
            b = xcd + 8 * (slot / units);
            hq = slot % units;
        } else {
            b = L / units;
            hq = L % units;
        }
        head = FULL ? hq : hq >> 1;
        qs = FULL ? 0 : hq & 1;
    }
    const int T = a.t;
    const int nch = (T + 31) >> 5;                   // 32-frame chunks
    const int tkp = nch * 32;
    const int vs = tkp + 4;                          // halfs per V^T row: (tkp / 2 + 2) dwords = 2 * odd multiple: conflict-free b64 reads
    const int hsplit = (nch + 1) / 2;                // query half 0 = chunks [0, hsplit), half 1 = [hsplit, nch)
    const int pq0 = FULL ? 0 : (qs == 0 ? 0 : hsplit);     // chunks whose Q rows this workgroup projects: [pq0, pq1)
    const int pq1 = FULL ? nch : (qs == 0 ? hsplit : nch);
    _Float16* sK = tf_smem;                          // [tkp][72]
    _Float16* sVt = sK + (size_t)tkp * TF_KS;        // [64][vs]
    _Float16* sA = sVt + (size_t)TF_DH * vs;         // [2][32][264]
    _Float16* sQ = sA + 2 * 32 * TF_AS;              // [192][64], swizzled (tf_q)
    const float* xb = a.x + (int64_t)b * T * TF_C;
    const int hd = a.heads * TF_DH;

    // ---- this wave's projection weights: one 32-feature tile of K (waves 0, 1), V (2, 3) or Q (4, 5), K = 256 in registers
    // (waves 6, 7 only take part in the staging: giving them the Q tiles of every other chunk balances the MFMAs over the SIMDs
    // but costs two more 16 KB weight fetches per workgroup -- measured slower)
    const int role = wid >> 1;                       // 0 K, 1 V, 2 Q, 3 none
    half8 wf[16];
    float bias_e[16];
    if (role < 3) {
        const int part = role == 0 ? 1 : (role == 1 ? 2 : 0);           // row block of the fused q | k | v weight
        const int frow = part * hd + head * TF_DH + (wid & 1) * 32;
        // fragment-ordered image (astts_op_tfm_pack_qkv): [32-feature tile][k-step][lane][8 halfs] -- one wave instruction reads
        // 1 KB of consecutive bytes (the row-major image puts a lane's 16 bytes 512 bytes apart: 64 sectors per instruction, a
        // quarter of each used; the 96 KB of weights a workgroup takes into registers cost 5.5 us that way)
        const _Float16* wp = a.w + ((int64_t)(frow >> 5) * 16 * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < 16; ++s) wf[s] = *reinterpret_cast<const half8*>(wp + (int64_t)s * 64 * 8);
        // bias of the features this lane's accumulator elements hold.  K / Q tiles: feature on the element index
        // (f_e = (e & 3) + 8 (e >> 2) + 4 hh); V tiles: feature on the lane (c)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int f = role == 1 ? c : (e & 3) + 8 * (e >> 2) + 4 * hh;
            bias_e[e] = a.bias ? a.bias[frow + f] : 0.0f;
        }
    }

    // ---- phase 1: normalise + project, chunk by chunk
    const int srow = tid >> 4, sseg = (tid & 15) * 16;      // staging: 16 threads per frame, 16 channels each
    // PF chunks of rows in flight per thread (PF x 4 float4); a workgroup reads its whole sequence (T x 1 KB of fp32 rows, shared
    // through L2 with the 15 other workgroups of the sequence)
    constexpr int PF = 2;
    float4 xr[PF][4];
    auto load_chunk = [&](int ch, float4 (&r)[4]) {
        const int fr = min(ch * 32 + srow, T - 1);
        const float* p = xb + (int64_t)fr * TF_C + sseg;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(p + 4 * i);
    };
    auto stage_chunk = [&](int buf, const float4 (&r)[4]) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += (r[i].x + r[i].y) + (r[i].z + r[i].w);
        s = row16_sum(s);
        const float mean = s * (1.0f / TF_C);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dx = r[i].x - mean, dy = r[i].y - mean, dz = r[i].z - mean, dw = r[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        q = row16_sum(q);
        const float rstd = rsqrtf(q * (1.0f / TF_C) + a.eps);
        half8 h0, h1;
        h0[0] = (_Float16)((r[0].x - mean) * rstd); h0[1] = (_Float16)((r[0].y - mean) * rstd);
        h0[2] = (_Float16)((r[0].z - mean) * rstd); h0[3] = (_Float16)((r[0].w - mean) * rstd);
        h0[4] = (_Float16)((r[1].x - mean) * rstd); h0[5] = (_Float16)((r[1].y - mean) * rstd);
        h0[6] = (_Float16)((r[1].z - mean) * rstd); h0[7] = (_Float16)((r[1].w - mean) * rstd);
        h1[0] = (_Float16)((r[2].x - mean) * rstd); h1[1] = (_Float16)((r[2].y - mean) * rstd);
        h1[2] = (_Float16)((r[2].z - mean) * rstd); h1[3] = (_Float16)((r[2].w - mean) * rstd);
        h1[4] = (_Float16)((r[3].x - mean) * rstd); h1[5] = (_Float16)((r[3].y - mean) * rstd);
        h1[6] = (_Float16)((r[3].z - mean) * rstd); h1[7] = (_Float16)((r[3].w - mean) * rstd);
        _Float16* d = sA + (size_t)buf * 32 * TF_AS + srow * TF_AS + sseg;
        *reinterpret_cast<half8*>(d) = h0;
        *reinterpret_cast<half8*>(d + 8) = h1;
    };
    auto project_chunk = [&](int ch) {
        const int buf = ch & 1;
        const bool q_chunk = ch >= pq0 && ch < pq1;
        const bool q_parked = FULL && ch >= hsplit;      // second half's Q rows: to the output tensor's rows for now
        const int qbase = FULL ? 0 : pq0;
        if (role < 2 || (role == 2 && q_chunk)) {
            const _Float16* ap = sA + (size_t)buf * 32 * TF_AS + c * TF_AS + 8 * hh;
            float16v acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = bias_e[e];
            if (role == 1) {                            // V: frame on the element index, feature on the lane
#pragma unroll
                for (int s0 = 0; s0 < 16; s0 += 8) {    // eight fragment reads in flight, then their MFMAs (a read issued right
                    half8 af[8];                        // before its MFMA exposes the LDS latency sixteen times per chunk)
#pragma unroll
                    for (int s = 0; s < 8; ++s) af[s] = *reinterpret_cast<const half8*>(ap + 16 * (s0 + s));
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], wf[s0 + s], acc, 0, 0, 0);
                }
                _Float16* vp = sVt + (size_t)((wid & 1) * 32 + c) * vs + ch * 32 + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 h4;
                    h4[0] = (_Float16)acc[4 * g]; h4[1] = (_Float16)acc[4 * g + 1]; h4[2] = (_Float16)acc[4 * g + 2]; h4[3] = (_Float16)acc[4 * g + 3];
                    *reinterpret_cast<half4*>(vp + 8 * g) = h4;
                }
            } else {                                    // K / Q: feature on the element index, frame on the lane
#pragma unroll
                for (int s0 = 0; s0 < 16; s0 += 8) {
                    half8 af[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) af[s] = *reinterpret_cast<const half8*>(ap + 16 * (s0 + s));
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s0 + s], af[s], acc, 0, 0, 0);
                }
                _Float16* kp = sK + (size_t)(ch * 32 + c) * TF_KS + (wid & 1) * 32 + 4 * hh;
                const int qr = (ch - qbase) * 32 + c;
                const int fr = ch * 32 + c;
                _Float16* park = a.out + ((int64_t)b * T + min(fr, T - 1)) * hd + head * TF_DH + (wid & 1) * 32 + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 h4;
                    h4[0] = (_Float16)acc[4 * g]; h4[1] = (_Float16)acc[4 * g + 1]; h4[2] = (_Float16)acc[4 * g + 2]; h4[3] = (_Float16)acc[4 * g + 3];
                    if (role == 0) *reinterpret_cast<half4*>(kp + 8 * g) = h4;
                    else if (!q_parked) *reinterpret_cast<half4*>(sQ + tf_q(qr, (wid & 1) * 32 + 4 * hh + 8 * g)) = h4;
                    else if (fr < T) *reinterpret_cast<half4*>(park + 8 * g) = h4;
                }
            }
        }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < nch) load_chunk(i, xr[i]);
    stage_chunk(0, xr[0]);
    if (PF < nch) load_chunk(PF, xr[0]);
    __syncthreads();
    // chunk ch is projected from buffer ch & 1 while chunk ch + 1 is normalised into the other buffer and chunk ch + 1 + PF is
    // requested into the register set that just emptied (register set of chunk k: k mod PF; the loop is unrolled by PF so that
    // every set index is a compile-time constant)
    for (int ch0 = 0; ch0 < nch; ch0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int ch = ch0 + u;
            if (ch < nch) {                             // workgroup-uniform
                if (ch + 1 < nch) stage_chunk((ch + 1) & 1, xr[(u + 1) % PF]);
                if (ch + 1 + PF < nch) load_chunk(ch + 1 + PF, xr[(u + 1) % PF]);
                project_chunk(ch);
                __syncthreads();
            }
        }
    }

    // ---- phase 2: 32-query tiles, keys and values straight from LDS.  Waves 0..3 own tiles 0..3.  Waves w and w + 4 share a
    // SIMD, so a fifth / sixth tile given whole to waves 4 / 5 would load SIMDs 0 / 1 with 22 key tiles against 11 on SIMDs
    // 2 / 3: instead the key range of tile 4 (5) is split between waves 4 and 6 (5 and 7) -- four ways over waves 4..7 when there
    // are five tiles -- and the owner merges the helpers' unnormalised partials through LDS (the staging buffers are free now):
    // 16.5 tile-units per SIMD.
    unsigned pf_keep[3];
    *seq_out = b;
    *unit_out = FULL ? head * 2 : head * 2 + qs;
    for (int hf = FULL ? 0 : qs; hf < (FULL ? 2 : qs + 1); ++hf) {
    const int qch0 = hf == 0 ? 0 : hsplit;           // this pass's query chunks [qch0, qch1)
    const int qch1 = hf == 0 ? hsplit : nch;
    if (FULL && hf == 1) {
        // the parked Q rows come back: every wave's stores of phase 1 are ordered before the barrier, the loads behind it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();                             // (also: pass 0's owners are done with the Q buffer as their transpose buffer)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int rows = (nch - hsplit) * 32;
        for (int i = tid; i < rows * 8; i += 512) {
            const int r = i >> 3, pc = i & 7;
            const int fr = min(hsplit * 32 + r, T - 1);
            const half8 v = *reinterpret_cast<const half8*>(a.out + ((int64_t)b * T + fr) * hd + head * TF_DH + pc * 8);
            *reinterpret_cast<half8*>(sQ + tf_q(r, pc * 8)) = v;
        }
        __syncthreads();
    }
    const int ntile = qch1 - qch0;
    int tile = -1, part = 0, parts = 1;
    if (wid < 4) {
        if (wid < ntile) tile = wid;
    } else if (!a.balance) {
        if (wid < ntile) tile = wid;
    } else if (ntile == 5) {
        tile = 4; part = wid - 4; parts = 4;
    } else if (ntile == 6) {
        tile = 4 + (wid & 1); part = (wid - 4) >> 1; parts = 2;
    }
    if (hf == (FULL ? 0 : qs)) {   // L2 prefetch of the next launches' weights: the workgroups of one XCD (ids congruent mod 8) split each range between them, one
        // 128-byte line per thread and range (a range beyond 512 lines per slot -- none today -- is left to its own launch).  Untracked asm
        // loads (xlane.h prefetch_line): the volatile loads used before compiled to system-scope flat loads with an immediate
        // s_waitcnt vmcnt(0) each -- the waves owning query tiles 0 and 1 sat out three HBM misses in a row before their first score
        const unsigned slot = blockIdx.x >> 3, nslots = max(gridDim.x >> 3, 1u);
        asm volatile("" ::"s"(a.pf_bytes[3]));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const unsigned lines = a.pf[r] ? (a.pf_bytes[r] + 127) >> 7 : 0u;
            const unsigned per = (lines + nslots - 1) / nslots;
            const unsigned ln = slot * per + tid;
            const char* base = a.pf[r] ? a.pf[r] : reinterpret_cast<const char*>(a.x);
            prefetch_line(base + (tid < per && ln < lines ? (size_t)ln << 7 : (size_t)0), pf_keep[r]);
        }
    }
    const int len = a.lens ? min(a.lens[b], T) : T;     // (here, where it is first needed: at the top its scalar load's wait stood in front of
                                                        // the weight and row loads -- an s_waitcnt lgkmcnt covers every scalar load in flight)
    const int nkt = (len + 31) >> 5;                    // key tiles with at least one valid key
    const int jb0 = tile >= 0 ? (nkt * part / parts) * 32 : 0;
    const int jb1 = tile >= 0 ? min((nkt * (part + 1) / parts) * 32, len) : 0;
    const int tsafe = tile >= 0 ? tile : 0;
    const int q0 = (qch0 + tsafe) * 32;                 // first frame of this wave's query tile
    _Float16* qrow = sQ + (size_t)(tsafe * 32) * TF_QS;   // this tile's Q rows (32 rows: the swizzle is tile-local); reused by its owner as the output transpose buffer
    half8 qf[4];
    {
        const float qscale = a.scale * 1.44269504088896341f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const half8 raw = *reinterpret_cast<const half8*>(qrow + tf_q(c, 16 * s + 8 * hh));
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[s][i] = (_Float16)((float)raw[i] * qscale);
        }
    }
    float16v ot[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        ot[0][e] = 0.0f;
        ot[1][e] = 0.0f;
    }
    float m_run = -INFINITY, l_run = 0.0f;
    for (int jb = jb0; jb < jb1; jb += 32) {
        float16v st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.0f;
        half8 kf[4];
        half4 vlo[4], vhi[4];                           // this key tile's K and V^T fragments: all requested before the first MFMA
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const half8*>(sK + (size_t)(jb + c) * TF_KS + 16 * s + 8 * hh);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const _Float16* vrow = sVt + (size_t)(dt * 32 + c) * vs + jb + 16 * s + 4 * hh;
                vlo[dt * 2 + s] = *reinterpret_cast<const half4*>(vrow);
                vhi[dt * 2 + s] = *reinterpret_cast<const half4*>(vrow + 8);
            }
#pragma unroll
        for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qf[s], st, 0, 0, 0);
        if (jb + 32 > len) {                            // wave-uniform: only the last key tile is ragged
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
        if (__builtin_amdgcn_ballot_w64(mloc > m_run) != 0) {
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
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
            l_run += p;
            pf[e >> 3][e & 7] = (_Float16)p;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half4 lo = vlo[dt * 2 + s], hi = vhi[dt * 2 + s];
                half8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], ot[dt], 0, 0, 0);
            }
        }
    }
    // ---- helpers hand their partial (O^T, running maximum, running sum) to the tile's owner
    float* sM = reinterpret_cast<float*>(sA);            // [3 slots][34][64] fp32 in the (now idle) staging buffers
    if (part > 0) {
        float* mp = sM + (size_t)(ntile == 5 ? part - 1 : wid - 6) * 34 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            mp[e * 64] = ot[0][e];
            mp[(16 + e) * 64] = ot[1][e];
        }
        mp[32 * 64] = m_run;
        mp[33 * 64] = l_run;
    }
    __syncthreads();
    if (!(tile < 0 || part > 0)) {
    for (int hp = 1; hp < parts; ++hp) {
        const float* mp = sM + (size_t)(ntile == 5 ? hp - 1 : wid - 4) * 34 * 64 + lane;
        const float m_p = mp[32 * 64], l_p = mp[33 * 64];
        const float m_new = fmaxf(m_run, m_p);
        if (m_new > -INFINITY) {                         // lane-local: a lane's maximum belongs to its query
            const float sa = __builtin_amdgcn_exp2f(m_run - m_new), sb = __builtin_amdgcn_exp2f(m_p - m_new);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                ot[0][e] = ot[0][e] * sa + mp[e * 64] * sb;
                ot[1][e] = ot[1][e] * sa + mp[(16 + e) * 64] * sb;
            }
            l_run = l_run * sa + l_p * sb;
            m_run = m_new;
        }
    }
    // ---- O^T (dims on the element index, query on the lane) -> this tile's Q rows in LDS (no longer needed) -> coalesced rows
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 h4;
            h4[0] = (_Float16)(ot[dt][4 * g] * inv); h4[1] = (_Float16)(ot[dt][4 * g + 1] * inv);
            h4[2] = (_Float16)(ot[dt][4 * g + 2] * inv); h4[3] = (_Float16)(ot[dt][4 * g + 3] * inv);
            *reinterpret_cast<half4*>(qrow + tf_q(c, dt * 32 + 8 * g + 4 * hh)) = h4;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 8 + (lane >> 3), seg = (lane & 7) * 8;
        const int fr = q0 + r;
        if (fr < T)
            *reinterpret_cast<half8*>(a.out + ((int64_t)b * T + fr) * hd + head * TF_DH + seg) = *reinterpret_cast<const half8*>(qrow + tf_q(r, seg));
    }
    }
    }   // query halves
    prefetch_keep(pf_keep[0]);                         // the prefetch destinations stay reserved to the end
    prefetch_keep(pf_keep[1]);
    prefetch_keep(pf_keep[2]);
}

// Leading parameters = what the first loads (weight fragments, the sequence's rows) need: preloaded into SGPRs by the command
// processor (-amdgpu-kernarg-preload-count, csrc/Makefile; a by-value struct is not), the struct carries the rest.
template <bool FULL>
__global__ __launch_bounds__(512, 2) void tfm_attn_fused(const float* p_x, const _Float16* p_w, const float* p_bias, const int* p_lens, int p_b, int p_heads,
                                                         int p_t, float p_eps, float p_scale, TfmAttnArgs a_in) {
    TfmAttnArgs a = a_in;
    a.x = p_x; a.w = p_w; a.bias = p_bias; a.lens = p_lens; a.b = p_b; a.heads = p_heads; a.t = p_t; a.eps = p_eps; a.scale = p_scale;
    int seq, unit;
    tfm_attn_body<FULL>(a, &seq, &unit);
}

// row-major fp16 [rows][k] (astts_op_pack_weight image) -> fragment order [rows / 32][k / 16 k-steps][64 lanes][8]:
// lane (c = lane & 31, hh = lane >> 5) of k-step s holds W[32 tile + c][16 s + 8 hh + j]: one wave instruction loads 1 KB of
// consecutive bytes
__global__ void tfm_pack_frag(const _Float16* __restrict__ w, _Float16* __restrict__ out, int rows, int k) {
    const int64_t total = (int64_t)rows * k;
    const int ksteps = k >> 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), ln = (int)((i >> 3) & 63);
        const int64_t q = i >> 9;
        const int s = (int)(q % ksteps);
        const int64_t tile = q / ksteps;
        out[i] = w[(tile * 32 + (ln & 31)) * k + 16 * s + 8 * (ln >> 5) + j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// LayerNorm + Linear(256 -> hidden) + GELU + Linear(hidden -> 256) + residual of one BasicTransformerBlock in ONE launch.
//
// Why: as three launches (layernorm_rows 5.1 us, gemm_ring 256->1024 + GELU 11.8 us, gemm_ring 1024->256 + residual 9.4 us,
// 3 x 1.45 us of boundary) the feed-forward part of a block takes ~31 us for 5.8 GFLOP at 16 x 344 rows: each of the three pays
// the ~5 us a dependent kernel needs to get going on fresh data.  Fused, a workgroup owns 32 rows: it normalises them once into
// LDS, streams the 1 MB of W1 | W2 (fragment order: every wave instruction is one 1 KB burst, 32 KB per wave in flight) through
// registers, and the 32 x hidden intermediate never leaves the CU.
//
// Workgroup = 32 rows, 8 waves.  hidden is walked in chunks of 256:
//   stage 1  wave w: H[:, 32 (8 j + w) ..+32] = gelu(A W1^T + b1): 16 MFMA 32x32x16 (weights as the A operand: hidden feature
//            on the accumulator element, row on the lane), fp16 to LDS (double buffered), one barrier
//   stage 2  wave w: Y[:, 32 w ..+32] += H_chunk W2^T: 16 MFMA (H as the A operand: row on the element, output feature on the
//            lane -> 128-byte row segments in the epilogue)
// software-pipelined: stage 1 of chunk j + 1 is issued before stage 2 of chunk j, whose MFMAs are woven with the GELU of chunk
// j + 1 (the GELU's VALU work is the larger share of a chunk's issue cycles).  The next chunk's W1 / W2 fragments are requested
// right after the MFMAs that consumed the current ones.
//
// Tried and dropped: 64 rows x half of the hidden features per workgroup (half the weight stream per CU), the two halves joined
// through L2 with an arrival flag: 27 us against 19 -- the weight stream is ~3.4 us of this kernel, not its bound, and the hand-over
// (with agent-scope fences: buffer_wbl2 / buffer_inv, 65 us) costs more than it saves.
static constexpr int FF_HS = 256 + 8;      // halfs per staged H row

struct TfmFfnArgs {
    const float* x;          // [m][256] fp32 residual stream
    const _Float16* w1;      // [hidden][256] fp16, LayerNorm scale folded in, fragment order
    const float* b1;         // [hidden] (W1 beta + b1)
    const _Float16* w2;      // [256][hidden] fp16, fragment order
    const float* b2;         // [256] or null
    float* out;              // [m][256]: x' + W2 gelu(W1 LN(x') + b1) + b2
    const _Float16* attn;    // WO: [m][k0] fp16 attention output; x' = x + attn Wo^T + bo (otherwise x' = x)
    const _Float16* wo;      // WO: [256][k0] fp16, fragment order
    const float* bo;         // WO: [256] or null
    int64_t m;
    int hidden, k0;
    float eps;
    const char* pf;          // L2 prefetch of the next launch's weights (one range), touched one 128-byte line per thread
    unsigned pf_bytes;
};

// WO: the attention's output projection + residual (the launch between tfm_attn_fused and this one: 8.5 us + boundary) runs as a
// prologue here: x' = x + attn Wo^T + bo is formed in LDS (fp32), normalised from there, and added back in the epilogue; it is
// never written to memory.
// The kernel body for the 32-row tile [m0, mend) (mend - m0 <= 32: rows at or beyond mend are neither read as themselves nor
// written); rot_seed spreads the chunk order over the workgroups of an XCD.
template <bool WO>
__device__ __forceinline__ void tfm_ffn_body(const TfmFfnArgs& a, const int64_t m0, const int64_t mend, const unsigned rot_seed) {
    extern __shared__ __attribute__((aligned(16))) _Float16 tf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    _Float16* sA = tf_smem;                          // [32][264] normalised rows
    _Float16* sH = sA + 32 * TF_AS;                  // [2][32][264] one 256-wide chunk of the hidden activations
    float* sB1 = reinterpret_cast<float*>(sH + 2 * 32 * FF_HS);   // [hidden]
    const int nchunk = a.hidden >> 8;
    const int ksteps2 = a.hidden >> 4;               // k-steps of a W2 tile

    // ---- rows first (needed first), then the first chunk's weights.  16 threads per row; thread i of a row takes channels
    // 4 i + 64 k (k = 0..3): every wave instruction reads 256 consecutive bytes of each of its 4 rows
    const int srow = tid >> 4, sseg = (tid & 15) * 4;
    float4 r[4];
    {
        const float* p = a.x + min(m0 + srow, mend - 1) * TF_C + sseg;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(p + 64 * i);
    }
    // The workgroups of one XCD (block ids congruent mod 8) start at different chunks and walk them cyclically: every block's
    // weights are cold in L2 (56 blocks x 2 MB per estimator pass), and with one common order all workgroups wait for the same
    // HBM lines chunk after chunk; rotated, the whole weight image is requested at once.  (The sum over chunks is taken in that
    // order.)
    const int rot = (int)(rot_seed % (unsigned)nchunk);
    auto wrap = [&](int j) { return j >= nchunk ? j - nchunk : j; };
    half8 w1f[16], w2f[16];
    const _Float16* w1p = a.w1 + ((int64_t)wid * 16 * 64 + lane) * 8;             // hidden tile 8 j + wid: + j * 8 tiles
    const _Float16* w2p = a.w2 + ((int64_t)wid * ksteps2 * 64 + lane) * 8;        // output tile wid, k-step 16 j + s
    auto load_w1 = [&](int j) {
        const _Float16* np = w1p + (int64_t)j * 8 * 16 * 512;
#pragma unroll
        for (int s = 0; s < 16; ++s) w1f[s] = *reinterpret_cast<const half8*>(np + (int64_t)s * 512);
    };
    auto load_w2 = [&](int j) {
        const _Float16* np = w2p + (int64_t)j * 16 * 512;
#pragma unroll
        for (int s = 0; s < 16; ++s) w2f[s] = *reinterpret_cast<const half8*>(np + (int64_t)s * 512);
    };
    float* sX = sB1 + a.hidden;                      // WO: [32][260] fp32 x'
    if constexpr (WO) {
        // ---- x' = x + attn Wo^T + bo.  The attention rows go to LDS where the hidden chunks will live later; Wo streams through
        // the W2 registers in halves of 256 input features while W1's first chunk is already on its way.
        const int as0 = a.k0 + 8;                    // halfs per staged attention row
        _Float16* sAtt = sH;                         // [32][k0 + 8] (k0 <= 512: fits the two H buffers)
        const int nh = a.k0 >> 8;
        const _Float16* wop = a.wo + ((int64_t)wid * (a.k0 >> 4) * 64 + lane) * 8;
        half8 at[4];
        {
            const _Float16* p = a.attn + min(m0 + srow, mend - 1) * a.k0 + (tid & 15) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i * 128 < a.k0) at[i] = *reinterpret_cast<const half8*>(p + 128 * i);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) w2f[s] = *reinterpret_cast<const half8*>(wop + (int64_t)s * 512);
        load_w1(rot);
        for (int i = tid; i < a.hidden; i += 512) sB1[i] = a.b1 ? a.b1[i] : 0.0f;
        float16v acc0;
        {
            const float* bp = a.bo + wid * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = a.bo ? *reinterpret_cast<const float4*>(bp + 8 * g) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                acc0[4 * g] = b4.x; acc0[4 * g + 1] = b4.y; acc0[4 * g + 2] = b4.z; acc0[4 * g + 3] = b4.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(sX + srow * 260 + sseg + 64 * i) = r[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i * 128 < a.k0) *reinterpret_cast<half8*>(sAtt + srow * as0 + (tid & 15) * 8 + 128 * i) = at[i];
        __syncthreads();
        const _Float16* atp = sAtt + c * as0 + 8 * hh;
        for (int h = 0; h < nh; ++h) {
#pragma unroll
            for (int s0 = 0; s0 < 16; s0 += 8) {
                half8 af[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) af[s] = *reinterpret_cast<const half8*>(atp + 16 * (16 * h + s0 + s));
#pragma unroll
                for (int s = 0; s < 8; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[s0 + s], af[s], acc0, 0, 0, 0);
            }
            if (h + 1 < nh) {
                const _Float16* np = wop + (int64_t)(h + 1) * 16 * 512;
#pragma unroll
                for (int s = 0; s < 16; ++s) w2f[s] = *reinterpret_cast<const half8*>(np + (int64_t)s * 512);
            }
        }
        load_w2(rot);
        {
            float* xp = sX + c * 260 + wid * 32 + 4 * hh;    // feature (e & 3) + 8 (e >> 2) + 4 hh of this wave's tile, row c
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = *reinterpret_cast<float4*>(xp + 8 * g);
                v.x += acc0[4 * g]; v.y += acc0[4 * g + 1]; v.z += acc0[4 * g + 2]; v.w += acc0[4 * g + 3];
                *reinterpret_cast<float4*>(xp + 8 * g) = v;
            }
        }
        __syncthreads();                             // x' complete; the attention rows are dead (the H buffers are free)
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(sX + srow * 260 + sseg + 64 * i);
    } else {
        load_w1(rot);
        load_w2(rot);
        for (int i = tid; i < a.hidden; i += 512) sB1[i] = a.b1 ? a.b1[i] : 0.0f;
    }
    {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += (r[i].x + r[i].y) + (r[i].z + r[i].w);
        s = row16_sum(s);
        const float mean = s * (1.0f / TF_C);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dx = r[i].x - mean, dy = r[i].y - mean, dz = r[i].z - mean, dw = r[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        q = row16_sum(q);
        const float rstd = rsqrtf(q * (1.0f / TF_C) + a.eps);
        _Float16* d = sA + srow * TF_AS + sseg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            half4 h4;
            h4[0] = (_Float16)((r[i].x - mean) * rstd); h4[1] = (_Float16)((r[i].y - mean) * rstd);
            h4[2] = (_Float16)((r[i].z - mean) * rstd); h4[3] = (_Float16)((r[i].w - mean) * rstd);
            *reinterpret_cast<half4*>(d + 64 * i) = h4;
        }
    }
    __syncthreads();

    unsigned pf_keep;
    {   // workgroups of one XCD (ids congruent mod 8) split the range; see tfm_attn_fused
        const unsigned slot = blockIdx.x >> 3, nslots = max((gridDim.x + 7) >> 3, 1u);
        const unsigned lines = a.pf ? (a.pf_bytes + 127) >> 7 : 0u;
        const unsigned per = (lines + nslots - 1) / nslots;
        const unsigned ln = slot * per + tid;
        const char* base = a.pf ? a.pf : reinterpret_cast<const char*>(a.x);
        prefetch_line(base + (tid < per && ln < lines ? (size_t)ln << 7 : (size_t)0), pf_keep);
    }
    float16v acc2;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[e] = 0.0f;
    const _Float16* ap = sA + c * TF_AS + 8 * hh;
    // stage 1 of chunk j: this wave's 32 hidden features (bias in the accumulator)
    auto stage1 = [&](int j, float16v& acc1) {
        const float* bp = sB1 + (j * 8 + wid) * 32 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
            acc1[4 * g] = b4.x; acc1[4 * g + 1] = b4.y; acc1[4 * g + 2] = b4.z; acc1[4 * g + 3] = b4.w;
        }
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 8) {
            half8 af[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) af[s] = *reinterpret_cast<const half8*>(ap + 16 * (s0 + s));
#pragma unroll
            for (int s = 0; s < 8; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[s0 + s], af[s], acc1, 0, 0, 0);
        }
    };
    auto store_h = [&](int buf, const float (&gl)[16]) {
        _Float16* hp = sH + (size_t)buf * 32 * FF_HS + c * FF_HS + wid * 32 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 h4;
            h4[0] = (_Float16)gl[4 * g]; h4[1] = (_Float16)gl[4 * g + 1]; h4[2] = (_Float16)gl[4 * g + 2]; h4[3] = (_Float16)gl[4 * g + 3];
            *reinterpret_cast<half4*>(hp + 8 * g) = h4;
        }
    };
    {
        float16v acc1;
        stage1(rot, acc1);
        if (nchunk > 1) load_w1(wrap(rot + 1));
        float gl[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) gl[e] = gelu_erf_fast(acc1[e]);
        store_h(0, gl);
    }
    __syncthreads();
    for (int it = 0; it + 1 < nchunk; ++it) {
        const int jn = wrap(it + 1 + rot);
        float16v acc1;
        stage1(jn, acc1);
        if (it + 2 < nchunk) load_w1(wrap(jn + 1));
        // stage 2 of chunk `it` with the GELU of chunk it + 1 woven in by hand: every MFMA is followed by one element's GELU
        // (~15 VALU instructions that issue while the MFMA runs).  The empty asm statements tie the MFMA's operand and the GELU's
        // input / result to their place in the instruction stream (sched_barrier alone orders nothing before instruction
        // selection; the MFMA reads hf[s], so it cannot sink below the second one).
        const _Float16* hp = sH + (size_t)(it & 1) * 32 * FF_HS + c * FF_HS + 8 * hh;
        float gl[16];
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 8) {
            half8 hf[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) hf[s] = *reinterpret_cast<const half8*>(hp + 16 * (s0 + s));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float gx = acc1[s0 + s];
                asm volatile("" : "+v"(hf[s]), "+v"(gx));
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf[s], w2f[s0 + s], acc2, 0, 0, 0);
                float gy = gelu_erf_fast(gx);
                asm volatile("" : "+v"(gy), "+v"(hf[s]));
                gl[s0 + s] = gy;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        load_w2(jn);
        store_h((it + 1) & 1, gl);
        __syncthreads();
    }
    // ---- last stage 2; the residual rows are requested before it.  Element e holds row (e & 3) + 8 (e >> 2) + 4 hh, the lane's
    // output feature is 32 wid + c: 128-byte row segments
    const int f = wid * 32 + c;
    float xr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int re = (e & 3) + 8 * (e >> 2) + 4 * hh;
        xr[e] = WO ? sX[re * 260 + f] : a.x[min(m0 + re, mend - 1) * TF_C + f];
    }
    const float b2 = a.b2 ? a.b2[f] : 0.0f;
    {
        const _Float16* hp = sH + (size_t)((nchunk - 1) & 1) * 32 * FF_HS + c * FF_HS + 8 * hh;
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 8) {
            half8 hf[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) hf[s] = *reinterpret_cast<const half8*>(hp + 16 * (s0 + s));
#pragma unroll
            for (int s = 0; s < 8; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf[s], w2f[s0 + s], acc2, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (row < mend) a.out[row * TF_C + f] = (xr[e] + b2) + acc2[e];
    }
    prefetch_keep(pf_keep);
}

template <bool WO>
__global__ __launch_bounds__(512, 1) void tfm_ffn_fused(const float* p_x, const _Float16* p_w1, const _Float16* p_attn, const _Float16* p_wo, int64_t p_m,
                                                        int p_hidden, int p_k0, float p_eps, TfmFfnArgs a_in) {
    TfmFfnArgs a = a_in;      // leading parameters: preloaded into SGPRs (see tfm_attn_fused)
    a.x = p_x; a.w1 = p_w1; a.attn = p_attn; a.wo = p_wo; a.m = p_m; a.hidden = p_hidden; a.k0 = p_k0; a.eps = p_eps;
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    tfm_ffn_body<WO>(a, m0, min(m0 + 32, a.m), blockIdx.x >> 3);
}

}  // namespace astts

using namespace astts;

extern "C" {

int astts_op_tfm_pack_frag(const void* w_f16, void* out_f16, int32_t rows, int32_t k, astts_stream_t stream) {
    ASTTS_REQUIRE(w_f16 && out_f16 && w_f16 != out_f16, ASTTS_ERR_INVALID, "astts_op_tfm_pack_frag: null / aliased pointer");
    ASTTS_REQUIRE(k >= 16 && k % 16 == 0 && rows >= 32 && rows % 32 == 0, ASTTS_ERR_UNSUPPORTED,
                  "astts_op_tfm_pack_frag: rows=%d k=%d (rows must be a multiple of 32, k of 16)", rows, k);
    hipLaunchKernelGGL(tfm_pack_frag, dim3(256), dim3(256), 0, (hipStream_t)stream, (const _Float16*)w_f16, (_Float16*)out_f16, rows, k);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

/* 1 when astts_op_tfm_ffn_fused serves this shape (channels 256, hidden a multiple of 256 up to 4096), else 0 */
int astts_op_tfm_ffn_fused_supported(int32_t c, int32_t hidden) {
    return c == TF_C && hidden >= 256 && hidden % 256 == 0 && hidden <= 4096 ? 1 : 0;
}

int astts_op_tfm_ffn_fused(const float* x, const void* w1_frag_f16, const float* b1, const void* w2_frag_f16, const float* b2, float* out,
                           int64_t m, int32_t c, int32_t hidden, float eps, const void* attn_f16, const void* wo_frag_f16,
                           const float* bo, int32_t k0, astts_stream_t stream) {
    return astts_op_tfm_ffn_fused_pf(x, w1_frag_f16, b1, w2_frag_f16, b2, out, m, c, hidden, eps, attn_f16, wo_frag_f16, bo, k0, nullptr, 0,
                                     stream);
}

int astts_op_tfm_ffn_fused_pf(const float* x, const void* w1_frag_f16, const float* b1, const void* w2_frag_f16, const float* b2, float* out,
                              int64_t m, int32_t c, int32_t hidden, float eps, const void* attn_f16, const void* wo_frag_f16,
                              const float* bo, int32_t k0, const void* pf_ptr, uint32_t pf_bytes, astts_stream_t stream) {
    ASTTS_REQUIRE(x && w1_frag_f16 && w2_frag_f16 && out, ASTTS_ERR_INVALID, "astts_op_tfm_ffn_fused: null pointer");
    ASTTS_REQUIRE(astts_op_tfm_ffn_fused_supported(c, hidden), ASTTS_ERR_UNSUPPORTED,
                  "astts_op_tfm_ffn_fused: c=%d hidden=%d (channels 256, hidden a multiple of 256 <= 4096)", c, hidden);
    ASTTS_REQUIRE(m >= 1 && m <= ((int64_t)1 << 31) * 32 - 32 &&
                      (((uintptr_t)x | (uintptr_t)w1_frag_f16 | (uintptr_t)w2_frag_f16 | (uintptr_t)out) & 15) == 0,
                  ASTTS_ERR_INVALID, "astts_op_tfm_ffn_fused: m out of range or operands not 16-byte aligned");
    const bool wo = attn_f16 != nullptr;
    if (wo) {
        ASTTS_REQUIRE(wo_frag_f16 && (((uintptr_t)attn_f16 | (uintptr_t)wo_frag_f16 | (uintptr_t)bo) & 15) == 0, ASTTS_ERR_INVALID,
                      "astts_op_tfm_ffn_fused: attention rows without a projection weight, or operands not 16-byte aligned");
        ASTTS_REQUIRE(k0 == 256 || k0 == 512, ASTTS_ERR_UNSUPPORTED, "astts_op_tfm_ffn_fused: k0=%d (256 or 512)", k0);
    }
    static std::once_flag attr;     // several host threads launch (PipelinedSynth): nobody may launch before the attribute is set
    std::call_once(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tfm_ffn_fused<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tfm_ffn_fused<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(32 * TF_AS + 2 * 32 * FF_HS) * sizeof(_Float16) + (size_t)hidden * sizeof(float) + (wo ? 32 * 260 * sizeof(float) : 0);
    TfmFfnArgs a{x, (const _Float16*)w1_frag_f16, b1, (const _Float16*)w2_frag_f16, b2, out, (const _Float16*)attn_f16,
                 (const _Float16*)wo_frag_f16, bo, m, hidden, k0, eps, nullptr, 0u};
    static const bool pf_on = !(getenv("ASTTS_TFM_PREFETCH") && atoi(getenv("ASTTS_TFM_PREFETCH")) == 0);
    if (pf_ptr && pf_on) {
        a.pf = (const char*)pf_ptr;
        a.pf_bytes = pf_bytes;
    }
    const bool prof = prof_begin(ASTTS_PROF_GEMM_TILE, st, 4.0 * (double)m * TF_C * hidden + (wo ? 2.0 * (double)m * TF_C * k0 : 0.0));
    if (wo) hipLaunchKernelGGL(tfm_ffn_fused<true>, dim3((unsigned)((m + 31) / 32)), dim3(512), lds, st, a.x, a.w1, a.attn, a.wo, a.m, a.hidden, a.k0, a.eps, a);
    else hipLaunchKernelGGL(tfm_ffn_fused<false>, dim3((unsigned)((m + 31) / 32)), dim3(512), lds, st, a.x, a.w1, a.attn, a.wo, a.m, a.hidden, a.k0, a.eps, a);
    if (prof) prof_end(ASTTS_PROF_GEMM_TILE, st);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

/* 1 when astts_op_tfm_attn_fused serves this shape (channels 256, head dim 64, t <= 384), else 0 */
int astts_op_tfm_attn_fused_supported(int32_t c, int32_t heads, int32_t t) {
    return c == TF_C && heads >= 1 && heads <= 16 && t >= 1 && t <= TF_MAX_T ? 1 : 0;
}

int astts_op_tfm_attn_fused(const float* x, const void* wqkv_frag_f16, const float* bias, const int32_t* lens, void* out_f16, int32_t b,
                            int32_t heads, int32_t t, int32_t c, float eps, float scale, astts_stream_t stream) {
    return astts_op_tfm_attn_fused_pf(x, wqkv_frag_f16, bias, lens, out_f16, b, heads, t, c, eps, scale, nullptr, nullptr, 0, stream);
}

int astts_op_tfm_attn_fused_pf(const float* x, const void* wqkv_frag_f16, const float* bias, const int32_t* lens, void* out_f16, int32_t b,
                               int32_t heads, int32_t t, int32_t c, float eps, float scale, const void* const* pf_ptrs,
                               const uint32_t* pf_bytes, int32_t n_pf, astts_stream_t stream) {
    ASTTS_REQUIRE(n_pf >= 0 && n_pf <= 3 && (n_pf == 0 || (pf_ptrs && pf_bytes)), ASTTS_ERR_INVALID, "astts_op_tfm_attn_fused_pf: n_pf=%d", n_pf);
    ASTTS_REQUIRE(x && wqkv_frag_f16 && out_f16, ASTTS_ERR_INVALID, "astts_op_tfm_attn_fused: null pointer");
    ASTTS_REQUIRE(astts_op_tfm_attn_fused_supported(c, heads, t), ASTTS_ERR_UNSUPPORTED,
                  "astts_op_tfm_attn_fused: c=%d heads=%d t=%d (channels 256, t <= %d)", c, heads, t, TF_MAX_T);
    ASTTS_REQUIRE(b >= 1 && (((uintptr_t)x | (uintptr_t)wqkv_frag_f16 | (uintptr_t)out_f16) & 15) == 0, ASTTS_ERR_INVALID,
                  "astts_op_tfm_attn_fused: operands must be 16-byte aligned");
    const int nch = (t + 31) / 32, tkp = nch * 32;
    const size_t lds = ((size_t)tkp * TF_KS + (size_t)TF_DH * (tkp + 4) + 2 * 32 * TF_AS + (size_t)TF_QROWS * TF_QS) * sizeof(_Float16);
    static std::once_flag attr;
    std::call_once(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tfm_attn_fused<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tfm_attn_fused<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    TfmAttnArgs a{x, (const _Float16*)wqkv_frag_f16, bias, lens, (_Float16*)out_f16, b, heads, t, eps, scale, getenv("ASTTS_TFM_BALANCE") ? atoi(getenv("ASTTS_TFM_BALANCE")) : 1,
                  {nullptr, nullptr, nullptr}, {0u, 0u, 0u}};
    static const bool pf_on = !(getenv("ASTTS_TFM_PREFETCH") && atoi(getenv("ASTTS_TFM_PREFETCH")) == 0);
    for (int i = 0; i < n_pf && pf_on; ++i) {
        a.pf[i] = (const char*)pf_ptrs[i];
        a.pf_bytes[i] = pf_ptrs[i] ? pf_bytes[i] : 0u;
    }
    hipStream_t st = (hipStream_t)stream;
    // profiled with the attention kind: ALGORITHMIC flops (q, k, v projected once + attention; the second projection of K and V by
    // the other query half's workgroup is this kernel's overhead, not work)
    const double flops = (double)b * heads * (3.0 * 2.0 * t * 64.0 * TF_C + 4.0 * (double)t * t * TF_DH);
    const bool prof = prof_begin(ASTTS_PROF_ATTN_FLASH, st, flops);
    // more than one round of (query half, head, sequence) workgroups on the 256 CUs: one workgroup per (head, sequence) takes both halves
    // (K / V projected once; bit-identical results).  ASTTS_TFM_ATTN_FULL=0 / 1 forces a form (tests, A/B).
    const char* fe = getenv("ASTTS_TFM_ATTN_FULL");       // read per call: the tests switch forms inside one process
    const int full_env = fe ? atoi(fe) : -1;
    const bool full = full_env >= 0 ? full_env != 0 : 2 * heads * b > 256;
    if (full) hipLaunchKernelGGL(tfm_attn_fused<true>, dim3(heads * b), dim3(512), lds, st, a.x, a.w, a.bias, a.lens, a.b, a.heads, a.t, a.eps, a.scale, a);
    else hipLaunchKernelGGL(tfm_attn_fused<false>, dim3(2 * heads * b), dim3(512), lds, st, a.x, a.w, a.bias, a.lens, a.b, a.heads, a.t, a.eps, a.scale, a);
    if (prof) prof_end(ASTTS_PROF_ATTN_FLASH, st);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
