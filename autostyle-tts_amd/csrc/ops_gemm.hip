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
// cin_pad multiple of 32, n_pad multiple of 128), out: fp32 [B*t_out, ldc].
// epilogue: (+bias[n]) -> activation -> *alpha -> *row_scale[m] -> +residual[m, n]
//
// gemm_tile:   block tile (WM*TM*32) x (WN*TN*32) x 32, 4 waves, v_mfma_f32_32x32x16_f16, fp32->fp16
//              conversion while staging through LDS (80-byte padded rows: conflict-free b128 reads),
//              register prefetch of the next K tile under the MFMAs of the current one.
// gemm_skinny: M <= 32 (decode steps, conditioning MLPs): weights streamed straight to VGPRs, the 4
//              waves of a block split K line by line and reduce through LDS -- weight-bandwidth bound.
#include "common.h"

namespace astts {

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_GELU = 3, ACT_MISH = 4, ACT_ELU = 5, ACT_TANH = 6, ACT_LEAKY = 7 };

__device__ __forceinline__ float apply_act(float x, int act, float slope) {
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_MISH: {
            const float sp = (x > 20.0f) ? x : log1pf(__expf(x));
            return x * tanhf(sp);
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
};

static constexpr int BK = 32;
static constexpr int LDS_ROW = 40;  // halfs per staged row (32 + 8 pad = 80 bytes)

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_tile(GemmArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_CHUNKS = BM * 2 / 256;  // 16-float chunks per thread for the X tile
    constexpr int B_CHUNKS = BN * 2 / 256;  // 16-half chunks per thread for the W tile
    static_assert(WM * WN == 4, "4 waves");
    static_assert(A_CHUNKS >= 1 && (B_CHUNKS >= 1 || BN == 64 || BN == 32), "tile too small");
    __shared__ __attribute__((aligned(16))) _Float16 sa[2][BM * LDS_ROW];
    __shared__ __attribute__((aligned(16))) _Float16 sb[2][BN * LDS_ROW];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int ktot = a.taps * a.cin_pad;
    const int nkt = ktot / BK;
    const bool vec_ok = (a.lda & 3) == 0 && ((uintptr_t)a.x & 15) == 0;

    // ---- per-thread staging coordinates
    int a_row[A_CHUNKS], a_seg[A_CHUNKS];
    int64_t a_base[A_CHUNKS];  // row index of (b, t*stride - pad) in X, before the tap offset
    int a_t[A_CHUNKS];         // t*stride - pad
    bool a_live[A_CHUNKS];
#pragma unroll
    for (int c = 0; c < A_CHUNKS; ++c) {
        const int id = tid + c * 256;
        a_row[c] = id >> 1;
        a_seg[c] = (id & 1) * 16;
        const int64_t m = m0 + a_row[c];
        a_live[c] = m < a.m;
        const int64_t b = a_live[c] ? m / a.t_out : 0;
        const int t = a_live[c] ? (int)(m - b * a.t_out) : 0;
        a_t[c] = t * a.stride - a.pad;
        a_base[c] = b * a.t_in;
    }
    constexpr int BCH = (B_CHUNKS >= 1) ? B_CHUNKS : 1;
    const bool b_active = (BN * 2 >= 256) || (tid < BN * 2);
    int b_row[BCH], b_seg[BCH];
#pragma unroll
    for (int c = 0; c < BCH; ++c) {
        const int id = tid + c * 256;
        b_row[c] = id >> 1;
        b_seg[c] = (id & 1) * 16;
    }

    float4 ra[A_CHUNKS][4];
    half8 rb[BCH][2];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int tap = k0 / a.cin_pad;
        const int c0 = k0 - tap * a.cin_pad;
#pragma unroll
        for (int c = 0; c < A_CHUNKS; ++c) {
            const int ts = a_t[c] + tap * a.dil;
            const bool ok = a_live[c] && ts >= 0 && ts < a.t_in;
            const int ch = c0 + a_seg[c];
            const float* src = a.x + (a_base[c] + ts) * (int64_t)a.lda + ch;
            if (ok && vec_ok && ch + 16 <= a.cin) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[c][j] = *reinterpret_cast<const float4*>(src + 4 * j);
            } else {
                float tmp[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) tmp[j] = (ok && ch + j < a.cin) ? src[j] : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[c][j] = make_float4(tmp[4 * j], tmp[4 * j + 1], tmp[4 * j + 2], tmp[4 * j + 3]);
            }
        }
        if (b_active) {
#pragma unroll
            for (int c = 0; c < BCH; ++c) {
                const _Float16* src = a.w + (int64_t)(n0 + b_row[c]) * ktot + k0 + b_seg[c];
                rb[c][0] = *reinterpret_cast<const half8*>(src);
                rb[c][1] = *reinterpret_cast<const half8*>(src + 8);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int c = 0; c < A_CHUNKS; ++c) {
            half8 h0, h1;
            h0[0] = (_Float16)ra[c][0].x; h0[1] = (_Float16)ra[c][0].y; h0[2] = (_Float16)ra[c][0].z; h0[3] = (_Float16)ra[c][0].w;
            h0[4] = (_Float16)ra[c][1].x; h0[5] = (_Float16)ra[c][1].y; h0[6] = (_Float16)ra[c][1].z; h0[7] = (_Float16)ra[c][1].w;
            h1[0] = (_Float16)ra[c][2].x; h1[1] = (_Float16)ra[c][2].y; h1[2] = (_Float16)ra[c][2].z; h1[3] = (_Float16)ra[c][2].w;
            h1[4] = (_Float16)ra[c][3].x; h1[5] = (_Float16)ra[c][3].y; h1[6] = (_Float16)ra[c][3].z; h1[7] = (_Float16)ra[c][3].w;
            _Float16* dst = &sa[buf][a_row[c] * LDS_ROW + a_seg[c]];
            *reinterpret_cast<half8*>(dst) = h0;
            *reinterpret_cast<half8*>(dst + 8) = h1;
        }
        if (b_active) {
#pragma unroll
            for (int c = 0; c < BCH; ++c) {
                _Float16* dst = &sb[buf][b_row[c] * LDS_ROW + b_seg[c]];
                *reinterpret_cast<half8*>(dst) = rb[c][0];
                *reinterpret_cast<half8*>(dst + 8) = rb[c][1];
            }
        }
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
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const half8*>(&sa[buf][((wm * TM + i) * 32 + r) * LDS_ROW + ks * 16 + h * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const half8*>(&sb[buf][((wn * TN + j) * 32 + r) * LDS_ROW + ks * 16 + h * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns column n (coalesced 128-byte rows), 16 rows per accumulator
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + r;
        const bool n_ok = n < a.n;
        const float bias = (a.bias && n_ok) ? a.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n_ok && m < a.m) {
                    float v = apply_act(acc[i][j][e] + bias, a.act, a.slope) * a.alpha;
                    if (a.row_scale) v *= a.row_scale[m];
                    if (a.residual) v += a.residual[m * a.ldr + n];
                    a.out[m * a.ldc + n] = v;
                }
            }
        }
    }
}

// M <= 32, no conv addressing (taps == 1, stride 1): out[m, n] for a 32-column slice per block.
__global__ __launch_bounds__(256) void gemm_skinny(GemmArgs a) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int ktot = a.cin_pad;
    const int lines = ktot >> 6;  // 64-element K lines (cin_pad is a multiple of 64 for this kernel)
    float16v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int mrow = r < a.m ? r : (int)a.m - 1;
    const float* xrow = a.x + (int64_t)mrow * a.lda + h * 32;
    const _Float16* wrow = a.w + (int64_t)(n0 + r) * ktot + h * 32;
    for (int line = wid; line < lines; line += 4) {
        const int k0 = line * 64;
        half8 fb[4], fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const half8*>(wrow + k0 + i * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = k0 + h * 32 + i * 8;
            float t[8];
            if (ch + 8 <= a.cin && (a.lda & 3) == 0) {
                const float4 v0 = *reinterpret_cast<const float4*>(xrow + k0 + i * 8);
                const float4 v1 = *reinterpret_cast<const float4*>(xrow + k0 + i * 8 + 4);
                t[0] = v0.x; t[1] = v0.y; t[2] = v0.z; t[3] = v0.w; t[4] = v1.x; t[5] = v1.y; t[6] = v1.z; t[7] = v1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = (ch + j < a.cin) ? xrow[k0 + i * 8 + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) fa[i][j] = (_Float16)t[j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[i], acc, 0, 0, 0);
    }
    if (wid > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wid - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wid == 0) {
        const int n = n0 + r;
        const bool n_ok = n < a.n;
        const float bias = (a.bias && n_ok) ? a.bias[n] : 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
            float v = acc[e] + red[0][e][lane] + red[1][e][lane] + red[2][e][lane];
            if (n_ok && m < a.m) {
                v = apply_act(v + bias, a.act, a.slope) * a.alpha;
                if (a.row_scale) v *= a.row_scale[m];
                if (a.residual) v += a.residual[(int64_t)m * a.ldr + n];
                a.out[(int64_t)m * a.ldc + n] = v;
            }
        }
    }
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

}  // namespace astts

using namespace astts;

extern "C" {

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
    ASTTS_REQUIRE(x && w_f16 && out, ASTTS_ERR_INVALID, "astts_op_gemm: null pointer");
    ASTTS_REQUIRE(m >= 1 && n >= 1 && cin >= 1 && taps >= 1, ASTTS_ERR_INVALID,
                  "astts_op_gemm: bad shape m=%lld n=%d cin=%d taps=%d", (long long)m, n, cin, taps);
    ASTTS_REQUIRE(cin_pad >= cin && cin_pad % 64 == 0, ASTTS_ERR_INVALID,
                  "astts_op_gemm: cin_pad=%d must be a multiple of 64 and >= cin=%d", cin_pad, cin);
    ASTTS_REQUIRE(t_in >= 1 && t_out >= 1 && m % t_out == 0 && stride >= 1 && dil >= 1, ASTTS_ERR_INVALID,
                  "astts_op_gemm: bad conv geometry t_in=%d t_out=%d stride=%d dil=%d", t_in, t_out, stride, dil);
    ASTTS_REQUIRE(act >= ACT_NONE && act <= ACT_LEAKY, ASTTS_ERR_INVALID, "astts_op_gemm: act=%d", act);
    GemmArgs a{x, (const _Float16*)w_f16, bias, residual, row_scale, out, m, n, cin, cin_pad, taps,
               lda, ldc, ldr, t_in, t_out, stride, dil, pad, act, alpha, slope};
    hipStream_t st = (hipStream_t)stream;
    const bool plain = taps == 1 && stride == 1 && pad == 0 && t_in == t_out;
    if (m <= 32 && plain) {
        hipLaunchKernelGGL(gemm_skinny, dim3((n + 31) / 32), dim3(256), 0, st, a);
    } else if (n <= 32) {
        hipLaunchKernelGGL((gemm_tile<4, 1, 1, 1>), dim3((unsigned)cdiv(m, 128), (n + 31) / 32), dim3(256), 0, st, a);
    } else if (n <= 64 || (n % 128 != 0 && n % 128 <= 64 && n < 256)) {
        hipLaunchKernelGGL((gemm_tile<2, 2, 2, 1>), dim3((unsigned)cdiv(m, 128), (n + 63) / 64), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((gemm_tile<2, 2, 2, 2>), dim3((unsigned)cdiv(m, 128), (n + 127) / 128), dim3(256), 0, st, a);
    }
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
