// ops_frontend.hip -- the small operators of the LEARNED half of the frontend (gfx950): the CAM++ speaker network and the
// speech tokenizer's quantiser input (astts/frontend_nets.py).
//
// The reference runs both networks inside CosyVoice(model_dir) through onnxruntime on every load_wav'd prompt
// (/root/reference/tts_with_rag.py:159,179-195, /root/reference/tts_with_style_and_timbre.py:83-93): campplus.onnx on the
// 80-bin Kaldi fbank, speech_tokenizer_v1.onnx on the 128-bin Whisper log-mel.  Their contractions run on the GEMM / conv /
// attention operators of this library; what is here is the glue those networks need and the synthesis path does not:
//   affine_act       eval-mode BatchNorm (per-channel scale / shift) + ReLU in FRONT of a convolution (the D-TDNN layers are
//                    pre-activation: the same features are normalised differently by every layer, so the norm cannot be folded
//                    into the producer), fp16 out = the next GEMM's operand
//   freq_unfold      the 3 x 3 convolutions of the FCM head as 3-tap convolutions over time: the three frequency rows of a
//                    window side by side in the channel dimension ([B, F, T, C] -> [B, F_out, T, 3 C], stride along F)
//   ftc_to_tfc       [B, F, T, C] -> [B, T, F C]: the head's output as TDNN input features
//   cam_context      context-aware mask input: mean over time + mean over the frame's 100-frame segment
//   cam_gate         y * sigmoid(m[segment]) written into the dense block's concatenation buffer
//   stats_pool       mean and unbiased standard deviation over time
//   l2_normalize     rows to unit length (the tokenizer's quantiser works on normalised encoder frames)
// All are HBM streams over a few hundred frames: one pass, coalesced rows, fixed summation orders (results do not depend on
// the launch geometry).
#include "common.h"

namespace astts {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// y[r, c] = act(x[r, c] * scale[c] + shift[c]); scale / shift may be null (identity); act: 0 none, 1 relu
template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void affine_act_k(const InT* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, OutT* __restrict__ y, int64_t ldy, int64_t rows, int c,
                                                    int act) {
    const int64_t total = rows * (int64_t)c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const int k = (int)(i - r * c);
        float v = (float)x[r * ldx + k];
        if (scale) v = fmaf(v, scale[k], shift[k]);
        if (act == 1) v = fmaxf(v, 0.0f);
        y[r * ldy + k] = (OutT)v;
    }
}

// y[b, fo, t, kf * C + c] = x[b, fo * sf + kf - (nkf - 1) / 2, t, c]  (0 outside [0, F_in))
template <typename InT>
__global__ __launch_bounds__(256) void freq_unfold_k(const InT* __restrict__ x, _Float16* __restrict__ y, int b, int f_in, int t, int c,
                                                     int f_out, int sf, int nkf) {
    const int64_t total = (int64_t)b * f_out * t * nkf * c;
    const int pad = (nkf - 1) / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int cc = (int)(r % c); r /= c;
        const int kf = (int)(r % nkf); r /= nkf;
        const int tt = (int)(r % t); r /= t;
        const int fo = (int)(r % f_out);
        const int bb = (int)(r / f_out);
        const int fi = fo * sf + kf - pad;
        float v = 0.0f;
        if (fi >= 0 && fi < f_in) v = (float)x[(((int64_t)bb * f_in + fi) * t + tt) * c + cc];
        y[i] = (_Float16)v;
    }
}

__global__ __launch_bounds__(256) void ftc_to_tfc_k(const float* __restrict__ x, float* __restrict__ y, int b, int f, int t, int c) {
    const int64_t total = (int64_t)b * f * t * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int cc = (int)(r % c); r /= c;
        const int ff = (int)(r % f); r /= f;
        const int tt = (int)(r % t);
        const int bb = (int)(r / t);
        y[i] = x[(((int64_t)bb * f + ff) * t + tt) * c + cc];
    }
}

// ctx[b, s, c] = mean_t h[b, t, c] + mean_{t in segment s} h[b, t, c]   (segments of seg_len frames, the last one shorter:
// F.avg_pool1d(ceil_mode=True) divides by the frames that exist).  One block per batch row; thread = (row group, channel);
// a segment's rows are summed per row group, the groups in order -- the same sums for every launch.
template <typename InT>
__global__ __launch_bounds__(1024) void cam_context_k(const InT* __restrict__ h, int64_t ldh, float* __restrict__ ctx, int t, int c,
                                                      int seg_len, int nseg) {
    extern __shared__ float part[];          // [groups][c]
    const int b = blockIdx.x;
    const int groups = blockDim.x / c;
    const int g = threadIdx.x / c, k = threadIdx.x - g * c;
    const InT* hb = h + (int64_t)b * t * ldh;
    float total = 0.0f;                      // (group 0's threads)
    for (int s = 0; s < nseg; ++s) {
        const int t0 = s * seg_len, t1 = min(t, t0 + seg_len);
        float acc = 0.0f;
        if (g < groups)
            for (int tt = t0 + g; tt < t1; tt += groups) acc += (float)hb[(int64_t)tt * ldh + k];
        if (g < groups) part[g * c + k] = acc;
        __syncthreads();
        if (g == 0) {
            float sum = 0.0f;
            for (int gg = 0; gg < groups; ++gg) sum += part[gg * c + k];
            total += sum;
            ctx[((int64_t)b * nseg + s) * c + k] = sum / (float)(t1 - t0);
        }
        __syncthreads();
    }
    if (g == 0) {
        const float mean = total / (float)t;
        for (int s = 0; s < nseg; ++s) ctx[((int64_t)b * nseg + s) * c + k] += mean;
    }
}

// out[b, t, col0 + k] = y[b, t, k] * sigmoid(m[b, t / seg_len, k])
__global__ __launch_bounds__(256) void cam_gate_k(const float* __restrict__ y, const float* __restrict__ m, float* __restrict__ out,
                                                  int64_t ldo, int b, int t, int c, int seg_len, int nseg) {
    const int64_t total = (int64_t)b * t * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int k = (int)(r % c); r /= c;
        const int tt = (int)(r % t);
        const int bb = (int)(r / t);
        const float z = m[((int64_t)bb * nseg + tt / seg_len) * c + k];
        const float sg = 1.0f / (1.0f + __expf(-z));
        out[((int64_t)bb * t + tt) * ldo + k] = y[i] * sg;
    }
}

// out[b, k] = mean_t x[b, t, k], out[b, c + k] = unbiased std; block = (batch row, 64 channels), 4 row groups; two passes
// (mean first, then the squared deviations) so that long, nearly constant channels keep their digits
__global__ __launch_bounds__(256) void stats_pool_k(const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int t, int c) {
    __shared__ float part[4][64];
    __shared__ float s_mean[64];
    const int b = blockIdx.y, k = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const bool live = k < c;
    const float* xb = x + (int64_t)b * t * ldx;
    float acc = 0.0f;
    if (live)
        for (int tt = g; tt < t; tt += 4) acc += xb[(int64_t)tt * ldx + k];
    part[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0) s_mean[threadIdx.x] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) / (float)t;
    __syncthreads();
    const float mean = s_mean[threadIdx.x & 63];
    acc = 0.0f;
    if (live)
        for (int tt = g; tt < t; tt += 4) {
            const float d = xb[(int64_t)tt * ldx + k] - mean;
            acc = fmaf(d, d, acc);
        }
    __syncthreads();
    part[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && live) {
        const float ss = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        out[(int64_t)b * 2 * c + k] = mean;
        out[(int64_t)b * 2 * c + c + k] = t > 1 ? sqrtf(ss / (float)(t - 1)) : 0.0f;
    }
}

// y[r, :] = x[r, :] / max(|x[r, :]|, eps): one wave per row
__global__ __launch_bounds__(256) void l2_normalize_k(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int c, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * c;
    float s = 0.0f;
    for (int k = lane; k < c; k += 64) s = fmaf(xr[k], xr[k], s);
    const float inv = 1.0f / fmaxf(sqrtf(wsum(s)), eps);
    for (int k = lane; k < c; k += 64) y[row * c + k] = xr[k] * inv;
}

// x[b, t, k] -= mean_t x[b, t, k] in place (the speaker network's input: Kaldi fbank minus its mean over time); block = (64 channels,
// batch row), 4 row groups, sums in a fixed order
__global__ __launch_bounds__(256) void sub_time_mean_k(float* __restrict__ x, int t, int c) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, k = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const bool live = k < c;
    float* xb = x + (int64_t)b * t * c;
    float acc = 0.0f;
    if (live)
        for (int tt = g; tt < t; tt += 4) acc += xb[(int64_t)tt * c + k];
    part[g][threadIdx.x & 63] = acc;
    __syncthreads();
    const int l = threadIdx.x & 63;
    const float mean = ((part[0][l] + part[1][l]) + (part[2][l] + part[3][l])) / (float)t;
    if (live)
        for (int tt = g; tt < t; tt += 4) xb[(int64_t)tt * c + k] -= mean;
}

static inline unsigned grid_for(int64_t total, int threads = 256) {
    int64_t g = cdiv(total, threads);
    return (unsigned)(g < 1 ? 1 : (g > 65535 * 4 ? 65535 * 4 : g));
}

}  // namespace astts

using namespace astts;

extern "C" {

int astts_op_affine_act(const void* x, int32_t x_f16, int64_t ldx, const float* scale, const float* shift, void* y, int32_t y_f16,
                        int64_t ldy, int64_t rows, int32_t c, int32_t act, astts_stream_t stream) {
    ASTTS_REQUIRE(x && y && rows >= 0 && c >= 1 && ldx >= c && ldy >= c, ASTTS_ERR_INVALID, "astts_op_affine_act: bad arguments");
    ASTTS_REQUIRE((scale == nullptr) == (shift == nullptr), ASTTS_ERR_INVALID, "astts_op_affine_act: scale and shift come together");
    ASTTS_REQUIRE(act == 0 || act == 1, ASTTS_ERR_INVALID, "astts_op_affine_act: act %d (0 none, 1 relu)", act);
    if (rows == 0) return ASTTS_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned g = grid_for(rows * c);
#define AA(IN, OUT) hipLaunchKernelGGL((affine_act_k<IN, OUT>), dim3(g), dim3(256), 0, st, (const IN*)x, ldx, scale, shift, (OUT*)y, ldy, rows, c, act)
    if (x_f16 && y_f16) AA(_Float16, _Float16);
    else if (x_f16) AA(_Float16, float);
    else if (y_f16) AA(float, _Float16);
    else AA(float, float);
#undef AA
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_freq_unfold(const void* x, int32_t x_f16, void* y_f16, int32_t b, int32_t f_in, int32_t t, int32_t c, int32_t f_out, int32_t sf,
                         int32_t nkf, astts_stream_t stream) {
    ASTTS_REQUIRE(x && y_f16 && b >= 1 && f_in >= 1 && t >= 1 && c >= 1 && f_out >= 1 && sf >= 1 && (nkf == 1 || nkf == 3), ASTTS_ERR_INVALID,
                  "astts_op_freq_unfold: bad arguments");
    ASTTS_REQUIRE((f_out - 1) * sf - (nkf - 1) / 2 < f_in, ASTTS_ERR_INVALID, "astts_op_freq_unfold: f_out %d beyond the input rows", f_out);
    hipStream_t st = (hipStream_t)stream;
    const unsigned g = grid_for((int64_t)b * f_out * t * nkf * c);
    if (x_f16)
        hipLaunchKernelGGL((freq_unfold_k<_Float16>), dim3(g), dim3(256), 0, st, (const _Float16*)x, (_Float16*)y_f16, b, f_in, t, c, f_out, sf, nkf);
    else
        hipLaunchKernelGGL((freq_unfold_k<float>), dim3(g), dim3(256), 0, st, (const float*)x, (_Float16*)y_f16, b, f_in, t, c, f_out, sf, nkf);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_ftc_to_tfc(const float* x, float* y, int32_t b, int32_t f, int32_t t, int32_t c, astts_stream_t stream) {
    ASTTS_REQUIRE(x && y && x != y && b >= 1 && f >= 1 && t >= 1 && c >= 1, ASTTS_ERR_INVALID, "astts_op_ftc_to_tfc: bad arguments");
    hipLaunchKernelGGL(ftc_to_tfc_k, dim3(grid_for((int64_t)b * f * t * c)), dim3(256), 0, (hipStream_t)stream, x, y, b, f, t, c);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_cam_context(const void* h, int32_t h_f16, int64_t ldh, float* ctx, int32_t b, int32_t t, int32_t c, int32_t seg_len,
                         astts_stream_t stream) {
    ASTTS_REQUIRE(h && ctx && b >= 1 && t >= 1 && c >= 1 && c <= 1024 && seg_len >= 1 && ldh >= c, ASTTS_ERR_INVALID,
                  "astts_op_cam_context: bad arguments");
    const int groups = 1024 / c;
    const int nseg = (int)cdiv(t, seg_len);
    hipStream_t st = (hipStream_t)stream;
    if (h_f16)
        hipLaunchKernelGGL((cam_context_k<_Float16>), dim3(b), dim3(groups * c), sizeof(float) * groups * c, st, (const _Float16*)h, ldh, ctx, t, c,
                           seg_len, nseg);
    else
        hipLaunchKernelGGL((cam_context_k<float>), dim3(b), dim3(groups * c), sizeof(float) * groups * c, st, (const float*)h, ldh, ctx, t, c, seg_len,
                           nseg);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_cam_gate(const float* y, const float* m, float* out, int64_t ldo, int32_t b, int32_t t, int32_t c, int32_t seg_len,
                      astts_stream_t stream) {
    ASTTS_REQUIRE(y && m && out && b >= 1 && t >= 1 && c >= 1 && seg_len >= 1 && ldo >= c, ASTTS_ERR_INVALID, "astts_op_cam_gate: bad arguments");
    hipLaunchKernelGGL(cam_gate_k, dim3(grid_for((int64_t)b * t * c)), dim3(256), 0, (hipStream_t)stream, y, m, out, ldo, b, t, c, seg_len,
                       (int)cdiv(t, seg_len));
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_stats_pool(const float* x, int64_t ldx, float* out, int32_t b, int32_t t, int32_t c, astts_stream_t stream) {
    ASTTS_REQUIRE(x && out && b >= 1 && t >= 1 && c >= 1 && ldx >= c, ASTTS_ERR_INVALID, "astts_op_stats_pool: bad arguments");
    hipLaunchKernelGGL(stats_pool_k, dim3((unsigned)cdiv(c, 64), b), dim3(256), 0, (hipStream_t)stream, x, ldx, out, t, c);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_sub_time_mean(float* x, int32_t b, int32_t t, int32_t c, astts_stream_t stream) {
    ASTTS_REQUIRE(x && b >= 1 && t >= 1 && c >= 1, ASTTS_ERR_INVALID, "astts_op_sub_time_mean: bad arguments");
    hipLaunchKernelGGL(sub_time_mean_k, dim3((unsigned)cdiv(c, 64), b), dim3(256), 0, (hipStream_t)stream, x, t, c);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_l2_normalize(const float* x, float* y, int64_t rows, int32_t c, float eps, astts_stream_t stream) {
    ASTTS_REQUIRE(x && y && rows >= 0 && c >= 1, ASTTS_ERR_INVALID, "astts_op_l2_normalize: bad arguments");
    if (rows == 0) return ASTTS_OK;
    hipLaunchKernelGGL(l2_normalize_k, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, c, eps);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
