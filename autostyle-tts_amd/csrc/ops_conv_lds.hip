// ops_conv_lds.hip -- the HiFT vocoder's resblock convolutions (the reference's hot loop #4, SURVEY.md a15: HiFTGenerator's
// ResBlocks behind cosyvoice.inference_tts_with_st, tts_with_rag.py:195) as an LDS-staged kernel: C -> C channels (128 or
// 256), 3 / 7 / 11 taps, dilation 1 / 3 / 5, "same" padding, channels-last activations.
//
// Why: as implicit GEMMs (gemm_tile) these convolutions fetch every input row once per tap from L2 (154 us for a 50 GFLOP
// stage-2 convolution: 13 % of the MFMA peak) and the Snake activation in front of each of them is a launch of its own that
// reads and writes the whole 113 MB tensor.  Here a workgroup stages its (BM + halo) x C input tile ONCE -- Snake applied once
// per element on the way in, fp16 in LDS -- and takes the taps as row-shifted fragment reads; the weights (fragment order:
// astts_op_conv_pack_frag) stream from L2 through registers one (tap, 128-channel slice) ahead.  Epilogue: bias, residual,
// and the resblock mean (acc_out = [acc_out +] scale * y) in the same pass.
//
// Workgroup = BM output frames of one sequence x all C output channels: (BM / 64) x (C / 64) waves, wave tile 64 x 64
// (2 x 2 MFMA 32x32x16 tiles: every A fragment read from LDS feeds two MFMAs, half an LDS fragment per MFMA).
//   C = 128: BM = 128, 4 waves, tile 178 x 136 halfs = 48 KB: TWO workgroups per CU (212 VGPRs), one stages while the other
//            computes -- with BM = 256 / 8 waves / one workgroup per CU a stage-2 convolution spent ~70 of its 89-140 us in
//            staging and epilogue with the MFMAs idle;
//   C = 256: BM = 128, 8 waves, tile 178 x 264 halfs = 94 KB (stage 1: 28 MB tensors, MFMA-heavier).
#include "common.h"

#include <algorithm>

namespace astts {

struct ConvLdsArgs {
    const void* x;            // [b][l][c] fp32 or fp16
    const float* alpha;       // [c] Snake parameter of the input activation, or null (no activation)
    const _Float16* w;        // [taps][c / 32][c / 16][64][8] fp16 (astts_op_conv_pack_frag)
    const float* bias;        // [c] or null
    const float* res;         // [b][l][c] fp32 residual or null
    void* y;                  // [b][l][c] fp32 / fp16: conv + bias + res, or null
    float* acc;               // [b][l][c] fp32: acc = (acc_add ? acc : 0) + acc_scale * (conv + bias + res), or null
    int l, taps, dil;
    int x_f16, y_f16, acc_add;
    float acc_scale;
    const int* lens;          // [b] frames of each sequence (ragged batches; null: l).  Frames at or beyond read as zero -- every sequence
                              // convolves as if it were alone with the zero padding behind it -- and are not written.
};

__device__ __forceinline__ float snakef(float x, float al, float inv) {
    const float sn = __sinf(al * x);
    return x + sn * sn * inv;
}

template <int C, int BM>
__global__ __launch_bounds__((BM / 64) * (C / 64) * 64, (BM / 64) * (C / 64) == 4 ? 2 : 1) void conv_lds(ConvLdsArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 cl_smem[];
    constexpr int RS = C + 8;                         // halfs per staged row
    constexpr int WN = C / 64;                        // waves along the output channels
    constexpr int KC = C / 128;                       // 128-channel slices per tap
    constexpr int NT = (BM / 64) * (C / 64) * 64;     // threads: one wave per 64 x 64 output tile
    constexpr int CV = C / 4;                         // float4 columns per row
    constexpr int RPP = NT / CV;                      // rows staged per pass of the workgroup
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int wn = wid % WN, wm = wid / WN;
    const int bb = blockIdx.y;
    const int t0 = blockIdx.x * BM;
    const int halo = a.dil * (a.taps - 1) / 2;
    const int sr = BM + 2 * halo;                     // staged rows: frames t0 - halo .. t0 + BM + halo - 1
    const int64_t seq = (int64_t)bb * a.l;
    const int lim = a.lens ? min(a.lens[bb], a.l) : a.l;      // this sequence's own length
    if (t0 >= lim) return;                            // a tile wholly behind the end of its (short) sequence: nothing to compute

    // ---- this wave's first weight unit (tap 0, slice 0) goes out before the staging
    half8 wf[2][16];                                  // [buffer][n-tile * 8 + k-step]
    const _Float16* wbase = a.w + ((int64_t)(wn * 2) * (C / 16) * 64 + lane) * 8;
    auto load_unit = [&](int u, half8 (&dst)[16]) {   // unit u = tap * KC + slice
        const int tap = u / KC, kc = u - tap * KC;
        const _Float16* p = wbase + ((int64_t)tap * (C / 32) * (C / 16) + kc * 8) * 512;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) dst[nt * 8 + ks] = *reinterpret_cast<const half8*>(p + ((int64_t)nt * (C / 16) + ks) * 512);
    };
    // (the Snake parameters of this thread's columns first: their wait must not cover the weight loads behind them)
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.alpha) a4 = *reinterpret_cast<const float4*>(a.alpha + (tid % CV) * 4);
    load_unit(0, wf[0]);

    // ---- staging: Snake once per element, fp16 rows in LDS; rows outside the sequence are zero (the convolution pads the
    // ACTIVATED signal, and snake(0) = 0 anyway)
    {
        const int col = (tid % CV) * 4, r0 = tid / CV;
        float al[4] = {a4.x, a4.y, a4.z, a4.w}, inv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.alpha) {
#pragma unroll
            for (int j = 0; j < 4; ++j) inv[j] = 1.0f / (al[j] + 1e-9f);
        }
        constexpr int SU = 12;                        // rows per thread in flight (a 178-row tile in two trips)
        for (int rb = r0; rb < sr; rb += RPP * SU) {
            // every row's load is UNCONDITIONAL (frame index clamped into the sequence, zeroed afterwards): a load inside the bounds
            // check is waited for at the end of its block, which made the SU loads "in flight" SU dependent round trips
            float4 v[SU];
            if (a.x_f16) {
                half4 hv[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int t = min(max(t0 - halo + rb + u * RPP, 0), a.l - 1);
                    hv[u] = *reinterpret_cast<const half4*>((const _Float16*)a.x + (seq + t) * C + col);
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) v[u] = make_float4((float)hv[u][0], (float)hv[u][1], (float)hv[u][2], (float)hv[u][3]);
            } else {
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int t = min(max(t0 - halo + rb + u * RPP, 0), a.l - 1);
                    v[u] = *reinterpret_cast<const float4*>((const float*)a.x + (seq + t) * C + col);
                }
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int r = rb + u * RPP;
                const int t = t0 - halo + r;
                if (!(r < sr && t >= 0 && t < lim)) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int r = rb + u * RPP;
                if (r < sr) {
                    float4 o = v[u];
                    if (a.alpha) o = make_float4(snakef(o.x, al[0], inv[0]), snakef(o.y, al[1], inv[1]), snakef(o.z, al[2], inv[2]), snakef(o.w, al[3], inv[3]));
                    half4 h4;
                    h4[0] = (_Float16)o.x; h4[1] = (_Float16)o.y; h4[2] = (_Float16)o.z; h4[3] = (_Float16)o.w;
                    *reinterpret_cast<half4*>(cl_smem + (size_t)r * RS + col) = h4;
                }
            }
        }
    }
    __syncthreads();

    // ---- main loop over (tap, slice) units; the next unit's weights are requested before the current unit's MFMAs
    float16v acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.0f;
    const int nunits = a.taps * KC;
    const _Float16* arow = cl_smem + (size_t)(wm * 64 + c) * RS + 8 * hh;
    auto compute_unit = [&](int u, const half8 (&w)[16]) {
        const int tap = u / KC, kc = u - tap * KC;
        const _Float16* ap = arow + (size_t)(tap * a.dil) * RS + kc * 128;
#pragma unroll
        for (int ks0 = 0; ks0 < 8; ks0 += 4) {
            half8 af[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) af[mt][ks] = *reinterpret_cast<const half8*>(ap + (size_t)(mt * 32) * RS + 16 * (ks0 + ks));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][ks], w[nt * 8 + ks0 + ks], acc[mt][nt], 0, 0, 0);
        }
    };
    for (int u = 0; u < nunits; u += 2) {             // two units per trip: the buffer index stays a compile-time constant
        if (u + 1 < nunits) load_unit(u + 1, wf[1]);
        compute_unit(u, wf[0]);
        if (u + 2 < nunits) load_unit(u + 2, wf[0]);
        if (u + 1 < nunits) compute_unit(u + 1, wf[1]);
    }

    // ---- epilogue: element e of tile (mt, nt) holds frame wm 64 + mt 32 + (e & 3) + 8 (e >> 2) + 4 hh, the lane's channel is
    // wn 64 + nt 32 + c.  Stored from there a wave instruction writes two 64-byte (fp16) pieces -- 37 of the 78 us of a 3-tap
    // convolution; instead each wave transposes its tile through its own 8.7 KB of LDS (the input tile is dead) 32 frames at a
    // time and moves whole rows: 16-byte loads of the residual / accumulator, 16- or 8-byte stores.
    __syncthreads();                                  // every wave is done reading the staged input
    float* tr = reinterpret_cast<float*>(cl_smem) + (size_t)wid * 32 * 68;       // [32 frames][64 + 4] fp32, this wave's
    const int er = lane >> 4, ec = (lane & 15) * 4;   // read-back: 16 lanes per frame, 4 frames per instruction
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + wn * 64 + ec);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int tb = t0 + wm * 64 + mt * 32;        // first frame of this 32-frame slab
        float4 rv[8], pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {                 // residual / accumulator rows requested before the transpose
            const int t = tb + er + 4 * i;
            const int64_t o = (seq + min(t, a.l - 1)) * C + wn * 64 + ec;
            rv[i] = a.res ? *reinterpret_cast<const float4*>(a.res + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            pv[i] = (a.acc && a.acc_add) ? *reinterpret_cast<const float4*>(a.acc + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) tr[((e & 3) + 8 * (e >> 2) + 4 * hh) * 68 + nt * 32 + c] = acc[mt][nt][e];
        __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = tb + er + 4 * i;
            const float4 v4 = *reinterpret_cast<const float4*>(tr + (er + 4 * i) * 68 + ec);
            if (t < lim) {
                const int64_t o = (seq + t) * C + wn * 64 + ec;
                const float4 v = make_float4(v4.x + bias4.x + rv[i].x, v4.y + bias4.y + rv[i].y, v4.z + bias4.z + rv[i].z, v4.w + bias4.w + rv[i].w);
                if (a.y) {
                    if (a.y_f16) {
                        half4 h4;
                        h4[0] = (_Float16)v.x; h4[1] = (_Float16)v.y; h4[2] = (_Float16)v.z; h4[3] = (_Float16)v.w;
                        *reinterpret_cast<half4*>((_Float16*)a.y + o) = h4;
                    } else {
                        *reinterpret_cast<float4*>((float*)a.y + o) = v;
                    }
                }
                if (a.acc)
                    *reinterpret_cast<float4*>(a.acc + o) = make_float4(pv[i].x + a.acc_scale * v.x, pv[i].y + a.acc_scale * v.y,
                                                                        pv[i].z + a.acc_scale * v.z, pv[i].w + a.acc_scale * v.w);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);           // the read-back is complete before the next slab overwrites the buffer
        __builtin_amdgcn_wave_barrier();
    }
}

// row-major fp16 conv weight [rows][taps][k] (PackedWeight image of a Conv1d: astts_op_pack_weight of [cout, taps, cin]) ->
// [taps][rows / 32][k / 16][64 lanes][8]: lane (c, hh) of k-step s of tap j holds W[32 tile + c][j][16 s + 8 hh + i]
__global__ void conv_pack_frag(const _Float16* __restrict__ w, _Float16* __restrict__ out, int rows, int taps, int k) {
    const int64_t total = (int64_t)rows * taps * k;
    const int ksteps = k >> 4;
    const int64_t per_tap = (int64_t)rows * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i / per_tap);
        const int64_t r = i - (int64_t)tap * per_tap;
        const int j = (int)(r & 7), ln = (int)((r >> 3) & 63);
        const int64_t q = r >> 9;
        const int s = (int)(q % ksteps);
        const int64_t tile = q / ksteps;
        out[i] = w[((tile * 32 + (ln & 31)) * taps + tap) * k + 16 * s + 8 * (ln >> 5) + j];
    }
}

}  // namespace astts

using namespace astts;

extern "C" {

int astts_op_conv_pack_frag(const void* w_f16, void* out_f16, int32_t rows, int32_t taps, int32_t k, astts_stream_t stream) {
    ASTTS_REQUIRE(w_f16 && out_f16 && w_f16 != out_f16, ASTTS_ERR_INVALID, "astts_op_conv_pack_frag: null / aliased pointer");
    ASTTS_REQUIRE(k >= 16 && k % 16 == 0 && rows >= 32 && rows % 32 == 0 && taps >= 1 && taps <= 64, ASTTS_ERR_UNSUPPORTED,
                  "astts_op_conv_pack_frag: rows=%d taps=%d k=%d (rows a multiple of 32, k of 16)", rows, taps, k);
    hipLaunchKernelGGL(conv_pack_frag, dim3(256), dim3(256), 0, (hipStream_t)stream, (const _Float16*)w_f16, (_Float16*)out_f16, rows, taps, k);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

/* 1 when astts_op_conv1d_snake serves this shape: c in {128, 256}, odd taps, dil * (taps - 1) / 2 <= 25 */
int astts_op_conv1d_snake_supported(int32_t c, int32_t taps, int32_t dil) {
    return (c == 128 || c == 256) && taps >= 1 && (taps & 1) && dil >= 1 && dil * (taps - 1) / 2 <= 25 ? 1 : 0;
}

int astts_op_conv1d_snake(const void* x, int32_t x_f16, const float* alpha, const void* w_frag_f16, const float* bias, const float* res,
                          void* y, int32_t y_f16, float* acc, float acc_scale, int32_t acc_add, int32_t b, int32_t l, int32_t c,
                          int32_t taps, int32_t dil, astts_stream_t stream) {
    return astts_op_conv1d_snake_lens(x, x_f16, alpha, w_frag_f16, bias, res, y, y_f16, acc, acc_scale, acc_add, b, l, c, taps, dil, nullptr, stream);
}

int astts_op_conv1d_snake_lens(const void* x, int32_t x_f16, const float* alpha, const void* w_frag_f16, const float* bias, const float* res,
                               void* y, int32_t y_f16, float* acc, float acc_scale, int32_t acc_add, int32_t b, int32_t l, int32_t c,
                               int32_t taps, int32_t dil, const int32_t* lens, astts_stream_t stream) {
    ASTTS_REQUIRE(x && w_frag_f16 && (y || acc), ASTTS_ERR_INVALID, "astts_op_conv1d_snake: null pointer");
    ASTTS_REQUIRE(astts_op_conv1d_snake_supported(c, taps, dil), ASTTS_ERR_UNSUPPORTED,
                  "astts_op_conv1d_snake: c=%d taps=%d dil=%d (c 128 or 256, odd taps, halo <= 25)", c, taps, dil);
    ASTTS_REQUIRE(b >= 1 && l >= 1 && (((uintptr_t)x | (uintptr_t)w_frag_f16 | (uintptr_t)alpha | (uintptr_t)bias | (uintptr_t)res |
                                        (uintptr_t)y | (uintptr_t)acc) & 15) == 0, ASTTS_ERR_INVALID,
                  "astts_op_conv1d_snake: bad shape b=%d l=%d or operands not 16-byte aligned", b, l);
    ASTTS_REQUIRE(x != y && x != (const void*)acc, ASTTS_ERR_INVALID, "astts_op_conv1d_snake: the output may not alias the input (halo rows)");
    static std::once_flag attr;     // several host threads launch (PipelinedSynth): nobody may launch before the attribute is set
    std::call_once(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lds<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lds<256, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    ConvLdsArgs a{x, alpha, (const _Float16*)w_frag_f16, bias, res, y, acc, l, taps, dil, x_f16, y_f16, acc_add, acc_scale, lens};
    hipStream_t st = (hipStream_t)stream;
    const int halo = dil * (taps - 1) / 2;
    const bool prof = prof_begin(ASTTS_PROF_GEMM_TILE, st, 2.0 * (double)b * l * c * c * taps);
    if (c == 128) {
        const size_t lds = std::max((size_t)(128 + 2 * halo) * (128 + 8) * sizeof(_Float16), (size_t)4 * 32 * 68 * sizeof(float));   // input tile | 4 waves' transpose buffers
        hipLaunchKernelGGL((conv_lds<128, 128>), dim3((unsigned)((l + 127) / 128), b), dim3(256), lds, st, a);
    } else {
        const size_t lds = std::max((size_t)(128 + 2 * halo) * (256 + 8) * sizeof(_Float16), (size_t)8 * 32 * 68 * sizeof(float));
        hipLaunchKernelGGL((conv_lds<256, 128>), dim3((unsigned)((l + 127) / 128), b), dim3(512), lds, st, a);
    }
    if (prof) prof_end(ASTTS_PROF_GEMM_TILE, st);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
