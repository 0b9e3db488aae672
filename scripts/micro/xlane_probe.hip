// xlane_probe.hip -- csrc/xlane.h against __shfl_xor: same bits for every offset, and the latency of a 64-lane all-reduce.
//   hipcc -O3 --offload-arch=gfx950 xlane_probe.hip -o xlane_probe
#include "../../autostyle-tts_amd/csrc/xlane.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace astts;

__global__ void check(const float* in, float* out_ref, float* out_new) {
    const float v = in[threadIdx.x];
    float* r = out_ref + threadIdx.x * 8;
    float* n = out_new + threadIdx.x * 8;
    r[0] = __shfl_xor(v, 1, 64); n[0] = lane_xor<1>(v);
    r[1] = __shfl_xor(v, 2, 64); n[1] = lane_xor<2>(v);
    r[2] = __shfl_xor(v, 4, 64); n[2] = lane_xor<4>(v);
    r[3] = __shfl_xor(v, 8, 64); n[3] = lane_xor<8>(v);
    r[4] = __shfl_xor(v, 16, 64); n[4] = lane_xor<16>(v);
    r[5] = __shfl_xor(v, 32, 64); n[5] = lane_xor<32>(v);
    float s = v;
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    r[6] = s; n[6] = wave_sum_desc(v);
    float m = v;
    for (int off = 8; off <= 32; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    r[7] = m; n[7] = xmax<32>(xmax<16>(xmax<8>(v)));
}

template <int MODE>
__global__ void lat(const float* in, float* out, int iters, long long* cyc) {
    float v = in[threadIdx.x];
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { float s = v; for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64); v = s * 0.015625f; }
        else v = wave_sum_desc(v) * 0.015625f;
    }
    const long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float *in, *r, *n; long long* cyc;
    hipMalloc(&in, 512 * 4); hipMalloc(&r, 512 * 8 * 4); hipMalloc(&n, 512 * 8 * 4); hipMalloc(&cyc, 8);
    std::vector<float> h(512);
    srand(5);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(512), 0, 0, in, r, n);
    std::vector<float> hr(512 * 8), hn(512 * 8);
    hipMemcpy(hr.data(), r, hr.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hn.data(), n, hn.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < hr.size(); ++i)
        if (memcmp(&hr[i], &hn[i], 4)) { if (bad < 10) printf("mismatch thread %zu slot %zu: %g vs %g\n", i / 8, i % 8, hr[i], hn[i]); ++bad; }
    printf("bit mismatches: %d of %zu\n", bad, hr.size());
    for (int mode = 0; mode < 2; ++mode) {
        long long c;
        for (int w = 0; w < 2; ++w) {
            if (mode == 0) hipLaunchKernelGGL(lat<0>, dim3(1), dim3(64), 0, 0, in, r, 1000, cyc);
            else hipLaunchKernelGGL(lat<1>, dim3(1), dim3(64), 0, 0, in, r, 1000, cyc);
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%s: %.1f clock64 ticks per 64-lane all-reduce (100 MHz ticks -> %.0f ns)\n", mode ? "dpp / permlane swap" : "__shfl_xor (ds_bpermute)", c / 1000.0, c / 1000.0 * 10.0);
    }
    return bad != 0;
}
