// Stand-alone timing of the GEMM kernels on the flow decoder's projection shapes, with parts of gemm_ring compiled
// out (-DRING_SKIP_MFMA / -DRING_SKIP_EPI / -DRING_SKIP_LOAD) to see where a block's time goes.
#include "../../autostyle-tts_amd/csrc/runtime.hip"
#include "../../autostyle-tts_amd/csrc/ops_gemm.hip"
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
int main() {
    struct S { int m, k, n, out16, res; } shapes[] = {{5504, 256, 1536, 1, 0}, {5504, 256, 1024, 1, 0}, {5504, 512, 256, 0, 1},
                                                     {5504, 1024, 256, 0, 1}, {11008, 256, 1536, 1, 0}, {11008, 1024, 256, 0, 1},
                                                     {90816, 256, 1536, 1, 0}, {90816, 256, 1024, 1, 0}, {90816, 512, 256, 0, 1}, {90816, 1024, 256, 0, 1}};
    hipStream_t st; CK(hipStreamCreate(&st));
    for (auto s : shapes) {
        _Float16 *x, *w; float *out, *res, *bias;
        CK(hipMalloc(&x, (size_t)s.m * s.k * 2)); CK(hipMalloc(&w, (size_t)((s.n + 127) / 128 * 128) * s.k * 2));
        CK(hipMalloc(&out, (size_t)s.m * s.n * 4)); CK(hipMalloc(&res, (size_t)s.m * s.n * 4)); CK(hipMalloc(&bias, s.n * 4));
        CK(hipMemset(x, 0x11, (size_t)s.m * s.k * 2)); CK(hipMemset(w, 0x11, (size_t)((s.n + 127) / 128 * 128) * s.k * 2));
        CK(hipMemset(res, 0, (size_t)s.m * s.n * 4)); CK(hipMemset(bias, 0, s.n * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 200;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) {
                int rc = astts_op_gemm_ex(x, 1, w, bias, s.res ? res : nullptr, nullptr, out, s.out16, s.m, s.n, s.k, s.k, 1, s.k, s.n, s.res ? s.n : 0,
                                          s.m, s.m, 1, 1, 0, 0, 1.0f, 0.1f, st);
                if (rc) { printf("error %s\n", astts_last_error_string()); return 1; }
            }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("m=%6d k=%5d n=%5d out16=%d res=%d: %7.2f us  %7.1f TFLOP/s\n", s.m, s.k, s.n, s.out16, s.res, ms * 1e3 / iters,
                            2.0 * s.m * s.n * s.k / (ms * 1e3 / iters) / 1e6);
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(out)); CK(hipFree(res)); CK(hipFree(bias));
    }
    return 0;
}
