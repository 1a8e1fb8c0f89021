// C ABI of librobo_hip_diag.so: hardware self-checks and micro-benchmarks (include/robo_hip_diag.h).
// NOT part of the product library: tests, bench.py's roofline block and tools/ load it next to librobo_hip.so
// (it links against it for the handle layouts and the instrumented diagonal-block launcher).
#include <vector>

#include "../common.h"
#include "../../../include/robo_hip_diag.h"

using namespace robo;

extern "C" {

int32_t robo_selftest_mfma_layout(robo_ctx* ctx, double* out_max_err) {
    if (!ctx || !out_max_err) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_mfma_selftest(ctx, out_max_err);
}

int32_t robo_microbench_mfma_f64(robo_ctx* ctx, int32_t iters, double* out_tflops) {
    if (!ctx || !out_tflops || iters < 1) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_mfma_microbench(ctx, iters, out_tflops, nullptr, nullptr, nullptr);
}

int32_t robo_selftest_diag_timeline(robo_gp* g, const double* theta, double* out13) {
    if (!g || !theta || !out13) return ROBO_BAD_ARGUMENT;
    if (!g->has_data) return ROBO_NOT_FITTED;
    {   // the gram matrix at theta into the GP's own buffer (public entry point; the host copy is discarded)
        std::vector<double> tmp((size_t)g->n * g->n);
        const int st = robo_gp_get_gram(g, theta, tmp.data());
        if (st != ROBO_OK) return st;
    }
    long long* d = nullptr;
    ROBO_HIP_CHECK(hipMalloc((void**)&d, 56 * sizeof(long long)));
    ROBO_HIP_CHECK(hipMemsetAsync(d, 0, 56 * sizeof(long long), g->ctx->stream));
    { const int st = launch_diag_timeline(g, d); if (st != ROBO_OK) return st; }
    long long h[56];
    ROBO_HIP_CHECK(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, g->ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
    ROBO_HIP_CHECK(hipFree(d));
    for (int i = 0; i < 13; ++i) out13[i] = (double)(h[i] - h[0]);
    for (int i = 0; i < 4; ++i) out13[13 + i] = (double)(h[16 + i] - h[16]);   // panel kernel
    // per interval s = 0..6 of the pivot wave: C1 + C2 done, through Bb(s), potf2(s+1) done (out has 17 + 21 entries)
    for (int sb = 0; sb < 7; ++sb)
        for (int i = 0; i < 3; ++i) out13[17 + 3 * sb + i] = (double)(h[24 + 4 * sb + i] - h[0]);
    return ROBO_OK;
}

int32_t robo_microbench_gemm_f64(robo_ctx* ctx, int32_t variant, int32_t wgs, int32_t k, int32_t reps,
                                 double* out_tflops) {
    if (!ctx || !out_tflops) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_gemm_microbench(ctx, variant, wgs, k, reps, out_tflops);
}

int32_t robo_microbench_mfma_f64_detail(robo_ctx* ctx, int32_t iters, double* out3) {
    if (!ctx || !out3 || iters < 1) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_mfma_microbench(ctx, iters, out3, out3 + 1, out3 + 2, out3 + 3);
}

int32_t robo_selftest_stretch_move(robo_ctx* ctx, const double* c, const double* s, const double* u, double a, int32_t P,
                                   int32_t n, double* out_z, double* out_q, double* out_lnpdiff) {
    if (!ctx || !c || !s || !u || !out_z || !out_q || !out_lnpdiff || n < 1) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_stretch_probe(ctx, c, s, u, a, P, n, out_z, out_q, out_lnpdiff);
}

int32_t robo_diag_clock_sample_begin(robo_ctx* ctx, int32_t window_us) {
    if (!ctx) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_clock_sampler(ctx, window_us);
}

int32_t robo_diag_clock_sample_end(robo_ctx* ctx, double* out3) {
    if (!ctx || !out3) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    return collect_clock_sampler(out3);
}

}  // extern "C"
