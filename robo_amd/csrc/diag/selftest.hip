// Hardware self-checks used by the tests and by bench.py's roofline block:
//   * one v_mfma_f64_16x16x4_f64 against a scalar product with ASYMMETRIC operands (a
//     transposed fragment map would pass a symmetric test);
//   * an MFMA-only loop to measure the fp64 matrix-pipe ceiling of the board the bench runs
//     on (the local microarchitecture guide lists no fp64 MFMA peak; the datasheet figure is
//     78.6 TFLOP/s).
#include "../common.h"
#include "../mcmc_dev.h"

namespace robo {

__global__ __launch_bounds__(64) void mfma_layout_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                         double* __restrict__ C) {
    // A: 16x4 row-major, B: 4x16 row-major, C: 16x16 row-major
    const int l = threadIdx.x;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    acc = mfma_f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

__global__ __launch_bounds__(256) void mfma_bench_kernel(double* __restrict__ sink, int iters, double seed,
                                                         long long* __restrict__ clocks) {
    const int l = threadIdx.x & 63;
    const long long t0 = clock64(), w0 = wall_clock64();
    const double a = seed + 1e-3 * l, b = seed - 1e-3 * l;
    v4d c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
    v4d c4 = {0, 0, 0, 0}, c5 = {1, 1, 1, 1}, c6 = {2, 2, 2, 2}, c7 = {3, 3, 3, 3};
    for (int i = 0; i < iters; ++i) {
        c0 = mfma_f64(a, b, c0);
        c1 = mfma_f64(a, b, c1);
        c2 = mfma_f64(a, b, c2);
        c3 = mfma_f64(a, b, c3);
        c4 = mfma_f64(a, b, c4);
        c5 = mfma_f64(a, b, c5);
        c6 = mfma_f64(a, b, c6);
        c7 = mfma_f64(a, b, c7);
    }
    const v4d s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) sink[threadIdx.x] = s[0];   // keep the chain live
    const long long t1 = clock64(), w1 = wall_clock64();
    if (clocks && blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = t1 - t0;   // shader clock ticks (s_memtime)
        clocks[1] = w1 - w0;   // constant 100 MHz ticks (s_memrealtime)
    }
}

// one accumulator: every MFMA depends on the previous one (the shape of the small products in the
// diagonal-block kernel)
__global__ __launch_bounds__(64) void mfma_chain_kernel(double* __restrict__ sink, int iters, double seed,
                                                        long long* __restrict__ clocks) {
    const int l = threadIdx.x & 63;
    const double a = seed + 1e-3 * l, b = seed - 1e-3 * l;
    v4d c0 = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = mfma_f64(a, b, c0);
        c0 = mfma_f64(a, b, c0);
        c0 = mfma_f64(a, b, c0);
        c0 = mfma_f64(a, b, c0);
    }
    if (c0[0] + c0[1] == 12345.678) sink[threadIdx.x] = c0[0];
    const long long t1 = clock64();
    if (threadIdx.x == 0) clocks[0] = t1 - t0;
}

// The stretch move's arithmetic (mcmc_dev.h: mcmc_stretch_z / mcmc_stretch_q / mcmc_lnpdiff, the functions the chain
// kernels inline) on arrays: tests compare the results with NumPy's bit for bit on the MI355X, and tests/test_isa.py
// reads these kernels' machine code -- stretch_q_probe_kernel must hold v_mul_f64 / v_add_f64 and no v_fma_f64.
__global__ __launch_bounds__(256) void stretch_q_probe_kernel(const double* __restrict__ c, const double* __restrict__ s,
                                                              const double* __restrict__ z, double* __restrict__ q, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) q[i] = mcmc_stretch_q(c[i], s[i], z[i]);
}
__global__ __launch_bounds__(256) void stretch_z_probe_kernel(const double* __restrict__ u, double a, double* __restrict__ z,
                                                              int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) z[i] = mcmc_stretch_z(a, u[i]);
}
// (P - 1) * lz + lp_new - lp_old with lz given (the device's log is not NumPy's: only the arithmetic around it is probed)
__global__ __launch_bounds__(256) void lnpdiff_probe_kernel(const double* __restrict__ lz, const double* __restrict__ lpn,
                                                            const double* __restrict__ lpo, int P, double* __restrict__ out,
                                                            int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = mcmc_lnpdiff(P, lz[i], lpn[i], lpo[i]);
}

int launch_stretch_probe(robo_ctx* ctx, const double* h_c, const double* h_s, const double* h_u, double a, int P, int n,
                         double* h_z, double* h_q, double* h_d) {
    double* d = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    ROBO_HIP_CHECK(hipMalloc(&d, 6 * bytes));
    double *dc = d, *ds = d + n, *du = d + 2 * (size_t)n, *dz = d + 3 * (size_t)n, *dq = d + 4 * (size_t)n, *dd = d + 5 * (size_t)n;
    ROBO_HIP_CHECK(hipMemcpyAsync(dc, h_c, bytes, hipMemcpyHostToDevice, ctx->stream));
    ROBO_HIP_CHECK(hipMemcpyAsync(ds, h_s, bytes, hipMemcpyHostToDevice, ctx->stream));
    ROBO_HIP_CHECK(hipMemcpyAsync(du, h_u, bytes, hipMemcpyHostToDevice, ctx->stream));
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(stretch_z_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)du, a, dz, n);
    hipLaunchKernelGGL(stretch_q_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)dc, (const double*)ds,
                       (const double*)dz, dq, n);
    // the accept statistic on (u as "log z", c, s)
    hipLaunchKernelGGL(lnpdiff_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)du, (const double*)dc,
                       (const double*)ds, P, dd, n);
    ROBO_LAUNCH_CHECK();
    ROBO_HIP_CHECK(hipMemcpyAsync(h_z, dz, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ROBO_HIP_CHECK(hipMemcpyAsync(h_q, dq, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ROBO_HIP_CHECK(hipMemcpyAsync(h_d, dd, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ROBO_HIP_CHECK(hipFree(d));
    return ROBO_OK;
}

int launch_mfma_selftest(robo_ctx* ctx, double* out_err) {
    double hA[64], hB[64], hC[256];
    for (int i = 0; i < 64; ++i) {
        hA[i] = 0.25 * i + 1.0 + 0.01 * (i % 7);
        hB[i] = 3.0 - 0.125 * i + 0.02 * (i % 5);
    }
    double* d = nullptr;
    ROBO_HIP_CHECK(hipMalloc(&d, (64 + 64 + 256) * sizeof(double)));
    ROBO_HIP_CHECK(hipMemcpyAsync(d, hA, sizeof(hA), hipMemcpyHostToDevice, ctx->stream));
    ROBO_HIP_CHECK(hipMemcpyAsync(d + 64, hB, sizeof(hB), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)d,
                       (const double*)(d + 64), d + 128);
    ROBO_HIP_CHECK(hipMemcpyAsync(hC, d + 128, sizeof(hC), hipMemcpyDeviceToHost, ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ROBO_HIP_CHECK(hipFree(d));
    double err = 0.0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
            const double e = s - hC[i * 16 + j];
            if ((e < 0 ? -e : e) > err) err = e < 0 ? -e : e;
        }
    *out_err = err;
    return ROBO_OK;
}

int launch_mfma_microbench(robo_ctx* ctx, int iters, double* out_tflops, double* out_cycles_per_mfma,
                           double* out_shader_mhz, double* out_chain) {
    const int blocks = 4096;
    double* sink = nullptr;
    ROBO_HIP_CHECK(hipMalloc(&sink, (256 + 4) * sizeof(double)));
    long long* clocks = reinterpret_cast<long long*>(sink + 256);
    hipEvent_t e0, e1;
    ROBO_HIP_CHECK(hipEventCreate(&e0));
    ROBO_HIP_CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_bench_kernel, dim3(blocks), dim3(256), 0, ctx->stream, sink, iters / 4 + 1, 0.5,
                       (long long*)nullptr);   // warm-up
    ROBO_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(mfma_bench_kernel, dim3(blocks), dim3(256), 0, ctx->stream, sink, iters, 0.5, clocks);
    ROBO_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    ROBO_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    ROBO_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    ROBO_HIP_CHECK(hipEventDestroy(e0));
    ROBO_HIP_CHECK(hipEventDestroy(e1));
    long long hclk[2] = {0, 0};
    ROBO_HIP_CHECK(hipMemcpy(hclk, clocks, sizeof(hclk), hipMemcpyDeviceToHost));
    // shader clock under full-chip MFMA load (block 0, wave 0 of the timed launch)
    if (out_shader_mhz) *out_shader_mhz = hclk[1] > 0 ? (double)hclk[0] / ((double)hclk[1] / 100.0) : 0.0;
    // issue interval: ONE wave alone on the chip, 8 independent accumulators
    hipLaunchKernelGGL(mfma_bench_kernel, dim3(1), dim3(64), 0, ctx->stream, sink, iters, 0.5, clocks);
    ROBO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ROBO_HIP_CHECK(hipMemcpy(hclk, clocks, sizeof(hclk), hipMemcpyDeviceToHost));
    if (out_cycles_per_mfma) *out_cycles_per_mfma = (double)hclk[0] / ((double)iters * 8.0);
    hipLaunchKernelGGL(mfma_chain_kernel, dim3(1), dim3(64), 0, ctx->stream, sink, iters, 0.5, clocks);
    ROBO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ROBO_HIP_CHECK(hipMemcpy(hclk, clocks, sizeof(hclk), hipMemcpyDeviceToHost));
    ROBO_HIP_CHECK(hipFree(sink));
    if (out_chain) *out_chain = (double)hclk[0] / ((double)iters * 4.0);
    const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2048.0;
    *out_tflops = flops / ((double)ms * 1e-3) / 1e12;
    return ROBO_OK;
}

}  // namespace robo
