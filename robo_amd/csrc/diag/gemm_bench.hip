// GEMM-core microbenchmarks (measurement infrastructure, not on the product path): the LDS-staged
// NT core of gemm_f64.h against a barrier-free variant that streams MFMA fragments straight from a
// fragment-packed operand layout.  Same shape as one TRSM block-row step: `wgs` workgroups, each
// C(128 x 128) = A_wg(128 x K) * B(128 x K)^T with B shared by all.
//
// What it established (MI355X, r01n):
//  * under an operand-streaming fp64 MFMA load the shader clock sits at 2.0-2.2 GHz, not the 2.4 GHz
//    the 78.6 TFLOP/s datasheet peak assumes (the register-only MFMA loop of selftest.hip holds
//    2.37 GHz): the clock-limited ceiling for this kernel shape is ~66-70 TFLOP/s;
//  * in isolation the packed variant wins (K = 2048: 58.0 -> 64.0, K = 4096: 62.1 -> 64.7 TFLOP/s),
//    but built into the posterior step kernel (packed V panels, packed -L and Linv copies) it was
//    2 % SLOWER than the LDS core in a same-box A/B (17.65 vs 18.0 ms per 65 536 candidates): the
//    scattered 8-byte stores of the packed tile and a less favourable slope at large K ate the
//    gain.  The product path keeps the LDS core; the packed core lives on here only.
#include "../common.h"
#include "../gemm_f64.h"

namespace robo {
// ---- fragment-packed operands: barrier-free streaming straight into MFMA fragments --------------
// The posterior solve streams each operand exactly once per workgroup, so LDS staging buys no reuse
// beyond what L1/L2 already give the four waves -- it only costs ds traffic and a barrier per
// k-tile.  Operands on that path (V panels, -L block rows, Linv blocks) are therefore kept in a
// layout where one wave-wide 16-byte load IS a pair of MFMA fragments:
//   element (r, k) of a (rows x K) matrix  ->  ((r >> 4) * (K / 8) + (k >> 3)) * 128
//                                              + 2 * ((r & 15) + 16 * (k & 3)) + ((k >> 2) & 1)
// i.e. 1 KB groups of (16 rows x 8 k); lane l of a wave reads double2 number l of a group:
// .x is its A (or B^T) entry for k-block 2g, .y for k-block 2g + 1 (fragment maps above).
__host__ __device__ __forceinline__ size_t pk_index(int r, int k, int K) {
    return ((size_t)(r >> 4) * (size_t)(K >> 3) + (size_t)(k >> 3)) * 128 + 2 * ((r & 15) + 16 * (k & 3)) +
           ((k >> 2) & 1);
}

struct Frag8 {
    double2 a0, a1, a2, a3, b0, b1, b2, b3;
};

// a / b: this lane's double2 in group 0 of the wave's first 16-row block; sa / sb: double2 stride
// between consecutive 16-row blocks ((K / 8) * 64); g: k-group
__device__ __forceinline__ Frag8 frag_load(const double2* __restrict__ a, size_t sa, const double2* __restrict__ b,
                                           size_t sb, int g) {
    Frag8 f;
    const size_t o = (size_t)g * 64;
    f.a0 = a[o];
    f.a1 = a[o + sa];
    f.a2 = a[o + 2 * sa];
    f.a3 = a[o + 3 * sa];
    f.b0 = b[o];
    f.b1 = b[o + sb];
    f.b2 = b[o + 2 * sb];
    f.b3 = b[o + 3 * sb];
    return f;
}

__device__ __forceinline__ void frag_mfma(const Frag8 f, Acc& acc) {
    const double ax[4] = {f.a0.x, f.a1.x, f.a2.x, f.a3.x}, bx[4] = {f.b0.x, f.b1.x, f.b2.x, f.b3.x};
    const double ay[4] = {f.a0.y, f.a1.y, f.a2.y, f.a3.y}, by[4] = {f.b0.y, f.b1.y, f.b2.y, f.b3.y};
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc.t[tm][tn] = mfma_f64(ax[tm], bx[tn], acc.t[tm][tn]);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc.t[tm][tn] = mfma_f64(ay[tm], by[tn], acc.t[tm][tn]);
    __builtin_amdgcn_s_setprio(0);
}

// acc += A[128 rows, ka0 : ka0 + kn] * B[128 rows, kb0 : kb0 + kn]^T on packed operands.
// Ap / Bp: packed 128-row panels with Ka / Kb columns; ka0, kb0, kn multiples of 16.  No LDS, no
// barriers: every wave streams its own 64 x kn (A) and 64 x kn (B) fragments, two k-groups in
// flight (the compiler barriers pin the prefetch ahead of the MFMA cluster -- without them the
// loads are sunk next to their uses; the prefetch index is clamped rather than branched around
// because a conditional load makes the waitcnt pass drain the queue at the join).
__device__ __forceinline__ void gemm_pk(const double* __restrict__ Ap, int Ka, int ka0, const double* __restrict__ Bp,
                                        int Kb, int kb0, int kn, Acc& acc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wy = wave >> 1, wx = wave & 1;
    const int ng = kn >> 3;
    if (ng <= 0) return;
    const size_t sa = (size_t)(Ka >> 3) * 64, sb = (size_t)(Kb >> 3) * 64;
    const double2* a = reinterpret_cast<const double2*>(Ap) + (size_t)(wy * 4) * sa + (size_t)(ka0 >> 3) * 64 + lane;
    const double2* b = reinterpret_cast<const double2*>(Bp) + (size_t)(wx * 4) * sb + (size_t)(kb0 >> 3) * 64 + lane;
    Frag8 f0 = frag_load(a, sa, b, sb, 0), f1;
    for (int g = 0; g < ng; g += 2) {
        f1 = frag_load(a, sa, b, sb, g + 1);
        asm volatile("" ::: "memory");
        frag_mfma(f0, acc);
        asm volatile("" ::: "memory");
        f0 = frag_load(a, sa, b, sb, g + 2 < ng ? g + 2 : ng - 1);
        asm volatile("" ::: "memory");
        frag_mfma(f1, acc);
        asm volatile("" ::: "memory");
    }
}


__global__ void fill_kernel(double* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        p[i] = ((double)(h & 0xffffff) / 16777216.0 - 0.5) * 1e-2;
    }
}

__global__ __launch_bounds__(256, 2) void gemm_lds_bench_kernel(const double* __restrict__ A,
                                                                const double* __restrict__ B, int K,
                                                                double* __restrict__ C, long long* __restrict__ clk) {
    const long long t0 = clock64(), w0 = wall_clock64();
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    Acc acc;
    acc_zero(acc);
    gemm_nt<4, false>(A + (size_t)blockIdx.x * NB * K, K, B, K, 0, K, acc, smem);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[((size_t)blockIdx.x * NB + acc_row(tm, r)) * NB + acc_col(tn)] = acc.t[tm][tn][r];
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) {
        clk[0] = clock64() - t0;
        clk[1] = wall_clock64() - w0;
    }
}

__global__ __launch_bounds__(256, 2) void gemm_direct_bench_kernel(const double* __restrict__ Ap,
                                                                   const double* __restrict__ Bp, int K,
                                                                   double* __restrict__ C, long long* __restrict__ clk) {
    const long long t0 = clock64(), w0 = wall_clock64();
    Acc acc;
    acc_zero(acc);
    gemm_pk(Ap + (size_t)blockIdx.x * NB * K, K, 0, Bp, K, 0, K, acc);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[((size_t)blockIdx.x * NB + acc_row(tm, r)) * NB + acc_col(tn)] = acc.t[tm][tn][r];
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) {
        clk[0] = clock64() - t0;
        clk[1] = wall_clock64() - w0;
    }
}

int launch_gemm_microbench(robo_ctx* ctx, int variant, int wgs, int K, int reps, double* out_tflops) {
    if (K % 16 || wgs <= 0 || reps <= 0) return ROBO_BAD_ARGUMENT;
    double *A = nullptr, *B = nullptr, *C = nullptr;
    const size_t na = (size_t)wgs * NB * K, nb = (size_t)NB * K, nc = (size_t)wgs * NB * NB;
    ROBO_HIP_CHECK(hipMalloc(&A, na * 8));
    ROBO_HIP_CHECK(hipMalloc(&B, nb * 8));
    ROBO_HIP_CHECK(hipMalloc(&C, nc * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, ctx->stream, A, na, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, ctx->stream, B, nb, 2u);
    long long* clk = nullptr;
    ROBO_HIP_CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    ROBO_HIP_CHECK(hipEventCreate(&e0));
    ROBO_HIP_CHECK(hipEventCreate(&e1));
    for (int it = 0; it <= reps; ++it) {
        if (it == 1) ROBO_HIP_CHECK(hipEventRecord(e0, ctx->stream));
        if (variant == 0)
            hipLaunchKernelGGL(gemm_lds_bench_kernel, dim3(wgs), dim3(256), 0, ctx->stream, (const double*)A,
                               (const double*)B, K, C, clk);
        else
            hipLaunchKernelGGL(gemm_direct_bench_kernel, dim3(wgs), dim3(256), 0, ctx->stream, (const double*)A,
                               (const double*)B, K, C, clk);
    }
    ROBO_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    ROBO_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    ROBO_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(A);
    hipFree(B);
    hipFree(C);
    long long h[2] = {0, 1};
    ROBO_HIP_CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    hipFree(clk);
    out_tflops[1] = (double)h[0] / ((double)h[1] / 100.0);   // shader MHz seen by workgroup 0 (100 MHz wall clock)
    out_tflops[0] = 2.0 * wgs * (double)NB * NB * K * reps / ((double)ms * 1e-3) / 1e12;
    return ROBO_OK;
}

// ---- shader clock while ANOTHER kernel runs -------------------------------------------------------------------------
// The fp64 peak of SURVEY 8(d) (78.6 TFLOP/s) is 32 flop/clk/SIMD at 2.4 GHz; under a chip-wide MFMA + LDS load the
// part does not hold 2.4 GHz (gemm_lds_bench_kernel above: ~2.05 GHz).  One wave per sampler workgroup sleeps through
// a window of the 100 MHz wall clock on its own stream and reports shader cycles / wall ticks of that window: launched
// right before a posterior step on the library stream, it measures the clock the block-row solve actually runs at.
__global__ __launch_bounds__(64) void clock_sampler_kernel(long long window_ticks, long long* __restrict__ out) {
    const long long w0 = wall_clock64(), t0 = clock64();
    long long w;
    do {
        __builtin_amdgcn_s_sleep(64);
        w = wall_clock64();
    } while (w - w0 < window_ticks);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = clock64() - t0;
        out[2 * blockIdx.x + 1] = w - w0;
    }
}

constexpr int SAMPLER_WGS = 8;
static hipStream_t g_sampler_stream = nullptr;
static long long* g_sampler_out = nullptr;       // pinned host memory, device-visible

int launch_clock_sampler(robo_ctx* ctx, int window_us) {
    if (window_us < 1 || window_us > 2000000) return ROBO_BAD_ARGUMENT;
    if (!g_sampler_stream) ROBO_HIP_CHECK(hipStreamCreateWithFlags(&g_sampler_stream, hipStreamNonBlocking));
    if (!g_sampler_out) ROBO_HIP_CHECK(hipHostMalloc((void**)&g_sampler_out, 2 * SAMPLER_WGS * sizeof(long long)));
    for (int i = 0; i < 2 * SAMPLER_WGS; ++i) g_sampler_out[i] = 0;
    (void)ctx;
    hipLaunchKernelGGL(clock_sampler_kernel, dim3(SAMPLER_WGS), dim3(64), 0, g_sampler_stream,
                       (long long)window_us * 100, g_sampler_out);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// out3 = {mean, min, max} shader MHz over the sampler workgroups
int collect_clock_sampler(double* out3) {
    if (!g_sampler_stream || !g_sampler_out) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipStreamSynchronize(g_sampler_stream));
    double sum = 0.0, lo = 1e30, hi = 0.0;
    int cnt = 0;
    for (int i = 0; i < SAMPLER_WGS; ++i) {
        if (g_sampler_out[2 * i + 1] <= 0) continue;
        const double mhz = (double)g_sampler_out[2 * i] / ((double)g_sampler_out[2 * i + 1] / 100.0);
        sum += mhz;
        lo = mhz < lo ? mhz : lo;
        hi = mhz > hi ? mhz : hi;
        ++cnt;
    }
    if (cnt == 0) return ROBO_RUNTIME_ERROR;
    out3[0] = sum / cnt;
    out3[1] = lo;
    out3[2] = hi;
    return ROBO_OK;
}

}  // namespace robo
