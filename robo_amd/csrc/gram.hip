// K1 gram_build and K4 cross_gram: pairwise scaled squared distances from LDS-tiled
// X blocks, covariance function, coalesced fp64 row stores.
//
// Replaces the kernel-matrix assembly inside george.GP.compute / GP.predict
// (reference call sites robo/models/gaussian_process.py:119,155,280).
//
// Both kernels use 64x64 output tiles, 256 threads, a 4x4 register micro-tile per thread.
// Distances are accumulated as sum_d (x_d - x'_d)^2 on inputs pre-scaled by 1/sqrt(m_d)
// (direct differences, not the |x|^2+|x'|^2-2xx' expansion: exact 0 on the diagonal, no
// cancellation -- these values feed a Cholesky with sigma^2 = 1e-3).
//
// HBM traffic per launch: gram   8 * n_pad*(n_pad+64)/2 bytes written, 8*n*D read
//                         cross  8 * rows*n_pad written, 8*(rows+n)*D read
#include "common.h"
#include "gram_tile.h"      // GT, GD, GLD, pair_cov_dot
#include "kern_math.h"

namespace robo {

__global__ __launch_bounds__(256) void scale_inputs_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                           const double* __restrict__ inv_sqrt_m, long long rows_real,
                                                           long long rows_pad, int dim, size_t out_stride,
                                                           size_t ism_stride) {
    out += (size_t)blockIdx.y * out_stride;          // batch coordinate: one scaling per theta
    inv_sqrt_m += (size_t)blockIdx.y * ism_stride;
    const long long total = rows_pad * dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / dim;
        const int d = (int)(i - r * dim);
        // pad rows replicate row 0 so that every lane of the tiled kernels computes on finite data
        const long long src = r < rows_real ? r : 0;
        out[i] = rows_real > 0 ? in[src * dim + d] * inv_sqrt_m[d] : 0.0;
    }
}

// Single-theta fits: the sample (covariance parameters, noise, mean) and the inverse square-root metrics arrive as
// KERNEL ARGUMENTS (2.1 KB by value) instead of through a pinned staging buffer and a host-to-device copy launch --
// the copy alone was ~10 % of a fit at BO-typical N <= 100.  Block 0 leaves device copies for the kernels that follow
// (gram: FitSample; posterior: candidate scaling by the same metrics).
__global__ __launch_bounds__(256) void scale_inputs_theta_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                                 const ThetaArgs ta, long long rows_real,
                                                                 long long rows_pad, int dim,
                                                                 double* __restrict__ ism_out,
                                                                 FitSample* __restrict__ sp_out) {
    if (blockIdx.x == 0) {
        for (int d = threadIdx.x; d < dim; d += blockDim.x) ism_out[d] = ta.ism[d];
        if (threadIdx.x == 0) *sp_out = ta.sp;
    }
    const long long total = rows_pad * dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / dim;
        const int d = (int)(i - r * dim);
        const long long src = r < rows_real ? r : 0;
        out[i] = rows_real > 0 ? in[src * dim + d] * ta.ism[d] : 0.0;
    }
}

int launch_scale_inputs_theta(robo_ctx* ctx, const double* d_in, double* d_out, const ThetaArgs& ta, int64_t rows_real,
                              int64_t rows_pad, int dim, double* d_ism_out, FitSample* d_sp_out) {
    const long long total = (long long)rows_pad * dim;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_inputs_theta_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_in, d_out, ta,
                       (long long)rows_real, (long long)rows_pad, dim, d_ism_out, d_sp_out);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// cov[a][b] = k(Xi[i0 + ty*4 + a], Xj[j0 + tx*4 + b]) ; Xi/Xj row-major (rows, dim), pre-scaled
template <class T, int KIND>
__device__ __forceinline__ void pair_cov(const CovParams& cp, const double* __restrict__ Xi,
                                         const double* __restrict__ Xj, long long i0, long long j0, double* sI,
                                         double* sJ, double (&cov)[4][4]) {
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, dim = cp.dim;
    constexpr bool fab = KIND == ROBO_KERNEL_FABOLAS;
    T acc[4][4], uu[4][4];
    T ss[fab ? 4 : 1][fab ? 4 : 1];      // Fabolas: sum of the per-dimension exponents (matern52_1d_split)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            cov_init<T, KIND>(cp, acc[a][b], uu[a][b]);
            if (fab) ss[a][b] = T(0);
        }
    for (int d0 = 0; d0 < dim; d0 += GD) {
        // stage 64 rows x 16 dims of both blocks, transposed to [d][row]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + e * 256;
            const int row = idx >> 4, d = idx & 15;
            const bool ok = d0 + d < dim;
            sI[d * GLD + row] = ok ? Xi[(i0 + row) * dim + d0 + d] : 0.0;
            sJ[d * GLD + row] = ok ? Xj[(j0 + row) * dim + d0 + d] : 0.0;
        }
        __syncthreads();
        const int dn = dim - d0 < GD ? dim - d0 : GD;
        for (int d = 0; d < dn; ++d) {
            T xi[4], xj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                xi[a] = (T)sI[d * GLD + ty * 4 + a];
                xj[a] = (T)sJ[d * GLD + tx * 4 + a];
            }
            if (fab && d0 + d < dim - 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) matern52_1d_split(xi[a] - xj[b], acc[a][b], ss[fab ? a : 0][fab ? b : 0]);
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) cov_step<T, KIND>(cp, d0 + d, xi[a], xj[b], acc[a][b], uu[a][b]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (fab) acc[a][b] *= exp_np(-ss[fab ? a : 0][fab ? b : 0]);
            cov[a][b] = cov_finish<T, KIND>(cp, acc[a][b], uu[a][b]);
        }
}

// K[i][j] for j-tile <= i-tile.  Rows/cols >= n: row n is the augmented right-hand side
// (y - mean), the rest identity, so that one Cholesky also yields z = L^-1 (y - mean)
// as row n of the factor (DESIGN.md "augmented row").
// (six workgroups per CU = 80 VGPRs; the Fabolas product kernel needs more live values -- one Matern factor per pair and
// dimension -- and spilled 92 registers under that cap: three per CU for it)
// WPC: workgroups per CU the register allocation is asked to allow (0: the default, six -- three for the Fabolas kernel)
template <class T, int KIND>
__global__ __launch_bounds__(256, KIND == ROBO_KERNEL_FABOLAS ? 3 : 6) void gram_kernel(const double* __restrict__ Xs, size_t xs_stride,
                                                   const double* __restrict__ y, double* __restrict__ K,
                                                   size_t k_stride, int n, int n_pad,
                                                   const FitSample* __restrict__ sp, int* __restrict__ fail,
                                                   unsigned* __restrict__ prog, double* __restrict__ Linv,
                                                   size_t linv_stride, int nbf) {
    // the factorisation's failure flag of this sample starts at 0 (one memset launch less in front of the Cholesky)
    if (blockIdx.x == 0 && threadIdx.x == 0) fail[blockIdx.y] = 0;
    // ... and so do the follower hand-off's words (potrf.hip: progress words to zero, every panel's W_77 slot to the sentinel
    // that makes its entries their own flag) -- the launch that used to do this was 5 us of an 84-us ensemble half-step
    if (prog) {
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < PROG_STRIDE; i += 256) prog[(size_t)blockIdx.y * PROG_STRIDE + i] = 0u;
        if ((int)blockIdx.x < nbf)
            Linv[(size_t)blockIdx.y * linv_stride + (size_t)blockIdx.x * NB * NB +
                 (size_t)(7 * 16 + (threadIdx.x >> 4)) * NB + 7 * 16 + (threadIdx.x & 15)] = __longlong_as_double(FOLLOW_SENTINEL);
    }
    __shared__ double sI[GD * GLD];
    __shared__ double sJ[GD * GLD];
    __shared__ double sN[2 * GT];
    Xs += (size_t)blockIdx.y * xs_stride;
    K += (size_t)blockIdx.y * k_stride;
    const CovParams cp = sp[blockIdx.y].cov;
    const double noise = sp[blockIdx.y].noise, mean_c = sp[blockIdx.y].mean_c;
    int bi, bj;
    tri_tile(blockIdx.x, bi, bj);
    const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;
    double cov[4][4];
    if constexpr (sizeof(T) == 8 && KIND != ROBO_KERNEL_FABOLAS) pair_cov_dot<KIND>(cp, Xs, i0, j0, sI, sJ, sN, cov);
    else pair_cov<T, KIND>(cp, Xs, Xs, i0, j0, sI, sJ, cov);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    // which columns of the tile this thread's four entries per row are (gram_tile.h: the fp64 stationary kernels' tiles use
    // gram_col, the others four consecutive columns); c0 / c2: first column of the entry pairs (0, 1) and (2, 3)
    constexpr bool DOT = sizeof(T) == 8 && KIND != ROBO_KERNEL_FABOLAS;
    const int c0 = DOT ? gram_col(tx, 0) : 4 * tx, c2 = DOT ? gram_col(tx, 2) : 4 * tx + 2;
    if (bi != bj && (int)i0 + GT <= n) {
        // interior tile (all rows and columns are training points, no diagonal entry): the values as they are --
        // a workgroup-uniform branch instead of ~10 compare/select instructions per entry (the kernel is VALU bound)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            double* row = K + (size_t)(i0 + ty * 4 + a) * n_pad + j0;
            *reinterpret_cast<double2*>(row + c0) = make_double2(cov[a][0], cov[a][1]);
            *reinterpret_cast<double2*>(row + c2) = make_double2(cov[a][2], cov[a][3]);
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int gi = (int)i0 + ty * 4 + a;
        double v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gj = (int)j0 + (b < 2 ? c0 + b : c2 + b - 2);
            double val;
            if (gi < n && gj < n) {
                val = cov[a][b];
                if (gi == gj) val += noise;
            } else if (gi == gj) {
                val = 1.0;
            } else if (gi == n && gj < n) {
                val = y[gj] - mean_c;
            } else if (gj == n && gi < n) {
                val = y[gi] - mean_c;
            } else {
                val = 0.0;
            }
            v[b] = val;
        }
        double* row = K + (size_t)gi * n_pad + j0;
        *reinterpret_cast<double2*>(row + c0) = make_double2(v[0], v[1]);
        *reinterpret_cast<double2*>(row + c2) = make_double2(v[2], v[3]);
    }
}


// Three further forms of K1 were built, measured slower on the MI355X and REMOVED from the product in round 5 (history:
// NOTES.md r02u / r03d / r04c-g, git 6e2ae1d): x.x' on the matrix pipe with fragment-ordered global operands (37.2 vs 33.8 us
// at N = 4096 D = 16), 32 x 64 tiles (36.9 us), persistent workgroups with prefetch (42.0 us), and gram_kernel compiled for
// 4 / 5 / 7 / 8 workgroups per CU (34.3 / 34.2 / 41.5 / 45.2 us against 34.6).  The kernel is bound by fp64 VALU issue (69.6
// instructions per pair); none of the structural variants moved that.

// V[c - c0][j] = k(xc_c, x_j) for j < n, 0 for n <= j < n_pad
template <class T, int KIND>
__global__ __launch_bounds__(256) void cross_gram_kernel(const double* __restrict__ Xcs,
                                                         const double* __restrict__ Xs, double* __restrict__ V,
                                                         long long c0, int n, int n_pad, CovParams cp) {
    __shared__ double sI[GD * GLD];
    __shared__ double sJ[GD * GLD];
    const long long i0 = (long long)blockIdx.y * GT;   // candidate tile (chunk local)
    const long long j0 = (long long)blockIdx.x * GT;   // train tile
    double cov[4][4];
    pair_cov<T, KIND>(cp, Xcs, Xs, c0 + i0, j0, sI, sJ, cov);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long long ci = i0 + ty * 4 + a;
        double v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gj = (int)j0 + tx * 4 + b;
            v[b] = gj < n ? cov[a][b] : 0.0;
        }
        double2* dst = reinterpret_cast<double2*>(V + (size_t)ci * n_pad + j0 + tx * 4);
        dst[0] = make_double2(v[0], v[1]);
        dst[1] = make_double2(v[2], v[3]);
    }
}

int launch_scale_inputs(robo_ctx* ctx, const double* d_in, double* d_out, const double* d_inv_sqrt_metric,
                        int64_t rows_real, int64_t rows_pad, int dim, int S, size_t out_stride, size_t ism_stride) {
    const long long total = (long long)rows_pad * dim;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_inputs_kernel, dim3(blocks, S), dim3(256), 0, ctx->stream, d_in, d_out,
                       d_inv_sqrt_metric, (long long)rows_real, (long long)rows_pad, dim, out_stride, ism_stride);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// instantiate per (precision, kind); K is robo_kernel_kind
#define ROBO_DISPATCH_COV(fp32, kind, CALL)                                        \
    do {                                                                           \
        if (fp32) {                                                                \
            if ((kind) == ROBO_KERNEL_MATERN52_ARD) { CALL(float, ROBO_KERNEL_MATERN52_ARD); }      \
            else if ((kind) == ROBO_KERNEL_RBF_ARD) { CALL(float, ROBO_KERNEL_RBF_ARD); }           \
            else { CALL(float, ROBO_KERNEL_FABOLAS); }                             \
        } else {                                                                   \
            if ((kind) == ROBO_KERNEL_MATERN52_ARD) { CALL(double, ROBO_KERNEL_MATERN52_ARD); }     \
            else if ((kind) == ROBO_KERNEL_RBF_ARD) { CALL(double, ROBO_KERNEL_RBF_ARD); }          \
            else { CALL(double, ROBO_KERNEL_FABOLAS); }                            \
        }                                                                          \
    } while (0)

// samples s0 .. s0 + ns - 1 of the batch (default: all of it) on `stream` (default: the context's)
int launch_gram(robo_gp* gp, const FitBuffers& fb, hipStream_t stream, int s0, int ns) {
    const int T = gp->n_pad / GT;
    const int tiles = T * (T + 1) / 2;
    const int nb = gp->n_pad / NB, nbf = (gp->n % NB == 0 && nb > 1) ? nb - 1 : nb;     // factored panels (launch_potrf)
    if (!stream) stream = gp->ctx->stream;
    if (ns < 0) ns = fb.S - s0;
#define ROBO_GRAM_CALL(TYPE, KIND)                                                                              \
    hipLaunchKernelGGL((gram_kernel<TYPE, KIND>), dim3(tiles, ns), dim3(256), 0, stream,                        \
                       fb.Xs + (size_t)s0 * fb.xs_stride, fb.xs_stride, (const double*)gp->d_y,                 \
                       fb.K + (size_t)s0 * fb.k_stride, fb.k_stride, gp->n, gp->n_pad, fb.sp + s0, fb.fail + s0,           \
                       fb.prog ? fb.prog + (size_t)s0 * PROG_STRIDE : (unsigned*)nullptr,                                \
                       fb.Linv + (size_t)s0 * fb.linv_stride, fb.linv_stride, nbf)
    ROBO_DISPATCH_COV(gp->fp32_gram, gp->kind, ROBO_GRAM_CALL);
#undef ROBO_GRAM_CALL
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_cross_gram(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn, double* d_out) {
    // cn is a multiple of NB (=2*GT); d_out: (cn x n_pad) destination, the solve workspace by default
    const dim3 grid(gp->n_pad / GT, (unsigned)(cn / GT));
    if (!d_out) d_out = cand->d_V;
#define ROBO_CROSS_CALL(TYPE, KIND)                                                                             \
    hipLaunchKernelGGL((cross_gram_kernel<TYPE, KIND>), grid, dim3(256), 0, gp->ctx->stream,                    \
                       (const double*)cand->d_Xcs, (const double*)gp->d_Xs, d_out, (long long)c0, gp->n,        \
                       gp->n_pad, gp->cov)
    ROBO_DISPATCH_COV(gp->fp32_gram, gp->kind, ROBO_CROSS_CALL);
#undef ROBO_CROSS_CALL
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
