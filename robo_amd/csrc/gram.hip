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
                                                                 FitSample* __restrict__ sp_out,
                                                                 int* __restrict__ tile_counter) {
    if (blockIdx.x == 0) {
        for (int d = threadIdx.x; d < dim; d += blockDim.x) ism_out[d] = ta.ism[d];
        if (threadIdx.x == 0) {
            *sp_out = ta.sp;
            *tile_counter = 0;     // the persistent gram kernel's tile hand-out starts at 0 (no memset launch)
        }
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
    int* d_counter = ctx->d_fail + 2;     // d_fail[0]: failure flag; [2]: tile counter of gram_persistent_kernel
    const long long total = (long long)rows_pad * dim;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_inputs_theta_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_in, d_out, ta,
                       (long long)rows_real, (long long)rows_pad, dim, d_ism_out, d_sp_out, d_counter);
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
template <class T, int KIND, int WPC = 0>
__global__ __launch_bounds__(256, WPC > 0 ? WPC : (KIND == ROBO_KERNEL_FABOLAS ? 3 : 6)) void gram_kernel(const double* __restrict__ Xs, size_t xs_stride,
                                                   const double* __restrict__ y, double* __restrict__ K,
                                                   size_t k_stride, int n, int n_pad,
                                                   const FitSample* __restrict__ sp, int* __restrict__ fail) {
    // the factorisation's failure flag of this sample starts at 0 (one memset launch less in front of the Cholesky)
    if (blockIdx.x == 0 && threadIdx.x == 0) fail[blockIdx.y] = 0;
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
    if (bi != bj && (int)i0 + GT <= n) {
        // interior tile (all rows and columns are training points, no diagonal entry): the values as they are --
        // a workgroup-uniform branch instead of ~10 compare/select instructions per entry (the kernel is VALU bound)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            double2* dst = reinterpret_cast<double2*>(K + (size_t)(i0 + ty * 4 + a) * n_pad + j0 + tx * 4);
            dst[0] = make_double2(cov[a][0], cov[a][1]);
            dst[1] = make_double2(cov[a][2], cov[a][3]);
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int gi = (int)i0 + ty * 4 + a;
        double v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gj = (int)j0 + tx * 4 + b;
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
        double2* dst = reinterpret_cast<double2*>(K + (size_t)gi * n_pad + j0 + tx * 4);
        dst[0] = make_double2(v[0], v[1]);
        dst[1] = make_double2(v[2], v[3]);
    }
}


// ---- K1 on 32 x 64 tiles (r04) ------------------------------------------------------------------------------------------
// gram_kernel's 2211 tiles at N = 4096 meet 1536 workgroup slots (6 per CU): the launch runs as two rounds of workgroups
// where 1.44 would do, and a workgroup's lifetime is a dependent chain (coordinates -> LDS -> distance passes -> rsq ->
// 13-term Horner -> stores).  Half-height tiles -- 32 rows x 64 columns, a 2 x 4 micro-tile per thread, 4422 tiles, eight
// workgroups per CU -- quantise the tail in units of half the work and halve every workgroup's chain.  Same arithmetic
// per entry, in the same order, as gram_kernel<double, KIND> (pair_cov_dot): K is bit-identical.
// MEASURED (r04d, MI355X, HIP events around the kernel): N = 4096 D = 16 36.9 us against gram_kernel's 33.8; N = 2048 17.9
// against 19.8; N = 8192 D = 64 251 against 184 -- the column block is staged once per 2048 pairs instead of once per 4096,
// and that costs more than the finer tail gains: the launch is not lost to round quantisation.  Tested option (tuning
// gram_half = 1), default off.
constexpr int HT = 32;          // tile rows
constexpr int HLD = HT + 2;

__device__ __forceinline__ void half_tile(int t, int& bi, int& bj) {
    // row blocks 2 p and 2 p + 1 (32 rows each) both own column blocks 0 .. p (64 columns each): 2 (p + 1) tiles per pair
    int p = (int)((sqrt(4.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((p + 1) * (p + 2) <= t) ++p;
    while (p * (p + 1) > t) --p;
    const int rem = t - p * (p + 1);
    bi = 2 * p + (rem > p ? 1 : 0);
    bj = rem > p ? rem - (p + 1) : rem;
}

template <int KIND>
__global__ __launch_bounds__(256, 8) void gram_half_kernel(const double* __restrict__ Xs, size_t xs_stride,
                                                           const double* __restrict__ y, double* __restrict__ K,
                                                           size_t k_stride, int n, int n_pad,
                                                           const FitSample* __restrict__ sp, int* __restrict__ fail) {
    if (blockIdx.x == 0 && threadIdx.x == 0) fail[blockIdx.y] = 0;
    __shared__ double sI[GD * HLD];
    __shared__ double sJ[GD * GLD];
    __shared__ double sN[HT + GT];
    Xs += (size_t)blockIdx.y * xs_stride;
    K += (size_t)blockIdx.y * k_stride;
    const CovParams cp = sp[blockIdx.y].cov;
    const double noise = sp[blockIdx.y].noise, mean_c = sp[blockIdx.y].mean_c;
    int bi, bj;
    half_tile(blockIdx.x, bi, bj);
    const long long i0 = (long long)bi * HT, j0 = (long long)bj * GT;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, dim = cp.dim;
    double dot[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) dot[a][b] = 0.0;
    double nrm = 0.0;                 // threads 0..31: |x_i|^2 of row i0 + t; 32..95: |x_j|^2 of row j0 + t - 32
    for (int d0 = 0; d0 < dim; d0 += GD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + e * 256;
            const int row = idx >> 4, d = idx & 15;
            const bool ok = d0 + d < dim;
            if (e < 2) sI[d * HLD + row] = ok ? Xs[(i0 + row) * dim + d0 + d] : 0.0;
            sJ[d * GLD + row] = ok ? Xs[(j0 + row) * dim + d0 + d] : 0.0;
        }
        __syncthreads();
        const int dn = dim - d0 < GD ? dim - d0 : GD;
        if (t < HT) {
            for (int d = 0; d < dn; ++d) {
                const double x = sI[d * HLD + t];
                nrm = fma(x, x, nrm);
            }
        } else if (t < HT + GT) {
            for (int d = 0; d < dn; ++d) {
                const double x = sJ[d * GLD + t - HT];
                nrm = fma(x, x, nrm);
            }
        }
#pragma unroll 4
        for (int d = 0; d < dn; ++d) {
            double xi[2], xj[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) xi[a] = sI[d * HLD + ty * 2 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) xj[b] = sJ[d * GLD + tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) dot[a][b] = fma(xi[a], xj[b], dot[a][b]);
        }
        __syncthreads();
    }
    if (t < HT + GT) sN[t] = nrm;
    __syncthreads();
    double ni[2], nj[4];
#pragma unroll
    for (int a = 0; a < 2; ++a) ni[a] = sN[ty * 2 + a];
#pragma unroll
    for (int b = 0; b < 4; ++b) nj[b] = sN[HT + tx * 4 + b];
    double cov[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double r2 = fma(-2.0, dot[a][b], ni[a] + nj[b]);
            r2 = r2 > 0.0 ? r2 : 0.0;
            if (i0 + ty * 2 + a == j0 + tx * 4 + b) r2 = 0.0;          // the diagonal is exact
            cov[a][b] = cov_finish<double, KIND>(cp, r2, 0.0);
        }
    if (j0 + GT <= i0 && (int)i0 + HT <= n) {
        // interior tile: no diagonal entry, every row and column a training point (workgroup-uniform branch)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            double2* dst = reinterpret_cast<double2*>(K + (size_t)(i0 + ty * 2 + a) * n_pad + j0 + tx * 4);
            dst[0] = make_double2(cov[a][0], cov[a][1]);
            dst[1] = make_double2(cov[a][2], cov[a][3]);
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int gi = (int)i0 + ty * 2 + a;
        double v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gj = (int)j0 + tx * 4 + b;
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
        double2* dst = reinterpret_cast<double2*>(K + (size_t)gi * n_pad + j0 + tx * 4);
        dst[0] = make_double2(v[0], v[1]);
        dst[1] = make_double2(v[2], v[3]);
    }
}

// ---- K1 with the pair dot products on the matrix pipe (r04) --------------------------------------------------------------
// gram_kernel spends 16 of its 69 fp64 VALU instructions per pair (D = 16) on x_i . x_j and stages both coordinate blocks
// through LDS behind two barriers per tile.  Here r2 = |x_i|^2 + |x_j|^2 - 2 x_i . x_j takes the cross term from
// v_mfma_f64_16x16x4_f64 -- ceil(D / 4) instructions per 16 x 16 pair tile on the otherwise idle matrix pipe -- with both
// operands read straight from global memory in fragment order (X is L2 resident: 0.5 MB at N = 4096); no LDS, no barrier,
// every wave runs on its own.  Wave w of a workgroup owns rows 16 w .. 16 w + 15 of the 64 x 64 tile and its four 16 x 16
// sub-tiles.  Row norms come from the operand fragments themselves (two cross-lane adds), the diagonal is exact by
// construction, r2 is clamped at 0: the same error model as pair_cov_dot (|x|^2 eps <= 1e-15 absolute in r2); entries
// differ from gram_kernel's in the last bits only (the dot product's association), K and the oracle's agree to rtol 1e-13.
// Fragment maps (gemm_f64.h): A/B operand lane l holds row l & 15, k = l >> 4; C reg r of lane l = (row (l >> 4) + 4 r,
// col l & 15).
// MEASURED SLOWER than gram_kernel (r04c: 37.2 vs 33.8 us at N = 4096 D = 16; 311 vs 182 us at N = 8192 D = 64) although it
// issues ~28 % fewer fp64 VALU instructions per pair (tools/isa_count.py): kept as a tested option, default off.
template <int KIND>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double* __restrict__ Xs, size_t xs_stride,
                                                        const double* __restrict__ y, double* __restrict__ K,
                                                        size_t k_stride, int n, int n_pad,
                                                        const FitSample* __restrict__ sp, int* __restrict__ fail) {
    if (blockIdx.x == 0 && threadIdx.x == 0) fail[blockIdx.y] = 0;
    Xs += (size_t)blockIdx.y * xs_stride;
    K += (size_t)blockIdx.y * k_stride;
    const CovParams cp = sp[blockIdx.y].cov;
    const double noise = sp[blockIdx.y].noise, mean_c = sp[blockIdx.y].mean_c;
    int bi, bj;
    tri_tile(blockIdx.x, bi, bj);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, dim = cp.dim;
    const int i0 = bi * GT + wave * 16, j0 = bj * GT;      // first row of this wave's strip, first column of the tile
    const int lr = lane & 15, lk = lane >> 4;
    v4d acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
    double na = 0.0, nb[4] = {0.0, 0.0, 0.0, 0.0};
    const double* pa = Xs + (size_t)(i0 + lr) * dim + lk;
    const double* pb = Xs + (size_t)(j0 + lr) * dim + lk;
    for (int k0 = 0; k0 < dim; k0 += 4) {
        const bool ok = k0 + lk < dim;
        const double a = ok ? pa[k0] : 0.0;
        double b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = ok ? pb[(size_t)c * 16 * dim + k0] : 0.0;
        na = fma(a, a, na);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            nb[c] = fma(b[c], b[c], nb[c]);
            acc[c] = mfma_f64(a, b[c], acc[c]);
        }
    }
    // |x|^2 of row (lane & 15): the four k-slices of a row sit in lanes l, l + 16, l + 32, l + 48
    na += __shfl_xor(na, 16);
    na += __shfl_xor(na, 32);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        nb[c] += __shfl_xor(nb[c], 16);
        nb[c] += __shfl_xor(nb[c], 32);
    }
    double ni[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ni[r] = __shfl(na, lk + 4 * r);      // norm of C-layout row (l >> 4) + 4 r
    double val[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double r2 = fma(-2.0, acc[c][r], ni[r] + nb[c]);
            val[c][r] = r2 > 0.0 ? r2 : 0.0;
        }
    if (bi != bj && bi * GT + GT <= n) {
        // interior tile (all rows and columns are training points, no diagonal entry): a workgroup-uniform branch, the
        // values as they are
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                K[(size_t)(i0 + lk + 4 * r) * n_pad + j0 + c * 16 + lr] = cov_finish<double, KIND>(cp, val[c][r], 0.0);
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = i0 + lk + 4 * r, gj = j0 + c * 16 + lr;
            double v = cov_finish<double, KIND>(cp, gi == gj ? 0.0 : val[c][r], 0.0);     // the diagonal is exact
            if (gi < n && gj < n) {
                if (gi == gj) v += noise;
            } else if (gi == gj) {
                v = 1.0;
            } else if (gi == n && gj < n) {
                v = y[gj] - mean_c;
            } else if (gj == n && gi < n) {
                v = y[gi] - mean_c;
            } else {
                v = 0.0;
            }
            K[(size_t)gi * n_pad + gj] = v;
        }
}

// ---- K1, single theta, fp64 stationary kernels: PERSISTENT workgroups -------------------------------------------------
// gram_kernel above launches one short workgroup per 64 x 64 tile (2211 at N = 4096): every resident workgroup goes
// through  load coordinates -> barrier -> 16 LDS-fed distance passes -> sqrt / exp -> store  in the same phase at the
// same time, the dispatcher has 8844 waves to start, and the fp64 VALU -- the bounding resource, 69 instructions per
// pair -- was busy for less than half of the kernel's 76k cycles (r02 PMC and ablations).  Here WPC
// workgroups per CU stay for the whole kernel and take tiles from an atomic counter (heaviest-first is irrelevant: all
// tiles cost the same); the NEXT tile's coordinates are requested before the current tile's math and land in registers
// while it runs (double-buffered LDS images), stores drain behind the following tile.  Same arithmetic per entry as
// gram_kernel<double, KIND> (pair_cov_dot): bit-identical K.
// MEASURED SLOWER (r03d, MI355X, N = 4096 D = 16): 42.0 us with 3..8 workgroups per CU, 54 us with one, against 34.5 us
// for gram_kernel (N = 8192 D = 64: 195 vs 185 us; N = 2048: 23 vs 17.5 us).  The kernel is bound by its fp64 VALU work
// (69 instructions per pair = 15.8 us at 100 % issue), not by phase alignment or dispatch: at 122 VGPRs only four waves
// per SIMD are resident instead of six, and that costs more than the prefetch hides.  Kept as an option (tuning key
// gram_persistent = workgroups per CU, default 0 = off) with its equality test; see DESIGN.md section 4.
struct TileRegs {
    double v[8];    // 64 rows x 16 dims of the i block (0..3) and of the j block (4..7): thread t holds [row (t + 256 e) >> 4][d = t & 15]
};

__device__ __forceinline__ TileRegs tile_coords_load(const double* __restrict__ X, long long i0, long long j0, int dim,
                                                     int d0) {
    const int t = threadIdx.x, d = t & 15;
    const bool ok = d0 + d < dim;
    TileRegs r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = (t + e * 256) >> 4;
        r.v[e] = ok ? X[(i0 + row) * dim + d0 + d] : 0.0;
        r.v[4 + e] = ok ? X[(j0 + row) * dim + d0 + d] : 0.0;
    }
    return r;
}

__device__ __forceinline__ void tile_coords_stage(const TileRegs& r, double* sI, double* sJ) {
    const int t = threadIdx.x, d = t & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = (t + e * 256) >> 4;
        sI[d * GLD + row] = r.v[e];
        sJ[d * GLD + row] = r.v[4 + e];
    }
}

template <int KIND>
__global__ __launch_bounds__(256, 4) void gram_persistent_kernel(const double* __restrict__ Xs,
                                                                 const double* __restrict__ y, double* __restrict__ K,
                                                                 int n, int n_pad, const FitSample* __restrict__ sp,
                                                                 int* __restrict__ fail, int* __restrict__ counter,
                                                                 int tiles) {
    __shared__ double sI[2][GD * GLD];
    __shared__ double sJ[2][GD * GLD];
    __shared__ double sN[2 * GT];
    __shared__ int sNext;
    if (blockIdx.x == 0 && threadIdx.x == 0) fail[0] = 0;
    const CovParams cp = sp[0].cov;
    const double noise = sp[0].noise, mean_c = sp[0].mean_c;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, dim = cp.dim;
    const double* sMine = nullptr;
    int cur = (int)blockIdx.x;                 // the first gridDim.x tiles are handed out statically
    if (cur >= tiles) return;
    int bi, bj;
    tri_tile(cur, bi, bj);
    TileRegs regs = tile_coords_load(Xs, (long long)bi * GT, (long long)bj * GT, dim, 0);
    int buf = 0;
    for (;;) {
        const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;
        // claim the next tile and request its first coordinate chunk: in flight during this tile's math
        if (t == 0) sNext = atomicAdd(counter, 1) + (int)gridDim.x;
        tile_coords_stage(regs, sI[buf], sJ[buf]);
        __syncthreads();                        // staged coordinates + sNext visible
        const int nxt = sNext;
        int nbi = 0, nbj = 0;
        TileRegs nregs;
        if (nxt < tiles) {
            tri_tile(nxt, nbi, nbj);
            nregs = tile_coords_load(Xs, (long long)nbi * GT, (long long)nbj * GT, dim, 0);
        }
        // ---- distances of this tile (pair_cov_dot, chunk by chunk; chunk 0 is already staged)
        double dot[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) dot[a][b] = 0.0;
        double nrm = 0.0;
        for (int d0 = 0; d0 < dim; d0 += GD) {
            double* cI = sI[buf];
            double* cJ = sJ[buf];
            if (d0 > 0) {                       // dim > 16: further chunks are staged synchronously
                __syncthreads();
                const TileRegs more = tile_coords_load(Xs, i0, j0, dim, d0);
                tile_coords_stage(more, cI, cJ);
                __syncthreads();
            }
            sMine = t < GT ? cI : cJ;
            const int dn = dim - d0 < GD ? dim - d0 : GD;
            if (t < 2 * GT)
                for (int d = 0; d < dn; ++d) {
                    const double x = sMine[d * GLD + (t & (GT - 1))];
                    nrm = fma(x, x, nrm);
                }
            for (int d = 0; d < dn; ++d) {
                double xi[4], xj[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    xi[a] = cI[d * GLD + ty * 4 + a];
                    xj[a] = cJ[d * GLD + tx * 4 + a];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) dot[a][b] = fma(xi[a], xj[b], dot[a][b]);
            }
        }
        __syncthreads();                        // sN of the previous tile fully read
        if (t < 2 * GT) sN[t] = nrm;
        __syncthreads();
        double ni[4], nj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            ni[a] = sN[ty * 4 + a];
            nj[a] = sN[GT + tx * 4 + a];
        }
        double cov[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double r2 = fma(-2.0, dot[a][b], ni[a] + nj[b]);
                r2 = r2 > 0.0 ? r2 : 0.0;
                if (i0 + ty * 4 + a == j0 + tx * 4 + b) r2 = 0.0;     // the diagonal is exact
                cov[a][b] = cov_finish<double, KIND>(cp, r2, 0.0);
            }
        if (bi != bj && (int)i0 + GT <= n) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double2* dst = reinterpret_cast<double2*>(K + (size_t)(i0 + ty * 4 + a) * n_pad + j0 + tx * 4);
                dst[0] = make_double2(cov[a][0], cov[a][1]);
                dst[1] = make_double2(cov[a][2], cov[a][3]);
            }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int gi = (int)i0 + ty * 4 + a;
                double v[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int gj = (int)j0 + tx * 4 + b;
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
                double2* dst = reinterpret_cast<double2*>(K + (size_t)gi * n_pad + j0 + tx * 4);
                dst[0] = make_double2(v[0], v[1]);
                dst[1] = make_double2(v[2], v[3]);
            }
        }
        if (nxt >= tiles) break;
        regs = nregs;
        bi = nbi;
        bj = nbj;
        buf ^= 1;
    }
}

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

int launch_gram(robo_gp* gp, const FitBuffers& fb) {
    const int T = gp->n_pad / GT;
    const int tiles = T * (T + 1) / 2;
    const Tuning& tune = gp->ctx->tune;
    if (fb.S == 1 && fb.K == gp->d_K && !gp->fp32_gram && gp->kind != ROBO_KERNEL_FABOLAS && tune.gram_persistent > 0 &&
        tiles > gp->ctx->num_cu * 2) {
        // single theta, fp64, stationary kernel (the headline path): persistent workgroups; the counter was zeroed by
        // launch_scale_inputs_theta, which always precedes a single-theta gram build
        const int wpc = tune.gram_persistent;
        int grid = gp->ctx->num_cu * wpc;
        if (grid > tiles) grid = tiles;
        int* counter = gp->ctx->d_fail + 2;
        if (gp->kind == ROBO_KERNEL_MATERN52_ARD)
            hipLaunchKernelGGL(gram_persistent_kernel<ROBO_KERNEL_MATERN52_ARD>, dim3(grid), dim3(256), 0, gp->ctx->stream,
                               fb.Xs, (const double*)gp->d_y, fb.K, gp->n, gp->n_pad, fb.sp, fb.fail, counter, tiles);
        else
            hipLaunchKernelGGL(gram_persistent_kernel<ROBO_KERNEL_RBF_ARD>, dim3(grid), dim3(256), 0, gp->ctx->stream,
                               fb.Xs, (const double*)gp->d_y, fb.K, gp->n, gp->n_pad, fb.sp, fb.fail, counter, tiles);
        ROBO_LAUNCH_CHECK();
        return ROBO_OK;
    }
    // A/B option (tuning gram_mfma = 1): the pair dot products on the matrix pipe -- MEASURED SLOWER (r04c, MI355X: N = 4096
    // D = 16 37.2 vs 33.8 us, N = 8192 D = 64 311 vs 182 us: its operands come from global memory in fragment order, 8-byte
    // strided loads, and a 16 x 16 x 4 fp64 MFMA per 4 dimensions does not amortise them)
    if (!gp->fp32_gram && gp->kind != ROBO_KERNEL_FABOLAS && tune.gram_mfma > 0) {
        if (gp->kind == ROBO_KERNEL_MATERN52_ARD)
            hipLaunchKernelGGL(gram_mfma_kernel<ROBO_KERNEL_MATERN52_ARD>, dim3(tiles, fb.S), dim3(256), 0, gp->ctx->stream,
                               fb.Xs, fb.xs_stride, (const double*)gp->d_y, fb.K, fb.k_stride, gp->n, gp->n_pad, fb.sp,
                               fb.fail);
        else
            hipLaunchKernelGGL(gram_mfma_kernel<ROBO_KERNEL_RBF_ARD>, dim3(tiles, fb.S), dim3(256), 0, gp->ctx->stream,
                               fb.Xs, fb.xs_stride, (const double*)gp->d_y, fb.K, fb.k_stride, gp->n, gp->n_pad, fb.sp,
                               fb.fail);
        ROBO_LAUNCH_CHECK();
        return ROBO_OK;
    }
    if (!gp->fp32_gram && gp->kind != ROBO_KERNEL_FABOLAS && tune.gram_half > 0) {
        // A/B option: 32 x 64 tiles, bit-identical entries (measured slower at N >= 4096, see gram_half_kernel)
        const int P2 = gp->n_pad / GT;                       // pairs of 32-row blocks
        const int half_tiles = P2 * (P2 + 1);
        if (gp->kind == ROBO_KERNEL_MATERN52_ARD)
            hipLaunchKernelGGL(gram_half_kernel<ROBO_KERNEL_MATERN52_ARD>, dim3(half_tiles, fb.S), dim3(256), 0,
                               gp->ctx->stream, fb.Xs, fb.xs_stride, (const double*)gp->d_y, fb.K, fb.k_stride, gp->n,
                               gp->n_pad, fb.sp, fb.fail);
        else
            hipLaunchKernelGGL(gram_half_kernel<ROBO_KERNEL_RBF_ARD>, dim3(half_tiles, fb.S), dim3(256), 0,
                               gp->ctx->stream, fb.Xs, fb.xs_stride, (const double*)gp->d_y, fb.K, fb.k_stride, gp->n,
                               gp->n_pad, fb.sp, fb.fail);
        ROBO_LAUNCH_CHECK();
        return ROBO_OK;
    }
    if (!gp->fp32_gram && gp->kind == ROBO_KERNEL_MATERN52_ARD && tune.gram_occ >= 4 && tune.gram_occ <= 8 &&
        tune.gram_occ != 6) {
        // A/B option: the same kernel compiled for 4 / 5 / 7 / 8 workgroups per CU (default six = 80 VGPRs).  Measured
        // (r04f, N = 4096 D = 16): seven 41.5 us, eight 45.2 us (64 VGPRs, 76 bytes of scratch) against 34.6 us
#define ROBO_GRAM_OCC(W)                                                                                             \
    hipLaunchKernelGGL((gram_kernel<double, ROBO_KERNEL_MATERN52_ARD, W>), dim3(tiles, fb.S), dim3(256), 0,          \
                       gp->ctx->stream, fb.Xs, fb.xs_stride, (const double*)gp->d_y, fb.K, fb.k_stride, gp->n,       \
                       gp->n_pad, fb.sp, fb.fail)
        switch (tune.gram_occ) {
            case 4: ROBO_GRAM_OCC(4); break;
            case 5: ROBO_GRAM_OCC(5); break;
            case 7: ROBO_GRAM_OCC(7); break;
            default: ROBO_GRAM_OCC(8); break;
        }
#undef ROBO_GRAM_OCC
        ROBO_LAUNCH_CHECK();
        return ROBO_OK;
    }
#define ROBO_GRAM_CALL(TYPE, KIND)                                                                              \
    hipLaunchKernelGGL((gram_kernel<TYPE, KIND>), dim3(tiles, fb.S), dim3(256), 0, gp->ctx->stream, fb.Xs,      \
                       fb.xs_stride, (const double*)gp->d_y, fb.K, fb.k_stride, gp->n, gp->n_pad, fb.sp, fb.fail)
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
