// Analytic gradient of the GP log marginal likelihood w.r.t. the log-space hyper-parameters
// (SURVEY.md section 8 row f1).
//
// Replaces the body of GaussianProcess.grad_nll (robo/models/gaussian_process.py:168-191):
//     K_inv = solver.apply_inverse(eye(N));  Kg = kernel.gradient(X) (N, N, P) ++ eye(N)
//     A = outer(alpha, alpha) - K_inv;       g = 0.5 einsum('ijk,ij', Kg, A)
// The reference materialises K^-1 (N x N) and the (N, N, P) kernel-gradient tensor on the host
// (2.3 GB at N = 4096, D = 16).  Here, after the factorisation the fit already did:
//   1. W = L^-1 by divide and conquer on the 128-blocks (triinv_kernel): with the diagonal-block
//      inverses of the factorisation as the base, level s merges pairs of inverted s x s diagonal
//      blocks,  [[A, 0], [B, C]]^-1 = [[A^-1, 0], [-C^-1 B A^-1, C^-1]],  as two NT GEMMs per pair --
//      log2(N / 128) levels of chip-wide GEMM launches (N^3/3 MFMA flops) instead of N / 128
//      dependent block rows with at most N / 128 workgroups each (the block-row TRSM on the identity
//      took 11 ms at N = 4096; this takes ~1 ms).  Both W and V = W^T are kept (dual store), the
//      scratch product lives in the unused upper triangle of W.  alpha = V z by a GEMV.
//   2. kinv_tile_kernel: A = alpha alpha^T - W^T W, lower 128 x 128 tiles, NT GEMM over k >= tile row
//      (N^3/3 MFMA flops); A is the only N x N intermediate (fp64, n_pad^2).
//   3. grad_reduce_kernel: per lower 64 x 64 tile, the pair kernel of gram.hip once more -- r^2,
//      k, dk/dr^2 -- and sum_ij A_ij dK_ij/dtheta_p for every p; the (N, N, P) tensor never exists.
//      Deterministic: fixed-order block reductions into part[p][tile], summed by grad_final_kernel.
// As in the reference (:178-182) the "gradient" of the Gram matrix w.r.t. the last entry of theta
// (log sigma^2) is the identity, i.e. the last component is d/d sigma^2 (DESIGN.md "Mirrored quirks").
#include "common.h"
#include "gemm_f64.h"
#include "kern_math.h"

namespace robo {

constexpr int RT = 64;        // tile edge of the reduction pass
constexpr int RD = 16;        // dims per LDS pass
constexpr int RLD = RT + 2;

// base of the recursion: W_ii = Linv_i, V_ii = Linv_i^T (columns >= n of V zeroed: the augmented row
// and the identity padding of the factor take no part in K^-1)
__global__ __launch_bounds__(256) void triinv_base_kernel(const double* __restrict__ Linv, int n,
                                                          double* __restrict__ W, double* __restrict__ V, int ld) {
    const int b = blockIdx.x;
    const double* src = Linv + (size_t)b * NB * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e >> 7, c = e & 127;
        const double v = src[e];
        W[(size_t)(b * NB + r) * ld + b * NB + c] = v;
        V[(size_t)(b * NB + c) * ld + b * NB + r] = (b * NB + r) < n ? v : 0.0;
    }
}

// one level of the merge, sb = 128-blocks per already inverted diagonal block.  Pair p: A = blocks
// [2 p sb, 2 p sb + sb), C = the next sb blocks (clipped to nbk), B = L[C, A].
//   phase 0:  T^T = A^-T B^T     tile (j-block of A, i-block of C):  sum_k V[j][k] L[i][k], k in A, k >= j
//             -> upper triangle of W (rows A, columns C), scratch
//   phase 1:  X = -C^-1 T        tile (i-block of C, j-block of A):  sum_k W[i][k] T^T[j][k], k in C, k <= i
//             -> W[C, A] = X,  V[A, C] = X^T (columns >= n zeroed)
// TM: 32 TM rows of the output tile per workgroup.  The contraction lengths of a level are triangular (1 .. sb blocks), and
// at the top levels there are no more 128-row tiles than CUs (sb = 16: 256 tiles, the longest 16 blocks deep = 260 us per
// phase with half of the matrix pipes idle): 32-row sub-tiles give the dispatcher 4x the workgroups to balance
// (r03: W = L^-1 at N = 4096 1.23 -> ms).  Same k-order per entry, hence the same bits for every TM.
template <int TM>
__global__ __launch_bounds__(256, 2) void triinv_kernel(int phase, int sb, int nbk, int n,
                                                        const double* __restrict__ L, double* __restrict__ W,
                                                        double* __restrict__ V, int ld) {
    __shared__ double smem[gemm_smem_doubles<TM>()];
    constexpr int SPLIT = 4 / TM;
    const int a0 = 2 * (int)blockIdx.y * sb, c0 = a0 + sb;
    int nc = nbk - c0;
    nc = nc > sb ? sb : nc;
    if (nc <= 0) return;
    const int tile = (int)blockIdx.x / SPLIT, h = (int)blockIdx.x % SPLIT;
    const int tj = tile % sb, ti = tile / sb;   // block inside A, block inside C
    if (ti >= nc) return;
    const size_t rowA = (size_t)(a0 + tj) * NB, rowC = (size_t)(c0 + ti) * NB;
    const size_t sub = (size_t)h * (32 * TM);    // first row of this workgroup's sub-tile
    AccT<TM> acc;
    acc_zero(acc);
    if (phase == 0) {
        gemm_nt<TM, false>(V + (rowA + sub) * ld + (size_t)a0 * NB, ld, L + rowC * ld + (size_t)a0 * NB, ld, tj * NB,
                           sb * NB, acc, smem);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    W[(rowA + sub + acc_row<TM>(tm, r)) * ld + rowC + acc_col(tn)] = acc.t[tm][tn][r];
    } else {
        gemm_nt<TM, true>(W + (rowC + sub) * ld + (size_t)c0 * NB, ld, W + rowA * ld + (size_t)c0 * NB, ld, 0,
                          (ti + 1) * NB, acc, smem);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t i = rowC + sub + acc_row<TM>(tm, r), j = rowA + acc_col(tn);
                    const double x = acc.t[tm][tn][r];
                    W[i * ld + j] = x;
                    V[j * ld + i] = (int)i < n ? x : 0.0;
                }
    }
}

// alpha[c] = sum_{k < n} V[c][k] z[k]  (= (W^T z)_c = (K^-1 (y - mean))_c), one wavefront per row
__global__ __launch_bounds__(256) void alpha_kernel(const double* __restrict__ V, int ld, const double* __restrict__ z,
                                                    int n, int rows, double* __restrict__ alpha) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= rows) return;
    const double* v = V + (size_t)c * ld;
    double s = 0.0;
    for (int k = (c / NB) * NB + lane; k < n; k += 64) s = fma(v[k], z[k], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) alpha[c] = c < n ? s : 0.0;
}

// A[i][j] = alpha_i alpha_j - sum_k V[i][k] V[j][k]   (V[c][k] = (L^-1)[k][c], zero for k < c)
__global__ __launch_bounds__(256, 2) void kinv_tile_kernel(const double* __restrict__ V, int ldv, int kend,
                                                           const double* __restrict__ alpha,
                                                           double* __restrict__ A, int lda) {
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    int ta, tb;
    tri_tile(blockIdx.x, ta, tb);
    Acc acc;
    acc_zero(acc);
    gemm_nt_128<false>(V + (size_t)ta * NB * ldv, ldv, V + (size_t)tb * NB * ldv, ldv, ta * NB, kend, acc, smem);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ta * NB + acc_row(tm, r);
            const double ai = alpha[i];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const int j = tb * NB + acc_col(tn);
                A[(size_t)i * lda + j] = ai * alpha[j] - acc.t[tm][tn][r];
            }
        }
}

// sum over the workgroup of up to RD per-thread values; result to out[idx(d)] by threads 0 .. cnt-1
__device__ __forceinline__ void block_sum_store(const double (&v)[RD], int cnt, double* sred, double* __restrict__ out,
                                                size_t stride, size_t off) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();   // sred free
#pragma unroll
    for (int d = 0; d < RD; ++d) {
        double s = v[d];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) sred[wave * RD + d] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
        const int d = threadIdx.x;
        out[(size_t)d * stride + off] = (sred[d] + sred[RD + d]) + (sred[2 * RD + d] + sred[3 * RD + d]);
    }
}

// part[p][tile] = sum over the tile's pairs (i, j < n) of  w A_ij dK_ij/dtheta_p,  w = 2 off the
// block diagonal (only lower tiles are visited).  theta layout: [log amp, log m_d ..., (log a,
// log b for the Fabolas kernel), log sigma^2].
template <int KIND>
__global__ __launch_bounds__(256) void grad_reduce_kernel(const double* __restrict__ Xs, const double* __restrict__ A,
                                                          int lda, int n, CovParams cp, double* __restrict__ part,
                                                          int ntiles) {
    __shared__ double sI[RD * RLD];
    __shared__ double sJ[RD * RLD];
    __shared__ double sred[4 * RD];
    const bool fab = KIND == ROBO_KERNEL_FABOLAS;
    int bi, bj;
    tri_tile(blockIdx.x, bi, bj);
    const long long i0 = (long long)bi * RT, j0 = (long long)bj * RT;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, dim = cp.dim;
    const int P = fab ? dim + 3 : dim + 2;
    // pass 1: scaled squared distances (product of unit Matern factors for the Fabolas kernel)
    double acc[4][4], uu[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cov_init<double, KIND>(cp, acc[a][b], uu[a][b]);
    for (int d0 = 0; d0 < dim; d0 += RD) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + e * 256, row = idx >> 4, d = idx & 15;
            const bool ok = d0 + d < dim;
            sI[d * RLD + row] = ok ? Xs[(i0 + row) * dim + d0 + d] : 0.0;
            sJ[d * RLD + row] = ok ? Xs[(j0 + row) * dim + d0 + d] : 0.0;
        }
        __syncthreads();
        const int dn = dim - d0 < RD ? dim - d0 : RD;
        for (int d = 0; d < dn; ++d) {
            double xi[4], xj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                xi[a] = sI[d * RLD + ty * 4 + a];
                xj[a] = sJ[d * RLD + tx * 4 + a];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) cov_step<double, KIND>(cp, d0 + d, xi[a], xj[b], acc[a][b], uu[a][b]);
        }
    }
    // weights: e_ij = w A_ij (amp f'(r^2))  [stationary kernels]  or  w A_ij K_ij  [Fabolas]
    const double w = bi == bj ? 1.0 : 2.0;
    double e[4][4];
    double sc[RD];
#pragma unroll
    for (int d = 0; d < RD; ++d) sc[d] = 0.0;   // 0: amp, 1: noise, 2: log a, 3: log b
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const long long gi = i0 + ty * 4 + a, gj = j0 + tx * 4 + b;
            const bool ok = gi < n && gj < n;
            const double aij = ok ? A[(size_t)gi * lda + gj] : 0.0;
            const double c = w * aij;
            const double k = cov_finish<double, KIND>(cp, acc[a][b], uu[a][b]);
            sc[0] += c * k;
            if (gi == gj) sc[1] += aij;
            if (KIND == ROBO_KERNEL_MATERN52_ARD) {
                const double s = sqrt(5.0 * acc[a][b]);
                e[a][b] = c * (-cp.amp * (5.0 / 6.0) * (1.0 + s) * exp(-s));
            } else if (KIND == ROBO_KERNEL_RBF_ARD) {
                e[a][b] = c * (-0.5 * k);
            } else {
                const double B = cp.blr_a + cp.blr_b * uu[a][b];
                e[a][b] = c * k;
                sc[2] += e[a][b] * cp.blr_a / B;
                sc[3] += e[a][b] * cp.blr_b * uu[a][b] / B;
            }
        }
    // pass 2: per-dimension sums, RD dimensions at a time
    for (int d0 = 0; d0 < dim; d0 += RD) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + q * 256, row = idx >> 4, d = idx & 15;
            const bool ok = d0 + d < dim;
            sI[d * RLD + row] = ok ? Xs[(i0 + row) * dim + d0 + d] : 0.0;
            sJ[d * RLD + row] = ok ? Xs[(j0 + row) * dim + d0 + d] : 0.0;
        }
        __syncthreads();
        double gd[RD];
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            double xi[4], xj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                xi[a] = sI[d * RLD + ty * 4 + a];
                xj[a] = sJ[d * RLD + tx * 4 + a];
            }
            double g = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double df = xi[a] - xj[b], s = df * df;
                    if (fab) {
                        const double r = sqrt(5.0 * s);
                        g += e[a][b] * ((5.0 / 6.0) * (1.0 + r) * s / (1.0 + r + 5.0 * s / 3.0));
                    } else {
                        g -= e[a][b] * s;   // d r^2 / d log m_d = -(x_d - x'_d)^2 / m_d
                    }
                }
            gd[d] = g;
        }
        // metric parameters: theta index 1 + d for d < (fab ? dim - 1 : dim)
        const int n_metric = fab ? dim - 1 : dim;
        int cnt = n_metric - d0;
        cnt = cnt < 0 ? 0 : (cnt > RD ? RD : cnt);
        block_sum_store(gd, cnt, sred, part + (size_t)(1 + d0) * ntiles, (size_t)ntiles, blockIdx.x);
    }
    // scalars: amp -> 0, noise -> P - 1, Fabolas log a / log b -> dim, dim + 1
    __syncthreads();
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            double s = sc[d];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) sred[wave * RD + d] = s;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            const int d = threadIdx.x;
            const double s = (sred[d] + sred[RD + d]) + (sred[2 * RD + d] + sred[3 * RD + d]);
            int p = -1;
            if (d == 0) p = 0;
            else if (d == 1) p = P - 1;
            else if (fab) p = dim + (d - 2);
            if (p >= 0) part[(size_t)p * ntiles + blockIdx.x] = s;
        }
    }
}

// out[p] = 0.5 sum_tiles part[p][tile]   (fixed order: strided per thread, then a tree)
__global__ __launch_bounds__(256) void grad_final_kernel(const double* __restrict__ part, int ntiles,
                                                         double* __restrict__ out, double* __restrict__ out_host) {
    __shared__ double s[256];
    const double* row = part + (size_t)blockIdx.x * ntiles;
    double a = 0.0;
    for (int i = threadIdx.x; i < ntiles; i += 256) a += row[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[blockIdx.x] = 0.5 * s[0];
        if (out_host) out_host[blockIdx.x] = 0.5 * s[0];   // pinned, device-visible: no copy launch for P doubles
    }
}

// d_V, d_A: (n_pad x n_pad) workspaces (d_A doubles as W during the inversion); d_alpha: n_pad;
// d_part: P x tiles64; d_out: P.  Asynchronous on the context's stream.
// W = L^-1 (lower block triangle of d_W; its strictly upper blocks hold scratch) and V = W^T of the current factor
int launch_triinv(robo_gp* gp, double* d_W, double* d_V) {
    hipStream_t st = gp->ctx->stream;
    const int n = gp->n, n_pad = gp->n_pad, nbk = (n + NB - 1) / NB;
    hipLaunchKernelGGL(triinv_base_kernel, dim3(nbk), dim3(256), 0, st, (const double*)gp->d_Linv, n, d_W, d_V, n_pad);
    for (int sb = 1; sb < nbk; sb *= 2) {
        const unsigned pairs = (unsigned)((nbk + 2 * sb - 1) / (2 * sb));
        const bool narrow = (long long)sb * sb * pairs <= 2LL * gp->ctx->num_cu;   // too few 128-row tiles to fill the chip
        for (int phase = 0; phase < 2; ++phase) {
            if (narrow)
                hipLaunchKernelGGL(triinv_kernel<1>, dim3((unsigned)(sb * sb * 4), pairs), dim3(256), 0, st, phase, sb, nbk,
                                   n, (const double*)gp->d_K, d_W, d_V, n_pad);
            else
                hipLaunchKernelGGL(triinv_kernel<4>, dim3((unsigned)(sb * sb), pairs), dim3(256), 0, st, phase, sb, nbk, n,
                                   (const double*)gp->d_K, d_W, d_V, n_pad);
        }
    }
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_grad_loglik(robo_gp* gp, double* d_V, double* d_A, double* d_alpha, double* d_part, double* d_out,
                       double* h_out) {
    hipStream_t st = gp->ctx->stream;
    const int n = gp->n, n_pad = gp->n_pad, nbk = (n + NB - 1) / NB;
    double* d_W = d_A;   // dead before kinv_tile_kernel writes A
    {
        const int s = launch_triinv(gp, d_W, d_V);
        if (s != ROBO_OK) return s;
    }
    hipLaunchKernelGGL(alpha_kernel, dim3((nbk * NB + 3) / 4), dim3(256), 0, st, (const double*)d_V, n_pad,
                       (const double*)(gp->d_K + (size_t)n * n_pad), n, nbk * NB, d_alpha);
    hipLaunchKernelGGL(kinv_tile_kernel, dim3(nbk * (nbk + 1) / 2), dim3(256), 0, st, (const double*)d_V, n_pad,
                       nbk * NB, (const double*)d_alpha, d_A, n_pad);
    const int t64 = (n + RT - 1) / RT, ntiles = t64 * (t64 + 1) / 2;
    const int P = gp->kind == ROBO_KERNEL_FABOLAS ? gp->dim + 3 : gp->dim + 2;
#define ROBO_GRAD_CALL(KIND)                                                                               \
    hipLaunchKernelGGL(grad_reduce_kernel<KIND>, dim3(ntiles), dim3(256), 0, st, (const double*)gp->d_Xs, \
                       (const double*)d_A, n_pad, n, gp->cov, d_part, ntiles)
    if (gp->kind == ROBO_KERNEL_MATERN52_ARD) ROBO_GRAD_CALL(ROBO_KERNEL_MATERN52_ARD);
    else if (gp->kind == ROBO_KERNEL_RBF_ARD) ROBO_GRAD_CALL(ROBO_KERNEL_RBF_ARD);
    else ROBO_GRAD_CALL(ROBO_KERNEL_FABOLAS);
#undef ROBO_GRAD_CALL
    hipLaunchKernelGGL(grad_final_kernel, dim3(P), dim3(256), 0, st, (const double*)d_part, ntiles, d_out, h_out);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
