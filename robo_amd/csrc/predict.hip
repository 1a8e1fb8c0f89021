// K5 batched predictive variance/mean: blocked forward substitution V = L^-1 K_*^T for a
// chunk of candidates, in place in the cross-gram workspace, plus the per-candidate
// reductions  q_c = |v_c|^2  and  mu_c = v_c . z  (z = L^-1 (y - mean), row n of L).
//
// Replaces george.GP.predict's  Kxs K^-1 Kxs^T  and  Kxs alpha
// (reference call site robo/models/gaussian_process.py:280-286) -- diagonal only:
//     var_c  = k(x_c, x_c) - q_c          mean_c = mu_c + mean
// (the reference builds the full M x M covariance and takes np.diag; the values are the
// same, the O(M^2) work is not done.)
//
// Layout: V is (chunk, n_pad) row-major, candidate-major ("Vt"), so every product is the
// NT form of gemm_f64.h with the candidates as the 128 output rows:
//   step i:  T   = V[:, blk i] - V[:, 0:i*128] * L[blk i, 0:i*128]^T      (K = i*128)
//            V_i = T * Linv_i^T                                           (K = 128)
// One launch per block row i (a launch boundary is the only inter-workgroup ordering the
// algorithm needs: step i reads columns < i*128 written by the SAME workgroup earlier).
// Flops per candidate: n_pad^2 + n_pad*128 MFMA flops.
#include "common.h"
#include "gemm_f64.h"
#include "kern_math.h"

namespace robo {

__global__ __launch_bounds__(256) void trsm_step_kernel(double* __restrict__ V, int ldv, const double* __restrict__ L,
                                                        int ld, const double* __restrict__ Linv, int i, int n,
                                                        double* __restrict__ q, double* __restrict__ mu,
                                                        long long c0) {
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    double* Vrow = V + (size_t)blockIdx.x * NB * ldv;
    double* Vt = Vrow + (size_t)i * NB;   // the tile being solved
    Acc acc;
    if (i > 0) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc.t[tm][tn][r] = Vt[(size_t)acc_row(tm, r) * ldv + acc_col(tn)];
        gemm_nt_128<true>(Vrow, ldv, L + (size_t)i * NB * ld, ld, 0, i * NB, acc, smem);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) Vt[(size_t)acc_row(tm, r) * ldv + acc_col(tn)] = acc.t[tm][tn][r];
        __syncthreads();   // T visible to the whole workgroup (global memory, workgroup scope)
    }
    acc_zero(acc);
    gemm_nt_128<false>(Vt, ldv, Linv + (size_t)i * NB * NB, NB, 0, NB, acc, smem);

    // epilogue: store V_i (columns >= n zeroed: augmented row + padding), reduce |v|^2 and v.z
    const int lane = threadIdx.x & 63, wx = (threadIdx.x >> 6) & 1;
    const double* z = L + (size_t)n * ld;
    double zc[4];
    bool live[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int gn = i * NB + acc_col(tn);
        live[tn] = gn < n;
        zc[tn] = live[tn] ? z[gn] : 0.0;
    }
    double* red = smem;   // [2][2][128]
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double sq = 0.0, sz = 0.0;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const double v = live[tn] ? acc.t[tm][tn][r] : 0.0;
                Vt[(size_t)acc_row(tm, r) * ldv + acc_col(tn)] = v;
                sq = fma(v, v, sq);
                sz = fma(v, zc[tn], sz);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                sq += __shfl_xor(sq, o);
                sz += __shfl_xor(sz, o);
            }
            if ((lane & 15) == 0) {
                red[(0 * 2 + wx) * NB + acc_row(tm, r)] = sq;
                red[(1 * 2 + wx) * NB + acc_row(tm, r)] = sz;
            }
        }
    __syncthreads();
    if (threadIdx.x < NB) {
        const long long c = c0 + (long long)blockIdx.x * NB + threadIdx.x;
        const double sq = red[0 * NB + threadIdx.x] + red[1 * NB + threadIdx.x];
        const double sz = red[2 * NB + threadIdx.x] + red[3 * NB + threadIdx.x];
        if (i == 0) {
            q[c] = sq;
            mu[c] = sz;
        } else {
            q[c] += sq;
            mu[c] += sz;
        }
    }
}

// mean/var from the reductions, with the reference's output transform and variance floor
// (robo/models/gaussian_process.py:282-294)
__global__ __launch_bounds__(256) void post_kernel(const double* __restrict__ q, const double* __restrict__ mu,
                                                   const double* __restrict__ Xcs, double* __restrict__ mean,
                                                   double* __restrict__ var, long long c0, long long cn, CovParams cp,
                                                   double mean_c, double y_mean, double y_std) {
    const long long i = c0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c0 + cn) return;
    double m = mu[i] + mean_c;
    double v = cov_self(cp, Xcs[i * cp.dim + cp.dim - 1]) - q[i];
    m = m * y_std + y_mean;
    v = v * (y_std * y_std);
    const double eps = 2.220446049250313e-16;
    v = v < eps ? eps : v;   // np.clip(var, eps, inf); NaN propagates like np.clip
    mean[i] = m;
    var[i] = v;
}

// cov[c][c'] = (k(x_c, x_c') - v_c . v_c') * y_std^2   for c, c' < m  (small m)
__global__ __launch_bounds__(256) void cov_kernel(const double* __restrict__ V, int ldv,
                                                  const double* __restrict__ Xcs, CovParams cp, double y_std,
                                                  long long m, double* __restrict__ cov) {
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    const long long r0 = (long long)blockIdx.y * NB, q0 = (long long)blockIdx.x * NB;
    Acc acc;
    acc_zero(acc);
    gemm_nt_128<false>(V + (size_t)r0 * ldv, ldv, V + (size_t)q0 * ldv, ldv, 0, ldv, acc, smem);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long a = r0 + acc_row(tm, r), b = q0 + acc_col(tn);
                if (a < m && b < m)
                    cov[a * m + b] = (cov_rows(cp, Xcs + a * cp.dim, Xcs + b * cp.dim) - acc.t[tm][tn][r]) *
                                     (y_std * y_std);
            }
}

int launch_trsm(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn) {
    // block rows that hold training points; a trailing block holding only the augmented row and
    // identity padding (n % 128 == 0) solves to zeros and is skipped (its V columns stay the
    // zeros cross_gram_kernel wrote)
    const int nbk = (gp->n + NB - 1) / NB;
    for (int i = 0; i < nbk; ++i) {
        hipLaunchKernelGGL(trsm_step_kernel, dim3((unsigned)(cn / NB)), dim3(256), 0, gp->ctx->stream, cand->d_V,
                           gp->n_pad, (const double*)gp->d_K, gp->n_pad, (const double*)gp->d_Linv, i, gp->n,
                           cand->d_q, cand->d_mu, (long long)c0);
    }
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_post(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn) {
    hipLaunchKernelGGL(post_kernel, dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, gp->ctx->stream,
                       (const double*)cand->d_q, (const double*)cand->d_mu, (const double*)cand->d_Xcs, cand->d_mean,
                       cand->d_var, (long long)c0, (long long)cn, gp->cov, gp->mean_c, gp->y_mean, gp->y_std);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_cov(robo_gp* gp, robo_cand* cand, double* d_cov) {
    const unsigned t = (unsigned)(cand->m_pad / NB);
    hipLaunchKernelGGL(cov_kernel, dim3(t, t), dim3(256), 0, gp->ctx->stream, (const double*)cand->d_V, gp->n_pad,
                       (const double*)cand->d_Xcs, gp->cov, gp->y_std, (long long)cand->m, d_cov);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
