// Gradients of the predictive mean and variance with respect to the test inputs.
//
// The reference's acquisition derivatives (robo/acquisition_functions/ei.py:80-85, pi.py:65-71, lcb.py:66-68)
// and its gradient-based incumbent search (robo/util/posterior_optimization.py:38-40,96-104) call
// ``model.predictive_gradients(X)``, which NO model in the reference tree implements; this is the device side of
// robo_amd's GaussianProcess.predictive_gradients.
//
// With v = L^-1 k_*(x) and z = L^-1 (y - mean):   mu = v . z + mean,   var = k(x,x) - |v|^2, hence
//     d mu / d x_d  = w_d . z,        d var / d x_d = d k(x,x) / d x_d - 2 v . w_d,       w_d = L^-1 (d k_* / d x_d).
// The D derivative vectors d k_*/d x_d are therefore D more right-hand sides of the SAME blocked forward
// substitution the posterior uses: every candidate becomes D + 1 consecutive rows of the solve workspace
// [k_*, d k_*/d x_1, ..., d k_*/d x_D], the MFMA block-row kernel (predict.hip) solves them all, its epilogue
// already delivers w_d . z (the "mu" reduction of row d) and |v|^2, and one dot product per (candidate, d) of rows
// that are still resident gives v . w_d.  No back-substitution, no K^-1.
#include "common.h"
#include "kern_math.h"

namespace robo {

// V[(c (D+1) + e) * ldv + j]:  e = 0: k(x_c, x_j);  e = 1 + d: d k(x_c, x_j) / d xs_{c,d}  (xs = scaled coordinate);
// columns j >= n (augmented row, padding) and pseudo-rows of candidates >= m are zero.
__global__ __launch_bounds__(256) void cross_grad_kernel(const double* __restrict__ Xcs, const double* __restrict__ Xs,
                                                         double* __restrict__ V, long long c_first, long long c_count,
                                                         long long rows_pad, int n, int ldv, CovParams cp) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const long long cl = blockIdx.y;                 // candidate within this batch
    const int D = cp.dim, E = D + 1;
    if (j >= ldv) return;
    if (cl >= c_count) {                              // tail rows up to the 128-row padding
        const long long row = c_count * E + (cl - c_count);
        if (row < rows_pad) V[(size_t)row * ldv + j] = 0.0;
        return;
    }
    double* out = V + (size_t)cl * E * ldv + j;
    if (j >= n) {
        for (int e = 0; e < E; ++e) out[(size_t)e * ldv] = 0.0;
        return;
    }
    const double* xi = Xcs + (size_t)(c_first + cl) * D;
    const double* xj = Xs + (size_t)j * D;
    if (cp.kind == ROBO_KERNEL_FABOLAS) {
        // k = amp prod_d m52(df_d^2) (a + b u u');  d/d xs_d = k (m52'/m52)(df_d^2) 2 df_d;  d/d u = amp prod b u'
        double prod = 1.0;
        for (int d = 0; d < D - 1; ++d) {
            const double df = xi[d] - xj[d];
            prod *= matern52_1d(df);
        }
        const double u = xi[D - 1], up = xj[D - 1];
        const double lin = cp.blr_a + cp.blr_b * u * up;
        const double k = cp.amp * prod * lin;
        out[0] = k;
        for (int d = 0; d < D - 1; ++d) {
            const double df = xi[d] - xj[d], s2 = df * df, t = sqrt(5.0 * s2);
            // m52'(s2) / m52(s2) = -(5/6) (1 + t) / (1 + t + 5 s2 / 3)
            out[(size_t)(1 + d) * ldv] = k * (-(5.0 / 6.0) * (1.0 + t) / (1.0 + t + 5.0 * s2 / 3.0)) * 2.0 * df;
        }
        out[(size_t)D * ldv] = cp.amp * prod * cp.blr_b * up;
        return;
    }
    double r2 = 0.0;
    for (int d = 0; d < D; ++d) {
        const double df = xi[d] - xj[d];
        r2 = fma(df, df, r2);
    }
    double k, dk;   // k = amp f(r2), dk = amp f'(r2)
    if (cp.kind == ROBO_KERNEL_MATERN52_ARD) {
        const double t = sqrt(5.0 * r2), e = exp(-t);
        k = cp.amp * (1.0 + t + 5.0 * r2 / 3.0) * e;
        dk = -cp.amp * (5.0 / 6.0) * (1.0 + t) * e;
    } else {
        k = cp.amp * exp(-0.5 * r2);
        dk = -0.5 * k;
    }
    out[0] = k;
    for (int d = 0; d < D; ++d) out[(size_t)(1 + d) * ldv] = dk * 2.0 * (xi[d] - xj[d]);
}

// one workgroup per candidate: mean, var, d mean / d x, d var / d x in the GP's (normalised) input space with the
// output transform applied.  q / mu are the block-row kernel's reductions per pseudo-row.
__global__ __launch_bounds__(256) void predgrad_post_kernel(const double* __restrict__ V, int ldv, int ncols,
                                                            const double* __restrict__ q, const double* __restrict__ mu,
                                                            const double* __restrict__ Xcs,
                                                            const double* __restrict__ inv_sqrt_metric,
                                                            long long c_first, CovParams cp, double mean_c, double y_mean,
                                                            double y_std, double* __restrict__ mean,
                                                            double* __restrict__ var, double* __restrict__ dmean,
                                                            double* __restrict__ dvar) {
    __shared__ double red[4];
    const long long cl = blockIdx.x, c = c_first + cl;
    const int D = cp.dim, E = D + 1;
    const double* v0 = V + (size_t)cl * E * ldv;
    const double u = Xcs[(size_t)c * D + D - 1];
    if (threadIdx.x == 0) {
        double m = mu[cl * E] + mean_c;
        double v = cov_self(cp, u) - q[cl * E];
        m = m * y_std + y_mean;
        v = v * (y_std * y_std);
        const double eps = 2.220446049250313e-16;
        mean[c] = m;
        var[c] = v < eps ? eps : v;       // the reference's floor (gaussian_process.py:290-294); the gradient is the
    }                                     // unclipped function's
    for (int d = 0; d < D; ++d) {
        const double* vd = v0 + (size_t)(1 + d) * ldv;
        double s = 0.0;
        for (int j = threadIdx.x; j < ncols; j += 256) s = fma(v0[j], vd[j], s);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double dot = (red[0] + red[1]) + (red[2] + red[3]);
            // d k(x,x) / d xs_d: 0 for the stationary kernels; Fabolas: 2 amp b u on the fidelity column
            const double dself = (cp.kind == ROBO_KERNEL_FABOLAS && d == D - 1) ? 2.0 * cp.amp * cp.blr_b * u : 0.0;
            const double ism = inv_sqrt_metric[d];      // d xs_d / d x_d
            dmean[(size_t)c * D + d] = mu[cl * E + 1 + d] * ism * y_std;
            dvar[(size_t)c * D + d] = (dself - 2.0 * dot) * ism * (y_std * y_std);
        }
        __syncthreads();
    }
}

int launch_cross_grad(robo_gp* gp, const double* d_Xcs, double* d_V, int64_t c_first, int64_t c_count, int64_t rows_pad) {
    const int E = gp->dim + 1;
    const long long tail = rows_pad - c_count * E;      // zero rows up to the padding
    const dim3 grid((unsigned)((gp->n_pad + 255) / 256), (unsigned)(c_count + tail));
    hipLaunchKernelGGL(cross_grad_kernel, grid, dim3(256), 0, gp->ctx->stream, d_Xcs, (const double*)gp->d_Xs, d_V,
                       (long long)c_first, (long long)c_count, (long long)rows_pad, gp->n, gp->n_pad, gp->cov);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_predgrad_post(robo_gp* gp, const double* d_V, const double* d_q, const double* d_mu, const double* d_Xcs,
                         int64_t c_first, int64_t c_count, double* d_mean, double* d_var, double* d_dmean,
                         double* d_dvar) {
    const int ncols = (gp->n + NB - 1) / NB * NB;        // the block rows the solve touched
    hipLaunchKernelGGL(predgrad_post_kernel, dim3((unsigned)c_count), dim3(256), 0, gp->ctx->stream, d_V, gp->n_pad, ncols,
                       d_q, d_mu, d_Xcs, (const double*)gp->d_theta, (long long)c_first, gp->cov, gp->mean_c, gp->y_mean,
                       gp->y_std, d_mean, d_var, d_dmean, d_dvar);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
