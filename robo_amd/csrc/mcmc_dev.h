// Device-side pieces of the hyper-parameter chain shared by mcmc.hip and the fused one-block step (potrf.hip).
#pragma once
#include "common.h"

namespace robo {

// The priors the library evaluates itself (robo_gp_mcmc_run's prior_kind):
//   1  robo/priors/default_priors.py:7-37 (DefaultPrior): lognormal on theta[0], tophat on theta[1 .. P-2], horseshoe on
//      the noise; par = {lognormal loc, sigma, tophat min, max, horseshoe scale}
//   2  robo/priors/env_priors.py:8-54 (EnvPrior, the Fabolas kernels' prior, wired at robo/fmin/fabolas.py:120-127): the
//      tophat covers only the n_ls length scales theta[1 .. n_ls]; the n_lr parameters of the Bayesian-linear-regression
//      kernel behind them each ADD NormalPrior.lnprob -- which is a pdf, not a log-pdf (robo/priors/base_prior.py:357,
//      mirrored: SURVEY A.3 #10) -- in the reference's order of summation; par = {.. the five above .., n_ls, n_lr,
//      normal mean, normal sigma}
constexpr int PRIOR_PAR = 9;
__device__ __forceinline__ double prior_lnprob(int kind, const double* th, int P, const double* par) {
    const double ninf = -__builtin_huge_val();
    const double yv = th[0] - par[0];
    double lp;
    if (yv > 0.0) {
        const double ly = log(yv);
        lp = -(ly * ly) / (2.0 * par[1] * par[1]) - ly - log(par[1] * sqrt(2.0 * M_PI));
    } else {
        lp = ninf;
    }
    const int ls_end = kind == 2 ? 1 + (int)par[5] : P - 1;
    for (int p = 1; p < ls_end; ++p)
        if (th[p] < par[2] || th[p] > par[3]) lp = ninf;
    if (kind == 2) {
        const int lr_end = ls_end + (int)par[6];
        for (int p = ls_end; p < lr_end && p < P - 1; ++p) {
            const double u = (th[p] - par[7]) / par[8];
            lp += exp(-0.5 * (u * u)) / (par[8] * sqrt(2.0 * M_PI));
        }
    }
    const double noise = th[P - 1];
    const double r = par[4] / exp(noise);
    double hs = log(log(1.0 + 3.0 * (r * r)));
    if (noise == 0.0) hs = __builtin_huge_val();
    return lp + hs;
}

// emcee 2's stretch move (EnsembleSampler._propose_stretch), operation for operation and unfused:
//     zz = ((a - 1.) * rand + 1) ** 2. / a            (NumPy's x ** 2. is x * x)
//     q  = c - zz * (c - s)
__device__ __forceinline__ double mcmc_stretch_z(double a, double u) {
    const double t = rn_add(rn_mul(rn_sub(a, 1.0), u), 1.0);
    return rn_div(rn_mul(t, t), a);
}
__device__ __forceinline__ double mcmc_stretch_q(double c, double s, double z) { return rn_sub(c, rn_mul(z, rn_sub(c, s))); }

// The proposal of walker w by the whole block (thread p <-> parameter p): q into sq[0 .. P), z into *sz, the inverse
// square-root metrics of q (of the unit kernel when q violates the reference's |theta| <= 20 bound) into sism[0 .. D).
// Returns that bound test.  Contains barriers: every thread of the block calls it.  z and q are formed from
// rn_* operations (common.h): no fused multiply-add, as in NumPy / emcee  zz = ((a - 1) u + 1)**2 / a,  q = c - zz (c - s).
// start == 1: the walker itself (first evaluation of the start positions), walker `first + w`
// start == 0: stretch move of walker w of half h at step it
__device__ __forceinline__ bool mcmc_block_proposal(const McmcState& st, int start, int first, int h, int it, int w,
                                                    double* sq, double* sism, double* sz, int* sbad) {
    const int P = st.P, D = st.D, half = st.k / 2;
    if (threadIdx.x == 0) {
        double z = 1.0;
        if (!start) {
            const size_t r = ((size_t)it * 2 + h) * half + w;
            z = mcmc_stretch_z(st.a, st.d_uz[r]);
        }
        *sz = z;
        *sbad = 0;
    }
    __syncthreads();
    bool bad = false;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        double q;
        if (start) {
            q = st.d_pos[(size_t)(first + w) * P + p];
        } else {
            const size_t r = ((size_t)it * 2 + h) * half + w;
            const double s = st.d_pos[(size_t)(h * half + w) * P + p];
            const double c = st.d_pos[(size_t)((1 - h) * half + st.d_partner[r]) * P + p];
            q = mcmc_stretch_q(c, s, *sz);
        }
        sq[p] = q;
        bad = bad || !(q >= -20.0 && q <= 20.0);          // also true for NaN / inf
    }
    if (bad) *sbad = 1;
    __syncthreads();                                       // sq is complete
    const bool ok = *sbad == 0;
    const int n_metric = st.kind == ROBO_KERNEL_FABOLAS ? D - 1 : D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) sism[d] = d < n_metric ? exp(-0.5 * (ok ? sq[1 + d] : 0.0)) : 1.0;
    __syncthreads();
    return ok;
}

// the batch fit's per-sample inputs of proposal q (api.hip theta_to_sample)
__device__ __forceinline__ FitSample mcmc_fit_sample(const McmcState& st, const double* sq, bool ok) {
    const bool fab = st.kind == ROBO_KERNEL_FABOLAS;
    FitSample sp;
    sp.cov.kind = st.kind;
    sp.cov.dim = st.D;
    sp.cov.amp = exp(ok ? sq[0] : 0.0);
    sp.cov.blr_a = fab ? exp(ok ? sq[st.D] : 0.0) : 0.0;
    sp.cov.blr_b = fab ? exp(ok ? sq[st.D + 1] : 0.0) : 0.0;
    sp.noise = exp(ok ? sq[st.P - 1] : 0.0) + JITTER;
    sp.mean_c = st.mean_c;
    return sp;
}

// log-probability of a proposal from its prior term and the fit's (z.z, log det, failure flag); the reference's protocol
__device__ __forceinline__ double mcmc_lnprob(double prior, int fail, double quad, double logdet, int n) {
    double lp = prior;
    if (lp > -__builtin_huge_val()) {          // (a +inf prior stays +inf unless the fit fails, as on the host)
        const double ll = fail != 0 ? -__builtin_huge_val() : -0.5 * (quad + logdet + (double)n * log(2.0 * M_PI));
        lp = ll + lp;
    }
    return lp;
}

// emcee 2's accept statistic  lnpdiff = (ndim - 1) log z + lnp(q) - lnp(s)  in its order of operations, unfused
__device__ __forceinline__ double mcmc_lnpdiff(int P, double log_z, double lp_new, double lp_old) {
    return rn_sub(rn_add(rn_mul((double)P - 1.0, log_z), lp_new), lp_old);
}

}  // namespace robo
