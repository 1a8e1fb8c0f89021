// The hyper-parameter chain of GaussianProcessMCMC, resident on the device.
//
// Replaces the loop the reference drives in robo/models/gaussian_process_mcmc.py:114-142
//     sampler = emcee.EnsembleSampler(n_hypers, ndim, self.loglikelihood);  sampler.run_mcmc(p0, n_steps, rstate0=rng)
// i.e. emcee 2's EnsembleSampler._propose_stretch (Goodman & Weare stretch move) around one george fit per walker
// (:168-202).  robo_gp_loglik_batch already evaluates half an ensemble per call; what stayed on the host was the
// proposal, the prior, the accept test and -- per half-step -- one upload, one synchronisation and ~100 us of Python.
// At a Bayesian-optimisation-typical N = 200 that host share is more than the device's (r03w: 191 us per half-step,
// ~85 of them on the device).  Here the whole chain is ONE sequence of launches on the library's stream:
//   mcmc_propose_kernel   q_w = c_j - z (c_j - s_w) for the walkers of the active half, the reference's bounds
//                         protocol (any |theta_p| > 20 -> -inf, gaussian_process_mcmc.py:185-190), log prior, and the
//                         batch fit's inputs (FitSample + inverse square-root metrics) straight into its device buffers
//   scale / gram / potrf  the batched fit of gram.hip / potrf.hip, unchanged (likelihood terms stay on the device)
//   mcmc_accept_kernel    lnp(q) = log-likelihood + log prior, emcee's accept test  (ndim - 1) log z + lnp(q) - lnp(s) >
//                         log u, walker / log-probability / acceptance-count update, chain record after the second half
// The random numbers do not depend on the chain, so the caller draws them up front in emcee 2's order (per half-step:
// rand for z, randint for the partners, rand for the accept test) -- same stream, same chain as the reference's sampler
// up to the rounding of exp / log on the device.  q is formed without fused multiply-adds (numpy has none).
#include "common.h"

namespace robo {

// robo/priors/default_priors.py:7-37 through robo_amd/priors/priors.py: lognormal on theta[0], tophat on the length
// scales, horseshoe on the noise; par = {lognormal loc, sigma, tophat min, max, horseshoe scale}
__device__ __forceinline__ double default_prior_lnprob(const double* th, int P, const double* par) {
    const double ninf = -__builtin_huge_val();
    const double yv = th[0] - par[0];
    double lp;
    if (yv > 0.0) {
        const double ly = log(yv);
        lp = -(ly * ly) / (2.0 * par[1] * par[1]) - ly - log(par[1] * sqrt(2.0 * M_PI));
    } else {
        lp = ninf;
    }
    for (int p = 1; p < P - 1; ++p)
        if (th[p] < par[2] || th[p] > par[3]) lp = ninf;
    const double noise = th[P - 1];
    const double r = par[4] / exp(noise);
    double hs = log(log(1.0 + 3.0 * (r * r)));
    if (noise == 0.0) hs = __builtin_huge_val();
    return lp + hs;
}

// theta (P) -> the batch fit's per-sample inputs (api.hip theta_to_sample); !ok: the unit kernel (theta = 0), which keeps
// the slot of a rejected-by-bounds proposal numerically harmless
__device__ __forceinline__ void theta_to_sample_dev(const double* q, bool ok, int kind, int D, int P, double mean_c,
                                                    FitSample* sp, double* ism) {
    const bool fab = kind == ROBO_KERNEL_FABOLAS;
    const int n_metric = fab ? D - 1 : D;
    for (int d = 0; d < n_metric; ++d) ism[d] = exp(-0.5 * (ok ? q[1 + d] : 0.0));
    if (fab) ism[D - 1] = 1.0;
    sp->cov.kind = kind;
    sp->cov.dim = D;
    sp->cov.amp = exp(ok ? q[0] : 0.0);
    sp->cov.blr_a = fab ? exp(ok ? q[D] : 0.0) : 0.0;
    sp->cov.blr_b = fab ? exp(ok ? q[D + 1] : 0.0) : 0.0;
    sp->noise = exp(ok ? q[P - 1] : 0.0) + JITTER;
    sp->mean_c = mean_c;
}

// start == 1: the walkers themselves (first evaluation of the start positions), `first`..`first + ns`
// start == 0: stretch-move proposals of half `h` at step *it
__global__ __launch_bounds__(256) void mcmc_propose_kernel(McmcState st, int start, int first, int h) {
    const int ns = start ? st.ns_eval : st.k / 2;
    const int it = start ? 0 : *st.d_it;
    for (int w = threadIdx.x; w < ns; w += blockDim.x) {
        double* q = st.d_q + (size_t)w * st.P;
        double z = 1.0;
        if (start) {
            for (int p = 0; p < st.P; ++p) q[p] = st.d_pos[(size_t)(first + w) * st.P + p];
        } else {
            const int half = st.k / 2;
            const size_t r = ((size_t)it * 2 + h) * half + w;
            const double* s = st.d_pos + (size_t)(h * half + w) * st.P;
            const double* c = st.d_pos + (size_t)((1 - h) * half + st.d_partner[r]) * st.P;
            const double t = __dadd_rn(__dmul_rn(st.a - 1.0, st.d_uz[r]), 1.0);
            z = __ddiv_rn(__dmul_rn(t, t), st.a);
            for (int p = 0; p < st.P; ++p) q[p] = __dsub_rn(c[p], __dmul_rn(z, __dsub_rn(c[p], s[p])));
        }
        st.d_z[w] = z;
        bool ok = true;
        for (int p = 0; p < st.P; ++p) ok = ok && (q[p] >= -20.0 && q[p] <= 20.0);     // also false for NaN / inf
        double prior = 0.0;
        if (ok && st.prior_kind == 1) prior = default_prior_lnprob(q, st.P, st.prior_par);
        st.d_prior[w] = ok ? prior : -__builtin_huge_val();
        theta_to_sample_dev(q, ok, st.kind, st.D, st.P, st.mean_c, st.d_sp + w, st.d_ism + (size_t)w * st.D);
    }
}

__global__ __launch_bounds__(256) void mcmc_accept_kernel(McmcState st, int start, int first, int h) {
    const int ns = start ? st.ns_eval : st.k / 2;
    const int it = start ? 0 : *st.d_it;
    const int half = st.k / 2;
    const double cst = (double)st.n * log(2.0 * M_PI);
    for (int w = threadIdx.x; w < ns; w += blockDim.x) {
        double lp = st.d_prior[w];
        if (lp > -__builtin_huge_val()) {          // (a +inf prior stays +inf unless the fit fails, as on the host)
            const double ll = st.d_fail[w] != 0 ? -__builtin_huge_val()
                                                : -0.5 * (st.d_out[2 * w] + st.d_out[2 * w + 1] + cst);
            lp = ll + lp;
        }
        if (lp != lp) atomicOr(st.d_err, 1);       // emcee: "lnprob returned NaN."
        if (start) {
            if (lp == __builtin_huge_val()) atomicOr(st.d_err, 2);   // "The initial lnprob was +inf."
            st.d_lnp[first + w] = lp;
            continue;
        }
        const int sw = h * half + w;
        const size_t r = ((size_t)it * 2 + h) * half + w;
        const double lnpdiff = ((double)st.P - 1.0) * log(st.d_z[w]) + lp - st.d_lnp[sw];
        if (lnpdiff > log(st.d_ua[r])) {
            const double* q = st.d_q + (size_t)w * st.P;
            for (int p = 0; p < st.P; ++p) st.d_pos[(size_t)sw * st.P + p] = q[p];
            st.d_lnp[sw] = lp;
            st.d_nacc[sw] += 1;
        }
    }
    if (start || h == 0) return;
    __syncthreads();
    // end of ensemble step `it`: record the chain, advance the step counter
    if (st.d_chain)
        for (int i = threadIdx.x; i < st.k * st.P; i += blockDim.x) {
            const int w = i / st.P, p = i - w * st.P;
            st.d_chain[((size_t)w * st.n_steps + it) * st.P + p] = st.d_pos[i];
        }
    if (st.d_lnprob)
        for (int w = threadIdx.x; w < st.k; w += blockDim.x) st.d_lnprob[(size_t)w * st.n_steps + it] = st.d_lnp[w];
    __syncthreads();
    if (threadIdx.x == 0) *st.d_it = it + 1;
}

int launch_mcmc_propose(robo_ctx* ctx, const McmcState& st, int start, int first, int h) {
    hipLaunchKernelGGL(mcmc_propose_kernel, dim3(1), dim3(256), 0, ctx->stream, st, start, first, h);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_mcmc_accept(robo_ctx* ctx, const McmcState& st, int start, int first, int h) {
    hipLaunchKernelGGL(mcmc_accept_kernel, dim3(1), dim3(256), 0, ctx->stream, st, start, first, h);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
