// The hyper-parameter chain of GaussianProcessMCMC, resident on the device.
//
// Replaces the loop the reference drives in robo/models/gaussian_process_mcmc.py:114-142
//     sampler = emcee.EnsembleSampler(n_hypers, ndim, self.loglikelihood);  sampler.run_mcmc(p0, n_steps, rstate0=rng)
// i.e. emcee 2's EnsembleSampler._propose_stretch (Goodman & Weare stretch move) around one george fit per walker
// (:168-202).  robo_gp_loglik_batch already evaluates half an ensemble per call; what stayed on the host was the
// proposal, the prior, the accept test and -- per half-step -- one upload, one synchronisation and ~100 us of Python.
// At a Bayesian-optimisation-typical N = 200 that host share is more than the device's (r03w: 191 us per half-step,
// ~85 of them on the device).  Here the whole chain is ONE sequence of launches on the library's stream:
//   mcmc_propose_scale_kernel  q_w = c_j - z (c_j - s_w) for the walkers of the active half, the reference's bounds
//                         protocol (any |theta_p| > 20 -> -inf, gaussian_process_mcmc.py:185-190), log prior, the batch
//                         fit's FitSample, and the inputs scaled by the proposal's metrics (scale_inputs_kernel's job)
//   gram / potrf          the batched fit of gram.hip / potrf.hip, unchanged (likelihood terms stay on the device)
//   mcmc_accept_kernel    lnp(q) = log-likelihood + log prior, emcee's accept test  (ndim - 1) log z + lnp(q) - lnp(s) >
//                         log u, walker / log-probability / acceptance-count update, chain record after the second half
// One-block problems (N <= 126, a Bayesian-optimisation run's own sizes) take all of that in ONE launch per half-step,
// one workgroup per walker with the gram tiles written straight into the factorisation's LDS image
// (potrf.hip: mcmc_block_step_kernel; same decisions, likelihoods within an ulp of this file's form).
// The random numbers do not depend on the chain, so the caller draws them up front in emcee 2's order (per half-step:
// rand for z, randint for the partners, rand for the accept test) -- same stream, same chain as the reference's sampler
// up to the rounding of exp / log on the device.  z and q are formed by the rn_* operations of common.h: the compiler
// may not contract them into fused multiply-adds (numpy has none; round 5's q was a v_fma_f64, tests/test_isa.py).
#include "common.h"
#include "mcmc_dev.h"

namespace robo {

// Proposal + input scaling in one launch, grid (blocks over the rows of X, walkers of the half): EVERY block forms its
// walker's proposal (thread p <-> parameter p), the bounds test and the inverse square-root metrics in LDS, then scales
// its share of X for the gram kernel (what scale_inputs_kernel does from a metrics array in memory); block 0 of each
// walker also leaves q, z, the log prior and the batch fit's FitSample behind.  (r03z: as two kernels -- one thread per
// walker for the proposal -- the proposal alone took 14.6 us of a 54 us half-step at N = 100.)
// start == 1: the walkers themselves (first evaluation of the start positions), `first`..`first + ns`
// start == 0: stretch-move proposals of half `h` at step `it`
__global__ __launch_bounds__(256) void mcmc_propose_scale_kernel(McmcState st, int start, int first, int h, int it,
                                                                 const double* __restrict__ X, double* __restrict__ Xs,
                                                                 long long rows_real, long long rows_pad,
                                                                 size_t xs_stride) {
    __shared__ double sq[MAX_DIM + 8];
    __shared__ double sism[MAX_DIM];
    __shared__ double sz;
    __shared__ int sbad;
    const int w = blockIdx.y, P = st.P, D = st.D;
    const bool ok = mcmc_block_proposal(st, start, first, h, it, w, sq, sism, &sz, &sbad);
    if (blockIdx.x == 0) {
        for (int p = threadIdx.x; p < P; p += blockDim.x) st.d_q[(size_t)w * P + p] = sq[p];
        for (int d = threadIdx.x; d < D; d += blockDim.x) st.d_ism[(size_t)w * D + d] = sism[d];
        if (threadIdx.x == 0) {
            st.d_z[w] = sz;
            double prior = 0.0;
            if (ok && st.prior_kind != 0) prior = prior_lnprob(st.prior_kind, sq, P, st.prior_par);
            st.d_prior[w] = ok ? prior : -__builtin_huge_val();
            st.d_sp[w] = mcmc_fit_sample(st, sq, ok);
        }
    }
    // scale_inputs_kernel's arithmetic: pad rows replicate row 0
    double* out = Xs + (size_t)w * xs_stride;
    const long long total = rows_pad * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / D;
        const int d = (int)(i - r * D);
        const long long src = r < rows_real ? r : 0;
        out[i] = rows_real > 0 ? X[src * D + d] * sism[d] : 0.0;
    }
}

__global__ __launch_bounds__(256) void mcmc_accept_kernel(McmcState st, int start, int first, int h, int it) {
    const int ns = start ? st.ns_eval : st.k / 2;
    const int half = st.k / 2;
    for (int w = threadIdx.x; w < ns; w += blockDim.x) {
        const double lp = mcmc_lnprob(st.d_prior[w], st.d_fail[w], st.d_out[2 * w], st.d_out[2 * w + 1], st.n);
        if (lp != lp) atomicOr(st.d_err, 1);       // emcee: "lnprob returned NaN."
        if (st.d_fail[w] < 0) atomicOr(st.d_err, 4);   // a panel follower's hand-off timed out (potrf.hip): not a rejection
        if (start) {
            if (lp == __builtin_huge_val()) atomicOr(st.d_err, 2);   // "The initial lnprob was +inf."
            st.d_lnp[first + w] = lp;
            continue;
        }
        const int sw = h * half + w;
        const size_t r = ((size_t)it * 2 + h) * half + w;
        const double lnpdiff = mcmc_lnpdiff(st.P, log(st.d_z[w]), lp, st.d_lnp[sw]);
        if (lnpdiff > log(st.d_ua[r])) {
            const double* q = st.d_q + (size_t)w * st.P;
            for (int p = 0; p < st.P; ++p) st.d_pos[(size_t)sw * st.P + p] = q[p];
            st.d_lnp[sw] = lp;
            st.d_nacc[sw] += 1;
        }
    }
    if (start || h == 0) return;
    __syncthreads();
    // end of ensemble step `it`: record the chain
    if (st.d_chain)
        for (int i = threadIdx.x; i < st.k * st.P; i += blockDim.x) {
            const int w = i / st.P, p = i - w * st.P;
            st.d_chain[((size_t)w * st.n_steps + it) * st.P + p] = st.d_pos[i];
        }
    if (st.d_lnprob)
        for (int w = threadIdx.x; w < st.k; w += blockDim.x) st.d_lnprob[(size_t)w * st.n_steps + it] = st.d_lnp[w];
}

// Tail of a multi-block ensemble half-step in ONE launch, one workgroup per walker (r06): the likelihood terms of the
// factorisation's tail -- potrf_inverse_kernel's per-block shares added block by block as loglik_finish_kernel adds them, the
// same operations in the same order, hence the same bits -- and then mcmc_accept_kernel's test, walker update and chain record
// for THIS walker.  Three launches (shares, finish, accept: ~14 us of an 83-us half-step at N = 200) become one.
__global__ __launch_bounds__(256) void mcmc_tail_kernel(McmcState st, int start, int first, int h, int it,
                                                        const double* __restrict__ K, size_t k_stride, int ld, int nbf,
                                                        const int* __restrict__ fail) {
    __shared__ double red[4];
    __shared__ int sacc;
    const int w = blockIdx.x, tid = threadIdx.x, n = st.n, P = st.P;
    const double* Ks = K + (size_t)w * k_stride;
    double sq = 0.0, sl = 0.0;
    for (int kb = 0; kb < nbf; ++kb) {
        const int r = kb * NB + tid;
        double q = 0.0, lg = 0.0;
        if (tid < NB && r < n) {
            const double zi = Ks[(size_t)n * ld + r];
            q = zi * zi;
            lg = log(Ks[(size_t)r * ld + r]);
        }
        for (int o = 32; o > 0; o >>= 1) {
            q += __shfl_xor(q, o);
            lg += __shfl_xor(lg, o);
        }
        if ((tid & 63) == 0 && tid < NB) {
            red[tid >> 6] = q;
            red[2 + (tid >> 6)] = lg;
        }
        __syncthreads();
        sq += red[0] + red[1];
        sl += red[2] + red[3];
        __syncthreads();
    }
    const int half = st.k / 2, sw = start ? first + w : h * half + w;
    if (tid == 0) {
        const int f = fail[w];
        const double lp = mcmc_lnprob(st.d_prior[w], f, sq, 2.0 * sl, n);
        if (lp != lp) atomicOr(st.d_err, 1);       // emcee: "lnprob returned NaN."
        if (f < 0) atomicOr(st.d_err, 4);          // a panel follower's hand-off timed out (potrf.hip)
        int acc = 0;
        if (start) {
            if (lp == __builtin_huge_val()) atomicOr(st.d_err, 2);   // "The initial lnprob was +inf."
            st.d_lnp[sw] = lp;
        } else {
            const size_t r = ((size_t)it * 2 + h) * half + w;
            const double lnpdiff = mcmc_lnpdiff(P, log(st.d_z[w]), lp, st.d_lnp[sw]);
            if (lnpdiff > log(st.d_ua[r])) {
                acc = 1;
                st.d_lnp[sw] = lp;
                st.d_nacc[sw] += 1;
            }
            if (st.d_lnprob) st.d_lnprob[(size_t)sw * st.n_steps + it] = st.d_lnp[sw];
        }
        sacc = acc;
    }
    __syncthreads();
    if (start) return;
    const bool acc = sacc != 0;
    const double* q = st.d_q + (size_t)w * P;
    for (int p = tid; p < P; p += 256) {
        double* pp = st.d_pos + (size_t)sw * P + p;
        const double v = acc ? q[p] : *pp;
        if (acc) *pp = v;
        if (st.d_chain) st.d_chain[((size_t)sw * st.n_steps + it) * P + p] = v;
    }
}

int launch_mcmc_tail(robo_ctx* ctx, const McmcState& st, int start, int first, int h, int it, const double* d_K, size_t k_stride,
                     int ld, int nbf, const int* d_fail) {
    const int ns = start ? st.ns_eval : st.k / 2;
    hipLaunchKernelGGL(mcmc_tail_kernel, dim3(ns), dim3(256), 0, ctx->stream, st, start, first, h, it, d_K, k_stride, ld, nbf,
                       d_fail);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_mcmc_propose_scale(robo_ctx* ctx, const McmcState& st, int start, int first, int h, int it, const double* d_X,
                              double* d_Xs, int64_t rows_real, int64_t rows_pad, size_t xs_stride) {
    const int ns = start ? st.ns_eval : st.k / 2;
    int blocks = (int)((rows_pad * st.D + 255) / 256);
    if (blocks > 64) blocks = 64;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mcmc_propose_scale_kernel, dim3(blocks, ns), dim3(256), 0, ctx->stream, st, start, first, h, it,
                       d_X, d_Xs, (long long)rows_real, (long long)rows_pad, xs_stride);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_mcmc_accept(robo_ctx* ctx, const McmcState& st, int start, int first, int h, int it) {
    hipLaunchKernelGGL(mcmc_accept_kernel, dim3(1), dim3(256), 0, ctx->stream, st, start, first, h, it);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo

// ---- the random numbers of n_steps ensemble steps, drawn like numpy.random.RandomState does (host code) --------------
// emcee 2 draws per half-step  rand(k/2) -> randint(k/2, size=k/2) -> rand(k/2)  from a legacy RandomState
// (gaussian_process_mcmc.py:117 hands it the model's stream).  The chain above needs them all up front; 1200 NumPy calls
// for a 200-step chain were ~2 ms of a 17 ms Bayesian-optimisation iteration.  Legacy RandomState is frozen by NumPy's
// compatibility policy: MT19937; random_sample = (a >> 5) * 2^26 + (b >> 6) over 2^53 from two 32-bit outputs; randint
// below 2^32 = masked rejection on 32-bit outputs (no output consumed when the range is a single value).
static inline uint32_t mt_next(uint32_t* key, int32_t* pos) {
    constexpr int N = 624, M = 397;
    if (*pos >= N) {
        for (int kk = 0; kk < N; ++kk) {
            const uint32_t y = (key[kk] & 0x80000000u) | (key[(kk + 1) % N] & 0x7fffffffu);
            key[kk] = key[(kk + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        *pos = 0;
    }
    uint32_t y = key[(*pos)++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

extern "C" int32_t robo_mcmc_draws(uint32_t* mt_key, int32_t* mt_pos, int32_t n_steps, int32_t half, double* u_stretch,
                                   int32_t* partner, double* u_accept) {
    if (!mt_key || !mt_pos || n_steps < 0 || half < 1 || *mt_pos < 0 || *mt_pos > 624) return ROBO_BAD_ARGUMENT;
    if (n_steps > 0 && (!u_stretch || !partner || !u_accept)) return ROBO_BAD_ARGUMENT;
    const uint32_t rng = (uint32_t)half - 1u;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    auto uniform = [&]() {
        const uint32_t a = mt_next(mt_key, mt_pos) >> 5, b = mt_next(mt_key, mt_pos) >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    };
    for (size_t hs = 0; hs < (size_t)n_steps * 2; ++hs) {
        double* uz = u_stretch + hs * half;
        double* ua = u_accept + hs * half;
        int32_t* pa = partner + hs * half;
        for (int w = 0; w < half; ++w) uz[w] = uniform();
        for (int w = 0; w < half; ++w) {
            uint32_t v = 0;
            if (rng != 0u) {
                do v = mt_next(mt_key, mt_pos) & mask;
                while (v > rng);
            }
            pa[w] = (int32_t)v;
        }
        for (int w = 0; w < half; ++w) ua[w] = uniform();
    }
    return ROBO_OK;
}
