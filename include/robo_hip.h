/*
 * robo_hip.h -- C ABI of librobo_hip.so: the MI355X (gfx950) GP-posterior + acquisition
 * hot path of automl/RoBO.
 *
 * This is the drop-in boundary.  The reference has no native
 * code; the arithmetic it delegates to the george C++ library through the call sites
 * listed below is what these entry points replace.  A RoBO maintainer binds them with
 * ctypes; robo_amd/_lib.py is exactly that binding.
 *
 * Conventions
 *   - every function returns a robo_status (int32); 0 = OK
 *   - all matrices are row-major (C order) IEEE fp64, exactly what NumPy hands over
 *   - "host" pointers are caller-owned and only read/written during the call
 *   - handles own all device memory; nothing is allocated per call on the hot path
 *   - a handle is not thread-safe; one HIP stream per context; no callbacks
 *   - theta is the reference's log-space hyper-parameter vector
 *       [log amp, log m_1 .. log m_D, log sigma^2]           (P = D + 2; ROBO_KERNEL_FABOLAS: see enum)
 *     (robo/models/gaussian_process.py:110-114,151-152; robo/priors/default_priors.py:28-35)
 */
#ifndef ROBO_HIP_H
#define ROBO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct robo_ctx robo_ctx;   /* one device + one HIP stream + event slots            */
typedef struct robo_gp robo_gp;     /* one GP: training data, Cholesky factor, solve vector */
typedef struct robo_cand robo_cand; /* one device-resident candidate batch + its workspace  */

enum robo_status {
    ROBO_OK = 0,
    ROBO_NOT_POSITIVE_DEFINITE = 1, /* -> np.linalg.LinAlgError (gaussian_process.py:120,156;
                                          gaussian_process_mcmc.py:194-197)                  */
    ROBO_NOT_FITTED = 2,            /* -> Exception('Model has to be trained first!')
                                          (gaussian_process.py:241,273,322)                  */
    ROBO_BAD_SHAPE = 3,             /* the reference asserts (base_model.py:68-70,76)        */
    ROBO_RUNTIME_ERROR = 4,         /* HIP error; text in robo_last_error_string()           */
    ROBO_BAD_ARGUMENT = 5
};

enum robo_kernel_kind {
    ROBO_KERNEL_MATERN52_ARD = 0, /* amp * george.kernels.Matern52Kernel(metric, ndim=D)
                                     (robo/fmin/bayesian_optimization.py:75-81)              */
    ROBO_KERNEL_RBF_ARD = 1,      /* amp * george.kernels.ExpSquaredKernel(metric, ndim=D)   */
    ROBO_KERNEL_FABOLAS = 2       /* amp * prod_d Matern52Kernel(m_d, axes=d) * BayesianLinearRegressionKernel(
                                     log_a, log_b, axes=D)   (robo/fmin/fabolas.py:104-117).  dim = D + 1: the
                                     last input column is the basis-transformed fidelity u (fabolas_gp.py:122-126);
                                     theta = [log amp, log m_1..m_D, log_a, log_b, log sigma^2]  (P = dim + 3)    */
};

enum robo_acq_kind {
    ROBO_ACQ_EI = 0,     /* robo/acquisition_functions/ei.py:65-88      */
    ROBO_ACQ_LOG_EI = 1, /* robo/acquisition_functions/log_ei.py:74-120 */
    ROBO_ACQ_PI = 2,     /* robo/acquisition_functions/pi.py:57-63      */
    ROBO_ACQ_LCB = 3     /* robo/acquisition_functions/lcb.py:62-65     */
};

/* flags reported by the acquisition kernels (host shim applies the reference's guards) */
#define ROBO_FLAG_ZERO_SIGMA 1u   /* some s == 0           (ei.py:72-74)  */
#define ROBO_FLAG_NEGATIVE_EI 2u  /* some EI < 0           (ei.py:86-88)  */
#define ROBO_FLAG_NAN 4u          /* some acquisition value is NaN        */

/* ---- context ---------------------------------------------------------------------- */
int32_t robo_device_count(int32_t* out_n);
/* hip_stream: an existing hipStream_t to launch on (e.g. torch's current stream), or NULL
 * to let the library create its own non-blocking stream.                                  */
int32_t robo_ctx_create(int32_t device, void* hip_stream, robo_ctx** out);
/* Handles created on a context (robo_gp, robo_cand, robo_comm, robo_multi) keep it alive: robo_ctx_destroy on a context
 * that still has such handles only marks it; its stream, events and scratch are released with the last of them.  (A
 * garbage-collected binding cannot promise to finalise a context after everything that lives on it.)                     */
int32_t robo_ctx_destroy(robo_ctx* ctx);
/* contexts whose resources are still held (created and not yet released): diagnostics / tests                            */
int32_t robo_ctx_live_count(int32_t* out_n);
int32_t robo_ctx_synchronize(robo_ctx* ctx);
int32_t robo_ctx_device_name(robo_ctx* ctx, char* buf, int32_t buf_len);
/* HIP-event timing on the context's stream (what bench.py uses).  Slots 0..19 are the
 * caller's; the library itself records 24..27 around the phases of the posterior evaluation
 * (cross-gram, triangular solve, post) of the last candidate chunk and, when switched on with
 * robo_ctx_set_phase_events (off by default: four event packets cost a 1.8 ms fit ~30 us; the
 * environment variable ROBO_PHASE_EVENTS=1 switches them on at context creation), 19..23 around
 * the phases of robo_gp_fit (19 -> 21 gram kernel, 20 -> 21 gram phase, 21 -> 22 Cholesky,
 * 22 -> 23 log-likelihood).                                                                 */
int32_t robo_ctx_event_record(robo_ctx* ctx, int32_t slot);
int32_t robo_ctx_event_elapsed_ms(robo_ctx* ctx, int32_t slot_begin, int32_t slot_end, float* out_ms);
int32_t robo_ctx_set_phase_events(robo_ctx* ctx, int32_t on);
/* Tuning knobs of a context.  They are read from the environment ONCE, when the context is created (variable
 * ROBO_<KEY in upper case>), and changed afterwards only through this call; value INT64_MIN restores the default, key
 * "env" re-reads all variables.  Keys: ws_bytes (solve workspace per candidate handle, default 6 GiB),
 * winv_max / winv_min_blocks (batches of at most winv_max candidates on a factor of at least winv_min_blocks 128-row
 * blocks are evaluated through the explicit inverse factor, winv.hip; default 32768 / 6, measured r03b; 0 = never),
 * winv_cond_max (... while cond_inf(L) <= this, default 1e5: robo_gp_factor_cond), winv_rows (form of that product: -1 auto,
 * 0 chunked units + reduction pass, 1 one workgroup per (candidate tile, block row) over the whole contraction range),
 * trsm_pair (1: two block rows of the solve per launch on one read of V; default 0: one), trsm_small_max, trsm_small_narrow,
 * trsm_small_deep, trsm_rows, predict_stepwise, mcmc_block_step, potrf_fused, potrf_tm4_min,
 * potrf_max_wg, potrf_group, potrf_split (sub-batches of a batched factorisation on their own streams, default 3, from
 * potrf_split_min = 12 panels on), potrf_lead, potrf_thin_last (kernel-variant selection; A/B runs and tests).         */
int32_t robo_ctx_set_tuning(robo_ctx* ctx, const char* key, int64_t value);
const char* robo_last_error_string(void);
const char* robo_version_string(void);

/* ---- GP fit: replaces george.GP.compute + log_likelihood ---------------------------
 * call sites: robo/models/gaussian_process.py:119,122,155,159,173;
 *             robo/models/gaussian_process_mcmc.py:195,200-202                          */
int32_t robo_gp_create(robo_ctx* ctx, int32_t kernel_kind, int32_t n_max, int32_t dim, robo_gp** out);
int32_t robo_gp_destroy(robo_gp* gp);
/* X: (n, dim) inputs as the GP sees them (already [0,1]-normalised by the caller,
 * gaussian_process.py:91); y: (n,) targets.  Uploaded once; thousands of theta are then
 * evaluated on the same data (the MCMC / L-BFGS loops).                                   */
int32_t robo_gp_set_data(robo_gp* gp, const double* X, const double* y, int32_t n);
/* un-normalisation applied to predictions: mu*y_std + y_mean, var*y_std^2
 * (gaussian_process.py:282-284).  Default (0, 1).                                          */
int32_t robo_gp_set_output_transform(robo_gp* gp, double y_mean, double y_std);
/* 0 (default): covariance entries in fp64.  1: evaluated in fp32 and widened -- the mixed
 * precision "fp32 K-build + fp64 Cholesky" of BASELINE.json config 5 (applies to K and K*).  */
int32_t robo_gp_set_precision(robo_gp* gp, int32_t fp32_gram);
/* number of theta entries for a kernel kind and input dimension                              */
int32_t robo_theta_size(int32_t kernel_kind, int32_t dim);
/* K = k_theta(X,X) + (sigma^2 + 1.25e-12) I ; L = chol(K) ; z = L^-1 (y - mean_c);
 * loglik = -1/2 (z.z + 2 sum log L_ii + n log 2pi).  On ROBO_NOT_POSITIVE_DEFINITE
 * *out_fail_col is the 0-based failing column and the GP is left unfitted.
 * out_loglik / out_fail_col may be NULL.                                                   */
int32_t robo_gp_fit(robo_gp* gp, const double* theta, double mean_c, double* out_loglik, int32_t* out_fail_col);
/* Analytic gradient of the log marginal likelihood w.r.t. theta (log-space), fitted at theta on
 * return like robo_gp_fit.  Replaces the body of GaussianProcess.grad_nll
 * (robo/models/gaussian_process.py:168-191: K^-1 via solver.apply_inverse, kernel.gradient (N,N,P),
 * 0.5 einsum('ijk,ij', Kg, alpha alpha^T - K^-1)) without forming K^-1 or the gradient tensor on
 * the host.  out_grad[theta_size]: d loglik / d theta_p for the kernel parameters; the LAST entry is
 * 0.5 tr(alpha alpha^T - K^-1) = d loglik / d sigma^2 (not d / d log sigma^2) exactly as the
 * reference computes it (:178-182 use the identity as the noise "gradient").  Prior gradients and
 * the sign flip of grad_nll stay with the caller.                                               */
int32_t robo_gp_grad_loglik(robo_gp* gp, const double* theta, double mean_c, double* out_loglik,
                            double* out_grad, int32_t* out_fail_col);

/* S independent likelihood evaluations on the same data in ONE batched pass (every kernel of
 * the fit runs with S x the workgroups); out_status[s] is a robo_status.  Works on a separate
 * batch workspace; the GP itself is left UNFITTED (call robo_gp_fit for the theta to keep).    */
int32_t robo_gp_loglik_batch(robo_gp* gp, const double* thetas, int32_t S, double mean_c, double* out_loglik,
                             int32_t* out_status);

/* The hyper-parameter chain of GaussianProcessMCMC.train, resident on the device: replaces
 *     sampler = emcee.EnsembleSampler(n_hypers, ndim, self.loglikelihood); sampler.run_mcmc(p0, n_steps, rstate0=rng)
 * (robo/models/gaussian_process_mcmc.py:114-142) INCLUDING emcee 2's stretch move, the prior and the accept test -- one
 * sequence of launches, no host round trip between half-steps.  lnprob(theta) = log p(y | X, theta) + log prior(theta)
 * with the reference's protocol (:185-202: any |theta_p| > 20, or a factorisation that fails -> -inf).
 * prior_kind 0 = none, 1 = robo.priors.default_priors.DefaultPrior with prior_par = {lognormal loc, lognormal sigma,
 * tophat min, tophat max, horseshoe scale}, 2 = robo.priors.env_priors.EnvPrior (robo/priors/env_priors.py:8-54, the prior
 * robo/fmin/fabolas.py:120-127 gives FabolasGPMCMC, robo/models/fabolas_gp.py:49-77) with prior_par = {those five, n_ls,
 * n_lr, normal mean, normal sigma}: tophat on theta[1 .. n_ls] only, plus NormalPrior.lnprob -- a pdf, as in the
 * reference (robo/priors/base_prior.py:357) -- of each of the n_lr regression parameters behind them.  The random numbers of a stretch-move chain do not depend on its state: the
 * caller draws them in emcee 2's order -- per step and half-ensemble rand(k/2) for z, randint(k/2) for the partners,
 * rand(k/2) for the accept test -- into u_stretch / partner / u_accept, each [n_steps][2][k/2].
 * pos (k x P) and lnp (k): in = start positions (lnp evaluated here when eval_start != 0), out = final state;
 * out_chain (k x n_steps x P), out_lnprob (k x n_steps), out_accepted (k) nullable.  a = 2 in emcee.
 * ROBO_BAD_ARGUMENT with emcee's message if a log-probability is NaN or the initial one +inf.  Leaves the GP unfitted. */
int32_t robo_gp_mcmc_run(robo_gp* gp, double mean_c, int32_t prior_kind, const double* prior_par, int32_t n_walkers,
                         int32_t n_steps, double a, const double* u_stretch, const int32_t* partner,
                         const double* u_accept, int32_t eval_start, double* pos, double* lnp, double* out_chain,
                         double* out_lnprob, int64_t* out_accepted);
/* Host helper of robo_gp_mcmc_run: the random numbers of n_steps ensemble steps exactly as a legacy
 * numpy.random.RandomState produces them in emcee 2's order -- per half-step rand(half), randint(half, size=half),
 * rand(half) -- from the MT19937 state (key[624], pos) of RandomState.get_state(), which is advanced in place (hand it back
 * with set_state).  Outputs [n_steps][2][half].  Pure host code; replaces 6 n_steps NumPy calls.                     */
int32_t robo_mcmc_draws(uint32_t* mt_key, int32_t* mt_pos, int32_t n_steps, int32_t half, double* u_stretch,
                        int32_t* partner, double* u_accept);
/* The per-sample model fits of GaussianProcessMCMC.train (gaussian_process_mcmc.py:149-164: one
 * GaussianProcess per hyper-parameter sample, each a gp.compute on the SAME X, y) as one batched pass
 * that KEEPS the factors: gps[0] holds the training data (robo_gp_set_data); afterwards every gps[s]
 * with out_status[s] == ROBO_OK is a fitted handle at thetas[s] (own copy of the data, factor, diagonal
 * block inverses, scaled inputs), bit-identical to robo_gp_fit(gps[s], thetas[s]).  Samples whose K is not
 * positive definite are left unfitted with out_status[s] = ROBO_NOT_POSITIVE_DEFINITE (the caller applies
 * the reference's noise x 10 retry, gaussian_process.py:120-122, to those).  All handles: same context,
 * kernel kind, dim, n_max >= n, pairwise distinct.                                                   */
int32_t robo_gp_fit_batch(robo_gp* const* gps, int32_t S, const double* thetas, double mean_c, double* out_loglik,
                          int32_t* out_status);
/* copy the lower Cholesky factor (n x n, row-major, upper zeroed) back -- diagnostics/tests */
int32_t robo_gp_get_factor(robo_gp* gp, double* out_L);
int32_t robo_gp_get_gram(robo_gp* gp, const double* theta, double* out_K); /* K incl. noise, n x n  */
/* cond_inf(L) = |L|_inf |L^-1|_inf of the current factor, EXACT (row sums of L and of the explicit inverse W, which is
 * built for this factor if it has not been yet: ~1 ms at N = 4096, then cached until the next fit).  It is the number
 * that decides whether small candidate batches of robo_gp_predict* / robo_acq_eval* (the reference's
 * gp.predict, robo/models/gaussian_process.py:280-294) may go through W (forward error ~eps cond) or stay on the
 * block-row substitution: the explicit inverse is used while it is <= the context's `winv_cond_max` (default 1e5).
 * out[0] = cond_inf(L), out[1] = min L_ii, out[2] = max L_ii.  Diagnostics / tests.                         */
int32_t robo_gp_factor_cond(robo_gp* gp, double* out);
/* Start building that explicit inverse NOW, asynchronously (returns without waiting for the device): called right after the
 * final fit of GaussianProcess.train (robo/models/gaussian_process.py:119) so that the build (0.9 ms at N = 4096) runs while
 * the host prepares the next acquisition maximisation (robo/maximizers/random_sampling.py:38-47 draws its 500 candidates in
 * a Python loop) and the first robo_gp_predict* / robo_acq_eval* of a small batch finds W in place.  A no-op where a small
 * batch would not use W (factor of fewer than winv_min_blocks blocks, fp32 K-build, diagonal ratio beyond the bound) and
 * on a handle that has never evaluated a small batch through W (nothing is allocated or built speculatively for models
 * that only see large batches): from the second iteration of a loop on.                                            */
int32_t robo_gp_prefetch_inverse(robo_gp* gp);

/* ---- candidates --------------------------------------------------------------------- */
/* Xc: (m, dim) candidates in the GP's (normalised) input space; copied H2D here, once.    */
int32_t robo_cand_create(robo_ctx* ctx, const double* Xc, int64_t m, int32_t dim, robo_cand** out);
int32_t robo_cand_destroy(robo_cand* cand);
/* a new batch of the SAME size into an existing handle (H2D only; buffers and solve workspace are kept): what a BO
 * loop does once per iteration with robo/maximizers/random_sampling.py:38-47's fresh candidate matrix            */
int32_t robo_cand_set_points(robo_cand* cand, const double* Xc, int64_t m);
/* device-side generation, no H2D: uniform [0,1)^dim, counter-based (Philox-4x32-10)         */
int32_t robo_cand_create_uniform(robo_ctx* ctx, int64_t m, int32_t dim, uint64_t seed, robo_cand** out);
/* RandomSampling.maximize's candidate recipe on the device (random_sampling.py:38-47), in the
 * normalised space: rows < n_uniform uniform, the rest N(loc, scale) clipped to [0,1]
 * (loc = normalised incumbent, scale_d = 0.1 / (upper_d - lower_d)).                           */
int32_t robo_cand_create_random(robo_ctx* ctx, int64_t m, int32_t dim, uint64_t seed, int64_t n_uniform,
                                const double* loc, const double* scale, robo_cand** out);
/* points first_index .. first_index + m - 1 of a (scrambled) Sobol' sequence in [0,1)^dim, generated on the device
 * from the direction numbers sv (dim x bits, row-major) and the digital shift (dim):
 *   x_k[d] = (shift[d] ^ XOR_{b in gray(k)} sv[d][b]) / 2^bits.   BASELINE config 5 names 2^20 Sobol candidates
 * (scipy.stats.qmc.Sobol(d=64, scramble=True, seed=0).random_base2(20); the reference itself has no
 * Sobol sampler); with sv / shift of a SciPy engine the sequence equals engine.random() bit for bit, and a rank of a
 * candidate shard generates its own slice through first_index.                                            */
int32_t robo_cand_create_sobol(robo_ctx* ctx, int64_t m, int32_t dim, const uint64_t* sv, const uint64_t* shift,
                               int32_t bits, uint64_t first_index, robo_cand** out);
int32_t robo_cand_get_points(robo_cand* cand, double* out_Xc);
/* candidates per pass of the solve workspace of the LAST posterior evaluated on this handle (a multiple of 128; the
 * whole padded batch when it fits ROBO_WS_BYTES, default 6 GiB); 0 before the first evaluation.  Diagnostics: lets a
 * caller (tests) see that a batch was evaluated in several passes.                                              */
int32_t robo_cand_workspace_chunk(robo_cand* cand, int64_t* out_chunk);
/* name of the kernel that ran the solve of the last posterior on this handle ("" before the first): the library picks
 * it from the batch size and precision; bench.py labels its roofline block with it.  Diagnostics.                */
int32_t robo_cand_last_solve_kernel(robo_cand* cand, char* buf, int32_t buf_len);
int32_t robo_cand_get_point(robo_cand* cand, int64_t index, double* out_x); /* one row: the winner */

/* ---- posterior: replaces george.GP.predict (gaussian_process.py:280-294) ------------ */
/* mean (m,), var (m,): diagonal only; var floored at DBL_EPSILON after the output
 * transform, exactly gaussian_process.py:282-294.  Either output may be NULL.
 * The host-array forms (robo_gp_predict, robo_acq_eval) keep the candidate handle they build inside the
 * robo_gp between calls of one batch size <= 16384 (the reference's 500 candidates per iteration, the 1 x D
 * calls of its single-point maximisers): a repeated call only re-uploads the points.                       */
int32_t robo_gp_predict_cand(robo_gp* gp, robo_cand* cand, double* out_mean, double* out_var);
int32_t robo_gp_predict(robo_gp* gp, const double* Xc, int64_t m, double* out_mean, double* out_var);
/* GaussianProcessMCMC.predict (gaussian_process_mcmc.py:205-249): mixture over S fitted GPs,
 * m = mean_s mu_s, v = var_s(mu_s) + mean_s(var_s), floored at DBL_EPSILON; NumPy's operation
 * order along the sample axis.                                                                */
int32_t robo_gp_predict_mixture_cand(robo_gp* const* gps, int32_t S, robo_cand* cand, double* out_mean,
                                     double* out_var);
/* full covariance (m x m), small m only (predict(full_cov=True), predict_variance,
 * sample_functions: gaussian_process.py:221-248,298-332)                                    */
/* Gradients of the posterior w.r.t. the test inputs: what model.predictive_gradients(X) supplies to the
 * reference's acquisition derivatives (robo/acquisition_functions/ei.py:80-85, pi.py:65-71, lcb.py:66-68) and to
 * robo/util/posterior_optimization.py:38-40,96-104 (no model in the reference tree implements it).
 * Xc (m x D) in the GP's normalised input space; out_dmean / out_dvar (m x D, row-major): derivatives of the
 * values out_mean / out_var w.r.t. those coordinates (output transform applied; the variance floor is not
 * differentiated).  Cost: D + 1 right-hand sides of the blocked forward substitution per point.            */
int32_t robo_gp_predict_grad(robo_gp* gp, const double* Xc, int64_t m, double* out_mean, double* out_var,
                             double* out_dmean, double* out_dvar);
int32_t robo_gp_predict_cov(robo_gp* gp, const double* Xc, int64_t m, double* out_mean, double* out_cov);

/* ---- acquisition: replaces EI/LogEI/PI/LCB.compute + the argmax of
 * RandomSampling.maximize (robo/maximizers/random_sampling.py:48-50) ------------------ */
/* out_acq (m,) may be NULL when only the maximiser is wanted.  argmax follows np.argmax:
 * first index of the maximum, NaN counts as maximal.  par = xi (EI/LogEI/PI) or kappa (LCB);
 * eta = incumbent value (ignored by LCB).                                                     */
int32_t robo_acq_eval_cand(robo_gp* gp, int32_t acq_kind, double par, double eta, robo_cand* cand, double* out_acq,
                           double* out_max, int64_t* out_argmax, uint32_t* out_flags);
int32_t robo_acq_eval(robo_gp* gp, int32_t acq_kind, double par, double eta, const double* Xc, int64_t m,
                      double* out_acq, double* out_max, int64_t* out_argmax, uint32_t* out_flags);
/* the element-wise half alone, for (mean, var) produced by any other BaseModel plugin
 * (the reference's acquisition classes accept every model, e.g. test/dummy_model.py)          */
int32_t robo_acq_eval_moments(robo_ctx* ctx, int32_t acq_kind, double par, double eta, const double* mean,
                              const double* var, int64_t m, double* out_acq, double* out_max, int64_t* out_argmax,
                              uint32_t* out_flags);
/* MarginalizationGPMCMC.compute (marginalization.py:115-121): mean over S fitted GPs of
 * the per-sample acquisition, accumulated in sample order (= NumPy's axis-0 mean).
 * etas (S): the incumbent value of every sample -- each estimator asks ITS OWN sub-model
 * (marginalization.py:40, ei.py:68): equal entries for plain GPs (observed minimum), different
 * ones for FabolasGP sub-models (predicted minimum of the projected points, fabolas_gp.py:141-164). */
int32_t robo_acq_eval_marginal_cand(robo_gp* const* gps, int32_t S, int32_t acq_kind, double par, const double* etas,
                                    robo_cand* cand, double* out_acq, double* out_max, int64_t* out_argmax,
                                    uint32_t* out_flags);
/* partial form for sample-sharded multi-GPU runs: returns sum_s acq_s (no division)          */
int32_t robo_acq_eval_sum_cand(robo_gp* const* gps, int32_t S, int32_t acq_kind, double par, const double* etas,
                               robo_cand* cand, double* out_acq_sum, uint32_t* out_flags);

/* ---- entropy search: replaces InformationGain.innovations/_dh_fun/compute ---------------
 * (robo/acquisition_functions/information_gain.py:87-125,169-203,253-272), batched over candidates.
 * rep: the Nb <= 64 representer points as a candidate batch (same normalised space).  The EP
 * state (log p_min and its derivatives, robo/util/epmgp.py:11-81) is host input:
 *   logP (Nb), lmb (Nb) log proposal values, W (Np <= 512) standard-normal quantiles,
 *   dlogPdMu (Nb,Nb), dlogPdSigma (Nb, Nb(Nb+1)/2) lower-triangle packed, dlogPdMudMu (Nb,Nb,Nb).
 * sn2 = model noise (get_noise()).  out_dh (m,) may be NULL; argmax as robo_acq_eval.         */
int32_t robo_ig_eval_cand(robo_gp* gp, robo_cand* cand, robo_cand* rep, int32_t n_outcomes, double sn2,
                          const double* logP, const double* lmb, const double* W, const double* dlogPdMu,
                          const double* dlogPdSigma, const double* dlogPdMudMu, double* out_dh, double* out_max,
                          int64_t* out_argmax);
/* Information gain PER UNIT COST with its argmax on the device: replaces InformationGainPerUnitCost.compute
 * (robo/acquisition_functions/information_gain_per_unit_cost.py:91-104)
 *     log_cost = self.cost_model.predict(X)[0];  dh = InformationGain.compute(X);  dh / (np.exp(log_cost) + overhead)
 * and the maximiser's argmax over it (robo/maximizers/random_sampling.py:48-50).  cost_gp: the fitted cost model;
 * cost_cand: the SAME m candidates in the cost model's input space (Fabolas: linear basis on the fidelity column where
 * the objective model has (1 - s)^2, robo/fmin/fabolas.py:130-131), a second handle on the same context.
 * out_values (m,) nullable; out_max / out_argmax as robo_acq_eval.                                               */
int32_t robo_ig_eval_per_cost_cand(robo_gp* gp, robo_cand* cand, robo_cand* rep, int32_t n_outcomes, double sn2,
                                   const double* logP, const double* lmb, const double* W, const double* dlogPdMu,
                                   const double* dlogPdSigma, const double* dlogPdMudMu, robo_gp* cost_gp,
                                   robo_cand* cost_cand, double overhead, double* out_values, double* out_max,
                                   int64_t* out_argmax);
/* the same from innovations inputs supplied by any other model: s (m, nb) covariances between each
 * candidate and the representer points, v (m,) predictive variances                              */
int32_t robo_ig_eval_moments(robo_ctx* ctx, int64_t m, int32_t nb, int32_t n_outcomes, double sn2, const double* s,
                             const double* v, const double* logP, const double* lmb, const double* W,
                             const double* dlogPdMu, const double* dlogPdSigma, const double* dlogPdMudMu,
                             double* out_dh);
/* posterior covariances cov(x_c, z_b) (m, nref <= 64), floored at DBL_EPSILON like the reference's
 * predict(full_cov=True) -> predict_variance path (gaussian_process.py:243-246,290-294)            */
int32_t robo_gp_cross_cov(robo_gp* gp, robo_cand* cand, robo_cand* ref, double* out_cov);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (SURVEY.md 8b "robo_comm_init + _sharded variants", 8e) -------
 * The reference is single-process; these entry points shard its two independent axes -- the candidate batch of
 * RandomSampling.maximize (robo/maximizers/random_sampling.py:42-50) and the hyper-parameter samples of
 * MarginalizationGPMCMC.compute (robo/acquisition_functions/marginalization.py:115-121) -- and run the one small
 * exchange each of them needs as an RCCL all-gather on the context's stream, device pointers on both sides.  Every
 * rank holds its own replica of the fitted GP(s) (the fit is deterministic).  librccl.so is loaded on first use.
 * COLLECTIVE: every rank of the communicator must make the same call; a rank whose local half fails still takes
 * part in the exchange (as an empty shard) and then returns its error.                                            */
typedef struct robo_comm robo_comm;
#define ROBO_COMM_ID_BYTES 128
/* rank 0 creates the id (ncclGetUniqueId); every rank must receive the same 128 bytes out of band (a file, a TCP
 * store, MPI, torch.distributed.broadcast_object_list ...) before robo_comm_init                                  */
int32_t robo_comm_create_id(void* out_id);
int32_t robo_comm_init(robo_ctx* ctx, int32_t rank, int32_t world, const void* id, robo_comm** out);
int32_t robo_comm_destroy(robo_comm* comm);
int32_t robo_comm_info(robo_comm* comm, int32_t* out_rank, int32_t* out_world);
/* all-gather of `count` host doubles per rank -> recv (world x count), rank-major: the winning point of a
 * device-generated shard, consistency checks, timings -- the few-hundred-byte side channel                         */
int32_t robo_comm_allgather(robo_comm* comm, const double* send, int64_t count, double* recv);
/* candidate shard: robo_acq_eval_cand on this rank's candidates (global index of candidate c = global_offset + c),
 * then the all-gather of the per-rank incumbents (32 B per rank: max, index, flags, status) and np.argmax's tie-break across them on the device.
 * out_max / out_argmax: the GLOBAL maximum and its global index, identical on every rank; out_owner_rank: the rank
 * whose shard holds it; out_flags: OR over all ranks; out_acq (nullable): this rank's m values.                    */
int32_t robo_acq_eval_cand_sharded(robo_comm* comm, robo_gp* gp, int32_t acq_kind, double par, double eta,
                                   robo_cand* cand, int64_t global_offset, double* out_acq, double* out_max,
                                   int64_t* out_argmax, int32_t* out_owner_rank, uint32_t* out_flags);
/* candidate shard of the information gain per unit cost (BASELINE config 4: 65 536 candidates over 8 GPUs):
 * robo_ig_eval_per_cost_cand on this rank's candidates, then the same exchange as robo_acq_eval_cand_sharded.     */
int32_t robo_ig_eval_per_cost_cand_sharded(robo_comm* comm, robo_gp* gp, robo_cand* cand, robo_cand* rep,
                                           int32_t n_outcomes, double sn2, const double* logP, const double* lmb,
                                           const double* W, const double* dlogPdMu, const double* dlogPdSigma,
                                           const double* dlogPdMudMu, robo_gp* cost_gp, robo_cand* cost_cand,
                                           double overhead, int64_t global_offset, double* out_values, double* out_max,
                                           int64_t* out_argmax, int32_t* out_owner_rank);
/* sample shard: this rank's S_local fitted GPs (S_total over all ranks; S_local may be 0) on ALL candidates; the
 * per-rank partial sums are all-gathered (m doubles per rank) and added in rank order on the device, divided by
 * S_total, argmax.  Outputs as robo_acq_eval_marginal_cand, identical on every rank; equal to the single-process
 * sample-order accumulation up to fp64 re-association (partial sums are formed per shard).                         */
int32_t robo_acq_eval_marginal_cand_sharded(robo_comm* comm, robo_gp* const* gps, int32_t S_local, int32_t S_total,
                                            int32_t acq_kind, double par, const double* etas, robo_cand* cand,
                                            double* out_acq, double* out_max, int64_t* out_argmax,
                                            uint32_t* out_flags);

/* ---- multi-GPU, ONE process: G contexts of the calling process, one per device (SURVEY.md 8b "Threading: single process,
 * one context per device") ------------------------------------------------------------------------------------------------
 * The reference is one process with one objective evaluation per iteration (robo/solver/bayesian_optimization.py:156-203).
 * A robo_multi fans the shards of the same two axes as the _sharded entry points above -- candidates
 * (robo/maximizers/random_sampling.py:42-50) and hyper-parameter samples (robo/acquisition_functions/marginalization.py:
 * 115-121, robo/models/gaussian_process_mcmc.py:149-164,235-247) -- over its contexts: one worker thread per device runs the
 * ordinary single-device entry point on that device's context and stream, all devices concurrently, and the G results are
 * reduced -- candidate shards on the host (32 bytes per device arrive through each context's pinned read-back anyway; np.argmax
 * tie-break: NaN maximal, larger value, lower global index), sample shards on the FIRST context's device (peer copies over
 * xGMI, then the rank-ordered sum / mixture kernels of the single-device and _sharded forms: same bits as those).
 * No collective is involved: a failing device cannot hang the others; every call waits for all devices and returns the
 * first failing device's status.  Arrays indexed "[G]" have one entry per context, in the order given to robo_multi_create;
 * every handle of slot g must live on context g.  Several contexts may share a device (tests on a one-GPU box).
 * ROBO_MULTI_THREADS=0 (read at creation) runs the per-device halves one after the other on the caller's thread.        */
typedef struct robo_multi robo_multi;
int32_t robo_multi_create(robo_ctx* const* ctxs, int32_t n_ctx, robo_multi** out);
int32_t robo_multi_destroy(robo_multi* multi);
/* out_devices[G] (nullable): HIP device of every slot; out_threads: worker threads in use (0 = caller's thread)            */
int32_t robo_multi_info(robo_multi* multi, int32_t* out_n, int32_t* out_devices, int32_t* out_threads);
/* the same training data / the same fit on every device (a replica per device; the fit is deterministic: replicas whose
 * status or log-likelihood bits differ are reported as ROBO_RUNTIME_ERROR).  gaussian_process.py:119,122,155          */
int32_t robo_gp_set_data_multi(robo_multi* multi, robo_gp* const* gps, const double* X, const double* y, int32_t n);
int32_t robo_gp_fit_multi(robo_multi* multi, robo_gp* const* gps, const double* theta, double mean_c, double* out_loglik,
                          int32_t* out_fail_col);
/* robo_gp_loglik_batch with the S thetas split contiguously over the devices (the first S % G devices take one more):
 * the walkers of an ensemble half-step (gaussian_process_mcmc.py:114-142,168-202).  gps[G]: one handle per device, each
 * holding the data.                                                                                                       */
int32_t robo_gp_loglik_batch_multi(robo_multi* multi, robo_gp* const* gps, const double* thetas, int32_t S, double mean_c,
                                   double* out_loglik, int32_t* out_status);
/* robo_gp_fit_batch per device: gps[S_total] in device order (S_dev[0] handles of slot 0 first, ...), thetas / outputs in
 * the same order; the FIRST handle of every device's group holds the training data on that device.
 * gaussian_process_mcmc.py:149-164                                                                                         */
int32_t robo_gp_fit_batch_multi(robo_multi* multi, robo_gp* const* gps, const int32_t* S_dev, const double* thetas,
                                double mean_c, double* out_loglik, int32_t* out_status);
/* candidate shard: robo_acq_eval_cand on every device's shard (cands[g] NULL = empty shard), gps[g] = that device's replica;
 * global index of candidate c of slot g = global_offsets[g] + c.  out_acq (nullable): the values of all shards, slot after
 * slot.  out_owner: the slot whose shard holds the maximum; out_flags: OR over the devices.                              */
int32_t robo_acq_eval_cand_multi(robo_multi* multi, robo_gp* const* gps, int32_t acq_kind, double par, double eta,
                                 robo_cand* const* cands, const int64_t* global_offsets, double* out_acq, double* out_max,
                                 int64_t* out_argmax, int32_t* out_owner, uint32_t* out_flags);
/* candidate shard of robo_ig_eval_cand (InformationGain.compute, information_gain.py:87-125,253-272): reps[g] = the
 * representer points on device g, the EP state is the same host input for every device                                     */
int32_t robo_ig_eval_cand_multi(robo_multi* multi, robo_gp* const* gps, robo_cand* const* cands, robo_cand* const* reps,
                                int32_t n_outcomes, double sn2, const double* logP, const double* lmb, const double* W,
                                const double* dlogPdMu, const double* dlogPdSigma, const double* dlogPdMudMu,
                                const int64_t* global_offsets, double* out_dh, double* out_max, int64_t* out_argmax,
                                int32_t* out_owner);
/* candidate shard of robo_ig_eval_per_cost_cand (BASELINE config 4); reps / cost_gps / cost_cands: per-device replicas     */
int32_t robo_ig_eval_per_cost_cand_multi(robo_multi* multi, robo_gp* const* gps, robo_cand* const* cands,
                                         robo_cand* const* reps, int32_t n_outcomes, double sn2, const double* logP,
                                         const double* lmb, const double* W, const double* dlogPdMu,
                                         const double* dlogPdSigma, const double* dlogPdMudMu, robo_gp* const* cost_gps,
                                         robo_cand* const* cost_cands, double overhead, const int64_t* global_offsets,
                                         double* out_values, double* out_max, int64_t* out_argmax, int32_t* out_owner);
/* sample shard of robo_acq_eval_marginal_cand: gps / etas [S_total] in device order, cands[g] = ALL m candidates on device g
 * (NULL allowed where S_dev[g] == 0, g > 0).  Partial sums are formed per device and added in device order: the result equals
 * robo_acq_eval_marginal_cand_sharded with the same shards bit for bit, and the single-device accumulation up to fp64
 * re-association.                                                                                                          */
int32_t robo_acq_eval_marginal_cand_multi(robo_multi* multi, robo_gp* const* gps, const int32_t* S_dev, int32_t acq_kind,
                                          double par, const double* etas, robo_cand* const* cands, double* out_acq,
                                          double* out_max, int64_t* out_argmax, uint32_t* out_flags);
/* sample shard of robo_gp_predict_mixture_cand: the per-sample posteriors are gathered on the first device in sample order
 * and mixed there by the same kernel: identical to the single-device mixture bit for bit.                                  */
int32_t robo_gp_predict_mixture_cand_multi(robo_multi* multi, robo_gp* const* gps, const int32_t* S_dev,
                                           robo_cand* const* cands, double* out_mean, double* out_var);

#ifdef __cplusplus
}
#endif
#endif /* ROBO_HIP_H */
