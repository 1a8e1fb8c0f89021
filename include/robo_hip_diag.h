/* librobo_hip_diag.so -- hardware self-checks and micro-benchmarks for the MI355X hot path of robo_amd.
 *
 * Measurement / test infrastructure, deliberately NOT in librobo_hip.so (the product library, include/robo_hip.h):
 * loaded by tests/, bench.py's roofline block and tools/ only.  Handles come from librobo_hip.so.            */
#ifndef ROBO_HIP_DIAG_H
#define ROBO_HIP_DIAG_H
#include "robo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- self tests / micro benchmarks (used by tests and bench.py, not by the product) --- */
/* runs one v_mfma_f64_16x16x4_f64 with asymmetric operands, returns max |D - A*B|          */
int32_t robo_selftest_mfma_layout(robo_ctx* ctx, double* out_max_err);
/* issues `iters` dependent-free MFMA f64 per wave on every CU; returns TFLOP/s               */
int32_t robo_microbench_mfma_f64(robo_ctx* ctx, int32_t iters, double* out_tflops);
/* shader-clock offsets of the phase boundaries of one potrf_diag_kernel (panel 0 of the gram
 * matrix at theta): load, potf2(0), sub-panel(0), steps 0..6, inverse, write-back [13]; the panel
 * kernel below it [4]; per interval s = 0..6 of the pivot wave: C1 + C2 done, barrier Bb passed,
 * potf2(s+1) done [21]                                                                        */
int32_t robo_selftest_diag_timeline(robo_gp* gp, const double* theta, double* out38);
/* out3[4] = {full-chip TFLOP/s, shader cycles per MFMA of one wave alone (8 independent
 * accumulators), shader MHz under load, cycles per MFMA in a fully dependent chain}            */
int32_t robo_microbench_mfma_f64_detail(robo_ctx* ctx, int32_t iters, double* out3);
/* GEMM-core microbenchmark in the shape of one posterior block-row step: wgs workgroups, each
 * C(128x128) = A_wg(128xK) B(128xK)^T.  variant 0 = LDS-staged core (gemm_f64.h), 1 = barrier-free
 * fragment streaming from a packed operand layout.  out2 = {TFLOP/s, shader MHz}.  Measurement only. */
int32_t robo_microbench_gemm_f64(robo_ctx* ctx, int32_t variant, int32_t wgs, int32_t k, int32_t reps,
                                 double* out2);

/* The stretch move of the hyper-parameter chain on arrays, through the very device functions the chain kernels inline
 * (csrc/mcmc_dev.h): z[i] = ((a - 1) u[i] + 1)^2 / a,  q[i] = c[i] - z[i] (c[i] - s[i]),
 * lnpdiff[i] = (P - 1) u[i] + c[i] - s[i] -- each operation rounded once, none fused, as NumPy / emcee 2 compute them
 * (/root/reference robo/models/gaussian_process_mcmc.py:126-135 drives emcee's _propose_stretch).  Tests compare the
 * three arrays with NumPy's bit for bit on the MI355X (round 5's q was contracted to a v_fma_f64).                 */
int32_t robo_selftest_stretch_move(robo_ctx* ctx, const double* c, const double* s, const double* u, double a, int32_t P,
                                   int32_t n, double* out_z, double* out_q, double* out_lnpdiff);

/* Shader clock while other work runs: _begin launches eight one-wave sampler workgroups on a private stream; each
 * sleeps through window_us of the 100 MHz wall clock and records the shader cycles that passed.  _end waits for them:
 * out3 = {mean, min, max} shader MHz.  bench.py brackets one posterior step with it, so that the roofline block can
 * state the fp64 peak AT THE CLOCK THE KERNEL RAN AT next to the nominal 2.4 GHz figure.                          */
int32_t robo_diag_clock_sample_begin(robo_ctx* ctx, int32_t window_us);
int32_t robo_diag_clock_sample_end(robo_ctx* ctx, double* out3);

#ifdef __cplusplus
}
#endif
#endif
