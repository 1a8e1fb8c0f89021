// Device-side scalar math: covariance functions and the normal-distribution pieces of
// EI / LogEI / PI.  Formulas are the ones oracle/gp_oracle.py pins against scipy.
#pragma once
#include "common.h"

namespace robo {

// Covariance functions (SURVEY.md A.2 -- the project's stated contract for the un-vendored george):
//   MATERN52_ARD  amp (1 + sqrt(5 r2) + 5 r2 / 3) exp(-sqrt(5 r2)),  r2 = sum_d (x_d - x'_d)^2 / m_d
//   RBF_ARD       amp exp(-r2 / 2)
//   FABOLAS       amp prod_{d < D} matern52((x_d - x'_d)^2 / m_d) * (a + b u u'),  u = last coordinate
//                 (robo/fmin/fabolas.py:104-117: one 1-D Matern-5/2 per input dimension times the
//                  Bayesian-linear-regression kernel on the basis-transformed fidelity column)
// Inputs arrive pre-scaled by 1/sqrt(m_d) (the fidelity column unscaled).  A pair is reduced
// dimension by dimension: cov_init -> cov_step per dimension -> cov_finish.  T = double, or
// float for the mixed-precision K-build of BASELINE config 5 (widened to fp64 afterwards).
template <class T>
__device__ __forceinline__ T matern52_unit(T r2) {
    const T s = sqrt(T(5) * r2);
    return (T(1) + s + T(5) * r2 / T(3)) * exp(-s);
}

// fp64 specialisation.  K1 (gram_kernel) is bound by fp64 VALU issue, not by HBM (r02i PMC: 111 VALU instructions per
// pair, >= 61 % of the kernel's cycles with every one at full rate), and about half of those instructions were
// the library expansions of sqrt() and exp() with their denormal / overflow / special-value handling.  Here the
// argument ranges are known: t = 5 r2 >= 0 finite, and exp only ever sees -s in (-inf, 0].
//   sqrt:  y = v_rsq_f64(t) with one third-order correction, s0 = t y, s = s0 + (t - s0^2) y / 2   (<= 1 ulp; t = 0 -> 0)
//   exp:   n = rint(-s log2 e), r = -s - n ln2 (two-constant Cody-Waite), |r| <= 0.347, degree-12 Taylor
//          polynomial by Horner (remainder 0.347^13 / 13! = 1.7e-16), scaled by 2^n with v_ldexp_f64;
//          below -745 the result underflows to 0 like the library's.
// Same formula as before, so values agree with the oracle to a few ulp (tests: rtol 1e-13 on K).
__device__ __forceinline__ double pos_sqrt(double t) {
    double y = __builtin_amdgcn_rsq(t);                   // ~2^-23 relative
    const double e = fma(-t * y, y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);                // third-order correction: ~e^3
    const double s0 = t * y;
    const double s = fma(fma(-s0, s0, t), 0.5 * y, s0);   // final rounding step
    return t > 0.0 ? s : 0.0;                             // rsq(0) = inf: 0 * inf
}

__device__ __forceinline__ double exp_nonpos(double x) {   // x <= 0
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);     // ln2 hi (low bits zero: n * hi is exact)
    r = fma(n, -1.90821492927058770002e-10, r);            // ln2 lo
    double p = 2.08767569878680989792e-09;                  // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);              // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);              // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);              // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);              // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);              // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);              // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);              // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);              // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);              // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return x < -745.2 ? 0.0 : ldexp(p, (int)n);
}

// exp of a non-positive argument: the trimmed fp64 form above, the library's for fp32
template <class T>
__device__ __forceinline__ T exp_np(T x) {
    return exp(x);
}
#ifndef ROBO_TRIMMED_MATH
#define ROBO_TRIMMED_MATH 1     // 0: the library's sqrt()/exp() everywhere (A/B builds)
#endif
#if ROBO_TRIMMED_MATH
template <>
__device__ __forceinline__ double exp_np<double>(double x) {
    return exp_nonpos(x);
}

template <>
__device__ __forceinline__ double matern52_unit<double>(double r2) {
    const double t = 5.0 * r2;
    const double s = pos_sqrt(t);
    return (1.0 + s + t * (1.0 / 3.0)) * exp_nonpos(-s);
}
#endif

// One factor of the Fabolas product kernel, matern52(df^2), in the distance itself: sqrt(5 df^2) = sqrt(5) |df| -- no
// square root (the v_rsq_f64 + correction sequence was 8 of the ~36 instructions per pair and dimension).  Split form
// for the tiled gram kernels: the polynomial factor and the exponent separately, so that a pair needs ONE exp over
// the sum of its exponents instead of one per dimension (another 20 of the 36):
//     prod_d (1 + s_d + s_d^2 / 3) exp(-s_d)  =  [prod_d (1 + s_d + s_d^2 / 3)] exp(-sum_d s_d),   s_d = sqrt(5) |df_d|
constexpr double SQRT5 = 2.23606797749978969641;
template <class T>
__device__ __forceinline__ T matern52_1d(T df) {
    const T s = T(SQRT5) * fabs(df);
    return (T(1) + s + s * s * T(1.0 / 3.0)) * exp_np(-s);
}
template <class T>
__device__ __forceinline__ void matern52_1d_split(T df, T& poly, T& expo) {
    const T s = T(SQRT5) * fabs(df);
    poly *= T(1) + s + s * s * T(1.0 / 3.0);
    expo += s;
}

// KIND < 0: decided at run time from p.kind (cold paths); otherwise compiled in (the tiled
// gram kernels are instantiated per kind: a run-time branch in their inner loop cost 2x)
template <class T, int KIND = -1>
__device__ __forceinline__ void cov_init(const CovParams& p, T& acc, T& uu) {
    const int kind = KIND < 0 ? p.kind : KIND;
    acc = kind == ROBO_KERNEL_FABOLAS ? T(1) : T(0);
    uu = T(0);
}

template <class T, int KIND = -1>
__device__ __forceinline__ void cov_step(const CovParams& p, int d, T xi, T xj, T& acc, T& uu) {
    const int kind = KIND < 0 ? p.kind : KIND;
    if (kind == ROBO_KERNEL_FABOLAS) {
        if (d == p.dim - 1) {
            uu = xi * xj;
        } else {
            acc *= matern52_1d(xi - xj);
        }
    } else {
        const T df = xi - xj;
        acc = fma(df, df, acc);
    }
}

template <class T, int KIND = -1>
__device__ __forceinline__ double cov_finish(const CovParams& p, T acc, T uu) {
    const int kind = KIND < 0 ? p.kind : KIND;
    if (kind == ROBO_KERNEL_MATERN52_ARD) return (double)(T(p.amp) * matern52_unit(acc));
    if (kind == ROBO_KERNEL_RBF_ARD) return (double)(T(p.amp) * exp_np(T(-0.5) * acc));
    return (double)(T(p.amp) * acc * (T(p.blr_a) + T(p.blr_b) * uu));
}

// prior variance k(x, x) of a (scaled) point whose fidelity coordinate is u
__device__ __forceinline__ double cov_self(const CovParams& p, double u) {
    return p.kind == ROBO_KERNEL_FABOLAS ? p.amp * (p.blr_a + p.blr_b * u * u) : p.amp;
}

// scalar path (cov_kernel): both points given as rows of `dim` scaled coordinates
__device__ __forceinline__ double cov_rows(const CovParams& p, const double* __restrict__ xi,
                                           const double* __restrict__ xj) {
    double acc, uu;
    cov_init(p, acc, uu);
    for (int d = 0; d < p.dim; ++d) cov_step(p, d, xi[d], xj[d], acc, uu);
    return cov_finish(p, acc, uu);
}

// map linear lower-triangular tile index to (bi, bj), bj <= bi
__device__ __forceinline__ void tri_tile(int t, int& bi, int& bj) {
    int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    while (i * (i + 1) / 2 > t) --i;
    bi = i;
    bj = t - i * (i + 1) / 2;
}

constexpr double SQRT1_2 = 0.70710678118654752440;
constexpr double SQRT_2PI = 2.50662827463100050242;
constexpr double LOG_SQRT_2PI = 0.91893853320467274178;

__device__ __forceinline__ double norm_cdf(double z) { return 0.5 * erfc(-z * SQRT1_2); }
__device__ __forceinline__ double norm_pdf(double z) { return exp(-z * z / 2.0) / SQRT_2PI; }
__device__ __forceinline__ double norm_logpdf(double z) { return -z * z / 2.0 - LOG_SQRT_2PI; }
// log Phi(z): erfcx form in the left tail (no underflow down to z ~ -1e150), log1p on the right
__device__ __forceinline__ double norm_logcdf(double z) {
    if (z < -1.0) return log(0.5 * erfcx(-z * SQRT1_2)) - 0.5 * z * z;
    return log1p(-0.5 * erfc(z * SQRT1_2));
}

// robo/acquisition_functions/ei.py:70-78 (the batch-level guards are applied by the host shim
// from the flags word)
__device__ __forceinline__ double acq_ei(double m, double v, double eta, double par) {
    const double s = sqrt(v);
    const double z = (eta - m - par) / s;
    return s * (z * norm_cdf(z) + norm_pdf(z));
}

// robo/acquisition_functions/log_ei.py:74-120, branch for branch
__device__ __forceinline__ double acq_log_ei(double m, double v, double eta, double par) {
    const double f_min = eta - par;
    const double s = sqrt(v);
    const double z = (f_min - m) / s;
    const double ninf = -__builtin_huge_val();
    if (fabs(f_min - m) == 0.0) return s > 0.0 ? log(s) + norm_logpdf(z) : ninf;
    if (s == 0.0) return m < f_min ? log(f_min - m) : ninf;
    const double b = log(s) + norm_logpdf(z);
    if (f_min > m) {
        const double a = log(f_min - m) + norm_logcdf(z);
        return fmax(a, b) + log(1.0 + exp(-fabs(b - a)));
    }
    const double a = log(m - f_min) + norm_logcdf(z);
    if (a >= b) return ninf;
    return b + log(1.0 - exp(a - b));
}

// robo/acquisition_functions/pi.py:57-63
__device__ __forceinline__ double acq_pi(double m, double v, double eta, double par) {
    return norm_cdf((eta - m - par) / sqrt(v));
}

// robo/acquisition_functions/lcb.py:65
__device__ __forceinline__ double acq_lcb(double m, double v, double par) { return -(m - par * sqrt(v)); }

}  // namespace robo
