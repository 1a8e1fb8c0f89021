// Device-side scalar math: covariance functions and the normal-distribution pieces of
// EI / LogEI / PI.  Formulas are the ones oracle/gp_oracle.py pins against scipy.
#pragma once
#include "common.h"

namespace robo {

// k(r^2) for amp * Matern52Kernel / amp * ExpSquaredKernel, r^2 = sum_d (x_d - x'_d)^2 / m_d
// (SURVEY.md A.2).  r2 == 0 gives exactly amp.
__device__ __forceinline__ double cov_from_r2(int kind, double amp, double r2) {
    if (kind == ROBO_KERNEL_MATERN52_ARD) {
        const double s = sqrt(5.0 * r2);
        return amp * (1.0 + s + 5.0 * r2 / 3.0) * exp(-s);
    }
    return amp * exp(-0.5 * r2);
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
