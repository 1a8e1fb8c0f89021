// K6 acquisition element-wise + argmax, K7 sample reduction, and the device-side
// candidate generator.
//
// Replaces EI/LogEI/PI/LCB.compute (robo/acquisition_functions/ei.py:65-88,
// log_ei.py:74-120 -- a Python per-point loop in the reference --, pi.py:57-63,
// lcb.py:62-65), the mean over hyper-parameter samples of
// MarginalizationGPMCMC.compute (marginalization.py:115-121) and the
// `X[y.argmax()]` of RandomSampling.maximize (robo/maximizers/random_sampling.py:48-50).
//
// argmax semantics = np.argmax: first index of the maximum; NaN is maximal.  Reductions
// are wavefront-shuffle trees with an index tie-break, so the result does not depend on
// the launch geometry.
#include "common.h"
#include "kern_math.h"

namespace robo {

struct Best {
    double v;
    long long i;   // -1 = empty
};

__device__ __forceinline__ bool better(const Best& a, const Best& b) {
    if (a.i < 0) return false;
    if (b.i < 0) return true;
    const bool an = isnan(a.v), bn = isnan(b.v);
    if (an != bn) return an;
    if (an) return a.i < b.i;
    if (a.v != b.v) return a.v > b.v;
    return a.i < b.i;
}

__device__ __forceinline__ Best block_best(Best x, Best* sh) {
    for (int o = 32; o > 0; o >>= 1) {
        Best y;
        y.v = __shfl_xor(x.v, o);
        y.i = __shfl_xor(x.i, o);
        if (better(y, x)) x = y;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    Best r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
        if (better(sh[w], r)) r = sh[w];
    return r;
}

// mode 0: acq[i] = a         (plain)
// mode 1: sum[i]  = a        (first hyper-parameter sample)
// mode 2: sum[i] += a        (next samples; fixed sample order = NumPy's axis-0 accumulation)
__global__ __launch_bounds__(256) void acq_kernel(const double* __restrict__ mean, const double* __restrict__ var,
                                                  double* __restrict__ acq, double* __restrict__ acq_sum, long long m,
                                                  int kind, double par, double eta, int mode,
                                                  double* __restrict__ part_val, long long* __restrict__ part_idx,
                                                  unsigned* __restrict__ flags) {
    __shared__ Best sh[4];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    Best b;
    b.v = 0.0;
    b.i = -1;
    unsigned f = 0;
    if (i < m) {
        const double mu = mean[i], v = var[i];
        double a;
        if (kind == ROBO_ACQ_EI) {
            a = acq_ei(mu, v, eta, par);
            // z*Phi(z) + phi(z) cancels to O(phi/z^2); below DBL_MIN (z < -37.5) the sign of the
            // rounded sum is libm noise.  Such values are returned as +0 instead of tripping the
            // reference's `EI < 0 -> ValueError` guard (ei.py:86-88); see DESIGN.md "Conscious fixes".
            if (a < 0.0 && a > -2.2250738585072014e-308) a = 0.0;
            if (a < 0.0) f |= ROBO_FLAG_NEGATIVE_EI;
        } else if (kind == ROBO_ACQ_LOG_EI) {
            a = acq_log_ei(mu, v, eta, par);
        } else if (kind == ROBO_ACQ_PI) {
            a = acq_pi(mu, v, eta, par);
        } else {
            a = acq_lcb(mu, v, par);
        }
        if (sqrt(v) == 0.0) f |= ROBO_FLAG_ZERO_SIGMA;
        if (isnan(a)) f |= ROBO_FLAG_NAN;
        if (mode == 0) acq[i] = a;
        else if (mode == 1) acq_sum[i] = a;
        else acq_sum[i] += a;
        b.v = a;
        b.i = i;
    }
    unsigned long long any = __ballot(f != 0);
    if (any != 0ull && f != 0) atomicOr(flags, f);
    const Best r = block_best(b, sh);
    if (threadIdx.x == 0) {
        part_val[blockIdx.x] = r.v;
        part_idx[blockIdx.x] = r.i;
    }
}

// vals[i] / div -> optional out[i]; per-block best of the divided values
__global__ __launch_bounds__(256) void scale_best_kernel(const double* __restrict__ vals, double* __restrict__ out,
                                                         long long m, double div, double* __restrict__ part_val,
                                                         long long* __restrict__ part_idx) {
    __shared__ Best sh[4];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    Best b;
    b.v = 0.0;
    b.i = -1;
    if (i < m) {
        const double a = vals[i] / div;
        if (out) out[i] = a;
        b.v = a;
        b.i = i;
    }
    const Best r = block_best(b, sh);
    if (threadIdx.x == 0) {
        part_val[blockIdx.x] = r.v;
        part_idx[blockIdx.x] = r.i;
    }
}

// single workgroup: best over the per-block partials -> part_val[n_part], part_idx[n_part]
__global__ __launch_bounds__(256) void best_final_kernel(double* __restrict__ part_val,
                                                         long long* __restrict__ part_idx, int n_part) {
    __shared__ Best sh[4];
    Best b;
    b.v = 0.0;
    b.i = -1;
    for (int p = threadIdx.x; p < n_part; p += 256) {
        Best c;
        c.v = part_val[p];
        c.i = part_idx[p];
        if (better(c, b)) b = c;
    }
    const Best r = block_best(b, sh);
    if (threadIdx.x == 0) {
        part_val[n_part] = r.v;
        part_idx[n_part] = r.i;
    }
}

// (max, argmax, flags) of a finished evaluation straight into pinned host memory; the flag word is cleared for the
// next evaluation.  One launch instead of three device-to-host copies and a memset: a quarter of the launches of an
// acquisition over a small batch (the reference's 500 candidates, its 1 x D single-point maximisers).
__global__ void report_best_kernel(const double* __restrict__ best_val, const long long* __restrict__ best_idx,
                                   unsigned* __restrict__ flags, double* __restrict__ host) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    host[0] = *best_val;
    reinterpret_cast<long long*>(host)[1] = *best_idx;
    reinterpret_cast<unsigned*>(host + 2)[0] = *flags;
    *flags = 0u;
}

int launch_report_best(robo_cand* cand, double* h_pinned) {
    hipLaunchKernelGGL(report_best_kernel, dim3(1), dim3(64), 0, cand->ctx->stream,
                       (const double*)(cand->d_part_val + cand->n_part),
                       (const long long*)(cand->d_part_idx + cand->n_part), cand->d_flags, h_pinned);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// GaussianProcessMCMC.predict mixture (robo/models/gaussian_process_mcmc.py:235-247):
//   m = mean_s mu_s ;  v = var_s(mu_s) + mean_s(var_s), floored at eps.
// Same operation order as NumPy on an (S, M) array reduced along axis 0: sequential sums over s,
// np.var as the mean of squared deviations from the (already rounded) mean.
__global__ __launch_bounds__(256) void mixture_kernel(const double* __restrict__ mu_all,
                                                      const double* __restrict__ var_all, long long stride, int S,
                                                      long long m, double* __restrict__ out_mean,
                                                      double* __restrict__ out_var) {
#pragma clang fp contract(off)   // NumPy rounds d*d before adding; an fma would differ in the last bit
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double sm = 0.0, sv = 0.0;
    for (int s = 0; s < S; ++s) {
        sm += mu_all[s * stride + i];
        sv += var_all[s * stride + i];
    }
    const double mean = sm / (double)S;
    double sd = 0.0;
    for (int s = 0; s < S; ++s) {
        const double d = mu_all[s * stride + i] - mean;
        sd += d * d;                      // NumPy: multiply(x, x) then add -- two roundings, no fma
    }
    double v = sd / (double)S + sv / (double)S;
    const double eps = 2.220446049250313e-16;
    v = v < eps ? eps : v;
    out_mean[i] = mean;
    out_var[i] = v;
}

int launch_mixture(robo_cand* cand, int S) {
    hipLaunchKernelGGL(mixture_kernel, dim3((unsigned)((cand->m + 255) / 256)), dim3(256), 0, cand->ctx->stream,
                       (const double*)cand->d_mu_all, (const double*)cand->d_var_all, (long long)cand->m_pad, S,
                       (long long)cand->m, cand->d_mean, cand->d_var);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// ---- Philox-4x32-10 counter-based uniforms (device-side candidate generation; the
// large-M maximiser row of SURVEY.md section 8f) ---------------------------------------------
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void uniform_kernel(double* __restrict__ out, long long m, long long m_pad, int dim,
                                                      unsigned long long seed) {
    const long long total = m_pad * dim;
    // one Philox block (4 x 32 bit) gives two 53-bit uniforms
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p * 2 < total;
         p += (long long)gridDim.x * blockDim.x) {
        unsigned c[4] = {(unsigned)p, (unsigned)(p >> 32), 0u, 0u};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const unsigned long long a = ((unsigned long long)c[0] << 32 | c[1]) >> 11;
        const unsigned long long b = ((unsigned long long)c[2] << 32 | c[3]) >> 11;
        const double ua = (double)a * (1.0 / 9007199254740992.0), ub = (double)b * (1.0 / 9007199254740992.0);
        const long long e = p * 2;
        out[e] = ua;
        if (e + 1 < total) out[e + 1] = ub;
    }
    (void)m;
}

// RandomSampling's candidate recipe generated on the device, in the GP's normalised input space
// (robo/maximizers/random_sampling.py:38-47): rows < n_uniform uniform in [0,1)^dim, the rest
// N(loc, scale) clipped to [0,1] (loc = normalised incumbent, scale = 0.1 / (upper - lower): the
// reference's absolute sigma of 0.1 in raw units).  Box-Muller on the Philox stream.
__global__ __launch_bounds__(256) void random_candidates_kernel(double* __restrict__ out, long long m_pad, int dim,
                                                                unsigned long long seed, long long n_uniform,
                                                                const double* __restrict__ loc,
                                                                const double* __restrict__ scale) {
    const long long total = m_pad * dim;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p * 2 < total;
         p += (long long)gridDim.x * blockDim.x) {
        unsigned c[4] = {(unsigned)p, (unsigned)(p >> 32), 0x52425321u, 0u};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const unsigned long long a = ((unsigned long long)c[0] << 32 | c[1]) >> 11;
        const unsigned long long b = ((unsigned long long)c[2] << 32 | c[3]) >> 11;
        const double ua = (double)a * (1.0 / 9007199254740992.0), ub = (double)b * (1.0 / 9007199254740992.0);
        // one Box-Muller pair from (ua, ub): used for elements that fall in the local cloud
        const double rad = sqrt(-2.0 * log(1.0 - ua)), ang = 6.283185307179586476925 * ub;
        const double g0 = rad * cos(ang), g1 = rad * sin(ang);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long e = p * 2 + h;
            if (e >= total) break;
            const long long row = e / dim;
            const int d = (int)(e - row * dim);
            double v = h == 0 ? ua : ub;
            if (row >= n_uniform) {
                v = loc[d] + scale[d] * (h == 0 ? g0 : g1);
                v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            }
            out[e] = v;
        }
    }
}

int launch_random_candidates(robo_ctx* ctx, double* d_out, int64_t m_pad, int dim, uint64_t seed, int64_t n_uniform,
                             const double* d_loc, const double* d_scale) {
    const long long pairs = ((long long)m_pad * dim + 1) / 2;
    int blocks = (int)((pairs + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(random_candidates_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_out, (long long)m_pad,
                       dim, (unsigned long long)seed, (long long)n_uniform, d_loc, d_scale);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_acq(robo_ctx* ctx, robo_cand* cand, int acq_kind, double par, double eta, bool accumulate, bool first) {
    const int blocks = (int)((cand->m + 255) / 256);
    const int mode = accumulate ? (first ? 1 : 2) : 0;
    hipLaunchKernelGGL(acq_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)cand->d_mean,
                       (const double*)cand->d_var, cand->d_acq, cand->d_acq_sum, (long long)cand->m, acq_kind, par,
                       eta, mode, cand->d_part_val, cand->d_part_idx, cand->d_flags);
    hipLaunchKernelGGL(best_final_kernel, dim3(1), dim3(256), 0, ctx->stream, cand->d_part_val,
                       cand->d_part_idx, blocks);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_argmax(robo_cand* cand, const double* d_vals, double div) {
    const int blocks = (int)((cand->m + 255) / 256);
    hipLaunchKernelGGL(scale_best_kernel, dim3(blocks), dim3(256), 0, cand->ctx->stream, d_vals, cand->d_acq,
                       (long long)cand->m, div, cand->d_part_val, cand->d_part_idx);
    hipLaunchKernelGGL(best_final_kernel, dim3(1), dim3(256), 0, cand->ctx->stream, cand->d_part_val,
                       cand->d_part_idx, blocks);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// Points first .. first + m - 1 of a (scrambled) Sobol' sequence, straight from the direction numbers:
//     x_k[d] = (shift[d] ^ XOR_{b : bit b of gray(k) is set} sv[d][b]) * 2^-bits,     gray(k) = k ^ (k >> 1)
// which is what SciPy's qmc.Sobol produces point after point (its _draw walks the Gray code incrementally); with
// sv / shift taken from a scipy engine the device sequence is bit-identical to engine.random(), any slice of it
// (fast_forward) included -- BASELINE config 5's 2^20 x 64 candidates never exist on the host and the 8 ranks of a
// candidate shard generate their own slices.  Pad rows replicate point `first`.
__global__ __launch_bounds__(256) void sobol_kernel(double* __restrict__ out, long long m, long long m_pad, int dim,
                                                    const unsigned long long* __restrict__ sv,
                                                    const unsigned long long* __restrict__ shift, int bits,
                                                    unsigned long long first, double scale) {
    const long long total = m_pad * dim;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / dim;
        const int d = (int)(e - row * dim);
        const unsigned long long k = first + (unsigned long long)(row < m ? row : 0);
        unsigned long long g = k ^ (k >> 1), x = shift[d];
        const unsigned long long* v = sv + (size_t)d * bits;
        while (g) {
            const int b = __ffsll((long long)g) - 1;
            x ^= v[b];
            g &= g - 1;
        }
        out[e] = (double)x * scale;
    }
}

int launch_sobol(robo_ctx* ctx, double* d_out, int64_t m, int64_t m_pad, int dim, const unsigned long long* d_sv,
                 const unsigned long long* d_shift, int bits, uint64_t first) {
    const long long total = (long long)m_pad * dim;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    double scale = 1.0;
    for (int b = 0; b < bits; ++b) scale *= 0.5;
    hipLaunchKernelGGL(sobol_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_out, (long long)m, (long long)m_pad, dim,
                       d_sv, d_shift, bits, (unsigned long long)first, scale);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_uniform(robo_ctx* ctx, double* d_out, int64_t m, int64_t m_pad, int dim, uint64_t seed) {
    const long long pairs = ((long long)m_pad * dim + 1) / 2;
    int blocks = (int)((pairs + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(uniform_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_out, (long long)m,
                       (long long)m_pad, dim, (unsigned long long)seed);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
