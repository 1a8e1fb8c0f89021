// Entropy-search information gain, batched over candidates.
//
// Replaces the per-candidate Python loop of InformationGain.compute
// (robo/acquisition_functions/information_gain.py:87-125): for every candidate x the reference
// calls `innovations` (:253-272 -- one 1-point predict plus one (Nb+1)-point full-covariance
// predict, i.e. an N x (Nb+1) Cholesky solve per candidate) and `_dh_fun` (:169-203 -- Nb = 50,
// Np = 400 tensor algebra).  Here, for a whole candidate batch:
//   cross_cov_kernel   s_c = cov(x_c, z_b) = k(x_c, z_b) - v_c . v_zb   (fp64 MFMA, NT GEMM over the
//                      V = L^-1 K* rows the posterior kernels already produced; K = n_pad)
//   ig_features_kernel F_c[a Nb + b] = s_ca s_cb
//   gemm_nt_kernel     Q_c = F_c G^T: the two quadratic forms per belief location i,
//                      q1_i = s^T dlogPdMudMu_i s  and  q2_i = sum_{a>=b} dlogPdSigma_i[ab] s_a s_b
//                      (fp64 MFMA, K = Nb^2)
//   ig_dh_kernel       innovation scalings, predicted log p_min for the Np hallucinated outcomes,
//                      renormalisation (log-sum-exp), entropy change, mean over Np
// The EP state (logP, dlogPdMu, dlogPdSigma, dlogPdMudMu) comes from the host
// (robo_amd/util/epmgp.py): Nb = 50 sequential site updates, once per update().
#include "common.h"
#include "gemm_f64.h"
#include "kern_math.h"

namespace robo {

constexpr int IG_Q2_OFF = 64;   // column offset of the q2 block in the 128-wide GEMM output

// s[c][b] = max(eps, y_std^2 (k(x_c, z_b) - v_c . v_zb))  for b < nb, 0 beyond
// (the clip at eps is the reference's: predict(full_cov=True) clips the WHOLE covariance matrix,
//  off-diagonals included, gaussian_process.py:290-294, and predict_variance reads it back)
// TM: 32 TM candidates per workgroup -- a batch of 8192 is only 64 tiles of 128 (0.75 ms of a config-4 step with
// three quarters of the chip idle, r03e); the k-order of every entry's products is the same, hence the same bits
template <int TM>
__global__ __launch_bounds__(256) void cross_cov_kernel(const double* __restrict__ Vc, int ldv,
                                                        const double* __restrict__ Vr, int ldr, int kend,
                                                        const double* __restrict__ Xcs, long long c0,
                                                        const double* __restrict__ Xrs, int nb, CovParams cp,
                                                        double y_std, double* __restrict__ S) {
    __shared__ double smem[gemm_smem_doubles<TM>()];
    const long long r0 = (long long)blockIdx.x * (32 * TM);
    AccT<TM> acc;
    acc_zero(acc);
    gemm_nt<TM, false>(Vc + (size_t)r0 * ldv, ldv, Vr, ldr, 0, kend, acc, smem);
    const double eps = 2.220446049250313e-16;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long a = r0 + acc_row<TM>(tm, r);
                const int b = acc_col(tn);
                double v = 0.0;
                if (b < nb) {
                    v = (cov_rows(cp, Xcs + (size_t)(c0 + a) * cp.dim, Xrs + (size_t)b * cp.dim) - acc.t[tm][tn][r]) *
                        (y_std * y_std);
                    v = v < eps ? eps : v;
                }
                S[(size_t)(c0 + a) * NB + b] = v;
            }
}

// F[c][a nb + b] = s_ca s_cb, zero padded to kf columns
__global__ __launch_bounds__(256) void ig_features_kernel(const double* __restrict__ S, long long c0, long long rows,
                                                          int nb, int kf, double* __restrict__ F) {
    const long long total = rows * kf;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long c = i / kf;
        const int e = (int)(i - c * kf);
        double v = 0.0;
        if (e < nb * nb) {
            const int a = e / nb, b = e - a * nb;
            const double* s = S + (size_t)(c0 + c) * NB;
            v = s[a] * s[b];
        }
        F[i] = v;
    }
}

// C (rows x 128) = A (rows x K) * B (128 x K)^T; 32 TM rows per workgroup (TM = 1 while 128-row tiles would leave CUs
// idle: 8192 candidates are 64 of them; same k-order per entry, same bits)
template <int TM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const double* __restrict__ A, int lda,
                                                      const double* __restrict__ B, int ldb, int K,
                                                      double* __restrict__ C, int ldc) {
    __shared__ double smem[gemm_smem_doubles<TM>()];
    const size_t r0 = (size_t)blockIdx.x * (32 * TM);
    AccT<TM> acc;
    acc_zero(acc);
    gemm_nt<TM, false>(A + r0 * lda, lda, B, ldb, 0, K, acc, smem);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(r0 + acc_row<TM>(tm, r)) * ldc + acc_col(tn)] = acc.t[tm][tn][r];
}

// one workgroup per candidate.  consts: [logP (64) | lmb (64) | W (npts) | dlogPdMu (nb x nb)]
__global__ __launch_bounds__(256) void ig_dh_kernel(const double* __restrict__ S, const double* __restrict__ var,
                                                    const double* __restrict__ Q, long long q_row0, long long c0,
                                                    long long m, int nb, int npts, double sn2, double H,
                                                    const double* __restrict__ consts, double* __restrict__ out) {
    __shared__ double sa[64], ssto[64], ss[64], slmb[64];
    __shared__ double sred[256];
    __shared__ int sflag;
    const long long c = c0 + blockIdx.x;
    if (c >= m) return;
    const int tid = threadIdx.x;
    const double* logP = consts;
    const double* lmb = consts + 64;
    const double* W = consts + 128;
    const double* dMu = consts + 128 + npts;
    const double* s = S + (size_t)c * NB;
    const double* q = Q + (size_t)(q_row0 + blockIdx.x) * NB;
    // innovations (information_gain.py:253-272): v = predictive variance at x, v_ = v - sn2 (sic),
    // norm_cov = s / v_, dM = norm_cov sqrt(v + 1e-10), dV = -norm_cov s^T
    const double v = var[c];
    const double v_ = v - sn2;
    const double sc_m = sqrt(v + 1e-10) / v_;          // dM_j = sc_m s_j
    if (tid < 64) ss[tid] = tid < nb ? s[tid] : 0.0;
    if (tid == 0) sflag = 0;
    __syncthreads();
    if (tid < 64) {
        double a = 0.0, st = 0.0;
        if (tid < nb) {
            double dot = 0.0;
            for (int j = 0; j < nb; ++j) dot = fma(dMu[tid * nb + j], ss[j], dot);
            st = sc_m * dot;
            // trterm_i = dM^T dlogPdMudMu_i dM = sc_m^2 q1_i ; dlogPdSigma_i . vec(dV) = -q2_i / v_
            a = logP[tid] + (-q[IG_Q2_OFF + tid] / v_) + 0.5 * (sc_m * sc_m) * q[tid];
        }
        sa[tid] = a;
        ssto[tid] = st;
        slmb[tid] = tid < nb ? lmb[tid] : 0.0;
    }
    __syncthreads();
    // first pass: per-outcome log-sum-exp; the reference falls back to the column maxima for ALL
    // outcomes if ANY log-sum-exp is infinite (information_gain.py:190-192)
    double lse[2], mx[2];
    int cnt = 0;
    bool bad = false;
    for (int p = tid; p < npts; p += 256, ++cnt) {
        const double w = W[p];
        double m_ = -__builtin_huge_val();
        for (int i = 0; i < nb; ++i) m_ = fmax(m_, sa[i] + ssto[i] * w);
        double sum = 0.0;
        for (int i = 0; i < nb; ++i) sum += exp(sa[i] + ssto[i] * w - m_);
        const double l = m_ + log(sum);
        mx[cnt] = m_;
        lse[cnt] = l;
        if (isinf(l)) bad = true;
    }
    if (bad) sflag = 1;
    __syncthreads();
    const bool use_max = sflag != 0;
    double part = 0.0;
    cnt = 0;
    for (int p = tid; p < npts; p += 256, ++cnt) {
        const double w = W[p];
        const double lsel = use_max ? mx[cnt] : lse[cnt];
        double acc = 0.0;
        for (int i = 0; i < nb; ++i) {
            const double lp = sa[i] + ssto[i] * w - lsel;
            acc += exp(lp) * (lp + slmb[i]);
        }
        part += acc + H;          // dHp_p = sum_i exp(lp)(lp + lmb) + H
    }
    sred[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sred[tid] += sred[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        double dh = sred[0] / (double)npts;
        if (isnan(dh) || (isinf(dh) && dh > 0)) dh = -1.7976931348623157e308;   // information_gain.py:119-120
        out[c] = dh;
    }
}

int launch_cross_cov(robo_gp* gp, robo_cand* cand, robo_cand* rep, int64_t c0, int64_t cn, double* d_S) {
#define ROBO_CC_CALL(TM)                                                                                         \
    hipLaunchKernelGGL(cross_cov_kernel<TM>, dim3((unsigned)(cn / (32 * TM))), dim3(256), 0, gp->ctx->stream,  \
                       (const double*)cand->d_V, gp->n_pad, (const double*)rep->d_V, gp->n_pad,                 \
                       (gp->n + NB - 1) / NB * NB, (const double*)cand->d_Xcs, (long long)c0,                   \
                       (const double*)rep->d_Xcs, (int)rep->m, gp->cov, gp->y_std, d_S)
    if (cn / NB >= 2 * gp->ctx->num_cu) ROBO_CC_CALL(4);
    else ROBO_CC_CALL(1);
#undef ROBO_CC_CALL
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_ig_dh(robo_ctx* ctx, const double* d_S, const double* d_var, double* d_F, double* d_Q, const double* d_G,
                 const double* d_consts, int64_t c0, int64_t cn, int64_t m, int nb, int npts, int kf, double sn2,
                 double H, double* d_out) {
    long long total = (long long)cn * kf;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(ig_features_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_S, (long long)c0, (long long)cn,
                       nb, kf, d_F);
    if (cn / NB >= 2 * ctx->num_cu)
        hipLaunchKernelGGL(gemm_nt_kernel<4>, dim3((unsigned)(cn / NB)), dim3(256), 0, ctx->stream, (const double*)d_F, kf,
                           d_G, kf, kf, d_Q, NB);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<1>, dim3((unsigned)(cn / 32)), dim3(256), 0, ctx->stream, (const double*)d_F, kf,
                           d_G, kf, kf, d_Q, NB);
    int64_t live = m - c0 < cn ? m - c0 : cn;
    if (live > 0)
        hipLaunchKernelGGL(ig_dh_kernel, dim3((unsigned)live), dim3(256), 0, ctx->stream, d_S, d_var,
                           (const double*)d_Q, (long long)0, (long long)c0, (long long)m, nb, npts, sn2, H, d_consts,
                           d_out);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// information gain per unit cost (robo/acquisition_functions/information_gain_per_unit_cost.py:91-104):
//     acquisition_value = dh / (exp(log_cost) + overhead)
// with log_cost the cost model's predictive mean; in place on the gains
__global__ __launch_bounds__(256) void per_cost_kernel(double* __restrict__ dh, const double* __restrict__ log_cost,
                                                       double overhead, long long m) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dh[i] = dh[i] / (exp(log_cost[i]) + overhead);
}

int launch_per_cost(robo_ctx* ctx, double* d_dh, const double* d_log_cost, double overhead, int64_t m) {
    hipLaunchKernelGGL(per_cost_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_dh, d_log_cost,
                       overhead, (long long)m);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
