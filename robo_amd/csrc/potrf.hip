// K2 blocked Cholesky (right-looking, panel width NB = 128) and K3 log-likelihood.
//
// Replaces the factorisation inside george.GP.compute and GP.log_likelihood
// (reference call sites robo/models/gaussian_process.py:119,155,159;
//  robo/models/gaussian_process_mcmc.py:195,200-202).
//
// Per panel k (block column k of the n_pad x n_pad lower matrix, in place):
//   potrf_diag_kernel   1 workgroup : L_kk = chol(A_kk), W_k = L_kk^-1            (LDS resident)
//   potrf_panel_kernel  nb-k-1 WGs  : A_ik <- A_ik * W_k^T          (fp64 MFMA, gemm_f64.h)
//   potrf_syrk_kernel   tri tiles   : A_ij <- A_ij - A_ik * A_jk^T  (fp64 MFMA, gemm_f64.h)
// Because row n of the matrix is the augmented right-hand side (gram.hip), the finished
// factor holds z = L^-1 (y - mean) in row n: loglik_kernel only reduces z.z and log diag.
//
// Flops: n_pad^3/3 (+ n_pad*NB^2 for the explicit block inverses); the trailing update
// is the MFMA-bound part, the 128-wide diagonal kernel the latency-bound part.
#include "common.h"
#include "gemm_f64.h"

namespace robo {

// ------------------------------------------------------------------------------------
// Diagonal block: 128x128, processed as 8x8 sub-blocks of 16x16 kept block-packed in LDS
// (lower blocks only: 36 blocks each for L and for W = L^-1).
// ------------------------------------------------------------------------------------
constexpr int SB = 16;                 // sub-block edge
constexpr int NSB = NB / SB;           // 8
constexpr int NBLK = NSB * (NSB + 1) / 2;   // 36
constexpr int BLK = SB * SB;           // 256 doubles

__device__ __forceinline__ int blk_off(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * BLK; }

// LDS ops of one wave execute in order; this only stops the compiler from moving a
// cross-lane LDS read above the write it depends on (no instruction is emitted).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// C-layout <-> LDS 16x16 block (row-major, ld 16)
__device__ __forceinline__ v4d blk_load_c(const double* b, int lane) {
    v4d c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = b[((lane >> 4) + 4 * r) * SB + (lane & 15)];
    return c;
}
__device__ __forceinline__ void blk_store_c(double* b, int lane, v4d c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) b[((lane >> 4) + 4 * r) * SB + (lane & 15)] = c[r];
}
// acc += sgn * A(16x16) * B^T(16x16)   ("NT": both blocks indexed [row][k])
template <bool NEG>
__device__ __forceinline__ v4d blk_mma_nt(const double* A, const double* B, int lane, v4d acc) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        double a = A[(lane & 15) * SB + kk * 4 + (lane >> 4)];
        const double b = B[(lane & 15) * SB + kk * 4 + (lane >> 4)];
        if (NEG) a = -a;
        acc = mfma_f64(a, b, acc);
    }
    return acc;
}
// acc += sgn * A(16x16) * B(16x16)     ("NN": B indexed [k][col])
template <bool NEG>
__device__ __forceinline__ v4d blk_mma_nn(const double* A, const double* B, int lane, v4d acc) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        double a = A[(lane & 15) * SB + kk * 4 + (lane >> 4)];
        const double b = B[(kk * 4 + (lane >> 4)) * SB + (lane & 15)];
        if (NEG) a = -a;
        acc = mfma_f64(a, b, acc);
    }
    return acc;
}

// broadcast lane `src`'s double to the whole wave through SGPRs (v_readlane_b32 x2; `src` is a
// compile-time constant after unrolling) -- no LDS round trip, unlike ds_bpermute
__device__ __forceinline__ double bcast_lane(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// One wave: unblocked Cholesky of the 16x16 block Ld (lower part valid) fused with its
// inverse.  Lane i (mod 16) owns row i of L and column i of W = L^-1 in registers; columns
// are exchanged with readlane broadcasts.  The inverse rides in the same unrolled loop
// (row k of W only needs row k of L, complete after step k), so its broadcasts fill the
// latency gaps of the pivot chain (rsqrt -> scale -> rank-1 update -> next pivot).
// g0 = global index of the block's first row; rows >= n_real have their pivot forced to 1
// (augmented row and identity padding).  Writes L (upper zeroed) to Ld, L^-1 (upper
// zeroed) to Wd.  Returns the first failing global column + 1, or 0.
__device__ __forceinline__ int potf2_inv_16(double* Ld, double* Wd, int lane, int g0, int n_real) {
    const int row = lane & 15;
    double a[SB], x[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) a[j] = j <= row ? Ld[row * SB + j] : 0.0;
    int fail = 0;
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        double p = bcast_lane(a[k], k);   // pivot after the previous rank-1 updates
        const bool forced = g0 + k >= n_real;
        if (forced) p = 1.0;
        if (!(p > 0.0)) {                 // also catches NaN
            if (fail == 0) fail = g0 + k + 1;
            p = 1.0;
        }
        const double ri = rsqrt(p);
        const double lik = row == k ? p * ri : a[k] * ri;   // rows < k hold garbage here, never read
        a[k] = lik;
#pragma unroll
        for (int j = k + 1; j < SB; ++j) {
            const double ljk = bcast_lane(lik, j);
            a[j] = fma(-lik, ljk, a[j]);
        }
        // W[k][c] for this lane's column c = row:  (delta_kc - sum_{j<k} L[k][j] W[j][c]) / L[k][k]
        double acc = row == k ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j) {
            const double lkj = bcast_lane(a[j], k);         // L[k][j]: lane k, register j
            acc = fma(-lkj, x[j], acc);
        }
        x[k] = k < row ? 0.0 : acc * ri;
    }
    if (lane < SB) {
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            Ld[row * SB + j] = j <= row ? a[j] : 0.0;
            Wd[j * SB + row] = x[j];          // W[j][row]; zero for j < row
        }
    }
    return fail;
}

__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ K, int ld, int k, int n_real,
                                                         double* __restrict__ Linv, int* __restrict__ fail) {
    __shared__ double sL[NBLK * BLK];
    __shared__ double sW[NBLK * BLK];
    __shared__ double sT[4 * BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* Kd = K + ((size_t)k * NB) * ld + (size_t)k * NB;

    // ---- load the 36 lower sub-blocks ------------------------------------------------
    for (int bi = 0; bi < NSB; ++bi)
        for (int bj = 0; bj <= bi; ++bj)
            sL[blk_off(bi, bj) + tid] = Kd[(size_t)(bi * SB + (tid >> 4)) * ld + bj * SB + (tid & 15)];
    __syncthreads();

    // ---- right-looking factorisation over 16-wide sub-panels, with look-ahead: after the
    // sub-panel solve of step s, wave 0 updates the next diagonal block and factors it while
    // waves 1-3 apply the rest of the trailing update (the pivot chain is the critical path).
    if (wave == 0) {
        const int f = potf2_inv_16(sL + blk_off(0, 0), sW + blk_off(0, 0), lane, k * NB, n_real);
        if (f != 0 && lane == 0 && *fail == 0) *fail = f;
    }
    __syncthreads();
    for (int s = 0; s < NSB - 1; ++s) {
        // sub-panel: L_is = A_is * W_ss^T
        for (int bi = s + 1 + wave; bi < NSB; bi += 4) {
            double* A = sL + blk_off(bi, s);
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            acc = blk_mma_nt<false>(A, sW + blk_off(s, s), lane, acc);
            blk_store_c(A, lane, acc);   // in place: every operand read precedes the MFMA result
        }
        __syncthreads();
        // trailing: A_ij -= L_is * L_js^T  for s < j <= i
        const int rem = NSB - 1 - s;
        const int cnt = rem * (rem + 1) / 2;     // tile 0 is (s+1, s+1)
        if (wave == 0) {
            double* C = sL + blk_off(s + 1, s + 1);
            v4d acc = blk_load_c(C, lane);
            acc = blk_mma_nt<true>(sL + blk_off(s + 1, s), sL + blk_off(s + 1, s), lane, acc);
            blk_store_c(C, lane, acc);
            wave_lds_fence();
            const int f = potf2_inv_16(C, sW + blk_off(s + 1, s + 1), lane, k * NB + (s + 1) * SB, n_real);
            if (f != 0 && lane == 0 && *fail == 0) *fail = f;
        } else {
            for (int t = wave; t < cnt; t += 3) {      // t = 1 .. cnt-1 over waves 1..3
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
                const int jj = t - ii * (ii + 1) / 2;
                const int bi = s + 1 + ii, bj = s + 1 + jj;
                double* C = sL + blk_off(bi, bj);
                v4d acc = blk_load_c(C, lane);
                acc = blk_mma_nt<true>(sL + blk_off(bi, s), sL + blk_off(bj, s), lane, acc);
                blk_store_c(C, lane, acc);
            }
        }
        __syncthreads();
    }

    // ---- W = L^-1 block column by block column (columns are independent) -----------------
    // W_ij = -W_ii * sum_{k=j}^{i-1} L_ik W_kj ;  wave w handles columns w and 7 - w
    for (int pass = 0; pass < 2; ++pass) {
        const int j = pass == 0 ? wave : NSB - 1 - wave;
        double* T = sT + wave * BLK;
        for (int i = j + 1; i < NSB; ++i) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int kb = j; kb < i; ++kb) acc = blk_mma_nn<false>(sL + blk_off(i, kb), sW + blk_off(kb, j), lane, acc);
            blk_store_c(T, lane, acc);
            wave_lds_fence();
            v4d w = {0.0, 0.0, 0.0, 0.0};
            w = blk_mma_nn<true>(sW + blk_off(i, i), T, lane, w);
            blk_store_c(sW + blk_off(i, j), lane, w);
            wave_lds_fence();
        }
    }
    __syncthreads();

    // ---- write back: L into K (lower blocks), W as a dense 128x128 row-major block --------
    double* Wg = Linv + (size_t)k * NB * NB;
    for (int bi = 0; bi < NSB; ++bi)
        for (int bj = 0; bj < NSB; ++bj) {
            const int r = bi * SB + (tid >> 4), c = bj * SB + (tid & 15);
            if (bj <= bi) {
                Kd[(size_t)r * ld + c] = sL[blk_off(bi, bj) + tid];
                Wg[r * NB + c] = sW[blk_off(bi, bj) + tid];
            } else {
                Wg[r * NB + c] = 0.0;
            }
        }
}

// A_ik <- A_ik * W_k^T for the rows below the diagonal block of panel k, 32 rows per workgroup
// (4x more workgroups than 128-row tiles: with <= 32 block rows the panel would otherwise
// occupy an eighth of the chip for a full 512-MFMA-deep tile)
__global__ __launch_bounds__(256) void potrf_panel_kernel(double* __restrict__ K, int ld, int k,
                                                          const double* __restrict__ Linv) {
    __shared__ double smem[gemm_smem_doubles<1>()];
    const size_t row0 = (size_t)(k + 1) * NB + (size_t)blockIdx.x * 32;
    double* A = K + row0 * ld + (size_t)k * NB;
    const double* W = Linv + (size_t)k * NB * NB;
    AccT<1> acc;
    acc_zero(acc);
    gemm_nt<1, false>(A, ld, W, NB, 0, NB, acc, smem);
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) A[(size_t)acc_row<1>(0, r) * ld + acc_col(tn)] = acc.t[0][tn][r];
}

// A_ij <- A_ij - A_ik * A_jk^T   for k < j <= i (right-looking trailing update, K = 128).
// Tile height 32*TM is chosen per step by the launcher: a 128x128x128 tile is 512 MFMAs deep
// (>= 13.6 us per wave), so late steps with few blocks use shorter tiles to cover the chip.
// (A two-level variant with 512-deep updates was measured slower: its strip updates put
// <= 32 workgroups on the critical path.)
template <int TM>
__global__ __launch_bounds__(256) void potrf_syrk_kernel(double* __restrict__ K, int ld, int k) {
    __shared__ double smem[gemm_smem_doubles<TM>()];
    constexpr int SPLIT = 4 / TM;                    // row sub-tiles per 128-row block
    int ii, jj;
    {
        const int t = blockIdx.x / SPLIT;
        int q = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((q + 1) * (q + 2) / 2 <= t) ++q;
        while (q * (q + 1) / 2 > t) --q;
        ii = q;
        jj = t - q * (q + 1) / 2;
    }
    const int i = k + 1 + ii, j = k + 1 + jj, h = blockIdx.x % SPLIT;
    const size_t row0 = (size_t)i * NB + (size_t)h * (32 * TM);
    const double* A = K + row0 * ld + (size_t)k * NB;
    const double* B = K + ((size_t)j * NB) * ld + (size_t)k * NB;
    double* C = K + row0 * ld + (size_t)j * NB;
    AccT<TM> acc;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.t[tm][tn][r] = C[(size_t)acc_row<TM>(tm, r) * ld + acc_col(tn)];
    gemm_nt<TM, true>(A, ld, B, ld, 0, NB, acc, smem);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)acc_row<TM>(tm, r) * ld + acc_col(tn)] = acc.t[tm][tn][r];
}

// out[0] = z.z, out[1] = 2 sum_{i<n} log L_ii   (z = row n of the factor); fixed summation order
__global__ __launch_bounds__(256) void loglik_kernel(const double* __restrict__ K, int ld, int n,
                                                     double* __restrict__ out) {
    __shared__ double sq[4], sl[4];
    double q = 0.0, l = 0.0;
    const double* z = K + (size_t)n * ld;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double zi = z[i];
        q = fma(zi, zi, q);
        l += log(K[(size_t)i * ld + i]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        q += __shfl_xor(q, o);
        l += __shfl_xor(l, o);
    }
    if ((threadIdx.x & 63) == 0) {
        sq[threadIdx.x >> 6] = q;
        sl[threadIdx.x >> 6] = l;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (sq[0] + sq[1]) + (sq[2] + sq[3]);
        out[1] = 2.0 * ((sl[0] + sl[1]) + (sl[2] + sl[3]));
    }
}

int launch_potrf(robo_gp* gp) {
    robo_ctx* ctx = gp->ctx;
    const int ld = gp->n_pad, nb = gp->n_pad / NB;
    ROBO_HIP_CHECK(hipMemsetAsync(ctx->d_fail, 0, sizeof(int), ctx->stream));
    for (int k = 0; k < nb; ++k) {
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, ctx->stream, gp->d_K, ld, k, gp->n, gp->d_Linv,
                           ctx->d_fail);
        const int rem = nb - k - 1;
        if (rem > 0) {
            hipLaunchKernelGGL(potrf_panel_kernel, dim3(rem * 4), dim3(256), 0, ctx->stream, gp->d_K, ld, k,
                               (const double*)gp->d_Linv);
            const int blocks = rem * (rem + 1) / 2;
            // measured (N = 4096): 128-row tiles 43 us/step at 384..528 blocks, 64-row tiles slower
            // (55 us: B panel re-read twice), 32-row tiles 16 us vs 21 us once blocks < 96
            if (blocks >= 96)
                hipLaunchKernelGGL(potrf_syrk_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, gp->d_K, ld, k);
            else
                hipLaunchKernelGGL(potrf_syrk_kernel<1>, dim3(blocks * 4), dim3(256), 0, ctx->stream, gp->d_K, ld, k);
        }
    }
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_loglik(robo_gp* gp) {
    hipLaunchKernelGGL(loglik_kernel, dim3(1), dim3(256), 0, gp->ctx->stream, (const double*)gp->d_K, gp->n_pad,
                       gp->n, gp->ctx->d_scalars);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
