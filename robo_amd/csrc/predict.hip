// K5 batched predictive variance/mean: blocked forward substitution V = L^-1 K_*^T for a
// chunk of candidates, in place in the cross-gram workspace, plus the per-candidate
// reductions  q_c = |v_c|^2  and  mu_c = v_c . z  (z = L^-1 (y - mean), row n of L).
//
// Replaces george.GP.predict's  Kxs K^-1 Kxs^T  and  Kxs alpha
// (reference call site robo/models/gaussian_process.py:280-286) -- diagonal only:
//     var_c  = k(x_c, x_c) - q_c          mean_c = mu_c + mean
// (the reference builds the full M x M covariance and takes np.diag; the values are the
// same, the O(M^2) work is not done.)
//
// Layout: V is (chunk, n_pad) row-major, candidate-major ("Vt"): both operands of every product have the
// contraction index contiguous (the NT form of gemm_f64.h).
//   step i:  T   = V[:, blk i] - V[:, 0:i*128] * L[blk i, 0:i*128]^T      (K = i*128)
//            V_i = T * Linv_i^T                                           (the 36 lower 16x16 blocks of Linv_i)
// One launch per block row i (a launch boundary is the only inter-workgroup ordering the
// algorithm needs: step i reads columns < i*128 written by the SAME workgroup earlier).
// MFMA flops per candidate: n_pad^2 - n_pad*128 + 72 n_pad.
#include "common.h"
#include "gemm_f64.h"
#include "kern_math.h"

namespace robo {

// ---- block-row step with the cross-gram tile generated in registers, TRANSPOSED tile ---------------------------
// Per 128 candidates (one workgroup) and block row i the step is
//     T   = K*_i - V[:, 0:128 i] L[blk i, 0:128 i]^T          (128 candidates x 128 training points)
//     V_i = T Linv_ii^T
// The workgroup holds T TRANSPOSED in its accumulators: rows = training points, columns = candidates, every wave
// all 128 rows of its own 32 candidates (8 x 2 MFMA tiles).  In that orientation
//   * T never goes back to memory between the two products: register r of an accumulator tile holds rows
//     (lane >> 4) + 4 r, which is exactly the B operand of k-step r of  V_i^T = Linv_ii T^T  (fragment maps in
//     gemm_f64.h) -- the first version stored T (128 KB per workgroup), waited for the store and staged it back
//     through LDS (phase stamps r02z: 27k of 174k fixed cycles per block row, plus a 56k-cycle second GEMM);
//   * only the 36 lower-triangular 16x16 blocks of Linv_ii are multiplied (288 MFMAs per wave instead of 512); its
//     A-operand fragments come pre-packed from the fit (linv_pack_kernel: one coalesced 512-byte load per
//     fragment, read straight from L2 -- no LDS, no barrier in the solve);
//   * result row block cb (cb = 7 .. 0) is final as soon as its (cb + 1) column blocks are accumulated, so its
//     store and its share of the |v|^2, v.z reductions overlap the MFMAs of the next row block; with the pi16 row
//     permutation baked into the packed fragments a lane holds four consecutive entries of a candidate's row
//     (two 16-byte stores), and the reductions are register sums + two cross-lane adds.
// One launch per block row is kept on purpose: a fully fused (persistent, all block rows per workgroup) variant was
// measured 2x SLOWER -- the launch boundary keeps the 512 workgroups in lock-step on the same L block row, which is
// what makes L an L2 hit; free-running workgroups drift apart and stream L from the Infinity Cache instead (r01k:
// 36 ms vs 18.4 ms per 65 536 candidates).

struct AccTt {
    v4d t[8][2];   // [training-row block][candidate block of this wave]
};

// K*_i^T into the accumulators: rows = 128 scaled training points (Xt), columns = 128 scaled candidates (Xc);
// rows >= n_valid are written as 0.  smem: [GD][GXL] for each operand.
constexpr int GEN_GD = 16, GEN_GXL = NB + 2;     // dimensions per staged chunk, leading dimension of the [d][row] images

template <int KIND>
__device__ __forceinline__ void gen_tile_init(AccTt& acc) {
    const bool fab = KIND == ROBO_KERNEL_FABOLAS;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc.t[rb][mb] = fab ? v4d{1.0, 1.0, 1.0, 1.0} : v4d{0.0, 0.0, 0.0, 0.0};
}

// one staged chunk of dimensions [d0, d0 + 16): sC / sX hold candidates / training points as [d][row]; `wave` = which 32
// candidates of the tile this wave owns
template <int KIND>
__device__ __forceinline__ void gen_tile_accumulate(const CovParams& cp, const double* sC, const double* sX, int d0,
                                                    int wave, AccTt& acc) {
    const int lane = threadIdx.x & 63, dim = cp.dim;
    const bool fab = KIND == ROBO_KERNEL_FABOLAS;
    const int rbase = lane >> 4, cbase = wave * 32 + (lane & 15);
    const int dn = dim - d0 < GEN_GD ? dim - d0 : GEN_GD;
    for (int d = 0; d < dn; ++d) {
        if (fab && d0 + d == dim - 1) break;   // fidelity column: handled in the finish
        double xi[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) xi[mb] = sC[d * GEN_GXL + cbase + mb * 16];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double xj = sX[d * GEN_GXL + rb * 16 + rbase + 4 * r];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const double df = xi[mb] - xj;
                    if (fab) acc.t[rb][mb][r] *= matern52_1d(df);
                    else acc.t[rb][mb][r] = fma(df, df, acc.t[rb][mb][r]);
                }
            }
    }
}

// covariance function of the accumulated distances; the last staged chunk still holds the fidelity column (dim - 1)
// for the Fabolas kernel; rows >= n_valid become 0
template <int KIND>
__device__ __forceinline__ void gen_tile_finish(const CovParams& cp, const double* sC, const double* sX, int n_valid,
                                                int wave, AccTt& acc) {
    const int lane = threadIdx.x & 63, dim = cp.dim;
    const bool fab = KIND == ROBO_KERNEL_FABOLAS;
    const int rbase = lane >> 4, cbase = wave * 32 + (lane & 15);
    const int dl = (dim - 1) % GEN_GD;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rb * 16 + rbase + 4 * r;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                double uu = 0.0;
                if (fab) uu = sC[dl * GEN_GXL + cbase + mb * 16] * sX[dl * GEN_GXL + row];
                const double v = cov_finish<double, KIND>(cp, acc.t[rb][mb][r], uu);
                acc.t[rb][mb][r] = row < n_valid ? v : 0.0;
            }
        }
}

// K*_i^T into the accumulators: rows = 128 scaled training points (Xt), columns = 128 scaled candidates (Xc);
// rows >= n_valid are written as 0.  smem: [GD][GXL] for each operand.  256 threads.
template <int KIND>
__device__ __forceinline__ void gen_cross_tile_t(const CovParams& cp, const double* __restrict__ Xc,
                                                 const double* __restrict__ Xt, int n_valid, double* smem,
                                                 AccTt& acc) {
    constexpr int GD = GEN_GD, GXL = GEN_GXL;
    double* sC = smem;
    double* sX = smem + GD * GXL;
    const int t = threadIdx.x, wave = t >> 6, dim = cp.dim;
    gen_tile_init<KIND>(acc);
    for (int d0 = 0; d0 < dim; d0 += GD) {
        __syncthreads();   // previous chunk (or previous user of smem) fully consumed
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = t + e * 256;          // 128 rows x 16 dims
            const int row = idx >> 4, d = idx & 15;
            const bool ok = d0 + d < dim;
            sC[d * GXL + row] = ok ? Xc[(size_t)row * dim + d0 + d] : 0.0;
            sX[d * GXL + row] = ok ? Xt[(size_t)row * dim + d0 + d] : 0.0;
        }
        __syncthreads();
        gen_tile_accumulate<KIND>(cp, sC, sX, d0, wave, acc);
    }
    gen_tile_finish<KIND>(cp, sC, sX, n_valid, wave, acc);
    __syncthreads();   // smem free for the GEMM stages
}

// acc(8 x 2 tiles per wave) -= A[0:128, k-tile] * B[0:128, k-tile]^T restricted to the wave's 32 B rows
__device__ __forceinline__ void tile_mfma_t(const double* sA, const double* sB, AccTt& acc, int wave) {
    const int lane = threadIdx.x & 63;
    const double* pa = sA + (lane & 15) * LDS_LD + (lane >> 4);
    const double* pb = sB + (wave * 32 + (lane & 15)) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        double a[8], b[2];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) a[rb] = pa[rb * 16 * LDS_LD + kk * 4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) b[mb] = -pb[mb * 16 * LDS_LD + kk * 4];
        if (ROBO_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc.t[rb][mb] = mfma_f64(a[rb], b[mb], acc.t[rb][mb]);
        if (ROBO_SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
}

// acc -= A[:, 0:kend] * B[:, 0:kend]^T  (A: 128 rows of L, B: the workgroup's 128 rows of V; both k-contiguous);
// staging and pipeline of gemm_nt<4, .> (gemm_f64.h), wave tiling 1 x 4 instead of 2 x 2
__device__ __forceinline__ void gemm_t(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                       int kend, AccTt& acc, double* smem) {
    constexpr int SA = stage_a<4>(), ST = SA + STAGE_B;
    const int nk = kend / BK;
    if (nk <= 0) return;
    Tile4 ra = tile_load_regs<128>(A, lda, 0);
    Tile4 rb = tile_load_regs<128>(B, ldb, 0);
    tile_store_lds<128>(smem, ra);
    tile_store_lds<128>(smem + SA, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        double* cur = smem + (kt & 1) * ST;
        double* nxt = smem + ((kt + 1) & 1) * ST;
        const bool more = kt + 1 < nk;
        if (more) {
            ra = tile_load_regs<128>(A, lda, (kt + 1) * BK);
            rb = tile_load_regs<128>(B, ldb, (kt + 1) * BK);
        }
        tile_mfma_t(cur, cur + SA, acc, threadIdx.x >> 6);
        if (more) {
            tile_store_lds<128>(nxt, ra);
            tile_store_lds<128>(nxt + SA, rb);
        }
        __syncthreads();
    }
}

// V_i^T = Linv_ii T^T from the accumulators, stores and reductions (see the header of this section).
// Wp: packed fragments of block i; z: row n of L at column 128 i; Vt: the workgroup's V rows at column 128 i.
__device__ __forceinline__ void solve_store_reduce_t(const AccTt& T, const double* __restrict__ Wp,
                                                     const double* __restrict__ z, int n_valid, double* __restrict__ Vt,
                                                     int ldv, double* __restrict__ q, double* __restrict__ mu,
                                                     bool first, int wave) {
    constexpr int PF = 8;   // fragments in flight (L2 hits): 16 MFMAs = 1024 cycles of cover
    const int lane = threadIdx.x & 63, g = lane >> 4, nn = lane & 15;
    const double* wp = Wp + lane;
    double wf[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) wf[s] = wp[s * 64];
    double sq[2] = {0.0, 0.0}, sz[2] = {0.0, 0.0};
    double* vrow[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) vrow[mb] = Vt + (size_t)(wave * 32 + mb * 16 + nn) * ldv + 4 * g;
#pragma unroll
    for (int cbi = 0; cbi < 8; ++cbi) {
        const int cb = 7 - cbi;
        v4d o[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int jb = 0; jb <= cb; ++jb)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int s = wp_offset(cb) + 4 * jb + kk;
                const double a = wf[s % PF];
                if (s + PF < WP_FRAGS) wf[s % PF] = wp[(s + PF) * 64];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) o[mb] = mfma_f64(a, T.t[jb][mb][kk], o[mb]);
            }
        // rows 16 cb + 4 g + r (r = 0..3) of the result for candidates nn (+16) of this wave
        const int c0 = 16 * cb + 4 * g;
        double zc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) zc[r] = c0 + r < n_valid ? z[c0 + r] : 0.0;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = c0 + r < n_valid ? o[mb][r] : 0.0;
                sq[mb] = fma(v[r], v[r], sq[mb]);
                sz[mb] = fma(v[r], zc[r], sz[mb]);
            }
            double2* dst = reinterpret_cast<double2*>(vrow[mb] + 16 * cb);
            dst[0] = make_double2(v[0], v[1]);
            dst[1] = make_double2(v[2], v[3]);
        }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        sq[mb] += __shfl_xor(sq[mb], 16);
        sz[mb] += __shfl_xor(sz[mb], 16);
        sq[mb] += __shfl_xor(sq[mb], 32);
        sz[mb] += __shfl_xor(sz[mb], 32);
        if (g == 0) {
            const int c = wave * 32 + mb * 16 + nn;
            if (first) {
                q[c] = sq[mb];
                mu[c] = sz[mb];
            } else {
                q[c] += sq[mb];
                mu[c] += sz[mb];
            }
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void trsm_step_gen_kernel(const double* __restrict__ Xcs,
                                                               const double* __restrict__ Xs, double* __restrict__ V,
                                                               int ldv, const double* __restrict__ L, int ld,
                                                               const double* __restrict__ LinvP, int i0, int i1, int n,
                                                               double* __restrict__ q, double* __restrict__ mu,
                                                               long long c0, CovParams cp) {
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    double* Vrow = V + (size_t)blockIdx.x * NB * ldv;
    const long long cw = c0 + (long long)blockIdx.x * NB;   // first candidate of this workgroup
    // block rows i0 .. i1-1 in one launch (ROBO_TRSM_ROWS, default 1): a block row only reads columns this same
    // workgroup wrote, so the launch boundary is not needed for correctness -- it is what keeps the workgroups in
    // step on the same rows of L (see above); a few rows per launch trade a little of that for fewer launch
    // ramps/tails
    for (int i = i0; i < i1; ++i) {
        AccTt acc;
        gen_cross_tile_t<KIND>(cp, Xcs + (size_t)cw * cp.dim, Xs + (size_t)i * NB * cp.dim, n - i * NB, smem, acc);
        if (i > 0) gemm_t(L + (size_t)i * NB * ld, ld, Vrow, ldv, i * NB, acc, smem);
        solve_store_reduce_t(acc, LinvP + (size_t)i * WP_BLOCK, L + (size_t)n * ld + (size_t)i * NB, n - i * NB,
                             Vrow + (size_t)i * NB, ldv, q + cw, mu + cw, i == 0, threadIdx.x >> 6);
        __syncthreads();   // V_i (global, workgroup scope) before the next block row stages it
    }
}


// ---- TWO block rows per launch on ONE read of V (r03j..m) -- MEASURED SLOWER, kept as a tested option (tuning key
// trsm_pair, default 0) -------------------------------------------------------------------------------------------------
// The block-row step above re-reads, for block row i, the columns < 128 i of V = L^-1 K_*^T that the same workgroup
// wrote in earlier launches: 33 GB per 65 536-candidate evaluation at N = 4096 (2.2 TB/s of HBM for the whole step).
// Timing experiments with the one-row kernel (wrong numbers, same arithmetic, same-session A/B, profiles/r03m_*):
// every workgroup reading the V rows of eight fixed tiles (V served from L2) 16.0 instead of 17.4 ms; pairs / quadruples
// / octets of neighbouring workgroups reading the same rows (1/2, 1/4, 1/8 of the HBM stream) +2.6 / +3.7 / +4.1 %;
// V read as if it were stored k-tile-blocked (16 KB contiguous per k-tile instead of 128-byte row pieces) -1.4 %.  So
// the stream costs ~5 % under a kernel whose matrix pipes are busy 86 % of the time, and half of it is worth +2.6 %.
// THIS kernel halves it and lost 3 % (3.78 vs 3.90 M evals/s): one workgroup of eight waves per CU instead of two
// independent ones of four (every k-tile barrier now stops all eight waves of the CU, where two workgroups fill each
// other's bubbles), and a tail in which half of the waves wait (solve i -> last update of row i+1 -> solve i+1).
// Here a workgroup of eight waves owns 128 candidates and TWO block rows (i, i + 1): waves 0-3 hold T_i^T, waves 4-7
// T_{i+1}^T (each wave all 128 rows of its block row for its own 32 candidates, as above), and every staged k-tile of
// V (columns < 128 i) feeds both -- V is read once per pair of block rows, the launch count halves.  Then
//   waves 0-3:  V_i = T_i Linv_ii^T, stored;               (waves 4-7 wait)
//   all waves stage, waves 4-7 multiply:  T_{i+1} -= V_i L[i+1, i]^T   (128 deep; V_i comes back from L2)
//   waves 4-7:  V_{i+1} = T_{i+1} Linv_{i+1,i+1}^T, stored.
// Every entry receives the same products in the same (ascending k) order as in the one-row step: same bits.
constexpr int PAIR_ST = (2 * NB + NB) * LDS_LD;                 // doubles per stage: 256 rows of L | 128 rows of V
constexpr int PAIR_SMEM_DOUBLES = 2 * PAIR_ST;                  // 110.6 KB: one workgroup (8 waves) per CU

// k-tile loads with 512 threads, every thread active (no masked loads, no zero-filled registers):
//   256-row operand: thread t takes row t / 2, 8 doubles (half t % 2);  128-row operand: row t / 4, 4 doubles (quarter t % 4)
struct Tile2 {
    double2 a, b;
};
__device__ __forceinline__ Tile4 tile_load_256(const double* __restrict__ G, int ld, int k0) {
    const int row = threadIdx.x >> 1, kh = (threadIdx.x & 1) * 8;
    const double2* p = reinterpret_cast<const double2*>(G + (size_t)row * ld + k0 + kh);
    Tile4 t;
    t.a = p[0];
    t.b = p[1];
    t.c = p[2];
    t.d = p[3];
    return t;
}
__device__ __forceinline__ void tile_store_256(double* S, const Tile4 t) {
    const int row = threadIdx.x >> 1, kh = (threadIdx.x & 1) * 8;
    double2* p = reinterpret_cast<double2*>(S + row * LDS_LD + kh);
    p[0] = t.a;
    p[1] = t.b;
    p[2] = t.c;
    p[3] = t.d;
}
__device__ __forceinline__ Tile2 tile_load_128q(const double* __restrict__ G, int ld, int k0) {
    const int row = threadIdx.x >> 2, kq = (threadIdx.x & 3) * 4;
    const double2* p = reinterpret_cast<const double2*>(G + (size_t)row * ld + k0 + kq);
    Tile2 t;
    t.a = p[0];
    t.b = p[1];
    return t;
}
__device__ __forceinline__ void tile_store_128q(double* S, const Tile2 t) {
    const int row = threadIdx.x >> 2, kq = (threadIdx.x & 3) * 4;
    double2* p = reinterpret_cast<double2*>(S + row * LDS_LD + kq);
    p[0] = t.a;
    p[1] = t.b;
}

// acc -= A[this wave's 128 rows, 0:kend] * B[0:128, 0:kend]^T; pipeline of gemm_t with 512 threads.
// WIDE: A has 256 rows (both block rows; the wave multiplies rows a_row0 ..), else 128 (the last 128 columns of block
// row i + 1's update, multiplied by half 1 only).  MULTIPLY = false: the same loads, stores and barriers without the
// products -- the code path of the waves that only help staging (their accumulators are dead by then, and a separate
// path is what tells the register allocator so).
template <bool WIDE, bool MULTIPLY>
__device__ __forceinline__ void gemm_pair(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                          int kend, AccTt* acc, double* smem, int a_row0, int wq) {
    constexpr int SA = 2 * NB * LDS_LD;
    const int nk = kend / BK;
    if (nk <= 0) return;
    Tile4 ra;
    Tile2 ra2, rb = tile_load_128q(B, ldb, 0);
    if (WIDE) ra = tile_load_256(A, lda, 0);
    else ra2 = tile_load_128q(A, lda, 0);
    if (WIDE) tile_store_256(smem, ra);
    else tile_store_128q(smem, ra2);
    tile_store_128q(smem + SA, rb);
    __syncthreads();
#pragma unroll 1   // (the narrow call has a compile-time trip count of 8: fully unrolled it spilled ~1000 registers)
    for (int kt = 0; kt < nk; ++kt) {
        double* cur = smem + (kt & 1) * PAIR_ST;
        double* nxt = smem + ((kt + 1) & 1) * PAIR_ST;
        const bool more = kt + 1 < nk;
        if (more) {
            if (WIDE) ra = tile_load_256(A, lda, (kt + 1) * BK);
            else ra2 = tile_load_128q(A, lda, (kt + 1) * BK);
            rb = tile_load_128q(B, ldb, (kt + 1) * BK);
        }
        if (MULTIPLY) tile_mfma_t(cur + a_row0 * LDS_LD, cur + SA, *acc, wq);
        if (more) {
            if (WIDE) tile_store_256(nxt, ra);
            else tile_store_128q(nxt, ra2);
            tile_store_128q(nxt + SA, rb);
        }
        __syncthreads();
    }
}

template <int KIND>
__global__ __launch_bounds__(512, 1) void trsm_pair_gen_kernel(const double* __restrict__ Xcs,
                                                               const double* __restrict__ Xs, double* __restrict__ V,
                                                               int ldv, const double* __restrict__ L, int ld,
                                                               const double* __restrict__ LinvP, int i, int n,
                                                               double* __restrict__ q, double* __restrict__ mu,
                                                               long long c0, CovParams cp) {
    __shared__ double smem[PAIR_SMEM_DOUBLES];
    // (wave-uniform by construction; readfirstlane tells the compiler, so that the two halves' code paths are scalar
    // branches with disjoint register live ranges instead of exec-masked regions)
    const int t = threadIdx.x, wave8 = __builtin_amdgcn_readfirstlane(t >> 6), half = wave8 >> 2, wq = wave8 & 3,
              dim = cp.dim;
    double* Vrow = V + (size_t)blockIdx.x * NB * ldv;
    const long long cw = c0 + (long long)blockIdx.x * NB;
    const int row = i + half;                                   // this wave's block row
    AccTt acc;
    {   // ---- both cross-gram tiles: candidates staged once, the two blocks of training points side by side
        constexpr int GD = GEN_GD, GXL = GEN_GXL;
        double* sC = smem;
        double* sX = smem + GD * GXL;                           // [2][GD][GXL]
        const double* Xc = Xcs + (size_t)cw * dim;
        const double* Xt = Xs + (size_t)i * NB * dim;           // 256 consecutive training points
        gen_tile_init<KIND>(acc);
        for (int d0 = 0; d0 < dim; d0 += GD) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = t + e * 512, r = idx >> 4, d = idx & 15;     // 128 candidates x 16 dims
                sC[d * GXL + r] = d0 + d < dim ? Xc[(size_t)r * dim + d0 + d] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = t + e * 512, r = idx >> 4, d = idx & 15;     // 256 training points x 16 dims
                sX[(r >> 7) * GD * GXL + d * GXL + (r & 127)] = d0 + d < dim ? Xt[(size_t)r * dim + d0 + d] : 0.0;
            }
            __syncthreads();
            gen_tile_accumulate<KIND>(cp, sC, sX + half * GD * GXL, d0, wq, acc);
        }
        gen_tile_finish<KIND>(cp, sC, sX + half * GD * GXL, n - row * NB, wq, acc);
        __syncthreads();
    }
    // ---- columns < 128 i of V against both block rows of L
    if (i > 0) gemm_pair<true, true>(L + (size_t)i * NB * ld, ld, Vrow, ldv, i * NB, &acc, smem, half * NB, wq);
    const double* Ltail = L + (size_t)(i + 1) * NB * ld + (size_t)i * NB;      // L[i+1, i]
    // ---- the tail.  ONE inlined copy of the solve, reached by half 0 right away and by half 1 after its last update
    // (two copies made the register allocator spill ~210 registers, a quarter of the accumulators inside the k-loop
    // above; with one copy whose accumulators die in it: none).  The hardware barrier only counts arrivals, so the two
    // halves meet at DIFFERENT s_barrier instructions: both execute 1 + 1 + 8 of them.
    //   half 0:                         solve row i   | B | stage-only loop (1 + 8 barriers)
    //   half 1:  | B | last 128 columns of row i+1's update (1 + 8 barriers; V_i comes back from L2)   solve row i+1
    if (half == 1) {
        __syncthreads();   // V_i and row i's share of q / mu are stored (global, workgroup scope)
        gemm_pair<false, true>(Ltail, ld, Vrow + (size_t)i * NB, ldv, NB, &acc, smem, 0, wq);
    }
    solve_store_reduce_t(acc, LinvP + (size_t)row * WP_BLOCK, L + (size_t)n * ld + (size_t)row * NB, n - row * NB,
                         Vrow + (size_t)row * NB, ldv, q + cw, mu + cw, row == 0, wq);
    if (half == 0) {
        __syncthreads();
        gemm_pair<false, false>(Ltail, ld, Vrow + (size_t)i * NB, ldv, NB, nullptr, smem, 0, wq);
    }
}

// ---- small candidate batches: 32 candidates per workgroup ---------------------------------------------------------
// With fewer than ~2 workgroups of 128 candidates per CU the step above is bound by ONE wave's chain of 512 i + 288
// MFMAs per block row (the reference's default 500 random candidates, the 50 representer points of entropy search,
// a Fabolas incumbent projection or an 8192-candidate shard of config 4 occupy 4, 1, 32 and 64 of 256 CUs and all
// take the same 11 ms at N = 4096).  Here a workgroup owns 32 candidates and its four waves split the 128 training
// rows of the tile (2 x 2 MFMA tiles each): 128 i MFMAs per wave and block row in the update; for the solve the tile
// goes through LDS once ([candidate block][row][16]: a fragment read is 16 consecutive doubles per lane group, two
// groups 128 bytes apart in bank space) and wave w forms result row blocks w and 7 - w (9 of the 36 sub-blocks each).
// Same arithmetic per element as the 128-candidate step -- products accumulate in the same order, and the |v|^2 /
// v.z reductions are redone from an LDS copy of the result in exactly its order -- so both give the same bits.
// MB = candidate blocks of 16 per workgroup: 2 (32 candidates), or 1 (16 candidates: twice the workgroups, used while
// that still means at most two per CU -- two short chains per CU overlap where one leaves the matrix pipe idle)
// DK = k-tiles of 16 per staging stage (2: one barrier per 32 contraction steps, but one workgroup per CU)
template <int MB, int DK> constexpr int small_stage() { return DK * (NB + 16 * MB) * LDS_LD; }   // DK x (L tile | V tile)
template <int MB, int DK> constexpr int small_smem_doubles() {   // T image + result image, or the two staging stages
    return 2 * MB * NB * 16 > 2 * small_stage<MB, DK>() ? 2 * MB * NB * 16 : 2 * small_stage<MB, DK>();
}

template <int MB>
struct AccS {
    v4d t[2][MB];   // [row block 2 wave + rbl][candidate block]
};

template <int KIND, int MB>
__device__ __forceinline__ void gen_cross_tile_s(const CovParams& cp, const double* __restrict__ Xc,
                                                 const double* __restrict__ Xt, int n_valid, double* smem,
                                                 AccS<MB>& acc) {
    constexpr int SC = 16 * MB, GD = 16, GXL = NB + 2, GCL = SC + 2;
    double* sX = smem;                 // [GD][GXL] training points
    double* sC = smem + GD * GXL;      // [GD][GCL] candidates
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, dim = cp.dim;
    const bool fab = KIND == ROBO_KERNEL_FABOLAS;
#pragma unroll
    for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc.t[rbl][mb] = fab ? v4d{1.0, 1.0, 1.0, 1.0} : v4d{0.0, 0.0, 0.0, 0.0};
    const int rbase = wave * 32 + (lane >> 4), cbase = lane & 15;
    for (int d0 = 0; d0 < dim; d0 += GD) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = t + e * 256, row = idx >> 4, d = idx & 15;
            sX[d * GXL + row] = d0 + d < dim ? Xt[(size_t)row * dim + d0 + d] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < MB; ++e) {
            const int idx = t + e * 256, row = idx >> 4, d = idx & 15;
            sC[d * GCL + row] = d0 + d < dim ? Xc[(size_t)row * dim + d0 + d] : 0.0;
        }
        __syncthreads();
        const int dn = dim - d0 < GD ? dim - d0 : GD;
        for (int d = 0; d < dn; ++d) {
            if (fab && d0 + d == dim - 1) break;
            double xi[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) xi[mb] = sC[d * GCL + cbase + mb * 16];
#pragma unroll
            for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double xj = sX[d * GXL + rbl * 16 + rbase + 4 * r];
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const double df = xi[mb] - xj;
                        if (fab) acc.t[rbl][mb][r] *= matern52_1d(df);
                        else acc.t[rbl][mb][r] = fma(df, df, acc.t[rbl][mb][r]);
                    }
                }
        }
    }
    const int dl = (dim - 1) % GD;
#pragma unroll
    for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rbl * 16 + rbase + 4 * r;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                double uu = 0.0;
                if (fab) uu = sC[dl * GCL + cbase + mb * 16] * sX[dl * GXL + row];
                const double v = cov_finish<double, KIND>(cp, acc.t[rbl][mb][r], uu);
                acc.t[rbl][mb][r] = row < n_valid ? v : 0.0;
            }
        }
    __syncthreads();
}

// acc -= A[128 rows, 0:kend] * B[16 MB rows, 0:kend]^T, wave w on rows 32 w .. 32 w + 31
template <int MB, int SMALL_DK>
__device__ __forceinline__ void gemm_s(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                       int kend, AccS<MB>& acc, double* smem) {
    // With 8 (MB = 1) or 16 MFMAs per wave and k-tile the loop is bound by the barrier -> LDS store -> fragment read
    // round trip per stage, not by the matrix pipe (per-launch trace: 1850 cycles per k-tile for 1024 of MFMA), so a
    // stage holds SMALL_DK k-tiles: with 2, half the round trips (M <= 2048 at N = 4096: 3.08 -> 2.54 ms) at 83-92 KB
    // of LDS, i.e. one workgroup per CU -- the launcher uses 1 where two workgroups per CU are needed.
    constexpr int SA = NB * LDS_LD, SC = 16 * MB, SUB = (NB + SC) * LDS_LD, SMALL_STAGE = small_stage<MB, SMALL_DK>();
    const int nst = kend / (BK * SMALL_DK), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // kend = 128 i
    if (nst <= 0) return;
    Tile4 ra[SMALL_DK], rb[SMALL_DK];
#pragma unroll
    for (int u = 0; u < SMALL_DK; ++u) {
        ra[u] = tile_load_regs<128>(A, lda, u * BK);
        rb[u] = tile_load_regs<SC>(B, ldb, u * BK);
    }
#pragma unroll
    for (int u = 0; u < SMALL_DK; ++u) {
        tile_store_lds<128>(smem + u * SUB, ra[u]);
        tile_store_lds<SC>(smem + u * SUB + SA, rb[u]);
    }
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const double* cur = smem + (st & 1) * SMALL_STAGE;
        double* nxt = smem + ((st + 1) & 1) * SMALL_STAGE;
        const bool more = st + 1 < nst;
        if (more) {
#pragma unroll
            for (int u = 0; u < SMALL_DK; ++u) {
                ra[u] = tile_load_regs<128>(A, lda, ((st + 1) * SMALL_DK + u) * BK);
                rb[u] = tile_load_regs<SC>(B, ldb, ((st + 1) * SMALL_DK + u) * BK);
            }
        }
#pragma unroll
        for (int u = 0; u < SMALL_DK; ++u) {
            const double* pa = cur + u * SUB + (wave * 32 + (lane & 15)) * LDS_LD + (lane >> 4);
            const double* pb = cur + u * SUB + SA + (lane & 15) * LDS_LD + (lane >> 4);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double a[2], b[MB];
#pragma unroll
                for (int rbl = 0; rbl < 2; ++rbl) a[rbl] = pa[rbl * 16 * LDS_LD + kk * 4];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) b[mb] = -pb[mb * 16 * LDS_LD + kk * 4];
#pragma unroll
                for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc.t[rbl][mb] = mfma_f64(a[rbl], b[mb], acc.t[rbl][mb]);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < SMALL_DK; ++u) {
                tile_store_lds<128>(nxt + u * SUB, ra[u]);
                tile_store_lds<SC>(nxt + u * SUB + SA, rb[u]);
            }
        }
        __syncthreads();
    }
}

template <int KIND, int MB, int DK>
__global__ __launch_bounds__(256) void trsm_step_small_kernel(const double* __restrict__ Xcs,
                                                              const double* __restrict__ Xs, double* __restrict__ V,
                                                              int ldv, const double* __restrict__ L, int ld,
                                                              const double* __restrict__ LinvP, int i, int n,
                                                              double* __restrict__ q, double* __restrict__ mu,
                                                              long long c0, CovParams cp) {
    constexpr int SC = 16 * MB;
    __shared__ double smem[small_smem_doubles<MB, DK>()];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, nn = lane & 15;
    double* Vrow = V + (size_t)blockIdx.x * SC * ldv;
    const long long cw = c0 + (long long)blockIdx.x * SC;
    const int n_valid = n - i * NB;
    AccS<MB> acc;
    gen_cross_tile_s<KIND, MB>(cp, Xcs + (size_t)cw * cp.dim, Xs + (size_t)i * NB * cp.dim, n_valid, smem, acc);
    if (i > 0) gemm_s<MB, DK>(L + (size_t)i * NB * ld, ld, Vrow, ldv, i * NB, acc, smem);
    // ---- T^T -> LDS: sT[mb][row][16]
    double* sT = smem;
    double* sO = smem + MB * NB * 16;
#pragma unroll
    for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sT[(mb * NB + wave * 32 + rbl * 16 + g + 4 * r) * 16 + nn] = acc.t[rbl][mb][r];
    __syncthreads();
    // ---- result row blocks w and 7 - w:  out[cb] = sum_{jb <= cb} Linv[cb][jb] T[jb]
    const double* wp = LinvP + (size_t)i * WP_BLOCK + lane;
    const double* z = L + (size_t)n * ld + (size_t)i * NB;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int cb = half == 0 ? wave : 7 - wave;
        v4d o[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) o[mb] = v4d{0.0, 0.0, 0.0, 0.0};
        const double* wf = wp + (size_t)wp_offset(cb) * 64;
        for (int jb = 0; jb <= cb; ++jb) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double a = wf[(4 * jb + kk) * 64];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    o[mb] = mfma_f64(a, sT[(mb * NB + 16 * jb + 4 * kk + g) * 16 + nn], o[mb]);
            }
        }
        const int r0 = 16 * cb + 4 * g;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = r0 + r < n_valid ? o[mb][r] : 0.0;
                sO[(mb * NB + r0 + r) * 16 + nn] = v[r];
            }
            double2* dst = reinterpret_cast<double2*>(Vrow + (size_t)(mb * 16 + nn) * ldv + (size_t)i * NB + r0);
            dst[0] = make_double2(v[0], v[1]);
            dst[1] = make_double2(v[2], v[3]);
        }
    }
    __syncthreads();
    // ---- |v|^2 and v.z per candidate in the summation order of the 128-candidate step: wave mb, lane (g, nn)
    if (wave < MB) {
        const int mb = wave;
        double sq = 0.0, sz = 0.0;
#pragma unroll
        for (int cbi = 0; cbi < 8; ++cbi) {
            const int r0 = 16 * (7 - cbi) + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = sO[(mb * NB + r0 + r) * 16 + nn];
                const double zc = r0 + r < n_valid ? z[r0 + r] : 0.0;
                sq = fma(v, v, sq);
                sz = fma(v, zc, sz);
            }
        }
        sq += __shfl_xor(sq, 16);
        sz += __shfl_xor(sz, 16);
        sq += __shfl_xor(sq, 32);
        sz += __shfl_xor(sz, 32);
        if (g == 0) {
            const long long c = cw + mb * 16 + nn;
            if (i == 0) {
                q[c] = sq;
                mu[c] = sz;
            } else {
                q[c] += sq;
                mu[c] += sz;
            }
        }
    }
}

// The same block-row step on a cross-gram that already lies in V (cross_gram_kernel: fp32 covariance entries of
// BASELINE config 5, representer points, gradient right-hand sides): the tile is read from V into the transposed
// accumulator layout instead of being generated.
__global__ __launch_bounds__(256, 2) void trsm_step_kernel(double* __restrict__ V, int ldv, const double* __restrict__ L,
                                                           int ld, const double* __restrict__ LinvP, int i, int n,
                                                           double* __restrict__ q, double* __restrict__ mu,
                                                           long long c0) {
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    double* Vrow = V + (size_t)blockIdx.x * NB * ldv;
    const long long cw = c0 + (long long)blockIdx.x * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    AccTt acc;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const double* src = Vrow + (size_t)(wave * 32 + mb * 16 + (lane & 15)) * ldv + (size_t)i * NB + (lane >> 4);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.t[rb][mb][r] = src[rb * 16 + 4 * r];
    }
    if (i > 0) gemm_t(L + (size_t)i * NB * ld, ld, Vrow, ldv, i * NB, acc, smem);
    solve_store_reduce_t(acc, LinvP + (size_t)i * WP_BLOCK, L + (size_t)n * ld + (size_t)i * NB, n - i * NB,
                         Vrow + (size_t)i * NB, ldv, q + cw, mu + cw, i == 0, threadIdx.x >> 6);
}

// mean/var from the reductions, with the reference's output transform and variance floor
// (robo/models/gaussian_process.py:282-294)
__global__ __launch_bounds__(256) void post_kernel(const double* __restrict__ q, const double* __restrict__ mu,
                                                   const double* __restrict__ Xcs, double* __restrict__ mean,
                                                   double* __restrict__ var, long long c0, long long cn, CovParams cp,
                                                   double mean_c, double y_mean, double y_std) {
    const long long i = c0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c0 + cn) return;
    double m = mu[i] + mean_c;
    double v = cov_self(cp, Xcs[i * cp.dim + cp.dim - 1]) - q[i];
    m = m * y_std + y_mean;
    v = v * (y_std * y_std);
    const double eps = 2.220446049250313e-16;
    v = v < eps ? eps : v;   // np.clip(var, eps, inf); NaN propagates like np.clip
    mean[i] = m;
    var[i] = v;
}

// cov[c][c'] = (k(x_c, x_c') - v_c . v_c') * y_std^2   for c, c' < m  (small m)
__global__ __launch_bounds__(256) void cov_kernel(const double* __restrict__ V, int ldv, int kend,
                                                  const double* __restrict__ Xcs, CovParams cp, double y_std,
                                                  long long m, double* __restrict__ cov) {
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    const long long r0 = (long long)blockIdx.y * NB, q0 = (long long)blockIdx.x * NB;
    Acc acc;
    acc_zero(acc);
    gemm_nt_128<false>(V + (size_t)r0 * ldv, ldv, V + (size_t)q0 * ldv, ldv, 0, kend, acc, smem);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long a = r0 + acc_row(tm, r), b = q0 + acc_col(tn);
                if (a < m && b < m)
                    cov[a * m + b] = (cov_rows(cp, Xcs + a * cp.dim, Xcs + b * cp.dim) - acc.t[tm][tn][r]) *
                                     (y_std * y_std);
            }
}

int launch_trsm(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn) {
    // block rows that hold training points; a trailing block holding only the augmented row and
    // identity padding (n % 128 == 0) solves to zeros and is skipped (its V columns stay the
    // zeros cross_gram_kernel wrote)
    const int nbk = (gp->n + NB - 1) / NB;
    cand->solve_kernel = "trsm_step_kernel";
    for (int i = 0; i < nbk; ++i) {
        hipLaunchKernelGGL(trsm_step_kernel, dim3((unsigned)(cn / NB)), dim3(256), 0, gp->ctx->stream, cand->d_V,
                           gp->n_pad, (const double*)gp->d_K, gp->n_pad, (const double*)gp->d_LinvP, i, gp->n,
                           cand->d_q, cand->d_mu, (long long)c0);
    }
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// block-row steps with the cross-gram tile generated in registers (fp64 covariance entries)
int launch_predict_fused(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn) {
    const int nbk = (gp->n + NB - 1) / NB;
    const dim3 grid((unsigned)(cn / NB));
    const Tuning& tune = gp->ctx->tune;
    {   // batches that leave most CUs without a 128-candidate workgroup: 32 candidates per workgroup
        if (cn <= tune.trsm_small_max) {
            // 16 candidates per workgroup while that is at most one workgroup per CU, 32 beyond; two k-tiles per
            // staging stage (83-92 KB of LDS) while one workgroup per CU covers the batch (the knobs force either)
            const int64_t ncu = gp->ctx->num_cu;
            const bool narrow = tune.trsm_small_narrow >= 0 ? tune.trsm_small_narrow != 0 : cn / 16 <= ncu;
            const bool deep = tune.trsm_small_deep >= 0 ? tune.trsm_small_deep != 0 : cn / (narrow ? 16 : 32) <= ncu;
            const dim3 sgrid((unsigned)(cn / (narrow ? 16 : 32)));
            cand->solve_kernel = "trsm_step_small_kernel";
#define ROBO_SMALL_LAUNCH(KIND, MB, DK)                                                                        \
    hipLaunchKernelGGL((trsm_step_small_kernel<KIND, MB, DK>), sgrid, dim3(256), 0, gp->ctx->stream,           \
                       (const double*)cand->d_Xcs, (const double*)gp->d_Xs, cand->d_V, gp->n_pad,             \
                       (const double*)gp->d_K, gp->n_pad, (const double*)gp->d_LinvP, i, gp->n, cand->d_q,     \
                       cand->d_mu, (long long)c0, gp->cov)
#define ROBO_SMALL_CALL(KIND)                                                                                  \
    do {                                                                                                       \
        if (narrow && deep) ROBO_SMALL_LAUNCH(KIND, 1, 2);                                                     \
        else if (narrow) ROBO_SMALL_LAUNCH(KIND, 1, 1);                                                        \
        else if (deep) ROBO_SMALL_LAUNCH(KIND, 2, 2);                                                          \
        else ROBO_SMALL_LAUNCH(KIND, 2, 1);                                                                    \
    } while (0)
            for (int i = 0; i < nbk; ++i) {
                if (gp->kind == ROBO_KERNEL_MATERN52_ARD) ROBO_SMALL_CALL(ROBO_KERNEL_MATERN52_ARD);
                else if (gp->kind == ROBO_KERNEL_RBF_ARD) ROBO_SMALL_CALL(ROBO_KERNEL_RBF_ARD);
                else ROBO_SMALL_CALL(ROBO_KERNEL_FABOLAS);
            }
#undef ROBO_SMALL_CALL
#undef ROBO_SMALL_LAUNCH
            ROBO_LAUNCH_CHECK();
            return ROBO_OK;
        }
    }
#define ROBO_STEP_CALL(KIND)                                                                                   \
    hipLaunchKernelGGL(trsm_step_gen_kernel<KIND>, grid, dim3(256), 0, gp->ctx->stream,                        \
                       (const double*)cand->d_Xcs, (const double*)gp->d_Xs, cand->d_V, gp->n_pad,             \
                       (const double*)gp->d_K, gp->n_pad, (const double*)gp->d_LinvP, i,                      \
                       (i + rows < nbk ? i + rows : nbk), gp->n, cand->d_q, cand->d_mu, (long long)c0, gp->cov)
    cand->solve_kernel = "trsm_step_gen_kernel";
    const int rows = tune.trsm_rows < 1 ? 1 : tune.trsm_rows;
    int i_first = 0;
    if (tune.trsm_pair != 0 && nbk >= 2) {
        // pairs of block rows on one read of V; an odd last block row takes the one-row step
        cand->solve_kernel = "trsm_pair_gen_kernel";
#define ROBO_PAIR_CALL(KIND)                                                                                   \
    hipLaunchKernelGGL(trsm_pair_gen_kernel<KIND>, grid, dim3(512), 0, gp->ctx->stream,                        \
                       (const double*)cand->d_Xcs, (const double*)gp->d_Xs, cand->d_V, gp->n_pad,             \
                       (const double*)gp->d_K, gp->n_pad, (const double*)gp->d_LinvP, i, gp->n, cand->d_q,    \
                       cand->d_mu, (long long)c0, gp->cov)
        for (int i = 0; i + 1 < nbk; i += 2) {
            if (gp->kind == ROBO_KERNEL_MATERN52_ARD) ROBO_PAIR_CALL(ROBO_KERNEL_MATERN52_ARD);
            else if (gp->kind == ROBO_KERNEL_RBF_ARD) ROBO_PAIR_CALL(ROBO_KERNEL_RBF_ARD);
            else ROBO_PAIR_CALL(ROBO_KERNEL_FABOLAS);
        }
#undef ROBO_PAIR_CALL
        i_first = nbk & ~1;
    }
    for (int i = i_first; i < nbk; i += rows) {
        if (gp->kind == ROBO_KERNEL_MATERN52_ARD) ROBO_STEP_CALL(ROBO_KERNEL_MATERN52_ARD);
        else if (gp->kind == ROBO_KERNEL_RBF_ARD) ROBO_STEP_CALL(ROBO_KERNEL_RBF_ARD);
        else ROBO_STEP_CALL(ROBO_KERNEL_FABOLAS);
    }
#undef ROBO_STEP_CALL
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_post(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn) {
    hipLaunchKernelGGL(post_kernel, dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, gp->ctx->stream,
                       (const double*)cand->d_q, (const double*)cand->d_mu, (const double*)cand->d_Xcs, cand->d_mean,
                       cand->d_var, (long long)c0, (long long)cn, gp->cov, gp->mean_c, gp->y_mean, gp->y_std);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_cov(robo_gp* gp, robo_cand* cand, double* d_cov) {
    const unsigned t = (unsigned)(cand->m_pad / NB);
    // only the block rows that were solved hold data (the fused kernel never touches the rest)
    hipLaunchKernelGGL(cov_kernel, dim3(t, t), dim3(256), 0, gp->ctx->stream, (const double*)cand->d_V, gp->n_pad,
                       (gp->n + NB - 1) / NB * NB, (const double*)cand->d_Xcs, gp->cov, gp->y_std, (long long)cand->m, d_cov);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
