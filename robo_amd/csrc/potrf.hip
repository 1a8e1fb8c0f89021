// K2 blocked Cholesky (right-looking, panel width NB = 128) and K3 log-likelihood.
//
// Replaces the factorisation inside george.GP.compute and GP.log_likelihood
// (reference call sites robo/models/gaussian_process.py:119,155,159;
//  robo/models/gaussian_process_mcmc.py:195,200-202).
//
// Per panel k (block column k of the n_pad x n_pad lower matrix, in place):
//   potrf_diag_kernel   1 workgroup : L_kk = chol(A_kk), W_k = L_kk^-1            (LDS resident)
//   potrf_panel_kernel  2(nb-k-1) WGs: A_ik <- A_ik * L_kk^-T by 16-column block substitution (fp64 MFMA, registers)
//   potrf_step_kernel   tri tiles   : A_ij <- A_ij - A_ik * A_jk^T  (fp64 MFMA, gemm_f64.h); the
//                                     workgroup of tile (k+1, k+1) then factors it (next diagonal block)
// Because row n of the matrix is the augmented right-hand side (gram.hip), the finished
// factor holds z = L^-1 (y - mean) in row n: loglik_kernel only reduces z.z and log diag.
//
// Flops: n_pad^3/3 (+ n_pad*NB^2 for the explicit block inverses); the trailing update
// is the MFMA-bound part, the 128-wide diagonal kernel the latency-bound part.
#include "common.h"
#include "gemm_f64.h"
#include "gram_tile.h"
#include "mcmc_dev.h"

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

// Element (r, c) of a 16x16 LDS block.  Rows are 16 doubles apart, so a plain row-major block puts
// the 16 rows of an MFMA A-fragment read (lane l -> row l & 15, column 4 kk + (l >> 4)) on only two
// 8-byte bank groups: an 8-way conflict on every fragment read (r01s: 1.4k cycles per 16x16x16
// product).  XOR-ing the column with (r & 14) spreads the 64 lanes of an A-fragment read, a
// B-fragment read and a C-layout access evenly over the 32 bank groups (2 lanes each = the minimum
// for a 512-byte wave access), at no cost in space -- the two images of the diagonal block already
// take 147 of the 160 KB.
__device__ __forceinline__ int bidx(int r, int c) { return r * SB + (c ^ (r & 14)); }

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
    for (int r = 0; r < 4; ++r) c[r] = b[bidx((lane >> 4) + 4 * r, lane & 15)];
    return c;
}
__device__ __forceinline__ void blk_store_c(double* b, int lane, v4d c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) b[bidx((lane >> 4) + 4 * r, lane & 15)] = c[r];
}
// acc += sgn * A(16x16) * B^T(16x16)   ("NT": both blocks indexed [row][k])
template <bool NEG>
__device__ __forceinline__ v4d blk_mma_nt(const double* A, const double* B, int lane, v4d acc) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        double a = A[bidx(lane & 15, kk * 4 + (lane >> 4))];
        const double b = B[bidx(lane & 15, kk * 4 + (lane >> 4))];
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
        double a = A[bidx(lane & 15, kk * 4 + (lane >> 4))];
        const double b = B[bidx(kk * 4 + (lane >> 4), lane & 15)];
        if (NEG) a = -a;
        acc = mfma_f64(a, b, acc);
    }
    return acc;
}

// fragments of a 16x16 LDS block for four consecutive MFMAs (k = 0..15): "row" form = element
// [lane & 15][4 kk + (lane >> 4)] (the A operand, and the B operand of an NT product), "col" form = element
// [4 kk + (lane >> 4)][lane & 15] (the B operand of an NN product)
struct Frag4 {
    double v[4];
};
__device__ __forceinline__ Frag4 frag_row(const double* A, int lane) {
    Frag4 f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f.v[kk] = A[bidx(lane & 15, kk * 4 + (lane >> 4))];
    return f;
}
__device__ __forceinline__ Frag4 frag_col(const double* B, int lane) {
    Frag4 f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f.v[kk] = B[bidx(kk * 4 + (lane >> 4), lane & 15)];
    return f;
}
template <bool NEG>
__device__ __forceinline__ v4d frag_mma(const Frag4& a, const Frag4& b, v4d acc) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = mfma_f64(NEG ? -a.v[kk] : a.v[kk], b.v[kk], acc);
    return acc;
}

// broadcast lane `src`'s double to the whole wave through SGPRs (v_readlane_b32 x2; `src` is a
// compile-time constant after unrolling)
__device__ __forceinline__ double bcast_lane(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// ---- 16x16 building blocks ------------------------------------------------------------
// Everything in the diagonal kernel is latency-bound on one pivot chain (128 sequential
// rsqrt -> scale -> update steps), so the blocks below are written for few instructions on that
// chain and for everything else to run on the other three waves meanwhile (measured with
// robo_selftest_diag_timeline: the first version spent 11.5k cycles per 16x16 potf2+inverse).

// One wave: unblocked Cholesky of the 16x16 block Ld (lower part valid), in place.
// Lane i (mod 16) owns row i in registers.  The dependent chain per column is kept to
//   pivot (readlane) -> rsqrt -> scale -> one FMA on the NEXT column (its multiplier comes through a
//   second readlane, not through LDS),
// so the next pivot never waits for the LDS exchange: the scaled column goes through a 16-double
// LDS buffer that every lane reads back uniformly (broadcast ds_read_b128) for the bulk rank-1
// update of columns >= k+2, which is issued one iteration later, in the shadow of the following
// column's rsqrt chain (r01s: 7.8k -> 6.4k cycles per 16x16; the lone wave is issue-bound).  1/L_kk is kept in rd[] for
// the forward substitutions.  g0 = global index of the block's first row; rows >= n_real have their
// pivot forced to 1 (augmented row and identity padding).  Returns the first failing global
// column + 1, or 0.
// 1 / sqrt(p) for a finite p > 0: v_rsq_f64 and one third-order correction r (1 + e/2 + 3 e^2/8), e = 1 - p r^2
// -- the sequence the compiler's rsqrt() expands to, minus its special-case selects (p = 0 / inf / NaN cannot
// reach this point: the pivot test above it replaces them, and an inf pivot only has to end in a flagged failure,
// which inf * 0 = NaN at the next pivot guarantees).
__device__ __forceinline__ double pivot_rsqrt(double p) {
    double r = __builtin_amdgcn_rsq(p);
    const double e = fma(-p * r, r, 1.0);
    return fma(r * e, fma(e, 0.375, 0.5), r);
}

template <bool GUARD>
__device__ __forceinline__ int potf2_16_impl(double* Ld, double* Wd, double* colbuf, int lane, int g0, int n_real) {
    const int row = lane & 15, grp = lane >> 4;
    // lanes 16..31 run the SAME instruction stream on different data: instead of row `row` of the block
    // they hold column `row` of W = L^-1, started as a unit vector -- the forward substitution
    // L w = e_c is exactly "scale entry k by 1/L_kk, subtract column k of L times it from the entries
    // below", i.e. the scale / fast-path / bulk-update instructions below with this lane's own
    // multiplier.  The inverse of the diagonal sub-block costs no instruction of its own.
    const bool isW = grp == 1;
    double a[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const double l = j <= row ? Ld[bidx(row, j)] : 0.0;
        a[j] = isW ? (j == row ? 1.0 : 0.0) : l;
    }
    int fail = 0;
    double lprev = 0.0;   // this lane's entry of the previous column (multiplier of the deferred bulk update)
    double* cbw = colbuf + grp * 2 * SB;   // every 16-lane group stores to its own copy; group 0's is read
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        double p = bcast_lane(a[k], k);   // pivot: complete (bulk updates <= k-2, fast path k-1)
        if (GUARD) {
            if (g0 + k >= n_real) p = 1.0;
            if (!(p > 0.0)) {             // also catches NaN
                if (fail == 0) fail = g0 + k + 1;
                p = 1.0;
            }
        }
        // Without the guard (every block that holds only training rows) the pivot is NOT tested on the chain: a
        // non-positive or NaN pivot makes v_rsq_f64 return NaN / inf, L_kk = p * rsqrt(p) comes out NaN, and so does
        // everything after it -- the failing column is read off the diagonal once the block is done (below).  Four
        // scalar/vector instructions and the select in front of the rsq less per pivot on an issue-bound wave.
        const double ri = pivot_rsqrt(p);
        if (k > 0) {
            // deferred bulk update by column k-1 (independent of the rsqrt chain above)
            const double* cbp = colbuf + ((k - 1) & 1) * SB;
#pragma unroll
            for (int j = k + 1; j < SB; ++j) a[j] = fma(-lprev, cbp[j], a[j]);
        }
        // rows < k (columns > k of W) hold garbage (zeros) here, never read
        // lane k's a[k] IS the pivot unless the guard / failure path replaced p, so without the guard the
        // scale needs no select (a failed factorisation is flagged; its numbers are garbage either way)
        const double lik = (GUARD && row == k && !isW) ? p * ri : a[k] * ri;
        a[k] = lik;
        cbw[(k & 1) * SB + row] = lik;    // no exec masking or branches on the chain
        if (k + 1 < SB) {
            const double l1 = bcast_lane(lik, k + 1);        // L[k+1][k]
            a[k + 1] = fma(-lik, l1, a[k + 1]);              // fast path: column k+1 is complete
        }
        lprev = lik;
        wave_lds_fence();
    }
    if (!GUARD) {
        // first column whose diagonal entry is not a positive finite number (lanes 0..15 hold L_rr in a[row])
        double diag = 0.0;
#pragma unroll
        for (int j = 0; j < SB; ++j) diag = row == j ? a[j] : diag;
        const bool bad = grp == 0 && !(diag > 0.0 && diag < 1.0e300);
        const unsigned long long m = __ballot(bad);
        if (m != 0ull) fail = g0 + (__ffsll((long long)m) - 1) + 1;
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        if (isW) Wd[bidx(j, row)] = a[j];                     // W[j][c], zero above the diagonal
        else Ld[bidx(row, j)] = j <= row ? a[j] : 0.0;
    }
    return fail;
}

// ---- all four 16-lane groups at work (r02w) -----------------------------------------------------------------
// The version above keeps a whole row (16 entries) per lane and uses two of the four lane groups (rows of L,
// columns of W); its lone wave is bound by instruction ISSUE (~34 instructions per pivot at ~10 cycles), and a third
// of them are the rank-1 update of up to 15 entries per lane.  Here a lane holds the entries of ONE COLUMN PARITY:
//     group 0: rows of L, even columns     group 2: rows of L, odd columns
//     group 1: columns of W, even rows     group 3: columns of W, odd rows
// i.e. 8 entries a[h] <-> second index j = 2 h + par, so the rank-1 update is at most 8 FMAs per pivot, and the
// scaled column goes through LDS de-interleaved ([even rows | odd rows]) so that a lane's operands are contiguous.
// The pivot chain does not pass through any lane's registers: the next diagonal entry with columns <= k-1 applied is
// broadcast as a uniform value d1 and  p_{k+1} = d1 - l_{k+1,k}^2  is one FMA on the broadcast l_{k+1,k} -- every
// entry of column k+1 (the diagonal one included) receives column k's contribution with the regular deferred update
// one iteration later, whose multiplier l_{row,k} a lane of the other parity reads back from the exchange buffer.
// colbuf: [4 groups][2 buffers][16], 1 KB.
template <bool GUARD>
__device__ __forceinline__ int potf2_16_split(double* Ld, double* Wd, double* colbuf, int lane, int g0, int n_real) {
    const int row = lane & 15, grp = lane >> 4, par = grp >> 1;
    const bool isW = (grp & 1) != 0;
    constexpr int H = SB / 2;
    double a[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int j = 2 * h + par;
        const double l = j <= row ? Ld[bidx(row, j)] : 0.0;
        a[h] = isW ? (j == row ? 1.0 : 0.0) : l;
    }
    const int pos = (row & 1) * H + (row >> 1);               // de-interleaved position of this lane's row
    double* cb_own = colbuf + grp * 2 * SB + pos;             // where this lane publishes its scaled entry
    const double* cb_mult = colbuf + (grp & 1) * 2 * SB + pos;   // + owner parity * 4 SB: this row's multiplier
    const double* cb_col = colbuf + par * H;                  // + owner parity * 4 SB: column values for own j's
    int fail = 0;
    double p = bcast_lane(a[0], 0);                           // L_00's pivot: group 0, lane 0
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        const int pk = k & 1, hk = k >> 1;
        // (A) the exchange-buffer reads of the deferred update by column k-1 are ISSUED first and consumed last: the
        // column was published at the end of the previous iteration, so they are a full LDS write -> read round trip
        // away, and the compiler's own order put the pivot's fma + rsq chain behind the first s_waitcnt on them
        // (r03 ISA: ~290 cycles per pivot = LDS round trip + rsq chain + hand-over, one after the other).  With the
        // scheduling barriers the rsq chain of pivot k runs while the reads are in flight.
        double lprev = 0.0, cv[H];
        if (k > 0) {
            const int pq = (k - 1) & 1, buf = (k - 1) & 1;
            lprev = cb_mult[pq * 4 * SB + buf * SB];
            const double* c = cb_col + pq * 4 * SB + buf * SB;
#pragma unroll
            for (int h = hk; h < H; ++h) cv[h] = c[h];
        }
        __builtin_amdgcn_sched_barrier(0);
        // (B) the pivot
        if (GUARD) {
            if (g0 + k >= n_real) p = 1.0;
            if (!(p > 0.0)) {             // also catches NaN
                if (fail == 0) fail = g0 + k + 1;
                p = 1.0;
            }
        }
        const double ri = pivot_rsqrt(p);
        __builtin_amdgcn_sched_barrier(0);
        // (C) deferred update by column k-1 (owner parity pq) of every own column j >= k
        if (k > 0) {
#pragma unroll
            for (int h = hk; h < H; ++h) {
                // h = hk is column k for the lanes of parity pk; for the other parity it is column k+1 (k even) or the
                // finished column k-1 (k odd), which must not be touched
                const double m = (pk == 1 && h == hk) ? (par == 1 ? lprev : 0.0) : lprev;
                a[h] = fma(-m, cv[h], a[h]);
            }
        }
        // the next diagonal entry (columns <= k-1 applied), uniform: lane (row k+1, L group of parity (k+1) & 1)
        double d1 = 0.0;
        if (k + 1 < SB) d1 = bcast_lane(a[(k + 1) >> 1], (k + 1) + 32 * ((k + 1) & 1));
        double lik = a[hk] * ri;                              // meaningful in the lanes of parity pk
        if (GUARD && row == k && !isW) lik = p * ri;
        a[hk] = par == pk ? lik : a[hk];
        cb_own[(k & 1) * SB] = lik;                           // the other parity's copies are never read
        if (k + 1 < SB) {
            const double l1 = bcast_lane(lik, (k + 1) + 32 * pk);   // L[k+1][k]
            p = fma(-l1, l1, d1);
        }
        wave_lds_fence();
    }
    if (!GUARD) {
        // first column whose diagonal entry is not a positive finite number
        double diag = 0.0;
#pragma unroll
        for (int h = 0; h < H; ++h) diag = (row >> 1) == h ? a[h] : diag;
        const bool bad = !isW && par == (row & 1) && !(diag > 0.0 && diag < 1.0e300);
        const unsigned long long m = __ballot(bad);
        if (m != 0ull) fail = g0 + ((__ffsll((long long)m) - 1) & 15) + 1;
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int j = 2 * h + par;
        if (isW) Wd[bidx(j, row)] = a[h];                     // W[j][c], zero above the diagonal
        else Ld[bidx(row, j)] = j <= row ? a[h] : 0.0;
    }
    return fail;
}

// ---- rank-1 updates on the matrix pipe, square-root-free (r03t): MEASURED SLOWER, kept as an A/B build ---------------
// Idea: keep the block in the MFMA ACCUMULATOR layout -- register r of lane l = T[(l >> 4) + 4 r][l & 15] -- in which
// row k of T is register k >> 2 of the sixteen lanes of group g = k & 3: exactly the lanes that hold k-slice g of BOTH
// operands of v_mfma_f64_16x16x4_f64 (A[i][g] in lane 16 g + i, B[g][j] in lane 16 g + j).  With the other slices
// zeroed, the rank-1 update of the whole block by pivot row k is ONE instruction whose operands are already in place:
// no exchange of the scaled column through LDS.  Elimination runs on the UPPER triangle, row by row (T = transpose of
// the stored lower block), in LDL^T form so that no square root sits between two pivots:
//     rinv_k = 1 / p_k;    T[i][j] -= T[k][i] (T[k][j] rinv_k)  for i, j > k;    p_{k+1} = T[k+1][k+1] - T[k][k+1]^2 rinv_k
// (the last line on the VALU from two broadcasts taken BEFORE the update is issued).  The inverse falls out of a second
// accumulator: X starts as the identity and receives the same eliminations, X -= w_k (row k of X), w_k = the B operand
// of the first update.  After the last pivot  L[j][i] = T[i][j] / sqrt(p_i),  W[i][j] = X[i][j] / sqrt(p_i).
// Correct (emulator suite + MI355X parity suite), and 5.8-6.0k cycles per 16 pivots against 3.8-4.3k for the
// four-group version (profiles/r03t_diag_timeline_mfma.txt): a VALU instruction (or readlane) that consumes the result
// of an fp64 MFMA waits ~250 cycles for it (accumulator-to-accumulator chaining is 64), and the operands of update k+1
// ARE the result of update k -- the matrix pipe cannot sit inside a per-pivot recurrence.
__device__ __forceinline__ double pivot_rcp(double p) {
    const double r = __builtin_amdgcn_rcp(p);
    const double e = fma(-p, r, 1.0);                   // r (1 + e + e^2): third order, like pivot_rsqrt
    return fma(r * e, 1.0 + e, r);
}

template <bool GUARD>
__device__ __forceinline__ int potf2_16_mfma(double* Ld, double* Wd, int lane, int g0, int n_real) {
    const int col = lane & 15, grp = lane >> 4;
    v4d t, x;
    double pj[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = grp + 4 * r;
        t[r] = col >= i ? Ld[bidx(col, i)] : 0.0;       // T[i][col] = stored lower entry [col][i]
        x[r] = col == i ? 1.0 : 0.0;
        pj[r] = 1.0;
    }
    int fail = 0;
    double p = bcast_lane(t[0], 0);
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        const int g = k & 3, r = k >> 2;
        if (GUARD) {
            if (g0 + k >= n_real) p = 1.0;              // augmented row / identity padding
            if (!(p > 0.0)) {                           // also catches NaN
                if (fail == 0) fail = g0 + k + 1;
                p = 1.0;
            }
        }
        const double nrinv = -pivot_rcp(p);
        pj[r] = grp == g ? p : pj[r];
        const double a = (grp == g && col > k) ? t[r] : 0.0;        // row k right of the diagonal, in slice g
        const double b = a * nrinv;                                  // -w_k
        const double bx = grp == g ? x[r] : 0.0;                     // row k of X
        if (k + 1 < SB) {
            // p_{k+1} from the entries as they are BEFORE this update (row k is final, T[k+1][k+1] has pivots < k)
            const double u = bcast_lane(t[r], 16 * g + k + 1);
            const double d = bcast_lane(t[(k + 1) >> 2], 16 * ((k + 1) & 3) + k + 1);
            p = fma(u * u, nrinv, d);
        }
        t = mfma_f64(a, b, t);
        x = mfma_f64(b, bx, x);
    }
    if (!GUARD) {
        // first non-positive (or non-finite) pivot: every lane of group g holds p_{g + 4 r} in pj[r]
        int first = SB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned long long m = __ballot(!(pj[r] > 0.0 && pj[r] < 1.0e300));
#pragma unroll
            for (int g = 3; g >= 0; --g)
                if (((m >> (16 * g)) & 1ull) != 0ull && 4 * r + g < first) first = 4 * r + g;
        }
        if (first < SB) fail = g0 + first + 1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = grp + 4 * r;
        const double ri = pivot_rsqrt(pj[r]);
        const double l = col == i ? pj[r] * ri : t[r] * ri;
        Ld[bidx(col, i)] = col >= i ? l : 0.0;          // L[col][i]; the upper triangle of the stored block is zero
        Wd[bidx(i, col)] = x[r] * ri;                   // W[i][col], zero above the diagonal
    }
    return fail;
}

#ifndef ROBO_POTF2
#define ROBO_POTF2 1            // 1: four lane groups (r02w); 2: rank-1 MFMA version (r03t: slower); 0: two lane groups
#endif
// the pivot guard for rows >= n_real only exists in the block(s) that hold the augmented row / padding
__device__ __forceinline__ int potf2_16(double* Ld, double* Wd, double* colbuf, int lane, int g0, int n_real) {
#if ROBO_POTF2 == 2
    if (g0 + SB <= n_real) return potf2_16_mfma<false>(Ld, Wd, lane, g0, n_real);
    return potf2_16_mfma<true>(Ld, Wd, lane, g0, n_real);
#elif ROBO_POTF2 == 1
    if (g0 + SB <= n_real) return potf2_16_split<false>(Ld, Wd, colbuf, lane, g0, n_real);
    return potf2_16_split<true>(Ld, Wd, colbuf, lane, g0, n_real);
#else
    if (g0 + SB <= n_real) return potf2_16_impl<false>(Ld, Wd, colbuf, lane, g0, n_real);
    return potf2_16_impl<true>(Ld, Wd, colbuf, lane, g0, n_real);
#endif
}

constexpr int TLD = SB + 2;   // padded leading dimension of the per-wave transposition scratch

// ---- the 128x128 diagonal block, block-packed in LDS -------------------------------------
// sL: 36 lower 16x16 blocks of A -> L in place; sW: 36 blocks of W = L^-1; sT: per-wave 16 x TLD
// scratch; sRd: task counters of the helper waves (8 ints); sCol: 4 x 2 x 16 column exchange buffers (wave 0).
//
// LEFT-LOOKING schedule (r02): wave 0 is the pivot wave and does nothing but the chain
//      potf2(s) -> L_{s+1,s} = A~_{s+1,s} W_ss^T -> A~_{s+1,s+1} -= L_{s+1,s} L_{s+1,s}^T -> potf2(s+1)
// i.e. two 16x16x16 MFMA products between consecutive 16-pivot chains.  Every other product runs on
// waves 1-3 in the shadow of a potf2:  a block A_ij is touched exactly twice -- once to receive ALL its
// updates  A~_ij = A_ij - sum_{c<j} L_ic L_jc^T  (accumulated in registers, one LDS round trip), once for
// its solve with W_jj.  The right-looking form of r01 re-read and re-wrote every trailing 16x16 block at
// every step (27, 20, 14 ... blocks on three waves: the early steps took 11-13k cycles against the pivot
// wave's 6.4k) and put a four-wave sub-panel phase plus a barrier on the chain.
// Interval s (two barriers, Ba at its start right after potf2(s), Bb in the middle):
//   wave 0     : C1  L_{s+1,s};  C2  pivot block (s+1,s+1) finished;  [Bb]  potf2(s+1)
//   waves 1-3  : solves L_{i,s}, i >= s+2   [Bb]   block column s+1 and pivot block (s+2,s+2) receive all
//                their updates (columns 0..s); the inverse advances by block row s, column by column (see the
//                task list in the loop).  One product per block of row 7 of W is left after the last pivot.
// ---- publication for the panel followers (potrf_step_follow_kernel) ---------------------------------------------------------
// With pub != nullptr the diagonal workgroup hands block column s of L_kk and W_ss to the OTHER workgroups of its launch as
// soon as they are final (after barrier Bb(s)), straight into their final places in K and in the inverse block -- so the
// write-back at the end goes away -- with write-through stores by the three helper waves, issued at the START of their
// half-interval (the pivot wave stores nothing: its chain is untouched).  The progress word COUNTS publications: every
// helper wave adds 1 once ITS stores of column s have left the CU -- no barrier between the three -- so column c is in memory
// when the word reads >= 3 (c + 1).  When: in the first intervals the helpers are the longer side of the interval (their
// update tasks, r05z_diag_timeline), so they drain and count AFTER their tasks, when the stores have long completed; from
// interval `early` on (potrf_pub_early, default 5: one or no task per wave) they have time to spare and count at once -- the followers then work on column s while the pivot
// wave runs potf2(s+1), and only the last column (16 x 16: W_77) is left when the diagonal block ends.  After the last pivot
// all four waves publish what the loop did not (column nsb - 1 and the identity padding) and add 1 each:
// the word ends at 3 (nsb - 1) + 4 = diag_prog_done(nsb).  Same arithmetic, same bits.
__host__ __device__ constexpr unsigned diag_prog_need(int c, int nsb) {      // value of the progress word from which column c is readable
    return c < nsb - 1 ? 3u * (unsigned)(c + 1) : 3u * (unsigned)(nsb - 1) + 4u;
}
__host__ __device__ constexpr int diag_nsb(int n_real, int kbase) {          // 16-row blocks of a diagonal block that are factored
    const int v = (n_real + 1 - kbase + SB - 1) / SB;
    return v < 1 ? 1 : (v > NSB ? NSB : v);
}
struct DiagPub {
    double* Kd;        // tile (k, k) in K (row-major, leading dimension ld)
    int ld;
    double* Wg;        // the 128 x 128 inverse block of panel k (its eight diagonal sub-blocks are written)
    unsigned* prog;    // progress word of panel k
    int early;         // first interval whose helper waves count right after publishing (see below)
};
// one wave: 16 x 16 LDS block (bidx layout) -> 16 rows of a row-major global matrix, write-through
__device__ __forceinline__ void blk_publish(const double* b, double* dst, int ld, int lane) {
    const int r = lane >> 2, c0 = (lane & 3) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) st_agent(dst + (size_t)r * ld + c0 + q, b[bidx(r, c0 + q)]);
}
// block column c of L (rows c .. 7) and W_cc, blocks dealt round-robin to `nw` waves (this wave: `w`)
// (Wc: the LDS block that holds W_cc, or nullptr: the identity -- a padding block of the rolling layout, which keeps no image of W)
__device__ __forceinline__ void diag_publish_column(const double* sL, const double* Wc, const DiagPub& pub, int c, int w,
                                                    int nw, int lane) {
    int t = 0;
    for (int bi = c; bi < NSB; ++bi, ++t)
        if (t % nw == w) blk_publish(sL + blk_off(bi, c), pub.Kd + (size_t)(bi * SB) * pub.ld + c * SB, pub.ld, lane);
    if (t % nw == w) {
        double* dst = pub.Wg + (size_t)(c * SB) * NB + c * SB;
        if (Wc) {
            blk_publish(Wc, dst, NB, lane);
        } else {
            const int r = lane >> 2, c0 = (lane & 3) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) st_agent(dst + (size_t)r * NB + c0 + q, r == c0 + q ? 1.0 : 0.0);
        }
    }
}

// ROLL (publishing callers only): sW is TWO 16 x 16 slots instead of a 36-block image -- W_ss lives in slot s & 1 from
// potf2(s) until it has been published (interval s), potf2(s + 2) may overwrite it a barrier later; the whole LDS image of a
// diagonal workgroup is then 80 KB (L image + 2 slots + exchange buffers): TWO workgroups per CU (r05g's layout, which had
// nowhere to put the W_ss; the publication gives them a place at once).
template <bool ROLL = false>
__device__ __forceinline__ void diag128_factor_invert(double* sL, double* sW, double* sT, double* sRd, double* sCol,
                                                      int kbase, int n_real, int* fail, long long* dbg,
                                                      const DiagPub* pub = nullptr) {
    auto wslot = [sW](int s_) { return ROLL ? sW + (s_ & 1) * BLK : sW + blk_off(s_, s_); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* ctr = reinterpret_cast<int*>(sRd);     // one task counter per interval
    if (tid >= 64 && tid < 64 + NSB) ctr[tid - 64] = 0;
    // 16-row blocks that hold training rows or the augmented row; the ones behind them are identity padding (their
    // factor and inverse are the identity and nothing couples them to the rest), so the chain stops there: at the
    // N < 128 of a Bayesian-optimisation run the single diagonal block is mostly padding (N = 30: 2 of 8 blocks).
    const int nsb = diag_nsb(n_real, kbase);
    if (!ROLL)
        for (int bi = nsb; bi < NSB; ++bi) sW[blk_off(bi, bi) + bidx(tid >> 4, tid & 15)] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
    // this lane's offsets inside a 16x16 block: fragment form [lane & 15][4 kk + (lane >> 4)], accumulator form
    // [(lane >> 4) + 4 r][lane & 15]
    int fo[4], co[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        fo[q] = bidx(lane & 15, q * 4 + (lane >> 4));
        co[q] = bidx((lane >> 4) + 4 * q, lane & 15);
    }
    if (wave == 0) {
        const int f = potf2_16(sL + blk_off(0, 0), wslot(0), sCol, lane, kbase, n_real);
        if (f != 0 && lane == 0 && *fail == 0) *fail = f;
    }
    __syncthreads();                                              // Ba(0)
    if (dbg && tid == 0) dbg[2] = clock64();
    for (int s = 0; s + 1 < nsb; ++s) {
        if (wave == 0) {
            // C1: the TRANSPOSE Q = L_{s+1,s}^T = W_ss A~_{s+1,s}^T.  In the MFMA accumulator layout register r
            // of lane l holds Q[(l >> 4) + 4 r][l & 15], which is at once the A fragment of columns 4r..4r+3 of Q^T
            // and the B fragment of rows 4r..4r+3 of Q:
            // C2: the pivot block's last update  T -= L L^T = Q^T Q  is four MFMAs straight from those registers,
            // with no trip through LDS between the two products of the chain.
            double* P = sL + blk_off(s + 1, s);
            double* C = sL + blk_off(s + 1, s + 1);
            v4d t = blk_load_c(C, lane);
            v4d q = {0.0, 0.0, 0.0, 0.0};
            q = blk_mma_nt<false>(wslot(s), P, lane, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) t = mfma_f64(-q[r], q[r], t);
            blk_store_c(C, lane, t);
            wave_lds_fence();   // (also orders the fragment reads of P before its overwrite)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[bidx(lane & 15, (lane >> 4) + 4 * r)] = q[r];   // L = Q^T for the helpers
            wave_lds_fence();
            if (dbg && tid == 0) dbg[24 + 4 * s] = clock64();     // C1 + C2 done (pivot wave)
        } else {
            // solves of block column s below the pivot wave's own block
            for (int bi = s + 2 + (wave - 1); bi < nsb; bi += 3) {
                double* A = sL + blk_off(bi, s);
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                acc = blk_mma_nt<false>(A, wslot(s), lane, acc);
                wave_lds_fence();
                blk_store_c(A, lane, acc);
            }
        }
        __syncthreads();                                          // Bb(s): block column s of L is final
        if (dbg && tid == 0 && s == 0) dbg[3] = clock64();
        if (dbg && tid == 0) dbg[24 + 4 * s + 1] = clock64();     // through Bb(s)
        if (pub && wave != 0) {
            diag_publish_column(sL, wslot(s), *pub, s, wave - 1, 3, lane);
            if (s >= pub->early) {
                drain_vmem();
                if (lane == 0) add_agent_u32(pub->prog, 1u);
            }
        }
        if (wave == 0) {
            const int f = potf2_16(sL + blk_off(s + 1, s + 1), wslot(s + 1), sCol, lane,
                                   kbase + (s + 1) * SB, n_real);
            if (f != 0 && lane == 0 && *fail == 0) *fail = f;
            if (dbg && tid == 0) dbg[24 + 4 * s + 2] = clock64() + (long long)(f == 12345678);   // potf2(s+1) done
        } else {
            // Work of the interval, handed out dynamically (an LDS counter per interval; a task is wave-sized):
            // one block of column s+1 (or the next pivot block) receives columns 0..s in one pass.
            // t = 0: (s+2, s+1) and t = 1: (s+2, s+2) are what the pivot wave needs first at the next
            // interval; t >= 2: (s+1+t, s+1).
            // The off-diagonal blocks of the inverse W = L^-1 are NOT formed here (r02f): advanced alongside the
            // factorisation they cost 112 more 16x16x16 products on these three waves and made the pivot wave
            // wait (84.7k cycles per diagonal block against 68.2k without them).  The panel solve needs only the
            // eight W_ss that fall out of potf2 (potrf_panel_kernel substitutes block column by block column);
            // the full inverses, which the posterior's TRSM and the likelihood gradient use, are produced for
            // all diagonal blocks at once by potrf_inverse_kernel after the factorisation.
            // Static hand-out (task t to wave 1 + t % 3), lane offsets computed once per kernel.  (Dynamic hand-out
            // through an LDS counter, pairing blocks that share an operand, and software-pipelined fragment loads
            // were all measured within noise of this: intervals 1-3 take 6.0-7.2k cycles against 4.5k for the pivot
            // wave alone, whatever the bookkeeping -- the pivot wave's own LDS exchange slows down while the helpers'
            // fragment reads share the LDS pipe.)
            const int ntask = s + 2 < nsb ? nsb - 1 - s : 0;
            for (int t = wave - 1; t < ntask; t += 3) {
                const int bi = t <= 1 ? s + 2 : s + 1 + t;
                const int bj = t == 1 ? s + 2 : s + 1;
                double* C = sL + blk_off(bi, bj);
                const double* Ai = sL + blk_off(bi, 0);
                const double* Bj = sL + blk_off(bj, 0);
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = C[co[r]];
                for (int c = 0; c <= s; ++c) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = mfma_f64(-Ai[c * BLK + fo[kk]], Bj[c * BLK + fo[kk]], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) C[co[r]] = acc[r];
            }
        }
        if (pub && wave != 0 && s < pub->early) {
            drain_vmem();                                         // this wave's share of column s has left the CU
            if (lane == 0) add_agent_u32(pub->prog, 1u);
        }
        __syncthreads();                                          // Ba(s+1)
        if (dbg && tid == 0) dbg[4 + s] = clock64();
    }
    if (pub) {
        // what the loop did not hand over: the last factored block column (nsb - 1) and the identity padding behind it
        for (int c = nsb - 1; c < NSB; ++c)
            diag_publish_column(sL, (ROLL && c >= nsb) ? nullptr : wslot(c), *pub, c, wave, 4, lane);
        drain_vmem();
        if (lane == 0) add_agent_u32(pub->prog, 1u);
    }
    if (dbg && tid == 0) dbg[11] = clock64();
}

constexpr int DIAG_SMEM_DOUBLES = 2 * NBLK * BLK + 4 * SB * TLD + NB + 8 * SB;   // 150 KB

struct DiagSmem {
    double *sL, *sW, *sT, *sRd, *sCol;
};
__device__ __forceinline__ DiagSmem diag_carve(double* base) {
    DiagSmem m;
    m.sL = base;
    m.sW = m.sL + NBLK * BLK;
    m.sT = m.sW + NBLK * BLK;
    m.sRd = m.sT + 4 * SB * TLD;
    m.sCol = m.sRd + NB;
    return m;
}

// L into K (lower sub-blocks); the eight W_ss = L_ss^-1 into the diagonal sub-blocks of the 128x128 row-major
// inverse block (its off-diagonal sub-blocks are filled in by potrf_inverse_kernel; the strictly upper ones were
// zeroed when the buffer was allocated and are never written)
__device__ __forceinline__ void diag_writeback(const DiagSmem& m, double* __restrict__ Kd, int ld,
                                               double* __restrict__ Wg) {
    const int tid = threadIdx.x;
    for (int bi = 0; bi < NSB; ++bi) {
        for (int bj = 0; bj <= bi; ++bj) {
            const int r = bi * SB + (tid >> 4), c = bj * SB + (tid & 15);
            Kd[(size_t)r * ld + c] = m.sL[blk_off(bi, bj) + bidx(tid >> 4, tid & 15)];
        }
        const int r = bi * SB + (tid >> 4), c = bi * SB + (tid & 15);
        Wg[r * NB + c] = m.sW[blk_off(bi, bi) + bidx(tid >> 4, tid & 15)];
    }
}

__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ K, size_t k_stride, int ld, int k,
                                                         int n_real, double* __restrict__ Linv, size_t linv_stride,
                                                         int* __restrict__ fail, long long* __restrict__ dbg,
                                                         double* __restrict__ ll_out, double* __restrict__ ll_host) {
    __shared__ double smem[DIAG_SMEM_DOUBLES];
    const DiagSmem m = diag_carve(smem);
    const int tid = threadIdx.x;
    const int smp = blockIdx.x;
    K += (size_t)blockIdx.x * k_stride;           // batch coordinate
    Linv += (size_t)blockIdx.x * linv_stride;
    fail += blockIdx.x;
    double* Kd = K + ((size_t)k * NB) * ld + (size_t)k * NB;
    if (dbg && tid == 0) dbg[0] = clock64();

    // ---- load the 36 lower sub-blocks ------------------------------------------------
    for (int bi = 0; bi < NSB; ++bi)
        for (int bj = 0; bj <= bi; ++bj)
            m.sL[blk_off(bi, bj) + bidx(tid >> 4, tid & 15)] =
                Kd[(size_t)(bi * SB + (tid >> 4)) * ld + bj * SB + (tid & 15)];
    __syncthreads();
    if (dbg && tid == 0) dbg[1] = clock64();

    diag128_factor_invert(m.sL, m.sW, m.sT, m.sRd, m.sCol, k * NB, n_real, fail, dbg);

    if (ll_out) {
        // Likelihood evaluation of a one-block problem (N < 128, the usual size of a BO run): the whole factor is in
        // LDS, so (z.z, 2 sum log L_ii, failure flag) leave from here -- no write-back, no tail and finishing launches.
        // Same operations in the same order as potrf_inverse_kernel + loglik_finish_kernel on one block.
        __syncthreads();
        double q = 0.0, lg = 0.0, dmin = __builtin_huge_val(), dmax = 0.0;
        if (tid < NB && tid < n_real) {
            const double zi = m.sL[blk_off(n_real >> 4, tid >> 4) + bidx(n_real & 15, tid & 15)];
            q = zi * zi;
            const double d = m.sL[blk_off(tid >> 4, tid >> 4) + bidx(tid & 15, tid & 15)];
            lg = log(d);
            dmin = dmax = d;
        }
        for (int o = 32; o > 0; o >>= 1) {
            q += __shfl_xor(q, o);
            lg += __shfl_xor(lg, o);
            dmin = fmin(dmin, __shfl_xor(dmin, o));
            dmax = fmax(dmax, __shfl_xor(dmax, o));
        }
        __syncthreads();
        double* red = m.sW;
        if ((tid & 63) == 0 && tid < NB) {
            red[tid >> 6] = q;
            red[2 + (tid >> 6)] = lg;
            red[4 + (tid >> 6)] = dmin;
            red[6 + (tid >> 6)] = dmax;
        }
        __syncthreads();
        if (tid == 0) {
            double sq = 0.0, sl = 0.0;
            sq += red[0] + red[1];
            sl += red[2] + red[3];
            ll_out[2 * smp] = sq;
            ll_out[2 * smp + 1] = 2.0 * sl;
            if (ll_host) {
                ll_host[5 * smp] = sq;
                ll_host[5 * smp + 1] = 2.0 * sl;
                ll_host[5 * smp + 2] = (double)*fail;
                ll_host[5 * smp + 3] = fmin(red[4], red[5]);
                ll_host[5 * smp + 4] = fmax(red[6], red[7]);
            }
        }
        return;
    }
    diag_writeback(m, Kd, ld, Linv + (size_t)k * NB * NB);
    if (dbg && tid == 0) dbg[12] = clock64();
}

// ---- one-block problems (N <= 126, the size of a Bayesian-optimisation run): a whole ensemble half-step per launch ---
// The device-resident chain of mcmc.hip is, per half-step, proposal + scaling | gram | this file's one-block
// factorisation-with-likelihood | accept: four launches of ~5 us each around ~8 us of work at N = 40 (r03z: 34 us per
// half-step).  Here one workgroup per walker does all of it: the proposal and its metrics in LDS (mcmc_dev.h), the gram
// tiles straight into the block-packed LDS image of the diagonal block (gram_tile.h: scaling while staging, the entries
// of scale_inputs_kernel + gram_kernel bit for bit), diag128_factor_invert, potrf_diag_kernel's likelihood reductions,
// the accept test and the walker's own chain record (a walker's entry for step `it` is final after ITS half-step).
// NG = 1: N <= 63, one 64 x 64 tile, 256 threads.  NG = 3: 64 <= N <= 126, 768 threads -- three groups of four waves
// compute the tiles (0,0), (1,0), (1,1) side by side (one after the other on four waves they took as long as the four
// launches, r03zf), then the upper two groups leave and the first one factors (a hardware barrier counts live waves).
template <int KIND, int NG>
__global__ __launch_bounds__(256 * NG) void mcmc_block_step_kernel(McmcState st, int start, int first, int h, int it,
                                                                   const double* __restrict__ X,
                                                                   const double* __restrict__ y) {
    __shared__ double smem[DIAG_SMEM_DOUBLES];
    __shared__ int sfail;
    const DiagSmem m = diag_carve(smem);
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, w = blockIdx.x, P = st.P, n = st.n;
    // the W image is unused until the first 16 x 16 factorisation writes its inverse: proposal and tile staging live there
    double* sq = m.sW;
    double* sism = sq + MAX_DIM + 8;
    double* sz = sism + MAX_DIM;
    int* sflag = reinterpret_cast<int*>(sz + 1);
    double* sI = sz + 2 + grp * (2 * GD * GLD + 2 * GT);     // per group: sI, sJ, sN
    double* sJ = sI + GD * GLD;
    double* sN = sJ + GD * GLD;
    static_assert(MAX_DIM + 8 + MAX_DIM + 2 + 3 * (2 * GD * GLD + 2 * GT) <= NBLK * BLK, "staging fits the W image");
    const bool ok = mcmc_block_proposal(st, start, first, h, it, w, sq, sism, sz, sflag);
    const FitSample sp = mcmc_fit_sample(st, sq, ok);            // uniform, in every thread's registers
    const double z = *sz;
    double prior = 0.0;
    if (threadIdx.x == 0) {
        if (ok && st.prior_kind != 0) prior = prior_lnprob(st.prior_kind, sq, P, st.prior_par);
        if (!ok) prior = -__builtin_huge_val();
        sfail = 0;
    }
    const double q0 = tid < P ? sq[tid] : 0.0, q1 = tid + 256 < P ? sq[tid + 256] : 0.0;   // (group 0) thread p keeps q[p]
    // ---- K into the LDS image: group g owns tile (0,0) / (1,0) / (1,1); rows / columns >= n as gram_kernel writes them
    {
        const int bi = grp == 0 ? 0 : 1, bj = grp == 2 ? 1 : 0;
        const int tx = tid & 15, ty = tid >> 4;
        double cov[4][4];
        pair_cov_dot<KIND>(sp.cov, X, (long long)bi * GT, (long long)bj * GT, sI, sJ, sN, cov, sism, (long long)n, tid);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int gi = bi * GT + ty * 4 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int gj = bj * GT + gram_col(tx, b);
                double val;
                if (gi < n && gj < n) {
                    val = cov[a][b];
                    if (gi == gj) val += sp.noise;
                } else if (gi == gj) {
                    val = 1.0;
                } else if (gi == n && gj < n) {
                    val = y[gj] - sp.mean_c;
                } else if (gj == n && gi < n) {
                    val = y[gi] - sp.mean_c;
                } else {
                    val = 0.0;
                }
                if ((gj >> 4) <= (gi >> 4)) m.sL[blk_off(gi >> 4, gj >> 4) + bidx(gi & 15, gj & 15)] = val;
            }
        }
    }
    __syncthreads();
    if (grp != 0) return;
    // (diag128_factor_invert stops after the sub-blocks that hold rows <= n: the blocks behind them are never read)
    diag128_factor_invert(m.sL, m.sW, m.sT, m.sRd, m.sCol, 0, n, &sfail, nullptr);
    // ---- (z.z, 2 sum log L_ii): the operations of potrf_diag_kernel's one-block branch, in its order
    __syncthreads();
    double qq = 0.0, lg = 0.0;
    if (tid < NB && tid < n) {
        const double zi = m.sL[blk_off(n >> 4, tid >> 4) + bidx(n & 15, tid & 15)];
        qq = zi * zi;
        lg = log(m.sL[blk_off(tid >> 4, tid >> 4) + bidx(tid & 15, tid & 15)]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        qq += __shfl_xor(qq, o);
        lg += __shfl_xor(lg, o);
    }
    __syncthreads();
    double* red = m.sW;
    if ((tid & 63) == 0 && tid < NB) {
        red[tid >> 6] = qq;
        red[2 + (tid >> 6)] = lg;
    }
    __syncthreads();
    // ---- accept test (mcmc_accept_kernel's, for this walker)
    const int half = st.k / 2, sw = start ? first + w : h * half + w;
    if (tid == 0) {
        const double lp = mcmc_lnprob(prior, sfail, red[0] + red[1], 2.0 * (red[2] + red[3]), n);
        if (lp != lp) atomicOr(st.d_err, 1);
        int acc = 0;
        if (start) {
            if (lp == __builtin_huge_val()) atomicOr(st.d_err, 2);
            st.d_lnp[sw] = lp;
        } else {
            const size_t r = ((size_t)it * 2 + h) * half + w;
            const double lnpdiff = mcmc_lnpdiff(P, log(z), lp, st.d_lnp[sw]);
            if (lnpdiff > log(st.d_ua[r])) {
                acc = 1;
                st.d_lnp[sw] = lp;
                st.d_nacc[sw] += 1;
            }
            if (st.d_lnprob) st.d_lnprob[(size_t)sw * st.n_steps + it] = st.d_lnp[sw];
        }
        *sflag = acc;
    }
    __syncthreads();
    if (start) return;
    const bool acc = *sflag != 0;
    for (int p = tid, e = 0; p < P; p += 256, ++e) {
        double* pp = st.d_pos + (size_t)sw * P + p;
        const double v = acc ? (e == 0 ? q0 : q1) : *pp;
        if (acc) *pp = v;
        if (st.d_chain) st.d_chain[((size_t)sw * st.n_steps + it) * P + p] = v;
    }
}

int launch_mcmc_block_step(robo_gp* gp, const McmcState& st, int start, int first, int h, int it) {
    const int ns = start ? st.ns_eval : st.k / 2;
    const bool one_tile = gp->n + 1 <= GT;
#define ROBO_BLOCK_STEP(KIND, NG)                                                                                     \
    hipLaunchKernelGGL((mcmc_block_step_kernel<KIND, NG>), dim3(ns), dim3(256 * NG), 0, gp->ctx->stream, st, start, first, \
                       h, it, (const double*)gp->d_X, (const double*)gp->d_y)
    if (gp->kind == ROBO_KERNEL_MATERN52_ARD) {
        if (one_tile) ROBO_BLOCK_STEP(ROBO_KERNEL_MATERN52_ARD, 1);
        else ROBO_BLOCK_STEP(ROBO_KERNEL_MATERN52_ARD, 3);
    } else {
        if (one_tile) ROBO_BLOCK_STEP(ROBO_KERNEL_RBF_ARD, 1);
        else ROBO_BLOCK_STEP(ROBO_KERNEL_RBF_ARD, 3);
    }
#undef ROBO_BLOCK_STEP
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// Panel solve  X L_kk^T = A_ik  for the rows below the diagonal block of panel k, by block forward substitution
// over the eight 16-column blocks with only the diagonal inverses W_ss (no 128x128 inverse is needed, so the
// diagonal-block kernel does not have to form one on the fit's critical path).  64 rows per workgroup, one
// 16-row strip per wave, entirely in registers:  with Y_s = X_s^T (16 x 16: panel column within block s by strip
// row) held in the MFMA accumulator layout,
//      Y_s = W_ss (A_s^T - sum_{c<s} L_sc Y_c)
// and register r of an accumulator-layout Y_c IS the B fragment of its rows 4r..4r+3, so every product takes its
// B operand straight from the previous results and its A operand (L_sc or W_ss, shared by the whole workgroup)
// from the LDS image of the diagonal block: 36 products = 144 MFMAs per strip, no LDS round trip on the chain.
constexpr int PANEL_SMEM_DOUBLES = NBLK * BLK;   // the 28 strictly lower blocks of L_kk + the 8 W_ss in the diagonal slots: 72 KB,
                                                 // two workgroups per CU (r05; 88 KB with the unused diagonal blocks of L: one)

// Within every 16-block the panel kernel indexes panel columns through the 4x4 index transpose pi(a) = (a >> 2) | ((a & 3) << 2)
// (an involution): register r of lane (i = l & 15, g = l >> 4) of an accumulator-layout Y_s is then X[strip row i][16 s + 4 g + r],
// i.e. FOUR CONSECUTIVE doubles of the strip's row -- the strip is loaded and stored with 16-byte accesses (r02o: the
// 8-byte column-strided form cost 11.2k cycles of loads and 7k of stores around a 11.8k-cycle chain).  The LDS images of
// L_sc and W_ss are permuted the same way in rows and columns when they are staged, which costs nothing.
__device__ __forceinline__ constexpr int pi16(int a) { return (a >> 2) | ((a & 3) << 2); }
__device__ __forceinline__ constexpr int tri_row(int b) {      // block index -> (bi, bj) of blk_off, compile time
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= b) ++i;
    return i;
}

__global__ __launch_bounds__(256) void potrf_panel_kernel(double* __restrict__ K, size_t k_stride, int ld, int k,
                                                          const double* __restrict__ Linv, size_t linv_stride,
                                                          long long* __restrict__ dbg) {
    __shared__ double smem[PANEL_SMEM_DOUBLES];
    double* sL = smem;
    K += (size_t)blockIdx.y * k_stride;
    Linv += (size_t)blockIdx.y * linv_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* Kd = K + ((size_t)k * NB) * ld + (size_t)k * NB;
    const double* Wg = Linv + (size_t)k * NB * NB;
    if (dbg && tid == 0 && blockIdx.x == 0) dbg[0] = clock64();
    // this strip's rows in the (column-permuted) accumulator layout: registers 0..3 of lane l <- A[row l & 15][16 s + 4 (l >> 4) + 0..3]
    const size_t row = (size_t)(k + 1) * NB + (size_t)blockIdx.x * 64 + wave * 16 + (lane & 15);
    double* Arow = K + row * ld + (size_t)k * NB + 4 * (lane >> 4);
    v4d y[NSB];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        const double2* p = reinterpret_cast<const double2*>(Arow + s * SB);
        const double2 lo = p[0], hi = p[1];
        y[s] = v4d{lo.x, lo.y, hi.x, hi.y};
    }
    // stage the strictly lower blocks of L_kk and, in the slots of its diagonal blocks (which the substitution never
    // reads), the eight W_ss: two adjacent columns per thread and step, rows and columns permuted
    {
        const int half = tid >> 7, pr = tid & 127, r = pr >> 3, c = (pr & 7) * 2;
        const int pos0 = bidx(pi16(r), pi16(c)), pos1 = bidx(pi16(r), pi16(c + 1));
        double2 v[NBLK / 2];
#pragma unroll
        for (int e = 0; e < NBLK / 2; ++e) {
            const int b0 = 2 * e, b1 = 2 * e + 1;        // this step's two blocks (threads 0..127 / 128..255)
            const int bi0 = tri_row(b0), bj0 = b0 - bi0 * (bi0 + 1) / 2, bi1 = tri_row(b1), bj1 = b1 - bi1 * (bi1 + 1) / 2;
            const int bi = half ? bi1 : bi0, bj = half ? bj1 : bj0;
            const double* src = bi == bj ? Wg + (bi * SB + r) * NB + bi * SB + c
                                         : Kd + (size_t)(bi * SB + r) * ld + bj * SB + c;
            v[e] = *reinterpret_cast<const double2*>(src);
        }
#pragma unroll
        for (int e = 0; e < NBLK / 2; ++e) {
            double* dst = smem + (2 * e + half) * BLK;
            dst[pos0] = v[e].x;
            dst[pos1] = v[e].y;
        }
    }
    __syncthreads();
    if (dbg && tid == 0 && blockIdx.x == 0) dbg[1] = clock64();
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        v4d t = y[s];
#pragma unroll
        for (int c = 0; c < s; ++c) {
            const Frag4 a = frag_row(sL + blk_off(s, c), lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) t = mfma_f64(-a.v[r], y[c][r], t);
        }
        const Frag4 w = frag_row(sL + blk_off(s, s), lane);
        v4d o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) o = mfma_f64(w.v[r], t[r], o);
        y[s] = o;
    }
    if (dbg && tid == 0 && blockIdx.x == 0) dbg[2] = clock64() + (long long)(y[NSB - 1][0] == 12345.678);
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        double2* p = reinterpret_cast<double2*>(Arow + s * SB);
        p[0] = make_double2(y[s][0], y[s][1]);
        p[1] = make_double2(y[s][2], y[s][3]);
    }
    if (dbg && tid == 0 && blockIdx.x == 0) dbg[3] = clock64();
}

// Off-diagonal 16x16 blocks of the inverses W = L_kk^-1 of ALL diagonal blocks at once (one workgroup per block,
// after the factorisation): the posterior's TRSM (predict.hip) and the likelihood gradient (gradient.hip) multiply
// with explicit 128 x 128 inverses.  Column by column: with S_ij = sum_{j<=c<i} L_ic W_cj kept in the W_ij slot
// until block row i is final, stage s does  W_sj = -W_ss S_sj  (j < s)  and then  S_ij += L_is W_sj  for every
// block row i > s -- 7, 13, 17, 19, 19, 17, 13 independent products in stages 0..6, handed out dynamically to the
// four waves, one barrier per half-stage.
//
// Tail of the factorisation, in the same launch (r02z: the separate pack and log-likelihood launches and the two
// device-to-host copies behind them cost a 1.8 ms fit ~25 us; one tiny finishing launch remains):
//   * Wp != nullptr: the inverse is also written as packed MFMA A-operand fragments (see linv_pack_kernel);
//   * every workgroup reduces its 128 rows' share of  sum log L_ii  and  z.z  (z = row n of the factor) into
//     ll_part (added up by loglik_finish_kernel).
__global__ __launch_bounds__(256) void potrf_inverse_kernel(const double* __restrict__ K, size_t k_stride, int ld,
                                                            double* __restrict__ Linv, size_t linv_stride, int n_real,
                                                            double* __restrict__ Wp, double* __restrict__ ll_part,
                                                            int want_inverse) {
    __shared__ double smem[2 * NBLK * BLK + 32];
    if (!want_inverse) {
        // likelihood evaluations (the MCMC / L-BFGS inner loops): only this block's share of the two sums -- at
        // BO-typical N < 128 the inverse was 20 of the 47 us of a batched pass
        const double* Ks = K + (size_t)blockIdx.y * k_stride;
        const int kb = blockIdx.x, t = threadIdx.x, r = kb * NB + t;
        double q = 0.0, lg = 0.0, dmin = __builtin_huge_val(), dmax = 0.0;
        if (t < NB && r < n_real) {
            const double zi = Ks[(size_t)n_real * ld + r];
            q = zi * zi;
            const double d = Ks[(size_t)r * ld + r];
            lg = log(d);
            dmin = dmax = d;
        }
        for (int o = 32; o > 0; o >>= 1) {
            q += __shfl_xor(q, o);
            lg += __shfl_xor(lg, o);
            dmin = fmin(dmin, __shfl_xor(dmin, o));
            dmax = fmax(dmax, __shfl_xor(dmax, o));
        }
        double* red = smem;
        if ((t & 63) == 0 && t < NB) {
            red[t >> 6] = q;
            red[2 + (t >> 6)] = lg;
            red[4 + (t >> 6)] = dmin;
            red[6 + (t >> 6)] = dmax;
        }
        __syncthreads();
        if (t == 0) {
            double* part = ll_part + ((size_t)blockIdx.y * gridDim.x + kb) * 4;
            part[0] = red[0] + red[1];
            part[1] = red[2] + red[3];
            part[2] = fmin(red[4], red[5]);
            part[3] = fmax(red[6], red[7]);
        }
        return;
    }
    double* sL = smem;
    double* sW = smem + NBLK * BLK;
    int* ctr = reinterpret_cast<int*>(smem + 2 * NBLK * BLK);
    K += (size_t)blockIdx.y * k_stride;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* Kd = K + ((size_t)k * NB) * ld + (size_t)k * NB;
    double* Wg = Linv + (size_t)blockIdx.y * linv_stride + (size_t)k * NB * NB;
    for (int bi = 0; bi < NSB; ++bi) {
        for (int bj = 0; bj <= bi; ++bj)
            sL[blk_off(bi, bj) + bidx(tid >> 4, tid & 15)] = Kd[(size_t)(bi * SB + (tid >> 4)) * ld + bj * SB + (tid & 15)];
        sW[blk_off(bi, bi) + bidx(tid >> 4, tid & 15)] = Wg[(bi * SB + (tid >> 4)) * NB + bi * SB + (tid & 15)];
    }
    if (tid < 2 * NSB) ctr[tid] = 0;
    __syncthreads();
    const v4d zero = {0.0, 0.0, 0.0, 0.0};
    // block rows behind the augmented row are identity padding: their off-diagonal inverse blocks are zero
    int nsb = (n_real + 1 - k * NB + SB - 1) / SB;
    nsb = nsb < 1 ? 1 : (nsb > NSB ? NSB : nsb);
    for (int s = 0; s + 1 < nsb; ++s) {
        // finalise block row s: W_sj = -W_ss S_sj, j < s
        for (;;) {
            int j = 0;
            if (lane == 0) j = atomicAdd(ctr + 2 * s, 1);
            j = __builtin_amdgcn_readlane(j, 0);
            if (j >= s) break;
            double* Wsj = sW + blk_off(s, j);
            v4d w = zero;
            w = blk_mma_nn<true>(sW + blk_off(s, s), Wsj, lane, w);
            wave_lds_fence();   // S_sj fully read before W_sj replaces it
            blk_store_c(Wsj, lane, w);
        }
        __syncthreads();
        // S_ij += L_is W_sj for i > s, j <= s: (7 - s)(s + 1) independent products
        const int rows = nsb - 1 - s, ntask = rows * (s + 1);
        for (;;) {
            int t = 0;
            if (lane == 0) t = atomicAdd(ctr + 2 * s + 1, 1);
            t = __builtin_amdgcn_readlane(t, 0);
            if (t >= ntask) break;
            const int j = t / rows, i = s + 1 + t % rows;
            double* Sij = sW + blk_off(i, j);
            const Frag4 a = frag_row(sL + blk_off(i, s), lane), b = frag_col(sW + blk_off(s, j), lane);
            v4d acc = j < s ? blk_load_c(Sij, lane) : zero;
            acc = frag_mma<false>(a, b, acc);
            blk_store_c(Sij, lane, acc);
        }
        __syncthreads();
    }
    // last block row: W_lj = -W_ll S_lj
    for (int j = wave; j < nsb - 1; j += 4) {
        double* S = sW + blk_off(nsb - 1, j);
        v4d w = zero;
        w = blk_mma_nn<true>(sW + blk_off(nsb - 1, nsb - 1), S, lane, w);
        wave_lds_fence();
        blk_store_c(S, lane, w);
    }
    __syncthreads();
    for (int bi = 1; bi < NSB; ++bi)
        for (int bj = 0; bj < bi; ++bj)
            Wg[(bi * SB + (tid >> 4)) * NB + bj * SB + (tid & 15)] =
                bi < nsb ? sW[blk_off(bi, bj) + bidx(tid >> 4, tid & 15)] : 0.0;
    if (Wp) {
        // one 16x16 block = 4 k-steps x 64 lanes = the 256 threads; nothing to decode
        double* wp = Wp + (size_t)k * WP_BLOCK + tid;                 // tid = 64 kk + lane
        const int src = bidx(pi16(lane & 15), 4 * wave + (lane >> 4));
#pragma unroll
        for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
            for (int jb = 0; jb <= cb; ++jb)
                wp[(wp_offset(cb) + 4 * jb) * 64] = (cb == jb || cb < nsb) ? sW[blk_off(cb, jb) + src] : 0.0;
    }
    // ---- log-likelihood terms of rows 128 k .. 128 k + 127 (rows >= n_real: augmented row / padding, no share)
    {
        double q = 0.0, lg = 0.0, dmin = __builtin_huge_val(), dmax = 0.0;
        const int r = k * NB + tid;
        if (tid < NB && r < n_real) {
            const double zi = K[(size_t)n_real * ld + r];
            q = zi * zi;
            const double d = sL[blk_off(tid >> 4, tid >> 4) + bidx(tid & 15, tid & 15)];
            lg = log(d);
            dmin = dmax = d;
        }
        for (int o = 32; o > 0; o >>= 1) {
            q += __shfl_xor(q, o);
            lg += __shfl_xor(lg, o);
            dmin = fmin(dmin, __shfl_xor(dmin, o));
            dmax = fmax(dmax, __shfl_xor(dmax, o));
        }
        __syncthreads();            // sW / the counters are dead: reuse the scratch
        double* red = smem + 2 * NBLK * BLK;
        if (lane == 0 && wave < 2) {
            red[wave] = q;
            red[2 + wave] = lg;
            red[4 + wave] = dmin;
            red[6 + wave] = dmax;
        }
        __syncthreads();
        if (tid == 0) {
            double* part = ll_part + ((size_t)blockIdx.y * gridDim.x + k) * 4;
            part[0] = red[0] + red[1];
            part[1] = red[2] + red[3];
            part[2] = fmin(red[4], red[5]);
            part[3] = fmax(red[6], red[7]);
        }
    }
}

// (z.z, 2 sum log L_ii) of every sample from the per-block partials, added in block order (a fixed summation order);
// with host_out (pinned, device-visible; [S][5]) the result, the failure flag and the extreme diagonal entries go
// straight to the host.
// A launch of its own: handing the partials over INSIDE the tail kernel (agent-scope release + arrival counter) was
// measured at +16 us -- the release writes back an L2 full of the factorisation's dirty lines (r02z).
__global__ __launch_bounds__(64) void loglik_finish_kernel(const double* __restrict__ ll_part, int nbk,
                                                           double* __restrict__ out, const int* __restrict__ fail,
                                                           double* __restrict__ host_out) {
    const int smp = blockIdx.x;
    if (threadIdx.x != 0) return;
    const double* part = ll_part + (size_t)smp * nbk * 4;
    double sq = 0.0, sl = 0.0, dmin = __builtin_huge_val(), dmax = 0.0;
    for (int b = 0; b < nbk; ++b) {
        sq += part[4 * b];
        sl += part[4 * b + 1];
        dmin = fmin(dmin, part[4 * b + 2]);
        dmax = fmax(dmax, part[4 * b + 3]);
    }
    out[2 * smp] = sq;
    out[2 * smp + 1] = 2.0 * sl;
    if (host_out) {
        host_out[5 * smp] = sq;
        host_out[5 * smp + 1] = 2.0 * sl;
        host_out[5 * smp + 2] = (double)fail[smp];
        host_out[5 * smp + 3] = dmin;      // extreme diagonal entries of L over the training rows (conditioning
        host_out[5 * smp + 4] = dmax;      // estimate for the explicit-inverse posterior, api.hip use_winv)
    }
}

// Tile (k+1, k+1) of the trailing update for the fused diagonal workgroup:  C_lower - P P^T  straight
// into the block-packed LDS image sL (P = A_{k+1,k}, 128 x 128).  Only the 36 lower 16x16 tiles are
// formed and both factors are the same panel (one staged operand): wave w owns block rows 7-w and w
// (8-w + w+1 = 9 tiles each), 288 MFMAs per wave against the 512 of a full 128 x 128 tile -- the
// lone-workgroup update in front of the pivot chain drops from 13.6 to ~8 us.
__device__ __forceinline__ void diag_tile_update(const double* __restrict__ P, const double* __restrict__ C, int ld,
                                                 double* stage, double* sL) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rA = 7 - wave, rB = wave;
    v4d acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int br = i <= rA ? rA : rB, bj = i <= rA ? i : i - rA - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[i][r] = C[(size_t)(br * SB + (lane >> 4) + 4 * r) * ld + bj * SB + (lane & 15)];
    }
    Tile4 rp = tile_load_regs<128>(P, ld, 0);
    tile_store_lds<128>(stage, rp);
    __syncthreads();
    for (int kt = 0; kt < NB / BK; ++kt) {
        const double* cur = stage + (kt & 1) * STAGE_B;
        double* nxt = stage + ((kt + 1) & 1) * STAGE_B;
        const bool more = kt + 1 < NB / BK;
        if (more) rp = tile_load_regs<128>(P, ld, (kt + 1) * BK);
        const double* f = cur + (lane & 15) * LDS_LD + (lane >> 4);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const double aA = -f[rA * SB * LDS_LD + kk * 4], aB = -f[rB * SB * LDS_LD + kk * 4];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int bj = i <= rA ? i : i - rA - 1;
                acc[i] = mfma_f64(i <= rA ? aA : aB, f[bj * SB * LDS_LD + kk * 4], acc[i]);
            }
        }
        if (more) tile_store_lds<128>(nxt, rp);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int br = i <= rA ? rA : rB, bj = i <= rA ? i : i - rA - 1;
        blk_store_c(sL + blk_off(br, bj), lane, acc[i]);
    }
}

// t-th tile of the lower triangle, row-major: t = ii (ii + 1) / 2 + jj
__device__ __forceinline__ void tri_decode(int t, int& ii, int& jj) {
    int q = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((q + 1) * (q + 2) / 2 <= t) ++q;
    while (q * (q + 1) / 2 > t) --q;
    ii = q;
    jj = t - q * (q + 1) / 2;
}

// 128-deep updates of the 128 x 128 tiles t = t0, t0 + stride, ... < ntiles of the trailing matrix by ONE workgroup
// (the fused step kernel runs one workgroup per CU: the diagonal block's LDS image sizes every workgroup).
// Phase stamps of the one-tile-per-workgroup version (r02y): read C + first operands 10.9k cycles, k-loop 42.5k
// (512 MFMAs per wave = 32.8k at 64 cycles; the chip holds ~2.0 GHz of its 2.4 under this load), store 4.4k; with every
// workgroup launched at the same instant, rounds of tiles simply add up (kernel trace: 26-29 us per round, 52-59 us
// for two, 81 us for three).  Here
//   * the next tile's C arrives in a second accumulator set while the current tile is multiplied (two 16 x 16 blocks
//     per k-tile, so an operand wait never waits for more than 1/8 of C), its first operand k-tile is requested before
//     the current tile is stored, and the stores drain under the next tile's MFMAs;
//   * the 32 k-steps of a tile are one software pipeline (a wave issues in order and is alone on its SIMD): fragments
//     of step g+1 are read before the MFMAs of step g, the next k-tile goes to LDS before the last step of the
//     current one, and the barrier + first fragment read of the next k-tile sit in the MIDDLE of that step's MFMAs.
// Measured: k-loop 42.5k -> 40.1k cycles, two tiles 52-59 -> 50 us, three 81 -> 75 us, fit 1.87 -> 1.82 ms.
// Same arithmetic as gemm_nt<4, true> on acc = C: products are accumulated on top of the stored value in ascending k.
// shift = 1: t indexes the triangle WITHOUT its first block row and column (tiles (ii + 1, jj + 1): the follower form of the
// step kernel, whose first-column tiles belong to the panel followers)
__device__ __forceinline__ void update_tiles_persistent(double* __restrict__ K, int ld, int k, int t0, int stride,
                                                        int ntiles, double* smem, int shift = 0) {
    constexpr int SA = stage_a<4>(), ST = SA + STAGE_B, NK = NB / BK;
    int ii, jj;
    tri_decode(t0, ii, jj);
    ii += shift;
    jj += shift;
    const double* A = K + ((size_t)(k + 1 + ii) * NB) * ld + (size_t)k * NB;
    const double* B = K + ((size_t)(k + 1 + jj) * NB) * ld + (size_t)k * NB;
    double* C = K + ((size_t)(k + 1 + ii) * NB) * ld + (size_t)(k + 1 + jj) * NB;
    Acc acc, nxt;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.t[tm][tn][r] = C[(size_t)acc_row<4>(tm, r) * ld + acc_col(tn)];
    Tile4 ra = tile_load_regs<128>(A, ld, 0), rb = tile_load_regs<128>(B, ld, 0);
    for (int t = t0;; t += stride) {
        const int tn_ = t + stride;
        const bool more = tn_ < ntiles;
        const double *An = A, *Bn = B;
        double* Cn = C;
        if (more) {
            tri_decode(tn_, ii, jj);
            ii += shift;
            jj += shift;
            An = K + ((size_t)(k + 1 + ii) * NB) * ld + (size_t)k * NB;
            Bn = K + ((size_t)(k + 1 + jj) * NB) * ld + (size_t)k * NB;
            Cn = K + ((size_t)(k + 1 + ii) * NB) * ld + (size_t)(k + 1 + jj) * NB;
        }
        tile_store_lds<128>(smem, ra);
        tile_store_lds<128>(smem + SA, rb);
        __syncthreads();
        // 32 k-steps (8 k-tiles x 4) as one software pipeline: fragments of step g+1 are read before the MFMAs of
        // step g; the next k-tile goes to LDS before the last step of the current one, and the barrier + the first
        // fragment read of the next k-tile sit in the MIDDLE of that step's 16 MFMAs
        {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int fa_off = ((wave >> 1) * 64 + (lane & 15)) * LDS_LD + (lane >> 4);
            const int fb_off = SA + ((wave & 1) * 64 + (lane & 15)) * LDS_LD + (lane >> 4);
            double fa[2][4], fb[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fa[0][q] = smem[fa_off + q * 16 * LDS_LD];
                fb[0][q] = smem[fb_off + q * 16 * LDS_LD];
            }
#pragma unroll
            for (int g = 0; g < 4 * NK; ++g) {
                const int kt = g >> 2, kk = g & 3, c = g & 1, n = c ^ 1;
                const double* cur = smem + (kt & 1) * ST;
                double* oth = smem + ((kt + 1) & 1) * ST;
                if (kk == 0) {
                    if (kt + 1 < NK) {
                        ra = tile_load_regs<128>(A, ld, (kt + 1) * BK);
                        rb = tile_load_regs<128>(B, ld, (kt + 1) * BK);
                    } else if (more) {
                        ra = tile_load_regs<128>(An, ld, 0);
                        rb = tile_load_regs<128>(Bn, ld, 0);
                    }
                    if (more) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int tm = (2 * kt + h) >> 2, tn = (2 * kt + h) & 3;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                nxt.t[tm][tn][r] = Cn[(size_t)acc_row<4>(tm, r) * ld + acc_col(tn)];
                        }
                    }
                }
                if (kk < 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        fa[n][q] = cur[fa_off + q * 16 * LDS_LD + (kk + 1) * 4];
                        fb[n][q] = cur[fb_off + q * 16 * LDS_LD + (kk + 1) * 4];
                    }
                } else if (kt + 1 < NK) {
                    tile_store_lds<128>(oth, ra);
                    tile_store_lds<128>(oth + SA, rb);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) fa[c][q] = -fa[c][q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc.t[tm][tn] = mfma_f64(fa[c][tm], fb[c][tn], acc.t[tm][tn]);
                __builtin_amdgcn_sched_barrier(0);
                if (kk == 3 && kt + 1 < NK) {
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        fa[n][q] = oth[fa_off + q * 16 * LDS_LD];
                        fb[n][q] = oth[fb_off + q * 16 * LDS_LD];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int tm = 2; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc.t[tm][tn] = mfma_f64(fa[c][tm], fb[c][tn], acc.t[tm][tn]);
            }
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) C[(size_t)acc_row<4>(tm, r) * ld + acc_col(tn)] = acc.t[tm][tn][r];
        if (!more) break;
        acc = nxt;
        A = An;
        B = Bn;
        C = Cn;
    }
}

// 32 TM rows of tile t (row-major index in the lower triangle of the trailing matrix) by this workgroup: the plain NT GEMM on
// top of the stored C, ascending k -- the same bits as the 128-row tile kernel leaves there
template <int TM>
__device__ __forceinline__ void update_subtile(double* __restrict__ K, int ld, int k, int t, int h, double* smem,
                                               int shift = 0) {
    int ii, jj;
    tri_decode(t, ii, jj);
    ii += shift;
    jj += shift;
    const size_t row0 = (size_t)(k + 1 + ii) * NB + (size_t)h * (32 * TM);
    const double* A = K + row0 * ld + (size_t)k * NB;
    const double* B = K + ((size_t)(k + 1 + jj) * NB) * ld + (size_t)k * NB;
    double* C = K + row0 * ld + (size_t)(k + 1 + jj) * NB;
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

// Trailing update of step k fused with the NEXT diagonal block:
//   A_ij <- A_ij - A_ik * A_jk^T   for k < j <= i (right-looking, K = 128), and workgroup 0 -- which
//   owns tile (k+1, k+1) -- goes on to factor and invert it (diag128_factor_invert) while the other
//   workgroups are still updating: the 66 us single-workgroup diagonal kernel, the fit's bottleneck,
//   runs in the shadow of the chip-wide update instead of after it.  (Doing the same with two
//   streams was slower: cross-queue event waits cost more than they hid, r01q.)
// Tile height 32*TM is chosen per step by the launcher: a 128x128x128 tile is 512 MFMAs deep
// (>= 13.6 us per wave), so late steps with few blocks use shorter tiles to cover the chip.
// (A two-level variant with 512-deep updates was measured slower: its strip updates put
// <= 32 workgroups on the critical path.)  Every workgroup is sized for the diagonal block's 150 KB
// of LDS, i.e. one per CU.
// FUSED = false is the plain trailing update (diagonal blocks by potrf_diag_kernel): used for batched
// fits, where S diagonal workgroups already run side by side and two update workgroups per CU matter
// more (measured, 27 thetas at N = 4096: 0.69 ms per theta unfused, 0.78 fused).
template <int TM, bool FUSED>
__global__ __launch_bounds__(256) void potrf_step_kernel(double* __restrict__ K, size_t k_stride, int ld, int k,
                                                         int n_real, double* __restrict__ Linv, size_t linv_stride,
                                                         int* __restrict__ fail, int kop, int depth, int first_col,
                                                         int ntiles, int thin_row) {
    // thin_row (batched updates only, else -1): the block row that holds nothing but the augmented row (n a multiple of
    // 128: row n is the first row of the last block, the rows below it are identity rows whose updates are zero) -- its
    // tiles are computed 32 rows high instead of 128
    // k: trailing base (tiles cover block rows/columns > k); kop: first block column of the panel
    // operand(s); depth: contraction length (128, or 256 = two panels at once, see launch_potrf);
    // first_col != 0: block column k+1 only
    __shared__ double smem[FUSED ? DIAG_SMEM_DOUBLES : gemm_smem_doubles<TM>()];
    K += (size_t)blockIdx.y * k_stride;
    if (FUSED && blockIdx.x == 0) {
        // ---- tile (k+1, k+1): update, then straight into the block-packed LDS image of the diagonal kernel
        const size_t d0 = (size_t)(k + 1) * NB;
        const double* A = K + d0 * ld + (size_t)k * NB;
        double* C = K + d0 * ld + d0;
        const DiagSmem m = diag_carve(smem);
        diag_tile_update(A, C, ld, m.sW, m.sL);   // staging in the (still unused) W image
        __syncthreads();
        diag128_factor_invert(m.sL, m.sW, m.sT, m.sRd, m.sCol, (k + 1) * NB, n_real, fail + blockIdx.y, nullptr);
        diag_writeback(m, C, ld, Linv + (size_t)blockIdx.y * linv_stride + (size_t)(k + 1) * NB * NB);
        return;
    }
    if (FUSED && TM == 4) {
        // workgroups 1 .. gridDim.x-1 share tiles 1 .. ntiles-1 (tile 0 is workgroup 0's)
        const int W = (int)gridDim.x - 1, Tp = ntiles - 1;
        int full_end = ntiles, split = 0, left = 0;
        if (first_col != 0 && Tp > W) {
            // tail split (first_col doubles as its switch on this path): with R full rounds of tiles per workgroup, the
            // `left` tiles of the ragged last round are cut into halves / quarters and spread over 2 / 4 times as many
            // workgroups -- the launch ends a (half / quarter tile) after the last full round instead of a whole tile
            // after it (k = 0: 527 tiles on 255 workgroups = 2 rounds + 17 tiles; k = 5..9: one round + 122 .. 20)
            const int R = Tp / W;
            left = Tp - R * W;
            if (left > 0) split = 4 * left <= W ? 4 : (2 * left <= W ? 2 : 0);
            if (split) full_end = 1 + R * W;
        }
        update_tiles_persistent(K, ld, k, (int)blockIdx.x, W, full_end, smem);
        if (split) {
            const int q = (int)blockIdx.x - 1;
            if (q < split * left) {
                __syncthreads();       // the last k-step's fragment reads before the sub-tile's staging stores
                if (split == 4) update_subtile<1>(K, ld, k, full_end + q / 4, q % 4, smem);
                else update_subtile<2>(K, ld, k, full_end + q / 2, q % 2, smem);
            }
        }
        return;
    }
    constexpr int SPLIT = 4 / TM;                    // row sub-tiles per 128-row block
    // sub-tile index; when fused, tile 0's sub-tiles belong to workgroup 0
    const int b = FUSED ? (int)blockIdx.x - 1 + SPLIT : (int)blockIdx.x;
    int ii, jj;
    {
        const int t = b / SPLIT;
        if (first_col) {
            ii = t;
            jj = 0;
        } else {
            int q = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while ((q + 1) * (q + 2) / 2 <= t) ++q;
            while (q * (q + 1) / 2 > t) --q;
            ii = q;
            jj = t - q * (q + 1) / 2;
        }
    }
    const int i = k + 1 + ii, j = k + 1 + jj, h = b % SPLIT;
    const size_t row0 = (size_t)i * NB + (size_t)h * (32 * TM);
    const double* A = K + row0 * ld + (size_t)kop * NB;
    const double* B = K + ((size_t)j * NB) * ld + (size_t)kop * NB;
    double* C = K + row0 * ld + (size_t)j * NB;
    if (!FUSED && i == thin_row) {
        // the augmented row's block: rows 0..31 of the tile (row n is its row 0), same products in the same order
        if (h != 0) return;
        AccT<1> acc1;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1.t[0][tn][r] = C[(size_t)acc_row<1>(0, r) * ld + acc_col(tn)];
        gemm_nt<1, true>(A, ld, B, ld, 0, depth, acc1, smem);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)acc_row<1>(0, r) * ld + acc_col(tn)] = acc1.t[0][tn][r];
        return;
    }
    AccT<TM> acc;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.t[tm][tn][r] = C[(size_t)acc_row<TM>(tm, r) * ld + acc_col(tn)];
    gemm_nt<TM, true>(A, ld, B, ld, 0, depth, acc, smem);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)acc_row<TM>(tm, r) * ld + acc_col(tn)] = acc.t[tm][tn][r];
}

// The LAST column of a panel is one block, W_77 = L_77^-1, and it is what every follower waits for when the diagonal block
// ends.  Its 256 entries are their own flag: the gram kernel -- which precedes every factorisation -- fills the slot with
// FOLLOW_SENTINEL (common.h) and zeroes the progress words, the diagonal workgroup's ordinary publication overwrites the
// sentinel, and a follower re-reads ITS element until it is no sentinel -- one round trip (the payload) instead of two
// (progress word, then payload) behind the last pivot.
constexpr long long W_SENTINEL = FOLLOW_SENTINEL;

// ---- the panel solve as a FOLLOWER of the diagonal block (r06) -----------------------------------------------------------
// potrf_panel_kernel's substitution, column by column instead of row by row: as soon as block column c of L_kk and W_cc are
// published (diag128_factor_invert with a DiagPub, progress word >= c + 1)
//      Y_c = W_cc t_c ;   t_s -= L_sc Y_c   for s > c
// -- every t_s still receives its updates in ascending c and every product its four MFMAs in the same order, so the strips
// are the panel kernel's bit for bit.  One 16-row strip per wave (64 rows per workgroup), operands of ONE column at a time
// in 16 KB of LDS (slot s: L_sc for s > c, slot c: W_cc; rows and columns permuted as in the panel kernel).
// The chain   panel (11 us) -> launch boundary -> tile update + 128 pivots   of the launch-per-phase factorisation loses its
// first link: the strips are final one hand-off (a few us) after the diagonal block's last pivot.
// NS: 16-row strips per wave (64 NS rows per workgroup; strip j of wave w = rows 64 j + 16 w ..): 1 in the single-theta step
// kernel (two followers per block row), 2 in the batched form (one follower per block row: half as many workgroups spin).
template <int NS = 1>
__device__ __forceinline__ void panel_follow(double* __restrict__ K, int ld, int kc, size_t row0, const unsigned* prog,
                                             const double* __restrict__ Wg, double* smem, int* fail, int n_real,
                                             int poll_sleep = 1) {
    // kc: block column being solved; row0: first of this workgroup's 64 NS rows; prog: progress word of panel kc;
    // poll_sleep: pauses of 128 cycles between two polls of the progress word (dozens of followers poll ONE word)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* Kd = K + ((size_t)kc * NB) * ld + (size_t)kc * NB;
    double* Arow[NS];
    v4d y[NS][NSB];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        Arow[j] = K + (row0 + j * 64 + wave * 16 + (lane & 15)) * ld + (size_t)kc * NB + 4 * (lane >> 4);
#pragma unroll
        for (int s = 0; s < NSB; ++s) {
            const double2* p = reinterpret_cast<const double2*>(Arow[j] + s * SB);
            const double2 lo = p[0], hi = p[1];
            y[j][s] = v4d{lo.x, lo.y, hi.x, hi.y};
        }
    }
    const int r = tid >> 4, c16 = tid & 15;
    const int pos = bidx(pi16(r), pi16(c16));
    int* sflag = reinterpret_cast<int*>(smem + NBLK * BLK);
    const int nsb = diag_nsb(n_real, kc * NB);
    // A hand-off is two round trips (the poll, then the payload: ~1.5 + ~2 us on a busy chip) -- more than the ~2.9 us per
    // column the diagonal block needs.  So every pass takes ALL the columns that are in memory by now: a follower that is
    // behind (it starts late: its own tile update comes first) catches up several columns per pass, and one that keeps
    // pace pays the two round trips per pass, not per column.  LDS: the panel kernel's image, slot blk_off(s, c) =
    // L_sc (s > c) or W_cc (s = c), rows and columns permuted.
    int c = 0;
    while (c < NSB) {
        if (c == NSB - 1) {
            // the last column: W_77's entries are their own flag (W_SENTINEL until the diagonal workgroup's store lands)
            const double* src = Wg + (size_t)((NSB - 1) * SB + r) * NB + (NSB - 1) * SB + c16;
            double v = 0.0;
            int ok = 0;
            for (unsigned spins = 0; spins <= PROG_SPIN_LIMIT; ++spins) {
                v = ld_agent(src);
                __syncthreads();         // (first pass: the previous pass's fragment reads are over; later: sflag was read)
                if (tid == 0) *sflag = 1;
                __syncthreads();
                if (__double_as_longlong(v) == W_SENTINEL) *sflag = 0;
                __syncthreads();
                ok = *sflag;
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) {
                if (tid == 0) *fail = -1;                 // (api.hip: reported as a run-time error, not as a matrix property)
                return;
            }
            smem[blk_off(NSB - 1, NSB - 1) + pos] = v;
            __syncthreads();
            const Frag4 w = frag_row(smem + blk_off(NSB - 1, NSB - 1), lane);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                v4d o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) o = mfma_f64(w.v[q], y[j][NSB - 1][q], o);
                double2* p = reinterpret_cast<double2*>(Arow[j] + (NSB - 1) * SB);
                p[0] = make_double2(o[0], o[1]);
                p[1] = make_double2(o[2], o[3]);
            }
            return;
        }
        if (tid == 0) {
            unsigned spins = 0, v;
            int hi = -1;
            const unsigned need = diag_prog_need(c, nsb);
            while ((v = ld_agent_u32(prog)) < need) {
                for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(2);
                if (++spins > PROG_SPIN_LIMIT) break;
            }
            if (v >= need) {
                hi = c;
                while (hi + 1 < NSB - 1 && v >= diag_prog_need(hi + 1, nsb)) ++hi;      // (the last column: above)
            }
            *sflag = hi;
        }
        __syncthreads();                 // (also: the previous pass's fragment reads are over)
        const int chi = *sflag;
        if (chi < 0) {                   // the producer never arrived: flag the factorisation, leave the strips alone
            if (tid == 0) *fail = -1;                 // (api.hip: reported as a run-time error, not as a matrix property)
            return;
        }
        // stage columns c .. chi: one element per thread and block, L1-bypassing loads, four blocks in flight per thread
        {
            int col = c, sl = c;             // blocks (s, col), col = c .. chi, s = col .. 7, column by column
            const int nblk = (chi - c + 1) * NSB - ((chi * (chi + 1)) / 2 - (c * (c - 1)) / 2);
            for (int i = 0; i < nblk; i += 4) {
                double v[4];
                int slot[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    slot[u] = -1;
                    if (i + u < nblk) {
                        slot[u] = blk_off(sl, col);
                        v[u] = sl == col ? ld_agent(Wg + (size_t)(col * SB + r) * NB + col * SB + c16)
                                         : ld_agent(Kd + (size_t)(sl * SB + r) * ld + col * SB + c16);
                        if (++sl == NSB) {
                            ++col;
                            sl = col;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (slot[u] >= 0) smem[slot[u] + pos] = v[u];
            }
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < NSB; ++cc) {
            if (cc < c || cc > chi) continue;
            {
                const Frag4 w = frag_row(smem + blk_off(cc, cc), lane);
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    v4d o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) o = mfma_f64(w.v[q], y[j][cc][q], o);
                    y[j][cc] = o;
                    // column cc of the strip is final: on its way to memory while the later columns are still being solved
                    double2* p = reinterpret_cast<double2*>(Arow[j] + cc * SB);
                    p[0] = make_double2(o[0], o[1]);
                    p[1] = make_double2(o[2], o[3]);
                }
            }
#pragma unroll
            for (int sl = cc + 1; sl < NSB; ++sl) {
                const Frag4 a = frag_row(smem + blk_off(sl, cc), lane);
#pragma unroll
                for (int j = 0; j < NS; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) y[j][sl] = mfma_f64(-a.v[q], y[j][cc][q], y[j][sl]);
            }
        }
        c = chi + 1;
    }
}

// Step k of the single-theta factorisation with the NEXT panel inside (potrf_follow):
//   workgroup 0            tile (k+1, k+1): update by panel k, 128 pivots -- publishing block column after block column
//   workgroups 1 .. nfol   two per block row i >= k+2: their 64 rows of tile (i, k+1) receive panel k (the 64-row sub-tile of
//                          the trailing update), then follow workgroup 0: panel k+1 is final a hand-off after its last pivot
//   the others             the remaining tiles (ii >= jj >= 1 of the trailing triangle), persistent as in potrf_step_kernel
// k = -1: the first diagonal block and panel 0 (nothing to update: workgroup 0 loads its tile, the followers their strips).
// Every workgroup carries the diagonal block's 150-KB LDS image, i.e. one per CU: with the grid <= the CU count all of them
// are resident and workgroup 0 -- dispatched first, waiting for nobody -- always makes progress; the followers' polls are bounded.
__global__ __launch_bounds__(256) void potrf_step_follow_kernel(double* __restrict__ K, size_t k_stride, int ld, int k,
                                                                int n_real, double* __restrict__ Linv, size_t linv_stride,
                                                                int* __restrict__ fail, unsigned* __restrict__ prog,
                                                                int nfol, int ntiles, int tail_split, int pub_early,
                                                                int fol_rows, int poll_sleep) {
    __shared__ double smem[DIAG_SMEM_DOUBLES];
    K += (size_t)blockIdx.y * k_stride;
    Linv += (size_t)blockIdx.y * linv_stride;
    fail += blockIdx.y;
    prog += (size_t)blockIdx.y * PROG_STRIDE + (k + 1);
    const int b = (int)blockIdx.x;
    const size_t d0 = (size_t)(k + 1) * NB;
    double* Wg = Linv + (size_t)(k + 1) * NB * NB;
    if (b == 0) {
        double* C = K + d0 * ld + d0;
        const DiagSmem m = diag_carve(smem);
        if (k >= 0) {
            diag_tile_update(K + d0 * ld + (size_t)k * NB, C, ld, m.sW, m.sL);   // staging in the (still unused) W image
        } else {
            const int tid = threadIdx.x;
            for (int bi = 0; bi < NSB; ++bi)
                for (int bj = 0; bj <= bi; ++bj)
                    m.sL[blk_off(bi, bj) + bidx(tid >> 4, tid & 15)] = C[(size_t)(bi * SB + (tid >> 4)) * ld + bj * SB + (tid & 15)];
        }
        __syncthreads();
        const DiagPub pub = {C, ld, Wg, prog, pub_early};
        diag128_factor_invert(m.sL, m.sW, m.sT, m.sRd, m.sCol, (k + 1) * NB, n_real, fail, nullptr, &pub);
        return;
    }
    if (b <= nfol) {
        if (fol_rows == 128) {
            // one follower per block row (two strips per wave): half as many workgroups leave the trailing update; the
            // whole 128-row tile first, so this follower starts ~10 us later and catches up several columns per pass
            const int ii = b;
            if (k >= 0) {
                update_subtile<4>(K, ld, k, ii * (ii + 1) / 2, 0, smem);
                __syncthreads();
            }
            panel_follow<2>(K, ld, k + 1, (size_t)(k + 1 + ii) * NB, prog, Wg, smem, fail, n_real, poll_sleep);
            return;
        }
        const int ii = 1 + (b - 1) / 2, h = (b - 1) & 1;           // block row k + 1 + ii, rows 64 h .. 64 h + 63
        if (k >= 0) {
            update_subtile<2>(K, ld, k, ii * (ii + 1) / 2, h, smem);
            __syncthreads();             // this workgroup's stores before its own strip loads (and the staging area's reuse)
        }
        panel_follow(K, ld, k + 1, (size_t)(k + 1 + ii) * NB + (size_t)h * 64, prog, Wg, smem, fail, n_real, poll_sleep);
        return;
    }
    // the remaining tiles of the trailing update: the triangle without its first block row and column
    const int q = b - 1 - nfol, W = (int)gridDim.x - 1 - nfol;
    int full_end = ntiles, split = 0, left = 0;
    if (tail_split != 0 && ntiles > W) {
        const int R = ntiles / W;
        left = ntiles - R * W;
        if (left > 0) split = 4 * left <= W ? 4 : (2 * left <= W ? 2 : 0);
        if (split) full_end = R * W;
    }
    if (q < full_end) update_tiles_persistent(K, ld, k, q, W, full_end, smem, 1);
    if (split && q < split * left) {
        __syncthreads();
        if (split == 4) update_subtile<1>(K, ld, k, full_end + q / 4, q % 4, smem, 1);
        else update_subtile<2>(K, ld, k, full_end + q / 2, q % 2, smem, 1);
    }
}

// Batched fits: diagonal block k and panel k of EVERY sample in one launch (potrf_batch_follow).  Grid (samples, 1 + block
// rows below k): the x index is the sample, so the S diagonal workgroups (y = 0) are dispatched before any follower, and a
// follower (y >= 1: all 128 rows of block row k + y, two strips per wave) waits for the diagonal workgroup of ITS sample only.
// Replaces potrf_diag_kernel + potrf_panel_kernel of the launch-per-phase form: the panel's ~12-40 us per step (S samples'
// strips behind a launch boundary) shrink to the hand-off behind the last pivot plus the followers that did not fit on the
// chip beside the diagonal workgroups (every workgroup carries the diagonal block's LDS image: one per CU) and run after them
// at the panel kernel's speed.  Same strips, bit for bit.
constexpr int ROLL_SMEM_DOUBLES = NBLK * BLK + 2 * BLK + NB + 8 * SB;   // L image + two W slots + counters + column exchange: 80 KB
template <bool ROLL>
__global__ __launch_bounds__(256, ROLL ? 2 : 1) void potrf_diag_follow_kernel(double* __restrict__ K, size_t k_stride, int ld, int k,
                                                                int n_real, double* __restrict__ Linv, size_t linv_stride,
                                                                int* __restrict__ fail, unsigned* __restrict__ prog,
                                                                int pub_early) {
    __shared__ double smem[ROLL ? ROLL_SMEM_DOUBLES : DIAG_SMEM_DOUBLES];
    const int smp = (int)blockIdx.x, role = (int)blockIdx.y;
    K += (size_t)smp * k_stride;
    Linv += (size_t)smp * linv_stride;
    fail += smp;
    prog += (size_t)smp * PROG_STRIDE + k;
    const size_t d0 = (size_t)k * NB;
    double* Wg = Linv + (size_t)k * NB * NB;
    if (role == 0) {
        double* C = K + d0 * ld + d0;
        DiagSmem m = diag_carve(smem);
        if (ROLL) {                       // sL | two W slots | counters | column exchange (no transposition scratch, no W image)
            m.sT = nullptr;
            m.sRd = m.sW + 2 * BLK;
            m.sCol = m.sRd + NB;
        }
        const int tid = threadIdx.x;
        for (int bi = 0; bi < NSB; ++bi)
            for (int bj = 0; bj <= bi; ++bj)
                m.sL[blk_off(bi, bj) + bidx(tid >> 4, tid & 15)] = C[(size_t)(bi * SB + (tid >> 4)) * ld + bj * SB + (tid & 15)];
        __syncthreads();
        const DiagPub pub = {C, ld, Wg, prog, pub_early};
        diag128_factor_invert<ROLL>(m.sL, m.sW, m.sT, m.sRd, m.sCol, k * NB, n_real, fail, nullptr, &pub);
        return;
    }
    static_assert(NBLK * BLK + 8 <= ROLL_SMEM_DOUBLES, "the follower's operand image fits the rolling layout");
    panel_follow<2>(K, ld, k, (size_t)(k + role) * NB, prog, Wg, smem, fail, n_real);
}

// ---- two-block problems (128 <= N <= 254): the same, ONE launch per ensemble half-step (r06) -------------------------------
// A Bayesian-optimisation run spends most of its life at these sizes, and the launch-per-phase half-step there was nine
// launches (proposal + scaling, gram, two diagonal blocks, panel, column update, likelihood shares, finish, accept): 93 us at
// N = 200, a third of it boundaries and phases that leave the chip idle.  Here one workgroup per walker runs the whole chain
//     proposal -> K (ten 64 x 64 tiles, three at a time) -> 128 pivots -> panel (eight strips) -> tile update -> 128 pivots
//     -> likelihood terms -> accept test
// with block (0, 0) of K written straight into the factorisation's LDS image and block row 1 through the walker's matrix of
// the batch workspace (read back by the same workgroup: L2).  The pieces are the launch path's own device functions in its
// order -- diag128_factor_invert, potrf_panel_kernel's substitution, diag_tile_update, potrf_inverse_kernel's likelihood
// shares added block by block -- so the log-likelihoods are the launch path's bit for bit and the chain is the same chain.
// MEASURED SLOWER (r06i, 52 walkers, D = 16, us per half-step, this kernel / launch path): N = 150 114 / 84, 200 122 / 91,
// 254 130 / 99 -- same walkers, same accept decisions.  One CU per walker runs in sequence what the launch path spreads
// over the chip (ten K tiles, eight panel strips, the tile update: ~35 us of the 120), the two 128-pivot chains (45 us) are
// on the path either way, and at 512 threads the compiler spills 150 registers in the panel and the tile update.  The nine
// launch boundaries it removes are worth ~15 us.  Kept as an option (mcmc_block_step = 3) with its tests; NOT the default.
template <int KIND, int NG>
__global__ __launch_bounds__(256 * NG) void mcmc_block2_step_kernel(McmcState st, int start, int first, int h, int it,
                                                               const double* __restrict__ X, const double* __restrict__ y,
                                                               double* __restrict__ Kws, size_t k_stride) {
    __shared__ double smem[DIAG_SMEM_DOUBLES];
    __shared__ int sfail;
    const DiagSmem m = diag_carve(smem);
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, w = blockIdx.x, P = st.P, n = st.n;
    constexpr int LD = 2 * NB;
    double* Kw = Kws + (size_t)w * k_stride;                 // this walker's 256 x 256 matrix: block row 1 lives there
    double* sq = m.sW;
    double* sism = sq + MAX_DIM + 8;
    double* sz = sism + MAX_DIM;
    int* sflag = reinterpret_cast<int*>(sz + 1);
    double* sI = sz + 2 + grp * (2 * GD * GLD + 2 * GT);     // per group: sI, sJ, sN
    double* sJ = sI + GD * GLD;
    double* sN = sJ + GD * GLD;
    const bool ok = mcmc_block_proposal(st, start, first, h, it, w, sq, sism, sz, sflag);
    const FitSample sp = mcmc_fit_sample(st, sq, ok);
    const double z = *sz;
    double prior = 0.0;
    if (threadIdx.x == 0) {
        if (ok && st.prior_kind != 0) prior = prior_lnprob(st.prior_kind, sq, P, st.prior_par);
        if (!ok) prior = -__builtin_huge_val();
        sfail = 0;
    }
    const double q0 = tid < P ? sq[tid] : 0.0, q1 = tid + 256 < P ? sq[tid + 256] : 0.0;   // (group 0) thread p keeps q[p]
    // ---- K: the ten lower 64 x 64 tiles of the 4 x 4 grid, NG per round (the tile routine's barriers are workgroup-wide:
    // a group without a tile in the last round recomputes tile 9 and drops it); entries as gram_kernel writes them.
    // NG = 2 (512 threads): the register budget of the tile update and the panel (256 per thread; at 768 threads the
    // compiler spilled 261 of them) costs one more round of tiles than NG = 3 would take.
#pragma nounroll
    for (int round = 0; round < (10 + NG - 1) / NG; ++round) {
        const int t = round * NG + grp;
        int bi, bj;
        tri_tile(t < 10 ? t : 9, bi, bj);
        const int tx = tid & 15, ty = tid >> 4;
        double cov[4][4];
        pair_cov_dot<KIND>(sp.cov, X, (long long)bi * GT, (long long)bj * GT, sI, sJ, sN, cov, sism, (long long)n, tid);
        if (t < 10) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int gi = bi * GT + ty * 4 + a;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int gj = bj * GT + gram_col(tx, b);
                    double val;
                    if (gi < n && gj < n) {
                        val = cov[a][b];
                        if (gi == gj) val += sp.noise;
                    } else if (gi == gj) {
                        val = 1.0;
                    } else if (gi == n && gj < n) {
                        val = y[gj] - sp.mean_c;
                    } else if (gj == n && gi < n) {
                        val = y[gi] - sp.mean_c;
                    } else {
                        val = 0.0;
                    }
                    if (gi < NB) {
                        if ((gj >> 4) <= (gi >> 4)) m.sL[blk_off(gi >> 4, gj >> 4) + bidx(gi & 15, gj & 15)] = val;
                    } else {
                        Kw[(size_t)gi * LD + gj] = val;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (grp != 0) return;
    const int lane = tid & 63, wave = tid >> 6;
    // ---- diagonal block 0
    diag128_factor_invert(m.sL, m.sW, m.sT, m.sRd, m.sCol, 0, n, &sfail, nullptr);
    __syncthreads();
    // block 0's log-diagonal share (all 128 rows are training rows here), before the image is reused
    double lg = tid < NB ? log(m.sL[blk_off(tid >> 4, tid >> 4) + bidx(tid & 15, tid & 15)]) : 0.0;
    // ---- the panel kernel's operand image in the W image: strictly lower blocks of L_00 and, in the diagonal slots, the W_ss,
    // rows and columns permuted (potrf_panel_kernel)
    {
        const int r = tid >> 4, c = tid & 15, src = bidx(r, c), pos = bidx(pi16(r), pi16(c));
        double wv[NSB];
#pragma unroll
        for (int s2 = 0; s2 < NSB; ++s2) wv[s2] = m.sW[blk_off(s2, s2) + src];
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < NSB; ++s2) m.sW[blk_off(s2, s2) + pos] = wv[s2];
        for (int bi = 1; bi < NSB; ++bi)
            for (int bj = 0; bj < bi; ++bj) m.sW[blk_off(bi, bj) + pos] = m.sL[blk_off(bi, bj) + src];
    }
    __syncthreads();
    // ---- panel: block row 1's eight 16-row strips, two per wave one after the other (the panel kernel's substitution)
#pragma nounroll
    for (int half = 0; half < 2; ++half) {
        double* Arow = Kw + (size_t)(NB + (wave + 4 * half) * 16 + (lane & 15)) * LD + 4 * (lane >> 4);
        v4d yv[NSB];
#pragma unroll
        for (int s2 = 0; s2 < NSB; ++s2) {
            const double2* p = reinterpret_cast<const double2*>(Arow + s2 * SB);
            const double2 lo = p[0], hi = p[1];
            yv[s2] = v4d{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int s2 = 0; s2 < NSB; ++s2) {
            v4d t = yv[s2];
#pragma unroll
            for (int c = 0; c < s2; ++c) {
                const Frag4 a = frag_row(m.sW + blk_off(s2, c), lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) t = mfma_f64(-a.v[r], yv[c][r], t);
            }
            const Frag4 wf = frag_row(m.sW + blk_off(s2, s2), lane);
            v4d o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) o = mfma_f64(wf.v[r], t[r], o);
            yv[s2] = o;
        }
#pragma unroll
        for (int s2 = 0; s2 < NSB; ++s2) {
            double2* p = reinterpret_cast<double2*>(Arow + s2 * SB);
            p[0] = make_double2(yv[s2][0], yv[s2][1]);
            p[1] = make_double2(yv[s2][2], yv[s2][3]);
        }
    }
    __syncthreads();                          // the strips are in memory for this workgroup's own reads
    // ---- block 0's shares of (z.z, sum log L_ii): potrf_inverse_kernel's likelihood branch, its order
    double part_q[2], part_l[2];
    double* red = m.sT;                       // (the transposition scratch is idle between the factorisations)
    {
        double q = 0.0;
        if (tid < NB) {
            const double zi = Kw[(size_t)n * LD + tid];
            q = zi * zi;
        } else {
            lg = 0.0;
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
        part_q[0] = red[0] + red[1];
        part_l[0] = red[2] + red[3];
        __syncthreads();
    }
    // ---- tile (1, 1) <- C - P P^T into the LDS image, second diagonal block
    diag_tile_update(Kw + (size_t)NB * LD, Kw + (size_t)NB * LD + NB, LD, m.sW, m.sL);
    __syncthreads();
    diag128_factor_invert(m.sL, m.sW, m.sT, m.sRd, m.sCol, NB, n, &sfail, nullptr);
    __syncthreads();
    {
        double q = 0.0;
        lg = 0.0;
        const int r = NB + tid;
        if (tid < NB && r < n) {
            const int zr = n - NB;
            const double zi = m.sL[blk_off(zr >> 4, tid >> 4) + bidx(zr & 15, tid & 15)];
            q = zi * zi;
            lg = log(m.sL[blk_off(tid >> 4, tid >> 4) + bidx(tid & 15, tid & 15)]);
        }
        for (int o = 32; o > 0; o >>= 1) {
            q += __shfl_xor(q, o);
            lg += __shfl_xor(lg, o);
        }
        __syncthreads();
        if ((tid & 63) == 0 && tid < NB) {
            red[tid >> 6] = q;
            red[2 + (tid >> 6)] = lg;
        }
        __syncthreads();
        part_q[1] = red[0] + red[1];
        part_l[1] = red[2] + red[3];
    }
    // ---- accept test (mcmc_accept_kernel's, for this walker); the sums block by block as loglik_finish_kernel adds them
    const int half_k = st.k / 2, sw = start ? first + w : h * half_k + w;
    if (tid == 0) {
        double sq_ = 0.0, sl_ = 0.0;
        sq_ += part_q[0];
        sl_ += part_l[0];
        sq_ += part_q[1];
        sl_ += part_l[1];
        const double lp = mcmc_lnprob(prior, sfail, sq_, 2.0 * sl_, n);
        if (lp != lp) atomicOr(st.d_err, 1);
        int acc = 0;
        if (start) {
            if (lp == __builtin_huge_val()) atomicOr(st.d_err, 2);
            st.d_lnp[sw] = lp;
        } else {
            const size_t r = ((size_t)it * 2 + h) * half_k + w;
            const double lnpdiff = mcmc_lnpdiff(P, log(z), lp, st.d_lnp[sw]);
            if (lnpdiff > log(st.d_ua[r])) {
                acc = 1;
                st.d_lnp[sw] = lp;
                st.d_nacc[sw] += 1;
            }
            if (st.d_lnprob) st.d_lnprob[(size_t)sw * st.n_steps + it] = st.d_lnp[sw];
        }
        *sflag = acc;
    }
    __syncthreads();
    if (start) return;
    const bool acc = *sflag != 0;
    for (int p = tid, e = 0; p < P; p += 256, ++e) {
        double* pp = st.d_pos + (size_t)sw * P + p;
        const double v = acc ? (e == 0 ? q0 : q1) : *pp;
        if (acc) *pp = v;
        if (st.d_chain) st.d_chain[((size_t)sw * st.n_steps + it) * P + p] = v;
    }
}

int launch_mcmc_block2_step(robo_gp* gp, const McmcState& st, int start, int first, int h, int it, double* d_K, size_t k_stride) {
    const int ns = start ? st.ns_eval : st.k / 2;
#define ROBO_BLOCK2_STEP(KIND)                                                                                          \
    hipLaunchKernelGGL((mcmc_block2_step_kernel<KIND, 2>), dim3(ns), dim3(512), 0, gp->ctx->stream, st, start, first, h, it, \
                       (const double*)gp->d_X, (const double*)gp->d_y, d_K, k_stride)
    if (gp->kind == ROBO_KERNEL_MATERN52_ARD) ROBO_BLOCK2_STEP(ROBO_KERNEL_MATERN52_ARD);
    else ROBO_BLOCK2_STEP(ROBO_KERNEL_RBF_ARD);
#undef ROBO_BLOCK2_STEP
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// Linv blocks -> packed A-operand fragments for the transposed block-row solve (predict.hip, trsm_step_t_kernel):
// fragment s = wp_offset(cb) + 4 jb + kk of diagonal block b, lane l:
//     Linv_b[16 cb + pi16(l & 15)][16 jb + 4 kk + (l >> 4)]
// (pi16 on the row slot: register r of lane group g of the product is then row 16 cb + 4 g + r of the result --
// four consecutive rows per lane, stored as two 16-byte pieces).
__device__ __forceinline__ double linv_pack_entry(const double* __restrict__ W, int idx) {
    const int f = idx >> 6, l = idx & 63;
    int cb = 7;
    while (f >= wp_offset(cb) + 4 * (cb + 1)) --cb;
    const int rel = f - wp_offset(cb), jb = rel >> 2, kk = rel & 3;
    return W[(size_t)(16 * cb + pi16(l & 15)) * NB + 16 * jb + 4 * kk + (l >> 4)];
}

__global__ __launch_bounds__(256) void linv_pack_kernel(const double* __restrict__ Linv, double* __restrict__ Wp) {
    const double* W = Linv + (size_t)blockIdx.x * NB * NB;
    double* out = Wp + (size_t)blockIdx.x * WP_BLOCK;
    for (int idx = threadIdx.x; idx < WP_BLOCK; idx += 256) out[idx] = linv_pack_entry(W, idx);
}

// robo_gp_fit_batch: the factors of a batched pass into the S handles they belong to, in ONE launch (grid.y = sample).
// Per handle this was eight stream operations -- copies of K, the inverse blocks, the scaled inputs, the metrics, the
// sample record, X and y, plus the fragment-packing launch: 416 operations for the 52 hyper-parameter samples of a
// Bayesian-optimisation iteration, 1.8 ms where the batched fit itself takes 0.1 (r03zy).
__global__ __launch_bounds__(256) void batch_keep_kernel(const KeepDst* __restrict__ dst, const double* __restrict__ bK,
                                                         size_t k_stride, const double* __restrict__ bLinv,
                                                         size_t linv_stride, const double* __restrict__ bXs,
                                                         size_t xs_stride, const double* __restrict__ bism,
                                                         const FitSample* __restrict__ bsp,
                                                         const double* __restrict__ X0, const double* __restrict__ y0,
                                                         int n, int np, int D) {
    const int s = blockIdx.y;
    const KeepDst d = dst[s];
    if (!d.ok) return;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    const double2* srcK = reinterpret_cast<const double2*>(bK + (size_t)s * k_stride);
    double2* dstK = reinterpret_cast<double2*>(d.K);
    for (size_t i = t; i < (size_t)np * np / 2; i += nt) dstK[i] = srcK[i];
    const double* srcL = bLinv + (size_t)s * linv_stride;
    for (size_t i = t; i < (size_t)np * NB; i += nt) d.Linv[i] = srcL[i];
    for (size_t i = t; i < (size_t)(np / NB) * WP_BLOCK; i += nt) {
        const size_t b = i / WP_BLOCK;
        d.LinvP[i] = linv_pack_entry(srcL + b * NB * NB, (int)(i - b * WP_BLOCK));
    }
    const double* srcX = bXs + (size_t)s * xs_stride;
    for (size_t i = t; i < (size_t)np * D; i += nt) d.Xs[i] = srcX[i];
    for (size_t i = t; i < (size_t)D; i += nt) d.theta[i] = bism[(size_t)s * D + i];
    if (t == 0) *d.sp = bsp[s];
    if (d.X) {          // handles other than gps[0]: the training data itself
        for (size_t i = t; i < (size_t)n * D; i += nt) d.X[i] = X0[i];
        for (size_t i = t; i < (size_t)n; i += nt) d.y[i] = y0[i];
    }
}

int launch_batch_keep(robo_gp* g0, const KeepDst* d_dst, int ns) {
    const size_t np = (size_t)g0->n_pad;
    size_t bx = (np * np / 2 + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(batch_keep_kernel, dim3((unsigned)bx, (unsigned)ns), dim3(256), 0, g0->ctx->stream, d_dst,
                       (const double*)g0->d_bK, np * np, (const double*)g0->d_bLinv, np * NB, (const double*)g0->d_bXs,
                       np * g0->dim, (const double*)g0->d_bism, (const FitSample*)g0->d_bsp, (const double*)g0->d_X,
                       (const double*)g0->d_y, g0->n, g0->n_pad, g0->dim);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

int launch_pack_linv(robo_gp* gp) {
    hipLaunchKernelGGL(linv_pack_kernel, dim3(gp->n_pad / NB), dim3(256), 0, gp->ctx->stream, (const double*)gp->d_Linv,
                       gp->d_LinvP);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// the follower form of the step kernel needs all its workgroups resident (one per CU): the diagonal workgroup, two followers
// per block row below the panel (`rows_below` of them), and at least a quarter of the chip left for the other tiles
static bool tiles_ok_for_follow(int rows_below, int max_wg) { return 1 + 2 * rows_below + max_wg / 4 <= max_wg; }

int launch_potrf(robo_gp* gp, const FitBuffers& fb, bool with_gram) {
    robo_ctx* ctx = gp->ctx;
    const int ld = gp->n_pad, nb = gp->n_pad / NB, S = fb.S;
    // fb.fail[0 .. S) is zeroed by the gram kernel, which always precedes the factorisation of its samples
    const Tuning& tune = ctx->tune;
    // fused step kernels (update + next diagonal block + next panel per launch): single fits, and batches whose factors are
    // so small that the per-theta CHAIN, not the matrix pipe, is what a pass costs (potrf_fused_panels: up to this many panels)
    // -1: by the workgroups of the first step's launch (per sample: the diagonal workgroup, two followers per block row, one
    // workgroup per other tile) against the chip -- up to three rounds of CUs: 26 walkers up to six panels (N <= 766).  (A
    // follower only ever waits for the diagonal workgroup of ITS sample, which precedes it in dispatch order: workgroups that
    // do not find a CU at once simply start later.)  r06w, us per half-step, grouped -> fused: N = 500 204 -> 182, 640 339 -> 331,
    // 760 385 -> 372; N = 1000 (eight panels, 4.4 rounds) 573 -> 641: slower, stays on the grouped path.
    // Measured (r06t, device chain, 26 walkers per half-step, us per half-step, grouped batched path -> fused): N = 200 86.8 ->
    // 84.2, 300 137.8 -> 131.7, 380 150.3 -> 142.4, same walkers and accept decisions (the fused form saves the column-update
    // launch of every step; its 128-deep updates at one workgroup per CU do not matter at three panels).
    const int fused_wgs = 1 + 2 * (nb - 1) + (nb - 1) * nb / 2;
    const bool fused = tune.potrf_fused != 0 &&
                       (S <= 2 || (tune.potrf_fused_panels < 0 ? S * fused_wgs <= 3 * ctx->num_cu : nb <= tune.potrf_fused_panels));
    bool gram_done = !with_gram;
    // n a multiple of 128 (every BASELINE size): the last block holds the augmented row ALONE.  Its diagonal entry is never
    // read -- z = L^-1 (y - mean) is complete once the last REAL panel has passed over row n, the likelihood needs z.z and the
    // real pivots only -- so that block is neither updated nor factored nor inverted: one link less in the chain
    // panel -> tile update -> 128 pivots -> next panel (r04: the 33rd link of the N = 4096 fit)
    const int nbf = (gp->n % NB == 0 && nb > 1) ? nb - 1 : nb;       // diagonal blocks that are factored
    // every launch of the factorisation as a function of (stream, first sample, samples): the batched path may run
    // sub-batches on their own streams (below)
    auto diag = [&](hipStream_t st, int s0, int ns, int kk) {
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(ns), dim3(256), 0, st, fb.K + (size_t)s0 * fb.k_stride, fb.k_stride, ld, kk,
                           gp->n, fb.Linv + (size_t)s0 * fb.linv_stride, fb.linv_stride, fb.fail + s0, (long long*)nullptr,
                           (double*)nullptr, (double*)nullptr);
    };
    auto panel = [&](hipStream_t st, int s0, int ns, int kk) {
        hipLaunchKernelGGL(potrf_panel_kernel, dim3((nb - kk - 1) * 2, ns), dim3(256), 0, st,
                           fb.K + (size_t)s0 * fb.k_stride, fb.k_stride, ld, kk,
                           (const double*)(fb.Linv + (size_t)s0 * fb.linv_stride), fb.linv_stride, (long long*)nullptr);
    };
#define ROBO_STEP(TM, F, ST, S0, NS, GRID, BASE, KOP, DEPTH, FIRST, NT)                                              \
    hipLaunchKernelGGL((potrf_step_kernel<TM, F>), dim3((GRID), (NS)), dim3(256), 0, (ST),                            \
                       fb.K + (size_t)(S0)*fb.k_stride, fb.k_stride, ld, (BASE), gp->n,                               \
                       fb.Linv + (size_t)(S0)*fb.linv_stride, fb.linv_stride, fb.fail + (S0), (KOP), (DEPTH), (FIRST), (NT),  \
                       (F) ? -1 : thin_row)
    // tile height, measured (N = 4096, S = 1): 128-row tiles 43 us/step at 384..528 blocks, 64-row tiles
    // slower (55 us: B panel re-read twice), 32-row tiles 16 us vs 21 us once blocks < 96
    const int thin_row = (nbf < nb && tune.potrf_thin_last != 0) ? nb - 1 : -1;   // the augmented row's own block (r05)
    auto update = [&](hipStream_t st, int s0, int ns, int tiles, int base, int kop, int depth, int first) {
        if (tiles * ns >= tune.potrf_batch_tm4_min) ROBO_STEP(4, false, st, s0, ns, tiles, base, kop, depth, first, 0);
        else if (tiles > 0) ROBO_STEP(1, false, st, s0, ns, tiles * 4, base, kop, depth, first, 0);
    };
    if (nb == 1 && !fb.want_inverse) {
        // one block, likelihood only: factor and reduce in one launch
        if (!gram_done) ROBO_TRY(launch_gram(gp, fb));
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(S), dim3(256), 0, ctx->stream, fb.K, fb.k_stride, ld, 0, gp->n, fb.Linv,
                           fb.linv_stride, fb.fail, (long long*)nullptr, fb.out, fb.host_out);
        ROBO_LAUNCH_CHECK();
        return ROBO_OK;
    }
    // test knobs (tests/: the emulator reaches the persistent multi-tile path at small N through them)
    const int tm4_min = tune.potrf_tm4_min, wg_cap = tune.potrf_max_wg;
    const int max_wg = wg_cap >= 2 ? wg_cap : ctx->num_cu;
    if (fused) {
        // single theta: trailing update of step k + diagonal block k+1 (workgroup 0) in one launch
        hipStream_t st = ctx->stream;
        if (!gram_done) ROBO_TRY(launch_gram(gp, fb));
        // potrf_follow: from step `ffrom` on the panel solve of column k+1 runs INSIDE step k's launch, following the
        // diagonal workgroup (potrf_step_follow_kernel); ffrom = -1: the first diagonal block and panel 0 too.  Needs every
        // workgroup resident: 1 + two followers per block row + at least a quarter of the chip for the other tiles.
        // Measured (r06, ms per fit, launch-per-phase -> followers from the first panel on): N = 4096 1.70 -> 1.52, 3000 1.22 ->
        // 1.13, 2048 0.83 -> 0.72, 1024 0.43 -> 0.37, 512 0.227 -> 0.214.  The followers are workgroups the trailing update does
        // not have: while a step is bound by its tiles (early steps of a large factor) they cost more than the panel launch
        // they replace -- N = 8192 (65 panels): followers from step 0 7.28, from step 16 6.11, from 24 / 32 5.88 against 6.06
        // without; with one 128-row follower per block row in the tile-bound steps (potrf_follow_rows = -1, below): from step
        // 0 6.08, 8 5.88, 16 5.80, 31 5.85.  potrf_follow_from = -2 (default): k >= nb - 2 - 3 max_wg / 16 (every step up to
        // N = 6144, step 15 at N = 8192).
        const bool can_follow = tune.potrf_follow != 0 && fb.prog != nullptr && max_wg >= 16 && nb <= PROG_STRIDE;
        int ffrom = nb;
        if (can_follow) {
            ffrom = tune.potrf_follow_from >= -1 ? tune.potrf_follow_from : nb - 2 - 3 * max_wg / 16;
            if (ffrom < -1) ffrom = -1;
            while (ffrom < nb && !tiles_ok_for_follow(nb - 1 - ffrom, max_wg)) ++ffrom;   // (explicit settings: residency)
        }
        auto follow = [&](int k) {
            // block rows below k+1: two followers each; the other tiles of the trailing triangle (none in front of panel 0)
            const int r1 = nb - k - 2, rest = k < 0 ? 0 : r1 * (r1 + 1) / 2;
            // followers per block row: two (64 rows each) or one (128 rows, potrf_follow_rows = 128; -1 = by step: one while
            // the other tiles need more than one round of the workgroups that two per row would leave)
            // Measured (r06k, N = 4096, us per step, two / one per row): step 0 70.2 / 60.9, 1 63.1 / 59.3, 2 57.2 / 52.4,
            // 3 56.8 / 51.8, 4 49.8 / 51.1, 8 45.6 / 50.8 -- one per row (its 128-row tile takes 24 us before it can follow)
            // pays while the other tiles need more than two rounds of the workgroups left
            int frows = tune.potrf_follow_rows == 128 ? 128 : 64;
            if (tune.potrf_follow_rows < 0 && rest > 2 * (max_wg - 1 - 2 * r1)) frows = 128;
            const int nfol = frows == 128 ? r1 : 2 * r1;
            int W = max_wg - 1 - nfol;
            W = W < 1 ? 1 : W;
            W = W < rest ? W : rest;
            hipLaunchKernelGGL(potrf_step_follow_kernel, dim3(1 + nfol + W, S), dim3(256), 0, st, fb.K, fb.k_stride,
                               ld, k, gp->n, fb.Linv, fb.linv_stride, fb.fail, fb.prog, nfol, rest,
                               tune.potrf_tail_split != 0 ? 1 : 0, tune.potrf_pub_early, frows,
                               tune.potrf_poll_sleep < 1 ? 1 : tune.potrf_poll_sleep);
        };

        bool panel_done = false;              // panel of the CURRENT column k already solved (by the previous follow step)
        if (ffrom < 0 && nbf >= 1 && nb > 1) {
            follow(-1);
            panel_done = true;
        } else {
            diag(st, 0, S, 0);
        }
        for (int k = 0; k + 1 < nb; ++k) {
            const int rem = nb - k - 1, tiles = rem * (rem + 1) / 2;
            if (!panel_done) panel(st, 0, S, k);
            panel_done = false;
            if (k + 1 >= nbf) break;          // what is left is the augmented row's own block: nothing to factor
            if (k >= ffrom) {
                follow(k);                    // update by panel k, diagonal block k+1, panel k+1
                panel_done = true;
                continue;
            }
            // 128-row tiles: one diagonal workgroup + at most (CUs - 1) persistent tile workgroups (one per CU: the
            // diagonal block's LDS image sizes every workgroup of the launch)
            if (tiles * S >= tm4_min)
                ROBO_STEP(4, true, st, 0, S, tiles < max_wg ? tiles : max_wg, k, k, NB, tune.potrf_tail_split != 0 ? 1 : 0,
                          tiles);
            else ROBO_STEP(1, true, st, 0, S, 1 + (tiles - 1) * 4, k, k, NB, 0, tiles);
        }
    } else {
        // batched thetas: the chip is full, and a 128-deep update is bound by reading and writing its C
        // tile (16 flop per byte), not by the MFMA pipe.  Panels are therefore taken in groups of G:
        // inside a group a block column only receives the group's earlier panels right before it is
        // factored (left-looking, one update of growing depth), and the rest of the trailing matrix
        // receives all G panels in ONE (128 G)-deep update -- 1/G of the C traffic.  Bitwise the same
        // factor: every element still accumulates its products in ascending k on top of the stored value.
        // measured in round 1, 27 thetas at N = 4096 on one stream (ms per theta): G=1 0.668, 2 0.596, 4 0.574, 6 0.566
        // potrf_group = 0 (default): by size -- measured with three streams (r05h, 26-27 thetas, ms per theta for groups of
        // 2 / 3 / 4 / 5): N = 1024 0.0290 / 0.0289 / 0.0300 / 0.0293, 1536 0.0535 / 0.0529 / 0.0538 / 0.0546, 2048 0.0946 /
        // 0.0927 / 0.0954 / 0.0942, 3072 0.2393 / 0.2307 / 0.2320 / 0.2322, 4096 0.4918 / 0.4791 / 0.4761 / 0.4708
        const int G = tune.potrf_group < 1 ? (nb <= 25 ? 3 : 5) : (tune.potrf_group > 16 ? 16 : tune.potrf_group);
        // `lead` = size of the FIRST group (1..G); all later groups hold G panels.  Where the group boundaries fall
        // changes which launches carry which products, never the order in which an element accumulates them.
        // potrf_batch_follow: diagonal block and panel of a step in ONE launch, the panel following the diagonal workgroups
        // through progress words (potrf_diag_follow_kernel); needs the batch's progress words (zeroed in front of the fork)
        // Measured (r06j, 26-27 thetas, three streams, ms per theta, launch-per-phase -> merged): N = 1024 0.0289 -> 0.0262,
        // 2048 0.0928 -> 0.0902, 3072 0.2288 -> 0.2324, 4096 0.4682 -> 0.4787 -- every workgroup of the merged launch carries
        // the diagonal block's 150-KB LDS image, so at large N the followers (one CU each) crowd out the other sub-batches'
        // update workgroups.  potrf_batch_follow = -1 (default): up to 17 panels (N <= 2048); 0 / 1: never / always.
        const bool bfollow = fb.prog != nullptr && nb <= PROG_STRIDE &&
                             (tune.potrf_batch_follow < 0 ? nb <= 17 : tune.potrf_batch_follow != 0);

        auto group = [&](hipStream_t st, int s0, int ns, int k0, int g) {
            for (int kk = k0; kk < k0 + g; ++kk) {
                if (kk >= nbf) break;         // the augmented row's own block
                // left-looking inside the group: block column kk <- panels k0 .. kk-1, all rows >= kk
                if (kk > k0) update(st, s0, ns, nb - kk, kk - 1, k0, (kk - k0) * NB, 1);
                if (bfollow && kk + 1 < nb) {
                    if (tune.potrf_batch_roll != 0)
                        hipLaunchKernelGGL(potrf_diag_follow_kernel<true>, dim3(ns, nb - kk), dim3(256), 0, st,
                                           fb.K + (size_t)s0 * fb.k_stride, fb.k_stride, ld, kk, gp->n,
                                           fb.Linv + (size_t)s0 * fb.linv_stride, fb.linv_stride, fb.fail + s0,
                                           fb.prog + (size_t)s0 * PROG_STRIDE, tune.potrf_pub_early);
                    else
                        hipLaunchKernelGGL(potrf_diag_follow_kernel<false>, dim3(ns, nb - kk), dim3(256), 0, st,
                                           fb.K + (size_t)s0 * fb.k_stride, fb.k_stride, ld, kk, gp->n,
                                           fb.Linv + (size_t)s0 * fb.linv_stride, fb.linv_stride, fb.fail + s0,
                                           fb.prog + (size_t)s0 * PROG_STRIDE, tune.potrf_pub_early);
                    continue;
                }
                diag(st, s0, ns, kk);
                if (kk + 1 < nb) panel(st, s0, ns, kk);
            }
            const int rem = nb - (k0 + g);                    // block rows/columns beyond the group
            if (rem > 0) update(st, s0, ns, rem * (rem + 1) / 2, k0 + g - 1, k0, g * NB, 0);
        };
        // Sub-batches on their own streams (potrf_split, r05): the latency-bound phases of a group (diagonal blocks on
        // S of 256 CUs, panels, the left-looking column updates) of one sub-batch run beside the chip-wide trailing
        // update of another IF their group boundaries do not coincide -- hence a different FIRST group per sub-batch
        // (sub-batch j leads with G - j G / splits panels; all later groups hold G).  Where the group boundaries fall
        // changes which launch carries which products, never the order in which an element accumulates them: same bits.
        // The host hands the groups out round-robin so that no stream's queue runs dry behind another's launches.
        int splits = tune.potrf_split < 1 ? 1 : (tune.potrf_split > ROBO_AUX_STREAMS + 1 ? ROBO_AUX_STREAMS + 1 : tune.potrf_split);
        // measured (r05b, 27 thetas, ms per theta, one stream -> three): N = 4096 0.512 -> 0.470, N = 2048 0.104 -> 0.092,
        // N = 1024 0.0293 -> 0.0316 (nine panels: the chains are too short to interleave); four streams: 0.58 at N = 4096
        if (S < 2 * splits || nb < tune.potrf_split_min) splits = 1;
        const bool gram_per_stream = !gram_done && splits > 1 && tune.potrf_gram_split != 0;
        // one gram launch for the whole batch goes IN FRONT of the fork: the side streams must not start before it is done
        if (!gram_done && !gram_per_stream) ROBO_TRY(launch_gram(gp, fb));
        if (splits > 1) {
            ROBO_TRY(ctx_aux_streams(ctx));
            ROBO_HIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
            for (int j = 1; j < splits; ++j) ROBO_HIP_CHECK(hipStreamWaitEvent(ctx->aux[j - 1], ctx->ev_fork, 0));
        }
        if (gram_per_stream) {
            // potrf_gram_split = 1: K of every sub-batch on ITS stream (the first sub-batch starts factoring after a third of
            // the covariance work, the others' fp64-VALU-bound gram kernels run beside its matrix-pipe phases).  MEASURED
            // (r05d, 27 thetas): no gain -- N = 4096 0.4711 vs 0.4700 ms per theta, N = 2048 0.0934 vs 0.0923 -- so the
            // default stays one gram launch for the whole batch
            for (int j = 0; j < splits; ++j) {
                const int s0 = (int)((long long)S * j / splits), s1 = (int)((long long)S * (j + 1) / splits);
                ROBO_TRY(launch_gram(gp, fb, j == 0 ? ctx->stream : ctx->aux[j - 1], s0, s1 - s0));
            }
        }
        int next_k0[ROBO_AUX_STREAMS + 1] = {0, 0, 0, 0};
        for (bool more = true; more;) {
            more = false;
            for (int j = 0; j < splits; ++j) {
                const int k0 = next_k0[j];
                if (k0 >= nb) continue;
                const int s0 = (int)((long long)S * j / splits), s1 = (int)((long long)S * (j + 1) / splits);
                int want = G;
                if (k0 == 0 && j > 0) {
                    want = tune.potrf_lead > 0 ? G - tune.potrf_lead * j : G - (G * j) / splits;
                    if (want < 1) want = 1;
                }
                const int g = nb - k0 < want ? nb - k0 : want;
                group(j == 0 ? ctx->stream : ctx->aux[j - 1], s0, s1 - s0, k0, g);
                next_k0[j] = k0 + g;
                more = more || next_k0[j] < nb;
            }
        }
        for (int j = 1; j < splits; ++j) {
            ROBO_HIP_CHECK(hipEventRecord(ctx->ev_join[j - 1], ctx->aux[j - 1]));
            ROBO_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_join[j - 1], 0));
        }
    }
#undef ROBO_STEP
    if (fb.skip_tail && !fb.want_inverse) return ROBO_OK;       // (the device chain's own tail kernel follows)
    // the explicit 128 x 128 inverses of all diagonal blocks, off the factorisation's critical path
    hipLaunchKernelGGL(potrf_inverse_kernel, dim3(nbf, S), dim3(256), 0, ctx->stream, (const double*)fb.K, fb.k_stride, ld,
                       fb.Linv, fb.linv_stride, gp->n, fb.LinvP, fb.ll_part, fb.want_inverse ? 1 : 0);
    hipLaunchKernelGGL(loglik_finish_kernel, dim3(S), dim3(64), 0, ctx->stream, (const double*)fb.ll_part, nbf, fb.out,
                       (const int*)fb.fail, fb.host_out);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

// one instrumented diagonal-block kernel on panel 0 of the current gram matrix
int launch_diag_timeline(robo_gp* gp, long long* d_stamps) {
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, gp->ctx->stream, gp->d_K, (size_t)0, gp->n_pad, 0,
                       gp->n, gp->d_Linv, (size_t)0, gp->ctx->d_fail, d_stamps, (double*)nullptr, (double*)nullptr);
    // ... and the panel solve below it (stamps 16..19: start, operands in LDS, chain done, stored)
    const int nb = gp->n_pad / NB;
    if (nb > 1)
        hipLaunchKernelGGL(potrf_panel_kernel, dim3((nb - 1) * 2, 1), dim3(256), 0, gp->ctx->stream, gp->d_K, (size_t)0,
                           gp->n_pad, 0, (const double*)gp->d_Linv, (size_t)0, d_stamps + 16);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
