// The 64 x 64 tile of the fp64 gram matrix (K1), shared by gram.hip and the fused ensemble step of small problems (potrf.hip).
#pragma once
#include "common.h"
#include "kern_math.h"

namespace robo {

constexpr int GT = 64;      // tile edge
constexpr int GD = 16;      // dims per LDS pass
constexpr int GLD = GT + 2; // LDS leading dimension (doubles)

// Column of the 64-wide tile that micro-tile entry b (0..3) of thread column tx (0..15) holds in pair_cov_dot's tiles.
// ROBO_GRAM_COLMAP = 1: {2 tx, 2 tx + 1, 32 + 2 tx, 33 + 2 tx} -- the sixteen lanes of a row then write 256 CONTIGUOUS bytes
// per 16-byte store instruction (two whole 128-byte lines) instead of sixteen 16-byte pieces at a 32-byte stride (every line
// of the row half-written by each of the two instructions); 0: the plain {4 tx .. 4 tx + 3} (A/B build).  Same entries, same
// arithmetic per entry.
#ifndef ROBO_GRAM_COLMAP
#define ROBO_GRAM_COLMAP 1
#endif
__host__ __device__ constexpr int gram_col(int tx, int b) {
#if ROBO_GRAM_COLMAP == 1
    return (b >> 1) * 32 + 2 * tx + (b & 1);
#else
    return 4 * tx + b;
#endif
}

// fp64 stationary kernels in K1: squared distances through  r2 = |x_i|^2 + |x_j|^2 - 2 x_i . x_j  with the row norms
// reduced once per tile from the staged coordinates -- one FMA per pair and dimension instead of a subtraction and an
// FMA.  K1 is bound by fp64 VALU issue (r02i PMC), and at D = 16 the 32 distance instructions were the largest
// block left after the sqrt/exp trimming.  The expansion's cancellation error is ~|x|^2 eps <= 1e-15 ABSOLUTE in r2
// (scaled coordinates, |x|^2 = O(1)), i.e. below one ulp of the O(1) covariance it feeds; r2 is clamped at 0 and the
// diagonal entries are exact by construction (gram_kernel sets r2 = 0 there).  The cross-gram kernels (posterior)
// and the fp32 K-build keep direct differences.
// scale == nullptr: X holds coordinates already scaled by the inverse square-root metrics (scale_inputs_kernel).
// scale != nullptr: X holds the RAW coordinates of rows_real rows and the scaling happens while staging -- the same
// product, the same replication of row 0 into the pad rows, as scale_inputs_kernel (bit-identical tile).
template <int KIND>
__device__ __forceinline__ void pair_cov_dot(const CovParams& cp, const double* __restrict__ X, long long i0,
                                             long long j0, double* sI, double* sJ, double* sN, double (&cov)[4][4],
                                             const double* scale = nullptr, long long rows_real = 0, int tg = -1) {
    // tg >= 0: index of this thread within the 256-thread group that owns the tile (several groups per workgroup, each
    // with its own staging buffers; the barriers below are workgroup-wide, so all groups make the same calls)
    const int t = tg >= 0 ? tg : (int)threadIdx.x, tx = t & 15, ty = t >> 4, dim = cp.dim;
    double dot[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) dot[a][b] = 0.0;
    double nrm = 0.0;                       // threads 0..63: |x_i|^2 of row i0 + t; 64..127: |x_j|^2 of row j0 + t - 64
    const double* sMine = t < GT ? sI : sJ;
    for (int d0 = 0; d0 < dim; d0 += GD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + e * 256;
            const int row = idx >> 4, d = idx & 15;
            const bool ok = d0 + d < dim;
            if (scale) {
                const long long ri = i0 + row < rows_real ? i0 + row : 0, rj = j0 + row < rows_real ? j0 + row : 0;
                sI[d * GLD + row] = (ok && rows_real > 0) ? X[ri * dim + d0 + d] * scale[d0 + d] : 0.0;
                sJ[d * GLD + row] = (ok && rows_real > 0) ? X[rj * dim + d0 + d] * scale[d0 + d] : 0.0;
            } else {
                sI[d * GLD + row] = ok ? X[(i0 + row) * dim + d0 + d] : 0.0;
                sJ[d * GLD + row] = ok ? X[(j0 + row) * dim + d0 + d] : 0.0;
            }
        }
        __syncthreads();
        const int dn = dim - d0 < GD ? dim - d0 : GD;
        if (t < 2 * GT)
            for (int d = 0; d < dn; ++d) {
                const double x = sMine[d * GLD + (t & (GT - 1))];
                nrm = fma(x, x, nrm);
            }
        for (int d = 0; d < dn; ++d) {
            double xi[4], xj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                xi[a] = sI[d * GLD + ty * 4 + a];
                xj[a] = sJ[d * GLD + gram_col(tx, a)];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) dot[a][b] = fma(xi[a], xj[b], dot[a][b]);
        }
        __syncthreads();
    }
    if (t < 2 * GT) sN[t] = nrm;
    __syncthreads();
    double ni[4], nj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        ni[a] = sN[ty * 4 + a];
        nj[a] = sN[GT + gram_col(tx, a)];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double r2 = fma(-2.0, dot[a][b], ni[a] + nj[b]);
            r2 = r2 > 0.0 ? r2 : 0.0;
            if (i0 + ty * 4 + a == j0 + gram_col(tx, b)) r2 = 0.0;     // the diagonal is exact
            cov[a][b] = cov_finish<double, KIND>(cp, r2, 0.0);
        }
}

}  // namespace robo
