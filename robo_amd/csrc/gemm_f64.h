// fp64 MFMA "NT" GEMM core:  acc(128x128) += A[0:128, k0:k1] * B[0:128, k0:k1]^T
//
// One workgroup (4 wave64, 2x2) owns a 128x128 output tile; each wave owns 64x64 =
// 4x4 v_mfma_f64_16x16x4_f64 tiles (64 accumulator f64 per lane).  Both operands are
// row-major with the contraction index contiguous ("NT"), which is the form every
// product on the hot path takes:
//     panel solve   A_ik * Linv_kk^T           (potrf.hip)
//     trailing      A_ij -= P_i * P_j^T        (potrf.hip)
//     TRSM update   T = K*_i - V * L_i^T       (predict.hip)
//     TRSM solve    V_i = T * Linv_ii^T        (predict.hip)
//     covariance    K** - V * V^T              (predict.hip)
//
// Fragment maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3):
//     A operand: lane l holds A[i = l & 15][k = l >> 4]
//     B operand: lane l holds B[k = l >> 4][j = l & 15]   (here B^T row j, column k)
//     C/D:       reg r of lane l is  (row = (l >> 4) + 4 r, col = l & 15)
//
// LDS: operand tiles are staged [128 rows][16 k] with leading dimension 18 doubles, so a
// fragment read (ds_read_b64, bank = double index mod 32 per 32-lane half) touches
// row*18 + {k, k+1}: 32 distinct banks.  Two stages (global->regs prefetch of tile t+1
// overlaps the MFMAs of tile t; one barrier per k-tile).  The fp64 MFMA issues at most
// once per 16+ cycles per SIMD, so eight ds_read_b64 per sixteen MFMAs and four 16-byte
// global loads per 64 MFMAs leave the matrix pipe as the only busy resource.
#pragma once
#include "common.h"

namespace robo {

constexpr int BK = 16;
constexpr int LDS_LD = BK + 2;
constexpr int STAGE = NB * LDS_LD;             // doubles per operand per stage
constexpr int GEMM_SMEM_DOUBLES = 4 * STAGE;   // [stage][A|B]

struct Acc {
    v4d t[4][4];
};

__device__ __forceinline__ void acc_zero(Acc& c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c.t[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
}

// position of this lane's accumulator element (tm, tn, r) inside the 128x128 tile
__device__ __forceinline__ int acc_row(int tm, int r) {
    const int lane = threadIdx.x & 63, wy = (threadIdx.x >> 6) >> 1;
    return wy * 64 + tm * 16 + (lane >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int tn) {
    const int lane = threadIdx.x & 63, wx = (threadIdx.x >> 6) & 1;
    return wx * 64 + tn * 16 + (lane & 15);
}

__device__ __forceinline__ void tile_load_regs(const double* __restrict__ G, int ld, int k0, double2 (&r)[4]) {
    const int row = threadIdx.x >> 1, kh = (threadIdx.x & 1) * 8;
    const double2* p = reinterpret_cast<const double2*>(G + (size_t)row * ld + k0 + kh);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = p[i];
}

__device__ __forceinline__ void tile_store_lds(double* S, const double2 (&r)[4]) {
    const int row = threadIdx.x >> 1, kh = (threadIdx.x & 1) * 8;
    double2* p = reinterpret_cast<double2*>(S + row * LDS_LD + kh);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = r[i];
}

template <bool NEG>
__device__ __forceinline__ void tile_mfma(const double* sA, const double* sB, Acc& acc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const double* pa = sA + (wy * 64 + (lane & 15)) * LDS_LD + (lane >> 4);
    const double* pb = sB + (wx * 64 + (lane & 15)) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = pa[t * 16 * LDS_LD + kk * 4];
            b[t] = pb[t * 16 * LDS_LD + kk * 4];
            if (NEG) a[t] = -a[t];
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc.t[tm][tn] = mfma_f64(a[tm], b[tn], acc.t[tm][tn]);
    }
}

// acc (+/-)= A[:, kbeg:kend] * B[:, kbeg:kend]^T.  A, B point at the first row of the
// 128-row operand panels; kbeg/kend are multiples of BK and uniform over the workgroup.
// smem: GEMM_SMEM_DOUBLES doubles, free for reuse on return (ends on a barrier).
template <bool NEG>
__device__ __forceinline__ void gemm_nt_128(const double* __restrict__ A, int lda, const double* __restrict__ B,
                                            int ldb, int kbeg, int kend, Acc& acc, double* smem) {
    const int nk = (kend - kbeg) / BK;
    if (nk <= 0) return;
    double2 ra[4], rb[4];
    tile_load_regs(A, lda, kbeg, ra);
    tile_load_regs(B, ldb, kbeg, rb);
    tile_store_lds(smem, ra);
    tile_store_lds(smem + STAGE, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        double* cur = smem + (kt & 1) * 2 * STAGE;
        double* nxt = smem + ((kt + 1) & 1) * 2 * STAGE;
        const bool more = kt + 1 < nk;
        if (more) {
            tile_load_regs(A, lda, kbeg + (kt + 1) * BK, ra);
            tile_load_regs(B, ldb, kbeg + (kt + 1) * BK, rb);
        }
        tile_mfma<NEG>(cur, cur + STAGE, acc);
        if (more) {
            tile_store_lds(nxt, ra);
            tile_store_lds(nxt + STAGE, rb);
        }
        __syncthreads();
    }
}

}  // namespace robo
