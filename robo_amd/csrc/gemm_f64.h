// fp64 MFMA "NT" GEMM core:  acc(BM x 128) += A[0:BM, k0:k1] * B[0:128, k0:k1]^T,  BM = 32*TM
//
// One workgroup (4 wave64, 2x2) owns a BM x 128 output tile; each wave owns (16*TM) x 64 =
// TM x 4 v_mfma_f64_16x16x4_f64 tiles (16*TM accumulator f64 per lane).  Both operands are
// row-major with the contraction index contiguous ("NT"), which is the form every product on
// the hot path takes:
//     panel solve   A_ik * Linv_kk^T           (potrf.hip, TM = 1: many small tiles, latency)
//     trailing      A_ij -= P_i * P_j^T        (potrf.hip, TM = 2: tail balance)
//     TRSM update   T = K*_i - V * L_i^T       (predict.hip, TM = 4: max operand reuse)
//     TRSM solve    V_i = T * Linv_ii^T        (predict.hip)
//     covariance    K** - V * V^T              (predict.hip)
//
// Fragment maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3):
//     A operand: lane l holds A[i = l & 15][k = l >> 4]
//     B operand: lane l holds B[k = l >> 4][j = l & 15]   (here B^T row j, column k)
//     C/D:       reg r of lane l is  (row = (l >> 4) + 4 r, col = l & 15)
//
// LDS: operand tiles are staged [rows][16 k] with leading dimension 18 doubles, so a
// fragment read (ds_read_b64, bank = double index mod 32 per 32-lane half) touches
// row*18 + {k, k+1}: 32 distinct banks.  Two stages (global->regs prefetch of tile t+1
// overlaps the MFMAs of tile t; one barrier per k-tile).  The fp64 MFMA issues once per 64
// cycles per SIMD (measured), so (TM+4) ds_read_b64 per 4*TM MFMAs and four 16-byte global
// loads per k-tile leave the matrix pipe as the only busy resource.  Compiled with
// -mllvm -amdgpu-mfma-vgpr-form: accumulators live in VGPRs (no v_accvgpr copies), <=256
// registers for TM = 4, so two workgroups share a CU.
#pragma once
#include "common.h"

namespace robo {

constexpr int BK = 16;
constexpr int LDS_LD = BK + 2;
constexpr int STAGE_B = NB * LDS_LD;                       // doubles of the B operand per stage
template <int TM> constexpr int stage_a() { return 32 * TM * LDS_LD; }
template <int TM> constexpr int gemm_smem_doubles() { return 2 * (stage_a<TM>() + STAGE_B); }
constexpr int GEMM_SMEM_DOUBLES = 2 * (32 * 4 * LDS_LD + STAGE_B);   // TM = 4

template <int TM>
struct AccT {
    v4d t[TM][4];
};
typedef AccT<4> Acc;

template <int TM>
__device__ __forceinline__ void acc_zero(AccT<TM>& c) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c.t[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
}

// position of this lane's accumulator element (tm, tn, r) inside the BM x 128 tile
template <int TM = 4>
__device__ __forceinline__ int acc_row(int tm, int r) {
    const int lane = threadIdx.x & 63, wy = (threadIdx.x >> 6) >> 1;
    return wy * (16 * TM) + tm * 16 + (lane >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int tn) {
    const int lane = threadIdx.x & 63, wx = (threadIdx.x >> 6) & 1;
    return wx * 64 + tn * 16 + (lane & 15);
}

// rows x 16 doubles of a row-major operand -> registers: thread t takes row t/2, half t%2
// 64 bytes of one operand row in flight between global memory and LDS (plain scalars, passed by
// value: as an array passed by reference the compiler kept it in scratch memory once the kernel
// around the k-loop grew, r01k)
struct Tile4 {
    double2 a, b, c, d;
};

template <int ROWS>
__device__ __forceinline__ Tile4 tile_load_regs(const double* __restrict__ G, int ld, int k0) {
    const int row = threadIdx.x >> 1, kh = (threadIdx.x & 1) * 8;
    Tile4 t;
    if (ROWS != 128) t.a = t.b = t.c = t.d = make_double2(0.0, 0.0);   // rows beyond the tile: defined, unused
    if (ROWS == 128 || row < ROWS) {
        const double2* p = reinterpret_cast<const double2*>(G + (size_t)row * ld + k0 + kh);
        t.a = p[0];
        t.b = p[1];
        t.c = p[2];
        t.d = p[3];
    }
    return t;
}

template <int ROWS>
__device__ __forceinline__ void tile_store_lds(double* S, const Tile4 t) {
    const int row = threadIdx.x >> 1, kh = (threadIdx.x & 1) * 8;
    if (ROWS == 128 || row < ROWS) {
        double2* p = reinterpret_cast<double2*>(S + row * LDS_LD + kh);
        p[0] = t.a;
        p[1] = t.b;
        p[2] = t.c;
        p[3] = t.d;
    }
}

// s_setprio(1) around each MFMA cluster: with two independent workgroups per CU the wave that has
// its operands ready keeps the matrix pipe while the other issues LDS/global traffic
// (A/B on MI355X, r01m: 18.45 -> 18.05 ms per 65 536 candidates, +2.5 %)
#ifndef ROBO_SETPRIO
#define ROBO_SETPRIO 1
#endif
// PIPE: the fragments of k-step kk+1 are requested before the MFMAs of k-step kk are issued.  A wave issues in
// order, so without it every k-step ends in  ds_read x (TM+4) -> s_waitcnt -> 4 TM MFMAs  with the matrix pipe idle
// for one LDS round trip; a second workgroup on the CU fills that bubble, a workgroup that is alone on its CU (the
// fused Cholesky step: the diagonal block's LDS image sizes every workgroup) cannot.
#ifndef ROBO_GEMM_PIPE
#define ROBO_GEMM_PIPE 0
#endif
template <int TM, bool NEG, bool PIPE = false>
__device__ __forceinline__ void tile_mfma(const double* sA, const double* sB, AccT<TM>& acc) {
    constexpr bool SETPRIO = ROBO_SETPRIO != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const double* pa = sA + (wy * (16 * TM) + (lane & 15)) * LDS_LD + (lane >> 4);
    const double* pb = sB + (wx * 64 + (lane & 15)) * LDS_LD + (lane >> 4);
    if (PIPE) {
        double a[2][TM], b[2][4];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[0][t] = pa[t * 16 * LDS_LD];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[0][t] = pb[t * 16 * LDS_LD];
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int c = kk & 1, n = c ^ 1;
            if (kk + 1 < BK / 4) {
#pragma unroll
                for (int t = 0; t < TM; ++t) a[n][t] = pa[t * 16 * LDS_LD + (kk + 1) * 4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[n][t] = pb[t * 16 * LDS_LD + (kk + 1) * 4];
            }
            if (NEG) {
#pragma unroll
                for (int t = 0; t < TM; ++t) a[c][t] = -a[c][t];
            }
            if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc.t[tm][tn] = mfma_f64(a[c][tm], b[c][tn], acc.t[tm][tn]);
            if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
        return;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        double a[TM], b[4];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            a[t] = pa[t * 16 * LDS_LD + kk * 4];
            if (NEG) a[t] = -a[t];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = pb[t * 16 * LDS_LD + kk * 4];
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc.t[tm][tn] = mfma_f64(a[tm], b[tn], acc.t[tm][tn]);
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
}

// acc (+/-)= A[:, kbeg:kend] * B[:, kbeg:kend]^T.  A (32*TM rows) and B (128 rows) point at the
// first row of the operand panels; kbeg/kend are multiples of BK and uniform over the
// workgroup.  smem: gemm_smem_doubles<TM>() doubles, free for reuse on return (ends on a barrier).
template <int TM, bool NEG>
__device__ __forceinline__ void gemm_nt(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                        int kbeg, int kend, AccT<TM>& acc, double* smem) {
    constexpr int SA = stage_a<TM>(), ST = SA + STAGE_B;
    const int nk = (kend - kbeg) / BK;
    if (nk <= 0) return;
    Tile4 ra = tile_load_regs<32 * TM>(A, lda, kbeg);
    Tile4 rb = tile_load_regs<128>(B, ldb, kbeg);
    tile_store_lds<32 * TM>(smem, ra);
    tile_store_lds<128>(smem + SA, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        double* cur = smem + (kt & 1) * ST;
        double* nxt = smem + ((kt + 1) & 1) * ST;
        const bool more = kt + 1 < nk;
        if (more) {
            ra = tile_load_regs<32 * TM>(A, lda, kbeg + (kt + 1) * BK);
            rb = tile_load_regs<128>(B, ldb, kbeg + (kt + 1) * BK);
        }
        tile_mfma<TM, NEG, ROBO_GEMM_PIPE != 0>(cur, cur + SA, acc);
        if (more) {
            tile_store_lds<32 * TM>(nxt, ra);
            tile_store_lds<128>(nxt + SA, rb);
        }
        __syncthreads();
    }
}

template <bool NEG>
__device__ __forceinline__ void gemm_nt_128(const double* __restrict__ A, int lda, const double* __restrict__ B,
                                            int ldb, int kbeg, int kend, Acc& acc, double* smem) {
    gemm_nt<4, NEG>(A, lda, B, ldb, kbeg, kend, acc, smem);
}

}  // namespace robo
