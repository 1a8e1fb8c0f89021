// Posterior of SMALL candidate batches through the explicit inverse factor W = L^-1.
//
// Replaces, for batches of at most a few thousand candidates, the same reference lines as predict.hip
// (george.GP.predict's  Kxs K^-1 Kxs^T  and  Kxs alpha,  robo/models/gaussian_process.py:280-286) -- what
// robo/maximizers/random_sampling.py:9,42 actually asks for is 500 candidates per call, the single-point maximisers
// ask for one, entropy search for 50 representer points.
//
// The block-row solve of predict.hip is a chain of n/128 dependent launches however few candidates there are
// (2.5 ms at N = 4096 for 1 .. 2048 candidates): forward substitution orders the block rows.  With W = L^-1
// (triinv_kernel of gradient.hip, once per factor, lazily) there is no order:
//       V = K_* W^T,     V[c][j] = sum_{k <= j} K_*[c][k] W[j][k],     q_c = |V_c|^2,  mu_c = V_c . z
// is ONE lower-triangular product.  Work units are (128 candidates) x (128 columns j) x (a chunk of KC 128-blocks of
// the contraction index k): the triangle makes block row j cost (j + 1) block products, so without the split over k
// a 500-candidate batch would be bound by the last block row of each candidate tile (32 products on one workgroup);
// with it the batch is ~320 units of at most 8 products, launched heaviest first.
//   winv_gemm_kernel     one unit: 128 x 128 x (128 KC) fp64 MFMA product (gemm_f64.h) -> its own tile of P
//   winv_reduce_kernel   per (candidate tile, block row): the unit tiles added IN CHUNK ORDER -> V tile (stored only
//                        for callers that consume V: entropy search, full covariance), |v|^2 and v.z of its 128 columns
//   winv_finish_kernel   per candidate: the block rows' partial sums added in block-row order -> q, mu
// The association of every V entry is therefore fixed by (n_pad, KC) alone: values do not depend on the workspace chunking or
// the launch order (bit-identical across them; they differ from the block-row solve's by rounding, ~1e-13 relative -- same
// tolerances against the oracle).  KC itself is picked from the handle's total batch (r04: the batch depth from eight
// candidate tiles on, half of it below -- the units of a small batch are its critical path), so values agree ACROSS
// batch-size classes to rounding, like across the other paths.
//
// Numerics: W carries a forward error of ~eps cond(L) (the block-row solve: eps cond of a 128-block), so the caller
// (api.hip) takes this path only while cond_inf(L) = |L|_inf |W|_inf -- measured here when W is built
// (winv_norm_kernel) -- stays below a bound and keeps the substitution otherwise.
#include <algorithm>
#include <vector>

#include "common.h"
#include "gemm_f64.h"

namespace robo {

__global__ __launch_bounds__(256, 2) void winv_gemm_kernel(const double* __restrict__ Ks, int ldk,
                                                           const double* __restrict__ W, int ldw,
                                                           const int4* __restrict__ units, int n_units,
                                                           double* __restrict__ P) {
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    const int ct = blockIdx.x;
    const int4 u = units[blockIdx.y];            // (block row j, first k-block, end k-block, canonical unit id)
    Acc acc;
    acc_zero(acc);
    gemm_nt_128<false>(Ks + (size_t)ct * NB * ldk, ldk, W + (size_t)u.x * NB * ldw, ldw, u.y * NB, u.z * NB, acc, smem);
    double* out = P + ((size_t)ct * n_units + u.w) * (NB * NB);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[acc_row(tm, r) * NB + acc_col(tn)] = acc.t[tm][tn][r];
}

// tile (candidate tile ct, block row j): wave w owns rows w, w + 4, ..; lane l the columns 2 l, 2 l + 1
template <bool STORE_V>
__global__ __launch_bounds__(256) void winv_reduce_kernel(const double* __restrict__ P, const int* __restrict__ prefix,
                                                          int n_units, const double* __restrict__ z, int n,
                                                          double* __restrict__ V, int ldv, double* __restrict__ qpart,
                                                          double* __restrict__ mupart, long long rows) {
    const int j = blockIdx.x, ct = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = prefix[j], nch = prefix[j + 1] - u0;
    const double* base = P + ((size_t)ct * n_units + u0) * (NB * NB) + 2 * lane;
    const int col = j * NB + 2 * lane;
    const bool ok0 = col < n, ok1 = col + 1 < n;
    const double z0 = ok0 ? z[col] : 0.0, z1 = ok1 ? z[col + 1] : 0.0;
    for (int i = 0; i < NB / 4; ++i) {
        const int row = wave + 4 * i;
        double2 s = *reinterpret_cast<const double2*>(base + row * NB);
        for (int ch = 1; ch < nch; ++ch) {
            const double2 t = *reinterpret_cast<const double2*>(base + (size_t)ch * (NB * NB) + row * NB);
            s.x += t.x;
            s.y += t.y;
        }
        s.x = ok0 ? s.x : 0.0;
        s.y = ok1 ? s.y : 0.0;
        const long long c = (long long)ct * NB + row;
        if (STORE_V) *reinterpret_cast<double2*>(V + (size_t)c * ldv + col) = s;
        double q = fma(s.y, s.y, s.x * s.x), m = fma(s.y, z1, s.x * z0);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            q += __shfl_xor(q, o);
            m += __shfl_xor(m, o);
        }
        if (lane == 0) {
            qpart[(size_t)j * rows + c] = q;
            mupart[(size_t)j * rows + c] = m;
        }
    }
}

// LARGE batches (enough (candidate tile, block row) pairs to fill the chip on their own): one workgroup per pair over the
// WHOLE contraction range, heaviest block rows first (the dispatcher hands the next workgroup to the first free slot:
// longest-processing-time scheduling, 0.97-0.99 balance from 4096 candidates up at N = 4096), and the block row's
// |v|^2 and v.z partial sums straight from the accumulators: no unit tiles in memory (0.33 ms of HBM traffic per 8192
// candidates at N = 4096), no chunk-reduction pass.  Every V entry is ONE accumulator chain in k order -- the association is
// fixed by n_pad alone; which of the two forms runs depends on the batch size (launch_predict_winv), as the switch to the
// block-row substitution does, so values agree across batch sizes to rounding (~1e-13 relative), not bit for bit.
template <bool STORE_V>
__global__ __launch_bounds__(256, 2) void winv_row_kernel(const double* __restrict__ Ks, int ldk,
                                                          const double* __restrict__ W, int ldw, int nbk,
                                                          const double* __restrict__ z, int n, double* __restrict__ V,
                                                          int ldv, double* __restrict__ qpart,
                                                          double* __restrict__ mupart, long long rows) {
    __shared__ double smem[GEMM_SMEM_DOUBLES];
    const int ct = blockIdx.x;
    const int j = nbk - 1 - (int)blockIdx.y;
    Acc acc;
    acc_zero(acc);
    gemm_nt_128<false>(Ks + (size_t)ct * NB * ldk, ldk, W + (size_t)j * NB * ldw, ldw, 0, (j + 1) * NB, acc, smem);
    const int lane = threadIdx.x & 63, wx = (threadIdx.x >> 6) & 1;
    double zc[4];
    bool okc[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int col = j * NB + acc_col(tn);
        okc[tn] = col < n;
        zc[tn] = okc[tn] ? z[col] : 0.0;
    }
    // this lane: 16 rows x 4 columns; a row's 128 columns live in 16 lanes (lane & 15) x 4 tn x 2 waves (wx)
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double q = 0.0, m = 0.0;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const double v = okc[tn] ? acc.t[tm][tn][r] : 0.0;
                if (STORE_V)
                    V[((size_t)ct * NB + acc_row(tm, r)) * ldv + (size_t)j * NB + acc_col(tn)] = v;
                q = fma(v, v, q);
                m = fma(v, zc[tn], m);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                q += __shfl_xor(q, o);
                m += __shfl_xor(m, o);
            }
            if ((lane & 15) == 0) {
                // [row][wx] pairs: the two column halves of a row are added below, wx = 0 first
                smem[(acc_row(tm, r) * 2 + wx) * 2] = q;
                smem[(acc_row(tm, r) * 2 + wx) * 2 + 1] = m;
            }
        }
    __syncthreads();
    if (threadIdx.x < NB) {
        const int row = threadIdx.x;
        const long long c = (long long)ct * NB + row;
        qpart[(size_t)j * rows + c] = smem[(row * 2) * 2] + smem[(row * 2 + 1) * 2];
        mupart[(size_t)j * rows + c] = smem[(row * 2) * 2 + 1] + smem[(row * 2 + 1) * 2 + 1];
    }
}

// A HANDFUL of candidates (the 1 x D calls of the reference's single-point maximisers, robo/maximizers/scipy_optimizer.py:44,
// differential_evolution.py:29, grid_search.py:60): V = K_* W^T is a matrix-vector product, bound by reading W once (67 MB at
// N = 4096) -- the 128-candidate tile of the GEMM forms spends 127/128 of its MFMAs on padding rows and still takes 0.18 ms.
// One workgroup per slab of GV_ROWS rows of W: thread t walks k = t, t + 256, ... <= the slab's last column with the slab's
// W entries (coalesced along k) and the MC candidates' K_* entries (L2-resident) in registers, GV_ROWS x MC partial sums per
// thread, one block reduction at the end; |v|^2 and v.z per slab, added up by winv_finish_kernel in slab order.
constexpr int GV_ROWS = 8;
template <int MC, bool STORE_V>
__global__ __launch_bounds__(256) void winv_gemv_kernel(const double* __restrict__ Ks, int ldk,
                                                        const double* __restrict__ W, int ldw,
                                                        const double* __restrict__ z, int n, double* __restrict__ V,
                                                        int ldv, double* __restrict__ qpart,
                                                        double* __restrict__ mupart, long long rows) {
    __shared__ double red[4][GV_ROWS * MC];
    // heaviest slabs (the last rows of W: longest contraction) first
    const int slab = (int)gridDim.x - 1 - (int)blockIdx.x;
    const int j0 = slab * GV_ROWS, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int kend = j0 >= n ? 0 : (j0 + GV_ROWS < n ? j0 + GV_ROWS : n);   // columns < kend (entries beyond a row's diagonal are 0)
    double acc[GV_ROWS][MC];
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
        for (int c = 0; c < MC; ++c) acc[r][c] = 0.0;
#pragma unroll 4
    for (int k = t; k < kend; k += 256) {
        double ks[MC], w[GV_ROWS];
#pragma unroll
        for (int c = 0; c < MC; ++c) ks[c] = Ks[(size_t)c * ldk + k];
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r) w[r] = (j0 + r < n && k <= j0 + r) ? W[(size_t)(j0 + r) * ldw + k] : 0.0;
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
            for (int c = 0; c < MC; ++c) acc[r][c] = fma(w[r], ks[c], acc[r][c]);
    }
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            double s = acc[r][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red[wave][r * MC + c] = s;
        }
    __syncthreads();
    // thread c < MC finishes candidate c: its GV_ROWS entries of V, their share of |v|^2 and v.z
    if (t < MC) {
        double q = 0.0, m = 0.0;
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r) {
            const int j = j0 + r;
            const double v = j < n ? (red[0][r * MC + t] + red[1][r * MC + t]) + (red[2][r * MC + t] + red[3][r * MC + t]) : 0.0;
            if (STORE_V) V[(size_t)t * ldv + j] = v;      // (0 for columns >= n)
            q = fma(v, v, q);
            m = fma(v, j < n ? z[j] : 0.0, m);
        }
        qpart[(size_t)slab * rows + t] = q;
        mupart[(size_t)slab * rows + t] = m;
    }
}

__global__ __launch_bounds__(256) void winv_finish_kernel(const double* __restrict__ qpart,
                                                          const double* __restrict__ mupart, int nbk, long long rows,
                                                          double* __restrict__ q, double* __restrict__ mu) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= rows) return;
    double a = 0.0, b = 0.0;
#pragma unroll 8      // (the partials were just written by other CUs: eight loads in flight instead of a chain of L2 misses)
    for (int j = 0; j < nbk; ++j) {
        a += qpart[(size_t)j * rows + c];
        b += mupart[(size_t)j * rows + c];
    }
    q[c] = a;
    mu[c] = b;
}

// |L|_inf and |W|_inf over the n training rows: one wave per row, max over rows through atomicMax on the bit pattern
// (non-negative doubles order like their bits)
__global__ __launch_bounds__(256) void winv_norm_kernel(const double* __restrict__ L, const double* __restrict__ W,
                                                        int ld, int n, unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* l = L + (size_t)row * ld;
    const double* w = W + (size_t)row * ld;
    double sl = 0.0, sw = 0.0;
    for (int j = lane; j <= row; j += 64) {
        sl += fabs(l[j]);
        sw += fabs(w[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sl += __shfl_xor(sl, o);
        sw += __shfl_xor(sw, o);
    }
    if (lane == 0) {
        // (a NaN anywhere in W makes its row sum NaN: the bit pattern of a NaN exceeds every finite one, the condition
        // number comes out NaN and the caller keeps the substitution)
        atomicMax(out, (unsigned long long)__double_as_longlong(sl));
        atomicMax(out + 1, (unsigned long long)__double_as_longlong(sw));
    }
}

// the matrix-vector form leaves n / 8 slab partials per candidate (512 at N = 4096): one WAVE per candidate adds them -- lane l the
// partials l, l + 64, ... in ascending order, then the butterfly -- instead of one thread walking all of them (a chain of
// dependent L2 misses: it cost more than the product itself)
__global__ __launch_bounds__(256) void winv_finish_few_kernel(const double* __restrict__ qpart,
                                                              const double* __restrict__ mupart, int nparts,
                                                              long long rows, int mc, double* __restrict__ q,
                                                              double* __restrict__ mu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < mc; c += 4) {
        double a = 0.0, b = 0.0;
        for (int p = lane; p < nparts; p += 64) {
            a += qpart[(size_t)p * rows + c];
            b += mupart[(size_t)p * rows + c];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            b += __shfl_xor(b, o);
        }
        if (lane == 0) {
            q[c] = a;
            mu[c] = b;
        }
    }
    for (long long c = mc + threadIdx.x; c < rows; c += 256) {       // padding candidates: defined values
        q[c] = 0.0;
        mu[c] = 0.0;
    }
}

// contraction blocks per unit: fixed by the number of block rows of the factor and the depth variant v (0: batches of eight
// candidate tiles and more; 1, 2: half / a quarter of that for smaller batches, whose critical path is one unit)
static int winv_kc(int nbk, int v) {
    const int kc = (nbk >= 16 ? 8 : (nbk >= 8 ? 4 : 2)) >> v;
    return kc < 1 ? 1 : kc;
}

// The asynchronous half of the W build: buffers and unit table for this factor's number of block rows, the triinv launches,
// the two row-sum reductions and the copy of their results into pinned memory -- nothing waits for the device (apart from
// the first-use staging of the unit table).  robo_gp_prefetch_inverse calls it right after a fit, so that the build runs
// while the host prepares its candidates; winv_ensure finishes it.
int winv_launch(robo_gp* gp) {
    hipStream_t st = gp->ctx->stream;
    const size_t np = (size_t)gp->n_pad_max;
    if (!gp->d_Winv) {
        ROBO_HIP_CHECK(hipMalloc((void**)&gp->d_Winv, np * np * sizeof(double)));
        gp->winv_gen = 0;
    }
    if (!gp->d_gV) ROBO_HIP_CHECK(hipMalloc((void**)&gp->d_gV, np * np * sizeof(double)));   // W^T, scratch of the build
    const int nbk = (gp->n + NB - 1) / NB;
    if (gp->winv_nbk != nbk) {
        gp->winv_nbk = 0;
        for (int v = 0; v < 3; ++v) {
            const int kc = winv_kc(nbk, v);
            std::vector<int> prefix(nbk + 1, 0);
            std::vector<int4> units;
            for (int j = 0; j < nbk; ++j) {
                prefix[j] = (int)units.size();
                for (int k0 = 0; k0 <= j; k0 += kc)
                    units.push_back(make_int4(j, k0, std::min(k0 + kc, j + 1), (int)units.size()));
            }
            prefix[nbk] = (int)units.size();
            // launch order: heaviest first (stable: ties keep the canonical order)
            std::stable_sort(units.begin(), units.end(), [](const int4& a, const int4& b) { return a.z - a.y > b.z - b.y; });
            if (gp->d_wunits[v]) ROBO_HIP_CHECK(hipFree(gp->d_wunits[v]));
            if (gp->d_wprefix[v]) ROBO_HIP_CHECK(hipFree(gp->d_wprefix[v]));
            gp->d_wunits[v] = gp->d_wprefix[v] = nullptr;
            ROBO_HIP_CHECK(hipMalloc((void**)&gp->d_wunits[v], units.size() * sizeof(int4)));
            ROBO_HIP_CHECK(hipMalloc((void**)&gp->d_wprefix[v], prefix.size() * sizeof(int)));
            ROBO_HIP_CHECK(hipMemcpyAsync(gp->d_wunits[v], units.data(), units.size() * sizeof(int4), hipMemcpyHostToDevice, st));
            ROBO_HIP_CHECK(hipMemcpyAsync(gp->d_wprefix[v], prefix.data(), prefix.size() * sizeof(int), hipMemcpyHostToDevice, st));
            ROBO_HIP_CHECK(hipStreamSynchronize(st));   // the staging vectors die with this scope
            gp->winv_units[v] = (int)units.size();
            gp->winv_kc[v] = kc;
        }
        gp->winv_nbk = nbk;
    }
    if (gp->winv_gen == gp->fit_gen || gp->winv_launched == gp->fit_gen) return ROBO_OK;
    const int s = launch_triinv(gp, gp->d_Winv, gp->d_gV);
    if (s != ROBO_OK) return s;
    // cond_inf(L) = |L|_inf |W|_inf, exact for the W just built: decides whether this factor may use W at all
    if (!gp->d_wnorm) ROBO_HIP_CHECK(hipMalloc((void**)&gp->d_wnorm, 2 * sizeof(double)));
    if (!gp->h_wnorm) ROBO_HIP_CHECK(hipHostMalloc((void**)&gp->h_wnorm, 2 * sizeof(double), 0));
    ROBO_HIP_CHECK(hipMemsetAsync(gp->d_wnorm, 0, 2 * sizeof(double), st));
    hipLaunchKernelGGL(winv_norm_kernel, dim3((unsigned)((gp->n + 3) / 4)), dim3(256), 0, st, (const double*)gp->d_K,
                       (const double*)gp->d_Winv, gp->n_pad, gp->n, reinterpret_cast<unsigned long long*>(gp->d_wnorm));
    ROBO_LAUNCH_CHECK();
    ROBO_HIP_CHECK(hipMemcpyAsync(gp->h_wnorm, gp->d_wnorm, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    gp->winv_launched = gp->fit_gen;
    return ROBO_OK;
}

// W of the current factor (built here unless a prefetch already launched it) and its condition number
int winv_ensure(robo_gp* gp) {
    if (gp->winv_gen == gp->fit_gen && gp->winv_nbk == (gp->n + NB - 1) / NB) return ROBO_OK;
    const int s = winv_launch(gp);
    if (s != ROBO_OK) return s;
    if (gp->winv_gen != gp->fit_gen) {
        ROBO_HIP_CHECK(hipStreamSynchronize(gp->ctx->stream));      // (returns at once when a prefetched build has finished)
        gp->winv_cond = gp->h_wnorm[0] * gp->h_wnorm[1];             // NaN (a broken W) compares false against every bound
        gp->winv_gen = gp->fit_gen;
    }
    return ROBO_OK;
}

static int grow(double** p, size_t* have, size_t need) {
    if (*have >= need) return ROBO_OK;
    if (*p) ROBO_HIP_CHECK(hipFree(*p));
    *p = nullptr;
    *have = 0;
    ROBO_HIP_CHECK(hipMalloc((void**)p, need));
    *have = need;
    return ROBO_OK;
}

int launch_predict_winv(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn, bool store_v) {
    hipStream_t st = gp->ctx->stream;
    const int n_pad = gp->n_pad, nbk = gp->winv_nbk;
    const unsigned cts = (unsigned)(cn / NB);
    // chunk depth of the unit form by the handle's TOTAL batch (not this workspace pass: a chunked workspace yields the values
    // of a single pass): eight candidate tiles and more take the batch depth, fewer take half of it -- a unit is the critical
    // path of a small batch.  Measured (r04x, MI355X, ms for depth full / half / quarter): N = 4096: 1..128 candidates 0.221 /
    // 0.182 / 0.222, 500 candidates 0.291 / 0.295 / 0.340; N = 2048, 500 candidates 0.214 / 0.159 / 0.169; N = 1000: 0.138 /
    // 0.120 / 0.132; from 2048 candidates up the full depth wins (0.72 / 0.75 / 0.89).  A quarter never wins (its reduction
    // pass reads twice the unit tiles): the third table is only reachable through the tuning key.
    const long long cts_total = cand->m_pad / NB;
    const int shift = gp->ctx->tune.winv_kc_shift;
    const int v = shift >= 0 ? (shift > 2 ? 2 : shift) : (cts_total >= 8 ? 0 : 1);
    const int nu = gp->winv_units[v];
    // whole contraction range per (candidate tile, block row) when those pairs fill the chip by themselves: decided from
    // the handle's TOTAL batch (not this workspace pass), so a chunked workspace yields the values of a single pass
    const long long slots = 2LL * gp->ctx->num_cu;
    const int rows_mode = gp->ctx->tune.winv_rows;                 // -1 auto, 0 never, 1 always (A/B, tests)
    const bool whole = rows_mode >= 0 ? rows_mode != 0 : (cand->m_pad / NB) * (long long)(nbk + 1) >= 2 * slots;
    // a handful of candidates: matrix-vector form (tuning winv_gemv: -1 auto = at most 8 candidates unless another form is
    // forced, 0 never, 1 whenever it applies)
    const int gemv_mode = gp->ctx->tune.winv_gemv;
    const bool gemv = (gemv_mode > 0 || (gemv_mode < 0 && rows_mode < 0 && shift < 0)) && cand->m <= 8 && cand->m_pad == NB;
    const int nslab = nbk * NB / GV_ROWS;       // through the last block row: its columns >= n are stored as 0, like the other forms
    const int nparts = gemv ? nslab : nbk;
    int s = grow(&cand->d_Ks, &cand->ks_bytes, (size_t)cn * n_pad * sizeof(double));
    if (s == ROBO_OK && !whole && !gemv) s = grow(&cand->d_P, &cand->p_bytes, (size_t)cts * nu * NB * NB * sizeof(double));
    if (s == ROBO_OK) s = grow(&cand->d_qpart, &cand->qpart_bytes, (size_t)2 * nparts * cn * sizeof(double));
    if (s != ROBO_OK) return s;
    double* qpart = cand->d_qpart;
    double* mupart = cand->d_qpart + (size_t)nparts * cn;
    s = launch_cross_gram(gp, cand, c0, cn, cand->d_Ks);
    if (s != ROBO_OK) return s;
    const double* z = gp->d_K + (size_t)gp->n * n_pad;
    if (gemv) {
        cand->solve_kernel = "winv_gemv_kernel";
#define ROBO_GEMV(MC, SV)                                                                                            \
    hipLaunchKernelGGL((winv_gemv_kernel<MC, SV>), dim3((unsigned)nslab), dim3(256), 0, st, (const double*)cand->d_Ks,  \
                       n_pad, (const double*)gp->d_Winv, n_pad, z, gp->n, cand->d_V, n_pad, qpart, mupart, (long long)cn)
        if (cand->m == 1) {
            if (store_v) ROBO_GEMV(1, true);
            else ROBO_GEMV(1, false);
        } else {
            if (store_v) ROBO_GEMV(8, true);
            else ROBO_GEMV(8, false);
        }
#undef ROBO_GEMV
    } else if (whole) {
        cand->solve_kernel = "winv_row_kernel";
        if (store_v)
            hipLaunchKernelGGL(winv_row_kernel<true>, dim3(cts, (unsigned)nbk), dim3(256), 0, st, (const double*)cand->d_Ks,
                               n_pad, (const double*)gp->d_Winv, n_pad, nbk, z, gp->n, cand->d_V, n_pad, qpart, mupart,
                               (long long)cn);
        else
            hipLaunchKernelGGL(winv_row_kernel<false>, dim3(cts, (unsigned)nbk), dim3(256), 0, st, (const double*)cand->d_Ks,
                               n_pad, (const double*)gp->d_Winv, n_pad, nbk, z, gp->n, cand->d_V, n_pad, qpart, mupart,
                               (long long)cn);
    } else {
        cand->solve_kernel = "winv_gemm_kernel";
        hipLaunchKernelGGL(winv_gemm_kernel, dim3(cts, (unsigned)nu), dim3(256), 0, st, (const double*)cand->d_Ks, n_pad,
                           (const double*)gp->d_Winv, n_pad, (const int4*)gp->d_wunits[v], nu, cand->d_P);
        if (store_v)
            hipLaunchKernelGGL(winv_reduce_kernel<true>, dim3((unsigned)nbk, cts), dim3(256), 0, st,
                               (const double*)cand->d_P, (const int*)gp->d_wprefix[v], nu, z, gp->n, cand->d_V, n_pad, qpart,
                               mupart, (long long)cn);
        else
            hipLaunchKernelGGL(winv_reduce_kernel<false>, dim3((unsigned)nbk, cts), dim3(256), 0, st,
                               (const double*)cand->d_P, (const int*)gp->d_wprefix[v], nu, z, gp->n, cand->d_V, n_pad, qpart,
                               mupart, (long long)cn);
    }
    if (gemv)
        hipLaunchKernelGGL(winv_finish_few_kernel, dim3(1), dim3(256), 0, st, (const double*)qpart, (const double*)mupart,
                           nparts, (long long)cn, cand->m == 1 ? 1 : 8, cand->d_q + c0, cand->d_mu + c0);
    else
        hipLaunchKernelGGL(winv_finish_kernel, dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, st, (const double*)qpart,
                           (const double*)mupart, nparts, (long long)cn, cand->d_q + c0, cand->d_mu + c0);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo
