// Shared declarations of librobo_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/robo_hip.h"

namespace robo {

// ---- tiling constants ---------------------------------------------------------------
// Everything N x N is stored row-major fp64 with leading dimension n_pad, a multiple of
// NB.  NB is the Cholesky panel width, the TRSM block and the GEMM workgroup tile edge.
constexpr int NB = 128;
constexpr int WG = 256;          // threads per workgroup of every tiled kernel (4 wave64)
constexpr double JITTER = 1.25e-12;  // george's diagonal jitter
constexpr int MAX_DIM = 256;
constexpr int PROG_STRIDE = 512;     // progress words per sample: one per panel (n_pad <= 65 536)

// v_mfma_f64_16x16x4_f64 accumulator: 4 f64 per lane
typedef double v4d __attribute__((vector_size(32)));

__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Single correctly rounded fp64 operations that are NEVER contracted into a fused multiply-add, for the places whose
// results must equal NumPy's bit for bit (the stretch-move proposal of the hyper-parameter chain, mcmc_dev.h).
// ROCm's __dmul_rn / __dadd_rn / __dsub_rn are the plain operators (__clang_hip_math.h) and hipcc's default
// -ffp-contract=fast-honor-pragmas fuses them: round 5's  c - z (c - s)  was compiled to  v_fma_f64 q = -z t + c  and
// the chains left the reference's after a few hundred steps on the MI355X.  The pragma removes the `contract` flag
// from the operation itself, so it stays unfused wherever it is inlined (tests/test_isa.py checks the machine code).
__device__ __forceinline__ double rn_add(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double rn_sub(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double rn_mul(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double rn_div(double a, double b) {
#pragma clang fp contract(off)
    return a / b;
}

// ---- hand-offs between workgroups of ONE launch (the factorisation's panel followers, potrf.hip) ------------------------
// Per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores: a payload another
// workgroup reads in the same launch is stored WRITE-THROUGH (relaxed agent-scope atomics = global_store ... sc1) and read
// with L1-bypassing loads of the same kind; the producer drains its stores (s_waitcnt vmcnt(0)) before ONE lane stores the
// progress word, the consumer polls that one word relaxed (MI355X guide, "inter-workgroup visibility": the sc1 / sc1 form).
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent_u32(const unsigned* p) {
    return __hip_atomic_load(const_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void add_agent_u32(unsigned* p, unsigned v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every vector-memory operation of this wave has completed (vmcnt(0); expcnt / lgkmcnt fields left at their maxima)
__device__ __forceinline__ void drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }
constexpr long long FOLLOW_SENTINEL = 0x7FF8DEAD00C0FFEELL;   // a NaN no arithmetic produces: "W_77 of this panel is not published yet"
constexpr unsigned PROG_SPIN_LIMIT = 1u << 24;   // bounded spin (~seconds): a hand-off that never arrives ends in a flagged failure

// covariance function of the current theta (see kern_math.h)
struct CovParams {
    int kind;   // robo_kernel_kind
    int dim;    // input dimensions (FABOLAS: the last one is the basis-transformed fidelity u)
    double amp, blr_a, blr_b;
};

// per-theta inputs of one fit; the fit kernels take an array of these and index it with the
// batch coordinate of the grid, so S hyper-parameter samples are factorised by ONE sequence of
// launches (S x more workgroups per launch: the MCMC inner loop of the reference evaluates
// n_hypers/2 independent likelihoods per ensemble half-step)
struct FitSample {
    CovParams cov;
    double noise;    // exp(theta[P-1]) + JITTER, added to the diagonal
    double mean_c;   // constant prior mean
};

// one theta as kernel arguments (single-sample fits, gram.hip)
struct ThetaArgs {
    FitSample sp;
    double ism[MAX_DIM];   // 1 / sqrt(metric_d)
};

void set_error(const char* fmt, ...);

#define ROBO_HIP_CHECK(expr)                                                                       \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            robo::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return ROBO_RUNTIME_ERROR;                                                             \
        }                                                                                          \
    } while (0)

#define ROBO_LAUNCH_CHECK()                                                                        \
    do {                                                                                           \
        hipError_t _e = hipGetLastError();                                                         \
        if (_e != hipSuccess) {                                                                    \
            robo::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return ROBO_RUNTIME_ERROR;                                                             \
        }                                                                                          \
    } while (0)

#define ROBO_TRY(expr)                \
    do {                              \
        int _s = (expr);              \
        if (_s != ROBO_OK) return _s; \
    } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace robo

// ---- handle layouts --------------------------------------------------------------------
namespace robo {
// Tuning knobs of a context: read ONCE from the environment when the context is created (ROBO_<NAME>), changed
// afterwards only through robo_ctx_set_tuning -- nothing on the hot path calls getenv.  -1 = automatic.
struct Tuning {
    long long ws_bytes;          // solve workspace per candidate handle (default 6 GiB)
    long long trsm_small_max;    // batches <= this use the 16/32-candidate block-row step
    int trsm_small_narrow;       // -1 auto, 0 / 1 force 32 / 16 candidates per workgroup
    int trsm_small_deep;         // -1 auto, 0 / 1 force 1 / 2 k-tiles per staging stage
    int trsm_rows;               // block rows per launch of the 128-candidate step
    int trsm_pair;               // 1: two block rows per launch on one read of V (trsm_pair_gen_kernel; measured slower: default 0)
    int predict_stepwise;        // 1: cross-gram in memory + trsm_step_kernel (A/B)
    long long winv_max;          // batches <= this (and >= winv_min_blocks block rows) go through W = L^-1 (0: never)
    int winv_min_blocks;
    int winv_gemv;               // explicit-inverse posterior of <= 8 candidates as a matrix-vector product: -1 auto, 0 never
    int winv_kc_shift;           // chunked explicit-inverse product: -1 auto (by batch size), 0 / 1 / 2 = the batch chunk depth, half, quarter
    int winv_rows;               // explicit-inverse product per (tile, block row) over the whole contraction range: -1 auto
                                 // (batches that fill the chip that way), 0 never (chunked units + reduction), 1 always
    long long winv_cond_max;     // ... while cond_inf(L) = |L|_inf |W|_inf stays below this (default 1e5)
    int potrf_fused;             // 0: never fuse the next diagonal block into the trailing update
    int potrf_fused_panels;      // ... also for batches of factors with at most this many panels (0: single fits only; -1 =
                                 // default: while the first step's launch is at most three rounds of CUs -- 26 walkers: N <= 766)
    int potrf_tm4_min, potrf_max_wg, potrf_group;
    int potrf_batch_tm4_min;     // batched updates: 128-row tiles from this many (tiles x samples) on, 32-row tiles below (96;
                                 // tests lower it so that the interpreter reaches the 128-row form at small N)
    int potrf_thin_last;         // batched fit: 32-row tiles for the block row that holds only the augmented row (1; 0 = A/B)
    int potrf_split;             // batched fit: sub-batches on their own streams with staggered group boundaries (1: one stream)
    int potrf_gram_split;        // ... with every sub-batch's gram kernel on its own stream (1) or one launch for all (0)
    int potrf_split_min;         // ... from this many panels on (default 12: N >= 1408)
    int potrf_lead;              // ... first-group size step between sub-batches (-1: G / splits)
    int potrf_tail_split;        // fused step: the ragged last round of 128-row tiles as half / quarter tiles on more workgroups
    int potrf_follow;            // single-theta fit: the panel solve of column k+1 FOLLOWS the diagonal block inside the step
                                 // kernel (progress words, potrf_step_follow_kernel) instead of its own launch (default 1; 0: the launch-per-phase form)
    int potrf_follow_from;       // ... from this step on (-1: the first panel too; -2 = default: by size, launch_potrf)
    int potrf_pub_early;         // ... the diagonal block's helper waves count a published column at once from this interval on
    int potrf_follow_rows;       // ... rows per follower workgroup: 64 (two per block row), 128 (one), -1: by step (launch_potrf)
    int potrf_poll_sleep;        // ... pauses (x 128 cycles) between two polls of a follower
    int potrf_batch_roll;        // ... with the diagonal workgroup's 80-KB rolling layout (two workgroups per CU); 0: the 150-KB image
    int potrf_batch_follow;      // batched fits: diagonal block + panel of a step in one launch, the panel following (-1 = default: up to 17 panels; 0 / 1)
    int mcmc_fused_tail;         // device chain, multi-block factors: likelihood terms + accept test in one launch (1)
    int mcmc_block_step;         // ensemble half-step in ONE launch: 2 (default) every one-block problem, 1 only N <= 63, 0 never;
                                 // 3: two-block problems (N <= 254) too -- built in r06, measured SLOWER than the launch path
};
void tuning_from_env(Tuning* t);
}  // namespace robo
struct robo_ctx;
namespace robo {
int ctx_aux_streams(robo_ctx* ctx);   // api.hip: create ctx->aux / ev_fork / ev_join once
void ctx_retain(robo_ctx* ctx);        // a handle was created on ctx
void ctx_release(robo_ctx* ctx);       // ... destroyed: frees a closing context with its last handle
}

constexpr int ROBO_AUX_STREAMS = 3;
struct robo_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    // side streams of the batched factorisation's sub-batches (potrf.hip), created on first use; fork / join events
    hipStream_t aux[ROBO_AUX_STREAMS];
    hipEvent_t ev_fork, ev_join[ROBO_AUX_STREAMS];
    bool aux_ready;
    // lifetime: handles created on this context (robo_gp, robo_cand, robo_comm, robo_multi) keep it alive.  robo_ctx_destroy
    // marks it `closing`; stream, events and scratch are released when the last such handle is destroyed (api.hip
    // ctx_retain / ctx_release).  A binding whose garbage collector finalises a context before the objects that live on it
    // (Python's cyclic GC gives no order) therefore cannot make a later robo_gp_destroy touch a dead stream.
    int users;
    bool closing;
    hipEvent_t events[32];
    bool phase_events;   // record the internal phase events of robo_gp_fit (robo_ctx_set_phase_events, default off)
    char name[256];
    int num_cu;
    robo::Tuning tune;
    // scratch shared by every call on this context
    double* d_scalars;   // [8]: quad, logdet, ...
    int* d_fail;         // first failing column + 1, or 0
    unsigned* d_prog;    // [2][PROG_STRIDE] progress words of the follower hand-off (potrf.hip), zeroed per factorisation
    double* h_pinned;    // small pinned staging (64 doubles)
};

namespace robo {
// Packed inverse diagonal blocks for the transposed block-row solve (predict.hip): the 36 lower 16x16 sub-blocks x 4
// k-steps of a 128x128 inverse as 64-lane MFMA A-operand fragments, in the order the solve consumes them (row block
// 7 first, column blocks ascending).
constexpr int WP_FRAGS = 144;
constexpr int WP_BLOCK = WP_FRAGS * 64;
__host__ __device__ constexpr int wp_offset(int cb) { return 4 * (36 - (cb + 1) * (cb + 2) / 2); }
}  // namespace robo

struct robo_gp {
    robo_ctx* ctx;
    int kind, dim, n_max;
    int n;          // training points
    int n_pad;      // round_up(n + 1, NB): row n is the augmented (y - mean) row, rest identity
    int n_pad_max;
    bool has_data, fitted;
    unsigned long long fit_gen;   // process-wide serial number of the factor this handle holds (0: none)
    struct robo_cand* host_cand;  // candidate handle behind the host-array entry points, kept between calls of one size
    bool fp32_gram;     // mixed precision: covariance entries evaluated in fp32 (BASELINE config 5)
    robo::CovParams cov;   // kind, dim, amp, blr_a, blr_b of the current theta
    double amp, noise, mean_c;
    double y_mean, y_std;
    double loglik;
    double* d_X;        // (n_max, dim) raw inputs
    double* d_Xs;       // (n_pad_max, dim) inputs scaled by 1/sqrt(metric_d)
    double* d_y;        // (n_max)
    double* d_K;        // (n_pad_max, n_pad_max) gram -> Cholesky factor in place (lower)
    double* d_Linv;     // (n_pad_max / NB) x NB x NB inverses of the diagonal blocks.  With n a multiple of NB the LAST block
                        // (the augmented row alone) is never factored: its inverse slots here and in d_LinvP are ZERO, not
                        // an inverse (robo_gp_set_data clears them); every consumer walks ceil(n / NB) block rows only
    double* d_LinvP;    // the same inverses as packed MFMA A-operand fragments (WP_BLOCK doubles per block, predict.hip)
    double* d_theta;    // inverse sqrt metric (dim) of the current theta
    robo::FitSample* d_sp;   // device copy of the current FitSample
    // batch workspace for robo_gp_loglik_batch (lazy, b_cap samples)
    int b_cap, b_npad;
    double *d_bK, *d_bLinv, *d_bXs, *d_bism, *d_bout, *h_bstage;
    robo::FitSample* d_bsp;
    int* d_bfail;
    unsigned* d_bprog;  // [b_cap][PROG_STRIDE] progress words of the batched follower form
    double *d_llpart, *d_bllpart;   // log-likelihood partials of the tail kernel (own factor / batch workspace)
    char* d_bkeep;                  // [b_cap] KeepDst records of robo_gp_fit_batch
    // workspace of robo_gp_grad_loglik (lazy, sized for n_pad_max): W^T, A = alpha alpha^T - K^-1, ...
    double *d_gV, *d_gA, *d_galpha, *d_gpart, *d_gout;
    // explicit inverse factor W = L^-1 for small candidate batches (winv.hip): lazy, rebuilt when the factor changes
    char* d_mcmc;                   // scratch of robo_gp_mcmc_run (one block, lazily grown)
    size_t mcmc_bytes;
    double* d_Winv;                 // (n_pad_max, n_pad_max) row-major, lower block triangle valid
    unsigned long long winv_gen;    // fit_gen the inverse was built for (0: none)
    // unit tables of the triangular product for winv_nbk block rows, one per contraction-chunk depth (the batch depth, half
    // and a quarter of it: small batches take shallower units -- more of them, shorter critical path)
    int* d_wunits[3];               // int4 per unit
    int* d_wprefix[3];              // first canonical unit of every block row (winv_nbk + 1 entries)
    int winv_units[3], winv_kc[3];
    int winv_nbk;
    double diag_min, diag_max;      // extreme diagonal entries of L over the training rows (0, 0: unknown)
    double winv_cond;               // cond_inf(L) = |L|_inf |W|_inf of the factor W was built for (winv_gen); 0: unknown
    double* d_wnorm;                // [2]: |L|_inf, |W|_inf (bit patterns, atomicMax)
    double* h_wnorm;                // pinned copy of the two norms (written by an asynchronous copy behind the build)
    unsigned long long winv_launched;   // fit_gen whose W build has been LAUNCHED (robo_gp_prefetch_inverse) but not yet read
};

struct robo_cand {
    robo_ctx* ctx;
    int dim;
    int64_t m;          // candidates
    int64_t m_pad;      // round_up(m, NB)
    double* d_Xc;       // (m_pad, dim) raw candidates (pad rows replicate row 0)
    double* d_Xcs;      // (m_pad, dim) scaled for the GP being evaluated
    double* d_V;        // (chunk, ldv) cross-gram -> L^-1 k_* in place
    int64_t chunk;      // candidates per workspace pass (multiple of NB)
    const char* solve_kernel;   // name of the kernel that ran the last posterior's solve (diagnostics: bench labels)
    int ldv;            // n_pad the workspace was sized for
    size_t v_bytes;
    const robo_gp* solved_gp;        // d_V holds L^-1 k_* of ALL the points for this factor (entropy search keeps
    unsigned long long solved_gen;   // the representer points' solve across compute() calls); 0 = not valid
    double* d_q;        // (m_pad) sum_n V^2
    double* d_mu;       // (m_pad) V . z
    double* d_mean;     // (m_pad) transformed mean
    double* d_var;      // (m_pad) transformed, floored variance
    double* d_acq;      // (m_pad)
    double* d_acq_sum;  // (m_pad) marginal accumulator
    // entropy search (lazy): cross-covariances with the representer points, quadratic-form
    // features / outputs, EP tensors
    double *d_S, *d_F, *d_Q, *d_G, *d_igc;
    size_t f_cap, q_cap, g_cap;
    double* h_igkey;    // host copy of the EP state whose device form d_G / d_igc hold (compared in full before an upload
    size_t igkey_len;   // is skipped: compute() is called many times per update(), information_gain.py:87-125 / :153-167)
    double* d_mu_all;   // (s_cap, m_pad) per-sample means/variances for the GP-MCMC mixture (lazy)
    double* d_var_all;
    int s_cap;
    // explicit-inverse path (winv.hip), lazy: cross-gram K_* (chunk x n_pad), per-unit product tiles, per-block-row
    // partial sums of |v|^2 and v.z
    double *d_Ks, *d_P, *d_qpart;
    size_t ks_bytes, p_bytes, qpart_bytes;
    double* d_part_val; // per-block argmax partials
    long long* d_part_idx;
    unsigned* d_flags;
    int n_part;
};

// ---- launchers implemented in the .hip files (all asynchronous on ctx->stream) ------------
namespace robo {
int launch_scale_inputs(robo_ctx* ctx, const double* d_in, double* d_out, const double* d_inv_sqrt_metric,
                        int64_t rows_real, int64_t rows_pad, int dim, int S = 1, size_t out_stride = 0,
                        size_t ism_stride = 0);
// the fit pipeline on S matrices at once (S = 1: the GP's own buffers)
struct FitBuffers {
    double* K; size_t k_stride;          // (n_pad x n_pad) per sample
    double* Linv; size_t linv_stride;    // (n_pad/128 x 128 x 128) per sample
    const double* Xs; size_t xs_stride;  // (n_pad x dim) per sample
    const FitSample* sp;                 // [S]
    int* fail;                           // [S]
    unsigned* prog;                      // [S][PROG_STRIDE] progress words (single-theta fits), or nullptr
    double* out;                         // [S][2]: z.z, 2 sum log diag
    double* ll_part;                     // [S][n_pad/128][4] per block: z.z and sum log L_ii shares, min / max L_ii
    double* LinvP;                       // packed inverse fragments of the GP's own factor, or nullptr (batch workspace)
    double* host_out;                    // pinned host [S][5]: z.z, 2 sum log diag, failure flag, min / max L_ii -- or nullptr
    bool want_inverse;                   // false: log-likelihood only (no explicit inverse blocks, no fragments)
    bool skip_tail;                      // the caller reduces the likelihood terms itself (mcmc.hip mcmc_tail_kernel)
    int S;
};
int launch_scale_inputs_theta(robo_ctx* ctx, const double* d_in, double* d_out, const ThetaArgs& ta, int64_t rows_real,
                              int64_t rows_pad, int dim, double* d_ism_out, FitSample* d_sp_out);
// destination of one sample of a batched fit that keeps its factors (potrf.hip batch_keep_kernel)
struct KeepDst {
    double *K, *Linv, *LinvP, *Xs, *theta;
    double *X, *y;            // nullptr for gps[0] (owns the training data already)
    FitSample* sp;
    int ok;
};
int launch_batch_keep(robo_gp* g0, const KeepDst* d_dst, int ns);
int launch_gram(robo_gp* gp, const FitBuffers& fb, hipStream_t stream = nullptr, int s0 = 0, int ns = -1);
// the device-resident hyper-parameter chain (mcmc.hip): everything the two kernels around the batched fit need
struct McmcState {
    int k, P, D, kind, n, n_steps, ns_eval, prior_kind;
    double a, mean_c;
    double prior_par[9];            // mcmc_dev.h PRIOR_PAR
    double *d_pos, *d_lnp, *d_q, *d_z, *d_prior;      // (k x P), (k), (k/2 x P), (k/2), (k/2)
    long long* d_nacc;                                // (k)
    int *d_it, *d_err;
    const double *d_uz, *d_ua;                        // (n_steps x 2 x k/2) each, emcee's draw order
    const int* d_partner;
    double *d_chain, *d_lnprob;                       // (k x n_steps x P), (k x n_steps); nullable
    FitSample* d_sp;                                  // the batched fit's inputs / outputs (api.hip batch_ensure)
    double* d_ism;
    const double* d_out;
    const int* d_fail;
};
int launch_mcmc_propose_scale(robo_ctx* ctx, const McmcState& st, int start, int first, int h, int it, const double* d_X,
                              double* d_Xs, int64_t rows_real, int64_t rows_pad, size_t xs_stride);
int launch_mcmc_accept(robo_ctx* ctx, const McmcState& st, int start, int first, int h, int it);
int launch_mcmc_tail(robo_ctx* ctx, const McmcState& st, int start, int first, int h, int it, const double* d_K, size_t k_stride,
                     int ld, int nbf, const int* d_fail);
int launch_mcmc_block_step(robo_gp* gp, const McmcState& st, int start, int first, int h, int it);
int launch_mcmc_block2_step(robo_gp* gp, const McmcState& st, int start, int first, int h, int it, double* d_K, size_t k_stride);
// with_gram: the gram matrices are built here too, every sub-batch's on the stream its factorisation runs on
int launch_potrf(robo_gp* gp, const FitBuffers& fb, bool with_gram = false);
int launch_diag_timeline(robo_gp* gp, long long* d_stamps);
int launch_cross_gram(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn, double* d_out = nullptr);
// posterior of a chunk through W = L^-1 (winv.hip): fills cand->d_q / d_mu (and d_V when store_v)
int winv_ensure(robo_gp* gp);
int winv_launch(robo_gp* gp);
int launch_per_cost(robo_ctx* ctx, double* d_dh, const double* d_log_cost, double overhead, int64_t m);
int launch_predict_winv(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn, bool store_v);
int launch_trsm(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn);
int launch_predict_fused(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn);
int launch_pack_linv(robo_gp* gp);
int launch_grad_loglik(robo_gp* gp, double* d_V, double* d_A, double* d_alpha, double* d_part, double* d_out,
                       double* h_out);
int launch_triinv(robo_gp* gp, double* d_W, double* d_V);   // W = L^-1 (and V = W^T) of the current factor
int launch_post(robo_gp* gp, robo_cand* cand, int64_t c0, int64_t cn);
int launch_cross_grad(robo_gp* gp, const double* d_Xcs, double* d_V, int64_t c_first, int64_t c_count, int64_t rows_pad);
int launch_predgrad_post(robo_gp* gp, const double* d_V, const double* d_q, const double* d_mu, const double* d_Xcs,
                         int64_t c_first, int64_t c_count, double* d_mean, double* d_var, double* d_dmean,
                         double* d_dvar);
int launch_acq(robo_ctx* ctx, robo_cand* cand, int acq_kind, double par, double eta, bool accumulate, bool first);
int launch_argmax(robo_cand* cand, const double* d_vals, double scale);
int launch_report_best(robo_cand* cand, double* h_pinned);
int launch_cov(robo_gp* gp, robo_cand* cand, double* d_cov);
int launch_mixture(robo_cand* cand, int S);
int launch_cross_cov(robo_gp* gp, robo_cand* cand, robo_cand* rep, int64_t c0, int64_t cn, double* d_S);
int launch_ig_dh(robo_ctx* ctx, const double* d_S, const double* d_var, double* d_F, double* d_Q, const double* d_G,
                 const double* d_consts, int64_t c0, int64_t cn, int64_t m, int nb, int npts, int kf, double sn2,
                 double H, double* d_out);
int launch_random_candidates(robo_ctx* ctx, double* d_out, int64_t m_pad, int dim, uint64_t seed, int64_t n_uniform,
                             const double* d_loc, const double* d_scale);
int launch_uniform(robo_ctx* ctx, double* d_out, int64_t m, int64_t m_pad, int dim, uint64_t seed);
int launch_sobol(robo_ctx* ctx, double* d_out, int64_t m, int64_t m_pad, int dim, const unsigned long long* d_sv,
                 const unsigned long long* d_shift, int bits, uint64_t first);
int launch_mfma_selftest(robo_ctx* ctx, double* out_err);
int launch_stretch_probe(robo_ctx* ctx, const double* h_c, const double* h_s, const double* h_u, double a, int P, int n,
                         double* h_z, double* h_q, double* h_d);
int launch_gemm_microbench(robo_ctx* ctx, int variant, int wgs, int K, int reps, double* out_tflops);
int launch_clock_sampler(robo_ctx* ctx, int window_us);
int collect_clock_sampler(double* out3);
int launch_mfma_microbench(robo_ctx* ctx, int iters, double* out_tflops, double* out_cycles_per_mfma,
                           double* out_shader_mhz, double* out_chain);
}  // namespace robo
