// C ABI of librobo_hip.so (see include/robo_hip.h for the contract and the reference call
// sites each entry point replaces).  Host-side orchestration only: every number is
// produced by the kernels in gram.hip / potrf.hip / predict.hip / acq.hip.
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <vector>

#include "common.h"

namespace robo {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- tuning knobs: environment -> context, once (robo_ctx_create); robo_ctx_set_tuning afterwards -------------------
struct TuneKey {
    const char* name;          // robo_ctx_set_tuning key; the environment variable is ROBO_<NAME in upper case>
    long long Tuning::*ll;
    int Tuning::*i;
    long long dflt;
};
static const TuneKey TUNE_KEYS[] = {
    {"ws_bytes", &Tuning::ws_bytes, nullptr, (long long)6 << 30},
    {"trsm_small_max", &Tuning::trsm_small_max, nullptr, 16384},
    {"trsm_small_narrow", nullptr, &Tuning::trsm_small_narrow, -1},
    {"trsm_small_deep", nullptr, &Tuning::trsm_small_deep, -1},
    {"trsm_rows", nullptr, &Tuning::trsm_rows, 1},
    {"trsm_pair", nullptr, &Tuning::trsm_pair, 0},
    {"predict_stepwise", nullptr, &Tuning::predict_stepwise, 0},
    {"winv_max", &Tuning::winv_max, nullptr, 32768},
    {"winv_min_blocks", nullptr, &Tuning::winv_min_blocks, 6},
    {"winv_cond_max", &Tuning::winv_cond_max, nullptr, 100000},
    {"winv_rows", nullptr, &Tuning::winv_rows, -1},
    {"winv_kc_shift", nullptr, &Tuning::winv_kc_shift, -1},
    {"winv_gemv", nullptr, &Tuning::winv_gemv, -1},
    {"potrf_fused", nullptr, &Tuning::potrf_fused, 1},
    {"potrf_fused_panels", nullptr, &Tuning::potrf_fused_panels, -1},
    {"potrf_tm4_min", nullptr, &Tuning::potrf_tm4_min, 96},
    {"potrf_max_wg", nullptr, &Tuning::potrf_max_wg, 0},
    {"potrf_group", nullptr, &Tuning::potrf_group, 0},
    {"potrf_tail_split", nullptr, &Tuning::potrf_tail_split, 1},
    {"potrf_follow", nullptr, &Tuning::potrf_follow, 1},
    {"potrf_follow_from", nullptr, &Tuning::potrf_follow_from, -2},
    {"potrf_pub_early", nullptr, &Tuning::potrf_pub_early, 6},
    {"potrf_follow_rows", nullptr, &Tuning::potrf_follow_rows, -1},
    {"potrf_poll_sleep", nullptr, &Tuning::potrf_poll_sleep, 1},
    {"potrf_batch_roll", nullptr, &Tuning::potrf_batch_roll, 0},
    {"potrf_batch_follow", nullptr, &Tuning::potrf_batch_follow, -1},
    {"potrf_batch_tm4_min", nullptr, &Tuning::potrf_batch_tm4_min, 96},
    {"potrf_thin_last", nullptr, &Tuning::potrf_thin_last, 1},
    {"potrf_split", nullptr, &Tuning::potrf_split, 3},
    {"potrf_gram_split", nullptr, &Tuning::potrf_gram_split, 0},
    {"potrf_split_min", nullptr, &Tuning::potrf_split_min, 12},
    {"potrf_lead", nullptr, &Tuning::potrf_lead, -1},
    {"mcmc_block_step", nullptr, &Tuning::mcmc_block_step, 2},
    {"mcmc_fused_tail", nullptr, &Tuning::mcmc_fused_tail, 1},
};

static void tune_set(Tuning* t, const TuneKey& k, long long v) {
    if (k.ll) t->*(k.ll) = v;
    else t->*(k.i) = (int)v;
}

void tuning_from_env(Tuning* t) {
    for (const TuneKey& k : TUNE_KEYS) {
        char env[64] = "ROBO_";
        size_t o = strlen(env);
        for (const char* p = k.name; *p && o + 1 < sizeof(env); ++p) env[o++] = (char)toupper((unsigned char)*p);
        env[o] = 0;
        const char* e = getenv(env);
        tune_set(t, k, (e && *e) ? (long long)atof(e) : k.dflt);
    }
    if (t->ws_bytes < 1) t->ws_bytes = (long long)6 << 30;   // callers round down to whole 128-candidate blocks
}

static std::mutex g_ctx_life;
static std::atomic<int> g_ctx_live{0};

static void ctx_free(robo_ctx* c) {
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (int i = 0; i < 32; ++i) hipEventDestroy(c->events[i]);
    if (c->aux_ready) {
        hipEventDestroy(c->ev_fork);
        for (int i = 0; i < ROBO_AUX_STREAMS; ++i) {
            hipStreamSynchronize(c->aux[i]);
            hipStreamDestroy(c->aux[i]);
            hipEventDestroy(c->ev_join[i]);
        }
    }
    hipFree(c->d_scalars);
    hipFree(c->d_fail);
    hipFree(c->d_prog);
    hipHostFree(c->h_pinned);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    --g_ctx_live;
}

void ctx_retain(robo_ctx* c) {
    std::lock_guard<std::mutex> lock(g_ctx_life);
    ++c->users;
}

void ctx_release(robo_ctx* c) {
    bool last;
    {
        std::lock_guard<std::mutex> lock(g_ctx_life);
        last = --c->users == 0 && c->closing;
    }
    if (last) ctx_free(c);
}

int ctx_aux_streams(robo_ctx* c) {
    if (c->aux_ready) return ROBO_OK;
    ROBO_HIP_CHECK(hipSetDevice(c->device));
    ROBO_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < ROBO_AUX_STREAMS; ++i) {
        ROBO_HIP_CHECK(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
        ROBO_HIP_CHECK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    c->aux_ready = true;
    return ROBO_OK;
}

static size_t workspace_bytes(const robo_ctx* c) { return (size_t)c->tune.ws_bytes; }

template <class T>
static int dev_alloc(T** p, size_t count) {
    ROBO_HIP_CHECK(hipMalloc((void**)p, (count ? count : 1) * sizeof(T)));
    return ROBO_OK;
}



}  // namespace robo

using namespace robo;

extern "C" {

const char* robo_last_error_string(void) { return g_err; }
const char* robo_version_string(void) { return "robo_hip 0.1 (gfx950, fp64 MFMA)"; }

int32_t robo_device_count(int32_t* out_n) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *out_n = n;
    return ROBO_OK;
}

int32_t robo_ctx_create(int32_t device, void* hip_stream, robo_ctx** out) {
    if (!out) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(device));
    robo_ctx* c = new robo_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->own_stream = false;
    } else {
        ROBO_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    for (int i = 0; i < 32; ++i) ROBO_HIP_CHECK(hipEventCreate(&c->events[i]));
    {   // internal phase events of robo_gp_fit (slots 19..23): off unless ROBO_PHASE_EVENTS=1 / set_phase_events
        const char* e = getenv("ROBO_PHASE_EVENTS");
        c->phase_events = e && atoi(e) != 0;
    }
    tuning_from_env(&c->tune);   // the only place the tuning variables are read
    hipDeviceProp_t prop;
    ROBO_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
    c->num_cu = prop.multiProcessorCount;
    ROBO_TRY(dev_alloc(&c->d_scalars, 8));
    ROBO_TRY(dev_alloc(&c->d_fail, 4));
    ROBO_TRY(dev_alloc(&c->d_prog, 2 * PROG_STRIDE));
    ROBO_HIP_CHECK(hipHostMalloc((void**)&c->h_pinned, (MAX_DIM + 64) * sizeof(double), 0));
    ++g_ctx_live;
    *out = c;
    return ROBO_OK;
}

int32_t robo_ctx_destroy(robo_ctx* c) {
    if (!c) return ROBO_OK;
    bool now;
    {
        std::lock_guard<std::mutex> lock(g_ctx_life);
        c->closing = true;
        now = c->users == 0;
    }
    if (now) ctx_free(c);        // otherwise with the last handle that lives on it (ctx_release)
    return ROBO_OK;
}

int32_t robo_ctx_live_count(int32_t* out_n) {
    if (!out_n) return ROBO_BAD_ARGUMENT;
    *out_n = g_ctx_live.load();
    return ROBO_OK;
}

int32_t robo_ctx_synchronize(robo_ctx* c) {
    if (!c) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(c->device));
    ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));
    return ROBO_OK;
}

int32_t robo_ctx_device_name(robo_ctx* c, char* buf, int32_t len) {
    if (!buf || len <= 0) return ROBO_BAD_ARGUMENT;
    snprintf(buf, (size_t)len, "%s", c->name);
    return ROBO_OK;
}

int32_t robo_ctx_event_record(robo_ctx* c, int32_t slot) {
    if (slot < 0 || slot >= 32) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(c->device));
    ROBO_HIP_CHECK(hipEventRecord(c->events[slot], c->stream));
    return ROBO_OK;
}

int32_t robo_ctx_set_phase_events(robo_ctx* c, int32_t on) {
    if (!c) return ROBO_BAD_ARGUMENT;
    c->phase_events = on != 0;
    return ROBO_OK;
}

int32_t robo_ctx_set_tuning(robo_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return ROBO_BAD_ARGUMENT;
    if (strcmp(key, "env") == 0) {          // re-read every ROBO_<NAME> variable
        tuning_from_env(&c->tune);
        return ROBO_OK;
    }
    for (const TuneKey& k : TUNE_KEYS)
        if (strcmp(key, k.name) == 0) {
            tune_set(&c->tune, k, value == INT64_MIN ? k.dflt : (long long)value);
            if (c->tune.ws_bytes < 1) c->tune.ws_bytes = (long long)6 << 30;
            return ROBO_OK;
        }
    set_error("robo_ctx_set_tuning: unknown key '%s'", key);
    return ROBO_BAD_ARGUMENT;
}

int32_t robo_ctx_event_elapsed_ms(robo_ctx* c, int32_t a, int32_t b, float* out_ms) {
    if (a < 0 || a >= 32 || b < 0 || b >= 32 || !out_ms) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipEventSynchronize(c->events[b]));
    ROBO_HIP_CHECK(hipEventElapsedTime(out_ms, c->events[a], c->events[b]));
    return ROBO_OK;
}

// ---------------------------------------------------------------------------------------
// GP
// ---------------------------------------------------------------------------------------
int32_t robo_gp_create(robo_ctx* ctx, int32_t kind, int32_t n_max, int32_t dim, robo_gp** out) {
    if (!ctx || !out) return ROBO_BAD_ARGUMENT;
    if (kind != ROBO_KERNEL_MATERN52_ARD && kind != ROBO_KERNEL_RBF_ARD && kind != ROBO_KERNEL_FABOLAS) {
        set_error("unknown kernel kind %d", kind);
        return ROBO_BAD_ARGUMENT;
    }
    if (n_max < 1 || dim < 1 || dim > MAX_DIM || (kind == ROBO_KERNEL_FABOLAS && dim < 2)) {
        set_error("bad shape n_max=%d dim=%d (dim <= %d)", n_max, dim, MAX_DIM);
        return ROBO_BAD_SHAPE;
    }
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    robo_gp* g = new robo_gp();
    memset(g, 0, sizeof(*g));
    g->ctx = ctx;
    g->kind = kind;
    g->dim = dim;
    g->n_max = n_max;
    g->n_pad_max = round_up(n_max + 1, NB);
    g->y_mean = 0.0;
    g->y_std = 1.0;
    const size_t np = (size_t)g->n_pad_max;
    ROBO_TRY(dev_alloc(&g->d_X, (size_t)n_max * dim));
    ROBO_TRY(dev_alloc(&g->d_Xs, np * dim));
    ROBO_TRY(dev_alloc(&g->d_y, (size_t)n_max));
    ROBO_TRY(dev_alloc(&g->d_K, np * np));
    ROBO_TRY(dev_alloc(&g->d_Linv, np * NB));
    // the strictly upper 16x16 sub-blocks of every inverted diagonal block are zero and never written
    ROBO_HIP_CHECK(hipMemset(g->d_Linv, 0, np * NB * sizeof(double)));
    ROBO_TRY(dev_alloc(&g->d_LinvP, (np / NB) * WP_BLOCK));
    ROBO_HIP_CHECK(hipMemset(g->d_LinvP, 0, (np / NB) * WP_BLOCK * sizeof(double)));
    ROBO_TRY(dev_alloc(&g->d_llpart, (np / NB) * 4));
    ROBO_TRY(dev_alloc(&g->d_theta, (size_t)dim + 8 + sizeof(FitSample) / sizeof(double)));
    g->d_sp = reinterpret_cast<FitSample*>(g->d_theta + dim + 8);
    ctx_retain(ctx);
    *out = g;
    return ROBO_OK;
}

int32_t robo_gp_destroy(robo_gp* g) {
    if (!g) return ROBO_OK;
    hipSetDevice(g->ctx->device);
    hipStreamSynchronize(g->ctx->stream);
    robo_cand_destroy(g->host_cand);
    hipFree(g->d_X);
    hipFree(g->d_Xs);
    hipFree(g->d_y);
    hipFree(g->d_K);
    hipFree(g->d_Linv);
    hipFree(g->d_LinvP);
    hipFree(g->d_llpart);
    hipFree(g->d_bllpart);
    hipFree(g->d_bkeep);
    hipFree(g->d_theta);
    hipFree(g->d_Winv);
    hipFree(g->d_wnorm);
    if (g->h_wnorm) hipHostFree(g->h_wnorm);
    hipFree(g->d_mcmc);
    for (int v = 0; v < 3; ++v) {
        hipFree(g->d_wunits[v]);
        hipFree(g->d_wprefix[v]);
    }
    hipFree(g->d_gV);
    hipFree(g->d_gA);
    hipFree(g->d_galpha);
    hipFree(g->d_gpart);
    hipFree(g->d_gout);
    hipFree(g->d_bK);
    hipFree(g->d_bLinv);
    hipFree(g->d_bXs);
    hipFree(g->d_bout);
    hipFree(g->d_bsp);
    hipFree(g->d_bfail);
    hipFree(g->d_bprog);
    if (g->h_bstage) hipHostFree(g->h_bstage);
    robo_ctx* ctx = g->ctx;
    delete g;
    ctx_release(ctx);
    return ROBO_OK;
}

int32_t robo_gp_set_data(robo_gp* g, const double* X, const double* y, int32_t n) {
    if (!g || !X || !y) return ROBO_BAD_ARGUMENT;
    if (n < 1 || n > g->n_max) {
        set_error("n=%d outside [1, n_max=%d]", n, g->n_max);
        return ROBO_BAD_SHAPE;
    }
    robo_ctx* c = g->ctx;
    ROBO_HIP_CHECK(hipSetDevice(c->device));
    ROBO_HIP_CHECK(hipMemcpyAsync(g->d_X, X, (size_t)n * g->dim * sizeof(double), hipMemcpyHostToDevice, c->stream));
    ROBO_HIP_CHECK(hipMemcpyAsync(g->d_y, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (n % NB == 0) {
        // the augmented row's own block (block n / NB) is never factored nor inverted (launch_potrf): its inverse slots
        // must not carry a former data set's entries -- the pack / keep kernels copy every block of the padded range
        const size_t blk = (size_t)(n / NB);
        ROBO_HIP_CHECK(hipMemsetAsync(g->d_Linv + blk * NB * NB, 0, (size_t)NB * NB * sizeof(double), c->stream));
        ROBO_HIP_CHECK(hipMemsetAsync(g->d_LinvP + blk * WP_BLOCK, 0, (size_t)WP_BLOCK * sizeof(double), c->stream));
    }
    ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));
    g->n = n;
    g->n_pad = round_up(n + 1, NB);
    g->has_data = true;
    g->fitted = false;
    return ROBO_OK;
}

int32_t robo_theta_size(int32_t kind, int32_t dim) { return kind == ROBO_KERNEL_FABOLAS ? dim + 3 : dim + 2; }

int32_t robo_gp_set_precision(robo_gp* g, int32_t fp32_gram) {
    if (!g) return ROBO_BAD_ARGUMENT;
    g->fp32_gram = fp32_gram != 0;
    g->fitted = false;
    return ROBO_OK;
}

int32_t robo_gp_set_output_transform(robo_gp* g, double y_mean, double y_std) {
    if (!g) return ROBO_BAD_ARGUMENT;
    g->y_mean = y_mean;
    g->y_std = y_std;
    return ROBO_OK;
}

// theta -> (FitSample, 1/sqrt(metric_d)); returns BAD_ARGUMENT for non-finite entries
static int theta_to_sample(const robo_gp* g, const double* theta, double mean_c, FitSample* sp, double* ism) {
    const int D = g->dim, P = robo_theta_size(g->kind, D);
    for (int p = 0; p < P; ++p)
        if (!std::isfinite(theta[p])) {
            set_error("theta[%d] is not finite", p);
            return ROBO_BAD_ARGUMENT;
        }
    const bool fab = g->kind == ROBO_KERNEL_FABOLAS;
    const int n_metric = fab ? D - 1 : D;
    for (int d = 0; d < n_metric; ++d) ism[d] = std::exp(-0.5 * theta[1 + d]);   // 1/sqrt(metric_d)
    if (fab) ism[D - 1] = 1.0;   // the fidelity column enters the linear kernel unscaled
    sp->cov.kind = g->kind;
    sp->cov.dim = D;
    sp->cov.amp = std::exp(theta[0]);
    sp->cov.blr_a = fab ? std::exp(theta[D]) : 0.0;
    sp->cov.blr_b = fab ? std::exp(theta[D + 1]) : 0.0;
    sp->noise = std::exp(theta[P - 1]) + JITTER;
    sp->mean_c = mean_c;
    return ROBO_OK;
}

static unsigned long long next_fit_gen() {
    static std::atomic<unsigned long long> gen{0};   // contexts of several devices fit on their own threads (multi.hip)
    return ++gen;
}

static FitBuffers own_buffers(robo_gp* g) {
    FitBuffers fb;
    fb.K = g->d_K; fb.k_stride = 0;
    fb.Linv = g->d_Linv; fb.linv_stride = 0;
    fb.Xs = g->d_Xs; fb.xs_stride = 0;
    fb.sp = g->d_sp;
    fb.fail = g->ctx->d_fail;
    fb.prog = g->ctx->d_prog;
    fb.out = g->ctx->d_scalars;
    fb.ll_part = g->d_llpart;
    fb.LinvP = g->d_LinvP;
    fb.host_out = g->ctx->h_pinned;
    fb.want_inverse = true;
    fb.skip_tail = false;
    fb.S = 1;
    return fb;
}

// stage theta, scale inputs, build the gram matrix into the GP's own buffers (asynchronous)
static int gp_build_gram(robo_gp* g, const double* theta, double mean_c) {
    robo_ctx* c = g->ctx;
    const int D = g->dim;
    ROBO_HIP_CHECK(hipSetDevice(c->device));
    ThetaArgs ta = {};
    ROBO_TRY(theta_to_sample(g, theta, mean_c, &ta.sp, ta.ism));
    g->cov = ta.sp.cov;
    g->amp = ta.sp.cov.amp;
    g->noise = ta.sp.noise;
    g->mean_c = mean_c;
    // theta travels as kernel arguments; block 0 of the scaling kernel leaves d_theta (metrics) and d_sp behind
    ROBO_TRY(launch_scale_inputs_theta(c, g->d_X, g->d_Xs, ta, g->n, g->n_pad, D, g->d_theta, g->d_sp));
    if (c->phase_events) ROBO_HIP_CHECK(hipEventRecord(c->events[19], c->stream));   // slot 19 -> 21: the gram kernel alone (K1)
    ROBO_TRY(launch_gram(g, own_buffers(g)));
    return ROBO_OK;
}

int32_t robo_gp_fit(robo_gp* g, const double* theta, double mean_c, double* out_loglik, int32_t* out_fail_col) {
    if (!g || !theta) return ROBO_BAD_ARGUMENT;
    if (!g->has_data) {
        set_error("robo_gp_fit before robo_gp_set_data");
        return ROBO_NOT_FITTED;
    }
    robo_ctx* c = g->ctx;
    g->fitted = false;
    // event slots 20..23: 20 -> 21 gram build, 21 -> 22 Cholesky, 22 -> 23 log-likelihood reduce
    if (c->phase_events) ROBO_HIP_CHECK(hipEventRecord(c->events[20], c->stream));
    ROBO_TRY(gp_build_gram(g, theta, mean_c));
    if (c->phase_events) ROBO_HIP_CHECK(hipEventRecord(c->events[21], c->stream));
    ROBO_TRY(launch_potrf(g, own_buffers(g)));
    if (c->phase_events) ROBO_HIP_CHECK(hipEventRecord(c->events[22], c->stream));
    if (c->phase_events) ROBO_HIP_CHECK(hipEventRecord(c->events[23], c->stream));
    // the tail kernel of the factorisation wrote (z.z, log det, failure flag) straight into the pinned buffer
    double* hp = c->h_pinned;
    ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));
    const int fail = (int)hp[2];
    if (fail < 0) {
        // a panel follower's bounded poll ran out (potrf.hip panel_follow): a workgroup of the step kernel never saw the
        // diagonal workgroup's progress -- not a property of the matrix
        if (out_loglik) *out_loglik = -HUGE_VAL;
        set_error("factorisation hand-off timed out (tuning potrf_follow=0 selects the launch-per-phase form)");
        return ROBO_RUNTIME_ERROR;
    }
    if (fail != 0) {
        if (out_fail_col) *out_fail_col = fail - 1;
        if (out_loglik) *out_loglik = -HUGE_VAL;
        set_error("matrix is not positive definite (column %d)", fail - 1);
        return ROBO_NOT_POSITIVE_DEFINITE;
    }
    const double quad = hp[0], logdet = hp[1];
    g->diag_min = hp[3];
    g->diag_max = hp[4];
    g->loglik = -0.5 * (quad + logdet + (double)g->n * std::log(2.0 * M_PI));
    g->fitted = true;
    g->fit_gen = next_fit_gen();
    if (out_loglik) *out_loglik = g->loglik;
    if (out_fail_col) *out_fail_col = -1;
    return ROBO_OK;
}

int32_t robo_gp_grad_loglik(robo_gp* g, const double* theta, double mean_c, double* out_loglik, double* out_grad,
                            int32_t* out_fail_col) {
    if (!g || !theta || !out_grad) return ROBO_BAD_ARGUMENT;
    ROBO_TRY(robo_gp_fit(g, theta, mean_c, out_loglik, out_fail_col));
    robo_ctx* c = g->ctx;
    const int P = robo_theta_size(g->kind, g->dim);
    if (!g->d_gV) ROBO_TRY(dev_alloc(&g->d_gV, (size_t)g->n_pad_max * g->n_pad_max));   // (shared with winv_ensure)
    if (!g->d_gA) {
        const size_t np = (size_t)g->n_pad_max, t64 = (np + 63) / 64;
        ROBO_TRY(dev_alloc(&g->d_gA, np * np));
        ROBO_TRY(dev_alloc(&g->d_galpha, np));
        ROBO_TRY(dev_alloc(&g->d_gpart, (size_t)P * (t64 * (t64 + 1) / 2)));
        ROBO_TRY(dev_alloc(&g->d_gout, (size_t)P));
    }
    // event slots 28 -> 29: everything after the factorisation (W^T, A, the reductions)
    ROBO_HIP_CHECK(hipEventRecord(c->events[28], c->stream));
    double* hg = c->h_pinned + 8;          // P <= MAX_DIM + 3 doubles behind the fit's three result slots
    ROBO_TRY(launch_grad_loglik(g, g->d_gV, g->d_gA, g->d_galpha, g->d_gpart, g->d_gout, hg));
    ROBO_HIP_CHECK(hipEventRecord(c->events[29], c->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));
    memcpy(out_grad, hg, (size_t)P * sizeof(double));
    return ROBO_OK;
}

// grow the batch workspace to hold S samples at the current n_pad
static int batch_ensure(robo_gp* g, int S) {
    if (g->b_cap >= S && g->b_npad == g->n_pad) return ROBO_OK;
    hipFree(g->d_bK); hipFree(g->d_bLinv); hipFree(g->d_bXs); hipFree(g->d_bout);
    hipFree(g->d_bsp); hipFree(g->d_bfail); hipFree(g->d_bllpart); hipFree(g->d_bkeep);   // d_bism lives in d_bsp's block
    hipFree(g->d_bprog);
    g->d_bprog = nullptr;
    if (g->h_bstage) hipHostFree(g->h_bstage);
    g->d_bK = g->d_bLinv = g->d_bXs = g->d_bism = g->d_bout = g->h_bstage = nullptr;
    g->d_bsp = nullptr; g->d_bfail = nullptr; g->d_bllpart = nullptr; g->d_bkeep = nullptr;
    g->b_cap = 0;
    const size_t np = (size_t)g->n_pad, D = (size_t)g->dim;
    ROBO_TRY(dev_alloc(&g->d_bK, (size_t)S * np * np));
    ROBO_TRY(dev_alloc(&g->d_bLinv, (size_t)S * np * NB));
    ROBO_HIP_CHECK(hipMemset(g->d_bLinv, 0, (size_t)S * np * NB * sizeof(double)));
    ROBO_TRY(dev_alloc(&g->d_bXs, (size_t)S * np * D));
    ROBO_TRY(dev_alloc(&g->d_bout, (size_t)S * 2));
    {   // [S x FitSample | S x D inverse sqrt metrics] in one block: one upload per pass
        char* blk = nullptr;
        ROBO_TRY(dev_alloc(&blk, (size_t)S * sizeof(FitSample) + (size_t)S * D * sizeof(double)));
        g->d_bsp = reinterpret_cast<FitSample*>(blk);
        g->d_bism = reinterpret_cast<double*>(g->d_bsp + S);
    }
    ROBO_TRY(dev_alloc(&g->d_bfail, (size_t)S));
    ROBO_TRY(dev_alloc(&g->d_bprog, (size_t)S * PROG_STRIDE));
    ROBO_TRY(dev_alloc(&g->d_bllpart, (size_t)S * (np / NB) * 4));
    ROBO_TRY(dev_alloc(&g->d_bkeep, (size_t)S * sizeof(KeepDst)));
    // pinned staging: [S x FitSample | S x D ism] up, [S x 5 doubles] down (written by the device)
    const size_t bytes = (size_t)S * (sizeof(FitSample) + D * sizeof(double) + 5 * sizeof(double)) + 64;
    ROBO_HIP_CHECK(hipHostMalloc((void**)&g->h_bstage, bytes, 0));
    g->b_cap = S;
    g->b_npad = g->n_pad;
    return ROBO_OK;
}

// S thetas on the training data of g, factorised by ONE sequence of launches per workspace chunk.  After each
// chunk `keep(s0, ns, status)` may copy the chunk's factors out of the strided batch workspace (it runs before
// the next chunk overwrites them).
static int fit_batch_core(robo_gp* g, const double* thetas, int32_t S, double mean_c, double* out_loglik,
                          int32_t* out_status, const std::function<int(int, int, const int*)>& keep) {
    robo_ctx* c = g->ctx;
    const int P = robo_theta_size(g->kind, g->dim), D = g->dim;
    const size_t np = (size_t)g->n_pad;
    ROBO_HIP_CHECK(hipSetDevice(c->device));
    // bound the workspace: sub-batches of at most `chunk` samples (S * n_pad^2 doubles each)
    size_t per = np * np * sizeof(double) + np * (NB + (size_t)D) * sizeof(double);
    int chunk = (int)(workspace_bytes(c) / per);
    if (chunk < 1) chunk = 1;
    if (chunk > S) chunk = S;
    ROBO_TRY(batch_ensure(g, chunk));
    for (int s0 = 0; s0 < S; s0 += chunk) {
        const int ns = S - s0 < chunk ? S - s0 : chunk;
        ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));   // staging buffer reuse
        const int cap = g->b_cap;                          // layout of the staging block and of its device twin
        FitSample* hsp = reinterpret_cast<FitSample*>(g->h_bstage);
        double* hism = reinterpret_cast<double*>(hsp + cap);
        double* hout = hism + (size_t)cap * D;             // [ns][5]: z.z, log det, failure flag, min / max L_ii
        std::vector<int> status(ns, ROBO_OK);
        for (int s = 0; s < ns; ++s) {
            const int st = theta_to_sample(g, thetas + (size_t)(s0 + s) * P, mean_c, hsp + s, hism + (size_t)s * D);
            status[s] = st;
            if (st != ROBO_OK) {   // keep the slot numerically harmless: unit kernel
                static const double zeros[MAX_DIM + 8] = {0};
                theta_to_sample(g, zeros, mean_c, hsp + s, hism + (size_t)s * D);
            }
        }
        ROBO_HIP_CHECK(hipMemcpyAsync(g->d_bsp, hsp, (size_t)cap * sizeof(FitSample) + (size_t)ns * D * sizeof(double),
                                      hipMemcpyHostToDevice, c->stream));
        FitBuffers fb;
        fb.K = g->d_bK; fb.k_stride = np * np;
        fb.prog = g->d_bprog;
        fb.Linv = g->d_bLinv; fb.linv_stride = np * NB;
        fb.Xs = g->d_bXs; fb.xs_stride = np * D;
        fb.sp = g->d_bsp;
        fb.fail = g->d_bfail;
        fb.out = g->d_bout;
        fb.ll_part = g->d_bllpart;
        fb.LinvP = nullptr;
        fb.host_out = hout;
        fb.want_inverse = (bool)keep;      // likelihoods only: the posterior's inverse blocks are not formed
        fb.skip_tail = false;
        fb.S = ns;
        ROBO_TRY(launch_scale_inputs(c, g->d_X, g->d_bXs, g->d_bism, g->n, g->n_pad, D, ns, np * D, (size_t)D));
        ROBO_TRY(launch_potrf(g, fb, true));   // gram + factorisation; its tail kernel also reduces the log-likelihood terms into fb.out
        ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));   // the finishing kernel wrote hout (pinned) itself
        for (int s = 0; s < ns; ++s) {
            double ll = -HUGE_VAL;
            if (status[s] == ROBO_OK) {
                if (hout[5 * s + 2] < 0.0) status[s] = ROBO_RUNTIME_ERROR;      // a follower's hand-off timed out (potrf.hip)
                else if (hout[5 * s + 2] != 0.0) status[s] = ROBO_NOT_POSITIVE_DEFINITE;
                else ll = -0.5 * (hout[5 * s] + hout[5 * s + 1] + (double)g->n * std::log(2.0 * M_PI));
            }
            out_loglik[s0 + s] = ll;
            if (out_status) out_status[s0 + s] = status[s];
        }
        if (keep) ROBO_TRY(keep(s0, ns, status.data()));
    }
    return ROBO_OK;
}

int32_t robo_gp_loglik_batch(robo_gp* g, const double* thetas, int32_t S, double mean_c, double* out_loglik,
                             int32_t* out_status) {
    if (!g || !thetas || S < 0 || !out_loglik) return ROBO_BAD_ARGUMENT;
    if (!g->has_data) {
        set_error("robo_gp_loglik_batch before robo_gp_set_data");
        return ROBO_NOT_FITTED;
    }
    if (S == 0) return ROBO_OK;
    g->fitted = false;   // the GP's own factor is not touched, but the call documents "unfitted after"
    return fit_batch_core(g, thetas, S, mean_c, out_loglik, out_status, nullptr);
}

// The whole ensemble chain on the device (mcmc.hip): no upload, synchronisation or host arithmetic between two half-steps.
int32_t robo_gp_mcmc_run(robo_gp* g, double mean_c, int32_t prior_kind, const double* prior_par, int32_t n_walkers,
                         int32_t n_steps, double a, const double* u_stretch, const int32_t* partner,
                         const double* u_accept, int32_t eval_start, double* pos, double* lnp, double* out_chain,
                         double* out_lnprob, int64_t* out_accepted) {
    if (!g || !pos || !lnp || n_walkers < 2 || (n_walkers & 1) || n_steps < 0 || !(a > 1.0)) return ROBO_BAD_ARGUMENT;
    if (n_steps > 0 && (!u_stretch || !partner || !u_accept)) return ROBO_BAD_ARGUMENT;
    if (prior_kind != 0 && ((prior_kind != 1 && prior_kind != 2) || !prior_par)) {
        set_error("robo_gp_mcmc_run: prior kind %d (0 = none, 1 = DefaultPrior, 2 = EnvPrior)", prior_kind);
        return ROBO_BAD_ARGUMENT;
    }
    if (prior_kind == 2) {
        const int P_ = robo_theta_size(g ? g->kind : 0, g ? g->dim : 1);
        const double n_ls = prior_par[5], n_lr = prior_par[6];
        if (!(n_ls >= 0 && n_lr >= 0 && n_ls == (double)(int)n_ls && n_lr == (double)(int)n_lr &&
              1 + (int)n_ls + (int)n_lr <= P_ - 1) || !(prior_par[8] > 0.0)) {
            set_error("robo_gp_mcmc_run: EnvPrior with n_ls=%g n_lr=%g sigma=%g does not fit %d hyper-parameters", n_ls,
                      n_lr, prior_par[8], P_);
            return ROBO_BAD_ARGUMENT;
        }
    }
    if (!g->has_data) {
        set_error("robo_gp_mcmc_run before robo_gp_set_data");
        return ROBO_NOT_FITTED;
    }
    robo_ctx* c = g->ctx;
    ROBO_HIP_CHECK(hipSetDevice(c->device));
    const int k = n_walkers, half = k / 2, D = g->dim, P = robo_theta_size(g->kind, D);
    const size_t np = (size_t)g->n_pad;
    {   // half an ensemble must fit the batch workspace in one pass (at the sizes of an MCMC fit it always does)
        const size_t per = np * np * sizeof(double) + np * (NB + (size_t)D) * sizeof(double);
        if ((size_t)half * per > workspace_bytes(c)) {
            set_error("robo_gp_mcmc_run: %d walkers of n_pad %zu exceed the batch workspace (ws_bytes)", half, np);
            return ROBO_BAD_SHAPE;
        }
    }
    g->fitted = false;
    ROBO_TRY(batch_ensure(g, half));
    // one scratch block: doubles first, then 64-bit counters, then ints
    const size_t nr = (size_t)n_steps * k;
    const size_t n_dbl = (size_t)k * P + k + (size_t)half * P + 2 * (size_t)half + 2 * nr + (size_t)k * n_steps * P + nr;
    const size_t bytes = n_dbl * sizeof(double) + (size_t)k * sizeof(long long) + (nr + 8) * sizeof(int);
    if (g->mcmc_bytes < bytes) {
        if (g->d_mcmc) ROBO_HIP_CHECK(hipFree(g->d_mcmc));
        g->d_mcmc = nullptr;
        g->mcmc_bytes = 0;
        ROBO_HIP_CHECK(hipMalloc((void**)&g->d_mcmc, bytes));
        g->mcmc_bytes = bytes;
    }
    McmcState st;
    memset(&st, 0, sizeof(st));
    double* d = reinterpret_cast<double*>(g->d_mcmc);
    st.d_pos = d; d += (size_t)k * P;
    st.d_lnp = d; d += k;
    st.d_q = d; d += (size_t)half * P;
    st.d_z = d; d += half;
    st.d_prior = d; d += half;
    double* d_uz = d; d += nr;
    double* d_ua = d; d += nr;
    st.d_chain = d; d += (size_t)k * n_steps * P;
    st.d_lnprob = d; d += nr;
    st.d_nacc = reinterpret_cast<long long*>(d);
    int* di = reinterpret_cast<int*>(st.d_nacc + k);   // [step counter, error flags, 6 spare | partners]
    st.d_it = di;
    st.d_err = di + 1;
    int* d_partner = di + 8;
    st.d_uz = d_uz; st.d_ua = d_ua; st.d_partner = d_partner;
    st.k = k; st.P = P; st.D = D; st.kind = g->kind; st.n = g->n; st.n_steps = n_steps; st.ns_eval = half;
    st.prior_kind = prior_kind; st.a = a; st.mean_c = mean_c;
    if (prior_kind != 0) for (int i = 0; i < (prior_kind == 2 ? 9 : 5); ++i) st.prior_par[i] = prior_par[i];
    st.d_sp = g->d_bsp; st.d_ism = g->d_bism; st.d_out = g->d_bout; st.d_fail = g->d_bfail;
    hipStream_t s = c->stream;
    ROBO_HIP_CHECK(hipMemcpyAsync(st.d_pos, pos, (size_t)k * P * sizeof(double), hipMemcpyHostToDevice, s));
    if (!eval_start) ROBO_HIP_CHECK(hipMemcpyAsync(st.d_lnp, lnp, (size_t)k * sizeof(double), hipMemcpyHostToDevice, s));
    if (nr > 0) {
        ROBO_HIP_CHECK(hipMemcpyAsync(d_uz, u_stretch, nr * sizeof(double), hipMemcpyHostToDevice, s));
        ROBO_HIP_CHECK(hipMemcpyAsync(d_ua, u_accept, nr * sizeof(double), hipMemcpyHostToDevice, s));
        ROBO_HIP_CHECK(hipMemcpyAsync(d_partner, partner, nr * sizeof(int), hipMemcpyHostToDevice, s));
    }
    ROBO_HIP_CHECK(hipMemsetAsync(st.d_nacc, 0, (size_t)k * sizeof(long long) + 8 * sizeof(int), s));
    FitBuffers fb;
    fb.K = g->d_bK; fb.k_stride = np * np;
    fb.prog = g->d_bprog;
    fb.Linv = g->d_bLinv; fb.linv_stride = np * NB;
    fb.Xs = g->d_bXs; fb.xs_stride = np * D;
    fb.sp = g->d_bsp;
    fb.fail = g->d_bfail;
    fb.out = g->d_bout;
    fb.ll_part = g->d_bllpart;
    fb.LinvP = nullptr;
    fb.host_out = nullptr;             // the likelihood terms are consumed on the device
    fb.want_inverse = false;
    fb.skip_tail = c->tune.mcmc_fused_tail != 0 && g->n_pad > NB;     // (one-block factors reduce inside their diagonal kernel)
    fb.S = half;
    // one-block problems: the whole half-step in one launch (potrf.hip mcmc_block_step_kernel; tuning: 0 never,
    // 1 only below 64 points, 2 = default: every one-block problem)
    const int bs = c->tune.mcmc_block_step;
    const bool one_block = (bs >= 2 ? g->n_pad == NB : (bs == 1 && g->n + 1 <= 64)) && !g->fp32_gram &&
                           g->kind != ROBO_KERNEL_FABOLAS;
    // two-block problems (128 <= N <= 254): likewise one launch, block row 1 through the batch workspace -- tuning value 3,
    // NOT the default: measured slower than the launch path (r06i: 122 vs 91 us per half-step at N = 200; potrf.hip)
    const bool two_block = bs >= 3 && g->n_pad == 2 * NB && g->n >= NB && !g->fp32_gram && g->kind != ROBO_KERNEL_FABOLAS;
    auto half_step = [&](int start, int first, int h, int it) -> int {
        if (one_block) return launch_mcmc_block_step(g, st, start, first, h, it);
        if (two_block) return launch_mcmc_block2_step(g, st, start, first, h, it, g->d_bK, np * np);
        ROBO_TRY(launch_mcmc_propose_scale(c, st, start, first, h, it, g->d_X, g->d_bXs, g->n, g->n_pad, np * D));
        ROBO_TRY(launch_potrf(g, fb, true));        // gram + factorisation
        if (fb.skip_tail) {                          // likelihood terms + accept test + chain record: one launch
            const int nbk = g->n_pad / NB, nbf = (g->n % NB == 0 && nbk > 1) ? nbk - 1 : nbk;
            return launch_mcmc_tail(c, st, start, first, h, it, g->d_bK, np * np, g->n_pad, nbf, g->d_bfail);
        }
        return launch_mcmc_accept(c, st, start, first, h, it);
    };
    if (eval_start) {
        ROBO_TRY(half_step(1, 0, 0, 0));
        ROBO_TRY(half_step(1, half, 0, 0));
    }
    for (int it = 0; it < n_steps; ++it) {
        ROBO_TRY(half_step(0, 0, 0, it));
        ROBO_TRY(half_step(0, 0, 1, it));
    }
    int herr[2] = {0, 0};
    std::vector<long long> hacc((size_t)k);
    ROBO_HIP_CHECK(hipMemcpyAsync(pos, st.d_pos, (size_t)k * P * sizeof(double), hipMemcpyDeviceToHost, s));
    ROBO_HIP_CHECK(hipMemcpyAsync(lnp, st.d_lnp, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, s));
    if (out_chain && nr > 0)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_chain, st.d_chain, (size_t)k * n_steps * P * sizeof(double), hipMemcpyDeviceToHost, s));
    if (out_lnprob && nr > 0)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_lnprob, st.d_lnprob, nr * sizeof(double), hipMemcpyDeviceToHost, s));
    ROBO_HIP_CHECK(hipMemcpyAsync(hacc.data(), st.d_nacc, (size_t)k * sizeof(long long), hipMemcpyDeviceToHost, s));
    ROBO_HIP_CHECK(hipMemcpyAsync(herr, st.d_it, sizeof(herr), hipMemcpyDeviceToHost, s));
    ROBO_HIP_CHECK(hipStreamSynchronize(s));
    if (out_accepted) for (int w = 0; w < k; ++w) out_accepted[w] = (int64_t)hacc[(size_t)w];
    if (herr[1] & 1) {
        set_error("lnprob returned NaN.");
        return ROBO_BAD_ARGUMENT;
    }
    if (herr[1] & 2) {
        set_error("The initial lnprob was +inf.");
        return ROBO_BAD_ARGUMENT;
    }
    if (herr[1] & 4) {
        set_error("factorisation hand-off timed out inside the chain (tuning potrf_batch_follow=0 selects the launch-per-phase form)");
        return ROBO_RUNTIME_ERROR;
    }
    return ROBO_OK;
}

int32_t robo_gp_fit_batch(robo_gp* const* gps, int32_t S, const double* thetas, double mean_c, double* out_loglik,
                          int32_t* out_status) {
    if (!gps || !thetas || S < 0 || !out_loglik || !out_status) return ROBO_BAD_ARGUMENT;
    if (S == 0) return ROBO_OK;
    robo_gp* g0 = gps[0];
    if (!g0) return ROBO_BAD_ARGUMENT;
    if (!g0->has_data) {
        set_error("robo_gp_fit_batch: gps[0] has no data (robo_gp_set_data)");
        return ROBO_NOT_FITTED;
    }
    robo_ctx* c = g0->ctx;
    const int D = g0->dim, n = g0->n;
    for (int s = 0; s < S; ++s) {
        robo_gp* g = gps[s];
        if (!g || g->ctx != c || g->kind != g0->kind || g->dim != D || g->n_max < n) {
            set_error("robo_gp_fit_batch: gps[%d] must share context, kernel kind and dim with gps[0] and hold n=%d rows",
                      s, n);
            return ROBO_BAD_SHAPE;
        }
        for (int t = 0; t < s; ++t)
            if (gps[t] == g) {
                set_error("robo_gp_fit_batch: gps[%d] and gps[%d] are the same handle", t, s);
                return ROBO_BAD_ARGUMENT;
            }
        g->fitted = false;
    }
    const size_t np = (size_t)g0->n_pad;
    const int P = robo_theta_size(g0->kind, D);
    auto keep = [&](int s0, int ns, const int* status) -> int {
        const double* hout = reinterpret_cast<const double*>(reinterpret_cast<const FitSample*>(g0->h_bstage) + g0->b_cap) +
                             (size_t)g0->b_cap * D;      // the batch's [ns][5] result block (fit_batch_core)
        // every kept factor goes to its handle in ONE launch (potrf.hip batch_keep_kernel)
        std::vector<KeepDst> dst((size_t)ns);
        for (int s = 0; s < ns; ++s) {
            KeepDst& d = dst[(size_t)s];
            memset(&d, 0, sizeof(d));
            if (status[s] != ROBO_OK) continue;
            robo_gp* g = gps[s0 + s];
            d.ok = 1;
            d.K = g->d_K; d.Linv = g->d_Linv; d.LinvP = g->d_LinvP; d.Xs = g->d_Xs; d.theta = g->d_theta; d.sp = g->d_sp;
            if (g != g0) {   // every handle ends up self-contained: same training data as gps[0]
                d.X = g->d_X;
                d.y = g->d_y;
            }
        }
        ROBO_HIP_CHECK(hipMemcpyAsync(g0->d_bkeep, dst.data(), (size_t)ns * sizeof(KeepDst), hipMemcpyHostToDevice, c->stream));
        ROBO_TRY(launch_batch_keep(g0, reinterpret_cast<const KeepDst*>(g0->d_bkeep), ns));
        for (int s = 0; s < ns; ++s) {
            if (status[s] != ROBO_OK) continue;
            robo_gp* g = gps[s0 + s];
            g->diag_min = hout[5 * s + 3];
            g->diag_max = hout[5 * s + 4];
            if (g != g0) {
                g->n = n;
                g->n_pad = g0->n_pad;
                g->has_data = true;
                g->fp32_gram = g0->fp32_gram;
            }
            FitSample sp;
            double ism[MAX_DIM];
            ROBO_TRY(theta_to_sample(g, thetas + (size_t)(s0 + s) * P, mean_c, &sp, ism));
            g->cov = sp.cov;
            g->amp = sp.cov.amp;
            g->noise = sp.noise;
            g->mean_c = mean_c;
            g->loglik = out_loglik[s0 + s];
            g->fitted = true;
            g->fit_gen = next_fit_gen();
        }
        ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));
        return ROBO_OK;
    };
    return fit_batch_core(g0, thetas, S, mean_c, out_loglik, out_status, keep);
}

int32_t robo_gp_get_factor(robo_gp* g, double* out_L) {
    if (!g || !out_L) return ROBO_BAD_ARGUMENT;
    if (!g->fitted) return ROBO_NOT_FITTED;
    ROBO_HIP_CHECK(hipSetDevice(g->ctx->device));     // (a process may drive several devices: multi.hip)
    const size_t np = (size_t)g->n_pad;
    std::vector<double> h(np * np);
    ROBO_HIP_CHECK(hipMemcpyAsync(h.data(), g->d_K, np * np * sizeof(double), hipMemcpyDeviceToHost, g->ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
    for (int i = 0; i < g->n; ++i)
        for (int j = 0; j < g->n; ++j) out_L[(size_t)i * g->n + j] = j <= i ? h[(size_t)i * np + j] : 0.0;
    return ROBO_OK;
}

int32_t robo_gp_get_gram(robo_gp* g, const double* theta, double* out_K) {
    if (!g || !theta || !out_K) return ROBO_BAD_ARGUMENT;
    if (!g->has_data) return ROBO_NOT_FITTED;
    g->fitted = false;   // d_K is overwritten
    ROBO_TRY(gp_build_gram(g, theta, 0.0));
    const size_t np = (size_t)g->n_pad;
    std::vector<double> h(np * np);
    ROBO_HIP_CHECK(hipMemcpyAsync(h.data(), g->d_K, np * np * sizeof(double), hipMemcpyDeviceToHost, g->ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
    for (int i = 0; i < g->n; ++i)
        for (int j = 0; j < g->n; ++j)
            out_K[(size_t)i * g->n + j] = j <= i ? h[(size_t)i * np + j] : h[(size_t)j * np + i];
    return ROBO_OK;
}

static bool winv_factor_ok(const robo_gp* g, int min_blocks);
static int winv_min_blocks_for(const robo_gp* g, long long m);

int32_t robo_gp_prefetch_inverse(robo_gp* g) {
    if (!g) return ROBO_BAD_ARGUMENT;
    if (!g->fitted) {
        set_error("Model has to be trained first!");
        return ROBO_NOT_FITTED;
    }
    // the same conditions under which a small batch would ask for W (winv_candidate), for the smallest batch that could
    // come (a handful of candidates: from three block rows on)
    if (!winv_factor_ok(g, winv_min_blocks_for(g, 1))) return ROBO_OK;
    // only for handles that HAVE served a small batch through W before (its buffers exist): a model that is only ever asked
    // for large batches never pays the two n_pad^2 buffers or the build
    if (!g->d_Winv) return ROBO_OK;
    ROBO_HIP_CHECK(hipSetDevice(g->ctx->device));
    return winv_launch(g);
}

int32_t robo_gp_factor_cond(robo_gp* g, double* out) {
    if (!g || !out) return ROBO_BAD_ARGUMENT;
    if (!g->fitted) {
        set_error("Model has to be trained first!");
        return ROBO_NOT_FITTED;
    }
    ROBO_HIP_CHECK(hipSetDevice(g->ctx->device));
    ROBO_TRY(winv_ensure(g));
    out[0] = g->winv_cond;
    out[1] = g->diag_min;
    out[2] = g->diag_max;
    return ROBO_OK;
}

// ---------------------------------------------------------------------------------------
// candidates
// ---------------------------------------------------------------------------------------
static int cand_alloc(robo_ctx* ctx, int64_t m, int32_t dim, robo_cand** out) {
    if (!ctx || !out) return ROBO_BAD_ARGUMENT;
    if (m < 1 || dim < 1 || dim > MAX_DIM) {
        set_error("bad candidate shape m=%lld dim=%d", (long long)m, dim);
        return ROBO_BAD_SHAPE;
    }
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    robo_cand* k = new robo_cand();
    memset(k, 0, sizeof(*k));
    k->ctx = ctx;
    k->dim = dim;
    k->m = m;
    k->m_pad = round_up64(m, NB);
    const size_t mp = (size_t)k->m_pad;
    ROBO_TRY(dev_alloc(&k->d_Xc, mp * dim));
    ROBO_TRY(dev_alloc(&k->d_Xcs, mp * dim));
    ROBO_TRY(dev_alloc(&k->d_q, mp));
    ROBO_TRY(dev_alloc(&k->d_mu, mp));
    ROBO_TRY(dev_alloc(&k->d_mean, mp));
    ROBO_TRY(dev_alloc(&k->d_var, mp));
    ROBO_TRY(dev_alloc(&k->d_acq, mp));
    ROBO_TRY(dev_alloc(&k->d_acq_sum, mp));
    k->n_part = (int)((m + 255) / 256);
    ROBO_TRY(dev_alloc(&k->d_part_val, (size_t)k->n_part + 1));
    ROBO_TRY(dev_alloc(&k->d_part_idx, (size_t)k->n_part + 1));
    ROBO_TRY(dev_alloc(&k->d_flags, 4));
    ROBO_HIP_CHECK(hipMemset(k->d_flags, 0, 4 * sizeof(unsigned)));   // cleared again by every read-back
    ctx_retain(ctx);
    *out = k;
    return ROBO_OK;
}

int32_t robo_cand_create(robo_ctx* ctx, const double* Xc, int64_t m, int32_t dim, robo_cand** out) {
    if (!Xc) return ROBO_BAD_ARGUMENT;
    robo_cand* k = nullptr;
    ROBO_TRY(cand_alloc(ctx, m, dim, &k));
    ROBO_HIP_CHECK(hipMemcpyAsync(k->d_Xc, Xc, (size_t)m * dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out = k;
    return ROBO_OK;
}

int32_t robo_cand_set_points(robo_cand* k, const double* Xc, int64_t m) {
    if (!k || !Xc) return ROBO_BAD_ARGUMENT;
    if (m != k->m) {
        set_error("robo_cand_set_points: batch holds %lld points, got %lld", (long long)k->m, (long long)m);
        return ROBO_BAD_SHAPE;
    }
    ROBO_HIP_CHECK(hipSetDevice(k->ctx->device));
    k->solved_gen = 0;
    ROBO_HIP_CHECK(hipMemcpyAsync(k->d_Xc, Xc, (size_t)m * k->dim * sizeof(double), hipMemcpyHostToDevice, k->ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(k->ctx->stream));   // the caller's buffer is only borrowed for the call
    return ROBO_OK;
}

int32_t robo_cand_create_uniform(robo_ctx* ctx, int64_t m, int32_t dim, uint64_t seed, robo_cand** out) {
    robo_cand* k = nullptr;
    ROBO_TRY(cand_alloc(ctx, m, dim, &k));
    ROBO_TRY(launch_uniform(ctx, k->d_Xc, m, k->m_pad, dim, seed));
    ROBO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out = k;
    return ROBO_OK;
}

int32_t robo_cand_create_sobol(robo_ctx* ctx, int64_t m, int32_t dim, const uint64_t* sv, const uint64_t* shift,
                               int32_t bits, uint64_t first_index, robo_cand** out) {
    if (!sv || !shift || bits < 1 || bits > 64) return ROBO_BAD_ARGUMENT;
    if (bits < 64 && (first_index + (uint64_t)m) > (1ull << bits)) {
        set_error("Sobol: points %llu .. %llu exceed 2^%d", (unsigned long long)first_index,
                  (unsigned long long)(first_index + (uint64_t)m), bits);
        return ROBO_BAD_SHAPE;
    }
    robo_cand* k = nullptr;
    ROBO_TRY(cand_alloc(ctx, m, dim, &k));
    // the direction numbers (dim x bits) and the digital shift (dim) ride in the still unused scaled-candidate buffer
    if ((size_t)dim * bits + dim > (size_t)k->m_pad * dim) {
        robo_cand_destroy(k);
        set_error("Sobol: batch too small to stage the direction numbers (m_pad %lld < bits + 1)", (long long)k->m_pad);
        return ROBO_BAD_SHAPE;
    }
    unsigned long long* d_sv = reinterpret_cast<unsigned long long*>(k->d_Xcs);
    unsigned long long* d_shift = d_sv + (size_t)dim * bits;
    int st = ROBO_OK;
    hipError_t e = hipMemcpyAsync(d_sv, sv, (size_t)dim * bits * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_shift, shift, (size_t)dim * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        set_error("Sobol: upload of the direction numbers failed: %s", hipGetErrorString(e));
        st = ROBO_RUNTIME_ERROR;
    }
    if (st == ROBO_OK) st = launch_sobol(ctx, k->d_Xc, m, k->m_pad, dim, d_sv, d_shift, bits, first_index);
    if (st == ROBO_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = ROBO_RUNTIME_ERROR;
    if (st != ROBO_OK) {
        robo_cand_destroy(k);
        return st;
    }
    *out = k;
    return ROBO_OK;
}

int32_t robo_cand_create_random(robo_ctx* ctx, int64_t m, int32_t dim, uint64_t seed, int64_t n_uniform,
                                const double* loc, const double* scale, robo_cand** out) {
    if (!loc || !scale || n_uniform < 0 || n_uniform > m) return ROBO_BAD_ARGUMENT;
    robo_cand* k = nullptr;
    ROBO_TRY(cand_alloc(ctx, m, dim, &k));
    // loc/scale ride in the (still unused) scaled-candidate buffer
    ROBO_HIP_CHECK(hipMemcpyAsync(k->d_Xcs, loc, (size_t)dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROBO_HIP_CHECK(hipMemcpyAsync(k->d_Xcs + dim, scale, (size_t)dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROBO_TRY(launch_random_candidates(ctx, k->d_Xc, k->m_pad, dim, seed, n_uniform, k->d_Xcs, k->d_Xcs + dim));
    ROBO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out = k;
    return ROBO_OK;
}

int32_t robo_cand_get_point(robo_cand* k, int64_t index, double* out_x) {
    if (!k || !out_x) return ROBO_BAD_ARGUMENT;
    if (index < 0 || index >= k->m) {
        set_error("candidate index %lld outside [0, %lld)", (long long)index, (long long)k->m);
        return ROBO_BAD_SHAPE;
    }
    ROBO_HIP_CHECK(hipSetDevice(k->ctx->device));
    ROBO_HIP_CHECK(hipMemcpyAsync(out_x, k->d_Xc + (size_t)index * k->dim, (size_t)k->dim * sizeof(double),
                                  hipMemcpyDeviceToHost, k->ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(k->ctx->stream));
    return ROBO_OK;
}

int32_t robo_cand_workspace_chunk(robo_cand* k, int64_t* out_chunk) {
    if (!k || !out_chunk) return ROBO_BAD_ARGUMENT;
    *out_chunk = k->chunk;
    return ROBO_OK;
}

int32_t robo_cand_last_solve_kernel(robo_cand* k, char* buf, int32_t buf_len) {
    if (!k || !buf || buf_len < 1) return ROBO_BAD_ARGUMENT;
    snprintf(buf, (size_t)buf_len, "%s", k->solve_kernel ? k->solve_kernel : "");
    return ROBO_OK;
}

int32_t robo_cand_get_points(robo_cand* k, double* out_Xc) {
    if (!k || !out_Xc) return ROBO_BAD_ARGUMENT;
    ROBO_HIP_CHECK(hipSetDevice(k->ctx->device));
    ROBO_HIP_CHECK(hipMemcpyAsync(out_Xc, k->d_Xc, (size_t)k->m * k->dim * sizeof(double), hipMemcpyDeviceToHost,
                                  k->ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(k->ctx->stream));
    return ROBO_OK;
}

int32_t robo_cand_destroy(robo_cand* k) {
    if (!k) return ROBO_OK;
    hipSetDevice(k->ctx->device);
    hipStreamSynchronize(k->ctx->stream);
    hipFree(k->d_Xc);
    hipFree(k->d_Xcs);
    hipFree(k->d_V);
    hipFree(k->d_q);
    hipFree(k->d_mu);
    hipFree(k->d_mean);
    hipFree(k->d_var);
    hipFree(k->d_acq);
    hipFree(k->d_acq_sum);
    hipFree(k->d_mu_all);
    hipFree(k->d_var_all);
    hipFree(k->d_S);
    hipFree(k->d_F);
    hipFree(k->d_Q);
    hipFree(k->d_G);
    hipFree(k->d_igc);
    free(k->h_igkey);
    hipFree(k->d_Ks);
    hipFree(k->d_P);
    hipFree(k->d_qpart);
    hipFree(k->d_part_val);
    hipFree(k->d_part_idx);
    hipFree(k->d_flags);
    robo_ctx* ctx = k->ctx;
    delete k;
    ctx_release(ctx);
    return ROBO_OK;
}

// size the (chunk x n_pad) solve workspace for this GP; grows, never shrinks
static int cand_ensure_workspace(robo_cand* k, int n_pad, bool single_chunk) {
    const size_t row = (size_t)n_pad * sizeof(double);
    int64_t chunk = (int64_t)(workspace_bytes(k->ctx) / row) / NB * NB;
    if (chunk < NB) chunk = NB;
    if (chunk > k->m_pad || single_chunk) chunk = k->m_pad;
    const size_t need = (size_t)chunk * row;
    if (need > k->v_bytes) {
        if (k->d_V) ROBO_HIP_CHECK(hipFree(k->d_V));
        k->d_V = nullptr;
        k->v_bytes = 0;
        ROBO_HIP_CHECK(hipMalloc((void**)&k->d_V, need));
        k->v_bytes = need;
    }
    k->chunk = chunk;
    k->ldv = n_pad;
    return ROBO_OK;
}

// K4 + K5: fills cand->d_mean / d_var (asynchronous).  after_chunk(c0, cn), if given, runs while
// the chunk's V = L^-1 K*^T is still in the workspace (cross-covariances for entropy search).
// Small batches on a well-conditioned factor go through the explicit inverse W = L^-1 (winv.hip): one triangular
// product instead of n / 128 dependent block-row launches.  W's forward error is ~eps cond(L) where the substitution's
// is ~eps cond of a 128-block, so the path is taken only while cond_inf(L) = |L|_inf |W|_inf -- EXACT, two row-sum
// reductions when W is built (winv_ensure) -- stays below winv_cond_max (default 1e5: the measured error of the mean through W is <= ~10 eps cond, i.e.
// <= 1.1e-10 = the stated absolute tolerance of the mean; tests/parity_checks.py check_winv_guard_sweep); beyond it the
// substitution stays.  The diagonal ratio
// max L_ii / min L_ii <= cond_2(L) is only the cheap pre-filter that avoids building a W that would be rejected.
// the factor-side half of the decision (everything but the batch size): shared by winv_candidate and
// robo_gp_prefetch_inverse, so that a prefetch is launched exactly for the factors a small batch would use W on.
// min_blocks: 3 for a handful of candidates (matrix-vector form), winv_min_blocks otherwise
static bool winv_factor_ok(const robo_gp* g, int min_blocks) {
    const Tuning& t = g->ctx->tune;
    if (g->fp32_gram || t.predict_stepwise || t.winv_max <= 0) return false;
    if ((g->n + NB - 1) / NB < min_blocks) return false;
    return g->diag_min > 0.0 && g->diag_max <= (double)t.winv_cond_max * g->diag_min;
}

static int winv_min_blocks_for(const robo_gp* g, long long m) {
    const Tuning& t = g->ctx->tune;
    // a handful of candidates (the matrix-vector form, winv.hip) pays from three block rows on: N = 300 0.044 vs 0.065 ms,
    // N = 500 0.053 vs 0.086 ms against the 32-candidate substitution (r04x); larger batches from winv_min_blocks on
    return (m <= 8 && t.winv_gemv != 0 && t.winv_min_blocks > 3) ? 3 : t.winv_min_blocks;
}

static bool winv_candidate(const robo_gp* g, const robo_cand* k) {
    if (k->m_pad > g->ctx->tune.winv_max) return false;
    return winv_factor_ok(g, winv_min_blocks_for(g, k->m));
}

static int decide_winv(robo_gp* g, const robo_cand* k, bool* use) {
    *use = false;
    if (!winv_candidate(g, k)) return ROBO_OK;
    ROBO_TRY(winv_ensure(g));                  // builds W for this factor if needed and measures cond_inf(L)
    *use = g->winv_cond > 0.0 && g->winv_cond <= (double)g->ctx->tune.winv_cond_max;
    return ROBO_OK;
}

// need_v: the caller consumes V = L^-1 K_*^T itself (cross-covariances, full covariance), not only its reductions
static int predict_core(robo_gp* g, robo_cand* k, bool single_chunk,
                        const std::function<int(int64_t, int64_t)>& after_chunk = nullptr, bool need_v = false) {
    if (!g || !k) return ROBO_BAD_ARGUMENT;
    if (!g->fitted) {
        set_error("Model has to be trained first!");
        return ROBO_NOT_FITTED;
    }
    if (k->dim != g->dim || k->ctx != g->ctx) {
        set_error("candidate batch (dim %d) does not match the GP (dim %d) or lives on another context", k->dim,
                  g->dim);
        return ROBO_BAD_SHAPE;
    }
    ROBO_HIP_CHECK(hipSetDevice(g->ctx->device));
    k->solved_gen = 0;
    ROBO_TRY(cand_ensure_workspace(k, g->n_pad, single_chunk));
    bool winv = false;
    ROBO_TRY(decide_winv(g, k, &winv));
    ROBO_TRY(launch_scale_inputs(g->ctx, k->d_Xc, k->d_Xcs, g->d_theta, k->m, k->m_pad, g->dim));
    // event slots 24..27 bracket the phases of the LAST chunk (bench.py reads them after a sync):
    //   24 -> 25 cross-gram, 25 -> 26 triangular solve (the MFMA kernel), 26 -> 27 post
    // (small batches are latency-bound and four event packets cost several microseconds: recorded for them only when
    // the phase events are switched on, robo_ctx_set_phase_events)
    hipStream_t st = g->ctx->stream;
    const bool ev = g->ctx->phase_events || k->m_pad > 16384;
    for (int64_t c0 = 0; c0 < k->m_pad; c0 += k->chunk) {
        const int64_t cn = k->m_pad - c0 < k->chunk ? k->m_pad - c0 : k->chunk;
        if (ev) ROBO_HIP_CHECK(hipEventRecord(g->ctx->events[24], st));
        if (winv) {
            if (ev) ROBO_HIP_CHECK(hipEventRecord(g->ctx->events[25], st));
            ROBO_TRY(launch_predict_winv(g, k, c0, cn, need_v || (bool)after_chunk));
        } else if (g->fp32_gram || g->ctx->tune.predict_stepwise) {
            // mixed precision (fp32 covariance entries) keeps the two-pass form
            ROBO_TRY(launch_cross_gram(g, k, c0, cn));
            if (ev) ROBO_HIP_CHECK(hipEventRecord(g->ctx->events[25], st));
            ROBO_TRY(launch_trsm(g, k, c0, cn));
        } else {
            if (ev) ROBO_HIP_CHECK(hipEventRecord(g->ctx->events[25], st));
            ROBO_TRY(launch_predict_fused(g, k, c0, cn));
        }
        if (ev) ROBO_HIP_CHECK(hipEventRecord(g->ctx->events[26], st));
        if (after_chunk) ROBO_TRY(after_chunk(c0, cn));
    }
    ROBO_TRY(launch_post(g, k, 0, k->m_pad));
    if (ev) ROBO_HIP_CHECK(hipEventRecord(g->ctx->events[27], st));
    return ROBO_OK;
}

int32_t robo_gp_predict_cand(robo_gp* g, robo_cand* k, double* out_mean, double* out_var) {
    ROBO_TRY(predict_core(g, k, false));
    hipStream_t st = g->ctx->stream;
    if (out_mean)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_mean, k->d_mean, (size_t)k->m * sizeof(double), hipMemcpyDeviceToHost, st));
    if (out_var)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_var, k->d_var, (size_t)k->m * sizeof(double), hipMemcpyDeviceToHost, st));
    ROBO_HIP_CHECK(hipStreamSynchronize(st));
    return ROBO_OK;
}

// per-sample posteriors of S fitted GPs on one candidate handle -> rows 0 .. S - 1 of the handle's (>= cap) x m_pad
// sample tables d_mu_all / d_var_all (asynchronous)
static int predict_samples(robo_gp* const* gps, int32_t S, robo_cand* k, int cap) {
    ROBO_HIP_CHECK(hipSetDevice(k->ctx->device));
    const size_t mp = (size_t)k->m_pad;
    if (cap < S) cap = S;
    if (k->s_cap < cap) {
        if (k->d_mu_all) ROBO_HIP_CHECK(hipFree(k->d_mu_all));
        if (k->d_var_all) ROBO_HIP_CHECK(hipFree(k->d_var_all));
        k->d_mu_all = k->d_var_all = nullptr;
        k->s_cap = 0;
        ROBO_TRY(dev_alloc(&k->d_mu_all, (size_t)cap * mp));
        ROBO_TRY(dev_alloc(&k->d_var_all, (size_t)cap * mp));
        k->s_cap = cap;
    }
    hipStream_t st = k->ctx->stream;
    for (int s = 0; s < S; ++s) {
        ROBO_TRY(predict_core(gps[s], k, false));
        ROBO_HIP_CHECK(hipMemcpyAsync(k->d_mu_all + (size_t)s * mp, k->d_mean, mp * sizeof(double), hipMemcpyDeviceToDevice, st));
        ROBO_HIP_CHECK(hipMemcpyAsync(k->d_var_all + (size_t)s * mp, k->d_var, mp * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    return ROBO_OK;
}

int32_t robo_gp_predict_mixture_cand(robo_gp* const* gps, int32_t S, robo_cand* k, double* out_mean,
                                     double* out_var) {
    if (!gps || S < 1 || !k) return ROBO_BAD_ARGUMENT;
    ROBO_TRY(predict_samples(gps, S, k, S));
    hipStream_t st = k->ctx->stream;
    ROBO_TRY(launch_mixture(k, S));
    if (out_mean)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_mean, k->d_mean, (size_t)k->m * sizeof(double), hipMemcpyDeviceToHost, st));
    if (out_var)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_var, k->d_var, (size_t)k->m * sizeof(double), hipMemcpyDeviceToHost, st));
    ROBO_HIP_CHECK(hipStreamSynchronize(st));
    return ROBO_OK;
}

// The candidate handle behind the host-array entry points (robo_gp_predict, robo_acq_eval).  Small batches -- the
// reference's 500 random candidates per iteration, the 1 x D calls of its single-point maximisers -- come back with
// the same size over and over: their handle (a dozen device allocations) is kept with the GP and only re-uploaded
// (0.14 -> ~0.08 ms per call at N <= 100).  Larger batches are created and destroyed per call as before.
static int host_cand(robo_gp* g, const double* Xc, int64_t m, robo_cand** out, bool* kept) {
    *kept = m <= 16384;
    if (!*kept) return robo_cand_create(g->ctx, Xc, m, g->dim, out);
    if (g->host_cand && g->host_cand->m == m) {
        ROBO_TRY(robo_cand_set_points(g->host_cand, Xc, m));
    } else {
        robo_cand_destroy(g->host_cand);
        g->host_cand = nullptr;
        ROBO_TRY(robo_cand_create(g->ctx, Xc, m, g->dim, &g->host_cand));
    }
    *out = g->host_cand;
    return ROBO_OK;
}

int32_t robo_gp_predict(robo_gp* g, const double* Xc, int64_t m, double* out_mean, double* out_var) {
    if (!g) return ROBO_BAD_ARGUMENT;
    if (!g->fitted) {
        set_error("Model has to be trained first!");
        return ROBO_NOT_FITTED;
    }
    robo_cand* k = nullptr;
    bool kept = false;
    ROBO_TRY(host_cand(g, Xc, m, &k, &kept));
    const int st = robo_gp_predict_cand(g, k, out_mean, out_var);
    if (!kept) robo_cand_destroy(k);
    return st;
}

int32_t robo_gp_predict_grad(robo_gp* g, const double* Xc, int64_t m, double* out_mean, double* out_var,
                             double* out_dmean, double* out_dvar) {
    if (!g || !Xc || !out_dmean || !out_dvar) return ROBO_BAD_ARGUMENT;
    if (!g->fitted) {
        set_error("Model has to be trained first!");
        return ROBO_NOT_FITTED;
    }
    const int D = g->dim, E = D + 1;
    robo_ctx* c = g->ctx;
    hipStream_t st = c->stream;
    robo_cand* kc = nullptr;   // the real candidates (upload + scaling + output buffers)
    ROBO_TRY(robo_cand_create(c, Xc, m, D, &kc));
    // candidates per pass: D + 1 workspace rows each, rows padded to the 128-row solve tile
    const size_t row_bytes = (size_t)g->n_pad * sizeof(double);
    int64_t per = (int64_t)(workspace_bytes(c) / row_bytes / NB * NB) / E;
    if (per < 1) per = 1;
    if (per > m) per = m;
    // cross_grad_kernel and predgrad_post_kernel put (a multiple of) the candidate index on grid.y / grid.x
    if (per + NB > 65535) per = 65535 - NB;
    const int64_t rows_pad = round_up64(per * E, NB);
    robo_cand* ws = nullptr;   // the pseudo-row solve workspace
    double *d_dm = nullptr, *d_dv = nullptr;
    int status = cand_alloc(c, rows_pad, 1, &ws);
    if (status == ROBO_OK) status = cand_ensure_workspace(ws, g->n_pad, true);
    if (status == ROBO_OK) status = dev_alloc(&d_dm, (size_t)m * D);
    if (status == ROBO_OK) status = dev_alloc(&d_dv, (size_t)m * D);
    if (status == ROBO_OK) status = launch_scale_inputs(c, kc->d_Xc, kc->d_Xcs, g->d_theta, kc->m, kc->m_pad, D);
    for (int64_t c0 = 0; status == ROBO_OK && c0 < m; c0 += per) {
        const int64_t cn = m - c0 < per ? m - c0 : per;
        const int64_t rp = round_up64(cn * E, NB);
        status = launch_cross_grad(g, kc->d_Xcs, ws->d_V, c0, cn, rp);
        if (status == ROBO_OK) status = launch_trsm(g, ws, 0, rp);
        if (status == ROBO_OK)
            status = launch_predgrad_post(g, ws->d_V, ws->d_q, ws->d_mu, kc->d_Xcs, c0, cn, kc->d_mean, kc->d_var, d_dm, d_dv);
    }
    if (status == ROBO_OK) {
        hipError_t e = hipSuccess;
        if (out_mean) e = hipMemcpyAsync(out_mean, kc->d_mean, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && out_var)
            e = hipMemcpyAsync(out_var, kc->d_var, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(out_dmean, d_dm, (size_t)m * D * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(out_dvar, d_dv, (size_t)m * D * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            set_error("robo_gp_predict_grad copy-out failed: %s", hipGetErrorString(e));
            status = ROBO_RUNTIME_ERROR;
        }
    } else {
        hipStreamSynchronize(st);
    }
    hipFree(d_dm);
    hipFree(d_dv);
    robo_cand_destroy(ws);
    robo_cand_destroy(kc);
    return status;
}

int32_t robo_gp_predict_cov(robo_gp* g, const double* Xc, int64_t m, double* out_mean, double* out_cov) {
    if (!g || !out_cov) return ROBO_BAD_ARGUMENT;
    if (!g->fitted) {
        set_error("Model has to be trained first!");
        return ROBO_NOT_FITTED;
    }
    if (m > 16384) {
        set_error("robo_gp_predict_cov is for small batches (m=%lld > 16384)", (long long)m);
        return ROBO_BAD_SHAPE;
    }
    robo_cand* k = nullptr;
    ROBO_TRY(robo_cand_create(g->ctx, Xc, m, g->dim, &k));
    int st = predict_core(g, k, true, nullptr, true);
    double* d_cov = nullptr;
    if (st == ROBO_OK && hipMalloc((void**)&d_cov, (size_t)m * m * sizeof(double)) != hipSuccess) {
        set_error("hipMalloc of the %lld x %lld covariance failed", (long long)m, (long long)m);
        st = ROBO_RUNTIME_ERROR;
    }
    if (st == ROBO_OK) st = launch_cov(g, k, d_cov);
    if (st == ROBO_OK) {
        hipStream_t s = g->ctx->stream;
        hipError_t e = hipMemcpyAsync(out_cov, d_cov, (size_t)m * m * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && out_mean)
            e = hipMemcpyAsync(out_mean, k->d_mean, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            set_error("predict_cov copy-out failed: %s", hipGetErrorString(e));
            st = ROBO_RUNTIME_ERROR;
        }
    }
    if (d_cov) hipFree(d_cov);
    robo_cand_destroy(k);
    return st;
}

// ---------------------------------------------------------------------------------------
// acquisition
// ---------------------------------------------------------------------------------------
static int check_acq_kind(int kind) {
    if (kind < ROBO_ACQ_EI || kind > ROBO_ACQ_LCB) {
        set_error("unknown acquisition kind %d", kind);
        return ROBO_BAD_ARGUMENT;
    }
    return ROBO_OK;
}

// The flag word of a candidate handle is OR-ed into by the acquisition kernels and cleared by the read-back's report
// kernel.  An error return between the two (a later sample's posterior failing, a launch failure) would leave stale
// ZERO_SIGMA / NEGATIVE_EI bits in a handle that lives on (kept host-array handles, representer points): clear them.
static int clear_flags_on_error(robo_cand* k, int status) {
    if (status != ROBO_OK && k && k->d_flags) hipMemsetAsync(k->d_flags, 0, 4 * sizeof(unsigned), k->ctx->stream);
    return status;
}

// D2H of (max, argmax, flags) [+ the acquisition vector] and the one synchronisation of the call
static int acq_read_back(robo_cand* k, const double* d_vec, double* out_vec, double* out_max, int64_t* out_argmax,
                         uint32_t* out_flags) {
    robo_ctx* c = k->ctx;
    double* hp = c->h_pinned;
    ROBO_TRY(launch_report_best(k, hp));   // (max, argmax, flags) -> pinned memory; clears the flag word
    if (out_vec)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_vec, d_vec, (size_t)k->m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (out_max) *out_max = hp[0];
    if (out_argmax) {
        long long i;
        memcpy(&i, hp + 1, sizeof(i));
        *out_argmax = (int64_t)i;
    }
    if (out_flags) {
        unsigned f;
        memcpy(&f, hp + 2, sizeof(f));
        *out_flags = f;
    }
    return ROBO_OK;
}

int32_t robo_acq_eval_cand(robo_gp* g, int32_t acq_kind, double par, double eta, robo_cand* k, double* out_acq,
                           double* out_max, int64_t* out_argmax, uint32_t* out_flags) {
    ROBO_TRY(check_acq_kind(acq_kind));
    ROBO_TRY(predict_core(g, k, false));
    ROBO_TRY(clear_flags_on_error(k, launch_acq(g->ctx, k, acq_kind, par, eta, false, false)));
    return clear_flags_on_error(k, acq_read_back(k, k->d_acq, out_acq, out_max, out_argmax, out_flags));
}

int32_t robo_acq_eval(robo_gp* g, int32_t acq_kind, double par, double eta, const double* Xc, int64_t m,
                      double* out_acq, double* out_max, int64_t* out_argmax, uint32_t* out_flags) {
    if (!g) return ROBO_BAD_ARGUMENT;
    if (!g->fitted) {
        set_error("Model has to be trained first!");
        return ROBO_NOT_FITTED;
    }
    robo_cand* k = nullptr;
    bool kept = false;
    ROBO_TRY(host_cand(g, Xc, m, &k, &kept));
    const int st = robo_acq_eval_cand(g, acq_kind, par, eta, k, out_acq, out_max, out_argmax, out_flags);
    if (!kept) robo_cand_destroy(k);
    return st;
}

static int acq_accumulate(robo_gp* const* gps, int32_t S, int32_t acq_kind, double par, const double* etas,
                          robo_cand* k) {
    if (!gps || S < 1 || !k || !etas) return ROBO_BAD_ARGUMENT;
    ROBO_TRY(check_acq_kind(acq_kind));
    ROBO_HIP_CHECK(hipSetDevice(k->ctx->device));
    for (int s = 0; s < S; ++s) {
        ROBO_TRY(clear_flags_on_error(k, predict_core(gps[s], k, false)));
        ROBO_TRY(clear_flags_on_error(k, launch_acq(k->ctx, k, acq_kind, par, etas[s], true, s == 0)));
    }
    return ROBO_OK;
}

int32_t robo_acq_eval_marginal_cand(robo_gp* const* gps, int32_t S, int32_t acq_kind, double par, const double* etas,
                                    robo_cand* k, double* out_acq, double* out_max, int64_t* out_argmax,
                                    uint32_t* out_flags) {
    ROBO_TRY(acq_accumulate(gps, S, acq_kind, par, etas, k));
    ROBO_TRY(clear_flags_on_error(k, launch_argmax(k, k->d_acq_sum, (double)S)));
    return clear_flags_on_error(k, acq_read_back(k, k->d_acq, out_acq, out_max, out_argmax, out_flags));
}

int32_t robo_acq_eval_sum_cand(robo_gp* const* gps, int32_t S, int32_t acq_kind, double par, const double* etas,
                               robo_cand* k, double* out_acq_sum, uint32_t* out_flags) {
    ROBO_TRY(acq_accumulate(gps, S, acq_kind, par, etas, k));
    ROBO_TRY(clear_flags_on_error(k, launch_argmax(k, k->d_acq_sum, 1.0)));
    return clear_flags_on_error(k, acq_read_back(k, k->d_acq_sum, out_acq_sum, nullptr, nullptr, out_flags));
}

int32_t robo_acq_eval_moments(robo_ctx* ctx, int32_t acq_kind, double par, double eta, const double* mean,
                              const double* var, int64_t m, double* out_acq, double* out_max, int64_t* out_argmax,
                              uint32_t* out_flags) {
    if (!ctx || !mean || !var) return ROBO_BAD_ARGUMENT;
    ROBO_TRY(check_acq_kind(acq_kind));
    robo_cand* k = nullptr;
    ROBO_TRY(cand_alloc(ctx, m, 1, &k));
    int st = ROBO_OK;
    hipError_t e = hipMemcpyAsync(k->d_mean, mean, (size_t)m * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(k->d_var, var, (size_t)m * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        set_error("robo_acq_eval_moments upload failed: %s", hipGetErrorString(e));
        st = ROBO_RUNTIME_ERROR;
    }
    if (st == ROBO_OK) st = launch_acq(ctx, k, acq_kind, par, eta, false, false);
    if (st == ROBO_OK) st = acq_read_back(k, k->d_acq, out_acq, out_max, out_argmax, out_flags);
    robo_cand_destroy(k);
    return st;
}

// ---------------------------------------------------------------------------------------
// entropy search: information gain of a candidate batch
// ---------------------------------------------------------------------------------------
static int ig_ensure(robo_cand* k, int kf) {
    const size_t mp = (size_t)k->m_pad;
    if (!k->d_S) ROBO_TRY(dev_alloc(&k->d_S, mp * NB));
    const size_t need_f = (size_t)k->chunk * kf;
    if (k->f_cap < need_f) {
        hipFree(k->d_F);
        k->d_F = nullptr;
        k->f_cap = 0;
        ROBO_TRY(dev_alloc(&k->d_F, need_f));
        k->f_cap = need_f;
    }
    if (k->q_cap < (size_t)k->chunk) {
        hipFree(k->d_Q);
        k->d_Q = nullptr;
        ROBO_TRY(dev_alloc(&k->d_Q, (size_t)k->chunk * NB));
        k->q_cap = (size_t)k->chunk;
    }
    if (k->g_cap < (size_t)kf) {
        hipFree(k->d_G);
        hipFree(k->d_igc);
        k->d_G = k->d_igc = nullptr;
        if (k->h_igkey) k->h_igkey[0] = -1.0;      // the device copies are gone: no cached EP state
        ROBO_TRY(dev_alloc(&k->d_G, (size_t)NB * kf));
        ROBO_TRY(dev_alloc(&k->d_igc, (size_t)128 + 512 + 64 * 64));
        k->g_cap = (size_t)kf;
    }
    return ROBO_OK;
}

// upload the EP state: consts = [logP (64) | lmb (64) | W (npts) | dlogPdMu (nb x nb)], G (128 x kf)
static int ig_upload(robo_cand* k, int nb, int npts, int kf, const double* logP, const double* lmb, const double* W,
                     const double* dlogPdMu, const double* dlogPdSigma, const double* dlogPdMudMu) {
    // The EP state changes once per update() of the acquisition function and is then evaluated on batch after batch:
    // the 2.5 MB re-layout + upload + synchronisation below is skipped when all six arrays equal, bit for bit, the
    // ones this handle's device copies were made from (0.1 ms of memcmp instead of ~0.5 ms per call at Nb = 50).
    const int ntri_k = nb * (nb + 1) / 2;
    const size_t lens[6] = {(size_t)nb, (size_t)nb, (size_t)npts, (size_t)nb * nb, (size_t)nb * ntri_k,
                            (size_t)nb * nb * nb};
    const double* srcs[6] = {logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu};
    size_t total = 2;
    for (size_t l : lens) total += l;
    if (k->h_igkey && k->igkey_len == total && k->h_igkey[0] == (double)nb && k->h_igkey[1] == (double)npts) {
        bool same = true;
        size_t off = 2;
        for (int a = 0; a < 6 && same; ++a) {
            same = memcmp(k->h_igkey + off, srcs[a], lens[a] * sizeof(double)) == 0;
            off += lens[a];
        }
        if (same) return ROBO_OK;
    }
    if (k->igkey_len != total) {
        free(k->h_igkey);
        k->h_igkey = (double*)malloc(total * sizeof(double));
        k->igkey_len = k->h_igkey ? total : 0;
    }
    if (k->h_igkey) {
        k->h_igkey[0] = -1.0;      // invalid until the upload below has been issued
        size_t off = 2;
        for (int a = 0; a < 6; ++a) {
            memcpy(k->h_igkey + off, srcs[a], lens[a] * sizeof(double));
            off += lens[a];
        }
    }
    std::vector<double> hc((size_t)128 + npts + (size_t)nb * nb, 0.0), hg((size_t)NB * kf, 0.0);
    for (int i = 0; i < nb; ++i) {
        hc[i] = logP[i];
        hc[64 + i] = lmb[i];
    }
    for (int p = 0; p < npts; ++p) hc[128 + p] = W[p];
    for (int i = 0; i < nb * nb; ++i) hc[128 + npts + i] = dlogPdMu[i];
    const int ntri = nb * (nb + 1) / 2;
    for (int i = 0; i < nb; ++i) {
        double* g1 = hg.data() + (size_t)i * kf;            // q1_i = s^T dlogPdMudMu_i s
        double* g2 = hg.data() + (size_t)(64 + i) * kf;     // q2_i = sum_{a>=b} dlogPdSigma_i[ab] s_a s_b
        for (int e = 0; e < nb * nb; ++e) g1[e] = dlogPdMudMu[(size_t)i * nb * nb + e];
        int idx = 0;
        for (int a = 0; a < nb; ++a)
            for (int b = 0; b <= a; ++b) g2[a * nb + b] = dlogPdSigma[(size_t)i * ntri + idx++];
    }
    hipStream_t st = k->ctx->stream;
    ROBO_HIP_CHECK(hipMemcpyAsync(k->d_igc, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice, st));
    ROBO_HIP_CHECK(hipMemcpyAsync(k->d_G, hg.data(), hg.size() * sizeof(double), hipMemcpyHostToDevice, st));
    ROBO_HIP_CHECK(hipStreamSynchronize(st));   // the staging vectors die with this scope
    if (k->h_igkey) {
        k->h_igkey[0] = (double)nb;
        k->h_igkey[1] = (double)npts;
    }
    return ROBO_OK;
}

static int ig_check(int nb, int npts) {
    if (nb < 2 || nb > 64 || npts < 1 || npts > 512) {
        set_error("information gain: Nb=%d must be in [2, 64], Np=%d in [1, 512]", nb, npts);
        return ROBO_BAD_SHAPE;
    }
    return ROBO_OK;
}

static double ig_entropy(int nb, const double* logP, const double* lmb) {
    double H = 0.0;
    for (int i = 0; i < nb; ++i) H -= std::exp(logP[i]) * (logP[i] + lmb[i]);
    return H;
}

int32_t robo_gp_cross_cov(robo_gp* g, robo_cand* k, robo_cand* rep, double* out_cov) {
    if (!g || !k || !rep || !out_cov) return ROBO_BAD_ARGUMENT;
    if (rep->m > 64) {
        set_error("cross-covariance reference set limited to 64 points (got %lld)", (long long)rep->m);
        return ROBO_BAD_SHAPE;
    }
    ROBO_TRY(predict_core(g, rep, true, nullptr, true));
    ROBO_TRY(cand_ensure_workspace(k, g->n_pad, false));
    ROBO_TRY(ig_ensure(k, 16));
    ROBO_TRY(predict_core(g, k, false, [&](int64_t c0, int64_t cn) { return launch_cross_cov(g, k, rep, c0, cn, k->d_S); }));
    std::vector<double> h((size_t)k->m * NB);
    ROBO_HIP_CHECK(hipMemcpyAsync(h.data(), k->d_S, h.size() * sizeof(double), hipMemcpyDeviceToHost, g->ctx->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
    for (int64_t c = 0; c < k->m; ++c)
        for (int64_t b = 0; b < rep->m; ++b) out_cov[c * rep->m + b] = h[(size_t)c * NB + b];
    return ROBO_OK;
}

// dH of every candidate of k into k->d_acq_sum (asynchronous)
static int ig_core(robo_gp* g, robo_cand* k, robo_cand* rep, int32_t npts, double sn2, const double* logP,
                   const double* lmb, const double* W, const double* dlogPdMu, const double* dlogPdSigma,
                   const double* dlogPdMudMu) {
    if (!g || !k || !rep || !logP || !lmb || !W || !dlogPdMu || !dlogPdSigma || !dlogPdMudMu) return ROBO_BAD_ARGUMENT;
    const int nb = (int)rep->m;
    ROBO_TRY(ig_check(nb, npts));
    const int kf = round_up(nb * nb, 16);
    // V of the representer points: kept across calls while the factor and the points stay the same (the reference
    // does its representer-point work once per update(), information_gain.py:127-167, not per compute())
    if (!(rep->solved_gen != 0 && rep->solved_gp == g && rep->solved_gen == g->fit_gen)) {
        ROBO_TRY(predict_core(g, rep, true, nullptr, true));
        rep->solved_gp = g;
        rep->solved_gen = g->fit_gen;
    }
    ROBO_TRY(cand_ensure_workspace(k, g->n_pad, false));
    ROBO_TRY(ig_ensure(k, kf));
    ROBO_TRY(ig_upload(k, nb, npts, kf, logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu));
    ROBO_TRY(predict_core(g, k, false, [&](int64_t c0, int64_t cn) { return launch_cross_cov(g, k, rep, c0, cn, k->d_S); }));
    const double H = ig_entropy(nb, logP, lmb);
    for (int64_t c0 = 0; c0 < k->m_pad; c0 += k->chunk) {
        const int64_t cn = k->m_pad - c0 < k->chunk ? k->m_pad - c0 : k->chunk;
        ROBO_TRY(launch_ig_dh(g->ctx, k->d_S, k->d_var, k->d_F, k->d_Q, k->d_G, k->d_igc, c0, cn, k->m, nb, npts, kf,
                              sn2, H, k->d_acq_sum));
    }
    return ROBO_OK;
}

// dH / (exp(log-cost mean) + overhead) of every candidate into k->d_acq (and the best of them into the argmax slots):
// the local half of robo_ig_eval_per_cost_cand and of its sharded form (comm.hip)
static int ig_per_cost_core(robo_gp* g, robo_cand* k, robo_cand* rep, int32_t npts, double sn2, const double* logP,
                            const double* lmb, const double* W, const double* dlogPdMu, const double* dlogPdSigma,
                            const double* dlogPdMudMu, robo_gp* cost_gp, robo_cand* cost_k, double overhead) {
    if (!cost_gp || !cost_k || !k) return ROBO_BAD_ARGUMENT;
    if (cost_k->m != k->m || cost_k->ctx != k->ctx || cost_gp->ctx != k->ctx || cost_k == k) {
        set_error("information gain per unit cost: the cost model's candidate handle must hold the same %lld candidates "
                  "(in the cost model's input space) on the same context", (long long)k->m);
        return ROBO_BAD_SHAPE;
    }
    ROBO_TRY(ig_core(g, k, rep, npts, sn2, logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu));
    ROBO_TRY(predict_core(cost_gp, cost_k, false));           // cost_k->d_mean: the cost model's (log-cost) mean
    ROBO_TRY(launch_per_cost(k->ctx, k->d_acq_sum, cost_k->d_mean, overhead, k->m));
    return launch_argmax(k, k->d_acq_sum, 1.0);
}

int32_t robo_ig_eval_cand(robo_gp* g, robo_cand* k, robo_cand* rep, int32_t npts, double sn2, const double* logP,
                          const double* lmb, const double* W, const double* dlogPdMu, const double* dlogPdSigma,
                          const double* dlogPdMudMu, double* out_dh, double* out_max, int64_t* out_argmax) {
    ROBO_TRY(ig_core(g, k, rep, npts, sn2, logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu));
    ROBO_TRY(launch_argmax(k, k->d_acq_sum, 1.0));
    return acq_read_back(k, k->d_acq, out_dh, out_max, out_argmax, nullptr);
}

int32_t robo_ig_eval_per_cost_cand(robo_gp* g, robo_cand* k, robo_cand* rep, int32_t npts, double sn2, const double* logP,
                                   const double* lmb, const double* W, const double* dlogPdMu, const double* dlogPdSigma,
                                   const double* dlogPdMudMu, robo_gp* cost_gp, robo_cand* cost_k, double overhead,
                                   double* out_values, double* out_max, int64_t* out_argmax) {
    ROBO_TRY(ig_per_cost_core(g, k, rep, npts, sn2, logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu, cost_gp, cost_k,
                              overhead));
    return acq_read_back(k, k->d_acq, out_values, out_max, out_argmax, nullptr);
}

int32_t robo_ig_eval_moments(robo_ctx* ctx, int64_t m, int32_t nb, int32_t npts, double sn2, const double* s,
                             const double* v, const double* logP, const double* lmb, const double* W,
                             const double* dlogPdMu, const double* dlogPdSigma, const double* dlogPdMudMu,
                             double* out_dh) {
    if (!ctx || !s || !v || !out_dh) return ROBO_BAD_ARGUMENT;
    ROBO_TRY(ig_check(nb, npts));
    const int kf = round_up(nb * nb, 16);
    robo_cand* k = nullptr;
    ROBO_TRY(cand_alloc(ctx, m, 1, &k));
    k->chunk = k->m_pad;
    int st = ig_ensure(k, kf);
    if (st == ROBO_OK) st = ig_upload(k, nb, npts, kf, logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu);
    if (st == ROBO_OK) {
        std::vector<double> hs((size_t)k->m_pad * NB, 0.0);
        for (int64_t c = 0; c < m; ++c)
            for (int b = 0; b < nb; ++b) hs[(size_t)c * NB + b] = s[c * nb + b];
        hipError_t e = hipMemcpyAsync(k->d_S, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(k->d_var, v, (size_t)m * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("robo_ig_eval_moments upload failed: %s", hipGetErrorString(e));
            st = ROBO_RUNTIME_ERROR;
        }
    }
    if (st == ROBO_OK)
        st = launch_ig_dh(ctx, k->d_S, k->d_var, k->d_F, k->d_Q, k->d_G, k->d_igc, 0, k->m_pad, m, nb, npts, kf, sn2,
                          ig_entropy(nb, logP, lmb), k->d_acq_sum);
    if (st == ROBO_OK) {
        hipError_t e = hipMemcpyAsync(out_dh, k->d_acq_sum, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = ROBO_RUNTIME_ERROR;
    }
    robo_cand_destroy(k);
    return st;
}

// ---------------------------------------------------------------------------------------
}  // extern "C"

// the local halves of the sharded entry points (comm.hip)
namespace robo {
int api_acq_local(robo_gp* g, int kind, double par, double eta, robo_cand* k) {
    ROBO_TRY(check_acq_kind(kind));
    ROBO_TRY(predict_core(g, k, false));
    return clear_flags_on_error(k, launch_acq(g->ctx, k, kind, par, eta, false, false));
}
int api_ig_per_cost_local(robo_gp* g, robo_cand* k, robo_cand* rep, int npts, double sn2, const double* const* ep,
                          robo_gp* cost_gp, robo_cand* cost_k, double overhead) {
    return clear_flags_on_error(k, ig_per_cost_core(g, k, rep, npts, sn2, ep[0], ep[1], ep[2], ep[3], ep[4], ep[5], cost_gp,
                                                     cost_k, overhead));
}
int api_acq_accumulate(robo_gp* const* gps, int S, int kind, double par, const double* etas, robo_cand* k) {
    return acq_accumulate(gps, S, kind, par, etas, k);
}
int api_acq_read_back(robo_cand* k, const double* d_vec, double* out_vec, double* out_max, int64_t* out_argmax,
                      uint32_t* out_flags) {
    return acq_read_back(k, d_vec, out_vec, out_max, out_argmax, out_flags);
}
int api_clear_flags(robo_cand* k, int status) { return clear_flags_on_error(k, status); }
int api_predict_samples(robo_gp* const* gps, int S, robo_cand* k, int cap) { return predict_samples(gps, S, k, cap); }
}  // namespace robo
