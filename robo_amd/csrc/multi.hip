// Several devices driven by ONE process (SURVEY.md section 8b "Threading: single process, one context per device"; 8e).
//
// The reference is one Python process (robo/solver/bayesian_optimization.py:156-203: one objective evaluation per
// iteration, one model, one maximiser).  comm.hip shards its two independent axes over one PROCESS per GPU; this file does
// the same over G contexts of the calling process, so that a robo.fmin caller reaches all GPUs of a node without a
// launcher: every _multi entry point fans its shards out to one worker thread per device (each runs the ordinary
// single-device entry point on its own context and stream, concurrently with the others), waits for all of them and
// reduces the G results:
//   * candidate shards (RandomSampling.maximize, information gain per unit cost): the per-device (max, index, flags) come
//     back through each context's pinned read-back exactly as in a single-device call; the np.argmax tie-break across
//     devices (NaN maximal, larger value, lower global index) is 32 bytes per device of HOST arithmetic -- an RCCL call
//     would add a collective launch and a second synchronisation per device to move numbers the host already holds;
//   * sample shards (MarginalizationGPMCMC.compute, GaussianProcessMCMC.predict): the per-device partial sums / sample
//     posteriors travel device to device (hipMemcpyPeerAsync over xGMI) to the first device and are reduced there by the
//     SAME kernels the one-process-per-GPU path uses (rank-ordered sum, mixture), so both multi-GPU forms agree bit for bit.
// No collective: a failing device cannot hang the others; the call returns the first failing device's status after all
// workers have finished.
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace robo {
int api_acq_accumulate(robo_gp* const* gps, int S, int kind, double par, const double* etas, robo_cand* k);
int api_acq_read_back(robo_cand* k, const double* d_vec, double* out_vec, double* out_max, int64_t* out_argmax,
                      uint32_t* out_flags);
int api_clear_flags(robo_cand* k, int status);
int api_predict_samples(robo_gp* const* gps, int S, robo_cand* k, int cap);
int launch_comm_pack_sum(hipStream_t st, const double* d_part, long long m, int have, const unsigned* d_flags, int status,
                         double* d_send);
int launch_comm_ordered_sum(hipStream_t st, const double* d_recv, long long stride, int world, long long m, double* d_total,
                            unsigned* d_flags, int* h_status);

// a C++ exception (an allocation failure inside an entry point) must not leave a worker thread: the caller would wait
// for ever, and an exception escaping a std::thread ends the process
static int guarded(const std::function<int()>& fn) {
    try {
        return fn();
    } catch (const std::exception& e) {
        set_error("exception in a device worker: %s", e.what());
    } catch (...) {
        set_error("exception in a device worker");
    }
    return ROBO_RUNTIME_ERROR;
}

struct MultiWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int status = ROBO_OK;
    char err[1024] = "";

    void loop() {
        for (;;) {
            std::function<int()> fn;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return has_job || quit; });
                if (quit) return;
                fn = std::move(job);
                has_job = false;
            }
            const int st = guarded(fn);
            {
                std::lock_guard<std::mutex> lock(mu);
                status = st;
                if (st != ROBO_OK) snprintf(err, sizeof(err), "%s", robo_last_error_string());
                done = true;
            }
            cv.notify_all();
        }
    }
};
}  // namespace robo

struct robo_multi {
    int G;
    std::vector<robo_ctx*> ctx;
    bool threads;
    std::vector<robo::MultiWorker*> w;
    // exchange buffers of the sample shard: one send buffer per device, the receive buffer on the first device
    std::vector<double*> d_send;
    std::vector<size_t> send_cap;
    double* d_recv;
    size_t recv_cap;
    double* h_pinned;   // [8] on the first device's context: status pair of the ordered sum at [4]
    // device g -> first device: 1 = peer access enabled (hipMemcpyPeerAsync moves the bytes over xGMI), 0 = the two devices
    // cannot address each other (hipDeviceCanAccessPeer said no, or ROBO_MULTI_NO_PEER=1): gathers are staged explicitly
    // through `h_stage` (pinned; D2H on the source device, H2D on the first device's stream)
    std::vector<int> peer;
    double* h_stage;
    size_t stage_cap;
    // the object is NOT re-entrant (one job slot per worker, shared exchange buffers): entry points hold this for the call
    std::mutex call_mu;
};

using namespace robo;

// run fn(g) for every device, concurrently; -> status of the first device that failed (its message becomes the caller's
// error string), ROBO_OK if none did.  Always waits for every device.
static int multi_run(robo_multi* m, const std::function<int(int)>& fn, std::vector<int>* statuses = nullptr) {
    std::vector<int> st((size_t)m->G, ROBO_OK);
    std::vector<std::string> msg((size_t)m->G);
    if (!m->threads) {
        for (int g = 0; g < m->G; ++g) {
            st[(size_t)g] = guarded([&fn, g] { return fn(g); });
            if (st[(size_t)g] != ROBO_OK) msg[(size_t)g] = robo_last_error_string();
        }
    } else {
        for (int g = 0; g < m->G; ++g) {
            MultiWorker* w = m->w[(size_t)g];
            {
                std::lock_guard<std::mutex> lock(w->mu);
                w->job = [&fn, g] { return fn(g); };
                w->has_job = true;
                w->done = false;
            }
            w->cv.notify_all();
        }
        for (int g = 0; g < m->G; ++g) {
            MultiWorker* w = m->w[(size_t)g];
            std::unique_lock<std::mutex> lock(w->mu);
            w->cv.wait(lock, [&] { return w->done; });
            st[(size_t)g] = w->status;
            if (w->status != ROBO_OK) msg[(size_t)g] = w->err;
        }
    }
    if (statuses) *statuses = st;
    for (int g = 0; g < m->G; ++g)
        if (st[(size_t)g] != ROBO_OK) {
            set_error("device %d (context %d of %d): %s", m->ctx[(size_t)g]->device, g, m->G, msg[(size_t)g].c_str());
            return st[(size_t)g];
        }
    return ROBO_OK;
}

// contiguous shard of n items for device g: the first n % G devices hold one more (robo_amd/sharding.py shard_range)
static void shard_range(int64_t n, int g, int G, int64_t* b, int64_t* e) {
    const int64_t base = n / G, rem = n % G;
    *b = g * base + (g < rem ? g : rem);
    *e = *b + base + (g < rem ? 1 : 0);
}

static int multi_reserve(robo_multi* m, size_t per_dev) {
    for (int g = 0; g < m->G; ++g) {
        if (m->send_cap[(size_t)g] >= per_dev) continue;
        ROBO_HIP_CHECK(hipSetDevice(m->ctx[(size_t)g]->device));
        if (m->d_send[(size_t)g]) ROBO_HIP_CHECK(hipFree(m->d_send[(size_t)g]));
        m->d_send[(size_t)g] = nullptr;
        m->send_cap[(size_t)g] = 0;
        ROBO_HIP_CHECK(hipMalloc((void**)&m->d_send[(size_t)g], per_dev * sizeof(double)));
        m->send_cap[(size_t)g] = per_dev;
    }
    if (m->recv_cap < per_dev) {
        ROBO_HIP_CHECK(hipSetDevice(m->ctx[0]->device));
        if (m->d_recv) ROBO_HIP_CHECK(hipFree(m->d_recv));
        m->d_recv = nullptr;
        m->recv_cap = 0;
        ROBO_HIP_CHECK(hipMalloc((void**)&m->d_recv, per_dev * (size_t)m->G * sizeof(double)));
        m->recv_cap = per_dev;
    }
    return ROBO_OK;
}

static int check_on(const robo_multi* m, int g, const robo_ctx* c, const char* who, const char* what) {
    if (c != m->ctx[(size_t)g]) {
        set_error("%s: %s of device slot %d lives on another context", who, what, g);
        return ROBO_BAD_ARGUMENT;
    }
    return ROBO_OK;
}

// np.argmax across the devices' incumbents: NaN maximal, then the larger value, then the lower global index
static bool better(double v, int64_t i, double bv, int64_t bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;
    if (!vn && v != bv) return v > bv;
    return i < bi;
}

extern "C" {

// `bytes` from device slot g's `src` to `dst` on the first device, ordered on the first device's stream.  Peers: one
// hipMemcpyPeerAsync (xGMI).  No peer access: staged through pinned host memory -- the source device's copy is complete
// before the first device's stream picks the bytes up, and the staging area is free again when this returns.
static int gather_to_first(robo_multi* m, int g, double* dst, const double* src, size_t bytes) {
    robo_ctx* c0 = m->ctx[0];
    robo_ctx* cg = m->ctx[(size_t)g];
    if (m->peer[(size_t)g]) {
        ROBO_HIP_CHECK(hipMemcpyPeerAsync(dst, c0->device, src, cg->device, bytes, c0->stream));
        return ROBO_OK;
    }
    if (bytes > m->stage_cap) {
        if (m->h_stage) ROBO_HIP_CHECK(hipHostFree(m->h_stage));
        m->h_stage = nullptr;
        m->stage_cap = 0;
        ROBO_HIP_CHECK(hipHostMalloc((void**)&m->h_stage, bytes, 0));
        m->stage_cap = bytes;
    }
    ROBO_HIP_CHECK(hipSetDevice(cg->device));
    ROBO_HIP_CHECK(hipMemcpyAsync(m->h_stage, src, bytes, hipMemcpyDeviceToHost, cg->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(cg->stream));
    ROBO_HIP_CHECK(hipSetDevice(c0->device));
    ROBO_HIP_CHECK(hipMemcpyAsync(dst, m->h_stage, bytes, hipMemcpyHostToDevice, c0->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(c0->stream));
    return ROBO_OK;
}

int32_t robo_multi_create(robo_ctx* const* ctxs, int32_t n_ctx, robo_multi** out) {
    if (!ctxs || !out || n_ctx < 1 || n_ctx > 64) return ROBO_BAD_ARGUMENT;
    for (int g = 0; g < n_ctx; ++g) {
        if (!ctxs[g]) return ROBO_BAD_ARGUMENT;
        for (int h = 0; h < g; ++h)
            if (ctxs[h] == ctxs[g]) {
                set_error("robo_multi_create: context %d is context %d again (one context per slot)", g, h);
                return ROBO_BAD_ARGUMENT;
            }
    }
    robo_multi* m = new robo_multi();
    m->G = n_ctx;
    m->ctx.assign(ctxs, ctxs + n_ctx);
    const char* e = getenv("ROBO_MULTI_THREADS");      // 0: run the per-device halves one after the other on the caller's thread
    m->threads = n_ctx > 1 && !(e && atoi(e) == 0);
    m->d_send.assign((size_t)n_ctx, nullptr);
    m->send_cap.assign((size_t)n_ctx, 0);
    m->d_recv = nullptr;
    m->recv_cap = 0;
    m->h_pinned = nullptr;
    if (hipSetDevice(ctxs[0]->device) != hipSuccess || hipHostMalloc((void**)&m->h_pinned, 8 * sizeof(double), 0) != hipSuccess) {
        set_error("robo_multi_create: pinned staging on device %d failed", ctxs[0]->device);
        delete m;
        return ROBO_RUNTIME_ERROR;
    }
    m->h_stage = nullptr;
    m->stage_cap = 0;
    m->peer.assign((size_t)n_ctx, 1);
    {
        const char* np = getenv("ROBO_MULTI_NO_PEER");
        const bool no_peer = np && atoi(np) != 0;
        for (int g = 1; g < n_ctx; ++g) {
            const int d0 = ctxs[0]->device, dg = ctxs[g]->device;
            if (dg == d0) continue;                       // two contexts on one device: an ordinary device-to-device copy
            int can = 0;
            if (no_peer || hipDeviceCanAccessPeer(&can, d0, dg) != hipSuccess || !can) {
                m->peer[(size_t)g] = 0;
                continue;
            }
            const hipError_t en = hipDeviceEnablePeerAccess(dg, 0);     // (current device: d0, set above)
            if (en != hipSuccess && en != hipErrorPeerAccessAlreadyEnabled) m->peer[(size_t)g] = 0;
            (void)hipGetLastError();
        }
    }
    if (m->threads)
        for (int g = 0; g < n_ctx; ++g) {
            MultiWorker* w = new MultiWorker();
            w->th = std::thread([w] { w->loop(); });
            m->w.push_back(w);
        }
    for (int g = 0; g < n_ctx; ++g) ctx_retain(ctxs[g]);      // the contexts outlive this object (common.h: lifetime)
    *out = m;
    return ROBO_OK;
}

int32_t robo_multi_destroy(robo_multi* m) {
    if (!m) return ROBO_OK;
    for (MultiWorker* w : m->w) {
        {
            std::lock_guard<std::mutex> lock(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        w->th.join();
        delete w;
    }
    for (int g = 0; g < m->G; ++g)
        if (m->d_send[(size_t)g]) {
            hipSetDevice(m->ctx[(size_t)g]->device);
            hipFree(m->d_send[(size_t)g]);
        }
    hipSetDevice(m->ctx[0]->device);
    if (m->d_recv) hipFree(m->d_recv);
    if (m->h_pinned) hipHostFree(m->h_pinned);
    if (m->h_stage) hipHostFree(m->h_stage);
    const std::vector<robo_ctx*> ctxs = m->ctx;
    delete m;
    for (robo_ctx* c : ctxs) ctx_release(c);
    return ROBO_OK;
}

int32_t robo_multi_info(robo_multi* m, int32_t* out_n, int32_t* out_devices, int32_t* out_threads) {
    if (!m) return ROBO_BAD_ARGUMENT;
    if (out_n) *out_n = m->G;
    if (out_devices)
        for (int g = 0; g < m->G; ++g) out_devices[g] = m->ctx[(size_t)g]->device;
    if (out_threads) *out_threads = m->threads ? m->G : 0;
    return ROBO_OK;
}

int32_t robo_gp_set_data_multi(robo_multi* m, robo_gp* const* gps, const double* X, const double* y, int32_t n) {
    if (!m || !gps || !X || !y) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    for (int g = 0; g < m->G; ++g) {
        if (!gps[g]) return ROBO_BAD_ARGUMENT;
        ROBO_TRY(check_on(m, g, gps[g]->ctx, "robo_gp_set_data_multi", "the GP"));
    }
    return multi_run(m, [&](int g) -> int { return (int)robo_gp_set_data(gps[g], X, y, n); });
}

int32_t robo_gp_fit_multi(robo_multi* m, robo_gp* const* gps, const double* theta, double mean_c, double* out_loglik,
                          int32_t* out_fail_col) {
    if (!m || !gps || !theta) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    for (int g = 0; g < m->G; ++g) {
        if (!gps[g]) return ROBO_BAD_ARGUMENT;
        ROBO_TRY(check_on(m, g, gps[g]->ctx, "robo_gp_fit_multi", "the GP"));
    }
    std::vector<double> ll((size_t)m->G, 0.0);
    std::vector<int32_t> col((size_t)m->G, 0);
    std::vector<int> st;
    const int first_bad = multi_run(m, [&](int g) -> int { return (int)robo_gp_fit(gps[g], theta, mean_c, &ll[(size_t)g], &col[(size_t)g]); },
                                    &st);
    if (out_fail_col) *out_fail_col = col[0];
    for (int g = 1; g < m->G; ++g)
        if (st[(size_t)g] != st[0] || (st[0] == ROBO_OK && memcmp(&ll[(size_t)g], &ll[0], sizeof(double)) != 0)) {
            // the fit is deterministic: replicas that disagree mean different data / different hardware state
            set_error("robo_gp_fit_multi: the replicas of devices %d and %d disagree (status %d / %d, log-likelihood %.17g / "
                      "%.17g)", m->ctx[0]->device, m->ctx[(size_t)g]->device, st[0], st[(size_t)g], ll[0], ll[(size_t)g]);
            for (int h = 0; h < m->G; ++h) gps[h]->fitted = false;
            return ROBO_RUNTIME_ERROR;
        }
    if (first_bad != ROBO_OK) return first_bad;
    if (out_loglik) *out_loglik = ll[0];
    return ROBO_OK;
}

int32_t robo_gp_loglik_batch_multi(robo_multi* m, robo_gp* const* gps, const double* thetas, int32_t S, double mean_c,
                                   double* out_loglik, int32_t* out_status) {
    if (!m || !gps || !thetas || S < 0 || !out_loglik) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    for (int g = 0; g < m->G; ++g) {
        if (!gps[g]) return ROBO_BAD_ARGUMENT;
        ROBO_TRY(check_on(m, g, gps[g]->ctx, "robo_gp_loglik_batch_multi", "the GP"));
    }
    const int P = robo_theta_size(gps[0]->kind, gps[0]->dim);
    return multi_run(m, [&](int g) -> int {
        int64_t b, e;
        shard_range(S, g, m->G, &b, &e);
        if (e <= b) return (int)ROBO_OK;
        return (int)robo_gp_loglik_batch(gps[g], thetas + (size_t)b * P, (int32_t)(e - b), mean_c, out_loglik + b,
                                         out_status ? out_status + b : nullptr);
    });
}

int32_t robo_gp_fit_batch_multi(robo_multi* m, robo_gp* const* gps, const int32_t* S_dev, const double* thetas, double mean_c,
                                double* out_loglik, int32_t* out_status) {
    if (!m || !gps || !S_dev || !thetas || !out_loglik || !out_status) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    std::vector<int> off((size_t)m->G + 1, 0);
    for (int g = 0; g < m->G; ++g) {
        if (S_dev[g] < 0) return ROBO_BAD_ARGUMENT;
        off[(size_t)g + 1] = off[(size_t)g] + S_dev[g];
        for (int s = off[(size_t)g]; s < off[(size_t)g + 1]; ++s) {
            if (!gps[s]) return ROBO_BAD_ARGUMENT;
            ROBO_TRY(check_on(m, g, gps[s]->ctx, "robo_gp_fit_batch_multi", "a GP"));
        }
    }
    if (off[(size_t)m->G] == 0) return ROBO_OK;
    const int P = robo_theta_size(gps[0]->kind, gps[0]->dim);
    return multi_run(m, [&](int g) -> int {
        const int o = off[(size_t)g], ns = S_dev[g];
        if (ns == 0) return (int)ROBO_OK;
        return (int)robo_gp_fit_batch(gps + o, ns, thetas + (size_t)o * P, mean_c, out_loglik + o, out_status + o);
    });
}

// reduce the per-device incumbents of a candidate shard on the host
static int reduce_best(robo_multi* m, const std::vector<double>& mx, const std::vector<int64_t>& am,
                       const std::vector<uint32_t>& fl, const std::vector<char>& have, const int64_t* global_offsets,
                       double* out_max, int64_t* out_argmax, int32_t* out_owner, uint32_t* out_flags) {
    double bv = 0.0;
    int64_t bi = -1;
    int owner = -1;
    uint32_t f = 0u;
    for (int g = 0; g < m->G; ++g) {
        if (!have[(size_t)g]) continue;
        f |= fl[(size_t)g];
        if (am[(size_t)g] < 0) continue;
        const int64_t gi = global_offsets[g] + am[(size_t)g];
        if (bi < 0 || better(mx[(size_t)g], gi, bv, bi)) {
            bv = mx[(size_t)g];
            bi = gi;
            owner = g;
        }
    }
    if (out_max) *out_max = bv;
    if (out_argmax) *out_argmax = bi;
    if (out_owner) *out_owner = owner;
    if (out_flags) *out_flags = f;
    return ROBO_OK;
}

int32_t robo_acq_eval_cand_multi(robo_multi* m, robo_gp* const* gps, int32_t acq_kind, double par, double eta,
                                 robo_cand* const* cands, const int64_t* global_offsets, double* out_acq, double* out_max,
                                 int64_t* out_argmax, int32_t* out_owner, uint32_t* out_flags) {
    if (!m || !gps || !cands || !global_offsets) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    std::vector<int64_t> pos((size_t)m->G + 1, 0);
    std::vector<char> have((size_t)m->G, 0);
    for (int g = 0; g < m->G; ++g) {
        pos[(size_t)g + 1] = pos[(size_t)g];
        if (!cands[g]) continue;                     // empty shard (fewer candidates than devices)
        if (!gps[g]) return ROBO_BAD_ARGUMENT;
        ROBO_TRY(check_on(m, g, gps[g]->ctx, "robo_acq_eval_cand_multi", "the GP"));
        ROBO_TRY(check_on(m, g, cands[g]->ctx, "robo_acq_eval_cand_multi", "the candidate shard"));
        have[(size_t)g] = 1;
        pos[(size_t)g + 1] += cands[g]->m;
    }
    std::vector<double> mx((size_t)m->G, 0.0);
    std::vector<int64_t> am((size_t)m->G, -1);
    std::vector<uint32_t> fl((size_t)m->G, 0u);
    ROBO_TRY(multi_run(m, [&](int g) -> int {
        if (!have[(size_t)g]) return (int)ROBO_OK;
        return (int)robo_acq_eval_cand(gps[g], acq_kind, par, eta, cands[g], out_acq ? out_acq + pos[(size_t)g] : nullptr,
                                       &mx[(size_t)g], &am[(size_t)g], &fl[(size_t)g]);
    }));
    return reduce_best(m, mx, am, fl, have, global_offsets, out_max, out_argmax, out_owner, out_flags);
}

int32_t robo_ig_eval_per_cost_cand_multi(robo_multi* m, robo_gp* const* gps, robo_cand* const* cands,
                                         robo_cand* const* reps, int32_t n_outcomes, double sn2, const double* logP,
                                         const double* lmb, const double* W, const double* dlogPdMu,
                                         const double* dlogPdSigma, const double* dlogPdMudMu, robo_gp* const* cost_gps,
                                         robo_cand* const* cost_cands, double overhead, const int64_t* global_offsets,
                                         double* out_values, double* out_max, int64_t* out_argmax, int32_t* out_owner) {
    if (!m || !gps || !cands || !reps || !cost_gps || !cost_cands || !global_offsets) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    std::vector<int64_t> pos((size_t)m->G + 1, 0);
    std::vector<char> have((size_t)m->G, 0);
    for (int g = 0; g < m->G; ++g) {
        pos[(size_t)g + 1] = pos[(size_t)g];
        if (!cands[g]) continue;
        if (!gps[g] || !reps[g] || !cost_gps[g] || !cost_cands[g]) return ROBO_BAD_ARGUMENT;
        ROBO_TRY(check_on(m, g, gps[g]->ctx, "robo_ig_eval_per_cost_cand_multi", "the GP"));
        ROBO_TRY(check_on(m, g, cands[g]->ctx, "robo_ig_eval_per_cost_cand_multi", "the candidate shard"));
        ROBO_TRY(check_on(m, g, cost_gps[g]->ctx, "robo_ig_eval_per_cost_cand_multi", "the cost model"));
        have[(size_t)g] = 1;
        pos[(size_t)g + 1] += cands[g]->m;
    }
    std::vector<double> mx((size_t)m->G, 0.0);
    std::vector<int64_t> am((size_t)m->G, -1);
    std::vector<uint32_t> fl((size_t)m->G, 0u);
    ROBO_TRY(multi_run(m, [&](int g) -> int {
        if (!have[(size_t)g]) return (int)ROBO_OK;
        return (int)robo_ig_eval_per_cost_cand(gps[g], cands[g], reps[g], n_outcomes, sn2, logP, lmb, W, dlogPdMu, dlogPdSigma,
                                               dlogPdMudMu, cost_gps[g], cost_cands[g], overhead,
                                               out_values ? out_values + pos[(size_t)g] : nullptr, &mx[(size_t)g],
                                               &am[(size_t)g]);
    }));
    return reduce_best(m, mx, am, fl, have, global_offsets, out_max, out_argmax, out_owner, nullptr);
}

int32_t robo_ig_eval_cand_multi(robo_multi* m, robo_gp* const* gps, robo_cand* const* cands, robo_cand* const* reps,
                                int32_t n_outcomes, double sn2, const double* logP, const double* lmb, const double* W,
                                const double* dlogPdMu, const double* dlogPdSigma, const double* dlogPdMudMu,
                                const int64_t* global_offsets, double* out_dh, double* out_max, int64_t* out_argmax,
                                int32_t* out_owner) {
    if (!m || !gps || !cands || !reps || !global_offsets) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    std::vector<int64_t> pos((size_t)m->G + 1, 0);
    std::vector<char> have((size_t)m->G, 0);
    for (int g = 0; g < m->G; ++g) {
        pos[(size_t)g + 1] = pos[(size_t)g];
        if (!cands[g]) continue;
        if (!gps[g] || !reps[g]) return ROBO_BAD_ARGUMENT;
        ROBO_TRY(check_on(m, g, gps[g]->ctx, "robo_ig_eval_cand_multi", "the GP"));
        ROBO_TRY(check_on(m, g, cands[g]->ctx, "robo_ig_eval_cand_multi", "the candidate shard"));
        ROBO_TRY(check_on(m, g, reps[g]->ctx, "robo_ig_eval_cand_multi", "the representer points"));
        have[(size_t)g] = 1;
        pos[(size_t)g + 1] += cands[g]->m;
    }
    std::vector<double> mx((size_t)m->G, 0.0);
    std::vector<int64_t> am((size_t)m->G, -1);
    std::vector<uint32_t> fl((size_t)m->G, 0u);
    ROBO_TRY(multi_run(m, [&](int g) -> int {
        if (!have[(size_t)g]) return (int)ROBO_OK;
        return (int)robo_ig_eval_cand(gps[g], cands[g], reps[g], n_outcomes, sn2, logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu,
                                      out_dh ? out_dh + pos[(size_t)g] : nullptr, &mx[(size_t)g], &am[(size_t)g]);
    }));
    return reduce_best(m, mx, am, fl, have, global_offsets, out_max, out_argmax, out_owner, nullptr);
}

int32_t robo_acq_eval_marginal_cand_multi(robo_multi* m, robo_gp* const* gps, const int32_t* S_dev, int32_t acq_kind,
                                          double par, const double* etas, robo_cand* const* cands, double* out_acq,
                                          double* out_max, int64_t* out_argmax, uint32_t* out_flags) {
    if (!m || !gps || !S_dev || !etas || !cands || !cands[0]) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    std::vector<int> off((size_t)m->G + 1, 0);
    const long long mm = (long long)cands[0]->m;
    for (int g = 0; g < m->G; ++g) {
        if (S_dev[g] < 0) return ROBO_BAD_ARGUMENT;
        off[(size_t)g + 1] = off[(size_t)g] + S_dev[g];
        if (S_dev[g] > 0 && !cands[g]) return ROBO_BAD_ARGUMENT;
        if (cands[g]) {
            ROBO_TRY(check_on(m, g, cands[g]->ctx, "robo_acq_eval_marginal_cand_multi", "the candidates"));
            if ((long long)cands[g]->m != mm) {
                set_error("robo_acq_eval_marginal_cand_multi: every device evaluates ALL candidates (device slot %d holds %lld, "
                          "slot 0 %lld)", g, (long long)cands[g]->m, mm);
                return ROBO_BAD_SHAPE;
            }
        }
    }
    const int S_total = off[(size_t)m->G];
    if (S_total < 1) return ROBO_BAD_ARGUMENT;
    ROBO_TRY(multi_reserve(m, (size_t)mm + 2));
    // per device: sum_s acq_s over ITS samples, packed with its flag word; one stream synchronisation per device
    const int st = multi_run(m, [&](int g) -> int {
        robo_ctx* c = m->ctx[(size_t)g];
        ROBO_HIP_CHECK(hipSetDevice(c->device));
        int status = ROBO_OK;
        const int ns = S_dev[g];
        if (ns > 0) status = api_acq_accumulate(gps + off[(size_t)g], ns, acq_kind, par, etas + off[(size_t)g], cands[g]);
        if (ns > 0 && status == ROBO_OK)
            status = launch_comm_pack_sum(c->stream, cands[g]->d_acq_sum, mm, 1, cands[g]->d_flags, ROBO_OK, m->d_send[(size_t)g]);
        else
            ROBO_HIP_CHECK(hipMemsetAsync(m->d_send[(size_t)g], 0, ((size_t)mm + 2) * sizeof(double), c->stream));
        if (ns > 0 && g != 0)   // the flag word travelled in the message (slot 0's is rewritten by the ordered sum)
            ROBO_HIP_CHECK(hipMemsetAsync(cands[g]->d_flags, 0, 4 * sizeof(unsigned), c->stream));
        ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));
        return status;
    });
    if (st != ROBO_OK) {
        for (int g = 0; g < m->G; ++g)
            if (cands[g]) {
                hipSetDevice(m->ctx[(size_t)g]->device);
                api_clear_flags(cands[g], st);
            }
        hipSetDevice(m->ctx[0]->device);
        hipStreamSynchronize(m->ctx[0]->stream);
        return st;
    }
    // gather on the first device (xGMI peer copies on ITS stream), add in device order with the kernel of the
    // one-process-per-GPU path, then the ordinary division, argmax and read-back
    robo_ctx* c0 = m->ctx[0];
    robo_cand* k0 = cands[0];
    ROBO_HIP_CHECK(hipSetDevice(c0->device));
    for (int g = 0; g < m->G; ++g)
        ROBO_TRY(gather_to_first(m, g, m->d_recv + (size_t)g * ((size_t)mm + 2), m->d_send[(size_t)g],
                                 ((size_t)mm + 2) * sizeof(double)));
    ROBO_TRY(api_clear_flags(k0, launch_comm_ordered_sum(c0->stream, m->d_recv, mm + 2, m->G, mm, k0->d_acq_sum, k0->d_flags,
                                                         reinterpret_cast<int*>(m->h_pinned + 4))));
    ROBO_TRY(api_clear_flags(k0, launch_argmax(k0, k0->d_acq_sum, (double)S_total)));
    return api_clear_flags(k0, api_acq_read_back(k0, k0->d_acq, out_acq, out_max, out_argmax, out_flags));
}

int32_t robo_gp_predict_mixture_cand_multi(robo_multi* m, robo_gp* const* gps, const int32_t* S_dev, robo_cand* const* cands,
                                           double* out_mean, double* out_var) {
    if (!m || !gps || !S_dev || !cands || !cands[0]) return ROBO_BAD_ARGUMENT;
    std::lock_guard<std::mutex> call_lock(m->call_mu);   // one call at a time per robo_multi (not re-entrant)
    std::vector<int> off((size_t)m->G + 1, 0);
    for (int g = 0; g < m->G; ++g) {
        if (S_dev[g] < 0 || (S_dev[g] > 0 && !cands[g])) return ROBO_BAD_ARGUMENT;
        off[(size_t)g + 1] = off[(size_t)g] + S_dev[g];
        if (cands[g]) {
            ROBO_TRY(check_on(m, g, cands[g]->ctx, "robo_gp_predict_mixture_cand_multi", "the candidates"));
            if (cands[g]->m != cands[0]->m) {
                set_error("robo_gp_predict_mixture_cand_multi: every device evaluates ALL candidates");
                return ROBO_BAD_SHAPE;
            }
        }
    }
    const int S_total = off[(size_t)m->G];
    if (S_total < 1) return ROBO_BAD_ARGUMENT;
    ROBO_TRY(multi_run(m, [&](int g) -> int {
        if (!cands[g] || (g != 0 && S_dev[g] == 0)) return (int)ROBO_OK;
        robo_ctx* c = m->ctx[(size_t)g];
        ROBO_TRY(api_predict_samples(gps + off[(size_t)g], S_dev[g], cands[g], g == 0 ? S_total : S_dev[g]));
        ROBO_HIP_CHECK(hipStreamSynchronize(c->stream));
        return (int)ROBO_OK;
    }));
    robo_ctx* c0 = m->ctx[0];
    robo_cand* k0 = cands[0];
    ROBO_HIP_CHECK(hipSetDevice(c0->device));
    const size_t mp = (size_t)k0->m_pad;
    for (int g = 1; g < m->G; ++g) {
        if (S_dev[g] == 0) continue;
        const size_t rows = (size_t)S_dev[g] * mp * sizeof(double), at = (size_t)off[(size_t)g] * mp;
        ROBO_TRY(gather_to_first(m, g, k0->d_mu_all + at, cands[g]->d_mu_all, rows));
        ROBO_TRY(gather_to_first(m, g, k0->d_var_all + at, cands[g]->d_var_all, rows));
    }
    ROBO_TRY(launch_mixture(k0, S_total));
    if (out_mean)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_mean, k0->d_mean, (size_t)k0->m * sizeof(double), hipMemcpyDeviceToHost, c0->stream));
    if (out_var)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_var, k0->d_var, (size_t)k0->m * sizeof(double), hipMemcpyDeviceToHost, c0->stream));
    ROBO_HIP_CHECK(hipStreamSynchronize(c0->stream));
    return ROBO_OK;
}

}  // extern "C"
