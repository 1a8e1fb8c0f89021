// The exchanges of the multi-GPU shards INSIDE the boundary (SURVEY.md sections 8b, 8e): one process per GPU, RCCL over
// xGMI, collectives issued on the library's stream with DEVICE pointers -- no host bounce, no torch in the data path.
//
// The reference is single-process Python; what is sharded here is its two independent axes:
//   * candidates  (robo/maximizers/random_sampling.py:42-50: one acq(X) call, X[y.argmax()]): every rank evaluates its
//     contiguous slice; exchange = all-gather of (max, global index, flags) -- 24 bytes per rank -- and the np.argmax
//     tie-break (lowest global index, NaN maximal) on the device;
//   * hyper-parameter samples (robo/acquisition_functions/marginalization.py:115-121: mean over model.models): every rank
//     accumulates sum_s acq_s over ITS samples on all candidates; exchange = all-gather of the M partial sums and a
//     RANK-ORDERED sum on the device (deterministic and identical on every rank; an all-reduce guarantees neither).
// librccl.so is loaded on first use (dlopen): single-GPU processes never pay for it.  ROBO_RCCL_LIB names another
// library with the same five entry points (the CPU test-suite's shared-memory stand-in, tests/hipemu/fake_rccl.cpp).
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

#include "common.h"

namespace robo {
int api_acq_local(robo_gp* g, int kind, double par, double eta, robo_cand* k);
int api_acq_accumulate(robo_gp* const* gps, int S, int kind, double par, const double* etas, robo_cand* k);
int api_acq_read_back(robo_cand* k, const double* d_vec, double* out_vec, double* out_max, int64_t* out_argmax,
                      uint32_t* out_flags);
int api_clear_flags(robo_cand* k, int status);
int api_ig_per_cost_local(robo_gp* g, robo_cand* k, robo_cand* rep, int npts, double sn2, const double* const* ep,
                          robo_gp* cost_gp, robo_cand* cost_k, double overhead);

// rccl.h, the five entry points used (signatures as in /opt/rocm/include/rccl/rccl.h:187,220,260,339,678)
struct NcclId { char internal[ROBO_COMM_ID_BYTES]; };
typedef int (*nccl_get_unique_id_t)(NcclId*);
typedef int (*nccl_comm_init_rank_t)(void** comm, int nranks, NcclId id, int rank);
typedef int (*nccl_all_gather_t)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t);
typedef int (*nccl_comm_destroy_t)(void* comm);
typedef const char* (*nccl_get_error_string_t)(int);
constexpr int NCCL_FLOAT64 = 8;

struct Rccl {
    void* handle;
    nccl_get_unique_id_t get_unique_id;
    nccl_comm_init_rank_t comm_init_rank;
    nccl_all_gather_t all_gather;
    nccl_comm_destroy_t comm_destroy;
    nccl_get_error_string_t get_error_string;
};

static Rccl* rccl() {
    static Rccl api = {};
    static std::mutex mu;                             // two contexts on two threads may create their communicators at once
    std::lock_guard<std::mutex> lock(mu);
    if (api.handle) return &api;
    const char* name = getenv("ROBO_RCCL_LIB");       // (not a hot path: once per process)
    if (!name || !*name) name = "librccl.so";
    // RTLD_LOCAL: a process that also runs torch.distributed has torch's own bundled librccl.so loaded; the five
    // symbols are taken from THIS handle and nothing is interposed
    void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h && strcmp(name, "librccl.so") == 0) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        set_error("cannot load %s: %s", name, dlerror());
        return nullptr;
    }
    api.get_unique_id = (nccl_get_unique_id_t)dlsym(h, "ncclGetUniqueId");
    api.comm_init_rank = (nccl_comm_init_rank_t)dlsym(h, "ncclCommInitRank");
    api.all_gather = (nccl_all_gather_t)dlsym(h, "ncclAllGather");
    api.comm_destroy = (nccl_comm_destroy_t)dlsym(h, "ncclCommDestroy");
    api.get_error_string = (nccl_get_error_string_t)dlsym(h, "ncclGetErrorString");
    if (!api.get_unique_id || !api.comm_init_rank || !api.all_gather || !api.comm_destroy || !api.get_error_string) {
        set_error("%s lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / "
                  "ncclGetErrorString", name);
        dlclose(h);
        return nullptr;
    }
    api.handle = h;
    return &api;
}

#define ROBO_TRY_COMM(expr)           \
    do {                              \
        int _s = (expr);              \
        if (_s != ROBO_OK) return _s; \
    } while (0)

#define ROBO_NCCL_CHECK(expr)                                                                          \
    do {                                                                                               \
        const int _r = (expr);                                                                         \
        if (_r != 0) {                                                                                 \
            robo::set_error("%s failed: %s (%s:%d)", #expr, rccl()->get_error_string(_r), __FILE__, __LINE__); \
            return ROBO_RUNTIME_ERROR;                                                                 \
        }                                                                                              \
    } while (0)

// ---- device side of the exchanges ------------------------------------------------------------------------------------
// this rank's message of the candidate shard: (best value, GLOBAL index as an exact fp64 < 2^53, flag word, status of
// this rank's local half -- ROBO_OK here; a rank whose half failed sends {0, -1, 0, its status} from the host)
constexpr int BEST_MSG = 4;
__global__ void comm_pack_best_kernel(const double* __restrict__ best_val, const long long* __restrict__ best_idx,
                                      const unsigned* __restrict__ flags, long long offset, double* __restrict__ send) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long i = *best_idx;
    send[0] = *best_val;
    send[1] = i < 0 ? -1.0 : (double)(i + offset);
    send[2] = (double)*flags;
    send[3] = 0.0;
}

// np.argmax over the ranks' incumbents (NaN maximal, then the larger value, then the lower global index), flags OR-ed;
// result straight into pinned host memory: [max, argmax (long long), flags (unsigned), owner rank (int), status of the
// first rank whose local half failed (int, ROBO_OK if none), that rank (int)]
__global__ void comm_best_kernel(const double* __restrict__ recv, int world, unsigned* __restrict__ flags,
                                 double* __restrict__ host) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double bv = 0.0;
    long long bi = -1;
    int owner = -1;
    unsigned f = 0u;
    int bad_status = 0, bad_rank = -1;
    for (int r = 0; r < world; ++r) {
        const double v = recv[BEST_MSG * r];
        const long long i = (long long)recv[BEST_MSG * r + 1];
        f |= (unsigned)recv[BEST_MSG * r + 2];
        if (bad_rank < 0 && recv[BEST_MSG * r + 3] != 0.0) {
            bad_status = (int)recv[BEST_MSG * r + 3];
            bad_rank = r;
        }
        if (i < 0) continue;
        bool take;
        if (bi < 0) take = true;
        else {
            const bool vn = isnan(v), bn = isnan(bv);
            if (vn != bn) take = vn;
            else if (!vn && v != bv) take = v > bv;
            else take = i < bi;
        }
        if (take) {
            bv = v;
            bi = i;
            owner = r;
        }
    }
    host[0] = bv;
    reinterpret_cast<long long*>(host)[1] = bi;
    reinterpret_cast<unsigned*>(host + 2)[0] = f;
    reinterpret_cast<int*>(host + 2)[1] = owner;
    reinterpret_cast<int*>(host + 3)[0] = bad_status;
    reinterpret_cast<int*>(host + 3)[1] = bad_rank;
    *flags = 0u;
}

// sample shard: this rank's partial sums + its flag word and the status of its local half behind them
__global__ __launch_bounds__(256) void comm_pack_sum_kernel(const double* __restrict__ part, long long m, int have,
                                                            const unsigned* __restrict__ flags, int status,
                                                            double* __restrict__ send) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) send[i] = have ? part[i] : 0.0;
    if (i == m) {
        send[m] = (double)*flags;
        send[m + 1] = (double)status;
    }
}

// total[i] = sum over ranks IN RANK ORDER of their partial sums; flags OR-ed into the handle's flag word; the status of
// the first rank whose local half failed (and that rank) into pinned host memory
__global__ __launch_bounds__(256) void comm_ordered_sum_kernel(const double* __restrict__ recv, long long stride,
                                                               int world, long long m, double* __restrict__ total,
                                                               unsigned* __restrict__ flags, int* __restrict__ host_status) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        double s = recv[i];
        for (int r = 1; r < world; ++r) s += recv[(size_t)r * stride + i];
        total[i] = s;
    }
    if (i == 0) {
        unsigned f = 0u;
        int bad_status = 0, bad_rank = -1;
        for (int r = 0; r < world; ++r) {
            f |= (unsigned)recv[(size_t)r * stride + m];
            if (bad_rank < 0 && recv[(size_t)r * stride + m + 1] != 0.0) {
                bad_status = (int)recv[(size_t)r * stride + m + 1];
                bad_rank = r;
            }
        }
        *flags = f;
        host_status[0] = bad_status;
        host_status[1] = bad_rank;
    }
}

// the two halves of the sample shard's exchange for callers outside this file (multi.hip: several devices of ONE process)
int launch_comm_pack_sum(hipStream_t st, const double* d_part, long long m, int have, const unsigned* d_flags, int status,
                         double* d_send) {
    hipLaunchKernelGGL(comm_pack_sum_kernel, dim3((unsigned)((m + 1 + 255) / 256)), dim3(256), 0, st, d_part, m, have, d_flags,
                       status, d_send);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}
int launch_comm_ordered_sum(hipStream_t st, const double* d_recv, long long stride, int world, long long m, double* d_total,
                            unsigned* d_flags, int* h_status) {
    hipLaunchKernelGGL(comm_ordered_sum_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_recv, stride, world, m,
                       d_total, d_flags, h_status);
    ROBO_LAUNCH_CHECK();
    return ROBO_OK;
}

}  // namespace robo

struct robo_comm {
    robo_ctx* ctx;
    int rank, world;
    void* nccl;
    double *d_send, *d_recv;
    size_t cap;          // doubles per rank the two buffers hold
    double* h_pinned;    // [8]: results of comm_best_kernel [0..3], status pair of the ordered sum [4]
};

using namespace robo;

static int comm_reserve(robo_comm* c, size_t per_rank) {
    if (c->cap >= per_rank) return ROBO_OK;
    if (c->d_send) ROBO_HIP_CHECK(hipFree(c->d_send));
    if (c->d_recv) ROBO_HIP_CHECK(hipFree(c->d_recv));
    c->d_send = c->d_recv = nullptr;
    c->cap = 0;
    ROBO_HIP_CHECK(hipMalloc((void**)&c->d_send, per_rank * sizeof(double)));
    ROBO_HIP_CHECK(hipMalloc((void**)&c->d_recv, per_rank * (size_t)c->world * sizeof(double)));
    c->cap = per_rank;
    return ROBO_OK;
}

extern "C" {

int32_t robo_comm_create_id(void* out_id) {
    if (!out_id) return ROBO_BAD_ARGUMENT;
    Rccl* api = rccl();
    if (!api) return ROBO_RUNTIME_ERROR;
    NcclId id;
    memset(&id, 0, sizeof(id));
    ROBO_NCCL_CHECK(api->get_unique_id(&id));
    memcpy(out_id, &id, sizeof(id));
    return ROBO_OK;
}

int32_t robo_comm_init(robo_ctx* ctx, int32_t rank, int32_t world, const void* id, robo_comm** out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return ROBO_BAD_ARGUMENT;
    Rccl* api = rccl();
    if (!api) return ROBO_RUNTIME_ERROR;
    ROBO_HIP_CHECK(hipSetDevice(ctx->device));
    robo_comm* c = new robo_comm();
    memset(c, 0, sizeof(*c));
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    NcclId nid;
    memcpy(&nid, id, sizeof(nid));
    const int r = api->comm_init_rank(&c->nccl, world, nid, rank);
    if (r != 0) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->get_error_string(r));
        delete c;
        return ROBO_RUNTIME_ERROR;
    }
    int st = comm_reserve(c, 64);
    if (st == ROBO_OK && hipHostMalloc((void**)&c->h_pinned, 8 * sizeof(double), 0) != hipSuccess) st = ROBO_RUNTIME_ERROR;
    ctx_retain(ctx);             // (released by robo_comm_destroy, also on the failure path right below)
    if (st != ROBO_OK) {
        robo_comm_destroy(c);
        return st;
    }
    *out = c;
    return ROBO_OK;
}

int32_t robo_comm_destroy(robo_comm* c) {
    if (!c) return ROBO_OK;
    hipSetDevice(c->ctx->device);
    hipStreamSynchronize(c->ctx->stream);
    if (c->nccl && rccl()) rccl()->comm_destroy(c->nccl);
    hipFree(c->d_send);
    hipFree(c->d_recv);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    robo_ctx* ctx = c->ctx;
    delete c;
    ctx_release(ctx);
    return ROBO_OK;
}

int32_t robo_comm_info(robo_comm* c, int32_t* out_rank, int32_t* out_world) {
    if (!c) return ROBO_BAD_ARGUMENT;
    if (out_rank) *out_rank = c->rank;
    if (out_world) *out_world = c->world;
    return ROBO_OK;
}

int32_t robo_comm_allgather(robo_comm* c, const double* send, int64_t count, double* recv) {
    if (!c || !send || !recv || count < 1) return ROBO_BAD_ARGUMENT;
    hipStream_t st = c->ctx->stream;
    ROBO_HIP_CHECK(hipSetDevice(c->ctx->device));
    ROBO_TRY_COMM(comm_reserve(c, (size_t)count));
    ROBO_HIP_CHECK(hipMemcpyAsync(c->d_send, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
    ROBO_NCCL_CHECK(rccl()->all_gather(c->d_send, c->d_recv, (size_t)count, NCCL_FLOAT64, c->nccl, st));
    ROBO_HIP_CHECK(hipMemcpyAsync(recv, c->d_recv, (size_t)count * c->world * sizeof(double), hipMemcpyDeviceToHost, st));
    ROBO_HIP_CHECK(hipStreamSynchronize(st));
    return ROBO_OK;
}

// The exchange of a candidate shard: this rank's incumbent (left in the handle's argmax slots by the local half, whose
// status is `status`) -> the global (max, argmax, owner, flags) on every rank.  No early return before the collective.
static int exchange_best(robo_comm* c, robo_cand* k, int status, int64_t global_offset, const char* who, double* out_acq,
                         double* out_max, int64_t* out_argmax, int32_t* out_owner_rank, uint32_t* out_flags) {
    hipStream_t st = c->ctx->stream;
    const bool ok = status == ROBO_OK;
    if (ok) {
        hipLaunchKernelGGL(comm_pack_best_kernel, dim3(1), dim3(64), 0, st, (const double*)(k->d_part_val + k->n_part),
                           (const long long*)(k->d_part_idx + k->n_part), (const unsigned*)k->d_flags,
                           (long long)global_offset, c->d_send);
    } else {
        // an empty shard for the others, and this rank's status: EVERY rank returns the failure (a job whose ranks
        // disagree about whether the call succeeded hangs in its next collective)
        const double none[BEST_MSG] = {0.0, -1.0, 0.0, (double)status};
        hipMemcpyAsync(c->d_send, none, sizeof(none), hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
    }
    ROBO_NCCL_CHECK(rccl()->all_gather(c->d_send, c->d_recv, BEST_MSG, NCCL_FLOAT64, c->nccl, st));
    hipLaunchKernelGGL(comm_best_kernel, dim3(1), dim3(64), 0, st, (const double*)c->d_recv, c->world, k->d_flags,
                       c->h_pinned);
    if (ok && out_acq)
        ROBO_HIP_CHECK(hipMemcpyAsync(out_acq, k->d_acq, (size_t)k->m * sizeof(double), hipMemcpyDeviceToHost, st));
    ROBO_HIP_CHECK(hipStreamSynchronize(st));
    if (!ok) return status;                          // (its own message stands in the error string)
    const double* hp = c->h_pinned;
    {
        int bad[2];
        memcpy(bad, hp + 3, sizeof(bad));
        if (bad[0] != ROBO_OK) {
            set_error("%s: the local half of rank %d failed with status %d", who, bad[1], bad[0]);
            return bad[0];
        }
    }
    if (out_max) *out_max = hp[0];
    if (out_argmax) {
        long long i;
        memcpy(&i, hp + 1, sizeof(i));
        *out_argmax = (int64_t)i;
    }
    unsigned f;
    int owner;
    memcpy(&f, hp + 2, sizeof(f));
    memcpy(&owner, reinterpret_cast<const char*>(hp + 2) + sizeof(unsigned), sizeof(owner));
    if (out_flags) *out_flags = f;
    if (out_owner_rank) *out_owner_rank = owner;
    return ROBO_OK;
}

int32_t robo_acq_eval_cand_sharded(robo_comm* c, robo_gp* g, int32_t acq_kind, double par, double eta, robo_cand* k,
                                   int64_t global_offset, double* out_acq, double* out_max, int64_t* out_argmax,
                                   int32_t* out_owner_rank, uint32_t* out_flags) {
    if (!c || !g || !k) return ROBO_BAD_ARGUMENT;
    if (g->ctx != c->ctx || k->ctx != c->ctx) {
        set_error("robo_acq_eval_cand_sharded: the GP, the candidates and the communicator must share one context");
        return ROBO_BAD_ARGUMENT;
    }
    ROBO_HIP_CHECK(hipSetDevice(c->ctx->device));
    const int status = api_acq_local(g, acq_kind, par, eta, k);
    return exchange_best(c, k, status, global_offset, "robo_acq_eval_cand_sharded", out_acq, out_max, out_argmax,
                         out_owner_rank, out_flags);
}

int32_t robo_ig_eval_per_cost_cand_sharded(robo_comm* c, robo_gp* g, robo_cand* k, robo_cand* rep, int32_t npts, double sn2,
                                           const double* logP, const double* lmb, const double* W, const double* dlogPdMu,
                                           const double* dlogPdSigma, const double* dlogPdMudMu, robo_gp* cost_gp,
                                           robo_cand* cost_k, double overhead, int64_t global_offset, double* out_values,
                                           double* out_max, int64_t* out_argmax, int32_t* out_owner_rank) {
    if (!c || !g || !k || !rep || !cost_gp || !cost_k) return ROBO_BAD_ARGUMENT;
    if (g->ctx != c->ctx || k->ctx != c->ctx || rep->ctx != c->ctx) {
        set_error("robo_ig_eval_per_cost_cand_sharded: the GPs, the candidates and the communicator must share one context");
        return ROBO_BAD_ARGUMENT;
    }
    ROBO_HIP_CHECK(hipSetDevice(c->ctx->device));
    const double* ep[6] = {logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu};
    const int status = api_ig_per_cost_local(g, k, rep, npts, sn2, ep, cost_gp, cost_k, overhead);
    return exchange_best(c, k, status, global_offset, "robo_ig_eval_per_cost_cand_sharded", out_values, out_max,
                         out_argmax, out_owner_rank, nullptr);
}

int32_t robo_acq_eval_marginal_cand_sharded(robo_comm* c, robo_gp* const* gps, int32_t S_local, int32_t S_total,
                                            int32_t acq_kind, double par, const double* etas, robo_cand* k,
                                            double* out_acq, double* out_max, int64_t* out_argmax,
                                            uint32_t* out_flags) {
    if (!c || !k || S_local < 0 || S_total < 1 || (S_local > 0 && (!gps || !etas))) return ROBO_BAD_ARGUMENT;
    if (k->ctx != c->ctx) {
        set_error("robo_acq_eval_marginal_cand_sharded: the candidates and the communicator must share one context");
        return ROBO_BAD_ARGUMENT;
    }
    hipStream_t st = c->ctx->stream;
    ROBO_HIP_CHECK(hipSetDevice(c->ctx->device));
    const long long m = (long long)k->m;
    // (out of device memory for the exchange buffers: nothing to send from -- the one error that leaves before the collective)
    ROBO_TRY_COMM(comm_reserve(c, (size_t)m + 2));
    int status = ROBO_OK;
    if (S_local > 0) status = api_acq_accumulate(gps, S_local, acq_kind, par, etas, k);
    const bool ok = status == ROBO_OK;
    // a failed rank still takes part in the collective (zeros): the others must not hang
    hipLaunchKernelGGL(comm_pack_sum_kernel, dim3((unsigned)((m + 1 + 255) / 256)), dim3(256), 0, st,
                       (const double*)k->d_acq_sum, m, (ok && S_local > 0) ? 1 : 0, (const unsigned*)k->d_flags, status,
                       c->d_send);
    ROBO_NCCL_CHECK(rccl()->all_gather(c->d_send, c->d_recv, (size_t)m + 2, NCCL_FLOAT64, c->nccl, st));
    hipLaunchKernelGGL(comm_ordered_sum_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st,
                       (const double*)c->d_recv, m + 2, c->world, m, k->d_acq_sum, k->d_flags,
                       reinterpret_cast<int*>(c->h_pinned + 4));
    // EVERY rank learns whether any rank's local half failed (a sum with a rank's samples missing must not be returned
    // as ROBO_OK anywhere): one stream synchronisation before the argmax
    hipStreamSynchronize(st);
    if (!ok) return api_clear_flags(k, status);
    {
        int bad[2];
        memcpy(bad, c->h_pinned + 4, sizeof(bad));
        if (bad[0] != ROBO_OK) {
            set_error("robo_acq_eval_marginal_cand_sharded: the local half of rank %d failed with status %d", bad[1],
                      bad[0]);
            return api_clear_flags(k, bad[0]);
        }
    }
    ROBO_TRY_COMM(api_clear_flags(k, launch_argmax(k, k->d_acq_sum, (double)S_total)));
    return api_clear_flags(k, api_acq_read_back(k, k->d_acq, out_acq, out_max, out_argmax, out_flags));
}

}  // extern "C"
