// hipemu runtime -- see hip/hip_runtime.h in this directory.  TEST-ONLY.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <chrono>
#include <cstdlib>
#include <algorithm>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------------------
// x86-64 SysV context switch (callee-saved registers only).
// ---------------------------------------------------------------------------------------
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

enum State { READY = 0, WAIT_BAR = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    void* sp;
    char* stack;
    Lane lane;
    int state;
    int linear;
    // wave-collective mailboxes
    unsigned char in[64];
    unsigned char out[64];
    int op;
};

static const size_t STACK_BYTES = 512 * 1024;
static std::vector<Fiber> g_fibers;
static Fiber* g_cur = nullptr;
static void* g_sched_sp = nullptr;
static dim3 g_block, g_bdim, g_gdim;
static std::function<void()> g_body;
static std::vector<char> g_dyn;
static int g_live = 0, g_bar_wait = 0;
static int g_wave_alive[64], g_wave_wait[64];  // up to 4096 threads / 64

Lane& cur() { return g_cur->lane; }
dim3& cur_block() { return g_block; }
dim3& cur_bdim() { return g_bdim; }
dim3& cur_gdim() { return g_gdim; }
void* dyn_smem() { return g_dyn.data(); }
int lane_id() { return g_cur->linear & 63; }

static void yield_to_sched() { hipemu_switch(&g_cur->sp, g_sched_sp); }

static void fiber_entry() {
    g_body();
    g_cur->state = DONE;
    yield_to_sched();
    std::abort();
}

static void release_barrier() {
    for (auto& f : g_fibers)
        if (f.state == WAIT_BAR) f.state = READY;
    g_bar_wait = 0;
}

void barrier() {
    g_cur->state = WAIT_BAR;
    ++g_bar_wait;
    if (g_bar_wait == g_live) {
        release_barrier();
        return;  // last arriver continues
    }
    yield_to_sched();
}

static void eval_wave(int w) {
    Fiber* base = &g_fibers[(size_t)w * 64];
    int n = (int)g_fibers.size() - w * 64;
    if (n > 64) n = 64;
    int op = -1;
    for (int l = 0; l < n; ++l)
        if (base[l].state == WAIT_WAVE) {
            if (op < 0) op = base[l].op;
            else if (op != base[l].op) {
                std::fprintf(stderr, "hipemu: wave %d lanes disagree on collective op (%d vs %d)\n", w, op, base[l].op);
                std::abort();
            }
        }
    switch (op) {
    case OP_SHFL:
        for (int l = 0; l < n; ++l) {
            if (base[l].state != WAIT_WAVE) continue;
            ShflIn in; std::memcpy(&in, base[l].in, sizeof(in));
            int s = in.src & 63;
            ShflIn sin; sin.bits = 0;
            if (s < n) std::memcpy(&sin, base[s].in, sizeof(sin));   // dead lane: stale bits, like HW
            std::memcpy(base[l].out, &sin.bits, 8);
        }
        break;
    case OP_BALLOT: {
        unsigned long long m = 0;
        for (int l = 0; l < n; ++l)
            if (base[l].state == WAIT_WAVE) { int p; std::memcpy(&p, base[l].in, 4); if (p) m |= 1ull << l; }
        for (int l = 0; l < n; ++l) std::memcpy(base[l].out, &m, 8);
        break;
    }
    case OP_MFMA_F64_16x16x4: {
        if (n != 64) { std::fprintf(stderr, "hipemu: MFMA in a partial wave\n"); std::abort(); }
        struct In { double a, b, c[4]; };
        double A[16][4], B[4][16];
        for (int l = 0; l < 64; ++l) {
            if (base[l].state != WAIT_WAVE) { std::fprintf(stderr, "hipemu: MFMA with inactive lane %d\n", l); std::abort(); }
            In in; std::memcpy(&in, base[l].in, sizeof(in));
            A[l & 15][l >> 4] = in.a;
            B[l >> 4][l & 15] = in.b;
        }
        for (int l = 0; l < 64; ++l) {
            In in; std::memcpy(&in, base[l].in, sizeof(in));
            double d[4];
            for (int r = 0; r < 4; ++r) {
                int row = (l >> 4) + 4 * r, col = l & 15;
                double acc = in.c[r];
                for (int k = 0; k < 4; ++k) acc = std::fma(A[row][k], B[k][col], acc);
                d[r] = acc;
            }
            std::memcpy(base[l].out, d, sizeof(d));
        }
        break;
    }
    case OP_WAVE_BARRIER:
        break;
    default:
        std::fprintf(stderr, "hipemu: unknown collective %d\n", op);
        std::abort();
    }
    for (int l = 0; l < n; ++l)
        if (base[l].state == WAIT_WAVE) base[l].state = READY;
    g_wave_wait[w] = 0;
}

void wave_collective(Op op, const void* in, void* out) {
    static const size_t in_bytes[] = {sizeof(ShflIn), 4, 48, 4};
    static const size_t out_bytes[] = {8, 8, 32, 0};
    Fiber* f = g_cur;
    int w = f->linear >> 6;
    std::memcpy(f->in, in, in_bytes[op]);
    f->op = op;
    f->state = WAIT_WAVE;
    ++g_wave_wait[w];
    if (g_wave_wait[w] == g_wave_alive[w]) eval_wave(w);   // last arriver evaluates and continues
    else yield_to_sched();
    std::memcpy(out, f->out, out_bytes[op]);
}

static void init_fiber(Fiber& f) {
    if (!f.stack) {
        f.stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == (char*)MAP_FAILED) { std::perror("hipemu mmap"); std::abort(); }
    }
    uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // fake return address (keeps entry's rsp = 8 mod 16)
    *--sp = (void*)&fiber_entry;     // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.sp = (void*)sp;
    f.state = READY;
}

// One interpreter for every emulated device and host thread: launches are serialised (the library's multi-device entry
// points launch from one worker thread per device, multi.hip).
// The same lock guards the stream queues of the deferred modes below.
std::recursive_mutex g_rt_mu;

void launch(std::function<void()> body, dim3 grid, dim3 block, size_t shmem) {
    std::lock_guard<std::recursive_mutex> lock(g_rt_mu);
    if (g_cur) { std::fprintf(stderr, "hipemu: nested launch\n"); std::abort(); }
    size_t nt = (size_t)block.x * block.y * block.z;
    if (nt == 0 || nt > 1024) { std::fprintf(stderr, "hipemu: bad block size %zu\n", nt); std::abort(); }
    if (g_fibers.size() < nt) g_fibers.resize(nt);
    static const int order = [] { const char* e = std::getenv("HIPEMU_ORDER"); return e ? std::atoi(e) : 0; }();
    unsigned long long rot = 0x9E3779B97F4A7C15ull;
    g_body = std::move(body);
    g_bdim = block; g_gdim = grid;
    g_dyn.assign(shmem + 64, 0);
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_block = dim3(bx, by, bz);
        // only the first nt fibers take part
        std::vector<Fiber>& F = g_fibers;
        for (size_t i = 0; i < nt; ++i) {
            F[i].linear = (int)i;
            F[i].lane.tid = dim3((unsigned)(i % block.x), (unsigned)((i / block.x) % block.y), (unsigned)(i / ((size_t)block.x * block.y)));
            init_fiber(F[i]);
        }
        for (size_t i = nt; i < F.size(); ++i) F[i].state = DONE;
        g_live = (int)nt; g_bar_wait = 0;
        int nw = (int)((nt + 63) / 64);
        for (int w = 0; w < nw; ++w) { int a = (int)nt - w * 64; g_wave_alive[w] = a > 64 ? 64 : a; g_wave_wait[w] = 0; }
        while (g_live > 0) {
            bool progressed = false;
            rot = rot * 6364136223846793005ull + 1442695040888963407ull;
            for (size_t k = 0; k < nt; ++k) {
                // HIPEMU_ORDER: which work-item runs next between two rendezvous points.  0: ascending (default);
                // 1: descending -- a consumer wave now runs BEFORE the producer wave it forgot to __syncthreads with;
                // 2: a different pseudo-random rotation and direction every pass
                size_t i = k;
                if (order == 1) i = nt - 1 - k;
                else if (order == 2) i = (rot & 1) ? (nt - 1 - (k + (size_t)(rot >> 1)) % nt) : (k + (size_t)(rot >> 1)) % nt;
                Fiber& f = F[i];
                if (f.state != READY) continue;
                progressed = true;
                g_cur = &f;
                hipemu_switch(&g_sched_sp, f.sp);
                g_cur = nullptr;
                if (f.state == DONE) {
                    --g_live;
                    int w = f.linear >> 6;
                    --g_wave_alive[w];
                    if (g_wave_alive[w] > 0 && g_wave_wait[w] == g_wave_alive[w]) eval_wave(w);
                    if (g_live > 0 && g_bar_wait == g_live) release_barrier();
                }
            }
            if (!progressed) {
                std::fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): live=%d at_barrier=%d\n", bx, by, bz, g_live, g_bar_wait);
                std::abort();
            }
        }
    }
    g_body = nullptr;
}

}  // namespace hipemu

// ---------------------------------------------------------------------------------------
// runtime API
// ---------------------------------------------------------------------------------------
// Streams and events.  Three schedules, all LEGAL orders of the work the host enqueued (in-stream order and
// hipStreamWaitEvent edges are always honoured), chosen by HIPEMU_ASYNC / hipemu_set_async():
//   0  every call executes at once, in issue order (the default);
//   1  deferred, "others first": a synchronisation point runs every runnable operation of the OTHER streams before the
//      stream it waits for -- a side stream that forgot to wait for its producer on the main stream reads stale data;
//   2  deferred, "others last": a synchronisation point runs only what the awaited stream (or event) transitively
//      depends on -- a main stream that forgot to join its side streams reads their output before it exists.
// The null stream executes at once in every mode (the library's streams are hipStreamNonBlocking: no implicit
// ordering with it).  hipFree / hipHostFree / hipDeviceSynchronize drain everything, as the runtime does.
struct hipemuEvent {
    std::chrono::steady_clock::time_point t;
    uint64_t enqueued = 0, completed = 0;
    int refs = 0;
    bool destroyed = false;
};
struct EmuOp {
    enum Kind { RUN, RECORD, WAIT } kind;
    std::function<void()> fn;
    hipemuEvent* ev;
    uint64_t seq;
};
struct hipemuStream { std::deque<EmuOp> q; };

using hipemu::g_rt_mu;
typedef std::lock_guard<std::recursive_mutex> RtLock;
static std::vector<hipemuStream*> g_streams;
static std::map<uintptr_t, size_t> g_allocs;            // every hipMalloc / hipHostMalloc range (device or pinned)
static int g_async = -1;

static int async_mode() {
    if (g_async < 0) {
        const char* e = std::getenv("HIPEMU_ASYNC");
        g_async = e ? std::atoi(e) : 0;
        if (g_async < 0 || g_async > 2) g_async = 0;
    }
    return g_async;
}
static bool tracked(const void* p) {
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return false;
    --it;
    return (uintptr_t)p < it->first + it->second;
}
static void unref(hipemuEvent* e) { if (--e->refs == 0 && e->destroyed) delete e; }
static bool runnable(hipemuStream* s) {
    if (s->q.empty()) return false;
    const EmuOp& op = s->q.front();
    return op.kind != EmuOp::WAIT || op.ev->completed >= op.seq;
}
static void exec_head(hipemuStream* s) {
    EmuOp op = std::move(s->q.front());
    s->q.pop_front();
    switch (op.kind) {
    case EmuOp::RUN: op.fn(); break;
    case EmuOp::RECORD:
        op.ev->t = std::chrono::steady_clock::now();
        if (op.ev->completed < op.seq) op.ev->completed = op.seq;
        unref(op.ev);
        break;
    case EmuOp::WAIT: unref(op.ev); break;
    }
}
static void satisfy(hipemuEvent* e, uint64_t seq, int depth);
static void step(hipemuStream* s, int depth) {           // the head of s, after whatever it waits for
    const EmuOp& op = s->q.front();
    if (op.kind == EmuOp::WAIT && op.ev->completed < op.seq) satisfy(op.ev, op.seq, depth + 1);
    exec_head(s);
}
static void satisfy(hipemuEvent* e, uint64_t seq, int depth) {
    if (depth > 256) { std::fprintf(stderr, "hipemu: event wait cycle\n"); std::abort(); }
    while (e->completed < seq) {
        hipemuStream* r = nullptr;
        for (hipemuStream* s : g_streams) {
            for (const EmuOp& op : s->q)
                if (op.kind == EmuOp::RECORD && op.ev == e && op.seq >= seq) { r = s; break; }
            if (r) break;
        }
        if (!r) { std::fprintf(stderr, "hipemu: wait for an event whose record is queued nowhere\n"); std::abort(); }
        step(r, depth);
    }
}
static void drain_stream(hipemuStream* s) { while (!s->q.empty()) step(s, 0); }
template <class Done> static void others_first_until(Done done, hipemuStream* target) {
    while (!done()) {
        hipemuStream* pick = nullptr;
        for (auto it = g_streams.rbegin(); it != g_streams.rend(); ++it)
            if (*it != target && runnable(*it)) { pick = *it; break; }
        if (!pick && target && runnable(target)) pick = target;
        if (!pick) { std::fprintf(stderr, "hipemu: deadlock: nothing runnable at a synchronisation point\n"); std::abort(); }
        exec_head(pick);
    }
}
static void drain_all() {
    if (async_mode() == 1)
        others_first_until([] { for (hipemuStream* s : g_streams) if (!s->q.empty()) return false; return true; }, nullptr);
    else
        for (hipemuStream* s : g_streams) drain_stream(s);
}
static bool deferred(hipStream_t s) { return s != nullptr && async_mode() != 0; }
static long g_deferred_total = 0;
static void enqueue(hipStream_t s, std::function<void()> fn) { ++g_deferred_total; s->q.push_back(EmuOp{EmuOp::RUN, std::move(fn), nullptr, 0}); }

namespace hipemu {
void launch_on(hipStream_t st, std::function<void()> body, dim3 grid, dim3 block, size_t shmem) {
    RtLock lock(g_rt_mu);
    if (!deferred(st)) { launch(std::move(body), grid, block, shmem); return; }
    enqueue(st, [body, grid, block, shmem]() { launch(body, grid, block, shmem); });
}
}  // namespace hipemu

extern "C" void hipemu_set_async(int mode) {
    RtLock lock(g_rt_mu);
    async_mode();
    drain_all();
    g_async = mode < 0 || mode > 2 ? 0 : mode;
}
extern "C" long hipemu_deferred_total() { RtLock lock(g_rt_mu); return g_deferred_total; }
extern "C" int hipemu_pending_ops() {
    RtLock lock(g_rt_mu);
    size_t n = 0;
    for (hipemuStream* s : g_streams) n += s->q.size();
    return (int)n;
}

// HIPEMU_GUARD=1: every allocation ends (to 32 bytes) against an inaccessible page and starts behind one, so that a
// kernel reading or writing past its buffer faults at the access instead of silently touching a neighbour.
struct GuardedAlloc { void* base; size_t len; };
static std::map<uintptr_t, GuardedAlloc> g_guarded;
static bool guard_mode() {
    static const bool on = [] { const char* e = std::getenv("HIPEMU_GUARD"); return e && std::atoi(e) != 0; }();
    return on;
}
hipError_t hipMalloc(void** p, size_t n) {
    void* q = nullptr;
    const size_t want = n ? n : 1;
    if (guard_mode()) {
        const size_t page = 4096, body = (want + 31) & ~(size_t)31, pages = (body + page - 1) / page;
        const size_t len = (pages + 2) * page;
        char* base = (char*)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base == (char*)MAP_FAILED) return hipErrorOutOfMemory;
        mprotect(base, page, PROT_NONE);
        mprotect(base + (pages + 1) * page, page, PROT_NONE);
        q = base + (pages + 1) * page - body;
        RtLock lock(g_rt_mu);
        g_guarded[(uintptr_t)q] = GuardedAlloc{base, len};
    } else if (posix_memalign(&q, 256, want)) {
        return hipErrorOutOfMemory;
    }
    std::memset(q, 0xCD, n);  // poison: uninitialised reads show up as garbage
    *p = q;
    RtLock lock(g_rt_mu);
    g_allocs[(uintptr_t)q] = want;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    RtLock lock(g_rt_mu);
    drain_all();                                         // the runtime synchronises the device before it frees
    g_allocs.erase((uintptr_t)p);
    auto it = g_guarded.find((uintptr_t)p);
    if (it != g_guarded.end()) {
        munmap(it->second.base, it->second.len);
        g_guarded.erase(it);
    } else {
        std::free(p);
    }
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
static hipError_t copy_async(void* d, const void* s, size_t n, hipStream_t st) {
    RtLock lock(g_rt_mu);
    if (!deferred(st)) { std::memmove(d, s, n); return hipSuccess; }
    if (!tracked(s)) {
        // pageable host source: staged before the call returns (the caller may reuse the buffer at once)
        auto snap = std::make_shared<std::vector<char>>((const char*)s, (const char*)s + n);
        enqueue(st, [d, snap, n]() { std::memcpy(d, snap->data(), n); });
    } else {
        enqueue(st, [d, s, n]() { std::memmove(d, s, n); });
    }
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { return copy_async(d, s, n, st); }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
    RtLock lock(g_rt_mu);
    if (!deferred(st)) { std::memset(d, v, n); return hipSuccess; }
    enqueue(st, [d, v, n]() { std::memset(d, v, n); });
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) {
    RtLock lock(g_rt_mu);
    *s = new hipemuStream();
    g_streams.push_back(*s);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) {
    RtLock lock(g_rt_mu);
    drain_stream(s);
    g_streams.erase(std::find(g_streams.begin(), g_streams.end(), s));
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    RtLock lock(g_rt_mu);
    if (!s) return hipSuccess;
    if (async_mode() == 1) others_first_until([s] { return s->q.empty(); }, s);
    else drain_stream(s);
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    RtLock lock(g_rt_mu);
    if (e->completed >= e->enqueued) return hipSuccess;  // nothing outstanding (or never recorded): no dependency
    if (!s) { satisfy(e, e->enqueued, 0); return hipSuccess; }
    ++e->refs;
    s->q.push_back(EmuOp{EmuOp::WAIT, nullptr, e, e->enqueued});
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { RtLock lock(g_rt_mu); drain_all(); return hipSuccess; }
// HIPEMU_DEVICES=G: G emulated devices sharing the host's memory (tests of the single-process multi-device entry points)
static int emu_device_count() {
    const char* e = std::getenv("HIPEMU_DEVICES");
    const int n = e ? std::atoi(e) : 1;
    return n < 1 ? 1 : n;
}
static thread_local int t_device = 0;
hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= emu_device_count()) return hipErrorInvalidDevice;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = emu_device_count(); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t st) { return copy_async(d, s, n, st); }
static bool no_peer() { const char* e = getenv("HIPEMU_NO_PEER"); return e && atoi(e) != 0; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = no_peer() ? 0 : 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return no_peer() ? hipErrorInvalidDevice : hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d) {
    if (d < 0 || d >= emu_device_count()) return hipErrorInvalidDevice;
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "hipemu (CPU lockstep interpreter, test only)");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "hipemu");
    p->multiProcessorCount = 1;
    p->totalGlobalMem = (size_t)8 << 30;
    p->clockRate = 1000000;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    RtLock lock(g_rt_mu);
    if (e->refs > 0) e->destroyed = true;                // released when the queued record / waits have run
    else delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    RtLock lock(g_rt_mu);
    ++e->enqueued;
    if (!deferred(s) && (!s || s->q.empty())) {
        e->t = std::chrono::steady_clock::now();
        e->completed = e->enqueued;
        return hipSuccess;
    }
    ++e->refs;
    s->q.push_back(EmuOp{EmuOp::RECORD, nullptr, e, e->enqueued});
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    RtLock lock(g_rt_mu);
    const uint64_t seq = e->enqueued;
    if (async_mode() == 1) others_first_until([e, seq] { return e->completed >= seq; }, nullptr);
    else satisfy(e, seq, 0);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    RtLock lock(g_rt_mu);
    if (a->completed < a->enqueued || b->completed < b->enqueued) return hipErrorNotReady;
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return hipSuccess; }

// ---------------------------------------------------------------------------------------
// erfcx(x) = exp(x^2) erfc(x): not in libm.  Accurate to a few ulp (test-only quality).
// ---------------------------------------------------------------------------------------
double erfcx(double x) {
    if (std::isnan(x)) return x;
    if (x < 0) {
        if (x < -26.7) return INFINITY;
        return 2.0 * std::exp(x * x) - erfcx(-x);
    }
    if (x < 6.0) {
        // exp(x^2) with the rounding error of x*x carried: x^2 = hi + lo
        double hi = x * x, lo = std::fma(x, x, -hi);
        return std::exp(hi) * std::erfc(x) * (1.0 + lo);
    }
    // asymptotic: 1/(x sqrt(pi)) * sum_k (-1)^k (2k-1)!! / (2x^2)^k
    double inv2x2 = 1.0 / (2.0 * x * x), term = 1.0, sum = 1.0;
    for (int k = 1; k < 60; ++k) {
        double nt = -term * (2 * k - 1) * inv2x2;
        if (std::fabs(nt) >= std::fabs(term) || std::fabs(nt) < 1e-18 * std::fabs(sum)) break;
        term = nt;
        sum += term;
    }
    return sum / (x * 1.7724538509055160273);
}
