// hipemu -- a TEST-ONLY stand-in for <hip/hip_runtime.h> so that the product's HIP
// sources (robo_amd/csrc/*.hip, unmodified, no #ifdefs) can be compiled with g++ and
// their index arithmetic / barrier structure / MFMA fragment maps exercised in the
// GPU-less build container.  It is NOT a backend: nothing under robo_amd/ loads the
// library built from it, it is slow (every lane is a fiber), and it proves nothing
// about performance or about the real hardware's fragment maps (those are taken from
// /opt/skills/guides/cdna_hip_programming.md section 3 and re-checked on the GPU by
// tests/test_gpu_kernels.py::test_mfma_layout_selftest).
//
// Model: one workgroup at a time; each work-item is a ucontext fiber; __syncthreads
// and the wave64 collectives (__shfl*, __ballot, MFMA) are rendezvous points.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

// ---- qualifiers -----------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_smem());
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotReady = 600,
       hipErrorPeerAccessAlreadyEnabled = 704 };
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    int clockRate;
};
enum { hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2 };

typedef double v4d_emu __attribute__((vector_size(32)));

struct alignas(16) double2 { double x, y; };
static inline double2 make_double2(double x, double y) { double2 r; r.x = x; r.y = y; return r; }
struct alignas(16) int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { int4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

namespace hipemu {
struct Lane {
    dim3 tid;
};
Lane& cur();
dim3& cur_block();
dim3& cur_bdim();
dim3& cur_gdim();
void* dyn_smem();
void launch(std::function<void()> body, dim3 grid, dim3 block, size_t shmem);
void launch_on(hipStream_t st, std::function<void()> body, dim3 grid, dim3 block, size_t shmem);
void barrier();
// generic wave64 collective: every live lane of the wave deposits `in` (nbytes) and gets
// `out` back once all arrived; `op` is evaluated once per rendezvous.
enum Op { OP_SHFL, OP_BALLOT, OP_MFMA_F64_16x16x4, OP_WAVE_BARRIER };
void wave_collective(Op op, const void* in, void* out);
int lane_id();
}  // namespace hipemu

#define threadIdx (hipemu::cur().tid)
#define blockIdx (hipemu::cur_block())
#define blockDim (hipemu::cur_bdim())
#define gridDim (hipemu::cur_gdim())
static const int warpSize = 64;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch_on((hipStream_t)(stream), [=]() mutable { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(shmem))

// ---- runtime API --------------------------------------------------------------------
hipError_t hipMalloc(void** p, size_t n);
template <class T> hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
template <class T> hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = 0);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = 0);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipDeviceSynchronize();
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipMemcpyPeerAsync(void* d, int d_dev, const void* s, int s_dev, size_t n, hipStream_t st = 0);
hipError_t hipDeviceCanAccessPeer(int* can, int dev, int peer);      // HIPEMU_NO_PEER=1: the emulated devices are no peers
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);

// ---- device intrinsics ------------------------------------------------------------------
static inline void __syncthreads() { hipemu::barrier(); }

namespace hipemu {
struct ShflIn { uint64_t bits; int src; };
template <class T> static inline T shfl_any(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    ShflIn in; in.bits = 0; std::memcpy(&in.bits, &v, sizeof(T)); in.src = src;
    uint64_t out = 0;
    wave_collective(OP_SHFL, &in, &out);
    T r; std::memcpy(&r, &out, sizeof(T));
    return r;
}
}  // namespace hipemu

template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::lane_id();
    int base = l & ~(width - 1);
    return hipemu::shfl_any(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = hipemu::lane_id();
    int base = l & ~(width - 1);
    int s = (l ^ mask);
    return hipemu::shfl_any(v, (s & ~(width - 1)) == base ? s : l);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int s = l + (int)d;
    return hipemu::shfl_any(v, ((s & ~(width - 1)) == (l & ~(width - 1))) ? s : l);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int s = l - (int)d;
    return hipemu::shfl_any(v, (s >= 0 && (s & ~(width - 1)) == (l & ~(width - 1))) ? s : l);
}
static inline unsigned long long __ballot(int pred) {
    int in = pred; unsigned long long out = 0;
    hipemu::wave_collective(hipemu::OP_BALLOT, &in, &out);
    return out;
}
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __all(int pred) {
    // only live lanes vote; a dead lane counts as "true"
    return __ballot(!pred) == 0ull;
}

// v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D reg r: row=(l>>4)+4r, col=l&15
// (cdna_hip_programming.md section 3 "f64 MFMA does NOT use these maps").
static inline v4d_emu __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, v4d_emu c, int, int, int) {
    struct { double a, b, c[4]; } in = {a, b, {c[0], c[1], c[2], c[3]}};
    double out[4];
    hipemu::wave_collective(hipemu::OP_MFMA_F64_16x16x4, &in, out);
    v4d_emu d = {out[0], out[1], out[2], out[3]};
    return d;
}

static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_fence(int, const char*) {}
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // only used on wave-uniform values
// LDS-direct load: lane l of the wave deposits `size` bytes at dst + l * size
static inline void __builtin_amdgcn_global_load_lds(const void* src, void* dst, unsigned size, int offset, unsigned) {
    std::memcpy(static_cast<char*>(dst) + offset + (size_t)(threadIdx.x & 63) * size, src, size);
}
// lanes are independent fibers here: the lock-step guarantee a real wave gives to
// "store; wave_barrier; load another lane's element" has to be an explicit rendezvous
static inline void __builtin_amdgcn_wave_barrier() {
    int in = 0; unsigned long long out;
    hipemu::wave_collective(hipemu::OP_WAVE_BARRIER, &in, &out);
}
static inline void __builtin_amdgcn_sched_barrier(int) {}
// v_rsq_f64 has ~2^-26 relative accuracy on the hardware; the interpreter returns the rounded value
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
// ROCm's __clang_hip_math.h defines these as the PLAIN operators (unless OCML_BASIC_ROUNDED_OPERATIONS): the compiler
// may contract them.  Mirrored exactly -- round 5 kept the intermediates in `volatile` here, stricter than the MI355X,
// and the interpreter could not see the fused stretch-move proposal the hardware executed.  The product sources use
// common.h's rn_* (contract(off)) where NumPy's rounding matters; the contracting interpreter build (build_emu.build_fma:
// clang++ -ffp-contract=fast-honor-pragmas -mfma, the device compiler's own front end and contraction rule on the host)
// fuses whatever the device compiler would be allowed to.
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __builtin_amdgcn_readlane(int v, int lane) { return hipemu::shfl_any(v, lane); }
static inline int __double2loint(double x) { uint64_t u; std::memcpy(&u, &x, 8); return (int)(uint32_t)u; }
static inline long long __double_as_longlong(double x) { long long u; std::memcpy(&u, &x, 8); return u; }
static inline double __longlong_as_double(long long u) { double x; std::memcpy(&x, &u, 8); return x; }
static inline int __double2hiint(double x) { uint64_t u; std::memcpy(&u, &x, 8); return (int)(uint32_t)(u >> 32); }
static inline double __hiloint2double(int hi, int lo) {
    uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double x; std::memcpy(&x, &u, 8); return x;
}
static inline long long clock64() { return 0; }
static inline long long wall_clock64() { return 0; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- device math missing from <cmath> ----------------------------------------------------
static inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
double erfcx(double x);
using std::erfc; using std::erf; using std::exp; using std::log; using std::log1p; using std::sqrt;
using std::fabs; using std::fma; using std::isnan; using std::isinf; using std::fmax; using std::fmin;
using std::isfinite; using std::cos; using std::sin;

// ---- atomics (single-threaded fibers: plain RMW) --------------------------------------
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
// scoped atomics (clang builtins on the device): workgroups run one at a time here
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
#ifndef __clang__   // (clang knows them as builtins on every target; its host lowering is an ordinary atomic)
template <class T> static inline T __hip_atomic_fetch_add(T* p, T v, int, int) { T o = *p; *p = o + v; return o; }
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T> static inline void __hip_atomic_store(T* p, T v, int, int) { *p = v; }
#endif
