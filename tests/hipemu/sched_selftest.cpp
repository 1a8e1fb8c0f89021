// TEST-ONLY: a two-stream fork/join whose dependencies can be left out on purpose, to check that the deferred schedules of
// hipemu.cpp (HIPEMU_ASYNC = 1 "others first", 2 "others last") expose exactly the dependency each one is meant to expose
// and that the immediate schedule (0) hides both.  tests/test_emu_schedules.py.
#include <hip/hip_runtime.h>

__global__ void sched_fill(double* p, double v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void sched_copy(const double* a, double* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

extern "C" void hipemu_set_async(int mode);

// main: A = 1 | fork | ........... | join | C = B | read C        side: | wait fork | B = A | record join |
extern "C" int sched_selftest(int mode, int with_fork_wait, int with_join_wait, double* out) {
    const int n = 64;
    hipemu_set_async(mode);
    hipStream_t main_s, side_s;
    hipEvent_t fork_e, join_e;
    double *A, *B, *C;
    hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&side_s, hipStreamNonBlocking);
    hipEventCreateWithFlags(&fork_e, hipEventDisableTiming);
    hipEventCreateWithFlags(&join_e, hipEventDisableTiming);
    hipMalloc(&A, n * sizeof(double)); hipMalloc(&B, n * sizeof(double)); hipMalloc(&C, n * sizeof(double));
    hipMemset(A, 0, n * sizeof(double)); hipMemset(B, 0, n * sizeof(double)); hipMemset(C, 0, n * sizeof(double));
    hipLaunchKernelGGL(sched_fill, dim3(1), dim3(64), 0, main_s, A, 1.0, n);
    hipEventRecord(fork_e, main_s);
    if (with_fork_wait) hipStreamWaitEvent(side_s, fork_e, 0);
    hipLaunchKernelGGL(sched_copy, dim3(1), dim3(64), 0, side_s, (const double*)A, B, n);
    hipEventRecord(join_e, side_s);
    if (with_join_wait) hipStreamWaitEvent(main_s, join_e, 0);
    hipLaunchKernelGGL(sched_copy, dim3(1), dim3(64), 0, main_s, (const double*)B, C, n);
    hipMemcpyAsync(out, C, n * sizeof(double), hipMemcpyDeviceToHost, main_s);
    hipStreamSynchronize(main_s);
    hipStreamSynchronize(side_s);
    hipFree(A); hipFree(B); hipFree(C);
    hipEventDestroy(fork_e); hipEventDestroy(join_e);
    hipStreamDestroy(main_s); hipStreamDestroy(side_s);
    hipemu_set_async(0);
    return 0;
}

// HIPEMU_GUARD=1: a kernel that stores one element past its buffer must fault at the store (tests/test_emu_schedules.py
// runs this in a child process and expects SIGSEGV with overrun = 1, a clean return with overrun = 0).
extern "C" int guard_probe(int overrun) {
    const int n = 100;                                   // 800 bytes: the buffer ends exactly at the guard page
    double* A;
    hipMalloc(&A, n * sizeof(double));
    hipLaunchKernelGGL(sched_fill, dim3(2), dim3(64), 0, (hipStream_t)0, A, 1.0, n + (overrun ? 1 : 0));
    hipFree(A);
    return 0;
}

// HIPEMU_ORDER: wave 1 consumes what wave 0 produced in LDS; without the barrier the result depends on which wave's
// work-items the interpreter runs first.  Returns the number of stale elements wave 1 saw.
__global__ void lds_handoff(double* out, int with_barrier) {
    __shared__ double buf[64];
    const int t = threadIdx.x;
    if (t < 64) buf[t] = 0.0;
    __syncthreads();
    if (t < 64) buf[t] = t + 1.0;
    if (with_barrier) __syncthreads();
    if (t >= 64) out[t - 64] = buf[t - 64];
}
extern "C" int order_probe(int with_barrier) {
    double* out;
    hipMalloc(&out, 64 * sizeof(double));
    hipLaunchKernelGGL(lds_handoff, dim3(1), dim3(128), 0, (hipStream_t)0, out, with_barrier);
    int stale = 0;
    for (int i = 0; i < 64; ++i) stale += out[i] != i + 1.0;
    hipFree(out);
    return stale;
}
