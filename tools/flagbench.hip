// Micro-benchmark (not product code): cost and correctness of a workgroup-to-workgroup hand-over inside one kernel on
// MI355X: 128 KB written by workgroup b, agent-scope release fence, flag; workgroup b+3 (another XCD: consecutive
// workgroups go to consecutive XCDs) spins on the flag, acquire fence, reads and checks the 128 KB.
// usage: flagbench [dirty_kb]   (extra dirty data per workgroup in L2 before the release)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void hand_over(double* data, double* junk, int* flags, long long* stamps, int* errs,
                                                 int dirty_doubles, int rep) {
    const int b = blockIdx.x, tid = threadIdx.x;
    double* mine = data + (size_t)b * 16384;
    long long t0 = clock64();
    for (int i = tid; i < dirty_doubles; i += 256) junk[(size_t)b * dirty_doubles + i] = (double)i;
    for (int i = tid; i < 16384; i += 256) mine[i] = (double)(b * 1000 + rep) + i;
    long long t1 = clock64();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + b, rep, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long long t2 = clock64();
    const int p = (b + 3) % gridDim.x;
    __shared__ int ok;
    if (tid == 0) {
        int budget = 1 << 22, v;
        while ((v = __hip_atomic_load(flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) != rep && --budget)
            __builtin_amdgcn_s_sleep(1);
        ok = v == rep;
    }
    __syncthreads();
    long long t3 = clock64();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const double* theirs = data + (size_t)p * 16384;
    int bad = 0;
    for (int i = tid; i < 16384; i += 256) bad += theirs[i] != (double)(p * 1000 + rep) + i;
    if (bad || !ok) atomicAdd(errs, bad + !ok);
    long long t4 = clock64();
    if (tid == 0) {
        long long* s = stamps + b * 4;
        s[0] = t1 - t0; s[1] = t2 - t1; s[2] = t3 - t2; s[3] = t4 - t3;
    }
}

int main(int argc, char** argv) {
    const int dirty_kb = argc > 1 ? atoi(argv[1]) : 0, G = argc > 2 ? atoi(argv[2]) : 256, dd = dirty_kb * 128;
    double *data, *junk; int *flags, *errs; long long* stamps;
    hipMalloc(&data, (size_t)G * 16384 * 8); hipMalloc(&junk, (size_t)G * (dd + 1) * 8);
    hipMalloc(&flags, G * 4); hipMalloc(&errs, 4); hipMalloc(&stamps, G * 4 * 8);
    hipMemset(flags, 0, G * 4); hipMemset(errs, 0, 4); hipMemset(data, 0, (size_t)G * 16384 * 8);
    for (int rep = 1; rep <= 3; ++rep) {
        hipLaunchKernelGGL(hand_over, dim3(G), dim3(256), 0, 0, data, junk, flags, stamps, errs, dd, rep);
        hipDeviceSynchronize();
        std::vector<long long> h(G * 4); int e = 0;
        hipMemcpy(h.data(), stamps, G * 4 * 8, hipMemcpyDeviceToHost); hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
        const char* nm[4] = {"write", "release+flag", "spin", "acquire+read+check"};
        printf("rep %d dirty %d KB errs %d:", rep, dirty_kb, e);
        for (int q = 0; q < 4; ++q) {
            std::vector<long long> v; for (int b = 0; b < G; ++b) v.push_back(h[b * 4 + q]);
            std::sort(v.begin(), v.end());
            printf("  %s med %lld max %lld", nm[q], v[G / 2], v[G - 1]);
        }
        printf("\n");
    }
    return 0;
}
