// Which memory / access kinds let a kernel on one stream observe counters bumped by a
// still-running kernel on another stream (MI355X, multi-XCD)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void producer(uint32_t *counters, unsigned long long *stamps, int iters, long long spin_ticks) {
    for (int it = 0; it < iters; it++) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&counters[it], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) stamps[it] = wall_clock64();
        }
    }
}
template <int MODE>
__global__ void waiter(uint32_t *counter, uint32_t target, uint32_t zero, unsigned long long *stamp) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            uint32_t v;
            if (MODE == 0) v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1) v = __hip_atomic_fetch_add(counter, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); v = *(volatile uint32_t *)counter; }
            if (v >= target) break;
            if (wall_clock64() - t0 > 100000000ull) break;
            __builtin_amdgcn_s_sleep(32);
        }
        *stamp = wall_clock64();
    }
}
int main() {
    const int iters = 8, blocks = 256;
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned long long *stamps; CK(hipHostMalloc(&stamps, 4096));
    const char *memn[] = {"hipMalloc", "finegrained", "uncached", "hostpinned"};
    for (int mem = 0; mem < 4; mem++) {
        uint32_t *ctr;
        if (mem == 0) CK(hipMalloc(&ctr, 4096));
        else if (mem == 1) CK(hipExtMallocWithFlags((void **)&ctr, 4096, hipDeviceMallocFinegrained));
        else if (mem == 2) CK(hipExtMallocWithFlags((void **)&ctr, 4096, hipDeviceMallocUncached));
        else CK(hipHostMalloc(&ctr, 4096));
        for (int mode = 0; mode < 3; mode++) {
            CK(hipMemset(ctr, 0, 4096)); CK(hipDeviceSynchronize());
            for (int i = 0; i < 64; i++) stamps[i] = 0;
            hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, sa, ctr, stamps, iters, 20000LL /* 200 us per iter */);
            if (mode == 0) hipLaunchKernelGGL(waiter<0>, dim3(1), dim3(64), 0, sb, ctr + 0, (uint32_t)blocks, 0u, stamps + 32);
            if (mode == 1) hipLaunchKernelGGL(waiter<1>, dim3(1), dim3(64), 0, sb, ctr + 0, (uint32_t)blocks, 0u, stamps + 32);
            if (mode == 2) hipLaunchKernelGGL(waiter<2>, dim3(1), dim3(64), 0, sb, ctr + 0, (uint32_t)blocks, 0u, stamps + 32);
            CK(hipDeviceSynchronize());
            printf("%-11s poll=%s : producer iter0 done at %8.1f us, last iter at %8.1f us; waiter released at %8.1f us\n", memn[mem],
                   mode == 0 ? "sc1-load " : mode == 1 ? "atomic-rmw" : "acq+load ", 0.0, (stamps[iters - 1] - stamps[0]) / 100.0,
                   ((long long)stamps[32] - (long long)stamps[0]) / 100.0);
        }
        if (mem == 3) CK(hipHostFree(ctr)); else CK(hipFree(ctr));
    }
    return 0;
}
