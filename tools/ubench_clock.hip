// Is the commit kernel slowed by a co-running memory-bound kernel through the CLOCK (power cap)
// rather than through a shared pipe?  Measures the shader clock (s_memtime / 100 MHz wall clock)
// of a VALU-bound persistent hashing kernel alone and beside a streaming copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "blake3.cuh"
using namespace zipk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) hash_persist(uint32_t *out, int iters, unsigned long long *stamps) {
    uint32_t l[8], r[8], h[8];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 8; i++) { l[i] = gid * 2654435761u + i; r[i] = gid ^ (i * 0x9E3779B9u); }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        blake3_node(l, r, h);
        for (int i = 0; i < 8; i++) { l[i] = h[i]; r[i] ^= h[7 - i]; }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (h[0] == 0x12345678u) out[gid] = h[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = w1 - w0; }
}
// same work, but 15 compressions unrolled per loop iteration: ~80 KB of straight-line code,
// like the commit kernel's hash phase (the instruction cache is 64 KB)
__global__ void __launch_bounds__(1024) hash_persist_unrolled(uint32_t *out, int iters, unsigned long long *stamps) {
    uint32_t l[8], r[8], h[8];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 8; i++) { l[i] = gid * 2654435761u + i; r[i] = gid ^ (i * 0x9E3779B9u); }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; it += 15) {
#pragma unroll
        for (int u = 0; u < 15; u++) {
            blake3_node(l, r, h);
            for (int i = 0; i < 8; i++) { l[i] = h[i]; r[i] ^= h[7 - i] + u; }
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (h[0] == 0x12345678u) out[gid] = h[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = w1 - w0; }
}
__global__ void __launch_bounds__(256) copy_stream(const uint4 *in, uint4 *out, size_t n, int reps) {
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    uint32_t *out; CK(hipMalloc(&out, 1 << 22));
    uint4 *a, *b; const size_t n = (size_t)1 << 26;  // 1 GiB each
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16));
    unsigned long long *st; CK(hipHostMalloc(&st, 256));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int iters = 260;  // ~1.2 ms of hashing
    for (int mode = 0; mode < 8; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            if (mode < 4) hipLaunchKernelGGL(hash_persist, dim3(256), dim3(1024), 0, sa, out, iters, st);
            else hipLaunchKernelGGL(hash_persist_unrolled, dim3(256), dim3(1024), 0, sa, out, 255, st);
            CK(hipEventRecord(e1, sa));
            const int cm = mode & 3;
            if (cm == 1) hipLaunchKernelGGL(copy_stream, dim3(512), dim3(256), 0, sb, a, b, n, 2);       // 2 blocks per CU, 4 GiB of traffic
            if (cm == 2) hipLaunchKernelGGL(copy_stream, dim3(1024), dim3(256), 0, sb, a, b, n, 3);      // 4 blocks per CU
            if (cm == 3) hipLaunchKernelGGL(copy_stream, dim3(256), dim3(256), 0, sb, a, b, n / 4, 2);   // light
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2)
                printf("%s mode %d (%s): hashing kernel %.3f ms, shader clock %.0f MHz\n", mode < 4 ? "looped  " : "unrolled", cm,
                       cm == 0 ? "alone" : cm == 1 ? "+copy 512 blocks" : cm == 2 ? "+copy 1024 blocks" : "+light copy", ms,
                       (double)st[0] / (double)st[1] * 100.0);
        }
    }
    return 0;
}
