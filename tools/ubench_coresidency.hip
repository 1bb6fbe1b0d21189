// What lets a small kernel on a second stream co-run with a persistent 256-WG kernel?
// Vary the persistent kernel's threads per WG and dynamic LDS; report when a probe kernel runs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
extern __shared__ unsigned char dyn[];
__global__ void __launch_bounds__(1024) busy(unsigned long long ticks, unsigned long long *stamp, int use_lds, float *sink) {
    const unsigned long long t0 = wall_clock64();
    float x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        for (int i = 0; i < 256; i++) x = x * 1.0001f + 0.5f;   // keep the VALU busy
    }
    if (use_lds) dyn[threadIdx.x] = (unsigned char)x;
    if (x == 12345.f) *sink = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
// same, but holding ~NV live registers per lane (VGPR pressure like the hashing kernel)
template <int NV>
__global__ void __launch_bounds__(1024) busy_regs(unsigned long long ticks, unsigned long long *stamp, float *sink) {
    const unsigned long long t0 = wall_clock64();
    float x[NV];
    for (int i = 0; i < NV; i++) x[i] = threadIdx.x + i;
    while (wall_clock64() - t0 < ticks) {
        for (int k = 0; k < 32; k++)
#pragma unroll
            for (int i = 0; i < NV; i++) x[i] = x[i] * 1.0001f + x[(i + 1) % NV];
    }
    float s = 0;
    for (int i = 0; i < NV; i++) s += x[i];
    dyn[threadIdx.x] = (unsigned char)s;
    if (s == 12345.f) *sink = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
__global__ void stamp_kernel(unsigned long long *out) { if (threadIdx.x == 0) *out = wall_clock64(); }
int main() {
    unsigned long long *st; CK(hipHostMalloc(&st, 4096)); float *sink; CK(hipMalloc(&sink, 64));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void *)busy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct Cfg { int threads; int lds; int blocks; } cfgs[] = {{1024, 0, 256}, {1024, 32 << 10, 256}, {1024, 64 << 10, 256}, {1024, 65 << 10, 256}, {1024, 100 << 10, 256},
                   {1024, 131712, 256}, {512, 131712, 256}, {256, 131712, 256}, {1024, 0, 512}, {512, 0, 256}};
    for (auto c : cfgs) {
        for (int i = 0; i < 8; i++) st[i] = 0;
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, sa, st + 0);
        hipLaunchKernelGGL(busy, dim3(c.blocks), dim3(c.threads), c.lds, sa, 100000ull /* 1 ms */, st, c.lds > 0, sink);
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, sb, st + 2);
        CK(hipDeviceSynchronize());
        printf("persistent %3d x %4d threads, LDS %6d B : busy kernel ended at %7.1f us, probe ran at %7.1f us\n", c.blocks, c.threads, c.lds,
               (st[1] - st[0]) / 100.0, ((long long)st[2] - (long long)st[0]) / 100.0);
    }
    CK(hipFuncSetAttribute((const void *)busy_regs<60>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)busy_regs<100>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int v = 0; v < 2; v++) {
        for (int i = 0; i < 8; i++) st[i] = 0;
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, sa, st + 0);
        if (v == 0) hipLaunchKernelGGL(busy_regs<60>, dim3(256), dim3(1024), 131712, sa, 100000ull, st, sink);
        else hipLaunchKernelGGL(busy_regs<100>, dim3(256), dim3(1024), 131712, sa, 100000ull, st, sink);
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, sb, st + 2);
        CK(hipDeviceSynchronize());
        printf("persistent 256 x 1024, LDS 131712, ~%d live VGPRs, ILP-rich VALU : ended at %7.1f us, probe ran at %7.1f us\n", v ? 100 : 60,
               (st[1] - st[0]) / 100.0, ((long long)st[2] - (long long)st[0]) / 100.0);
    }
    return 0;
}
