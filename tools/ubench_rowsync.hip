// How much does a workgroup barrier after every 15 compressions cost a 1024-thread workgroup per CU (the commit kernel's
// row structure: the waves of a SIMD are served oldest first, finish a hash phase one after the other, and the last one
// runs alone at its dependent-issue rate) -- and does a priority that FALLS with a wave's own progress even it out?
// Build: hipcc -O3 --offload-arch=gfx950 -I zinc_amd/csrc tools/ubench_rowsync.hip -o tools/ubench_rowsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "blake3.cuh"
using namespace zipk;

// MODE 0: no barrier (one stream of ROWS * 15 compressions)      1: barrier per row
//      2: barrier per row, s_setprio 1 for the first 8 compressions of a row, 0 for the rest
//      3: barrier per row, four levels: 3 / 2 / 1 / 0 for compressions 0-3 / 4-7 / 8-11 / 12-14
//      4: as 2, but 1 for the first 11
template <int MODE>
__global__ void __launch_bounds__(1024) rows(uint32_t *out, int nrows) {
    uint32_t l[8], r[8], h[8];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 8; i++) { l[i] = gid * 2654435761u + i; r[i] = gid ^ (i * 0x9E3779B9u); }
    for (int row = 0; row < nrows; row++) {
#pragma unroll
        for (int c = 0; c < 15; c++) {
            if (MODE == 2) { if (c == 0) __builtin_amdgcn_s_setprio(1); if (c == 8) __builtin_amdgcn_s_setprio(0); }
            if (MODE == 4) { if (c == 0) __builtin_amdgcn_s_setprio(1); if (c == 11) __builtin_amdgcn_s_setprio(0); }
            if (MODE == 3) {
                if (c == 0) __builtin_amdgcn_s_setprio(3);
                if (c == 4) __builtin_amdgcn_s_setprio(2);
                if (c == 8) __builtin_amdgcn_s_setprio(1);
                if (c == 12) __builtin_amdgcn_s_setprio(0);
            }
            blake3_node(l, r, h);
#pragma unroll
            for (int i = 0; i < 8; i++) { l[i] = h[i]; r[i] ^= h[7 - i]; }
        }
        if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (h[0] == 0x12345678u) out[gid] = h[1];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class F>
float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    uint32_t *out; CK(hipMalloc(&out, 1 << 24));
    const int nrows = 16;
    const char *names[] = {"no barrier", "barrier per row", "barrier + prio 1|0 (8|7)", "barrier + prio 3|2|1|0", "barrier + prio 1|0 (11|4)"};
    for (int rep = 0; rep < 2; rep++) {
        float ms[5];
        ms[0] = time_ms([&] { hipLaunchKernelGGL(rows<0>, dim3(256), dim3(1024), 0, 0, out, nrows); });
        ms[1] = time_ms([&] { hipLaunchKernelGGL(rows<1>, dim3(256), dim3(1024), 0, 0, out, nrows); });
        ms[2] = time_ms([&] { hipLaunchKernelGGL(rows<2>, dim3(256), dim3(1024), 0, 0, out, nrows); });
        ms[3] = time_ms([&] { hipLaunchKernelGGL(rows<3>, dim3(256), dim3(1024), 0, 0, out, nrows); });
        ms[4] = time_ms([&] { hipLaunchKernelGGL(rows<4>, dim3(256), dim3(1024), 0, 0, out, nrows); });
        for (int m = 0; m < 5; m++)
            printf("%-28s: %.3f ms for %.1fM compressions (16 rows x 15 per lane, 256 x 1024 lanes)  %+.1f %% vs no barrier\n", names[m], ms[m],
                   256.0 * 1024 * 15 * nrows / 1e6, (ms[m] / ms[0] - 1) * 100);
    }
    return 0;
}
