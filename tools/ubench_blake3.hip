// Micro-benchmarks that bound the commit kernel on the actual chip:
//   hash_only   : chained BLAKE3 node compressions in registers (int32 VALU ceiling)
//   hash_store  : one compression per 32-byte store, thread-contiguous (8 hashes = 256 B per lane)
//                 or lane-interleaved (coalesced) placement
// Build: hipcc -O3 --offload-arch=gfx950 -I zinc_amd/csrc tools/ubench_blake3.hip -o tools/ubench_blake3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "blake3.cuh"
using namespace zipk;

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) hash_only(uint32_t *out, int iters) {
    uint32_t l[8], r[8], h[8];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 8; i++) { l[i] = gid * 2654435761u + i; r[i] = gid ^ (i * 0x9E3779B9u); }
    for (int it = 0; it < iters; it++) {
        blake3_node(l, r, h);
        for (int i = 0; i < 8; i++) { l[i] = h[i]; r[i] ^= h[7 - i]; }
    }
    if (h[0] == 0x12345678u) out[gid] = h[1];  // keep the chain live
}

template <bool CONTIG>
__global__ void __launch_bounds__(1024) hash_store(uint32_t *out, int rows_per_block) {
    // mimics raa_commit_kernel's output phase: 1024 threads x 8 hashes per row
    const uint32_t tid = threadIdx.x;
    uint32_t l[8], r[8], h[8];
    for (int i = 0; i < 8; i++) { l[i] = tid * 2654435761u + i; r[i] = blockIdx.x ^ (i * 0x9E3779B9u); }
    for (int row = 0; row < rows_per_block; row++) {
        uint32_t *base = out + ((size_t)(blockIdx.x * rows_per_block + row) * 8192) * 8;
        for (int e = 0; e < 8; e++) {
            blake3_node(l, r, h);
            for (int i = 0; i < 8; i++) { l[i] = h[i]; r[i] ^= h[7 - i]; }
            const uint32_t idx = CONTIG ? tid * 8 + e : e * 1024 + tid;
            uint4 *d = reinterpret_cast<uint4 *>(base + (size_t)idx * 8);
            d[0] = make_uint4(h[0], h[1], h[2], h[3]);
            d[1] = make_uint4(h[4], h[5], h[6], h[7]);
        }
    }
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
    uint32_t *out; CK(hipMalloc(&out, (size_t)4096 * 8192 * 32));
    const double total = 67.1e6;  // compressions of a 2^24 commit
    {
        const int iters = 64, threads = (int)(total / iters);
        float ms = time_ms([&] { hipLaunchKernelGGL(hash_only<256>, dim3(threads / 256), dim3(256), 0, 0, out, iters); });
        printf("hash_only<256>  : %.3f ms for %.1fM compressions -> %.2f G/s\n", ms, (threads / 256 * 256.0) * iters / 1e6, (threads / 256 * 256.0) * iters / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(hash_only<1024>, dim3(threads / 1024), dim3(1024), 0, 0, out, iters); });
        printf("hash_only<1024> : %.3f ms\n", ms);
        // one 1024-thread block per CU (the commit kernel's residency at cw = 8192)
        ms = time_ms([&] { hipLaunchKernelGGL(hash_only<1024>, dim3(256), dim3(1024), 0, 0, out, 256); });
        printf("hash_only<1024> 256 blocks x 256 iters (1 block/CU): %.3f ms for %.1fM -> %.2f G/s\n", ms, 256 * 1024 * 256 / 1e6, 256.0 * 1024 * 256 / ms / 1e6);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(hash_store<true>, dim3(4096), dim3(1024), 0, 0, out, 1); });
        printf("hash_store contiguous-per-lane : %.3f ms for 33.5M compressions + 1 GiB of stores\n", ms);
        ms = time_ms([&] { hipLaunchKernelGGL(hash_store<false>, dim3(4096), dim3(1024), 0, 0, out, 1); });
        printf("hash_store lane-interleaved    : %.3f ms\n", ms);
        ms = time_ms([&] { hipLaunchKernelGGL(hash_store<true>, dim3(256), dim3(1024), 0, 0, out, 16); });
        printf("hash_store contiguous, 256 persistent blocks x 16 rows: %.3f ms\n", ms);
    }
    return 0;
}
