// Does the sumcheck round kernel's load pattern (each lane reads `per` consecutive 32-byte field elements of a table,
// i.e. 16-byte loads at a 32*per-byte lane stride) reach HBM bandwidth by itself?  Grid-stride over 2^nv points.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_aosload.hip -o tools/ubench_aosload && ./tools/ubench_aosload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PER, int K>
__global__ void __launch_bounds__(256) k(const uint64_t *const *tabs, uint64_t half, uint64_t *out) {
    uint64_t acc = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < half; b += (uint64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int t = 0; t < K; t++) {
            const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(tabs[t] + b * PER * 4);
#pragma unroll
            for (int i = 0; i < PER * 2; i++) {
                const ulonglong2 v = p[i];
                acc += v.x ^ v.y;
            }
        }
    if (acc == 0x1234567) out[0] = acc;
}

// the same bytes with wave-contiguous 16-byte loads (lane l reads piece l, l + 64, ...)
template <int PER, int K>
__global__ void __launch_bounds__(256) kc(const uint64_t *const *tabs, uint64_t half, uint64_t *out) {
    uint64_t acc = 0;
    const uint64_t pieces = half * PER * 2;  // 16-byte pieces per table
    for (int t = 0; t < K; t++) {
        const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(tabs[t]);
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pieces; i += (uint64_t)gridDim.x * blockDim.x) {
            const ulonglong2 v = p[i];
            acc += v.x ^ v.y;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}

int main() {
    const int nv = 24, K = 2;
    const uint64_t n = 1ull << nv;
    uint64_t *tab[4], **tabs_d, *out;
    for (int t = 0; t < K; t++) { hipMalloc(&tab[t], n * 32); hipMemset(tab[t], 1, n * 32); }
    hipMalloc(&tabs_d, sizeof(tab)); hipMemcpy(tabs_d, tab, sizeof(tab), hipMemcpyHostToDevice);
    hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char *name, double bytes) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-44s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    const double bytes = (double)K * n * 32;
    for (int blocks : {2048, 8192, 32768}) {
        printf("blocks %d\n", blocks);
        time([&] { hipLaunchKernelGGL((k<2, K>), dim3(blocks), dim3(256), 0, 0, tabs_d, n / 2, out); }, "  per-lane 64 B (round 1 pattern)", bytes);
        time([&] { hipLaunchKernelGGL((k<4, K>), dim3(blocks), dim3(256), 0, 0, tabs_d, n / 4, out); }, "  per-lane 128 B (fold pattern)", bytes);
        time([&] { hipLaunchKernelGGL((kc<2, K>), dim3(blocks), dim3(256), 0, 0, tabs_d, n / 2, out); }, "  wave-contiguous 16 B pieces", bytes);
    }
    return 0;
}
