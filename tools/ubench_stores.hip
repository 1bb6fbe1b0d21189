// Does a power-of-two row stride (256 KiB rows / 512 KiB trees) camp on memory channels?
// 4096 blocks x 1024 threads, each block writes one "row" of 8192 x 32 B in the commit
// kernel's strided pattern; row stride = 256 KiB + pad.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) store_rows(uint4 *out, size_t row_stride_u4, uint32_t seed) {
    uint4 *row = out + (size_t)blockIdx.x * row_stride_u4;
    const uint32_t t = threadIdx.x;
    for (int e = 0; e < 8; e++) {
        const uint32_t j = e * 1024 + t;
        const uint32_t v = j * 2654435761u + seed;
        row[2 * j] = make_uint4(v, v + 1, v + 2, v + 3);
        row[2 * j + 1] = make_uint4(v + 4, v + 5, v + 6, v + 7);
    }
}
__global__ void __launch_bounds__(256) store_linear(uint4 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
template <class F> float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; i++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms); }
    return best;
}
int main() {
    uint4 *out; CK(hipMalloc(&out, (size_t)4096 * (262144 + 65536)));
    const size_t pads[] = {0, 256, 4096, 4352, 65536 - 256};
    for (size_t pad : pads) {
        const size_t stride = (262144 + pad) / 16;
        float ms = time_ms([&] { hipLaunchKernelGGL(store_rows, dim3(4096), dim3(1024), 0, 0, out, stride, 1u); });
        printf("row stride 256 KiB + %6zu B : %.3f ms  -> %.2f TB/s\n", pad, ms, 1.073741824 / ms);
    }
    float ms = time_ms([&] { hipLaunchKernelGGL(store_linear, dim3(2048), dim3(256), 0, 0, out, (size_t)1 << 26); });
    printf("linear float4 stores 1 GiB      : %.3f ms  -> %.2f TB/s\n", ms, 1.073741824 / ms);
    return 0;
}
