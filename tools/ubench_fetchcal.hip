// What does rocprofv3's FETCH_SIZE count for the access patterns of this library?  (MI355X_MICROARCH.md, HBM
// section: exactly 1/2 of the bytes of a wide 16-B-per-lane streaming read; other widths uncalibrated.)
// Three kernels over buffers far larger than the 256 MiB Infinity Cache, each byte read exactly once:
//   cal_stream16   16 B per lane, consecutive lanes consecutive        (the guide's calibrated pattern)
//   cal_stream8     8 B per lane, consecutive                           (the commit kernel's witness reads)
//   cal_gather32   two lanes x 16 B per 32-byte node, nodes 128 B apart in random order
//                  (open_columns_kernel's sibling reads: one 32-byte node out of a 128-byte line)
//   cal_gather32d  the same with nodes 64 B apart (two nodes per line, read by different wave instructions)
// Run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- tools/ubench_fetchcal
// and compare FETCH_SIZE (KiB) per kernel with the "useful KiB" this program prints.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_fetchcal.hip -o tools/ubench_fetchcal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) cal_stream16(const uint4 *in, uint32_t *sink, size_t n) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) cal_stream8(const uint2 *in, uint32_t *sink, size_t n) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 v = in[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// node k of the visit order sits at byte (perm(k) * stride): perm = multiplication by an odd constant mod 2^bits
__global__ void __launch_bounds__(256) cal_gather32(const uint4 *in, uint32_t *sink, uint32_t bits, uint32_t stride16) {
    uint32_t acc = 0;
    const size_t n_nodes = (size_t)1 << bits;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes * 2; i += (size_t)gridDim.x * blockDim.x) {
        const size_t node = ((i >> 1) * 2654435761ull) & (n_nodes - 1);
        const uint4 v = in[node * stride16 + (i & 1)];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)4 << 30;  // 4 GiB
    uint4 *buf; uint32_t *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
    const dim3 grid(4096), block(256);
    hipLaunchKernelGGL(cal_stream16, grid, block, 0, 0, buf, sink, bytes / 16);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(cal_stream8, grid, block, 0, 0, reinterpret_cast<const uint2 *>(buf), sink, bytes / 8);
    CK(hipDeviceSynchronize());
    // 2^25 nodes x 128-byte spacing = 4 GiB footprint, 1 GiB useful
    hipLaunchKernelGGL(cal_gather32, grid, block, 0, 0, buf, sink, 25u, 8u);
    CK(hipDeviceSynchronize());
    // 2^26 nodes x 64-byte spacing = 4 GiB footprint, 2 GiB useful
    hipLaunchKernelGGL(cal_gather32, grid, block, 0, 0, buf, sink, 26u, 4u);
    CK(hipDeviceSynchronize());
    printf("useful KiB: cal_stream16 %zu, cal_stream8 %zu, cal_gather32 (128-B spacing) %zu, cal_gather32 (64-B spacing) %zu\n",
           bytes / 1024, bytes / 1024, ((size_t)32 << 25) / 1024, ((size_t)32 << 26) / 1024);
    return 0;
}
