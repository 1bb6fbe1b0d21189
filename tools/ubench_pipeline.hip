// When do the workgroups of the persistent commit kernel publish their chunks, and when does a
// waiter on another stream see them?  (2^24 geometry.)
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -DZIPK_DEBUG_STAMPS -I zinc_amd/csrc tools/ubench_pipeline.hip -o tools/ubench_pipeline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#include "kernels_commit.cuh"
using namespace zipk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) copy_stream(const uint4 *in, uint4 *out, size_t n, int reps) {
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// gather-like co-runner: scattered 32-byte reads (2 lanes per node), coalesced 16-byte writes
__global__ void __launch_bounds__(256) gather_stream(const uint4 *in, uint4 *out, size_t n_nodes, int reps) {
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes * 2; i += (size_t)gridDim.x * blockDim.x) {
            const size_t node = ((i >> 1) * 2654435761ull + r) % n_nodes;
            out[i] = in[node * 2 + (i & 1)];
        }
}
__global__ void stamp_kernel(unsigned long long *out) { if (threadIdx.x == 0) *out = wall_clock64(); }

int main(int argc, char **argv) {
    const uint32_t R = 4096, C = 4096, cw = 8192, G = 256;
    const uint32_t rpc = argc > 1 ? (uint32_t)atoi(argv[1]) : 4;  // rounds per chunk
    const int modes = argc > 2 ? atoi(argv[2]) : 4;
    std::vector<uint32_t> p1(cw), p2(cw);
    std::iota(p1.begin(), p1.end(), 0); std::iota(p2.begin(), p2.end(), 0);
    std::mt19937 g(1); std::shuffle(p1.begin(), p1.end(), g); std::shuffle(p2.begin(), p2.end(), g);
    std::vector<int64_t> ev((size_t)R * C);
    for (auto &x : ev) x = (int64_t)(((uint64_t)g() << 32) | g());
    CommitArgs a{};
    int64_t *evd; uint32_t *p1d, *p2d, *flag;
    unsigned long long *stamps_h; CK(hipHostMalloc(&stamps_h, (size_t)kStampRec * G * 8));
    for (size_t i = 0; i < (size_t)kStampRec * G; i++) stamps_h[i] = 0;
    uint4 *ca, *cb; const size_t cn = (size_t)1 << 26; CK(hipMalloc(&ca, cn * 16)); CK(hipMalloc(&cb, cn * 16)); CK(hipMemset(ca, 1, cn * 16));
    CK(hipMalloc(&evd, ev.size() * 8)); CK(hipMalloc(&p1d, cw * 4)); CK(hipMalloc(&p2d, cw * 4)); CK(hipMalloc(&flag, 64));
    CK(hipMalloc(&a.rows, (size_t)R * cw * 32)); CK(hipMalloc(&a.layers, (size_t)R * 2 * cw * 32)); CK(hipMalloc(&a.chunk_done, 256));
    CK(hipMemcpy(evd, ev.data(), ev.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(p1d, p1.data(), cw * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(p2d, p2.data(), cw * 4, hipMemcpyHostToDevice));
    a.evals = evd; a.perm1 = p1d; a.perm2 = p2d; a.row_len = C; a.cw = cw; a.nact = cw / 8;
    a.num_rows = R; a.rounds_per_chunk = rpc; a.stamps = stamps_h; CK(hipMalloc(&a.roots, R * 32));
    const size_t lds = 512 + (size_t)8 * (1024 + 4) * 12 + (size_t)C * 8 + 4 * kFinisherFlagWords;
    auto kh = raa_commit_kernel<8, true>;
    CK(hipFuncSetAttribute((const void *)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"alone", "+copy 512 blocks x 6 GiB", "+gather 512 blocks", "+gather 2048 blocks"};
    printf("rounds per chunk %u\n", rpc);
    for (int mode = 0; mode < modes; mode++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(a.chunk_done, 0, 256)); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(kh, dim3(G), dim3(1024), lds, sa, a);
            CK(hipEventRecord(e1, sa));
            if (mode == 1) hipLaunchKernelGGL(copy_stream, dim3(512), dim3(256), 0, sb, ca, cb, cn, 3);
            if (mode == 2) hipLaunchKernelGGL(gather_stream, dim3(512), dim3(256), 0, sb, ca, cb, cn / 2, 3);
            if (mode == 3) hipLaunchKernelGGL(gather_stream, dim3(2048), dim3(256), 0, sb, ca, cb, cn / 2, 3);
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        double pa = 0, pb = 0, pc = 0;
        for (uint32_t w = 0; w < G; w++) { pa += stamps_h[(size_t)kStampRec * w + 3]; pb += stamps_h[(size_t)kStampRec * w + 4]; pc += stamps_h[(size_t)kStampRec * w + 5]; }
        printf("%-28s commit %.3f ms | per-WG avg: encode phases %.1f us, hash phase %.1f us, chunk end+stage %.1f us\n", names[mode], ms,
               pa / G / 100.0, pb / G / 100.0, pc / G / 100.0);
    }
    return 0;
}
