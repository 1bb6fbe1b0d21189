// Microbenchmark: the library's mont_mul<4> (64-bit limbs, u128 products) against a product-scanning variant on
// 32-bit limbs with a 96-bit column accumulator (one v_mad_u64_u32 + one add-with-carry per partial product).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_montmul.hip -o tools/ubench_montmul && ./tools/ubench_montmul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../zinc_amd/csrc/kernels_open.cuh"
using namespace zipk;

// 8 x 32-bit limbs, Montgomery product a*b*R^-1 mod q (R = 2^256), canonical result.
// Column-wise (Comba): column k collects a_i*b_j (i+j=k) and m_i*q_j; acc = (hi32, lo64).
struct Acc96 {
    uint64_t lo;
    uint32_t hi;
    __device__ __forceinline__ void mac(uint32_t x, uint32_t y) {
        const uint64_t p = (uint64_t)x * y;
        const uint64_t s = lo + p;
        hi += s < p;
        lo = s;
    }
    __device__ __forceinline__ void shift32() {
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
};

__device__ __forceinline__ void mont_mul32(const uint32_t (&a)[8], const uint32_t (&b)[8], const uint32_t (&q)[8],
                                           uint32_t inv32, uint32_t (&out)[8]) {
    uint32_t m[8], t[9];
    Acc96 acc{0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc.mac(a[i], b[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) acc.mac(m[i], q[k - i]);
        m[k] = (uint32_t)acc.lo * inv32;
        acc.mac(m[k], q[0]);
        acc.shift32();
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
        for (int i = k - 7; i < 8; i++) acc.mac(a[i], b[k - i]);
#pragma unroll
        for (int i = k - 7; i < 8; i++) acc.mac(m[i], q[k - i]);
        t[k - 8] = (uint32_t)acc.lo;
        acc.shift32();
    }
    t[8] = (uint32_t)acc.lo;
    // conditional subtraction
    uint32_t d[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)t[i] - q[i] - borrow;
        d[i] = (uint32_t)x;
        borrow = (x >> 32) & 1;
    }
    const bool ge = t[8] != 0 || borrow == 0;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = ge ? d[i] : t[i];
}

// CIOS on 64-bit limbs: multiplication and reduction interleaved row by row
__device__ __forceinline__ void mont_mul_cios64(const uint64_t (&a)[4], const uint64_t (&b)[4], const FieldDev<4> &f,
                                                uint64_t (&out)[4]) {
    typedef unsigned __int128 u128;
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u128 x = (u128)a[j] * b[i] + t[j] + c;
            t[j] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        u128 y = (u128)t[4] + c;
        t[4] = (uint64_t)y;
        t[5] = (uint64_t)(y >> 64);
        const uint64_t m = t[0] * f.inv;
        u128 x = (u128)m * f.modulus[0] + t[0];
        c = (uint64_t)(x >> 64);
#pragma unroll
        for (int j = 1; j < 4; j++) {
            x = (u128)m * f.modulus[j] + t[j] + c;
            t[j - 1] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        y = (u128)t[4] + c;
        t[3] = (uint64_t)y;
        t[4] = t[5] + (uint64_t)(y >> 64);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = t[i];
    if (t[4] || geq_n<4>(out, f.modulus)) sub_n<4>(out, f.modulus);
}

__global__ void kcios(const uint64_t *in, uint64_t *out, int iters, FieldDev<4> f) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[4], b[4], t[4];
    for (int k = 0; k < 4; k++) { a[k] = in[i * 8 + k]; b[k] = in[i * 8 + 4 + k]; }
    for (int it = 0; it < iters; it++) {
        mont_mul_cios64(a, b, f, t);
        for (int k = 0; k < 4; k++) { a[k] = b[k]; b[k] = t[k]; }
    }
    for (int k = 0; k < 4; k++) out[i * 4 + k] = b[k];
}

__global__ void k64(const uint64_t *in, uint64_t *out, int iters, FieldDev<4> f) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[4], b[4], t[4];
    for (int k = 0; k < 4; k++) { a[k] = in[i * 8 + k]; b[k] = in[i * 8 + 4 + k]; }
    for (int it = 0; it < iters; it++) {
        mont_mul<4>(a, b, f, t);
        for (int k = 0; k < 4; k++) { a[k] = b[k]; b[k] = t[k]; }
    }
    for (int k = 0; k < 4; k++) out[i * 4 + k] = b[k];
}

__global__ void k32(const uint64_t *in, uint64_t *out, int iters, FieldDev<4> f, uint32_t inv32) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a[8], b[8], t[8], q[8];
    for (int k = 0; k < 4; k++) {
        a[2 * k] = (uint32_t)in[i * 8 + k]; a[2 * k + 1] = (uint32_t)(in[i * 8 + k] >> 32);
        b[2 * k] = (uint32_t)in[i * 8 + 4 + k]; b[2 * k + 1] = (uint32_t)(in[i * 8 + 4 + k] >> 32);
        q[2 * k] = (uint32_t)f.modulus[k]; q[2 * k + 1] = (uint32_t)(f.modulus[k] >> 32);
    }
    for (int it = 0; it < iters; it++) {
        mont_mul32(a, b, q, inv32, t);
        for (int k = 0; k < 8; k++) { a[k] = b[k]; b[k] = t[k]; }
    }
    for (int k = 0; k < 4; k++) out[i * 4 + k] = (uint64_t)b[2 * k] | ((uint64_t)b[2 * k + 1] << 32);
}

int main() {
    // the Stark prime 2^251 + 17 * 2^192 + 1
    FieldDev<4> f{};
    f.modulus[0] = 1; f.modulus[1] = 0; f.modulus[2] = 0; f.modulus[3] = 0x0800000000000011ull;
    uint64_t inv = 1;
    for (int i = 0; i < 63; i++) { inv *= inv; inv *= f.modulus[0]; }
    f.inv = (uint64_t)0 - inv;
    const uint32_t inv32 = (uint32_t)f.inv;
    const int blocks = 256 * 16, threads = 256, iters = 2000;
    const size_t n = (size_t)blocks * threads;
    std::vector<uint64_t> h(n * 8);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
    for (size_t i = 0; i < n * 2; i++) h[i * 4 + 3] &= 0x07FFFFFFFFFFFFFFull;  // < q
    uint64_t *in, *o1, *o2;
    hipMalloc(&in, n * 64); hipMalloc(&o1, n * 32); hipMalloc(&o2, n * 32);
    hipMemcpy(in, h.data(), n * 64, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        float ms1, ms2;
        hipEventRecord(e0); hipLaunchKernelGGL(k64, dim3(blocks), dim3(threads), 0, 0, in, o1, iters, f); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, in, o2, iters, f, inv32); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms2, e0, e1);
        float ms3;
        hipEventRecord(e0); hipLaunchKernelGGL(kcios, dim3(blocks), dim3(threads), 0, 0, in, o2, iters, f); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms3, e0, e1);
        {
            std::vector<uint64_t> ra(n * 4), rb(n * 4);
            hipMemcpy(ra.data(), o1, n * 32, hipMemcpyDeviceToHost);
            hipMemcpy(rb.data(), o2, n * 32, hipMemcpyDeviceToHost);
            printf("CIOS64 %s; ", memcmp(ra.data(), rb.data(), n * 32) ? "DIFFERS" : "identical");
        }
        hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, in, o2, iters, f, inv32); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms2, e0, e1);
        printf("64-bit limbs (library): %.2f ms (%.1f G/s)   CIOS 64-bit: %.2f ms (%.1f G/s)   32-bit Comba: %.2f ms (%.1f G/s)\n", ms1,
               n * (double)iters / ms1 / 1e6, ms3, n * (double)iters / ms3 / 1e6, ms2, n * (double)iters / ms2 / 1e6);
    }
    std::vector<uint64_t> r1(n * 4), r2(n * 4);
    hipMemcpy(r1.data(), o1, n * 32, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), o2, n * 32, hipMemcpyDeviceToHost);
    printf("results %s\n", memcmp(r1.data(), r2.data(), n * 32) ? "DIFFER" : "identical");
    return 0;
}
