// How do the double-rate VOP2 opcodes (v_xor_b32, v_add_u32: tools/ubench_valu_ops) and the single-rate VOP3 ones
// (v_alignbit_b32, v_add3_u32) mix on one SIMD?  Streams of 8 independent chains per wave, equal numbers of
// v_xor_b32 and v_alignbit_b32, grouped in runs of RUN instructions of one kind.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu_mix.hip -o tools/ubench_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y))
#define ROT(i) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[i]))
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y))
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z))

// RUN = 1: x0 ^, x0 rot, x1 ^, x1 rot ...   RUN = 8: x0..x7 ^, x0..x7 rot   RUN = 64: 8 rounds of xors, 8 rounds of rots
template <int RUN>
__global__ void __launch_bounds__(256) k_mix(uint32_t *out, int iters) {
    uint32_t x[8], y = threadIdx.x * 2654435761u + 1u;
    for (int c = 0; c < 8; c++) x[c] = threadIdx.x + c * 77u;
    for (int it = 0; it < iters; it++) {
        if (RUN == 1) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int c = 0; c < 8; c++) { XOR(c); ROT(c); }
            }
        } else if (RUN == 2) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int c = 0; c < 8; c += 2) { XOR(c); XOR(c + 1); ROT(c); ROT(c + 1); }
            }
        } else if (RUN == 4) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int c = 0; c < 8; c += 4) { XOR(c); XOR(c + 1); XOR(c + 2); XOR(c + 3); ROT(c); ROT(c + 1); ROT(c + 2); ROT(c + 3); }
            }
        } else if (RUN == 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int c = 0; c < 8; c++) XOR(c);
#pragma unroll
                for (int c = 0; c < 8; c++) ROT(c);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int c = 0; c < 8; c++) XOR(c);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int c = 0; c < 8; c++) ROT(c);
            }
        }
    }
    uint32_t s = 0;
    for (int c = 0; c < 8; c++) s ^= x[c];
    if (s == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the G function of BLAKE3 as the compiler emits it: four independent Gs interleaved instruction by instruction
// (12 instructions each: 2 add3, 2 add, 4 xor, 4 alignbit), or with each kind grouped four at a time
template <bool GROUPED>
__global__ void __launch_bounds__(256) k_g(uint32_t *out, int iters) {
    uint32_t a[4], b[4], c[4], d[4], y = threadIdx.x * 2654435761u + 1u, z = blockIdx.x ^ 0x9E3779B9u;
    for (int i = 0; i < 4; i++) { a[i] = threadIdx.x + i; b[i] = a[i] * 3; c[i] = a[i] * 5; d[i] = a[i] * 7; }
#define A3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(y))
#define XD(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d[i]) : "v"(a[i]))
#define RD(i, n) asm volatile("v_alignbit_b32 %0, %0, %0, " #n : "+v"(d[i]))
#define AC(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[i]) : "v"(d[i]))
#define XB(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(b[i]) : "v"(c[i]))
#define RB(i, n) asm volatile("v_alignbit_b32 %0, %0, %0, " #n : "+v"(b[i]))
#define ALL4(M) M(0); M(1); M(2); M(3)
#define ALL4N(M, n) M(0, n); M(1, n); M(2, n); M(3, n)
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (GROUPED) {
                ALL4(A3); ALL4(XD); ALL4N(RD, 16); ALL4(AC); ALL4(XB); ALL4N(RB, 12);
                ALL4(A3); ALL4(XD); ALL4N(RD, 8); ALL4(AC); ALL4(XB); ALL4N(RB, 7);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    A3(i); XD(i); RD(i, 16); AC(i); XB(i); RB(i, 12); A3(i); XD(i); RD(i, 8); AC(i); XB(i); RB(i, 7);
                }
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 4; i++) s ^= a[i] ^ b[i] ^ c[i] ^ d[i] ^ z;
    if (s == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(uint32_t *, int);
struct Case { const char *name; kern_t k; double insts_per_iter; };

int main() {
    uint32_t *out; CK(hipMalloc(&out, (size_t)1 << 26));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const Case cases[] = {{"xor/alignbit, runs of 1", k_mix<1>, 128}, {"xor/alignbit, runs of 2", k_mix<2>, 128},
                          {"xor/alignbit, runs of 4", k_mix<4>, 128}, {"xor/alignbit, runs of 8", k_mix<8>, 128},
                          {"xor/alignbit, runs of 64", k_mix<64>, 128},
                          {"BLAKE3 G x4, one G after the other (dependent chain of 12)", k_g<false>, 8 * 48},
                          {"BLAKE3 G x4, kinds grouped four at a time", k_g<true>, 8 * 48}};
    printf("| stream (half VOP2 double-rate, half VOP3 single-rate) | waves/SIMD | G wave-inst/s | cycles per inst per SIMD @2.4 GHz |\n|---|---|---|---|\n");
    const int iters = 2000;
    for (const Case &c : cases)
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;
            c.k<<<blocks, 256>>>(out, 10);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(a));
                c.k<<<blocks, 256>>>(out, iters);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            const double insts = (double)blocks * 4 * iters * c.insts_per_iter;
            const double gps = insts / (best * 1e-3) / 1e9;
            printf("| %s | %d | %.1f | %.2f |\n", c.name, wps, gps, (cus * 4 * 2.4e9) / (gps * 1e9));
        }
    return 0;
}
