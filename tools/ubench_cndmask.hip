// What does a lane select cost on gfx950?  v_cndmask_b32 (VCC / SGPR-pair mask) against v_bfi_b32 with a per-lane
// all-ones / all-zeros mask register, 8 independent chains per wave.  (tools/ubench_valu_ops measured v_cndmask_b32
// with an uninitialised VCC at 22 cycles: is that the instruction or the experiment?)
// Result (round 3): 23 cycles per instruction for the VCC form in THIS loop at any number of waves, 4.2 for the SGPR-pair
// form and for v_bfi_b32 -- but replacing the 133 VCC selects per row and wave of the commit kernel's butterfly by
// v_bfi_b32 did not change the kernel's row time (76 us either way): the micro-benchmark's figure does not carry over to
// compiled code, where a v_cmp feeds every group of selects.  Kept as a record of the dead end.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_cndmask.hip -o tools/ubench_cndmask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int CHAINS = 8, UNROLL = 16;

#define KERNEL(NAME, SETUP, ASM)                                                                           \
    __global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, int iters) {                              \
        uint32_t x[CHAINS], y = threadIdx.x * 2654435761u + 1u, m = (threadIdx.x & 1) ? 0xFFFFFFFFu : 0u;    \
        unsigned long long sm = 0x5555555555555555ull;                                                       \
        for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x + c * 77u;                                       \
        SETUP;                                                                                               \
        for (int it = 0; it < iters; it++) {                                                                 \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                             \
                _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(x[c]) : "v"(y), "v"(m), "s"(sm)); \
            }                                                                                                \
        }                                                                                                    \
        uint32_t s = 0;                                                                                      \
        for (int c = 0; c < CHAINS; c++) s ^= x[c];                                                          \
        if (s == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
    }

KERNEL(cnd_vcc, asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(m) : "vcc"), "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(cnd_sgpr, (void)0, "v_cndmask_b32_e64 %0, %0, %1, %3")
KERNEL(bfi, (void)0, "v_bfi_b32 %0, %2, %0, %1")
KERNEL(cnd_vcc_alt, asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(m) : "vcc"), "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL(xor_ref, (void)0, "v_xor_b32 %0, %0, %1")

typedef void (*kern_t)(uint32_t *, int);
struct Op { const char *name; kern_t k; };
int main() {
    uint32_t *out; CK(hipMalloc(&out, (size_t)1 << 26));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const Op ops[] = {{"v_cndmask_b32 (VCC set by v_cmp)", k_cnd_vcc}, {"v_cndmask_b32, operands swapped", k_cnd_vcc_alt},
                      {"v_cndmask_b32_e64 (SGPR pair mask)", k_cnd_sgpr}, {"v_bfi_b32 (VGPR mask)", k_bfi}, {"v_xor_b32 (reference)", k_xor_ref}};
    printf("| select | waves/SIMD | G wave-inst/s | cycles per inst per SIMD @2.4 GHz |\n|---|---|---|---|\n");
    const int iters = 2000;
    for (const Op &op : ops)
        for (int wps : {1, 4, 8}) {
            const int blocks = cus * wps;
            op.k<<<blocks, 256>>>(out, 10);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(a)); op.k<<<blocks, 256>>>(out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            const double insts = (double)blocks * 4 * iters * UNROLL * CHAINS;
            const double gps = insts / (best * 1e-3) / 1e9;
            printf("| %s | %d | %.1f | %.2f |\n", op.name, wps, gps, (cus * 4 * 2.4e9) / (gps * 1e9));
        }
    return 0;
}
