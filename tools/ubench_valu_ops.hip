// Issue rate of the int32 VALU opcodes a BLAKE3 compression is made of (and a few candidates to replace them), per
// opcode, at 1 / 2 / 8 waves per SIMD: G wave-instructions per second chip-wide and cycles per instruction per SIMD.
// Each wave runs CHAINS independent dependency chains of the opcode (8: latency hidden inside one wave as well).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu_ops.hip -o tools/ubench_valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int CHAINS = 8, UNROLL = 16;

#define OP_KERNEL(NAME, ASM)                                                                              \
    __global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, int iters) {                             \
        uint32_t x[CHAINS], y = threadIdx.x * 2654435761u + 1u, z = blockIdx.x ^ 0x9E3779B9u;              \
        for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x + c * 77u;                                      \
        for (int it = 0; it < iters; it++) {                                                                \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                            \
                _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(x[c]) : "v"(y), "v"(z)); \
            }                                                                                               \
        }                                                                                                   \
        uint32_t s = 0;                                                                                     \
        for (int c = 0; c < CHAINS; c++) s ^= x[c];                                                         \
        if (s == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = s;                               \
    }

OP_KERNEL(xor, "v_xor_b32 %0, %0, %1")
OP_KERNEL(add, "v_add_u32 %0, %0, %1")
OP_KERNEL(add3, "v_add3_u32 %0, %0, %1, %2")
OP_KERNEL(alignbit, "v_alignbit_b32 %0, %0, %0, 7")
OP_KERNEL(alignbit16, "v_alignbit_b32 %0, %0, %0, 16")
OP_KERNEL(perm, "v_perm_b32 %0, %0, %0, %1")
OP_KERNEL(xad, "v_xad_u32 %0, %0, %1, %2")
OP_KERNEL(or3, "v_or3_b32 %0, %0, %1, %2")
OP_KERNEL(bfi, "v_bfi_b32 %0, %0, %1, %2")
OP_KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
OP_KERNEL(lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
OP_KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
OP_KERNEL(mov, "v_mov_b32 %0, %1")
OP_KERNEL(fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP_KERNEL(add_f32, "v_add_f32 %0, %0, %1")
OP_KERNEL(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
OP_KERNEL(mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OP_KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %1")
OP_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP_KERNEL(bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
OP_KERNEL(mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP_KERNEL(xor_dpp, "v_xor_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

typedef void (*kern_t)(uint32_t *, int);
struct Op { const char *name; kern_t k; };

int main() {
    uint32_t *out; CK(hipMalloc(&out, (size_t)1 << 26));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const Op ops[] = {{"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_add3_u32", k_add3}, {"v_alignbit_b32 (7)", k_alignbit},
                      {"v_alignbit_b32 (16)", k_alignbit16}, {"v_perm_b32", k_perm}, {"v_xad_u32", k_xad}, {"v_or3_b32", k_or3},
                      {"v_bfi_b32", k_bfi}, {"v_lshl_add_u32", k_lshl_add}, {"v_lshl_or_b32", k_lshl_or}, {"v_and_or_b32", k_and_or},
                      {"v_mov_b32", k_mov}, {"v_fma_f32", k_fma_f32}, {"v_add_f32", k_add_f32}, {"v_pk_add_u16", k_pk_add_u16},
                      {"v_mad_u32_u24", k_mad_u24}, {"v_mul_lo_u32", k_mul_lo}, {"v_cndmask_b32", k_cndmask}, {"v_bitop3_b32", k_bitop3},
                      {"v_mov_b32_dpp quad_perm", k_mov_dpp}, {"v_xor_b32_dpp quad_perm", k_xor_dpp}};
    printf("%d CUs; %d chains per wave, cycles priced at the clock a fixed 2.4 GHz would give (see G inst/s for the raw figure)\n", cus, CHAINS);
    printf("| opcode | waves/SIMD | G wave-inst/s | cycles per inst per SIMD @2.4 GHz |\n|---|---|---|---|\n");
    const int iters = 2000;
    for (const Op &op : ops) {
        for (int wps : {1, 2, 8}) {
            const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD; wps blocks per CU
            op.k<<<blocks, 256>>>(out, 10);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(a));
                op.k<<<blocks, 256>>>(out, iters);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            const double insts = (double)blocks * 4 * iters * UNROLL * CHAINS;
            const double gps = insts / (best * 1e-3) / 1e9;
            printf("| %s | %d | %.1f | %.2f |\n", op.name, wps, gps, (cus * 4 * 2.4e9) / (gps * 1e9));
        }
    }
    return 0;
}
