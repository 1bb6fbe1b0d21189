// Open-side kernels: the two row combinations (proximity test over Z, evaluation
// phase over F_q) fused into one pass over the witness, and the column-opening
// gather that emits the reference's proof-stream wire format.
//
// Reference loops replaced:
//   combine_rows over Int<M>           src/zip/utils.rs:94-127 via src/zip/pcs/open_z.rs:103-112
//   map_to_field + combine_rows over F src/zip/pcs/open_z.rs:76-90, src/conversion.rs:86-100,
//                                      src/field.rs:536-568, src/field/config.rs:163-170
//   open_merkle_trees_for_column       src/zip/pcs/open_z.rs:124-143, src/zip/pcs/utils.rs:163-176,220-233,
//                                      src/zip/pcs_transcript.rs:115-135,198-211
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace zipk {

typedef __int128 i128;
typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------
// multi-limb helpers (little-endian uint64 limbs)
// ---------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ uint64_t add_n(uint64_t (&a)[N], const uint64_t (&b)[N]) {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const u128 t = (u128)a[i] + b[i] + carry;
        a[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    return carry;
}
template <int N>
__device__ __forceinline__ uint64_t sub_n(uint64_t (&a)[N], const uint64_t (&b)[N]) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const u128 t = (u128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}
template <int N>
__device__ __forceinline__ bool geq_n(const uint64_t (&a)[N], const uint64_t (&b)[N]) {
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
}

template <int FL>
struct FieldDev {
    uint64_t modulus[FL];
    uint64_t r2[FL];
    uint64_t inv;
};

// t (2 FL limbs) = a * b, row by row on 32-bit limbs: x * y + r + carry with 32-bit x, y, r, carry never overflows
// 64 bits, so every step is ONE v_mad_u64_u32 (which takes a 64-bit addend) plus the carry add.  The same loops on
// 64-bit limbs with 128-bit intermediates compile to the same number of multiplies, but chained through carry flags
// with a hazard s_nop behind most of them: 0.49 -> 0.40 ms for round 1 of the sumcheck at 2^24 (its three unreduced
// products per pair of table entries).
template <int FL>
__device__ __forceinline__ void mul_wide(const uint64_t (&a)[FL], const uint64_t (&b)[FL], uint64_t (&t)[2 * FL]) {
    constexpr int N = 2 * FL;
    uint32_t x[N], y[N], r[2 * N];
#pragma unroll
    for (int i = 0; i < FL; i++) {
        x[2 * i] = (uint32_t)a[i];
        x[2 * i + 1] = (uint32_t)(a[i] >> 32);
        y[2 * i] = (uint32_t)b[i];
        y[2 * i + 1] = (uint32_t)(b[i] >> 32);
    }
#pragma unroll
    for (int i = 0; i < 2 * N; i++) r[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const uint64_t p = (uint64_t)x[i] * y[j] + r[i + j] + carry;
            r[i + j] = (uint32_t)p;
            carry = (uint32_t)(p >> 32);
        }
        r[i + N] = carry;
    }
#pragma unroll
    for (int i = 0; i < 2 * FL; i++) t[i] = (uint64_t)r[2 * i] | ((uint64_t)r[2 * i + 1] << 32);
}

// Montgomery reduction of a 2*FL-limb value t < q*R: returns t * R^-1 mod q, canonical.
// Same arithmetic as src/field/biginteger.rs:532-560 + the conditional subtraction of
// src/field/config.rs:68-76 (carry branch for moduli without a spare bit).
template <int FL>
__device__ __forceinline__ void mont_redc(uint64_t (&t)[2 * FL], const FieldDev<FL> &f, uint64_t (&out)[FL]) {
    uint64_t carry2 = 0;
#pragma unroll
    for (int i = 0; i < FL; i++) {
        const uint64_t k = t[i] * f.inv;
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < FL; j++) {
            const u128 x = (u128)k * f.modulus[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
        const u128 y = (u128)t[i + FL] + carry + carry2;
        t[i + FL] = (uint64_t)y;
        carry2 = (uint64_t)(y >> 64);
    }
#pragma unroll
    for (int i = 0; i < FL; i++) out[i] = t[FL + i];
    if (carry2 || geq_n<FL>(out, f.modulus)) sub_n<FL>(out, f.modulus);
}

template <int FL>
__device__ __forceinline__ void mont_mul(const uint64_t (&a)[FL], const uint64_t (&b)[FL], const FieldDev<FL> &f,
                                         uint64_t (&out)[FL]) {
    // (the 64-bit-limb product here: with mul_wide's 32-bit rows inlined at every mont_mul the kernels that hold many
    // of them allocate more registers and lose a wave per SIMD -- sumcheck round 1 0.40 -> 0.79 ms, measured)
    uint64_t t[2 * FL];
#pragma unroll
    for (int i = 0; i < 2 * FL; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < FL; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < FL; j++) {
            const u128 x = (u128)a[i] * b[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
        t[i + FL] = carry;
    }
    mont_redc<FL>(t, f, out);
}

// y (FL+2 limbs, < q * 2^(64+16)) mod q, canonical:  REDC(y) * R^2 * R^-1.
template <int FL>
__device__ __forceinline__ void reduce_wide(const uint64_t (&y)[FL + 2], const FieldDev<FL> &f, uint64_t (&out)[FL]) {
    uint64_t t[2 * FL];
#pragma unroll
    for (int i = 0; i < 2 * FL; i++) t[i] = (i < FL + 2) ? y[i] : 0;
    uint64_t z[FL];
    mont_redc<FL>(t, f, z);
    mont_mul<FL>(z, f.r2, f, out);
}

// ---------------------------------------------------------------------------
// Fused row combinations.  One thread per witness column, blockIdx.y selects a
// chunk of rows; exact partial sums go to HBM and are folded by
// combine_finalize_kernel.
//   over Z :  u'[c]  = sum_r coeff[r] * w[r][c]        192-bit accumulator (|.| < 2^(126+16))
//   over Fq:  row[c] = sum_r q0_mont[r] * w[r][c] mod q  == the reference's
//             sum_r q0[r] (x) phi(w[r][c]) because phi(w) = w*R mod q and every
//             reference operation returns the canonical residue.
// Both sums run on UNSIGNED operands: with w' = w + 2^63 and c' = c + 2^63 (one xor each)
//   sum c w  = sum c' w' - 2^63 (sum_r w'[r][c] + sum_r c'[r]) + rows 2^126      (mod 2^192: the value fits)
//   sum q0 w = sum q0 w' - 2^63 sum_r q0[r]
// where the sums over r of c' and q0 do not depend on the column: one thread per chunk adds them up (part_k) and the
// fold applies them.  Every product step is then x * y + r + carry on 32-bit words = one v_mad_u64_u32 (+ the carry
// add), no sign handling, no masked add of q0 for negative w (round 2's form: 160 VALU instructions per element from
// 128-bit C arithmetic; this one: ~65).
// quirk_mod != 0 reproduces the reference's `%=` against a modulus read as a
// negative Int (see oracle/zip_oracle.c field_from_signed_words): |w| is first
// reduced modulo quirk_mod = 2^(64*FL) - q when that is < 2^64 (QUIRK: its own instance, the 64-bit division
// stays out of the common kernel).
// ---------------------------------------------------------------------------
struct CombineArgs {
    const int64_t *evals;   // [num_rows][row_len]
    const int64_t *coeffs;  // [num_rows]            (device)
    const uint64_t *q0;     // [num_rows][FL]        (device, Montgomery limbs)
    uint32_t num_rows, row_len, rows_per_chunk;
    uint32_t prio;  // s_setprio level: the kernel usually runs beside the (older, always ready) hashing waves of the commit
    uint64_t quirk_mod;
    uint64_t *part_int;  // [chunks][row_len][3]   sum_r c'_r w'_rc - 2^63 sum_r w'_rc   (mod 2^192)
    uint64_t *part_a;    // [chunks][row_len][FL+2] sum_r q0_r w'_rc
    uint64_t *part_k;    // [chunks][FL+3]          sum_r q0_r (FL + 1 limbs) | sum_r c'_r (2 limbs)
};

// Accumulators of the row combinations: a value is kept as sum_j (lo[j] + 2^64 top[j]) 2^(32 j), one 96-bit accumulator
// per 32-bit position.  A 32 x 32 -> 64-bit product goes to its position with ONE v_mad_u64_u32 (64-bit addend, carry
// out) and one add of the carry: no carry runs between positions inside the row loop, no zero-extended register pairs
// (the packed form -- x * y + r + carry per word -- cost 25 instructions per 32 x 64-bit product row, half of them moves).
template <int N>
struct WideAcc {
    uint64_t lo[N];
    uint32_t top[N];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < N; j++) { lo[j] = 0; top[j] = 0; }
    }
    // x * y0 to position J, x * y1 to position J + 1.  Written out: from C (lo += p; top += lo < p) hipcc makes five
    // instructions per product (multiply, 64-bit add, compare, select, add) where the hardware needs two -- the
    // multiply-add takes the accumulator as its 64-bit addend and hands the carry out.  gfx950 wants two wait states
    // between a VALU write of a carry register and the VALU read of it: the second product and one s_nop fill them.
    template <int J>
    __device__ __forceinline__ void mad2(uint32_t x, uint32_t y0, uint32_t y1) {
#ifndef ZIPK_COMBINE_C_ARITH
        uint64_t carry0;
        asm("v_mad_u64_u32 %0, %4, %5, %6, %0\n\t"
            "v_mad_u64_u32 %1, vcc, %5, %7, %1\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32_e64 %2, %4, 0, %2, %4\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc"
            : "+v"(lo[J]), "+v"(lo[J + 1]), "+v"(top[J]), "+v"(top[J + 1]), "=&s"(carry0)
            : "s"(x), "v"(y0), "v"(y1)
            : "vcc");
#else
        const uint64_t p0 = (uint64_t)x * y0, p1 = (uint64_t)x * y1;
        const uint64_t s0 = lo[J] + p0, s1 = lo[J + 1] + p1;
        top[J] += s0 < p0 ? 1u : 0u;
        top[J + 1] += s1 < p1 ? 1u : 0u;
        lo[J] = s0;
        lo[J + 1] = s1;
#endif
    }
    template <int J>
    __device__ __forceinline__ void add(uint64_t x) {
        const uint64_t s = lo[J] + x;
        top[J] += s < x ? 1u : 0u;
        lo[J] = s;
    }
    // the packed value, L 64-bit limbs (the caller knows it fits)
    template <int L>
    __device__ __forceinline__ void pack(uint64_t (&out)[L]) const {
        uint32_t w[2 * L + 3];
#pragma unroll
        for (int i = 0; i < 2 * L + 3; i++) w[i] = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const uint32_t part[3] = {(uint32_t)lo[j], (uint32_t)(lo[j] >> 32), top[j]};
            uint32_t carry = 0;
#pragma unroll
            for (int i = j; i < 2 * L; i++) {
                const uint64_t t = (uint64_t)w[i] + (i - j < 3 ? part[i - j] : 0u) + carry;
                w[i] = (uint32_t)t;
                carry = (uint32_t)(t >> 32);
            }
        }
#pragma unroll
        for (int i = 0; i < L; i++) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    }
};

// acc += x[0 .. NX) * (y0 + 2^32 y1), x in 32-bit words: products x[j] y0 at position j, x[j] y1 at position j + 1
template <int NX, int J = 0, int N>
__device__ __forceinline__ void mad_words(WideAcc<N> &acc, const uint32_t (&x)[NX], uint32_t y0, uint32_t y1) {
    if constexpr (J < NX) {
        acc.template mad2<J>(x[J], y0, y1);
        mad_words<NX, J + 1>(acc, x, y0, y1);
    }
}

#ifndef ZIPK_COMBINE_UNROLL
#define ZIPK_COMBINE_UNROLL 4
#endif
constexpr int kCombineUnroll = ZIPK_COMBINE_UNROLL;
template <int FL, bool DO_INT, bool DO_FIELD, bool QUIRK>
__global__ void __launch_bounds__(256) combine_rows_kernel(CombineArgs a) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t chunk = blockIdx.y;
    const uint32_t r0 = chunk * a.rows_per_chunk;
    const uint32_t r1 = min(r0 + a.rows_per_chunk, a.num_rows);
    // the wave that also adds up the chunk's column-independent sums (wave-uniform)
    const bool lead = __builtin_amdgcn_readfirstlane((uint32_t)(blockIdx.x == 0 && threadIdx.x < 64)) != 0;
    constexpr uint64_t kBias = 1ull << 63;

    WideAcc<3> P;        // sum c' w'        < rows 2^128
    WideAcc<1> W;        // sum w'           < rows 2^64
    WideAcc<2 * FL + 1> A;  // sum q0 w'     < rows 2^(64 FL + 64)
    P.clear();
    W.clear();
    A.clear();

    const bool live = col < a.row_len;
    const int64_t *p = a.evals + (size_t)r0 * a.row_len + (live ? col : 0);
    // (row r's coefficient and q0 entry are the same for every lane: read through the constant address space they
    // come by scalar loads, issued rows ahead, and are multiplied from SGPRs -- as vector loads each cost a round trip
    // that the one resident wave of a SIMD had nobody to hide behind; nothing writes them while this kernel runs)
    const auto *coeffs_k = (const __attribute__((address_space(4))) int64_t *)(uintptr_t)a.coeffs;
    const auto *q0_k = (const __attribute__((address_space(4))) uint64_t *)(uintptr_t)a.q0;
    auto one_row = [&](int64_t w, uint32_t r) {
        const uint64_t wb = (uint64_t)w ^ kBias;
        uint32_t w0 = (uint32_t)wb, w1 = (uint32_t)(wb >> 32);
        if (DO_INT) {
            const uint64_t cb = (uint64_t)coeffs_k[r] ^ kBias;  // wave-uniform: scalar load, SGPRs
            const uint32_t c[2] = {(uint32_t)cb, (uint32_t)(cb >> 32)};
            mad_words<2>(P, c, w0, w1);
            W.template add<0>(wb);
        }
        if (DO_FIELD) {
            if (QUIRK) {  // (the field half only: the combination over the integers sees the entry as it is)
                const uint64_t mag = (w < 0 ? (uint64_t)0 - (uint64_t)w : (uint64_t)w) % a.quirk_mod;
                const uint64_t qb = (uint64_t)((w < 0) ? -(int64_t)mag : (int64_t)mag) ^ kBias;
                w0 = (uint32_t)qb;
                w1 = (uint32_t)(qb >> 32);
            }
            const auto *q = q0_k + (size_t)r * FL;  // wave-uniform: scalar loads, SGPRs
            uint32_t qw[2 * FL];
#pragma unroll
            for (int i = 0; i < FL; i++) {
                qw[2 * i] = (uint32_t)q[i];
                qw[2 * i + 1] = (uint32_t)(q[i] >> 32);
            }
            mad_words<2 * FL>(A, qw, w0, w1);
        }
    };
    // (the witness entries of kCombineUnroll rows are requested before the first of them is used: beside the commit
    // kernel a SIMD holds ONE wave of this kernel, nobody else hides its loads)
    uint32_t r = r0;
    for (; r + kCombineUnroll <= r1; r += kCombineUnroll, p += (size_t)kCombineUnroll * a.row_len) {
        int64_t w[kCombineUnroll];
#pragma unroll
        for (int k = 0; k < kCombineUnroll; k++) w[k] = p[(size_t)k * a.row_len];
#pragma unroll
        for (int k = 0; k < kCombineUnroll; k++) one_row(w[k], r + k);
    }
    for (; r < r1; r++, p += a.row_len) one_row(*p, r);
    if (live) {
        const size_t slot = (size_t)chunk * a.row_len + col;
        if (DO_INT) {
            // P -= W << 63   (192-bit, wraps)
            uint64_t pl[3], ws[2], wl[3];
            P.template pack<3>(pl);
            W.template pack<2>(ws);
            wl[0] = ws[0] << 63;
            wl[1] = (ws[0] >> 1) | (ws[1] << 63);
            wl[2] = ws[1] >> 1;
            sub_n<3>(pl, wl);
#pragma unroll
            for (int i = 0; i < 3; i++) a.part_int[slot * 3 + i] = pl[i];
        }
        if (DO_FIELD) {
            uint64_t al[FL + 2];
            A.template pack<FL + 2>(al);
#pragma unroll
            for (int i = 0; i < FL + 2; i++) a.part_a[slot * (FL + 2) + i] = al[i];
        }
    }
    if (!lead) return;
    // the chunk's column-independent sums: the lanes share the rows, then meet through xor shuffles
    const uint32_t lane = threadIdx.x;
    uint64_t Q[FL + 1], C[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < FL + 1; i++) Q[i] = 0;
    for (uint32_t r = r0 + lane; r < r1; r += 64) {
        if (DO_FIELD) {
            uint64_t t[FL + 1];
#pragma unroll
            for (int i = 0; i < FL; i++) t[i] = a.q0[(size_t)r * FL + i];
            t[FL] = 0;
            add_n<FL + 1>(Q, t);
        }
        if (DO_INT) {
            const uint64_t t[2] = {(uint64_t)a.coeffs[r] ^ kBias, 0};
            add_n<2>(C, t);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint64_t t[FL + 1], u[2];
#pragma unroll
        for (int i = 0; i < FL + 1; i++) t[i] = (uint64_t)__shfl_xor((unsigned long long)Q[i], off, 64);
        u[0] = (uint64_t)__shfl_xor((unsigned long long)C[0], off, 64);
        u[1] = (uint64_t)__shfl_xor((unsigned long long)C[1], off, 64);
        add_n<FL + 1>(Q, t);
        add_n<2>(C, u);
    }
    if (lane == 0) {
        uint64_t *k = a.part_k + (size_t)chunk * (FL + 3);
#pragma unroll
        for (int i = 0; i < FL + 1; i++) k[i] = Q[i];
        k[FL + 1] = C[0];
        k[FL + 2] = C[1];
    }
}

struct FinalizeArgs {
    const uint64_t *part_int, *part_a, *part_k;
    uint32_t chunks, row_len, m_limbs;
    uint32_t num_rows;    // rows the partial sums cover (the bias terms of combine_rows_kernel scale with it)
    uint32_t prio;        // s_setprio level (see CombineArgs)
    uint64_t *uprime;     // [row_len][m_limbs] little-endian limbs, or null
    uint64_t *row_limbs;  // [row_len][FL] Montgomery little-endian limbs, or null
    uint8_t *row_be;      // [row_len][8*FL] big-endian bytes of the Montgomery value, or null
};

// 16 columns per workgroup, eight threads per column: each sums every eighth chunk (the fold is latency-bound --
// one thread per column walking all chunks left 16 waves on the whole chip waiting on one load after another),
// then the eight partial sums meet in LDS and the first thread of the column finishes.
// 16 columns = 9 KB of LDS: the kernel must fit into the 26 KB the persistent commit workgroups leave free on a
// CU, or it only starts when they end (32 columns = 30.7 KB waited 1.2 ms for exactly that).
constexpr uint32_t kFinalizeCols = 16, kFinalizeGroups = 8;

template <int FL, bool DO_INT, bool DO_FIELD>
// (capped at 88 VGPRs, so that a wave fits into the registers per lane the
// hinted commit kernel leaves free on every SIMD; at its natural 126 the kernel waited for the commit to end)
__global__ void __launch_bounds__(kFinalizeCols * kFinalizeGroups) __attribute__((amdgpu_num_vgpr(88)))
combine_finalize_kernel(FinalizeArgs a, FieldDev<FL> f) {
    __shared__ uint64_t sh_int[kFinalizeGroups][kFinalizeCols][3];
    __shared__ uint64_t sh_a[kFinalizeGroups][kFinalizeCols][FL + 2];
    __shared__ uint64_t sh_k[kFinalizeGroups][FL + 3];
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t lc = threadIdx.x % kFinalizeCols, g = threadIdx.x / kFinalizeCols;
    const uint32_t col = blockIdx.x * kFinalizeCols + lc;
    const bool valid = col < a.row_len;
    if (DO_INT) {
        uint64_t s[3] = {0, 0, 0};
        if (valid) {
#pragma unroll 4
            for (uint32_t c = g; c < a.chunks; c += kFinalizeGroups) {
                uint64_t t[3];
#pragma unroll
                for (int i = 0; i < 3; i++) t[i] = a.part_int[((size_t)c * a.row_len + col) * 3 + i];
                add_n<3>(s, t);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) sh_int[g][lc][i] = s[i];
    }
    if (DO_FIELD) {
        uint64_t A[FL + 2];
#pragma unroll
        for (int i = 0; i < FL + 2; i++) A[i] = 0;
        if (valid) {
#pragma unroll 4
            for (uint32_t c = g; c < a.chunks; c += kFinalizeGroups) {
                const size_t slot = (size_t)c * a.row_len + col;
                uint64_t t[FL + 2];
#pragma unroll
                for (int i = 0; i < FL + 2; i++) t[i] = a.part_a[slot * (FL + 2) + i];
                add_n<FL + 2>(A, t);
            }
        }
#pragma unroll
        for (int i = 0; i < FL + 2; i++) sh_a[g][lc][i] = A[i];
    }
    if (lc == 0) {  // the column-independent sums (part_k), every eighth chunk per group as well
        uint64_t kq[FL + 1], kc[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < FL + 1; i++) kq[i] = 0;
#pragma unroll 4
        for (uint32_t c = g; c < a.chunks; c += kFinalizeGroups) {
            uint64_t q[FL + 1], cs[2];
#pragma unroll
            for (int i = 0; i < FL + 1; i++) q[i] = a.part_k[(size_t)c * (FL + 3) + i];
            cs[0] = a.part_k[(size_t)c * (FL + 3) + FL + 1];
            cs[1] = a.part_k[(size_t)c * (FL + 3) + FL + 2];
            add_n<FL + 1>(kq, q);
            add_n<2>(kc, cs);
        }
#pragma unroll
        for (int i = 0; i < FL + 1; i++) sh_k[g][i] = kq[i];
        sh_k[g][FL + 1] = kc[0];
        sh_k[g][FL + 2] = kc[1];
    }
    __syncthreads();
    if (g != 0 || !valid) return;
    uint64_t K[FL + 3];  // sum_r q0_r (FL + 1 limbs) | sum_r c'_r (2 limbs), all chunks
    {
        uint64_t kq[FL + 1], kc[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < FL + 1; i++) kq[i] = 0;
        for (uint32_t k = 0; k < kFinalizeGroups; k++) {
            uint64_t q[FL + 1], cs[2] = {sh_k[k][FL + 1], sh_k[k][FL + 2]};
#pragma unroll
            for (int i = 0; i < FL + 1; i++) q[i] = sh_k[k][i];
            add_n<FL + 1>(kq, q);
            add_n<2>(kc, cs);
        }
#pragma unroll
        for (int i = 0; i < FL + 1; i++) K[i] = kq[i];
        K[FL + 1] = kc[0];
        K[FL + 2] = kc[1];
    }
    if (DO_INT) {
        uint64_t s[3] = {0, 0, 0};
        for (uint32_t k = 0; k < kFinalizeGroups; k++) {
            uint64_t t[3];
#pragma unroll
            for (int i = 0; i < 3; i++) t[i] = sh_int[k][lc][i];
            add_n<3>(s, t);
        }
        // - 2^63 sum_r c'_r + rows 2^126 (see combine_rows_kernel)
        uint64_t t[3] = {K[FL + 1] << 63, (K[FL + 1] >> 1) | (K[FL + 2] << 63), K[FL + 2] >> 1};
        sub_n<3>(s, t);
        uint64_t u[3] = {0, (uint64_t)(a.num_rows & 3u) << 62, (uint64_t)(a.num_rows >> 2)};
        add_n<3>(s, u);
        const uint64_t sign = (uint64_t)((int64_t)s[2] >> 63);
        if (a.uprime) {
            for (uint32_t i = 0; i < a.m_limbs; i++)  // write_integer: limbs little-endian, pcs_transcript.rs:115-123
                a.uprime[(size_t)col * a.m_limbs + i] = i < 3 ? s[i] : sign;
        }
    }
    if (DO_FIELD) {
        uint64_t A[FL + 2], Bs[FL + 2];
#pragma unroll
        for (int i = 0; i < FL + 2; i++) A[i] = 0;
        for (uint32_t k = 0; k < kFinalizeGroups; k++) {
            uint64_t t[FL + 2];
#pragma unroll
            for (int i = 0; i < FL + 2; i++) t[i] = sh_a[k][lc][i];
            add_n<FL + 2>(A, t);
        }
        // 2^63 sum_r q0_r
        Bs[0] = K[0] << 63;
#pragma unroll
        for (int i = 1; i < FL + 1; i++) Bs[i] = (K[i - 1] >> 1) | (K[i] << 63);
        Bs[FL + 1] = K[FL] >> 1;
        uint64_t ra[FL], rb[FL];
        reduce_wide<FL>(A, f, ra);
        reduce_wide<FL>(Bs, f, rb);
        if (!geq_n<FL>(ra, rb)) add_n<FL>(ra, f.modulus);  // wraps mod 2^(64 FL) when q has no spare bit; fine
        sub_n<FL>(ra, rb);
        if (a.row_limbs) {
#pragma unroll
            for (int i = 0; i < FL; i++) a.row_limbs[(size_t)col * FL + i] = ra[i];
        }
        if (a.row_be) {  // BigInt::to_bytes_be of the Montgomery value, pcs_transcript.rs:107-113
            uint64_t *dst = reinterpret_cast<uint64_t *>(a.row_be + (size_t)col * 8 * FL);
#pragma unroll
            for (int i = 0; i < FL; i++) dst[i] = __builtin_bswap64(ra[FL - 1 - i]);
        }
    }
}

// Sum of G partial results of a row-sharded open (one per GPU after the RCCL
// all-gather): exact 512-bit adds for u', modular adds for the evaluation row.
template <int FL>
__global__ void __launch_bounds__(256) sum_partials_kernel(const uint64_t *uparts, const uint64_t *fparts,
                                                           uint32_t G, uint32_t row_len, uint32_t m_limbs,
                                                           uint64_t *uprime, uint64_t *row_limbs, FieldDev<FL> f,
                                                           uint8_t *row_be = nullptr) {
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= row_len) return;
    if (uparts) {
        uint64_t carry = 0;
        for (uint32_t i = 0; i < m_limbs; i++) {
            u128 acc = carry;
            for (uint32_t g = 0; g < G; g++) acc += uparts[((size_t)g * row_len + col) * m_limbs + i];
            uprime[(size_t)col * m_limbs + i] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
    }
    if (fparts) {
        uint64_t acc[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) acc[i] = 0;
        for (uint32_t g = 0; g < G; g++) {
            uint64_t t[FL];
#pragma unroll
            for (int i = 0; i < FL; i++) t[i] = fparts[((size_t)g * row_len + col) * FL + i];
            const uint64_t c = add_n<FL>(acc, t);
            if (c || geq_n<FL>(acc, f.modulus)) sub_n<FL>(acc, f.modulus);
        }
        if (row_limbs) {
#pragma unroll
            for (int i = 0; i < FL; i++) row_limbs[(size_t)col * FL + i] = acc[i];
        }
        if (row_be) {  // BigInt::to_bytes_be of the Montgomery value, pcs_transcript.rs:107-113
            uint64_t *dst = reinterpret_cast<uint64_t *>(row_be + (size_t)col * 8 * FL);
#pragma unroll
            for (int i = 0; i < FL; i++) dst[i] = __builtin_bswap64(acc[FL - 1 - i]);
        }
    }
}

// ---------------------------------------------------------------------------
// Column openings in wire format.  For opened column i (index cols[i]):
//   num_rows x K limbs little-endian            rows[r*cw + col]            (open_z.rs:130-137)
//   num_rows x { be64(depth), depth x 32 B }    sibling (col >> k) ^ 1 of level k, leaf level first
//                                               (pcs/utils.rs:163-176, pcs_transcript.rs:198-211)
// Grid: (n_cols, row chunks).  Everything is moved as 8-byte words because a
// path record (8 + 32*depth bytes) is only 8-byte aligned in the stream.
// ---------------------------------------------------------------------------
struct OpenColsArgs {
    const uint64_t *rows;    // [num_rows][cw][K], or [num_rows][cw][2] when compact_rows (see CommitArgs)
    uint32_t compact_rows;
    const uint64_t *layers;  // [num_rows][2*cw][4]
    const uint32_t *cols;    // [n_cols] (device)
    const uint32_t *order;   // [n_cols] (device) or null: workgroup x -> opening index (see gather_order in zip_hip.hip)
    uint8_t *out;            // wire stream of the openings
    uint32_t num_rows, cw, depth, k_limbs;
    uint32_t row_lo, row_hi;  // rows handled by this launch (one pipeline chunk)
    uint32_t rows_per_block;  // even, or the launch has a single block row
    uint32_t prio;            // s_setprio level of the gather waves (tuning knob)
    // packed openings (CommitArgs.pk; open_columns_ilv_kernel): the column values and the siblings of levels 0..2 are
    // read from the packed block of the row's group of four at the ranks of pk_rank[opening] = {value, level 0, level 1,
    // level 2}, the siblings of levels >= 3 from the row-interleaved `layers` of such a commitment
    const uint8_t *pk;
    uint32_t pk_stride, pk_off0, pk_off1, pk_off2;
    const uint16_t *pk_rank;  // [n_cols][4] (device)
    // open_columns_ilv_kernel: everything workgroup x needs to know in ONE 16-byte entry (or null: order / cols / pk_rank)
    //   .x = opening | column << 16      .y = rank of the value | of the level-0 sibling << 16
    //   .z = rank of the level-1 sibling | of the level-2 sibling << 16
    const uint4 *wg_tab;
};

// Grid (n_cols, row blocks): blocks that run together share a narrow band of rows, so the
// upper tree levels of those rows are served from L2 / Infinity Cache while the leaf-level
// lines stream from HBM once.
// A path record (be64(depth) + depth sibling hashes) is only 8-byte aligned in the stream,
// so a block first assembles the byte-exact image of its records in LDS from aligned
// 16-byte loads (two lanes per 32-byte node -> one request), then streams the image out as
// aligned 16-byte stores, 1 KiB per wave instruction.
//
// The kernel usually runs BESIDE the VALU-bound persistent commit kernel, so every vector
// instruction it issues is taken from the hashing waves: each thread keeps one fixed
// (level, half) role, and walking the rows is one pointer increment per 16-byte load -- no
// divisions, no per-element address arithmetic.  SLOTS = lanes reserved per row
// (2*depth + 1 rounded up to 32 or 64).
template <int SLOTS>
__global__ void __launch_bounds__(256) open_columns_kernel(OpenColsArgs a) {
    extern __shared__ __align__(16) unsigned char img[];
    constexpr uint32_t K = 4;               // Int<4> column values (checked by zip_ctx_create)
    constexpr uint32_t RPP = 256 / SLOTS;   // rows per pass of the block
    if (a.prio) __builtin_amdgcn_s_setprio(2);  // memory-bound: do not queue behind older hashing waves
    const uint32_t ci = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const uint32_t col = a.cols[ci];
    const uint32_t d = a.depth, cw2 = 2u * a.cw;
    const uint32_t rec_bytes = 8 + 32 * d;
    const size_t col_bytes = (size_t)a.num_rows * (8 * K + rec_bytes);
    uint8_t *base = a.out + (size_t)ci * col_bytes;
    const uint32_t r0 = a.row_lo + blockIdx.y * a.rows_per_block;
    const uint32_t r1 = min(r0 + a.rows_per_block, a.row_hi);
    const uint32_t nrows = r1 - r0;

    // ---- phase 1: gather sibling hashes into the LDS image ----
    // lane role h < 2*depth: half (h & 1) of the level-(h >> 1) sibling (pcs/utils.rs:163-176);
    // h == 2*depth: the be64(depth) length prefix (pcs_transcript.rs:200-203).
    {
        const uint32_t h = threadIdx.x & (SLOTS - 1), rsub = threadIdx.x / SLOTS;
        const uint32_t lvl = h >> 1;
        const uint32_t node = cw2 - (cw2 >> lvl) + ((col >> lvl) ^ 1u);
        const uint64_t *src = a.layers + ((size_t)(r0 + rsub) * cw2 + node) * 4 + (h & 1) * 2;
        const size_t src_step = (size_t)RPP * cw2 * 4;
        unsigned char *dst = img + (size_t)rsub * rec_bytes + 8 + (size_t)h * 16;
        const uint32_t dst_step = RPP * rec_bytes;
        if (h < 2 * d) {
#pragma unroll 4
            for (uint32_t rr = rsub; rr < nrows; rr += RPP, src += src_step, dst += dst_step) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src);
                reinterpret_cast<uint64_t *>(dst)[0] = v.x;
                reinterpret_cast<uint64_t *>(dst)[1] = v.y;
            }
        } else if (h == 2 * d) {
            const uint64_t hdr = __builtin_bswap64((uint64_t)d);
            for (uint32_t rr = rsub; rr < nrows; rr += RPP)
                *reinterpret_cast<uint64_t *>(img + (size_t)rr * rec_bytes) = hdr;
        }
    }
    // ---- column values: rows[r*cw + col], K limbs little-endian (open_z.rs:130-137) ----
    {
        const uint32_t half = threadIdx.x & 1, rsub = threadIdx.x >> 1;  // two 16-byte halves per value
        if (rsub < nrows) {
            const uint32_t r = r0 + rsub;
            ulonglong2 v;
            if (a.compact_rows) {  // (w0, w1, w2, sign): the upper half of the Int<4> is the sign word four times
                const uint4 e = *reinterpret_cast<const uint4 *>(a.rows + ((size_t)r * a.cw + col) * 2);
                const uint64_t ss = ((uint64_t)e.w << 32) | e.w;
                v.x = half ? ss : ((uint64_t)e.y << 32) | e.x;
                v.y = half ? ss : ((uint64_t)e.w << 32) | e.z;
            } else {
                v = *reinterpret_cast<const ulonglong2 *>(a.rows + ((size_t)r * a.cw + col) * K + half * 2);
            }
            *reinterpret_cast<ulonglong2 *>(base + (size_t)r * 8 * K + half * 16) = v;
        }
    }
    __syncthreads();
    // ---- phase 2: stream the image out ----
    unsigned char *recs = base + (size_t)a.num_rows * 8 * K + (size_t)r0 * rec_bytes;
    const uint32_t total = nrows * rec_bytes;
    if ((reinterpret_cast<uintptr_t>(recs) & 15) == 0) {
        const uint32_t n16 = total / 16;
        for (uint32_t i = threadIdx.x; i < n16; i += 256)
            reinterpret_cast<uint4 *>(recs)[i] = reinterpret_cast<const uint4 *>(img)[i];
        if (threadIdx.x == 0 && (total & 15))
            reinterpret_cast<uint64_t *>(recs)[n16 * 2] = reinterpret_cast<const uint64_t *>(img)[n16 * 2];
    } else {
        for (uint32_t i = threadIdx.x; i < total / 8; i += 256)
            reinterpret_cast<uint64_t *>(recs)[i] = reinterpret_cast<const uint64_t *>(img)[i];
    }
}

// The same openings without the LDS image: every lane moves 16 bytes per row straight from where they
// are to where the wire format wants them.  The siblings of a record sit at 8 (mod 16) in every other record, so the
// stores are 16-byte stores to 8-byte-aligned addresses (global_store_dwordx4 only wants dword alignment); the lanes
// of a row still write one contiguous run.  Roles of the SLOTS lanes reserved per row:
//   h < 2 depth        half (h & 1) of the level-(h >> 1) sibling
//   h == 2 depth       the record's first 16 bytes: be64(depth) and, again, the first 8 bytes of the level-0 sibling
//                      (the same bytes lane 0 writes -- so that this lane, too, is "load 16, store 16")
//   h == 2 depth + 1,2 the two 16-byte halves of the column value (open_z.rs:130-137)
// One instruction stream for all of them (the role is a select on the loaded data); no LDS, no barrier, 21 VGPRs:
// nothing but registers bounds how many of its waves fit beside the commit kernel.  Used where the commit kernel
// leaves too little LDS for a useful image (cw = 16384); beside the 8-entry commit kernel its waves, which never wait
// at a barrier, take more issue slots from the hashing waves than they save (EXPERIMENTS.md).
struct __attribute__((packed, aligned(8))) oc_u128_a8 { uint64_t x, y; };

template <int SLOTS>
__global__ void __launch_bounds__(256) open_columns_stream_kernel(OpenColsArgs a) {
    constexpr uint32_t K = 4;               // Int<4> column values (checked by zip_ctx_create)
    constexpr uint32_t RPP = 256 / SLOTS;   // rows per pass of the block
    if (a.prio) __builtin_amdgcn_s_setprio(2);
    const uint32_t ci = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const uint32_t col = a.cols[ci];
    const uint32_t d = a.depth, cw2 = 2u * a.cw;
    const uint32_t rec_bytes = 8 + 32 * d;
    const size_t col_bytes = (size_t)a.num_rows * (8 * K + rec_bytes);
    uint8_t *base = a.out + (size_t)ci * col_bytes;
    const uint32_t r0 = a.row_lo + blockIdx.y * a.rows_per_block;
    const uint32_t r1 = min(r0 + a.rows_per_block, a.row_hi);
    const uint32_t nrows = r1 - r0;
    const uint32_t h = threadIdx.x & (SLOTS - 1), rsub = threadIdx.x / SLOTS;
    if (h > 2 * d + 2) return;
    const bool is_hdr = h == 2 * d, is_val = h > 2 * d;
    const uint32_t half = is_val ? h - 2 * d - 1 : (h & 1u);
    const uint32_t lvl = (is_hdr || is_val) ? 0u : h >> 1;
    const uint32_t row = r0 + rsub;
    // source: a tree node (or its packed copy), or the column's row entry
    const uint8_t *src;
    size_t src_step;
    if (is_val) {
        if (a.compact_rows) {
            src = reinterpret_cast<const uint8_t *>(a.rows + ((size_t)row * a.cw + col) * 2);
            src_step = (size_t)RPP * a.cw * 16;
        } else {
            src = reinterpret_cast<const uint8_t *>(a.rows + ((size_t)row * a.cw + col) * K + half * 2);
            src_step = (size_t)RPP * a.cw * 8 * K;
        }
    } else {
        const uint32_t node = cw2 - (cw2 >> lvl) + ((col >> lvl) ^ 1u);
        src = reinterpret_cast<const uint8_t *>(a.layers + ((size_t)row * cw2 + node) * 4) + (is_hdr ? 0u : half * 16u);
        src_step = (size_t)RPP * cw2 * 32;
    }
    uint8_t *dst = is_val ? base + (size_t)row * 8 * K + half * 16
                          : base + (size_t)a.num_rows * 8 * K + (size_t)row * rec_bytes + (is_hdr ? 0u : 8u + h * 16u);
    const size_t dst_step = is_val ? (size_t)RPP * 8 * K : (size_t)RPP * rec_bytes;
    const bool val_hi = is_val && half && a.compact_rows;  // the upper half of a 16-byte row entry's Int<4>: sign words
    const uint64_t hdr = __builtin_bswap64((uint64_t)d);
#pragma unroll 4
    for (uint32_t rr = rsub; rr < nrows; rr += RPP, src += src_step, dst += dst_step) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src);
        const uint64_t sw = (v.y >> 32) * 0x100000001ull;  // (the entry's sign word, twice)
        oc_u128_a8 o;
        o.x = is_hdr ? hdr : val_hi ? sw : v.x;
        o.y = is_hdr ? v.x : val_hi ? sw : v.y;
        *reinterpret_cast<oc_u128_a8 *>(dst) = o;
    }
}

// The openings with as few MEMORY INSTRUCTIONS as they can be made with (round 3, second attempt).  What a gather costs
// the VALU-bound commit kernel beside it is its vector-memory and LDS instructions (~35 SIMD cycles of hashing each,
// EXPERIMENTS.md), not its VALU ones: the first "lean" kernel (fewest VALU instructions, but split loads, an 8-byte
// header store and four dword stores for a sign word: 32 memory instructions per wave and 32 rows) slowed the commit
// kernel MORE than open_columns_kernel with its LDS image (16).  Here every lane that has a role issues exactly ONE
// 16-byte load and ONE 16-byte store per row, nothing else: 8 memory instructions per wave and 32 rows.
//   h < 2 depth      half (h & 1) of the level-(h >> 1) sibling: 16 bytes from the tree / the packed block to the record
//   h == 2 depth     the record's first 16 bytes: be64(depth), then the first 8 bytes of the level-0 sibling -- it loads
//                    what lane 0 loads (one request in the TA) and stores [header | low half]; lane 0 writes the same 8
//                    bytes again
//   h == 2 depth + 1 the value's low 16 bytes      h == 2 depth + 2: its high 16 bytes (a compact entry's sign word
//                    four times: three register moves)
// No LDS, no barrier; 8-byte-aligned records: global_store_dwordx4 at dword alignment; the walk over the rows is one
// 64-bit add each for the source and the destination pointer; four rows in flight per lane.
template <int SLOTS>
__global__ void __launch_bounds__(256) open_columns_lean_kernel(OpenColsArgs a) {
    constexpr uint32_t K = 4;               // Int<4> column values (checked by zip_ctx_create)
    constexpr uint32_t RPP = 256 / SLOTS;   // rows per pass of the block
    if (a.prio) __builtin_amdgcn_s_setprio(2);
    const uint32_t ci = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const uint32_t col = a.cols[ci];
    const uint32_t d = a.depth, cw2 = 2u * a.cw;
    const uint32_t rec_bytes = 8 + 32 * d;
    const size_t col_bytes = (size_t)a.num_rows * (8 * K + rec_bytes);
    uint8_t *base = a.out + (size_t)ci * col_bytes;
    const uint32_t r0 = a.row_lo + blockIdx.y * a.rows_per_block;
    const uint32_t r1 = min(r0 + a.rows_per_block, a.row_hi);
    const uint32_t nrows = r1 - r0;
    const uint32_t h = threadIdx.x & (SLOTS - 1), rsub = threadIdx.x / SLOTS;
    if (h > 2 * d + 2) return;
    const bool is_hdr = h == 2 * d, is_val = h > 2 * d;
    const uint32_t half = is_val ? h - 2 * d - 1 : is_hdr ? 0u : (h & 1u);
    const uint32_t lvl = (is_hdr || is_val) ? 0u : h >> 1;
    const uint32_t row = r0 + rsub;
    const bool val_hi_compact = is_val && half && a.compact_rows;
    // source: a tree node (or its packed copy), or the column's row entry
    const uint8_t *src;
    size_t src_step;
    if (is_val) {
        if (a.compact_rows) {
            src = reinterpret_cast<const uint8_t *>(a.rows + ((size_t)row * a.cw + col) * 2);
            src_step = (size_t)RPP * a.cw * 16;
        } else {
            src = reinterpret_cast<const uint8_t *>(a.rows + ((size_t)row * a.cw + col) * K + half * 2);
            src_step = (size_t)RPP * a.cw * 8 * K;
        }
    } else {
        const uint32_t node = cw2 - (cw2 >> lvl) + ((col >> lvl) ^ 1u);
        src = reinterpret_cast<const uint8_t *>(a.layers + ((size_t)row * cw2 + node) * 4) + half * 16u;
        src_step = (size_t)RPP * cw2 * 32;
    }
    uint8_t *dst = is_val ? base + (size_t)row * 8 * K + half * 16
                          : base + (size_t)a.num_rows * 8 * K + (size_t)row * rec_bytes + (is_hdr ? 0u : 8u + h * 16u);
    const size_t dst_step = is_val ? (size_t)RPP * 8 * K : (size_t)RPP * rec_bytes;
    const uint64_t hdr = __builtin_bswap64((uint64_t)d);
    const uint32_t hdr_lo = (uint32_t)hdr, hdr_hi = (uint32_t)(hdr >> 32);
    const uint32_t full = nrows / RPP, rest = nrows % RPP;  // wave-uniform: the loops below are scalar loops
    auto passes = [&](auto n_tag) {
        constexpr int N = decltype(n_tag)::value;
        uint4 v[N];
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = *reinterpret_cast<const uint4 *>(src + k * src_step);
#pragma unroll
        for (int k = 0; k < N; k++) {
            uint4 o = v[k];
            if (is_hdr) o = make_uint4(hdr_lo, hdr_hi, v[k].x, v[k].y);
            if (val_hi_compact) o = make_uint4(v[k].w, v[k].w, v[k].w, v[k].w);
            oc_u128_a8 w;
            w.x = ((uint64_t)o.y << 32) | o.x;
            w.y = ((uint64_t)o.w << 32) | o.z;
            *reinterpret_cast<oc_u128_a8 *>(dst + k * dst_step) = w;
        }
        src += N * src_step;
        dst += N * dst_step;
    };
    uint32_t p = 0;
    for (; p + 4 <= full; p += 4) passes(std::integral_constant<int, 4>{});
    for (; p < full; p++) passes(std::integral_constant<int, 1>{});
    if (rsub < rest) passes(std::integral_constant<int, 1>{});
}


// ---------------------------------------------------------------------------------------------------------------
// The openings of a PACKED commitment (zip_commit_hinted / zip_commit_open, round 4): everything an opening reads is
// row-interleaved in groups of four rows (CommitArgs.pk), so the sibling of level k of rows 4q .. 4q+3 is ONE whole
// 128-byte line -- eight lanes x 16 bytes -- where the natural layout has four 32-byte pieces of four lines 512 KB
// apart (55.7 M read requests of 33 useful bytes per launch at 2^24, profiles/round3_gather_pmc.md).
// Workgroup = (opening, block of rows), 256 threads.  Per pass of 32 rows (8 groups):
//   lane l of a wave:  group l / 8 of the pass, row (l % 8) / 2 of the group, half l % 2 of the 32-byte node
//   wave w:            levels w, w + 4, w + 8, w + 12 (< depth): one 1 KB wave instruction = 8 whole lines per level
//   wave 3 also:       the column values (64 lanes = 32 rows x 2 halves of the Int<4>, from the 16-byte entries)
//   wave 2 also:       the be64(depth) prefix of every record (pcs_transcript.rs:200-203)
// IMAGE: the byte-exact image of the block's records is assembled in LDS and streamed out as aligned 16-byte stores
// (a record is only 8-byte aligned in the stream); !IMAGE: every lane stores its 16 bytes where they belong
// (global_store_dwordx4 at dword alignment; no LDS, no barrier -- for the geometries whose commit kernel leaves no room
// for an image beside it).
// Reference: open_merkle_trees_for_column, src/zip/pcs/open_z.rs:124-143; MerkleProof::create_proof,
// src/zip/pcs/utils.rs:163-176; write_merkle_proof, src/zip/pcs_transcript.rs:198-211.
typedef uint32_t oc_u32x4 __attribute__((ext_vector_type(4)));

// WHOLE = 32 * P: a workgroup whose block is P whole passes of 32 rows (the case at every size the gather matters at) takes
// a branch-free path -- per-lane pointers set up once, the loads of ALL its levels and passes issued back to back before
// the first one is used.  (Round 4's first form predicated every load on `row in block && level < depth`: the compiler
// then cannot tell that the load of the previous level has been consumed and puts `s_waitcnt vmcnt(0)` in front of the next
// one's address arithmetic -- four dependent round trips per wave where one was meant.)  Any other block -- ragged
// ends, single rows -- takes the general loop below it.
typedef const oc_u32x4 __attribute__((address_space(1))) *oc_gptr;

template <bool IMAGE, int P>
__global__ void __launch_bounds__(256) open_columns_ilv_kernel(OpenColsArgs a) {
    extern __shared__ __align__(16) unsigned char img[];
    constexpr uint32_t K = 4;  // Int<4> column values (checked by zip_ctx_create)
    if (a.prio) __builtin_amdgcn_s_setprio(2);  // memory-bound: do not queue behind older hashing waves
    uint32_t ci, col, rk[4];
    if (a.wg_tab) {  // (one scalar 16-byte load instead of a chain of two dependent table reads)
        const uint4 t = a.wg_tab[blockIdx.x];
        ci = t.x & 0xFFFFu;
        col = t.x >> 16;
        rk[0] = t.y & 0xFFFFu;
        rk[1] = t.y >> 16;
        rk[2] = t.z & 0xFFFFu;
        rk[3] = t.z >> 16;
    } else {
        ci = a.order ? a.order[blockIdx.x] : blockIdx.x;
        col = a.cols[ci];
#pragma unroll
        for (int k = 0; k < 4; k++) rk[k] = a.pk_rank[ci * 4 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) rk[k] = __builtin_amdgcn_readfirstlane(rk[k]);
    const uint32_t d = a.depth, cw2 = 2u * a.cw;
    const uint32_t rec_bytes = 8 + 32 * d;
    const size_t col_bytes = (size_t)a.num_rows * (8 * K + rec_bytes);
    uint8_t *base = a.out + (size_t)ci * col_bytes;
    uint8_t *recs = base + (size_t)a.num_rows * 8 * K;  // record of row r at recs + r * rec_bytes
    const uint32_t r0 = a.row_lo + blockIdx.y * a.rows_per_block;
    const uint32_t r1 = min(r0 + a.rows_per_block, a.row_hi);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t sub = lane & 7u, gi = lane >> 3, half = lane & 1u;
    // wave-uniform per level j (k = wave + 4 j): byte offset of the level's line inside a group, and the group stride
    const uint8_t *src_base[4];
    size_t src_gstride[4];
    uint32_t src_off[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t k = wave + 4u * j;
        if (k < 3u) {
            const uint32_t off = k == 0 ? a.pk_off0 : k == 1 ? a.pk_off1 : a.pk_off2;
            const uint32_t rank = k == 0 ? rk[1] : k == 1 ? rk[2] : rk[3];
            src_base[j] = a.pk;
            src_gstride[j] = (size_t)4 * a.pk_stride;
            src_off[j] = 4u * off + rank * 128u;
        } else {
            const uint32_t node = cw2 - (cw2 >> k) + ((col >> k) ^ 1u);
            src_base[j] = reinterpret_cast<const uint8_t *>(a.layers);
            src_gstride[j] = (size_t)cw2 * 128;
            src_off[j] = 0;
            // (cw2 * 128 bytes per group: the node's offset can exceed 32 bits at cw = 16384 only by a factor the
            // size_t product below absorbs)
            src_base[j] += (size_t)node * 128;
        }
    }
    const uint32_t vrank = rk[0];
    const uint64_t hdr = __builtin_bswap64((uint64_t)d);
    // (uniform over the workgroup; depth >= 12: every wave has its first three levels)
    const bool whole = P > 0 && (r0 & 3u) == 0 && r1 - r0 == 32u * (uint32_t)P && d >= 12u;
    if (whole) {
        constexpr int PP = P > 0 ? P : 1;
        const size_t g0 = (size_t)(r0 >> 2) + gi;  // this lane's group of four rows in pass 0
        const uint8_t *ptr[4];
#pragma unroll
        for (int j = 0; j < 4; j++) ptr[j] = src_base[j] + g0 * src_gstride[j] + src_off[j] + sub * 16u;
        const bool has3 = wave + 12u < d;
        oc_u32x4 v[PP][4];
#pragma unroll
        for (int p = 0; p < PP; p++)
#pragma unroll
            for (int j = 0; j < 3; j++) v[p][j] = *(oc_gptr)(ptr[j] + (size_t)p * 8 * src_gstride[j]);
        if (has3) {
#pragma unroll
            for (int p = 0; p < PP; p++) v[p][3] = *(oc_gptr)(ptr[3] + (size_t)p * 8 * src_gstride[3]);
        }
        if (wave == 3u) {
            // column values (open_z.rs:130-137): lane -> (row lane / 2 of the pass, half of the Int<4>)
            oc_u32x4 e[PP];
#pragma unroll
            for (int p = 0; p < PP; p++) {
                const uint32_t vrow = r0 + 32u * p + (lane >> 1);
                e[p] = *(oc_gptr)(a.pk + (size_t)(vrow >> 2) * 4 * a.pk_stride + ((size_t)vrank * 4 + (vrow & 3u)) * 16);
            }
#pragma unroll
            for (int p = 0; p < PP; p++) {
                const uint32_t vrow = r0 + 32u * p + (lane >> 1);
                const uint4 o = half ? make_uint4(e[p].w, e[p].w, e[p].w, e[p].w) : make_uint4(e[p].x, e[p].y, e[p].z, e[p].w);
                *reinterpret_cast<uint4 *>(base + (size_t)vrow * 8 * K + half * 16) = o;
            }
        }
        if (wave == 2u && lane < 32u) {
#pragma unroll
            for (int p = 0; p < PP; p++) {
                if (IMAGE) *reinterpret_cast<uint64_t *>(img + (size_t)(32u * p + lane) * rec_bytes) = hdr;
                else *reinterpret_cast<uint64_t *>(recs + (size_t)(r0 + 32u * p + lane) * rec_bytes) = hdr;
            }
        }
        const uint32_t lrow = 4u * gi + (sub >> 1);  // row of the pass
        auto put = [&](int p, int j) {
            const uint32_t k = wave + 4u * j, rr = 32u * p + lrow;
            const uint64_t lo = ((uint64_t)v[p][j].y << 32) | v[p][j].x, hi = ((uint64_t)v[p][j].w << 32) | v[p][j].z;
            if (IMAGE) {
                uint64_t *dst = reinterpret_cast<uint64_t *>(img + rr * rec_bytes + 8 + k * 32u + half * 16u);
                dst[0] = lo;
                dst[1] = hi;
            } else {
                oc_u128_a8 o;
                o.x = lo;
                o.y = hi;
                *reinterpret_cast<oc_u128_a8 *>(recs + (size_t)(r0 + rr) * rec_bytes + 8 + k * 32u + half * 16u) = o;
            }
        };
#pragma unroll
        for (int p = 0; p < PP; p++)
#pragma unroll
            for (int j = 0; j < 3; j++) put(p, j);
        if (has3) {
#pragma unroll
            for (int p = 0; p < PP; p++) put(p, 3);
        }
    } else
    for (uint32_t rb = r0 & ~3u; rb < r1; rb += 32u) {
        const uint32_t row = rb + 4u * gi + (sub >> 1);
        const bool ok = row >= r0 && row < r1;
        const size_t grp = row >> 2;
        oc_u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (wave + 4u * j < d && ok)
                v[j] = *reinterpret_cast<const oc_u32x4 *>(src_base[j] + grp * src_gstride[j] + src_off[j] + sub * 16u);
        }
        // ---- column values: rows[r * cw + col], K limbs little-endian (open_z.rs:130-137), from the 16-byte entry
        //      (w0, w1, w2, sign): the upper half of the Int<4> is the sign word four times
        if (wave == 3u) {
            const uint32_t vrow = rb + (lane >> 1);
            if (vrow >= r0 && vrow < r1) {
                const oc_u32x4 e = *reinterpret_cast<const oc_u32x4 *>(a.pk + (size_t)(vrow >> 2) * 4 * a.pk_stride +
                                                                       ((size_t)vrank * 4 + (vrow & 3u)) * 16);
                const uint4 o = half ? make_uint4(e.w, e.w, e.w, e.w) : make_uint4(e.x, e.y, e.z, e.w);
                *reinterpret_cast<uint4 *>(base + (size_t)vrow * 8 * K + half * 16) = o;
            }
        }
        if (wave == 2u && lane < 32u) {
            const uint32_t hrow = rb + lane;
            if (hrow >= r0 && hrow < r1) {
                if (IMAGE) *reinterpret_cast<uint64_t *>(img + (size_t)(hrow - r0) * rec_bytes) = hdr;
                else *reinterpret_cast<uint64_t *>(recs + (size_t)hrow * rec_bytes) = hdr;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t k = wave + 4u * j;
            if (k < d && ok) {
                const uint64_t lo = ((uint64_t)v[j].y << 32) | v[j].x, hi = ((uint64_t)v[j].w << 32) | v[j].z;
                if (IMAGE) {
                    uint64_t *dst = reinterpret_cast<uint64_t *>(img + (size_t)(row - r0) * rec_bytes + 8 + k * 32u + half * 16u);
                    dst[0] = lo;
                    dst[1] = hi;
                } else {
                    oc_u128_a8 o;
                    o.x = lo;
                    o.y = hi;
                    *reinterpret_cast<oc_u128_a8 *>(recs + (size_t)row * rec_bytes + 8 + k * 32u + half * 16u) = o;
                }
            }
        }
    }
    if (!IMAGE) return;
    __syncthreads();
    // ---- stream the image out ----
    unsigned char *out = recs + (size_t)r0 * rec_bytes;
    const uint32_t total = (r1 - r0) * rec_bytes;
    if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const uint32_t n16 = total / 16;
        for (uint32_t i = threadIdx.x; i < n16; i += 256)
            reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(img)[i];
        if (threadIdx.x == 0 && (total & 15))
            reinterpret_cast<uint64_t *>(out)[n16 * 2] = reinterpret_cast<const uint64_t *>(img)[n16 * 2];
    } else {
        for (uint32_t i = threadIdx.x; i < total / 8; i += 256)
            reinterpret_cast<uint64_t *>(out)[i] = reinterpret_cast<const uint64_t *>(img)[i];
    }
}

}  // namespace zipk
