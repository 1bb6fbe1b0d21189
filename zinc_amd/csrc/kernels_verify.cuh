// Verifier-side kernels and the witness MLE evaluation (item 2).
//
// Reference loops replaced:
//   verify_testing / verify_column_testing   src/zip/pcs/verify_z.rs:60-127
//   verify_evaluation_z / verify_proximity_q_0 src/zip/pcs/verify_z.rs:129-188
//   ColumnOpening::verify_column, MerkleProof::verify  src/zip/pcs/utils.rs:178-210,235-249
//   encode_wide / encode_f (one row)           src/zip/code_raa.rs:107-138
//   FieldMap for Int<K>                        src/conversion.rs:86-100, src/field.rs:536-568
//   DenseMultilinearExtension::evaluate        src/poly_f/mle/dense.rs:35-41 (via <q0-row, q1>)
//
// The proof stream has a fixed layout once every Merkle record carries be64(depth) as its length
// prefix (src/zip/pcs_transcript.rs:198-211), so every (column, row) opening is checked
// independently: one thread hashes one path and contributes one term to the two column inner
// products.  Records with another prefix are counted as malformed (the reference would lose its
// place in the stream there).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blake3.cuh"
#include "kernels_open.cuh"

namespace zipk {

// ---------------------------------------------------------------------------------------
// RAA encoding of ONE row: repeat, permute, accumulate, permute, accumulate
// (code_raa.rs:89-105).  O(cw) work, one workgroup.
//   FIELD = false: L-limb two's-complement integers (Int<M>).  The reference accumulates with
//                  checked additions (src/field/int.rs:122-134: overflow panics); the sums are
//                  formed exactly in L+1 limbs and *overflow is set when a prefix leaves L limbs.
//   FIELD = true : L-limb field elements, modular additions (config.rs:53-58).
// tmp: cw * L limbs of scratch in global memory.
// ---------------------------------------------------------------------------------------
template <int L, bool FIELD>
struct EncElem {
    static constexpr int W = FIELD ? L : L + 1;
    uint64_t v[W];
};

template <int L, bool FIELD>
__device__ __forceinline__ void enc_zero(EncElem<L, FIELD> &x) {
#pragma unroll
    for (int i = 0; i < EncElem<L, FIELD>::W; i++) x.v[i] = 0;
}
template <int L, bool FIELD>
__device__ __forceinline__ void enc_load(EncElem<L, FIELD> &x, const uint64_t *p) {
#pragma unroll
    for (int i = 0; i < L; i++) x.v[i] = p[i];
    if constexpr (!FIELD) x.v[L] = (uint64_t)((int64_t)x.v[L - 1] >> 63);  // sign extension
}
template <int L, bool FIELD>
__device__ __forceinline__ void enc_add(EncElem<L, FIELD> &a, const EncElem<L, FIELD> &b, const FieldDev<L> &f) {
    constexpr int W = EncElem<L, FIELD>::W;
    const uint64_t carry = add_n<W>(a.v, b.v);
    if constexpr (FIELD) {
        if (carry || geq_n<L>(a.v, f.modulus)) sub_n<L>(a.v, f.modulus);
    }
}
// stores the L limbs; returns true when an integer does not fit them
template <int L, bool FIELD>
__device__ __forceinline__ bool enc_store(const EncElem<L, FIELD> &x, uint64_t *p) {
#pragma unroll
    for (int i = 0; i < L; i++) p[i] = x.v[i];
    if constexpr (!FIELD) return x.v[L] != (uint64_t)((int64_t)x.v[L - 1] >> 63);
    return false;
}

template <int L, bool FIELD>
__global__ void __launch_bounds__(1024) encode_row_kernel(const uint64_t *in, uint32_t row_len, uint32_t cw,
                                                          const uint32_t *perm1, const uint32_t *perm2,
                                                          uint64_t *tmp, uint64_t *out, FieldDev<L> f,
                                                          uint32_t *overflow) {
    extern __shared__ __align__(16) unsigned char enc_smem[];
    using El = EncElem<L, FIELD>;
    El *tot = reinterpret_cast<El *>(enc_smem);  // [blockDim.x]
    const uint32_t T = blockDim.x, tid = threadIdx.x;
    const uint32_t per = (cw + T - 1) / T;
    const uint32_t j0 = tid * per, j1 = min(j0 + per, cw);
    bool ovf = false;
    for (int pass = 0; pass < 2; pass++) {
        // source of element j: pass 0 = repeated input row through pi1, pass 1 = tmp through pi2
        auto fetch = [&](uint32_t j, El &x) {
            if (pass == 0) enc_load<L, FIELD>(x, in + (size_t)(perm1[j] & (row_len - 1)) * L);
            else enc_load<L, FIELD>(x, tmp + (size_t)perm2[j] * L);
        };
        El sum;
        enc_zero<L, FIELD>(sum);
        for (uint32_t j = j0; j < j1; j++) {
            El x;
            fetch(j, x);
            enc_add<L, FIELD>(sum, x, f);
        }
        tot[tid] = sum;
        __syncthreads();
        // inclusive scan of the T chunk totals in place (Hillis-Steele, log2 T rounds) ...
        for (uint32_t off = 1; off < T; off <<= 1) {
            El x;
            enc_zero<L, FIELD>(x);
            if (tid >= off) x = tot[tid - off];
            __syncthreads();
            if (tid >= off) {
                El y = tot[tid];
                enc_add<L, FIELD>(y, x, f);
                tot[tid] = y;
            }
            __syncthreads();
        }
        // ... of which thread t needs the exclusive value: the sum of the chunks before its own
        El run;
        enc_zero<L, FIELD>(run);
        if (tid) run = tot[tid - 1];
        uint64_t *dst = pass == 0 ? tmp : out;
        // pass 1 reads tmp while pass 0 of no other thread writes it any more; pass 0 writes tmp
        // only after every thread has finished READING `in`: no hazard.  Pass 1 writes `out`.
        for (uint32_t j = j0; j < j1; j++) {
            El x;
            fetch(j, x);
            enc_add<L, FIELD>(run, x, f);
            ovf |= enc_store<L, FIELD>(run, dst + (size_t)j * L);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (ovf && overflow) atomicOr(overflow, 1u);
}

// ---------------------------------------------------------------------------------------
// FieldMap for a K = Int<4> column entry (conversion.rs:86-100 over field.rs:536-568):
// |v| reduced modulo q, Montgomery form, negated when v < 0.  `fq` describes 2^256 - q and is
// used instead of q for the reduction when the modulus has its top bit set in 4 limbs -- the
// reference's `%=` reads the modulus as a negative Int (oracle/zip_oracle.c,
// field_from_signed_words).
// ---------------------------------------------------------------------------------------
template <int FL>
__device__ __forceinline__ void field_from_int256(const uint64_t (&v)[4], const FieldDev<FL> &f,
                                                  const FieldDev<FL> &fq, bool quirk, uint64_t (&out)[FL]) {
    const bool neg = (int64_t)v[3] < 0;
    uint64_t mag[4];
    {
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const u128 t = (u128)0 - v[i] - borrow;
            mag[i] = neg ? (uint64_t)t : v[i];
            borrow = (uint64_t)(t >> 64) & 1;
        }
    }
    if constexpr (FL == 4) {
        if (quirk) {
            uint64_t t[4], wide[8];
            mont_mul<4>(mag, fq.r2, fq, t);  // |v| * R mod q'
#pragma unroll
            for (int i = 0; i < 8; i++) wide[i] = i < 4 ? t[i] : 0;
            mont_redc<4>(wide, fq, mag);  // |v| mod q'
        }
        mont_mul<4>(mag, f.r2, f, out);
    } else {
        uint64_t lo[FL], hi[FL], a[FL], b[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) {
            lo[i] = mag[i];
            hi[i] = (FL + i < 4) ? mag[FL + i] : 0;
        }
        mont_mul<FL>(lo, f.r2, f, a);   // lo * R
        mont_mul<FL>(hi, f.r2, f, b);   // hi * R
        mont_mul<FL>(b, f.r2, f, hi);   // hi * R^2 = (hi * 2^(64 FL)) * R
        const uint64_t c = add_n<FL>(a, hi);
        if (c || geq_n<FL>(a, f.modulus)) sub_n<FL>(a, f.modulus);
#pragma unroll
        for (int i = 0; i < FL; i++) out[i] = a[i];
    }
    if (neg) {
        bool zero = true;
#pragma unroll
        for (int i = 0; i < FL; i++) zero &= out[i] == 0;
        if (!zero) {
            uint64_t q[FL];
#pragma unroll
            for (int i = 0; i < FL; i++) q[i] = f.modulus[i];
            sub_n<FL>(q, out);
#pragma unroll
            for (int i = 0; i < FL; i++) out[i] = q[i];
        }
    }
}

struct VerifyColsArgs {
    const uint8_t *wire;     // column section of the proof (device)
    const uint32_t *cols;    // [n_cols]
    const int64_t *coeffs;   // [num_rows] or null (num_rows == 1)
    const uint64_t *q0;      // [num_rows][FL] Montgomery, or null (num_rows == 1)
    const uint32_t *roots;   // [num_rows][8]
    uint32_t num_rows, depth, n_cols, quirk;
    uint64_t *part_int;      // [n_cols][row blocks][6]
    uint64_t *part_f;        // [n_cols][row blocks][FL]
    uint32_t *bad_merkle;    // [n_cols] paths that do not reach their root
    uint32_t *malformed;     // [n_cols] records whose length prefix is not be64(depth)
};

// Grid (n_cols, ceil(num_rows / 256)), one thread per opened entry.
template <int FL>
__global__ void __launch_bounds__(256) verify_columns_kernel(VerifyColsArgs a, FieldDev<FL> f, FieldDev<FL> fq) {
    __shared__ uint64_t red[256 * (6 + FL)];
    const uint32_t ci = blockIdx.x, tid = threadIdx.x;
    const uint32_t r = blockIdx.y * 256 + tid;
    const uint32_t col = a.cols[ci];
    const uint32_t d = a.depth, rec_bytes = 8 + 32 * d;
    const size_t col_bytes = (size_t)a.num_rows * (32 + rec_bytes);
    const uint8_t *base = a.wire + (size_t)ci * col_bytes;
    uint64_t si[6] = {0, 0, 0, 0, 0, 0}, sf[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) sf[i] = 0;
    if (r < a.num_rows) {
        uint64_t v[4];
        const uint64_t *vp = reinterpret_cast<const uint64_t *>(base + (size_t)r * 32);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = vp[i];
        // ---- coeffs[r] * expand(v) in 384-bit two's complement (verify_z.rs:114-120) ----
        if (a.coeffs) {
            const int64_t c = a.coeffs[r];
            const uint64_t cu = (uint64_t)c;
            uint64_t carry = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u128 x = (u128)v[i] * cu + carry;
                si[i] = (uint64_t)x;
                carry = (uint64_t)(x >> 64);
            }
            si[4] = carry;
            si[5] = 0;
            if ((int64_t)v[3] < 0) {  // v = v_u - 2^256
                const u128 x = (u128)si[4] - cu;
                si[4] = (uint64_t)x;
                si[5] -= (uint64_t)(x >> 64) & 1;
            }
            if (c < 0) {  // c = c_u - 2^64, with v sign-extended to 5 limbs above bit 64
                uint64_t borrow = 0;
                const uint64_t ext = (uint64_t)((int64_t)v[3] >> 63);
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const uint64_t sub = i < 4 ? v[i] : ext;
                    const u128 x = (u128)si[i + 1] - sub - borrow;
                    si[i + 1] = (uint64_t)x;
                    borrow = (uint64_t)(x >> 64) & 1;
                }
            }
        }
        // ---- q0[r] (x) phi(v)  (verify_z.rs:176-183) ----
        uint64_t e[FL];
        field_from_int256<FL>(v, f, fq, a.quirk != 0, e);
        if (a.q0) {
            uint64_t q[FL];
#pragma unroll
            for (int i = 0; i < FL; i++) q[i] = a.q0[(size_t)r * FL + i];
            mont_mul<FL>(q, e, f, sf);
        } else {
#pragma unroll
            for (int i = 0; i < FL; i++) sf[i] = e[i];
        }
        // ---- MerkleProof::verify (pcs/utils.rs:178-210) ----
        const uint64_t *rec = reinterpret_cast<const uint64_t *>(base + (size_t)a.num_rows * 32 + (size_t)r * rec_bytes);
        if (rec[0] != __builtin_bswap64((uint64_t)d)) {
            atomicAdd(&a.malformed[ci], 1u);
        } else {
            uint32_t cur[8];
            blake3_leaf_limbs<4>(v, cur);
            uint32_t index = col;
            for (uint32_t l = 0; l < d; l++) {
                uint32_t sib[8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint64_t w = rec[1 + 4 * l + i];
                    sib[2 * i] = (uint32_t)w;
                    sib[2 * i + 1] = (uint32_t)(w >> 32);
                }
                uint32_t m[16], h[8];
                const bool right = index & 1u;  // current node is the right child
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    m[i] = right ? sib[i] : cur[i];
                    m[8 + i] = right ? cur[i] : sib[i];
                }
                blake3_block64(m, h);
#pragma unroll
                for (int i = 0; i < 8; i++) cur[i] = h[i];
                index >>= 1;
            }
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 8; i++) ok &= cur[i] == a.roots[(size_t)r * 8 + i];
            if (!ok) atomicAdd(&a.bad_merkle[ci], 1u);
        }
    }
    // ---- block sums: 384-bit wrap-around adds and modular adds ----
#pragma unroll
    for (int i = 0; i < 6; i++) red[tid * (6 + FL) + i] = si[i];
#pragma unroll
    for (int i = 0; i < FL; i++) red[tid * (6 + FL) + 6 + i] = sf[i];
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            uint64_t x[6], y[6], p[FL], q[FL];
#pragma unroll
            for (int i = 0; i < 6; i++) { x[i] = red[tid * (6 + FL) + i]; y[i] = red[(tid + s) * (6 + FL) + i]; }
#pragma unroll
            for (int i = 0; i < FL; i++) { p[i] = red[tid * (6 + FL) + 6 + i]; q[i] = red[(tid + s) * (6 + FL) + 6 + i]; }
            add_n<6>(x, y);
            const uint64_t c = add_n<FL>(p, q);
            if (c || geq_n<FL>(p, f.modulus)) sub_n<FL>(p, f.modulus);
#pragma unroll
            for (int i = 0; i < 6; i++) red[tid * (6 + FL) + i] = x[i];
#pragma unroll
            for (int i = 0; i < FL; i++) red[tid * (6 + FL) + 6 + i] = p[i];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const size_t slot = (size_t)ci * gridDim.y + blockIdx.y;
#pragma unroll
        for (int i = 0; i < 6; i++) a.part_int[slot * 6 + i] = red[i];
#pragma unroll
        for (int i = 0; i < FL; i++) a.part_f[slot * FL + i] = red[6 + i];
    }
}

// Per opened column: fold the row-block partials and compare with the encoded combined rows.
// flags[ci]: bit 0 = proximity test over Z failed (verify_z.rs:122-125), bit 1 = over F_q (:184-186).
template <int FL>
__global__ void __launch_bounds__(256) verify_finalize_kernel(const uint64_t *part_int, const uint64_t *part_f,
                                                              uint32_t blocks, const uint32_t *cols, uint32_t n_cols,
                                                              const uint64_t *enc_int, uint32_t m_limbs,
                                                              const uint64_t *enc_f, uint32_t *flags, FieldDev<FL> f) {
    const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n_cols) return;
    const uint32_t col = cols[ci];
    uint32_t fl = 0;
    uint64_t s[6] = {0, 0, 0, 0, 0, 0}, p[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) p[i] = 0;
    for (uint32_t b = 0; b < blocks; b++) {
        uint64_t y[6], q[FL];
#pragma unroll
        for (int i = 0; i < 6; i++) y[i] = part_int[((size_t)ci * blocks + b) * 6 + i];
#pragma unroll
        for (int i = 0; i < FL; i++) q[i] = part_f[((size_t)ci * blocks + b) * FL + i];
        add_n<6>(s, y);
        const uint64_t c = add_n<FL>(p, q);
        if (c || geq_n<FL>(p, f.modulus)) sub_n<FL>(p, f.modulus);
    }
    if (enc_int) {
        const uint64_t sign = (uint64_t)((int64_t)s[5] >> 63);
        for (uint32_t i = 0; i < m_limbs; i++)
            if (enc_int[(size_t)col * m_limbs + i] != (i < 6 ? s[i] : sign)) fl |= 1u;
    }
#pragma unroll
    for (int i = 0; i < FL; i++)
        if (enc_f[(size_t)col * FL + i] != p[i]) fl |= 2u;
    flags[ci] = fl;
}

// out = sum_c a[c] (x) b[c]  (Montgomery values; inner_product, src/zip/utils.rs).  One workgroup.
template <int FL>
__global__ void __launch_bounds__(1024) field_dot_kernel(const uint64_t *a, const uint64_t *b, uint32_t n,
                                                         uint64_t *out, FieldDev<FL> f) {
    __shared__ uint64_t red[1024 * FL];
    const uint32_t tid = threadIdx.x;
    uint64_t acc[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) acc[i] = 0;
    for (uint32_t c = tid; c < n; c += blockDim.x) {
        uint64_t x[FL], y[FL], t[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) { x[i] = a[(size_t)c * FL + i]; y[i] = b[(size_t)c * FL + i]; }
        mont_mul<FL>(x, y, f, t);
        const uint64_t cy = add_n<FL>(acc, t);
        if (cy || geq_n<FL>(acc, f.modulus)) sub_n<FL>(acc, f.modulus);
    }
#pragma unroll
    for (int i = 0; i < FL; i++) red[tid * FL + i] = acc[i];
    __syncthreads();
    for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) {
            uint64_t p[FL], q[FL];
#pragma unroll
            for (int i = 0; i < FL; i++) { p[i] = red[tid * FL + i]; q[i] = red[(tid + s) * FL + i]; }
            const uint64_t cy = add_n<FL>(p, q);
            if (cy || geq_n<FL>(p, f.modulus)) sub_n<FL>(p, f.modulus);
#pragma unroll
            for (int i = 0; i < FL; i++) red[tid * FL + i] = p[i];
        }
        __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < FL; i++) out[i] = red[i];
    }
}

// read_field_elements (pcs_transcript.rs:138-160): big-endian bytes -> Montgomery limbs, no range
// check in the reference; *noncanonical counts elements >= q (see zip_verify in zip_hip.h).
template <int FL>
__global__ void __launch_bounds__(256) decode_field_row_kernel(const uint8_t *row_be, uint32_t n, uint64_t *limbs,
                                                               uint32_t *noncanonical, FieldDev<FL> f) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    uint64_t v[FL];
    const uint64_t *src = reinterpret_cast<const uint64_t *>(row_be + (size_t)c * 8 * FL);
#pragma unroll
    for (int i = 0; i < FL; i++) v[i] = __builtin_bswap64(src[FL - 1 - i]);
    if (geq_n<FL>(v, f.modulus)) atomicAdd(noncanonical, 1u);
#pragma unroll
    for (int i = 0; i < FL; i++) limbs[(size_t)c * FL + i] = v[i];
}

// Diagnostic: FieldMap of arbitrary Int<4> values (the verifier's column entries are attacker
// chosen 256-bit integers; honest ones never exceed 96 bits), checked against the oracle in tests.
template <int FL>
__global__ void __launch_bounds__(256) field_map_int256_kernel(const uint64_t *vals, uint32_t n, uint64_t *out,
                                                               FieldDev<FL> f, FieldDev<FL> fq, uint32_t quirk) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t v[4], e[FL];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = vals[(size_t)i * 4 + k];
    field_from_int256<FL>(v, f, fq, quirk != 0, e);
#pragma unroll
    for (int k = 0; k < FL; k++) out[(size_t)i * FL + k] = e[k];
}

}  // namespace zipk
