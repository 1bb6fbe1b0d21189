// The O(n) field loops of SpartanProver::prove around its two sumchecks (BASELINE configs[4]:
// ZincProver end to end), i.e. everything src/zinc/prover.rs does between the transcript calls:
//
//   field_map_i64_kernel     z_ccs / matrix entries -> F_q      prover.rs:236, sparse_matrix.rs:38-58
//                                                              (FieldMap for Int<1>: conversion.rs:86-100)
//   eq_table_kernel          build_eq_x_r                      sumcheck/utils.rs:102-177
//   spmv_rows_kernel         calculate_Mz_mles: M_k z          zinc/utils.rs:121-135, ccs/utils.rs:47-76
//   second_table_kernel      sum_k gamma^k * compute_eval_table_sparse(M_k, eq(r_x))
//                                                              sparse_matrix.rs:165-182 + prover.rs:279-290
//   field_dot_partials_kernel  V_s[k] = Mz_k(r_x) = <Mz_k, eq(r_x)>   prover.rs:330-347
//   matrix_eval_partials_kernel  mle[M_k](r_x, r_y), the verifier's V_xy   verifier.rs:248-261
//
// All of it is exact arithmetic on canonical Montgomery residues, so summation order is free and the
// results are the reference's bit for bit.  The matrices are CSR for M z (one thread per row) and the
// transposed CSC for the column table (one thread per column: the reference's scatter-add becomes a
// gather, no atomics).  HBM-bound integer work; nothing here is a GEMM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_sumcheck.cuh"

namespace zipk {

// phi(v) for a 64-bit signed v: |v| (mod 2^(64 FL) - q first when the modulus has its top bit set and that
// difference fits a limb -- the reference reads the modulus as a signed Int, see make_field), times R,
// negated for v < 0.
template <int FL>
__device__ __forceinline__ void field_from_i64(int64_t v, const FieldDev<FL> &f, uint64_t quirk_mod, uint64_t (&out)[FL]) {
    uint64_t mag[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) mag[i] = 0;
    mag[0] = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    if (quirk_mod) mag[0] %= quirk_mod;
    mont_mul<FL>(mag, f.r2, f, out);
    if (v < 0) {
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

// out[i] = phi(in[i]) for i < n_in, 0 for n_in <= i < n_out (Vec::resize with zero / MLE padding)
template <int FL>
__global__ void __launch_bounds__(256) field_map_i64_kernel(const int64_t *in, uint64_t n_in, uint64_t n_out, uint64_t *out,
                                                            FieldDev<FL> f, uint64_t quirk_mod) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t e[FL];
        if (i < n_in) {
            field_from_i64<FL>(in[i], f, quirk_mod, e);
        } else {
#pragma unroll
            for (int k = 0; k < FL; k++) e[k] = 0;
        }
        fe_store<FL>(out + i * FL, e);
    }
}

// eq(x, r)[i] = prod_j (bit_j(i) ? r_j : 1 - r_j), variable 0 = least significant bit of i.
// r: nv Montgomery elements in HBM.  The reference builds the table by doubling; products of exact
// residues do not depend on the order.
// (large tables are built from two small ones, see eq_outer_kernel)
template <int FL>
__global__ void __launch_bounds__(256) eq_table_kernel(const uint64_t *r, uint32_t nv, uint64_t *out, FieldDev<FL> f) {
    __shared__ uint64_t fac[2][32][FL];  // [bit][j]
    const uint32_t tid = threadIdx.x;
    if (tid < nv) {
        uint64_t rj[FL], one[FL], unit[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) unit[i] = 0;
        unit[0] = 1;
        mont_mul<FL>(unit, f.r2, f, one);  // R mod q
        fe_load<FL>(rj, r + (size_t)tid * FL);
        fe_sub<FL>(one, rj, f);
#pragma unroll
        for (int i = 0; i < FL; i++) { fac[0][tid][i] = one[i]; fac[1][tid][i] = rj[i]; }
    }
    __syncthreads();
    const uint64_t n = (uint64_t)1 << nv;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + tid; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t acc[FL];
#pragma unroll
        for (int k = 0; k < FL; k++) acc[k] = fac[i & 1][0][k];
        for (uint32_t j = 1; j < nv; j++) {
            uint64_t x[FL], t[FL];
#pragma unroll
            for (int k = 0; k < FL; k++) x[k] = fac[(i >> j) & 1][j][k];
            mont_mul<FL>(acc, x, f, t);
#pragma unroll
            for (int k = 0; k < FL; k++) acc[k] = t[k];
        }
        fe_store<FL>(out + i * FL, acc);
    }
}

// eq over nv variables as the outer product of the tables over the low nv_lo and the remaining high
// variables: out[i] = lo[i mod 2^nv_lo] (x) hi[i >> nv_lo] -- one multiplication per entry instead of nv - 1.
template <int FL>
__global__ void __launch_bounds__(256) eq_outer_kernel(const uint64_t *lo, const uint64_t *hi, uint32_t nv_lo, uint64_t n,
                                                       uint64_t *out, FieldDev<FL> f) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t a[FL], b[FL], t[FL];
        fe_load<FL>(a, lo + (i & (((uint64_t)1 << nv_lo) - 1)) * FL);
        fe_load<FL>(b, hi + (i >> nv_lo) * FL);
        mont_mul<FL>(a, b, f, t);
        fe_store<FL>(out + i * FL, t);
    }
}

// out[row] = sum_e vals[e] (x) z[col[e]]  (row < n_rows), 0 for n_rows <= row < m
template <int FL>
__global__ void __launch_bounds__(256) spmv_rows_kernel(const uint32_t *row_ptr, const uint32_t *col_idx, const uint64_t *vals,
                                                        const uint64_t *z, uint32_t n_rows, uint32_t m, uint64_t *out,
                                                        FieldDev<FL> f) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    uint64_t acc[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) acc[i] = 0;
    if (row < n_rows) {
        for (uint32_t e = row_ptr[row]; e < row_ptr[row + 1]; e++) {
            uint64_t v[FL], x[FL], t[FL];
            fe_load<FL>(v, vals + (size_t)e * FL);
            fe_load<FL>(x, z + (size_t)col_idx[e] * FL);
            mont_mul<FL>(x, v, f, t);
            fe_add<FL>(acc, t, f);
        }
    }
    fe_store<FL>(out + (size_t)row * FL, acc);
}

// ---- transposing a CSR matrix on the device (for second_table_kernel) -------------------------------
// counts[col + 1] += 1 per entry; an inclusive scan over the m + 1 counters gives col_ptr; then every entry is
// placed at an atomically advanced per-column cursor.  The order inside a column is whatever the atomics give --
// the column sums are exact field sums, so it does not matter.
__global__ void __launch_bounds__(256) csc_count_kernel(const uint32_t *col_idx, uint32_t nnz, uint32_t *counts) {
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += gridDim.x * blockDim.x)
        atomicAdd(&counts[col_idx[e] + 1], 1u);
}

constexpr uint32_t kScanBlock = 1024, kScanItems = 4, kScanTile = kScanBlock * kScanItems;

// inclusive scan of one tile in LDS; returns this thread's four results and the tile total
__device__ __forceinline__ uint32_t tile_inclusive_scan(uint32_t (&v)[kScanItems], uint32_t *lds) {
    const uint32_t tid = threadIdx.x;
#pragma unroll
    for (uint32_t i = 1; i < kScanItems; i++) v[i] += v[i - 1];
    lds[tid] = v[kScanItems - 1];
    __syncthreads();
    for (uint32_t off = 1; off < kScanBlock; off <<= 1) {
        const uint32_t add = tid >= off ? lds[tid - off] : 0u;
        __syncthreads();
        lds[tid] += add;
        __syncthreads();
    }
    const uint32_t before = tid ? lds[tid - 1] : 0u;
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) v[i] += before;
    return lds[kScanBlock - 1];
}

// phase 1: tile sums; phase 3 (offsets != nullptr): in-place inclusive scan with the tile's offset added
__global__ void __launch_bounds__(kScanBlock) scan_tiles_kernel(uint32_t *data, uint32_t n, uint32_t *tile_sums,
                                                                const uint32_t *offsets) {
    __shared__ uint32_t lds[kScanBlock];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) v[i] = base + i < n ? data[base + i] : 0u;
    const uint32_t total = tile_inclusive_scan(v, lds);
    if (offsets) {
        const uint32_t off = offsets[blockIdx.x];
#pragma unroll
        for (uint32_t i = 0; i < kScanItems; i++)
            if (base + i < n) data[base + i] = v[i] + off;
    } else if (threadIdx.x == 0) {
        tile_sums[blockIdx.x] = total;
    }
}

// phase 2: exclusive scan of the tile sums (one workgroup, tile after tile)
__global__ void __launch_bounds__(kScanBlock) scan_sums_kernel(uint32_t *sums, uint32_t n) {
    __shared__ uint32_t lds[kScanBlock];
    uint32_t carry = 0;
    for (uint32_t start = 0; start < n; start += kScanTile) {
        const uint32_t base = start + threadIdx.x * kScanItems;
        uint32_t v[kScanItems], orig[kScanItems];
#pragma unroll
        for (uint32_t i = 0; i < kScanItems; i++) orig[i] = v[i] = base + i < n ? sums[base + i] : 0u;
        const uint32_t total = tile_inclusive_scan(v, lds);
#pragma unroll
        for (uint32_t i = 0; i < kScanItems; i++)
            if (base + i < n) sums[base + i] = v[i] - orig[i] + carry;
        carry += total;
        __syncthreads();
    }
}

template <int FL>
__global__ void __launch_bounds__(256) csc_fill_kernel(const uint32_t *row_ptr, const uint32_t *col_idx, const uint64_t *vals,
                                                       uint32_t n_rows, uint32_t *cursor, uint32_t *row_idx, uint64_t *vals_t) {
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < n_rows; row += gridDim.x * blockDim.x)
        for (uint32_t e = row_ptr[row]; e < row_ptr[row + 1]; e++) {
            const uint32_t dst = atomicAdd(&cursor[col_idx[e]], 1u);
            row_idx[dst] = row;
            uint64_t v[FL];
            fe_load<FL>(v, vals + (size_t)e * FL);
            fe_store<FL>(vals_t + (size_t)dst * FL, v);
        }
}

constexpr int kCcsMaxMatrices = 7;

struct SecondTableArgs {
    const uint32_t *col_ptr[kCcsMaxMatrices];  // CSC of M_k: n_cols + 1
    const uint32_t *row_idx[kCcsMaxMatrices];
    const uint64_t *vals[kCcsMaxMatrices];     // Montgomery, CSC order
    uint32_t t, m;
    const uint64_t *eq;                        // eq(r_x): m elements
    uint64_t *out;                             // m elements
};

// out[col] = sum_k gamma^k * sum_{rows of column col} eq[row] (x) M_k[row][col], folded as the reference
// does (highest k first: lin = lin * gamma + table_k[col]).
template <int FL>
__global__ void __launch_bounds__(256) second_table_kernel(SecondTableArgs a, const uint64_t *gamma_d, FieldDev<FL> f) {
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= a.m) return;
    uint64_t gamma[FL], lin[FL];
    fe_load<FL>(gamma, gamma_d);
#pragma unroll
    for (int i = 0; i < FL; i++) lin[i] = 0;
    for (int k = (int)a.t - 1; k >= 0; k--) {
        uint64_t acc[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) acc[i] = 0;
        for (uint32_t e = a.col_ptr[k][col]; e < a.col_ptr[k][col + 1]; e++) {
            uint64_t v[FL], x[FL], t[FL];
            fe_load<FL>(v, a.vals[k] + (size_t)e * FL);
            fe_load<FL>(x, a.eq + (size_t)a.row_idx[k][e] * FL);
            mont_mul<FL>(x, v, f, t);
            fe_add<FL>(acc, t, f);
        }
        uint64_t t[FL];
        mont_mul<FL>(lin, gamma, f, t);
        fe_add<FL>(t, acc, f);
#pragma unroll
        for (int i = 0; i < FL; i++) lin[i] = t[i];
    }
    fe_store<FL>(a.out + (size_t)col * FL, lin);
}

// mle[M](r_x, r_y) = sum_{row, e} eq_x[row] (x) M[row][col_e] (x) eq_y[col_e]
// (DenseMultilinearExtension::from_matrix + evaluate, src/poly_f/mle/dense.rs:69-87: index = rows * col + row, the
// low s variables select the row).  One thread per row, block sums to `partials`, then sumcheck_reduce_kernel.
template <int FL>
__global__ void __launch_bounds__(256) matrix_eval_partials_kernel(const uint32_t *row_ptr, const uint32_t *col_idx,
                                                                   const uint64_t *vals, const uint64_t *eq_x,
                                                                   const uint64_t *eq_y, uint32_t n_rows, uint64_t *partials,
                                                                   FieldDev<FL> f) {
    __shared__ uint64_t red[256 * FL];
    const uint32_t tid = threadIdx.x;
    uint64_t acc[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) acc[i] = 0;
    for (uint32_t row = blockIdx.x * blockDim.x + tid; row < n_rows; row += gridDim.x * blockDim.x) {
        uint64_t inner[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) inner[i] = 0;
        for (uint32_t e = row_ptr[row]; e < row_ptr[row + 1]; e++) {
            uint64_t v[FL], y[FL], t[FL];
            fe_load<FL>(v, vals + (size_t)e * FL);
            fe_load<FL>(y, eq_y + (size_t)col_idx[e] * FL);
            mont_mul<FL>(v, y, f, t);
            fe_add<FL>(inner, t, f);
        }
        uint64_t x[FL], t[FL];
        fe_load<FL>(x, eq_x + (size_t)row * FL);
        mont_mul<FL>(inner, x, f, t);
        fe_add<FL>(acc, t, f);
    }
    fe_store<FL>(red + (size_t)tid * FL, acc);
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            uint64_t p[FL], q[FL];
            fe_load<FL>(p, red + (size_t)tid * FL);
            fe_load<FL>(q, red + (size_t)(tid + s) * FL);
            fe_add<FL>(p, q, f);
            fe_store<FL>(red + (size_t)tid * FL, p);
        }
        __syncthreads();
    }
    if (tid < FL) partials[(size_t)blockIdx.x * FL + tid] = red[tid];
}

// partials[block] = sum over the block's share of a[i] (x) b[i]; summed by sumcheck_reduce_kernel (ne = 1)
template <int FL>
__global__ void __launch_bounds__(256) field_dot_partials_kernel(const uint64_t *a, const uint64_t *b, uint64_t n,
                                                                 uint64_t *partials, FieldDev<FL> f) {
    __shared__ uint64_t red[256 * FL];
    const uint32_t tid = threadIdx.x;
    uint64_t acc[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) acc[i] = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + tid; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x[FL], y[FL], t[FL];
        fe_load<FL>(x, a + i * FL);
        fe_load<FL>(y, b + i * FL);
        mont_mul<FL>(x, y, f, t);
        fe_add<FL>(acc, t, f);
    }
    fe_store<FL>(red + (size_t)tid * FL, acc);
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            uint64_t p[FL], q[FL];
            fe_load<FL>(p, red + (size_t)tid * FL);
            fe_load<FL>(q, red + (size_t)(tid + s) * FL);
            fe_add<FL>(p, q, f);
            fe_store<FL>(red + (size_t)tid * FL, p);
        }
        __syncthreads();
    }
    if (tid < FL) partials[(size_t)blockIdx.x * FL + tid] = red[tid];
}

}  // namespace zipk
