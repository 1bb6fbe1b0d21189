// Sumcheck prover rounds for the two combination functions ZincProver uses.
//
// Reference loops replaced:
//   IPForMLSumcheck::prove_round          src/sumcheck/prover.rs:62-180 with comb_fn = product of the MLE
//                                         values (second sumcheck, zinc/prover.rs:300) or the CCS form
//                                         (sum_t c_t prod_{j in S_t} v_j) * v_last (first sumcheck,
//                                         sumcheck_polynomial_comb_fn_1, zinc/utils.rs:77-94)
//   DenseMultilinearExtension::fix_variables  src/poly_f/mle/dense.rs:142-168
//
// One pass per round: the fold with the previous challenge (fix_variables) is fused into the
// evaluation of the next round polynomial, so a round reads each table once and writes the half-size
// folded tables (ping-pong buffers; the caller's tables are never written).  Everything is exact
// field arithmetic on canonical Montgomery residues, so the per-thread / per-block partial sums
// give the same evaluations as the reference's Rayon fold in any order.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "kernels_open.cuh"

namespace zipk {

constexpr int kSumcheckMaxMles = 4;
constexpr int kSumcheckMaxDegree = 4;

template <int FL>
__device__ __forceinline__ void fe_add(uint64_t (&a)[FL], const uint64_t (&b)[FL], const FieldDev<FL> &f) {
    const uint64_t c = add_n<FL>(a, b);
    if (c || geq_n<FL>(a, f.modulus)) sub_n<FL>(a, f.modulus);
}
template <int FL>
__device__ __forceinline__ void fe_sub(uint64_t (&a)[FL], const uint64_t (&b)[FL], const FieldDev<FL> &f) {
    if (sub_n<FL>(a, b)) {  // a < b: add the modulus back (wraps when q has no spare bit; the sum is exact)
        uint64_t q[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) q[i] = f.modulus[i];
        add_n<FL>(a, q);
    }
}
template <int FL>
__device__ __forceinline__ void fe_load(uint64_t (&a)[FL], const uint64_t *p) {
#pragma unroll
    for (int i = 0; i < FL; i++) a[i] = p[i];
}
template <int FL>
__device__ __forceinline__ void fe_store(uint64_t *p, const uint64_t (&a)[FL]) {
#pragma unroll
    for (int i = 0; i < FL; i++) p[i] = a[i];
}

constexpr int kSumcheckMaxTerms = 8;

// ---- lazy reduction of the per-thread sums ------------------------------------------------------------------
// The round message is a SUM of Montgomery products over the thread's hypercube points (prover.rs:128-150): instead of
// reducing every product (REDC costs more than the schoolbook product itself) the unreduced 2 FL-limb products are
// added up in a (2 FL + 1)-limb accumulator and reduced once per evaluation point and thread.  The canonical residue
// of the sum is unique, so the message is bit for bit the reference's.  Round 1 of the degree-2 product sumcheck at
// 2^24: 0.67 -> 0.46 ms (software-pipelined loads on top changed nothing: the 64-bit multiplies bound it).
// (mul_wide: kernels_open.cuh)
template <int FL>
__device__ __forceinline__ void acc_wide_add(uint64_t (&acc)[2 * FL + 1], const uint64_t (&t)[2 * FL]) {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 2 * FL; i++) {
        const u128 x = (u128)acc[i] + t[i] + carry;
        acc[i] = (uint64_t)x;
        carry = (uint64_t)(x >> 64);
    }
    acc[2 * FL] += carry;
}
// acc (sum of up to 2^64 products of canonical residues) -> acc * R^-1 mod q, canonical
template <int FL>
__device__ __forceinline__ void acc_wide_reduce(uint64_t (&acc)[2 * FL + 1], const FieldDev<FL> &f, uint64_t (&out)[FL]) {
    // Montgomery steps on the FL low limbs, carries run through the whole accumulator: afterwards
    // acc[FL .. 2 FL] = (acc + m q) / R  ==  acc R^-1 (mod q), below (number of products + 1) * q
#pragma unroll
    for (int i = 0; i < FL; i++) {
        const uint64_t k = acc[i] * f.inv;
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < FL; j++) {
            const u128 x = (u128)k * f.modulus[j] + acc[i + j] + carry;
            acc[i + j] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
#pragma unroll
        for (int j = i + FL; j <= 2 * FL; j++) {
            const u128 x = (u128)acc[j] + carry;
            acc[j] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
    }
    uint64_t y[FL + 2];
#pragma unroll
    for (int i = 0; i <= FL; i++) y[i] = acc[FL + i];
    y[FL + 1] = 0;
    reduce_wide<FL>(y, f, out);  // y mod q, canonical (y < 2^64 q << q R)
}

// f(integral_constant<int, E>) for E = FIRST .. LAST, unrolled by construction
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void sc_static_for(F &&f) {
    if constexpr (FIRST <= LAST) {
        f(std::integral_constant<int, FIRST>{});
        sc_static_for<FIRST + 1, LAST>(f);
    }
}

template <int FL>
struct SumcheckRoundArgs {
    const uint64_t *src[kSumcheckMaxMles];  // tables of this round's input (2*half entries, or 4*half when fold)
    uint64_t *dst[kSumcheckMaxMles];        // folded tables (2*half entries) when fold
    uint64_t r[FL];                         // previous round's challenge (Montgomery), when fold
    uint32_t degree, fold;
    uint64_t half;                          // number of hypercube points b of this round: 2^(nv - round)
    uint64_t *partials;                     // [gridDim.x][degree + 1][FL]
    uint32_t *done;                         // arrival counter (zero between launches): the last workgroup folds the
                                            // partials; nullptr = sumcheck_reduce_kernel does it in a second launch
    uint64_t *evals_out;                    // [degree + 1][FL], device memory or host-mapped pinned memory
    uint32_t *host_flag;                    // host-mapped word set to `seq` after the message (the host polls it), or null
    uint32_t seq;
    // combination function: n_terms == 0: the product of all MLE values; otherwise
    //   (sum_t coeff[t] * prod_{j in term_mask[t]} vals[j]) * vals[K - 1]   (zinc/utils.rs:77-94)
    uint32_t n_terms;
    uint32_t term_mask[kSumcheckMaxTerms];
    uint64_t coeff[kSumcheckMaxTerms][FL];
    // coefficients that are 1 or -1 (every R1CS-shaped CCS: c = [1, -1]) cost no multiplication:
    // 0 = general, 1 = one, 2 = minus one.  `one` = R mod q, for a term with an empty product.
    uint32_t coeff_kind[kSumcheckMaxTerms];
    uint64_t one[FL];
};

// c = the combination function of one point without its last factor: the product of val[0 .. K-2] (n_terms == 0), or
// sum_t coeff[t] * prod_{j in term_mask[t]} val[j] (the CCS form; coefficients 1 / -1 cost no multiplication)
template <int FL, int K>
__device__ __forceinline__ void sumcheck_comb(const SumcheckRoundArgs<FL> &a, const FieldDev<FL> &f, const uint64_t (&val)[K][FL],
                                              uint64_t (&c)[FL]) {
    if (a.n_terms == 0) {
#pragma unroll
        for (int i = 0; i < FL; i++) c[i] = K > 1 ? val[0][i] : a.one[i];
#pragma unroll
        for (int k = 1; k < K - 1; k++) {
            uint64_t t[FL];
            mont_mul<FL>(c, val[k], f, t);
#pragma unroll
            for (int i = 0; i < FL; i++) c[i] = t[i];
        }
    } else {
        uint64_t sum[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) sum[i] = 0;
        for (uint32_t tt = 0; tt < a.n_terms; tt++) {
            uint64_t term[FL];
            const uint32_t kind = a.coeff_kind[tt];  // wave-uniform
            bool have = kind == 0;
            if (have) {
#pragma unroll
                for (int i = 0; i < FL; i++) term[i] = a.coeff[tt][i];
            }
            const uint32_t m = a.term_mask[tt];
#pragma unroll
            for (int k = 0; k < K; k++) {
                if ((m >> k) & 1u) {  // wave-uniform
                    if (have) {
                        uint64_t t[FL];
                        mont_mul<FL>(term, val[k], f, t);
#pragma unroll
                        for (int i = 0; i < FL; i++) term[i] = t[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < FL; i++) term[i] = val[k][i];
                        have = true;
                    }
                }
            }
            if (!have) {
#pragma unroll
                for (int i = 0; i < FL; i++) term[i] = a.one[i];
            }
            if (kind == 2) fe_sub<FL>(sum, term, f);
            else fe_add<FL>(sum, term, f);
        }
#pragma unroll
        for (int i = 0; i < FL; i++) c[i] = sum[i];  // eq() is the last MLE
    }
}

template <int FL>
__device__ __forceinline__ void sumcheck_last_block_folds(const SumcheckRoundArgs<FL> &a, const FieldDev<FL> &f, uint64_t *red,
                                                          uint32_t ne, uint32_t tid) {
    // The workgroup that arrives last folds the per-block partials into the round message (one launch per
    // round instead of two).  Release: every wave makes its stores visible device-wide before the counter is
    // bumped; acquire: the folding workgroup fences again before it reads the other workgroups' partials.
    // Only for small grids: the device-wide release makes every wave write the dirty L2 back, which costs the early
    // rounds (hundreds of MB of freshly folded tables in the cache) far more than the second launch saves.
    if (!a.done) return;
    __shared__ uint32_t is_last;
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(a.done, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    for (uint32_t e = 0; e < ne; e++) {
        uint64_t sum[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) sum[i] = 0;
        for (uint32_t b = tid; b < gridDim.x; b += 256) {
            uint64_t p[FL];
            fe_load<FL>(p, a.partials + ((size_t)b * ne + e) * FL);
            fe_add<FL>(sum, p, f);
        }
        __syncthreads();  // red is reused
        fe_store<FL>(red + (size_t)tid * FL, sum);
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
        if (tid < FL) a.evals_out[(size_t)e * FL + tid] = red[tid];
    }
    if (tid == 0) {
        *a.done = 0;  // the next round's launch is ordered after this one on the stream
        if (a.host_flag) {  // the message was stored by this wave (tid < FL): order it before the flag, system-wide
            __threadfence_system();
            __hip_atomic_store(a.host_flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// K = number of MLEs, DEG = degree of the round polynomial (evaluations at 0..DEG); template
// parameters so that the per-MLE values and the per-point accumulators live in registers.
template <int FL, int K, int DEG>
__global__ void __launch_bounds__(256) sumcheck_round_kernel(SumcheckRoundArgs<FL> a, FieldDev<FL> f) {
    extern __shared__ __align__(16) unsigned char sc_smem[];
    uint64_t *red = reinterpret_cast<uint64_t *>(sc_smem);  // [256][DEG + 1][FL]
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t ne = DEG + 1;
    uint64_t wacc[DEG + 1][2 * FL + 1];  // unreduced sums of the LAST product of every point (lazy reduction, above)
#pragma unroll
    for (int e = 0; e <= DEG; e++)
#pragma unroll
        for (int i = 0; i <= 2 * FL; i++) wacc[e][i] = 0;
    uint64_t rr[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) rr[i] = a.r[i];

    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + tid; b < a.half; b += (uint64_t)gridDim.x * blockDim.x) {
        // the values of MLE k at t = 0 and t = 1 (the two table entries); from t = 2 on `val` holds the increment per
        // unit of t and `nxt` walks: v1 + step, ..  (prover.rs:128-150) -- the same canonical residues as adding the
        // increment from t = 0, two modular additions per MLE and pair cheaper
        uint64_t val[K][FL], nxt[K][FL];
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint64_t v1[FL];
            if (a.fold) {  // fix_variables with the previous challenge: p'[j] = p[2j] + r (p[2j+1] - p[2j])
                uint64_t l[FL], d[FL], t[FL];
                const uint64_t *s = a.src[k] + (size_t)(4 * b) * FL;
                fe_load<FL>(l, s);
                fe_load<FL>(d, s + FL);
                fe_sub<FL>(d, l, f);
                mont_mul<FL>(d, rr, f, t);
                fe_add<FL>(l, t, f);
#pragma unroll
                for (int i = 0; i < FL; i++) val[k][i] = l[i];
                fe_load<FL>(l, s + 2 * FL);
                fe_load<FL>(d, s + 3 * FL);
                fe_sub<FL>(d, l, f);
                mont_mul<FL>(d, rr, f, t);
                fe_add<FL>(l, t, f);
#pragma unroll
                for (int i = 0; i < FL; i++) v1[i] = l[i];
                fe_store<FL>(a.dst[k] + (size_t)(2 * b) * FL, val[k]);
                fe_store<FL>(a.dst[k] + (size_t)(2 * b + 1) * FL, v1);
            } else {
                const uint64_t *s = a.src[k] + (size_t)(2 * b) * FL;
                fe_load<FL>(val[k], s);
                fe_load<FL>(v1, s + FL);
            }
#pragma unroll
            for (int i = 0; i < FL; i++) nxt[k][i] = v1[i];
        }
        // (the evaluation point is a COMPILE-TIME index: behind a loop hipcc would not unroll -- K * DEG large -- the
        // accumulators wacc[e] became a stack object, 288-368 bytes of scratch per lane in the CCS shapes)
        auto point = [&](auto e_tag, const uint64_t (&val)[K][FL]) {
            constexpr int e = decltype(e_tag)::value;
            uint64_t c[FL], w[2 * FL];  // the point's value = c * val[K - 1], added up unreduced
            sumcheck_comb<FL, K>(a, f, val, c);
            mul_wide<FL>(c, val[K - 1], w);
            acc_wide_add<FL>(wacc[e], w);
        };
        point(std::integral_constant<int, 0>{}, val);
        if constexpr (DEG >= 1) point(std::integral_constant<int, 1>{}, nxt);
        if constexpr (DEG >= 2) {
#pragma unroll
            for (int k = 0; k < K; k++) {  // val <- step = v1 - v0
                uint64_t d[FL];
#pragma unroll
                for (int i = 0; i < FL; i++) d[i] = nxt[k][i];
                fe_sub<FL>(d, val[k], f);
#pragma unroll
                for (int i = 0; i < FL; i++) val[k][i] = d[i];
            }
            sc_static_for<2, DEG>([&](auto e_tag) {
#pragma unroll
                for (int k = 0; k < K; k++) fe_add<FL>(nxt[k], val[k], f);
                point(e_tag, nxt);
            });
        }
    }
    uint64_t acc[DEG + 1][FL];
#pragma unroll
    for (int e = 0; e <= DEG; e++) acc_wide_reduce<FL>(wacc[e], f, acc[e]);
    // block sum
#pragma unroll
    for (int e = 0; e <= DEG; e++)
#pragma unroll
        for (int i = 0; i < FL; i++) red[((size_t)tid * ne + e) * FL + i] = acc[e][i];
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            for (uint32_t e = 0; e < ne; e++) {
                uint64_t p[FL], q[FL];
                fe_load<FL>(p, red + ((size_t)tid * ne + e) * FL);
                fe_load<FL>(q, red + ((size_t)(tid + s) * ne + e) * FL);
                fe_add<FL>(p, q, f);
                fe_store<FL>(red + ((size_t)tid * ne + e) * FL, p);
            }
        }
        __syncthreads();
    }
    if (tid < ne * FL) a.partials[(size_t)blockIdx.x * ne * FL + tid] = red[tid];
    sumcheck_last_block_folds<FL>(a, f, red, ne, tid);
}

// ---- degree 3 (the first sumcheck of ZincProver: (M0 z * M1 z - M2 z) * eq) with FOUR LANES PER POINT ------------------
// sumcheck_round_kernel<4, 4, 3> holds four (2 FL + 1)-limb accumulators and both table entries of four MLEs per thread:
// 210 VGPRs, two waves per SIMD -- and with ONE wave per SIMD it takes 1.7x as long (tools/exp_sumcheck_occupancy.py): it
// is bound by the latency of its dependent multiply chains, not by the VALU.  Here the four evaluation points 0..3 of a
// hypercube point b are the four lanes of a quad: a lane keeps ONE accumulator and the K values at ITS point
// (v0, v1, v1 + step, v1 + 2 step: the same canonical residues the reference's running sum produces, prover.rs:128-150),
// <= 128 VGPRs, four waves per SIMD.  The lanes of a quad read the same table entries (one cache line, one request); in a
// folding round lane k folds MLE k (fix_variables, dense.rs:142-168) and the quad shares the folded pairs with DPP
// quad_perm broadcasts, so no multiplication is done twice.  Same partials layout and last-block fold as above.
template <int KK>
__device__ __forceinline__ uint64_t quad_bcast(uint64_t x) {
    constexpr int CTRL = KK * 0x55;  // quad_perm:[KK, KK, KK, KK]
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, CTRL, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), CTRL, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

template <int FL, int K, int DEG = 3>
__global__ void __launch_bounds__(256, 4) sumcheck_round_quad_kernel(SumcheckRoundArgs<FL> a, FieldDev<FL> f) {
    static_assert(DEG == 2 || DEG == 3, "a quad holds the points 0..3: degree 3, or degree 2 with its fourth lane idle");
    extern __shared__ __align__(16) unsigned char sc_smem[];
    uint64_t *red = reinterpret_cast<uint64_t *>(sc_smem);  // [256][FL]
    constexpr uint32_t ne = DEG + 1;  // (degree 2: lane 3 of a quad computes a fourth point nobody reads)
    const uint32_t tid = threadIdx.x, e = tid & 3u;
    uint64_t wacc[2 * FL + 1];
#pragma unroll
    for (int i = 0; i <= 2 * FL; i++) wacc[i] = 0;
    uint64_t rr[FL];
#pragma unroll
    for (int i = 0; i < FL; i++) rr[i] = a.r[i];
    const uint64_t is0 = e == 0u ? ~0ull : 0ull, ge2 = e >= 2u ? ~0ull : 0ull, ge3 = e >= 3u ? ~0ull : 0ull;
    // the MLE this lane folds (lane k of a quad: MLE k)
    const uint64_t *my_src = a.src[0];
    uint64_t *my_dst = a.dst[0];
#pragma unroll
    for (int k = 1; k < K; k++) {
        if (e == (uint32_t)k) {
            my_src = a.src[k];
            my_dst = a.dst[k];
        }
    }
    // the value at this lane's point from the entries at t = 0 and t = 1
    auto at_point = [&](const uint64_t (&v0)[FL], const uint64_t (&v1)[FL], uint64_t (&out)[FL]) {
        uint64_t step[FL], s2[FL], s3[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) {
            step[i] = v1[i];
            out[i] = (v0[i] & is0) | (v1[i] & ~is0);
        }
        fe_sub<FL>(step, v0, f);
#pragma unroll
        for (int i = 0; i < FL; i++) {
            s2[i] = step[i] & ge2;  // (adding 0 leaves a canonical residue as it is)
            s3[i] = step[i] & ge3;
        }
        fe_add<FL>(out, s2, f);
        fe_add<FL>(out, s3, f);
    };
    const uint64_t per_pass = (uint64_t)gridDim.x * 64u;
    for (uint64_t b = (uint64_t)blockIdx.x * 64u + (tid >> 2); b < a.half; b += per_pass) {
        uint64_t val[K][FL];
        if (a.fold) {
            uint64_t p0[FL], p1[FL];
#pragma unroll
            for (int i = 0; i < FL; i++) p0[i] = p1[i] = 0;
            if (e < (uint32_t)K) {  // p'[j] = p[2j] + r (p[2j+1] - p[2j])
                uint64_t d[FL], t[FL];
                const uint64_t *s = my_src + (size_t)(4 * b) * FL;
                fe_load<FL>(p0, s);
                fe_load<FL>(d, s + FL);
                fe_sub<FL>(d, p0, f);
                mont_mul<FL>(d, rr, f, t);
                fe_add<FL>(p0, t, f);
                fe_load<FL>(p1, s + 2 * FL);
                fe_load<FL>(d, s + 3 * FL);
                fe_sub<FL>(d, p1, f);
                mont_mul<FL>(d, rr, f, t);
                fe_add<FL>(p1, t, f);
                fe_store<FL>(my_dst + (size_t)(2 * b) * FL, p0);
                fe_store<FL>(my_dst + (size_t)(2 * b + 1) * FL, p1);
            }
            sc_static_for<0, K - 1>([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                uint64_t v0[FL], v1[FL];
#pragma unroll
                for (int i = 0; i < FL; i++) {
                    v0[i] = quad_bcast<k>(p0[i]);
                    v1[i] = quad_bcast<k>(p1[i]);
                }
                at_point(v0, v1, val[k]);
            });
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                uint64_t v0[FL], v1[FL];
                const uint64_t *s = a.src[k] + (size_t)(2 * b) * FL;
                fe_load<FL>(v0, s);
                fe_load<FL>(v1, s + FL);
                at_point(v0, v1, val[k]);
            }
        }
        uint64_t c[FL], w[2 * FL];
        sumcheck_comb<FL, K>(a, f, val, c);
        mul_wide<FL>(c, val[K - 1], w);
        acc_wide_add<FL>(wacc, w);
    }
    uint64_t acc[FL];
    acc_wide_reduce<FL>(wacc, f, acc);
    fe_store<FL>(red + (size_t)tid * FL, acc);
    __syncthreads();
    for (uint32_t s = 128; s >= 4; s >>= 1) {  // lanes with the same point: tid mod 4 survives every halving
        if (tid < s) {
            uint64_t p[FL], q[FL];
            fe_load<FL>(p, red + (size_t)tid * FL);
            fe_load<FL>(q, red + (size_t)(tid + s) * FL);
            fe_add<FL>(p, q, f);
            fe_store<FL>(red + (size_t)tid * FL, p);
        }
        __syncthreads();
    }
    if (tid < ne * FL) a.partials[(size_t)blockIdx.x * ne * FL + tid] = red[tid];  // red[e][FL], e = 0..DEG
    sumcheck_last_block_folds<FL>(a, f, red, ne, tid);
}

// evaluations[e] = sum over the blocks' partials (one workgroup; blocks <= a few thousand)
template <int FL>
__global__ void __launch_bounds__(256) sumcheck_reduce_kernel(const uint64_t *partials, uint32_t blocks, uint32_t ne,
                                                              uint64_t *evaluations, FieldDev<FL> f,
                                                              uint32_t *host_flag = nullptr, uint32_t seq = 0) {
    __shared__ uint64_t red[256 * FL];
    const uint32_t tid = threadIdx.x;
    for (uint32_t e = 0; e < ne; e++) {
        uint64_t acc[FL];
#pragma unroll
        for (int i = 0; i < FL; i++) acc[i] = 0;
        for (uint32_t b = tid; b < blocks; b += 256) {
            uint64_t p[FL];
            fe_load<FL>(p, partials + ((size_t)b * ne + e) * FL);
            fe_add<FL>(acc, p, f);
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
        if (tid < FL) evaluations[(size_t)e * FL + tid] = red[tid];
        __syncthreads();
    }
    if (host_flag && tid == 0) {  // as in sumcheck_round_kernel
        __threadfence_system();
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace zipk
