// BLAKE3 single-block hashing for gfx950, one hash per lane.
//
// Replaces the `blake3` crate calls of the reference Merkle tree:
//   leaf:  blake3::hash(Int<K>::to_bytes())          src/zip/pcs/utils.rs:87-93, src/field/int.rs:201-210
//   node:  Hasher::update(l) ; update(r) ; finalize   src/zip/pcs/utils.rs:95-118
// Both are exactly one compression: cv = IV, counter = 0, flags = CHUNK_START|CHUNK_END|ROOT,
// block_len = message length (public BLAKE3 specification).
//
// The kernel is VALU bound (12 VALU ops per G, 56 G per compression), so the
// whole compression is unrolled with a compile-time message schedule: the state
// and the message live in VGPRs, the permutation between rounds costs nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The compression in a fixed instruction order that alternates gfx950's double-rate VOP2 and single-rate VOP3
// integer opcodes (tools/gen_blake3_sched.py, profiles/round3_valu_issue.md): 3.4 instead of 4.0 issue cycles per
// instruction.  -DZIPK_B3_COMPILER_SCHED leaves the order to hipcc (A/B measurements).
#include "blake3_sched.inc"

namespace zipk {

#define ZIPK_B3_IV0 0x6A09E667u
#define ZIPK_B3_IV1 0xBB67AE85u
#define ZIPK_B3_IV2 0x3C6EF372u
#define ZIPK_B3_IV3 0xA54FF53Au
#define ZIPK_B3_IV4 0x510E527Fu
#define ZIPK_B3_IV5 0x9B05688Cu
#define ZIPK_B3_IV6 0x1F83D9ABu
#define ZIPK_B3_IV7 0x5BE0CD19u
#define ZIPK_B3_FLAGS 0x0Bu  // CHUNK_START | CHUNK_END | ROOT

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) {
    return __builtin_rotateright32(x, n);  // v_alignbit_b32
}

#define ZIPK_G(a, b, c, d, mx, my) \
    a = a + b + (mx);              \
    d = rotr(d ^ a, 16);           \
    c = c + d;                     \
    b = rotr(b ^ c, 12);           \
    a = a + b + (my);              \
    d = rotr(d ^ a, 8);            \
    c = c + d;                     \
    b = rotr(b ^ c, 7);

#define ZIPK_ROUND(m, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    ZIPK_G(v0, v4, v8, v12, m[s0], m[s1])                                                   \
    ZIPK_G(v1, v5, v9, v13, m[s2], m[s3])                                                   \
    ZIPK_G(v2, v6, v10, v14, m[s4], m[s5])                                                  \
    ZIPK_G(v3, v7, v11, v15, m[s6], m[s7])                                                  \
    ZIPK_G(v0, v5, v10, v15, m[s8], m[s9])                                                  \
    ZIPK_G(v1, v6, v11, v12, m[s10], m[s11])                                                \
    ZIPK_G(v2, v7, v8, v13, m[s12], m[s13])                                                 \
    ZIPK_G(v3, v4, v9, v14, m[s14], m[s15])

// One compression of a <=64-byte message held as 16 little-endian words
// (zero padded).  h receives the 8 digest words (little-endian serialisation).
__device__ __forceinline__ void blake3_block(const uint32_t (&m)[16], uint32_t block_len,
                                             uint32_t (&h)[8]) {
    uint32_t v0 = ZIPK_B3_IV0, v1 = ZIPK_B3_IV1, v2 = ZIPK_B3_IV2, v3 = ZIPK_B3_IV3;
    uint32_t v4 = ZIPK_B3_IV4, v5 = ZIPK_B3_IV5, v6 = ZIPK_B3_IV6, v7 = ZIPK_B3_IV7;
    uint32_t v8 = ZIPK_B3_IV0, v9 = ZIPK_B3_IV1, v10 = ZIPK_B3_IV2, v11 = ZIPK_B3_IV3;
    uint32_t v12 = 0u, v13 = 0u, v14 = block_len, v15 = ZIPK_B3_FLAGS;
    ZIPK_ROUND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    ZIPK_ROUND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
    ZIPK_ROUND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
    ZIPK_ROUND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
    ZIPK_ROUND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
    ZIPK_ROUND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
    ZIPK_ROUND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
    h[0] = v0 ^ v8;
    h[1] = v1 ^ v9;
    h[2] = v2 ^ v10;
    h[3] = v3 ^ v11;
    h[4] = v4 ^ v12;
    h[5] = v5 ^ v13;
    h[6] = v6 ^ v14;
    h[7] = v7 ^ v15;
}

// One compression of a 64-byte message / of a 32-byte message (words 8..15 zero), in the scheduled order.
__device__ __forceinline__ void blake3_block64(const uint32_t (&m)[16], uint32_t (&h)[8]) {
#ifdef ZIPK_B3_COMPILER_SCHED
    blake3_block(m, 64u, h);
#else
    blake3_sched_node(m, h);
#endif
}
__device__ __forceinline__ void blake3_block32(const uint32_t (&m8)[8], uint32_t (&h)[8]) {
#ifdef ZIPK_B3_COMPILER_SCHED
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = i < 8 ? m8[i] : 0u;
    blake3_block(m, 32u, h);
#else
    blake3_sched_half(m8, h);
#endif
}

// Parent node = BLAKE3(left.bytes || right.bytes), a plain 64-byte message
// (NOT BLAKE3's parent-node mode): src/zip/pcs/utils.rs:107-112.
__device__ __forceinline__ void blake3_node(const uint32_t (&l)[8], const uint32_t (&r)[8],
                                            uint32_t (&h)[8]) {
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        m[i] = l[i];
        m[8 + i] = r[i];
    }
    blake3_block64(m, h);
}

// Leaf of a codeword entry whose 256-bit two's-complement value is the sign
// extension of the 96-bit (d0, d1, d2).  Message = limbs in little-endian ORDER,
// each limb big-endian (src/field/int.rs:201-210):
//   m[2i] = bswap32(hi32(limb_i)), m[2i+1] = bswap32(lo32(limb_i)).
__device__ __forceinline__ void blake3_leaf_sext96(uint32_t d0, uint32_t d1, uint32_t d2,
                                                   uint32_t (&h)[8]) {
    const uint32_t s = (uint32_t)((int32_t)d2 >> 31);
    const uint32_t m[8] = {__builtin_bswap32(d1), __builtin_bswap32(d0), s, __builtin_bswap32(d2), s, s, s, s};
    blake3_block32(m, h);
}

// Generic leaf of LIMBS 64-bit limbs (1..8), used by the standalone Merkle entry
// point (benches/zip_benches.rs:80-98 hashes random full-width Int<4> leaves).
template <int LIMBS>
__device__ __forceinline__ void blake3_leaf_limbs(const uint64_t (&limb)[LIMBS], uint32_t (&h)[8]) {
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < LIMBS) {
            m[2 * i] = __builtin_bswap32((uint32_t)(limb[i] >> 32));
            m[2 * i + 1] = __builtin_bswap32((uint32_t)limb[i]);
        } else {
            m[2 * i] = 0u;
            m[2 * i + 1] = 0u;
        }
    }
    if constexpr (LIMBS == 4) {  // the Zip leaves (Int<4>); other widths only occur in the standalone zip_merkle_trees
        const uint32_t m8[8] = {m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]};
        blake3_block32(m8, h);
    } else if constexpr (LIMBS == 8) {
        blake3_block64(m, h);
    } else {
        blake3_block(m, 8u * LIMBS, h);
    }
}

}  // namespace zipk
