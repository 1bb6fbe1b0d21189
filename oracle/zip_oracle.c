/*
 * zip_oracle.c -- CPU restatement of the Zip PCS hot path of NethermindEth/zinc.
 * TEST INFRASTRUCTURE ONLY (see zip_oracle.h for the parity status of each part).
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference repository root).  Written for clarity, not speed; rows / columns
 * are spread over OpenMP threads the way the reference spreads them over Rayon.
 */
#include "zip_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

#define MAXW 16 /* widest integer handled (limbs) */

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================== */
/* BLAKE3, one block.  Public specification (blake3 crate 1.8.2 is the       */
/* reference's dependency; call sites src/zip/pcs/utils.rs:90,107-112).      */
/* ======================================================================== */
static const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                  0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_ROOT = 8 };

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static inline void b3_g(uint32_t *v, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    v[a] = v[a] + v[b] + mx;
    v[d] = rotr32(v[d] ^ v[a], 16);
    v[c] = v[c] + v[d];
    v[b] = rotr32(v[b] ^ v[c], 12);
    v[a] = v[a] + v[b] + my;
    v[d] = rotr32(v[d] ^ v[a], 8);
    v[c] = v[c] + v[d];
    v[b] = rotr32(v[b] ^ v[c], 7);
}

static void b3_compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter,
                        uint32_t block_len, uint32_t flags, uint32_t out[8]) {
    uint32_t v[16], m[16], t[16];
    for (int i = 0; i < 8; i++) v[i] = cv[i];
    for (int i = 0; i < 4; i++) v[8 + i] = B3_IV[i];
    v[12] = (uint32_t)counter;
    v[13] = (uint32_t)(counter >> 32);
    v[14] = block_len;
    v[15] = flags;
    memcpy(m, block, sizeof m);
    for (int r = 0; r < 7; r++) {
        b3_g(v, 0, 4, 8, 12, m[0], m[1]);
        b3_g(v, 1, 5, 9, 13, m[2], m[3]);
        b3_g(v, 2, 6, 10, 14, m[4], m[5]);
        b3_g(v, 3, 7, 11, 15, m[6], m[7]);
        b3_g(v, 0, 5, 10, 15, m[8], m[9]);
        b3_g(v, 1, 6, 11, 12, m[10], m[11]);
        b3_g(v, 2, 7, 8, 13, m[12], m[13]);
        b3_g(v, 3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]];
        memcpy(m, t, sizeof m);
    }
    for (int i = 0; i < 8; i++) out[i] = v[i] ^ v[i + 8];
}

int orc_blake3_hash_block(const uint8_t *msg, size_t len, uint8_t out[32]) {
    if (len > 64) return ORC_ERR_PARAM;
    uint8_t blk[64];
    uint32_t m[16], h[8];
    memset(blk, 0, sizeof blk);
    if (len) memcpy(blk, msg, len);
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)blk[4 * i] | ((uint32_t)blk[4 * i + 1] << 8) |
               ((uint32_t)blk[4 * i + 2] << 16) | ((uint32_t)blk[4 * i + 3] << 24);
    b3_compress(B3_IV, m, 0, (uint32_t)len, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT, h);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)h[i];
        out[4 * i + 1] = (uint8_t)(h[i] >> 8);
        out[4 * i + 2] = (uint8_t)(h[i] >> 16);
        out[4 * i + 3] = (uint8_t)(h[i] >> 24);
    }
    return ORC_OK;
}

/* ======================================================================== */
/* Keccak-256 (sha3 crate 0.10.8 `Keccak256`; src/transcript.rs:2,17)        */
/* ======================================================================== */
static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14,
                                   27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KECCAK_PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4,
                                   15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};

static inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

static void keccak_f(uint64_t st[25]) {
    uint64_t bc[5], t;
    for (int round = 0; round < 24; round++) {
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = KECCAK_PIL[i];
            uint64_t b = st[j];
            st[j] = rotl64(t, KECCAK_ROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= KECCAK_RC[round];
    }
}

void orc_keccak_init(orc_keccak *k) { memset(k, 0, sizeof *k); }

static void keccak_absorb_block(orc_keccak *k, const uint8_t *blk) {
    for (int i = 0; i < 17; i++) {
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) w |= (uint64_t)blk[8 * i + b] << (8 * b);
        k->st[i] ^= w;
    }
    keccak_f(k->st);
}

void orc_keccak_update(orc_keccak *k, const uint8_t *data, size_t len) {
    while (len) {
        size_t take = 136 - k->buflen;
        if (take > len) take = len;
        memcpy(k->buf + k->buflen, data, take);
        k->buflen += (uint32_t)take;
        data += take;
        len -= take;
        if (k->buflen == 136) {
            keccak_absorb_block(k, k->buf);
            k->buflen = 0;
        }
    }
}

void orc_keccak_finalize_copy(const orc_keccak *k, uint8_t domain, uint8_t out[32]) {
    orc_keccak c = *k;
    memset(c.buf + c.buflen, 0, 136 - c.buflen);
    c.buf[c.buflen] ^= domain;
    c.buf[135] ^= 0x80;
    keccak_absorb_block(&c, c.buf);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(c.st[i] >> (8 * b));
}

/* src/transcript.rs:40-55 -- counter-mode expansion on a CLONE of the hasher. */
void orc_tr_get_random_bytes(orc_keccak *k, size_t length, uint8_t *out) {
    size_t have = 0;
    int32_t counter = 0;
    while (have < length) {
        orc_keccak tmp = *k;
        uint8_t cb[4] = {(uint8_t)(counter >> 24), (uint8_t)(counter >> 16),
                         (uint8_t)(counter >> 8), (uint8_t)counter};
        uint8_t h[32];
        orc_keccak_update(&tmp, cb, 4);
        orc_keccak_finalize_copy(&tmp, 0x01, h);
        size_t take = length - have < 32 ? length - have : 32;
        memcpy(out + have, h, take);
        have += take;
        counter++;
    }
}

/* src/transcript.rs:142-155 */
void orc_tr_get_integer_challenge(orc_keccak *k, uint32_t n_limbs, uint64_t *out) {
    for (uint32_t i = 0; i < n_limbs; i++) {
        uint8_t ch[8], tag;
        orc_tr_get_random_bytes(k, 8, ch);
        tag = 0x12;
        orc_keccak_update(k, &tag, 1);
        orc_keccak_update(k, ch, 8);
        tag = 0x34;
        orc_keccak_update(k, &tag, 1);
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) w |= (uint64_t)ch[b] << (8 * b);
        out[i] = w;
    }
}

/* src/transcript.rs:183-185 */
uint64_t orc_tr_get_u64(orc_keccak *k) {
    uint64_t w;
    orc_tr_get_integer_challenge(k, 1, &w);
    return w;
}

/* ======================================================================== */
/* Unsigned / signed multi-limb helpers (crypto-bigint 0.6.1 semantics:      */
/* two's complement, checked add/mul panic on overflow; src/field/int.rs)    */
/* ======================================================================== */
static int ul_cmp(const uint64_t *a, const uint64_t *b, uint32_t n) {
    for (uint32_t i = n; i-- > 0;) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return 0;
}
static uint64_t ul_add(uint64_t *a, const uint64_t *b, uint32_t n) { /* a += b, returns carry */
    u128 c = 0;
    for (uint32_t i = 0; i < n; i++) {
        c += (u128)a[i] + b[i];
        a[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
static uint64_t ul_sub(uint64_t *a, const uint64_t *b, uint32_t n) { /* a -= b, returns borrow */
    uint64_t borrow = 0;
    for (uint32_t i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static int ul_is_zero(const uint64_t *a, uint32_t n) {
    for (uint32_t i = 0; i < n; i++)
        if (a[i]) return 0;
    return 1;
}
static void ul_neg(uint64_t *a, uint32_t n) { /* two's complement negate */
    uint64_t c = 1;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t t = ~a[i] + c;
        c = (c && t == 0) ? 1 : 0;
        a[i] = t;
    }
}
static uint32_t ul_bits(const uint64_t *a, uint32_t n) {
    for (uint32_t i = n; i-- > 0;) {
        if (a[i]) return 64 * i + (64 - (uint32_t)__builtin_clzll(a[i]));
    }
    return 0;
}
/* rem = a mod m (unsigned, n limbs, m != 0); binary long division */
static void ul_mod(const uint64_t *a, const uint64_t *m, uint32_t n, uint64_t *rem) {
    if (ul_cmp(a, m, n) < 0) {
        memcpy(rem, a, 8 * n);
        return;
    }
    uint64_t r[MAXW + 1];
    memset(r, 0, sizeof r);
    uint64_t mm[MAXW + 1];
    memcpy(mm, m, 8 * n);
    mm[n] = 0;
    uint32_t bits = ul_bits(a, n);
    for (uint32_t b = bits; b-- > 0;) {
        /* r = (r << 1) | bit b of a */
        for (uint32_t i = n + 1; i-- > 1;) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
        r[0] = (r[0] << 1) | ((a[b / 64] >> (b % 64)) & 1);
        if (ul_cmp(r, mm, n + 1) >= 0) ul_sub(r, mm, n + 1);
    }
    memcpy(rem, r, 8 * n);
}

static inline int wi_neg_p(const uint64_t *a, uint32_t n) { return (int)(a[n - 1] >> 63); }

/* From<&Int<M>> for Int<N> (crypto-bigint resize = sign extension), int.rs:194-199 */
static void wi_sext(uint64_t *out, uint32_t no, const uint64_t *in, uint32_t ni) {
    uint64_t fill = wi_neg_p(in, ni) ? ~0ULL : 0ULL;
    for (uint32_t i = 0; i < no; i++) out[i] = i < ni ? in[i] : fill;
}
/* a += b, signed; returns 1 on signed overflow (reference would panic, int.rs:122-134) */
static int wi_add_checked(uint64_t *a, const uint64_t *b, uint32_t n) {
    int sa = wi_neg_p(a, n), sb = wi_neg_p(b, n);
    ul_add(a, b, n);
    return (sa == sb) && (wi_neg_p(a, n) != sa);
}
/* out = a * b, signed n-limb, returns 1 if the product does not fit (int.rs:73-80) */
static int wi_mul_checked(uint64_t *out, const uint64_t *a, const uint64_t *b, uint32_t n) {
    uint64_t ma[MAXW], mb[MAXW], prod[2 * MAXW];
    int sa = wi_neg_p(a, n), sb = wi_neg_p(b, n);
    memcpy(ma, a, 8 * n);
    memcpy(mb, b, 8 * n);
    if (sa) ul_neg(ma, n);
    if (sb) ul_neg(mb, n);
    memset(prod, 0, 16 * n);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t carry = 0;
        if (!ma[i]) continue;
        for (uint32_t j = 0; j < n; j++) {
            u128 t = (u128)ma[i] * mb[j] + prod[i + j] + carry;
            prod[i + j] = (uint64_t)t;
            carry = (uint64_t)(t >> 64);
        }
        prod[i + n] = carry;
    }
    int ovf = !ul_is_zero(prod + n, n);
    int neg = sa != sb;
    if (prod[n - 1] >> 63) {
        /* magnitude >= 2^(64n-1): only representable as exactly MIN when negative */
        int is_min = neg && (prod[n - 1] == (1ULL << 63));
        for (uint32_t i = 0; i + 1 < n && is_min; i++)
            if (prod[i]) is_min = 0;
        if (!is_min) ovf = 1;
    }
    memcpy(out, prod, 8 * n);
    if (neg) ul_neg(out, n);
    return ovf;
}

/* ======================================================================== */
/* Montgomery field (src/field/config.rs, src/field/biginteger.rs)           */
/* ======================================================================== */
static inline uint64_t mac_with_carry(uint64_t a, uint64_t b, uint64_t c, uint64_t *carry) {
    u128 t = (u128)a + (u128)b * c + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

/* config.rs:68-76 */
static void field_reduce(const orc_field *f, uint64_t *a, int carry) {
    if (f->has_spare_bit) {
        if (ul_cmp(a, f->modulus, f->fl) >= 0) ul_sub(a, f->modulus, f->fl);
    } else if (carry || ul_cmp(a, f->modulus, f->fl) >= 0) {
        ul_sub(a, f->modulus, f->fl);
    }
}

/* config.rs:163-170 = mul_naive (biginteger.rs:448-464) + montgomery_reduction (:532-560) */
void orc_field_mul(const orc_field *f, uint64_t *a, const uint64_t *b) {
    const uint32_t N = f->fl;
    uint64_t lo[ORC_MAX_FL] = {0}, hi[ORC_MAX_FL] = {0};
    for (uint32_t i = 0; i < N; i++) {
        uint64_t carry = 0;
        for (uint32_t j = 0; j < N; j++) {
            uint32_t k = i + j;
            if (k >= N)
                hi[k - N] = mac_with_carry(hi[k - N], a[i], b[j], &carry);
            else
                lo[k] = mac_with_carry(lo[k], a[i], b[j], &carry);
        }
        hi[i] = carry;
    }
    uint64_t carry2 = 0;
    for (uint32_t i = 0; i < N; i++) {
        uint64_t tmp = lo[i] * f->inv;
        uint64_t carry = 0;
        (void)mac_with_carry(lo[i], tmp, f->modulus[0], &carry); /* mac!: low word discarded */
        for (uint32_t j = 1; j < N; j++) {
            uint32_t k = i + j;
            if (k >= N)
                hi[k - N] = mac_with_carry(hi[k - N], tmp, f->modulus[j], &carry);
            else
                lo[k] = mac_with_carry(lo[k], tmp, f->modulus[j], &carry);
        }
        u128 t = (u128)hi[i] + carry + carry2; /* adc! */
        hi[i] = (uint64_t)t;
        carry2 = (uint64_t)(t >> 64);
    }
    memcpy(a, hi, 8 * N);
    field_reduce(f, a, carry2 != 0);
}

/* config.rs:53-58 */
void orc_field_add(const orc_field *f, uint64_t *a, const uint64_t *b) {
    uint64_t c = ul_add(a, b, f->fl);
    field_reduce(f, a, c != 0);
}
/* config.rs:60-66 */
void orc_field_sub(const orc_field *f, uint64_t *a, const uint64_t *b) {
    if (ul_cmp(b, a, f->fl) > 0) ul_add(a, f->modulus, f->fl);
    ul_sub(a, b, f->fl);
}
/* arithmetic.rs:130-149 */
void orc_field_neg(const orc_field *f, uint64_t *a) {
    if (ul_is_zero(a, f->fl)) return;
    uint64_t t[ORC_MAX_FL];
    memcpy(t, a, 8 * f->fl);
    memcpy(a, f->modulus, 8 * f->fl);
    ul_sub(a, t, f->fl);
}

/* x = 2x mod q for x in [0,q), handling moduli without a spare bit */
static void field_dbl_plain(const orc_field *f, uint64_t *x) {
    uint64_t top = x[f->fl - 1] >> 63;
    for (uint32_t i = f->fl; i-- > 1;) x[i] = (x[i] << 1) | (x[i - 1] >> 63);
    x[0] <<= 1;
    if (top || ul_cmp(x, f->modulus, f->fl) >= 0) ul_sub(x, f->modulus, f->fl);
}

/* config.rs:174-186 (+ :196-214 for inv) */
int orc_field_new(orc_field *f, uint32_t fl, const uint64_t *modulus) {
    if (fl == 0 || fl > ORC_MAX_FL || !(modulus[0] & 1)) return ORC_ERR_PARAM;
    memset(f, 0, sizeof *f);
    f->fl = fl;
    memcpy(f->modulus, modulus, 8 * fl);
    f->has_spare_bit = (modulus[fl - 1] >> 63) == 0;
    uint64_t inv = 1;
    for (int i = 0; i < 63; i++) {
        inv *= inv;
        inv *= modulus[0];
    }
    f->inv = (uint64_t)0 - inv;
    /* R = 2^(64 fl) mod q, R2 = 2^(128 fl) mod q by repeated doubling of 1 */
    uint64_t x[ORC_MAX_FL] = {0};
    x[0] = 1;
    if (ul_cmp(x, f->modulus, fl) >= 0) ul_sub(x, f->modulus, fl); /* q == 1 */
    for (uint32_t i = 0; i < 64 * fl; i++) field_dbl_plain(f, x);
    memcpy(f->r, x, 8 * fl);
    for (uint32_t i = 0; i < 64 * fl; i++) field_dbl_plain(f, x);
    memcpy(f->r2, x, 8 * fl);
    return ORC_OK;
}

/* Shared tail of every FieldMap impl (conversion.rs:9-46, field.rs:536-568):
 * `words` is an Int<W> (signed!) with W = max(source limbs, fl); the modulus is
 * ALSO read as a signed Int<W>, so a modulus with its top bit set acts as
 * 2^(64W) - q inside `%=` (crypto-bigint Int::rem: |lhs| mod |rhs|, sign of lhs),
 * and `F::B::from(Int)` takes the absolute value (biginteger.rs:805-816).  Then
 * the value is multiplied by R^2 (Montgomery form). */
static void field_from_signed_words(const orc_field *f, const uint64_t *words, uint32_t W,
                                    uint64_t *out) {
    uint64_t val[MAXW] = {0}, mod[MAXW] = {0}, rem[MAXW] = {0};
    memcpy(val, words, 8 * W);
    if (wi_neg_p(val, W)) ul_neg(val, W);
    for (uint32_t i = 0; i < W; i++) mod[i] = i < f->fl ? f->modulus[i] : 0;
    if (wi_neg_p(mod, W)) ul_neg(mod, W);
    ul_mod(val, mod, W, rem);
    for (uint32_t i = 0; i < f->fl; i++) out[i] = rem[i];
    orc_field_mul(f, out, f->r2);
}

/* FieldMap for T: Integer (conversion.rs:86-100) over BigInt<M> (field.rs:536-568) */
void orc_field_from_int(const orc_field *f, const uint64_t *v, uint32_t n, uint64_t *out) {
    uint32_t W = n > f->fl ? n : f->fl;
    uint64_t mag[MAXW] = {0}, words[MAXW] = {0};
    int neg = wi_neg_p(v, n);
    memcpy(mag, v, 8 * n);
    if (neg) ul_neg(mag, n); /* Integer::abs -> Uint */
    for (uint32_t i = 0; i < W; i++) words[i] = i < n ? mag[i] : 0;
    field_from_signed_words(f, words, W, out);
    if (neg) orc_field_neg(f, out);
}

void orc_field_from_i64(const orc_field *f, int64_t v, uint64_t *out) {
    uint64_t w = (uint64_t)v;
    orc_field_from_int(f, &w, 1, out);
}

/* impl_field_map_for_int!(u128 / u64), conversion.rs:9-46 */
void orc_field_from_u128(const orc_field *f, uint64_t lo, uint64_t hi, uint64_t *out) {
    uint64_t words[MAXW] = {0};
    words[0] = lo;
    if (f->fl > 1) words[1] = hi;
    field_from_signed_words(f, words, f->fl, out);
}

/* transcript.rs:72-86 */
static void tr_get_challenge_limbs(orc_keccak *k, uint64_t lo[2], uint64_t hi[2]) {
    uint8_t ch[32], tag;
    orc_keccak_finalize_copy(k, 0x01, ch);
    uint64_t w[4];
    for (int i = 0; i < 4; i++) {
        w[i] = 0;
        for (int b = 0; b < 8; b++) w[i] = (w[i] << 8) | ch[8 * i + b];
    }
    /* lo = u128::from_be_bytes(ch[0..16]) -> low 64 bits are ch[8..16] */
    lo[0] = w[1];
    lo[1] = w[0];
    hi[0] = w[3];
    hi[1] = w[2];
    tag = 0x00;
    orc_keccak_update(k, &tag, 1);
    orc_keccak_update(k, ch, 32);
    tag = 0x01;
    orc_keccak_update(k, &tag, 1);
}

/* transcript.rs:88-133 */
void orc_tr_get_challenge(orc_keccak *k, const orc_field *f, uint64_t *out) {
    uint64_t lo[2], hi[2];
    tr_get_challenge_limbs(k, lo, hi);
    uint32_t cbits = ul_bits(f->modulus, f->fl) - 1;
    if (f->fl == 1) {
        uint64_t mask = (1ULL << cbits) - 1;
        orc_field_from_u128(f, lo[0] & mask, 0, out); /* u64 map_to_field */
        return;
    }
    if (cbits < 128) {
        uint64_t l0 = lo[0], l1 = lo[1];
        if (cbits < 64) {
            l0 &= (1ULL << cbits) - 1;
            l1 = 0;
        } else if (cbits > 64) {
            l1 &= (1ULL << (cbits - 64)) - 1;
        } else {
            l1 = 0;
        }
        orc_field_from_u128(f, l0, l1, out);
        return;
    }
    uint64_t h0 = hi[0], h1 = hi[1];
    if (cbits < 256) {
        uint32_t keep = cbits - 128;
        if (keep < 64) {
            h0 &= (1ULL << keep) - 1;
            h1 = 0;
        } else if (keep > 64) {
            h1 &= (1ULL << (keep - 64)) - 1;
        } else {
            h1 = 0;
        }
    }
    /* two_to_128 = BigInt::from_bits_le(196 bits, bit 128 set).map_to_field */
    uint64_t words[MAXW] = {0}, two128[ORC_MAX_FL], a[ORC_MAX_FL], b[ORC_MAX_FL];
    if (f->fl > 2) words[2] = 1;
    field_from_signed_words(f, words, f->fl, two128);
    orc_field_from_u128(f, lo[0], lo[1], a);
    orc_field_from_u128(f, h0, h1, b);
    orc_field_mul(f, two128, b);
    orc_field_add(f, a, two128);
    memcpy(out, a, 8 * f->fl);
}

static void limbs_to_be(const uint64_t *v, uint32_t n, uint8_t *out) {
    for (uint32_t i = 0; i < n; i++)
        for (int b = 0; b < 8; b++) out[8 * (n - 1 - i) + (7 - b)] = (uint8_t)(v[i] >> (8 * b));
}

/* field.rs:360-378 (Initialized branch) */
void orc_tr_absorb_field(orc_keccak *k, const orc_field *f, const uint64_t *val) {
    uint8_t buf[8 * ORC_MAX_FL], tag;
    tag = 0x3;
    orc_keccak_update(k, &tag, 1);
    limbs_to_be(f->modulus, f->fl, buf);
    orc_keccak_update(k, buf, 8 * f->fl);
    tag = 0x5;
    orc_keccak_update(k, &tag, 1);
    tag = 0x1;
    orc_keccak_update(k, &tag, 1);
    limbs_to_be(val, f->fl, buf);
    orc_keccak_update(k, buf, 8 * f->fl);
    tag = 0x3;
    orc_keccak_update(k, &tag, 1);
}

/* sumcheck/utils.rs:117-177.  r[0] is bound to the least significant index bit. */
int orc_build_eq_x_r(const orc_field *f, const uint64_t *r, uint32_t nvars, uint64_t *out) {
    const uint32_t fl = f->fl;
    if (nvars == 0) return ORC_ERR_PARAM;
    uint64_t *buf = (uint64_t *)malloc((size_t)8 * fl << nvars);
    if (!buf) return ORC_ERR_PARAM;
    /* innermost call: r.len()==1 on the LAST coordinate: [1 - r, r] */
    uint64_t one[ORC_MAX_FL];
    memcpy(one, f->r, 8 * fl);
    const uint64_t *rl = r + (size_t)fl * (nvars - 1);
    memcpy(out, one, 8 * fl);
    orc_field_sub(f, out, rl);
    memcpy(out + fl, rl, 8 * fl);
    size_t len = 2;
    for (uint32_t t = nvars - 1; t-- > 0;) {
        const uint64_t *rt = r + (size_t)fl * t;
        memcpy(buf, out, 8 * fl * len);
        for (size_t i = 0; i < 2 * len; i++) {
            uint64_t bi[ORC_MAX_FL], tmp[ORC_MAX_FL];
            memcpy(bi, buf + fl * (i >> 1), 8 * fl);
            memcpy(tmp, rt, 8 * fl);
            orc_field_mul(f, tmp, bi); /* r[0] * b_i */
            if ((i & 1) == 0) {
                orc_field_sub(f, bi, tmp);
                memcpy(out + fl * i, bi, 8 * fl);
            } else {
                memcpy(out + fl * i, tmp, 8 * fl);
            }
        }
        len *= 2;
    }
    free(buf);
    return ORC_OK;
}

/* ======================================================================== */
/* rand 0.9.2 `StdRng::seed_from_u64` + `SliceRandom::shuffle`               */
/* Restatement of the published algorithm of the un-vendored crates          */
/* (zip/utils.rs:139-142 is the only call site).  Pinned piece by piece by   */
/* published vectors (tests/golden/rand_vectors.json, tests/test_oracle_kats.py): */
/*   ChaCha12 block           draft-strombergson-chacha-test-vectors TC1     */
/*   StdRng = ChaCha12, word / counter layout, next_u64                      */
/*                            rand's own test_stdrng_construction            */
/*   PCG32 step + XSH-RR      O'Neill's pcg32 demo (seed 42, stream 54)      */
/*   IncreasingUniform shuffle + Canon's-method random_range                 */
/*                            rand's value_stability_slice (Pcg32(414))      */
/*   rand_core seed_from_u64 (advance-then-output PCG32 with the fixed       */
/*   increment, 4 bytes little-endian per step)                              */
/*                            rand_pcg's test_lcg64xsh32_construction:       */
/*                            Lcg64Xsh32::seed_from_u64(0).next_u64()        */
/* ======================================================================== */
typedef struct {
    uint32_t key[8];
    uint64_t counter;
    uint32_t buf[16];
    int idx;
} chacha12_rng;

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);

static void chacha12_block(chacha12_rng *g) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865; s[1] = 0x3320646e; s[2] = 0x79622d32; s[3] = 0x6b206574;
    for (int i = 0; i < 8; i++) s[4 + i] = g->key[i];
    s[12] = (uint32_t)g->counter;
    s[13] = (uint32_t)(g->counter >> 32);
    s[14] = 0;
    s[15] = 0;
    memcpy(x, s, sizeof x);
    for (int r = 0; r < 6; r++) {
        CHACHA_QR(x[0], x[4], x[8], x[12]);
        CHACHA_QR(x[1], x[5], x[9], x[13]);
        CHACHA_QR(x[2], x[6], x[10], x[14]);
        CHACHA_QR(x[3], x[7], x[11], x[15]);
        CHACHA_QR(x[0], x[5], x[10], x[15]);
        CHACHA_QR(x[1], x[6], x[11], x[12]);
        CHACHA_QR(x[2], x[7], x[8], x[13]);
        CHACHA_QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; i++) g->buf[i] = x[i] + s[i];
    g->counter++;
    g->idx = 0;
}

/* PCG XSH-RR 64/32 output function and LCG step (O'Neill); rand_core and rand_pcg both build on them */
static inline uint32_t pcg32_output(uint64_t state) {
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    return (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
}
static inline uint64_t pcg32_step(uint64_t state, uint64_t inc) { return state * 6364136223846793005ULL + inc; }

static void chacha12_seed_from_u64(chacha12_rng *g, uint64_t state) {
    /* rand_core SeedableRng::seed_from_u64: the state is advanced FIRST, then the PCG output function of the new
     * state gives 4 seed bytes (little-endian), eight times */
    for (int i = 0; i < 8; i++) {
        state = pcg32_step(state, 11634580027462260723ULL);
        g->key[i] = pcg32_output(state);
    }
    g->counter = 0;
    g->idx = 16;
}

static uint32_t chacha12_next_u32(void *p) {
    chacha12_rng *g = (chacha12_rng *)p;
    if (g->idx >= 16) chacha12_block(g);
    return g->buf[g->idx++];
}

/* rand_pcg::Lcg64Xsh32 (= Pcg32), the generator rand's own value-stability tests run on */
typedef struct { uint64_t state, inc; } pcg32_rng;
static void pcg32_new(pcg32_rng *g, uint64_t state, uint64_t stream) {
    g->inc = (stream << 1) | 1;
    g->state = pcg32_step(state + g->inc, g->inc);
}
static uint32_t pcg32_next_u32(void *p) {
    pcg32_rng *g = (pcg32_rng *)p;
    const uint64_t old = g->state;
    g->state = pcg32_step(old, g->inc);
    return pcg32_output(old);
}

typedef uint32_t (*u32_source)(void *);

/* UniformInt<u32>::sample_single_inclusive(0, bound-1): Canon's method, one retry */
static uint32_t rand_range_u32(u32_source next, void *g, uint32_t bound) {
    uint64_t m = (uint64_t)next(g) * bound;
    uint32_t result = (uint32_t)(m >> 32), lo_order = (uint32_t)m;
    if (lo_order > (uint32_t)(0u - bound)) {
        uint64_t m2 = (uint64_t)next(g) * bound;
        uint32_t new_hi = (uint32_t)(m2 >> 32);
        uint32_t sum = lo_order + new_hi;
        result += (sum < lo_order); /* checked_add(..).is_none() */
    }
    return result;
}

/* SliceRandom::shuffle of [0, len) (rand 0.9 seq/slice.rs + seq/increasing_uniform.rs) over any u32 source */
static void shuffle_perm(u32_source next, void *g, uint32_t len, uint32_t *perm) {
    for (uint32_t i = 0; i < len; i++) perm[i] = i;
    if (len <= 1) return;
    /* IncreasingUniform::new(rng, 0) */
    uint32_t n = 0, chunk = 0;
    uint8_t chunk_remaining = 1;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t next_n = n + 1, result;
        uint8_t next_rem;
        if (chunk_remaining > 0) {
            next_rem = (uint8_t)(chunk_remaining - 1);
        } else {
            /* calculate_bound_u32(next_n) */
            uint32_t product = next_n, current = next_n + 1;
            for (;;) {
                uint64_t p = (uint64_t)product * current;
                if (p >> 32) break;
                product = (uint32_t)p;
                current++;
            }
            uint8_t remaining = (uint8_t)(current - next_n);
            chunk = rand_range_u32(next, g, product);
            next_rem = (uint8_t)(remaining - 1);
        }
        if (next_rem == 0) {
            result = chunk;
        } else {
            result = chunk % next_n;
            chunk /= next_n;
        }
        chunk_remaining = next_rem;
        n = next_n;
        uint32_t t = perm[i];
        perm[i] = perm[result];
        perm[result] = t;
    }
}

void orc_shuffle_seeded_perm(uint64_t seed, uint32_t len, uint32_t *perm) {
    chacha12_rng g;
    chacha12_seed_from_u64(&g, seed);
    shuffle_perm(chacha12_next_u32, &g, len, perm);
}

/* ---- known-answer hooks for the pieces above (tests/test_oracle_kats.py) ---- */
void orc_kat_chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    chacha12_rng g;
    memcpy(g.key, key, sizeof g.key);
    g.counter = counter;
    chacha12_block(&g);
    memcpy(out, g.buf, sizeof g.buf);
}
/* StdRng::from_seed(seed).next_u64() ... : n consecutive u64 (two u32 words each, low word first) */
void orc_kat_stdrng_from_seed_u64(const uint8_t seed[32], uint32_t n, uint64_t *out) {
    chacha12_rng g;
    for (int i = 0; i < 8; i++)
        g.key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) |
                   ((uint32_t)seed[4 * i + 3] << 24);
    g.counter = 0;
    g.idx = 16;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t lo = chacha12_next_u32(&g), hi = chacha12_next_u32(&g);
        out[i] = lo | (hi << 32);
    }
}
void orc_kat_pcg32(uint64_t state, uint64_t stream, uint32_t n, uint32_t *out) {
    pcg32_rng g;
    pcg32_new(&g, state, stream);
    for (uint32_t i = 0; i < n; i++) out[i] = pcg32_next_u32(&g);
}
/* [0, len).shuffle(&mut Pcg32::new(state, stream)) */
void orc_kat_shuffle_pcg32(uint64_t state, uint64_t stream, uint32_t len, uint32_t *perm) {
    pcg32_rng g;
    pcg32_new(&g, state, stream);
    shuffle_perm(pcg32_next_u32, &g, len, perm);
}
/* the 32-byte seed seed_from_u64 expands `state` to (its first 16 bytes seed rand_pcg's construction vector) */
void orc_kat_seed_from_u64(uint64_t state, uint32_t key_out[8]) {
    chacha12_rng g;
    chacha12_seed_from_u64(&g, state);
    memcpy(key_out, g.key, sizeof g.key);
}

/* ======================================================================== */
/* RAA code (src/zip/code_raa.rs:89-171)                                     */
/* ======================================================================== */
int orc_raa_encode_row(const uint64_t *row, uint32_t in_limbs, uint32_t row_len, uint32_t rep,
                       const uint32_t *perm1, const uint32_t *perm2, uint64_t *out,
                       uint32_t L) {
    const size_t cw = (size_t)row_len * rep;
    int ovf = 0;
    uint64_t *a = (uint64_t *)malloc(8 * L * cw), *b = (uint64_t *)malloc(8 * L * cw);
    if (!a || !b) { free(a); free(b); return ORC_ERR_PARAM; }
    /* repeat (:142-152) with Out::from(&In) = sign extension */
    for (size_t j = 0; j < cw; j++) wi_sext(a + L * j, L, row + (size_t)in_limbs * (j % row_len), in_limbs);
    /* shuffle_seeded(perm_1_seed) expressed through its permutation table */
    for (size_t j = 0; j < cw; j++) memcpy(b + L * j, a + L * perm1[j], 8 * L);
    /* accumulate (:164-171) */
    for (size_t i = 1; i < cw; i++) ovf |= wi_add_checked(b + L * i, b + L * (i - 1), L);
    for (size_t j = 0; j < cw; j++) memcpy(a + L * j, b + L * perm2[j], 8 * L);
    for (size_t i = 1; i < cw; i++) ovf |= wi_add_checked(a + L * i, a + L * (i - 1), L);
    memcpy(out, a, 8 * L * cw);
    free(a);
    free(b);
    return ovf ? ORC_ERR_OVERFLOW : ORC_OK;
}

void orc_raa_encode_row_field(const orc_field *f, const uint64_t *row, uint32_t row_len,
                              uint32_t rep, const uint32_t *perm1, const uint32_t *perm2,
                              uint64_t *out) {
    const size_t cw = (size_t)row_len * rep;
    const uint32_t L = f->fl;
    uint64_t *a = (uint64_t *)malloc(8 * L * cw), *b = (uint64_t *)malloc(8 * L * cw);
    for (size_t j = 0; j < cw; j++) memcpy(a + L * j, row + (size_t)L * (j % row_len), 8 * L);
    for (size_t j = 0; j < cw; j++) memcpy(b + L * j, a + L * perm1[j], 8 * L);
    for (size_t i = 1; i < cw; i++) orc_field_add(f, b + L * i, b + L * (i - 1));
    for (size_t j = 0; j < cw; j++) memcpy(a + L * j, b + L * perm2[j], 8 * L);
    for (size_t i = 1; i < cw; i++) orc_field_add(f, a + L * i, a + L * (i - 1));
    memcpy(out, a, 8 * L * cw);
    free(a);
    free(b);
}

/* ======================================================================== */
/* Merkle tree (src/zip/pcs/utils.rs:67-210)                                 */
/* ======================================================================== */
/* ToBytes for Int<N> (int.rs:201-210): limbs in LE order, each limb big-endian */
static void leaf_bytes(const uint64_t *leaf, uint32_t limbs, uint8_t *out) {
    for (uint32_t i = 0; i < limbs; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(leaf[i] >> (8 * (7 - b)));
}

int orc_merkle_tree(uint32_t depth, const uint64_t *leaves, uint32_t leaf_limbs, uint8_t *layers) {
    if (leaf_limbs == 0 || leaf_limbs > 8) return ORC_ERR_PARAM;
    const size_t nleaves = (size_t)1 << depth;
    for (size_t i = 0; i < nleaves; i++) { /* compute_leaves_hashes :87-93 */
        uint8_t msg[64];
        leaf_bytes(leaves + (size_t)leaf_limbs * i, leaf_limbs, msg);
        orc_blake3_hash_block(msg, 8 * leaf_limbs, layers + 32 * i);
    }
    size_t offset = 0; /* merklize_leaves_hashes :95-118 */
    for (uint32_t d = depth; d >= 1; d--) {
        size_t width = (size_t)1 << d;
        const uint8_t *cur = layers + 32 * offset;
        uint8_t *next = layers + 32 * (offset + width);
        for (size_t i = 0; i < width / 2; i++) orc_blake3_hash_block(cur + 64 * i, 64, next + 32 * i);
        offset += width;
    }
    return ORC_OK;
}

void orc_merkle_path(uint32_t depth, const uint8_t *layers, uint32_t leaf, uint8_t *path) {
    size_t offset = 0; /* :163-176 */
    uint32_t n = 0;
    for (uint32_t d = depth; d >= 1; d--) {
        size_t width = (size_t)1 << d;
        size_t idx = (leaf >> (depth - d)) ^ 1;
        memcpy(path + 32 * n++, layers + 32 * (offset + idx), 32);
        offset += width;
    }
}

int orc_merkle_verify(uint32_t depth, const uint8_t *path, const uint8_t root[32],
                      const uint64_t *leaf, uint32_t leaf_limbs, uint32_t leaf_index) {
    uint8_t cur[32], msg[64]; /* :178-210 */
    leaf_bytes(leaf, leaf_limbs, msg);
    orc_blake3_hash_block(msg, 8 * leaf_limbs, cur);
    uint32_t index = leaf_index;
    for (uint32_t l = 0; l < depth; l++) {
        if ((index & 1) == 0) {
            memcpy(msg, cur, 32);
            memcpy(msg + 32, path + 32 * l, 32);
        } else {
            memcpy(msg, path + 32 * l, 32);
            memcpy(msg + 32, cur, 32);
        }
        orc_blake3_hash_block(msg, 64, cur);
        index /= 2;
    }
    return memcmp(cur, root, 32) == 0 ? ORC_OK : ORC_ERR_PROOF;
}

/* ======================================================================== */
/* params / commit (code_raa.rs:35-86, structs.rs:79-91, commit.rs:50-183)   */
/* ======================================================================== */
static uint32_t ilog2_u64(uint64_t x) { return 63 - (uint32_t)__builtin_clzll(x); }
static uint64_t isqrt_u64(uint64_t x) {
    uint64_t r = 0;
    for (int b = 31; b >= 0; b--) {
        uint64_t t = r | (1ULL << b);
        if (t * t <= x) r = t;
    }
    return r;
}
static uint64_t next_pow2_u64(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

int orc_params_init(orc_params *p, uint32_t num_vars, uint32_t n_limbs, uint32_t rep,
                    const uint32_t *perm1, const uint32_t *perm2) {
    memset(p, 0, sizeof *p);
    uint64_t poly_size = 1ULL << num_vars;
    p->num_vars = num_vars;
    p->row_len = (uint32_t)next_pow2_u64(isqrt_u64(poly_size));        /* code_raa.rs:43 */
    p->rep = rep;
    p->codeword_len = p->row_len * rep;                                  /* :113-115 */
    p->num_rows = (uint32_t)next_pow2_u64(poly_size / p->row_len);       /* structs.rs:82 */
    p->depth = ilog2_u64(next_pow2_u64(p->codeword_len));                /* commit.rs:67 */
    p->n_limbs = n_limbs;
    p->k_limbs = 4 * n_limbs;
    p->m_limbs = 8 * n_limbs;
    p->perm1 = perm1;
    p->perm2 = perm2;
    p->num_column_opening = 1000; /* code.rs:229-231 */
    p->num_proximity_testing = 1; /* code.rs:239-241 */
    /* width assertion code_raa.rs:53-72 */
    uint32_t rep_log = ilog2_u64(next_pow2_u64(rep));
    uint32_t nv_even = (num_vars % 2 == 0) ? num_vars : num_vars + 1;
    uint32_t width = 64 * n_limbs + nv_even + 2 * rep_log;
    if (64 * p->k_limbs < width) return ORC_ERR_PARAM;
    if (p->k_limbs > MAXW || p->m_limbs > MAXW) return ORC_ERR_PARAM;
    return ORC_OK;
}

int orc_commit(const orc_params *p, const uint64_t *evals, uint64_t *rows, uint8_t *layers,
               uint8_t *roots) {
    const size_t cw = p->codeword_len, K = p->k_limbs;
    const size_t tree_hashes = ((size_t)2 << p->depth) - 1;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
    for (uint32_t r = 0; r < p->num_rows; r++) {
        uint64_t *row_out = rows + (size_t)r * cw * K;
        int e = orc_raa_encode_row(evals + (size_t)r * p->row_len * p->n_limbs, p->n_limbs,
                                   p->row_len, p->rep, p->perm1, p->perm2, row_out, (uint32_t)K);
        if (e) err |= 1;
        if (layers) {
            uint8_t *tl = layers + (size_t)r * tree_hashes * 32;
            orc_merkle_tree(p->depth, row_out, (uint32_t)K, tl);
            if (roots) memcpy(roots + 32 * (size_t)r, tl + 32 * (tree_hashes - 1), 32);
        }
    }
    return err ? ORC_ERR_OVERFLOW : ORC_OK;
}

/* commit.rs:50-87 row by row, keeping only the roots and the opening blocks of the picked columns
 * (open_z.rs:124-143, pcs/utils.rs:163-176,220-233, pcs_transcript.rs:115-123,198-211) */
int orc_commit_open_columns(const orc_params *p, const uint64_t *evals, const uint32_t *cols, uint32_t n_cols,
                            uint8_t *roots, uint8_t *blocks) {
    const size_t cw = p->codeword_len, K = p->k_limbs, R = p->num_rows, d = p->depth;
    const size_t tree_hashes = ((size_t)2 << d) - 1;
    const size_t rec = 8 + 32 * d, col_bytes = R * (8 * K + rec);
    int err = 0;
    for (uint32_t i = 0; i < n_cols; i++)
        if (cols[i] >= cw) return ORC_ERR_PARAM;
#pragma omp parallel reduction(| : err)
    {
        uint64_t *row_out = (uint64_t *)malloc(cw * K * 8);
        uint8_t *tl = (uint8_t *)malloc(tree_hashes * 32);
        if (!row_out || !tl) err |= 2;
#pragma omp for schedule(dynamic, 1)
        for (uint32_t r = 0; r < R; r++) {
            if (!row_out || !tl) continue;
            if (orc_raa_encode_row(evals + (size_t)r * p->row_len * p->n_limbs, p->n_limbs, p->row_len, p->rep,
                                   p->perm1, p->perm2, row_out, (uint32_t)K))
                err |= 1;
            orc_merkle_tree(p->depth, row_out, (uint32_t)K, tl);
            if (roots) memcpy(roots + 32 * (size_t)r, tl + 32 * (tree_hashes - 1), 32);
            for (uint32_t i = 0; i < n_cols; i++) {
                uint8_t *blk = blocks + (size_t)i * col_bytes;
                uint8_t *val = blk + (size_t)r * 8 * K; /* write_integer: limbs little-endian, pcs_transcript.rs:115-123 */
                for (size_t l = 0; l < K; l++) {
                    const uint64_t w = row_out[(size_t)cols[i] * K + l];
                    for (int b = 0; b < 8; b++) val[8 * l + b] = (uint8_t)(w >> (8 * b));
                }
                uint8_t *rp = blk + R * 8 * K + (size_t)r * rec;
                for (int k = 0; k < 8; k++) rp[k] = (uint8_t)((uint64_t)d >> (8 * (7 - k)));
                orc_merkle_path(p->depth, tl, cols[i], rp + 8);
            }
        }
        free(row_out);
        free(tl);
    }
    if (err & 2) return ORC_ERR_ALLOC;
    return err ? ORC_ERR_OVERFLOW : ORC_OK;
}

/* zip/utils.rs:94-127 with F = Int<M>; coeffs/evals expanded first (open_z.rs:104-110) */
int orc_combine_rows_int(const uint64_t *coeffs, uint32_t coeff_limbs, const uint64_t *evals,
                         uint32_t eval_limbs, uint32_t num_rows, uint32_t row_len,
                         uint64_t *out, uint32_t M) {
    int err = 0;
#pragma omp parallel for schedule(static) reduction(| : err)
    for (uint32_t c = 0; c < row_len; c++) {
        uint64_t acc[MAXW] = {0}, a[MAXW], b[MAXW], prod[MAXW];
        for (uint32_t r = 0; r < num_rows; r++) {
            wi_sext(a, M, coeffs + (size_t)coeff_limbs * r, coeff_limbs);
            wi_sext(b, M, evals + (size_t)eval_limbs * ((size_t)r * row_len + c), eval_limbs);
            err |= wi_mul_checked(prod, a, b, M);
            err |= wi_add_checked(acc, prod, M);
        }
        memcpy(out + (size_t)M * c, acc, 8 * M);
    }
    return err ? ORC_ERR_OVERFLOW : ORC_OK;
}

/* open_z.rs:76-90: evaluations.map_to_field, then combine_rows over F */
void orc_combine_rows_field(const orc_field *f, const uint64_t *q0, const uint64_t *evals,
                            uint32_t eval_limbs, uint32_t num_rows, uint32_t row_len,
                            uint64_t *out) {
    const uint32_t fl = f->fl;
#pragma omp parallel for schedule(static)
    for (uint32_t c = 0; c < row_len; c++) {
        uint64_t acc[ORC_MAX_FL] = {0}, e[ORC_MAX_FL], t[ORC_MAX_FL];
        for (uint32_t r = 0; r < num_rows; r++) {
            orc_field_from_int(f, evals + (size_t)eval_limbs * ((size_t)r * row_len + c), eval_limbs, e);
            memcpy(t, q0 + (size_t)fl * r, 8 * fl);
            orc_field_mul(f, t, e); /* coeff * &eval */
            orc_field_add(f, acc, t);
        }
        memcpy(out + (size_t)fl * c, acc, 8 * fl);
    }
}

size_t orc_proof_len(const orc_params *p, uint32_t fl) {
    size_t len = 0; /* commit.rs:712-737 */
    if (p->num_rows > 1) len += (size_t)p->num_proximity_testing * p->row_len * 8 * p->m_limbs;
    len += (size_t)p->num_column_opening * p->num_rows * (8 * p->k_limbs + 8 + 32 * (size_t)p->depth);
    len += (size_t)p->row_len * 8 * fl;
    return len;
}

/* ------------------------------------------------------------- proof stream */
typedef struct {
    uint8_t *buf;
    size_t cap, pos;
    int err;
} wstream;
static void ws_write(wstream *s, const void *src, size_t n) {
    if (s->pos + n > s->cap) { s->err = 1; return; }
    memcpy(s->buf + s->pos, src, n);
    s->pos += n;
}
/* pcs_transcript.rs:115-123: limbs little-endian */
static void ws_write_integer(wstream *s, const uint64_t *v, uint32_t limbs) {
    for (uint32_t i = 0; i < limbs; i++) {
        uint8_t b[8];
        for (int k = 0; k < 8; k++) b[k] = (uint8_t)(v[i] >> (8 * k));
        ws_write(s, b, 8);
    }
}

/* pcs_transcript.rs:174-179 */
static uint32_t squeeze_challenge_idx(orc_keccak *fs, const orc_field *f, uint32_t cap) {
    uint64_t ch[ORC_MAX_FL];
    orc_tr_get_challenge(fs, f, ch);
    uint32_t num = (uint32_t)ch[0]; /* first 4 LE bytes of the Montgomery value */
    return num % cap;
}

int orc_open(const orc_params *p, const orc_field *f, const uint64_t *evals, const uint64_t *rows,
             const uint8_t *layers, const uint64_t *point, orc_keccak *fs, uint8_t *proof,
             size_t proof_cap, size_t *proof_len, uint32_t *cols_out, uint64_t *coeffs_out) {
    wstream ws = {proof, proof_cap, 0, 0};
    const uint32_t R = p->num_rows, C = p->row_len, cw = p->codeword_len, fl = f->fl;
    const size_t tree_hashes = ((size_t)2 << p->depth) - 1;
    int rc = ORC_OK;

    /* prove_testing_phase, open_z.rs:93-122 */
    if (R > 1) {
        for (uint32_t t = 0; t < p->num_proximity_testing; t++) {
            uint64_t *coeffs = (uint64_t *)malloc((size_t)8 * p->n_limbs * R);
            uint64_t *comb = (uint64_t *)malloc((size_t)8 * p->m_limbs * C);
            for (uint32_t r = 0; r < R; r++)
                orc_tr_get_integer_challenge(fs, p->n_limbs, coeffs + (size_t)p->n_limbs * r);
            if (coeffs_out && t == 0) memcpy(coeffs_out, coeffs, (size_t)8 * p->n_limbs * R);
            if (orc_combine_rows_int(coeffs, p->n_limbs, evals, p->n_limbs, R, C, comb, p->m_limbs))
                rc = ORC_ERR_OVERFLOW;
            for (uint32_t c = 0; c < C; c++) ws_write_integer(&ws, comb + (size_t)p->m_limbs * c, p->m_limbs);
            free(coeffs);
            free(comb);
        }
    }
    for (uint32_t i = 0; i < p->num_column_opening; i++) {
        uint32_t col = squeeze_challenge_idx(fs, f, cw);
        if (cols_out) cols_out[i] = col;
        /* open_merkle_trees_for_column, open_z.rs:124-143 */
        for (uint32_t r = 0; r < R; r++)
            ws_write_integer(&ws, rows + ((size_t)r * cw + col) * p->k_limbs, p->k_limbs);
        for (uint32_t r = 0; r < R; r++) { /* pcs/utils.rs:220-233 + pcs_transcript.rs:198-211 */
            uint8_t path[32 * 40], lenb[8];
            uint64_t d = p->depth;
            for (int k = 0; k < 8; k++) lenb[k] = (uint8_t)(d >> (8 * (7 - k)));
            orc_merkle_path(p->depth, layers + (size_t)r * tree_hashes * 32, col, path);
            ws_write(&ws, lenb, 8);
            ws_write(&ws, path, 32 * (size_t)p->depth);
        }
    }

    /* prove_evaluation_phase, open_z.rs:62-91 */
    uint64_t *row = (uint64_t *)malloc((size_t)8 * fl * C);
    if (R > 1) {
        uint32_t lr = ilog2_u64(R);
        uint64_t *q0 = (uint64_t *)malloc((size_t)8 * fl * R);
        orc_build_eq_x_r(f, point + (size_t)fl * (p->num_vars - lr), lr, q0); /* pcs/utils.rs:279-292 */
        orc_combine_rows_field(f, q0, evals, p->n_limbs, R, C, row);
        free(q0);
    } else {
        for (uint32_t c = 0; c < C; c++)
            orc_field_from_int(f, evals + (size_t)p->n_limbs * c, p->n_limbs, row + (size_t)fl * c);
    }
    for (uint32_t c = 0; c < C; c++) { /* write_field_element, pcs_transcript.rs:107-113 */
        uint8_t be[8 * ORC_MAX_FL];
        orc_tr_absorb_field(fs, f, row + (size_t)fl * c);
        limbs_to_be(row + (size_t)fl * c, fl, be);
        ws_write(&ws, be, 8 * fl);
    }
    free(row);
    if (proof_len) *proof_len = ws.pos;
    if (ws.err) return ORC_ERR_TRANSCRIPT;
    return rc;
}

/* --------------------------------------------------------------- verifier */
typedef struct {
    const uint8_t *buf;
    size_t len, pos;
    int err;
} rstream;
static const uint8_t *rs_read(rstream *s, size_t n) {
    if (s->pos + n > s->len) { s->err = 1; return NULL; }
    const uint8_t *p = s->buf + s->pos;
    s->pos += n;
    return p;
}
static int rs_read_integer(rstream *s, uint64_t *v, uint32_t limbs) {
    for (uint32_t i = 0; i < limbs; i++) {
        const uint8_t *b = rs_read(s, 8);
        if (!b) return 1;
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) w |= (uint64_t)b[k] << (8 * k);
        v[i] = w;
    }
    return 0;
}

/* utils.rs inner_product over Int<M> (zip/utils.rs:12-24) */
static int inner_product_int(const uint64_t *a, const uint64_t *b, uint32_t n, uint32_t M,
                             uint64_t *out) {
    int err = 0;
    uint64_t prod[MAXW];
    memset(out, 0, 8 * M);
    for (uint32_t i = 0; i < n; i++) {
        err |= wi_mul_checked(prod, a + (size_t)M * i, b + (size_t)M * i, M);
        if (i == 0) memcpy(out, prod, 8 * M);
        else err |= wi_add_checked(out, prod, M);
    }
    return err;
}

int orc_verify(const orc_params *p, const orc_field *f, const uint8_t *roots,
               const uint64_t *point, const uint64_t *eval, orc_keccak *fs,
               const uint8_t *proof, size_t proof_len, int check_merkle) {
    rstream rs = {proof, proof_len, 0, 0};
    const uint32_t R = p->num_rows, C = p->row_len, cw = p->codeword_len, fl = f->fl;
    const uint32_t K = p->k_limbs, M = p->m_limbs, ncol = p->num_column_opening;
    int rc = ORC_OK;
    uint64_t *coeffs_m = NULL, *enc_comb = NULL;
    uint32_t *cols = (uint32_t *)malloc(4 * (size_t)ncol);
    uint64_t *colvals = (uint64_t *)malloc((size_t)8 * K * R * ncol);

    /* verify_testing, verify_z.rs:60-105 (num_proximity_testing == 1 supported) */
    if (R > 1 && p->num_proximity_testing > 0) {
        uint64_t *coeffs = (uint64_t *)malloc((size_t)8 * p->n_limbs * R);
        uint64_t *comb = (uint64_t *)malloc((size_t)8 * M * C);
        coeffs_m = (uint64_t *)malloc((size_t)8 * M * R);
        enc_comb = (uint64_t *)malloc((size_t)8 * M * cw);
        for (uint32_t r = 0; r < R; r++)
            orc_tr_get_integer_challenge(fs, p->n_limbs, coeffs + (size_t)p->n_limbs * r);
        for (uint32_t c = 0; c < C; c++)
            if (rs_read_integer(&rs, comb + (size_t)M * c, M)) rc = ORC_ERR_TRANSCRIPT;
        for (uint32_t r = 0; r < R; r++) wi_sext(coeffs_m + (size_t)M * r, M, coeffs + (size_t)p->n_limbs * r, p->n_limbs);
        if (rc == ORC_OK &&
            orc_raa_encode_row(comb, M, C, p->rep, p->perm1, p->perm2, enc_comb, M) != ORC_OK)
            rc = ORC_ERR_OVERFLOW;
        free(coeffs);
        free(comb);
    }
    for (uint32_t i = 0; i < ncol && rc == ORC_OK; i++) {
        uint32_t col = squeeze_challenge_idx(fs, f, cw);
        cols[i] = col;
        uint64_t *cv = colvals + (size_t)K * R * i;
        for (uint32_t r = 0; r < R; r++)
            if (rs_read_integer(&rs, cv + (size_t)K * r, K)) rc = ORC_ERR_TRANSCRIPT;
        if (rc) break;
        /* verify_column_testing :107-127 */
        uint64_t lhs[MAXW];
        if (R > 1) {
            uint64_t *ce = (uint64_t *)malloc((size_t)8 * M * R);
            for (uint32_t r = 0; r < R; r++) wi_sext(ce + (size_t)M * r, M, cv + (size_t)K * r, K);
            if (inner_product_int(coeffs_m, ce, R, M, lhs)) rc = ORC_ERR_OVERFLOW;
            free(ce);
            if (rc == ORC_OK && memcmp(lhs, enc_comb + (size_t)M * col, 8 * M) != 0) rc = ORC_ERR_PROOF;
        }
        /* ColumnOpening::verify_column, pcs/utils.rs:235-249.  The reference discards
         * the result (verify_z.rs:99) and stops reading the column's remaining proofs
         * at the first failure; check_merkle=1 turns a failure into a rejection.  The
         * stream is consumed sequentially; the hashing of a column is spread over threads. */
        {
            const uint8_t **paths = (const uint8_t **)malloc(sizeof(uint8_t *) * R);
            uint64_t *plens = (uint64_t *)malloc(8 * (size_t)R);
            uint32_t nread = 0;
            for (uint32_t r = 0; r < R && rc == ORC_OK; r++) {
                const uint8_t *lb = rs_read(&rs, 8);
                if (!lb) { rc = ORC_ERR_TRANSCRIPT; break; }
                uint64_t plen = 0;
                for (int k = 0; k < 8; k++) plen = (plen << 8) | lb[k];
                if (plen > 64) { rc = ORC_ERR_TRANSCRIPT; break; }
                const uint8_t *path = rs_read(&rs, 32 * (size_t)plen);
                if (!path) { rc = ORC_ERR_TRANSCRIPT; break; }
                paths[r] = path;
                plens[r] = plen;
                nread = r + 1;
            }
            if (rc == ORC_OK) {
                int64_t first_bad = -1;
#pragma omp parallel for schedule(static)
                for (uint32_t r = 0; r < nread; r++) {
                    if (orc_merkle_verify((uint32_t)plens[r], paths[r], roots + 32 * (size_t)r, cv + (size_t)K * r, K,
                                          col) != ORC_OK) {
#pragma omp critical
                        if (first_bad < 0 || (int64_t)r < first_bad) first_bad = r;
                    }
                }
                if (first_bad >= 0) {
                    if (check_merkle) {
                        rc = ORC_ERR_PROOF;
                    } else {
                        /* faithful mode: the reference returns from verify_column at the first bad
                         * path, leaving the rest of this column's proofs unread in the stream */
                        size_t unread = 0;
                        for (uint32_t r = (uint32_t)first_bad + 1; r < nread; r++) unread += 8 + 32 * (size_t)plens[r];
                        rs.pos -= unread;
                    }
                }
            }
            free(paths);
            free(plens);
        }
    }

    /* verify_evaluation_z :129-163 */
    if (rc == ORC_OK) {
        uint64_t *rowf = (uint64_t *)malloc((size_t)8 * fl * C);
        uint64_t *encf = (uint64_t *)malloc((size_t)8 * fl * cw);
        for (uint32_t c = 0; c < C && rc == ORC_OK; c++) { /* read_field_element */
            const uint8_t *b = rs_read(&rs, 8 * fl);
            if (!b) { rc = ORC_ERR_TRANSCRIPT; break; }
            for (uint32_t i = 0; i < fl; i++) {
                uint64_t w = 0;
                for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (fl - 1 - i) + k];
                rowf[(size_t)fl * c + i] = w;
            }
            orc_tr_absorb_field(fs, f, rowf + (size_t)fl * c);
        }
        if (rc == ORC_OK) {
            orc_raa_encode_row_field(f, rowf, C, p->rep, p->perm1, p->perm2, encf);
            uint32_t lr = ilog2_u64(R), lc = p->num_vars - lr;
            uint64_t *q0 = (uint64_t *)calloc((size_t)fl * R, 8);
            uint64_t *q1 = (uint64_t *)calloc((size_t)fl << lc, 8);
            if (lr) orc_build_eq_x_r(f, point + (size_t)fl * lc, lr, q0);
            if (lc) orc_build_eq_x_r(f, point, lc, q1);
            uint64_t acc[ORC_MAX_FL] = {0}, t[ORC_MAX_FL];
            uint32_t nq1 = lc ? (1u << lc) : 0;
            for (uint32_t c = 0; c < C && c < nq1; c++) {
                memcpy(t, rowf + (size_t)fl * c, 8 * fl);
                orc_field_mul(f, t, q1 + (size_t)fl * c);
                orc_field_add(f, acc, t);
            }
            if (memcmp(acc, eval, 8 * fl) != 0) rc = ORC_ERR_PROOF;
            for (uint32_t i = 0; i < ncol && rc == ORC_OK; i++) { /* verify_proximity_q_0 :165-188 */
                const uint64_t *cv = colvals + (size_t)K * R * i;
                uint64_t lhs[ORC_MAX_FL] = {0}, e[ORC_MAX_FL];
                if (R > 1) {
                    for (uint32_t r = 0; r < R; r++) {
                        orc_field_from_int(f, cv + (size_t)K * r, K, e);
                        memcpy(t, q0 + (size_t)fl * r, 8 * fl);
                        orc_field_mul(f, t, e);
                        orc_field_add(f, lhs, t);
                    }
                } else {
                    orc_field_from_int(f, cv, K, lhs);
                }
                if (memcmp(lhs, encf + (size_t)fl * cols[i], 8 * fl) != 0) rc = ORC_ERR_PROOF;
            }
            free(q0);
            free(q1);
        }
        free(rowf);
        free(encf);
    }
    free(coeffs_m);
    free(enc_comb);
    free(cols);
    free(colvals);
    return rc;
}

/* poly_f/mle/dense.rs evaluate == <eq(point), phi(evals)> (prover.rs:317-319) */
void orc_mle_eval_field(const orc_field *f, const uint64_t *evals, uint32_t eval_limbs,
                        uint32_t num_vars, const uint64_t *point, uint64_t *out) {
    const uint32_t fl = f->fl;
    size_t n = (size_t)1 << num_vars;
    uint64_t acc[ORC_MAX_FL] = {0};
    if (num_vars == 0) {
        orc_field_from_int(f, evals, eval_limbs, out);
        return;
    }
    uint64_t *eq = (uint64_t *)malloc(8 * fl * n);
    orc_build_eq_x_r(f, point, num_vars, eq);
    for (size_t i = 0; i < n; i++) {
        uint64_t e[ORC_MAX_FL];
        orc_field_from_int(f, evals + eval_limbs * i, eval_limbs, e);
        orc_field_mul(f, e, eq + fl * i);
        orc_field_add(f, acc, e);
    }
    memcpy(out, acc, 8 * fl);
    free(eq);
}

/* ------------------------------------------------------------------ sumcheck prover */
/* comb_fn over the values of the MLEs at one point: the product of all of them (n_terms == 0,
 * zinc/prover.rs:300) or sumcheck_polynomial_comb_fn_1 (zinc/utils.rs:77-94) */
#define SC_MAX_MLES 32
static void sc_comb(const orc_field *f, uint64_t (*vals)[ORC_MAX_FL], uint32_t n_mles, uint32_t n_terms,
                    const uint32_t *term_masks, const uint64_t *coeffs, int times_last, uint64_t *out) {
    const uint32_t fl = f->fl;
    if (n_terms == 0) {
        memcpy(out, vals[0], 8 * fl);
        for (uint32_t k = 1; k < n_mles; k++) orc_field_mul(f, out, vals[k]);
        return;
    }
    uint64_t result[ORC_MAX_FL] = {0}, term[ORC_MAX_FL];
    for (uint32_t t = 0; t < n_terms; t++) {
        memcpy(term, coeffs + (size_t)t * fl, 8 * fl);
        for (uint32_t j = 0; j < n_mles; j++)
            if ((term_masks[t] >> j) & 1u) orc_field_mul(f, term, vals[j]);
        orc_field_add(f, result, term);
    }
    if (times_last) orc_field_mul(f, result, vals[n_mles - 1]); /* eq() is the last MLE */
    memcpy(out, result, 8 * fl);
}

static int sc_prove(const orc_field *f, uint64_t *mles, uint32_t n_mles, uint32_t nvars, uint32_t degree,
                    uint32_t n_terms, const uint32_t *term_masks, const uint64_t *coeffs, int times_last, orc_keccak *tr,
                    uint64_t *msgs_out, uint64_t *randomness_out) {
    const uint32_t fl = f->fl;
    if (nvars == 0 || n_mles == 0 || n_mles > SC_MAX_MLES || degree > 8) return ORC_ERR_PARAM;
    const size_t n = (size_t)1 << nvars;
    uint64_t t[ORC_MAX_FL];
    /* sumcheck.rs:64-76: nvars and degree enter the transcript as field elements (u128 map) */
    orc_field_from_u128(f, nvars, 0, t);
    orc_tr_absorb_field(tr, f, t);
    orc_field_from_u128(f, degree, 0, t);
    orc_tr_absorb_field(tr, f, t);
    for (uint32_t round = 1; round <= nvars; round++) {
        const size_t half = (size_t)1 << (nvars - round);
        if (round > 1) { /* prover.rs:68-86: fix the next variable at the previous challenge */
            const uint64_t *r = randomness_out + (size_t)fl * (round - 2);
            for (uint32_t k = 0; k < n_mles; k++) {
                uint64_t *poly = mles + (size_t)k * n * fl;
                for (size_t b = 0; b < 2 * half; b++) { /* dense.rs:155-164 */
                    uint64_t left[ORC_MAX_FL], a[ORC_MAX_FL];
                    memcpy(left, poly + (2 * b) * fl, 8 * fl);
                    memcpy(a, poly + (2 * b + 1) * fl, 8 * fl);
                    orc_field_sub(f, a, left);
                    orc_field_mul(f, a, r);
                    orc_field_add(f, left, a);
                    memcpy(poly + b * fl, left, 8 * fl);
                }
            }
        }
        /* prover.rs:119-156: evaluations of the round polynomial at 0..degree */
        uint64_t evals[9][ORC_MAX_FL];
        memset(evals, 0, sizeof evals);
        for (size_t b = 0; b < half; b++) {
            uint64_t v0[SC_MAX_MLES][ORC_MAX_FL], vals[SC_MAX_MLES][ORC_MAX_FL], step[SC_MAX_MLES][ORC_MAX_FL], c[ORC_MAX_FL];
            for (uint32_t k = 0; k < n_mles; k++)
                memcpy(v0[k], mles + ((size_t)k * n + 2 * b) * fl, 8 * fl);
            sc_comb(f, v0, n_mles, n_terms, term_masks, coeffs, times_last, c);
            orc_field_add(f, evals[0], c);
            if (degree > 0) {
                for (uint32_t k = 0; k < n_mles; k++) {
                    memcpy(vals[k], mles + ((size_t)k * n + 2 * b + 1) * fl, 8 * fl);
                    memcpy(step[k], vals[k], 8 * fl);
                    orc_field_sub(f, step[k], v0[k]);
                }
                sc_comb(f, vals, n_mles, n_terms, term_masks, coeffs, times_last, c);
                orc_field_add(f, evals[1], c);
                for (uint32_t e = 2; e <= degree; e++) {
                    for (uint32_t k = 0; k < n_mles; k++) orc_field_add(f, vals[k], step[k]);
                    sc_comb(f, vals, n_mles, n_terms, term_masks, coeffs, times_last, c);
                    orc_field_add(f, evals[e], c);
                }
            }
        }
        uint64_t *msg = msgs_out + (size_t)(round - 1) * (degree + 1) * fl;
        for (uint32_t e = 0; e <= degree; e++) {
            memcpy(msg + (size_t)e * fl, evals[e], 8 * fl);
            orc_tr_absorb_field(tr, f, evals[e]); /* sumcheck.rs:100 absorb_slice */
        }
        uint64_t *r = randomness_out + (size_t)fl * (round - 1);
        orc_tr_get_challenge(tr, f, r);  /* sample_round, verifier.rs:150-154 */
        orc_tr_absorb_field(tr, f, r);   /* sumcheck.rs:103 */
    }
    return ORC_OK;
}

int orc_sumcheck_prove(const orc_field *f, uint64_t *mles, uint32_t n_mles, uint32_t nvars, uint32_t degree,
                       uint32_t n_terms, const uint32_t *term_masks, const uint64_t *coeffs, orc_keccak *tr,
                       uint64_t *msgs_out, uint64_t *randomness_out) {
    return sc_prove(f, mles, n_mles, nvars, degree, n_terms, term_masks, coeffs, 1, tr, msgs_out, randomness_out);
}

/* comb(vals) = sum_p coeffs[p] * prod_{j in masks[p]} vals[j]: rand_poly_comb_fn (src/sumcheck/utils.rs:67-78), the
 * combination function of benches/sumcheck_benches.rs and src/sumcheck/tests.rs; up to 32 MLEs */
int orc_sumcheck_prove_products(const orc_field *f, uint64_t *mles, uint32_t n_mles, uint32_t nvars, uint32_t degree,
                                uint32_t n_products, const uint32_t *masks, const uint64_t *coeffs, orc_keccak *tr,
                                uint64_t *msgs_out, uint64_t *randomness_out) {
    if (n_products == 0) return ORC_ERR_PARAM;
    return sc_prove(f, mles, n_mles, nvars, degree, n_products, masks, coeffs, 0, tr, msgs_out, randomness_out);
}

int orc_sumcheck_prove_product(const orc_field *f, uint64_t *mles, uint32_t n_mles, uint32_t nvars,
                               uint32_t degree, orc_keccak *tr, uint64_t *msgs_out,
                               uint64_t *randomness_out) {
    return orc_sumcheck_prove(f, mles, n_mles, nvars, degree, 0, NULL, NULL, tr, msgs_out, randomness_out);
}

/* ==================================================================================================
 * Spartan prover / verifier of ZincProver (src/zinc/prover.rs, src/zinc/verifier.rs), CPU restatement.
 * Test infrastructure like everything in this file.
 * ================================================================================================== */

/* Field inverse by Fermat (q prime).  The reference's Div goes through its own inverse; the result is the
 * unique inverse either way. */
void orc_field_inv(const orc_field *f, const uint64_t *a, uint64_t *out) {
    uint64_t e[ORC_MAX_FL], two[ORC_MAX_FL] = {2}, acc[ORC_MAX_FL], base[ORC_MAX_FL];
    memcpy(e, f->modulus, 8 * f->fl);
    ul_sub(e, two, f->fl); /* q - 2 */
    memcpy(acc, f->r, 8 * f->fl); /* one */
    memcpy(base, a, 8 * f->fl);
    for (uint32_t i = 0; i < 64 * f->fl; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) orc_field_mul(f, acc, base);
        uint64_t sq[ORC_MAX_FL];
        memcpy(sq, base, 8 * f->fl);
        orc_field_mul(f, base, sq);
    }
    memcpy(out, acc, 8 * f->fl);
}

/* interpolate_uni_poly (src/sumcheck/verifier.rs:161-303): the value at x of the unique polynomial of
 * degree < len through (i, p_i[i]), i = 0..len-1.  Same early returns for x in {0..len-1}; the general
 * case is the same Lagrange sum (the reference only organises the denominators to save divisions). */
void orc_interpolate_uni_poly(const orc_field *f, const uint64_t *p_i, uint32_t len, const uint64_t *x,
                              uint64_t *out) {
    const uint32_t fl = f->fl;
    uint64_t j[ORC_MAX_FL] = {0}, evals[33][ORC_MAX_FL], prod[ORC_MAX_FL];
    memcpy(prod, x, 8 * fl);
    memcpy(evals[0], x, 8 * fl);
    for (uint32_t i = 1; i < len; i++) { /* :176-185 */
        if (!memcmp(x, j, 8 * fl)) {
            memcpy(out, p_i + (size_t)(i - 1) * fl, 8 * fl);
            return;
        }
        orc_field_add(f, j, f->r);
        memcpy(evals[i], x, 8 * fl);
        orc_field_sub(f, evals[i], j);
        orc_field_mul(f, prod, evals[i]);
    }
    if (!memcmp(x, j, 8 * fl)) { /* :187-189 */
        memcpy(out, p_i + (size_t)(len - 1) * fl, 8 * fl);
        return;
    }
    uint64_t res[ORC_MAX_FL] = {0};
    for (uint32_t i = 0; i < len; i++) {
        /* denom_i = prod_{k != i} (i - k) as a field element */
        uint64_t den[ORC_MAX_FL], t[ORC_MAX_FL];
        memcpy(den, f->r, 8 * fl);
        for (uint32_t k = 0; k < len; k++) {
            if (k == i) continue;
            orc_field_from_i64(f, (int64_t)i - (int64_t)k, t);
            orc_field_mul(f, den, t);
        }
        orc_field_mul(f, den, evals[i]); /* denom_i * (x - i) */
        orc_field_inv(f, den, t);
        orc_field_mul(f, t, prod);
        orc_field_mul(f, t, p_i + (size_t)i * fl);
        orc_field_add(f, res, t);
    }
    memcpy(out, res, 8 * fl);
}

/* MLSumcheck::verify_as_subprotocol (src/sumcheck.rs:116-160) with verify_round and
 * check_and_generate_subclaim (src/sumcheck/verifier.rs:61-143).  msgs: nvars * (degree + 1) elements.
 * Returns ORC_OK and (point, expected_evaluation), or ORC_ERR_PROOF for SumCheckFailed. */
int orc_sumcheck_verify(const orc_field *f, uint32_t nvars, uint32_t degree, const uint64_t *claimed_sum,
                        const uint64_t *msgs, orc_keccak *tr, uint64_t *point_out, uint64_t *expected_out) {
    const uint32_t fl = f->fl;
    if (degree + 1 > 33) return ORC_ERR_PARAM;
    uint64_t t[ORC_MAX_FL];
    orc_field_from_u128(f, nvars, 0, t);
    orc_tr_absorb_field(tr, f, t);
    orc_field_from_u128(f, degree, 0, t);
    orc_tr_absorb_field(tr, f, t);
    if (nvars == 0) { /* :138-144 */
        orc_tr_absorb_field(tr, f, claimed_sum);
        memcpy(expected_out, claimed_sum, 8 * fl);
        return ORC_OK;
    }
    for (uint32_t i = 0; i < nvars; i++) { /* :155-160 */
        for (uint32_t e = 0; e <= degree; e++)
            orc_tr_absorb_field(tr, f, msgs + ((size_t)i * (degree + 1) + e) * fl);
        uint64_t *r = point_out + (size_t)i * fl;
        orc_tr_get_challenge(tr, f, r);
        orc_tr_absorb_field(tr, f, r);
    }
    uint64_t expected[ORC_MAX_FL];
    memcpy(expected, claimed_sum, 8 * fl);
    for (uint32_t i = 0; i < nvars; i++) { /* verifier.rs:110-137 */
        const uint64_t *ev = msgs + (size_t)i * (degree + 1) * fl;
        uint64_t s[ORC_MAX_FL];
        memcpy(s, ev, 8 * fl);
        if (degree > 0) orc_field_add(f, s, ev + fl);
        if (memcmp(s, expected, 8 * fl)) return ORC_ERR_PROOF;
        orc_interpolate_uni_poly(f, ev, degree + 1, point_out + (size_t)i * fl, expected);
    }
    memcpy(expected_out, expected, 8 * fl);
    return ORC_OK;
}

/* the shape ZincProver's own code supports (see the header): square, power-of-two, MLE list == matrices */
static int ccs_check(const orc_ccs *c) {
    if (!c || !c->M || !c->S_masks || !c->c || c->t < 1 || c->t > 7 || c->q < 1 || c->q > 8) return ORC_ERR_PARAM;
    if (c->s != c->s_prime || c->s < 1 || c->s > 28) return ORC_ERR_PARAM;
    if (c->m != (1u << c->s) || c->n != c->m) return ORC_ERR_PARAM;
    uint32_t pos = 0;
    for (uint32_t i = 0; i < c->q; i++) { /* prepare_lin_sumcheck_polynomial pushes S[i]'s MLEs in order */
        if (c->c[i] == 0) return ORC_ERR_PARAM;
        for (uint32_t j = 0; j < c->t; j++)
            if ((c->S_masks[i] >> j) & 1) {
                if (j != pos) return ORC_ERR_PARAM;
                pos++;
            }
    }
    if (pos != c->t) return ORC_ERR_PARAM;
    for (uint32_t k = 0; k < c->t; k++)
        if (c->M[k].n_rows > c->m || c->M[k].n_cols != c->n) return ORC_ERR_PARAM;
    return ORC_OK;
}

/* mat_vec_mul (src/ccs/utils.rs:47-76) after SparseMatrix::map_to_field (src/sparse_matrix.rs:38-58),
 * padded to 2^s like DenseMultilinearExtension::from_evaluations_vec */
static void ccs_mz(const orc_field *f, const orc_sparse *M, const uint64_t *z_f, uint32_t m, uint64_t *out) {
    const uint32_t fl = f->fl;
    memset(out, 0, (size_t)m * fl * 8);
    for (uint32_t row = 0; row < M->n_rows; row++) {
        uint64_t acc[ORC_MAX_FL] = {0};
        for (uint32_t e = M->row_ptr[row]; e < M->row_ptr[row + 1]; e++) {
            uint64_t v[ORC_MAX_FL], p[ORC_MAX_FL];
            orc_field_from_i64(f, M->values[e], v);
            memcpy(p, z_f + (size_t)M->col_idx[e] * fl, 8 * fl);
            orc_field_mul(f, p, v);
            orc_field_add(f, acc, p);
        }
        memcpy(out + (size_t)row * fl, acc, 8 * fl);
    }
}

int orc_ccs_mz(const orc_field *f, const orc_ccs *ccs, const int64_t *z, uint32_t z_len, uint64_t *mz_out) {
    int rc = ccs_check(ccs);
    if (rc) return rc;
    if (z_len > ccs->m) return ORC_ERR_PARAM;
    const uint32_t fl = f->fl, m = ccs->m;
    uint64_t *z_f = calloc((size_t)m * fl, 8);
    if (!z_f) return ORC_ERR_ALLOC;
    for (uint32_t i = 0; i < z_len; i++) orc_field_from_i64(f, z[i], z_f + (size_t)i * fl);
    for (uint32_t k = 0; k < ccs->t; k++) ccs_mz(f, &ccs->M[k], z_f, m, mz_out + (size_t)k * m * fl);
    free(z_f);
    return ORC_OK;
}

/* sum_k gamma^k * compute_eval_table_sparse(M_k, eq_rx) (src/sparse_matrix.rs:165-182; the fold of
 * src/zinc/prover.rs:279-290) */
int orc_ccs_second_table(const orc_field *f, const orc_ccs *ccs, const uint64_t *eq_rx, const uint64_t *gamma,
                         uint64_t *out) {
    int rc = ccs_check(ccs);
    if (rc) return rc;
    const uint32_t fl = f->fl, m = ccs->m;
    uint64_t *tab = calloc((size_t)m * fl, 8);
    if (!tab) return ORC_ERR_ALLOC;
    memset(out, 0, (size_t)m * fl * 8);
    for (int32_t k = (int32_t)ccs->t - 1; k >= 0; k--) { /* .rev().fold: lin = lin * gamma + table_k */
        const orc_sparse *M = &ccs->M[k];
        memset(tab, 0, (size_t)m * fl * 8);
        for (uint32_t row = 0; row < M->n_rows; row++)
            for (uint32_t e = M->row_ptr[row]; e < M->row_ptr[row + 1]; e++) {
                uint64_t v[ORC_MAX_FL], p[ORC_MAX_FL];
                orc_field_from_i64(f, M->values[e], v);
                memcpy(p, eq_rx + (size_t)row * fl, 8 * fl);
                orc_field_mul(f, p, v);
                orc_field_add(f, tab + (size_t)M->col_idx[e] * fl, p);
            }
        for (uint32_t i = 0; i < m; i++) {
            orc_field_mul(f, out + (size_t)i * fl, gamma);
            orc_field_add(f, out + (size_t)i * fl, tab + (size_t)i * fl);
        }
    }
    free(tab);
    return ORC_OK;
}

static void dot_field(const orc_field *f, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
    uint64_t acc[ORC_MAX_FL] = {0};
    for (size_t i = 0; i < n; i++) {
        uint64_t p[ORC_MAX_FL];
        memcpy(p, a + i * f->fl, 8 * f->fl);
        orc_field_mul(f, p, b + i * f->fl);
        orc_field_add(f, acc, p);
    }
    memcpy(out, acc, 8 * f->fl);
}

/* SpartanProver::prove (src/zinc/prover.rs:130-161) after prepare_for_random_field_piop (:172-239).
 *   z: x || 1 || w (Statement_Z::get_z_vector), z_len <= m; zero-extended to m (:230-232)
 * Outputs: msgs1 s*(d+2) elements, r_x s, msgs2 s*3, r_y s, V_s t. */
int orc_spartan_prove(const orc_field *f, const orc_ccs *ccs, const int64_t *z, uint32_t z_len, orc_keccak *tr,
                      uint64_t *msgs1, uint64_t *r_x, uint64_t *msgs2, uint64_t *r_y, uint64_t *V_s) {
    int rc = ccs_check(ccs);
    if (rc) return rc;
    if (z_len > ccs->m) return ORC_ERR_PARAM;
    const uint32_t fl = f->fl, m = ccs->m, s = ccs->s, t = ccs->t;
    const size_t tab = (size_t)m * fl;
    uint64_t *mz = calloc(tab * t, 8), *g = calloc(tab * (t + 1), 8), *eq = calloc(tab, 8), *two = calloc(tab * 2, 8);
    uint64_t *coeffs = calloc((size_t)ccs->q * fl, 8), *beta = calloc((size_t)s * fl, 8);
    rc = ORC_ERR_ALLOC;
    if (!mz || !g || !eq || !two || !coeffs || !beta) goto done;
    /* sumcheck_1 (:242-259): beta (zinc/utils.rs:100-106), Mz MLEs, g = [Mz.., eq(beta)], degree d + 1 */
    orc_keccak_update(tr, (const uint8_t *)"beta_s", 6);
    for (uint32_t i = 0; i < s; i++) orc_tr_get_challenge(tr, f, beta + (size_t)i * fl);
    if ((rc = orc_ccs_mz(f, ccs, z, z_len, mz))) goto done;
    memcpy(g, mz, tab * t * 8);
    if ((rc = orc_build_eq_x_r(f, beta, s, g + tab * t))) goto done;
    for (uint32_t i = 0; i < ccs->q; i++) orc_field_from_i64(f, ccs->c[i], coeffs + (size_t)i * fl);
    if ((rc = orc_sumcheck_prove(f, g, t + 1, s, ccs->d + 1, ccs->q, ccs->S_masks, coeffs, tr, msgs1, r_x))) goto done;
    /* sumcheck_2 (:261-303) */
    uint64_t gamma[ORC_MAX_FL];
    orc_keccak_update(tr, (const uint8_t *)"gamma", 5);
    orc_tr_get_challenge(tr, f, gamma);
    if ((rc = orc_build_eq_x_r(f, r_x, s, eq))) goto done;
    if ((rc = orc_ccs_second_table(f, ccs, eq, gamma, two))) goto done;
    for (uint32_t i = 0; i < m; i++) /* z_mle.map_to_field (:149) */
        if (i < z_len) orc_field_from_i64(f, z[i], two + tab + (size_t)i * fl);
    if ((rc = orc_sumcheck_prove(f, two, 2, s, 2, 0, NULL, NULL, tr, msgs2, r_y))) goto done;
    /* calculate_V_s (:330-347): Mz_k evaluated at r_x */
    for (uint32_t k = 0; k < t; k++) dot_field(f, mz + tab * k, eq, m, V_s + (size_t)k * fl);
    rc = ORC_OK;
done:
    free(mz); free(g); free(eq); free(two); free(coeffs); free(beta);
    return rc;
}

/* lin_comb_V_s (src/zinc/verifier.rs:212-219) */
static void lin_comb(const orc_field *f, const uint64_t *gamma, const uint64_t *v, uint32_t n, uint64_t *out) {
    uint64_t res[ORC_MAX_FL] = {0};
    for (int32_t i = (int32_t)n - 1; i >= 0; i--) {
        orc_field_mul(f, res, gamma);
        orc_field_add(f, res, v + (size_t)i * f->fl);
    }
    memcpy(out, res, 8 * f->fl);
}

/* SpartanVerifier::verify (src/zinc/verifier.rs:105-139).  Returns ORC_OK with the verification points
 * (r_x, r_y, e_y, gamma) or ORC_ERR_PROOF. */
int orc_spartan_verify(const orc_field *f, const orc_ccs *ccs, const uint64_t *msgs1, const uint64_t *msgs2,
                       const uint64_t *V_s, orc_keccak *tr, uint64_t *r_x, uint64_t *r_y, uint64_t *e_y,
                       uint64_t *gamma) {
    int rc = ccs_check(ccs);
    if (rc) return rc;
    const uint32_t fl = f->fl, s = ccs->s;
    uint64_t beta[32][ORC_MAX_FL], zero[ORC_MAX_FL] = {0}, sv[ORC_MAX_FL];
    orc_keccak_update(tr, (const uint8_t *)"beta_s", 6);
    for (uint32_t i = 0; i < s; i++) orc_tr_get_challenge(tr, f, beta[i]);
    if ((rc = orc_sumcheck_verify(f, s, ccs->d + 1, zero, msgs1, tr, r_x, sv))) return rc; /* :142-162 */
    /* verify_linearization_claim (:164-187) */
    uint64_t e[ORC_MAX_FL], sum[ORC_MAX_FL] = {0};
    memcpy(e, f->r, 8 * fl);
    for (uint32_t i = 0; i < s; i++) { /* eq_eval, sumcheck/utils.rs:81-95 */
        uint64_t xy[ORC_MAX_FL], term[ORC_MAX_FL];
        memcpy(xy, r_x + (size_t)i * fl, 8 * fl);
        orc_field_mul(f, xy, beta[i]);
        memcpy(term, xy, 8 * fl);
        orc_field_add(f, term, xy);
        orc_field_sub(f, term, r_x + (size_t)i * fl);
        orc_field_sub(f, term, beta[i]);
        orc_field_add(f, term, f->r);
        orc_field_mul(f, e, term);
    }
    for (uint32_t i = 0; i < ccs->q; i++) {
        uint64_t term[ORC_MAX_FL];
        orc_field_from_i64(f, ccs->c[i], term);
        for (uint32_t j = 0; j < ccs->t; j++)
            if ((ccs->S_masks[i] >> j) & 1) orc_field_mul(f, term, V_s + (size_t)j * fl);
        orc_field_add(f, sum, term);
    }
    orc_field_mul(f, e, sum);
    if (memcmp(e, sv, 8 * fl)) return ORC_ERR_PROOF;
    orc_keccak_update(tr, (const uint8_t *)"gamma", 5);
    orc_tr_get_challenge(tr, f, gamma);
    uint64_t claimed[ORC_MAX_FL];
    lin_comb(f, gamma, V_s, ccs->t, claimed);
    return orc_sumcheck_verify(f, ccs->s_prime, 2, claimed, msgs2, tr, r_y, e_y); /* :189-210 */
}

/* The final check of verify_pcs_proof (src/zinc/verifier.rs:248-269):
 *   lin_comb(gamma, [mle(M_k)(r_x, r_y)]_k) * v == e_y,
 * with DenseMultilinearExtension::from_matrix (src/poly_f/mle/dense.rs:69-87: index = rows * col + row, so the
 * low s variables select the row) evaluated as sum val * eq(r_x)[row] * eq(r_y)[col]. */
int orc_ccs_eval_matrices(const orc_field *f, const orc_ccs *ccs, const uint64_t *r_x, const uint64_t *r_y,
                          uint64_t *v_xy) {
    int rc = ccs_check(ccs);
    if (rc) return rc;
    const uint32_t fl = f->fl, m = ccs->m;
    uint64_t *ex = calloc((size_t)m * fl, 8), *ey = calloc((size_t)m * fl, 8);
    if (!ex || !ey) { free(ex); free(ey); return ORC_ERR_ALLOC; }
    orc_build_eq_x_r(f, r_x, ccs->s, ex);
    orc_build_eq_x_r(f, r_y, ccs->s_prime, ey);
    for (uint32_t k = 0; k < ccs->t; k++) {
        const orc_sparse *M = &ccs->M[k];
        uint64_t acc[ORC_MAX_FL] = {0};
        for (uint32_t row = 0; row < M->n_rows; row++)
            for (uint32_t e = M->row_ptr[row]; e < M->row_ptr[row + 1]; e++) {
                uint64_t p[ORC_MAX_FL];
                orc_field_from_i64(f, M->values[e], p);
                orc_field_mul(f, p, ex + (size_t)row * fl);
                orc_field_mul(f, p, ey + (size_t)M->col_idx[e] * fl);
                orc_field_add(f, acc, p);
            }
        memcpy(v_xy + (size_t)k * fl, acc, 8 * fl);
    }
    free(ex); free(ey);
    return ORC_OK;
}

int orc_spartan_final_check(const orc_field *f, const orc_ccs *ccs, const uint64_t *r_x, const uint64_t *r_y,
                            const uint64_t *gamma, const uint64_t *v, const uint64_t *e_y) {
    uint64_t vxy[8 * ORC_MAX_FL], lhs[ORC_MAX_FL];
    int rc = orc_ccs_eval_matrices(f, ccs, r_x, r_y, vxy);
    if (rc) return rc;
    lin_comb(f, gamma, vxy, ccs->t, lhs); /* lin_comb_V_s over V_xy */
    orc_field_mul(f, lhs, v);
    return memcmp(lhs, e_y, 8 * f->fl) ? ORC_ERR_PROOF : ORC_OK;
}
