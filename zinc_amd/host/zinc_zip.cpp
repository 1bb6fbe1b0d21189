// Host side of the Zip commit/open path (see zinc_zip.hpp).  Only sequential, tiny work lives
// here; MultilinearZip::commit / open forward every O(n) loop to libzip_hip.so.
#include "zinc_zip.hpp"

#include <chrono>
#include <cstdio>

#include <algorithm>
#include <list>
#include <map>
#include <set>
#include <mutex>

#include <sys/mman.h>

namespace zinc {

namespace {
// Freed proof- / witness-sized blocks are kept (exact sizes recur from proof to proof; at most 4 GiB) instead of
// going back to the kernel: a fresh 383 MiB block costs as much in first-touch page faults as its PCIe transfer.
struct BigBlockCache {
    std::mutex mu;
    std::multimap<size_t, void *> blocks;
    size_t bytes = 0;
    std::set<void *> pinned;  // blocks registered with the HIP runtime (unregistered before they are freed)
};
BigBlockCache g_big;
void big_block_free(void *p) {  // g_big.mu held
    if (g_big.pinned.erase(p)) zip_host_unregister(p);
    std::free(p);
}
constexpr size_t kBigBlock = (size_t)16 << 20, kBigCacheCap = (size_t)4 << 30;
}  // namespace

void *byte_stream_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes >= kBigBlock) {
        {
            std::lock_guard<std::mutex> g(g_big.mu);
            auto it = g_big.blocks.find(bytes);
            if (it != g_big.blocks.end()) {
                p = it->second;
                g_big.blocks.erase(it);
                g_big.bytes -= bytes;
                return p;
            }
        }
        if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
        (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
        // pinned once, reused for every later proof of this size: the device then writes the proof straight into
        // it (ZINC_HOST_NO_PIN=1 keeps the blocks pageable)
        static const bool pin = std::getenv("ZINC_HOST_NO_PIN") == nullptr;
        if (pin && zip_host_register(p, bytes) == ZIP_OK) {
            std::lock_guard<std::mutex> g(g_big.mu);
            g_big.pinned.insert(p);
        }
        return p;
    }
    p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
}

void byte_stream_free(void *p, size_t bytes) {
    if (p && bytes >= kBigBlock) {
        std::lock_guard<std::mutex> g(g_big.mu);
        if (g_big.bytes + bytes <= kBigCacheCap) {
            g_big.blocks.emplace(bytes, p);
            g_big.bytes += bytes;
            return;
        }
        big_block_free(p);
        return;
    }
    std::free(p);
}

void byte_stream_release_cache() {
    std::lock_guard<std::mutex> g(g_big.mu);
    for (auto &kv : g_big.blocks) big_block_free(kv.second);
    g_big.blocks.clear();
    g_big.bytes = 0;
}

using u128 = unsigned __int128;

// ============================================================================ Keccak-256
// sha3 crate `Keccak256` (src/transcript.rs:2): Keccak-f[1600], rate 136, domain byte 0x01.
static inline uint64_t rol64(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }

void Keccak256::permute(uint64_t a[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // theta, rho + pi, chi, iota with the 25 lanes in registers (lane (x, y) = a[x + 5y]; unrolled by a script from the
    // rotation table 0 1 62 28 27 / 36 44 6 55 20 / 3 10 43 25 39 / 41 45 15 21 8 / 18 2 61 56 14): four times the speed
    // of the loop form, and the transcript is on the critical path of every sumcheck round and every opening
    uint64_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7], a8 = a[8], a9 = a[9], a10 = a[10], a11 = a[11], a12 = a[12], a13 = a[13], a14 = a[14], a15 = a[15], a16 = a[16], a17 = a[17], a18 = a[18], a19 = a[19], a20 = a[20], a21 = a[21], a22 = a[22], a23 = a[23], a24 = a[24];
    for (int round = 0; round < 24; round++) {
        const uint64_t c0 = a0 ^ a5 ^ a10 ^ a15 ^ a20;
        const uint64_t c1 = a1 ^ a6 ^ a11 ^ a16 ^ a21;
        const uint64_t c2 = a2 ^ a7 ^ a12 ^ a17 ^ a22;
        const uint64_t c3 = a3 ^ a8 ^ a13 ^ a18 ^ a23;
        const uint64_t c4 = a4 ^ a9 ^ a14 ^ a19 ^ a24;
        const uint64_t d0 = c4 ^ rol64(c1, 1);
        const uint64_t d1 = c0 ^ rol64(c2, 1);
        const uint64_t d2 = c1 ^ rol64(c3, 1);
        const uint64_t d3 = c2 ^ rol64(c4, 1);
        const uint64_t d4 = c3 ^ rol64(c0, 1);
        const uint64_t b0 = (a0 ^ d0);
        const uint64_t b1 = rol64((a6 ^ d1), 44);
        const uint64_t b2 = rol64((a12 ^ d2), 43);
        const uint64_t b3 = rol64((a18 ^ d3), 21);
        const uint64_t b4 = rol64((a24 ^ d4), 14);
        const uint64_t b5 = rol64((a3 ^ d3), 28);
        const uint64_t b6 = rol64((a9 ^ d4), 20);
        const uint64_t b7 = rol64((a10 ^ d0), 3);
        const uint64_t b8 = rol64((a16 ^ d1), 45);
        const uint64_t b9 = rol64((a22 ^ d2), 61);
        const uint64_t b10 = rol64((a1 ^ d1), 1);
        const uint64_t b11 = rol64((a7 ^ d2), 6);
        const uint64_t b12 = rol64((a13 ^ d3), 25);
        const uint64_t b13 = rol64((a19 ^ d4), 8);
        const uint64_t b14 = rol64((a20 ^ d0), 18);
        const uint64_t b15 = rol64((a4 ^ d4), 27);
        const uint64_t b16 = rol64((a5 ^ d0), 36);
        const uint64_t b17 = rol64((a11 ^ d1), 10);
        const uint64_t b18 = rol64((a17 ^ d2), 15);
        const uint64_t b19 = rol64((a23 ^ d3), 56);
        const uint64_t b20 = rol64((a2 ^ d2), 62);
        const uint64_t b21 = rol64((a8 ^ d3), 55);
        const uint64_t b22 = rol64((a14 ^ d4), 39);
        const uint64_t b23 = rol64((a15 ^ d0), 41);
        const uint64_t b24 = rol64((a21 ^ d1), 2);
        a0 = b0 ^ (~b1 & b2);
        a1 = b1 ^ (~b2 & b3);
        a2 = b2 ^ (~b3 & b4);
        a3 = b3 ^ (~b4 & b0);
        a4 = b4 ^ (~b0 & b1);
        a5 = b5 ^ (~b6 & b7);
        a6 = b6 ^ (~b7 & b8);
        a7 = b7 ^ (~b8 & b9);
        a8 = b8 ^ (~b9 & b5);
        a9 = b9 ^ (~b5 & b6);
        a10 = b10 ^ (~b11 & b12);
        a11 = b11 ^ (~b12 & b13);
        a12 = b12 ^ (~b13 & b14);
        a13 = b13 ^ (~b14 & b10);
        a14 = b14 ^ (~b10 & b11);
        a15 = b15 ^ (~b16 & b17);
        a16 = b16 ^ (~b17 & b18);
        a17 = b17 ^ (~b18 & b19);
        a18 = b18 ^ (~b19 & b15);
        a19 = b19 ^ (~b15 & b16);
        a20 = b20 ^ (~b21 & b22);
        a21 = b21 ^ (~b22 & b23);
        a22 = b22 ^ (~b23 & b24);
        a23 = b23 ^ (~b24 & b20);
        a24 = b24 ^ (~b20 & b21);
        a0 ^= RC[round];
    }
    a[0] = a0;
    a[1] = a1;
    a[2] = a2;
    a[3] = a3;
    a[4] = a4;
    a[5] = a5;
    a[6] = a6;
    a[7] = a7;
    a[8] = a8;
    a[9] = a9;
    a[10] = a10;
    a[11] = a11;
    a[12] = a12;
    a[13] = a13;
    a[14] = a14;
    a[15] = a15;
    a[16] = a16;
    a[17] = a17;
    a[18] = a18;
    a[19] = a19;
    a[20] = a20;
    a[21] = a21;
    a[22] = a22;
    a[23] = a23;
    a[24] = a24;
}

void Keccak256::absorb_block(const uint8_t *blk) {
    for (int i = 0; i < 17; i++) {
        uint64_t w;
        std::memcpy(&w, blk + 8 * i, 8);  // little-endian host
        st_[i] ^= w;
    }
    permute(st_);
}

void Keccak256::update(const uint8_t *data, size_t len) {
    while (len) {
        const size_t take = std::min<size_t>(136 - buflen_, len);
        std::memcpy(buf_ + buflen_, data, take);
        buflen_ += (uint32_t)take;
        data += take;
        len -= take;
        if (buflen_ == 136) {
            absorb_block(buf_);
            buflen_ = 0;
        }
    }
}

std::array<uint8_t, 32> Keccak256::finalize() const {
    Keccak256 c = *this;
    std::memset(c.buf_ + c.buflen_, 0, 136 - c.buflen_);
    c.buf_[c.buflen_] ^= 0x01;
    c.buf_[135] ^= 0x80;
    c.absorb_block(c.buf_);
    std::array<uint8_t, 32> out;
    std::memcpy(out.data(), c.st_, 32);
    return out;
}

// ============================================================================ multi-limb helpers
namespace {

int cmp(const Limbs &a, const Limbs &b, uint32_t n) {
    for (uint32_t i = n; i-- > 0;)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
bool add_in_place(Limbs &a, const Limbs &b, uint32_t n) {
    u128 c = 0;
    for (uint32_t i = 0; i < n; i++) {
        c += (u128)a[i] + b[i];
        a[i] = (uint64_t)c;
        c >>= 64;
    }
    return c != 0;
}
void sub_in_place(Limbs &a, const Limbs &b, uint32_t n) {
    uint64_t borrow = 0;
    for (uint32_t i = 0; i < n; i++) {
        const u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
bool is_zero(const Limbs &a, uint32_t n) {
    for (uint32_t i = 0; i < n; i++)
        if (a[i]) return false;
    return true;
}
void negate(Limbs &a, uint32_t n) {  // two's complement
    uint64_t carry = 1;
    for (uint32_t i = 0; i < n; i++) {
        a[i] = ~a[i] + carry;
        carry = (carry && a[i] == 0) ? 1 : 0;
    }
}
uint32_t bit_length(const Limbs &a, uint32_t n) {
    for (uint32_t i = n; i-- > 0;)
        if (a[i]) return 64 * i + 64 - (uint32_t)__builtin_clzll(a[i]);
    return 0;
}
// a mod m for n-limb unsigned values (schoolbook shift-subtract; only used off the hot path)
Limbs umod(const Limbs &a, const Limbs &m, uint32_t n) {
    if (cmp(a, m, n) < 0) return a;
    Limbs r{};
    for (uint32_t b = bit_length(a, n); b-- > 0;) {
        const bool top = r[n - 1] >> 63;
        for (uint32_t i = n; i-- > 1;) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
        r[0] = (r[0] << 1) | ((a[b / 64] >> (b % 64)) & 1);
        if (top || cmp(r, m, n) >= 0) sub_in_place(r, m, n);
    }
    return r;
}

// Tail shared by every FieldMap impl: the value and the MODULUS are read as signed Int<W>
// (F::I = Int<N>, src/field.rs:280), `%=` is crypto-bigint's Int::rem (|lhs| mod |rhs|, sign of
// lhs) and BigInt::from(Int) takes the magnitude (biginteger.rs:805-816); then times R^2.
Limbs from_signed_words(const FieldConfig &f, Limbs words) {
    const uint32_t n = f.limbs;
    if (words[n - 1] >> 63) negate(words, n);
    Limbs mod = f.modulus;
    if (mod[n - 1] >> 63) negate(mod, n);
    Limbs v = umod(words, mod, n);
    f.mul_assign(v, f.r2);
    return v;
}

}  // namespace

// ============================================================================ FieldConfig
FieldConfig FieldConfig::make(const uint64_t *modulus, uint32_t limbs) {
    if (limbs == 0 || limbs > kMaxLimbs || !(modulus[0] & 1)) throw std::logic_error("FieldConfig: bad modulus");
    FieldConfig f;
    f.limbs = limbs;
    for (uint32_t i = 0; i < limbs; i++) f.modulus[i] = modulus[i];
    f.modulus_has_spare_bit = (modulus[limbs - 1] >> 63) == 0;
    uint64_t inv = 1;  // config.rs:196-214
    for (int i = 0; i < 63; i++) {
        inv *= inv;
        inv *= modulus[0];
    }
    f.inv = 0 - inv;
    Limbs x{};
    x[0] = 1;
    auto dbl = [&](Limbs &v) {
        const bool top = v[limbs - 1] >> 63;
        for (uint32_t i = limbs; i-- > 1;) v[i] = (v[i] << 1) | (v[i - 1] >> 63);
        v[0] <<= 1;
        if (top || cmp(v, f.modulus, limbs) >= 0) sub_in_place(v, f.modulus, limbs);
    };
    for (uint32_t i = 0; i < 64 * limbs; i++) dbl(x);
    f.r = x;
    for (uint32_t i = 0; i < 64 * limbs; i++) dbl(x);
    f.r2 = x;
    return f;
}

void FieldConfig::reduce_modulus(Limbs &a, bool carry) const {
    if (modulus_has_spare_bit) {
        if (cmp(a, modulus, limbs) >= 0) sub_in_place(a, modulus, limbs);
    } else if (carry || cmp(a, modulus, limbs) >= 0) {
        sub_in_place(a, modulus, limbs);
    }
}

// mul_naive (biginteger.rs:448-464) then montgomery_reduction (:532-560)
void FieldConfig::mul_assign(Limbs &a, const Limbs &b) const {
    const uint32_t N = limbs;
    uint64_t t[2 * kMaxLimbs] = {0};
    for (uint32_t i = 0; i < N; i++) {
        uint64_t carry = 0;
        for (uint32_t j = 0; j < N; j++) {
            const u128 x = (u128)a[i] * b[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
        t[i + N] = carry;
    }
    uint64_t carry2 = 0;
    for (uint32_t i = 0; i < N; i++) {
        const uint64_t k = t[i] * inv;
        uint64_t carry = 0;
        for (uint32_t j = 0; j < N; j++) {
            const u128 x = (u128)k * modulus[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
        const u128 y = (u128)t[i + N] + carry + carry2;
        t[i + N] = (uint64_t)y;
        carry2 = (uint64_t)(y >> 64);
    }
    for (uint32_t i = 0; i < N; i++) a[i] = t[N + i];
    reduce_modulus(a, carry2 != 0);
}

void FieldConfig::add_assign(Limbs &a, const Limbs &b) const { reduce_modulus(a, add_in_place(a, b, limbs)); }

void FieldConfig::sub_assign(Limbs &a, const Limbs &b) const {
    if (cmp(b, a, limbs) > 0) add_in_place(a, modulus, limbs);
    sub_in_place(a, b, limbs);
}

void FieldConfig::neg(Limbs &a) const {
    if (is_zero(a, limbs)) return;
    Limbs t = modulus;
    sub_in_place(t, a, limbs);
    a = t;
}

Limbs FieldConfig::inverse(const Limbs &a) const {
    Limbs e = modulus, two{};
    two[0] = 2;
    sub_in_place(e, two, limbs);  // q - 2
    Limbs acc = r, base = a;
    for (uint32_t i = 0; i < 64 * limbs; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) mul_assign(acc, base);
        const Limbs sq = base;
        mul_assign(base, sq);
    }
    return acc;
}

uint32_t FieldConfig::num_bits() const { return bit_length(modulus, limbs); }

zip_field FieldConfig::to_abi() const {
    zip_field z{};
    z.limbs = limbs;
    for (uint32_t i = 0; i < kMaxLimbs; i++) z.modulus[i] = i < limbs ? modulus[i] : 0;
    return z;
}

Limbs map_to_field(const FieldConfig &f, int64_t v) {
    Limbs w{};
    w[0] = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;  // Integer::abs
    Limbs r = from_signed_words(f, w);
    if (v < 0) f.neg(r);
    return r;
}

Limbs map_to_field_u128(const FieldConfig &f, uint64_t lo, uint64_t hi) {
    Limbs w{};
    w[0] = lo;
    if (f.limbs > 1) w[1] = hi;
    return from_signed_words(f, w);
}

std::vector<Limbs> build_eq_x_r(const FieldConfig &f, const Limbs *r, uint32_t nvars) {
    if (nvars == 0) throw std::logic_error("r length is 0");  // ArithErrors::InvalidParameters
    std::vector<Limbs> buf(2);
    buf[0] = f.r;  // F::one()
    f.sub_assign(buf[0], r[nvars - 1]);
    buf[1] = r[nvars - 1];
    for (uint32_t t = nvars - 1; t-- > 0;) {
        std::vector<Limbs> res(buf.size() * 2);
        for (size_t i = 0; i < res.size(); i++) {
            Limbs tmp = r[t];
            f.mul_assign(tmp, buf[i >> 1]);
            if (i & 1) {
                res[i] = tmp;
            } else {
                res[i] = buf[i >> 1];
                f.sub_assign(res[i], tmp);
            }
        }
        buf.swap(res);
    }
    return buf;
}

// ============================================================================ KeccakTranscript
std::vector<uint8_t> KeccakTranscript::get_random_bytes(size_t length) {
    std::vector<uint8_t> out;
    out.reserve(length + 32);
    for (int32_t counter = 0; out.size() < length; counter++) {
        Keccak256 tmp = hasher_;
        tmp.update({(uint8_t)(counter >> 24), (uint8_t)(counter >> 16), (uint8_t)(counter >> 8), (uint8_t)counter});
        const auto h = tmp.finalize();
        out.insert(out.end(), h.begin(), h.end());
    }
    out.resize(length);
    return out;
}

void KeccakTranscript::get_integer_challenge(uint32_t n_limbs, uint64_t *out) {
    for (uint32_t i = 0; i < n_limbs; i++) {
        const auto ch = get_random_bytes(8);
        hasher_.update({0x12});
        hasher_.update(ch.data(), 8);
        hasher_.update({0x34});
        uint64_t w;
        std::memcpy(&w, ch.data(), 8);  // u64::from_le_bytes
        out[i] = w;
    }
}

std::vector<int64_t> KeccakTranscript::get_integer_challenges_i64(size_t n) {
    std::vector<int64_t> v(n);
    for (size_t i = 0; i < n; i++) {
        uint64_t w;
        get_integer_challenge(1, &w);
        v[i] = (int64_t)w;
    }
    return v;
}

static void to_be_bytes(const Limbs &v, uint32_t n, uint8_t *out) {
    for (uint32_t i = 0; i < n; i++)
        for (int b = 0; b < 8; b++) out[8 * (n - 1 - i) + (7 - b)] = (uint8_t)(v[i] >> (8 * b));
}

void KeccakTranscript::absorb_random_field(const FieldConfig &f, const Limbs &v) {
    uint8_t buf[8 * kMaxLimbs];
    hasher_.update({0x3});
    to_be_bytes(f.modulus, f.limbs, buf);
    hasher_.update(buf, 8 * f.limbs);
    hasher_.update({0x5});
    hasher_.update({0x1});
    to_be_bytes(v, f.limbs, buf);
    hasher_.update(buf, 8 * f.limbs);
    hasher_.update({0x3});
}

Limbs KeccakTranscript::get_challenge(const FieldConfig &f) {
    const auto ch = hasher_.finalize();  // get_challenge_limbs, transcript.rs:72-86
    auto be64 = [&](int off) {
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) w = (w << 8) | ch[off + b];
        return w;
    };
    uint64_t lo0 = be64(8), lo1 = be64(0), hi0 = be64(24), hi1 = be64(16);
    hasher_.update({0x00});
    hasher_.update(ch.data(), 32);
    hasher_.update({0x01});

    const uint32_t cbits = f.num_bits() - 1;
    auto mask128 = [](uint64_t &w0, uint64_t &w1, uint32_t keep) {
        if (keep < 64) {
            w0 &= (1ULL << keep) - 1;
            w1 = 0;
        } else if (keep == 64) {
            w1 = 0;
        } else if (keep < 128) {
            w1 &= (1ULL << (keep - 64)) - 1;
        }
    };
    if (f.limbs == 1) return map_to_field_u128(f, lo0 & ((1ULL << cbits) - 1), 0);
    if (cbits < 128) {
        mask128(lo0, lo1, cbits);
        return map_to_field_u128(f, lo0, lo1);
    }
    if (cbits < 256) mask128(hi0, hi1, cbits - 128);
    Limbs w{};
    if (f.limbs > 2) w[2] = 1;  // BigInt::from_bits_le(bit 128)
    Limbs two128 = from_signed_words(f, w);
    Limbs a = map_to_field_u128(f, lo0, lo1);
    const Limbs b = map_to_field_u128(f, hi0, hi1);
    f.mul_assign(two128, b);
    f.add_assign(a, two128);
    return a;
}

namespace zip {

// ============================================================================ shuffle_seeded
// rand 0.9.2: StdRng = ChaCha12, seed_from_u64 = PCG32 expansion, SliceRandom::shuffle =
// IncreasingUniform Fisher-Yates with Canon's-method random_range.  Every piece pinned by published vectors
// (tests/golden/rand_vectors.json; the seed expansion directly through zinc_kat_seed_from_u64).
namespace {
struct ChaCha12 {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int idx = 16;
    // rand_core::SeedableRng::seed_from_u64: PCG32 steps, the state advanced before each output
    static void seed_words(uint64_t state, uint32_t *out, int n) {
        for (int i = 0; i < n; i++) {
            state = state * 6364136223846793005ULL + 11634580027462260723ULL;
            const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
            out[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
        }
    }
    explicit ChaCha12(uint64_t state) { seed_words(state, key, 8); }
    static inline uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    void refill() {
        uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574}, x[16];
        for (int i = 0; i < 8; i++) s[4 + i] = key[i];
        s[12] = (uint32_t)counter;
        s[13] = (uint32_t)(counter >> 32);
        s[14] = s[15] = 0;
        std::memcpy(x, s, sizeof x);
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < 6; r++) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) buf[i] = x[i] + s[i];
        counter++;
        idx = 0;
    }
    uint32_t next_u32() {
        if (idx >= 16) refill();
        return buf[idx++];
    }
    uint32_t random_range(uint32_t bound) {
        const uint64_t m = (uint64_t)next_u32() * bound;
        uint32_t result = (uint32_t)(m >> 32);
        const uint32_t lo = (uint32_t)m;
        if (lo > (uint32_t)(0u - bound)) {
            const uint32_t hi2 = (uint32_t)(((uint64_t)next_u32() * bound) >> 32);
            result += (uint32_t)(lo + hi2 < lo);
        }
        return result;
    }
};
}  // namespace

void kat_seed_from_u64(uint64_t seed, uint32_t *words, uint32_t n_words) { ChaCha12::seed_words(seed, words, (int)n_words); }

std::vector<uint32_t> shuffle_seeded_perm(uint64_t seed, uint32_t len) {
    std::vector<uint32_t> perm(len);
    for (uint32_t i = 0; i < len; i++) perm[i] = i;
    if (len <= 1) return perm;
    ChaCha12 rng(seed);
    uint32_t n = 0, chunk = 0;
    uint8_t remaining = 1;  // IncreasingUniform::new(rng, 0)
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t next_n = n + 1;
        uint8_t next_rem;
        if (remaining > 0) {
            next_rem = remaining - 1;
        } else {
            uint32_t product = next_n, current = next_n + 1;
            while (((uint64_t)product * current) >> 32 == 0) product *= current++;
            chunk = rng.random_range(product);
            next_rem = (uint8_t)(current - next_n - 1);
        }
        uint32_t j;
        if (next_rem == 0) {
            j = chunk;
        } else {
            j = chunk % next_n;
            chunk /= next_n;
        }
        remaining = next_rem;
        n = next_n;
        std::swap(perm[i], perm[j]);
    }
    return perm;
}

// ============================================================================ RaaCode
static uint64_t isqrt(uint64_t x) {
    uint64_t r = 0;
    for (int b = 31; b >= 0; b--) {
        const uint64_t t = r | (1ULL << b);
        if (t * t <= x) r = t;
    }
    return r;
}
static uint64_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}
static uint32_t ilog2(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }

RaaCode RaaCode::make(const LinearCodeSpec &spec, uint64_t poly_size, SeedSource &transcript) {
    RaaCode c;
    const uint32_t num_vars = ilog2(poly_size);
    c.row_len = (uint32_t)next_pow2(isqrt(1ULL << num_vars));  // code_raa.rs:43
    c.repetition_factor = spec.repetition_factor;
    c.num_column_opening = spec.num_column_opening;
    c.num_proximity_testing = spec.num_proximity_testing;
    // width assertion, code_raa.rs:53-72 (N = Int<1>: 64 bits, K = Int<4>: 256 bits)
    const uint32_t rep_log = ilog2(next_pow2(spec.repetition_factor));
    const uint32_t nv_even = (num_vars & 1) ? num_vars + 1 : num_vars;
    const uint32_t width = 64 + nv_even + 2 * rep_log;
    if (256 < width)
        throw std::logic_error("Cannot fit " + std::to_string(width) + "-bit wide codeword entries in 256 bits integers");
    c.perm_1_seed = transcript.get_u64();
    c.perm_2_seed = transcript.get_u64();
    return c;
}

// ============================================================================ PcsTranscript
void PcsTranscript::write_field_elements(const FieldConfig &f, const Limbs *elems, size_t n) {
    uint8_t be[8 * kMaxLimbs];
    for (size_t i = 0; i < n; i++) {
        fs_transcript.absorb_random_field(f, elems[i]);  // common_field_element
        to_be_bytes(elems[i], f.limbs, be);
        append(be, 8 * f.limbs);
    }
}

size_t PcsTranscript::squeeze_challenge_idx(const FieldConfig &f, size_t cap) {
    const Limbs ch = fs_transcript.get_challenge(f);
    return (size_t)(uint32_t)ch[0] % cap;  // first 4 little-endian bytes of the Montgomery value
}

// ============================================================================ MultilinearZip
static void check(zip_ctx *ctx, int32_t rc, const char *what) {
    if (rc == ZIP_OK) return;
    const std::string detail = ctx ? zip_ctx_last_error(ctx) : "";
    const std::string msg = std::string(what) + ": " + zip_strerror(rc) + (detail.empty() ? "" : " (" + detail + ")");
    if (rc == ZIP_ERR_SHAPE) throw std::logic_error(msg);  // the reference panics (assert_eq!)
    if (rc == ZIP_ERR_INVALID_PARAM) throw ZipError(ZipError::InvalidPcsParam, msg);
    throw ZipError(ZipError::Device, msg);
}

namespace {
struct CtxKey {
    int device;
    uint32_t num_vars, row_len, rep;
    uint64_t seed1, seed2;
    bool operator==(const CtxKey &o) const {
        return device == o.device && num_vars == o.num_vars && row_len == o.row_len && rep == o.rep &&
               seed1 == o.seed1 && seed2 == o.seed2;
    }
};
constexpr size_t kCtxCacheSize = 4;
std::mutex &ctx_cache_mu() {
    static std::mutex m;
    return m;
}
std::list<std::pair<CtxKey, std::shared_ptr<zip_ctx>>> &ctx_cache() {
    static std::list<std::pair<CtxKey, std::shared_ptr<zip_ctx>>> c;
    return c;
}

}  // namespace

MultilinearZipParams MultilinearZip::setup(uint64_t poly_size, const RaaCode &code, int device) {
    if (poly_size == 0 || (poly_size & (poly_size - 1))) throw std::logic_error("assertion failed: poly_size.is_power_of_two()");
    MultilinearZipParams pp;
    pp.num_vars = ilog2(poly_size);
    pp.num_rows = (uint32_t)next_pow2((1ULL << pp.num_vars) / code.row_len);  // structs.rs:82
    pp.linear_code = code;
    // The shim's job: expand the two seeds once (the reference re-runs the shuffle for every row).
    pp.perm1 = shuffle_seeded_perm(code.perm_1_seed, code.codeword_len());
    pp.perm2 = shuffle_seeded_perm(code.perm_2_seed, code.codeword_len());
    // cached device context for this (device, geometry, seeds)
    const CtxKey key{device, pp.num_vars, code.row_len, code.repetition_factor, code.perm_1_seed, code.perm_2_seed};
    {
        std::lock_guard<std::mutex> g(ctx_cache_mu());
        auto &cache = ctx_cache();
        for (auto it = cache.begin(); it != cache.end(); ++it) {
            if (it->first == key) {
                pp.ctx = it->second;
                cache.splice(cache.begin(), cache, it);  // most recently used first
                return pp;
            }
        }
    }
    zip_params zp{};
    zp.num_vars = pp.num_vars;
    zp.row_len = code.row_len;
    zp.num_rows = pp.num_rows;
    zp.codeword_len = code.codeword_len();
    zp.rep = code.repetition_factor;
    zp.n_limbs = 1;
    zp.k_limbs = 4;
    zp.m_limbs = 8;
    zp.perm1 = pp.perm1.data();
    zp.perm2 = pp.perm2.data();
    zp.device = device;
    zip_ctx *ctx = nullptr;
    check(nullptr, zip_ctx_create(&zp, &ctx), "zip_ctx_create");
    pp.ctx = std::shared_ptr<zip_ctx>(ctx, zip_ctx_destroy);
    {
        std::lock_guard<std::mutex> g(ctx_cache_mu());
        auto &cache = ctx_cache();
        cache.emplace_front(key, pp.ctx);
        while (cache.size() > kCtxCacheSize) cache.pop_back();
    }
    return pp;
}

void MultilinearZip::release_cached_contexts() {
    {
        std::lock_guard<std::mutex> g(ctx_cache_mu());
        ctx_cache().clear();
    }
    byte_stream_release_cache();   // and the host blocks kept for the next proof
    zip_release_cached_memory();   // and the device / pinned blocks of the per-proof handles
}

// validate_input (pcs/utils.rs:24-58)
static void validate_input(const char *function, uint32_t param_num_vars, uint32_t poly_num_vars, const size_t *point_len) {
    if (param_num_vars < poly_num_vars)
        throw ZipError(ZipError::InvalidPcsParam, std::string("Too many variates of poly to ") + function +
                                                      " (param supports variates up to " + std::to_string(param_num_vars) +
                                                      " but got " + std::to_string(poly_num_vars) + ")");
    if (point_len && *point_len != poly_num_vars)
        throw ZipError(ZipError::InvalidPcsParam, "Invalid point (expect point to have " + std::to_string(poly_num_vars) +
                                                      " variates but got " + std::to_string(*point_len) + ")");
}

static std::pair<MultilinearZipData, MultilinearZipCommitment> commit_impl(const MultilinearZipParams &pp,
                                                                          const int64_t *evals, size_t n_evals,
                                                                          uint32_t poly_num_vars, bool merkle) {
    validate_input("commit", pp.num_vars, poly_num_vars, nullptr);
    MultilinearZipCommitment comm;
    if (merkle) comm.roots.resize(pp.num_rows);
    zip_commitment *h = nullptr;
    check(pp.ctx.get(),
          zip_commit(pp.ctx.get(), evals, n_evals, ZIP_MEM_HOST, merkle ? 1 : 0,
                     merkle ? reinterpret_cast<uint8_t *>(comm.roots.data()) : nullptr, &h),
          "zip_commit");
    MultilinearZipData data;
    data.ctx = pp.ctx;
    data.handle = std::shared_ptr<zip_commitment>(h, zip_commitment_free);
    return {std::move(data), std::move(comm)};
}

std::pair<MultilinearZipData, MultilinearZipCommitment> MultilinearZip::commit(const MultilinearZipParams &pp,
                                                                               const int64_t *evals, size_t n_evals,
                                                                               uint32_t poly_num_vars) {
    return commit_impl(pp, evals, n_evals, poly_num_vars, true);
}

MultilinearZipData MultilinearZip::commit_no_merkle(const MultilinearZipParams &pp, const int64_t *evals, size_t n_evals,
                                                    uint32_t poly_num_vars) {
    return commit_impl(pp, evals, n_evals, poly_num_vars, false).first;
}

void MultilinearZip::open(const MultilinearZipParams &pp, const int64_t *evals, size_t n_evals, uint32_t poly_num_vars,
                          const MultilinearZipData &commit_data, const Limbs *point, size_t point_len,
                          const FieldConfig &field, PcsTranscript &transcript) {
    validate_input("open", pp.num_vars, poly_num_vars, &point_len);
    const uint32_t row_len = pp.linear_code.row_len, num_rows = pp.num_rows, cw = pp.linear_code.codeword_len();
    if (n_evals != (size_t)row_len * num_rows) throw std::logic_error("evaluations do not fill the matrix");
    zip_ctx *ctx = pp.ctx.get();

    // ---- everything the transcript yields BEFORE the first absorb (write_integers and
    // write_merkle_proof never absorb: pcs_transcript.rs:115-135,198-211), in the reference's order
    std::vector<int64_t> coeffs;
    if (num_rows > 1) {  // prove_testing_phase, open_z.rs:100-113 (num_proximity_testing == 1 on this path)
        if (pp.linear_code.num_proximity_testing != 1)
            throw ZipError(ZipError::InvalidPcsParam, "only one proximity test is supported on the device path");
        coeffs = transcript.fs_transcript.get_integer_challenges_i64(num_rows);
    }
    std::vector<uint32_t> cols(pp.linear_code.num_column_opening);
    for (auto &c : cols) c = (uint32_t)transcript.squeeze_challenge_idx(field, cw);  // open_z.rs:116-120
    // left_point_to_tensor (pcs/utils.rs:279-292): eq over the LAST log2(num_rows) coordinates
    std::vector<uint64_t> q0;
    if (num_rows > 1) {
        const uint32_t lr = ilog2(num_rows);
        const auto eq = build_eq_x_r(field, point + (point_len - lr), lr);
        q0.resize((size_t)num_rows * field.limbs);
        for (uint32_t r = 0; r < num_rows; r++)
            for (uint32_t k = 0; k < field.limbs; k++) q0[(size_t)r * field.limbs + k] = eq[r][k];
    }
    // ---- one device call produces the whole stream of open() ----
    const zip_field zf = field.to_abi();
    const size_t len = zip_proof_len(ctx, (uint32_t)cols.size(), field.limbs);
    const size_t at = transcript.stream.size();
    transcript.stream.resize(at + len);  // uninitialised (ByteStream); the library's copy threads touch the pages
    check(ctx,
          zip_open(commit_data.handle.get(), evals, ZIP_MEM_HOST, coeffs.empty() ? nullptr : coeffs.data(), cols.data(),
                   (uint32_t)cols.size(), q0.empty() ? nullptr : q0.data(), &zf, transcript.stream.data() + at,
                   ZIP_MEM_HOST),
          "zip_open");
    // ---- write_field_elements absorbs each element of the evaluation row (pcs_transcript.rs:107-113);
    // the bytes are already in the stream, only the Fiat-Shamir state still has to follow
    const uint8_t *row_be = transcript.stream.data() + at + len - (size_t)row_len * field.limbs * 8;
    for (uint32_t c = 0; c < row_len; c++) {
        Limbs v{};
        const uint8_t *b = row_be + (size_t)c * field.limbs * 8;
        for (uint32_t i = 0; i < field.limbs; i++) {
            uint64_t w = 0;
            for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (field.limbs - 1 - i) + k];
            v[i] = w;
        }
        transcript.fs_transcript.absorb_random_field(field, v);
    }
}

// point_to_tensor (pcs/utils.rs:252-276): q_0 over the last log2(num_rows) coordinates, q_1 over the rest;
// an empty half gives an EMPTY vector (MLE::zero()).
static void point_to_tensor(const FieldConfig &field, uint32_t num_rows, const Limbs *point, size_t point_len,
                            std::vector<uint64_t> &q0, std::vector<uint64_t> &q1) {
    const uint32_t lr = ilog2(num_rows);
    const size_t lc = point_len - lr;
    auto flat = [&](const std::vector<Limbs> &eq, std::vector<uint64_t> &out) {
        out.resize(eq.size() * field.limbs);
        for (size_t i = 0; i < eq.size(); i++)
            for (uint32_t k = 0; k < field.limbs; k++) out[i * field.limbs + k] = eq[i][k];
    };
    q0.clear();
    q1.clear();
    if (lr) flat(build_eq_x_r(field, point + lc, lr), q0);
    if (lc) flat(build_eq_x_r(field, point, (uint32_t)lc), q1);
}

void MultilinearZip::verify(const MultilinearZipParams &vp, const MultilinearZipCommitment &comm, const Limbs *point,
                            size_t point_len, const Limbs &eval, PcsTranscript &transcript, const FieldConfig &field) {
    validate_input("verify", vp.num_vars, vp.num_vars, &point_len);
    const uint32_t row_len = vp.linear_code.row_len, num_rows = vp.num_rows, cw = vp.linear_code.codeword_len();
    if (comm.roots.size() != num_rows) throw ZipError(ZipError::InvalidPcsOpen, "commitment has the wrong number of roots");
    zip_ctx *ctx = vp.ctx.get();
    // the challenges in the order verify_testing draws them (verify_z.rs:69-90); reads never absorb
    std::vector<int64_t> coeffs;
    if (num_rows > 1) {
        if (vp.linear_code.num_proximity_testing != 1)
            throw ZipError(ZipError::InvalidPcsParam, "only one proximity test is supported on the device path");
        coeffs = transcript.fs_transcript.get_integer_challenges_i64(num_rows);
    }
    std::vector<uint32_t> cols(vp.linear_code.num_column_opening);
    for (auto &c : cols) c = (uint32_t)transcript.squeeze_challenge_idx(field, cw);
    std::vector<uint64_t> q0, q1;
    point_to_tensor(field, num_rows, point, point_len, q0, q1);
    const zip_field zf = field.to_abi();
    const size_t len = zip_proof_len(ctx, (uint32_t)cols.size(), field.limbs);
    const size_t avail = transcript.read_size() - transcript.read_pos;
    zip_verify_report rep{};
    check(ctx,
          zip_verify(ctx, comm.roots.empty() ? nullptr : comm.roots[0].data(), transcript.read_data() + transcript.read_pos,
                     ZIP_MEM_HOST, avail, coeffs.empty() ? nullptr : coeffs.data(), cols.data(), (uint32_t)cols.size(),
                     q0.empty() ? nullptr : q0.data(), q1.empty() ? nullptr : q1.data(), eval.data(), &zf, &rep),
          "zip_verify");
    switch (rep.verdict) {
        case ZIP_VERIFY_ACCEPT: break;
        case ZIP_VERIFY_PROXIMITY_TESTING:
        case ZIP_VERIFY_PROXIMITY_Q0: throw ZipError(ZipError::InvalidPcsOpen, "Proximity failure");
        case ZIP_VERIFY_EVAL_CONSISTENCY: throw ZipError(ZipError::InvalidPcsOpen, "Evaluation consistency failure");
        case ZIP_VERIFY_MERKLE: throw ZipError(ZipError::InvalidPcsOpen, "Merkle proof verification failed");
        case ZIP_VERIFY_OVERFLOW: throw std::logic_error("attempt to add with overflow (encode_wide of the combined row)");
        default: throw ZipError(ZipError::Transcript, "Failed to read the proof stream");
    }
    // read_field_elements absorbed every element of the evaluation row (pcs_transcript.rs:138-160)
    const uint8_t *row_be = transcript.read_data() + transcript.read_pos + len - (size_t)row_len * field.limbs * 8;
    for (uint32_t c = 0; c < row_len; c++) {
        Limbs v{};
        const uint8_t *b = row_be + (size_t)c * field.limbs * 8;
        for (uint32_t i = 0; i < field.limbs; i++) {
            uint64_t w = 0;
            for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (field.limbs - 1 - i) + k];
            v[i] = w;
        }
        transcript.fs_transcript.absorb_random_field(field, v);
    }
    transcript.read_pos += len;
}

Limbs MultilinearZip::evaluate(const MultilinearZipParams &pp, const int64_t *evals, size_t n_evals, const Limbs *point,
                               size_t point_len, const FieldConfig &field) {
    if (point_len != pp.num_vars)
        throw ZipError(ZipError::InvalidPcsParam, "IncorrectLength: the point does not have num_vars coordinates");
    if (n_evals != (size_t)pp.linear_code.row_len * pp.num_rows) throw std::logic_error("evaluations do not fill the matrix");
    std::vector<uint64_t> q0, q1;
    point_to_tensor(field, pp.num_rows, point, point_len, q0, q1);
    const zip_field zf = field.to_abi();
    Limbs v{};
    check(pp.ctx.get(),
          zip_mle_eval(pp.ctx.get(), evals, ZIP_MEM_HOST, q0.empty() ? nullptr : q0.data(), q1.empty() ? nullptr : q1.data(),
                       &zf, v.data()),
          "zip_mle_eval");
    return v;
}

}  // namespace zip

namespace {
// ZINC_HOST_TIMING=1: stage times on stderr (tools/zinc_prover_times.py, tools/prover_pcs_step.py)
struct PcsStageTimer {
    bool on = std::getenv("ZINC_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[zinc]   pcs %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
}  // namespace

zip::ZipProof zip::commit_z_mle_and_prove_evaluation(const LinearCodeSpec &lc_spec, const int64_t *z_evals, size_t m,
                                                     const Limbs *r_y, size_t r_y_len, KeccakTranscript &transcript,
                                                     const FieldConfig &config, int device) {
    PcsStageTimer timer;
    KeccakSeedSource seeds(transcript);
    const RaaCode linear_code = RaaCode::make(lc_spec, m, seeds);               // prover.rs:313
    const MultilinearZipParams param = MultilinearZip::setup(m, linear_code, device);  // :314
    timer.lap("RaaCode::new + setup");
    auto committed = MultilinearZip::commit(param, z_evals, m, param.num_vars);  // :315
    timer.lap("commit");
    PcsTranscript pcs_transcript;                                                // :316 (fresh)
    ZipProof out;
    {   // :317-319, z_mle.map_to_field(config).evaluate(r_y): over the witness the commit left on the device
        if (r_y_len != param.num_vars)
            throw ZipError(ZipError::InvalidPcsParam, "IncorrectLength: the point does not have num_vars coordinates");
        std::vector<uint64_t> q0, q1;
        point_to_tensor(config, param.num_rows, r_y, r_y_len, q0, q1);
        const zip_field zf = config.to_abi();
        check(param.ctx.get(),
              zip_commitment_mle_eval(committed.first.handle.get(), q0.empty() ? nullptr : q0.data(),
                                      q1.empty() ? nullptr : q1.data(), &zf, out.v.data()),
              "zip_commitment_mle_eval");
    }
    timer.lap("evaluate");
    MultilinearZip::open(param, z_evals, m, param.num_vars, committed.first, r_y, r_y_len, config, pcs_transcript);  // :320
    timer.lap("open");
    out.z_comm = std::move(committed.second);
    out.pcs_proof = pcs_transcript.into_proof();
    return out;
}

namespace {
sumcheck::ProverOutput prove_as_subprotocol_impl(KeccakTranscript &transcript, const std::vector<const uint64_t *> &mles,
                                                 uint32_t nvars, uint32_t degree, const zip_sumcheck_comb *comb,
                                                 const FieldConfig &config, int device,
                                                 zip_mem_kind kind = ZIP_MEM_HOST) {
    // sumcheck.rs:64-76 (FIELD_LIMBS > 1: the u128 map)
    transcript.absorb_random_field(config, map_to_field_u128(config, nvars, 0));
    transcript.absorb_random_field(config, map_to_field_u128(config, degree, 0));
    sumcheck::ProverOutput out;
    if (nvars == 0) return out;  // :77-92: empty proof
    const zip_field zf = config.to_abi();
    zip_sumcheck *raw = nullptr;
    int32_t rc = zip_sumcheck_init(device, mles.data(), kind, (uint32_t)mles.size(), nvars, degree, comb, &zf, &raw);
    if (rc) throw ZipError(rc == ZIP_ERR_INVALID_PARAM ? ZipError::InvalidPcsParam : ZipError::Device,
                           std::string("zip_sumcheck_init: ") + zip_strerror(rc));
    std::unique_ptr<zip_sumcheck, void (*)(zip_sumcheck *)> s(raw, zip_sumcheck_free);
    std::vector<uint64_t> evals((size_t)(degree + 1) * config.limbs);
    Limbs r{};
    for (uint32_t round = 0; round < nvars; round++) {  // :97-106
        rc = zip_sumcheck_round(s.get(), round ? r.data() : nullptr, evals.data());
        if (rc) throw ZipError(ZipError::Device, std::string("zip_sumcheck_round: ") + zip_sumcheck_last_error(s.get()));
        std::vector<Limbs> msg(degree + 1);
        for (uint32_t e = 0; e <= degree; e++) {
            for (uint32_t i = 0; i < config.limbs; i++) msg[e][i] = evals[(size_t)e * config.limbs + i];
            transcript.absorb_random_field(config, msg[e]);  // absorb_slice
        }
        out.proof.msgs.push_back(std::move(msg));
        r = transcript.get_challenge(config);   // sample_round
        transcript.absorb_random_field(config, r);
        out.randomness.push_back(r);
    }
    return out;
}
}  // namespace

sumcheck::ProverOutput sumcheck::prove_as_subprotocol_product(KeccakTranscript &transcript,
                                                              const std::vector<const uint64_t *> &mles, uint32_t nvars,
                                                              uint32_t degree, const FieldConfig &config, int device) {
    return prove_as_subprotocol_impl(transcript, mles, nvars, degree, nullptr, config, device);
}

namespace {
zip_sumcheck_comb make_comb(size_t n_mles, const std::vector<Limbs> &c, const std::vector<std::vector<uint32_t>> &S,
                            const FieldConfig &config) {
    if (c.size() != S.size()) throw std::logic_error("ccs.c and ccs.S differ in length");
    zip_sumcheck_comb comb{};
    for (size_t t = 0; t < c.size(); t++) {
        bool zero = true;  // terms with a zero coefficient are skipped (zinc/utils.rs:80-82)
        for (uint32_t i = 0; i < config.limbs; i++) zero &= c[t][i] == 0;
        if (zero) continue;
        if (comb.n_terms == 8) throw ZipError(ZipError::InvalidPcsParam, "more than 8 non-zero CCS terms");
        for (uint32_t j : S[t]) {
            if (j >= n_mles) throw std::logic_error("index out of bounds: ccs.S refers to a missing MLE");
            comb.term_mask[comb.n_terms] |= 1u << j;
        }
        for (uint32_t i = 0; i < config.limbs; i++) comb.coeff[comb.n_terms][i] = c[t][i];
        comb.n_terms++;
    }
    if (comb.n_terms == 0) throw ZipError(ZipError::InvalidPcsParam, "no non-zero CCS term");
    return comb;
}
}  // namespace

sumcheck::ProverOutput sumcheck::prove_as_subprotocol_ccs(KeccakTranscript &transcript,
                                                          const std::vector<const uint64_t *> &mles, uint32_t nvars,
                                                          uint32_t degree, const std::vector<Limbs> &c,
                                                          const std::vector<std::vector<uint32_t>> &S,
                                                          const FieldConfig &config, int device) {
    const zip_sumcheck_comb comb = make_comb(mles.size(), c, S, config);
    return prove_as_subprotocol_impl(transcript, mles, nvars, degree, &comb, config, device);
}

sumcheck::ProverOutput sumcheck::prove_as_subprotocol_products(
    KeccakTranscript &transcript, const std::vector<const uint64_t *> &mles, uint32_t nvars, uint32_t degree,
    const std::vector<std::pair<Limbs, std::vector<uint32_t>>> &products, const FieldConfig &config, int device) {
    transcript.absorb_random_field(config, map_to_field_u128(config, nvars, 0));  // sumcheck.rs:64-76
    transcript.absorb_random_field(config, map_to_field_u128(config, degree, 0));
    ProverOutput out;
    if (nvars == 0) return out;
    if (products.empty()) {  // comb_fn == 0 (sumcheck/tests.rs:525-557): every round polynomial is the zero constant
        for (uint32_t round = 0; round < nvars; round++) {
            std::vector<Limbs> msg(degree + 1);
            for (const Limbs &e : msg) transcript.absorb_random_field(config, e);
            out.proof.msgs.push_back(std::move(msg));
            const Limbs r = transcript.get_challenge(config);
            transcript.absorb_random_field(config, r);
            out.randomness.push_back(r);
        }
        return out;
    }
    const zip_field zf = config.to_abi();
    struct Free {
        void operator()(zip_sumcheck *p) const { zip_sumcheck_free(p); }
    };
    std::vector<std::unique_ptr<zip_sumcheck, Free>> provers;
    for (const auto &[coeff, indices] : products) {
        if (indices.empty() || indices.size() > 4)
            throw ZipError(ZipError::InvalidPcsParam, "a product needs 1..4 multiplicands");
        std::vector<const uint64_t *> tables;
        for (uint32_t j : indices) {
            if (j >= mles.size()) throw std::logic_error("index out of bounds: a product refers to a missing MLE");
            tables.push_back(mles[j]);
        }
        zip_sumcheck_comb comb{};  // coeff * prod_{j < K-1} vals[j] * vals[K-1]
        comb.n_terms = 1;
        comb.term_mask[0] = (1u << (tables.size() - 1)) - 1u;
        for (uint32_t i = 0; i < config.limbs; i++) comb.coeff[0][i] = coeff[i];
        zip_sumcheck *raw = nullptr;
        const int32_t rc = zip_sumcheck_init(device, tables.data(), ZIP_MEM_HOST, (uint32_t)tables.size(), nvars, degree, &comb, &zf, &raw);
        if (rc) throw ZipError(rc == ZIP_ERR_INVALID_PARAM ? ZipError::InvalidPcsParam : ZipError::Device,
                               std::string("zip_sumcheck_init: ") + zip_strerror(rc));
        provers.emplace_back(raw);
    }
    std::vector<uint64_t> evals((size_t)(degree + 1) * config.limbs);
    Limbs r{};
    for (uint32_t round = 0; round < nvars; round++) {
        for (auto &p : provers)
            if (zip_sumcheck_round_begin(p.get(), round ? r.data() : nullptr))
                throw ZipError(ZipError::Device, std::string("zip_sumcheck_round_begin: ") + zip_sumcheck_last_error(p.get()));
        std::vector<Limbs> msg(degree + 1);
        for (auto &p : provers) {
            if (zip_sumcheck_round_end(p.get(), evals.data()))
                throw ZipError(ZipError::Device, std::string("zip_sumcheck_round_end: ") + zip_sumcheck_last_error(p.get()));
            for (uint32_t e = 0; e <= degree; e++) {
                Limbs t{};
                for (uint32_t i = 0; i < config.limbs; i++) t[i] = evals[(size_t)e * config.limbs + i];
                config.add_assign(msg[e], t);
            }
        }
        for (uint32_t e = 0; e <= degree; e++) transcript.absorb_random_field(config, msg[e]);
        out.proof.msgs.push_back(std::move(msg));
        r = transcript.get_challenge(config);
        transcript.absorb_random_field(config, r);
        out.randomness.push_back(r);
    }
    return out;
}

// ---------------------------------------------------------------------------- CCS
ccs::SparseMatrix ccs::SparseMatrix::from_coeffs(uint32_t n_rows, uint32_t n_cols,
                                                 const std::vector<std::vector<std::pair<int64_t, uint32_t>>> &coeffs) {
    if (coeffs.size() > n_rows) throw std::logic_error("more coefficient rows than n_rows");
    SparseMatrix M;
    M.n_rows = n_rows;
    M.n_cols = n_cols;
    M.row_ptr.assign(1, 0);
    for (uint32_t r = 0; r < n_rows; r++) {
        if (r < coeffs.size())
            for (const auto &[v, col] : coeffs[r]) {
                M.values.push_back(v);
                M.col_idx.push_back(col);
            }
        M.row_ptr.push_back((uint32_t)M.col_idx.size());
    }
    return M;
}

std::vector<int64_t> ccs::Statement_Z::get_z_vector(const std::vector<int64_t> &w) const {
    std::vector<int64_t> z;
    z.reserve(public_input.size() + w.size() + 1);
    z.insert(z.end(), public_input.begin(), public_input.end());
    z.push_back(1);
    z.insert(z.end(), w.begin(), w.end());
    return z;
}

// ---------------------------------------------------------------------------- ZincProver
IntVec ZincProver::get_z_ccs(const int64_t *x, size_t l, const int64_t *w, size_t w_len, size_t m) {
    const size_t len = l + 1 + w_len;
    IntVec z(len <= m ? m : len);  // prover.rs:230-232: resize(ccs.m) only when not longer
    if (l) std::memcpy(z.data(), x, l * 8);
    z[l] = 1;
    if (w_len) std::memcpy(z.data() + l + 1, w, w_len * 8);
    if (z.size() > len) std::memset(z.data() + len, 0, (z.size() - len) * 8);
    return z;
}

namespace {
// ZINC_HOST_TIMING=1: stage times of the prover on stderr (tools/zinc_prover_times.py)
struct StageTimer {
    bool on = std::getenv("ZINC_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[zinc] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
void ccs_check(zip_ccs *h, int32_t rc, const char *what) {
    if (rc == ZIP_OK) return;
    const std::string msg = std::string(what) + ": " + (h ? zip_ccs_last_error(h) : zip_strerror(rc));
    if (rc == ZIP_ERR_SHAPE) throw std::logic_error(msg);  // the reference panics / LengthsNotEqual
    throw ZipError(rc == ZIP_ERR_INVALID_PARAM ? ZipError::InvalidPcsParam : ZipError::Device, msg);
}
const uint64_t *ccs_table(zip_ccs *h, zip_ccs_table_kind which, uint32_t index) {
    const uint64_t *p = nullptr;
    ccs_check(h, zip_ccs_table(h, which, index, &p), "zip_ccs_table");
    return p;
}
}  // namespace

PreparedCcs::PreparedCcs(const ccs::Statement_Z &statement, const ccs::CCS_Z &ccs, const FieldConfig &config, int device)
    : t_((uint32_t)ccs.t), s_((uint32_t)ccs.s), limbs_(config.limbs), modulus_(config.modulus), device_(device) {
    if (ccs.s == 0 || ccs.s > 28 || ccs.m != ((size_t)1 << ccs.s) || ccs.n != ccs.m || ccs.s_prime != ccs.s)
        throw std::logic_error("assertion failed: rx.len() == num_rows (compute_eval_table_sparse): m == n == 2^s is required");
    if (statement.constraints.size() != ccs.t) throw std::logic_error("CCS sizes and the statement disagree");
    const zip_field zf = config.to_abi();
    std::vector<zip_sparse_matrix> mats;
    for (const auto &M : statement.constraints) mats.push_back(M.to_abi());
    ccs_check(nullptr, zip_ccs_create(device, mats.data(), t_, s_, &zf, &h_), "zip_ccs_create");
}
PreparedCcs::PreparedCcs(const zip_sparse_matrix *matrices, uint32_t t, uint32_t s, const FieldConfig &config, int device)
    : t_(t), s_(s), limbs_(config.limbs), modulus_(config.modulus), device_(device) {
    const zip_field zf = config.to_abi();
    ccs_check(nullptr, zip_ccs_create(device, matrices, t, s, &zf, &h_), "zip_ccs_create");
}
PreparedCcs::~PreparedCcs() { zip_ccs_free(h_); }

std::pair<SpartanProof, std::vector<Limbs>> ZincProver::spartan_prove(const ccs::Statement_Z &statement,
                                                                      const int64_t *z_ccs, size_t z_len,
                                                                      KeccakTranscript &transcript, const ccs::CCS_Z &ccs,
                                                                      const FieldConfig &config, PreparedCcs *prepared) const {
    // the shape the reference's prover supports (see the header)
    if (ccs.s == 0 || ccs.s > 28 || ccs.m != ((size_t)1 << ccs.s) || ccs.n != ccs.m || ccs.s_prime != ccs.s)
        throw std::logic_error("assertion failed: rx.len() == num_rows (compute_eval_table_sparse): m == n == 2^s is required");
    if (statement.constraints.size() != ccs.t || ccs.S.size() != ccs.q || ccs.c.size() != ccs.q)
        throw std::logic_error("CCS sizes and the statement disagree");
    if (z_len > ccs.m) throw std::logic_error("LengthsNotEqual: M.n_cols != z.len()");
    std::vector<std::vector<uint32_t>> S(ccs.q);
    {
        size_t pos = 0;
        for (size_t i = 0; i < ccs.q; i++) {
            if (ccs.c[i] == 0) throw ZipError(ZipError::InvalidPcsParam, "a zero CCS coefficient shifts the MLE list the combination function indexes");
            for (size_t j : ccs.S[i]) {
                if (j != pos++) throw ZipError(ZipError::InvalidPcsParam, "ccs.S must list the matrices 0..t-1 in order");
                S[i].push_back((uint32_t)j);
            }
        }
        if (pos != ccs.t) throw ZipError(ZipError::InvalidPcsParam, "ccs.S must list the matrices 0..t-1 in order");
    }
    const uint32_t s = (uint32_t)ccs.s, t = (uint32_t)ccs.t;
    StageTimer timer;
    std::unique_ptr<PreparedCcs> own;
    if (!prepared) {  // prepare_for_random_field_piop: ccs.map_to_field / statement.map_to_field
        own = std::make_unique<PreparedCcs>(statement, ccs, config, device_);
        prepared = own.get();
        timer.lap("PreparedCcs (zip_ccs_create)");
    } else if (prepared->t_ != t || prepared->s_ != s || prepared->limbs_ != config.limbs ||
               prepared->modulus_ != config.modulus || prepared->device_ != device_) {
        throw std::logic_error("the prepared CCS belongs to another circuit, field or device");
    }
    std::lock_guard<std::mutex> one_proof(prepared->mu_);
    zip_ccs *dev = prepared->h_;
    // z_ccs -> F_q (prover.rs:236); calculate_Mz_mles
    ccs_check(dev, zip_ccs_set_z(dev, z_ccs, z_len, ZIP_MEM_HOST), "zip_ccs_set_z");
    timer.lap("zip_ccs_set_z");

    SpartanProof proof;
    // ---- sumcheck_1 (prover.rs:242-259)
    transcript.absorb(reinterpret_cast<const uint8_t *>("beta_s"), 6);  // squeeze_beta_challenges, zinc/utils.rs:100-106
    std::vector<uint64_t> flat((size_t)s * config.limbs);
    for (uint32_t i = 0; i < s; i++) {
        const Limbs b = transcript.get_challenge(config);
        std::copy(b.begin(), b.begin() + config.limbs, flat.begin() + (size_t)i * config.limbs);
    }
    ccs_check(dev, zip_ccs_eq_table(dev, flat.data(), 0), "zip_ccs_eq_table");
    std::vector<const uint64_t *> g;  // prepare_lin_sumcheck_polynomial: [Mz_0 .. Mz_{t-1}, eq(beta)], degree d + 1
    for (uint32_t k = 0; k < t; k++) g.push_back(ccs_table(dev, ZIP_CCS_MZ, k));
    g.push_back(ccs_table(dev, ZIP_CCS_EQ, 0));
    std::vector<Limbs> c_f;
    for (int64_t c : ccs.c) c_f.push_back(map_to_field(config, c));  // CCS_Z::map_to_field, ccs_z.rs:147
    const zip_sumcheck_comb comb = make_comb(g.size(), c_f, S, config);
    auto sc1 = prove_as_subprotocol_impl(transcript, g, s, (uint32_t)ccs.d + 1, &comb, config, device_, ZIP_MEM_DEVICE);
    proof.linearization_sumcheck = std::move(sc1.proof);
    timer.lap("sumcheck_1");
    const std::vector<Limbs> &r_x = sc1.randomness;

    // ---- sumcheck_2 (prover.rs:261-303)
    transcript.absorb(reinterpret_cast<const uint8_t *>("gamma"), 5);  // squeeze_gamma_challenge, zinc/utils.rs:112-118
    const Limbs gamma = transcript.get_challenge(config);
    for (uint32_t i = 0; i < s; i++) std::copy(r_x[i].begin(), r_x[i].begin() + config.limbs, flat.begin() + (size_t)i * config.limbs);
    std::vector<uint64_t> vs((size_t)t * config.limbs);
    ccs_check(dev, zip_ccs_second_table(dev, flat.data(), gamma.data(), vs.data()), "zip_ccs_second_table");
    timer.lap("zip_ccs_second_table");
    const std::vector<const uint64_t *> two{ccs_table(dev, ZIP_CCS_SECOND, 0), ccs_table(dev, ZIP_CCS_Z_FIELD, 0)};
    auto sc2 = prove_as_subprotocol_impl(transcript, two, s, 2, nullptr, config, device_, ZIP_MEM_DEVICE);
    proof.second_sumcheck = std::move(sc2.proof);
    timer.lap("sumcheck_2");

    // ---- calculate_V_s (prover.rs:330-347), computed with the second table
    for (uint32_t k = 0; k < t; k++) {
        Limbs v{};
        for (uint32_t i = 0; i < config.limbs; i++) v[i] = vs[(size_t)k * config.limbs + i];
        proof.V_s.push_back(v);
    }
    return {std::move(proof), std::move(sc2.randomness)};
}

ZincProof ZincProver::prove(const ccs::Statement_Z &statement, const ccs::Witness_Z &wit, KeccakTranscript &transcript,
                            const ccs::CCS_Z &ccs, const FieldConfig &config, std::vector<Limbs> *r_y_out,
                            PreparedCcs *prepared) const {
    const IntVec z_ccs = get_z_ccs(statement, wit, ccs);
    return prove_z(statement, z_ccs.data(), z_ccs.size(), transcript, ccs, config, r_y_out, prepared);
}

ZincProof ZincProver::prove_z(const ccs::Statement_Z &statement, const int64_t *z_ccs, size_t z_len, KeccakTranscript &transcript,
                              const ccs::CCS_Z &ccs, const FieldConfig &config, std::vector<Limbs> *r_y_out,
                              PreparedCcs *prepared) const {
    auto [spartan_proof, r_y] = spartan_prove(statement, z_ccs, z_len, transcript, ccs, config, prepared);
    // commit_z_mle_and_prove_evaluation (prover.rs:305-327); z_mle = from_evaluations_slice(s_prime, z_ccs)
    const size_t n_mle = (size_t)1 << ccs.s_prime;
    IntVec padded;
    const int64_t *z_mle = z_ccs;
    if (z_len != n_mle) {
        padded.resize(n_mle);
        std::memcpy(padded.data(), z_ccs, std::min(z_len, n_mle) * 8);
        if (z_len < n_mle) std::memset(padded.data() + z_len, 0, (n_mle - z_len) * 8);
        z_mle = padded.data();
    }
    ZincProof out;
    out.spartan_proof = std::move(spartan_proof);
    StageTimer timer;
    out.zip_proof = zip::commit_z_mle_and_prove_evaluation(lc_spec_, z_mle, ccs.m, r_y.data(), r_y.size(), transcript,
                                                           config, device_);
    timer.lap("commit_z_mle_and_prove_eval");
    if (r_y_out) *r_y_out = std::move(r_y);
    return out;
}

// ---------------------------------------------------------------------------- sumcheck verifier
Limbs sumcheck::interpolate_uni_poly(const FieldConfig &config, const std::vector<Limbs> &p_i, const Limbs &x) {
    const size_t len = p_i.size();
    std::vector<Limbs> evals{x};
    Limbs prod = x, j{};
    for (size_t i = 1; i < len; i++) {  // verifier.rs:176-185: early return when x is one of the nodes
        if (x == j) return p_i[i - 1];
        config.add_assign(j, config.r);
        Limbs tmp = x;
        config.sub_assign(tmp, j);
        evals.push_back(tmp);
        config.mul_assign(prod, tmp);
    }
    if (x == j) return p_i[len - 1];
    // sum_i p_i[i] * prod / ((x - i) * prod_{k != i} (i - k)): the reference only organises the denominators
    // so that it needs fewer divisions (:217-300); the value is that of the unique interpolant either way
    Limbs res{};
    for (size_t i = 0; i < len; i++) {
        Limbs den = config.r;
        for (size_t k = 0; k < len; k++)
            if (k != i) config.mul_assign(den, map_to_field(config, (int64_t)i - (int64_t)k));
        config.mul_assign(den, evals[i]);
        Limbs term = config.inverse(den);
        config.mul_assign(term, prod);
        config.mul_assign(term, p_i[i]);
        config.add_assign(res, term);
    }
    return res;
}

sumcheck::SubClaim sumcheck::verify_as_subprotocol(KeccakTranscript &transcript, uint32_t num_vars, uint32_t degree,
                                                   const Limbs &claimed_sum, const SumcheckProof &proof,
                                                   const FieldConfig &config) {
    transcript.absorb_random_field(config, map_to_field_u128(config, num_vars, 0));  // sumcheck.rs:124-136
    transcript.absorb_random_field(config, map_to_field_u128(config, degree, 0));
    SubClaim claim;
    if (num_vars == 0) {  // :138-144
        transcript.absorb_random_field(config, claimed_sum);
        claim.expected_evaluation = claimed_sum;
        return claim;
    }
    if (proof.msgs.size() != num_vars)
        throw SpartanError(SpartanError::InvalidProofLength, "sumcheck proof has " + std::to_string(proof.msgs.size()) +
                                                                 " rounds, expected " + std::to_string(num_vars));
    for (uint32_t i = 0; i < num_vars; i++) {  // :155-160, verify_round (verifier.rs:61-90)
        for (const Limbs &e : proof.msgs[i]) transcript.absorb_random_field(config, e);
        const Limbs r = transcript.get_challenge(config);
        transcript.absorb_random_field(config, r);
        claim.point.push_back(r);
    }
    Limbs expected = claimed_sum;  // check_and_generate_subclaim (verifier.rs:97-143)
    for (uint32_t i = 0; i < num_vars; i++) {
        const std::vector<Limbs> &ev = proof.msgs[i];
        if (ev.size() != (size_t)degree + 1) throw SpartanError(SpartanError::MaxDegreeExceeded, "MaxDegreeExceeded");
        Limbs sum = ev[0];
        if (degree > 0) config.add_assign(sum, ev[1]);
        if (sum != expected) throw SpartanError(SpartanError::SumCheckFailed, "sumcheck round " + std::to_string(i) + ": p(0) + p(1) != expected");
        expected = interpolate_uni_poly(config, ev, claim.point[i]);
    }
    claim.expected_evaluation = expected;
    return claim;
}

// ---------------------------------------------------------------------------- ZincVerifier
namespace {
Limbs lin_comb_V_s(const FieldConfig &config, const Limbs &gamma, const std::vector<Limbs> &V) {  // verifier.rs:212-219
    Limbs res{};
    for (size_t i = V.size(); i-- > 0;) {
        config.mul_assign(res, gamma);
        config.add_assign(res, V[i]);
    }
    return res;
}
}  // namespace

VerificationPoints ZincVerifier::spartan_verify(const SpartanProof &proof, const ccs::CCS_Z &ccs, KeccakTranscript &transcript,
                                                const FieldConfig &config) const {
    if (proof.V_s.size() != ccs.t || ccs.S.size() != ccs.q || ccs.c.size() != ccs.q)
        throw std::logic_error("index out of bounds: V_s / ccs.S / ccs.c sizes");  // the reference indexes V_s[j]
    transcript.absorb(reinterpret_cast<const uint8_t *>("beta_s"), 6);
    std::vector<Limbs> beta_s;
    for (size_t i = 0; i < ccs.s; i++) beta_s.push_back(transcript.get_challenge(config));
    // verify_linearization_proof (:142-162): degree d + 1, claimed sum 0
    const sumcheck::SubClaim first = sumcheck::verify_as_subprotocol(transcript, (uint32_t)ccs.s, (uint32_t)ccs.d + 1, Limbs{},
                                                                    proof.linearization_sumcheck, config);
    // verify_linearization_claim (:164-187)
    Limbs e = config.r;
    for (size_t i = 0; i < ccs.s; i++) {  // eq_eval, sumcheck/utils.rs:81-95
        Limbs xy = first.point[i];
        config.mul_assign(xy, beta_s[i]);
        Limbs term = xy;
        config.add_assign(term, xy);
        config.sub_assign(term, first.point[i]);
        config.sub_assign(term, beta_s[i]);
        config.add_assign(term, config.r);
        config.mul_assign(e, term);
    }
    Limbs sum{};
    for (size_t i = 0; i < ccs.q; i++) {
        Limbs term = map_to_field(config, ccs.c[i]);
        for (size_t j : ccs.S[i]) {
            if (j >= proof.V_s.size()) throw std::logic_error("index out of bounds: ccs.S refers to a missing V_s");
            config.mul_assign(term, proof.V_s[j]);
        }
        config.add_assign(sum, term);
    }
    config.mul_assign(e, sum);
    if (e != first.expected_evaluation) throw SpartanError(SpartanError::SumCheckFailed, "linearization claim: e * sum c_i prod V_s != s");
    VerificationPoints out;
    transcript.absorb(reinterpret_cast<const uint8_t *>("gamma"), 5);
    out.gamma = transcript.get_challenge(config);
    const Limbs claimed = lin_comb_V_s(config, out.gamma, proof.V_s);
    const sumcheck::SubClaim second = sumcheck::verify_as_subprotocol(transcript, (uint32_t)ccs.s_prime, 2, claimed,
                                                                     proof.second_sumcheck, config);  // :189-210
    out.rx_ry = first.point;
    out.rx_ry.insert(out.rx_ry.end(), second.point.begin(), second.point.end());
    out.e_y = second.expected_evaluation;
    return out;
}

void ZincVerifier::verify_pcs_proof(const ccs::Statement_Z &statement, const zip::MultilinearZipCommitment &z_comm,
                                    const Limbs &v, const uint8_t *pcs_proof, size_t pcs_proof_len,
                                    const VerificationPoints &points, const ccs::CCS_Z &ccs, KeccakTranscript &transcript,
                                    const FieldConfig &config, PreparedCcs *prepared) const {
    if (points.rx_ry.size() != ccs.s + ccs.s_prime) throw std::logic_error("range end index out of range for rx_ry");
    zip::KeccakSeedSource seeds(transcript);
    const zip::RaaCode linear_code = zip::RaaCode::make(lc_spec_, ccs.m, seeds);                  // :234
    const zip::MultilinearZipParams param = zip::MultilinearZip::setup(ccs.m, linear_code, device_);  // :235
    zip::PcsTranscript pcs_transcript = zip::PcsTranscript::from_proof_view(pcs_proof, pcs_proof_len);  // :236
    const Limbs *r_y = points.rx_ry.data() + ccs.s;
    zip::MultilinearZip::verify(param, z_comm, r_y, ccs.s_prime, v, pcs_transcript, config);  // :239-246
    // V_xy (:248-261) on the device
    std::unique_ptr<PreparedCcs> own;
    if (!prepared) {
        own = std::make_unique<PreparedCcs>(statement, ccs, config, device_);
        prepared = own.get();
    }
    std::vector<uint64_t> rx((size_t)ccs.s * config.limbs), ry((size_t)ccs.s_prime * config.limbs), vxy((size_t)ccs.t * config.limbs);
    for (size_t i = 0; i < ccs.s; i++) std::copy(points.rx_ry[i].begin(), points.rx_ry[i].begin() + config.limbs, rx.begin() + i * config.limbs);
    for (size_t i = 0; i < ccs.s_prime; i++) std::copy(r_y[i].begin(), r_y[i].begin() + config.limbs, ry.begin() + i * config.limbs);
    {
        std::lock_guard<std::mutex> one(prepared->mu_);
        ccs_check(prepared->h_, zip_ccs_eval_matrices(prepared->h_, rx.data(), ry.data(), vxy.data()), "zip_ccs_eval_matrices");
    }
    std::vector<Limbs> V_xy(ccs.t);
    for (size_t k = 0; k < ccs.t; k++)
        for (uint32_t i = 0; i < config.limbs; i++) V_xy[k][i] = vxy[k * config.limbs + i];
    Limbs lhs = lin_comb_V_s(config, points.gamma, V_xy);  // :264
    config.mul_assign(lhs, v);
    if (lhs != points.e_y)
        throw SpartanError(SpartanError::PcsVerification, "linear combination of powers of gamma and V_x != e_y");
}

void ZincVerifier::verify(const ccs::Statement_Z &statement, const ZincProof &proof, KeccakTranscript &transcript,
                          const ccs::CCS_Z &ccs, const FieldConfig &config, PreparedCcs *prepared) const {
    if (ccs.s == 0 || ccs.s > 28 || ccs.m != ((size_t)1 << ccs.s) || ccs.n != ccs.m || ccs.s_prime != ccs.s)
        throw std::logic_error("m == n == 2^s is required (see ZincProver)");
    const VerificationPoints points = spartan_verify(proof.spartan_proof, ccs, transcript, config);  // :62-64
    verify_pcs_proof(statement, proof.zip_proof, points, ccs, transcript, config, prepared);          // :66-73
}

}  // namespace zinc
