"""Host-side expansion of a permutation seed into the index table the C ABI takes.

Mirrors `shuffle_seeded` (src/zip/utils.rs:139-142): `StdRng::seed_from_u64(seed)`
then `SliceRandom::shuffle`, applied to the identity so that
`shuffle_seeded(x, seed)[j] == x[perm[j]]`.

rand 0.9.2 / rand_chacha are not vendored in the reference and no reference test pins a
permutation, so this restates the crates' published algorithm (PCG32 seed expansion ->
ChaCha12 block RNG -> IncreasingUniform Fisher-Yates with Canon's-method `random_range`).
Every piece is pinned by a published vector (tests/golden/rand_vectors.json,
tests/test_host_mirror.py) -- since round 3 also `seed_from_u64`, by rand_pcg's own
construction vector (`Lcg64Xsh32::seed_from_u64(0).next_u64()`).  In the Rust integration the shim calls the real
`shuffle_seeded` on `[0..codeword_len)` and passes the table through the FFI, so nothing
on the GPU side depends on this file.
"""
import numpy as np

_M32 = 0xFFFFFFFF
_M64 = 0xFFFFFFFFFFFFFFFF


def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & _M32


def _pcg32_output(state):
    """PCG XSH-RR 64/32 (O'Neill)."""
    xorshifted = (((state >> 18) ^ state) >> 27) & _M32
    rot = state >> 59
    return ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & _M32


class _RangeMixin:
    def random_range_u32(self, bound):
        """UniformInt<u32>::sample_single_inclusive(0, bound - 1): Canon's method, one retry."""
        m = self.next_u32() * bound
        result, lo = m >> 32, m & _M32
        if lo > ((-bound) & _M32):
            new_hi = (self.next_u32() * bound) >> 32
            result += 1 if lo + new_hi > _M32 else 0
        return result


def seed_from_u64_words(seed_u64, n_words):
    """rand_core::SeedableRng::seed_from_u64: the u64 seed expanded to n_words little-endian 32-bit words of seed
    (PCG32 steps with increment 11634580027462260723, the state advanced BEFORE each output).  Pinned by rand_pcg's
    construction vector (tests/golden/rand_vectors.json: pcg32_seed_from_u64)."""
    state, words = seed_u64 & _M64, []
    for _ in range(n_words):
        state = (state * 6364136223846793005 + 11634580027462260723) & _M64
        words.append(_pcg32_output(state))
    return words


class Pcg32(_RangeMixin):
    """rand_pcg::Lcg64Xsh32::new(state, stream): the generator rand's own value-stability tests use."""

    def __init__(self, state, stream):
        self.inc = ((stream << 1) | 1) & _M64
        self.state = ((state + self.inc) * 6364136223846793005 + self.inc) & _M64

    @classmethod
    def from_seed(cls, seed16):
        """Lcg64Xsh32::from_seed: state = LE u64 of bytes 0..8, increment = LE u64 of bytes 8..16 | 1."""
        g = cls.__new__(cls)
        g.inc = (int.from_bytes(bytes(seed16[8:16]), "little") | 1) & _M64
        g.state = ((int.from_bytes(bytes(seed16[:8]), "little") + g.inc) * 6364136223846793005 + g.inc) & _M64
        return g

    def next_u64(self):
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)

    def next_u32(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & _M64
        return _pcg32_output(old)


class ChaCha12Rng(_RangeMixin):
    """rand_chacha::ChaCha12Rng: 64-bit block counter, stream 0, words consumed in order."""

    def __init__(self, seed_u64=None, key_words=None):
        if key_words is not None:  # SeedableRng::from_seed: eight little-endian key words
            key = [int(w) & _M32 for w in key_words]
        else:
            key = seed_from_u64_words(seed_u64, 8)  # rand_core::SeedableRng::seed_from_u64
        self.key, self.counter, self.buf, self.idx = key, 0, [], 16

    def _block(self):
        s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + self.key + [
            self.counter & _M32, (self.counter >> 32) & _M32, 0, 0]
        x = list(s)

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 16)
            x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 8)
            x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 7)

        for _ in range(6):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        self.buf = [(a + b) & _M32 for a, b in zip(x, s)]
        self.counter += 1
        self.idx = 0

    def next_u32(self):
        if self.idx >= 16:
            self._block()
        v = self.buf[self.idx]
        self.idx += 1
        return v



def shuffle_seeded_perm(seed: int, length: int) -> np.ndarray:
    return shuffle_perm_with(ChaCha12Rng(seed), length)


def shuffle_perm_with(rng, length: int) -> np.ndarray:
    """SliceRandom::shuffle of [0, length) over any generator with next_u32 / random_range_u32."""
    perm = list(range(length))
    if length > 1:
        n, chunk, chunk_remaining = 0, 0, 1  # IncreasingUniform::new(rng, 0)
        for i in range(length):
            next_n = n + 1
            if chunk_remaining > 0:
                next_rem = chunk_remaining - 1
            else:
                product, current = next_n, next_n + 1  # calculate_bound_u32
                while product * current <= _M32:
                    product *= current
                    current += 1
                chunk = rng.random_range_u32(product)
                next_rem = (current - next_n) - 1
            if next_rem == 0:
                j = chunk
            else:
                j = chunk % next_n
                chunk //= next_n
            chunk_remaining, n = next_rem, next_n
            perm[i], perm[j] = perm[j], perm[i]
    return np.array(perm, dtype=np.uint32)
