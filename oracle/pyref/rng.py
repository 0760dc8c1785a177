"""Python restatement of the RNGs / Fiat-Shamir transcript behind src/marlin (oracle; test infra only).

Reference anchors: /root/reference/src/marlin/mod.rs:13 (FS = SimpleHashFiatShamirRng<Blake2s, ChaChaRng>),
:33-35 (generate_rand = ark_std::test_rng()).  The crates themselves (rand 0.8 / rand_chacha 0.3.1 /
blake2 0.9 / ark-std 0.3 / ark-marlin) are not vendored; behaviour restated from SURVEY.md Appendix
A.1/A.8 [U].  ChaCha is pinned by the RFC 7539 / djb zero-key keystream vector in tests/.
"""
import hashlib
from .bls12_377 import R, Q, Fq2

MASK32 = 0xFFFFFFFF


def _rotl(v, c):
    return ((v << c) & MASK32) | (v >> (32 - c))


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & MASK32
    s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & MASK32
    s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & MASK32
    s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & MASK32
    s[b] = _rotl(s[b] ^ s[c], 7)


def chacha_block(key_words, counter, stream, rounds):
    """djb ChaCha with 64-bit block counter (words 12,13) and 64-bit stream id (words 14,15)."""
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & MASK32, (counter >> 32) & MASK32, stream & MASK32, (stream >> 32) & MASK32]
    s = list(init)
    for _ in range(rounds // 2):
        _qr(s, 0, 4, 8, 12)
        _qr(s, 1, 5, 9, 13)
        _qr(s, 2, 6, 10, 14)
        _qr(s, 3, 7, 11, 15)
        _qr(s, 0, 5, 10, 15)
        _qr(s, 1, 6, 11, 12)
        _qr(s, 2, 7, 8, 13)
        _qr(s, 3, 4, 9, 14)
    return [(s[i] + init[i]) & MASK32 for i in range(16)]


class ChaChaRng:
    """rand_chacha::ChaChaXRng = BlockRng over a 64-word (4-block) buffer."""

    BUF = 64

    def __init__(self, seed32, rounds):
        assert len(seed32) == 32
        self.key = [int.from_bytes(seed32[4 * i:4 * i + 4], "little") for i in range(8)]
        self.rounds = rounds
        self.counter = 0
        self.results = [0] * self.BUF
        self.index = self.BUF  # empty

    def _generate(self):
        out = []
        for _ in range(4):
            out += chacha_block(self.key, self.counter, 0, self.rounds)
            self.counter += 1
        self.results = out

    def _generate_and_set(self, index):
        self._generate()
        self.index = index

    def next_u32(self):
        if self.index >= self.BUF:
            self._generate_and_set(0)
        v = self.results[self.index]
        self.index += 1
        return v

    def next_u64(self):
        ln = self.BUF
        idx = self.index
        if idx < ln - 1:
            self.index += 2
            return (self.results[idx + 1] << 32) | self.results[idx]
        if idx >= ln:
            self._generate_and_set(2)
            return (self.results[1] << 32) | self.results[0]
        x = self.results[ln - 1]
        self._generate_and_set(1)
        y = self.results[0]
        return (y << 32) | x

    # ---- rand::distributions::Standard
    def gen_bool(self):
        return (self.next_u32() >> 31) == 1

    def gen_u128(self):
        x = self.next_u64()
        y = self.next_u64()
        return (y << 64) | x

    # ---- ark_ff UniformRand: accepted limbs ARE the Montgomery representation
    def _rand_mont(self, nlimbs, shave, modulus):
        while True:
            limbs = [self.next_u64() for _ in range(nlimbs)]
            limbs[-1] &= 0xFFFFFFFFFFFFFFFF >> shave
            v = sum(l << (64 * i) for i, l in enumerate(limbs))
            if v < modulus:
                return v

    def rand_fr(self):
        """Returns the standard-form value of the sampled element."""
        m = self._rand_mont(4, 3, R)
        return m * pow(1 << 256, -1, R) % R

    def rand_fq(self):
        m = self._rand_mont(6, 7, Q)
        return m * pow(1 << 384, -1, Q) % Q

    def rand_fq2(self):
        c0 = self.rand_fq()
        c1 = self.rand_fq()
        return Fq2(c0, c1)


TEST_RNG_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)


def test_rng():
    """ark_std::test_rng(): StdRng (rand 0.8 = ChaCha12) from a fixed seed."""
    return ChaChaRng(TEST_RNG_SEED, 12)


def blake2s(data):
    return hashlib.blake2s(data, digest_size=32).digest()


class FiatShamirRng:
    """ark_marlin::SimpleHashFiatShamirRng<Blake2s, ChaChaRng> (ChaChaRng = ChaCha20)."""

    def __init__(self, init_bytes):
        self.seed = blake2s(bytes(init_bytes))
        self.r = ChaChaRng(self.seed, 20)

    def absorb(self, data):
        self.seed = blake2s(bytes(data) + self.seed)
        self.r = ChaChaRng(self.seed, 20)

    def rand_fr(self):
        return self.r.rand_fr()

    def gen_u128(self):
        return self.r.gen_u128()
