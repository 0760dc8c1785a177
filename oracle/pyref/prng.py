"""xoshiro256** input generator shared by the golden-vector script, the tests and bench.py
(SURVEY.md §8d "Standalone kernel inputs": seed 0x53574D41524C494E "SWMARLIN", 4 limbs per draw,
top 3 bits masked, rejection above r).  Test/bench infrastructure; inputs only, no reference logic."""
import numpy as np

from .bls12_377 import R

MASK64 = 0xFFFFFFFFFFFFFFFF
SEED = 0x53574D41524C494E


def _rotl(x, k):
    return ((x << k) & MASK64) | (x >> (64 - k))


class Xoshiro256ss:
    def __init__(self, seed=SEED):
        # splitmix64 expansion of the 64-bit seed
        s = []
        x = seed & MASK64
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & MASK64
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
            s.append(z ^ (z >> 31))
        self.s = s

    def next_u64(self):
        s = self.s
        result = (_rotl((s[1] * 5) & MASK64, 7) * 9) & MASK64
        t = (s[1] << 17) & MASK64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = _rotl(s[3], 45)
        return result

    def fr(self):
        while True:
            limbs = [self.next_u64() for _ in range(4)]
            limbs[3] &= MASK64 >> 3
            v = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
            if v < R:
                return v


def fr_array(n, seed=SEED):
    """n uniform Fr values as an (n, 4) uint64 array of little-endian limbs (vectorised; numpy PCG seeded
    from `seed`, rejection-sampled).  Used for inputs too large for the pure-Python generator."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((n, 4), dtype=np.uint64)
    filled = 0
    r_limbs = [(R >> (64 * i)) & MASK64 for i in range(4)]
    while filled < n:
        m = max(1024, int((n - filled) * 1.8))
        cand = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64)
        cand[:, 3] &= np.uint64(MASK64 >> 3)
        # lexicographic compare cand < R from the top limb down
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (cand[:, i] < np.uint64(r_limbs[i]))
            eq &= cand[:, i] == np.uint64(r_limbs[i])
        good = cand[lt]
        take = min(len(good), n - filled)
        out[filled:filled + take] = good[:take]
        filled += take
    return out
