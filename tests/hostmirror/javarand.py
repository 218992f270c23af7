"""java.util.Random for the host layer (fold assignment must follow the reference's stream:
happy.coding.math.Randoms.seed(n) -> new java.util.Random(n); uniform() -> nextDouble()).  Published JDK algorithm:
48-bit LCG, multiplier 0x5DEECE66D, addend 0xB; nextDouble = (next(26) << 27 + next(27)) * 2^-53."""
import numpy as np

_MULT, _MASK = 0x5DEECE66D, (1 << 48) - 1


class JavaRandom:
    def __init__(self, seed):
        self.seed = (int(seed) ^ _MULT) & _MASK

    def _next(self, bits):
        self.seed = (self.seed * _MULT + 0xB) & _MASK
        return self.seed >> (48 - bits)

    def next_double(self):
        return ((self._next(26) << 27) + self._next(27)) * (1.0 / (1 << 53))

    def doubles(self, n):
        out = np.empty(n)
        for i in range(n):
            out[i] = self.next_double()
        return out
