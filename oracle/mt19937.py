"""MT19937 as consumed by ``torch.rand`` on the CPU generator (oracle; test infrastructure).

Follows the published Matsumoto-Nishimura MT19937 (init_genrand + genrand_int32); torch's
CPUGeneratorImpl uses exactly this engine (at::mt19937), and fp32 ``torch.rand`` maps each
32-bit draw to ``(x & 0xFFFFFF) * 2**-24`` serially in row-major order (SURVEY.md A.1 item 7,
re-verified in make_golden.py against torch itself).  The reference consumes it through
models/shapley.py:69,114,133.
"""
from __future__ import annotations

import numpy as np

N, M = 624, 397
_UPPER, _LOWER, _MAG = np.uint32(0x80000000), np.uint32(0x7FFFFFFF), np.uint32(0x9908B0DF)


class MT19937:
    def __init__(self, seed: int):
        mt = np.zeros(N, dtype=np.uint64)
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, N):
            mt[i] = (1812433253 * (int(mt[i - 1]) ^ (int(mt[i - 1]) >> 30)) + i) & 0xFFFFFFFF
        self.mt = mt.astype(np.uint32)
        self.pos = N  # next draw triggers a twist

    @classmethod
    def from_state(cls, mt: np.ndarray, pos: int) -> "MT19937":
        o = cls.__new__(cls)
        o.mt = np.asarray(mt, dtype=np.uint32).copy()
        o.pos = int(pos)
        return o

    def _twist(self) -> None:
        mt = self.mt

        def step(lo: int, hi: int) -> None:  # mt[lo:hi] from mt[i], mt[i+1], mt[(i+M)%N]
            idx = np.arange(lo, hi)
            y = (mt[idx] & _UPPER) | (mt[(idx + 1) % N] & _LOWER)
            mt[idx] = mt[(idx + M) % N] ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), _MAG, np.uint32(0))

        # each phase only reads words already final for it (old for [0,227), new after)
        step(0, N - M)
        step(N - M, 2 * (N - M))
        step(2 * (N - M), N - 1)
        step(N - 1, N)
        self.pos = 0

    def raw(self, count: int) -> np.ndarray:
        out = np.empty(count, dtype=np.uint32)
        done = 0
        while done < count:
            if self.pos >= N:
                self._twist()
            take = min(N - self.pos, count - done)
            y = self.mt[self.pos:self.pos + take].copy()
            y ^= y >> np.uint32(11)
            y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
            y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
            y ^= y >> np.uint32(18)
            out[done:done + take] = y
            self.pos += take
            done += take
        return out

    def rand_f32(self, *shape: int) -> np.ndarray:
        """== torch.rand(*shape) (fp32, CPU generator)."""
        n = int(np.prod(shape)) if shape else 1
        r = self.raw(n)
        return ((r & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(shape)
