#!/usr/bin/env python3
"""CPU check of the dropout masks' hash (csrc/common.h: otr_rand32, round 5) restated in numpy: keep rate, serial / cross-step /
cross-site correlations and row / column means of a [7968, 256] mask at p = 0.1 against what an i.i.d. source gives."""
import numpy as np


def rand32(seed, idx):
    s0, s1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over='ignore'):
        k0 = np.uint32(s0 * np.uint32(0x9E3779B1)) ^ s1
        k1 = np.uint32(s1 * np.uint32(0x85EBCA77)) ^ np.uint32(s0 >> np.uint32(15)) ^ np.uint32(0xC2B2AE3D)
        x = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ k0
        x = x + ((idx >> np.uint64(32)).astype(np.uint32) * np.uint32(0x27D4EB2F))
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x ^= k1
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def main():
    thr = np.uint32(int(0.1 * 2 ** 32))
    seed, inc, n = 0x5EED, 0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF, 7968 * 256
    idx = np.arange(n, dtype=np.uint64)
    print('i.i.d.: column-mean std %.5f, row-mean std %.5f, correlation noise %.5f' % ((0.09 / 7968) ** 0.5, (0.09 / 256) ** 0.5, n ** -0.5))
    prev = None
    for step in range(6):
        k = (rand32(seed, idx) >= thr).astype(np.float64)
        line = 'step %d keep %.5f lag1 %+.5f lag256 %+.5f colmean std %.5f rowmean std %.5f' % (
            step, k.mean(), np.corrcoef(k[:-1], k[1:])[0, 1], np.corrcoef(k[:-256], k[256:])[0, 1], k.reshape(-1, 256).mean(0).std(),
            k.reshape(-1, 256).mean(1).std())
        if prev is not None:
            line += ' vs previous step %+.5f' % np.corrcoef(k, prev)[0, 1]
        site2 = (rand32(seed, idx + np.uint64(n)) >= thr).astype(np.float64)
        print(line + ' vs next site %+.5f' % np.corrcoef(k, site2)[0, 1])
        prev, seed = k, (seed + inc) & 0xFFFFFFFFFFFFFFFF


if __name__ == '__main__':
    main()
