#!/usr/bin/env python3
"""Which rows of an octahedral grid could leave the chirp-z (Bluestein) FFT kernels, by weight (grid points)?
VERDICT r4 #3: count before building.  Row i (1..H) has NLOEN = 20 + 4 (i - 1) points, half-length sz = NLOEN / 2 (complex transform
of the real row).  Classes:
  direct   sz = product of at most three radices from EMI_MR_RADICES (emi_mr_body.h: 2..16, 17, 19, 23): k_fft_*_mr today
  +29/31   the same with 29 and 31 allowed (DESIGN section 9 iv: needs a 168-register variant)
  rader    sz = c * p, p prime > 23 with p - 1 a product of at most three existing radices (cyclic convolution of length p - 1 on
           the direct passes), c a product of at most two existing radices (or 1)
  rader2   as rader but p - 1 only needs to be 23-smooth with any number of factors (would need more passes)
usage: chirpz_share.py [H ...]   (default 1280 2560 400)"""
import sys
from itertools import product

RAD = [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 23]


def products(radices, nmax, maxf):
    out = {1}
    cur = {1}
    for _ in range(maxf):
        cur = {a * r for a in cur for r in radices if a * r <= nmax}
        out |= cur
    return out


def is_prime(n):
    if n < 2:
        return False
    i = 2
    while i * i <= n:
        if n % i == 0:
            return False
        i += 1
    return True


def smooth(n, b):
    for p in range(2, b + 1):
        while n % p == 0:
            n //= p
    return n == 1


def main():
    Hs = [int(a) for a in sys.argv[1:]] or [1280, 2560, 400]
    for H in Hs:
        nmax = 2 * (5 + H)
        d3 = products(RAD, nmax, 3)
        d3x = products(RAD + [29, 31], nmax, 3)
        c2 = products(RAD, nmax, 2)
        tot = 0
        w = dict(direct=0, x2931=0, rader=0, rader2=0, rader_prime_only=0)
        lens = dict(direct=0, x2931=0, rader=0, rader2=0)
        ex = []
        for i in range(1, H + 1):
            n = 20 + 4 * (i - 1)
            sz = n // 2
            tot += n
            if sz in d3:
                w["direct"] += n
                lens["direct"] += 1
                continue
            if sz in d3x:
                w["x2931"] += n
                lens["x2931"] += 1
                continue
            # largest prime factor
            m, p = sz, 0
            f = 2
            while f * f <= m:
                while m % f == 0:
                    p = max(p, f)
                    m //= f
                f += 1
            if m > 1:
                p = max(p, m)
            c = sz // p
            if sz % (p * p) != 0 and c in c2:
                if (p - 1) in d3:
                    w["rader"] += n
                    lens["rader"] += 1
                    if len(ex) < 12 or i > H - 6:
                        ex.append((n, sz, c, p))
                elif smooth(p - 1, 23):
                    w["rader2"] += n
                    lens["rader2"] += 1
        print("O%d: %d row lengths, %.3e points per hemisphere" % (H, H, tot))
        acc = 0.0
        for k in ("direct", "x2931", "rader", "rader2"):
            acc += w[k]
            print("  %-8s %5d lengths  %5.1f %% of the grid   (cumulative off chirp-z: %5.1f %%)" % (k, lens[k], 100.0 * w[k] / tot, 100.0 * acc / tot))
        print("  examples (n, sz, c, p):", ex[:8], "...", ex[-4:])


if __name__ == "__main__":
    main()
