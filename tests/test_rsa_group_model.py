"""CPU (-m "not gpu"): a Python-integer model of csrc/rsa_quad.hip.h — G lanes per signature, 19 limbs of 28 bits per
lane, a 38-column window of 64-bit accumulators per lane, quotient digits from lane 0, the low half of a window handed one
lane down every 19 steps, no conditional subtraction — step for step as the kernel does it, with the kernel's register
widths asserted (every accumulator < 2^64, every operand limb <= 2^28).  It pins the algorithm and its bounds on the
host: results must equal pow(s, 65537, n) for random and for worst-case operands (moduli and signatures of all-ones
limbs), for four lanes (<= 2048 bits) and eight (<= 4096 bits), and the cached constant must be what the pre-pass
derives from 2^(2 * 2048 NL) mod n with two 32-bit-radix Montgomery products."""
import random

import pytest

QL, MASK = 19, (1 << 28) - 1
U64 = 1 << 64


def to_lanes(x, G):
    limbs = [(x >> (28 * t)) & MASK for t in range(G * QL)]
    return [limbs[QL * p:QL * (p + 1)] for p in range(G)]


def from_lanes(v):
    return sum(l << (28 * (QL * p + j)) for p, lane in enumerate(v) for j, l in enumerate(lane))


def qmont_columns(a, b, n, ninv, G):
    """qmont_columns<G>: returns W[p][0..18] (lazy columns of the result)."""
    W = [[0] * (2 * QL) for _ in range(G)]
    B = [list(x) for x in b]
    for _blk in range(G):
        for r in range(QL):
            bd = B[0][r]                                             # g_bcast0
            for p in range(G):
                for k in range(QL):
                    base = 0 if (k == QL - 1 and r > 0) else W[p][k + r]      # first touch of a high column
                    W[p][k + r] = a[p][k] * bd + base
                    assert W[p][k + r] < U64
            m = ((W[0][r] & 0xFFFFFFFF) * ninv) & MASK                # lane 0's quotient digit, broadcast
            for p in range(G):
                for k in range(QL):
                    W[p][k + r] = n[p][k] * m + W[p][k + r]
                    assert W[p][k + r] < U64
                W[p][r + 1] += W[p][r] >> 28
                assert W[p][r + 1] < U64
                W[p][r] &= MASK
            assert W[0][r] == 0                                      # reduced
        recv = [[W[(p + 1) % G][j] for j in range(QL)] for p in range(G)]      # g_rotdown of the finished low columns
        for p in range(G):
            for j in range(QL):
                W[p][j] = W[p][QL + j] + recv[p][j]
                assert W[p][j] < U64
        B = [B[(p + 1) % G] for p in range(G)]                       # g_rotdown of the multiplier digits
    return W


def qnorm(W, G, cross):
    out = [[0] * QL for _ in range(G)]
    carry = [0] * G
    for p in range(G):
        c = 0
        for j in range(QL):
            t = W[p][j] + c
            assert t < U64
            out[p][j] = t & MASK
            c = t >> 28
        carry[p] = c
    for _ in range(cross):
        cin = [0] + carry[:-1]                                       # g_fromprev, lane 0 masked
        for p in range(G):
            t = out[p][0] + cin[p]
            out[p][0] = t & MASK
            c = t >> 28
            for j in range(1, QL):
                t = out[p][j] + c
                assert t < (1 << 32)
                out[p][j] = t & MASK
                c = t >> 28
            carry[p] = c
    assert carry[G - 1] == 0                                         # nothing beyond 532 G bits
    last = [0] + carry[:-1]
    for p in range(G):
        out[p][0] += last[p]
        assert out[p][0] <= MASK + 1
    return out


def cond_sub(acc, nn, G):
    """The last step of rsa_group_wave: acc (exact limbs, < 2n) minus n when acc >= n.  Per lane the sign of the highest
    differing limb; the highest differing lane of the group decides acc >= n, the lanes below a lane decide its borrow."""
    c = []
    for p in range(G):
        d = 0
        for j in reversed(range(QL)):
            if d == 0:
                d = (acc[p][j] > nn[p][j]) - (acc[p][j] < nn[p][j])
        c.append(d)
    gt = sum(1 << p for p in range(G) if c[p] > 0)
    lt = sum(1 << p for p in range(G) if c[p] < 0)
    if gt < lt:
        return acc
    out = []
    for p in range(G):
        low = (1 << p) - 1
        borrow = 1 if (lt & low) > (gt & low) else 0
        lane = []
        for j in range(QL):
            t = acc[p][j] - nn[p][j] - borrow
            lane.append(t & MASK)
            borrow = 1 if t < 0 else 0
        out.append(lane)
    return out


def group_modexp(s, n, G):
    Rbits = 28 * QL * G
    rr = to_lanes(pow(2, 2 * Rbits, n), G)
    ninv = (-pow(n, -1, 1 << 28)) & MASK
    nn = to_lanes(n, G)
    acc = to_lanes(s, G)
    plain = acc
    # s R, sixteen squarings -> s^65536 R; the last product takes the PLAIN s: (s^65536 R) s / R = s^65537, out of the
    # Montgomery domain without a product by one
    for step in range(18):
        b = rr if step == 0 else (plain if step == 17 else acc)
        acc = qnorm(qmont_columns(acc, b, nn, ninv, G), G, G - 1 if step == 17 else 1)
        assert from_lanes(acc) < 2 * n                               # no conditional subtraction: values stay below 2n
    assert all(l <= MASK for lane in acc for l in lane)              # exact limbs
    em = cond_sub(acc, nn, G)                                        # < n + n^2 / R: one subtraction at most
    return from_lanes(em)


def rand_odd(bits, rng):
    return rng.getrandbits(bits) | (1 << (bits - 1)) | 1


@pytest.mark.parametrize("G,bits", [(4, 2048), (4, 1024), (4, 1537), (8, 4096), (8, 3072), (8, 2049)])
def test_group_modexp_matches_pow(G, bits):
    rng = random.Random(1000 * G + bits)
    for trial in range(2 if G == 8 else 3):
        n = rand_odd(bits, rng)
        s = rng.randrange(n)
        assert group_modexp(s, n, G) == pow(s, 65537, n), (G, bits, trial)


@pytest.mark.parametrize("G,bits", [(4, 2048), (8, 4096)])
def test_group_modexp_worst_case_limbs(G, bits):
    n = (1 << bits) - 1                                              # all-ones limbs (odd; not a product of two primes, irrelevant here)
    for s in (n - 1, n - 2, (1 << bits) - (1 << (bits - 28)) - 1, 1, 0):
        assert group_modexp(s, n, G) == pow(s, 65537, n), (G, hex(s)[:20])
    n = (1 << (bits - 1)) + 1                                        # the smallest modulus of this length
    assert group_modexp(n - 1, n, G) == pow(n - 1, 65537, n)


@pytest.mark.parametrize("G,bits", [(4, 2048), (4, 1031), (8, 4096), (8, 2100)])
def test_conditional_subtraction(G, bits):
    """The kernel's last step on values a signature that verifies never produces (acc >= n): equal, one more, all borrows
    (n with zero low limbs), the largest value the last product can leave."""
    rng = random.Random(31 * G + bits)
    ns = [rand_odd(bits, rng), (1 << (bits - 1)) + 1, (1 << bits) - 1, (1 << (bits - 1)) + (1 << (28 * QL)) + 1]
    for n in ns:
        for x in (0, 1, n - 1, n, n + 1, n + (1 << (28 * QL)) - 1, n + (n >> 80), 2 * n - 1, rng.randrange(n), n + rng.randrange(n)):
            if x >= 1 << (28 * QL * G):
                continue
            got = from_lanes(cond_sub(to_lanes(x, G), to_lanes(n, G), G))
            assert got == (x - n if x >= n else x), (G, bits, hex(x)[:18])


@pytest.mark.parametrize("NL,G", [(1, 4), (2, 8)])
def test_cached_constant_derivation(NL, G):
    """rsa_kernel.hip.h: mont(R^2, 2^c) = 2^c R, mont(R^2, 2^c R) = 2^c R^2 (radix R = 2^(2048 NL), c = 160 / 320) must be
    R'^2 mod n for R' = 2^(532 G)."""
    rng = random.Random(7 + NL)
    n = rand_odd(2048 * NL - 5, rng)
    R = 1 << (2048 * NL)
    Rinv = pow(R, -1, n)
    mont = lambda x, y: x * y * Rinv % n
    rr = R * R % n
    c = 2 * (532 * G - 2048 * NL)
    assert c == (160 if NL == 1 else 320)
    r2 = mont(rr, mont(rr, 1 << c))
    assert r2 == pow(2, 2 * 532 * G, n)
