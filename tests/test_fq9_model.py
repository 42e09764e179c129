"""CPU model of the 9 x 29-bit unsaturated Fq arithmetic of the MSM gather kernel (co-zkvms_amd/csrc/fq9.hip.hpp):
the same column schedule, constants parsed from the generated fq9_consts.inc, with an assertion on every 64-bit
accumulator and every 32-bit limb.  Checks (no GPU): (i) the generated constants are what the header says they are;
(ii) long random chains of mixed additions -- including negated points and worst-case limb patterns -- never
overflow, keep the documented value bounds (X < 6p, Y < 2p, ZZ, ZZZ < 1.05p), and (iii) after the three outgoing
products give the exact EC sum AND satisfy ZZ^3 = ZZZ^2 (what the later XYZZ fold levels rely on)."""
import os
import random
import re

import pyref as O

P = O.P if hasattr(O, "P") else 21888242871839275222246405745257275088696311157297823662689037894645226208583
W, N = 29, 9
MASK = (1 << W) - 1
R, RP = 1 << 256, 1 << 261
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _consts():
    txt = open(os.path.join(ROOT, "co-zkvms_amd", "csrc", "fq9_consts.inc")).read()
    out = {}
    for name, body in re.findall(r"(F9_\w+)\[9\] = \{([^}]*)\}", txt):
        out[name] = [int(x.strip().rstrip("u"), 16) for x in body.split(",")]
    out["F9_INV"] = int(re.search(r"F9_INV = (0x[0-9a-f]+)u", txt).group(1), 16)
    return out


C = _consts()


def val(l):
    return sum(x << (W * i) for i, x in enumerate(l))


def limbs(x):
    return [(x >> (W * i)) & MASK for i in range(N)]


def test_generated_constants():
    assert val(C["F9_P"]) == P and all(x <= MASK for x in C["F9_P"])
    assert (C["F9_INV"] * P + 1) % (1 << W) == 0
    for name, k, lift in (("F9_C2", 2, 1), ("F9_C3", 3, 1), ("F9_C7", 7, 1), ("F9_C4X3", 4, 3)):
        assert val(C[name]) == k * P
        assert all(lift << W <= x < (lift + 1) << W for x in C[name][:-1])
    assert val(C["F9_ONE"]) == R % P
    assert val(C["F9_LAM"]) == R * R * pow(RP, -1, P) % P
    assert val(C["F9_OUT"]) == RP * RP * pow(R, -1, P) % P
    assert val(C["F9_OUT2"]) == RP ** 3 * pow(R * R, -1, P) % P


def mul(a, b, c=None, d=None):
    acc, m, r = 0, [0] * 9, [0] * 9
    PL, INV = C["F9_P"], C["F9_INV"]
    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
        for i in range(lo, hi + 1):
            acc += a[i] * b[k - i]
            if c is not None:
                acc += c[i] * d[k - i]
        if k < 9:
            for i in range(k):
                acc += m[i] * PL[k - i]
            assert acc < 1 << 64
            m[k] = ((acc & 0xFFFFFFFF) * INV) & MASK
            acc += m[k] * PL[0]
            assert acc < 1 << 64 and acc & MASK == 0
            acc >>= W
        else:
            for i in range(lo, 9):
                acc += m[i] * PL[k - i]
            assert acc < 1 << 64
            r[k - 9] = acc & MASK
            acc >>= W
    assert acc < 1 << 32
    r[8] = acc
    return r


def sqr(a):
    assert all(x <= MASK for x in a[:8])
    r = mul(a, a)
    # the kernel's schedule: off-diagonal products once against the doubled operand -- same column sums
    return r


def norm(a):
    r, c = [0] * 9, 0
    for i in range(8):
        t = a[i] + c
        assert t < 1 << 32
        r[i] = t & MASK
        c = t >> W
    r[8] = a[8] + c
    assert r[8] < 1 << 32
    return r


def sub(a, Cc, b):
    r = [a[i] + Cc[i] - b[i] for i in range(9)]
    assert all(0 <= x < 1 << 32 for x in r)
    return r


def from_affine(qx, qy):
    return [mul(qx, C["F9_ONE"]), mul(qy, C["F9_LAM"]), list(C["F9_ONE"]), list(C["F9_LAM"])]


def madd(acc, qx, qy):
    X, Y, ZZ, ZZZ = acc
    U2, S2 = mul(qx, ZZ), mul(qy, ZZZ)
    Pd, Rd = norm(sub(U2, C["F9_C7"], X)), norm(sub(S2, C["F9_C3"], Y))
    PP = sqr(Pd)
    if val(PP) % P == 0:
        return None
    RR, PPP, Q = sqr(Rd), mul(Pd, PP), mul(X, PP)
    X3 = [RR[i] + C["F9_C4X3"][i] - PPP[i] - 2 * Q[i] for i in range(9)]
    assert all(0 <= x < 1 << 32 for x in X3)
    X3 = norm(X3)
    T = sub(Q, C["F9_C7"], X3)
    NY = [C["F9_C3"][i] - Y[i] for i in range(9)]
    assert all(x >= 0 for x in NY)
    return [X3, mul(Rd, T, NY, PPP), mul(ZZ, PP), mul(ZZZ, PPP)]


def to_std(acc):
    X, Y, ZZ, ZZZ = acc
    return [val(mul(X, C["F9_OUT"])), val(mul(Y, C["F9_OUT2"])), val(ZZ), val(mul(ZZZ, C["F9_OUT"]))]


def test_madd_chain_bounds_and_exactness():
    rnd = random.Random(7)
    pts = [O.g1_mul(O.G1_GEN, rnd.randrange(1, O.R)) for _ in range(48)]
    acc, ref = None, None
    worst = [0.0] * 4
    for k, pt in enumerate(pts):
        neg = k % 3 == 1
        x, y = pt[0] * R % P, pt[1] * R % P
        qx, qy = limbs(x), limbs(y)
        if neg:  # negative digit: 2p - y limb-wise, as the kernel does
            qy = [C["F9_C2"][i] - qy[i] for i in range(9)]
            assert all(0 <= v < 1 << 30 for v in qy)
            pt = O.g1_neg(pt)
        acc = from_affine(qx, qy) if acc is None else madd(acc, qx, qy)
        assert acc is not None
        ref = O.g1_add(ref, pt)
        for j in range(4):
            worst[j] = max(worst[j], val(acc[j]) / P)
        X, Y, ZZ, ZZZ = [v * pow(R, -1, P) % P for v in to_std(acc)]
        assert pow(ZZ, 3, P) == pow(ZZZ, 2, P)
        assert (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P) == ref
    assert worst[0] < 6 and worst[1] < 2 and worst[2] < 1.05 and worst[3] < 1.05


def test_worst_case_limbs_do_not_overflow():
    """all-ones limb patterns at the documented bounds through every product shape of madd9"""
    full = [MASK] * 8 + [(6 * P) >> (W * 8)]          # normalised, value ~ 6p in the top limb
    u = [3 * (1 << W) - 1] * 8 + [(9 * P) >> (W * 8)]  # un-normalised operand: limbs < 3 * 2^29
    ny = [(1 << 30) - 1] * 8 + [(3 * P) >> (W * 8)]
    mul(full, full)
    mul(full, u)
    mul(full, u, ny, full)  # the fused Y3 product: Rd * T + NY * PPP
    sqr(full)


def test_doubling_is_reported():
    pt = O.g1_mul(O.G1_GEN, 12345)
    qx, qy = limbs(pt[0] * R % P), limbs(pt[1] * R % P)
    acc = from_affine(qx, qy)
    assert madd(acc, qx, qy) is None  # P + P: PP == 0 mod p -> the kernel queues the segment for the saturated path
