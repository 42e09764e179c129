"""
ORACLE (test infrastructure, NOT product code) -- exact big-integer CPU restatement of the
co-zkvms sumcheck + polynomial-commitment hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (co-zkvms_amd/) never imports it and fails loudly without its HIP library.

PARITY STATUS: "parity unpinned" by reference *outputs*: the reference is Rust, cannot be built
here (no cargo/rustc, six git-pinned forks absent) and its tests hold no golden vectors for this
path.  The oracle is pinned by (1) the three constants the reference does hold
(TWO_INV `snarks-core/src/field.rs:5-7`; the Montgomery R mod r bytes in
`co-noir-spartan/noir-r1cs/noir_proof_scheme.json:7`; rank mapping `mpc-net/src/rep3/mod.rs:29-32`),
(2) EIP-196 BN254 G1 vectors, and (3) the algebraic identities the reference's own test
(`co-jolt/src/poly/commitment/pst13.rs:498-546`) checks.  Python ints are exact, so every value
below is the mathematically unique answer; an independent C restatement (oracle/c/) is
cross-checked against this file.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
import hashlib

# --------------------------------------------------------------------------------------------
# BN254 fields (ark-bn254; SURVEY.md 8 conventions)
# --------------------------------------------------------------------------------------------
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr modulus r
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq modulus p
MONT = 1 << 256
# snarks-core/src/field.rs:5-7
TWO_INV = 0x183227397098d014dc2822db40c0ac2e9419f4243cdcb848a1f0fac9f8000001
assert TWO_INV == (R + 1) // 2 and (2 * TWO_INV) % R == 1
# co-noir-spartan/noir-r1cs/noir_proof_scheme.json:7 starts with LE bytes of R mod r
R_MONT_ONE = MONT % R
assert R_MONT_ONE == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb


def to_mont(x, mod):
    return (x * MONT) % mod


def from_mont(x, mod):
    return (x * pow(MONT, -1, mod)) % mod


def limbs64(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs64(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


# --------------------------------------------------------------------------------------------
# deterministic PRNG used by every harness (SplitMix64 -> rejection sample < modulus)
# --------------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def field(self, mod=R):
        while True:
            v = self.next() | (self.next() << 64) | (self.next() << 128) | ((self.next() & ((1 << 62) - 1)) << 192)
            if v < mod:
                return v


# --------------------------------------------------------------------------------------------
# G1: y^2 = x^3 + 3 over Fq, affine tuples (x, y) or None for infinity (ark-ec short Weierstrass)
# --------------------------------------------------------------------------------------------
G1_GEN = (1, 2)


def g1_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - 3) % P == 0


def g1_neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


# Jacobian helpers so that big loops do not pay one inversion per addition
def _jac_double(Pj):
    X, Y, Z = Pj
    if Z == 0:
        return Pj
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(Pj, q):
    if q is None:
        return Pj
    X1, Y1, Z1 = Pj
    if Z1 == 0:
        return (q[0], q[1], 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = q[0] * Z1Z1 % P
    S2 = q[1] * Z1 * Z1Z1 % P
    if U2 == X1:
        if S2 == Y1:
            return _jac_double(Pj)
        return (1, 1, 0)
    H = (U2 - X1) % P
    HH = H * H % P
    I = 4 * HH % P
    J = H * I % P
    r = 2 * (S2 - Y1) % P
    V = X1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * Y1 * J) % P
    Z3 = ((Z1 + H) * (Z1 + H) - Z1Z1 - HH) % P
    return (X3, Y3, Z3)


def _jac_to_affine(Pj):
    X, Y, Z = Pj
    if Z == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def g1_mul(pt, k):
    """double-and-add; k taken mod r"""
    k %= R
    acc = (1, 1, 0)
    if pt is None or k == 0:
        return None
    for bit in bin(k)[2:]:
        acc = _jac_double(acc)
        if bit == '1':
            acc = _jac_add_affine(acc, pt)
    return _jac_to_affine(acc)


def msm_naive(bases, scalars):
    """C = sum_i s_i * G_i  -- definition of VariableBaseMSM (call sites pst13.rs:286-294,461-469)."""
    acc = None
    for b, s in zip(bases, scalars):
        acc = g1_add(acc, g1_mul(b, s))
    return acc


def msm_pippenger(bases, scalars, c=None):
    """Bucket method as arkworks `msm_bigint` / jolt-core `msm`: signed c-bit digits,
    2^(c-1) buckets per window filled in input order, running-sum reduction, Horner combine.
    Any correct MSM yields the same affine point; this is a second, structurally different check."""
    n = min(len(bases), len(scalars))
    if n == 0:
        return None
    if c is None:
        c = 3 if n < 32 else (n.bit_length() - 1) * 69 // 100 + 2
    nbits = 254
    nwin = (nbits + c - 1) // c + 1
    digits = []
    for s in scalars[:n]:
        s %= R
        d = []
        carry = 0
        for w in range(nwin):
            v = ((s >> (w * c)) & ((1 << c) - 1)) + carry
            if v > (1 << (c - 1)):
                v -= 1 << c
                carry = 1
            else:
                carry = 0
            d.append(v)
        assert carry == 0
        digits.append(d)
    total = (1, 1, 0)
    for w in reversed(range(nwin)):
        for _ in range(c):
            total = _jac_double(total)
        buckets = [(1, 1, 0)] * (1 << (c - 1))
        for i in range(n):
            d = digits[i][w]
            if d > 0:
                buckets[d - 1] = _jac_add_affine(buckets[d - 1], bases[i])
            elif d < 0:
                buckets[-d - 1] = _jac_add_affine(buckets[-d - 1], g1_neg(bases[i]))
        run = (1, 1, 0)
        acc = (1, 1, 0)
        for b in reversed(buckets):
            run = _jac_add_jac(run, b)
            acc = _jac_add_jac(acc, run)
        total = _jac_add_jac(total, acc)
    return _jac_to_affine(total)


def _jac_add_jac(a, b):
    if a[2] == 0:
        return b
    if b[2] == 0:
        return a
    X1, Y1, Z1 = a
    X2, Y2, Z2 = b
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return _jac_double(a)
        return (1, 1, 0)
    H = (U2 - U1) % P
    I = 4 * H * H % P
    J = H * I % P
    r = 2 * (S2 - S1) % P
    V = U1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * S1 * J) % P
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % P
    return (X3, Y3, Z3)


# --------------------------------------------------------------------------------------------
# Rep3 / additive shares (mpc-types/src/protocols/rep3/arithmetic/{types,ops}.rs)
# a share is a tuple (a, b): a = this party's additive share, b = previous party's (types.rs:22-29)
# --------------------------------------------------------------------------------------------
def rep3_share(v, rng):
    """mpc-core/src/protocols/rep3/arithmetic.rs:21-33"""
    t0 = rng.field()
    t1 = rng.field()
    t2 = (v - t0 - t1) % R
    return [(t0, t2), (t1, t0), (t2, t1)]


def rep3_open(shares):
    """rep3 combine: a0 + a1 + a2 (arithmetic.rs:35-37 uses a+b of one party + b of the prev)"""
    return (shares[0][0] + shares[1][0] + shares[2][0]) % R


def rep3_add(x, y):
    return ((x[0] + y[0]) % R, (x[1] + y[1]) % R)


def rep3_sub(x, y):
    return ((x[0] - y[0]) % R, (x[1] - y[1]) % R)


def rep3_mul_public(x, c):
    """ops.rs:80-101"""
    return (x[0] * c % R, x[1] * c % R)


def rep3_local_mul(x, y):
    """Share x Share -> Additive: a.a*b.a + a.a*b.b + a.b*b.a  (ops.rs:71-78)"""
    return (x[0] * y[0] + x[0] * y[1] + x[1] * y[0]) % R


def rep3_into_additive(x):
    """(a+b) * TWO_INV  (types.rs:76-81)"""
    return (x[0] + x[1]) * TWO_INV % R


def rep3_promote_from_trivial(v, party):
    """types.rs:90-96: P0 (v,0), P1 (0,v), P2 (0,0)"""
    return [(v % R, 0), (0, v % R), (0, 0)][party]


def additive_promote_from_trivial(v, party):
    """mpc-types/src/protocols/additive/types.rs:47-53: P0 holds v, others 0"""
    return v % R if party == 0 else 0


def plain_as_share(v):
    """'plain prover' mode: a single party whose share is the value itself.  Used with mode='plain'
    helpers below (share = (v, None))."""
    return v % R


# generic helpers over the two share modes: rep3 shares are 2-tuples, plain values are ints
def sh_add(x, y):
    return rep3_add(x, y) if isinstance(x, tuple) else (x + y) % R


def sh_sub(x, y):
    return rep3_sub(x, y) if isinstance(x, tuple) else (x - y) % R


def sh_mul_public(x, c):
    return rep3_mul_public(x, c) if isinstance(x, tuple) else x * c % R


def sh_local_mul(x, y):
    return rep3_local_mul(x, y) if isinstance(x, tuple) else x * y % R


def sh_zero(like):
    return (0, 0) if isinstance(like, tuple) else 0


def sh_into_additive(x):
    return rep3_into_additive(x) if isinstance(x, tuple) else x % R


# --------------------------------------------------------------------------------------------
# public polynomial helpers (jolt-core, out of tree: EqPolynomial / SplitEqPolynomial / UniPoly;
# restated from SURVEY App. B/C and the in-tree call sites)
# --------------------------------------------------------------------------------------------
def eq_evals(r):
    """EqPolynomial::evals(r): big-endian -- r[0] pairs with the MOST significant index bit
    (use: dense_mlpoly.rs:149-153)."""
    ev = [1]
    for rj in r:
        nxt = []
        for e in ev:
            hi = e * rj % R
            nxt.append((e - hi) % R)
            nxt.append(hi)
        ev = nxt
    return ev


class SplitEq:
    """SplitEqPolynomial (jolt-core poly/split_eq_poly.rs; used dense_interleaved_poly.rs:218-303,
    grand_product.rs:192).  new(w): m = len/2, E2 = evals(w[..m]), E1 = evals(w[m..])."""

    def __init__(self, w):
        m = len(w) // 2
        self.num_vars = len(w)
        self.E2 = eq_evals(w[:m])
        self.E1 = eq_evals(w[m:])
        self.E1_len = len(self.E1)
        self.E2_len = len(self.E2)

    def get_num_vars(self):
        return self.num_vars

    def length(self):
        return self.E1_len * self.E2_len if self.E1_len > 1 else self.E2_len

    def bind(self, r):
        if self.E1_len == 1:
            n = self.E2_len // 2
            for i in range(n):
                self.E2[i] = (self.E2[2 * i] + r * (self.E2[2 * i + 1] - self.E2[2 * i])) % R
            self.E2_len = n
        else:
            n = self.E1_len // 2
            for i in range(n):
                self.E1[i] = (self.E1[2 * i] + r * (self.E1[2 * i + 1] - self.E1[2 * i])) % R
            self.E1_len = n
            if self.E1_len == 1:
                for i in range(self.E2_len):
                    self.E2[i] = self.E2[i] * self.E1[0] % R


def unipoly_from_evals(evals):
    """UniPoly::from_evals: the unique polynomial of degree len-1 through (i, evals[i]).
    Returned low->high coefficients (co-jolt/src/poly/unipoly.rs:5-13 applies it share-wise)."""
    n = len(evals)
    coeffs = [0] * n
    for i in range(n):
        # Lagrange basis l_i(x) = prod_{j!=i} (x - j)/(i - j)
        num = [1]
        den = 1
        for j in range(n):
            if j == i:
                continue
            num = [0] + num  # multiply by x
            for k in range(len(num) - 1):
                num[k] = (num[k] - j * num[k + 1]) % R
            den = den * (i - j) % R
        s = evals[i] * pow(den, -1, R) % R
        for k in range(n):
            coeffs[k] = (coeffs[k] + s * num[k]) % R
    return coeffs


def unipoly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def unipoly_compress(coeffs):
    """CompressedUniPoly: drop the linear term (subprotocols/sumcheck.rs:146-148)"""
    return [coeffs[0]] + list(coeffs[2:])


# --------------------------------------------------------------------------------------------
# harness transcript (the reference's KeccakTranscript lives out of tree in jolt-core and only the
# coordinator hashes -- SURVEY App. C; the harness uses SHA-256, documented in DESIGN.md)
# --------------------------------------------------------------------------------------------
class Transcript:
    def __init__(self, label=b"cozk"):
        self.state = hashlib.sha256(label).digest()
        self.n_rounds = 0

    def _absorb(self, data):
        self.state = hashlib.sha256(self.state + self.n_rounds.to_bytes(4, "little") + data).digest()
        self.n_rounds += 1

    def append_scalar(self, x):
        self._absorb((x % R).to_bytes(32, "little"))

    def append_scalars(self, xs):
        self._absorb(b"".join((x % R).to_bytes(32, "little") for x in xs))

    def append_point(self, pt):
        if pt is None:
            self._absorb(b"\x00" * 64)
        else:
            self._absorb(pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little"))

    def challenge_scalar(self):
        """128-bit challenge like jolt's challenge_scalar (16 bytes)"""
        self._absorb(b"challenge")
        return int.from_bytes(self.state[:16], "little")

    def challenge_vector(self, n):
        return [self.challenge_scalar() for _ in range(n)]


# --------------------------------------------------------------------------------------------
# Rep3DensePolynomial (co-jolt/src/poly/dense_mlpoly.rs)
# --------------------------------------------------------------------------------------------
LOW_TO_HIGH = 0
HIGH_TO_LOW = 1


def dense_bind(coeffs, r, order):
    """dense_mlpoly.rs:310-378 `bind`: out[i] = lo + r*(hi-lo) per share limb.
    LowToHigh pairs (2i,2i+1); HighToLow pairs (i, i+n).  (bind_parallel's unbound HighToLow
    omits `left +` at :446-450 -- SURVEY 9: latent bug, the correct formula is implemented.)"""
    n = len(coeffs) // 2
    out = []
    for i in range(n):
        lo, hi = (coeffs[2 * i], coeffs[2 * i + 1]) if order == LOW_TO_HIGH else (coeffs[i], coeffs[i + n])
        out.append(sh_add(lo, sh_mul_public(sh_sub(hi, lo), r)))
    return out


def dense_sumcheck_evals(coeffs, index, degree, order):
    """dense_mlpoly.rs:113-147: evaluations at 0, 2, 3, ... (degree values)"""
    n = len(coeffs)
    if order == LOW_TO_HIGH:
        e0, e1 = coeffs[2 * index], coeffs[2 * index + 1]
    else:
        e0, e1 = coeffs[index], coeffs[index + n // 2]
    evals = [e0]
    if degree == 1:
        return evals
    m = sh_sub(e1, e0)
    ev = e1
    for _ in range(1, degree):
        ev = sh_add(ev, m)
        evals.append(ev)
    return evals


def dense_evaluate_at_chi(coeffs, chis):
    """dense_mlpoly.rs:160-181: sum_i into_additive(share_i) * chi_i -> additive share"""
    acc = 0
    for c, chi in zip(coeffs, chis):
        acc = (acc + sh_into_additive(c) * chi) % R
    return acc


def dense_batch_evaluate(polys, r):
    """dense_mlpoly.rs:183-192"""
    eq = eq_evals(r)
    return [dense_evaluate_at_chi(p, eq) for p in polys], eq


def dense_linear_combination(polys, coeffs):
    """dense_mlpoly.rs:195-226: sum_k coeff_k * poly_k, shorter polys contribute to their prefix"""
    max_len = max(len(p) for p in polys)
    out = [sh_zero(polys[0][0])] * max_len
    out = list(out)
    for c, p in zip(coeffs, polys):
        for i, v in enumerate(p):
            out[i] = sh_add(out[i], sh_mul_public(v, c))
    return out


def dense_dot_product_with_public(coeffs, other):
    """dense_mlpoly.rs:228-234"""
    acc = sh_zero(coeffs[0])
    for a, b in zip(coeffs, other):
        acc = sh_add(acc, sh_mul_public(a, b))
    return acc


def public_bind(coeffs, r, order):
    n = len(coeffs) // 2
    if order == LOW_TO_HIGH:
        return [(coeffs[2 * i] + r * (coeffs[2 * i + 1] - coeffs[2 * i])) % R for i in range(n)]
    return [(coeffs[i] + r * (coeffs[i + n] - coeffs[i])) % R for i in range(n)]


# --------------------------------------------------------------------------------------------
# Rep3DenseInterleavedPolynomial (co-jolt/src/poly/dense_interleaved_poly.rs)
# --------------------------------------------------------------------------------------------
def _get(chunk, i, zero):
    return chunk[i] if i < len(chunk) else zero


def interleaved_bind(coeffs, r):
    """dense_interleaved_poly.rs:155-195: 4 inputs -> 2 outputs, zero-padded ragged tail;
    new len = next_multiple_of(len,4)/2"""
    zero = sh_zero(coeffs[0])
    out = []
    for c in range(0, len(coeffs), 4):
        ch = coeffs[c:c + 4]
        u = [_get(ch, k, zero) for k in range(4)]
        out.append(sh_add(u[0], sh_mul_public(sh_sub(u[2], u[0]), r)))
        out.append(sh_add(u[1], sh_mul_public(sh_sub(u[3], u[1]), r)))
    return out


def interleaved_uninterleave(coeffs):
    """:104-120"""
    left = coeffs[0::2]
    right = coeffs[1::2]
    if len(right) < len(left):
        right = right + [sh_zero(coeffs[0])]
    return left, right


def _cubic_terms(ch, zero, eq3):
    left = (_get(ch, 0, zero), _get(ch, 2, zero))
    right = (_get(ch, 1, zero), _get(ch, 3, zero))
    m_l = sh_sub(left[1], left[0])
    m_r = sh_sub(right[1], right[0])
    l2 = sh_add(left[1], m_l)
    l3 = sh_add(l2, m_l)
    r2 = sh_add(right[1], m_r)
    r3 = sh_add(r2, m_r)
    return (sh_local_mul(left[0], right[0]) * eq3[0] % R,
            sh_local_mul(l2, r2) * eq3[1] % R,
            sh_local_mul(l3, r3) * eq3[2] % R)


def _eq3(e0, e1):
    m = (e1 - e0) % R
    e2 = (e1 + m) % R
    return (e0, e2, (e2 + m) % R)


def interleaved_compute_cubic_evals(coeffs, eq, previous_claim):
    """dense_interleaved_poly.rs:210-365 -> [g(0), g(1)=claim-g(0), g(2), g(3)] additive."""
    zero = sh_zero(coeffs[0])
    s = [0, 0, 0]
    if eq.E1_len == 1:
        # zip(self.par_chunks(4), eq_poly.E2.par_chunks(2)) -- E2 is the untruncated Vec (:221-222)
        nchunks = min((len(coeffs) + 3) // 4, len(eq.E2) // 2)
        for k in range(nchunks):
            ch = coeffs[4 * k:4 * k + 4]
            t = _cubic_terms(ch, zero, _eq3(eq.E2[2 * k], eq.E2[2 * k + 1]))
            s = [(s[i] + t[i]) % R for i in range(3)]
    else:
        E1e = [_eq3(eq.E1[2 * j], eq.E1[2 * j + 1]) for j in range(eq.E1_len // 2)]
        n = len(coeffs)
        npow = 1 << max(0, (n - 1).bit_length())
        chunk_size = max(npow // eq.E2_len, 1)
        for x2 in range(eq.E2_len):
            Px2 = coeffs[x2 * chunk_size:(x2 + 1) * chunk_size]
            if not Px2:
                break
            inner = [0, 0, 0]
            for j, e3 in enumerate(E1e):
                ch = Px2[4 * j:4 * j + 4]
                if not ch:
                    break
                t = _cubic_terms(ch, zero, e3)
                inner = [(inner[i] + t[i]) % R for i in range(3)]
            s = [(s[i] + inner[i] * eq.E2[x2]) % R for i in range(3)]
    return [s[0], (previous_claim - s[0]) % R, s[1], s[2]]


def interleaved_compute_cubic(coeffs, eq, previous_claim):
    """returns the 4 additive coefficient shares sent to the coordinator (sumcheck.rs:108-110)"""
    return unipoly_from_evals(interleaved_compute_cubic_evals(coeffs, eq, previous_claim))


def interleaved_layer_output_local(coeffs):
    """local part of layer_output (:122-141): out[j] = L[j] x R[j] (additive, unmasked)"""
    left, right = interleaved_uninterleave(coeffs)
    return [sh_local_mul(l, r) for l, r in zip(left, right)]


def rep3_mul_vec(parties_left, parties_right, masks):
    """rep3::arithmetic::mul_vec for all three parties at once (out-of-tree co-snarks; semantics
    SURVEY App. C): c.a = local(a,b) + mask_i ; c.b = c.a of the previous party (ring reshare,
    mpc-core/src/protocols/rep3/arithmetic.rs:144-164).  masks[i][j] must satisfy sum_i = 0."""
    n = len(parties_left[0])
    ca = [[(rep3_local_mul(parties_left[p][j], parties_right[p][j]) + masks[p][j]) % R for j in range(n)]
          for p in range(3)]
    return [[(ca[p][j], ca[(p + 2) % 3][j]) for j in range(n)] for p in range(3)]


def zero_masks(n, rng):
    """three zero-sum mask vectors (mask_i = PRF(k_{i,next}) - PRF(k_{prev,i}))"""
    k = [[rng.field() for _ in range(n)] for _ in range(3)]
    return [[(k[p][j] - k[(p + 2) % 3][j]) % R for j in range(n)] for p in range(3)]


# --------------------------------------------------------------------------------------------
# PST13 (co-jolt/src/poly/commitment/pst13.rs + ark-poly-commit multilinear_pc)
# --------------------------------------------------------------------------------------------
def pst_setup(nv, rng, g=G1_GEN):
    """MultilinearPC::setup: powers_of_g[i] = { g^{eq(t[i..], b)} : b in {0,1}^{nv-i} } with ark's
    little-endian variable order: index bit j of b pairs with t[i + j]."""
    t = [rng.field() for _ in range(nv)]
    powers = []
    for i in range(nv):
        # little-endian eq table over t[i..]: bit j of index <-> t[i+j]
        ev = [1]
        for tj in t[i:]:
            ev = [e * (1 - tj) % R for e in ev] + [e * tj % R for e in ev]
        powers.append([g1_mul(g, e) for e in ev])
    return {"nv": nv, "t": t, "g": g, "powers_of_g": powers}


def pst_commit(ck, evals):
    """pst13.rs:282-296: MSM(powers_of_g[0][..len], evals)"""
    return msm_naive(ck["powers_of_g"][0][:len(evals)], evals)


def pst_open(ck, evals, point):
    """pst13.rs:428-474 `open` (point already reversed by the caller, :134,358)."""
    nv = ck["nv"]
    assert len(evals) == 1 << nv
    r = list(evals)
    proofs = []
    for i in range(nv):
        k = nv - i
        p = point[i]
        half = 1 << (k - 1)
        q = [(r[2 * b + 1] - r[2 * b]) % R for b in range(half)]
        r = [(r[2 * b] * (1 - p) + r[2 * b + 1] * p) % R for b in range(half)]
        scalars = [q[x >> 1] for x in range(1 << k)]
        proofs.append(msm_naive(ck["powers_of_g"][i], scalars))
    return proofs, r[0]


def pst_check_with_trapdoor(ck, commitment, point, value, proofs):
    """Pairing-free restatement of MultilinearPC::check (pst13.rs:367-385): with the trapdoor t
    known, e(C - g^v, h) = prod e(pi_i, h^{t_i - p_i}) becomes C - v*g == sum_i (t_i - p_i) * pi_i."""
    lhs = g1_add(commitment, g1_neg(g1_mul(ck["g"], value)))
    rhs = None
    for ti, pi, prf in zip(ck["t"], point, proofs):
        rhs = g1_add(rhs, g1_mul(prf, (ti - pi) % R))
    return lhs == rhs


def pst_evaluate_le(evals, point):
    """ark DenseMultilinearExtension evaluate: point[0] binds index bit 0"""
    r = list(evals)
    for p in point:
        r = [(r[2 * b] + p * (r[2 * b + 1] - r[2 * b])) % R for b in range(len(r) // 2)]
    return r[0]


# --------------------------------------------------------------------------------------------
# dense grand product: worker + coordinator (co-jolt/src/subprotocols/{grand_product,sumcheck}.rs)
# Runs `nparties` in {1 (plain prover), 3 (Rep3)} in lock-step, playing the coordinator itself
# (message schedule: SURVEY App. D).
# --------------------------------------------------------------------------------------------
def combine_additive(vecs):
    """mpc-core/src/protocols/additive.rs:103-114"""
    return [sum(col) % R for col in zip(*vecs)]


def gp_construct(leaves_per_party, batch_size, mask_rng):
    """Rep3BatchedDenseGrandProduct::construct (grand_product.rs:239-255)."""
    nparties = len(leaves_per_party)
    n = len(leaves_per_party[0])
    assert n % batch_size == 0
    per = n // batch_size
    assert per & (per - 1) == 0
    num_layers = per.bit_length() - 1
    layers = [[list(l) for l in leaves_per_party]]
    for _ in range(num_layers - 1):
        prev = layers[-1]
        if nparties == 1:
            layers.append([interleaved_layer_output_local(prev[0])])
        else:
            lr = [interleaved_uninterleave(prev[p]) for p in range(3)]
            masks = zero_masks(len(lr[0][0]), mask_rng)
            layers.append(rep3_mul_vec([x[0] for x in lr], [x[1] for x in lr], masks))
    return layers


def gp_claimed_outputs(layers):
    """grand_product.rs:266-272: last layer chunks(2) -> L x R (additive)"""
    out = []
    for top in layers[-1]:
        out.append([sh_local_mul(top[2 * i], top[2 * i + 1]) for i in range(len(top) // 2)])
    return out


def gp_prove(layers, transcript, record=None):
    """prove_grand_product_worker + cooridinate_prove_grand_product (grand_product.rs:56-130),
    prove_layer / coordinate_prove_layer (:143-217), prove_sumcheck (sumcheck.rs:96-131,134-165).
    Returns (proof dict, r_grand_product)."""
    nparties = len(layers[0])
    outs = gp_claimed_outputs(layers)
    outputs = combine_additive(outs)
    transcript.append_scalars(outputs)
    # DensePolynomial::new_padded(outputs).evaluate(r)
    padded = list(outputs)
    while len(padded) & (len(padded) - 1):
        padded.append(0)
    nv_out = len(padded).bit_length() - 1
    r = transcript.challenge_vector(nv_out)
    claim_pub = sum(e * v for e, v in zip(eq_evals(r), padded)) % R
    claims = [additive_promote_from_trivial(claim_pub, p) for p in range(nparties)]
    proof = {"outputs": outputs, "layers": []}
    work = [[list(l) for l in layer] for layer in layers]
    for layer in reversed(work):
        eqs = [SplitEq(r) for _ in range(nparties)]
        num_rounds = eqs[0].get_num_vars()
        r_sumcheck = []
        round_polys = []
        prev_claims = list(claims)
        for _round in range(num_rounds):
            msgs = [interleaved_compute_cubic(layer[p], eqs[p], prev_claims[p]) for p in range(nparties)]
            if record is not None:
                record.append(("cubic", [list(m) for m in msgs]))
            poly = combine_additive(msgs)
            comp = unipoly_compress(poly)
            transcript.append_scalars(comp)
            r_j = transcript.challenge_scalar()
            r_sumcheck.append(r_j)
            nxt = unipoly_eval(poly, r_j)
            for p in range(nparties):
                layer[p] = interleaved_bind(layer[p], r_j)
                eqs[p].bind(r_j)
            prev_claims = [additive_promote_from_trivial(nxt, p) for p in range(nparties)]
            round_polys.append(comp)
        # final claims (sumcheck.rs:53-75: coordinator sums the `.a` components)
        finals = [(layer[p][0], layer[p][1]) for p in range(nparties)]
        if nparties == 3:
            left = sum(f[0][0] for f in finals) % R
            right = sum(f[1][0] for f in finals) % R
        else:
            left, right = finals[0][0] % R, finals[0][1] % R
        transcript.append_scalar(left)
        transcript.append_scalar(right)
        r = list(reversed(r_sumcheck))
        r_layer = transcript.challenge_scalar()
        # worker: claim = add_mul_public(left, right-left, r_layer).into_additive() (:211-213)
        claims = [sh_into_additive(sh_add(finals[p][0], sh_mul_public(sh_sub(finals[p][1], finals[p][0]), r_layer)))
                  for p in range(nparties)]
        r.append(r_layer)
        proof["layers"].append({"round_polys": round_polys, "left": left, "right": right})
    return proof, r


def gp_verify(proof, batch_size, transcript):
    """plain verifier for the dense GKR proof: replays the transcript and checks every round
    g(0)+g(1)=claim and the layer reduction; returns (final_claim, r)."""
    outputs = proof["outputs"]
    transcript.append_scalars(outputs)
    padded = list(outputs)
    while len(padded) & (len(padded) - 1):
        padded.append(0)
    nv_out = len(padded).bit_length() - 1
    r = transcript.challenge_vector(nv_out)
    claim = sum(e * v for e, v in zip(eq_evals(r), padded)) % R
    for lp in proof["layers"]:
        rs = []
        e = claim
        for comp in lp["round_polys"]:
            # decompress: c1 = e - 2*c0 - c2 - c3
            c0 = comp[0]
            rest = comp[1:]
            c1 = (e - 2 * c0 - sum(rest)) % R
            poly = [c0, c1] + list(rest)
            transcript.append_scalars(comp)
            r_j = transcript.challenge_scalar()
            rs.append(r_j)
            e = unipoly_eval(poly, r_j)
        left, right = lp["left"], lp["right"]
        # eq(r_layer_point, rs_reversed) * left * right must equal e
        eqv = 1
        for a, b in zip(r, reversed(rs)):
            eqv = eqv * ((a * b + (1 - a) * (1 - b)) % R) % R
        if eqv * left % R * right % R != e:
            return None
        transcript.append_scalar(left)
        transcript.append_scalar(right)
        r = list(reversed(rs))
        r_layer = transcript.challenge_scalar()
        claim = (left + r_layer * (right - left)) % R
        r.append(r_layer)
    return claim, r


# --------------------------------------------------------------------------------------------
# opening accumulator (co-jolt/src/poly/opening_proof.rs)
# --------------------------------------------------------------------------------------------
def opening_compute_quadratic(openings, coeffs, remaining_rounds, previous_claim):
    """opening_proof.rs:364-437.  openings: list of dicts {poly (bound share list), eq (bound public
    list), point_len, claim (share)}.  Returns UniPoly coeffs (low->high) from evals [e0, prev-e0, e2]."""
    e0s, e2s = [], []
    for op in openings:
        if remaining_rounds <= op["point_len"]:
            poly, eq = op["poly"], op["eq"]
            half = len(poly) // 2
            ev0 = 0
            ev2 = 0
            for i in range(half):
                ev0 = (ev0 + sh_into_additive(sh_mul_public(poly[i], eq[i]))) % R
                pb = sh_sub(sh_add(poly[i + half], poly[i + half]), poly[i])
                eb = (2 * eq[i + half] - eq[i]) % R
                ev2 = (ev2 + sh_into_additive(sh_mul_public(pb, eb))) % R
            e0s.append(ev0)
            e2s.append(ev2)
        else:
            rem = remaining_rounds - op["point_len"] - 1
            sc = sh_into_additive(op["claim"]) * (1 << rem) % R
            e0s.append(sc)
            e2s.append(sc)
    c0 = sum(e * c for e, c in zip(e0s, coeffs)) % R
    c2 = sum(e * c for e, c in zip(e2s, coeffs)) % R
    return unipoly_from_evals([c0, (previous_claim - c0) % R, c2])


def opening_reduce(openings_per_party, transcript, record=None):
    """prove_batch_opening_reduction + coordinator loop (opening_proof.rs:181-235,293-361).
    openings_per_party[p] = list of {poly, eq, point, claim}.  Returns (r, combined claims, proof)."""
    nparties = len(openings_per_party)
    rho = transcript.challenge_scalar()
    nopen = len(openings_per_party[0])
    rho_pows = [1]
    for _ in range(1, nopen):
        rho_pows.append(rho_pows[-1] * rho % R)
    state = []
    for p in range(nparties):
        st = []
        for op in openings_per_party[p]:
            st.append({"poly": list(op["poly"]), "eq": list(op["eq"]), "point_len": len(op["point"]),
                       "claim": op["claim"], "nv": len(op["poly"]).bit_length() - 1})
        state.append(st)
    max_nv = max(o["nv"] for o in state[0])
    es = []
    for p in range(nparties):
        e = 0
        for c, o in zip(rho_pows, state[p]):
            cl = o["claim"]
            if o["nv"] != max_nv:
                cl = sh_mul_public(cl, 1 << (max_nv - o["nv"]))
            e = (e + sh_into_additive(cl) * c) % R
        es.append(e)
    r = []
    comps = []
    for rnd in range(max_nv):
        remaining = max_nv - rnd
        msgs = [opening_compute_quadratic(state[p], rho_pows, remaining, es[p]) for p in range(nparties)]
        if record is not None:
            record.append(("quad", [list(m) for m in msgs]))
        poly = combine_additive(msgs)
        comp = unipoly_compress(poly)
        transcript.append_scalars(comp)
        r_j = transcript.challenge_scalar()
        r.append(r_j)
        new_claim = unipoly_eval(poly, r_j)
        es = [additive_promote_from_trivial(new_claim, p) for p in range(nparties)]
        for p in range(nparties):
            for o in state[p]:
                if remaining <= o["point_len"]:
                    o["eq"] = public_bind(o["eq"], r_j, HIGH_TO_LOW)
                    o["poly"] = dense_bind(o["poly"], r_j, HIGH_TO_LOW)
        comps.append(comp)
    claims = combine_additive([[sh_into_additive(o["poly"][0]) for o in state[p]] for p in range(nparties)])
    transcript.append_scalars(claims)
    return r, claims, {"round_polys": comps, "claims": claims, "rho": rho}


# --------------------------------------------------------------------------------------------
# synthetic data streams shared with the engine's `cozk_vec_fill_random`
# (co-zkvms_amd/csrc/capi.hip k_fill_random_*): element i has its own SplitMix64 stream
# --------------------------------------------------------------------------------------------
_STREAM_MUL = 0xD1342543DE82EF95


def synthetic_fr(seed, n, max_bits=0):
    out = []
    for i in range(n):
        v = SplitMix64((seed + i * _STREAM_MUL) & 0xFFFFFFFFFFFFFFFF).field(R)
        if 0 < max_bits < 254:
            v &= (1 << max_bits) - 1
        out.append(v)
    return out


def synthetic_small(seed, n, bits):
    return [SplitMix64((seed + i * _STREAM_MUL) & 0xFFFFFFFFFFFFFFFF).next() & ((1 << bits) - 1) for i in range(n)]


# --------------------------------------------------------------------------------------------
# ark-serialize uncompressed encodings (the wire format of mpc-net, mpc-net/src/rep3/quic/worker.rs:187-219, and of the
# proof structs PST13Commitment{nv, g_product} / Proof{proofs}, co-jolt/src/poly/commitment/pst13.rs:397-401).
# From upstream knowledge of ark-serialize / ark-ec 0.5 (out of tree; the reference holds no serialized bytes to pin
# them): Fr = 32-byte LE canonical integer; usize / Vec length = u64 LE; G1Affine uncompressed = x || y with SWFlags in
# the two spare top bits of the last byte: bit 6 = infinity (x = y = 0), bit 7 = y > -y (`to_flags()` is passed to
# y.serialize_with_flags in both modes); readers ignore bit 7 when uncompressed.
# --------------------------------------------------------------------------------------------
def ser_u64(v):
    return int(v).to_bytes(8, "little")


def ser_fr(x):
    return (x % R).to_bytes(32, "little")


def ser_vec_fr(v):
    return ser_u64(len(v)) + b"".join(ser_fr(x) for x in v)


def ser_g1(pt):
    if pt is None:
        return b"\x00" * 63 + b"\x40"
    x, y = pt
    out = bytearray(x.to_bytes(32, "little") + y.to_bytes(32, "little"))
    if y > (P - y) % P:
        out[63] |= 0x80
    return bytes(out)


def deser_g1(b):
    """inverse of ser_g1 with arkworks' Validate::Yes checks: canonical coordinates, flags, on-curve"""
    assert len(b) == 64
    flags = b[63] & 0xC0
    x = int.from_bytes(b[:32], "little")
    y = int.from_bytes(b[32:], "little") & ((1 << 254) - 1)
    if flags == 0xC0 or x >= P or y >= P:
        raise ValueError("invalid point encoding")
    if flags & 0x40:
        if x or y:
            raise ValueError("infinity with non-zero coordinates")
        return None
    if not g1_is_on_curve((x, y)):
        raise ValueError("not on the curve")
    return (x, y)


# --------------------------------------------------------------------------------------------
# keyed PRF of the engine (co-zkvms_amd/csrc/prf.hip.hpp): element j of a stream is one ChaCha12 block.
# The reference draws shares and masks from ChaCha12 streams keyed with 32-byte seeds the parties exchange
# (mpc-types/src/protocols/rep3.rs:29,177; mpc-core/src/protocols/rep3/network.rs:190-211); random access
# per element replaces the sequential stream.  ChaCha itself is pinned by the RFC 8439 2.3.2 block vector
# (20 rounds; tests/test_oracle.py) -- the 12-round variant differs only in the loop count.
# --------------------------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def _rotl32(v, n):
    return ((v << n) | (v >> (32 - n))) & _M32


def _chacha_qr(x, a, b, c, d):
    x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl32(x[d] ^ x[a], 16)
    x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl32(x[b] ^ x[c], 12)
    x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl32(x[d] ^ x[a], 8)
    x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl32(x[b] ^ x[c], 7)


def chacha_block_words(state, rounds):
    """the ChaCha block function on a 16-word state: `rounds` / 2 double rounds + feed-forward"""
    x = list(state)
    for _ in range(rounds // 2):
        _chacha_qr(x, 0, 4, 8, 12); _chacha_qr(x, 1, 5, 9, 13); _chacha_qr(x, 2, 6, 10, 14); _chacha_qr(x, 3, 7, 11, 15)
        _chacha_qr(x, 0, 5, 10, 15); _chacha_qr(x, 1, 6, 11, 12); _chacha_qr(x, 2, 7, 8, 13); _chacha_qr(x, 3, 4, 9, 14)
    return [(a + b) & _M32 for a, b in zip(x, state)]


_CHACHA_CONST = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
_PRF_DOMAIN = 0x4B5A4F43  # "COZK"


def prf_block(key, counter, attempt):
    """prf.hip.hpp chacha12_block: state = constants | key | counter (64 bit) | domain | attempt"""
    assert len(key) == 32
    k = [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)]
    st = _CHACHA_CONST + k + [counter & _M32, (counter >> 32) & _M32, _PRF_DOMAIN, attempt]
    return chacha_block_words(st, 12)


def prf_fr(key, j):
    """PRF(key, j): the first of (words 0..7, words 8..15) of block (j, attempt), top word masked to 30 bits,
    that is below r; next attempt if neither is.  Canonical integer (the engine returns its Montgomery form)."""
    attempt = 0
    while True:
        w = prf_block(key, j, attempt)
        for half in range(2):
            ws = w[8 * half:8 * half + 8]
            ws[7] &= 0x3FFFFFFF
            v = sum(x << (32 * i) for i, x in enumerate(ws))
            if v < R:
                return v
        attempt += 1


def prf_fr_vec(key, counter, n):
    return [prf_fr(key, counter + i) for i in range(n)]


def harness_prf_key(seed, idx):
    """keys of the synthetic harness runs (csrc/host/prover.hpp harness_prf_key): a real host supplies the 32-byte
    seeds its parties exchanged; the harness expands (run seed, key index) so all parties agree without an exchange"""
    s = (seed ^ (0xC0DEC0DE + idx * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    g = SplitMix64(s)
    return b"".join(g.next().to_bytes(8, "little") for _ in range(4))


def rep3_share_vec(v, key0, key1, counter=0):
    """cozk_rep3_share_vec for all three parties (rep3::share_field_element, mpc-core/src/protocols/rep3/
    arithmetic.rs:21-33): t0 = PRF(key0, .), t1 = PRF(key1, .), t2 = v - t0 - t1; P0 (t0, t2), P1 (t1, t0), P2 (t2, t1)"""
    t0 = prf_fr_vec(key0, counter, len(v))
    t1 = prf_fr_vec(key1, counter, len(v))
    t2 = [(a - b - c) % R for a, b, c in zip(v, t0, t1)]
    return [list(zip(t0, t2)), list(zip(t1, t0)), list(zip(t2, t1))]


# --------------------------------------------------------------------------------------------
# generic product sumcheck (prove_arbitrary_worker) and co-noir-spartan rounds
# --------------------------------------------------------------------------------------------
def prod_round_evals(polys, degree):
    """co-jolt/src/subprotocols/sumcheck.rs:189-215 with comb_func = product of the polynomials' evaluations
    (at most one shared factor): sum over i of prod_j sumcheck_evals(P_j, i, degree, HighToLow)[e];
    points 0, 2, 3, .., degree.  Shared x public = mul_public then into_additive."""
    half = len(polys[0]) // 2
    acc = [0] * degree
    for i in range(half):
        evs = [dense_sumcheck_evals(p, i, degree, HIGH_TO_LOW) for p in polys]
        for e in range(degree):
            prod = None
            for ev in evs:
                v = ev[e]
                if prod is None:
                    prod = v
                elif isinstance(prod, tuple):
                    prod = rep3_mul_public(prod, v)
                elif isinstance(v, tuple):
                    prod = rep3_mul_public(v, prod)
                else:
                    prod = prod * v % R
            acc[e] = (acc[e] + sh_into_additive(prod)) % R
    return acc


def spartan_first_round_evals(za, zb, zc, pub):
    """co-noir-spartan/co-spartan/src/sumcheck.rs:171-280 (before the additive mask): X = 0..3"""
    half = len(pub) // 2
    out = [0, 0, 0, 0]
    for b in range(half):
        a0, b0, c0, p0 = za[2 * b], zb[2 * b], zc[2 * b], pub[2 * b]
        sa, sb, sc = sh_sub(za[2 * b + 1], a0), sh_sub(zb[2 * b + 1], b0), sh_sub(zc[2 * b + 1], c0)
        sp = (pub[2 * b + 1] - p0) % R
        for t in range(4):
            term = sh_local_mul(a0, b0) * p0 - sh_into_additive(sh_mul_public(c0, p0))
            out[t] = (out[t] + term) % R
            a0, b0, c0, p0 = sh_add(a0, sa), sh_add(b0, sb), sh_add(c0, sc), (p0 + sp) % R
    return out


def spartan_second_round_evals(z, pa, pb, pc, coef):
    """co-noir-spartan/co-spartan/src/sumcheck.rs:282-395 (before the Rep3 mask): X = 0..2, share-valued"""
    half = len(pa) // 2
    out = [sh_zero(z[0])] * 3
    out = list(out)
    for b in range(half):
        z0 = z[2 * b]
        sz = sh_sub(z[2 * b + 1], z0)
        a0, b0, c0 = pa[2 * b], pb[2 * b], pc[2 * b]
        sa, sb, sc = (pa[2 * b + 1] - a0) % R, (pb[2 * b + 1] - b0) % R, (pc[2 * b + 1] - c0) % R
        for t in range(3):
            lin = (a0 * coef[0] + b0 * coef[1] + c0 * coef[2]) % R
            out[t] = sh_add(out[t], sh_mul_public(z0, lin))
            z0, a0, b0, c0 = sh_add(z0, sz), (a0 + sa) % R, (b0 + sb) % R, (c0 + sc) % R
    return out


def sparse_matvec(entries, z, nrows):
    """co-noir-spartan/co-spartan/src/worker.rs:153-182: out[row] += z[col] * val for (row, col, val) in entries"""
    out = [sh_zero(z[0])] * nrows
    out = list(out)
    for row, col, val in entries:
        out[row] = sh_add(out[row], sh_mul_public(z[col], val))
    return out


# --------------------------------------------------------------------------------------------
# K11: memory-checking leaf fingerprints
# --------------------------------------------------------------------------------------------
def fingerprint_leaves(cols, col_coeffs, polys, poly_coeffs, constant, party=None):
    """compute_leaves (co-jolt/src/jolt/vm/bytecode/worker.rs:57-100, read_write_memory/worker.rs:207-300):
    leaf[i] = sum_k c_k * cols[k][i] + sum_j d_j * polys[j][i] + constant.  cols: lists of small integers (compact
    public columns); polys: lists of ints (public) or (a, b) tuples (Rep3 shares); party None = plain prover,
    otherwise the public part enters through add_public (party 0: a, party 1: b; rep3::arithmetic::add_public)."""
    n = min([len(c) for c in cols] + [len(p) for p in polys])
    out = []
    for i in range(n):
        pub = constant % R
        for c, col in zip(col_coeffs, cols):
            pub = (pub + c * col[i]) % R
        sa = sb = 0
        for d, p in zip(poly_coeffs, polys):
            v = p[i]
            if isinstance(v, tuple):
                sa = (sa + d * v[0]) % R
                sb = (sb + d * v[1]) % R
            else:
                pub = (pub + d * v) % R
        if party is None:
            assert sb == 0
            out.append((sa + pub) % R)
        else:
            out.append(((sa + (pub if party == 0 else 0)) % R, (sb + (pub if party == 1 else 0)) % R))
    return out
