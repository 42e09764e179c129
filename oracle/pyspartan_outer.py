"""ORACLE (test infrastructure -- never imported by the product): exact big-integer restatement of co-jolt's Spartan OUTER
sumcheck over the sparse interleaved Az / Bz / Cz,

    sum_x eq(tau, x) * (Az(x) * Bz(x) - Cz(x)) = 0,

following the reference:

  Rep3SpartanInterleavedPolynomial::new            co-jolt/src/poly/spartan_interleaved_poly.rs:40-172 (uniform and cross-step
                                                   constraints evaluated row by row into (index, SharedOrPublic) lists)
  first_sumcheck_round / subsequent_sumcheck_round :189-385, :386-612 (quadratic evaluations at 0 and infinity per block of 6,
                                                   Gruen split-eq weights, sparse binding)
  final_sumcheck_evals                             :648-664
  process_eq_sumcheck_round_worker / coordinate_eq_sumcheck_round   co-jolt/src/subprotocols/sumcheck_spartan.rs:14-79
  prove_spartan_cubic_sumcheck                     co-jolt/src/r1cs/spartan/worker.rs:277-300
  SharedOrPublic                                   co-jolt/src/utils/shared_or_public.rs:15-290
  LC::evaluate_row_rep3_mixed                      co-jolt/src/r1cs/ops.rs:18-37

Out of tree (jolt-core, restated from upstream knowledge; parity unpinned): GruenSplitEqPolynomial, UniPoly::
from_linear_times_quadratic_with_hint, the r1cs builder types (Constraint, OffsetEqConstraint, LC) -- the constraint
system is an INPUT here (the concrete Jolt constraints live in jolt-core).

Values: ('P', v) public, ('S', share) shared (share = (a, b) tuple, or an int for the plain prover), ('A', v) additive."""
import pyref as O

R = O.R


# ------------------------------------------------------------------------------------------------ SharedOrPublic
def _add_public_share(x, c, party):
    if not isinstance(x, tuple):
        return (x + c) % R
    if party == 0:
        return ((x[0] + c) % R, x[1])
    if party == 1:
        return (x[0], (x[1] + c) % R)
    return x


def _add_public_additive(x, c, party):
    return (x + c) % R if party == 0 else x % R


def sp_zero_public():
    return ("P", 0)


def sp_into_additive(x, party):
    if x[0] == "P":
        return x[1] % R if party == 0 else 0
    if x[0] == "S":
        return O.sh_into_additive(x[1])
    return x[1] % R


def sp_add(x, y, party):
    kx, ky = x[0], y[0]
    if kx == "S" and ky == "S":
        return ("S", O.sh_add(x[1], y[1]))
    if kx == "S" and ky == "P":
        return ("S", _add_public_share(x[1], y[1], party))
    if kx == "P" and ky == "S":
        return ("S", _add_public_share(y[1], x[1], party))
    if kx == "P" and ky == "P":
        return ("P", (x[1] + y[1]) % R)
    if kx == "A" and ky == "A":
        return ("A", (x[1] + y[1]) % R)
    if kx == "A" and ky == "P":
        return ("A", _add_public_additive(x[1], y[1], party))
    if kx == "P" and ky == "A":
        return ("A", _add_public_additive(y[1], x[1], party))
    if kx == "A" and ky == "S":
        return ("A", (x[1] + O.sh_into_additive(y[1])) % R)
    return ("A", (O.sh_into_additive(x[1]) + y[1]) % R)


def sp_neg(x):
    if x[0] == "S":
        return ("S", O.sh_sub(O.sh_zero(x[1]), x[1]))
    return (x[0], (-x[1]) % R)


def sp_sub(x, y, party):
    return sp_add(x, sp_neg(y), party)


def sp_mul(x, y):
    kx, ky = x[0], y[0]
    if kx == "P" and ky == "P":
        return ("P", x[1] * y[1] % R)
    if kx == "S" and ky == "P":
        return ("S", O.sh_mul_public(x[1], y[1]))
    if kx == "P" and ky == "S":
        return ("S", O.sh_mul_public(y[1], x[1]))
    if kx == "S" and ky == "S":
        return ("A", O.sh_local_mul(x[1], y[1]))
    if kx == "A" and ky == "P":
        return ("A", x[1] * y[1] % R)
    if kx == "P" and ky == "A":
        return ("A", x[1] * y[1] % R)
    raise ValueError("multiplication of additive shares is not allowed")


def sp_mul_public(x, c):
    return sp_mul(x, ("P", c % R))


def sp_mul_mul_public(x, y, c):
    kx, ky = x[0], y[0]
    if kx == "S" and ky == "S":
        return ("A", O.sh_local_mul(x[1], y[1]) * c % R)
    if kx == "S" and ky == "P":
        return ("A", O.sh_into_additive(x[1]) * (y[1] * c % R) % R)
    if kx == "P" and ky == "S":
        return ("A", O.sh_into_additive(y[1]) * (x[1] * c % R) % R)
    return sp_mul(x, sp_mul(y, ("P", c % R)))


def sp_is_public_zero(x):
    return x[0] == "P" and x[1] % R == 0


# ------------------------------------------------------------------------------------------------ constraints
def eval_lc(lc, polys, row, party):
    """LC::evaluate_row_rep3_mixed (r1cs/ops.rs:18-37): lc = [(var or None, coeff)]; polys[var] = ('P', list) | ('S', list)"""
    acc = sp_zero_public()
    for var, coeff in lc:
        if var is None:
            term = ("P", coeff % R)
        else:
            kind, col = polys[var]
            term = sp_mul_public((kind, col[row]), coeff % R)
        acc = sp_add(acc, term, party)
    return acc


def lc_constant(lc):
    return sum(c for v, c in lc if v is None) % R


def eval_offset_lc(olc, polys, step, next_step, party):
    """eval_offset_lc_rep3_mixed (spartan_interleaved_poly.rs:666-684): olc = (offset flag, lc)"""
    off, lc = olc
    if not off:
        return eval_lc(lc, polys, step, party)
    if next_step is not None:
        return eval_lc(lc, polys, next_step, party)
    return ("P", lc_constant(lc))


def build_sparse(uniform, cross, polys, padded, num_steps, party):
    """Rep3SpartanInterleavedPolynomial::new (:40-172) -> sorted list of (index, value); dense_len = num_steps * padded"""
    coeffs = []
    for step in range(num_steps):
        for ci, (a, b, c) in enumerate(uniform):
            gi = 3 * (step * padded + ci)
            az = sp_zero_public()
            if a:
                az = eval_lc(a, polys, step, party)
                if not sp_is_public_zero(az):
                    coeffs.append((gi, az))
            bz = sp_zero_public()
            if b:
                bz = eval_lc(b, polys, step, party)
                if not sp_is_public_zero(bz):
                    coeffs.append((gi + 1, bz))
            if sp_is_public_zero(az) and sp_is_public_zero(bz):
                continue
            if az[0] == "S" and bz[0] == "S":
                coeffs.append((gi + 2, eval_lc(c, polys, step, party)))
            else:
                coeffs.append((gi + 2, sp_mul(az, bz)))
        nxt = step + 1 if step + 1 < num_steps else None
        for ci, (a, b, cond) in enumerate(cross):
            gi = 3 * (step * padded + len(uniform) + ci)
            az = sp_sub(eval_offset_lc(a, polys, step, nxt, party), eval_offset_lc(b, polys, step, nxt, party), party)
            coeffs.append((gi, az))
            if az[0] == "P" and az[1] % R != 0:
                continue
            coeffs.append((gi + 1, eval_offset_lc(cond, polys, step, nxt, party)))
    return coeffs


# ------------------------------------------------------------------------------------------------ Gruen split-eq
class GruenSplitEq:
    """jolt-core GruenSplitEqPolynomial (out of tree): w = [w_out | w_in | w_last]; tables over w_out / w_in (cached for
    every prefix), the variable being bound is handled as a linear factor, bound variables fold into current_scalar"""

    def __init__(self, w):
        n = len(w)
        m = n // 2
        self.w = list(w)
        self.current_index = n
        self.current_scalar = 1
        w_out, w_in = w[:m], w[m:n - 1]
        self.E_out_vec = [O.eq_evals(w_out[:k]) for k in range(len(w_out) + 1)]
        self.E_in_vec = [O.eq_evals(w_in[:k]) for k in range(len(w_in) + 1)]

    def E_in_current(self):
        return self.E_in_vec[-1]

    def E_out_current(self):
        return self.E_out_vec[-1]

    def bind(self, r):
        w = self.w[self.current_index - 1]
        self.current_scalar = self.current_scalar * ((1 - w - r + 2 * w * r) % R) % R
        self.current_index -= 1
        if len(self.w) // 2 < self.current_index:
            self.E_in_vec.pop()
        elif 0 < self.current_index:
            self.E_out_vec.pop()


def cubic_from_linear_times_quadratic_with_hint(l0, l1, t0, tinf, hint):
    """UniPoly::from_linear_times_quadratic_with_hint: s(X) = (l0 + l1 X)(t0 + t1 X + tinf X^2) with s(0) + s(1) = hint.
    Linear in (t0, tinf, hint): applied to additive shares it gives additive shares of the coefficients."""
    inv = pow((l0 + l1) % R, -1, R)
    t1 = ((hint - l0 * t0) * inv - t0 - tinf) % R
    return [l0 * t0 % R, (l0 * t1 + l1 * t0) % R, (l0 * tinf + l1 * t1) % R, l1 * tinf % R]


# ------------------------------------------------------------------------------------------------ rounds
def _blocks(coeffs):
    out, cur = [], []
    for c in coeffs:
        if cur and cur[0][0] // 6 != c[0] // 6:
            out.append(cur)
            cur = []
        cur.append(c)
    if cur:
        out.append(cur)
    return out


def quadratic_evals(coeffs, eq, party, first):
    """the (t(0), t(infinity)) of a round (:200-275 first round: t(0) = 0 by construction; :410-505 afterwards)"""
    E_in, E_out = eq.E_in_current(), eq.E_out_current()
    nbits = len(E_in).bit_length() - 1
    mask = (1 << nbits) - 1
    t0 = tinf = 0
    for blk in _blocks(coeffs):
        bi = blk[0][0] // 6
        e = E_out[bi >> nbits] * E_in[bi & mask] % R
        block = [sp_zero_public()] * 6
        for idx, val in blk:
            block[idx % 6] = val
        az, bz, cz0 = (block[0], block[3]), (block[1], block[4]), block[2]
        azi, bzi = sp_sub(az[1], az[0], party), sp_sub(bz[1], bz[0], party)
        if first:
            if sp_is_public_zero(azi) and sp_is_public_zero(bzi):
                continue
            tinf = (tinf + sp_into_additive(sp_mul_mul_public(azi, bzi, 1), party) * e) % R
        else:
            t0 = (t0 + (sp_into_additive(sp_mul(az[0], bz[0]), party) - sp_into_additive(cz0, party)) * e) % R
            tinf = (tinf + sp_into_additive(sp_mul(azi, bzi), party) * e) % R
    return t0, tinf


def bind_sparse(coeffs, r, party):
    """the binding pass of either round (:289-370, :520-600): low + r (high - low) for each of Az, Bz, Cz present in a block"""
    out = []
    for blk in _blocks(coeffs):
        bi = blk[0][0] // 6
        pair = {0: [None, None], 1: [None, None], 2: [None, None]}
        for idx, val in blk:
            pair[idx % 3][(idx % 6) // 3] = val
        for k in range(3):
            lo, hi = pair[k]
            if lo is None and hi is None:
                continue
            lo = lo if lo is not None else sp_zero_public()
            hi = hi if hi is not None else sp_zero_public()
            out.append((3 * bi + k, sp_add(lo, sp_mul_public(sp_sub(hi, lo, party), r), party)))
    return out


def final_evals(coeffs, party):
    """final_sumcheck_evals (:648-664)"""
    ev = [0, 0, 0]
    for idx, val in coeffs[:3]:
        if idx < 3:
            ev[idx] = sp_into_additive(val, party)
    return ev


def prove(uniform, cross, polys_per_party, padded, num_steps, tau, transcript):
    """prove_spartan_cubic_sumcheck (r1cs/spartan/worker.rs:277-300) + the coordinator's coordinate_eq_sumcheck_round
    loop; returns (proof dict, challenges in round order)"""
    np_ = len(polys_per_party)
    coeffs = [build_sparse(uniform, cross, polys_per_party[p], padded, num_steps, p) for p in range(np_)]
    eqs = [GruenSplitEq(tau) for _ in range(np_)]
    claims = [0] * np_
    rs, comps = [], []
    for rnd in range(len(tau)):
        msgs = []
        for p in range(np_):
            t0, tinf = quadratic_evals(coeffs[p], eqs[p], p, rnd == 0)
            eq = eqs[p]
            sw = eq.current_scalar * eq.w[eq.current_index - 1] % R
            msgs.append(cubic_from_linear_times_quadratic_with_hint((eq.current_scalar - sw) % R, (2 * sw - eq.current_scalar) % R, t0, tinf, claims[p]))
        poly = O.combine_additive(msgs)
        comp = O.unipoly_compress(poly)
        transcript.append_scalars(comp)
        r_i = transcript.challenge_scalar()
        rs.append(r_i)
        nxt = O.unipoly_eval(poly, r_i)
        comps.append(comp)
        for p in range(np_):
            claims[p] = O.additive_promote_from_trivial(nxt, p)
            eqs[p].bind(r_i)
            coeffs[p] = bind_sparse(coeffs[p], r_i, p)
    fin = O.combine_additive([final_evals(coeffs[p], p) for p in range(np_)])
    transcript.append_scalars(fin)
    return {"round_polys": comps, "claims": fin}, rs


def verify(proof, tau, transcript):
    """the verifier's outer-sumcheck check (jolt-core UniformSpartanProof::verify, out of tree): replay, then
    claim == eq(tau, r) (Az(r) Bz(r) - Cz(r)) at the reversed challenge list.  Returns the challenges or None."""
    claim = 0
    rs = []
    for comp in proof["round_polys"]:
        if len(comp) != 3:
            return None
        c1 = (claim - 2 * comp[0] - sum(comp[1:])) % R
        poly = [comp[0], c1] + list(comp[1:])
        transcript.append_scalars(comp)
        r_i = transcript.challenge_scalar()
        rs.append(r_i)
        claim = O.unipoly_eval(poly, r_i)
    az, bz, cz = proof["claims"]
    transcript.append_scalars(proof["claims"])
    pt = list(reversed(rs))
    eqv = 1
    for a, b in zip(tau, pt):
        eqv = eqv * ((a * b + (1 - a) * (1 - b)) % R) % R
    if eqv * ((az * bz - cz) % R) % R != claim:
        return None
    return rs


def dense_azbzcz(uniform, cross, plain_cols, padded, num_steps):
    """Az, Bz, Cz in the clear as dense vectors (row = step * padded + constraint): what the claims must be the MLEs of"""
    L = num_steps * padded
    az, bz, cz = [0] * L, [0] * L, [0] * L
    polys = [("P", c) for c in plain_cols]
    for step in range(num_steps):
        nxt = step + 1 if step + 1 < num_steps else None
        for ci, (a, b, c) in enumerate(uniform):
            row = step * padded + ci
            az[row] = eval_lc(a, polys, step, 0)[1] if a else 0
            bz[row] = eval_lc(b, polys, step, 0)[1] if b else 0
            cz[row] = az[row] * bz[row] % R
        for ci, (a, b, cond) in enumerate(cross):
            row = step * padded + len(uniform) + ci
            az[row] = (eval_offset_lc(a, polys, step, nxt, 0)[1] - eval_offset_lc(b, polys, step, nxt, 0)[1]) % R
            bz[row] = eval_offset_lc(cond, polys, step, nxt, 0)[1]
    return az, bz, cz


# ------------------------------------------------------------------------------------------------ the harness instance
def synthetic_system():
    """csrc/host/outer_harness.hpp outer_build_system: 5 uniform + 2 cross-step constraints over 14 columns, 8 rows per step"""
    C = None
    uniform = [([(0, 1), (1, 2), (C, 3)], [(2, 1), (3, -1)], [(6, 1)]),
               ([(4, 1), (C, 1)], [(1, 1), (2, 1)], [(7, 1)]),
               ([(4, 1)], [(5, 1), (C, -1)], [(8, 1)]),
               ([(5, 1)], [(0, 1)], [(9, 1)]),
               ([], [(1, 1)], [])]
    cross = [((False, [(11, 1)]), (True, [(10, 1)]), (False, [(12, 1)])),
             ((False, [(0, 1), (C, 5)]), (True, [(13, 1), (C, 5)]), (False, [(12, 1)]))]
    return uniform, cross, 8


IS_PUBLIC = [False, False, False, False, True, True, False, False, True, False, False, False, True, False]


def synthetic_columns(seed, n):
    """outer_build_clear: the dealer's view of the 14 columns"""
    c = [None] * 14
    for v in range(4):
        c[v] = O.synthetic_fr(seed + 100 * (v + 1), n)
    c[4] = O.synthetic_small(seed + 501, n, 8)
    c[5] = O.synthetic_small(seed + 502, n, 1)
    c[12] = O.synthetic_small(seed + 503, n, 1)
    c[12][n - 1] = 0
    c[11] = O.synthetic_fr(seed + 600, n)
    c[6] = [(c[0][x] + 2 * c[1][x] + 3) * (c[2][x] - c[3][x]) % R for x in range(n)]
    c[7] = [(c[4][x] + 1) * (c[1][x] + c[2][x]) % R for x in range(n)]
    c[8] = [c[4][x] * (c[5][x] - 1) % R for x in range(n)]
    c[9] = [c[5][x] * c[0][x] % R for x in range(n)]
    c[10] = [7 if x == 0 else c[11][x - 1] for x in range(n)]
    c[13] = [9 if x == 0 else c[0][x - 1] for x in range(n)]
    return c


def party_columns(seed, cols, nparties):
    """per party the ('P' | 'S', list) columns: shared columns through rep3_share_vec with the harness keys"""
    out = [[None] * 14 for _ in range(nparties)]
    for v in range(14):
        if IS_PUBLIC[v]:
            for p in range(nparties):
                out[p][v] = ("P", cols[v])
        elif nparties == 1:
            out[0][v] = ("S", cols[v])
        else:
            s = seed + 100 * (v + 1)
            sh = O.rep3_share_vec(cols[v], O.harness_prf_key(s, 101), O.harness_prf_key(s, 102))
            for p in range(3):
                out[p][v] = ("S", sh[p])
    return out


def run(cfg):
    """the pipeline of csrc/host/outer_harness.hpp: returns proof bytes + verified"""
    import hashlib
    nparties = 1 if cfg["mode"] == "plain" else 3
    n = 1 << cfg["log_steps"]
    uniform, cross, padded = synthetic_system()
    cols = synthetic_columns(cfg["seed"], n)
    polys = party_columns(cfg["seed"], cols, nparties)
    tr = O.Transcript(b"cozk-spartan-outer")
    nv = cfg["log_steps"] + 3
    tau = tr.challenge_vector(nv)
    proof, rs = prove(uniform, cross, polys, padded, n, tau, tr)
    vt = O.Transcript(b"cozk-spartan-outer")
    vtau = vt.challenge_vector(nv)
    v = verify(proof, vtau, vt)
    ok = v == rs
    if ok:
        az, bz, cz = dense_azbzcz(uniform, cross, cols, padded, n)
        eq = O.eq_evals(list(reversed(rs)))
        ok = [sum(a * b for a, b in zip(eq, vec)) % R for vec in (az, bz, cz)] == proof["claims"]
    blob = O.ser_u64(len(proof["round_polys"])) + b"".join(O.ser_vec_fr(c) for c in proof["round_polys"]) + O.ser_vec_fr(proof["claims"])
    return {"proof_bytes": blob, "digest": hashlib.sha256(blob).hexdigest(), "verified": ok, "proof": proof, "rs": rs}


# ================================================================================================ the WHOLE Spartan worker
# Rep3UniformSpartanProver::prove (co-jolt/src/r1cs/spartan/worker.rs:63-273) + Rep3UniformSpartanCoordinator::prove_rep3
# (r1cs/spartan/coordinator.rs:27-136): outer sumcheck (above), inner sumcheck over y = (shift bit | const bit | variable),
# shift sumcheck over the steps, and the two batch_evaluate / append claim exchanges.
#
# Out of tree (jolt-core, restated from upstream knowledge; parity unpinned): EqPlusOnePolynomial::evals,
# UniformSpartanKey::evaluate_matrix_mle_partial / evaluate_matrix_mle_full / evaluate_z_mle_with_segment_evals and the plain
# verifier UniformSpartanProof::verify.  They are fixed here by the identities the in-tree worker relies on:
#   Az(rx) + rlc Bz(rx) + rlc^2 Cz(rx) = sum_y ABC(rx_constr, y) z(y || rx_step)        (worker.rs:107-170)
# with z's constant column = 1 in the non-shifted half only (bind_z[num_vars_uniform] = 1, :154).
import pyjolt_r1cs as J


def eq_plus_one_evals(r):
    """EqPlusOnePolynomial::evals(r, None) -> (eq(r, .), eq_plus_one(r, .)): eq_plus_one(x, y) = 1 iff y = x + 1 as
    big-endian integers, x < 2^l - 1 (no wrap-around), multilinear in x.  For y != 0 with k trailing zero bits:
    x = y - 1 has its low k bits set, bit k clear and the high bits of y: value = prod_{low k} r_j * (1 - r_k) * eq(r_high, y_high)"""
    l = len(r)
    n = 1 << l
    eq = O.eq_evals(r)
    out = [0] * n
    for y in range(1, n):
        k = (y & -y).bit_length() - 1
        v = 1
        for j in range(k):  # low bits: r index l - 1 - j
            v = v * r[l - 1 - j] % R
        v = v * ((1 - r[l - 1 - k]) % R) % R
        hi = y >> (k + 1)
        m = l - k - 1
        for j in range(m):
            bit = (hi >> (m - 1 - j)) & 1
            v = v * (r[j] if bit else (1 - r[j]) % R) % R
        out[y] = v
    return eq, out


def matrix_mle_partial(uniform, cross, padded, nvars_padded, rx_constr, rlc):
    """key.evaluate_matrix_mle_partial(rx_constr, rx_step, rlc) (used worker.rs:123-126): the 4 V entries of
    ABC(rx_constr, .) = A + rlc B + rlc^2 C, V = nvars_padded: [variables | constant at V | shifted variables | (unused)]"""
    V = nvars_padded
    eq = O.eq_evals(rx_constr)
    out = [0] * (4 * V)
    r1, r2 = rlc % R, rlc * rlc % R

    def add(lc, row, w, shifted=False, sign=1):
        for var, c in lc:
            col = V if var is None else (var + (2 * V if shifted else 0))
            out[col] = (out[col] + sign * c * w % R * eq[row]) % R

    for ci, (a, b, c) in enumerate(uniform):
        add(a, ci, 1)
        add(b, ci, r1)
        add(c, ci, r2)
    for ci, (a, b, cond) in enumerate(cross):
        row = len(uniform) + ci
        # Az = a - b, Bz = cond, Cz = 0; an offset LC reads the NEXT step: its variables land in the shifted half, its
        # constant stays in the (non-shifted) constant column (sum_t eq(rx_step, t) = 1; at the last step an offset LC is
        # its constant, eval_offset_lc_rep3_mixed spartan_interleaved_poly.rs:666-684, and eq_plus_one has no wrap-around)
        for (off, lcx), w, sign in ((a, 1, 1), (b, 1, -1), (cond, r1, 1)):
            for var, cf in lcx:
                col = V if var is None else (var + (2 * V if off else 0))
                out[col] = (out[col] + sign * cf * w % R * eq[row]) % R
    return out


# ---- MixedPolynomial (co-jolt/src/poly/mixed_polynomial.rs:12-193)
def mixed_sumcheck_evals(evals, index, degree, party):
    """sumcheck_evals, HighToLow (:48-59): values at 0, 2, .., degree"""
    half = len(evals) // 2
    out = [evals[index]]
    if degree == 1:
        return out
    ev = evals[index + half]
    m = sp_sub(ev, out[0], party)
    for _ in range(1, degree):
        ev = sp_add(ev, m, party)
        out.append(ev)
    return out


def mixed_bind_top(evals, r, party):
    """bound_poly_var_top (:78-89)"""
    n = len(evals) // 2
    return [sp_add(evals[i], sp_mul_public(sp_sub(evals[i + n], evals[i], party), r), party) for i in range(n)]


def unipoly_from_evals_deg2(e0, e1, e2):
    return O.unipoly_from_evals([e0, e1, e2])


def prove_arbitrary_mixed(claims, num_rounds, polys_per_party, transcript):
    """prove_arbitrary_worker (subprotocols/sumcheck.rs:168-246) with comb_func = (p0 * p1).into_additive, degree 2, over
    MixedPolynomials, all parties in lock step with coordinate_prove_arbitrary (:134-165).  claims: per party additive.
    Returns (compressed polys, r, per party final evals as additive)"""
    np_ = len(polys_per_party)
    prev = list(claims)
    comps, rs = [], []
    for _ in range(num_rounds):
        msgs = []
        for p in range(np_):
            polys = polys_per_party[p]
            half = len(polys[0]) // 2
            acc = [0, 0]
            for i in range(half):
                ev = [mixed_sumcheck_evals(pl, i, 2, p) for pl in polys]
                for j in range(2):
                    acc[j] = (acc[j] + sp_into_additive(sp_mul(ev[0][j], ev[1][j]), p)) % R
            pts = [acc[0], (prev[p] - acc[0]) % R, acc[1]]
            msgs.append(O.unipoly_from_evals(pts))
        poly = O.combine_additive(msgs)
        comp = O.unipoly_compress(poly)
        transcript.append_scalars(comp)
        r_j = transcript.challenge_scalar()
        rs.append(r_j)
        comps.append(comp)
        nxt = O.unipoly_eval(poly, r_j)
        for p in range(np_):
            prev[p] = O.additive_promote_from_trivial(nxt, p)
            polys_per_party[p] = [mixed_bind_top(pl, r_j, p) for pl in polys_per_party[p]]
    finals = [[sp_into_additive(pl[0], p) for pl in polys_per_party[p]] for p in range(np_)]
    return comps, rs, finals


def _dot_public(col, pub, party):
    """Rep3MultilinearPolynomial::dot_product_with_public (multilinear_polynomial.rs; dense_mlpoly.rs:228-234): a public
    polynomial gives a public value, a shared one a Rep3 share"""
    kind, vals = col
    if kind == "P":
        return ("P", sum(v * w for v, w in zip(vals, pub)) % R)
    acc = O.sh_zero(vals[0])
    for v, w in zip(vals, pub):
        acc = O.sh_add(acc, O.sh_mul_public(v, w))
    return ("S", acc)


def _batch_evaluate_additive(polys, chis, party):
    """batch_evaluate + into_additive of every claim (worker.rs:243-272)"""
    return [sp_into_additive(_dot_public(col, chis, party), party) for col in polys]


def receive_claims(parts, transcript):
    """Rep3ProverOpeningAccumulator::receive_claims (opening_proof.rs:108-128) -> (claims, rho, batched claim)"""
    claims = O.combine_additive(parts)
    rho = transcript.challenge_scalar()
    pw, batched = 1, 0
    for c in claims:
        batched = (batched + pw * c) % R
        pw = pw * rho % R
    return claims, rho, batched


def prove_full(uniform, cross, padded, polys_per_party, num_steps, transcript, appends=None):
    """the whole Rep3UniformSpartanProver::prove with its coordinator.  Returns the proof dict.  appends(point) -> claims: the
    opening accumulator of a surrounding flow (oracle/pyflow.py); None: the two claim exchanges alone."""
    np_ = len(polys_per_party)
    nvars = len(polys_per_party[0])
    steps_bits = num_steps.bit_length() - 1
    constr_bits = padded.bit_length() - 1
    V = 1
    while V < nvars:
        V <<= 1
    tau = transcript.challenge_vector(steps_bits + constr_bits)
    outer, rs = prove(uniform, cross, polys_per_party, padded, num_steps, tau, transcript)
    outer_r = list(reversed(rs))
    rx_step, rx_constr = outer_r[:steps_bits], outer_r[steps_bits:]
    rlc = transcript.challenge_scalar()
    az, bz, cz = outer["claims"]
    claim_inner = (az + rlc * bz + rlc * rlc * cz) % R
    eq_step, eqp1_step = eq_plus_one_evals(rx_step)
    abc = matrix_mle_partial(uniform, cross, padded, V, rx_constr, rlc)
    inner_polys = []
    for p in range(np_):
        bind_z = [sp_zero_public()] * (2 * V)
        bind_shift = [sp_zero_public()] * (2 * V)
        for i, col in enumerate(polys_per_party[p]):
            bind_z[i] = _dot_public(col, eq_step, p)
            bind_shift[i] = _dot_public(col, eqp1_step, p)
        bind_z[V] = ("P", 1)
        inner_polys.append([[("P", v) for v in abc], bind_z + bind_shift])
    inner_rounds = (4 * V).bit_length() - 1
    inner_comps, inner_r, _ = prove_arbitrary_mixed([O.additive_promote_from_trivial(claim_inner, p) for p in range(np_)], inner_rounds, inner_polys, transcript)
    # shift sumcheck
    ry_var = inner_r[1:]
    eq_ry = O.eq_evals(ry_var)
    shift_polys, shift_claims = [], []
    for p in range(np_):
        zry = []
        for t in range(num_steps):
            acc = sp_zero_public()
            for i, (kind, vals) in enumerate(polys_per_party[p]):
                acc = sp_add(acc, sp_mul_public((kind, vals[t]), eq_ry[i]), p)  # scale_coeff + sum_for (worker.rs:196-205)
            zry.append(acc)
        shift_polys.append([zry, [("P", v) for v in eqp1_step]])
        shift_claims.append(sum(sp_into_additive(sp_mul(a, b), p) for a, b in zip(zry, shift_polys[p][1])) % R)
    shift_claim = sum(shift_claims) % R  # combine_additive_share; NOT appended to the transcript (coordinator.rs:113-117)
    shift_comps, shift_r, _ = prove_arbitrary_mixed(shift_claims, steps_bits, shift_polys, transcript)
    if appends is not None:
        witness_evals, shift_evals = appends(rx_step), appends(shift_r)
        rho1 = rho2 = batched1 = batched2 = None
    else:
        chis1 = O.eq_evals(rx_step)
        parts = [_batch_evaluate_additive(polys_per_party[p], chis1, p) for p in range(np_)]
        witness_evals, rho1, batched1 = receive_claims(parts, transcript)
        chis2 = O.eq_evals(shift_r)
        parts = [_batch_evaluate_additive(polys_per_party[p], chis2, p) for p in range(np_)]
        shift_evals, rho2, batched2 = receive_claims(parts, transcript)
    return {"outer": outer, "outer_r": rs, "inner_polys": inner_comps, "inner_r": inner_r, "shift_claim": shift_claim, "shift_polys": shift_comps,
            "shift_r": shift_r, "witness_evals": witness_evals, "shift_witness_evals": shift_evals, "rlc": rlc, "tau": tau,
            "rho": [rho1, rho2], "batched": [batched1, batched2]}


def _verify_rounds(comps, claim, degree, transcript):
    rs = []
    for comp in comps:
        if len(comp) != degree:
            return None, None
        c1 = (claim - 2 * comp[0] - sum(comp[1:])) % R
        poly = [comp[0], c1] + list(comp[1:])
        transcript.append_scalars(comp)
        r_j = transcript.challenge_scalar()
        rs.append(r_j)
        claim = O.unipoly_eval(poly, r_j)
    return claim, rs


def _eq_point(a, b):
    e = 1
    for x, y in zip(a, b):
        e = e * ((x * y + (1 - x) * (1 - y)) % R) % R
    return e


def eq_plus_one_point(x, y):
    """EqPlusOnePolynomial::evaluate (x, y big-endian points): sum over k of [low k bits: x = 1, y = 0][bit k: x = 0, y = 1][high: equal]"""
    l = len(x)
    acc = 0
    for k in range(l):
        v = 1
        for j in range(k):
            v = v * x[l - 1 - j] % R * ((1 - y[l - 1 - j]) % R) % R
        v = v * ((1 - x[l - 1 - k]) % R) % R * y[l - 1 - k] % R
        for j in range(l - k - 1):
            v = v * ((x[j] * y[j] + (1 - x[j]) * (1 - y[j])) % R) % R
        acc = (acc + v) % R
    return acc


def verify_full(proof, uniform, cross, padded, nvars, num_steps, transcript):
    """the plain verifier's sumcheck checks (jolt-core UniformSpartanProof::verify, out of tree): returns True / False.
    The opened witness evaluations themselves are checked by the opening proof (or, in the harness, against the dealer's
    columns)."""
    steps_bits = num_steps.bit_length() - 1
    constr_bits = padded.bit_length() - 1
    V = 1
    while V < nvars:
        V <<= 1
    tau = transcript.challenge_vector(steps_bits + constr_bits)
    rs = verify(proof["outer"], tau, transcript)
    if rs is None:
        return False
    outer_r = list(reversed(rs))
    rx_step, rx_constr = outer_r[:steps_bits], outer_r[steps_bits:]
    rlc = transcript.challenge_scalar()
    az, bz, cz = proof["outer"]["claims"]
    claim = (az + rlc * bz + rlc * rlc * cz) % R
    inner_rounds = (4 * V).bit_length() - 1
    if len(proof["inner_polys"]) != inner_rounds:
        return False
    fin, inner_r = _verify_rounds(proof["inner_polys"], claim, 2, transcript)
    if fin is None:
        return False
    ry_var = inner_r[1:]
    eq_ry = O.eq_evals(ry_var)
    z_eval = (sum(eq_ry[i] * proof["witness_evals"][i] for i in range(nvars)) + eq_ry[V]) % R
    z_comb = ((1 - inner_r[0]) * z_eval + inner_r[0] * proof["shift_claim"]) % R
    abc = matrix_mle_partial(uniform, cross, padded, V, rx_constr, rlc)
    eq_y = O.eq_evals(inner_r)
    abc_eval = sum(a * e for a, e in zip(abc, eq_y)) % R
    if abc_eval * z_comb % R != fin:
        return False
    if len(proof["shift_polys"]) != steps_bits:
        return False
    fin, shift_r = _verify_rounds(proof["shift_polys"], proof["shift_claim"], 2, transcript)
    if fin is None:
        return False
    z_shift = sum(eq_ry[i] * proof["shift_witness_evals"][i] for i in range(nvars)) % R
    if z_shift * eq_plus_one_point(rx_step, shift_r) % R != fin:
        return False
    # receive_claims x 2
    for _ in range(2):
        transcript.challenge_scalar()
    return rx_step, shift_r


def jolt_party_columns(seed, cols, nparties):
    """per party the ('P' | 'S', list) columns of the Jolt-shaped system: shared columns through rep3_share_vec with the
    harness keys (seed + 100 (v + 1), 101 / 102)"""
    nv = len(cols)
    out = [[None] * nv for _ in range(nparties)]
    for v in range(nv):
        if J.IS_PUBLIC[v]:
            for p in range(nparties):
                out[p][v] = ("P", cols[v])
        elif nparties == 1:
            out[0][v] = ("S", cols[v])
        else:
            s = seed + 100 * (v + 1)
            sh = O.rep3_share_vec(cols[v], O.harness_prf_key(s, 101), O.harness_prf_key(s, 102))
            for p in range(3):
                out[p][v] = ("S", sh[p])
    return out


def serialize_full(proof):
    blob = O.ser_u64(len(proof["outer"]["round_polys"])) + b"".join(O.ser_vec_fr(c) for c in proof["outer"]["round_polys"]) + O.ser_vec_fr(proof["outer"]["claims"])
    blob += O.ser_u64(len(proof["inner_polys"])) + b"".join(O.ser_vec_fr(c) for c in proof["inner_polys"])
    blob += O.ser_fr(proof["shift_claim"])
    blob += O.ser_u64(len(proof["shift_polys"])) + b"".join(O.ser_vec_fr(c) for c in proof["shift_polys"])
    blob += O.ser_vec_fr(proof["witness_evals"]) + O.ser_vec_fr(proof["shift_witness_evals"])
    return blob


def run_full(cfg):
    """the pipeline of csrc/host/outer_harness.hpp with cfg.system = jolt (full = 1): the whole Spartan worker on the
    Jolt-shaped constraint system; returns proof bytes + verified"""
    import hashlib
    nparties = 1 if cfg["mode"] == "plain" else 3
    n = 1 << cfg["log_steps"]
    if cfg.get("system", "jolt") == "jolt":
        uniform, cross, padded = J.build_system()
        cols = J.synthetic_columns(cfg["seed"], n)
        polys = jolt_party_columns(cfg["seed"], cols, nparties)
    else:
        uniform, cross, padded = synthetic_system()
        cols = synthetic_columns(cfg["seed"], n)
        polys = party_columns(cfg["seed"], cols, nparties)
    tr = O.Transcript(b"cozk-spartan")
    proof = prove_full(uniform, cross, padded, polys, n, tr)
    vt = O.Transcript(b"cozk-spartan")
    v = verify_full(proof, uniform, cross, padded, len(cols), n, vt)
    ok = bool(v)
    if ok:
        rx_step, shift_r = v
        for point, claims in ((rx_step, proof["witness_evals"]), (shift_r, proof["shift_witness_evals"])):
            eq = O.eq_evals(point)
            ok = ok and [sum(a * b for a, b in zip(eq, c)) % R for c in cols] == claims
    blob = serialize_full(proof)
    return {"proof_bytes": blob, "digest": hashlib.sha256(blob).hexdigest(), "verified": ok, "proof": proof}
