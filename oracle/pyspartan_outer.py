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
