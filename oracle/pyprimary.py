"""ORACLE (test infrastructure -- never imported by the product): exact big-integer restatement of Lasso's PRIMARY
SUMCHECK of co-jolt's instruction lookups,

    sum_x eq(r, x) * ( sum_i flag_i(x) * g_i(E_1(x), .., E_alpha(x)) - lookup_output(x) ) = 0,

following the reference:

  prove_primary_sumcheck_inner / primary_sumcheck_prover_message / precompute_evals
                                   co-jolt/src/jolt/vm/instruction_lookups/worker.rs:375-720 (LowToHigh binding, `degree`
                                   evaluations at 0, 2, .., degree per round; the coordinator inserts claim - e(0))
  prove_primary_sumcheck_rep3      co-jolt/src/jolt/vm/instruction_lookups/coordinator.rs:97-150
  combine_lookups_rep3_batched     the three collation FORMS a sha2-style trace needs:
      CONCAT   (AND / OR / XOR / ADD-like)   co-jolt/src/jolt/instruction/and.rs:89-101 ->
                                             concatenate_lookups_rep3_batched, co-jolt/src/utils/instruction_utils.rs:26-47
      PRODUCT  (BEQ-like)                    co-jolt/src/jolt/instruction/beq.rs:106-130 -> rep3::arithmetic::product_many,
                                             mpc-core/src/protocols/rep3/arithmetic.rs:86-102
      LTU      (SLTU-like)                   co-jolt/src/jolt/instruction/sltu.rs:139-170 (mul_vec chain over the EQ
                                             subtable evaluations, reshare_additive_many of the sum)
  utils/future.rs:48-69 (fufill_batched) is what batches the multiplications of one level into one mul_vec.

Three parties run in lock-step; `mul` = local product (+ a zero-sharing mask, zero here: masks cancel in every sum the
coordinator forms) followed by the ring reshare (c.b = previous party's c.a).  Plain prover: one party, shares are ints.
The instruction SET itself (27 RV32I instructions, their subtables and memory maps) lives out of tree in jolt-core: the
harness takes the instruction table as an input (form + memory list per instruction), parity unpinned beyond that."""
import pyref as O

R = O.R

CONCAT, PRODUCT, LTU = 0, 1, 2


class Instr:
    """form: CONCAT (mems = C memories, `bits` = operand bits per chunk), PRODUCT (mems = the factors),
    LTU (mems = C LTU memories followed by C - 1 EQ memories)"""

    def __init__(self, form, mems, bits=0):
        self.form, self.mems, self.bits = form, list(mems), bits

    def g_degree(self):
        if self.form == CONCAT:
            return 1
        if self.form == PRODUCT:
            return len(self.mems)
        return (len(self.mems) + 1) // 2  # LTU: C


def sumcheck_degree(instrs):
    """sumcheck_poly_degree (worker.rs:701-708): max g degree + 2 (eq and flag)"""
    return max(i.g_degree() for i in instrs) + 2


# ------------------------------------------------------------------------------------------------ rep3 helpers
def _mul_vec(xs, ys):
    """rep3::arithmetic::mul_vec for all parties: xs[p][j] x ys[p][j] -> Rep3 shares (zero masks)"""
    np_ = len(xs)
    if np_ == 1:
        return [[x * y % R for x, y in zip(xs[0], ys[0])]]
    ca = [[O.rep3_local_mul(x, y) for x, y in zip(xs[p], ys[p])] for p in range(3)]
    return [[(ca[p][j], ca[(p + 2) % 3][j]) for j in range(len(ca[p]))] for p in range(3)]


def _reshare_additive_many(adds):
    """reshare_additive_many (mpc-core/src/protocols/rep3/arithmetic.rs:152-164): (a, previous party's a)"""
    if len(adds) == 1:
        return [list(adds[0])]
    return [[(adds[p][j], adds[(p + 2) % 3][j]) for j in range(len(adds[p]))] for p in range(3)]


def combine_lookups_batched(instr, vals):
    """combine_lookups_rep3_batched: vals[p][m][j] = party p's evaluation of the instruction's m-th memory for item j
    -> per party a list of Rep3 shares (plain: values) of g over the items"""
    np_ = len(vals)
    n = len(vals[0][0])
    if instr.form == CONCAT:  # instruction_utils.rs:26-47
        C = len(instr.mems)
        shift = 1 << instr.bits
        out = []
        for p in range(np_):
            sums = list(vals[p][C - 1])
            weight = shift
            for m in range(C - 2, -1, -1):
                sums = [O.sh_add(s, O.sh_mul_public(v, weight)) for s, v in zip(sums, vals[p][m])]
                weight = weight * shift % R
            out.append(sums)
        return out
    if instr.form == PRODUCT:  # product_many: fold of mul_vec
        acc = [list(vals[p][0]) for p in range(np_)]
        for m in range(1, len(instr.mems)):
            acc = _mul_vec(acc, [vals[p][m] for p in range(np_)])
        return acc
    # LTU (sltu.rs:139-170)
    C = (len(instr.mems) + 1) // 2
    ltu = [[vals[p][i] for i in range(C)] for p in range(np_)]
    eq = [[vals[p][C + i] for i in range(C - 1)] for p in range(np_)]
    sums = [[O.sh_into_additive(x) for x in ltu[p][0]] for p in range(np_)]
    eq_prods = [list(eq[p][0]) for p in range(np_)]
    for i in range(1, C - 1):
        for p in range(np_):
            sums[p] = [(s + O.sh_local_mul(l, e)) % R for s, l, e in zip(sums[p], ltu[p][i], eq_prods[p])]
        eq_prods = _mul_vec(eq_prods, [eq[p][i] for p in range(np_)])
    fin = [[(s + O.sh_local_mul(l, e)) % R for s, l, e in zip(sums[p], ltu[p][C - 1], eq_prods[p])] for p in range(np_)]
    return _reshare_additive_many(fin)


def g_plain(instr, e):
    """g_i on plain values e[m] of its memories"""
    if instr.form == CONCAT:
        C = len(instr.mems)
        return sum(e[m] << (instr.bits * (C - 1 - m)) for m in range(C)) % R
    if instr.form == PRODUCT:
        v = 1
        for x in e:
            v = v * x % R
        return v
    C = (len(instr.mems) + 1) // 2
    s, prod = 0, 1
    for i in range(C):
        s = (s + e[i] * prod) % R
        if i < C - 1:
            prod = prod * e[C + i] % R
    return s


# ------------------------------------------------------------------------------------------------ one round
def prover_message(instrs, eq, flags, E, outs):
    """primary_sumcheck_prover_message (worker.rs:454-598): eq, flags public coefficient lists; E[p][m], outs[p] the parties'
    share lists -> per party `degree` additive evaluations at 0, 2, .., degree"""
    np_ = len(E)
    degree = sumcheck_degree(instrs)
    half = len(eq) // 2
    L2H = O.LOW_TO_HIGH
    eq_ev = [O.dense_sumcheck_evals(eq, i, degree, L2H) for i in range(half)]
    out_ev = [[O.dense_sumcheck_evals(outs[p], i, degree, L2H) for i in range(half)] for p in range(np_)]
    flag_ev = [[O.dense_sumcheck_evals(f, i, degree, L2H) for f in flags] for i in range(half)]
    used = [[[k for k, v in enumerate(fe) if v % R != 0] for fe in flag_ev[i]] for i in range(half)]
    inner = [[[0] * degree for _ in range(half)] for _ in range(np_)]
    for ii, instr in enumerate(instrs):
        items = [(i, k) for i in range(half) for k in used[i][ii]]
        if not items:
            continue
        vals = []
        for p in range(np_):
            per_mem = []
            for m in instr.mems:
                ev_cache = {}
                col = []
                for (i, k) in items:
                    if i not in ev_cache:
                        ev_cache[i] = O.dense_sumcheck_evals(E[p][m], i, degree, L2H)
                    col.append(ev_cache[i][k])
                per_mem.append(col)
            vals.append(per_mem)
        coll = combine_lookups_batched(instr, vals)
        for p in range(np_):
            for j, (i, k) in enumerate(items):
                inner[p][i][k] = (inner[p][i][k] + O.sh_into_additive(coll[p][j]) * flag_ev[i][ii][k]) % R
    msgs = []
    for p in range(np_):
        ev = [0] * degree
        for i in range(half):
            for k in range(degree):
                ev[k] = (ev[k] + (inner[p][i][k] - O.sh_into_additive(out_ev[p][i][k])) * eq_ev[i][k]) % R
        msgs.append(ev)
    return msgs


def from_evals_with_claim(evals, claim):
    """coordinator.rs:131-132: insert claim - e(0) at position 1, interpolate through 0, 1, 2, .., degree"""
    pts = [evals[0], (claim - evals[0]) % R] + list(evals[1:])
    return O.unipoly_from_evals(pts)


def prove(instrs, r_eq, flags, E, outs, transcript):
    """prove_primary_sumcheck_inner (worker.rs:375-452) + prove_primary_sumcheck_rep3 (coordinator.rs:97-150); returns
    (proof dict, r (in round order), final evals dict)"""
    np_ = len(E)
    eq = O.eq_evals(r_eq)
    flags = [list(f) for f in flags]
    E = [[list(m) for m in E[p]] for p in range(np_)]
    outs = [list(outs[p]) for p in range(np_)]
    num_rounds = len(r_eq)
    claim = 0
    rs, comps = [], []
    for _ in range(num_rounds):
        msgs = prover_message(instrs, eq, flags, E, outs)
        total = O.combine_additive(msgs)
        poly = from_evals_with_claim(total, claim)
        comp = O.unipoly_compress(poly)
        transcript.append_scalars(comp)
        r_j = transcript.challenge_scalar()
        rs.append(r_j)
        claim = O.unipoly_eval(poly, r_j)
        comps.append(comp)
        eq = O.public_bind(eq, r_j, O.LOW_TO_HIGH)
        flags = [O.public_bind(f, r_j, O.LOW_TO_HIGH) for f in flags]
        E = [[O.dense_bind(m, r_j, O.LOW_TO_HIGH) for m in E[p]] for p in range(np_)]
        outs = [O.dense_bind(outs[p], r_j, O.LOW_TO_HIGH) for p in range(np_)]
    # final claims (worker.rs:427-452): E(r), flags(r), lookup_outputs(r) -- additive, combined by the coordinator
    E_evals = O.combine_additive([[O.sh_into_additive(m[0]) for m in E[p]] for p in range(np_)])
    flag_evals = [f[0] for f in flags]
    out_eval = sum(O.sh_into_additive(outs[p][0]) for p in range(np_)) % R
    openings = E_evals + flag_evals + [out_eval]
    transcript.append_scalars(openings)
    return {"round_polys": comps, "openings": openings}, rs, {"claim": claim, "eq": eq[0]}


def verify(instrs, r_eq, proof, n_mem, transcript):
    """plain verifier of the primary sumcheck (jolt-core InstructionLookupsProof::verify_primary_sumcheck, out of tree):
    replay, then claim == eq(r_eq, r) * (sum_i flag_i(r) g_i(E(r)) - out(r)).  Returns the point (round order) or None."""
    degree = sumcheck_degree(instrs)
    claim = 0
    rs = []
    for comp in proof["round_polys"]:
        if len(comp) != degree:
            return None
        c1 = (claim - 2 * comp[0] - sum(comp[1:])) % R
        poly = [comp[0], c1] + list(comp[1:])
        transcript.append_scalars(comp)
        r_j = transcript.challenge_scalar()
        rs.append(r_j)
        claim = O.unipoly_eval(poly, r_j)
    op = proof["openings"]
    transcript.append_scalars(op)
    E_evals, flag_evals, out_eval = op[:n_mem], op[n_mem:n_mem + len(instrs)], op[-1]
    # LowToHigh binding: round j binds index bit j, i.e. the LAST variable first: the point is the reversed challenge list
    pt = list(reversed(rs))
    eqv = 1
    for a, b in zip(r_eq, pt):
        eqv = eqv * ((a * b + (1 - a) * (1 - b)) % R) % R
    acc = 0
    for f, instr in zip(flag_evals, instrs):
        acc = (acc + f * g_plain(instr, [E_evals[m] for m in instr.mems])) % R
    if eqv * ((acc - out_eval) % R) % R != claim:
        return None
    return rs
