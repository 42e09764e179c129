"""CPU tier for the whole co-jolt Spartan worker's oracle (oracle/pyjolt_r1cs.py, oracle/pyspartan_outer.py prove_full):
the reference's constraint set (co-jolt/src/r1cs/constraints.rs:39-257) in its table form, the synthetic satisfying trace, and
the identities the out-of-tree pieces (EqPlusOnePolynomial, evaluate_matrix_mle_partial) are fixed by.  Parity unpinned by
reference outputs (the reference holds no Spartan fixture); these pin the restatement to the algebra the in-tree worker
relies on (r1cs/spartan/worker.rs:100-235)."""
import pyjolt_r1cs as J
import pyref as O
import pyspartan_outer as S

R = O.R


def test_constraint_set_shape_and_satisfaction():
    uniform, cross, padded = J.build_system()
    # 27 + 12 binary, pack_be, 2 operands, 5 load / store / lui, add, sub, product + mul, move, assert, 2 concat, 4 x (relevant
    # chunk + query chunk), 2 x (product + conditional write), next_pc_jump, should_branch, next_pc (constraints.rs:43-222)
    assert len(uniform) == 39 + 1 + 2 + 5 + 2 + 2 + 1 + 1 + 2 + 8 + 4 + 1 + 1 + 1 == 70
    assert len(cross) == 2 and padded == 128 and J.NUM_INPUTS == 78
    assert sum(J.IS_PUBLIC) == 9 + 12 + 27
    for n in (1, 2, 16, 256):
        cols = J.synthetic_columns(3 + n, n)
        assert J.check_satisfied(uniform, cross, cols, n) is None
    # a trace that breaks a constraint is caught: flip a lookup output on an assert row
    n = 64
    cols = J.synthetic_columns(9, n)
    t = next(t for t in range(n) if cols[J.IDX["Op_Assert"]][t] == 1)
    cols[J.IDX["LookupOutput"]][t] = 0
    assert J.check_satisfied(uniform, cross, cols, n) is not None
    # every instruction kind of the generator occurs, every circuit flag is set somewhere, padding closes the trace
    cols = J.synthetic_columns(11, 1024)
    assert all(any(cols[J.IDX["Op_" + f]]) for f in J.CIRCUIT_FLAGS)
    assert sum(1 for i in J.INSTRUCTIONS if any(cols[J.IDX["I_" + i]])) >= 20
    assert cols[J.IDX["Bytecode_ELFAddress"]][-1] == 0 and cols[J.IDX["NextPC"]][-1] == J.PC_START_ADDRESS + 4


def test_eq_plus_one_is_the_shift():
    """sum_t eq(r, t) f(t + 1) over t < n - 1 == sum_y eq_plus_one(r, y) f(y): what makes z_shift the next step's witness"""
    rng = O.SplitMix64(77)
    for l in (1, 3, 6):
        n = 1 << l
        r = [rng.field() for _ in range(l)]
        f = [rng.field() for _ in range(n)]
        eq, eqp1 = S.eq_plus_one_evals(r)
        assert eq == O.eq_evals(r)
        assert sum(eq[t] * f[t + 1] for t in range(n - 1)) % R == sum(a * b for a, b in zip(eqp1, f)) % R
        y = [rng.field() for _ in range(l)]
        ey = O.eq_evals(y)
        assert S.eq_plus_one_point(r, y) == sum(a * b for a, b in zip(eqp1, ey)) % R


def test_matrix_mle_partial_identity():
    """Az(rx) + rlc Bz(rx) + rlc^2 Cz(rx) == sum_y ABC(rx_constr, y) z(y || rx_step) with z's constant column = 1 in the
    non-shifted half only (worker.rs:107-170) -- for the Jolt set and the toy system, cross-step constraints included"""
    rng = O.SplitMix64(5)
    for system in ("jolt", "toy"):
        if system == "jolt":
            uniform, cross, padded = J.build_system()
            n = 32
            cols = J.synthetic_columns(4, n)
        else:
            uniform, cross, padded = S.synthetic_system()
            n = 16
            cols = S.synthetic_columns(4, n)
        nv = len(cols)
        V = 1
        while V < nv:
            V <<= 1
        az, bz, cz = S.dense_azbzcz(uniform, cross, cols, padded, n)
        sb, cb = n.bit_length() - 1, padded.bit_length() - 1
        rx = [rng.field() for _ in range(sb + cb)]
        rlc = rng.field()
        eq = O.eq_evals(rx)
        lhs = sum(e * (a + rlc * b + rlc * rlc * c) for e, a, b, c in zip(eq, az, bz, cz)) % R
        rx_step, rx_constr = rx[:sb], rx[sb:]
        es, ep = S.eq_plus_one_evals(rx_step)
        z = [0] * (4 * V)
        for i, col in enumerate(cols):
            z[i] = sum(a * b for a, b in zip(col, es)) % R
            z[2 * V + i] = sum(a * b for a, b in zip(col, ep)) % R
        z[V] = 1
        abc = S.matrix_mle_partial(uniform, cross, padded, V, rx_constr, rlc)
        assert sum(a * b for a, b in zip(abc, z)) % R == lhs


def test_whole_spartan_pipeline_plain_equals_rep3_and_verifies():
    for cfg in (dict(log_steps=0, seed=3), dict(log_steps=2, seed=5), dict(log_steps=4, seed=8), dict(log_steps=3, seed=2, system="toy")):
        a = S.run_full(dict(cfg, mode="plain"))
        b = S.run_full(dict(cfg, mode="rep3"))
        assert a["verified"] and b["verified"]
        assert a["proof_bytes"] == b["proof_bytes"]
    # a tampered proof is rejected
    uniform, cross, padded = J.build_system()
    n = 4
    cols = J.synthetic_columns(5, n)
    polys = S.jolt_party_columns(5, cols, 1)
    proof = S.prove_full(uniform, cross, padded, polys, n, O.Transcript(b"cozk-spartan"))
    assert S.verify_full(proof, uniform, cross, padded, len(cols), n, O.Transcript(b"cozk-spartan"))
    bad = dict(proof, shift_claim=(proof["shift_claim"] + 1) % R)
    assert not S.verify_full(bad, uniform, cross, padded, len(cols), n, O.Transcript(b"cozk-spartan"))
    bad = dict(proof, witness_evals=[(proof["witness_evals"][0] + 1) % R] + proof["witness_evals"][1:])
    assert not S.verify_full(bad, uniform, cross, padded, len(cols), n, O.Transcript(b"cozk-spartan"))
