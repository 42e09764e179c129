"""GPU parity of co-jolt's Spartan outer sumcheck over Az / Bz / Cz (co-jolt/src/poly/spartan_interleaved_poly.rs,
co-jolt/src/r1cs/spartan/worker.rs:63-100,277-300) through the C ABI:
  * k_r1cs_rows: the dense Az / Bz / Cz the device builds from the witness columns open to the values of the oracle's sparse
    (index, SharedOrPublic) list (every case of ::new: shared x shared, public x shared, public x public, zero / empty LCs,
    cross-step constraints incl. the last step's constant-only evaluation), and equal the clear Az, Bz, Cz;
  * every round: the sum over the parties of the cubic coefficients == the sum of the oracle's (sparse walk, Gruen split-eq,
    from_linear_times_quadratic_with_hint), the final claims too;
  * whole proofs bit-identical to oracle/pyspartan_outer.py (plain and 3-party Rep3) at small sizes; at 2^14 / 2^18 steps
    the built-in verifier (rounds, final check, claims == the multilinear extensions of the clear Az, Bz, Cz) and
    Rep3 == plain."""
import hashlib
import importlib

import pytest

import pyref as O
import pyspartan_outer as S

pytestmark = pytest.mark.gpu
R = O.R


def _open(vals, nparties):
    if nparties == 1:
        return [v % R for v in vals[0]]
    return [(a[0] + b[0] + c[0]) % R for a, b, c in zip(*vals)]


@pytest.mark.parametrize("log_steps,nparties", [(0, 1), (1, 3), (3, 3), (3, 1), (5, 3)])
def test_rows_and_every_round_match_the_sparse_oracle(cozk, ctx, log_steps, nparties):
    OU = importlib.import_module("co-zkvms_amd.outer")
    n = 1 << log_steps
    seed = 40 + log_steps
    uniform, cross, padded = S.synthetic_system()
    cols = S.synthetic_columns(seed, n)
    polys = S.party_columns(seed, cols, nparties)
    rng = O.SplitMix64(17)
    nv = log_steps + 3
    tau = [rng.field() for _ in range(nv)]
    mode = "plain" if nparties == 1 else "rep3"
    devs = []
    for p in range(nparties):
        dp = [cozk.Rep3DensePolynomial.new(ctx, col) for _kind, col in polys[p]]
        devs.append(OU.SpartanOuter(ctx, mode, p, uniform, cross, dp, padded, tau))
    # rows: dense shares open to the clear Az, Bz, Cz (= what the oracle's sparse list holds, zeros elsewhere)
    az, bz, cz = S.dense_azbzcz(uniform, cross, cols, padded, n)
    got = [d.download() for d in devs]
    for q, clear in enumerate((az, bz, cz)):
        assert _open([g[q] for g in got], nparties) == clear
    sparse = [S.build_sparse(uniform, cross, polys[p], padded, n, p) for p in range(nparties)]
    for p in range(nparties):
        for idx, val in sparse[p]:
            assert (az, bz, cz)[idx % 3][idx // 3] == sum(S.sp_into_additive(
                [s for s in sparse[q] if s[0] == idx][0][1], q) for q in range(nparties)) % R
        break  # the index pattern is the same for every party
    # rounds
    eqs = [S.GruenSplitEq(tau) for _ in range(nparties)]
    claims = [0] * nparties
    r_prev = None
    for rnd in range(nv):
        ref_msgs = []
        for p in range(nparties):
            t0, tinf = S.quadratic_evals(sparse[p], eqs[p], p, rnd == 0)
            eq = eqs[p]
            sw = eq.current_scalar * eq.w[eq.current_index - 1] % R
            ref_msgs.append(S.cubic_from_linear_times_quadratic_with_hint((eq.current_scalar - sw) % R, (2 * sw - eq.current_scalar) % R, t0, tinf, claims[p]))
        dev_msgs = [devs[p].round(r_prev, claims[p]) for p in range(nparties)]
        ref_poly, dev_poly = O.combine_additive(ref_msgs), O.combine_additive(dev_msgs)
        assert dev_poly == ref_poly, rnd
        if nparties == 1:
            assert dev_msgs == ref_msgs
        r_prev = rng.field()
        nxt = O.unipoly_eval(ref_poly, r_prev)
        for p in range(nparties):
            claims[p] = O.additive_promote_from_trivial(nxt, p)
            eqs[p].bind(r_prev)
            sparse[p] = S.bind_sparse(sparse[p], r_prev, p)
    fin_dev = O.combine_additive([devs[p].final_evals(r_prev) for p in range(nparties)])
    # final_sumcheck_evals reads the list AFTER the last bind
    assert fin_dev == O.combine_additive([S.final_evals(sparse[p], p) for p in range(nparties)])
    for d in devs:
        d.free()


@pytest.mark.parametrize("mode", ["plain", "rep3"])
@pytest.mark.parametrize("log_steps,seed", [(0, 1), (2, 5), (4, 9), (6, 11)])
def test_small_proofs_bit_identical_to_the_oracle(cozk, mode, log_steps, seed):
    OU = importlib.import_module("co-zkvms_amd.outer")
    h = OU.OuterHarness(mode=mode, log_steps=log_steps, seed=seed)
    res = h.prove(verify=True)
    assert res.verified == 1, h.last_error()
    ref = S.run(dict(mode=mode, log_steps=log_steps, seed=seed))
    assert ref["verified"]
    got = h.proof_bytes(res)
    assert hashlib.sha256(got).hexdigest() == bytes(res.proof_digest).hex()
    assert got == ref["proof_bytes"]
    h.close()


@pytest.mark.parametrize("log_steps", [14, 18])
def test_large_verifies_and_rep3_equals_plain(cozk, log_steps):
    OU = importlib.import_module("co-zkvms_amd.outer")
    digs = {}
    for mode in ("plain", "rep3"):
        h = OU.OuterHarness(mode=mode, log_steps=log_steps, seed=2026)
        r = h.prove(verify=True)
        assert r.verified == 1, h.last_error()
        assert bytes(h.prove(verify=False).proof_digest) == bytes(r.proof_digest)
        digs[mode] = bytes(r.proof_digest)
        h.close()
    assert digs["plain"] == digs["rep3"]
