"""CPU tests (no GPU): the oracle itself.
  * pins of the exact big-int oracle: the constants the reference holds (TWO_INV, R mod r, rank map),
    EIP-196 G1 vectors, MSM naive == Pippenger, Rep3 identities, GKR / PST13 verification identities
    (what pst13.rs:498-546 `test_combine_commitments` checks);
  * the independent plain-C restatement against the big-int oracle and the committed golden vectors;
  * whole-pipeline proof bytes: C == Python for plain and Rep3, and plain == Rep3."""
import json
import os

import numpy as np
import pytest

import coracle
import pyharness
import pyref as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))


def _limbs(vals, mod):
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = O.to_mont(v, mod)
        out[i] = [(m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    return out


def _ints(arr, mod):
    return [O.from_mont(O.from_limbs64(r), mod) for r in arr]


def test_reference_constants():
    # snarks-core/src/field.rs:5-7
    assert O.TWO_INV == 0x183227397098d014dc2822db40c0ac2e9419f4243cdcb848a1f0fac9f8000001 == (O.R + 1) // 2
    # co-noir-spartan/noir-r1cs/noir_proof_scheme.json:7 -- interner bytes start with R mod r (LE)
    assert (O.MONT % O.R).to_bytes(32, "little").hex().startswith("fbffff4f1c3496ac")
    # mpc-net/src/rep3/mod.rs:29-32,52-85: global_worker_id = worker * 3 + party
    assert [w * 3 + p for w in range(2) for p in range(3)] == [0, 1, 2, 3, 4, 5]
    # EIP-196 vectors
    assert [hex(v) for v in O.g1_add(O.G1_GEN, O.G1_GEN)] == GOLD["g1"]["two_g"]
    assert O.g1_mul(O.G1_GEN, 3) == (3353031288059533942658390886683067124040920775575537747144343083137631628272,
                                     19321533766552368860946552437480515441416830039777911637913418824951667761761)
    assert O.g1_mul(O.G1_GEN, O.R) is None and O.g1_mul(O.G1_GEN, O.R - 1) == O.g1_neg(O.G1_GEN)


def test_chacha_block_rfc8439_vector_and_prf():
    """the keyed PRF behind every share / mask (csrc/prf.hip.hpp) is the ChaCha block function; pin the restatement
    with the RFC 8439 section 2.3.2 block test vector (20 rounds: key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:
    00:00:00:00), then check the 12-round PRF: C restatement == Python restatement, range, determinism, zero-sum masks"""
    key = bytes(range(32))
    st = O._CHACHA_CONST + [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)] + [1, 0x09000000, 0x4A000000, 0]
    assert O.chacha_block_words(st, 20) == [
        0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
        0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    k1, k2 = O.harness_prf_key(7, 0), O.harness_prf_key(7, 1)
    assert len(k1) == 32 and k1 != k2
    v = O.prf_fr_vec(k1, 1000, 200)
    assert all(0 <= x < O.R for x in v) and len(set(v)) == 200
    assert v[5:10] == O.prf_fr_vec(k1, 1005, 5) and v != O.prf_fr_vec(k2, 1000, 200)
    assert _ints(coracle.prf_fr(k1, 1000, 200), O.R) == v
    # a counter beyond 32 bits reaches the high counter word
    assert _ints(coracle.prf_fr(k2, (1 << 40) + 3, 4), O.R) == O.prf_fr_vec(k2, (1 << 40) + 3, 4)
    # rep3_share_vec: the a components open to the secret, b is the previous party's a
    sec = O.synthetic_fr(9, 17)
    sh = O.rep3_share_vec(sec, k1, k2, counter=3)
    assert [(sh[0][i][0] + sh[1][i][0] + sh[2][i][0]) % O.R for i in range(17)] == sec
    assert all(sh[p][i][1] == sh[(p + 2) % 3][i][0] for p in range(3) for i in range(17))


def test_rep3_identities():
    rng = O.SplitMix64(3)
    x, y = rng.field(), rng.field()
    sx, sy = O.rep3_share(x, rng), O.rep3_share(y, rng)
    assert O.rep3_open(sx) == x
    # local products sum to the product (ops.rs:71-78); into_additive sums to the secret (types.rs:76-81)
    assert sum(O.rep3_local_mul(sx[p], sy[p]) for p in range(3)) % O.R == x * y % O.R
    assert sum(O.rep3_into_additive(sx[p]) for p in range(3)) % O.R == x
    c = rng.field()
    assert sum(O.rep3_into_additive(O.rep3_promote_from_trivial(c, p)) for p in range(3)) % O.R == c
    assert sum(O.additive_promote_from_trivial(c, p) for p in range(3)) % O.R == c


def test_msm_pippenger_equals_naive_and_golden():
    g = GOLD["msm"]
    pts = [(int(p[0], 16), int(p[1], 16)) for p in g["points"]]
    sc = [int(s, 16) for s in g["scalars"]]
    exp = tuple(int(v, 16) for v in g["result"])
    assert O.msm_naive(pts, sc) == exp == O.msm_pippenger(pts, sc) == O.msm_pippenger(pts, sc, c=7)
    # C oracle
    xy = np.concatenate([_limbs([p[0] for p in pts], O.P), _limbs([p[1] for p in pts], O.P)], axis=1)
    out, inf = coracle.msm(xy, np.zeros(len(pts), dtype=np.uint8), _limbs(sc, O.R))
    assert inf == 0 and (_ints([out[:4]], O.P)[0], _ints([out[4:]], O.P)[0]) == exp
    # with an infinity base and a cancelling pair
    pts2 = pts[:3] + [pts[0]]
    inf2 = np.array([0, 1, 0, 0], dtype=np.uint8)
    sc2 = [5, 7, 9, O.R - 5]
    xy2 = np.concatenate([_limbs([p[0] for p in pts2], O.P), _limbs([p[1] for p in pts2], O.P)], axis=1)
    out, inf = coracle.msm(xy2, inf2, _limbs(sc2, O.R))
    assert (_ints([out[:4]], O.P)[0], _ints([out[4:]], O.P)[0]) == O.g1_mul(pts[2], 9)


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_c_field_ops_against_golden(field):
    mod = O.R if field == "fr" else O.P
    rows = GOLD[field]
    a = [int(r["a"], 16) for r in rows]
    b = [int(r["b"], 16) for r in rows]
    A, B = _limbs(a, mod), _limbs(b, mod)
    for op, key in ((0, "add"), (1, "sub"), (2, "mul")):
        got = _ints(coracle.fp_binop(1 if field == "fq" else 0, op, A, B), mod)
        assert got == [int(r[key], 16) for r in rows]


def test_c_fast_fq_product_and_xyzz_msm():
    """round 3 CPU baseline: the unrolled no-carry Fq product == the textbook CIOS == the golden products (edge operands 0, 1,
    p - 1, R, >= 2^253 included); the XYZZ Pippenger (now orc_msm's path) on repeated bases / infinity / cancelling points"""
    rows = GOLD["fq"]
    a = [int(r["a"], 16) for r in rows] + [0, 1, O.P - 1, O.P - 1, (1 << 253) + 7]
    b = [int(r["b"], 16) for r in rows] + [5, O.P - 1, O.P - 1, 1, O.P - 2]
    A, B = _limbs(a, O.P), _limbs(b, O.P)
    got = _ints(coracle.fq_mul_fast(A, B), O.P)
    assert got == [x * y % O.P for x, y in zip(a, b)]
    assert got == _ints(coracle.fp_binop(1, 2, A, B), O.P)
    rng = O.SplitMix64(99)
    g = O.G1_GEN
    base = [O.g1_mul(g, rng.field()) for _ in range(5)]
    pts = base + [base[0], O.g1_neg(base[1]), base[2], base[2]] + [(0, 0)]
    inf = np.array([0] * 9 + [1], dtype=np.uint8)
    sc = [rng.field() for _ in range(5)] + [3, 3, O.R - 1, 1, 12345]
    sc[6] = sc[1]  # s * P + s * (-P) cancels inside one bucket
    xy = np.concatenate([_limbs([p[0] for p in pts], O.P), _limbs([p[1] for p in pts], O.P)], axis=1)
    out, oinf = coracle.msm(xy, inf, _limbs(sc, O.R))
    want = O.msm_naive([None if i == 9 else p for i, p in enumerate(pts)], sc)
    assert oinf == 0 and (_ints([out[:4]], O.P)[0], _ints([out[4:]], O.P)[0]) == want


def test_gkr_and_pst_identities_small():
    rng = O.SplitMix64(9)
    leaves = [rng.field() for _ in range(32)]
    L1 = O.gp_construct([leaves], 2, None)
    sh = [O.rep3_share(v, rng) for v in leaves]
    L3 = O.gp_construct([[s[p] for s in sh] for p in range(3)], 2, O.SplitMix64(1))
    p1, r1 = O.gp_prove(L1, O.Transcript())
    p3, r3 = O.gp_prove(L3, O.Transcript())
    assert p1 == p3 and r1 == r3
    claim, r = O.gp_verify(p1, 2, O.Transcript())
    assert sum(e * x for e, x in zip(O.eq_evals(r), leaves)) % O.R == claim
    # PST13: commit(RLC) == RLC(commit), open + check (pst13.rs:498-546)
    ck = O.pst_setup(3, rng)
    polys = [[rng.field() for _ in range(8)] for _ in range(3)]
    rho = rng.field()
    pw = [1, rho, rho * rho % O.R]
    agg = [sum(c * p[i] for c, p in zip(pw, polys)) % O.R for i in range(8)]
    comb = None
    for c, p in zip(pw, polys):
        comb = O.g1_add(comb, O.g1_mul(O.pst_commit(ck, p), c))
    assert comb == O.pst_commit(ck, agg)
    pt = [rng.field() for _ in range(3)]
    prf, val = O.pst_open(ck, agg, list(reversed(pt)))
    assert val == sum(e * v for e, v in zip(O.eq_evals(pt), agg)) % O.R  # big-endian evaluate == reversed-point PST value
    assert O.pst_check_with_trapdoor(ck, comb, list(reversed(pt)), val, prf)


def test_pipeline_c_equals_python_equals_golden():
    for row in GOLD["pipelines"]:
        cfg = row["cfg"]
        res, proof = coracle.pipeline(cfg)
        assert bytes(res.digest).hex() == row["digest"] and len(proof) == row["proof_len"]
    # plain == rep3 for the same witness
    d = {r["cfg"]["mode"] + str(r["cfg"]["seed"]): r["digest"] for r in GOLD["pipelines"]}
    assert d["plain42"] == d["rep342"] and d["plain7"] == d["rep37"]
    # the Python pipeline regenerates the first fixture (slow path, one config)
    cfg = GOLD["pipelines"][0]["cfg"]
    assert pyharness.run(cfg)["digest"] == GOLD["pipelines"][0]["digest"]


def test_spartan_pipeline_oracle_self_consistent():
    """oracle/pyspartan.py: the synthetic R1CS is satisfied, both sumchecks, the matrix evaluation and the PST13
    opening verify; the digest is deterministic in the seed"""
    import pyspartan
    a = pyspartan.run(dict(log_n=4, seed=11))
    b = pyspartan.run(dict(log_n=4, seed=11))
    c = pyspartan.run(dict(log_n=4, seed=12))
    assert a["verified"] and c["verified"]
    assert a["digest"] == b["digest"] != c["digest"]
