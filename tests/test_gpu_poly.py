"""GPU parity of the polynomial seam (bind / evals / RLC / GKR layer ops) against oracle/pyref.py.
Bit-exact bar; both share modes (Rep3 shares and plain values); ragged tails as the reference
handles them (dense_interleaved_poly.rs:160-177,232-247)."""
import pytest

import pyref as O

pytestmark = pytest.mark.gpu


def _shares(rng, n, mode):
    if mode == "rep3":
        return [(rng.field(), rng.field()) for _ in range(n)]
    return [rng.field() for _ in range(n)]


@pytest.mark.parametrize("mode", ["rep3", "plain"])
@pytest.mark.parametrize("n", [2, 4, 8, 1024])
def test_dense_bind_both_orders_first_and_rebind(cozk, ctx, mode, n):
    rng = O.SplitMix64(n + (mode == "plain"))
    for order in (cozk.LOW_TO_HIGH, cozk.HIGH_TO_LOW):
        coeffs = _shares(rng, n, mode)
        p = cozk.Rep3DensePolynomial.new(ctx, coeffs)
        ref = coeffs
        while len(ref) > 1:
            r = rng.field()
            p.bind(r, order)
            ref = O.dense_bind(ref, r, order)
            assert p.coeffs() == ref and len(p) == len(ref)
        assert p.final_sumcheck_claim() == ref[0]
    # mixed orders on one polynomial (first bind L2H, then H2L in place, then L2H again)
    coeffs = _shares(rng, 16, mode)
    p = cozk.Rep3DensePolynomial.new(ctx, coeffs)
    ref = coeffs
    for order in (cozk.LOW_TO_HIGH, cozk.HIGH_TO_LOW, cozk.LOW_TO_HIGH):
        r = rng.field()
        p.bind(r, order)
        ref = O.dense_bind(ref, r, order)
        assert p.coeffs() == ref


@pytest.mark.parametrize("mode", ["rep3", "plain"])
def test_batch_evaluate_dot_and_lincomb(cozk, ctx, mode):
    rng = O.SplitMix64(5)
    nv = 9
    n = 1 << nv
    polys_ref = [_shares(rng, n, mode) for _ in range(3)] + [_shares(rng, n // 4, mode)]
    polys = [cozk.Rep3DensePolynomial.new(ctx, c) for c in polys_ref]
    r = [rng.field() for _ in range(nv)]
    evals, eq = cozk.Rep3DensePolynomial.batch_evaluate(polys[:3], r)
    ref_evals, ref_eq = O.dense_batch_evaluate(polys_ref[:3], r)
    assert eq.to_ints() == ref_eq
    assert evals == ref_evals
    # 0/1 chi short-circuit path of evaluate_at_chi_optimized gives the same value
    chi01 = [rng.next() & 1 for _ in range(n)]
    got = polys[0].evaluate_at_chi(cozk.Vec.from_ints(ctx, chi01))
    assert got == O.dense_evaluate_at_chi(polys_ref[0], chi01)
    pub = [rng.field() for _ in range(n)]
    assert polys[1].dot_product_with_public(cozk.Vec.from_ints(ctx, pub)) == O.dense_dot_product_with_public(polys_ref[1], pub)
    cf = [rng.field() for _ in range(4)]
    lc = cozk.Rep3DensePolynomial.linear_combination(polys, cf)
    assert lc.coeffs() == O.dense_linear_combination(polys_ref, cf)


def test_lincomb_public_into_shared_by_party(cozk, ctx):
    """public polynomial inside a shared RLC enters via add_public: P0 -> a, P1 -> b, P2 -> none
    (multilinear_polynomial.rs:221-231; shared_or_public.rs:150-152)"""
    rng = O.SplitMix64(9)
    n = 64
    sh = [(rng.field(), rng.field()) for _ in range(n)]
    pub = [rng.field() for _ in range(n)]
    cf = [rng.field(), rng.field()]
    ps = cozk.Rep3DensePolynomial.new(ctx, sh)
    pp = cozk.Rep3DensePolynomial.new(ctx, pub)
    for party in range(3):
        got = cozk.Rep3DensePolynomial.linear_combination([ps, pp], cf, party_id=party).coeffs()
        exp = []
        for s, v in zip(sh, pub):
            a, b = s[0] * cf[0] % O.R, s[1] * cf[0] % O.R
            if party == 0:
                a = (a + v * cf[1]) % O.R
            if party == 1:
                b = (b + v * cf[1]) % O.R
            exp.append((a, b))
        assert got == exp


def test_chunk_view_and_share_a_commit(cozk, ctx):
    rng = O.SplitMix64(4)
    coeffs = _shares(rng, 64, "rep3")
    p = cozk.Rep3DensePolynomial.new(ctx, coeffs)
    c = p.chunk(16, 16)
    assert c.coeffs() == coeffs[16:32]
    assert p.copy_share_a().to_ints() == [x[0] for x in coeffs]
    r = rng.field()
    c.bind(r, cozk.LOW_TO_HIGH)
    assert c.coeffs() == O.dense_bind(coeffs[16:32], r, O.LOW_TO_HIGH)
    assert p.coeffs() == coeffs  # parent untouched


def test_eq_evals_and_spliteq(cozk, ctx):
    rng = O.SplitMix64(2)
    for nv in (0, 1, 2, 5, 8):
        r = [rng.field() for _ in range(nv)]
        assert cozk.eq_evals(ctx, r).to_ints() == O.eq_evals(r)


@pytest.mark.parametrize("mode", ["rep3", "plain"])
@pytest.mark.parametrize("length", [2, 4, 6, 8, 12, 64, 96, 1024, 2048 + 8])
def test_layer_bind_and_cubic_all_rounds(cozk, ctx, mode, length):
    """runs every sumcheck round of one layer: compute_cubic (nested Dao-Thaler case while E1 is
    unbound, then the linear-time case), bind, eq bind -- comparing each message with the oracle."""
    rng = O.SplitMix64(length * 3 + (mode == "plain"))
    coeffs = _shares(rng, length, mode)
    nodes = (length + 1) // 2
    nv = max(0, (nodes - 1).bit_length())
    w = [rng.field() for _ in range(nv)]
    layer = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    eq = cozk.SplitEqPolynomial(ctx, w)
    ref = list(coeffs)
    ref_eq = O.SplitEq(w)
    claim = rng.field()
    for _ in range(nv):
        assert eq.lens() == (ref_eq.E1_len, ref_eq.E2_len)
        got = layer.compute_cubic(eq, claim)
        exp = O.interleaved_compute_cubic(ref, ref_eq, claim)
        assert got == exp
        r = rng.field()
        layer.bind(r)
        eq.bind(r)
        ref = O.interleaved_bind(ref, r)
        ref_eq.bind(r)
        assert layer.coeffs() == ref
        claim = rng.field()
    if len(ref) == 2:
        fc = layer.final_claims()
        assert fc == (ref[0], ref[1])


@pytest.mark.parametrize("mode", ["rep3", "plain"])
@pytest.mark.parametrize("length", [4, 6, 96, 2048 + 8, 1 << 15, (1 << 15) + 12])
def test_layer_round_fused_equals_separate_calls(cozk, ctx, mode, length):
    """cozk_layer_round (bind + eq bind + compute_cubic per call; single-launch kernel for layers <= 8192 elements,
    so the two larger sizes cross from the separate-kernel path into it) gives the same round messages and the
    same bound layer as the separate calls; the small sizes are also checked against the oracle"""
    rng = O.SplitMix64(length * 5 + (mode == "plain"))
    coeffs = _shares(rng, length, mode)
    nodes = (length + 1) // 2
    nv = max(0, (nodes - 1).bit_length())
    w = [rng.field() for _ in range(nv)]
    fused = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    sep = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    eq_f, eq_s = cozk.SplitEqPolynomial(ctx, w), cozk.SplitEqPolynomial(ctx, w)
    small = length <= 4096
    ref, ref_eq = list(coeffs), O.SplitEq(w)
    r = None
    for _ in range(nv):
        claim = rng.field()
        if r is not None:
            sep.bind(r)
            eq_s.bind(r)
            if small:
                ref = O.interleaved_bind(ref, r)
                ref_eq.bind(r)
        got = fused.round(eq_f, r, claim)
        assert got == sep.compute_cubic(eq_s, claim)
        assert eq_f.lens() == eq_s.lens()
        if small:
            assert got == O.interleaved_compute_cubic(ref, ref_eq, claim)
        r = rng.field()
    if r is not None:
        fused.bind(r)
        sep.bind(r)
    assert fused.coeffs() == sep.coeffs()


@pytest.mark.parametrize("mode", ["rep3", "plain"])
@pytest.mark.parametrize("length", [4, 6, 96, 2048, 4096 + 8, 1 << 14])
def test_layer_prove_rounds_matches_separate_calls(cozk, ctx, mode, length):
    """cozk_layer_prove_rounds (whole round loop behind the ABI; per-round launches above 2048 elements, then the
    resident mailbox kernel down to the final claims) sends the same round polynomials and ends in the same final
    claims as compute_cubic / bind / final_claims called one by one"""
    rng = O.SplitMix64(length * 7 + (mode == "plain"))
    coeffs = _shares(rng, length, mode)
    nodes = (length + 1) // 2
    nv = max(0, (nodes - 1).bit_length())
    w = [rng.field() for _ in range(nv)]
    claim0 = rng.field()
    rs = [rng.field() for _ in range(nv)]
    claims = [rng.field() for _ in range(nv)]
    sep = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    eq_s = cozk.SplitEqPolynomial(ctx, w)
    expected, c = [], claim0
    for j in range(nv):
        expected.append(sep.compute_cubic(eq_s, c))
        sep.bind(rs[j])
        eq_s.bind(rs[j])
        c = claims[j]
    one = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    eq_o = cozk.SplitEqPolynomial(ctx, w)
    seen = []

    def exchange(rnd, cf):
        seen.append(cf)
        return rs[rnd], claims[rnd]

    got_r, (left, right) = one.prove_rounds(eq_o, claim0, nv, exchange)
    assert seen == expected
    assert got_r == rs
    if nv:
        assert (left, right) == sep.final_claims()
        assert one.coeffs() == sep.coeffs()


@pytest.mark.parametrize("mode", ["rep3", "plain"])
def test_layer_output_local_masks_and_claimed_outputs(cozk, ctx, mode):
    rng = O.SplitMix64(31)
    coeffs = _shares(rng, 64, mode)
    layer = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    plain = layer.layer_output_local().to_ints()
    assert plain == O.interleaved_layer_output_local(coeffs)
    assert layer.claimed_outputs() == O.interleaved_layer_output_local(coeffs)
    # masked: mask_j = PRF(seed_self, ctr + j) - PRF(seed_prev, ctr + j)
    got = layer.layer_output_local(masked=True, seed_self=11, seed_prev=22, counter=5).to_ints()
    ms = O.synthetic_fr(11, 40)[5:5 + 32]
    mp = O.synthetic_fr(22, 40)[5:5 + 32]
    assert got == [(v + a - b) % O.R for v, a, b in zip(plain, ms, mp)]


def test_rep3_layer_output_reconstructs_product(cozk, ctx):
    """three parties on one GPU: local product + zero-sum masks, ring reshare by device copies;
    reconstructed next layer == products of the reconstructed inputs (mul_vec semantics)."""
    rng = O.SplitMix64(77)
    n = 128
    vals = [rng.field() for _ in range(n)]
    sh = [O.rep3_share(v, rng) for v in vals]
    seeds = [101, 202, 303]  # seed_i is shared between party i and party i+1
    ca = []
    for p in range(3):
        layer = cozk.Rep3DenseInterleavedPolynomial.new(ctx, [s[p] for s in sh])
        ca.append(layer.layer_output_local(masked=True, seed_self=seeds[p], seed_prev=seeds[(p + 2) % 3], counter=0).to_ints())
    recon = [(ca[0][j] + ca[1][j] + ca[2][j]) % O.R for j in range(n // 2)]
    assert recon == [vals[2 * j] * vals[2 * j + 1] % O.R for j in range(n // 2)]


def test_open_quadratic_and_pst_fold(cozk, ctx):
    rng = O.SplitMix64(13)
    n = 256
    polys_ref = [[(rng.field(), rng.field()) for _ in range(n)], [(rng.field(), rng.field()) for _ in range(n // 2)]]
    eqs_ref = [[rng.field() for _ in range(n)], [rng.field() for _ in range(n // 2)]]
    polys = [cozk.Rep3DensePolynomial.new(ctx, c) for c in polys_ref]
    eqs = [cozk.Rep3DensePolynomial.new(ctx, c) for c in eqs_ref]
    got = cozk.open_quadratic_evals(polys, eqs)
    for (g0, g2), p, e in zip(got, polys_ref, eqs_ref):
        h = len(p) // 2
        e0 = sum(O.rep3_into_additive(O.rep3_mul_public(p[i], e[i])) for i in range(h)) % O.R
        e2 = sum(O.rep3_into_additive(O.rep3_mul_public(O.rep3_sub(O.rep3_add(p[i + h], p[i + h]), p[i]),
                                                        (2 * e[i + h] - e[i]) % O.R)) for i in range(h)) % O.R
        assert (g0, g2) == (e0, e2)
    # PST fold
    r = [rng.field() for _ in range(64)]
    pt = rng.field()
    q = cozk.Vec.alloc(ctx, 32)
    rn = cozk.Vec.alloc(ctx, 32)
    cozk.pst_fold(ctx, cozk.Vec.from_ints(ctx, r), pt, q, rn)
    assert q.to_ints() == [(r[2 * b + 1] - r[2 * b]) % O.R for b in range(32)]
    assert rn.to_ints() == [(r[2 * b] * (1 - pt) + r[2 * b + 1] * pt) % O.R for b in range(32)]
