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
@pytest.mark.parametrize("length", [2, 4, 6, 8, 12, 64, 96, 1024, 2048 + 8, 4096 + 24, 3 * 4096 + 40, 1 << 14])
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
@pytest.mark.parametrize("length", [4, 6, 96, 2048 + 8, 8192 + 72, 1 << 15, (1 << 15) + 12, 3 * (1 << 14) + 520])
def test_layer_round_fused_equals_separate_calls(cozk, ctx, mode, length):
    """cozk_layer_round (bind + eq bind + compute_cubic per call; single-launch kernel for layers <= 8192 elements,
    so the larger sizes cross from the separate-kernel path into it) gives the same round messages and the
    same bound layer as the separate calls, and the oracle's.  The larger sizes run the 9 x 29 kernels with the LDS-staged
    whole-line accesses; the ragged ones mix staged waves with the direct path of the last, partial wave."""
    rng = O.SplitMix64(length * 5 + (mode == "plain"))
    coeffs = _shares(rng, length, mode)
    nodes = (length + 1) // 2
    nv = max(0, (nodes - 1).bit_length())
    w = [rng.field() for _ in range(nv)]
    fused = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    sep = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    eq_f, eq_s = cozk.SplitEqPolynomial(ctx, w), cozk.SplitEqPolynomial(ctx, w)
    small = True  # every size against the oracle (Python: ~0.1 s per round at 2^15)
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


@pytest.mark.parametrize("resident", [True, False])
@pytest.mark.parametrize("mode", ["rep3", "plain"])
@pytest.mark.parametrize("length", [4, 6, 96, 2048, 4096 + 8, 1 << 14])
def test_layer_prove_rounds_matches_separate_calls(cozk, ctx, mode, length, resident):
    """cozk_layer_prove_rounds (whole round loop behind the ABI; per-round launches above 2048 elements, then -- with
    resident rounds on -- the resident mailbox kernel down to the final claims) sends the same round polynomials and
    ends in the same final claims as compute_cubic / bind / final_claims called one by one"""
    ctx.set_resident_rounds(resident)
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

    try:
        got_r, (left, right) = one.prove_rounds(eq_o, claim0, nv, exchange)
    finally:
        ctx.set_resident_rounds(None)
    assert seen == expected
    assert got_r == rs
    if nv:
        assert (left, right) == sep.final_claims()
        assert one.coeffs() == sep.coeffs()


def test_resident_watchdog_falls_back_to_per_round_launches(cozk, ctx, monkeypatch):
    """a round callback slower than the resident kernel's watchdog (COZK_RESIDENT_TIMEOUT_S) used to fail the proof
    ("the resident kernel gave up waiting for the host"); now the remaining rounds of the call run as one launch per
    round from the state the kernel left: same round polynomials, same final claims.  The stall is put in the middle of
    the resident tail and again in its last round (the kernel then leaves before its final bind)."""
    import time
    monkeypatch.setenv("COZK_RESIDENT_TIMEOUT_S", "1")
    rng = O.SplitMix64(4242)
    length = 512
    coeffs = _shares(rng, length, "rep3")
    nv = 8
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
    for stall_round in (3, nv - 1):
        one = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
        eq_o = cozk.SplitEqPolynomial(ctx, w)
        seen = []

        def exchange(rnd, cf):
            seen.append(cf)
            if rnd == stall_round:
                time.sleep(1.6)
            return rs[rnd], claims[rnd]

        ctx.set_resident_rounds(True)
        try:
            got_r, (left, right) = one.prove_rounds(eq_o, claim0, nv, exchange)
        finally:
            ctx.set_resident_rounds(None)
        assert seen == expected and got_r == rs
        assert (left, right) == sep.final_claims()
        assert one.coeffs() == sep.coeffs()


def test_two_interdependent_contexts_default_flags(cozk):
    """the failure of round 1's suite logs, designed out: two provers in ONE process on ONE GPU that need each other's
    round messages (each callback waits at a barrier until the other prover has sent its round polynomial), both
    contexts created with DEFAULT flags.  The automatic default keeps resident round kernels off as soon as a second
    context lives on the device, so neither prover blocks the other's launches; both finish with the same transcript."""
    import threading
    n_prov = 2
    ctxs = [cozk.Context(0) for _ in range(n_prov)]
    rng = O.SplitMix64(2718)
    length, nv = 1024, 9
    coeffs = _shares(rng, length, "plain")
    w = [rng.field() for _ in range(nv)]
    claim0 = rng.field()
    rs = [rng.field() for _ in range(nv)]
    claims = [rng.field() for _ in range(nv)]
    barrier = threading.Barrier(n_prov, timeout=60)
    seen = [[] for _ in range(n_prov)]
    results, errors = [None] * n_prov, []

    def run(i):
        try:
            layer = cozk.Rep3DenseInterleavedPolynomial.new(ctxs[i], coeffs)
            eq = cozk.SplitEqPolynomial(ctxs[i], w)

            def exchange(rnd, cf):
                seen[i].append(cf)
                barrier.wait()  # the "coordinator" answers only when every prover's message of this round is in
                return rs[rnd], claims[rnd]

            results[i] = layer.prove_rounds(eq, claim0, nv, exchange)
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append(e)
            barrier.abort()

    th = [threading.Thread(target=run, args=(i,)) for i in range(n_prov)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errors, errors
    assert all(not t.is_alive() for t in th)
    assert seen[0] == seen[1] and len(seen[0]) == nv
    assert results[0] == results[1] and results[0][0] == rs
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("mode", ["rep3", "plain"])
def test_layer_output_local_masks_and_claimed_outputs(cozk, ctx, mode):
    rng = O.SplitMix64(31)
    coeffs = _shares(rng, 64, mode)
    layer = cozk.Rep3DenseInterleavedPolynomial.new(ctx, coeffs)
    plain = layer.layer_output_local().to_ints()
    assert plain == O.interleaved_layer_output_local(coeffs)
    assert layer.claimed_outputs() == O.interleaved_layer_output_local(coeffs)
    # masked: mask_j = PRF(key_self, ctr + j) - PRF(key_prev, ctr + j)  (keyed ChaCha12, csrc/prf.hip.hpp)
    ks, kp = O.harness_prf_key(11, 0), O.harness_prf_key(22, 0)
    got = layer.layer_output_local(masked=True, key_self=ks, key_prev=kp, counter=5).to_ints()
    ms, mp = O.prf_fr_vec(ks, 5, 32), O.prf_fr_vec(kp, 5, 32)
    assert got == [(v + a - b) % O.R for v, a, b in zip(plain, ms, mp)]


def test_rep3_layer_output_reconstructs_product(cozk, ctx):
    """three parties on one GPU: local product + zero-sum masks, ring reshare by device copies;
    reconstructed next layer == products of the reconstructed inputs (mul_vec semantics)."""
    rng = O.SplitMix64(77)
    n = 128
    vals = [rng.field() for _ in range(n)]
    sh = [O.rep3_share(v, rng) for v in vals]
    seeds = [O.harness_prf_key(101, p) for p in range(3)]  # key_i is shared between party i and party i+1
    ca = []
    for p in range(3):
        layer = cozk.Rep3DenseInterleavedPolynomial.new(ctx, [s[p] for s in sh])
        ca.append(layer.layer_output_local(masked=True, key_self=seeds[p], key_prev=seeds[(p + 2) % 3], counter=0).to_ints())
    recon = [(ca[0][j] + ca[1][j] + ca[2][j]) % O.R for j in range(n // 2)]
    assert recon == [vals[2 * j] * vals[2 * j + 1] % O.R for j in range(n // 2)]


@pytest.mark.parametrize("n", [1, 2, 255, 1024])
def test_rep3_mul_vec_local_matches_oracle(cozk, ctx, n):
    """cozk_rep3_mul_vec_local (the generic rep3::arithmetic::mul_vec, local half) against the oracle: every party's
    c.a = a.a*b.a + a.a*b.b + a.b*b.a (mpc-types/src/protocols/rep3/arithmetic/ops.rs:71-78) + its zero-sharing
    mask; after the ring reshare (b = previous party's a) the result is exactly O.rep3_mul_vec, which opens to the
    element-wise product; the plain mode is the plain product"""
    rng = O.SplitMix64(900 + n)
    xs, ys = [rng.field() for _ in range(n)], [rng.field() for _ in range(n)]
    X, Y = [O.rep3_share(v, rng) for v in xs], [O.rep3_share(v, rng) for v in ys]
    keys = [O.harness_prf_key(77, p) for p in range(3)]
    ctr = 40
    prf = [O.prf_fr_vec(keys[p], ctr, n) for p in range(3)]
    masks = [[(prf[p][j] - prf[(p + 2) % 3][j]) % O.R for j in range(n)] for p in range(3)]
    exp = O.rep3_mul_vec([[s[p] for s in X] for p in range(3)], [[s[p] for s in Y] for p in range(3)], masks)
    ca = []
    for p in range(3):
        xa, xb = cozk.Vec.from_ints(ctx, [s[p][0] for s in X]), cozk.Vec.from_ints(ctx, [s[p][1] for s in X])
        ya, yb = cozk.Vec.from_ints(ctx, [s[p][0] for s in Y]), cozk.Vec.from_ints(ctx, [s[p][1] for s in Y])
        unmasked = cozk.rep3_mul_vec_local(ctx, xa, xb, ya, yb).to_ints()
        assert unmasked == [O.rep3_local_mul(X[j][p], Y[j][p]) for j in range(n)]
        ca.append(cozk.rep3_mul_vec_local(ctx, xa, xb, ya, yb, key_self=keys[p], key_prev=keys[(p + 2) % 3], counter=ctr).to_ints())
        assert ca[p] == [e[0] for e in exp[p]]
    got = [[(ca[p][j], ca[(p + 2) % 3][j]) for j in range(n)] for p in range(3)]  # the ring reshare
    assert got == exp
    assert [O.rep3_open([got[p][j] for p in range(3)]) for j in range(n)] == [x * y % O.R for x, y in zip(xs, ys)]
    pl = cozk.rep3_mul_vec_local(ctx, cozk.Vec.from_ints(ctx, xs), None, cozk.Vec.from_ints(ctx, ys), None).to_ints()
    assert pl == [x * y % O.R for x, y in zip(xs, ys)]


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
