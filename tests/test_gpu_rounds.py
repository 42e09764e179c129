"""GPU parity of the remaining round kernels: the generic product round of prove_arbitrary_worker
(co-jolt/src/subprotocols/sumcheck.rs:168-246) and co-noir-spartan's two sumcheck rounds + zero_round
(co-spartan/src/sumcheck.rs:171-395, worker.rs:153-182), each run for ALL rounds of a sumcheck with the
oracle alongside (bit-exact)."""
import pytest

import pyref as O

pytestmark = pytest.mark.gpu


def _sh(rng, n, mode):
    return [(rng.field(), rng.field()) for _ in range(n)] if mode == "rep3" else [rng.field() for _ in range(n)]


@pytest.mark.parametrize("mode", ["rep3", "plain"])
@pytest.mark.parametrize("m,degree", [(2, 2), (3, 3), (1, 1), (4, 4)])
def test_prove_arbitrary_rounds(cozk, ctx, mode, m, degree):
    """product of m polynomials, one of them shared (the Spartan inner/shift and output-check shapes)"""
    rng = O.SplitMix64(17 * m + degree + (mode == "plain"))
    n = 256
    ref = [[rng.field() for _ in range(n)] for _ in range(m - 1)] + [_sh(rng, n, mode)]
    # put the shared factor in the middle when possible
    if m >= 3:
        ref[1], ref[-1] = ref[-1], ref[1]
    polys = [cozk.Rep3DensePolynomial.new(ctx, c) for c in ref]
    while len(ref[0]) > 1:
        got = cozk.prod_sumcheck_evals(polys, degree)
        assert got == O.prod_round_evals(ref, degree)
        r = rng.field()
        for p in polys:
            p.bind(r, cozk.HIGH_TO_LOW)
        ref = [O.dense_bind(c, r, O.HIGH_TO_LOW) for c in ref]
    assert [p.final_sumcheck_claim() for p in polys] == [c[0] for c in ref]


@pytest.mark.parametrize("mode", ["rep3", "plain"])
def test_spartan_first_and_second_sumcheck_all_rounds(cozk, ctx, mode):
    rng = O.SplitMix64(404 + (mode == "plain"))
    nv = 8
    n = 1 << nv
    za, zb, zc = _sh(rng, n, mode), _sh(rng, n, mode), _sh(rng, n, mode)
    eq = [rng.field() for _ in range(n)]
    P = [cozk.Rep3DensePolynomial.new(ctx, c) for c in (za, zb, zc, eq)]
    ref = [za, zb, zc, eq]
    for _ in range(nv):
        assert cozk.spartan_first_round(*P) == O.spartan_first_round_evals(*ref)
        r = rng.field()
        for p in P:
            p.bind(r, cozk.LOW_TO_HIGH)  # fix_variables
        ref = [O.dense_bind(c, r, O.LOW_TO_HIGH) for c in ref]
    # sumcheck #2
    z = _sh(rng, n, mode)
    pa, pb, pc = ([rng.field() for _ in range(n)] for _ in range(3))
    coef = [rng.field() for _ in range(3)]
    Z = cozk.Rep3DensePolynomial.new(ctx, z)
    A, B, C = (cozk.Rep3DensePolynomial.new(ctx, c) for c in (pa, pb, pc))
    for _ in range(nv):
        got = cozk.spartan_second_round(Z, A, B, C, coef)
        exp = O.spartan_second_round_evals(z, pa, pb, pc, coef)
        if mode == "plain":
            assert [g[0] for g in got] == exp
        else:
            assert got == exp
        r = rng.field()
        for p in (Z, A, B, C):
            p.bind(r, cozk.LOW_TO_HIGH)
        z, pa, pb, pc = (O.dense_bind(c, r, O.LOW_TO_HIGH) for c in (z, pa, pb, pc))


@pytest.mark.parametrize("mode", ["rep3", "plain"])
def test_spartan_zero_round_sparse_matvec(cozk, ctx, mode):
    rng = O.SplitMix64(55)
    nrows = ncols = 128
    z = _sh(rng, ncols, mode)
    rows = []
    for r in range(nrows):
        k = rng.next() % 4  # 0..3 non-zeros per row, including empty rows
        rows.append(sorted({rng.next() % ncols for _ in range(k)}))
    row_ptr, col, va, vb, vc = [0], [], [], [], []
    ea, eb, ec = [], [], []
    for r, cs in enumerate(rows):
        for c in cs:
            a, b, cc = rng.field(), rng.field() % 7, rng.field()
            col.append(c)
            va.append(a)
            vb.append(b)
            vc.append(cc)
            ea.append((r, c, a))
            eb.append((r, c, b))
            ec.append((r, c, cc))
        row_ptr.append(len(col))
    V = cozk.Vec
    za, zb, zc = cozk.sparse_matvec3(V.from_ints(ctx, row_ptr, cozk.SCALAR_U32), V.from_ints(ctx, col, cozk.SCALAR_U32),
                                     V.from_ints(ctx, va), V.from_ints(ctx, vb), V.from_ints(ctx, vc),
                                     cozk.Rep3DensePolynomial.new(ctx, z))
    assert za.coeffs() == O.sparse_matvec(ea, z, nrows)
    assert zb.coeffs() == O.sparse_matvec(eb, z, nrows)
    assert zc.coeffs() == O.sparse_matvec(ec, z, nrows)


@pytest.mark.parametrize("party", [None, 0, 1, 2])
def test_fingerprint_leaves_k11(cozk, ctx, party):
    """compute_leaves (K11): bytecode-shaped fingerprint gamma a + gamma^2 v0 + ... + shared term - tau over compact
    u8 / u16 / u32 / u64 columns, one shared and one public Fr polynomial; read leaves at offset 0, write leaves
    (+ gamma^7) right behind them in the same buffer, as the batch of two circuits is laid out"""
    rng = O.SplitMix64(900 + (party or 0) * 3 + (party is None))
    n = 300  # not a multiple of the block size
    mode = "plain" if party is None else "rep3"
    kinds = [cozk.SCALAR_U8, cozk.SCALAR_U16, cozk.SCALAR_U32, cozk.SCALAR_U64]
    bits = {cozk.SCALAR_U8: 8, cozk.SCALAR_U16: 16, cozk.SCALAR_U32: 32, cozk.SCALAR_U64: 64}
    cols_ref = [[rng.next() & ((1 << bits[k]) - 1) for _ in range(n)] for k in kinds]
    cols_ref[3][0] = (1 << 64) - 1
    cols = [cozk.Vec.from_ints(ctx, c, kind=k) for c, k in zip(cols_ref, kinds)]
    gamma, tau = rng.field(), rng.field()
    g = [pow(gamma, e, O.R) for e in range(1, 8)]
    shared = _sh(rng, n, mode)
    public = [rng.field() for _ in range(n)]
    polys_ref = [shared, public]
    polys = [cozk.Rep3DensePolynomial.new(ctx, shared), cozk.Rep3DensePolynomial.new(ctx, public)]
    out_a = cozk.Vec.alloc(ctx, 2 * n)
    out_b = cozk.Vec.alloc(ctx, 2 * n) if mode == "rep3" else None
    pid = party or 0
    cozk.fingerprint_leaves(ctx, cols, g[:4], polys, [g[4], g[5]], (-tau) % O.R, mode, pid, out_a, out_b, offset=0, n=n)
    cozk.fingerprint_leaves(ctx, cols, g[:4], polys, [g[4], g[5]], (g[6] - tau) % O.R, mode, pid, out_a, out_b, offset=n, n=n)
    read = O.fingerprint_leaves(cols_ref, g[:4], polys_ref, [g[4], g[5]], (-tau) % O.R, party)
    write = O.fingerprint_leaves(cols_ref, g[:4], polys_ref, [g[4], g[5]], (g[6] - tau) % O.R, party)
    a = out_a.to_ints()
    if mode == "rep3":
        b = out_b.to_ints()
        assert list(zip(a, b)) == read + write
    else:
        assert a == read + write


def test_fingerprint_leaves_parties_reconstruct_plain(cozk, ctx):
    """the three parties' Rep3 leaves open to the plain prover's leaves (public part added exactly once)"""
    rng = O.SplitMix64(911)
    n = 64
    col = [rng.next() & 0xffff for _ in range(n)]
    v = [rng.field() for _ in range(n)]
    t0, t1 = [rng.field() for _ in range(n)], [rng.field() for _ in range(n)]
    t2 = [(x - y - z) % O.R for x, y, z in zip(v, t0, t1)]
    shares = [list(zip(t0, t2)), list(zip(t1, t0)), list(zip(t2, t1))]
    gamma, tau = rng.field(), rng.field()
    plain = O.fingerprint_leaves([col], [gamma], [v], [gamma * gamma % O.R], (-tau) % O.R, None)
    a_sum = [0] * n
    for p in range(3):
        out_a, out_b = cozk.Vec.alloc(ctx, n), cozk.Vec.alloc(ctx, n)
        cozk.fingerprint_leaves(ctx, [cozk.Vec.from_ints(ctx, col, kind=cozk.SCALAR_U16)], [gamma], [cozk.Rep3DensePolynomial.new(ctx, shares[p])],
                                [gamma * gamma % O.R], (-tau) % O.R, "rep3", p, out_a, out_b)
        a_sum = [(x + y) % O.R for x, y in zip(a_sum, out_a.to_ints())]
    assert a_sum == plain
