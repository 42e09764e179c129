"""GPU parity: field arithmetic and the MSM seam against the exact big-int oracle (oracle/pyref.py).
Bar: bit-exact (integer work).  Calls go through the C ABI (ctypes)."""
import numpy as np
import pytest

import pyref as O

pytestmark = pytest.mark.gpu

EDGE = [0, 1, 2, O.R - 1, O.R - 2, O.R_MONT_ONE, (1 << 253) + 12345, (1 << 253) - 1, O.TWO_INV]


def _rand_fr(rng, n, mod=O.R):
    return [rng.field(mod) for _ in range(n)]


@pytest.mark.parametrize("base_field", [False, True])
def test_field_binops_bit_exact(cozk, ctx, base_field):
    mod = O.P if base_field else O.R
    rng = O.SplitMix64(11 + base_field)
    edge = [e % mod for e in EDGE] + [mod - 1, mod - 2]
    a = edge + _rand_fr(rng, 2000, mod)
    b = list(reversed(edge)) + _rand_fr(rng, 2000, mod)
    # values are uploaded as Montgomery limbs of the right field
    A = cozk.Vec.from_numpy(ctx, cozk.fr_to_mont_limbs(a, mod))
    B = cozk.Vec.from_numpy(ctx, cozk.fr_to_mont_limbs(b, mod))
    for op, f in ((cozk.OP_ADD, lambda x, y: (x + y) % mod), (cozk.OP_SUB, lambda x, y: (x - y) % mod),
                  (cozk.OP_MUL, lambda x, y: x * y % mod)):
        got = cozk.mont_limbs_to_int(A.binop(op, B, base_field).to_numpy(), mod)
        assert got == [f(x, y) for x, y in zip(a, b)]


def test_fill_random_matches_oracle_stream(cozk, ctx):
    v = cozk.Vec.random(ctx, 300, seed=99).to_ints()
    assert v == O.synthetic_fr(99, 300)
    v8 = cozk.Vec.random(ctx, 100, seed=5, kind=cozk.SCALAR_U8).to_ints()
    assert v8 == O.synthetic_small(5, 100, 8)
    flags = cozk.Vec.random(ctx, 100, seed=6, kind=cozk.SCALAR_U8, max_bits=1).to_ints()
    assert flags == O.synthetic_small(6, 100, 1) and set(flags) <= {0, 1}


def _bases(rng, n):
    return [O.g1_mul(O.G1_GEN, rng.field()) for _ in range(n)]


@pytest.mark.parametrize("precompute", [True, False])
@pytest.mark.parametrize("n", [1, 2, 3, 17, 256, 1024])
def test_msm_matches_oracle(cozk, ctx, n, precompute):
    rng = O.SplitMix64(1000 + n)
    pts = _bases(rng, n)
    sc = _rand_fr(rng, n)
    # edge scalars from SURVEY 8c: zero, one, r-1, small
    for i, e in enumerate([0, 1, O.R - 1, 7, 65535, 65536, 32768, 32769][:n]):
        sc[i] = e
    B = cozk.Bases.upload(ctx, pts, precompute=precompute)
    got = B.msm(cozk.Vec.from_ints(ctx, sc))
    assert got == O.msm_naive(pts, sc)
    # host-scalar form of the seam
    assert B.msm_host(cozk.fr_to_mont_limbs(sc)) == got


def test_msm_infinity_repeats_and_cancellation(cozk, ctx):
    rng = O.SplitMix64(77)
    g = O.g1_mul(O.G1_GEN, 5)
    pts = [g, g, O.g1_neg(g), None, g, O.g1_mul(O.G1_GEN, 9)] + [g] * 200
    sc = [3, 3, 6, 12345, O.R - 1, 0] + [1] * 200
    B = cozk.Bases.upload(ctx, pts, precompute=True)
    assert B.msm(cozk.Vec.from_ints(ctx, sc)) == O.msm_naive(pts, sc)
    # everything cancels -> infinity
    pts2 = [g, O.g1_neg(g)]
    assert cozk.Bases.upload(ctx, pts2, precompute=False).msm(cozk.Vec.from_ints(ctx, [5, 5])) is None
    # all-zero scalars -> infinity
    assert B.msm(cozk.Vec.from_ints(ctx, [0] * len(pts))) is None


@pytest.mark.parametrize("kind,bits", [("U32", 32), ("U32", 20), ("U64", 64), ("U64", 33)])
def test_vec_narrow_keeps_values_and_the_commitment(cozk, ctx, kind, bits):
    """cozk_vec_narrow (msm_field_elements' dispatch on the scalars' bit length, pst13.rs:286-294): an FR vector of small values as a
    U32 / U64 vector -- same values, same MSM result; a value that does not fit is refused"""
    n = 600
    rng = O.SplitMix64(900 + bits)
    vals = [rng.field() & ((1 << bits) - 1) for _ in range(n)]
    vals[0], vals[1], vals[2] = 0, (1 << bits) - 1, 1
    v = cozk.Vec.from_ints(ctx, vals)
    k = getattr(cozk, "SCALAR_" + kind)
    nv = v.narrow(k)
    assert nv.to_ints() == vals
    pts = _bases(rng, n)
    B = cozk.Bases.upload(ctx, pts, precompute=True)
    assert B.msm(nv) == B.msm(v) == O.msm_naive(pts, vals)
    width = 32 if kind == "U32" else 64
    bad = list(vals)
    bad[n // 2] = 1 << width
    with pytest.raises(cozk.CozkError):
        cozk.Vec.from_ints(ctx, bad).narrow(k)
    bad[n // 2] = O.R - 1
    with pytest.raises(cozk.CozkError):
        cozk.Vec.from_ints(ctx, bad).narrow(k)


@pytest.mark.parametrize("kind,bits", [("U8", 8), ("U16", 16), ("U32", 32), ("U64", 64), ("U8", 1)])
def test_msm_small_scalar_kinds(cozk, ctx, kind, bits):
    n = 700
    rng = O.SplitMix64(4242)
    pts = _bases(rng, 64) * 11  # repeated points are legal SRS input
    pts = pts[:n]
    k = getattr(cozk, "SCALAR_" + kind)
    v = cozk.Vec.random(ctx, n, seed=31 + bits, kind=k, max_bits=bits if bits in (1,) else 0)
    sc = v.to_ints()
    B = cozk.Bases.upload(ctx, pts, precompute=True)
    assert B.msm(v) == O.msm_naive(pts, sc)


def test_msm_i64(cozk, ctx):
    rng = O.SplitMix64(5)
    pts = _bases(rng, 50)
    vals = [(-1) ** i * (rng.next() >> (1 + i % 40)) for i in range(50)]
    vals[0] = -(1 << 63)
    vals[1] = (1 << 63) - 1
    v = cozk.Vec.from_ints(ctx, vals, kind=cozk.SCALAR_I64)
    B = cozk.Bases.upload(ctx, pts, precompute=True)
    assert B.msm(v) == O.msm_naive(pts, [x % O.R for x in vals])


def test_batch_msm_mixed_kinds_and_slice(cozk, ctx):
    n = 512
    rng = O.SplitMix64(8)
    pts = _bases(rng, 40) * 14
    pts = pts[:n + 10]
    B = cozk.Bases.upload(ctx, pts, precompute=True)
    vecs = [cozk.Vec.random(ctx, n, seed=1), cozk.Vec.random(ctx, n, seed=2, kind=cozk.SCALAR_U16),
            cozk.Vec.random(ctx, n, seed=3, kind=cozk.SCALAR_U8, max_bits=1), cozk.Vec.random(ctx, n, seed=4)]
    got = B.batch_msm(vecs, offset=10)
    for g, v in zip(got, vecs):
        assert g == O.msm_naive(pts[10:10 + n], v.to_ints())


def test_bases_from_scalars_and_pair_sums(cozk, ctx):
    s = O.synthetic_fr(3, 64)
    S = cozk.Vec.from_ints(ctx, s)
    B = cozk.Bases.from_scalars(ctx, S, precompute=False)
    pts = B.download()
    assert pts == [O.g1_mul(O.G1_GEN, x) for x in s]
    ps = B.pair_sums(precompute=False).download()
    assert ps == [O.g1_add(pts[2 * i], pts[2 * i + 1]) for i in range(32)]


def test_msm_linearity_large(cozk, ctx):
    """size-independent property at 2^16: MSM(a) + MSM(b) == MSM(a+b) (the identity that
    pst13.rs:498-546 `test_combine_commitments` checks), bases generated on device."""
    n = 1 << 16
    B = cozk.Bases.from_scalars(ctx, cozk.Vec.random(ctx, n, seed=123), precompute=True)
    a = cozk.Vec.random(ctx, n, seed=1)
    b = cozk.Vec.random(ctx, n, seed=2)
    c = a.binop(cozk.OP_ADD, b)
    pa, pb, pc = B.batch_msm([a, b, c])
    assert ctx.g1_sum([pa, pb]) == pc
    assert O.g1_is_on_curve(pc) and pc is not None


def test_msm_repeated_bases_take_the_exception_path(cozk, ctx):
    """every base is the same point, so inside a bucket every addition after the first is P + P or P - P: the
    9x29 gather kernel must hand ALL those segments to the saturated fix-up kernel.  Expected result by scalar
    arithmetic: (sum of scalars) * G."""
    rng = O.SplitMix64(31337)
    n = 1 << 12
    g = O.g1_mul(O.G1_GEN, rng.field())
    sc = _rand_fr(rng, n)
    sc[5] = (-sc[4]) % O.R      # exact cancellation inside buckets
    sc[7] = sc[6]               # identical digit patterns: doublings in every window
    B = cozk.Bases.upload(ctx, [g] * n, precompute=True)
    got = B.msm(cozk.Vec.from_ints(ctx, sc))
    assert got == O.g1_mul(g, sum(sc) % O.R)
    # small scalars on repeated bases: one hot bucket made of equal points only
    flags = [rng.next() & 1 for _ in range(n)]
    got = B.msm(cozk.Vec.from_ints(ctx, flags, kind=cozk.SCALAR_U8))
    assert got == O.g1_mul(g, sum(flags) % O.R)


def test_prf_stream_matches_oracle(cozk, ctx):
    """cozk_vec_fill_prf: element j = one ChaCha12 block keyed with the 32-byte key (csrc/prf.hip.hpp) == pyref.prf_fr;
    257 elements cover both halves of a block and the second-attempt path (6 % of the draws)"""
    key = O.harness_prf_key(11, 3)
    for ctr in (0, 12345, (1 << 33) + 7):
        assert cozk.Vec.prf(ctx, 257, key, counter=ctr).to_ints() == O.prf_fr_vec(key, ctr, 257)


def test_rep3_share_vec_matches_oracle_sharing(cozk, ctx):
    """witness scatter on the device: party p's (a, b) of rep3::share_field_element with t0 = PRF(key0, .),
    t1 = PRF(key1, .); the three a components open to the secret, and b is the previous party's a"""
    n = 257
    v = O.synthetic_fr(4040, n)
    V = cozk.Vec.from_ints(ctx, v)
    k0, k1 = O.harness_prf_key(51, 0), O.harness_prf_key(52, 0)
    exp = O.rep3_share_vec(v, k0, k1, counter=9)
    got = []
    for p in range(3):
        a, b = V.rep3_share(k0, k1, p, counter=9)
        got.append((a.to_ints(), b.to_ints()))
        assert got[p] == ([x[0] for x in exp[p]], [x[1] for x in exp[p]])
    assert [(x + y + z) % O.R for x, y, z in zip(got[0][0], got[1][0], got[2][0])] == v


def test_rep3_scatter_device_to_device(cozk, ctx):
    """cozk_rep3_scatter: the dealer's context generates a party's shares and they arrive as vectors of the PARTY's context
    (another context on the same GPU here; a peer copy over xGMI when the devices differ -- 2 GPUs needed for that leg):
    equal to cozk_rep3_share_vec / the oracle, usable by the party's own stream right away"""
    import torch
    n = 1000
    v = O.synthetic_fr(808, n)
    V = cozk.Vec.from_ints(ctx, v)
    k0, k1 = O.harness_prf_key(61, 0), O.harness_prf_key(62, 0)
    exp = O.rep3_share_vec(v, k0, k1, counter=4)
    devs = [0] + ([1] if torch.cuda.device_count() > 1 else [])
    for d in devs:
        party_ctx = cozk.Context(d)
        for p in range(3):
            a, b = V.rep3_scatter(k0, k1, p, party_ctx, counter=4)
            assert a.ctx is party_ctx
            assert (a.to_ints(), b.to_ints()) == ([x[0] for x in exp[p]], [x[1] for x in exp[p]])
            assert a.binop(cozk.OP_ADD, b).to_ints() == [(x[0] + x[1]) % O.R for x in exp[p]]  # the party computes on them
            a.free()
            b.free()
        party_ctx.close()


def test_rep3_scatter_peer_copy_two_gpus(cozk, ctx):
    """the hipMemcpyPeer leg of cozk_rep3_scatter (dealer on GPU 0, party on GPU 1): skipped on this pool's 1-GPU boxes"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    n = 1 << 16
    V = cozk.Vec.random(ctx, n, seed=31)
    k0, k1 = O.harness_prf_key(71, 0), O.harness_prf_key(72, 0)
    party_ctx = cozk.Context(1)
    for p in range(3):
        a, b = V.rep3_scatter(k0, k1, p, party_ctx, counter=7)
        ea, eb = V.rep3_share(k0, k1, p, counter=7)
        assert a.ctx is party_ctx and a.to_ints() == ea.to_ints() and b.to_ints() == eb.to_ints()
        for v in (a, b, ea, eb):
            v.free()
    party_ctx.close()


def test_rep3_scatter_orders_against_the_party_stream(cozk, ctx):
    """the scatter's outputs are blocks of the PARTY's allocator but are written from the dealer's stream.  A block
    the party freed a moment ago may still be read by a kernel in flight on the party's stream: the scatter must not
    overwrite it early.  Queue reads of X on the party's stream, free X, scatter a vector of the same size straight after
    (the allocator hands X's block back), and check what the queued reads produced."""
    n = 1 << 20
    party_ctx = cozk.Context(0)
    X = cozk.Vec.random(party_ctx, n, seed=4242)
    acc = X.binop(cozk.OP_ADD, X)
    for _ in range(8):  # a queue of kernels that all read X
        nxt = acc.binop(cozk.OP_ADD, X)
        acc.free()
        acc = nxt
    X.free()
    V = cozk.Vec.random(ctx, n, seed=77)
    k0, k1 = O.harness_prf_key(63, 0), O.harness_prf_key(64, 0)
    a, b = V.rep3_scatter(k0, k1, 1, party_ctx, counter=0)
    X2 = cozk.Vec.random(party_ctx, n, seed=4242)
    want = X2.to_ints()
    got = acc.to_ints()
    assert got[:64] == [10 * x % O.R for x in want[:64]]
    assert got[-64:] == [10 * x % O.R for x in want[-64:]]
    ea, eb = V.rep3_share(k0, k1, 1, counter=0)
    assert a.to_ints()[:32] == ea.to_ints()[:32] and b.to_ints()[-32:] == eb.to_ints()[-32:]
    for v in (a, b, X2, acc, ea, eb, V):
        v.free()
    party_ctx.close()
