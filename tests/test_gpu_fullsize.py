"""Size-independent properties at BASELINE.json's full single-GPU size (config 2: 2^20 cycles) -- the oracle cannot
finish these sizes in seconds, so the checks are the domain's own identities:
  * MSM linearity over 2^20 points: MSM(a) + MSM(b) == MSM(a + b) for uniform Fr scalars, and for u16 / 0-1 flag
    columns against their Fr embeddings (all scalar kinds must agree on the same values);
  * a sum of MSM results over a batch == one MSM of the coefficient-wise sum (a checksum of checksums);
  * the whole hot path at the bench workload: the built-in verifier accepts (every sumcheck round, the final GKR
    claim against a direct evaluation of the leaves, the opening reduction, the PST13 opening with the trapdoor),
    the proof is deterministic across steps."""
import pytest

import pyref as O

pytestmark = pytest.mark.gpu
N = 1 << 20


@pytest.fixture(scope="module")
def srs(cozk, ctx):
    b = cozk.Bases.from_scalars(ctx, cozk.Vec.random(ctx, N, seed=4242))
    yield b
    b.free()


def test_msm_linearity_2p20(cozk, ctx, srs):
    a = cozk.Vec.random(ctx, N, seed=1)
    b = cozk.Vec.random(ctx, N, seed=2)
    s = a.binop(cozk.OP_ADD, b)
    pa, pb, ps = srs.batch_msm([a, b, s])
    assert ps == O.g1_add(pa, pb)
    assert ps is not None and O.g1_is_on_curve(ps)


def _to_montgomery(cozk, ctx, plain_vec):
    """the vector's limbs hold plain integers x (which read as Montgomery values mean x / R); a Montgomery product
    with the constant vector whose limbs are R * R mod r gives x * R, i.e. x in Montgomery form -- on the device"""
    import numpy as np
    rr = cozk.fr_to_mont_limbs([pow(2, 256, cozk.FR_MOD)])  # limbs of R * R mod r
    return plain_vec.binop(cozk.OP_MUL, cozk.Vec.from_numpy(ctx, np.repeat(rr, len(plain_vec), axis=0)))


def test_small_kinds_equal_their_fr_embedding_2p20(cozk, ctx, srs):
    import numpy as np
    for kind, bits in ((cozk.SCALAR_U16, 0), (cozk.SCALAR_U8, 1), (cozk.SCALAR_U32, 0)):
        v = cozk.Vec.random(ctx, N, seed=77 + kind, kind=kind, max_bits=bits)
        limbs = np.zeros((N, 4), dtype=np.uint64)
        limbs[:, 0] = v.to_numpy().astype(np.uint64)
        emb = _to_montgomery(cozk, ctx, cozk.Vec.from_numpy(ctx, limbs))
        got_small, got_fr = srs.batch_msm([v, emb])
        assert got_small == got_fr and got_small is not None


def test_batch_checksum_of_checksums_2p20(cozk, ctx, srs):
    vecs = [cozk.Vec.random(ctx, N, seed=100 + i) for i in range(5)]
    total = vecs[0]
    for v in vecs[1:]:
        total = total.binop(cozk.OP_ADD, v)
    outs = srs.batch_msm(vecs + [total])
    acc = None
    for p in outs[:-1]:
        acc = O.g1_add(acc, p)
    assert acc == outs[-1]


def test_bench_workload_verifies_and_is_deterministic(cozk):
    h = cozk.Harness(mode="plain", log_n=20, n_fr=64, n_u16=32, n_u32=16, n_flags=16, n_small=0, gp_batch=8, gp_log_leaves=21, seed=2026)
    r1 = h.prove(verify=True)
    assert r1.verified == 1, h.last_error()
    r2 = h.prove(verify=False)
    assert bytes(r1.proof_digest) == bytes(r2.proof_digest)
    h.close()
