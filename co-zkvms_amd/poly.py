"""Host-side mirror of the reference's polynomial interfaces over the C ABI:

    Rep3DensePolynomial              co-jolt/src/poly/dense_mlpoly.rs
    Rep3DenseInterleavedPolynomial   co-jolt/src/poly/dense_interleaved_poly.rs
    SplitEqPolynomial                jolt-core (used dense_interleaved_poly.rs:218-303)

`mode` = MODE_REP3 (shares {a, b}) or MODE_PLAIN (plain prover / public polynomial).
Values cross as canonical Python ints; shares as (a, b) tuples.
"""
import ctypes

import numpy as np

from . import _lib as L
from .engine import Vec, fr_to_mont_limbs, mont_limbs_to_int


def _fr(x):
    return fr_to_mont_limbs([x])[0]


def _ptr_array(objs):
    return (ctypes.c_void_p * len(objs))(*[o.h for o in objs])


class Rep3DensePolynomial:
    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    # ---- constructors
    @classmethod
    def from_vec_shares(cls, ctx, a, b=None):
        """`from_vec_shares(a, b)` (dense_mlpoly.rs:77-84); a, b: Vec (device) -- b None => plain"""
        h = ctypes.c_void_p()
        mode = L.MODE_REP3 if b is not None else L.MODE_PLAIN
        ctx.check(ctx._l.cozk_poly_create(ctx.h, mode, a.h, b.h if b is not None else None, ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def new(cls, ctx, coeffs):
        """`new(coeffs)`: list of (a, b) shares, or list of ints for a plain polynomial"""
        if coeffs and isinstance(coeffs[0], tuple):
            return cls.from_vec_shares(ctx, Vec.from_ints(ctx, [c[0] for c in coeffs]),
                                       Vec.from_ints(ctx, [c[1] for c in coeffs]))
        return cls.from_vec_shares(ctx, Vec.from_ints(ctx, coeffs))

    @classmethod
    def random(cls, ctx, n, seed, mode=L.MODE_REP3):
        a = Vec.random(ctx, n, seed)
        b = Vec.random(ctx, n, seed + 0x1000003) if mode == L.MODE_REP3 else None
        return cls.from_vec_shares(ctx, a, b)

    def chunk(self, offset, length):
        """split_poly chunk view (dense_mlpoly.rs:275-301)"""
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx._l.cozk_poly_chunk(self.ctx.h, self.h, offset, length, ctypes.byref(h)))
        return Rep3DensePolynomial(self.ctx, h)

    # ---- accessors
    def __len__(self):
        return self.ctx._l.cozk_poly_len(self.h)

    def len(self):
        return len(self)

    @property
    def mode(self):
        return self.ctx._l.cozk_poly_mode(self.h)

    def get_num_vars(self):
        return len(self).bit_length() - 1

    def coeffs(self):
        n = len(self)
        a = np.empty((n, 4), dtype=np.uint64)
        b = np.empty((n, 4), dtype=np.uint64)
        self.ctx.check(self.ctx._l.cozk_poly_download(self.ctx.h, self.h, a.ctypes.data, b.ctypes.data))
        if self.mode == L.MODE_REP3:
            return list(zip(mont_limbs_to_int(a), mont_limbs_to_int(b)))
        return mont_limbs_to_int(a)

    def copy_share_a(self):
        """zero-copy device view of the `a` components (dense_mlpoly.rs:103-110)"""
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx._l.cozk_poly_share_view(self.ctx.h, self.h, 0, ctypes.byref(h)))
        v = Vec(self.ctx, h, L.SCALAR_FR)
        v._keepalive = self
        return v

    def get_bound_coeff(self, index):
        a = np.zeros(4, dtype=np.uint64)
        b = np.zeros(4, dtype=np.uint64)
        self.ctx.check(self.ctx._l.cozk_poly_get_coeff(self.ctx.h, self.h, index, a.ctypes.data, b.ctypes.data))
        if self.mode == L.MODE_REP3:
            return (mont_limbs_to_int(a)[0], mont_limbs_to_int(b)[0])
        return mont_limbs_to_int(a)[0]

    def final_sumcheck_claim(self):
        assert len(self) == 1
        return self.get_bound_coeff(0)

    # ---- PolynomialBinding
    def bind(self, r, order):
        rr = _fr(r)
        self.ctx.check(self.ctx._l.cozk_poly_bind(self.ctx.h, self.h, rr.ctypes.data, order))

    bind_parallel = bind

    # ---- evaluation
    @staticmethod
    def batch_evaluate(polys, r):
        """(dense_mlpoly.rs:183-192) -> (additive evals, eq Vec)"""
        ctx = polys[0].ctx
        eq = eq_evals(ctx, r)
        return Rep3DensePolynomial.batch_evaluate_at_chi(polys, eq), eq

    @staticmethod
    def batch_evaluate_at_chi(polys, chi):
        ctx = polys[0].ctx
        k = len(polys)
        out = np.zeros((k, 4), dtype=np.uint64)
        ctx.check(ctx._l.cozk_poly_batch_evaluate_at_chi(ctx.h, _ptr_array(polys), k, chi.h, out.ctypes.data))
        return mont_limbs_to_int(out)

    def evaluate_at_chi(self, chi):
        return Rep3DensePolynomial.batch_evaluate_at_chi([self], chi)[0]

    def evaluate(self, r):
        return Rep3DensePolynomial.batch_evaluate([self], r)[0][0]

    def dot_product_with_public(self, other):
        a = np.zeros(4, dtype=np.uint64)
        b = np.zeros(4, dtype=np.uint64)
        self.ctx.check(self.ctx._l.cozk_poly_dot_product_with_public(self.ctx.h, self.h, other.h, a.ctypes.data,
                                                                     b.ctypes.data))
        if self.mode == L.MODE_REP3:
            return (mont_limbs_to_int(a)[0], mont_limbs_to_int(b)[0])
        return mont_limbs_to_int(a)[0]

    @staticmethod
    def linear_combination(polys, coeffs, out_mode=None, party_id=0):
        ctx = polys[0].ctx
        if out_mode is None:
            out_mode = L.MODE_REP3 if any(p.mode == L.MODE_REP3 for p in polys) else L.MODE_PLAIN
        cf = fr_to_mont_limbs(coeffs)
        h = ctypes.c_void_p()
        ctx.check(ctx._l.cozk_poly_linear_combination(ctx.h, _ptr_array(polys), cf.ctypes.data, len(polys), out_mode,
                                                      party_id, ctypes.byref(h)))
        return Rep3DensePolynomial(ctx, h)

    def free(self):
        if self.h:
            self.ctx._l.cozk_poly_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def eq_evals(ctx, r):
    """EqPolynomial::evals(r) as a device Vec"""
    rr = fr_to_mont_limbs(r) if len(r) else np.zeros((0, 4), dtype=np.uint64)
    h = ctypes.c_void_p()
    ctx.check(ctx._l.cozk_eq_evals(ctx.h, rr.ctypes.data if len(r) else None, len(r), ctypes.byref(h)))
    return Vec(ctx, h, L.SCALAR_FR)


def eq_plus_one_evals(ctx, r):
    """EqPlusOnePolynomial::evals(r, None).1 (cozk_eq_plus_one_evals; co-jolt/src/r1cs/spartan/worker.rs:116) as a device Vec"""
    rr = fr_to_mont_limbs(r) if len(r) else np.zeros((0, 4), dtype=np.uint64)
    h = ctypes.c_void_p()
    ctx.check(ctx._l.cozk_eq_plus_one_evals(ctx.h, rr.ctypes.data if len(r) else None, len(r), ctypes.byref(h)))
    return Vec(ctx, h, L.SCALAR_FR)


def batch_dot_public(polys, pubs):
    """bind_z / bind_shift_z (worker.rs:139-152): dot_product_with_public of every polynomial with 1 or 2 public Vecs in one
    pass -> out[p][q] = (a, b) share (b = 0 for a plain / public polynomial)"""
    ctx = polys[0].ctx
    k, nq = len(polys), len(pubs)
    out = np.zeros((k * nq * 2, 4), dtype=np.uint64)
    ctx.check(ctx._l.cozk_poly_batch_dot_public(ctx.h, _ptr_array(polys), k, _ptr_array(pubs), nq, out.ctypes.data))
    v = mont_limbs_to_int(out)
    return [[(v[(p * nq + q) * 2], v[(p * nq + q) * 2 + 1]) for q in range(nq)] for p in range(k)]


def open_quadratic_evals(polys, eqs):
    """inner sums of compute_quadratic (opening_proof.rs:374-414) -> [(eval_0, eval_2)] additive"""
    ctx = polys[0].ctx
    k = len(polys)
    out = np.zeros((2 * k, 4), dtype=np.uint64)
    ctx.check(ctx._l.cozk_open_quadratic_evals(ctx.h, _ptr_array(polys), _ptr_array(eqs), k, out.ctypes.data))
    v = mont_limbs_to_int(out)
    return [(v[2 * i], v[2 * i + 1]) for i in range(k)]


def fingerprint_leaves(ctx, cols, col_coeffs, polys, poly_coeffs, constant, mode, party_id, out_a, out_b=None, offset=0, n=None):
    """compute_leaves (K11): leaf = sum c_k col_k + sum d_j poly_j + constant into out_a/out_b[offset ..]"""
    if n is None:
        n = min([len(c) for c in cols] + [len(p) for p in polys])
    cc = fr_to_mont_limbs(list(col_coeffs)) if cols else np.zeros((1, 4), dtype=np.uint64)
    pc = fr_to_mont_limbs(list(poly_coeffs)) if polys else np.zeros((1, 4), dtype=np.uint64)
    ca = (ctypes.c_void_p * max(1, len(cols)))(*[c.h for c in cols])
    ctx.check(ctx._l.cozk_fingerprint_leaves(ctx.h, ca, cc.ctypes.data, len(cols), _ptr_array(polys) if polys else None, pc.ctypes.data, len(polys),
                                             _fr(constant).ctypes.data, L.MODE_REP3 if mode == "rep3" else L.MODE_PLAIN, party_id, out_a.h,
                                             out_b.h if out_b is not None else None, offset, n))


def pst_fold(ctx, r_vec, p, q_vec, r_next):
    pp = _fr(p)
    ctx.check(ctx._l.cozk_pst_fold(ctx.h, r_vec.h, pp.ctypes.data, q_vec.h, r_next.h))


class SplitEqPolynomial:
    def __init__(self, ctx, w):
        self.ctx = ctx
        ww = fr_to_mont_limbs(w) if len(w) else np.zeros((0, 4), dtype=np.uint64)
        h = ctypes.c_void_p()
        ctx.check(ctx._l.cozk_spliteq_new(ctx.h, ww.ctypes.data if len(w) else None, len(w), ctypes.byref(h)))
        self.h = h
        self.num_vars = len(w)

    def get_num_vars(self):
        return self.num_vars

    def lens(self):
        a = ctypes.c_size_t()
        b = ctypes.c_size_t()
        self.ctx._l.cozk_spliteq_lens(self.h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def bind(self, r):
        rr = _fr(r)
        self.ctx.check(self.ctx._l.cozk_spliteq_bind(self.ctx.h, self.h, rr.ctypes.data))

    def free(self):
        if self.h:
            self.ctx._l.cozk_spliteq_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Rep3DenseInterleavedPolynomial:
    def __init__(self, ctx, handle, mode):
        self.ctx = ctx
        self.h = handle
        self.mode = mode

    @classmethod
    def from_vecs(cls, ctx, a, b=None, take_ownership=False):
        mode = L.MODE_REP3 if b is not None else L.MODE_PLAIN
        h = ctypes.c_void_p()
        ctx.check(ctx._l.cozk_layer_create(ctx.h, mode, a.h, b.h if b is not None else None,
                                           1 if take_ownership else 0, ctypes.byref(h)))
        return cls(ctx, h, mode)

    @classmethod
    def new(cls, ctx, coeffs):
        if coeffs and isinstance(coeffs[0], tuple):
            return cls.from_vecs(ctx, Vec.from_ints(ctx, [c[0] for c in coeffs]), Vec.from_ints(ctx, [c[1] for c in coeffs]))
        return cls.from_vecs(ctx, Vec.from_ints(ctx, coeffs))

    def clone(self):
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx._l.cozk_layer_clone(self.ctx.h, self.h, ctypes.byref(h)))
        return Rep3DenseInterleavedPolynomial(self.ctx, h, self.mode)

    def __len__(self):
        return self.ctx._l.cozk_layer_len(self.h)

    def len(self):
        return len(self)

    def coeffs(self):
        n = len(self)
        a = np.empty((n, 4), dtype=np.uint64)
        b = np.empty((n, 4), dtype=np.uint64)
        self.ctx.check(self.ctx._l.cozk_layer_download(self.ctx.h, self.h, a.ctypes.data, b.ctypes.data))
        if self.mode == L.MODE_REP3:
            return list(zip(mont_limbs_to_int(a), mont_limbs_to_int(b)))
        return mont_limbs_to_int(a)

    def bind(self, r, party_id=None):
        rr = _fr(r)
        self.ctx.check(self.ctx._l.cozk_layer_bind(self.ctx.h, self.h, rr.ctypes.data))

    def compute_cubic(self, eq_poly, previous_round_claim, party_id=None):
        """-> 4 additive coefficient shares (UniPoly<AdditiveShare>.coeffs)"""
        pc = _fr(previous_round_claim)
        out = np.zeros((4, 4), dtype=np.uint64)
        self.ctx.check(self.ctx._l.cozk_layer_compute_cubic(self.ctx.h, self.h, eq_poly.h, pc.ctypes.data, out.ctypes.data))
        return mont_limbs_to_int(out)

    def round(self, eq_poly, r, previous_round_claim):
        """one prove_sumcheck round in one call: bind layer + eq with the previous challenge r (None in round 0),
        then compute_cubic -> 4 additive coefficient shares"""
        pc = _fr(previous_round_claim)
        rr = _fr(r) if r is not None else None
        out = np.zeros((4, 4), dtype=np.uint64)
        self.ctx.check(self.ctx._l.cozk_layer_round(self.ctx.h, self.h, eq_poly.h, rr.ctypes.data if rr is not None else None,
                                                    pc.ctypes.data, out.ctypes.data))
        return mont_limbs_to_int(out)

    def prove_rounds(self, eq_poly, claim, num_rounds, exchange):
        """the whole prove_sumcheck round loop (cozk_layer_prove_rounds): `exchange(round, coeffs) -> (r, next_claim)` is
        the host's star exchange.  Returns (challenges, (left, right) final claims)."""
        cb_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                ctypes.POINTER(ctypes.c_uint64))
        err = []

        def _cb(_u, rnd, coeffs, r_out, nc_out):
            try:
                cf = mont_limbs_to_int(np.ctypeslib.as_array(coeffs, shape=(4, 4)).copy())
                r, nc = exchange(rnd, cf)
                rr, cc = _fr(r), _fr(nc)
                for i in range(4):
                    r_out[i] = int(rr.reshape(-1)[i])
                    nc_out[i] = int(cc.reshape(-1)[i])
                return 0
            except Exception as e:  # never unwind into C
                err.append(e)
                return 1

        cb = cb_t(_cb)
        pc = _fr(claim)
        rs = np.zeros((max(1, num_rounds), 4), dtype=np.uint64)
        fc = np.zeros((4, 4), dtype=np.uint64)
        rc = self.ctx._l.cozk_layer_prove_rounds(self.ctx.h, self.h, eq_poly.h, pc.ctypes.data, num_rounds, cb, None, rs.ctypes.data, fc.ctypes.data)
        if err:
            raise err[0]
        self.ctx.check(rc)
        f = mont_limbs_to_int(fc)
        left, right = ((f[0], f[1]), (f[2], f[3])) if self.mode == L.MODE_REP3 else (f[0], f[2])
        return mont_limbs_to_int(rs)[:num_rounds], (left, right)

    def final_claims(self, party_id=None):
        out = np.zeros((4, 4), dtype=np.uint64)
        self.ctx.check(self.ctx._l.cozk_layer_final_claims(self.ctx.h, self.h, out.ctypes.data))
        v = mont_limbs_to_int(out)
        if self.mode == L.MODE_REP3:
            return (v[0], v[1]), (v[2], v[3])
        return v[0], v[2]

    def layer_output_local(self, masked=False, key_self=None, key_prev=None, counter=0):
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx._l.cozk_layer_output_local(self.ctx.h, self.h, 1 if masked else 0, L.prf_key(key_self), L.prf_key(key_prev),
                                                           counter, ctypes.byref(h)))
        return Vec(self.ctx, h, L.SCALAR_FR)

    def claimed_outputs(self):
        n = len(self) // 2
        out = np.zeros((n, 4), dtype=np.uint64)
        self.ctx.check(self.ctx._l.cozk_layer_claimed_outputs(self.ctx.h, self.h, out.ctypes.data))
        return mont_limbs_to_int(out)

    def free(self):
        if self.h:
            self.ctx._l.cozk_layer_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def rep3_mul_vec_local(ctx, xa, xb, ya, yb, key_self=None, key_prev=None, counter=0):
    """rep3::arithmetic::mul_vec, local half (cozk_rep3_mul_vec_local): out[j] = x[j] x y[j] as an additive share,
    plus the zero-sharing mask PRF(key_self, counter + j) - PRF(key_prev, counter + j) when keys are given.
    xb = yb = None: plain vectors."""
    mode = L.MODE_PLAIN if xb is None else L.MODE_REP3
    masked = key_self is not None
    h = ctypes.c_void_p()
    ctx.check(ctx._l.cozk_rep3_mul_vec_local(ctx.h, mode, xa.h, xb.h if xb is not None else None, ya.h, yb.h if yb is not None else None,
                                             1 if masked else 0, L.prf_key(key_self), L.prf_key(key_prev), counter, ctypes.byref(h)))
    return Vec(ctx, h, L.SCALAR_FR)


def prod_sumcheck_evals(polys, degree):
    """one round of prove_arbitrary_worker's evaluation loop for a product of polynomials
    (co-jolt/src/subprotocols/sumcheck.rs:189-215): additive evaluations at x = 0, 2, .., degree"""
    ctx = polys[0].ctx
    out = np.zeros((degree, 4), dtype=np.uint64)
    ctx.check(ctx._l.cozk_prod_sumcheck_evals(ctx.h, _ptr_array(polys), len(polys), degree, out.ctypes.data))
    return mont_limbs_to_int(out)


def spartan_first_round(za, zb, zc, pub):
    """co-spartan first_sumcheck_prove_round evaluations at X = 0..3 (unmasked additive)"""
    ctx = za.ctx
    out = np.zeros((4, 4), dtype=np.uint64)
    ctx.check(ctx._l.cozk_spartan_first_round(ctx.h, za.h, zb.h, zc.h, pub.h, out.ctypes.data))
    return mont_limbs_to_int(out)


def spartan_second_round(z, a, b, c, coef):
    """co-spartan second_sumcheck_prove_round Rep3 evaluations at X = 0..2 (unmasked) -> [(a, b)] * 3"""
    ctx = z.ctx
    cf = fr_to_mont_limbs(coef)
    oa = np.zeros((3, 4), dtype=np.uint64)
    ob = np.zeros((3, 4), dtype=np.uint64)
    ctx.check(ctx._l.cozk_spartan_second_round(ctx.h, z.h, a.h, b.h, c.h, cf.ctypes.data, oa.ctypes.data, ob.ctypes.data))
    return list(zip(mont_limbs_to_int(oa), mont_limbs_to_int(ob)))


def sparse_matvec3(row_ptr, col, val_a, val_b, val_c, z):
    """co-spartan zero_round: (za, zb, zc) = (A, B, C) z on shares; CSR inputs are device Vecs"""
    ctx = z.ctx
    ha, hb, hc = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    ctx.check(ctx._l.cozk_sparse_matvec3(ctx.h, row_ptr.h, col.h, val_a.h, val_b.h, val_c.h, z.h, ctypes.byref(ha),
                                         ctypes.byref(hb), ctypes.byref(hc)))
    return Rep3DensePolynomial(ctx, ha), Rep3DensePolynomial(ctx, hb), Rep3DensePolynomial(ctx, hc)
