"""co-jolt Spartan outer-sumcheck harness over the C ABI (`cozk_outer_harness_*`) and a thin wrapper of the
kernel-level entry points (`cozk_outer_*`) for the parity tests."""
import ctypes

import numpy as np

from . import _lib as L
from .engine import fr_to_mont_limbs, mont_limbs_to_int


class OuterConfig(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("log_steps", ctypes.c_int), ("devices", ctypes.c_int * 3), ("seed", ctypes.c_uint64),
                ("system", ctypes.c_int), ("full", ctypes.c_int)]


class OuterResult(ctypes.Structure):
    _fields_ = [("verified", ctypes.c_int), ("wall_ms", ctypes.c_double), ("t_build_ms", ctypes.c_double), ("t_prove_ms", ctypes.c_double),
                ("t_worker_ms", ctypes.c_double), ("t_outer_ms", ctypes.c_double), ("t_inner_ms", ctypes.c_double), ("t_shift_ms", ctypes.c_double),
                ("t_openings_ms", ctypes.c_double), ("bytes_star_up", ctypes.c_uint64), ("bytes_star_down", ctypes.c_uint64),
                ("star_messages", ctypes.c_uint64), ("proof_len", ctypes.c_uint64), ("proof_digest", ctypes.c_uint8 * 32)]


class LC(ctypes.Structure):
    _fields_ = [("first_term", ctypes.c_int), ("n_terms", ctypes.c_int), ("offset", ctypes.c_int)]


class R1CS(ctypes.Structure):
    _fields_ = [("term_var", ctypes.POINTER(ctypes.c_int)), ("term_coeff", ctypes.POINTER(ctypes.c_int64)), ("n_terms", ctypes.c_size_t),
                ("uniform", ctypes.POINTER(LC)), ("n_uniform", ctypes.c_size_t), ("cross", ctypes.POINTER(LC)), ("n_cross", ctypes.c_size_t),
                ("padded_num_constraints", ctypes.c_size_t)]


OUTER_SYMBOLS = ["cozk_outer_harness_create", "cozk_outer_harness_error", "cozk_outer_harness_destroy", "cozk_outer_harness_prove",
                 "cozk_outer_harness_proof_bytes", "cozk_outer_create", "cozk_outer_free", "cozk_outer_len", "cozk_outer_download", "cozk_outer_round",
                 "cozk_outer_final_evals", "cozk_eq_plus_one_evals", "cozk_poly_batch_dot_public"]
_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


def _decl():
    l = L.lib()
    l.cozk_outer_harness_create.restype = _i
    l.cozk_outer_harness_create.argtypes = [ctypes.POINTER(OuterConfig), ctypes.POINTER(_vp)]
    l.cozk_outer_harness_error.restype = ctypes.c_char_p
    l.cozk_outer_harness_error.argtypes = [_vp]
    l.cozk_outer_harness_destroy.restype = _i
    l.cozk_outer_harness_destroy.argtypes = [_vp]
    l.cozk_outer_harness_prove.restype = _i
    l.cozk_outer_harness_prove.argtypes = [_vp, _i, ctypes.POINTER(OuterResult)]
    l.cozk_outer_harness_proof_bytes.restype = _i
    l.cozk_outer_harness_proof_bytes.argtypes = [_vp, _vp, _sz]
    l.cozk_outer_create.restype = _i
    l.cozk_outer_create.argtypes = [_vp, _i, _i, ctypes.POINTER(R1CS), _vp, _sz, _vp, _sz, ctypes.POINTER(_vp)]
    l.cozk_outer_free.restype = _i
    l.cozk_outer_free.argtypes = [_vp]
    l.cozk_outer_len.restype = _sz
    l.cozk_outer_len.argtypes = [_vp]
    l.cozk_outer_download.restype = _i
    l.cozk_outer_download.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    l.cozk_outer_round.restype = _i
    l.cozk_outer_round.argtypes = [_vp, _vp, _vp, _vp, _vp]
    l.cozk_outer_final_evals.restype = _i
    l.cozk_outer_final_evals.argtypes = [_vp, _vp, _vp, _vp]
    return l


class OuterHarness:
    def __init__(self, mode="plain", log_steps=4, devices=(0, 0, 0), seed=1, system="toy", full=False):
        """system = "jolt": the reference's constraint set (70 + 2 constraints, 78 inputs, 128 rows per step); full: the whole
        Spartan worker (outer + inner + shift sumchecks + the two opening appends)"""
        self._l = _decl()
        cfg = OuterConfig()
        cfg.system = 1 if system == "jolt" else 0
        cfg.full = 1 if full else 0
        cfg.mode = L.MODE_PLAIN if mode == "plain" else L.MODE_REP3
        cfg.log_steps = log_steps
        cfg.devices = (ctypes.c_int * 3)(*devices)
        cfg.seed = seed
        h = _vp()
        rc = self._l.cozk_outer_harness_create(ctypes.byref(cfg), ctypes.byref(h))
        self.h = h
        if rc != L.OK:
            msg = (self._l.cozk_outer_harness_error(h) or b"?").decode() if h else "?"
            if h:
                self._l.cozk_outer_harness_destroy(h)
                self.h = None
            raise L.CozkError(rc, msg)

    def prove(self, verify=True):
        res = OuterResult()
        rc = self._l.cozk_outer_harness_prove(self.h, 1 if verify else 0, ctypes.byref(res))
        if rc != L.OK:
            raise L.CozkError(rc, (self._l.cozk_outer_harness_error(self.h) or b"?").decode())
        return res

    def proof_bytes(self, res):
        buf = (ctypes.c_uint8 * int(res.proof_len))()
        rc = self._l.cozk_outer_harness_proof_bytes(self.h, buf, int(res.proof_len))
        if rc != L.OK:
            raise L.CozkError(rc, "proof_bytes")
        return bytes(buf)

    def last_error(self):
        return (self._l.cozk_outer_harness_error(self.h) or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            self._l.cozk_outer_harness_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SpartanOuter:
    """`cozk_outer`: uniform / cross = lists of (a, b, c) with LCs as [(var or None, coeff)] (cross LCs are (offset, lc));
    polys = the witness columns (Rep3DensePolynomial objects, PLAIN = public)"""

    def __init__(self, ctx, mode, party, uniform, cross, polys, padded, tau):
        self._l = _decl()
        self.ctx = ctx
        self.mode = L.MODE_REP3 if mode == "rep3" else L.MODE_PLAIN
        tv, tc, lcs_u, lcs_c = [], [], [], []

        def add(lc, offset=0):
            first = len(tv)
            for var, coeff in lc:
                tv.append(-1 if var is None else var)
                tc.append(coeff)
            return LC(first, len(lc), offset)

        for a, b, c in uniform:
            lcs_u += [add(a), add(b), add(c)]
        for a, b, cond in cross:
            lcs_c += [add(a[1], 1 if a[0] else 0), add(b[1], 1 if b[0] else 0), add(cond[1], 1 if cond[0] else 0)]
        self._tv = (ctypes.c_int * max(1, len(tv)))(*tv)
        self._tc = (ctypes.c_int64 * max(1, len(tc)))(*tc)
        self._u = (LC * max(1, len(lcs_u)))(*lcs_u)
        self._c = (LC * max(1, len(lcs_c)))(*lcs_c)
        sys = R1CS(self._tv, self._tc, len(tv), self._u, len(uniform), self._c, len(cross), padded)
        arr = (_vp * len(polys))(*[p.h for p in polys])
        t = fr_to_mont_limbs(tau)
        h = _vp()
        ctx.check(self._l.cozk_outer_create(ctx.h, self.mode, party, ctypes.byref(sys), arr, len(polys), t.ctypes.data, len(tau), ctypes.byref(h)))
        self.h = h

    def __len__(self):
        return self._l.cozk_outer_len(self.h)

    def download(self):
        n = len(self)
        bufs = [np.zeros((n, 4), dtype=np.uint64) for _ in range(6)]
        self.ctx.check(self._l.cozk_outer_download(self.ctx.h, self.h, *[b.ctypes.data for b in bufs]))
        v = [mont_limbs_to_int(b) for b in bufs]
        if self.mode == L.MODE_REP3:
            return [list(zip(v[0], v[1])), list(zip(v[2], v[3])), list(zip(v[4], v[5]))]
        return [v[0], v[2], v[4]]

    def round(self, r, claim):
        rr = fr_to_mont_limbs([r])[0] if r is not None else None
        cl = fr_to_mont_limbs([claim])[0]
        out = np.zeros((4, 4), dtype=np.uint64)
        self.ctx.check(self._l.cozk_outer_round(self.ctx.h, self.h, rr.ctypes.data if rr is not None else None, cl.ctypes.data, out.ctypes.data))
        return mont_limbs_to_int(out)

    def final_evals(self, r):
        rr = fr_to_mont_limbs([r])[0]
        out = np.zeros((3, 4), dtype=np.uint64)
        self.ctx.check(self._l.cozk_outer_final_evals(self.ctx.h, self.h, rr.ctypes.data, out.ctypes.data))
        return mont_limbs_to_int(out)

    def free(self):
        if getattr(self, "h", None):
            self._l.cozk_outer_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
