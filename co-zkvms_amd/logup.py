"""co-noir-spartan's public lookup round on the device (`cozk_hash_tuple`, `cozk_logup_h`, `cozk_vec_boost_degree`,
`cozk_prodlist_*`): thin ctypes wrappers for the parity tests and for hosts that drive the round themselves."""
import ctypes

import numpy as np

from . import _lib as L
from .engine import Vec, fr_to_mont_limbs, mont_limbs_to_int

LOGUP_SYMBOLS = ["cozk_vec_gather", "cozk_hash_tuple", "cozk_logup_h", "cozk_vec_boost_degree", "cozk_prodlist_create", "cozk_prodlist_free", "cozk_prodlist_degree",
                 "cozk_prodlist_round", "cozk_prodlist_final"]
_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


def _decl():
    l = L.lib()
    l.cozk_vec_gather.restype = _i
    l.cozk_vec_gather.argtypes = [_vp, _vp, _vp, _sz, ctypes.POINTER(_vp)]
    l.cozk_hash_tuple.restype = _i
    l.cozk_hash_tuple.argtypes = [_vp, _vp, _vp, _vp, _sz, ctypes.POINTER(_vp)]
    l.cozk_logup_h.restype = _i
    l.cozk_logup_h.argtypes = [_vp, _vp, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]
    l.cozk_vec_boost_degree.restype = _i
    l.cozk_vec_boost_degree.argtypes = [_vp, _vp, _i, ctypes.POINTER(_vp)]
    l.cozk_prodlist_create.restype = _i
    l.cozk_prodlist_create.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp, _sz, ctypes.POINTER(_vp)]
    l.cozk_prodlist_free.restype = _i
    l.cozk_prodlist_free.argtypes = [_vp]
    l.cozk_prodlist_degree.restype = _i
    l.cozk_prodlist_degree.argtypes = [_vp]
    l.cozk_prodlist_round.restype = _i
    l.cozk_prodlist_round.argtypes = [_vp, _vp, _vp, _vp]
    l.cozk_prodlist_final.restype = _i
    l.cozk_prodlist_final.argtypes = [_vp, _vp, _vp, _vp]
    return l


def gather(ctx, idx, src, n_out):
    """eq_tilde_{rx,ry}: out[j] = src[idx[j]] (0xffffffff: no entry -> 0), zero padding up to n_out"""
    l = _decl()
    iv = Vec.from_ints(ctx, idx, kind=L.SCALAR_U32)
    h = _vp()
    ctx.check(l.cozk_vec_gather(ctx.h, iv.h, src.h, n_out, ctypes.byref(h)))
    return Vec(ctx, h, L.SCALAR_FR)


def hash_tuple(ctx, idx, eq, v_msg, n_out):
    l = _decl()
    iv = Vec.from_ints(ctx, idx, kind=L.SCALAR_U32)
    h = _vp()
    ctx.check(l.cozk_hash_tuple(ctx.h, iv.h, eq.h, fr_to_mont_limbs([v_msg])[0].ctypes.data, n_out, ctypes.byref(h)))
    return Vec(ctx, h, L.SCALAR_FR)


def logup_h(ctx, values, m, x):
    l = _decl()
    phi, h = _vp(), _vp()
    ctx.check(l.cozk_logup_h(ctx.h, values.h, m.h if m is not None else None, fr_to_mont_limbs([x])[0].ctypes.data, ctypes.byref(phi), ctypes.byref(h)))
    return Vec(ctx, phi, L.SCALAR_FR), Vec(ctx, h, L.SCALAR_FR)


def boost_degree(ctx, v, new_num_vars):
    l = _decl()
    h = _vp()
    ctx.check(l.cozk_vec_boost_degree(ctx.h, v.h, new_num_vars, ctypes.byref(h)))
    return Vec(ctx, h, L.SCALAR_FR)


class ProdList:
    """ListOfProductsOfPolynomials + IPForMLSumcheck prover state: polys = Vecs, products = [(coef, [poly index, ...])]"""

    def __init__(self, ctx, polys, products):
        self._l = _decl()
        self.ctx = ctx
        self.n_polys = len(polys)
        arr = (_vp * len(polys))(*[p.h for p in polys])
        coefs = fr_to_mont_limbs([c for c, _ in products])
        counts = (ctypes.c_int * len(products))(*[len(f) for _, f in products])
        flat = [j for _, f in products for j in f]
        fidx = (ctypes.c_int * len(flat))(*flat)
        h = _vp()
        ctx.check(self._l.cozk_prodlist_create(ctx.h, arr, len(polys), coefs.ctypes.data, counts, fidx, len(products), ctypes.byref(h)))
        self.h = h
        self.degree = self._l.cozk_prodlist_degree(h)

    def round(self, r=None):
        rr = fr_to_mont_limbs([r])[0] if r is not None else None
        out = np.zeros((self.degree + 1, 4), dtype=np.uint64)
        self.ctx.check(self._l.cozk_prodlist_round(self.ctx.h, self.h, rr.ctypes.data if rr is not None else None, out.ctypes.data))
        return mont_limbs_to_int(out)

    def final(self, r):
        out = np.zeros((self.n_polys, 4), dtype=np.uint64)
        self.ctx.check(self._l.cozk_prodlist_final(self.ctx.h, self.h, fr_to_mont_limbs([r])[0].ctypes.data, out.ctypes.data))
        return mont_limbs_to_int(out)

    def free(self):
        if getattr(self, "h", None):
            self._l.cozk_prodlist_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
