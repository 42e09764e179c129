"""Worker drivers (`cozk_worker_*`) with a host-supplied transport: the C++ round loops of
csrc/host/prover.hpp run against `cozk_star_net` callbacks provided from Python (tests play the
coordinator inside the callbacks; bench/hosts can put torch.distributed or sockets behind them)."""
import ctypes

import numpy as np

from . import _lib as L
from .engine import fr_to_mont_limbs, mont_limbs_to_int

_SEND = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
_RECV = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t))
_RESHARE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


class StarNet(ctypes.Structure):
    _fields_ = [("user", ctypes.c_void_p), ("send_response", _SEND), ("receive_request", _RECV)]


class RingNet(ctypes.Structure):
    _fields_ = [("user", ctypes.c_void_p), ("reshare", _RESHARE), ("stream_ordered", ctypes.c_int)]


class WorkerParams(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("party", ctypes.c_int), ("key_self", ctypes.c_uint8 * 32), ("key_prev", ctypes.c_uint8 * 32),
                ("mask_counter", ctypes.c_uint64)]


WORKER_SYMBOLS = ["cozk_worker_prove_grand_product", "cozk_worker_prove_arbitrary", "cozk_worker_spartan_first_sumcheck",
                  "cozk_worker_spartan_second_sumcheck"]


class CallbackStar:
    """wraps two Python callables: on_send(bytes) and on_recv() -> bytes"""

    def __init__(self, on_send, on_recv):
        self.error = None

        def _send(_u, p, n):
            try:
                on_send(ctypes.string_at(p, n))
                return 0
            except Exception as e:  # never unwind into C
                self.error = e
                return 1

        def _recv(_u, buf, cap, out_len):
            try:
                b = on_recv()
                if len(b) > cap:
                    return 2
                ctypes.memmove(buf, b, len(b))
                out_len[0] = len(b)
                return 0
            except Exception as e:
                self.error = e
                return 1

        self._s, self._r = _SEND(_send), _RECV(_recv)
        self.net = StarNet(None, self._s, self._r)


def _params(mode, party=0, key_self=None, key_prev=None, counter=0):
    ks = (ctypes.c_uint8 * 32)(*(L.prf_key(key_self) or bytes(32)))
    kp = (ctypes.c_uint8 * 32)(*(L.prf_key(key_prev) or bytes(32)))
    return WorkerParams(L.MODE_PLAIN if mode == "plain" else L.MODE_REP3, party, ks, kp, counter)


def _check(ctx, rc, star):
    if rc != L.OK:
        if star.error is not None:
            raise star.error
        ctx.check(rc)


def prove_grand_product(ctx, layer, batch_size, star, mode="plain", ring=None, **kw):
    l = L.lib()
    p = _params(mode, **kw)
    cap = 64
    out = np.zeros((cap, 4), dtype=np.uint64)
    n = ctypes.c_size_t()
    rc = l.cozk_worker_prove_grand_product(ctx.h, ctypes.byref(p), ctypes.byref(star.net), ctypes.byref(ring) if ring else None, layer.h,
                                           ctypes.c_size_t(batch_size), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap), ctypes.byref(n))
    _check(ctx, rc, star)
    return mont_limbs_to_int(out[:n.value])


def prove_arbitrary(ctx, polys, degree, claim, num_rounds, star, mode="plain", **kw):
    l = L.lib()
    p = _params(mode, **kw)
    arr = (ctypes.c_void_p * len(polys))(*[q.h for q in polys])
    cl = fr_to_mont_limbs([claim])[0]
    out_r = np.zeros((max(1, num_rounds), 4), dtype=np.uint64)
    out_f = np.zeros((len(polys), 4), dtype=np.uint64)
    rc = l.cozk_worker_prove_arbitrary(ctx.h, ctypes.byref(p), ctypes.byref(star.net), arr, ctypes.c_size_t(len(polys)), degree,
                                       cl.ctypes.data_as(ctypes.c_void_p), num_rounds, out_r.ctypes.data_as(ctypes.c_void_p),
                                       out_f.ctypes.data_as(ctypes.c_void_p))
    _check(ctx, rc, star)
    return mont_limbs_to_int(out_r[:num_rounds]), mont_limbs_to_int(out_f)


def spartan_first_sumcheck(ctx, za, zb, zc, eq, star, mode="plain", **kw):
    l = L.lib()
    p = _params(mode, **kw)
    nv = len(eq).bit_length() - 1
    pt = np.zeros((max(1, nv), 4), dtype=np.uint64)
    fin = np.zeros((4, 4), dtype=np.uint64)
    rc = l.cozk_worker_spartan_first_sumcheck(ctx.h, ctypes.byref(p), ctypes.byref(star.net), za.h, zb.h, zc.h, eq.h,
                                              pt.ctypes.data_as(ctypes.c_void_p), fin.ctypes.data_as(ctypes.c_void_p))
    _check(ctx, rc, star)
    return mont_limbs_to_int(pt[:nv]), mont_limbs_to_int(fin)


def spartan_second_sumcheck(ctx, z, a, b, c, coef, star, mode="plain", **kw):
    l = L.lib()
    p = _params(mode, **kw)
    nv = len(z).bit_length() - 1
    cf = fr_to_mont_limbs(coef)
    pt = np.zeros((max(1, nv), 4), dtype=np.uint64)
    fin = np.zeros((4, 4), dtype=np.uint64)
    rc = l.cozk_worker_spartan_second_sumcheck(ctx.h, ctypes.byref(p), ctypes.byref(star.net), z.h, a.h, b.h, c.h,
                                               cf.ctypes.data_as(ctypes.c_void_p), pt.ctypes.data_as(ctypes.c_void_p),
                                               fin.ctypes.data_as(ctypes.c_void_p))
    _check(ctx, rc, star)
    return mont_limbs_to_int(pt[:nv]), mont_limbs_to_int(fin)
