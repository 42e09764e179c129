"""Instruction-lookups harness over the C ABI (`cozk_lookups_*`): SURVEY 8(f)1 restated synthetically -- the toggled /
sparse batched grand product of Lasso's read / write memory checking (co-jolt/src/subprotocols/sparse_grand_product.rs)
on the GPU(s), coordinator + verifier on the calling thread -- and thin wrappers of the toggle-layer entry points
(`cozk_toggle_*`) for the kernel-level parity tests."""
import ctypes

import numpy as np

from . import _lib as L
from .engine import Vec, fr_to_mont_limbs, mont_limbs_to_int


class LookupsConfig(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("log_n", ctypes.c_int), ("n_pairs", ctypes.c_int), ("density_pct", ctypes.c_int),
                ("devices", ctypes.c_int * 3), ("seed", ctypes.c_uint64), ("log_workers", ctypes.c_int), ("primary", ctypes.c_int), ("mix", ctypes.c_int)]


class LookupsResult(ctypes.Structure):
    _fields_ = [("verified", ctypes.c_int), ("wall_ms", ctypes.c_double), ("t_primary_ms", ctypes.c_double), ("t_construct_ms", ctypes.c_double),
                ("t_prove_ms", ctypes.c_double),
                ("t_worker_ms", ctypes.c_double), ("bytes_star_up", ctypes.c_uint64), ("bytes_star_down", ctypes.c_uint64),
                ("bytes_ring", ctypes.c_uint64), ("star_messages", ctypes.c_uint64), ("proof_len", ctypes.c_uint64),
                ("proof_digest", ctypes.c_uint8 * 32)]


LOOKUPS_SYMBOLS = ["cozk_lookups_create", "cozk_lookups_error", "cozk_lookups_destroy", "cozk_lookups_prove", "cozk_lookups_proof_bytes",
                   "cozk_toggle_create", "cozk_toggle_free", "cozk_toggle_batch", "cozk_toggle_len", "cozk_toggle_layer_output", "cozk_toggle_bind",
                   "cozk_toggle_round", "cozk_toggle_final_claims", "cozk_toggle_download"]

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


def _decl():
    l = L.lib()
    l.cozk_lookups_create.restype = _i
    l.cozk_lookups_create.argtypes = [ctypes.POINTER(LookupsConfig), ctypes.POINTER(_vp)]
    l.cozk_lookups_error.restype = ctypes.c_char_p
    l.cozk_lookups_error.argtypes = [_vp]
    l.cozk_lookups_destroy.restype = _i
    l.cozk_lookups_destroy.argtypes = [_vp]
    l.cozk_lookups_prove.restype = _i
    l.cozk_lookups_prove.argtypes = [_vp, _i, ctypes.POINTER(LookupsResult)]
    l.cozk_lookups_proof_bytes.restype = _i
    l.cozk_lookups_proof_bytes.argtypes = [_vp, _vp, _sz]
    l.cozk_toggle_create.restype = _i
    l.cozk_toggle_create.argtypes = [_vp, _i, _vp, _sz, _vp, _vp, _i, ctypes.POINTER(_vp)]
    l.cozk_toggle_free.restype = _i
    l.cozk_toggle_free.argtypes = [_vp]
    l.cozk_toggle_batch.restype = _sz
    l.cozk_toggle_batch.argtypes = [_vp]
    l.cozk_toggle_len.restype = _sz
    l.cozk_toggle_len.argtypes = [_vp]
    l.cozk_toggle_layer_output.restype = _i
    l.cozk_toggle_layer_output.argtypes = [_vp, _vp, _i, ctypes.POINTER(_vp)]
    l.cozk_toggle_bind.restype = _i
    l.cozk_toggle_bind.argtypes = [_vp, _vp, _vp]
    l.cozk_toggle_round.restype = _i
    l.cozk_toggle_round.argtypes = [_vp, _vp, _vp, _vp, _i, _vp]
    l.cozk_toggle_final_claims.restype = _i
    l.cozk_toggle_final_claims.argtypes = [_vp, _vp, _vp, _vp, _vp]
    l.cozk_toggle_download.restype = _i
    l.cozk_toggle_download.argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]
    return l


class LookupsHarness:
    def __init__(self, mode="plain", log_n=6, n_pairs=2, density_pct=25, devices=(0, 0, 0), seed=1, primary=False, log_workers=0, mix="uniform"):
        self._l = _decl()
        cfg = LookupsConfig()
        cfg.mix = 1 if mix == "sha2" else 0
        cfg.mode = L.MODE_PLAIN if mode == "plain" else L.MODE_REP3
        cfg.log_n, cfg.n_pairs, cfg.density_pct = log_n, n_pairs, density_pct
        cfg.devices = (ctypes.c_int * 3)(*devices)
        cfg.seed = seed
        cfg.primary = 1 if primary else 0
        cfg.log_workers = log_workers
        h = _vp()
        rc = self._l.cozk_lookups_create(ctypes.byref(cfg), ctypes.byref(h))
        self.h = h
        if rc != L.OK:
            msg = (self._l.cozk_lookups_error(h) or b"?").decode() if h else "?"
            if h:
                self._l.cozk_lookups_destroy(h)
                self.h = None
            raise L.CozkError(rc, msg)

    def prove(self, verify=True):
        res = LookupsResult()
        rc = self._l.cozk_lookups_prove(self.h, 1 if verify else 0, ctypes.byref(res))
        if rc != L.OK:
            raise L.CozkError(rc, (self._l.cozk_lookups_error(self.h) or b"?").decode())
        return res

    def proof_bytes(self, res):
        buf = (ctypes.c_uint8 * int(res.proof_len))()
        rc = self._l.cozk_lookups_proof_bytes(self.h, buf, int(res.proof_len))
        if rc != L.OK:
            raise L.CozkError(rc, "proof_bytes")
        return bytes(buf)

    def last_error(self):
        return (self._l.cozk_lookups_error(self.h) or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            self._l.cozk_lookups_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ToggleLayer:
    """Rep3BatchedGrandProductToggleLayer on the device (`cozk_toggle`): `flags` = one list of 0/1 per PAIR of circuits,
    `fingerprints` = one list per circuit of ints (plain) or (a, b) tuples (Rep3)"""

    def __init__(self, ctx, flags, fingerprints):
        from .poly import Rep3DenseInterleavedPolynomial, SplitEqPolynomial  # noqa: F401 - same module family
        self._l = _decl()
        self.ctx = ctx
        self.mode = L.MODE_REP3 if isinstance(fingerprints[0][0], tuple) else L.MODE_PLAIN
        self._flag_vecs = [Vec.from_ints(ctx, f, kind=L.SCALAR_U8) for f in flags]
        flat = [v for row in fingerprints for v in row]
        if self.mode == L.MODE_REP3:
            fa, fb = Vec.from_ints(ctx, [v[0] for v in flat]), Vec.from_ints(ctx, [v[1] for v in flat])
        else:
            fa, fb = Vec.from_ints(ctx, flat), None
        arr = (_vp * len(flags))(*[v.h for v in self._flag_vecs])
        h = _vp()
        ctx.check(self._l.cozk_toggle_create(ctx.h, self.mode, arr, len(flags), fa.h, fb.h if fb is not None else None, 0, ctypes.byref(h)))
        self.h = h

    def layer_output(self, party=0):
        from .poly import Rep3DenseInterleavedPolynomial
        h = _vp()
        self.ctx.check(self._l.cozk_toggle_layer_output(self.ctx.h, self.h, party, ctypes.byref(h)))
        return Rep3DenseInterleavedPolynomial(self.ctx, h, self.mode)

    def bind(self, r):
        rr = fr_to_mont_limbs([r])[0]
        self.ctx.check(self._l.cozk_toggle_bind(self.ctx.h, self.h, rr.ctypes.data))

    def round(self, eq, r=None, party=0):
        """bind with r (None in the first round), then this party's additive g(0), g(2), g(3)"""
        rr = fr_to_mont_limbs([r])[0] if r is not None else None
        out = np.zeros((3, 4), dtype=np.uint64)
        self.ctx.check(self._l.cozk_toggle_round(self.ctx.h, self.h, eq.h, rr.ctypes.data if rr is not None else None, party, out.ctypes.data))
        return mont_limbs_to_int(out)

    def final_claims(self):
        fl, pa, pb = (np.zeros(4, dtype=np.uint64) for _ in range(3))
        self.ctx.check(self._l.cozk_toggle_final_claims(self.ctx.h, self.h, fl.ctypes.data, pa.ctypes.data, pb.ctypes.data))
        f, a, b = (mont_limbs_to_int(x.reshape(1, 4))[0] for x in (fl, pa, pb))
        return f, ((a, b) if self.mode == L.MODE_REP3 else a)

    def download(self):
        nf, npn = _sz(), _sz()
        self.ctx.check(self._l.cozk_toggle_download(self.ctx.h, self.h, None, None, None, ctypes.byref(nf), ctypes.byref(npn)))
        fl = np.zeros((nf.value, 4), dtype=np.uint64)
        pa = np.zeros((npn.value, 4), dtype=np.uint64)
        pb = np.zeros((npn.value, 4), dtype=np.uint64)
        self.ctx.check(self._l.cozk_toggle_download(self.ctx.h, self.h, fl.ctypes.data, pa.ctypes.data, pb.ctypes.data, None, None))
        a = mont_limbs_to_int(pa)
        fps = list(zip(a, mont_limbs_to_int(pb))) if self.mode == L.MODE_REP3 else a
        return mont_limbs_to_int(fl), fps

    def free(self):
        if getattr(self, "h", None):
            self._l.cozk_toggle_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
