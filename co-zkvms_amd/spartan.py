"""co-noir-spartan harness over the C ABI (`cozk_spartan_*`): BASELINE config 4 restated -- the
worker side of SpartanProverWorker::prove (co-noir-spartan/co-spartan/src/worker.rs:119-300) on a satisfied
synthetic R1CS, coordinator + verifier on the calling thread."""
import ctypes

from . import _lib as L


class SpartanConfig(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("log_n", ctypes.c_int), ("precompute", ctypes.c_int), ("devices", ctypes.c_int * 3),
                ("seed", ctypes.c_uint64), ("lookup_round", ctypes.c_int), ("log_pub_workers", ctypes.c_int)]


class SpartanResult(ctypes.Structure):
    _fields_ = [("verified", ctypes.c_int), ("wall_ms", ctypes.c_double), ("t_zero_round_ms", ctypes.c_double),
                ("t_commit_ms", ctypes.c_double), ("t_sumcheck1_ms", ctypes.c_double), ("t_matrix_build_ms", ctypes.c_double),
                ("t_sumcheck2_ms", ctypes.c_double), ("t_open_ms", ctypes.c_double), ("t_worker_ms", ctypes.c_double),
                ("bytes_star_up", ctypes.c_uint64), ("bytes_star_down", ctypes.c_uint64), ("star_messages", ctypes.c_uint64),
                ("proof_len", ctypes.c_uint64), ("proof_digest", ctypes.c_uint8 * 32), ("t_lookup_ms", ctypes.c_double),
                ("pub_workers", ctypes.c_int), ("pub_star_messages", ctypes.c_uint64), ("pub_bytes_up", ctypes.c_uint64),
                ("pub_bytes_down", ctypes.c_uint64)]


SPARTAN_SYMBOLS = ["cozk_spartan_create", "cozk_spartan_error", "cozk_spartan_destroy", "cozk_spartan_prove", "cozk_spartan_proof_bytes"]


def _decl():
    l = L.lib()
    l.cozk_spartan_create.restype = ctypes.c_int
    l.cozk_spartan_create.argtypes = [ctypes.POINTER(SpartanConfig), ctypes.POINTER(ctypes.c_void_p)]
    l.cozk_spartan_error.restype = ctypes.c_char_p
    l.cozk_spartan_error.argtypes = [ctypes.c_void_p]
    l.cozk_spartan_destroy.restype = ctypes.c_int
    l.cozk_spartan_destroy.argtypes = [ctypes.c_void_p]
    l.cozk_spartan_prove.restype = ctypes.c_int
    l.cozk_spartan_prove.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(SpartanResult)]
    l.cozk_spartan_proof_bytes.restype = ctypes.c_int
    l.cozk_spartan_proof_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    return l


class SpartanHarness:
    def __init__(self, mode="plain", log_n=10, precompute=True, devices=(0, 0, 0), seed=1, lookup_round=False, log_pub_workers=0):
        self._l = _decl()
        cfg = SpartanConfig()
        cfg.mode = L.MODE_PLAIN if mode == "plain" else L.MODE_REP3
        cfg.log_n = log_n
        cfg.precompute = 1 if precompute else 0
        cfg.devices = (ctypes.c_int * 3)(*devices)
        cfg.seed = seed
        cfg.lookup_round = 1 if lookup_round else 0
        cfg.log_pub_workers = log_pub_workers
        h = ctypes.c_void_p()
        rc = self._l.cozk_spartan_create(ctypes.byref(cfg), ctypes.byref(h))
        self.h = h
        if rc != L.OK:
            msg = (self._l.cozk_spartan_error(h) or b"?").decode() if h else "?"
            if h:
                self._l.cozk_spartan_destroy(h)
                self.h = None
            raise L.CozkError(rc, msg)

    def prove(self, verify=True):
        res = SpartanResult()
        rc = self._l.cozk_spartan_prove(self.h, 1 if verify else 0, ctypes.byref(res))
        if rc != L.OK:
            raise L.CozkError(rc, (self._l.cozk_spartan_error(self.h) or b"?").decode())
        return res

    def proof_bytes(self, res):
        buf = (ctypes.c_uint8 * int(res.proof_len))()
        rc = self._l.cozk_spartan_proof_bytes(self.h, buf, int(res.proof_len))
        if rc != L.OK:
            raise L.CozkError(rc, "proof_bytes")
        return bytes(buf)

    def last_error(self):
        return (self._l.cozk_spartan_error(self.h) or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            self._l.cozk_spartan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
